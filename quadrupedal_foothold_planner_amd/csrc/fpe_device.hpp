// fpe_device.hpp — plain-data types shared by the host side of the engine and the gfx950 kernels.
#pragma once
#include <cstdint>

#include "../../include/fpe.h"
#include "fpe_gridmath.hpp"

namespace fpe {

// Canonical device map: both layers row-major, (i, j) at i * cols + j, start index (0, 0).
struct DevMap {
    MapGeom g;
    const float* trav;  // "traversability" layer (cpp:2057, 2138, 1650)
    const float* elev;  // "elevation" layer (cpp:2532-2533)
};

// SpiralIterator visiting order as a rank table (fpe_spiral.cpp): entry k = k-th visited offset
// relative to the centre index; ringStart[d] = first entry of ring d (rings 0..maxRing).
struct SpiralLut {
    const int16_t* di;
    const int16_t* dj;
    const uint8_t* ring;       // ring index of entry k
    const int32_t* ringStart;  // [maxRing + 2]
    int32_t maxRing;
};

constexpr int kMaxFootOffsets = 128;

// Per-call constants derived on the host from fpe_params with the reference's typing
// (initialize(), cpp:340-421; fpe_host.cpp::derive_constants).
struct PlanConsts {
    float footRadius, thrDefault, thrCandidate, searchRadius;
    double rf, rf2;  // double(footRadius), pow(rf, 2)
    double LbHalf, WbHalfNeg, WbHalfPos;
    double biasX[4], biasY[4];
    double stepHalf, step, stepQuarter;
    double h, drift;
    int32_t RF_FIRST;
    int32_t tileH, tileW;  // LDS tile half-width / width in cells
    uint32_t tileWMagic;   // fastdiv magic of tileW
    float maxSearchRadius; // radius the tile was sized for
    int32_t groupOverride; // 0 = automatic lanes-per-leg; 16 / 64 force it (tuning knob FPE_PLAN_GROUP)
    // Cell offsets of a CELL-CENTRED foot disc (checkCirclePolygonFoothold's CircleIterator around
    // a spiral candidate), valid only when footRobust != 0: the host proved that no lattice offset
    // lies within rounding distance of the radius, so the f64 per-candidate bounding-box walk
    // visits exactly {candidate + offset} ∩ map (fpe_host.cpp::derive_foot_offsets).
    unsigned long long* trace;  // profiling-only: per-phase s_memtime stamps (FPE_TRACE_PTR); null in production
    int32_t nFoot;
    int32_t footRobust;
    int32_t footReach;     // cells a foot disc can reach from its centre cell (max |offset|, or ceil(rf/res)+1)
    // foot radius >= 0.9 * resolution: the middle cell of an UNCLAMPED 3x3 CircleIterator bounding box is inside the
    // disc whatever the (continuous) centre — proof in fpe_host.cpp::derive_constants — so the 8-lane kernel walks
    // the other eight cells in one round and takes the middle one for granted (disc_issue / disc_consume)
    int32_t midCellInside;
    int8_t footDa[kMaxFootOffsets];
    int8_t footDb[kMaxFootOffsets];
};

// Tile flag bits (one byte per cell in LDS).
enum : uint8_t {
    kFlagInMap = 1,      // cell index inside the map
    kFlagFinite = 2,     // isfinite(traversability)            GridMap::isValid
    kFlagBelowDef = 4,   // traversability < defaultFootholdThreshold   (raw compare, NaN -> false)
    kFlagBelowCand = 8,  // traversability < candidateFootholdThreshold (raw compare)
    kFlagFail = 16       // finite && (below candidate threshold || outside the search polygon)
};

}  // namespace fpe
