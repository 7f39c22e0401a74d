// fpe_device.hpp — plain-data types shared by the host side of the engine and the gfx950 kernels.
#pragma once
#include <cstdint>

#include <hip/hip_runtime.h>

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
    int32_t total;             // ringStart[maxRing + 1]: entries of the whole table (spares the kernels a dependent load)
    // entry k as one word, (di & 0xFF) | (dj & 0xFF) << 8 | ring << 16, padded with (0, 0, 255) to a multiple of 32 entries
    // plus one more group of 32: the 8-lane bit-window kernels read four consecutive entries per lane and round (uint4)
    const uint32_t* packed;
    // The first sixteen ranks (ring 0, ring 1, seven cells of ring 2: all within two rows / columns of the centre) by ROW, for
    // the 3x3-only kernels' LDS-free first rounds (fpe_bits.hpp::leg_fast8m): words 0-4 = row offsets di = -2..2, five 5-bit
    // fields each (column offset dj = -2..2 -> the rank of (di, dj), 31 = not among the first sixteen); words 6-7 / 8-9 = di + 2
    // / dj + 2 of rank q in the 4-bit field q of a 64-bit word.  Built by fpe_create from the same table.
    const uint32_t* fast16;
};
constexpr int kFast16Words = 12;

constexpr int kMaxFootOffsets = 128;

// Per-call constants derived on the host from fpe_params with the reference's typing
// (initialize(), cpp:340-421; fpe_host.cpp::derive_constants).
// Engine-level tuning / test knobs (fpe_set_tuning); seeded once from the environment in fpe_create.
struct Tuning {
    int32_t planGroup = 0;
    int32_t literalDiscs = 0;
    int32_t noMidVariant = 0;
    int32_t noBits = 0;
    int32_t serviceOverlap = 1;  // small plan + opt calls: the opt chain beside the plan kernel on the flags it will most likely produce (fpe_engine.cpp, plan_host); 0: one after the other
    int32_t servicePoll = 1;     // one-pose overlapped calls: the host polls the chain's completion word in the pinned arena instead of waiting for the stream's signal
    int32_t serviceOptGate = 2;  // fpe_plan_service*: 2 enforce (default: the handler's behaviour), 1 advisory (chain runs, reported), 0 exact gates only (no opt chain) (include/fpe.h)
};

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
    int32_t groupOverride; // 0 = automatic lanes-per-leg; 4 / 8 / 16 / 64 / 65 force it (fpe_set_tuning "plan_group")
    int32_t noMidVariant;  // fpe_set_tuning "no_mid_variant": never launch the 3x3-only kernel variants
    int32_t noBits;        // fpe_set_tuning "no_bits": never launch the bit-window kernels
    // Half-width (cells) of the bit window around getIndex(search centre), or 0 when the host could not prove the
    // bit-window kernels exact for these parameters on this map (fpe_host.cpp::bits_window_halfwidth)
    int32_t winH;
    // A box corner strictly inside the map whose predicted cell coordinate ((x - org) - pos) * (1 / res) is farther
    // than cornerEps from an integer has the index -trunc(prediction): the reference's boundPositionToRange rewrite
    // and index division change the quotient by far less (fpe_host.cpp::bits_window_halfwidth, `slack`).
    double cornerEps;
    // getGaitCycleSearchGridMap's submap (cpp:2339-2345): isos_.length x isos_.width (cpp:384-394)
    double isosLen, isosWid;
    // Cell offsets of a CELL-CENTRED foot disc (checkCirclePolygonFoothold's CircleIterator around
    // a spiral candidate), valid only when footRobust != 0: the host proved that no lattice offset
    // lies within rounding distance of the radius, so the f64 per-candidate bounding-box walk
    // visits exactly {candidate + offset} ∩ map (fpe_host.cpp::derive_foot_offsets).
    unsigned long long* trace;  // profiling-only (-DFPE_TRACE builds): per-phase s_memtime stamps; null in production
    int32_t nFoot;
    int32_t footRobust;
    int32_t footReach;     // cells a foot disc can reach from its centre cell (max |offset|, or ceil(rf/res)+1)
    // foot radius >= 0.9 * resolution: the middle cell of an UNCLAMPED 3x3 CircleIterator bounding box is inside the
    // disc whatever the (continuous) centre — proof in fpe_host.cpp::derive_constants — so the 8-lane kernel walks
    // the other eight cells in one round and takes the middle one for granted (disc_issue / disc_consume)
    int32_t midCellInside;
    int8_t footDa[kMaxFootOffsets];
    int8_t footDb[kMaxFootOffsets];
    // The same disc as row intervals (bit-window erosion): the offsets with row offset +-a are the columns
    // [-hwList[hwIdx[a]], +hwList[hwIdx[a]]], a = 0..footReach; nHW distinct half-widths (0: the table is not of that
    // form, or has more than kMaxHW distinct widths — the kernels then walk the offsets one by one).
    int32_t nHW;
    int8_t hwList[4];
    int8_t hwIdx[16];
    // SpiralIterator constants of the DEFAULT search radius (fpe_params.searchRadius; a pose may override it per leg):
    // nRings = ceil(R / res), spiral rank-table entries of rings 0..nRings — spares every wavefront a division and a
    // dependent table load in its prologue
    int32_t defNRings, defNCand;
};
constexpr int kMaxHW = 4;

// Bit planes of one map snapshot for one (defaultFootholdThreshold, candidateFootholdThreshold) pair
// (fpe_bits.hpp).  One uint4 per 32 columns of a row: x = D  (trav < thrDefault, raw compare: NaN 0, -inf 1),
// y = Df (finite && trav < thrDefault), z = C (finite && trav < thrCandidate), w = F (finite); bit b of a word =
// column 32 * word + b.  TILED: the groups of 8 consecutive rows x one word are one 128-byte line (a search window
// is a few dozen rows of one or two words: row-major planes spend a line per window row, 32-64 bytes of it used);
// tiles in row-major tile order.  Rows -1 and `rows`, kBitPadW word groups left of column 0 and everything right of
// the last column are zero ("not in the map"), so a window hanging over the map edge reads zeros without a bounds test.
#ifndef FPE_BITS_TILED
#define FPE_BITS_TILED 1  // 0: row-major planes (round-2 layout, kept for A/B measurements)
#endif
constexpr int kBitPadW = 4;
struct BitMap {
    const uint4* words;  // group of (row i, word w) at bit_group_index(i, w, strideW)
    int32_t strideW;     // nw + 2 * kBitPadW
    int32_t nw;          // ceil(cols / 32)
};
// row groups (tiles of 8 rows) covering the rows -1 .. rows
__host__ __device__ constexpr int bit_row_groups(int rows) { return ((rows + 1) >> 3) + 1; }
__host__ __device__ __forceinline__ size_t bit_group_index(int i, int w, int strideW) {
#if FPE_BITS_TILED
    const int r1 = i + 1;
    return ((static_cast<size_t>(r1 >> 3) * strideW + static_cast<size_t>(w + kBitPadW)) << 3) + static_cast<size_t>(r1 & 7);
#else
    return static_cast<size_t>(i + 1) * strideW + static_cast<size_t>(w + kBitPadW);
#endif
}

// Producer filters (fpe_filters.hpp): parameters of the published default chain and the eight
// canonical row-major output layers in HBM.
struct FilterConsts {
    double normalRadius, slopeCritical, stepCritical, stepFirstRadius, stepSecondRadius;
    int32_t stepCriticalCells;
    double roughnessCritical, roughnessRadius;
};
struct FilterLayers {
    float *nx, *ny, *nz, *slope, *stepHeight, *step, *rough, *trav;
};

// Opt track (fpe_opt.hpp): the file-scope NLopt globals of the reference (cpp:28-51) and what
// initialize() / gridmapCallback derive for them, with the reference's typing (fpe_host.cpp::derive_opt_constants).
struct OptConsts {
    double w1, w2, w3, w4, wr, wc;
    double ctol;
    double lengthBase, skew, mapResolution;  // cpp:497-498, 514
    double t1, t2, t3, t4;                   // cpp:1156-1159
    double lbOverRes, skew2OverRes;          // lengthBase/mapResolution, 2*skew/mapResolution (cpp:69-72): loop invariants
    double lfRow0, rhRow0;                   // lfCurrentRow / rhCurrentRow at entry of the call (cpp:36)
    int32_t useConstraints;
    int32_t colLoA, colUpA;                  // xBounds of x2 = x8 (cpp:1063-1064)
    int32_t colLoB, colUpB;                  // xBounds of x4 = x6 (cpp:1065-1066)
    int32_t pad;
};
constexpr long long kMaxLatticePoints = 1ll << 24;  // row points the build-defined optimiser enumerates (oracle: same)

// Tile flag bits (one byte per cell in LDS).
enum : uint8_t {
    kFlagInMap = 1,      // cell index inside the map
    kFlagFinite = 2,     // isfinite(traversability)            GridMap::isValid
    kFlagBelowDef = 4,   // traversability < defaultFootholdThreshold   (raw compare, NaN -> false)
    kFlagBelowCand = 8,  // traversability < candidateFootholdThreshold (raw compare)
    kFlagFail = 16       // finite && (below candidate threshold || outside the search polygon)
};

}  // namespace fpe
