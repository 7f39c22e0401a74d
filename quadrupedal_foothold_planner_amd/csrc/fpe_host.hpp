// fpe_host.hpp — host-side logic of the engine that needs no GPU: parameter typing, derived
// constants, the spiral rank table, tile sizing and GlobalFootholds message assembly.
#pragma once
#include <cstdint>
#include <vector>

#include "fpe_device.hpp"

namespace fpe {

// Upper bound of SpiralIterator rings the rank table is built for (ceil(R/res) must not exceed it).
constexpr int kMaxRings = 96;

struct SpiralTable {
    std::vector<int16_t> di, dj;
    std::vector<uint8_t> ring;
    std::vector<int32_t> ringStart;  // [nRings + 2]
    int maxRing = 0;
};
// SpiralIterator visiting order for rings 0..nRings as index offsets from the centre cell
// (grid_map SpiralIterator::generateRing; consumed from the back — SURVEY.md App. A.5).
void build_spiral_table(int nRings, SpiralTable& out);

// initialize() constants with the reference's float/double typing (cpp:340-421, 2693).
void derive_constants(const fpe_params& p, const MapGeom& g, float maxSearchRadius, const Tuning& tuning, PlanConsts& out);

// Offsets of a cell-centred disc of radius double(footRadius) and the proof that they are
// rounding-robust on this map (see PlanConsts::footRobust).
void derive_foot_offsets(float footRadius, const MapGeom& g, PlanConsts& out);

// Tile half-width (cells) that covers: every spiral candidate's foot disc, the default disc and
// the centroid rectangle for search radii up to maxSearchRadius.
int tile_halfwidth(float maxSearchRadius, float footRadius, double resolution);

// Half-width of the window that provably contains every cell a leg search can touch (bit-window kernels), or 0
// when one of the proofs those kernels rest on does not hold for these parameters on this map.
int bits_window_halfwidth(const PlanConsts& c, const MapGeom& g);

// number of rings ceil(double(R)/res) as SpiralIterator computes it
int spiral_rings(float searchRadius, double resolution);

int validate_params(const fpe_params& p);

// Opt track: the file-scope NLopt globals (cpp:28-51, 497-498, 514), the constraint thresholds
// t1..t4 (cpp:1156-1159) and the column bounds of xBounds (cpp:1063-1066, 528-529), with the reference's typing.
// FPE_E_INVALID_ARG when a weight / scale / tolerance is not finite.
int derive_opt_constants(const fpe_params& p, const fpe_opt_params& op, const MapGeom& g, const PlanConsts& pc, OptConsts& out);

// GlobalFootholds message content from one pose's plan outputs (cpp:591-699, 1378-1396, 1574).
void assemble_global_footholds(const fpe_foothold* nominal, const uint8_t* cycleOk, const double* stance,
                               int nCycles, fpe_global_footholds* msg);
// centroidGlobalFootholdsMsg_ content of one call (cpp:709-727, 1444-1462): its own bookkeeping, not the nominal one
void assemble_centroid_footholds(const fpe_centroid_foothold* centroid, const uint8_t* cycleOk, const double* stance,
                                 int nCycles, fpe_global_footholds* msg);
void assemble_track_report(const double* resultXYZ, const uint8_t* cycleOk, const double* stance, int nCycles,
                           const fpe_params& params, fpe_track_report* rep);
// optGlobalFootholdsMsg_ content of one call (cpp:737-755, 1510-1532): bookkeeping as the centroid message
void assemble_opt_footholds(const fpe_opt_foothold* opt, const uint8_t* cycleOk, const double* stance, int nCycles,
                            fpe_global_footholds* msg);
// centroidFeetCenterPath as the reference fills it (cpp:792 and cpp:946): per cycle the centroid track's feet centre,
// then the opt track's
void interleave_centroid_path(fpe_track_report* centroid, const fpe_track_report& opt);

}  // namespace fpe
