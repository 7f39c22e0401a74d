// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see fpo_planner.hpp / fpo_gridmap.hpp).
// Plain C entry points so tests/, smoke() and bench.py's cpu_baseline leg can drive the CPU
// restatement through ctypes.  Record layouts are documented in oracle/fpo.py (numpy dtypes).
#include <algorithm>
#include <atomic>
#include <cstring>
#include <thread>
#include <vector>

#include "fpo_planner.hpp"

using namespace fpo;

extern "C" {

struct fpo_map {
    int32_t rows, cols;
    double resolution;
    double position[2];
    const float* traversability;  // rows*cols
    const float* elevation;       // rows*cols
    int32_t row_major;            // 1: (i,j) at i*cols+j ; 0: column-major i+j*rows (grid_map native)
};

struct fpo_leg {  // 32 bytes
    int32_t row, col;
    double x, y;
    float z;
    uint8_t valid, source, pad[2];
};
struct fpo_centroid {  // 32 bytes
    double x, y;
    float z;
    int32_t row, col;
    uint8_t code, pad[3];
};
struct fpo_query {  // one checkFoothold call
    double cx, cy;
    float search_radius;
    int32_t n_vertices;
    double vx[8], vy[8];
};

static GridMap buildMap(const fpo_map* m) {
    GridMap g;
    g.size = {m->rows, m->cols};
    g.res = m->resolution;
    g.length = {static_cast<double>(m->rows) * g.res, static_cast<double>(m->cols) * g.res};
    g.position = {m->position[0], m->position[1]};
    const size_t n = (size_t)m->rows * m->cols;
    g.trav.resize(n);
    g.elev.resize(n);
    if (m->row_major) {
        for (int i = 0; i < m->rows; ++i)
            for (int j = 0; j < m->cols; ++j) {
                g.trav[(size_t)i + (size_t)j * m->rows] = m->traversability[(size_t)i * m->cols + j];
                g.elev[(size_t)i + (size_t)j * m->rows] = m->elevation[(size_t)i * m->cols + j];
            }
    } else {
        std::memcpy(g.trav.data(), m->traversability, n * sizeof(float));
        std::memcpy(g.elev.data(), m->elevation, n * sizeof(float));
    }
    return g;
}

// Opaque prepared map (avoids rebuilding the column-major copy for every call).
void* fpo_map_create(const fpo_map* m) { return new GridMap(buildMap(m)); }
void fpo_map_destroy(void* h) { delete static_cast<GridMap*>(h); }

static void planRange(const GridMap& map, const Params& p, const PoseSpec* poses, int b0, int b1, int nCycles,
                      fpo_leg* nominal, fpo_centroid* centroid, double* defaultNext, uint8_t* cycleOk,
                      double* stance) {
    PlanOutput out;
    for (int b = b0; b < b1; ++b) {
        planGlobalFootholds(map, p, poses[b], nCycles, out);
        for (int k = 0; k < nCycles * 4; ++k) {
            const size_t o = (size_t)b * nCycles * 4 + k;
            if (nominal) {
                fpo_leg& r = nominal[o];
                std::memset(&r, 0, sizeof(r));
                const LegResult& s = out.nominal[k];
                r.row = s.row; r.col = s.col; r.x = s.x; r.y = s.y; r.z = s.z; r.valid = s.valid; r.source = s.source;
            }
            if (centroid) {
                fpo_centroid& r = centroid[o];
                std::memset(&r, 0, sizeof(r));
                const CentroidResult& s = out.centroid[k];
                r.x = s.x; r.y = s.y; r.z = s.z; r.row = s.row; r.col = s.col; r.code = s.code;
            }
            if (defaultNext) {
                defaultNext[o * 3 + 0] = out.defaultNext[k].x;
                defaultNext[o * 3 + 1] = out.defaultNext[k].y;
                defaultNext[o * 3 + 2] = out.defaultNext[k].z;
            }
        }
        if (cycleOk)
            for (int g = 0; g < nCycles; ++g) cycleOk[(size_t)b * nCycles + g] = out.cycleOk[g];
        if (stance)
            for (int l = 0; l < 4; ++l) {
                stance[((size_t)b * 4 + l) * 3 + 0] = out.stance[l].x;
                stance[((size_t)b * 4 + l) * 3 + 1] = out.stance[l].y;
                stance[((size_t)b * 4 + l) * 3 + 2] = out.stance[l].z;
            }
    }
}

// Chained plan for B poses.  threads <= 1: caller thread; otherwise std::thread over pose blocks.
int fpo_plan(const void* mapHandle, const Params* params, const PoseSpec* poses, int B, int nCycles,
             int threads, fpo_leg* nominal, fpo_centroid* centroid, double* defaultNext, uint8_t* cycleOk,
             double* stance) {
    const GridMap& map = *static_cast<const GridMap*>(mapHandle);
    if (threads <= 1) {
        planRange(map, *params, poses, 0, B, nCycles, nominal, centroid, defaultNext, cycleOk, stance);
        return 0;
    }
    std::atomic<int> nextBlock{0};
    const int block = 16;
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t)
        pool.emplace_back([&]() {
            for (;;) {
                int b0 = nextBlock.fetch_add(block);
                if (b0 >= B) break;
                planRange(map, *params, poses, b0, std::min(B, b0 + block), nCycles, nominal, centroid,
                          defaultNext, cycleOk, stance);
            }
        });
    for (auto& th : pool) th.join();
    return 0;
}

// status[b] bit 0: getGaitCycleSearchGridMap fails in the first gait cycle of pose b (service returns false)
int fpo_pose_status(const void* mapHandle, const Params* params, const PoseSpec* poses, int B, uint8_t* status) {
    const GridMap& map = *static_cast<const GridMap*>(mapHandle);
    const Constants c = makeConstants(*params);
    for (int b = 0; b < B; ++b) {
        const double sx[4] = {c.LbHalf, -c.LbHalf, -c.LbHalf, c.LbHalf};
        const double sy[4] = {c.WbHalfNeg, c.WbHalfNeg, c.WbHalfPos, c.WbHalfPos};
        Point3 cur[4];
        for (int l = 0; l < 4; ++l) {  // stance (cpp:350-378) then setFirstGait (cpp:2693)
            Point3 s;
            s.x = sx[l]; s.y = sy[l]; s.z = 0;
            s.x += poses[b].pose[0]; s.y += poses[b].pose[1]; s.z += poses[b].pose[2];
            cur[l] = s;
            cur[l].x = s.x - c.stepHalf;
        }
        status[b] = getGaitCycleSearchGridMap(map, *params, cur, poses[b].pose[1], 0.0) ? 0 : 1;
    }
    return 0;
}

// The LATERAL side of the handler's gate (cpp:2307-2349, 931-934): getGaitCycleSearchGridMap's getSubmap is centred at
// (x of the opt track's next feet centre, initialPose_[1] + ajustedPose_[1]); every condition of getSubmap is a
// conjunction of an x-only and a y-only condition, and the y of cycle g depends on g only (the drift, cpp:1578).  With x
// at the map's centre (where the x conditions hold) the call fails iff the y side fails: fail[b] = first such cycle of
// pose b, 255 = none.  A request whose y side fails in cycle g is refused by the reference in cycle g at the latest,
// whatever its optimiser does (include/fpe.h, fpe_service_gate).
int fpo_gate_lateral(const void* mapHandle, const Params* params, const PoseSpec* poses, int B, int nCycles, uint8_t* fail) {
    const GridMap& map = *static_cast<const GridMap*>(mapHandle);
    for (int b = 0; b < B; ++b) {
        fail[b] = 255;
        double ajustedPoseY = 0.0;  // cpp:759
        for (int g = 0; g < nCycles; ++g) {
            // four feet whose polygon centre has x = map centre - stepLength, so that getGaitCycleSearchGridMap's
            // p.x = feetCenter.x + stepLength lands near the map's centre (any x whose x side passes serves)
            const double cx = map.position.x - static_cast<double>(params->stepLength);
            Point3 cur[4];
            const double dx[4] = {0.2, -0.2, -0.2, 0.2}, dy[4] = {-0.1, -0.1, 0.1, 0.1};
            for (int l = 0; l < 4; ++l) {
                cur[l].x = cx + dx[l];
                cur[l].y = dy[l];
                cur[l].z = 0.0;
            }
            if (!getGaitCycleSearchGridMap(map, *params, cur, poses[b].pose[1], ajustedPoseY)) {
                fail[b] = static_cast<uint8_t>(g);
                break;
            }
            ajustedPoseY += params->lateralDrift;  // cpp:1578
        }
    }
    return 0;
}

// Evaluation products of ONE trot plan: per track k (0 centroid, 1 nominal)
//   path[k]: up to nCycles x 3 doubles, dist[k] / speed[k]: up to 2*nCycles doubles; counts[k] = {nPath, nKpi}.
int fpo_plan_products(const void* mapHandle, const Params* params, const PoseSpec* pose, int nCycles, double* path0,
                      double* path1, double* dist0, double* dist1, double* speed0, double* speed1, int32_t* counts) {
    const GridMap& map = *static_cast<const GridMap*>(mapHandle);
    PlanOutput out;
    planGlobalFootholds(map, *params, *pose, nCycles, out);
    double* path[2] = {path0, path1};
    double* dist[2] = {dist0, dist1};
    double* speed[2] = {speed0, speed1};
    for (int k = 0; k < 2; ++k) {
        counts[2 * k] = (int32_t)out.feetCenterPath[k].size();
        counts[2 * k + 1] = (int32_t)out.feetDistance[k].size();
        for (size_t i = 0; i < out.feetCenterPath[k].size(); ++i) {
            path[k][3 * i] = out.feetCenterPath[k][i].x;
            path[k][3 * i + 1] = out.feetCenterPath[k][i].y;
            path[k][3 * i + 2] = out.feetCenterPath[k][i].z;
        }
        for (size_t i = 0; i < out.feetDistance[k].size(); ++i) {
            dist[k][i] = out.feetDistance[k][i];
            speed[k][i] = out.cogSpeed[k][i];
        }
    }
    return 0;
}

// n independent checkFoothold calls (cpp:2001) with arbitrary polygons.
int fpo_search_legs(const void* mapHandle, const Params* params, const fpo_query* q, int n, fpo_leg* out) {
    const GridMap& map = *static_cast<const GridMap*>(mapHandle);
    for (int k = 0; k < n; ++k) {
        Polygon poly;
        for (int v = 0; v < q[k].n_vertices; ++v) poly.addVertex({q[k].vx[v], q[k].vy[v]});
        LegResult r;
        checkFoothold(map, {q[k].cx, q[k].cy}, params->footRadius, q[k].search_radius, poly, *params, r);
        std::memset(&out[k], 0, sizeof(fpo_leg));
        out[k].row = r.row; out[k].col = r.col; out[k].x = r.x; out[k].y = r.y; out[k].z = r.z;
        out[k].valid = r.valid; out[k].source = r.source;
    }
    return 0;
}

int fpo_centroid_method(const void* mapHandle, const Params* params, double x, double y, float searchRadius,
                        fpo_centroid* out) {
    const GridMap& map = *static_cast<const GridMap*>(mapHandle);
    CentroidResult r;
    checkFootholdUseCentroidMethod(map, {x, y}, searchRadius, *params, r);
    std::memset(out, 0, sizeof(*out));
    out->x = r.x; out->y = r.y; out->z = r.z; out->row = r.row; out->col = r.col; out->code = r.code;
    return 0;
}

float fpo_mean_height(const void* mapHandle, double x, double y, float radius, double h) {
    return getFootholdMeanHeight(*static_cast<const GridMap*>(mapHandle), {x, y}, radius, h);
}

// ---- grid_map semantics probes (KATs / property tests) ------------------------------------------
int fpo_spiral_cells(const void* mapHandle, double cx, double cy, double radius, int32_t* ij, int maxCells) {
    const GridMap& map = *static_cast<const GridMap*>(mapHandle);
    int n = 0;
    for (SpiralIterator it(map, {cx, cy}, radius); !it.isPastEnd(); ++it) {
        if (n < maxCells) {
            ij[2 * n] = (*it).i;
            ij[2 * n + 1] = (*it).j;
        }
        ++n;
    }
    return n;
}
int fpo_circle_cells(const void* mapHandle, double cx, double cy, double radius, int32_t* ij, int maxCells) {
    const GridMap& map = *static_cast<const GridMap*>(mapHandle);
    int n = 0;
    for (CircleIterator it(map, {cx, cy}, radius); !it.isPastEnd(); ++it) {
        if (n < maxCells) {
            ij[2 * n] = (*it).i;
            ij[2 * n + 1] = (*it).j;
        }
        ++n;
    }
    return n;
}
int fpo_get_index(const void* mapHandle, double x, double y, int32_t* ij) {
    const GridMap& map = *static_cast<const GridMap*>(mapHandle);
    Idx2 idx;
    bool ok = map.getIndex({x, y}, idx);
    ij[0] = idx.i;
    ij[1] = idx.j;
    return ok ? 1 : 0;
}
int fpo_get_position(const void* mapHandle, int i, int j, double* xy) {
    const GridMap& map = *static_cast<const GridMap*>(mapHandle);
    Vec2 p{0, 0};
    bool ok = map.getPosition({i, j}, p);
    xy[0] = p.x;
    xy[1] = p.y;
    return ok ? 1 : 0;
}
// out: [tl_i, tl_j, rows, cols], pos_len: [px, py, lx, ly]
int fpo_submap_info(const void* mapHandle, double x, double y, double lx, double ly, int32_t* out, double* posLen) {
    const GridMap& map = *static_cast<const GridMap*>(mapHandle);
    SubmapInfo info;
    bool ok = getSubmapInformation(info, {x, y}, {lx, ly}, map.length, map.position, map.res, map.size);
    if (!ok) return 0;
    out[0] = info.topLeft.i; out[1] = info.topLeft.j; out[2] = info.size.i; out[3] = info.size.j;
    posLen[0] = info.position.x; posLen[1] = info.position.y; posLen[2] = info.length.x; posLen[3] = info.length.y;
    return 1;
}
int fpo_polygon_inside(const double* vx, const double* vy, int n, double x, double y) {
    Polygon poly;
    for (int v = 0; v < n; ++v) poly.addVertex({vx[v], vy[v]});
    return poly.isInside({x, y}) ? 1 : 0;
}
void fpo_polygon_center(const double* feet12, double* out3) {
    Point3 f[4];
    for (int l = 0; l < 4; ++l) {
        f[l].x = feet12[l * 3];
        f[l].y = feet12[l * 3 + 1];
        f[l].z = feet12[l * 3 + 2];
    }
    Point3 c = getPolygonCenter(f[0], f[1], f[2], f[3]);
    out3[0] = c.x; out3[1] = c.y; out3[2] = c.z;
}
void fpo_constants(const Params* p, double* out /*[14]*/) {
    Constants c = makeConstants(*p);
    out[0] = c.LbHalf; out[1] = c.WbHalfNeg; out[2] = c.WbHalfPos;
    for (int l = 0; l < 4; ++l) { out[3 + l] = c.biasX[l]; out[7 + l] = c.biasY[l]; }
    out[11] = c.stepHalf; out[12] = c.step; out[13] = c.stepQuarter;
}
// as-written cost emulation: single-threaded plan with by-value map copies; returns the copy count
unsigned long long fpo_plan_as_written(const void* mapHandle, const Params* params, const PoseSpec* poses, int B, int nCycles,
                                       fpo_leg* nominal) {
    const GridMap& map = *static_cast<const GridMap*>(mapHandle);
    setEmulateByValueCopies(true);
    planRange(map, *params, poses, 0, B, nCycles, nominal, nullptr, nullptr, nullptr, nullptr);
    const unsigned long long n = byValueCopyCount();
    setEmulateByValueCopies(false);
    return n;
}

// ---- the opt track (fpo_opt.cpp) ------------------------------------------------------------------------------
struct fpo_opt_foothold {  // 32 bytes: RF/RH/LH/LF_footholdResult_opt (cpp:1283-1314)
    double x, y;
    float z;
    int32_t row, col;  // (int)x[2k], (int)x[2k+1]: the gaitMap_ index the position was taken from (cpp:1287-1312)
    uint8_t foot_id, gait_cycle_id, committed, pad;
};
struct fpo_opt_cycle {  // 240 bytes; same layout as fpe_opt_cycle (include/fpe.h)
    int32_t gait_top_left[2], gait_size[2];
    int32_t nominal_index[8], centroid_index[8];
    int32_t traversable_row[2][4];
    int32_t x_lower[8], x_upper[8];
    int32_t x[8];
    double minf, lf_current_row, rh_current_row;
    uint8_t centroid_code[4];
    uint8_t gate_failed, committed, solver_status, pad;
};

// cycleOk: [B * nCycles] footholdValidation_ of the nominal track (from fpo_plan).  footholds: [B * nCycles * 4],
// cycles: [B * nCycles], gateFailCycle: [B] (255 = none).  Cycles from the failing one on stay zero.
int fpo_plan_opt(const void* mapHandle, const Params* params, const OptParams* op, const PoseSpec* poses, int B, int nCycles,
                 const uint8_t* cycleOk, fpo_opt_foothold* footholds, fpo_opt_cycle* cycles, uint8_t* gateFailCycle) {
    const GridMap& map = *static_cast<const GridMap*>(mapHandle);
    OptOutput out;
    for (int b = 0; b < B; ++b) {
        planOptTrack(map, *params, *op, poses[b], nCycles, cycleOk + (size_t)b * nCycles, out);
        if (gateFailCycle) gateFailCycle[b] = out.gateFailCycle < 0 ? 255 : static_cast<uint8_t>(out.gateFailCycle);
        for (int g = 0; g < nCycles; ++g) {
            const OptCycle& oc = out.cycles[(size_t)g];
            const bool ran = poses[b].gait == 0 && (out.gateFailCycle < 0 || g < out.gateFailCycle);
            if (cycles) {
                fpo_opt_cycle& r = cycles[(size_t)b * nCycles + g];
                std::memset(&r, 0, sizeof(r));
                r.gate_failed = oc.gateFailed;
                if (ran) {
                    for (int k = 0; k < 2; ++k) { r.gait_top_left[k] = oc.gaitTopLeft[k]; r.gait_size[k] = oc.gaitSize[k]; }
                    for (int k = 0; k < 8; ++k) {
                        r.nominal_index[k] = oc.nominalIndex[k]; r.centroid_index[k] = oc.centroidIndex[k];
                        r.x_lower[k] = oc.xLower[k]; r.x_upper[k] = oc.xUpper[k];
                        r.x[k] = static_cast<int32_t>(oc.x[k]);
                    }
                    for (int k = 0; k < 4; ++k) {
                        r.traversable_row[0][k] = oc.traversableRow[0][k]; r.traversable_row[1][k] = oc.traversableRow[1][k];
                        r.centroid_code[k] = oc.centroidCode[k];
                    }
                    r.minf = oc.minf; r.lf_current_row = oc.lfCurrentRow; r.rh_current_row = oc.rhCurrentRow;
                    r.committed = oc.committed; r.solver_status = oc.solverStatus;
                }
            }
            if (footholds) {
                static const int slot[4] = {2, 1, 3, 0};  // position of RF,RH,LH,LF in the optimiser's order LF,RH,RF,LH
                for (int l = 0; l < 4; ++l) {
                    fpo_opt_foothold& f = footholds[((size_t)b * nCycles + g) * 4 + l];
                    std::memset(&f, 0, sizeof(f));
                    f.foot_id = static_cast<uint8_t>(l);
                    f.gait_cycle_id = static_cast<uint8_t>(g);
                    if (!ran) continue;
                    f.x = oc.result[l].x; f.y = oc.result[l].y; f.z = static_cast<float>(oc.result[l].z);
                    f.row = static_cast<int32_t>(oc.x[2 * slot[l]]); f.col = static_cast<int32_t>(oc.x[2 * slot[l] + 1]);
                    f.committed = oc.committed;
                }
            }
        }
    }
    return 0;
}

// Evaluation products of ONE pose's opt track: path [nCycles x 3] (feet centres pushed onto centroidFeetCenterPath,
// cpp:946), dist / speed [2 * nCycles]; counts = {nPath, nKpi, gateFailCycle (-1 none)}.
int fpo_plan_opt_products(const void* mapHandle, const Params* params, const OptParams* op, const PoseSpec* pose, int nCycles,
                          const uint8_t* cycleOk, double* path, double* dist, double* speed, int32_t* counts) {
    const GridMap& map = *static_cast<const GridMap*>(mapHandle);
    OptOutput out;
    planOptTrack(map, *params, *op, *pose, nCycles, cycleOk, out);
    counts[0] = (int32_t)out.feetCenterPath.size();
    counts[1] = (int32_t)out.feetDistance.size();
    counts[2] = out.gateFailCycle;
    for (size_t i = 0; i < out.feetCenterPath.size(); ++i) {
        path[3 * i] = out.feetCenterPath[i].x; path[3 * i + 1] = out.feetCenterPath[i].y; path[3 * i + 2] = out.feetCenterPath[i].z;
    }
    for (size_t i = 0; i < out.feetDistance.size(); ++i) {
        dist[i] = out.feetDistance[i];
        speed[i] = out.cogSpeed[i];
    }
    return 0;
}

// One pose's opt track with the optimiser's x of the first nForced cycles supplied by the caller (planOptTrack's forcedX): the
// cycles' records (problem of every cycle that ran, x as truncated by the chain) and the gate verdict.  Test infrastructure of
// tests/golden/make_cobyla_golden.py (an external optimiser drives the literal chain).
int fpo_plan_opt_forced(const void* mapHandle, const Params* params, const OptParams* op, const PoseSpec* pose, int nCycles,
                        const uint8_t* cycleOk, const double* forcedX, int nForced, fpo_opt_cycle* cycles, int32_t* gateFailCycle) {
    const GridMap& map = *static_cast<const GridMap*>(mapHandle);
    OptOutput out;
    planOptTrack(map, *params, *op, *pose, nCycles, cycleOk, out, forcedX, nForced);
    *gateFailCycle = out.gateFailCycle;
    for (int g = 0; g < nCycles; ++g) {
        const OptCycle& oc = out.cycles[(size_t)g];
        fpo_opt_cycle& r = cycles[g];
        std::memset(&r, 0, sizeof(r));
        r.gate_failed = oc.gateFailed;
        if (!(out.gateFailCycle < 0 || g < out.gateFailCycle)) continue;
        for (int k = 0; k < 2; ++k) { r.gait_top_left[k] = oc.gaitTopLeft[k]; r.gait_size[k] = oc.gaitSize[k]; }
        for (int k = 0; k < 8; ++k) {
            r.nominal_index[k] = oc.nominalIndex[k]; r.centroid_index[k] = oc.centroidIndex[k];
            r.x_lower[k] = oc.xLower[k]; r.x_upper[k] = oc.xUpper[k];
            r.x[k] = static_cast<int32_t>(oc.x[k]);
        }
        for (int k = 0; k < 4; ++k) {
            r.traversable_row[0][k] = oc.traversableRow[0][k]; r.traversable_row[1][k] = oc.traversableRow[1][k];
            r.centroid_code[k] = oc.centroidCode[k];
        }
        r.minf = oc.minf; r.lf_current_row = oc.lfCurrentRow; r.rh_current_row = oc.rhCurrentRow;
        r.committed = oc.committed; r.solver_status = oc.solverStatus;
    }
    return 0;
}

// the build-defined optimiser alone (tests: hand-made problems)
int fpo_solve_lattice(const OptParams* op, const int32_t* nominalIndex, const int32_t* centroidIndex, const int32_t* lo,
                      const int32_t* up, double lengthBase, double skew, double mapResolution, double lfCurrentRow,
                      double rhCurrentRow, double* x, double* minf) {
    return solveLattice(*op, nominalIndex, centroidIndex, lo, up, lengthBase, skew, mapResolution, lfCurrentRow, rhCurrentRow, x, minf);
}

// checkFootholdUseCentroidMethod on a SUBMAP of the map (the opt track's use, cpp:1010): the submap is
// getSubmap(subCentre, subLength); out6 = {code, beginRow, endRow, row, col, submapOk}.
int fpo_centroid_on_submap(const void* mapHandle, const Params* params, double sx, double sy, double slx, double sly, double x,
                           double y, float searchRadius, fpo_centroid* out, int32_t* out6) {
    const GridMap& map = *static_cast<const GridMap*>(mapHandle);
    bool ok = false;
    GridMap sub = map.getSubmap({sx, sy}, {slx, sly}, ok, nullptr, true);
    std::memset(out, 0, sizeof(*out));
    for (int k = 0; k < 6; ++k) out6[k] = 0;
    if (!ok) return 0;
    CentroidResult r;
    int b = 0, e = 0;
    checkFootholdUseCentroidMethod(sub, {x, y}, searchRadius, *params, r, &map, &b, &e);
    out->x = r.x; out->y = r.y; out->z = r.z; out->row = r.row; out->col = r.col; out->code = r.code;
    out6[0] = r.code; out6[1] = b; out6[2] = e; out6[3] = r.row; out6[4] = r.col; out6[5] = 1;
    return 1;
}

int fpo_sizeof(int which) {
    switch (which) {
        case 0: return (int)sizeof(Params);
        case 1: return (int)sizeof(PoseSpec);
        case 2: return (int)sizeof(fpo_leg);
        case 3: return (int)sizeof(fpo_centroid);
        case 4: return (int)sizeof(fpo_query);
        case 5: return (int)sizeof(fpo_map);
        case 6: return (int)sizeof(OptParams);
        case 7: return (int)sizeof(fpo_opt_foothold);
        case 8: return (int)sizeof(fpo_opt_cycle);
    }
    return -1;
}

}  // extern "C"
