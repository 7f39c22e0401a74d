// fpe_kernels.hip — hand-written HIP for gfx950 (MI355X, wave64): the foothold-search hot path.
//
// Layout: one workgroup (256 threads = 4 wavefronts) per body pose; wavefront w searches leg w
// (RF, RH, LH, LF) — the device counterpart of the reference's 4 std::thread(checkFoothold)
// (cpp:863-909).  The gait-cycle chain of globalFootholdPlan (cpp:762-1579) runs inside the
// workgroup: feet of the three tracks live in LDS, every cycle ends with a commit/skip decision.
// Each wavefront stages the traversability window around its search centre into LDS as one flag
// byte per cell (coalesced row reads), then evaluates the default disc, the spiral candidates
// (rank table + wave ballot = argmin of SpiralIterator rank), the centroid rectangle and the mean
// heights from those flags.  No MFMA: nothing here is a contraction.
//
// All geometry is f64 in the reference's expression order (fpe_gridmath.hpp); compile with
// -ffp-contract=off.  Reference citations: "cpp:" = foothold_planner/src/FootholdPlanner.cpp.
#include <hip/hip_runtime.h>

#include "fpe_device.hpp"

namespace fpe {

namespace {

constexpr int kWave = 64;

__device__ __forceinline__ int lane_id() { return static_cast<int>(threadIdx.x) & (kWave - 1); }
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ bool wave_any(bool p) { return __ballot(p) != 0ull; }

// A search centre must be finite and of sane magnitude.  The reference has no such test: a
// non-finite centre (reachable once the centroid track has committed its "no case" (0,0,0)
// results, cpp:1777-1944, and the feet polygon degenerates) sends NaN through
// getIndexFromPosition's (int) cast, which is undefined behaviour.  Engine and oracle both define
// it as "no cell is visited": invalid leg, getSubmap failure, mean height = h.
__device__ __forceinline__ bool centre_usable(double x, double y) { return fabs(x) <= 1e9 && fabs(y) <= 1e9; }

// ---- per-wave search context (all members wave-uniform) ---------------------------------------
struct LegCtx {
    double cx, cy;   // search centre = centroid-track next position of this leg (cpp:861-862)
    double R, R2;    // double(float searchRadius), R*R (SpiralIterator radiusSquare_)
    float Rf;        // the float radius (centroid rectangle uses searchRadius_*2 in f32, cpp:1616)
    int nRings;      // ceil(R / res)
    int ici, icj;    // getIndex(centre)
    int ti0, tj0;    // tile origin (cell index of tile[0])
    int nv;          // polygon vertex count
    const double* vx;  // polygon vertices (LDS)
    const double* vy;
};

__device__ __forceinline__ uint8_t tile_at(const uint8_t* tile, const PlanConsts& pc, const LegCtx& c, int i, int j) {
    const int a = i - c.ti0, b = j - c.tj0;
    // cells a search can touch are inside the tile by construction (tileH); anything else is
    // treated as "outside the map" instead of reading out of bounds
    if (static_cast<unsigned>(a) >= static_cast<unsigned>(pc.tileW) ||
        static_cast<unsigned>(b) >= static_cast<unsigned>(pc.tileW))
        return 0;
    return tile[a * pc.tileW + b];
}

// Stage the flag tile: coalesced reads of the traversability rows, one byte per cell to LDS.
__device__ void stage_tile(const DevMap& m, const PlanConsts& pc, const LegCtx& c, uint8_t* tile, int lane) {
    const int W = pc.tileW;
    const int n = W * W;
    for (int t = lane; t < n; t += kWave) {
        const int a = static_cast<int>(fastdiv(static_cast<uint32_t>(t), pc.tileWMagic));
        const int b = t - a * W;
        const int i = c.ti0 + a, j = c.tj0 + b;
        uint8_t f = 0;
        if (in_range(i, j, m.g.rows, m.g.cols)) {
            const float v = m.trav[static_cast<size_t>(i) * m.g.cols + j];
            f = kFlagInMap;
            if (__builtin_isfinite(v)) f |= kFlagFinite;
            if (v < pc.thrDefault) f |= kFlagBelowDef;
            if (v < pc.thrCandidate) f |= kFlagBelowCand;
        }
        tile[t] = f;
    }
}

// Second pass, only when the spiral search runs: fold the polygon test into kFlagFail.
// checkCirclePolygonFoothold (cpp:2132-2138): a FINITE cell fails when it is below the candidate
// threshold or its centre is outside the polygon; non-finite cells never fail.
__device__ void fold_polygon(const DevMap& m, const PlanConsts& pc, const LegCtx& c, uint8_t* tile, int lane) {
    const int W = pc.tileW;
    const int n = W * W;
    for (int t = lane; t < n; t += kWave) {
        uint8_t f = tile[t];
        if ((f & kFlagInMap) && (f & kFlagFinite)) {
            bool fail = (f & kFlagBelowCand) != 0;
            if (!fail) {
                const int a = static_cast<int>(fastdiv(static_cast<uint32_t>(t), pc.tileWMagic));
                const int b = t - a * W;
                const double px = cell_pos(m.g.baseX, m.g.res, c.ti0 + a);
                const double py = cell_pos(m.g.baseY, m.g.res, c.tj0 + b);
                fail = !polygon_inside(c.vx, c.vy, c.nv, px, py);
            }
            if (fail) tile[t] = f | kFlagFail;
        }
    }
}

// getFootholdMeanHeight (cpp:2520-2554) for one disc, cooperatively: lanes test membership and
// load elevation, then the f32 sum is accumulated in the CircleIterator's row-major order.
__device__ float mean_height_wave(const DevMap& m, const PlanConsts& pc, double cx, double cy, int lane) {
    if (!centre_usable(cx, cy)) return static_cast<float>(static_cast<double>(0.0f) + pc.h);  // no cell visited
    const BBox bb = circle_bbox(m.g, cx, cy, pc.rf);
    const int ni = uni(bb.ni), nj = uni(bb.nj), i0 = uni(bb.i0), j0 = uni(bb.j0);
    const int nb = ni * nj;
    const uint32_t magic = fastdiv_magic(static_cast<uint32_t>(nj));
    float sum = 0.0f, last = 0.0f;
    int cnt = 0;
    for (int base = 0; base < nb; base += kWave) {
        const int t = base + lane;
        bool vis = false;
        float v = 0.0f;
        if (t < nb) {
            const int a = static_cast<int>(fastdiv(static_cast<uint32_t>(t), magic));
            const int i = i0 + a, j = j0 + (t - a * nj);
            if (in_range(i, j, m.g.rows, m.g.cols) && cell_in_disc(m.g, i, j, cx, cy, pc.rf2)) {
                vis = true;
                const float e = m.elev[static_cast<size_t>(i) * m.g.cols + j];
                v = __builtin_isfinite(e) ? e : 0.0f;  // cpp:2532-2537
            }
        }
        unsigned long long mask = __ballot(vis);
        while (mask) {
            const int l = __builtin_ctzll(mask);
            mask &= mask - 1;
            const float e = __shfl(v, l);
            last = e;
            if (e < 10) {  // cpp:2539
                cnt++;
                sum = sum + e;
            }
        }
    }
    const float mean = (cnt != 0) ? (sum / cnt) : last;           // cpp:2547-2551
    return static_cast<float>(static_cast<double>(mean) + pc.h);  // cpp:2553 (float + double)
}

// checkDefaultFoothold (cpp:2039-2082): valid iff the disc around the CONTINUOUS centre visits at
// least one cell and no finite visited cell is below defaultFootholdThreshold_.
__device__ bool default_check_wave(const DevMap& m, const PlanConsts& pc, const LegCtx& c, const uint8_t* tile, int lane) {
    const BBox bb = circle_bbox(m.g, c.cx, c.cy, pc.rf);
    const int ni = uni(bb.ni), nj = uni(bb.nj), i0 = uni(bb.i0), j0 = uni(bb.j0);
    const int nb = ni * nj;
    const uint32_t magic = fastdiv_magic(static_cast<uint32_t>(nj));
    bool any = false, fail = false;
    for (int base = 0; base < nb; base += kWave) {
        const int t = base + lane;
        if (t < nb) {
            const int a = static_cast<int>(fastdiv(static_cast<uint32_t>(t), magic));
            const int i = i0 + a, j = j0 + (t - a * nj);
            if (in_range(i, j, m.g.rows, m.g.cols) && cell_in_disc(m.g, i, j, c.cx, c.cy, pc.rf2)) {
                any = true;
                const uint8_t f = tile_at(tile, pc, c, i, j);
                if ((f & kFlagFinite) && (f & kFlagBelowDef)) fail = true;
            }
        }
    }
    return wave_any(any) && !wave_any(fail);
}

// checkCirclePolygonFoothold (cpp:2117-2163) for the cell-centred disc of candidate (i, j),
// one lane per candidate.
__device__ __forceinline__ bool candidate_disc_ok(const DevMap& m, const PlanConsts& pc, const LegCtx& c,
                                                  const uint8_t* tile, int i, int j) {
    const double fx = cell_pos(m.g.baseX, m.g.res, i);
    const double fy = cell_pos(m.g.baseY, m.g.res, j);
    const BBox bb = circle_bbox(m.g, fx, fy, pc.rf);
    bool any = false;
    for (int a = 0; a < bb.ni; ++a) {
        for (int b = 0; b < bb.nj; ++b) {
            const int qi = bb.i0 + a, qj = bb.j0 + b;
            if (in_range(qi, qj, m.g.rows, m.g.cols) && cell_in_disc(m.g, qi, qj, fx, fy, pc.rf2)) {
                if (tile_at(tile, pc, c, qi, qj) & kFlagFail) return false;
                any = true;
            }
        }
    }
    return any;
}

// checkCandidateFoothold (cpp:2085-2114): first valid cell in SpiralIterator order.  Lane k of a
// round evaluates the candidate of rank base+k; the lowest set ballot bit is the argmin of rank.
__device__ bool candidate_search_wave(const DevMap& m, const PlanConsts& pc, const SpiralLut& lut, const LegCtx& c,
                                      const uint8_t* tile, int lane, int& wi, int& wj) {
    const int nR = c.nRings < lut.maxRing ? c.nRings : lut.maxRing;
    const int M = uni(lut.ringStart[nR + 1]);
    for (int base = 0; base < M; base += kWave) {
        const int k = base + lane;
        bool ok = false;
        int i = 0, j = 0;
        if (k < M) {
            i = c.ici + lut.di[k];
            j = c.icj + lut.dj[k];
            ok = in_range(i, j, m.g.rows, m.g.cols);
            if (ok) {
                // SpiralIterator::generateRing filters rings nRings-1 and nRings by isInside;
                // the centre cell (ring 0) is pushed unfiltered by the constructor
                const int r = lut.ring[k];
                if (r >= 1 && (r == c.nRings || r + 1 == c.nRings))
                    ok = cell_in_disc(m.g, i, j, c.cx, c.cy, c.R2);
            }
            if (ok) ok = candidate_disc_ok(m, pc, c, tile, i, j);
        }
        const unsigned long long mask = __ballot(ok);
        if (mask) {
            const int l = __builtin_ctzll(mask);
            wi = __shfl(i, l);
            wj = __shfl(j, l);
            return true;
        }
    }
    return false;
}

struct CentroidOut {
    double x, y;
    float z;
    int row, col;
    int code;
};

// checkFootholdUseCentroidMethod (cpp:1605-1997) on the rectangle (2R x R) around the centre.
// zCentre = mean height at the centre (the whole-region-valid result reuses it, cpp:1687).
__device__ CentroidOut centroid_wave(const DevMap& m, const PlanConsts& pc, const LegCtx& c, const uint8_t* tile,
                                     float zCentre, int lane) {
    CentroidOut o;
    o.x = 0.0;
    o.y = 0.0;
    o.z = 0.0f;
    o.row = -1;
    o.col = -1;
    o.code = 5;
    const double lx = static_cast<double>(c.Rf * 2);  // cpp:1616: float * int stays float
    const double ly = static_cast<double>(c.Rf);      // cpp:1617
    const Submap s = submap_info(m.g, c.cx, c.cy, lx, ly);
    if (!s.ok) {  // cpp:1628-1631
        o.code = 6;
        return o;
    }
    const int ni = uni(s.ni), nj = uni(s.nj), i0 = uni(s.i0), j0 = uni(s.j0);
    const int ncell = ni * nj;
    const uint32_t magic = fastdiv_magic(static_cast<uint32_t>(nj));
    // whole-region test, cpp:1649-1658 (raw `<`: NaN passes)
    bool below = false;
    for (int t = lane; t < ncell; t += kWave) {
        const int a = static_cast<int>(fastdiv(static_cast<uint32_t>(t), magic));
        if (tile_at(tile, pc, c, i0 + a, j0 + (t - a * nj)) & kFlagBelowDef) below = true;
    }
    const bool whole = ncell > 0 && !wave_any(below);
    if (whole) {  // cpp:1684-1689
        o.x = c.cx;
        o.y = c.cy;
        o.z = zCentre;
        o.row = c.ici;
        o.col = c.icj;
        o.code = 0;
        return o;
    }
    // row scan, cpp:1717-1750 (in-bounds columns only): lane = row
    const int bottomRow = ni - 1, rightCol = nj - 1;
    int minRow = 0, maxRow = 0, k = 0;
    for (int rbase = 0; rbase < ni; rbase += kWave) {
        const int r = rbase + lane;
        bool blocked = false;
        if (r < ni) {
            int cnt = 0;
            for (int cc = 0; cc < nj; ++cc)
                if (tile_at(tile, pc, c, i0 + r, j0 + cc) & kFlagBelowDef) ++cnt;
            blocked = cnt > ((rightCol + 1) * 0.5);  // cpp:1743
        }
        const unsigned long long mask = __ballot(blocked);
        if (mask) {
            if (k == 0) minRow = rbase + __builtin_ctzll(mask);
            maxRow = rbase + 63 - __builtin_clzll(mask);
            k += __builtin_popcountll(mask);
        }
    }
    int newRow, newCol;
    if (minRow == 0 && maxRow != bottomRow) {  // case 1, cpp:1777-1786
        newRow = static_cast<int>(floor((maxRow + bottomRow + 1) * 0.5));
        newCol = static_cast<int>(floor((rightCol + 1) * 0.5));
        o.code = 1;
    } else if (minRow != 0 && maxRow != bottomRow) {  // case 2, cpp:1843-1886
        if ((minRow - 0) >= (bottomRow - maxRow)) {
            newRow = static_cast<int>(ceil(minRow * 0.5));
            o.code = 2;
        } else {
            newRow = static_cast<int>(floor((maxRow + bottomRow) * 0.5));
            o.code = 3;
        }
        newCol = static_cast<int>(floor((rightCol + 0) * 0.5));
    } else if (minRow != 0 && maxRow == bottomRow) {  // case 3, cpp:1944-1952
        newRow = static_cast<int>(ceil(minRow * 0.5));
        newCol = static_cast<int>(floor((rightCol + 0) * 0.5));
        o.code = 4;
    } else {
        return o;  // first and last row blocked: no branch taken, result stays (0,0,0)
    }
    // map.getPosition(newIndex) on the SUBMAP (cpp:1816), height on the full map (cpp:1820)
    o.x = cell_pos(s.baseX, m.g.res, newRow);
    o.y = cell_pos(s.baseY, m.g.res, newCol);
    o.z = mean_height_wave(m, pc, o.x, o.y, lane);
    o.row = index_of(o.x, m.g.orgX, m.g.posX, m.g.res);
    o.col = index_of(o.y, m.g.orgY, m.g.posY, m.g.res);
    return o;
}

struct NominalOut {
    int row, col;
    double x, y;
    float z;
    int valid, source;
};

// checkFoothold (cpp:2001-2036).  zCentre = getFootholdMeanHeight at the DEFAULT centre, which the
// reference uses even when a spiral candidate was chosen (cpp:2029).
__device__ NominalOut nominal_wave(const DevMap& m, const PlanConsts& pc, const SpiralLut& lut, const LegCtx& c,
                                   uint8_t* tile, float zCentre, int lane) {
    NominalOut o;
    o.x = c.cx;  // cpp:2016-2017
    o.y = c.cy;
    o.z = 0.0f;
    o.row = -1;
    o.col = -1;
    o.valid = 0;
    o.source = 2;
    if (default_check_wave(m, pc, c, tile, lane)) {  // cpp:2012
        o.valid = 1;
        o.source = 0;
        o.row = c.ici;
        o.col = c.icj;
        o.z = zCentre;
        return o;
    }
    fold_polygon(m, pc, c, tile, lane);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    int wi = 0, wj = 0;
    if (candidate_search_wave(m, pc, lut, c, tile, lane, wi, wj)) {  // cpp:2022
        o.valid = 1;
        o.source = 1;
        o.row = wi;
        o.col = wj;
        o.x = cell_pos(m.g.baseX, m.g.res, wi);  // cpp:2105-2107
        o.y = cell_pos(m.g.baseY, m.g.res, wj);
        o.z = zCentre;
    }
    return o;
}

__device__ __forceinline__ void init_ctx(LegCtx& c, const DevMap& m, const PlanConsts& pc, double cx, double cy, float Rf) {
    c.cx = cx;
    c.cy = cy;
    c.Rf = Rf;
    c.R = static_cast<double>(Rf);
    c.R2 = c.R * c.R;
    c.nRings = static_cast<int>(static_cast<unsigned int>(ceil(c.R / m.g.res)));
    c.ici = index_of(cx, m.g.orgX, m.g.posX, m.g.res);
    c.icj = index_of(cy, m.g.orgY, m.g.posY, m.g.res);
    c.ti0 = c.ici - pc.tileH;
    c.tj0 = c.icj - pc.tileH;
}

__device__ __forceinline__ void store_foothold(fpe_foothold* dst, const NominalOut& o, int leg, int cycle) {
    fpe_foothold f;
    f.row = o.row;
    f.col = o.col;
    f.x = o.x;
    f.y = o.y;
    f.z = o.z;
    f.valid = static_cast<uint8_t>(o.valid);
    f.source = static_cast<uint8_t>(o.source);
    f.foot_id = static_cast<uint8_t>(leg);
    f.gait_cycle_id = static_cast<uint8_t>(cycle);
    *dst = f;
}

// getPolygonCenter (cpp:2421-2463): feet[leg][xyz] in LDS.
__device__ __forceinline__ void polygon_center(const double (*feet)[3], double& ox, double& oy, double& oz) {
    double x1 = feet[0][0], y1 = feet[0][1];
    double x2 = feet[1][0], y2 = feet[1][1];
    double sum_x = 0, sum_y = 0, sum_s = 0;
#pragma unroll
    for (int i = 1; i <= 2; i++) {
        const double x3 = feet[i + 1][0], y3 = feet[i + 1][1];  // i=1: LH, i=2: LF
        const double s = ((x2 - x1) * (y3 - y1) - (x3 - x1) * (y2 - y1)) / 2.0;
        sum_x += (x1 + x2 + x3) * s;
        sum_y += (y1 + y2 + y3) * s;
        sum_s += s;
        x2 = x3;
        y2 = y3;
    }
    ox = sum_x / sum_s / 3.0;
    oy = sum_y / sum_s / 3.0;
    oz = (feet[0][2] + feet[1][2] + feet[2][2] + feet[3][2]) / 4.0;
}

// LDS carve for one pose-workgroup (all offsets multiples of 16).
struct PoseShared {
    double cur[3][4][3];   // current feet of the default / centroid / nominal tracks (cpp:1338, 1413, 1480)
    double nxt[3][4][3];   // this phase's results per track
    double polyX[4][8];    // search polygon vertices per leg
    double polyY[4][8];
    int valid[4];
    int pad[4];
};

}  // namespace

// ---- chained plan kernel ------------------------------------------------------------------------
// grid = B poses, block = 256 (wave w = leg w).  Dynamic LDS = sizeof(PoseShared) + 4 * tileW^2.
__global__ __launch_bounds__(256) void plan_chained_kernel(DevMap m, PlanConsts pc, SpiralLut lut,
                                                            const fpe_pose* __restrict__ poses, int B, int nCycles,
                                                            fpe_plan_out out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    PoseShared& sh = *reinterpret_cast<PoseShared*>(smem);
    const int b = blockIdx.x;
    if (b >= B) return;
    const int leg = static_cast<int>(threadIdx.x) >> 6;
    const int lane = lane_id();
    const int tileBytes = (pc.tileW * pc.tileW + 15) & ~15;
    uint8_t* tile = smem + sizeof(PoseShared) + static_cast<size_t>(leg) * tileBytes;

    const fpe_pose pose = poses[b];
    const double x0 = pose.position[0], y0 = pose.position[1], z0 = pose.position[2];
    const int gait = pose.gait;
    float Rf = pose.leg_search_radius[leg];
    if (!(Rf > 0.0f)) Rf = pc.searchRadius;
    const int polyKind = pose.leg_polygon_kind[leg];
    const bool radiusOk = Rf <= pc.maxSearchRadius;

    // initial stance (cpp:350-378) and first-gait shift (setFirstGait, cpp:2679-2699)
    if (threadIdx.x < 4) {
        const int l = threadIdx.x;
        double sx = (l == 0 || l == 3) ? pc.LbHalf : -pc.LbHalf;
        double sy = (l <= 1) ? pc.WbHalfNeg : pc.WbHalfPos;
        double sz = 0;
        sx += x0;
        sy += y0;
        sz += z0;
        if (out.stance) {
            double* st = out.stance + (static_cast<size_t>(b) * 4 + l) * 3;
            st[0] = sx;
            st[1] = sy;
            st[2] = sz;
        }
        for (int t = 0; t < 3; ++t) {
            sh.cur[t][l][0] = sx - pc.stepHalf;
            sh.cur[t][l][1] = sy;
            sh.cur[t][l][2] = sz;
        }
    }
    __syncthreads();

    double adjY = 0.0;  // ajustedPose_[1], cpp:759
    const int nPhases = (gait == 1) ? 4 : 1;
    const double advance = (gait == 1) ? pc.stepQuarter : pc.step;

    for (int g = 0; g < nCycles; ++g) {
        bool cycleOk = true;
        for (int ph = 0; ph < nPhases; ++ph) {
            unsigned mask = 0xFu;
            if (gait == 1) {
                // swing order LF,RH,RF,LH (RF_FIRST=false) or RF,LH,LF,RH (build-defined walk)
                const int order = pc.RF_FIRST ? ((0) | (2 << 2) | (3 << 4) | (1 << 6)) : ((3) | (1 << 2) | (0 << 4) | (2 << 6));
                mask = 1u << ((order >> (2 * ph)) & 3);
            }
            const bool active = (mask >> leg) & 1u;
            if (active) {
                // next default positions of this leg on the three tracks (cpp:2190-2213, 2265-2284)
                double nx[3], ny[3];
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    double cx, cy, cz;
                    polygon_center(sh.cur[t], cx, cy, cz);
                    const double Nx = cx + advance;  // cpp:2199
                    const double Ny = y0 + adjY;     // cpp:2201
                    nx[t] = Nx + pc.biasX[leg];      // cpp:2414-2415
                    ny[t] = Ny + pc.biasY[leg];
                }
                // search polygon from the NOMINAL track (cpp:2235-2244, getSearchPolygon cpp:2496-2517)
                if (lane == 0) {
                    const double r = static_cast<double>(Rf);
                    if (polyKind == 0) {
                        sh.polyX[leg][0] = nx[2] + r;  sh.polyY[leg][0] = ny[2] + 0.5 * r;
                        sh.polyX[leg][1] = nx[2] + r;  sh.polyY[leg][1] = ny[2] - 0.5 * r;
                        sh.polyX[leg][2] = nx[2] - r;  sh.polyY[leg][2] = ny[2] - 0.5 * r;
                        sh.polyX[leg][3] = nx[2] - r;  sh.polyY[leg][3] = ny[2] + 0.5 * r;
                    } else {
                        const double hx = 0.5 * r;
                        const double hy = (0.5 * r) * 0.8660254037844386;
                        sh.polyX[leg][0] = nx[2] + r;   sh.polyY[leg][0] = ny[2];
                        sh.polyX[leg][1] = nx[2] + hx;  sh.polyY[leg][1] = ny[2] - hy;
                        sh.polyX[leg][2] = nx[2] - hx;  sh.polyY[leg][2] = ny[2] - hy;
                        sh.polyX[leg][3] = nx[2] - r;   sh.polyY[leg][3] = ny[2];
                        sh.polyX[leg][4] = nx[2] - hx;  sh.polyY[leg][4] = ny[2] + hy;
                        sh.polyX[leg][5] = nx[2] + hx;  sh.polyY[leg][5] = ny[2] + hy;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();

                LegCtx c;
                init_ctx(c, m, pc, nx[1], ny[1], Rf);  // centre from the CENTROID track (cpp:861-862)
                c.nv = (polyKind == 0) ? 4 : 6;
                c.vx = sh.polyX[leg];
                c.vy = sh.polyY[leg];

                NominalOut no;
                CentroidOut co;
                float zDefault = 0.0f;
                if (radiusOk && !centre_usable(c.cx, c.cy)) {
                    no.row = no.col = -1; no.x = c.cx; no.y = c.cy; no.z = 0.0f; no.valid = 0; no.source = 2;
                    co.x = co.y = 0.0; co.z = 0.0f; co.row = co.col = -1; co.code = 6;
                    if (out.default_next) zDefault = mean_height_wave(m, pc, nx[0], ny[0], lane);
                } else if (radiusOk) {
                    stage_tile(m, pc, c, tile, lane);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    const float zCentre = mean_height_wave(m, pc, c.cx, c.cy, lane);
                    co = centroid_wave(m, pc, c, tile, zCentre, lane);       // cpp:818-821
                    no = nominal_wave(m, pc, lut, c, tile, zCentre, lane);   // cpp:863-903
                    if (out.default_next)                                    // cpp:2289-2301
                        zDefault = mean_height_wave(m, pc, nx[0], ny[0], lane);
                } else {
                    no.row = no.col = -1; no.x = c.cx; no.y = c.cy; no.z = 0.0f; no.valid = 0; no.source = 3;
                    co.x = co.y = 0.0; co.z = 0.0f; co.row = co.col = -1; co.code = 6;
                }
                if (lane == 0) {
                    sh.valid[leg] = no.valid;
                    sh.nxt[0][leg][0] = nx[0]; sh.nxt[0][leg][1] = ny[0]; sh.nxt[0][leg][2] = static_cast<double>(zDefault);
                    sh.nxt[1][leg][0] = co.x;  sh.nxt[1][leg][1] = co.y;  sh.nxt[1][leg][2] = static_cast<double>(co.z);
                    sh.nxt[2][leg][0] = no.x;  sh.nxt[2][leg][1] = no.y;  sh.nxt[2][leg][2] = static_cast<double>(no.z);
                    const size_t o = (static_cast<size_t>(b) * nCycles + g) * 4 + leg;
                    if (out.nominal) store_foothold(out.nominal + o, no, leg, g);
                    if (out.centroid) {
                        fpe_centroid_foothold cf;
                        cf.x = co.x; cf.y = co.y; cf.z = co.z; cf.row = co.row; cf.col = co.col;
                        cf.code = static_cast<uint8_t>(co.code); cf.pad[0] = cf.pad[1] = cf.pad[2] = 0;
                        out.centroid[o] = cf;
                    }
                    if (out.default_next) {
                        out.default_next[o * 3 + 0] = nx[0];
                        out.default_next[o * 3 + 1] = ny[0];
                        out.default_next[o * 3 + 2] = static_cast<double>(zDefault);
                    }
                }
            } else if (lane == 0) {
                sh.valid[leg] = 1;  // non-swing legs do not vote
            }
            __syncthreads();
            // footholdValidation_ = AND of the swing legs' flags (cpp:1323); commit or skip (cpp:1332-1576)
            const bool phaseOk = (sh.valid[0] & sh.valid[1] & sh.valid[2] & sh.valid[3]) != 0;
            __syncthreads();
            if (phaseOk && active && lane < 9) {
                const int t = lane / 3, k = lane - t * 3;
                sh.cur[t][leg][k] = sh.nxt[t][leg][k];
            }
            __syncthreads();
            cycleOk = cycleOk && phaseOk;
        }
        if (threadIdx.x == 0 && out.cycle_ok) out.cycle_ok[static_cast<size_t>(b) * nCycles + g] = cycleOk ? 1 : 0;
        adjY += pc.drift;  // cpp:1578
    }
}

// ---- open-loop per-leg search: one wavefront per checkFoothold call (hpp:94-100) ----------------
struct QueryShared {
    double polyX[4][8];
    double polyY[4][8];
};

__global__ __launch_bounds__(256) void search_legs_kernel(DevMap m, PlanConsts pc, SpiralLut lut,
                                                           const fpe_leg_query* __restrict__ queries, int n,
                                                           fpe_foothold* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    QueryShared& sh = *reinterpret_cast<QueryShared*>(smem);
    const int w = static_cast<int>(threadIdx.x) >> 6;
    const int lane = lane_id();
    const int q = blockIdx.x * 4 + w;
    if (q >= n) return;
    const int tileBytes = (pc.tileW * pc.tileW + 15) & ~15;
    uint8_t* tile = smem + sizeof(QueryShared) + static_cast<size_t>(w) * tileBytes;

    const fpe_leg_query* qp = queries + q;
    const int nv = qp->n_vertices;
    if (lane < 8) {
        sh.polyX[w][lane] = lane < nv ? qp->vx[lane] : 0.0;
        sh.polyY[w][lane] = lane < nv ? qp->vy[lane] : 0.0;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    const float Rf = qp->search_radius;
    LegCtx c;
    init_ctx(c, m, pc, qp->cx, qp->cy, Rf);
    c.nv = nv;
    c.vx = sh.polyX[w];
    c.vy = sh.polyY[w];
    NominalOut no;
    if (!centre_usable(c.cx, c.cy)) {
        no.row = no.col = -1; no.x = c.cx; no.y = c.cy; no.z = 0.0f; no.valid = 0; no.source = 2;
    } else if (Rf <= pc.maxSearchRadius && Rf >= 0.0f && nv >= 0 && nv <= FPE_MAX_POLYGON_VERTICES) {
        stage_tile(m, pc, c, tile, lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const float zCentre = mean_height_wave(m, pc, c.cx, c.cy, lane);
        no = nominal_wave(m, pc, lut, c, tile, zCentre, lane);
    } else {
        no.row = no.col = -1; no.x = c.cx; no.y = c.cy; no.z = 0.0f; no.valid = 0; no.source = 3;
    }
    if (lane == 0) store_foothold(out + q, no, 0, 0);
}

// ---- map ingest: grid_map_msgs layout -> canonical row-major start-index-0 layer ----------------
// src is column-major with circular-buffer start index (si, sj): unwrapped (i, j) lives at buffer
// index ((i + si) % rows, (j + sj) % cols) (grid_map getBufferIndexFromIndex).  32x32 tiles through
// LDS so that both the column-major reads and the row-major writes are coalesced.
__global__ __launch_bounds__(256) void canonicalise_layer_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                                  int rows, int cols, int si, int sj, int srcRowMajor) {
    __shared__ float t[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int iBase = blockIdx.y * 32, jBase = blockIdx.x * 32;
    if (srcRowMajor) {
        for (int r = ty; r < 32; r += 8) {
            const int i = iBase + r, j = jBase + tx;
            if (i < rows && j < cols) {
                int bi = i + si; if (bi >= rows) bi -= rows;
                int bj = j + sj; if (bj >= cols) bj -= cols;
                dst[static_cast<size_t>(i) * cols + j] = src[static_cast<size_t>(bi) * cols + bj];
            }
        }
        return;
    }
    // column-major source: consecutive threads walk i (contiguous in src)
    for (int c = ty; c < 32; c += 8) {
        const int i = iBase + tx, j = jBase + c;
        if (i < rows && j < cols) {
            int bi = i + si; if (bi >= rows) bi -= rows;
            int bj = j + sj; if (bj >= cols) bj -= cols;
            t[c][tx] = src[static_cast<size_t>(bi) + static_cast<size_t>(bj) * rows];
        }
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int i = iBase + r, j = jBase + tx;
        if (i < rows && j < cols) dst[static_cast<size_t>(i) * cols + j] = t[tx][r];
    }
}

// ---- launch wrappers (called from fpe_engine.cpp) -------------------------------------------------
size_t plan_lds_bytes(const PlanConsts& pc) {
    return sizeof(PoseShared) + 4 * static_cast<size_t>((pc.tileW * pc.tileW + 15) & ~15);
}
size_t search_lds_bytes(const PlanConsts& pc) {
    return sizeof(QueryShared) + 4 * static_cast<size_t>((pc.tileW * pc.tileW + 15) & ~15);
}

hipError_t launch_plan_chained(const DevMap& m, const PlanConsts& pc, const SpiralLut& lut, const fpe_pose* d_poses,
                               int B, int nCycles, const fpe_plan_out& d_out, hipStream_t stream) {
    const size_t lds = plan_lds_bytes(pc);
    hipLaunchKernelGGL(plan_chained_kernel, dim3(B), dim3(256), lds, stream, m, pc, lut, d_poses, B, nCycles, d_out);
    return hipGetLastError();
}

hipError_t launch_search_legs(const DevMap& m, const PlanConsts& pc, const SpiralLut& lut, const fpe_leg_query* d_q,
                              int n, fpe_foothold* d_out, hipStream_t stream) {
    const size_t lds = search_lds_bytes(pc);
    hipLaunchKernelGGL(search_legs_kernel, dim3((n + 3) / 4), dim3(256), lds, stream, m, pc, lut, d_q, n, d_out);
    return hipGetLastError();
}

hipError_t launch_canonicalise(const float* d_src, float* d_dst, int rows, int cols, int si, int sj, int srcRowMajor,
                               hipStream_t stream) {
    dim3 grid((cols + 31) / 32, (rows + 31) / 32);
    hipLaunchKernelGGL(canonicalise_layer_kernel, grid, dim3(256), 0, stream, d_src, d_dst, rows, cols, si, sj, srcRowMajor);
    return hipGetLastError();
}

hipError_t set_max_lds(size_t planBytes, size_t searchBytes) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(plan_chained_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(planBytes));
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(search_legs_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(searchBytes));
}

}  // namespace fpe
