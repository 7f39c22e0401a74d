// fpe_kernels.hip — hand-written HIP for gfx950 (MI355X, wave64): the foothold-search hot path.
//
// Work decomposition.  A body pose is one chain of gait cycles (globalFootholdPlan, cpp:762-1579);
// inside a cycle the four legs are independent (the reference's 4 std::thread(checkFoothold),
// cpp:863-909).  A leg is searched by a GROUP of G lanes (template parameter):
//   G = 8 : two poses per wavefront, lanes 8*l..8*l+7 of a pose's half = leg l; 64-thread workgroups, no
//           block barriers — the default for windows of <= 1024 cells (2 cm maps, 1 cm maps at R 0.1);
//   G = 64: plan_sequential_kernel, one wavefront per pose, the swing legs of a phase searched one after the
//           other with all 64 lanes — large windows (1 cm at R 0.15, 0.5 cm maps);
//   G = 4 / 16 and the four-wavefronts-per-pose form of G = 64 exist for measurement (fpe_set_tuning "plan_group").
// Direct kernels (this file's first part) read the f32 layers themselves: the default disc and the centroid
// rectangle straight through L1/L2, the spiral window of large foot discs staged into LDS as flag bytes.
// Bit-window kernels (second part, fpe_bits.hpp) read per-snapshot bit planes instead — one 32/64/96-bit row mask
// per window row and lane — and touch the f32 elevation layer only for the mean heights.
// Spiral candidates: lane = SpiralIterator rank, lowest set ballot bit = argmin of rank.
// No MFMA: nothing here is a contraction.
//
// All geometry is f64 in the reference's expression order (fpe_gridmath.hpp); compile with
// -ffp-contract=off.  Reference citations: "cpp:" = foothold_planner/src/FootholdPlanner.cpp.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <type_traits>

#include "fpe_device.hpp"

namespace fpe {

namespace {

// profiling-only timeline stamps (pc.trace != null): slot = ((block * 8 + cycle) * 16 + point)
// compiled in only with -DFPE_TRACE (scratch/trace.py); a no-op in the shipped library
__device__ __forceinline__ void stamp(const PlanConsts& pc, int cyc, int point) {
#ifdef FPE_TRACE
    if (pc.trace && blockIdx.x < 256 && cyc < 8 && threadIdx.x == 0)
        pc.trace[(static_cast<size_t>(blockIdx.x) * 8 + cyc) * 16 + point] = __builtin_readcyclecounter();
#ifdef FPE_TRACE_ALL_BLOCKS  // (start, end, hardware id) of EVERY block behind the 256 x 8 x 16 stage table: residency studies
    if (pc.trace && cyc == 6 && point >= 14 && threadIdx.x == 0 && blockIdx.x < 65536) {
        pc.trace[256 * 8 * 16 + static_cast<size_t>(blockIdx.x) * 4 + (point - 14)] = __builtin_readcyclecounter();
        if (point == 14) {
            unsigned hw;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            pc.trace[256 * 8 * 16 + static_cast<size_t>(blockIdx.x) * 4 + 2] = hw;
            pc.trace[256 * 8 * 16 + static_cast<size_t>(blockIdx.x) * 4 + 3] = xcc;
        }
    }
#endif
#else
    (void)pc;
    (void)cyc;
    (void)point;
#endif
}

// the same from whichever lane is the first active one (inside lane-divergent regions)
__device__ __forceinline__ void stamp_any(const PlanConsts& pc, int cyc, int point) {
#ifdef FPE_TRACE
    const int first = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x));
    if (pc.trace && blockIdx.x < 256 && cyc < 8 && static_cast<int>(threadIdx.x) == first)
        pc.trace[(static_cast<size_t>(blockIdx.x) * 8 + cyc) * 16 + point] = __builtin_readcyclecounter();
#else
    (void)pc;
    (void)cyc;
    (void)point;
#endif
}

__device__ __forceinline__ void stamp_value(const PlanConsts& pc, int cyc, int point, long long v) {
#ifdef FPE_TRACE
    if (pc.trace && blockIdx.x < 256 && cyc < 8 && threadIdx.x == 0)
        pc.trace[(static_cast<size_t>(blockIdx.x) * 8 + cyc) * 16 + point] = v;
#else
    (void)pc;
    (void)cyc;
    (void)point;
    (void)v;
#endif
}

// Keep a (wave-uniform) value in a vector register: the empty asm hides its uniformity from the compiler.
__device__ __forceinline__ double in_vgpr(double x) {
    asm volatile("" : "+v"(x));
    return x;
}

// A search centre must be finite and of sane magnitude.  The reference has no such test: a
// non-finite centre (reachable once the centroid track has committed its "no case" (0,0,0)
// results, cpp:1777-1944, and the feet polygon degenerates) sends NaN through
// getIndexFromPosition's (int) cast, which is undefined behaviour.  Engine and oracle both define
// it as "no cell is visited": invalid leg, getSubmap failure, mean height = h.
__device__ __forceinline__ bool centre_usable(double x, double y) { return fabs(x) <= 1e6 && fabs(y) <= 1e6; }

// ---- lane group ------------------------------------------------------------------------------------
template <int G>
struct Grp {
    int sub;    // lane index inside the group, 0..G-1
    int gbase;  // first lane of the group inside its wavefront
    __device__ __forceinline__ explicit Grp(int tid) : sub(tid & (G - 1)), gbase((tid & 63) & ~(G - 1)) {}
    __device__ __forceinline__ unsigned long long ballot(bool p) const {
        const unsigned long long m = __ballot(p);
        if (G == 64) return m;
        return (m >> gbase) & ((1ull << (G & 63)) - 1ull);
    }
    __device__ __forceinline__ bool any(bool p) const { return ballot(p) != 0ull; }
    // Value of lane l of the group.  Every caller derives l from a group ballot (or passes a constant), so l is uniform
    // over the group; with one group per wavefront that is a v_readlane (no LDS round trip) instead of a ds_bpermute.
    // Value of lane L (a compile-time constant) of the group: v_readlane with one group per wavefront, else a ds_swizzle in
    // bit mode (no address register, no index arithmetic: a ds_bpermute needs both)
    template <int L>
    __device__ __forceinline__ int bcast_c(int v) const {
        if constexpr (G == 64) return __builtin_amdgcn_readlane(v, L);
        else return __builtin_amdgcn_ds_swizzle(v, ((~(G - 1)) & 0x1F) | (L << 5));
    }
    template <class T>
    __device__ __forceinline__ T bcast(T v, int l) const {
        if constexpr (G == 64 && sizeof(T) == 4) {
            const int r = __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), __builtin_amdgcn_readfirstlane(l));
            return __builtin_bit_cast(T, r);
        } else {
            return __shfl(v, gbase + l);
        }
    }
};

// t / d and t % d for 0 <= t < 2^22, 1 <= d < 2^22 without the 32-bit udiv sequence: one f32
// reciprocal estimate and one exact integer fix-up step (the estimate is off by at most 1).
__device__ __forceinline__ void divmod_small(int t, int d, float dinv, int& q, int& r) {
    q = static_cast<int>(static_cast<float>(t) * dinv);
    r = t - q * d;
    if (r < 0) {
        q -= 1;
        r += d;
    } else if (r >= d) {
        q += 1;
        r -= d;
    }
}
// v_rcp_f32 (1 ulp) is enough: with t < 2^22 the estimate t * dinv is within 1 of t / d, which the fix-up absorbs
__device__ __forceinline__ float rcp_small(int d) { return __builtin_amdgcn_rcpf(static_cast<float>(d)); }

// Pose-level synchronisation: a pose owns one wavefront when G < 64 (LDS operations of a wavefront
// are executed in order, so only the compiler must not reorder) and four wavefronts when G = 64.
template <int G>
__device__ __forceinline__ void pose_sync() {
    if (G == 64) {
        __syncthreads();
    } else {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- lane-transposed corner arithmetic -------------------------------------------------------------------
// The f64 "scalar" geometry of a leg is dominated by boundPositionToRange + getIndexFromPosition of
// box corners: CircleIterator::findSubmapParameters (2 corners x 2 axes per disc) and
// getSubmapInformation (same for the centroid rectangle).  Instead of every lane repeating all of
// them, lane s of a group evaluates ONE quantity — box (s >> 2) & 3, corner/axis s & 3
// (0 top-left x, 1 top-left y, 2 bottom-right x, 3 bottom-right y) — and the integer results are
// gathered with cross-lane reads.  Same functions, same operands: bit-identical to the scalar form.
struct Box {
    double cx, cy, hx, hy;  // centre and half extents: corner = centre +- half extent
};
struct CornerVal {
    int idx;      // getIndexFromPosition of the bounded corner on this lane's axis
    bool within;  // checkIfPositionWithinMap of the bounded corner on this lane's axis
};
// Quantity k (0 top-left x, 1 top-left y, 2 bottom-right x, 3 bottom-right y) of one box.  raw: the box is a
// plain getIndexFromPosition of its centre (no boundPositionToRange), e.g. getIndex(search centre).
__device__ __forceinline__ CornerVal corner_quantity(const MapGeom& g, int k, const Box& b, bool raw) {
    const bool isY = (k & 1) != 0, isBR = (k & 2) != 0;
    const double c = isY ? b.cy : b.cx, h = isY ? b.hy : b.hx;
    const double org = isY ? g.orgY : g.orgX, pos = isY ? g.posY : g.posX, len = isY ? g.lenY : g.lenX;
    const double v = isBR ? c - h : c + h;
    const double bnd = raw ? c : bound_axis(v, org, pos, len);
    CornerVal r;
    // index_of_fast's test with a wave-uniform fallback: when any lane's predicted quotient is within rounding
    // distance of an integer the wavefront takes the true division (hipcc if-converts index_of_fast's own rare branch,
    // i.e. every call pays the division sequence)
    {
        const double n = (bnd - org) - pos;
        const double qf = n * g.rinv;
        double k2 = trunc(qf);
        const double fr = fabs(qf - k2);
        const double eps = fabs(qf) * 4.5e-16 + 1e-290;
        if (__ballot(!(fr > eps && fr < 1.0 - eps)) != 0ull) k2 = trunc(n / g.res);
        r.idx = -static_cast<int>(k2);
    }
    r.within = raw || within_axis(bnd, org, pos, len);
    return r;
}
__device__ __forceinline__ Box pick_box(bool second, const Box& a, const Box& b) {
    Box r;
    r.cx = second ? b.cx : a.cx;
    r.cy = second ? b.cy : a.cy;
    r.hx = second ? b.hx : a.hx;
    r.hy = second ? b.hy : a.hy;
    return r;
}

// NQ quantities (8 or 16) evaluated min(G,16) at a time by the lanes of a group.
template <int G, int NQ>
struct Corners {
    static constexpr int L = (G < 16) ? G : 16;          // lanes used per pass
    static constexpr int P = (NQ + L - 1) / L;           // passes
    int idx[P];
    unsigned within;                                      // bit id = within flag of quantity id

    // Quantity id = box (id >> 2) & 3, corner/axis id & 3; a box whose bit is set in rawBoxMask is raw.  Pass p
    // covers the L/4 boxes starting at p*L/4, so a lane chooses among 1, 2 or 4 boxes — never more than it must.
    __device__ __forceinline__ void eval(const MapGeom& mg, const Grp<G>& g, const Box& b0, const Box& b1, const Box& b2,
                                         const Box& b3, unsigned rawBoxMask) {
        within = 0u;
        const int lane = g.sub & (L - 1);
#pragma unroll
        for (int p = 0; p < P; ++p) {
            Box b;
            bool raw;
            if constexpr (L == 4) {
                const int q = p & 3;
                b = q == 0 ? b0 : (q == 1 ? b1 : (q == 2 ? b2 : b3));
                raw = ((rawBoxMask >> q) & 1u) != 0;
            } else if constexpr (L == 8) {
                const bool second = (lane & 4) != 0;
                const bool upper = ((2 * p) & 3) >= 2;
                b = upper ? pick_box(second, b2, b3) : pick_box(second, b0, b1);
                const unsigned two = (rawBoxMask >> ((2 * p) & 3)) & 3u;
                raw = second ? (two & 2u) != 0 : (two & 1u) != 0;
            } else {
                const int q = (lane >> 2) & 3;
                b = pick_box((q & 2) != 0, pick_box((q & 1) != 0, b0, b1), pick_box((q & 1) != 0, b2, b3));
                raw = ((rawBoxMask >> q) & 1u) != 0;
            }
            const CornerVal cv = corner_quantity(mg, lane & 3, b, raw);
            idx[p] = cv.idx;
            const unsigned bits = static_cast<unsigned>(g.ballot(cv.within)) & ((1u << L) - 1u);
            within |= bits << (p * L);
        }
    }
    // Quantity ID as seen by every lane of the group.  Groups of <= 32 lanes read it with ds_swizzle (bit mode:
    // keep the group bits of the lane id, force the low bits to ID % L): no LDS memory and no address VALU.
    template <int ID>
    __device__ __forceinline__ int get(const Grp<G>& g) const {
        if constexpr (G <= 32) {
            constexpr int pattern = ((~(G - 1)) & 0x1F) | ((ID % L) << 5);
            return __builtin_amdgcn_ds_swizzle(idx[ID / L], pattern);
        } else if constexpr (G == 64) {
            return __builtin_amdgcn_readlane(idx[ID / L], ID % L);  // one group per wavefront: a fixed lane, a scalar result
        } else {
            return g.bcast(idx[ID / L], ID % L);
        }
    }
    template <int Q>
    __device__ __forceinline__ BBox bbox(const Grp<G>& g) const {
        BBox b;
        b.i0 = get<4 * Q + 0>(g);
        b.j0 = get<4 * Q + 1>(g);
        b.ni = get<4 * Q + 2>(g) - b.i0 + 1;
        b.nj = get<4 * Q + 3>(g) - b.j0 + 1;
        return b;
    }
    __device__ __forceinline__ bool box_within(int q) const { return ((within >> (4 * q)) & 0xFu) == 0xFu; }
};

// Tail of getSubmapInformation once the four corner indices / within flags are known.
__device__ __forceinline__ Submap submap_from_corners(const MapGeom& g, const BBox& bb, bool allWithin, double px, double py) {
    Submap s;
    s.ok = false;
    s.i0 = bb.i0;
    s.j0 = bb.j0;
    s.ni = bb.ni;
    s.nj = bb.nj;
    s.baseX = s.baseY = 0.0;
    if (!allWithin) return s;
    if (!in_range(s.i0, s.j0, g.rows, g.cols)) return s;  // getPositionFromIndex(topLeft) range check
    // getBufferRegionsForSubmap: the region must fit the buffer — a corner bounded onto the far edge can round to the
    // index `size` (oracle/fpo_gridmap.hpp::getSubmap)
    if (s.i0 + s.ni > g.rows || s.j0 + s.nj > g.cols) return s;
    const double cornerX = cell_pos(g.baseX, g.res, s.i0) - (-(0.5 * g.res));
    const double cornerY = cell_pos(g.baseY, g.res, s.j0) - (-(0.5 * g.res));
    const double subLenX = static_cast<double>(s.ni) * g.res;
    const double subLenY = static_cast<double>(s.nj) * g.res;
    const double subOrgX = 0.5 * subLenX, subOrgY = 0.5 * subLenY;
    const double subPosX = cornerX - subOrgX, subPosY = cornerY - subOrgY;
    if (!(within_axis(px, subOrgX, subPosX, subLenX) && within_axis(py, subOrgY, subPosY, subLenY))) return s;
    s.baseX = subPosX + (subOrgX - 0.5 * g.res);
    s.baseY = subPosY + (subOrgY - 0.5 * g.res);
    s.ok = true;
    return s;
}

// ---- per-leg search context (group-uniform) -----------------------------------------------------------
struct LegCtx {
    double cx, cy;   // search centre = centroid-track next position of this leg (cpp:861-862)
    double R2;       // double(float searchRadius)^2 (SpiralIterator radiusSquare_)
    int nRings;      // ceil(R / res)
    int nCand;       // spiral table entries of rings 0..nRings
    int cyc;         // gait cycle (profiling stamps only)
    int ici, icj;    // getIndex(centre)
    int ti0, tj0;    // tile origin (cell index of tile[0])
    int nv;          // polygon vertex count
    const double* vx;  // polygon vertices (LDS)
    const double* vy;
    const int8_t* footDa;  // foot-disc offset table (LDS copy of PlanConsts::footDa/footDb)
    const int8_t* footDb;
    const int16_t* footOff;  // da * tileW + db of every table entry (LDS)
    bool rect;         // polygon is the reference rectangle: xlo/xhi/ylo/yhi hold its sides
    double xlo, xhi, ylo, yhi;
};

__device__ __forceinline__ uint8_t tile_at(const uint8_t* tile, int W, const LegCtx& c, int i, int j) {
    const int a = i - c.ti0, b = j - c.tj0;
    // cells a search can touch are inside the tile by construction (tileH); anything else reads as
    // "outside the map" instead of out of bounds
    if (static_cast<unsigned>(a) >= static_cast<unsigned>(W) || static_cast<unsigned>(b) >= static_cast<unsigned>(W))
        return 0;
    return tile[a * W + b];
}

// ---- LDS staging of the spiral window (large foot discs) ---------------------------------------------------
// The window around the search centre is staged into LDS as one flag byte per cell, with the
// per-cell verdict of checkCirclePolygonFoothold (cpp:2132-2138) folded in: a FINITE cell fails when
// it is below the candidate threshold or its centre is outside the search polygon; non-finite cells
// never fail.  Staging is lazy by square annuli: the spiral visits rings outward and usually stops
// after a few, so only the cells within reach of the rings examined so far are ever read from HBM.
__device__ __forceinline__ bool cell_outside_polygon(const LegCtx& c, double px, double py) {
    if (c.rect) {
        // the reference's rectangle (getSearchPolygon, cpp:2496-2517; vertices LU,RU,RD,LD): PNPOLY
        // reduces to these six comparisons on the same values — only the two vertical edges can
        // straddle py, and their intersection abscissae are exactly xhi and xlo
        const bool straddle = (c.ylo > py) != (c.yhi > py);
        return !(straddle && ((px < c.xhi) != (px < c.xlo)));
    }
    return !polygon_inside_fast(c.vx, c.vy, c.nv, px, py);
}

// LDS of one spiral window: tileW^2 flag bytes (rounded to 16), then two doubles per tile column — the
// abscissae at which the search polygon's boundary crosses that column (column_crossings).
__host__ __device__ __forceinline__ int tile_flag_bytes(const PlanConsts& pc) { return (pc.tileW * pc.tileW + 15) & ~15; }
__host__ __device__ __forceinline__ int tile_total_bytes(const PlanConsts& pc) { return tile_flag_bytes(pc) + pc.tileW * 16; }

// PNPOLY (Polygon::isInside, cpp:2138) counts, for a cell centre (px, py), the edges that straddle py and
// whose intersection abscissa xi = (vx[j]-vx[i])*(py-vy[i])/(vy[j]-vy[i]) + vx[i] lies beyond px.  py and
// therefore every xi depend on the tile COLUMN only, so they are evaluated once per column (same
// expressions, same operands) instead of once per cell: with at most two crossings per column — any convex
// polygon, in particular the reference rectangle and the hexagon — a cell is inside iff
// (px < X0) != (px < X1), a missing crossing being -inf.  Returns false when some column has more than two
// crossings (non-convex open-loop polygons): the caller then keeps the per-cell evaluation.
template <int G>
__device__ bool column_crossings(const DevMap& m, const PlanConsts& pc, const LegCtx& c, const Grp<G>& g, double* colX) {
    const double ninf = -__builtin_huge_val();
    bool over = false;
    for (int b = g.sub; b < pc.tileW; b += G) {
        const double py = cell_pos(m.g.baseY, m.g.res, c.tj0 + b);
        double X0 = ninf, X1 = ninf;
        if (c.rect) {
            // getSearchPolygon's rectangle (cpp:2496-2517): only the two vertical edges can straddle, and their
            // intersection abscissae are exactly xhi and xlo (see cell_outside_polygon)
            if ((c.ylo > py) != (c.yhi > py)) {
                X0 = c.xhi;
                X1 = c.xlo;
            }
        } else {
            int n = 0;
            for (int i = 0, j = c.nv - 1; i < c.nv; j = i++) {
                if ((c.vy[i] > py) != (c.vy[j] > py)) {
                    const double ex = c.vx[j] - c.vx[i];
                    const double t = py - c.vy[i];
                    double xi = c.vx[i];
                    if (!(ex == 0.0 && fabs(t) <= DBL_MAX)) xi = ex * t / (c.vy[j] - c.vy[i]) + c.vx[i];  // polygon_inside_fast
                    if (n == 0) X0 = xi;
                    else if (n == 1) X1 = xi;
                    ++n;
                }
            }
            over |= n > 2;
        }
        colX[2 * b] = X0;
        colX[2 * b + 1] = X1;
    }
    return !g.any(over);
}

// Stage the cells whose Chebyshev distance d from the tile centre satisfies lo < d <= hi (lo = -1 stages
// the centre block too).  The square annulus is ONE index space — top band, bottom band, then the left and
// right bands row by row — walked by all lanes with 2-4 independent loads in flight per lane:
// the staging of a large window is a chain of cache round trips, so what matters is how many of them
// overlap, not the arithmetic (cfg-5: 20 k of the 25 k clocks of a spiral search were staging waits).
template <int G>
__device__ void stage_annulus(const DevMap& m, const PlanConsts& pc, const LegCtx& c, uint8_t* tile, const Grp<G>& g,
                              int lo, int hi, bool useColX) {
    const double* colX = reinterpret_cast<const double*>(tile + tile_flag_bytes(pc));
    constexpr int kStageUnroll = 4;  // loads in flight per lane
    const int H = pc.tileH;
    if (hi > H) hi = H;
    if (hi <= lo) return;
    const int W = pc.tileW;
    const int o0 = H - hi, Wo = 2 * hi + 1;             // outer square [o0, o0 + Wo)
    const bool whole = lo < 0;
    const int i0 = whole ? o0 : H - lo;                 // inner square [i0, i0 + Wi) is already staged
    const int Wi = whole ? 0 : 2 * lo + 1;
    const int band = i0 - o0;                           // thickness of the annulus
    const int nTop = whole ? Wo * Wo : band * Wo;       // whole square: everything is "top band"
    const int nBot = whole ? 0 : band * Wo;
    const int sideW = 2 * band;
    const int nSide = whole ? 0 : Wi * sideW;
    const int N = nTop + nBot + nSide;
    const float woInv = rcp_small(Wo), swInv = rcp_small(sideW > 0 ? sideW : 1);
    for (int base = 0; base < N; base += G * kStageUnroll) {
        int a[kStageUnroll], b[kStageUnroll];
        float v[kStageUnroll];
        bool live[kStageUnroll], inMap[kStageUnroll];
#pragma unroll
        for (int u = 0; u < kStageUnroll; ++u) {
            const int t = base + u * G + g.sub;
            live[u] = t < N;
            inMap[u] = false;
            v[u] = 0.0f;
            a[u] = b[u] = 0;
            if (live[u]) {
                int qa, qb;
                if (t < nTop + nBot) {
                    const bool bottom = t >= nTop;
                    divmod_small(bottom ? t - nTop : t, Wo, woInv, qa, qb);
                    a[u] = (bottom ? i0 + Wi : o0) + qa;
                    b[u] = o0 + qb;
                } else {
                    divmod_small(t - nTop - nBot, sideW, swInv, qa, qb);
                    a[u] = i0 + qa;
                    b[u] = qb < band ? o0 + qb : i0 + Wi + (qb - band);
                }
                const int ci = c.ti0 + a[u], cj = c.tj0 + b[u];
                inMap[u] = in_range(ci, cj, m.g.rows, m.g.cols);
                if (inMap[u]) v[u] = m.trav[static_cast<size_t>(ci) * m.g.cols + cj];
            }
        }
#pragma unroll
        for (int u = 0; u < kStageUnroll; ++u) {
            if (!live[u]) continue;
            uint8_t f = 0;
            if (inMap[u]) {
                f = kFlagInMap;
                if (v[u] < pc.thrDefault) f |= kFlagBelowDef;
                if (v[u] < pc.thrCandidate) f |= kFlagBelowCand;
                if (__builtin_isfinite(v[u])) {
                    f |= kFlagFinite;
                    bool fail = (f & kFlagBelowCand) != 0;
                    if (!fail) {
                        const double px = cell_pos(m.g.baseX, m.g.res, c.ti0 + a[u]);
                        if (useColX) {
                            const double2 X = *reinterpret_cast<const double2*>(colX + 2 * b[u]);
                            fail = !((px < X.x) != (px < X.y));
                        } else {
                            fail = cell_outside_polygon(c, px, cell_pos(m.g.baseY, m.g.res, c.tj0 + b[u]));
                        }
                    }
                    if (fail) f |= kFlagFail;
                }
            }
            tile[a[u] * W + b[u]] = f;
        }
    }
}

// In-order f32 accumulation of getFootholdMeanHeight (cpp:2539-2545) over the visited lanes of one
// round: lanes are cells in the CircleIterator's row-major order.
template <int G>
__device__ __forceinline__ void accumulate_heights(const Grp<G>& g, bool vis, float v, float& sum, float& last, int& cnt) {
    unsigned long long mask = g.ballot(vis);
    while (mask) {
        const int l = __builtin_ctzll(mask);
        mask &= mask - 1;
        const float e = g.bcast(v, l);
        last = e;
        if (e < 10) {  // cpp:2539
            cnt++;
            sum = sum + e;
        }
    }
}

// The visited values of a round are compacted into LDS scratch in lane order (= CircleIterator
// order) and summed afterwards by reading them back four at a time — the f32 additions stay
// strictly sequential, only the cross-lane traffic is batched (3 % faster than a ballot/shuffle loop
// even for the 1-4 cells of a 2 cm foot disc; 5x for the 45-cell discs of a 0.5 cm map).
struct OrderedSum {
    float* scratch;  // >= (cells of the disc bounding box) floats, 16-byte aligned; null: ballot loop
    int n;
};
template <int G>
__device__ __forceinline__ void ordered_push(const Grp<G>& g, OrderedSum& os, bool vis, float v) {
    const unsigned long long mask = g.ballot(vis);
    const int rank = __builtin_popcountll(mask & ((1ull << g.sub) - 1ull));
    if (vis) os.scratch[os.n + rank] = v;
    os.n += __builtin_popcountll(mask);
}
__device__ __forceinline__ void ordered_step(float e, float& sum, float& last, int& cnt) {
    last = e;
    if (e < 10) {  // cpp:2539
        cnt++;
        sum = sum + e;
    }
}
__device__ __forceinline__ void ordered_finish(const OrderedSum& os, float& sum, float& last, int& cnt) {
    int t = 0;
    for (; t + 16 <= os.n; t += 16) {  // large discs: four independent 16-byte reads, then 16 dependent adds
        float4 q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) q[u] = *reinterpret_cast<const float4*>(os.scratch + t + 4 * u);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            ordered_step(q[u].x, sum, last, cnt);
            ordered_step(q[u].y, sum, last, cnt);
            ordered_step(q[u].z, sum, last, cnt);
            ordered_step(q[u].w, sum, last, cnt);
        }
    }
    for (; t + 4 <= os.n; t += 4) {
        const float4 q = *reinterpret_cast<const float4*>(os.scratch + t);
        ordered_step(q.x, sum, last, cnt);
        ordered_step(q.y, sum, last, cnt);
        ordered_step(q.z, sum, last, cnt);
        ordered_step(q.w, sum, last, cnt);
    }
    for (; t < os.n; ++t) ordered_step(os.scratch[t], sum, last, cnt);
}

// Sequential f32 sum of one value per lane over the lanes of a group, in lane order, computed
// redundantly by every lane: lane k's value arrives by ds_swizzle (bit mode: keep the group bits of
// the lane id, force the low bits to k — no LDS memory, no VALU), so a round costs G dependent
// v_add_f32 and nothing else.  Lanes that must not contribute pass -0.0f: s + (-0.0f) == s bit for
// bit for every s (round-to-nearest), which keeps the reference's add order over the visited cells.
template <int G, int K>
__device__ __forceinline__ float grp_lane_value(float v) {
    static_assert(G <= 32, "ds_swizzle bit mode permutes within 32 lanes");
    constexpr int pattern = ((~(G - 1)) & 0x1F) | (K << 5);
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), pattern));
}
template <int G, int K = 0, int End = G>  // lanes K .. End-1 of the group, in order
struct SeqSum {
    static __device__ __forceinline__ float run(float s, float x) {
        s = s + grp_lane_value<G, K>(x);
        return SeqSum<G, K + 1, End>::run(s, x);
    }
};
template <int G, int End>
struct SeqSum<G, End, End> {
    static __device__ __forceinline__ float run(float s, float) { return s; }
};

__device__ __forceinline__ float finish_mean(float sum, float last, int cnt, double h) {
    const float mean = (cnt != 0) ? (sum / cnt) : last;       // cpp:2547-2551
    return static_cast<float>(static_cast<double>(mean) + h);  // cpp:2553 (float + double)
}

// A finite cell fails checkCirclePolygonFoothold's test (cpp:2132-2138) when it is below the
// candidate threshold or its centre is outside the search polygon; non-finite cells never fail.
__device__ __forceinline__ bool cell_fails_direct(const DevMap& m, const PlanConsts& pc, const LegCtx& c, int i, int j) {
    const float v = m.trav[static_cast<size_t>(i) * m.g.cols + j];
    if (!__builtin_isfinite(v)) return false;
    if (v < pc.thrCandidate) return true;
    return cell_outside_polygon(c, cell_pos(m.g.baseX, m.g.res, i), cell_pos(m.g.baseY, m.g.res, j));
}

// checkCirclePolygonFoothold (cpp:2117-2163) for the cell-centred disc of candidate (i, j), one
// lane per candidate.  kTile: per-cell verdicts come from the LDS tile (stage_annulus), else
// they are evaluated on demand from the map (tiny discs: fewer cells than the tile has).
// Disc membership: the host-proved offset table when pc.footRobust, else the literal
// CircleIterator bounding-box walk in f64.
template <bool kTile>
__device__ __forceinline__ bool candidate_disc_ok(const DevMap& m, const PlanConsts& pc, const LegCtx& c,
                                                  const uint8_t* tile, int i, int j) {
    if (pc.footRobust) {
        if (pc.nFoot == 1) {  // the disc is the candidate's own cell (e.g. rf 0.02 on a 2 cm map)
            return kTile ? (tile_at(tile, pc.tileW, c, i, j) & kFlagFail) == 0 : !cell_fails_direct(m, pc, c, i, j);
        }
        if (kTile) {
            // every cell a disc can touch has been staged (stage_annulus keeps ring + footReach ahead of the
            // candidates; cells outside the map carry flag 0), so the disc is a branch-free OR over the
            // precomputed tile displacements of the offset table, eight independent LDS reads per trip
            const uint8_t* p = tile + (i - c.ti0) * pc.tileW + (j - c.tj0);
            unsigned acc = 0;
            int k = 0;
            for (; k + 8 <= pc.nFoot; k += 8) {
#pragma unroll
                for (int u = 0; u < 8; ++u) acc |= p[c.footOff[k + u]];
                if (acc & kFlagFail) return false;  // early exit at chunk granularity: a round whose candidates all
                                                    // fail stays short, a clean disc still reads 8 cells per trip
            }
            for (; k < pc.nFoot; ++k) acc |= p[c.footOff[k]];
            return (acc & kFlagFail) == 0;  // the candidate's own cell (offset 0,0) is always visited
        }
        for (int k = 0; k < pc.nFoot; ++k) {
            const int qi = i + c.footDa[k], qj = j + c.footDb[k];
            if (!in_range(qi, qj, m.g.rows, m.g.cols)) continue;
            if (cell_fails_direct(m, pc, c, qi, qj)) return false;
        }
        return true;  // the candidate's own cell (offset 0,0) is always visited
    }
    const double fx = cell_pos(m.g.baseX, m.g.res, i);
    const double fy = cell_pos(m.g.baseY, m.g.res, j);
    const BBox bb = circle_bbox_fast(m.g, fx, fy, pc.rf);
    bool any = false;
    for (int a = 0; a < bb.ni; ++a) {
        for (int b = 0; b < bb.nj; ++b) {
            const int qi = bb.i0 + a, qj = bb.j0 + b;
            if (in_range(qi, qj, m.g.rows, m.g.cols) && cell_in_disc(m.g, qi, qj, fx, fy, pc.rf2)) {
                const bool fail = kTile ? (tile_at(tile, pc.tileW, c, qi, qj) & kFlagFail) != 0 : cell_fails_direct(m, pc, c, qi, qj);
                if (fail) return false;
                any = true;
            }
        }
    }
    return any;
}

// checkCandidateFoothold (cpp:2085-2114): first valid cell in SpiralIterator order.  Lane k of a
// round evaluates the candidate of rank base+k; the lowest set ballot bit is the argmin of rank.
// The first kLutHeadRounds*G entries of the spiral rank table, held in registers for the whole
// chain (they are the same for every leg and cycle): the common spiral search ends within them, so
// it never waits on a table load.
constexpr int kLutHeadRounds = 2;
struct LutHead {
    int dij[kLutHeadRounds];   // di | dj << 16 of entry r*G + sub
    int ring[kLutHeadRounds];
};
template <int G>
__device__ __forceinline__ LutHead load_lut_head(const SpiralLut& lut, const Grp<G>& g) {
    LutHead h;
    const int total = lut.total;
#pragma unroll
    for (int r = 0; r < kLutHeadRounds; ++r) {
        const int k = min(r * G + g.sub, total - 1);
        h.dij[r] = (static_cast<int>(lut.di[k]) & 0xFFFF) | (static_cast<int>(lut.dj[k]) << 16);
        h.ring[r] = lut.ring[k];
    }
    return h;
}

template <int G, bool kTile>
__device__ bool candidate_search_grp(const DevMap& m, const PlanConsts& pc, const SpiralLut& lut, const LutHead& head,
                                     const LegCtx& c, uint8_t* tile, const Grp<G>& g, int& wi, int& wj) {
    const int M = c.nCand;
    int round = 0;
    int staged = -1;  // Chebyshev radius of the tile staged so far (kTile)
    bool useColX = false;
    // rank-table entries of the NEXT round are loaded while the current round is evaluated (from the
    // last register-held round on; searches that end earlier never touch the table)
    int nDi = 0, nDj = 0, nR = c.nRings;
#ifdef FPE_TRACE
    long long traceStage = 0;
#endif
    for (int base = 0; base < M; base += G, ++round) {
        const int k = base + g.sub;
        bool ok = false;
        int i = 0, j = 0;
        int di = 0, dj = 0, r = c.nRings;
        if (round < kLutHeadRounds) {
            if (k < M) {
                const int e = round == 0 ? head.dij[0] : head.dij[1];
                di = static_cast<int16_t>(e & 0xFFFF);
                dj = e >> 16;
                r = round == 0 ? head.ring[0] : head.ring[1];
            }
        } else {
            di = nDi;
            dj = nDj;
            r = nR;
        }
        if (round + 1 >= kLutHeadRounds) {
            const int kn = k + G;
            nR = c.nRings;
            if (kn < M) {
                nDi = lut.di[kn];
                nDj = lut.dj[kn];
                nR = lut.ring[kn];
            }
        }
        if constexpr (kTile) {
            // rings grow with rank: the last lane of the round holds the farthest ring it needs.  The
            // staged radius grows geometrically (x1.5) so that a long search needs O(log) staging steps
            // while a short one never stages much more than it reads (measured against per-round and
            // all-at-once staging on cfg-3/4/5).
            int need = g.bcast(r, G - 1) + pc.footReach;
            if (need > staged && staged >= 0) need = max(need, min(pc.tileH, staged + max(2, staged / 2)));
            if (need > staged) {
#ifdef FPE_TRACE
                const long long t0 = __builtin_readcyclecounter();
#endif
                if (staged < 0 && !c.rect) {  // the rectangle's own test is six compares: nothing to precompute
                    useColX = column_crossings(m, pc, c, g, reinterpret_cast<double*>(tile + tile_flag_bytes(pc)));
                    // the crossings are read by other lanes of the group (a group never spans wavefronts, and the
                    // LDS operations of a wavefront execute in order: only the compiler must not reorder)
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
                stage_annulus(m, pc, c, tile, g, staged, need, useColX);
                staged = need;
#ifdef FPE_TRACE
                traceStage += __builtin_readcyclecounter() - t0;
#endif
            }
        }
        if (k < M) {
            i = c.ici + di;
            j = c.icj + dj;
            ok = in_range(i, j, m.g.rows, m.g.cols);
            // SpiralIterator::generateRing filters rings nRings-1 and nRings by isInside;
            // the centre cell (ring 0) is pushed unfiltered by the constructor
            if (ok && r >= 1 && (r == c.nRings || r + 1 == c.nRings)) ok = cell_in_disc(m.g, i, j, c.cx, c.cy, c.R2);
            if (ok) ok = candidate_disc_ok<kTile>(m, pc, c, tile, i, j);
        }
        const unsigned long long mask = g.ballot(ok);
        if (mask) {
            const int l = __builtin_ctzll(mask);
            wi = g.bcast(i, l);
            wj = g.bcast(j, l);
#ifdef FPE_TRACE
            stamp_value(pc, c.cyc, 11, traceStage);
            stamp_value(pc, c.cyc, 12, round + 1);
#endif
            return true;
        }
    }
#ifdef FPE_TRACE
    stamp_value(pc, c.cyc, 11, traceStage);
    stamp_value(pc, c.cyc, 12, round);
#endif
    return false;
}

// ---- direct (L1/L2-served) passes of the common path ------------------------------------------------------
// At small windows the cells a leg needs in the common case — the default disc and the centroid
// rectangle, ~10^2 cells — are cheaper to read once straight from the cache hierarchy (lane = cell,
// lane = row) than to stage through LDS first; the LDS tile is kept for the spiral window of
// large foot discs (stage_annulus).

// One pass over a CircleIterator disc (centre c, bounding box bb), lanes = cells in row-major order:
// getFootholdMeanHeight (cpp:2520-2554) and, when kCheck, checkDefaultFoothold (cpp:2039-2082):
// valid iff >= 1 cell visited and no finite visited cell is below defaultFootholdThreshold_.
template <int G, bool kCheck>
__device__ __forceinline__ float disc_pass_direct(const DevMap& m, const PlanConsts& pc, double cx, double cy,
                                                  const BBox& bb, const Grp<G>& g, bool& defaultOk, float* scratch) {
    const int nb = bb.ni * bb.nj;
    const float njInv = rcp_small(bb.nj);
    float sum = 0.0f, last = 0.0f;
    int cnt = 0;
    bool any = false, fail = false;
    OrderedSum os{scratch, 0};
    for (int base = 0; base < nb; base += G) {
        const int t = base + g.sub;
        bool vis = false;
        float v = 0.0f;
        if (t < nb) {
            int a, bq;
            divmod_small(t, bb.nj, njInv, a, bq);
            const int i = bb.i0 + a, j = bb.j0 + bq;
            if (in_range(i, j, m.g.rows, m.g.cols) && cell_in_disc(m.g, i, j, cx, cy, pc.rf2)) {
                vis = true;
                const size_t off = static_cast<size_t>(i) * m.g.cols + j;
                const float e = m.elev[off];
                if (kCheck) {
                    const float tv = m.trav[off];
                    if (__builtin_isfinite(tv) && tv < pc.thrDefault) fail = true;  // cpp:2055-2057
                }
                v = __builtin_isfinite(e) ? e : 0.0f;  // cpp:2532-2537
            }
        }
        any |= vis;
        if (os.scratch) ordered_push(g, os, vis, v);
        else accumulate_heights(g, vis, v, sum, last, cnt);
    }
    if (os.scratch) ordered_finish(os, sum, last, cnt);
    if (kCheck) defaultOk = g.any(any) && !g.any(fail);
    return finish_mean(sum, last, cnt, pc.h);
}

// Two-phase form of disc_pass_direct for memory-level parallelism: disc_issue() computes
// membership and issues the loads of up to kDiscRounds*G cells; disc_consume() runs after other
// independent work (the centroid row scan) has overlapped their latency.  Larger discs fall back to
// the single-phase pass inside disc_consume().
constexpr int kDiscRounds = 4;  // up to 32 cells on 8 lanes: the 5x5 boxes of a 1 cm map stay pipelined
template <int G>
__device__ __forceinline__ constexpr int disc_rounds() { return G >= 64 ? 2 : kDiscRounds; }  // 64 lanes: 2 x 64 cells
struct DiscLoads {
    float e[kDiscRounds];  // elevation of this lane's cell in round r
    float t[kDiscRounds];  // traversability (centre disc only)
    int vis[kDiscRounds];  // 0/1, deliberately an int: a long-lived lane mask would occupy (and spill) a scalar register pair
    bool pipelined;
    bool mid;              // 3x3 form: round 0 holds cells 0-3 and 5-8, the middle cell is eMid / tMid
    float eMid, tMid;
};
// kMid: kernel variant for foot radii in [0.9, 1] x resolution, where every unclamped disc box is 3x3: the
// generic two-round issue is not compiled in at all (a wavefront with a clamped or smaller box takes the direct
// pass at consume time).  Besides the instructions, this removes the generic path's load destinations from the
// hot region: with both forms present the compiler's s_waitcnt insertion has to assume those loads pending at the
// join and stalls the 3x3 form on the row loads issued just before it (measured: 56.2 -> 53.7 us per launch).
// dy2tab (optional, 3x3 form only): (cell_pos(baseY, res, bb.j0 + k) - cy)^2 for k = 0..2, precomputed by the caller —
// the y side of a leg's geometry does not depend on the chain (fpe_bits.hpp, YEntry).
// kLoad = false: membership only (DiscLoads::vis / pipelined) — the caller defers the heights and reads the elevations
// itself later (one-wavefront-per-pose bit-window kernels, fpe_bits.hpp::flush_seqrec).
template <int G, bool kCheck, bool kMid = false, bool kLoad = true>
__device__ __forceinline__ void disc_issue(const DevMap& m, const PlanConsts& pc, double cx, double cy, const BBox& bb,
                                           const Grp<G>& g, DiscLoads& d, const double* dy2tab = nullptr) {
    const int nb = bb.ni * bb.nj;
    d.mid = false;
    d.eMid = d.tMid = 0.0f;
    if constexpr (G == 8) {
        // The usual disc of a 2 cm map is a 3x3 box: nine cells on eight lanes would cost a second round of
        // membership arithmetic for one cell.  The middle cell is inside the disc by construction
        // (PlanConsts::midCellInside), so the eight lanes take the other eight cells and every lane loads the
        // middle one without testing it.  The choice is made per wavefront: all its active legs must qualify.
        // "Unclamped": boundPositionToRange moves an outside corner onto the map edge, i.e. into the first or last
        // row / column — a box that stays clear of those was not clamped (and lies inside the map).
        const bool ok3 = pc.midCellInside != 0 && bb.ni == 3 && bb.nj == 3 && bb.i0 >= 1 && bb.j0 >= 1 &&
                         bb.i0 + 4 <= m.g.rows && bb.j0 + 4 <= m.g.cols;
        if (__ballot(!ok3) == 0ull) {
            d.mid = true;
            d.pipelined = true;
            const int t = g.sub + (g.sub >= 4 ? 1 : 0);
            const int a = t >= 6 ? 2 : (t >= 3 ? 1 : 0);
            const int i = bb.i0 + a, j = bb.j0 + (t - 3 * a);
#pragma unroll
            for (int r = 0; r < kDiscRounds; ++r) {
                d.vis[r] = false;
                d.e[r] = d.t[r] = 0.0f;
            }
            if (dy2tab) {
                const double dx = cell_pos(m.g.baseX, m.g.res, i) - cx;  // cell_in_disc with the column's dy * dy looked up
                d.vis[0] = ((dx * dx + dy2tab[t - 3 * a]) <= pc.rf2) ? 1 : 0;
            } else {
                d.vis[0] = cell_in_disc(m.g, i, j, cx, cy, pc.rf2) ? 1 : 0;  // the unclamped box lies inside the map
            }
            if constexpr (kLoad) {
                const size_t offM = static_cast<size_t>(bb.i0 + 1) * m.g.cols + (bb.j0 + 1);
                d.eMid = m.elev[offM];
                if (kCheck) d.tMid = m.trav[offM];
                if (d.vis[0]) {
                    const size_t off = static_cast<size_t>(i) * m.g.cols + j;
                    d.e[0] = m.elev[off];
                    if (kCheck) d.t[0] = m.trav[off];
                }
            }
            return;
        }
        if constexpr (kMid) {
            d.pipelined = false;  // rare (map border): disc_consume runs the direct pass
#pragma unroll
            for (int r = 0; r < kDiscRounds; ++r) {
                d.vis[r] = false;
                d.e[r] = d.t[r] = 0.0f;
            }
            return;
        }
    }
    d.pipelined = nb <= disc_rounds<G>() * G;
    if (!d.pipelined) return;
    const float njInv = rcp_small(bb.nj);
#pragma unroll
    for (int r = 0; r < disc_rounds<G>(); ++r) {
        const int t = r * G + g.sub;
        d.vis[r] = false;
        d.e[r] = 0.0f;
        d.t[r] = 0.0f;
        if (__ballot(t < nb) == 0ull) continue;  // wave-uniform: a round past every box of the wavefront
        // one level of control flow per round (nested per-lane tests are compiled into exec-mask mazes): the cell of slot t
        // by the branch-free form of divmod_small (estimate off by at most one), membership evaluated for every lane
        const int njS = max(bb.nj, 1);
        int a = static_cast<int>(static_cast<float>(t) * njInv);
        int bq = t - a * njS;
        const int fix = (bq >= njS ? 1 : 0) - (bq < 0 ? 1 : 0);
        a += fix;
        bq -= fix * njS;
        const int i = bb.i0 + a, j = bb.j0 + bq;
        const bool inMap = in_range(i, j, m.g.rows, m.g.cols), inDisc = cell_in_disc(m.g, i, j, cx, cy, pc.rf2);
        const bool vis = (t < nb) & inMap & inDisc;
        d.vis[r] = vis;
        if constexpr (kLoad) {
            if (vis) {
                const size_t off = static_cast<size_t>(i) * m.g.cols + j;
                d.e[r] = m.elev[off];
                if (kCheck) d.t[r] = m.trav[off];
            }
        }
    }
}
template <int G, bool kCheck, bool kMid = false>
__device__ __forceinline__ float disc_consume(const DevMap& m, const PlanConsts& pc, double cx, double cy, const BBox& bb,
                                              const Grp<G>& g, const DiscLoads& d, bool& defaultOk, float* scratch) {
    if (!d.pipelined) return disc_pass_direct<G, kCheck>(m, pc, cx, cy, bb, g, defaultOk, scratch);
    float sum = 0.0f, last = 0.0f;
    int cnt = 0;
    bool any = false, fail = false;
    if constexpr (G == 8) {
        if (d.mid) {  // wave-uniform (disc_issue): cells 0-3 = lanes 0-3, the middle cell, cells 5-8 = lanes 4-7
            const float v0 = __builtin_isfinite(d.e[0]) ? d.e[0] : 0.0f;  // cpp:2532-2537
            const float vm = __builtin_isfinite(d.eMid) ? d.eMid : 0.0f;
            if (kCheck) {  // cpp:2055-2057
                fail = (d.vis[0] && __builtin_isfinite(d.t[0]) && d.t[0] < pc.thrDefault) ||
                       (__builtin_isfinite(d.tMid) && d.tMid < pc.thrDefault);
            }
            const bool inc0 = d.vis[0] && v0 < 10, incM = vm < 10;  // cpp:2539
            cnt = __builtin_popcountll(g.ballot(inc0)) + (incM ? 1 : 0);
            const float x0 = inc0 ? v0 : -0.0f;
            sum = SeqSum<G, 0, 4>::run(sum, x0);
            sum = sum + (incM ? vm : -0.0f);
            sum = SeqSum<G, 4, 8>::run(sum, x0);
            if (__ballot(cnt == 0) != 0ull) {
                // every visited value was >= 10: the mean falls back to the LAST visited value (cpp:2547-2551) —
                // the highest visited lane of cells 5-8, else the middle cell
                const unsigned long long mv = g.ballot(d.vis[0]) >> 4;
                const int l = mv ? 4 + (63 - __builtin_clzll(mv)) : 0;
                const float lv = g.bcast(v0, l);
                last = mv ? lv : vm;
            }
            if (kCheck) defaultOk = !g.any(fail);  // the middle cell is always visited
            return finish_mean(sum, last, cnt, pc.h);
        }
        if constexpr (kMid) {  // the 3x3-only variant issues nothing else (disc_issue): rounds 1.. are never loaded
            __builtin_unreachable();
        }
    }
    if constexpr (G <= 16) {
        // small discs: no compaction at all — G dependent adds per round on swizzled lane values
        float v[kDiscRounds];
        bool anyVis = false;
#pragma unroll
        for (int r = 0; r < disc_rounds<G>(); ++r) {
            v[r] = __builtin_isfinite(d.e[r]) ? d.e[r] : 0.0f;                                          // cpp:2532-2537
            if (kCheck && d.vis[r] && __builtin_isfinite(d.t[r]) && d.t[r] < pc.thrDefault) fail = true;  // cpp:2055-2057
            const bool inc = d.vis[r] && v[r] < 10;                                                     // cpp:2539
            anyVis |= d.vis[r];
            cnt += __builtin_popcountll(g.ballot(inc));
            if (r == 0 || __ballot(d.vis[r]) != 0ull) sum = SeqSum<G>::run(sum, inc ? v[r] : -0.0f);
        }
        const bool groupVis = g.any(anyVis);
        if (__ballot(cnt == 0 && groupVis) != 0ull) {
            // every visited value was >= 10: the mean falls back to the LAST visited value (cpp:2547-2551)
            unsigned long long mk = 0ull;
            int lastRound = 0;
#pragma unroll
            for (int r = 0; r < disc_rounds<G>(); ++r) {
                const unsigned long long mr = g.ballot(d.vis[r]);
                if (mr) {
                    mk = mr;
                    lastRound = r;
                }
            }
            const int l = mk ? 63 - __builtin_clzll(mk) : 0;
#pragma unroll
            for (int r = 0; r < disc_rounds<G>(); ++r) {
                const float lv = g.bcast(v[r], l);
                if (mk && r == lastRound) last = lv;
            }
        }
        if (kCheck) defaultOk = groupVis && !g.any(fail);
        return finish_mean(sum, last, cnt, pc.h);
    }
    OrderedSum os{scratch, 0};
#pragma unroll
    for (int r = 0; r < disc_rounds<G>(); ++r) {
        const float v = __builtin_isfinite(d.e[r]) ? d.e[r] : 0.0f;                                 // cpp:2532-2537
        if (kCheck && d.vis[r] && __builtin_isfinite(d.t[r]) && d.t[r] < pc.thrDefault) fail = true;  // cpp:2055-2057
        any |= d.vis[r];
        if (os.scratch) ordered_push(g, os, d.vis[r], v);
        else accumulate_heights(g, d.vis[r], v, sum, last, cnt);
    }
    if (os.scratch) ordered_finish(os, sum, last, cnt);
    if (kCheck) defaultOk = g.any(any) && !g.any(fail);
    return finish_mean(sum, last, cnt, pc.h);
}

constexpr int kOnDemandMaxFoot = 4;

struct CentroidOut {
    double x, y;
    float z;
    int row, col;
    int code;
};

// checkFootholdUseCentroidMethod (cpp:1605-1997) on the rectangle s around the centre (flags
// already staged).  zCentre = mean height at the centre (whole-region-valid result, cpp:1687).
struct CentroidScan {
    bool whole;
    int minRow, maxRow;
};
// Row scan of the centroid rectangle, lane = row: per-row count of cells below the default threshold
// (raw `<`, NaN passes) gives both the whole-region test (cpp:1649-1658) and the first/last blocked
// rows (cpp:1717-1750; in-bounds columns only).
template <int G>
__device__ CentroidScan centroid_scan(const DevMap& m, const PlanConsts& pc, const Submap& s, const Grp<G>& g) {
    CentroidScan r0;
    r0.whole = false;
    r0.minRow = r0.maxRow = 0;
    if (!s.ok) return r0;
    const int ni = s.ni, nj = s.nj;
    const int rightCol = nj - 1;
    bool anyBelow = false;
    int minRow = 0, maxRow = 0, k = 0;
    for (int rbase = 0; rbase < ni; rbase += G) {
        const int r = rbase + g.sub;
        bool blocked = false;
        if (r < ni) {
            int cnt = 0;
            // the rectangle lies inside the map (clamped corners, s.ok): raw `<`, NaN passes (cpp:1653, 1736)
            const float* rowp = m.trav + static_cast<size_t>(s.i0 + r) * m.g.cols + s.j0;
            // columns in chunks of 8 independent loads (clamped index, predicated count): one memory
            // round trip per chunk instead of one per cell
            for (int c0 = 0; c0 < nj; c0 += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = rowp[min(c0 + u, nj - 1)];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (c0 + u < nj && v[u] < pc.thrDefault) ++cnt;
            }
            anyBelow |= cnt > 0;
            blocked = cnt > ((rightCol + 1) * 0.5);  // cpp:1743
        }
        const unsigned long long mask = g.ballot(blocked);
        if (mask) {
            if (k == 0) minRow = rbase + __builtin_ctzll(mask);
            maxRow = rbase + 63 - __builtin_clzll(mask);
            k += __builtin_popcountll(mask);
        }
    }
    r0.whole = ni * nj > 0 && !g.any(anyBelow);
    r0.minRow = minRow;
    r0.maxRow = maxRow;
    return r0;
}

// Two-phase form of centroid_scan: rows_issue() starts the loads of up to CH*G rows x 4*C4 columns right after
// the corner arithmetic; rows_finish() counts after the disc membership math has overlapped their latency.
// Larger rectangles fall back to centroid_scan().  The shape is a property of the kernel variant: the 3x3-only
// 8-lane kernel scans the 11x6 rectangle of a 2 cm map (2 chunks x 8 columns), the generic small-group kernels
// up to 24 rows x 12 columns (21x11 at 1 cm / R 0.1), the 64-lane kernels up to 64 rows x 24 columns (41x21 at 0.5 cm).
template <int CH, int C4>
struct RowLoads {
    float v[CH][4 * C4];
    bool pipelined;
    bool small;  // wave-uniform: every active leg's rectangle fits 2 chunks x 8 columns (the 11x6 of a 2 cm map)
};
template <int G, bool kMid>
struct RowShape {
    static constexpr int CH = (G >= 64) ? 1 : (kMid ? 2 : 3);
    static constexpr int C4 = (G >= 64) ? 6 : (kMid ? 2 : 3);
    typedef RowLoads<CH, C4> Loads;
};
[[maybe_unused]] constexpr int kRowOverreadBytes = 6 * 16;  // widest row read (C4 = 6); the layers carry this much tail padding
template <int G, int CH, int C4>
__device__ __forceinline__ void rows_issue(const DevMap& m, const Submap& s, const Grp<G>& g, RowLoads<CH, C4>& rl) {
    rl.pipelined = s.ok && s.ni <= CH * G && s.nj <= 4 * C4;
    rl.small = (CH > 2 || C4 > 2) && __ballot(!(s.ni <= 2 * G && s.nj <= 8)) == 0ull;
    if (!rl.pipelined) return;
#pragma unroll
    for (int ch = 0; ch < CH; ++ch) {
        if (ch >= 2 && rl.small) continue;
        const int r = min(ch * G + g.sub, s.ni - 1);  // clamped: idle lanes re-read the last row
        const float* rowp = m.trav + static_cast<size_t>(s.i0 + r) * m.g.cols + s.j0;
        // 16-byte loads at 4-byte alignment; columns >= nj are never counted, and the layer allocation carries
        // kRowOverreadBytes of tail padding so the over-read of the last row stays in bounds
        typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
#pragma unroll
        for (int q = 0; q < C4; ++q) {
            if (q >= 2 && rl.small) continue;
            const f4u a = *reinterpret_cast<const f4u*>(rowp + 4 * q);
            rl.v[ch][4 * q + 0] = a.x;
            rl.v[ch][4 * q + 1] = a.y;
            rl.v[ch][4 * q + 2] = a.z;
            rl.v[ch][4 * q + 3] = a.w;
        }
    }
}
template <int G, int CH, int C4>
__device__ __forceinline__ CentroidScan rows_finish(const DevMap& m, const PlanConsts& pc, const Submap& s,
                                                    const Grp<G>& g, const RowLoads<CH, C4>& rl) {
    if (!rl.pipelined) return centroid_scan(m, pc, s, g);
    CentroidScan r0;
    const int ni = s.ni, nj = s.nj, rightCol = nj - 1;
    bool anyBelow = false;
    int minRow = 0, maxRow = 0, k = 0;
#pragma unroll
    for (int ch = 0; ch < CH; ++ch) {
        if (ch >= 2 && rl.small) continue;
        const int r = ch * G + g.sub;
        int cnt = 0;
#pragma unroll
        for (int u = 0; u < 4 * C4; ++u) {
            if (u >= 8 && rl.small) continue;
            if (u < nj && rl.v[ch][u] < pc.thrDefault) ++cnt;  // raw `<`, NaN passes (cpp:1653, 1736)
        }
        const bool live = r < ni;
        anyBelow |= live && cnt > 0;
        const bool blocked = live && cnt > ((rightCol + 1) * 0.5);  // cpp:1743
        const unsigned long long mask = g.ballot(blocked);
        if (mask) {
            if (k == 0) minRow = ch * G + __builtin_ctzll(mask);
            maxRow = ch * G + 63 - __builtin_clzll(mask);
            k += __builtin_popcountll(mask);
        }
    }
    r0.whole = ni * nj > 0 && !g.any(anyBelow);
    r0.minRow = minRow;
    r0.maxRow = maxRow;
    return r0;
}

// checkFootholdUseCentroidMethod (cpp:1605-1997) given the row scan.  zCentre = mean height at the
// centre (the whole-region-valid result reuses it, cpp:1687).
// checkFootholdUseCentroidMethod (cpp:1605-1997) given the row scan, in two halves: centroid_begin()
// decides the case and ISSUES the loads of the result's foot disc; centroid_end() consumes them
// after independent work (default-track height, spiral search) has overlapped their latency.
// zCentre = mean height at the centre (the whole-region-valid result reuses it, cpp:1687).
struct CentroidPending {
    CentroidOut o;
    int needDisc;  // 0/1 (int on purpose, see DiscLoads::vis)
    BBox rb;
    DiscLoads dl;
};
template <int G, bool kMid = false>
__device__ void centroid_begin(const DevMap& m, const PlanConsts& pc, const LegCtx& c, const Submap& s,
                               const CentroidScan& sc, float zCentre, const Grp<G>& g, CentroidPending& cp) {
    CentroidOut& o = cp.o;
    cp.needDisc = 0;
    o.x = 0.0;
    o.y = 0.0;
    o.z = 0.0f;
    o.row = -1;
    o.col = -1;
    o.code = 5;
    if (!s.ok) {  // cpp:1628-1631
        o.code = 6;
        return;
    }
    const int bottomRow = s.ni - 1, rightCol = s.nj - 1;
    const int minRow = sc.minRow, maxRow = sc.maxRow;
    if (sc.whole) {  // cpp:1684-1689
        o.x = c.cx;
        o.y = c.cy;
        o.z = zCentre;
        o.row = c.ici;
        o.col = c.icj;
        o.code = 0;
        return;
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
        return;  // first and last row blocked: no branch taken, result stays (0,0,0)
    }
    // map.getPosition(newIndex) on the SUBMAP (cpp:1816), height on the full map (cpp:1820)
    o.x = cell_pos(s.baseX, m.g.res, newRow);
    o.y = cell_pos(s.baseY, m.g.res, newCol);
    // quantities 0-3: corners of the result's foot disc; 4-5: getIndex(result)
    const Box disc{o.x, o.y, pc.rf, pc.rf};
    Corners<G, 8> cr;
    cr.eval(m.g, g, disc, disc, disc, disc, 0x2u);
    cp.rb = cr.template bbox<0>(g);
    o.row = cr.template get<4>(g);
    o.col = cr.template get<5>(g);
    disc_issue<G, false, kMid>(m, pc, o.x, o.y, cp.rb, g, cp.dl);
    cp.needDisc = 1;
}
template <int G, bool kMid = false>
__device__ void centroid_end(const DevMap& m, const PlanConsts& pc, const Grp<G>& g, CentroidPending& cp, float* scratch) {
    if (cp.needDisc != 0) {
        bool unused;
        cp.o.z = disc_consume<G, false, kMid>(m, pc, cp.o.x, cp.o.y, cp.rb, g, cp.dl, unused, scratch);
    }
}

struct NominalOut {
    int row, col;
    double x, y;
    float z;
    int valid, source;
};

__device__ __forceinline__ void nominal_invalid(NominalOut& o, double cx, double cy, int source) {
    o.row = o.col = -1;
    o.x = cx;  // cpp:2016-2017
    o.y = cy;
    o.z = 0.0f;
    o.valid = 0;
    o.source = source;
}

// Per-leg constants that do not change along the chain.
struct LegConst {
    float Rf;       // float search radius (cpp:1616 uses searchRadius_*2 in f32)
    double R2;      // double(Rf)^2
    int nRings;     // ceil(double(Rf) / res), SpiralIterator::nRings_
    int nCand;      // spiral rank-table entries of rings 0..nRings
    double lx, ly;  // centroid rectangle (cpp:1616-1617)
};
__device__ __forceinline__ LegConst make_leg_const(float Rf, double res, const SpiralLut& lut) {
    LegConst k;
    k.Rf = Rf;
    const double R = static_cast<double>(Rf);
    k.R2 = R * R;
    k.nRings = static_cast<int>(static_cast<unsigned int>(ceil(R / res)));
    k.nCand = lut.ringStart[(k.nRings < lut.maxRing ? k.nRings : lut.maxRing) + 1];
    k.lx = static_cast<double>(Rf * 2);
    k.ly = static_cast<double>(Rf);
    return k;
}

// Default-track mean height request (getDefaultFootholds, cpp:2289-2301) riding along with a leg search.
struct DefaultDisc {
    int want;  // 0/1 (int on purpose, see DiscLoads::vis)
    double x, y;
    BBox bb;
    float z;
};

template <int G, bool kMid>
__device__ void spiral_search(const DevMap& m, const PlanConsts& pc, const SpiralLut& lut, const LutHead& head,
                              const LegCtx& c, uint8_t* tile, const Grp<G>& g, float zCentre, NominalOut& no);

// One leg: centroid method (cpp:1605-1997) + checkFoothold (cpp:2001-2036) around the same centre.
// bb = CircleIterator box of the centre disc, s = getSubmap geometry of the centroid rectangle (both
// from the corner lanes).  kCentroid=false skips the centroid track (open-loop fpe_search_legs).
template <int G, bool kCentroid, bool kMid = false>
__device__ void search_leg(const DevMap& m, const PlanConsts& pc, const SpiralLut& lut, const LutHead& head, LegCtx& c,
                           const LegConst& lk,
                           uint8_t* tile, const Grp<G>& g, const BBox& bb, const Submap& s, DefaultDisc& dflt,
                           NominalOut& no, CentroidOut& co) {
    c.R2 = lk.R2;
    c.nRings = lk.nRings;
    c.nCand = lk.nCand;
    c.ti0 = c.ici - pc.tileH;
    c.tj0 = c.icj - pc.tileH;
    // software pipeline: (1) row loads of the centroid rectangle, (2) disc membership math and disc
    // loads, (3) row counts, (4) centre disc -> default check and height, (5) centroid case + loads
    // of its result disc, (6) default-track height, (7) spiral search if needed, (8) centroid height
    typename RowShape<G, kMid>::Loads rl;
    if (kCentroid) rows_issue(m, s, g, rl);
    DiscLoads dc;
    disc_issue<G, true, kMid>(m, pc, c.cx, c.cy, bb, g, dc);
    DiscLoads dd;
    if (dflt.want != 0) disc_issue<G, false, kMid>(m, pc, dflt.x, dflt.y, dflt.bb, g, dd);
    stamp(pc, c.cyc, 3);
    CentroidScan sc;
    if (kCentroid) sc = rows_finish(m, pc, s, g, rl);
    stamp(pc, c.cyc, 4);
    bool defaultOk = true;
    float* scratch = reinterpret_cast<float*>(tile);  // the tile is idle outside the spiral search
    const float zCentre = disc_consume<G, true, kMid>(m, pc, c.cx, c.cy, bb, g, dc, defaultOk, scratch);  // cpp:2012 + cpp:2029
    stamp(pc, c.cyc, 5);
    CentroidPending cp;
    if (kCentroid) centroid_begin<G, kMid>(m, pc, c, s, sc, zCentre, g, cp);                          // cpp:818-821
    stamp(pc, c.cyc, 6);
    if (dflt.want != 0) {
        bool unused;
        dflt.z = disc_consume<G, false, kMid>(m, pc, dflt.x, dflt.y, dflt.bb, g, dd, unused, scratch);  // cpp:2289-2301
    }
    stamp(pc, c.cyc, 7);
    if (defaultOk) {
        no.valid = 1;
        no.source = 0;
        no.row = c.ici;
        no.col = c.icj;
        no.x = c.cx;  // cpp:2016-2017
        no.y = c.cy;
        no.z = zCentre;
    } else {
        spiral_search<G, kMid>(m, pc, lut, head, c, tile, g, zCentre, no);
    }
    if (kCentroid) {
        centroid_end<G, kMid>(m, pc, g, cp, scratch);
        co = cp.o;
    }
}

// checkCandidateFoothold half of checkFoothold (cpp:2022-2029) once the default foothold has failed.
template <int G, bool kMid>
__device__ void spiral_search(const DevMap& m, const PlanConsts& pc, const SpiralLut& lut, const LutHead& head,
                              const LegCtx& c, uint8_t* tile, const Grp<G>& g, float zCentre, NominalOut& no) {
    nominal_invalid(no, c.cx, c.cy, 2);
    int wi = 0, wj = 0;
    bool found;
    if (kMid || (pc.footRobust && pc.nFoot <= kOnDemandMaxFoot)) {  // the 3x3-only variant is launched for such discs only
        // tiny foot discs: evaluate the few cells a candidate needs straight from the map
        found = candidate_search_grp<G, false>(m, pc, lut, head, c, tile, g, wi, wj);  // cpp:2022
    } else {
        found = candidate_search_grp<G, true>(m, pc, lut, head, c, tile, g, wi, wj);
    }
    if (found) {
        no.valid = 1;
        no.source = 1;
        no.row = wi;
        no.col = wj;
        no.x = cell_pos(m.g.baseX, m.g.res, wi);  // cpp:2105-2107
        no.y = cell_pos(m.g.baseY, m.g.res, wj);
        no.z = zCentre;  // z at the DEFAULT centre even for a candidate (cpp:2029)
    }
}

// Result records of the 8-lane kernels leave as streaming stores (nt): nothing on the device reads them back within the
// launch, and a record left dirty in the L2 displaces map lines and is only written out when the kernel ends (14 MB per
// headline launch behind the last wavefront).  Measured: headline 30.25 -> 29.1 us, cfg-4 1.193 -> 1.151 ms.  The
// one-wavefront-per-pose kernels store one 32-byte record per leg from a single lane: streamed, those partial lines cost
// more than they save (cfg-3 +4 %, cfg-5 +1.5 %), so they keep ordinary stores (kStream = false).
typedef unsigned int fpe_v4u __attribute__((ext_vector_type(4)));
typedef unsigned int fpe_v2u __attribute__((ext_vector_type(2)));
template <bool kStream, class T>
__device__ __forceinline__ void store_record(T* dst, const T& v) {
    static_assert(sizeof(T) % 8 == 0, "records are stored in 8- or 16-byte pieces");
#ifdef FPE_DBG_NOSTORE  // measurement only: the store stays in the code (and everything it depends on) but never executes
    if (reinterpret_cast<uintptr_t>(dst) != 1) return;
#endif
    if constexpr (!kStream) {
        *dst = v;
    } else if constexpr (sizeof(T) % 16 == 0) {
        fpe_v4u w[sizeof(T) / 16];
        __builtin_memcpy(w, &v, sizeof(T));
#pragma unroll
        for (unsigned k = 0; k < sizeof(T) / 16; ++k) __builtin_nontemporal_store(w[k], reinterpret_cast<fpe_v4u*>(dst) + k);
    } else {
        fpe_v2u w[sizeof(T) / 8];
        __builtin_memcpy(w, &v, sizeof(T));
#pragma unroll
        for (unsigned k = 0; k < sizeof(T) / 8; ++k) __builtin_nontemporal_store(w[k], reinterpret_cast<fpe_v2u*>(dst) + k);
    }
}

template <bool kStream = false>
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
    store_record<kStream>(dst, f);
}

// The two exchange forms of a nominal foothold (include/fpe.h): the 16-byte fpe_selected_foothold and the 8-byte
// fpe_selected_packed (row / col biased by FPE_PACKED_BIAS in 14 bits each, valid and source above them).
template <bool kStream>
__device__ __forceinline__ void store_selected(const fpe_plan_out& out, size_t o, int row, int col, float z, int valid, int source, int leg,
                                               int cyc) {
    if (out.selected) {
        fpe_selected_foothold sf;
        sf.row = row; sf.col = col; sf.z = z;
        sf.valid = static_cast<uint8_t>(valid); sf.source = static_cast<uint8_t>(source);
        sf.foot_id = static_cast<uint8_t>(leg); sf.gait_cycle_id = static_cast<uint8_t>(cyc);
        store_record<kStream>(out.selected + o, sf);
    }
    if (out.selected_packed) {
        fpe_selected_packed sp;
        sp.cell = (static_cast<uint32_t>(row + FPE_PACKED_BIAS) & 0x3FFFu) | ((static_cast<uint32_t>(col + FPE_PACKED_BIAS) & 0x3FFFu) << 14) |
                  ((static_cast<uint32_t>(valid) & 1u) << 28) | ((static_cast<uint32_t>(source) & 3u) << 29);
        sp.z = z;
        store_record<kStream>(out.selected_packed + o, sp);
    }
}

// getPolygonCenter (cpp:2421-2463): feet[leg][xyz] in LDS.  Only the centre's x is computed: the next
// centre takes y from initialPose_/ajustedPose_ (cpp:2201, 2272) and getDefaultFootholdNext zeroes z
// (cpp:2411-2418), so the y and z of the centre never reach a result (two f64 divisions saved per track).
// x / 3.0, correctly rounded, in three operations instead of the division sequence (the feet-polygon centre sits at
// the head of every gait cycle's dependent chain).  With c = RN(1/3), q = RN(x * c) is within 1.5 ulp of x / 3, so
// r = x - 3 q is exact in an FMA (a small multiple of ulp(q)), and q + r * c = x/3 + (x/3 - q) * eps with |eps| <= 2^-53:
// the perturbation is < 2^-52 ulp while x / 3 is never closer than ulp / 6 to a rounding boundary (3 * midpoint is an
// odd multiple of half an ulp of the quotient's grid, x an even one), so the final FMA rounds to RN(x / 3).  Zero, NaN
// and values near the ends of the exponent range take the division itself (wave-uniform branch).
__device__ __forceinline__ double div3(double x) {
    const double ax = fabs(x);
    if (__ballot(!(ax > 1e-280 && ax < 1e300)) != 0ull) return x / 3.0;
    const double c = 0x1.5555555555555p-2;
    const double q = x * c;
    const double r = __builtin_fma(-3.0, q, x);
    return __builtin_fma(r, c, q);
}

__device__ __forceinline__ double polygon_center_x(const double (*feet)[3]) {
    const double x1 = feet[0][0], y1 = feet[0][1];
    double x2 = feet[1][0], y2 = feet[1][1];
    double sum_x = 0, sum_s = 0;
#pragma unroll
    for (int i = 1; i <= 2; i++) {
        const double x3 = feet[i + 1][0], y3 = feet[i + 1][1];  // i=1: LH, i=2: LF
        const double s = ((x2 - x1) * (y3 - y1) - (x3 - x1) * (y2 - y1)) / 2.0;
        sum_x += (x1 + x2 + x3) * s;
        sum_s += s;
        x2 = x3;
        y2 = y3;
    }
    return div3(sum_x / sum_s);
}

// getGaitCycleSearchGridMap (cpp:2307-2349) in the FIRST gait cycle: the opt track's current feet are the shifted
// stance like every other track's (setFirstGait, cpp:582-588), so the next feet centre is (centre.x + stepLength_,
// initialPose_[1] + 0) and the gate is getSubmap(centre, isos_.length x isos_.width) succeeding (cpp:2345-2349).
// Later cycles depend on the NLopt results and are not evaluated.
__device__ __forceinline__ uint8_t opt_gate_cycle0(const MapGeom& g, const PlanConsts& pc, double ctrX, double y0) {
    const double px = ctrX + pc.step;  // cpp:2327
    const double py = y0 + 0.0;        // cpp:2329 with ajustedPose_[1] = 0 (cpp:759)
    const bool ok = centre_usable(px, py) && submap_info(g, px, py, pc.isosLen, pc.isosWid).ok;
    return ok ? 0 : static_cast<uint8_t>(FPE_POSE_OPT_SUBMAP_FAILED);
}

// LDS of one pose (all offsets multiples of 16).
struct PoseShared {
    double cur[3][4][3];  // current feet of the default / centroid / nominal tracks (cpp:1338, 1413, 1480)
    double nxt[3][4][3];  // this phase's results per track
    double ctr[4];        // x of the feet-polygon centre of each track for this phase ([3] pads to 16 B)
    double polyX[4][8];   // search polygon vertices per leg
    double polyY[4][8];
    int valid[4];
    int pad[4];
    int8_t footDa[kMaxFootOffsets];
    int8_t footDb[kMaxFootOffsets];
    int16_t footOff[kMaxFootOffsets];
};

// Per-leg constants of a pose (do not change along the chain).
struct LegStatic {
    float Rf;       // search radius of this leg (fpe_pose override or fpe_params.searchRadius)
    int polyKind;   // 0 reference rectangle, 1 hexagon
    bool radiusOk;  // Rf within the radius the LDS tile was sized for
    LegConst lk;
    double biasX, biasY;  // defaultBias of the leg (cpp:403-421)
};
__device__ __forceinline__ LegStatic make_leg_static(const PlanConsts& pc, const fpe_pose* pp, int leg, double res,
                                                     const SpiralLut& lut) {
    LegStatic ls;
    ls.Rf = pp->leg_search_radius[leg];
    if (!(ls.Rf > 0.0f)) ls.Rf = pc.searchRadius;
    ls.polyKind = pp->leg_polygon_kind[leg];
    ls.radiusOk = ls.Rf <= pc.maxSearchRadius;
    ls.lk = make_leg_const(ls.Rf, res, lut);
    ls.biasX = pc.biasX[leg];
    ls.biasY = pc.biasY[leg];
    return ls;
}

// One swing leg of one phase: next default positions on the three tracks, search polygon, the leg
// search, results to LDS (for the commit decision) and to HBM.
// What a leg hands to the commit step: the next positions of the three tracks and its validity.
struct LegCommit {
    double v[3][3];  // [track][xyz]
    int valid;
};
// kDirect: the results stay in registers (LegCommit) for a commit decided by wave ballot (a pose lives in one
// wavefront); otherwise they are staged in PoseShared::nxt / valid for the LDS commit of the multi-wave forms.
template <int G, bool kMid = false, bool kDirect = false>
__device__ __forceinline__ void leg_phase(const DevMap& m, const PlanConsts& pc, const SpiralLut& lut, const LutHead& head,
                                          PoseShared& sh, uint8_t* tile, const Grp<G>& g, int leg, const LegStatic& ls,
                                          double y0, double adjY, double advance, int cyc, int nCycles, int b, bool live,
                                          const fpe_plan_out& out, LegCommit* lc = nullptr) {
    const float Rf = ls.Rf;
    const int polyKind = ls.polyKind;
    const bool radiusOk = ls.radiusOk;
    const LegConst& lk = ls.lk;
    const double biasX = ls.biasX, biasY = ls.biasY;
        // next default positions of this leg on the three tracks (cpp:2199-2213, 2270-2284)
        const double Ny = y0 + adjY;                         // cpp:2201
        const double nx0 = (sh.ctr[0] + advance) + biasX;  // cpp:2199, 2414
        const double nx1 = (sh.ctr[1] + advance) + biasX;
        const double nx2 = (sh.ctr[2] + advance) + biasX;
        const double ny = Ny + biasY;                        // identical on the three tracks
        // search polygon from the NOMINAL track (cpp:2235-2244, getSearchPolygon cpp:2496-2517)
        if (g.sub == 0) {
            const double r = static_cast<double>(Rf);
            double* vx = sh.polyX[leg];
            double* vy = sh.polyY[leg];
            if (polyKind == 0) {
                vx[0] = nx2 + r;  vy[0] = ny + 0.5 * r;
                vx[1] = nx2 + r;  vy[1] = ny - 0.5 * r;
                vx[2] = nx2 - r;  vy[2] = ny - 0.5 * r;
                vx[3] = nx2 - r;  vy[3] = ny + 0.5 * r;
            } else {
                const double hx = 0.5 * r;
                const double hy = (0.5 * r) * 0.8660254037844386;
                vx[0] = nx2 + r;   vy[0] = ny;
                vx[1] = nx2 + hx;  vy[1] = ny - hy;
                vx[2] = nx2 - hx;  vy[2] = ny - hy;
                vx[3] = nx2 - r;   vy[3] = ny;
                vx[4] = nx2 - hx;  vy[4] = ny + hy;
                vx[5] = nx2 + hx;  vy[5] = ny + hy;
            }
        }

        LegCtx c;
        c.cyc = cyc;
        c.cx = nx1;  // centre from the CENTROID track (cpp:861-862)
        c.cy = ny;
        c.nv = (polyKind == 0) ? 4 : 6;
        {
            const double r = static_cast<double>(Rf);
            c.rect = polyKind == 0;
            c.xhi = nx2 + r;        // vertices LU/RU x (cpp:2501-2503)
            c.xlo = nx2 - r;        // RD/LD x
            c.yhi = ny + 0.5 * r;   // LU/LD y
            c.ylo = ny - 0.5 * r;   // RU/RD y
        }
        c.vx = sh.polyX[leg];
        c.vy = sh.polyY[leg];
        c.footDa = sh.footDa;
        c.footDb = sh.footDb;
        c.footOff = sh.footOff;

        NominalOut no;
        CentroidOut co;
        float zDefault = 0.0f;
        BBox dbox;
        bool haveDbox = false;
        if (!radiusOk) {
            nominal_invalid(no, c.cx, c.cy, 3);
            co.x = co.y = 0.0; co.z = 0.0f; co.row = co.col = -1; co.code = 6;
        } else if (!centre_usable(c.cx, c.cy)) {
            nominal_invalid(no, c.cx, c.cy, 2);
            co.x = co.y = 0.0; co.z = 0.0f; co.row = co.col = -1; co.code = 6;
        } else {
            // corner lanes: box 0 = centre foot disc, box 1 = centroid rectangle (half extents
            // 0.5*lx, 0.5*ly: p -/+ (-0.5*l) == p +/- 0.5*l exactly), box 2 = default-track disc,
            // box 3 = getIndex(centre) (zero half extent)
            const Box b0{c.cx, c.cy, pc.rf, pc.rf}, b1{c.cx, c.cy, 0.5 * lk.lx, 0.5 * lk.ly};
            const Box b2{nx0, ny, pc.rf, pc.rf};
            Corners<G, 16> cs;
            cs.eval(m.g, g, b0, b1, b2, b0, 0x8u);
            const BBox bb = cs.template bbox<0>(g);
            const BBox rbox = cs.template bbox<1>(g);
            dbox = cs.template bbox<2>(g);
            c.ici = cs.template get<12>(g);
            c.icj = cs.template get<13>(g);
            const bool rectWithin = cs.box_within(1);
            const Submap sm = submap_from_corners(m.g, rbox, rectWithin, c.cx, c.cy);
            stamp(pc, cyc, 2);
            DefaultDisc dflt;
            dflt.want = (out.default_next != nullptr && centre_usable(nx0, ny)) ? 1 : 0;
            dflt.x = nx0;
            dflt.y = ny;
            dflt.bb = dbox;
            dflt.z = static_cast<float>(static_cast<double>(0.0f) + pc.h);  // value when no cell is visited
            search_leg<G, true, kMid>(m, pc, lut, head, c, lk, tile, g, bb, sm, dflt, no, co);
            zDefault = dflt.z;
            haveDbox = true;
            stamp(pc, cyc, 8);
        }
        if (out.default_next && !haveDbox) {  // cpp:2289-2301 (leg search skipped: radius / centre unusable)
            if (!centre_usable(nx0, ny)) {
                zDefault = static_cast<float>(static_cast<double>(0.0f) + pc.h);  // no cell visited
            } else {
                dbox = circle_bbox_fast(m.g, nx0, ny, pc.rf);
                bool unused;
                zDefault = disc_pass_direct<G, false>(m, pc, nx0, ny, dbox, g, unused, reinterpret_cast<float*>(tile));
            }
        }
        if constexpr (kDirect) {
            lc->valid = no.valid;
            lc->v[0][0] = nx0;   lc->v[0][1] = ny;    lc->v[0][2] = static_cast<double>(zDefault);
            lc->v[1][0] = co.x;  lc->v[1][1] = co.y;  lc->v[1][2] = static_cast<double>(co.z);
            lc->v[2][0] = no.x;  lc->v[2][1] = no.y;  lc->v[2][2] = static_cast<double>(no.z);
        }
        if (g.sub == 0) {
            if constexpr (!kDirect) {
                sh.valid[leg] = no.valid;
                sh.nxt[0][leg][0] = nx0;   sh.nxt[0][leg][1] = ny;    sh.nxt[0][leg][2] = static_cast<double>(zDefault);
                sh.nxt[1][leg][0] = co.x;  sh.nxt[1][leg][1] = co.y;  sh.nxt[1][leg][2] = static_cast<double>(co.z);
                sh.nxt[2][leg][0] = no.x;  sh.nxt[2][leg][1] = no.y;  sh.nxt[2][leg][2] = static_cast<double>(no.z);
            }
            if (live) {
                const size_t o = (static_cast<size_t>(b) * nCycles + cyc) * 4 + leg;
                if (out.nominal) store_foothold(out.nominal + o, no, leg, cyc);
                store_selected<false>(out, o, no.row, no.col, no.z, no.valid, no.source, leg, cyc);
                if (out.centroid) {
                    fpe_centroid_foothold cf;
                    cf.x = co.x; cf.y = co.y; cf.z = co.z; cf.row = co.row; cf.col = co.col;
                    cf.code = static_cast<uint8_t>(co.code); cf.pad[0] = cf.pad[1] = cf.pad[2] = 0;
                    store_record<false>(out.centroid + o, cf);
                }
                if (out.default_next) {
                    store_record<false>(out.default_next + o * 3 + 0, static_cast<double>(nx0));
                    store_record<false>(out.default_next + o * 3 + 1, static_cast<double>(ny));
                    store_record<false>(out.default_next + o * 3 + 2, static_cast<double>(zDefault));
                }
            }
        }
}

}  // namespace

// ---- chained plan kernel ------------------------------------------------------------------------------
// G lanes per leg; a pose owns 4*G consecutive threads; block = max(64, 4*G) threads holds
// PPB = blockDim / (4*G) poses.  Dynamic LDS per pose = sizeof(PoseShared) + 4 * tileBytes.
#ifndef FPE_MINWAVES
#define FPE_MINWAVES 4
#endif
template <int G, bool kMid = false>
__global__ __launch_bounds__(G == 64 ? 256 : 64, G == 64 ? 4 : (G == 16 ? FPE_MINWAVES : 2)) void plan_chained_kernel(DevMap mArg, PlanConsts pc, SpiralLut lut,
                                                                           const fpe_pose* __restrict__ poses, int B,
                                                                           int nCycles, fpe_plan_out out) {
    // The map geometry is wave-uniform and would live in 20 scalar registers; this kernel needs more uniform
    // state than the 102 SGPRs hold, and every spilled SGPR costs a v_readlane (+ wait states) per use.  The
    // doubles are only ever operands of vector f64 arithmetic, so they are parked in VGPRs instead.
    DevMap m = mArg;
    if constexpr (G <= 16) {
        m.g.res = in_vgpr(m.g.res);
        m.g.rinv = in_vgpr(m.g.rinv);
        m.g.lenX = in_vgpr(m.g.lenX);
        m.g.lenY = in_vgpr(m.g.lenY);
        m.g.posX = in_vgpr(m.g.posX);
        m.g.posY = in_vgpr(m.g.posY);
        m.g.orgX = in_vgpr(m.g.orgX);
        m.g.orgY = in_vgpr(m.g.orgY);
        m.g.baseX = in_vgpr(m.g.baseX);
        m.g.baseY = in_vgpr(m.g.baseY);
    }
    constexpr int kPoseThreads = 4 * G;
    constexpr int kBlock = (G == 64) ? 256 : 64;
    constexpr int kPPB = kBlock / kPoseThreads;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = static_cast<int>(threadIdx.x);
    const int slot = tid / kPoseThreads;
    const int leg = (tid / G) & 3;
    const Grp<G> g(tid);
    const int tileBytes = tile_total_bytes(pc);
    const size_t poseBytes = sizeof(PoseShared) + 4 * static_cast<size_t>(tileBytes);
    unsigned char* base = smem + static_cast<size_t>(slot) * poseBytes;
    PoseShared& sh = *reinterpret_cast<PoseShared*>(base);
    uint8_t* tile = base + sizeof(PoseShared) + static_cast<size_t>(leg) * tileBytes;

    int b = blockIdx.x * kPPB + slot;
    const bool live = b < B;  // padding poses of the last block run the chain on pose B-1, store nothing
    if (!live) b = B - 1;

    const fpe_pose* pp = poses + b;
    const double x0 = pp->position[0], y0 = pp->position[1], z0 = pp->position[2];
    const int gait = pp->gait;
    const LegStatic ls = make_leg_static(pc, pp, leg, m.g.res, lut);
    const LutHead head = load_lut_head(lut, g);

    for (int k = tid % kPoseThreads; k < pc.nFoot; k += kPoseThreads) {
        sh.footDa[k] = pc.footDa[k];
        sh.footDb[k] = pc.footDb[k];
        sh.footOff[k] = static_cast<int16_t>(pc.footDa[k] * pc.tileW + pc.footDb[k]);
    }
    // initial stance (cpp:350-378) and first-gait shift (setFirstGait, cpp:2679-2699)
    if (g.sub == 0) {
        double sx = (leg == 0 || leg == 3) ? pc.LbHalf : -pc.LbHalf;
        double sy = (leg <= 1) ? pc.WbHalfNeg : pc.WbHalfPos;
        double sz = 0;
        sx += x0;
        sy += y0;
        sz += z0;
        if (out.stance && live) {
            double* st = out.stance + (static_cast<size_t>(b) * 4 + leg) * 3;
            st[0] = sx;
            st[1] = sy;
            st[2] = sz;
        }
        for (int t = 0; t < 3; ++t) {
            sh.cur[t][leg][0] = sx - pc.stepHalf;
            sh.cur[t][leg][1] = sy;
            sh.cur[t][leg][2] = sz;
        }
    }
    pose_sync<G>();
    if (out.pose_status && live && leg == 0 && g.sub == 0)
        out.pose_status[b] = opt_gate_cycle0(m.g, pc, polygon_center_x(sh.cur[0]), y0);

    double adjY = 0.0;  // ajustedPose_[1], cpp:759
    const int nPhases = (gait == 1) ? 4 : 1;
    const double advance = (gait == 1) ? pc.stepQuarter : pc.step;
    // swing order LF,RH,RF,LH (RF_FIRST=false) or RF,LH,LF,RH (build-defined walk)
    const int walkOrder = pc.RF_FIRST ? ((0) | (2 << 2) | (3 << 4) | (1 << 6)) : ((3) | (1 << 2) | (0 << 4) | (2 << 6));

    for (int cyc = 0; cyc < nCycles; ++cyc) {
        bool cycleOk = true;
        for (int ph = 0; ph < nPhases; ++ph) {
            const unsigned mask = (gait == 1) ? (1u << ((walkOrder >> (2 * ph)) & 3)) : 0xFu;
            const bool active = (mask >> leg) & 1u;

            stamp(pc, cyc, 0);
            // feet-polygon centres: group t computes track t (getPolygonCenter, cpp:2191, 2265)
            if (leg < 3 && g.sub == 0) {
                sh.ctr[leg] = polygon_center_x(sh.cur[leg]);
            }
            pose_sync<G>();
            stamp(pc, cyc, 1);

            bool phaseOk;
            if constexpr (G <= 16) {
                // a pose lives in one wavefront: footholdValidation_ (cpp:1323) is a ballot over its lanes and the
                // committed positions go from registers straight to PoseShared::cur (cpp:1332-1576)
                LegCommit lc;
                lc.valid = 1;  // non-swing legs do not vote
                if (active) {
                    leg_phase<G, kMid, true>(m, pc, lut, head, sh, tile, g, leg, ls, y0, adjY, advance, cyc, nCycles, b, live, out, &lc);
                }
                stamp(pc, cyc, 9);
                constexpr int kPoseLanes = 4 * G;
                const unsigned long long poseMask = (kPoseLanes == 64) ? ~0ull : (((1ull << (kPoseLanes & 63)) - 1ull) << (slot * kPoseLanes));
                phaseOk = (__ballot(lc.valid == 0) & poseMask) == 0ull;
                if (phaseOk && active && g.sub == 0) {
#pragma unroll
                    for (int t = 0; t < 3; ++t)
#pragma unroll
                        for (int k = 0; k < 3; ++k) sh.cur[t][leg][k] = lc.v[t][k];
                }
            } else {
                if (active) {
                    leg_phase<G, kMid>(m, pc, lut, head, sh, tile, g, leg, ls, y0, adjY, advance, cyc, nCycles, b, live, out);
                } else if (g.sub == 0) {
                    sh.valid[leg] = 1;  // non-swing legs do not vote
                }
                pose_sync<G>();
                stamp(pc, cyc, 9);
                // footholdValidation_ = AND of the swing legs' flags (cpp:1323); commit or skip (cpp:1332-1576)
                phaseOk = (sh.valid[0] & sh.valid[1] & sh.valid[2] & sh.valid[3]) != 0;
                if (phaseOk && active) {
                    for (int e = g.sub; e < 9; e += G) {
                        const int t = e / 3, k = e - t * 3;
                        sh.cur[t][leg][k] = sh.nxt[t][leg][k];
                    }
                }
            }
            pose_sync<G>();
            cycleOk = cycleOk && phaseOk;
            stamp(pc, cyc, 10);
        }
        if (leg == 0 && g.sub == 0 && live && out.cycle_ok)
            out.cycle_ok[static_cast<size_t>(b) * nCycles + cyc] = cycleOk ? 1 : 0;
        adjY += pc.drift;  // cpp:1578
    }
}

// ---- chained plan, sequential-legs form (large windows) ------------------------------------------------------
// One wavefront per pose; the swing legs of a phase are searched one after the other, each with all
// 64 lanes.  For large spiral windows (1 cm / 0.5 cm maps) a leg has enough cells and candidates to
// fill a wavefront, the per-leg geometry becomes truly wave-uniform, there are no workgroup barriers
// and — for the 4-phase walk gait, where only one leg swings per phase — no idle wavefronts.
__global__ __launch_bounds__(64, 4) void plan_sequential_kernel(DevMap m, PlanConsts pc, SpiralLut lut,
                                                                const fpe_pose* __restrict__ poses, int B, int nCycles,
                                                                fpe_plan_out out) {
    constexpr int G = 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = static_cast<int>(threadIdx.x);
    const Grp<G> g(tid);
    PoseShared& sh = *reinterpret_cast<PoseShared*>(smem);
    uint8_t* tile = smem + sizeof(PoseShared);
    const int b = blockIdx.x;
    if (b >= B) return;
    const bool live = true;

    const fpe_pose* pp = poses + b;
    const double x0 = pp->position[0], y0 = pp->position[1], z0 = pp->position[2];
    const int gait = pp->gait;
    const LutHead head = load_lut_head(lut, g);
    for (int k = tid; k < pc.nFoot; k += G) {
        sh.footDa[k] = pc.footDa[k];
        sh.footDb[k] = pc.footDb[k];
        sh.footOff[k] = static_cast<int16_t>(pc.footDa[k] * pc.tileW + pc.footDb[k]);
    }
    // initial stance (cpp:350-378) and first-gait shift (setFirstGait, cpp:2679-2699): lane = leg
    if (tid < 4) {
        const int leg = tid;
        double sx = (leg == 0 || leg == 3) ? pc.LbHalf : -pc.LbHalf;
        double sy = (leg <= 1) ? pc.WbHalfNeg : pc.WbHalfPos;
        double sz = 0;
        sx += x0;
        sy += y0;
        sz += z0;
        if (out.stance) {
            double* st = out.stance + (static_cast<size_t>(b) * 4 + leg) * 3;
            st[0] = sx;
            st[1] = sy;
            st[2] = sz;
        }
        for (int t = 0; t < 3; ++t) {
            sh.cur[t][leg][0] = sx - pc.stepHalf;
            sh.cur[t][leg][1] = sy;
            sh.cur[t][leg][2] = sz;
        }
    }
    pose_sync<16>();
    if (out.pose_status && tid == 0) out.pose_status[b] = opt_gate_cycle0(m.g, pc, polygon_center_x(sh.cur[0]), y0);

    double adjY = 0.0;  // ajustedPose_[1], cpp:759
    const int nPhases = (gait == 1) ? 4 : 1;
    const double advance = (gait == 1) ? pc.stepQuarter : pc.step;
    const int walkOrder = pc.RF_FIRST ? ((0) | (2 << 2) | (3 << 4) | (1 << 6)) : ((3) | (1 << 2) | (0 << 4) | (2 << 6));

    for (int cyc = 0; cyc < nCycles; ++cyc) {
        bool cycleOk = true;
        for (int ph = 0; ph < nPhases; ++ph) {
            const unsigned mask = (gait == 1) ? (1u << ((walkOrder >> (2 * ph)) & 3)) : 0xFu;
            stamp(pc, cyc, 0);
            // feet-polygon centres: lane t computes track t (getPolygonCenter, cpp:2191, 2265)
            if (tid < 3) {
                sh.ctr[tid] = polygon_center_x(sh.cur[tid]);
            }
            if (tid < 4) sh.valid[tid] = 1;  // non-swing legs do not vote
            pose_sync<16>();
            stamp(pc, cyc, 1);
            for (int leg = 0; leg < 4; ++leg) {
                if (!((mask >> leg) & 1u)) continue;
                const LegStatic ls = make_leg_static(pc, pp, leg, m.g.res, lut);
                stamp(pc, cyc, 13);
                leg_phase<G>(m, pc, lut, head, sh, tile, g, leg, ls, y0, adjY, advance, cyc, nCycles, b, live, out);
                stamp(pc, cyc, 14);
            }
            pose_sync<16>();
            stamp(pc, cyc, 9);
            // footholdValidation_ = AND of the swing legs' flags (cpp:1323); commit or skip (cpp:1332-1576)
            const bool phaseOk = (sh.valid[0] & sh.valid[1] & sh.valid[2] & sh.valid[3]) != 0;
            if (phaseOk && tid < 36) {
                const int leg = tid / 9, e = tid - leg * 9;
                if ((mask >> leg) & 1u) {
                    const int t = e / 3, k = e - t * 3;
                    sh.cur[t][leg][k] = sh.nxt[t][leg][k];
                }
            }
            pose_sync<16>();
            cycleOk = cycleOk && phaseOk;
            stamp(pc, cyc, 10);
        }
        if (tid == 0 && out.cycle_ok) out.cycle_ok[static_cast<size_t>(b) * nCycles + cyc] = cycleOk ? 1 : 0;
        adjY += pc.drift;  // cpp:1578
    }
}

// ---- open-loop per-leg search: one wavefront per checkFoothold call (hpp:94-100) ----------------------
constexpr int kMaxQueriesPerBlock = 32;  // 256 threads / 8 lanes per query
struct QueryShared {
    double polyX[kMaxQueriesPerBlock][8];
    double polyY[kMaxQueriesPerBlock][8];
    int8_t footDa[kMaxFootOffsets];
    int8_t footDb[kMaxFootOffsets];
    int16_t footOff[kMaxFootOffsets];
};

// G lanes per query (8 for windows of <= 1024 cells, like the chained plan; else a whole wavefront), 256 / G queries
// per workgroup.
template <int G, bool kMid = false>
__global__ __launch_bounds__(256) void search_legs_kernel(DevMap m, PlanConsts pc, SpiralLut lut,
                                                           const fpe_leg_query* __restrict__ queries, int n,
                                                           fpe_foothold* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    QueryShared& sh = *reinterpret_cast<QueryShared*>(smem);
    const int tid = static_cast<int>(threadIdx.x);
    const int w = tid / G;
    const Grp<G> g(tid);
    const int q = blockIdx.x * (256 / G) + w;
    const int tileBytes = tile_total_bytes(pc);
    uint8_t* tile = smem + sizeof(QueryShared) + static_cast<size_t>(w) * tileBytes;
    for (int k = tid & 63; k < pc.nFoot; k += 64) {  // every wavefront writes the same bytes, then reads its own writes
        sh.footDa[k] = pc.footDa[k];
        sh.footDb[k] = pc.footDb[k];
        sh.footOff[k] = static_cast<int16_t>(pc.footDa[k] * pc.tileW + pc.footDb[k]);
    }
    if (q >= n) return;

    const fpe_leg_query* qp = queries + q;
    const int nv = qp->n_vertices;
    if (g.sub < 8) {
        sh.polyX[w][g.sub] = g.sub < nv ? qp->vx[g.sub] : 0.0;
        sh.polyY[w][g.sub] = g.sub < nv ? qp->vy[g.sub] : 0.0;
    }
    pose_sync<16>();  // the vertices are read by the other lanes of the group (same wavefront)
    const float Rf = qp->search_radius;
    LegCtx c;
    c.cx = qp->cx;
    c.cy = qp->cy;
    c.nv = nv;
    c.cyc = 99;
    c.rect = false;
    c.xlo = c.xhi = c.ylo = c.yhi = 0.0;
    c.vx = sh.polyX[w];
    c.vy = sh.polyY[w];
    c.footDa = sh.footDa;
    c.footDb = sh.footDb;
    c.footOff = sh.footOff;
    NominalOut no;
    CentroidOut co;
    if (!centre_usable(c.cx, c.cy)) {
        nominal_invalid(no, c.cx, c.cy, 2);
    } else if (Rf <= pc.maxSearchRadius && Rf >= 0.0f && nv >= 0 && nv <= FPE_MAX_POLYGON_VERTICES) {
        const LegConst lk = make_leg_const(Rf, m.g.res, lut);
        const LutHead head = load_lut_head(lut, g);
        const Box b0{c.cx, c.cy, pc.rf, pc.rf};
        Corners<G, 8> cs;
        cs.eval(m.g, g, b0, b0, b0, b0, 0x2u);  // quantities 4,5 = getIndex(centre)
        const BBox bb = cs.template bbox<0>(g);
        c.ici = cs.template get<4>(g);
        c.icj = cs.template get<5>(g);
        Submap sm;
        sm.ok = false;
        DefaultDisc dflt;
        dflt.want = 0;
        search_leg<G, false, kMid>(m, pc, lut, head, c, lk, tile, g, bb, sm, dflt, no, co);
    } else {
        nominal_invalid(no, c.cx, c.cy, 3);
    }
    if (g.sub == 0) store_foothold(out + q, no, 0, 0);
}

// ---- map ingest: grid_map_msgs layout -> canonical row-major start-index-0 layer --------------------
// src is column-major with circular-buffer start index (si, sj): unwrapped (i, j) lives at buffer
// index ((i + si) % rows, (j + sj) % cols) (grid_map getBufferIndexFromIndex).  64 x 64 tiles through
// LDS so that both the column-major reads and the row-major writes are coalesced (round 4: 32 x 32 tiles moved 128 bytes
// per wavefront instruction and 3.95 TB/s on two 4000 x 4000 layers; 64 x 64 with sixteen loads in flight per thread 4.6,
// 16-byte stores where the destination's rows allow them 4.7 — the plain wrapped copy of a row-major source runs at 4.5).
// Bit planes of the canonical layer while it passes through (round 5; SURVEY 8(f) N1 + the planes of fpe_bits.hpp): with
// `planes.words` set, every wavefront ballots the 64 columns of a destination row it holds anyway — lane = column — and lanes 0 / 1
// store the row's two word groups {D, Df, C, F}: the separate pass of build_bitmap_kernel over the traversability layer (13 us for
// 4000 x 4000) is gone.  Same predicates, same layout (bit_group_index) as build_bitmap_kernel; the buffer's padding is zeroed by the
// caller.  T = 64 only (a wavefront = one tile row).
struct CanonPlanes {
    uint4* words;  // null: no planes
    int strideW, nw;
    float thrD, thrC;
};
template <int T>
__device__ __forceinline__ void canon_plane_row(const CanonPlanes& pl, float v, bool in, int i, int rows, int jBase) {
    static_assert(T == 64, "a wavefront ballots one tile row of 64 columns");
    const bool fin = in && __builtin_isfinite(v);
    const bool d = in && v < pl.thrD;  // raw compare: NaN -> false, -inf -> true (cpp:1653, 1736)
    const bool c = fin && v < pl.thrC;
    const unsigned long long bD = __ballot(d), bDf = __ballot(d && fin), bC = __ballot(c), bF = __ballot(fin);
    const int lane = static_cast<int>(threadIdx.x) & 63;
    if (lane < 2 && i < rows) {
        const int w = jBase / 32 + lane;
        if (w < pl.nw) {
            uint4 o;
            o.x = static_cast<unsigned>(bD >> (32 * lane));
            o.y = static_cast<unsigned>(bDf >> (32 * lane));
            o.z = static_cast<unsigned>(bC >> (32 * lane));
            o.w = static_cast<unsigned>(bF >> (32 * lane));
            pl.words[bit_group_index(i, w, pl.strideW)] = o;
        }
    }
}
template <int T>
__global__ __launch_bounds__(256) void canonicalise_layer_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                                  int rows, int cols, int si, int sj, int srcRowMajor, CanonPlanes planes) {
    // T x T tiles, a wavefront instruction moves T consecutive floats of one source column / destination row (T = 64: 256
    // bytes); every thread has T * T / 256 loads in flight before the first LDS store
    __shared__ float t[T][T + 1];
    constexpr int kPer = T * T / 256;  // tile elements per thread
    const int tx = threadIdx.x & (T - 1), ty = threadIdx.x / T;  // T x (256 / T)
    constexpr int kStep = 256 / T;
    const int iBase = blockIdx.y * T, jBase = blockIdx.x * T;
    if (srcRowMajor) {
#pragma unroll
        for (int q = 0; q < kPer; ++q) {
            const int i = iBase + ty + kStep * q, j = jBase + tx;
            const bool in = i < rows && j < cols;
            float v = 0.0f;
            if (in) {
                int bi = i + si; if (bi >= rows) bi -= rows;
                int bj = j + sj; if (bj >= cols) bj -= cols;
                v = src[static_cast<size_t>(bi) * cols + bj];
                dst[static_cast<size_t>(i) * cols + j] = v;
            }
            if constexpr (T == 64) {
                if (planes.words) canon_plane_row<T>(planes, v, in, i, rows, jBase);  // (wave-uniform branch; every lane ballots)
            }
        }
        return;
    }
    // column-major source: consecutive threads walk i (contiguous in src)
    float v[kPer];
#pragma unroll
    for (int q = 0; q < kPer; ++q) {
        const int i = iBase + tx, j = jBase + ty + kStep * q;
        v[q] = 0.0f;
        if (i < rows && j < cols) {
            int bi = i + si; if (bi >= rows) bi -= rows;
            int bj = j + sj; if (bj >= cols) bj -= cols;
            v[q] = src[static_cast<size_t>(bi) + static_cast<size_t>(bj) * rows];
        }
    }
#pragma unroll
    for (int q = 0; q < kPer; ++q) t[ty + kStep * q][tx] = v[q];
    __syncthreads();
    if constexpr (T == 64) {
        if (planes.words) {  // the tile once more, lane = column of a destination row (t[column][row]: conflict-free reads)
#pragma unroll 4
            for (int q = 0; q < kPer; ++q) {
                const int r = ty + kStep * q, i = iBase + r, j = jBase + tx;
                canon_plane_row<T>(planes, t[tx][r], i < rows && j < cols, i, rows, jBase);
            }
        }
    }
#ifndef FPE_CANON_SCALAR_STORE
    if ((cols & 3) == 0) {  // rows of the destination start 16-byte aligned: four columns per thread, one 16-byte store
        constexpr int kGroups = T / 4, kRowsPerPass = 256 / kGroups;
        const int g4 = threadIdx.x % kGroups, r0 = threadIdx.x / kGroups;
#pragma unroll
        for (int q = 0; q < T / kRowsPerPass; ++q) {
            const int r = r0 + kRowsPerPass * q, i = iBase + r, j = jBase + 4 * g4;
            if (i < rows && j < cols) {  // (cols % 4 == 0: j + 3 < cols as well)
                float4 o;
                o.x = t[4 * g4 + 0][r]; o.y = t[4 * g4 + 1][r]; o.z = t[4 * g4 + 2][r]; o.w = t[4 * g4 + 3][r];
                *reinterpret_cast<float4*>(dst + static_cast<size_t>(i) * cols + j) = o;
            }
        }
        return;
    }
#endif
#pragma unroll
    for (int q = 0; q < kPer; ++q) {
        const int i = iBase + ty + kStep * q, j = jBase + tx;
        if (i < rows && j < cols) dst[static_cast<size_t>(i) * cols + j] = t[tx][ty + kStep * q];
    }
}

// ---- launch wrappers (called from fpe_engine.cpp) ---------------------------------------------------------
static size_t tile_bytes(const PlanConsts& pc) { return static_cast<size_t>(tile_total_bytes(pc)); }

// lanes per leg for a tile of tileW^2 cells: small windows (2 cm maps) put a whole pose in one
// wavefront; large windows give every leg its own wavefront
// lanes per leg: 8 for small windows (two poses per wavefront); large windows use the sequential-legs
// kernel (code 65: one wavefront per pose, 64 lanes per leg, legs in sequence).  64 = one wavefront per
// leg, four per pose (kept for measurement).
int plan_group_size(const PlanConsts& pc) {
    if (pc.groupOverride == 4 || pc.groupOverride == 8 || pc.groupOverride == 16 || pc.groupOverride == 64 ||
        pc.groupOverride == 65)
        return pc.groupOverride;
    return (pc.tileW * pc.tileW <= 1024) ? 8 : 65;
}
size_t plan_lds_bytes(const PlanConsts& pc) {
    const int G = plan_group_size(pc);
    if (G == 65) return sizeof(PoseShared) + tile_bytes(pc);  // sequential legs share one tile
    const int ppb = G >= 16 ? 1 : 64 / (4 * G);  // poses per 64-thread block
    return ppb * (sizeof(PoseShared) + 4 * tile_bytes(pc));
}
static int search_group_size(const PlanConsts& pc) { return (pc.tileW * pc.tileW <= 1024 && pc.groupOverride != 64) ? 8 : 64; }
size_t search_lds_bytes(const PlanConsts& pc) {
    return sizeof(QueryShared) + static_cast<size_t>(256 / search_group_size(pc)) * tile_bytes(pc);
}

// The 3x3-only variant of the 8-lane kernel: foot radius in [0.9, 1] x resolution (a disc box then spans at most
// three cells per axis) with a host-proved candidate foot disc of at most four cells (no LDS window: the staged
// spiral path is not compiled into it), e.g. the reference's yaml footRadius 0.02 on a 2 cm map.
// fpe_set_tuning("no_mid_variant", 1) keeps the generic kernel (tests run both).
static bool mid_variant(const PlanConsts& pc, double res) {
    return pc.midCellInside != 0 && pc.rf <= res && pc.footRobust != 0 && pc.nFoot <= kOnDemandMaxFoot &&
           pc.noMidVariant == 0;
}

hipError_t launch_plan_chained(const DevMap& m, const PlanConsts& pc, const SpiralLut& lut, const fpe_pose* d_poses,
                               int B, int nCycles, const fpe_plan_out& d_out, hipStream_t stream) {
    const size_t lds = plan_lds_bytes(pc);
    const int G = plan_group_size(pc);
    if (G == 65) {
        hipLaunchKernelGGL(plan_sequential_kernel, dim3(B), dim3(64), lds, stream, m, pc, lut, d_poses, B, nCycles, d_out);
    } else if (G == 4) {
        hipLaunchKernelGGL(plan_chained_kernel<4>, dim3((B + 3) / 4), dim3(64), lds, stream, m, pc, lut, d_poses, B, nCycles, d_out);
    } else if (G == 8 && mid_variant(pc, m.g.res)) {
        hipLaunchKernelGGL((plan_chained_kernel<8, true>), dim3((B + 1) / 2), dim3(64), lds, stream, m, pc, lut, d_poses, B, nCycles, d_out);
    } else if (G == 8) {
        hipLaunchKernelGGL(plan_chained_kernel<8>, dim3((B + 1) / 2), dim3(64), lds, stream, m, pc, lut, d_poses, B, nCycles, d_out);
    } else if (G == 16) {
        hipLaunchKernelGGL(plan_chained_kernel<16>, dim3(B), dim3(64), lds, stream, m, pc, lut, d_poses, B, nCycles, d_out);
    } else {
        hipLaunchKernelGGL(plan_chained_kernel<64>, dim3(B), dim3(256), lds, stream, m, pc, lut, d_poses, B, nCycles, d_out);
    }
    return hipGetLastError();
}

hipError_t launch_search_legs(const DevMap& m, const PlanConsts& pc, const SpiralLut& lut, const fpe_leg_query* d_q,
                              int n, fpe_foothold* d_out, hipStream_t stream) {
    const size_t lds = search_lds_bytes(pc);
    if (search_group_size(pc) == 8 && mid_variant(pc, m.g.res))
        hipLaunchKernelGGL((search_legs_kernel<8, true>), dim3((n + 31) / 32), dim3(256), lds, stream, m, pc, lut, d_q, n, d_out);
    else if (search_group_size(pc) == 8)
        hipLaunchKernelGGL(search_legs_kernel<8>, dim3((n + 31) / 32), dim3(256), lds, stream, m, pc, lut, d_q, n, d_out);
    else
        hipLaunchKernelGGL(search_legs_kernel<64>, dim3((n + 3) / 4), dim3(256), lds, stream, m, pc, lut, d_q, n, d_out);
    return hipGetLastError();
}

size_t bitmap_words(int rows, int cols, int* strideW, int* nw);  // fpe_bits.hpp (part two of this translation unit)
// d_planeWords: also build the bit planes of the destination layer for (thrDefault, thrCandidate) — the buffer zeroed here first
// (padding rows / word groups; recycled buffers are dirty) — or null.
hipError_t launch_canonicalise(const float* d_src, float* d_dst, int rows, int cols, int si, int sj, int srcRowMajor,
                               hipStream_t stream, uint32_t* d_planeWords, float thrDefault, float thrCandidate) {
#ifndef FPE_CANON_TILE
#define FPE_CANON_TILE 64
#endif
    CanonPlanes pl{nullptr, 0, 0, 0.0f, 0.0f};
    if (d_planeWords) {
        if (FPE_CANON_TILE != 64) return hipErrorInvalidValue;
        const size_t units = bitmap_words(rows, cols, &pl.strideW, &pl.nw);
        const hipError_t e = hipMemsetAsync(d_planeWords, 0, units * 4, stream);
        if (e != hipSuccess) return e;
        pl.words = reinterpret_cast<uint4*>(d_planeWords);
        pl.thrD = thrDefault;
        pl.thrC = thrCandidate;
    }
    dim3 grid((cols + FPE_CANON_TILE - 1) / FPE_CANON_TILE, (rows + FPE_CANON_TILE - 1) / FPE_CANON_TILE);
    hipLaunchKernelGGL(canonicalise_layer_kernel<FPE_CANON_TILE>, grid, dim3(256), 0, stream, d_src, d_dst, rows, cols, si, sj, srcRowMajor, pl);
    return hipGetLastError();
}

// ---- part two of this translation unit: the bit-window kernels ---------------------------------------------
#include "fpe_bits.hpp"
// ---- part three: the producer's filters (elevation -> traversability) ----------------------------------------
#include "fpe_filters.hpp"

#include "fpe_opt.hpp"

hipError_t set_max_lds(size_t planBytes, size_t searchBytes) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(plan_chained_kernel<16>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(planBytes));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(plan_chained_kernel<8>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(planBytes));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(plan_chained_kernel<8, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(planBytes));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(plan_chained_kernel<4>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(planBytes));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(plan_chained_kernel<64>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(planBytes));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(plan_sequential_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(planBytes));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(search_legs_kernel<8>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(searchBytes));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(search_legs_kernel<8, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(searchBytes));
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(search_legs_kernel<64>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(searchBytes));
}

}  // namespace fpe

#ifdef FPE_FUSED_TIMELINE
extern "C" int fpe_debug_timeline(void* out, size_t bytes) {
    return static_cast<int>(hipMemcpyFromSymbol(out, HIP_SYMBOL(fpe::g_fusedTimeline), bytes));
}
#endif
#ifdef FPE_OPT_TRACE
extern "C" int fpe_debug_opt_trace(void* out, size_t bytes) {
    return static_cast<int>(hipMemcpyFromSymbol(out, HIP_SYMBOL(fpe::g_optTrace), bytes));
}
#endif
#ifdef FPE_DBG_COUNT_WALKS
extern "C" int fpe_debug_walk_dbg(double* out) { return static_cast<int>(hipMemcpyFromSymbol(out, HIP_SYMBOL(fpe::g_walkDbg), 64 * 8 * 8)); }
extern "C" int fpe_debug_walk_counts(unsigned* out, int reset) {
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(fpe::g_walkWhy), 16);
    if (reset) { unsigned z[4] = {0, 0, 0, 0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(fpe::g_walkWhy), z, 16); }
    return static_cast<int>(e);
}
#endif

