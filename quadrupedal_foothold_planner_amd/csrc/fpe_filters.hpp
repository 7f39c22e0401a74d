// fpe_filters.hpp — part three of the fpe_kernels.hip translation unit: the PRODUCER of the path's input
// (SURVEY.md §8(f) N3), elevation layer -> traversability layer, as disc stencils on the canonical map in HBM.
//
// The reference starts the producer and subscribes to its output (launch/mapping.launch:12-13, launch/all.launch:21-22,
// FootholdPlanner.cpp:188); the package itself (leggedrobotics/traversability_estimation, README.md:29) and its filter
// configuration are not under /root/reference and no version is pinned, so this is a restatement of that package's
// published default chain (grid_map_filters NormalVectorsFilter, area method; traversability_estimation_filters
// SlopeFilter, StepFilter, RoughnessFilter; the (1/3)-weighted MathExpressionFilter).  PARITY UNPINNED; the arithmetic
// contract is oracle/fpo_filters.cpp (same f64 expression order, float layers, filters read the rounded normals back).
//
// Every filter is a walk over CircleIterator(cell centre, radius) per cell.  One workgroup = one 16 x 16 tile of cells,
// one cell per lane.  The source tile plus a halo of H = floor(r / res) + 1 cells goes through LDS once; the iterator's
// per-axis quantities — the bounding rows / columns of findSubmapParameters and the squared centre distances of
// isInside — depend on (cell row, row offset) and (cell column, column offset) only and are tabulated in LDS by the
// workgroup, so a visit costs one LDS read, one f64 add and one compare before the filter's own arithmetic.
// Bound: instruction issue (loop control, LDS reads and the f64 arithmetic of the published filters — points and a
// 3 x 3 covariance per cell; profiles/round3_filters.txt), not HBM: 4 B read and
// 4 B written per cell and layer against some thousand f64 operations per cell.
// (included inside namespace fpe of fpe_kernels.hip, like fpe_bits.hpp)
#pragma once

namespace {

#ifndef FPE_FUSED_WAVES
#define FPE_FUSED_WAVES 6
#endif
#ifndef FPE_FUSED_WAVES_2CM  // the published chain at 2 cm: four workgroups per CU (64 registers, 21 dwords of scratch in the walk phase): 0.056 -> 0.053 ms; at 1 cm the same trade loses (0.274 -> 0.278)
#define FPE_FUSED_WAVES_2CM 8
#endif
#ifdef FPE_FUSED_TIMELINE  // measurement builds only (scratch/): per-workgroup clock marks of the fused kernel
__device__ unsigned long long g_fusedTimeline[8192][16];
#define FPE_TL_MARK(k) do { if (threadIdx.x == 0 && blockIdx.x < 8192) g_fusedTimeline[blockIdx.x][k] = wall_clock64(); } while (0)
#else
#define FPE_TL_MARK(k) do { } while (0)
#endif
constexpr int kFT = 16;        // tile edge of the disc stencils
constexpr int kFilterMaxH = 24;  // largest halo the tables are sized for (e.g. r 0.115 m at 0.5 cm)

struct DiscLds {
    float* tile;        // (TR + 2H) x (TC + 2H) source cells, row-major, NaN outside the map
    double *xP, *yP;    // positions of the tile's rows / columns (halo included)
    double *dx2, *dy2;  // [TR][2H + 1] / [TC][2H + 1] squared centre distance per axis
    int *bi0, *bi1, *bj0, *bj1;  // [TR] / [TC] bounding rows / columns of the cell's iterator
    int H, WR, WC;      // halo, tile rows and columns with the halo
};
// (TR x TC: the workgroup's cells — kFT x kFT for the walking kernels; 16 x 16, 32 x 16 (rows x columns) or 32 x 32 for the
// row-moment and row-run kernels)
__host__ __device__ inline size_t disc_lds_bytes(int H, int TR = kFT, int TC = kFT) {
    const int WR = TR + 2 * H, WC = TC + 2 * H;
    return static_cast<size_t>(WR) * WC * 4 + static_cast<size_t>(WR + WC) * 8 + static_cast<size_t>(TR + TC) * (2 * H + 1) * 8 + 2 * (TR + TC) * 4 + 16;
}
__device__ __forceinline__ DiscLds disc_carve(char* base, int H, int TR = kFT, int TC = kFT) {
    DiscLds d;
    d.H = H;
    d.WR = TR + 2 * H;
    d.WC = TC + 2 * H;
    char* p = base;
    d.xP = reinterpret_cast<double*>(p); p += d.WR * 8;
    d.yP = reinterpret_cast<double*>(p); p += d.WC * 8;
    d.dx2 = reinterpret_cast<double*>(p); p += TR * (2 * H + 1) * 8;
    d.dy2 = reinterpret_cast<double*>(p); p += TC * (2 * H + 1) * 8;
    d.bi0 = reinterpret_cast<int*>(p); p += TR * 4;
    d.bi1 = reinterpret_cast<int*>(p); p += TR * 4;
    d.bj0 = reinterpret_cast<int*>(p); p += TC * 4;
    d.bj1 = reinterpret_cast<int*>(p); p += TC * 4;
    d.tile = reinterpret_cast<float*>(p);
    return d;
}
// The tables of a tile: positions of its rows / columns (halo included), the iterator's bounding box per row / column of the
// workgroup's cells, and the squared per-axis centre distances.  No barrier.
template <int TR, int TC, int HS>
__device__ __forceinline__ void disc_tables(const DiscLds& d, const MapGeom& g, int ti0, int tj0, double r, bool axisTables = true) {
    const int H = HS > 0 ? HS : d.H, WR = TR + 2 * H, WC = TC + 2 * H, t = threadIdx.x, D = 2 * H + 1;
    constexpr int kThreads = TR * TC;
    if (axisTables) {
        for (int k = t; k < WR + WC; k += kThreads) {
            if (k < WR) d.xP[k] = cell_pos(g.baseX, g.res, ti0 - H + k);
            else d.yP[k - WR] = cell_pos(g.baseY, g.res, tj0 - H + (k - WR));
        }
    }
    if (t < TR + TC) {  // CircleIterator::findSubmapParameters per axis (circle_bbox), clamped onto the halo
        const bool isRow = t < TR;
        const int l = isRow ? t : t - TR;
        const int idx = (isRow ? ti0 : tj0) + l, n = isRow ? g.rows : g.cols;
        const double c = cell_pos(isRow ? g.baseX : g.baseY, g.res, idx);
        const double org = isRow ? g.orgX : g.orgY, pos = isRow ? g.posX : g.posY, len = isRow ? g.lenX : g.lenY;
        int a = index_of_fast(bound_axis(c + r, org, pos, len), org, pos, g.res, g.rinv);
        int b = index_of_fast(bound_axis(c - r, org, pos, len), org, pos, g.res, g.rinv);
        a = max(a, max(idx - H, 0));
        b = min(b, min(idx + H, n - 1));
        (isRow ? d.bi0 : d.bj0)[l] = a;
        (isRow ? d.bi1 : d.bj1)[l] = b;
    }
    if (axisTables) {
        for (int k = t; k < (TR + TC) * D; k += kThreads) {  // CircleIterator::isInside, per axis
            const bool isRow = k < TR * D;
            const int e = isRow ? k : k - TR * D;
            const int l = e / D, o = e % D;
            const double base = isRow ? g.baseX : g.baseY;
            const int first = (isRow ? ti0 : tj0) - H + l;
            const double dd = cell_pos(base, g.res, first + o) - cell_pos(base, g.res, first + H);
            (isRow ? d.dx2 : d.dy2)[e] = dd * dd;
        }
    }
}
// Tables and source tile of the workgroup's cells [ti0, ti0 + kFT) x [tj0, tj0 + kFT); ends with a barrier.
// One phase, one barrier: every thread first REQUESTS its share of the tile (a wavefront takes every fourth tile row,
// lane = tile column: no index division, up to 16 loads in flight per lane — the former one-load-one-store loop waited
// for each of its five to ten loads in turn, and the kernels spent most of a wavefront's life there), computes its
// table entries from cell_pos itself while the loads fly (the same values the position arrays hold), then stores.
// kTables false: the tile and the bounding boxes only (the step filter's row runs need no per-axis distance tables).
// tilesOnly (run time, wave-uniform): the tile alone — the caller knows that nothing on its hot path reads a table (the moment
// phase of a disc without an offset on the circle) and builds them later if a cell has to walk (disc_tables, walk_phase).
// kCoherent: the source was written by OTHER workgroups of this launch (filter_chain_kernel): device-scope loads, which do not hit
// a stale line of this XCD's caches.
template <bool kTables = true, int TR = kFT, int TC = kFT, int HS = 0, bool kCoherent = false>
__device__ __forceinline__ void disc_setup(const DiscLds& d, const MapGeom& g, const float* __restrict__ src, int ti0, int tj0, double r, bool tileOnly = false) {
    // HS > 0: the halo is a compile-time constant (every loop bound, the divisions by 2 H + 1 and the number of tile rows a
    // lane requests are then constants); 0: run time
    const int H = HS > 0 ? HS : d.H, WR = TR + 2 * H, WC = TC + 2 * H, t = threadIdx.x;
    constexpr int kThreads = TR * TC;  // (the launch makes sure that a tile row is at most one wavefront load: TC + 2 H <= 64)
    // a wavefront instruction loads one tile row (WC > 32) or two (lanes 0-31 / 32-63); up to eight rows in flight per lane
    const int perInst = WC <= 32 ? 2 : 1, laneCols = WC <= 32 ? 32 : 64;
    const int col = t & (laneCols - 1), sub = (t & 63) / laneCols, wv = t >> 6;
    const int tj = tj0 - H + col;
    const bool colOk = col < WC && tj >= 0 && tj < g.cols;
    const int rowStep = (kThreads / 64) * perInst;  // rows per workgroup instruction
    const int firstRow = wv * perInst + sub;
    // rows a lane has in flight: all of the tile's when the halo is a constant (at most eight), else eight per round
    constexpr int kStepC = (kThreads / 64) * ((TC + 2 * HS) <= 32 ? 2 : 1);
    constexpr int kNeed = ((TR + 2 * HS) + kStepC - 1) / kStepC;
    constexpr int kBatch = HS > 0 ? (kNeed < 8 ? kNeed : 8) : 8;
    float v[kBatch];
    const auto request = [&](int base) {
#pragma unroll
        for (int q = 0; q < kBatch; ++q) {
            const int row = base + firstRow + rowStep * q, ti = ti0 - H + row;
            v[q] = __builtin_nanf("");
            if (row < WR && colOk && ti >= 0 && ti < g.rows) {
                if constexpr (kCoherent) v[q] = __hip_atomic_load(&src[static_cast<size_t>(ti) * g.cols + tj], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else v[q] = src[static_cast<size_t>(ti) * g.cols + tj];
            }
        }
    };
    // invalid cells (GridMap::isValid = isfinite) enter the tile as quiet NaNs: the walks test `z == z`, and the float
    // max / min of the step filter skip them by themselves
    const auto deposit = [&](int base) {
#pragma unroll
        for (int q = 0; q < kBatch; ++q) {
            const int row = base + firstRow + rowStep * q;
            if (row < WR && col < WC) d.tile[row * WC + col] = isfinite(v[q]) ? v[q] : __builtin_nanf("");
        }
    };
    request(0);
    if (!tileOnly) disc_tables<TR, TC, HS>(d, g, ti0, tj0, r, kTables);
    deposit(0);
    for (int base = rowStep * kBatch; base < WR; base += rowStep * kBatch) {  // (tiles of more rows than a round requests)
        request(base);
        deposit(base);
    }
    __syncthreads();
}
// The cell's iterator walk, in CircleIterator order (rows outer, columns inner): f(x, y, value) per member cell.
// isInside is monotone in the column distance on either side of the centre column (the squared distances are rounded
// monotonically), so the members of a row are one interval; its ends follow from the previous row's by a few tests
// instead of one test per cell of the bounding box.  Same members, same order as testing every cell.
template <class F>
__device__ __forceinline__ void disc_walk(const DiscLds& d, int li, int lj, int ti0, int tj0, double r2, F&& f) {
    const int H = d.H, W = d.WC, D = 2 * H + 1;
    const int i = ti0 + li, j = tj0 + lj;
    const int i0 = d.bi0[li], i1 = d.bi1[li];
    const int maxL = j - d.bj0[lj], maxR = d.bj1[lj] - j;  // the bounding box's columns either side of the centre column
    const int dyC = lj * D + H, dxC = li * D + H - i;       // dy2[dyC + dj], dx2[dxC + ii]
    const int colBase = H - tj0;
    int wL = 0, wR = 0;
    for (int ii = i0; ii <= i1; ++ii) {
        const double a = d.dx2[dxC + ii];
        if (!(a <= r2)) continue;  // (a + 0 <= r2: the centre column, dy2 = 0 exactly)
        wR = min(wR, maxR);
        wL = min(wL, maxL);
        while (wR > 0 && !(a + d.dy2[dyC + wR] <= r2)) --wR;
        while (wR < maxR && a + d.dy2[dyC + wR + 1] <= r2) ++wR;
        while (wL > 0 && !(a + d.dy2[dyC - wL] <= r2)) --wL;
        while (wL < maxL && a + d.dy2[dyC - wL - 1] <= r2) ++wL;
        const int ri = ii - ti0 + H;
        const double x = d.xP[ri];
        const int rowBase = ri * W + colBase;
        int jj = j - wL;
        const int jEnd = j + wR;
        for (; jj + 3 <= jEnd; jj += 4) {  // four members' LDS reads in flight before the first is consumed
            const float z0 = d.tile[rowBase + jj], z1 = d.tile[rowBase + jj + 1], z2 = d.tile[rowBase + jj + 2], z3 = d.tile[rowBase + jj + 3];
            const double y0 = d.yP[colBase + jj], y1 = d.yP[colBase + jj + 1], y2 = d.yP[colBase + jj + 2], y3 = d.yP[colBase + jj + 3];
            f(x, y0, z0);
            f(x, y1, z1);
            f(x, y2, z2);
            f(x, y3, z3);
        }
        for (; jj <= jEnd; ++jj) f(x, d.yP[colBase + jj], d.tile[rowBase + jj]);
    }
}

// v_max_f32 / v_min_f32 return the other operand when one is a quiet NaN (IEEE mode); written as instructions because
// fmaxf / fminf are compiled with a canonicalising v_max(x, x) in front of every operand.
__device__ __forceinline__ float max_skip_nan(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float min_skip_nan(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// One Jacobi rotation of the symmetric 3 x 3 eigenproblem on the (p, q) pair; r is the third index.
__device__ __forceinline__ void jacobi_rotate(double& app, double& aqq, double& apq, double& arp, double& arq, double (&vp)[3], double (&vq)[3]) {
    if (apq == 0.0) return;
    if (fabs(apq) <= 2.3e-18 * (fabs(app) + fabs(aqq))) {  // a rotation by less than 2^-58: nothing moves in double precision
        apq = 0.0;
        return;
    }
    const double theta = (aqq - app) / (2.0 * apq);
    const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
    const double c = 1.0 / sqrt(t * t + 1.0);
    const double s = t * c;
    app = app - t * apq;
    aqq = aqq + t * apq;
    apq = 0.0;
    const double rp = arp, rq = arq;
    arp = c * rp - s * rq;
    arq = s * rp + c * rq;
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        const double a = vp[m], b = vq[m];
        vp[m] = c * a - s * b;
        vq[m] = s * a + c * b;
    }
}

// The symmetric 3 x 3 eigenproblem of a cell's scatter matrix -> the normal (eigenvector of the smallest eigenvalue, first of
// equals; z axis for a rank-deficient matrix; pointing up), cyclic Jacobi as in oracle/fpo_filters.cpp.  wS / wL: smallest
// eigenvalue and the scale it is compared with.
__device__ __forceinline__ void normal_from_scatter(double a00, double a01, double a02, double a11, double a12, double a22, double& ex, double& ey,
                                                    double& ez, double& wS, double& wL) {
    double v0[3] = {1.0, 0.0, 0.0}, v1[3] = {0.0, 1.0, 0.0}, v2[3] = {0.0, 0.0, 1.0};  // columns of V
    for (int sweep = 0; sweep < 12; ++sweep) {
        const double off = fabs(a01) + fabs(a02) + fabs(a12);
        if (off == 0.0) break;
        jacobi_rotate(a00, a11, a01, a02, a12, v0, v1);  // (p, q, r) = (0, 1, 2)
        jacobi_rotate(a00, a22, a02, a01, a12, v0, v2);  // (0, 2, 1)
        jacobi_rotate(a11, a22, a12, a01, a02, v1, v2);  // (1, 2, 0)
    }
    // the eigenvector of the smallest eigenvalue (first of equals); rank-deficient covariance -> z axis
    wS = a00;
    ex = v0[0]; ey = v0[1]; ez = v0[2];
    if (a11 < wS) { wS = a11; ex = v1[0]; ey = v1[1]; ez = v1[2]; }
    if (a22 < wS) { wS = a22; ex = v2[0]; ey = v2[1]; ez = v2[2]; }
    wL = fmax(fmax(fmax(a00, 0.0), a11), a22);
    if (!(wS > 3.0 * DBL_EPSILON * wL)) { ex = 0.0; ey = 0.0; ez = 1.0; }
    if (ez < 0.0) { ex = -ex; ey = -ey; ez = -ez; }
}

// The same eigenvector without the Jacobi sweeps (round 4; the row-moment kernel only): the smallest root of the
// characteristic polynomial by Newton's iteration from 0 — for a symmetric positive semi-definite matrix the cubic has three
// real non-negative roots, and from the left of the smallest one Newton's steps rise monotonically to it, quadratically unless
// it is (nearly) double — then the cross product of two rows of A - lambda I.  The matrix is scaled to a largest diagonal
// entry of 1 first.  ~140 f64 instructions where five Jacobi sweeps take ~750 (two divisions and two square roots per
// rotation).  What the iteration is asked for is lambda to 4e-16 of the trace — the ABSOLUTE accuracy that decides the
// eigenvector (its error is the error of lambda over the gap to the next eigenvalue, the same conditioning Jacobi has).
// On 4 x 10^5 scatter matrices of the synthetic terrains (1 cm / 2 cm / 0.5 cm) the float normals equal Jacobi's bit for
// bit in every cell that converges within the iteration cap (scratch/eig_proto.py: 99.9-100 % of the cells; mean 3.9
// iterations per cell, 4.1 per wavefront); a cell that does not — eigenvalues within a few per cent of each other — reports
// false and takes the sweeps.  wS / wL as normal_from_scatter, wL being the trace (>= the largest eigenvalue, <= 3 x).
#ifndef FPE_NEWTON_CAP
#define FPE_NEWTON_CAP 32
#endif
__device__ __forceinline__ double rcp_refined(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    return __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
}
__device__ __forceinline__ bool normal_newton(double a00, double a01, double a02, double a11, double a12, double a22, double& ex, double& ey, double& ez,
                                              double& wS, double& wL, double& gapRel) {
    const double s = fmax(fmax(a00, a11), a22);
    const double inv = rcp_refined(s);
    const double b00 = a00 * inv, b01 = a01 * inv, b02 = a02 * inv, b11 = a11 * inv, b12 = a12 * inv, b22 = a22 * inv;
    const double c2 = (b00 + b11) + b22;
    const double m0 = __builtin_fma(b11, b22, -(b12 * b12)), m1 = __builtin_fma(b00, b22, -(b02 * b02)), m2 = __builtin_fma(b00, b11, -(b01 * b01));
    const double c1 = (m0 + m1) + m2;
    const double c0 = __builtin_fma(b02, __builtin_fma(b01, b12, -(b11 * b02)), __builtin_fma(b00, m0, -(b01 * __builtin_fma(b01, b22, -(b12 * b02)))));
    const double tol = 4e-16 * c2;
    // start: Halley's step from 0 (f(0) = -c0, f'(0) = c1, f''(0) = -2 c2) — one order better than Newton's c0 / c1, one
    // iteration less.  For a cubic with three real roots both steps stay left of the smallest root; were rounding to carry the
    // start past it, Newton's next step returns to the left (f is concave and increasing up to c2 / 3).
    double lam = c0 * c1 * rcp_refined(__builtin_fma(c1, c1, -(c0 * c2)));
    // degenerate input — no matrix, or the two smaller eigenvalues both at rounding level (c1 ~ lambda2 lambda3 when lambda1 ~ 0:
    // two members, collinear members): c0 / c1 would be a quotient of rounding noise.  Reported as not converged: the sweeps
    // and the rank test decide.
    bool done = !(s > 0.0) || !(c1 > 1e-9 * (c2 * c2));
    const bool bad = done;
    // (cap: beside a riser the two small eigenvalues lie within a per cent of each other — z variance 0.35 against 0.0476 and 0.0446
    // at 1 cm — and the iteration halves its error step by step until it is inside their gap: 13-20 steps.  At a cap of 16 such
    // cells fell to the sweeps and, their gap being below 1e-3, to the literal walks: 196 cells of the 1 cm probe map, each holding
    // its workgroup for 20-40 us.  The loop leaves when the wavefront's last lane is done.)
    for (int it = 0; it < FPE_NEWTON_CAP; ++it) {
        const double f = __builtin_fma(__builtin_fma(lam - c2, lam, c1), lam, -c0);
        const double fp = __builtin_fma(__builtin_fma(3.0, lam, -2.0 * c2), lam, c1);
        const double dl = f * rcp_refined(fp);
        if (!done) lam -= dl;
        // (stopping one step earlier, when the error a step LEAVES is below the tolerance, was tried in round 5: two more cells
        // in 45 000 campaign maps left the one-ulp bar — steep faces, where the eigenvector takes every bit of lambda)
        done = done || fabs(dl) <= tol;
        if (__ballot(!done) == 0ull) break;
    }
    // The scaled entries are formed AGAIN here from the caller's matrix instead of being kept across the iteration (the compiler is
    // kept from reusing the first set by an empty asm on the scale): six multiplications against twelve registers held through the
    // loop — with them the 2 cm instantiation (64 registers, eight wavefronts per SIMD) spilled four doubles of the caller's matrix
    // in EVERY cell, 32 B of scratch stores per thread and launch (47 MB written per 1000 x 1000 launch for 4 MB of
    // layers).  Same operands, same operation: the same values bit for bit.
#ifndef FPE_NEWTON_KEEP_SCALED
    double inv2 = inv;
    asm volatile("" : "+v"(inv2));
    const double d00 = a00 * inv2, d01 = a01 * inv2, d02 = a02 * inv2, d11 = a11 * inv2, d12 = a12 * inv2, d22 = a22 * inv2;
#else
    const double d00 = b00, d01 = b01, d02 = b02, d11 = b11, d12 = b12, d22 = b22;
#endif
    const double m00 = d00 - lam, m11 = d11 - lam, m22 = d22 - lam;
    // cross products of the rows of A - lambda I: where the lattice's x and y spread dominate the matrix (terrain below ~45
    // degrees on a disc that is not cut down to a sliver: entries d00, d11 of order 1 after the scaling) rows 0 and 1 give the
    // long product, squared length of order 1.  Below 0.05 — steep faces, whose z variance takes the scale and leaves rows 0
    // and 1 small and their product cancellation-limited (the round-5 campaign's seeds 2505080, 2514166: normals 2-9 float ulps
    // off with a looser bar), slivers at map borders — the other two are formed and the longest of the three taken, per lane:
    // a cell's value never depends on the cells it shares a wavefront with.
    const double p0 = __builtin_fma(d01, d12, -(d02 * m11)), p1 = __builtin_fma(d02, d01, -(m00 * d12)), p2 = __builtin_fma(m00, m11, -(d01 * d01));
    const double np = __builtin_fma(p0, p0, __builtin_fma(p1, p1, p2 * p2));
    double v0 = p0, v1 = p1, v2 = p2, nn = np;
    const bool shortP = !(np > 0.05);
    if (__ballot(shortP) != 0ull) {
        const double q0 = __builtin_fma(d01, m22, -(d02 * d12)), q1 = __builtin_fma(d02, d02, -(m00 * m22)), q2 = __builtin_fma(m00, d12, -(d01 * d02));
        const double t0 = __builtin_fma(m11, m22, -(d12 * d12)), t1 = __builtin_fma(d12, d02, -(d01 * m22)), t2 = __builtin_fma(d01, d12, -(m11 * d02));
        const double nq = __builtin_fma(q0, q0, __builtin_fma(q1, q1, q2 * q2)), nt = __builtin_fma(t0, t0, __builtin_fma(t1, t1, t2 * t2));
        if (shortP && nq > nn) { v0 = q0; v1 = q1; v2 = q2; nn = nq; }
        if (shortP && nt > nn) { v0 = t0; v1 = t1; v2 = t2; nn = nt; }
    }
    double y = __builtin_amdgcn_rsq(nn);  // 1 / sqrt(nn), refined twice
    y = y * __builtin_fma(-0.5 * nn, y * y, 1.5);
    y = y * __builtin_fma(-0.5 * nn, y * y, 1.5);
    ex = v0 * y;
    ey = v1 * y;
    ez = v2 * y;
    if (ez < 0.0) { ex = -ex; ey = -ey; ez = -ez; }
    wS = lam * s;
    wL = c2 * s;
    // the next eigenvalue (the other two roots: sum S, product P): an eigenvector is only as good as its eigenvalue is apart.
    // lam2 = (S - sqrt(S^2 - 4 P)) / 2 with the hardware's unrefined square root and reciprocal (1e-8: a threshold's worth);
    // the caller weighs the relative gap against the smallest component of the normal.  (The cheaper bound lam2 >= P / S is
    // useless where it matters: beside a riser two SMALL eigenvalues face one large one, and P / S falls below lam.)
    const double S = c2 - lam, P = __builtin_fma(-lam, S, c1);
    const double lam2 = 0.5 * (S - __builtin_amdgcn_sqrt(fmax(__builtin_fma(S, S, -4.0 * P), 0.0)));
    gapRel = (lam2 - lam) * __builtin_amdgcn_rcp(c2);
    return done && !bad && nn > 0.0 && lam >= 0.0 && gapRel > 1e-6;
}

// One cell of NormalVectorsFilter (area method) + SlopeFilter [+ RoughnessFilter of the same radius] by the LITERAL walks of
// the published filters (points, mean, scatter about the mean, plane distances: three passes over the iterator's members).
__device__ __forceinline__ void normals_cell_exact(const DiscLds& d, int li, int lj, int ti0, int tj0, double r, double slopeCritical, int fuseRough,
                                                   double roughCritical, float& ox, float& oy, float& oz, float& os, float& orough) {
    const double r2 = r * r;
    int np = 0;
    double sx = 0.0, sy = 0.0, sz = 0.0;
    disc_walk(d, li, lj, ti0, tj0, r2, [&](double x, double y, float z) {
        if (isfinite(z)) { ++np; sx += x; sy += y; sz += static_cast<double>(z); }
    });
    const double nd = static_cast<double>(np);
    const double mx = sx / nd, my = sy / nd, mz = sz / nd;
    double a00 = 0.0, a01 = 0.0, a02 = 0.0, a11 = 0.0, a12 = 0.0, a22 = 0.0;
    disc_walk(d, li, lj, ti0, tj0, r2, [&](double x, double y, float z) {
        if (isfinite(z)) {
            const double dx = x - mx, dy = y - my, dz = static_cast<double>(z) - mz;
            a00 += dx * dx; a01 += dx * dy; a02 += dx * dz;
            a11 += dy * dy; a12 += dy * dz; a22 += dz * dz;
        }
    });
    double ex, ey, ez, wS, wL;
    normal_from_scatter(a00, a01, a02, a11, a12, a22, ex, ey, ez, wS, wL);
    ox = static_cast<float>(ex);
    oy = static_cast<float>(ey);
    oz = static_cast<float>(ez);
    const double slope = acos(static_cast<double>(oz));  // SlopeFilter reads the float layer
    os = slope < slopeCritical ? static_cast<float>(1.0 - slope / slopeCritical) : 0.0f;
    if (fuseRough) {  // RoughnessFilter::update with the float normals just written
        const double normalX = ox, normalY = oy, normalZ = oz;
        const double planeParameter = mx * normalX + my * normalY + mz * normalZ;
        double sum = 0.0;
        disc_walk(d, li, lj, ti0, tj0, r2, [&](double x, double y, float z) {
            if (isfinite(z)) {
                const double dist = normalX * x + normalY * y + normalZ * static_cast<double>(z) - planeParameter;
                sum += dist * dist;
            }
        });
        const double roughness = sqrt(sum / (nd - 1.0));
        orough = roughness < roughCritical ? static_cast<float>(1.0 - roughness / roughCritical) : 0.0f;
    }
}

// NormalVectorsFilter (area method) + SlopeFilter; with fuseRough also the RoughnessFilter of the same radius (same
// members in the same order, hence the same point count and mean: one walk and one tile load less than its own launch).
__global__ __launch_bounds__(256) void filter_normals_kernel(MapGeom g, const float* __restrict__ elev, FilterLayers L, double r, int H, double slopeCritical,
                                                              int fuseRough, double roughCritical) {
    extern __shared__ __attribute__((aligned(16))) char ldsRaw[];
    const DiscLds d = disc_carve(ldsRaw, H);
    const int ti0 = blockIdx.y * kFT, tj0 = blockIdx.x * kFT;
    disc_setup(d, g, elev, ti0, tj0, r);
    const int li = threadIdx.x / kFT, lj = threadIdx.x % kFT;
    const int i = ti0 + li, j = tj0 + lj;
    if (i >= g.rows || j >= g.cols) return;
    const size_t cell = static_cast<size_t>(i) * g.cols + j;
    const float nanf = __builtin_nanf("");
    float ox = nanf, oy = nanf, oz = nanf, os = nanf, orough = nanf;
    if (isfinite(d.tile[(li + H) * d.WC + lj + H])) normals_cell_exact(d, li, lj, ti0, tj0, r, slopeCritical, fuseRough, roughCritical, ox, oy, oz, os, orough);
    L.nx[cell] = ox;
    L.ny[cell] = oy;
    L.nz[cell] = oz;
    L.slope[cell] = os;
    if (fuseRough) L.rough[cell] = orough;
}

// ---- the disc as a lattice SHAPE -------------------------------------------------------------------------------------------
// On the uniform lattice the members of a cell's disc are the same offsets for every cell — row offset o holds the columns
// |dc| <= w(o) — except for the few offsets that lie ON the circle ((0, R), (R, 0), (3, 4) R / 5 ...), where
// CircleIterator::isInside's rounding decides cell by cell.  step_shape sorts the offsets on the host: squared distance
// against r^2 with a relative margin of 1e-7 (position differences are good to 1e-12).
constexpr int kStepMaxClasses = 14;
constexpr int kStepMaxEdge = 24;
struct StepShape {
    int8_t rowW[2 * kFilterMaxH + 1 + 3];  // per row offset o + H: its robust half-width, -1: no robust member
    int8_t edgeR[kStepMaxEdge], edgeC[kStepMaxEdge];  // offsets on the circle
    uint32_t storeMask;                // bit w: some row has half-width w — the run is stored at that width; its class = the number of set bits below w
    int32_t nEdge, nClasses, wMax;
    int32_t ok;                        // 0: the shape does not fit the step kernels' tables (the walking kernels run)
    int32_t rowsOk;                    // 0: not even the rows' half-widths are to be trusted (map too far from the origin)
    uint64_t edgeRows;                 // bit o + H: row offset o holds an offset on the circle (complete also when the edge list overflowed)
};
// reach: the largest |coordinate| of a cell of the map — a position difference carries ~2 ulp of that, which decides how wide
// the band of offsets is that the per-cell arithmetic must settle.
__host__ inline StepShape step_shape(double r, double res, int H, double reach) {
    StepShape sp{};
    for (auto& c : sp.rowW) c = -1;
    sp.ok = 1;
    sp.rowsOk = 1;
    const double r2 = r * r, res2 = res * res;
    // relative margin on squared distances: 1e-7 near the origin; far from it the rounding of the positions themselves
    // (4 ulp of `reach` on a difference of one resolution, twice that on its square), with a factor of 16 to spare.
    // Beyond 1e-3 two neighbouring offsets of a row could both be in doubt: the walking kernels take over.
    const double rel = fmax(1e-7, 16.0 * 8.0 * 2.220446049250313e-16 * reach / res);
    if (rel > 1e-3) {
        sp.ok = sp.rowsOk = 0;
        return sp;
    }
    const double margin = rel * r2;
    for (int o = -H; o <= H; ++o) {
        int w = -1;
        for (int k = 0; k <= H; ++k) {
            const double d2 = (static_cast<double>(o) * o + static_cast<double>(k) * k) * res2;
            if (d2 < r2 - margin) {
                w = k;
            } else if (d2 <= r2 + margin) {  // on the circle: decided per cell
                sp.edgeRows |= 1ull << (o + H);
                for (int sgn = (k == 0 ? 1 : -1); sgn <= 1; sgn += 2) {
                    if (sp.nEdge >= kStepMaxEdge) {
                        sp.ok = 0;  // (the rows' half-widths below stay valid: the normals kernel tests the next column itself)
                    } else {
                        sp.edgeR[sp.nEdge] = static_cast<int8_t>(o);
                        sp.edgeC[sp.nEdge] = static_cast<int8_t>(sgn * k);
                        ++sp.nEdge;
                    }
                }
            }
        }
        sp.rowW[o + H] = static_cast<int8_t>(w);
        if (w >= 0) sp.storeMask |= 1u << w;
        if (w > sp.wMax) sp.wMax = w;
    }
    sp.nClasses = __builtin_popcount(sp.storeMask);
    if (sp.nClasses > kStepMaxClasses) sp.ok = 0;
    return sp;
}
// ---- the same three filters by ROW MOMENTS (round 4; the kernel is round 5's filter_fused_kernel, fpe_filters_fused.hpp) -------
// A cell's disc is at most 2H + 1 row intervals (disc_walk: the members of a row are one interval).  Everything the three
// filters need of the members — count, mean, the 3 x 3 scatter matrix about the mean, and the sum of squared plane
// distances, which for the plane through the mean is n^T A n — is a function of the members' MOMENTS, and a row interval's
// moments are differences of per-row prefix sums (count, sum of c, sum of c^2 as integers, c = tile column; sum of z', z'^2,
// c z' in f64, z' = z - z0 with z0 one elevation of the tile): two LDS reads per quantity and ROW instead of one visit per
// MEMBER and pass (81 members x 3 passes x ~30 f64 operations at 1 cm).  Coordinates enter as exact integers (the lattice),
// recentred at the cell; metres only scale the finished matrix.
// Not the oracle's summation order: the scatter matrix agrees with the two-pass walk to ~1e-14 relative (the bar of
// tests/test_gpu_filters.py is one float ulp and 99.99 % bit-identical cells).  Where that is not enough — a (nearly)
// rank-deficient matrix, smallest eigenvalue below 1e-10 of the scale: exact planes, flat synthetic ground, fewer than three
// members, where the rank test and the last bits of a tiny component depend on the summation order — the cell takes the
// literal walks above instead (wave-divergent; no natural terrain gets there).
constexpr int kMomentMaxH = 12;

// RoughnessFilter: needs the finished normal layers.
__global__ __launch_bounds__(256) void filter_roughness_kernel(MapGeom g, const float* __restrict__ elev, FilterLayers L, double r, int H, double critical) {
    extern __shared__ __attribute__((aligned(16))) char ldsRaw[];
    const DiscLds d = disc_carve(ldsRaw, H);
    const int ti0 = blockIdx.y * kFT, tj0 = blockIdx.x * kFT;
    disc_setup(d, g, elev, ti0, tj0, r);
    const int li = threadIdx.x / kFT, lj = threadIdx.x % kFT;
    const int i = ti0 + li, j = tj0 + lj;
    if (i >= g.rows || j >= g.cols) return;
    const size_t cell = static_cast<size_t>(i) * g.cols + j;
    float out = __builtin_nanf("");
    const float fx = L.nx[cell];
    if (isfinite(fx)) {
        const double normalX = fx, normalY = L.ny[cell], normalZ = L.nz[cell];
        const double r2 = r * r;
        int np = 0;
        double sx = 0.0, sy = 0.0, sz = 0.0;
        disc_walk(d, li, lj, ti0, tj0, r2, [&](double x, double y, float z) {
            if (isfinite(z)) { ++np; sx += x; sy += y; sz += static_cast<double>(z); }
        });
        const double nd = static_cast<double>(np);
        const double mx = sx / nd, my = sy / nd, mz = sz / nd;
        const double planeParameter = mx * normalX + my * normalY + mz * normalZ;
        double sum = 0.0;
        disc_walk(d, li, lj, ti0, tj0, r2, [&](double x, double y, float z) {
            if (isfinite(z)) {
                const double dist = normalX * x + normalY * y + normalZ * static_cast<double>(z) - planeParameter;
                sum += dist * dist;
            }
        });
        const double roughness = sqrt(sum / (nd - 1.0));
        out = roughness < critical ? static_cast<float>(1.0 - roughness / critical) : 0.0f;
    }
    L.rough[cell] = out;
}

// StepFilter, first iteration: step_height = max - min over the first window.
__global__ __launch_bounds__(256) void filter_step1_kernel(MapGeom g, const float* __restrict__ elev, FilterLayers L, double r, int H) {
    extern __shared__ __attribute__((aligned(16))) char ldsRaw[];
    const DiscLds d = disc_carve(ldsRaw, H);
    const int ti0 = blockIdx.y * kFT, tj0 = blockIdx.x * kFT;
    disc_setup(d, g, elev, ti0, tj0, r);
    const int li = threadIdx.x / kFT, lj = threadIdx.x % kFT;
    const int i = ti0 + li, j = tj0 + lj;
    if (i >= g.rows || j >= g.cols) return;
    float out = __builtin_nanf("");
    if (isfinite(d.tile[(li + H) * d.WC + lj + H])) {
        float hi = -__builtin_huge_valf(), lo = __builtin_huge_valf();
        disc_walk(d, li, lj, ti0, tj0, r * r, [&](double, double, float z) {
            hi = max_skip_nan(hi, z);
            lo = min_skip_nan(lo, z);
        });
        out = static_cast<float>(static_cast<double>(hi) - static_cast<double>(lo));  // the centre is a member: init holds
    }
    L.stepHeight[static_cast<size_t>(i) * g.cols + j] = out;
}

// StepFilter, second iteration, and the weighted sum of the three filters.
__global__ __launch_bounds__(256) void filter_step2_kernel(MapGeom g, FilterLayers L, double r, int H, double critical, float critDown, int nCritical) {
    extern __shared__ __attribute__((aligned(16))) char ldsRaw[];
    const DiscLds d = disc_carve(ldsRaw, H);
    const int ti0 = blockIdx.y * kFT, tj0 = blockIdx.x * kFT;
    disc_setup(d, g, L.stepHeight, ti0, tj0, r);
    const int li = threadIdx.x / kFT, lj = threadIdx.x % kFT;
    const int i = ti0 + li, j = tj0 + lj;
    if (i >= g.rows || j >= g.cols) return;
    const size_t cell = static_cast<size_t>(i) * g.cols + j;
    // double(sh) > critical for a float sh is sh > critDown = critical rounded DOWN to float (launch_filters); NaN (invalid)
    // compares false
    int nCells = 0;
    float seen = -1.0f;  // step heights are >= +0: the maximum over the valid members, -1 when there is none
    disc_walk(d, li, lj, ti0, tj0, r * r, [&](double, double, float sh) {
        seen = max_skip_nan(seen, sh);
        nCells += sh > critDown ? 1 : 0;
    });
    const bool valid = seen >= 0.0f;
    const float stepMaxF = valid ? seen : 0.0f;
    float out = __builtin_nanf("");
    if (valid) {
        const double stepMax = static_cast<double>(stepMaxF);
        const double step = fmin(stepMax, static_cast<double>(nCells) / static_cast<double>(nCritical) * stepMax);
        out = step < critical ? static_cast<float>(1.0 - step / critical) : 0.0f;
    }
    L.step[cell] = out;
    const float third = 1.0f / 3.0f;  // MathExpressionFilter on float matrices: (1.0 / 3.0) * (slope + step + roughness)
    L.trav[cell] = third * ((L.slope[cell] + out) + L.rough[cell]);
}

#include "fpe_filters_fused.hpp"

// ---- the step filter's two windows by ROW RUNS (round 4) -----------------------------------------------------------------
// Both iterations of the StepFilter reduce a disc with max / min (and a count): order-free, so only the SET of members
// matters.  On the uniform lattice the set is the same for every cell — row offset o holds the columns |dc| <= w(o) — except
// for the few offsets that lie ON the circle ((0, R), (R, 0), (3, 4) R / 5 ...), where CircleIterator::isInside's rounding
// decides cell by cell.  The host sorts the offsets (step_shape: squared distance against r^2 with a relative margin of
// 1e-7 — position differences are good to 1e-12); the kernel then
//   1. extends, for every (tile row, interior column), a run around the column one cell per side at a time up to the
//      largest half-width, keeping the running max / min (count) in registers and storing it at the half-widths some row
//      of the disc has ("classes") — 2 R + 1 LDS reads per run instead of one walk per cell;
//   2. folds, per cell, one stored run per disc row: 2 R + 1 reads of LDS for a disc of ~pi R^2 members;
//   3. tests the on-circle offsets with the iterator's own arithmetic (dx2 + dy2 <= r^2 on the tables of disc_setup, and
//      the bounding box of the cell) and folds the members among them one by one.
// Invalid cells are quiet NaNs in the tile: v_max / v_min skip them, `>` is false on them — what the iterator's isValid
// test does.  Same members as disc_walk by construction: bit-identical layers (tests/test_gpu_filters.py).
__host__ __device__ inline size_t step_lds_bytes(int H, int nClasses, int TR = kFT, int TC = kFT) {
    const int WR = TR + 2 * H;
    return ((disc_lds_bytes(H, TR, TC) + 15) & ~static_cast<size_t>(15)) + static_cast<size_t>(nClasses) * WR * TC * 8 + 128;
}

// kSecond false: step_height = max - min of the elevation over the first window.  true: the second window over the step
// heights (their maximum and the number above the critical value).  One tile's worth as a device function — every thread of
// the workgroup takes part in the barriers; `live`: the thread's cell lies inside the map.  Returns the fold of the window
// (hi; lo or cnt) and the centre value; the callers turn them into layers.
// HS > 0: the halo as a compile-time constant (round 5: the windows of the published default chain at 2 cm and 1 cm, 5 and 9
// cells) — the run extension and the fold over the disc's rows are unrolled, their LDS addresses immediates, the shape's
// per-row entries scalar registers; HS = 0: any halo at run time (the shape's tables go through LDS).
__device__ __forceinline__ float max3_skip_nan(float a, float b, float c) {  // v_max_f32(v_max_f32(a, b), c): quiet NaNs skipped
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float min3_skip_nan(float a, float b, float c) {
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
template <bool kSecond, int TR, int TC, int HS, bool kCoherent = false>
__device__ __forceinline__ void step_runs_phase(char* ldsRaw, const MapGeom& g, const float* __restrict__ src, double r, int Hrt, const StepShape& sp,
                                                float critDown, int ti0, int tj0, bool live, float& hiOut, float& loOut, int& cntOut, float& centreOut) {
    const int H = HS > 0 ? HS : Hrt;
    const DiscLds d = disc_carve(ldsRaw, H, TR, TC);
    const int WR = TR + 2 * H, WC = TC + 2 * H;
    // stored runs: [class][tile row][interior column] pairs (max, min) or (max, count bits)
    float2* runs = reinterpret_cast<float2*>(ldsRaw + ((disc_lds_bytes(H, TR, TC) + 15) & ~static_cast<size_t>(15)));
    // the shape's per-row half-widths and on-circle offsets in LDS (read in loops of run-time length: from the kernel arguments
    // every read is a scalar load the loop waits for)
    int8_t* shp = reinterpret_cast<int8_t*>(runs + static_cast<size_t>(sp.nClasses) * WR * TC);
    if (threadIdx.x < 2 * kFilterMaxH + 1) shp[threadIdx.x] = sp.rowW[threadIdx.x];
    else if (threadIdx.x >= 64 && threadIdx.x < 64 + kStepMaxEdge) shp[threadIdx.x] = sp.edgeR[threadIdx.x - 64];
    else if (threadIdx.x >= 96 && threadIdx.x < 96 + kStepMaxEdge) shp[threadIdx.x] = sp.edgeC[threadIdx.x - 96];
    const unsigned storeMask = sp.storeMask;
    // (with the per-axis distance tables: the on-circle offsets' tests below read them instead of recomputing two cell
    // positions per offset and cell)
    disc_setup<true, TR, TC, HS, kCoherent>(d, g, src, ti0, tj0, r);
    if (kSecond) FPE_TL_MARK(8);
    const float ninf = -__builtin_huge_valf(), pinf = __builtin_huge_valf();
    for (int e = threadIdx.x; e < WR * TC; e += TR * TC) {  // 1. one run per (tile row, interior column)
        const int row = e / TC, c = e - row * TC;
        const float* p = d.tile + row * WC + c + H;
        const float z = p[0];
        float hi = max_skip_nan(kSecond ? -1.0f : ninf, z), lo = kSecond ? 0.0f : min_skip_nan(pinf, z);
        int cnt = (kSecond && z > critDown) ? 1 : 0;
        float2* out = runs + row * TC + c;  // class k at out[k * WR * TC]; classes in the order of their widths
        if (storeMask & 1u) {
            *out = make_float2(hi, kSecond ? __int_as_float(cnt) : lo);
            out += WR * TC;
        }
        const auto extend = [&](int w) {
            const float a = p[-w], b = p[w];
            hi = max3_skip_nan(hi, a, b);
            if (kSecond) cnt += (a > critDown ? 1 : 0) + (b > critDown ? 1 : 0);
            else lo = min3_skip_nan(lo, a, b);
            if ((storeMask >> w) & 1u) {
                *out = make_float2(hi, kSecond ? __int_as_float(cnt) : lo);
                out += WR * TC;
            }
        };
        if constexpr (HS > 0) {
#pragma unroll
            for (int w = 1; w < HS; ++w)  // (a robust half-width is below the halo: the halo is one cell wider than the radius)
                if (w <= sp.wMax) extend(w);
        } else {
            for (int w = 1; w <= sp.wMax; ++w) extend(w);
        }
    }
    __syncthreads();
    if (kSecond) FPE_TL_MARK(9);
    const int li = threadIdx.x / TC, lj = threadIdx.x % TC;
    const int i = ti0 + li, j = tj0 + lj;
    const float centre = d.tile[(li + H) * WC + lj + H];
    float hi = kSecond ? -1.0f : ninf, lo = pinf;
    int cnt = 0;
    if (live && (kSecond || isfinite(centre))) {
        // 2. one stored run per row of the disc (the cell's bounding box never cuts a robust member: it spans every row and
        // column within r of the centre, and what lies outside the map is NaN in the tile)
        const float2* const mine = runs + li * TC + lj;  // row offset o, class k: mine[(k * WR + H + o) * TC]
        const auto fold = [&](int o, int w) {
            const int k = __builtin_popcount(storeMask & ((1u << w) - 1u));
            const float2 v = mine[(k * WR + H + o) * TC];
            hi = max_skip_nan(hi, v.x);
            if (kSecond) cnt += __float_as_int(v.y);
            else lo = min_skip_nan(lo, v.y);
        };
        if constexpr (HS > 0) {
#pragma unroll
            for (int o = -HS; o <= HS; ++o) {
                const int w = sp.rowW[o + HS];  // (a scalar register: the kernel argument at a constant index)
                if (w >= 0) fold(o, w);
            }
        } else {
            for (int o = -H; o <= H; ++o) {
                const int w = shp[o + H];
                if (w >= 0) fold(o, w);
            }
        }
        if (kSecond) FPE_TL_MARK(10);
        // 3. the offsets on the circle, by the iterator's own tests: bounding box, and the squared axis distances of
        // disc_setup's tables (the difference of two cell positions, squared — CircleIterator::isInside's operands)
        const double r2 = r * r;
        const int i0 = d.bi0[li], i1 = d.bi1[li], j0 = d.bj0[lj], j1 = d.bj1[lj];
        const int D = 2 * H + 1;
        const double* const dxRow = d.dx2 + li * D + H;
        const double* const dyRow = d.dy2 + lj * D + H;
        const float* const tc = d.tile + (li + H) * WC + lj + H;
        // four offsets at a time: their table and tile reads issue together (one offset after the other the loop waited for
        // three dependent LDS round trips each: 1.6 of a workgroup's 13.8 us at 2 cm); entries past nEdge are (0, 0) and masked
        constexpr int kG = 4;
        const auto edge_group = [&](int e0, auto&& offs) {
            int o[kG], oc[kG];
            double a[kG], b[kG];
            float z[kG];
#pragma unroll
            for (int k = 0; k < kG; ++k) offs(e0 + k, o[k], oc[k]);
#pragma unroll
            for (int k = 0; k < kG; ++k) {
                a[k] = dxRow[o[k]];
                b[k] = dyRow[oc[k]];
                z[k] = tc[o[k] * WC + oc[k]];
            }
#pragma unroll
            for (int k = 0; k < kG; ++k) {
                const int ii = i + o[k], jj = j + oc[k];
                const bool in = e0 + k < sp.nEdge && ii >= i0 && ii <= i1 && jj >= j0 && jj <= j1 && a[k] + b[k] <= r2;
                const float zk = in ? z[k] : __builtin_nanf("");
                hi = max_skip_nan(hi, zk);
                if (kSecond) cnt += zk > critDown ? 1 : 0;
                else lo = min_skip_nan(lo, zk);
            }
        };
        static_assert(kStepMaxEdge % kG == 0, "the edge list is read in whole groups");
        if constexpr (HS > 0) {  // (the offsets are scalars: kernel arguments at constant indices)
#pragma unroll
            for (int e0 = 0; e0 < kStepMaxEdge; e0 += kG)
                if (e0 < sp.nEdge) edge_group(e0, [&](int e, int& o, int& oc) { o = sp.edgeR[e]; oc = sp.edgeC[e]; });
        } else {
            for (int e0 = 0; e0 < sp.nEdge; e0 += kG) edge_group(e0, [&](int e, int& o, int& oc) { o = shp[64 + e]; oc = shp[96 + e]; });
        }
    }
    hiOut = hi;
    loOut = lo;
    cntOut = cnt;
    centreOut = centre;
}
// StepFilter's closing arithmetic on the second window's fold: the step layer's value (NaN where the window held no valid cell).
__device__ __forceinline__ float step_value(float hi, int cnt, double critical, int nCritical) {
    float out = __builtin_nanf("");
    if (hi >= 0.0f) {
        const double stepMax = static_cast<double>(hi);
        const double step = fmin(stepMax, static_cast<double>(cnt) / static_cast<double>(nCritical) * stepMax);
        out = step < critical ? static_cast<float>(1.0 - step / critical) : 0.0f;
    }
    return out;
}

// The two windows as launches of their own.  Tiles are numbered XCD-aware (xcd_tile): 1-D grid of tilesX * tilesY workgroups.
template <bool kSecond, int TR, int TC, int HS>
__global__ __launch_bounds__(TR * TC) void filter_step_runs_kernel(MapGeom g, const float* __restrict__ src, FilterLayers L, double r, int H, StepShape sp,
                                                                double critical, float critDown, int nCritical, int tilesX, int nTiles) {
    extern __shared__ __attribute__((aligned(16))) char ldsRaw[];
    int ty, tx;
    xcd_tile(tilesX, nTiles, ty, tx);
    const int ti0 = ty * TR, tj0 = tx * TC;
    const int i = ti0 + static_cast<int>(threadIdx.x) / TC, j = tj0 + static_cast<int>(threadIdx.x) % TC;
    const bool live = i < g.rows && j < g.cols;
    float hi, lo, centre;
    int cnt;
    step_runs_phase<kSecond, TR, TC, HS>(ldsRaw, g, src, r, H, sp, critDown, ti0, tj0, live, hi, lo, cnt, centre);
    if (!live) return;
    const size_t cell = static_cast<size_t>(i) * g.cols + j;
    if constexpr (!kSecond) {
        L.stepHeight[cell] = isfinite(centre) ? static_cast<float>(static_cast<double>(hi) - static_cast<double>(lo)) : __builtin_nanf("");
    } else {
        const float out = step_value(hi, cnt, critical, nCritical);
        L.step[cell] = out;
        const float third = 1.0f / 3.0f;  // MathExpressionFilter on float matrices: (1.0 / 3.0) * (slope + step + roughness)
        L.trav[cell] = third * ((L.slope[cell] + out) + L.rough[cell]);
    }
}

// ---- the chain's second launch (round 5): normals + slope + roughness by row moments and the StepFilter's second window over
// the same 16 x 16 (32 x 32) cells, then the weighted sum.  kStep 0: the moment phase alone (the step window is a launch
// of its own — its shape does not fit the row-run tables).  travOnly (run time, wave-uniform): store the traversability
// layer and nothing else.
// (512-thread workgroups: three of them per CU — six wavefronts per SIMD, 80 registers; without the hint the allocator takes 88
// and a third of the CU's wavefronts with them: 0.283 -> 0.341 ms at 1 cm)
// One tile of the fused launch: the second step window, the moment phase, the stores, the walking phase.  kChain: the tile's
// coordinates come from the caller (filter_chain_kernel) instead of the workgroup number.
template <int H, int TR, int TC, int HS, bool kChain>
__device__ __forceinline__ void fused_tile(char* ldsRaw, const MapGeom& g, const float* __restrict__ elev, const FilterLayers& L, double rN, double slopeCritical, double roughCritical,
                                           double invSlopeCritical, double invRoughCritical, const StepShape& sN, double r2nd, int h2nd, const StepShape& s2, double stepCritical,
                                           float critDown, int nCritical, int kStepFlags, int travOnly, int ty, int tx) {
    const int ti0 = ty * TR, tj0 = tx * TC;
    const int i = ti0 + static_cast<int>(threadIdx.x) / TC, j = tj0 + static_cast<int>(threadIdx.x) % TC;
    const bool live = i < g.rows && j < g.cols;
    float stepOut = 0.0f;
    FPE_TL_MARK(0);
#ifdef FPE_FUSED_TIMELINE
    if (threadIdx.x == 0 && blockIdx.x < 8192) {
        g_fusedTimeline[blockIdx.x][6] = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_ID
        g_fusedTimeline[blockIdx.x][7] = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // XCC_ID
    }
#endif
    // kStepFlags: bit 0 = the second step window rides in this launch, bit 1 = the first step window is at least as wide as the
    // normals' disc (a step height of exactly 0 then certifies a disc of equal elevations: normals_from_moments, `flat`)
    const int kStep = kStepFlags & 1;
    float stepHeightHere = __builtin_nanf("");  // the cell's own step height (first window)
    if (kStep) {
        float hi, lo;
        int cnt;
        step_runs_phase<true, TR, TC, HS, kChain>(ldsRaw, g, L.stepHeight, r2nd, h2nd, s2, critDown, ti0, tj0, live, hi, lo, cnt, stepHeightHere);
        stepOut = step_value(hi, cnt, stepCritical, nCritical);
        __syncthreads();  // the moment phase reuses the LDS
    } else if (live && (kStepFlags & 2)) {
        stepHeightHere = L.stepHeight[static_cast<size_t>(i) * g.cols + j];
    }
    const bool flat = (kStepFlags & 2) != 0 && stepHeightHere == 0.0f;
    FPE_TL_MARK(1);
    float ox, oy, oz, os, orough;
    const bool needWalk = moments_phase<H, TR, TC>(ldsRaw, g, elev, ti0, tj0, rN, travOnly == 0, flat, sN, slopeCritical, roughCritical, invSlopeCritical, invRoughCritical, live, ox,
                                                   oy, oz, os, orough);
    if (live && !needWalk) {
        const size_t cell = static_cast<size_t>(i) * g.cols + j;
        if (!travOnly) {
            L.nx[cell] = ox;
            L.ny[cell] = oy;
            L.nz[cell] = oz;
            L.slope[cell] = os;
            L.rough[cell] = orough;
        }
        if (kStep) {
            const float third = 1.0f / 3.0f;  // MathExpressionFilter on float matrices: (1.0 / 3.0) * (slope + step + roughness)
            L.trav[cell] = third * ((os + stepOut) + orough);
        }
    }
    if (live && kStep && !travOnly) L.step[static_cast<size_t>(i) * g.cols + j] = stepOut;
    FPE_TL_MARK(2);
    // the cells that take the literal walks (rank-deficient scatter, components at rounding level): a phase of their own
    walk_phase<H, TR, TC, kChain>(ldsRaw, needWalk, stepOut, ty, tx);
    FPE_TL_MARK(3);
}

template <int H, int TR, int TC, int HS>
__global__ __launch_bounds__(TR * TC) __attribute__((amdgpu_waves_per_eu(TR * TC == 512 ? ((H == 3 && HS == 5) ? FPE_FUSED_WAVES_2CM : FPE_FUSED_WAVES) : 4))) void filter_fused_kernel(MapGeom g, const float* __restrict__ elev, FilterLayers L, double rN, double slopeCritical,
                                                            double roughCritical, double invSlopeCritical, double invRoughCritical, StepShape sN, double r2nd, int h2nd, StepShape s2, double stepCritical,
                                                            float critDown, int nCritical, int kStepFlags, int travOnly, int tilesX, int nTiles) {
    extern __shared__ __attribute__((aligned(16))) char ldsRaw[];
    int ty, tx;
    xcd_tile(tilesX, nTiles, ty, tx);
    fused_tile<H, TR, TC, HS, false>(ldsRaw, g, elev, L, rN, slopeCritical, roughCritical, invSlopeCritical, invRoughCritical, sN, r2nd, h2nd, s2, stepCritical, critDown, nCritical,
                                     kStepFlags, travOnly, ty, tx);
}

// ---- the chain as ONE launch (round 6) --------------------------------------------------------------------------------------
// Every workgroup first computes the step heights of one tile (the first window: step_runs_phase<false>), publishes them (release
// at device scope, one flag per tile), then runs the fused tile `lag` positions BEHIND in its XCD's order, after the flags of that
// tile and its eight neighbours.  Order: XCD x = workgroup % 8 owns the strip of `cw` tile columns [x cw, (x + 1) cw), row-major
// inside the strip (position k = workgroup / 8: tile row k / cw, column x cw + k % cw).  A neighbour of position k - lag lies at
// most at position k - lag + 2 cw - 1 of its own strip, so with lag = 2 cw every flag a workgroup waits for is set by a
// workgroup with a SMALLER number, which the dispatcher has started before it (workgroups are dispatched in the order of their
// numbers) and which waits for nobody before it sets its flag: no cycle.  A wait that never ends (the assumption broken) traps
// after ~1 s of polling instead of hanging the device.  Flags hold the launch's epoch: nobody clears them.
template <int H, int TR, int TC, int HS>
__global__ __launch_bounds__(TR * TC) __attribute__((amdgpu_waves_per_eu(TR * TC == 512 ? ((H == 3 && HS == 5) ? FPE_FUSED_WAVES_2CM : FPE_FUSED_WAVES) : 4))) void filter_chain_kernel(MapGeom g, const float* __restrict__ elev, FilterLayers L, double rN, double slopeCritical,
                                                            double roughCritical, double invSlopeCritical, double invRoughCritical, StepShape sN, double r2nd, int h2nd, StepShape s2, double stepCritical,
                                                            float critDown, int nCritical, int kStepFlags, int travOnly, int tilesX, int nTiles, StepShape s1, double r1st,
                                                            unsigned* flags, unsigned epoch, int cw, int nK, int lag) {
    extern __shared__ __attribute__((aligned(16))) char ldsRaw[];
    const int x = static_cast<int>(blockIdx.x & 7u), k = static_cast<int>(blockIdx.x >> 3);
    const int tilesY = nTiles / tilesX;
    if (k < nK) {
        const int ty = k / cw, tx = x * cw + (k - ty * cw);
        if (tx < tilesX) {  // (wave-uniform)
            const int ti0 = ty * TR, tj0 = tx * TC;
            const int i = ti0 + static_cast<int>(threadIdx.x) / TC, j = tj0 + static_cast<int>(threadIdx.x) % TC;
            const bool live = i < g.rows && j < g.cols;
            float hi, lo, centre;
            int cnt;
            step_runs_phase<false, TR, TC, HS>(ldsRaw, g, elev, r1st, HS, s1, 0.0f, ti0, tj0, live, hi, lo, cnt, centre);
            // device-scope stores (written through to where every XCD reads them) and device-scope loads on the other side: a release /
            // acquire pair of FENCES at device scope writes back and invalidates the XCD's whole L2 per workgroup — measured: the
            // chain at 0.475 ms instead of 0.054
            if (live)
                __hip_atomic_store(&L.stepHeight[static_cast<size_t>(i) * g.cols + j],
                                   isfinite(centre) ? static_cast<float>(static_cast<double>(hi) - static_cast<double>(lo)) : __builtin_nanf(""), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wavefront's stores are acknowledged ...
            __syncthreads();                                   // ... before the one flag says so (and the next phase reuses the LDS)
            if (threadIdx.x == 0) __hip_atomic_store(&flags[ty * tilesX + tx], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    const int k2 = k - lag;
    if (k2 < 0) return;
    const int ty = k2 / cw, tx = x * cw + (k2 - ty * cw);
    if (tx >= tilesX) return;
    if (threadIdx.x < 9) {
        const int ny = ty + static_cast<int>(threadIdx.x) / 3 - 1, nx = tx + static_cast<int>(threadIdx.x) % 3 - 1;
        if (ny >= 0 && ny < tilesY && nx >= 0 && nx < tilesX) {
            const unsigned* f = &flags[ny * tilesX + nx];
            unsigned polls = 0;
            while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
                __builtin_amdgcn_s_sleep(8);
                if (++polls > (1u << 22)) __builtin_trap();
            }
        }
    }
    __syncthreads();
    fused_tile<H, TR, TC, HS, true>(ldsRaw, g, elev, L, rN, slopeCritical, roughCritical, invSlopeCritical, invRoughCritical, sN, r2nd, h2nd, s2, stepCritical, critDown, nCritical,
                                    kStepFlags, travOnly, ty, tx);
}

__host__ inline int filter_halo(double r, double res) { return static_cast<int>(r / res) + 1; }

}  // namespace

bool filters_supported(const FilterConsts& fc, const MapGeom& g) {
    const double rmax = std::fmax(std::fmax(fc.normalRadius, fc.roughnessRadius), std::fmax(fc.stepFirstRadius, fc.stepSecondRadius));
    return filter_halo(rmax, g.res) <= kFilterMaxH;
}
// What a chain with these parameters can do without the caller's layer buffer: 1 = the two-launch chain that stores
// step_height and traversability only (the fused kernel covers normals, slope, roughness and the second step window),
// 0 = the intermediate layers have to exist (some filter runs as a launch of its own).
struct FilterRoute {
    StepShape sN, s1, s2;
    int hN, hR, h1, h2;
    int tF;          // tile COLUMNS of the fused kernel (its rows: 32), 0: the moment form does not apply (walking kernels)
    int stepFused;   // the second step window rides in the fused kernel
    int t1, t2;      // tiles of the row-run launches: 32 = 32 x 32, 24 = 32 rows x 16 columns, 16 = 16 x 16
    size_t fusedBytes;
};
namespace {
template <int H, int TR, int TC, int HS = 0>
hipError_t launch_fused_one(const MapGeom& g, const FilterConsts& fc, const float* d_elev, const FilterLayers& L, const FilterRoute& rt, float critDown, int travOnly,
                            hipStream_t stream) {
    const void* fn = reinterpret_cast<const void*>(filter_fused_kernel<H, TR, TC, HS>);
    if (rt.fusedBytes > 48 * 1024) {
        const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(rt.fusedBytes));
        if (e != hipSuccess) return e;
    }
    const int tilesX = (g.cols + TC - 1) / TC, nTiles = tilesX * ((g.rows + TR - 1) / TR);
    hipLaunchKernelGGL((filter_fused_kernel<H, TR, TC, HS>), dim3(nTiles), dim3(TR * TC), rt.fusedBytes, stream, g, d_elev, L, fc.normalRadius, fc.slopeCritical,
                       fc.roughnessCritical, 1.0 / fc.slopeCritical, 1.0 / fc.roughnessCritical, rt.sN, fc.stepSecondRadius, rt.h2, rt.s2, fc.stepCritical, critDown, fc.stepCriticalCells,
                       rt.stepFused | (fc.normalRadius <= fc.stepFirstRadius ? 2 : 0), travOnly, tilesX, nTiles);
    return hipGetLastError();
}
// The chain as one launch (filter_chain_kernel): flags = one word per tile, holding the epoch of the last launch that wrote the tile.
template <int H, int TR, int TC, int HS>
hipError_t launch_chain_one(const MapGeom& g, const FilterConsts& fc, const float* d_elev, const FilterLayers& L, const FilterRoute& rt, float critDown, int travOnly,
                            unsigned* flags, unsigned epoch, hipStream_t stream) {
    const void* fn = reinterpret_cast<const void*>(filter_chain_kernel<H, TR, TC, HS>);
    size_t bytes = rt.fusedBytes;
    const size_t first = step_lds_bytes(rt.h1, rt.s1.nClasses, TR, TC);
    if (first > bytes) bytes = first;
    if (bytes > 150 * 1024) return hipErrorInvalidValue;
    if (bytes > 48 * 1024) {
        const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes));
        if (e != hipSuccess) return e;
    }
    const int tilesX = (g.cols + TC - 1) / TC, tilesY = (g.rows + TR - 1) / TR, nTiles = tilesX * tilesY;
    const int cw = (tilesX + 7) / 8, nK = cw * tilesY, lag = 2 * cw;
    hipLaunchKernelGGL((filter_chain_kernel<H, TR, TC, HS>), dim3(8 * (nK + lag)), dim3(TR * TC), bytes, stream, g, d_elev, L, fc.normalRadius, fc.slopeCritical,
                       fc.roughnessCritical, 1.0 / fc.slopeCritical, 1.0 / fc.roughnessCritical, rt.sN, fc.stepSecondRadius, rt.h2, rt.s2, fc.stepCritical, critDown, fc.stepCriticalCells,
                       rt.stepFused | (fc.normalRadius <= fc.stepFirstRadius ? 2 : 0), travOnly, tilesX, nTiles, rt.s1, fc.stepFirstRadius, flags, epoch, cw, nK, lag);
    return hipGetLastError();
}
// The fused kernel's tile: 32 rows x 16 columns (512 threads) for halos up to eight cells, 32 x 32 beyond.  Rows amortise: the
// row runs of the step window are one per (tile row with halo, interior column), the prefix scans one lane per tile row.
constexpr int kFusedRows = 32;
constexpr int kFusedSmallMaxH = 8;
int fused_tile(int H) { return H <= kFusedSmallMaxH ? 16 : 32; }
}  // namespace
FilterRoute filter_route(const FilterConsts& fc, const MapGeom& g) {
    FilterRoute rt{};
    rt.hN = filter_halo(fc.normalRadius, g.res);
    rt.hR = filter_halo(fc.roughnessRadius, g.res);
    rt.h1 = filter_halo(fc.stepFirstRadius, g.res);
    rt.h2 = filter_halo(fc.stepSecondRadius, g.res);
    const double reach = std::fmax(std::fabs(g.posX) + g.orgX, std::fabs(g.posY) + g.orgY) + g.res;
    static const int forcedTile = std::getenv("FPE_FILTER_TILE") ? std::atoi(std::getenv("FPE_FILTER_TILE")) : 0;
#ifndef FPE_FILTERS_WALK_ONLY
    rt.sN = step_shape(fc.normalRadius, g.res, rt.hN, reach);
    rt.s1 = step_shape(fc.stepFirstRadius, g.res, rt.h1, reach);
    rt.s2 = step_shape(fc.stepSecondRadius, g.res, rt.h2, reach);
    // The moment form takes the lattice as EXACT integers; the published filters (and the oracle) sum the cells' rounded f64
    // positions.  Near the origin the two agree far inside a float ulp; at a coordinate of `reach` a position carries an error of
    // ulp(reach) ~ 2e-16 reach against a spacing of res, and the normal follows it: beyond reach / res = 2e5 (2 km at 1 cm, 4e-11
    // relative) the literal walks run instead (at 3 000 km a handful of cells differed by up to 8 float ulps; tests).
    const bool latticeExact = reach / g.res < 2.0e5;
    const bool sameDisc = fc.roughnessRadius == fc.normalRadius;  // the published default chain: both 0.05 m
    if (sameDisc && rt.hN >= 1 && rt.hN <= kMomentMaxH && rt.sN.rowsOk && latticeExact) rt.tF = fused_tile(rt.hN);
#endif
    // Tile edge of the row-run launches: 32 (1024 threads) when a tile row with its halo is still one wavefront load and the
    // arrays fit the LDS, else 16.  FPE_FILTER_TILE=16 (environment: a measurement switch) keeps the small tile.
    // (32 x 16 where a 32-column tile row with its halo is more than one wavefront load or the runs outgrow the LDS: windows of
    // 15-24 cells — 0.08 m at 0.5 cm: against 16 x 16 the runs per cell drop from 3.1 to 2.1 and the CU holds eight wavefronts
    // instead of four)
    const auto tile_of = [&](int H, int nClasses) {
        if (forcedTile == 16 || g.rows < 64 || g.cols < 64) return 16;
        if (32 + 2 * H <= 64 && step_lds_bytes(H, nClasses, 32, 32) <= 150 * 1024) return 32;
        if (16 + 2 * H <= 64 && step_lds_bytes(H, nClasses, 32, 16) <= 150 * 1024) return 24;
        return 16;
    };
    rt.t1 = tile_of(rt.h1, rt.s1.nClasses);
#ifdef FPE_FILTER_T1_ENV  // (measurement builds: the first window's tile from the environment)
    if (std::getenv("FPE_FILTER_T1")) rt.t1 = std::atoi(std::getenv("FPE_FILTER_T1"));
#endif
    rt.t2 = tile_of(rt.h2, rt.s2.nClasses);
    if (rt.tF) {
        rt.fusedBytes = fused_moment_bytes(rt.hN, kFusedRows, rt.tF);
        const size_t stepBytes = step_lds_bytes(rt.h2, rt.s2.nClasses, kFusedRows, rt.tF);
        if (rt.s2.ok && rt.tF + 2 * rt.h2 <= 64 && stepBytes <= 150 * 1024) {
            rt.stepFused = 1;
            if (stepBytes > rt.fusedBytes) rt.fusedBytes = stepBytes;
        }
        if (rt.fusedBytes > 150 * 1024) rt.tF = rt.stepFused = 0;
    }
    return rt;
}
// 1: the chain can run without the seven intermediate layers (launch_filters with travOnly).
bool filters_trav_only_ok(const FilterConsts& fc, const MapGeom& g) {
    const FilterRoute rt = filter_route(fc, g);
    return rt.tF != 0 && rt.stepFused != 0;
}
// The chain on `stream`: step heights, then normals + slope + roughness + second step window + weighted sum in one launch
// (filter_fused_kernel); launches of their own for whatever the lattice forms do not cover.  travOnly: only L.stepHeight and
// L.trav are valid pointers (filters_trav_only_ok must have said yes).
hipError_t launch_filters(const MapGeom& g, const FilterConsts& fc, const float* d_elev, const FilterLayers& L, bool travOnly, hipStream_t stream) {
    const dim3 grid((g.cols + kFT - 1) / kFT, (g.rows + kFT - 1) / kFT), block(256);
    const FilterRoute rt = filter_route(fc, g);
    if (travOnly && !(rt.tF && rt.stepFused)) return hipErrorInvalidValue;
    const auto fits = [](const void* fn, size_t bytes) -> bool {  // beyond 48 KB of dynamic LDS a kernel has to be told
        if (bytes <= 48 * 1024) return true;
        if (bytes > 150 * 1024) return false;
        return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes)) == hipSuccess;
    };
    float critDown = static_cast<float>(fc.stepCritical);
    if (static_cast<double>(critDown) > fc.stepCritical) critDown = std::nextafterf(critDown, -HUGE_VALF);
    // one row-run launch (kSecond false: step heights from the elevation; true: step + weighted sum from the step heights)
    const auto launch_runs = [&](auto second, int tcode, int H, const StepShape& sp, const float* src, double radius) -> bool {
        constexpr bool kSecond = decltype(second)::value;
        const int TR = tcode == 16 ? 16 : 32, TC = tcode == 32 ? 32 : 16;
        const size_t bytes = step_lds_bytes(H, sp.nClasses, TR, TC);
        const int tilesX = (g.cols + TC - 1) / TC, nTiles = tilesX * ((g.rows + TR - 1) / TR);
        const double crit = kSecond ? fc.stepCritical : 0.0;
        const float cd = kSecond ? critDown : 0.0f;
        const int nCrit = kSecond ? fc.stepCriticalCells : 1;
#define FPE_RUNS(TRR, TCC, HS)                                                                                                                          \
    do {                                                                                                                                                \
        if (!fits(reinterpret_cast<const void*>(filter_step_runs_kernel<kSecond, TRR, TCC, HS>), bytes)) return false;                                  \
        hipLaunchKernelGGL((filter_step_runs_kernel<kSecond, TRR, TCC, HS>), dim3(nTiles), dim3(TRR * TCC), bytes, stream, g, src, L, radius, H, sp, crit, cd, \
                           nCrit, tilesX, nTiles);                                                                                                      \
        return true;                                                                                                                                    \
    } while (0)
        if (!sp.ok) return false;
        if (tcode == 32) {
            if (!kSecond && H == 5) FPE_RUNS(32, 32, 5);
            if (!kSecond && H == 9) FPE_RUNS(32, 32, 9);
            FPE_RUNS(32, 32, 0);
        }
        if (tcode == 24) {
#ifdef FPE_FILTER_T1_ENV
            if (!kSecond && H == 5) FPE_RUNS(32, 16, 5);
            if (!kSecond && H == 9) FPE_RUNS(32, 16, 9);
#endif
            if (H == 17) FPE_RUNS(32, 16, 17);  // the published windows (0.08 m) at 0.5 cm
            FPE_RUNS(32, 16, 0);
        }
        FPE_RUNS(16, 16, 0);
#undef FPE_RUNS
    };
#ifdef FPE_FILTER_CHAIN_EXPERIMENT  // (measurement build: the flags from a process-wide buffer — one chain at a time)
    if (rt.tF == 16 && rt.stepFused && rt.h1 == rt.h2 && rt.s1.ok && std::getenv("FPE_FILTER_CHAIN")) {
        static unsigned* gFlags = nullptr;
        static unsigned gEpoch = 0;
        if (!gFlags) {
            if (hipMalloc(&gFlags, 4u << 20) != hipSuccess || hipMemset(gFlags, 0, 4u << 20) != hipSuccess) return hipErrorOutOfMemory;
        }
        if (rt.hN == 3 && rt.h2 == 5) return launch_chain_one<3, kFusedRows, 16, 5>(g, fc, d_elev, L, rt, critDown, travOnly ? 1 : 0, gFlags, ++gEpoch, stream);
        if (rt.hN == 6 && rt.h2 == 9) return launch_chain_one<6, kFusedRows, 16, 9>(g, fc, d_elev, L, rt, critDown, travOnly ? 1 : 0, gFlags, ++gEpoch, stream);
    }
#endif
    // 1. step heights (first window)
    if (!launch_runs(std::false_type{}, rt.t1, rt.h1, rt.s1, d_elev, fc.stepFirstRadius))  // (the walking kernel writes L.stepHeight only: fine for travOnly too)
        hipLaunchKernelGGL(filter_step1_kernel, grid, block, disc_lds_bytes(rt.h1), stream, g, d_elev, L, fc.stepFirstRadius, rt.h1);
    // 2. normals + slope + roughness [+ second step window + weighted sum]
    bool stepDone = false;
    if (rt.tF) {
        hipError_t e = hipErrorInvalidValue;
        // the published default chain at 2 cm and 1 cm (normals 0.05 m, second step window 0.08 m): the step window's halo is a
        // compile-time constant too
        if (rt.stepFused && rt.hN == 3 && rt.h2 == 5) e = launch_fused_one<3, kFusedRows, 16, 5>(g, fc, d_elev, L, rt, critDown, travOnly ? 1 : 0, stream);
        else if (rt.stepFused && rt.hN == 6 && rt.h2 == 9) e = launch_fused_one<6, kFusedRows, 16, 9>(g, fc, d_elev, L, rt, critDown, travOnly ? 1 : 0, stream);
        else switch (rt.hN) {
#define FPE_FUSED_CASE(HH, TT) case HH: e = launch_fused_one<HH, kFusedRows, TT>(g, fc, d_elev, L, rt, critDown, travOnly ? 1 : 0, stream); break;
            FPE_FUSED_CASE(1, 16) FPE_FUSED_CASE(2, 16) FPE_FUSED_CASE(3, 16) FPE_FUSED_CASE(4, 16) FPE_FUSED_CASE(5, 16) FPE_FUSED_CASE(6, 16)
            FPE_FUSED_CASE(7, 16) FPE_FUSED_CASE(8, 16) FPE_FUSED_CASE(9, 32) FPE_FUSED_CASE(10, 32) FPE_FUSED_CASE(11, 32) FPE_FUSED_CASE(12, 32)
#undef FPE_FUSED_CASE
            default: break;
        }
        if (e != hipSuccess) return e;
        stepDone = rt.stepFused != 0;
    } else {
        const int fuse = fc.roughnessRadius == fc.normalRadius ? 1 : 0;
        hipLaunchKernelGGL(filter_normals_kernel, grid, block, disc_lds_bytes(rt.hN), stream, g, d_elev, L, fc.normalRadius, rt.hN, fc.slopeCritical, fuse,
                           fc.roughnessCritical);
        if (!fuse)
            hipLaunchKernelGGL(filter_roughness_kernel, grid, block, disc_lds_bytes(rt.hR), stream, g, d_elev, L, fc.roughnessRadius, rt.hR, fc.roughnessCritical);
    }
    // 3. second step window + weighted sum, when it did not ride in the fused launch
    if (!stepDone && !launch_runs(std::true_type{}, rt.t2, rt.h2, rt.s2, static_cast<const float*>(L.stepHeight), fc.stepSecondRadius))
        hipLaunchKernelGGL(filter_step2_kernel, grid, block, disc_lds_bytes(rt.h2), stream, g, L, fc.stepSecondRadius, rt.h2, fc.stepCritical, critDown,
                           fc.stepCriticalCells);
    return hipGetLastError();
}
