// fpe_filters_fused.hpp — round 5 of the producer's filters (included by fpe_filters.hpp inside its
// anonymous namespace): NormalVectorsFilter + SlopeFilter + RoughnessFilter by row moments with the halo as a TEMPLATE
// parameter — the moment phase of filter_fused_kernel, which also runs the StepFilter's second window over the same cells.
//
// All kernels of the chain are bound by the SIMDs' instruction issue (profiles/round4_filters.txt, round5_filters.txt:
// VALU-active = 4 clocks x VALU instructions per wavefront), so the lever is instructions per cell:
//   * H (halo) and the tile TR x TC are compile-time: the loop over the disc's rows is unrolled, a row's LDS addresses are
//     the thread's base plus an IMMEDIATE, the row offset is a constant operand of the integer multiply-adds;
//   * a row's members need the iterator's own test only where an offset lies ON the circle (step_shape's edge rows: none at
//     2 cm / r 0.05, seven of eleven rows at 1 cm) — the other rows take their robust half-width from an SGPR;
//   * the six prefix arrays of round 4 are two arrays of 16-byte records ({n | sum c << 16, sum c^2, sum z'} and {sum z'^2,
//     sum c z'}): four ds_read_b128 per row instead of twelve scalar reads, two integer subtractions instead of three;
//   * the prefix scans run one wavefront per record field group, lane = tile row (no divergence inside a wavefront); tiles
//     of 32 rows use 38-56 of the 64 lanes and halve the scans per cell against 16 x 16;
//   * the closing arithmetic uses refined reciprocals and an fdlibm-style acos instead of IEEE division / library calls, the
//     eigenvalue iteration starts from Halley's step and stops when the error LEFT by a step is below the tolerance;
//   * the workgroup -> tile map is XCD-aware (a contiguous band of tiles per XCD: a tile's halo is its neighbours' interior,
//     read through the same L2 — round 4 measured 6.2 x the layer in fabric reads, round 5 1.07 x).
// The f64 arithmetic on the moments is the round-4 kernel's, operation for operation (same prefix sums, same differences,
// same recentring); cells the moment form cannot decide still take the literal walks (normals_cell_exact).
#pragma once

struct __attribute__((aligned(16))) MomentA {
    uint32_t nC;  // valid cells left of this column | (sum of their tile columns) << 16
    uint32_t CC;  // sum of c^2
    double z;     // sum of z' = z - z0
};
struct __attribute__((aligned(16))) MomentB {
    double zz, cz;  // sum of z'^2, sum of c z'
};
static_assert(sizeof(MomentA) == 16 && sizeof(MomentB) == 16, "moment records are one ds_read_b128 each");

template <int H, int TR, int TC>
struct FusedLayout {
    static constexpr int WR = TR + 2 * H, WC = TC + 2 * H, W1 = WC + 1, D = 2 * H + 1;
    static constexpr size_t discBytes = (static_cast<size_t>(WR) * WC * 4 + static_cast<size_t>(WR + WC) * 8 + static_cast<size_t>(TR + TC) * D * 8 +
                                         2 * (TR + TC) * 4 + 16 + 15) & ~static_cast<size_t>(15);
    static constexpr size_t recBytes = static_cast<size_t>(WR) * W1 * 16;  // one record array: a record per (tile row, column prefix)
};
__host__ inline size_t fused_moment_bytes(int H, int TR, int TC) {
    const int WR = TR + 2 * H, W1 = TC + 2 * H + 1;
    return ((disc_lds_bytes(H, TR, TC) + 15) & ~static_cast<size_t>(15)) + 2 * static_cast<size_t>(WR) * W1 * 16;
}

// Workgroup -> tile, XCD-aware and bijective (the dispatcher is observed to place block b on XCD b % 8; a wrong guess costs
// speed only): XCD x takes a contiguous band of the row-major tile order.
__device__ __forceinline__ void xcd_tile(int tilesX, int nTiles, int& ty, int& tx) {
    const unsigned b = blockIdx.x, q = static_cast<unsigned>(nTiles) >> 3, rem = static_cast<unsigned>(nTiles) & 7u, x = b & 7u, k = b >> 3;
    const unsigned id = (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + k;
    ty = static_cast<int>(id / static_cast<unsigned>(tilesX));
    tx = static_cast<int>(id - static_cast<unsigned>(ty) * static_cast<unsigned>(tilesX));
}

// acos on [0, 1] (the float normal's z component, turned upwards) in f64: the rational approximation of fdlibm's e_acos.c
// (Sun Microsystems; pS / qS are its published coefficients), the quotient by a refined reciprocal and the square root by a
// refined reciprocal square root instead of IEEE division / sqrt sequences.  Within 1 ulp (f64) of the host's acos on 2 x 10^6
// random floats (scratch: no float layer value changed); ~35 instructions where the library call takes ~100.
__device__ __forceinline__ double sqrt_refined(double v) {  // v >= 0, finite; 0 -> 0
    double y = __builtin_amdgcn_rsq(v);
    double g = v * y, h = 0.5 * y;
    double e = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, e, g);
    h = __builtin_fma(h, e, h);
    const double dd = __builtin_fma(-g, g, v);
    g = __builtin_fma(dd, h, g);
    return v > 0.0 ? g : 0.0;
}
// A double constant as a SCALAR register pair (two s_mov: scalar instructions, which issue beside the vector unit's) instead of the
// two v_mov_b32 per use the compiler materialises it with in a kernel whose vector unit is the bound — a vector instruction may read one
// scalar operand, and a Horner step has exactly one constant.  86 of the 961 static vector instructions of the 2 cm instantiation's
// closing arithmetic were such moves (round 6).
__device__ __forceinline__ double in_sgpr(double v) {
    asm("" : "+s"(v));
    return v;
}
__device__ __forceinline__ double acos_unit(double x) {
    const double pS0 = in_sgpr(1.66666666666666657415e-01), pS1 = in_sgpr(-3.25565818622400915405e-01), pS2 = in_sgpr(2.01212532134862925881e-01),
                 pS3 = in_sgpr(-4.00555345006794114027e-02), pS4 = in_sgpr(7.91534994289814532176e-04), pS5 = in_sgpr(3.47933107596021167570e-05),
                 qS1 = in_sgpr(-2.40339491173441421878e+00), qS2 = in_sgpr(2.02094576023350569471e+00), qS3 = in_sgpr(-6.88283971605453293030e-01),
                 qS4 = in_sgpr(7.70381505559019352791e-02);
    const bool hiHalf = x >= 0.5;
    const double z = hiHalf ? (1.0 - x) * 0.5 : x * x;
    const double p = z * __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, pS5, pS4), pS3), pS2), pS1), pS0);
    const double q = __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, qS4, qS3), qS2), qS1), 1.0);
    const double rr = p * rcp_refined(q);
    const double s = sqrt_refined(z);
    const double up = 2.0 * __builtin_fma(s, rr, s);                                                // x >= 0.5: 2 asin(sqrt((1 - x) / 2))
    const double low = in_sgpr(1.57079632679489655800e+00) - (x - (in_sgpr(6.12323399573676603587e-17) - x * rr));   // x < 0.5: pi/2 - asin(x)
    return hiHalf ? up : low;
}

// Scatter moments of a cell's disc -> float normal, slope and roughness values: the closing arithmetic of
// filter_normals_moments_kernel (same thresholds for handing a cell to the literal walks) with the divisions as products by
// refined reciprocals and acos / sqrt as above — every value within an f64 ulp or two of the round-4 kernel's, far inside the
// float layers' rounding (tests/test_gpu_filters.py: one float ulp, >= 99.99 % of the cells bit-identical to the oracle).
// Returns true when the cell has to take the literal walks instead (the outputs are then not set): the caller queues it for
// the workgroup's walking phase (walk_phase below).
#ifndef FPE_NORMAL_MIN_GAP
#define FPE_NORMAL_MIN_GAP 3e-4
#endif
#ifndef FPE_NORMAL_MIN_COMPONENT_GAP
#define FPE_NORMAL_MIN_COMPONENT_GAP 1e-7
#endif
#ifdef FPE_DBG_COUNT_WALKS
__device__ unsigned g_walkWhy[4];
__device__ double g_walkDbg[64][8];
#endif
template <bool kSweeps>
__device__ __forceinline__ bool normals_from_moments(const MapGeom& g, double slopeCritical, double roughCritical, double invSlopeCritical,
                                                     double invRoughCritical, int N, int Sc, int Scc, int Sv, int Svv, int Svc, double Sz, double Szz, double Scz,
                                                     double Svz, bool normalsStored, bool flat, double zAbs, float& ox, float& oy, float& oz, float& os, float& orough) {
    const double nd = static_cast<double>(N);
    // `flat`: the first step window (a disc at least as wide as this one, same centre) found max == min — every member has the
    // centre's elevation, bit for bit.  The published filters then leave no choice: the covariance's z row is rounding noise of the
    // mean (|dz| <= (N + 1) eps |z|), its smallest eigenvalue fails the rank test against any xy scatter (and 0 > 0 fails it for
    // one member), the normal is the z axis exactly, acos(1.0f) = 0, and the roughness sqrt(N dz^2 / (N - 1)) vanishes in the float
    // result as long as it stays below 2^-26 of the critical value (two members at least: one gives 0 / 0 and takes the walk, as
    // do elevations too large for the bound).  Flat synthetic ground — every cell rank-deficient, every cell on the literal
    // walks until now — and the plateaus between the test terrain's risers stay on this path, in both chains.
    if (flat && N >= 2 && zAbs * (nd + 1.0) * 2.5e-16 < 1.4e-8 * roughCritical) {
        ox = 0.0f;
        oy = 0.0f;
        oz = 1.0f;
        os = 0.0 < slopeCritical ? 1.0f : 0.0f;
        orough = 1.0f;
        return false;
    }
    const double invN = rcp_refined(nd);
    const double Avv = static_cast<double>(N * Svv - Sv * Sv) * invN, Acc = static_cast<double>(N * Scc - Sc * Sc) * invN;
    const double Avc = static_cast<double>(N * Svc - Sv * Sc) * invN;
    const double Avz = Svz - static_cast<double>(Sv) * Sz * invN, Acz = Scz - static_cast<double>(Sc) * Sz * invN;
    const double Azz = fmax(Szz - Sz * Sz * invN, 0.0);
    const double res = g.res, res2 = res * res;
    const double a00 = res2 * Avv, a01 = res2 * Avc, a02 = -(res * Avz), a11 = res2 * Acc, a12 = -(res * Acz), a22 = Azz;
    double ex, ey, ez, eigS, eigL, gapRel;
    bool walk = false;
    // A cell whose iteration fails (no convergence inside the cap, eigenvalues within 1e-6, degenerate input):
    //   kSweeps (the 32 x 32 tiles of halos above eight cells: 0.5 cm maps) — the Jacobi sweeps on the moment matrix right here, and
    //     the literal walks only when its eigenvalues are closer than 1e-3, as in rounds 4-5: a walk over a 23-row disc costs a
    //     workgroup tens of microseconds, 2 118 cells of the 0.5 cm probe map fail the iteration and 78 of them have to walk
    //     (walking them all: 0.97 -> 1.27 ms), and at one 1 024-thread workgroup per CU registers do not bound the occupancy;
    //   otherwise (32 x 16 tiles: the published chain at 2 cm and 1 cm) — the LITERAL WALKS, the oracle's own arithmetic: the sweeps'
    //     eighteen registers of eigenvectors were live beside everything the cell still needs, and at the 2 cm instantiation's 64
    //     registers the compiler spilled the iteration's results around the (never taken) branch in EVERY cell — 32 B of scratch
    //     stores per thread and launch (47 MB written per 1000 x 1000 launch for 4 MB of layers).  127 instead of 71
    //     cells of the 1 cm probe map walk, none at 2 cm: no measurable time (profiles/round6_filters.txt).
    if constexpr (kSweeps) {
        if (!normal_newton(a00, a01, a02, a11, a12, a22, ex, ey, ez, eigS, eigL, gapRel)) {
            normal_from_scatter(a00, a01, a02, a11, a12, a22, ex, ey, ez, eigS, eigL);
            walk = !(gapRel > 1e-3);  // eigenvalues this close: the sweeps on THIS matrix and on the oracle's (another order of summation) part ways
        }
    } else {
        walk = !normal_newton(a00, a01, a02, a11, a12, a22, ex, ey, ez, eigS, eigL, gapRel);
    }
    // The matrix entries carry ~1e-15 of their scale (prefix differences instead of the oracle's two-pass sums), the eigenvector
    // that error over the relative gap to the next eigenvalue: dv ~ 1e-15 / gap.  A float component c is allowed its last bit
    // (the tests' bar: one ulp, 6e-8 |c|) but not two: dv must stay well below 6e-8 |c|, i.e. |c| x gap well above 1.7e-8.
    // Cells under 1e-7 take the literal walks — a component at rounding-noise level (symmetric neighbourhoods: exactly 0 here,
    // ~1e-17 by the oracle's order), or a small one beside a narrow gap (beside the synthetic terrain's risers the gap is 0.005;
    // campaign seeds 2512287, 2514166: components of 7e-6 and 2e-6 came out two and nine float ulps off) — and so do (nearly)
    // rank-deficient matrices (exact planes, flat synthetic ground, fewer than three members: the rank test and exact zeros
    // depend on the order of operations).  With the usual gap of 0.25 that is a component below 4e-7.  A wavefront walks when
    // ANY of its 64 cells asks for it and a walk is some thirty times a cell's usual work, so the product is as low as the
    // arithmetic allows (at 3e-7: 1.3e-4 of the synthetic terrain's cells, a fifth of the chain's time).
    // normalsStored false (the traversability-only chain): nobody reads the x and y components.  Slope and roughness see the
    // eigenvector's error at SECOND order (the z component of a near-vertical normal; n^T A n about its minimiser) or, on a steep
    // face, at dv = 1e-13 / gap against the z component's float ulp of 3e-8 — the small-component rule has nothing to protect, and
    // the cells it sends to the walks (symmetric neighbourhoods beside the synthetic terrain's risers: 9 us of the chain's 65 at
    // 2 cm, 32 of 300 at 1 cm, all of it the tail of the few workgroups that hold such a cell) stay on the moment path.  A gap
    // below 1e-4 is a genuine degeneracy (members on a line): the oracle's normal is then decided by its own rounding — walk.
    const double cMin = fmin(fmin(fabs(ex), fabs(ey)), fabs(ez));
#ifdef FPE_DBG_COUNT_WALKS
    if (walk) {
        const unsigned k = atomicAdd(&g_walkWhy[0], 1u);
        if (k < 64) { g_walkDbg[k][0] = static_cast<double>(N); g_walkDbg[k][1] = a00; g_walkDbg[k][2] = a11; g_walkDbg[k][3] = a22; g_walkDbg[k][4] = a01; g_walkDbg[k][5] = a02; g_walkDbg[k][6] = a12; g_walkDbg[k][7] = gapRel; }
    }
    else if (!(eigS > 1e-10 * eigL)) atomicAdd(&g_walkWhy[1], 1u);
    else if (normalsStored ? !(cMin * fmin(gapRel, 0.3) > FPE_NORMAL_MIN_COMPONENT_GAP) : !(gapRel > 1e-4)) atomicAdd(&g_walkWhy[2], 1u);
    else if (normalsStored && !(gapRel > FPE_NORMAL_MIN_GAP)) atomicAdd(&g_walkWhy[3], 1u);
#endif
    if (walk || !(eigS > 1e-10 * eigL)) return true;
    if (normalsStored ? !(cMin * fmin(gapRel, 0.3) > FPE_NORMAL_MIN_COMPONENT_GAP) : !(gapRel > 1e-4)) return true;
    // The stored normal of a cell whose two SMALL eigenvalues nearly coincide (a steep smooth face under a symmetric disc: both are
    // the lattice's own second moment) turns within their plane by dA / gap, and dA — prefix differences over a tile row — is ~3e-14
    // of the scale, not 1e-15: at a gap of 3.6e-5 every component is within 1e-9 of where the oracle puts it and rounds the other
    // way in 3 % of the cases (campaign seed 10090970, round 6: five such cells on one 56 x 63 map, over the cap of the tests'
    // `loose` class; mpmath says the oracle's rounding is the right one in all five).  Below 3e-4 the cell walks.
    if (normalsStored && !(gapRel > FPE_NORMAL_MIN_GAP)) return true;
    {
        // (Measured and not built, round 6: sending a cell to the walks when a component lies within 2e-15 .. 3e-14 / gap of the
        // midpoint between two floats — the `loose` class of the tests at its source.  One walk holds a 512-cell workgroup: +10 %
        // (2e-15) to +37 % (3e-14) on the 1 cm chain, and campaign seed 9184403's row of ten cells was still outside the widest.)
        ox = static_cast<float>(ex);
        oy = static_cast<float>(ey);
        oz = static_cast<float>(ez);
        const double slope = acos_unit(static_cast<double>(oz));  // SlopeFilter reads the float layer
        const double slopeRem = 1.0 - slope * invSlopeCritical;
        os = slope < slopeCritical ? static_cast<float>(slopeRem) : 0.0f;
        // RoughnessFilter: the plane through the mean with the FLOAT normal: sum of squared distances = n^T A n
        const double nx = ox, ny = oy, nz = oz;
        const double q = nx * (nx * a00 + 2.0 * (ny * a01 + nz * a02)) + ny * (ny * a11 + 2.0 * (nz * a12)) + nz * (nz * a22);
        const double roughness = sqrt_refined(fmax(q, 0.0) * rcp_refined(nd - 1.0));  // (N >= 3 here: fewer members are rank-deficient)
        const double roughRem = 1.0 - roughness * invRoughCritical;
        orough = roughness < roughCritical ? static_cast<float>(roughRem) : 0.0f;
        // A value within 1e-5 (roughness) / 1e-7 (slope) of its critical value is the REMAINDER of a cancellation: the moment form's
        // roughness is good to ~1e-13 of the critical value (n^T A n is itself a small difference of large moments), which is 2 float
        // ulps of a remainder of 1e-6 — and the side of the critical value (0 exactly, or a tiny positive value) is at stake too.
        // Campaign seed 9014219 (round 6): roughness 9.14e-7, 1.14e-13 = two ulps from the oracle, identical normals, on the round-5
        // kernels and on these.  The bar is frozen: such a cell takes the literal walks (a few cells per million).
        if (fabs(roughRem) < 1e-5 || fabs(slopeRem) < 1e-7) return true;
    }
    return false;
}

// The moment phase of one tile: tables and source tile (disc_setup, one barrier), the prefix records (one barrier), then the
// calling thread's cell.  `live`: the thread's cell is inside the map (every thread takes part in the barriers).
template <int H, int TR, int TC>
__device__ __forceinline__ bool moments_phase(char* ldsRaw, const MapGeom& g, const float* __restrict__ elev, int ti0, int tj0, double r, bool normalsStored, bool flat,
                                              const StepShape& sp, double slopeCritical, double roughCritical, double invSlopeCritical,
                                              double invRoughCritical, bool live, float& ox, float& oy, float& oz, float& os, float& orough) {
    using Lay = FusedLayout<H, TR, TC>;
    constexpr int WR = Lay::WR, W = Lay::WC, W1 = Lay::W1, D = Lay::D;
    static_assert(W <= 64 && WR <= 64, "a tile row is one wavefront load; the scans run one lane per tile row");
    static_assert(TR * TC >= 128, "two wavefronts scan the record fields");
    const DiscLds d = disc_carve(ldsRaw, H, TR, TC);
    MomentA* const PA = reinterpret_cast<MomentA*>(ldsRaw + Lay::discBytes);
    MomentB* const PB = reinterpret_cast<MomentB*>(ldsRaw + Lay::discBytes + Lay::recBytes);
    // A disc without an offset on the circle (the published chain at 2 cm) reads no table on this path — the rows' half-widths are
    // scalars of the shape: the tile alone, and walk_phase builds the tables for the workgroups that walk (round 6: the tables were
    // ~150 vector instructions of the first wavefront and ~40 of five more, in every workgroup).
    disc_setup<true, TR, TC, H>(d, g, elev, ti0, tj0, r, sp.edgeRows == 0ull);
    FPE_TL_MARK(11);
    // z0, the elevation the prefix sums are taken about: a VALID cell near the tile's centre (the sums' cancellation grows with the
    // square of the largest |z - z0| in the tile: the centre halves it against a corner, and a hole at one fixed cell must not
    // leave z0 = 0 — a map at an altitude of 100 m would then sum squares of 10^4).  Every wavefront looks at the same 64 cells —
    // the centre first, then a diagonal sweep through the interior — and takes the first valid one.
    double z0 = 0.0;
    {
        const int lane = threadIdx.x & 63;
        const int k = (lane * 37) % (TR * TC);  // lane 0 -> offset 0 = the centre; the others spread over the interior
        const int rr = (TR / 2 + k / TC) % TR, cq = (TC / 2 + k % TC) % TC;
        const float zs = d.tile[(H + rr) * W + H + cq];
        const unsigned long long fin = __ballot(zs == zs);
        if (fin != 0ull) z0 = static_cast<double>(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(zs), __builtin_ctzll(fin))));
    }
    {   // prefix records over the tile's columns: lane = tile row; the first wavefront scans the integer fields, the second the three
        // sums of z' — one conversion and one recentring per cell for all three (rounds 4-5 had a wavefront per sum, each converting
        // and recentring the cell again: 3 x 8 vector instructions per column where this takes 13; same additions in the same order)
        const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
        if (lane < WR && grp < 2) {
            const float* src = d.tile + lane * W;
            if (grp == 0) {
                uint2* dst = reinterpret_cast<uint2*>(PA + lane * W1);
                uint32_t nC = 0, CC = 0;
                dst[0] = make_uint2(0u, 0u);
#pragma unroll 4
                for (int c = 0; c < W; ++c) {
                    const float z = src[c];
                    const bool v = z == z;
                    nC += v ? (1u | (static_cast<uint32_t>(c) << 16)) : 0u;
                    CC += v ? static_cast<uint32_t>(c * c) : 0u;
                    dst[2 * (c + 1)] = make_uint2(nC, CC);
                }
            } else {
                typedef double f64x2s __attribute__((ext_vector_type(2)));
                double* dstZ = &PA[lane * W1].z;
                f64x2s* dstB = reinterpret_cast<f64x2s*>(PB + lane * W1);
                double accZ = 0.0, accZZ = 0.0, accCZ = 0.0;
                dstZ[0] = 0.0;
                dstB[0] = f64x2s{0.0, 0.0};
#pragma unroll 4
                for (int c = 0; c < W; ++c) {
                    const float z = src[c];
                    const double zz = z == z ? static_cast<double>(z) - z0 : 0.0;
                    accZ += zz;
                    accZZ += zz * zz;
                    accCZ += static_cast<double>(c) * zz;
                    dstZ[2 * (c + 1)] = accZ;
                    dstB[c + 1] = f64x2s{accZZ, accCZ};
                }
            }
        }
    }
    __syncthreads();
    FPE_TL_MARK(12);
    const int li = threadIdx.x / TC, lj = threadIdx.x % TC;
    const int i = ti0 + li, j = tj0 + lj;
    const float nanf = __builtin_nanf("");
    ox = oy = oz = os = orough = nanf;
    if (!live || !isfinite(d.tile[(li + H) * W + lj + H])) return false;
    const double r2 = r * r;
    const int i0 = d.bi0[li], i1 = d.bi1[li];
    const int maxL = j - d.bj0[lj], maxR = d.bj1[lj] - j;
    const int dyC = lj * D + H;
    const int cc = lj + H;  // the cell's own tile column; its tile row is li + H, the disc's first row li
    // byte offset of the record (first row of the disc, own column); a row adds (o + H) * W1 * 16 as an immediate of the
    // LDS instruction, its half-width comes from an SGPR: one VALU instruction per address
    const char* const recA = reinterpret_cast<const char*>(PA) + (li * W1 + cc) * 16;
    constexpr int kAB = static_cast<int>(Lay::recBytes);  // record B of an index lies kAB bytes behind record A
    uint32_t AnC = 0, ACC = 0;
    int Sv = 0, Svv = 0, SvC = 0;
    double Sz = 0.0, Szz = 0.0, Scz = 0.0, Svz = 0.0;
    const double ccD = static_cast<double>(cc);
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef double f64x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int oo = 0; oo < D; ++oo) {
        const int o = oo - H;
        const int w0 = sp.rowW[oo];
        const bool edge = ((sp.edgeRows >> oo) & 1u) != 0u;
        if (w0 < 0 && !edge) continue;  // (wave-uniform: the row holds no member for any cell)
        int lo, hi;                      // records [lo, hi) of the row, as byte offsets from recA + the row's immediate
        if (!edge) {
            lo = -16 * w0;
            hi = 16 * w0 + 16;
        } else {
            // a row with an offset on the circle: the iterator's own tests decide the row (offset (o, 0) on the circle) and
            // the column either side of the robust interval — filter_normals_moments_kernel's per-row logic
            const double a = d.dx2[li * D + oo];
            const int ii = i + o;
            const bool rowIn = ii >= i0 && ii <= i1 && a <= r2;
            const int k1 = w0 + 1 < H ? w0 + 1 : H;
            const int wR = min(w0 < 0 ? 0 : w0 + (a + d.dy2[dyC + k1] <= r2 ? 1 : 0), maxR);
            const int wL = min(w0 < 0 ? 0 : w0 + (a + d.dy2[dyC - k1] <= r2 ? 1 : 0), maxL);
            lo = -16 * wL;
            hi = rowIn ? 16 * wR + 16 : lo;
        }
        const char* const rowRec = recA + oo * W1 * 16;
        const u32x4 aH = *reinterpret_cast<const u32x4*>(rowRec + hi), aL = *reinterpret_cast<const u32x4*>(rowRec + lo);
        const f64x2 bH = *reinterpret_cast<const f64x2*>(rowRec + kAB + hi), bL = *reinterpret_cast<const f64x2*>(rowRec + kAB + lo);
        const uint32_t dnC = aH.x - aL.x;
        const int n = static_cast<int>(dnC & 0xFFFFu), cRow = static_cast<int>(dnC >> 16);
        AnC += dnC;
        ACC += aH.y - aL.y;
        Sv += o * n;
        Svv += o * o * n;
        SvC += o * cRow;
        const double zH = __hiloint2double(static_cast<int>(aH.w), static_cast<int>(aH.z)), zL = __hiloint2double(static_cast<int>(aL.w), static_cast<int>(aL.z));
        const double z = zH - zL, zz = bH.x - bL.x;
        const double cz = (bH.y - bL.y) - ccD * z;
        Sz += z;
        Szz += zz;
        Scz += cz;
        Svz += static_cast<double>(o) * z;
    }
    FPE_TL_MARK(13);
    const int N = static_cast<int>(AnC & 0xFFFFu), SC = static_cast<int>(AnC >> 16);
    const int Sc = SC - cc * N;
    const int Scc = static_cast<int>(ACC) - 2 * cc * SC + cc * cc * N;
    const int Svc = SvC - cc * Sv;
    return normals_from_moments<(TR * TC > 512)>(g, slopeCritical, roughCritical, invSlopeCritical, invRoughCritical, N, Sc, Scc, Sv, Svv, Svc, Sz, Szz, Scz, Svz, normalsStored, flat,
                                fabs(static_cast<double>(d.tile[(li + H) * W + lj + H])), ox, oy, oz, os, orough);
}

// The cell's iterator walk with the rows' column intervals taken from the lattice shape — the robust half-width of step_shape
// and, on rows with an offset on the circle, the iterator's own tests for the row and the column either side (the moment
// phase's logic: the same members) — instead of disc_walk's stepping from the previous row's interval, and with a row's cells
// FETCHED TOGETHER: the 2 H - 1 columns a row can hold are static slots whose LDS reads issue back to back, the visitor then
// runs over them in order with the cells outside the row's interval masked to NaN (a hole: every visitor skips holes).  A
// lone lane walking a cell is bound by LDS round trips, not by arithmetic: disc_walk's chain of dependent reads per row and
// per member made a walked cell cost 13 us at 2 cm and 32 us at 1 cm.  Same members, same order (rows outer, columns inner);
// cells outside the map are NaN in the tile already.  rowW: the shape's half-widths in LDS (the row loop is not unrolled).
template <int H, class F>
__device__ __forceinline__ void disc_walk_rows(const DiscLds& d, const int8_t* rowW, unsigned long long edgeRows, int li, int lj, int ti0, int tj0, double r2,
                                               F&& f) {
    constexpr int D = 2 * H + 1, kSlots = 2 * H - 1;  // a robust half-width is at most H - 2, one more column by the edge test
    const int WC = d.WC;
    const int i = ti0 + li, j = tj0 + lj;
    const int i0 = d.bi0[li], i1 = d.bi1[li];
    const int maxL = j - d.bj0[lj], maxR = d.bj1[lj] - j;
    const int dyC = lj * D + H;
    const double* const yRow = d.yP + lj + H;
    // (not unrolled: the walks are the kernel's cold path — one cell in 10^4 — and three unrolled copies of the visitor per row
    // were what pushed the 2 cm instantiation past its 64 registers)
#ifndef FPE_WALK_UNROLL
#pragma unroll 1
#endif
    for (int oo = 0; oo < D; ++oo) {
        const int o = oo - H;
        const int w0 = rowW[oo];
        const bool edge = ((edgeRows >> oo) & 1u) != 0u;
        if (w0 < 0 && !edge) continue;
        int wL = w0, wR = w0;
        bool rowIn = true;
        if (edge) {
            const double a = d.dx2[li * D + oo];
            const int ii = i + o;
            rowIn = ii >= i0 && ii <= i1 && a <= r2;
            const int k1 = w0 + 1 < H ? w0 + 1 : H;
            wR = min(w0 < 0 ? 0 : w0 + (a + d.dy2[dyC + k1] <= r2 ? 1 : 0), maxR);
            wL = min(w0 < 0 ? 0 : w0 + (a + d.dy2[dyC - k1] <= r2 ? 1 : 0), maxL);
        }
        if (!rowIn) wR = -1 - wL;  // empty interval
        const int ri = li + oo;
        const double x = d.xP[ri];
        const float* const zRow = d.tile + ri * WC + lj + H;
        constexpr int kChunk = 8;
#pragma unroll
        for (int c0 = 0; c0 < kSlots; c0 += kChunk) {
            float zz[kChunk];
            double yy[kChunk];
#pragma unroll
            for (int k = 0; k < kChunk; ++k) {
                const int dj = c0 + k - (H - 1);
                if (c0 + k < kSlots) {
                    zz[k] = zRow[dj];
                    yy[k] = yRow[dj];
                }
            }
#pragma unroll
            for (int k = 0; k < kChunk; ++k) {
                const int dj = c0 + k - (H - 1);
                if (c0 + k < kSlots) f(x, yy[k], (dj >= -wL && dj <= wR) ? zz[k] : __builtin_nanf(""));
            }
        }
    }
}
// normals_cell_exact (fpe_filters.hpp) on disc_walk_rows: the published filters' three passes, expression for expression.
// stash: four doubles of LDS of the calling lane's own (the walking phase's list area has them to spare): the mean and the member
// count wait there while the Jacobi sweeps run, and the cell's coordinates pass through an empty asm before every walk so that
// nothing derived from them is carried from one walk to the next — the sweeps' eighteen registers of eigenvectors beside three
// walks' worth of addresses were what the 2 cm instantiation's 64 registers could not hold (13 registers of scratch, cold but
// there; round 6: none).
template <int H>
__device__ __forceinline__ void normals_cell_exact_rows(const DiscLds& d, const int8_t* rowW, unsigned long long edgeRows, int li, int lj, int ti0, int tj0, double r, double slopeCritical,
                                                        double roughCritical, double* stash, float& ox, float& oy, float& oz, float& os, float& orough) {
    const double r2 = r * r;
    int np = 0;
    double sx = 0.0, sy = 0.0, sz = 0.0;
    disc_walk_rows<H>(d, rowW, edgeRows, li, lj, ti0, tj0, r2, [&](double x, double y, float z) {
        if (isfinite(z)) { ++np; sx += x; sy += y; sz += static_cast<double>(z); }
    });
    double nd = static_cast<double>(np);
    double mx = sx / nd, my = sy / nd, mz = sz / nd;
    stash[3] = nd;  // (not needed before the roughness: it waits in LDS from here)
    double a00 = 0.0, a01 = 0.0, a02 = 0.0, a11 = 0.0, a12 = 0.0, a22 = 0.0;
    asm volatile("" : "+v"(li), "+v"(lj));
    disc_walk_rows<H>(d, rowW, edgeRows, li, lj, ti0, tj0, r2, [&](double x, double y, float z) {
        if (isfinite(z)) {
            const double dx = x - mx, dy = y - my, dz = static_cast<double>(z) - mz;
            a00 += dx * dx; a01 += dx * dy; a02 += dx * dz;
            a11 += dy * dy; a12 += dy * dz; a22 += dz * dz;
        }
    });
    stash[0] = mx;
    stash[1] = my;
    stash[2] = mz;
    double ex, ey, ez, wS, wL;
    normal_from_scatter(a00, a01, a02, a11, a12, a22, ex, ey, ez, wS, wL);
    asm volatile("" : "+v"(li), "+v"(lj), "+v"(stash));
    mx = stash[0];
    my = stash[1];
    mz = stash[2];
    nd = stash[3];
    ox = static_cast<float>(ex);
    oy = static_cast<float>(ey);
    oz = static_cast<float>(ez);
    const double slope = acos(static_cast<double>(oz));  // SlopeFilter reads the float layer
    os = slope < slopeCritical ? static_cast<float>(1.0 - slope / slopeCritical) : 0.0f;
    const double normalX = ox, normalY = oy, normalZ = oz;  // RoughnessFilter::update with the float normals just written
    const double planeParameter = mx * normalX + my * normalY + mz * normalZ;
    double sum = 0.0;
    disc_walk_rows<H>(d, rowW, edgeRows, li, lj, ti0, tj0, r2, [&](double x, double y, float z) {
        if (isfinite(z)) {
            const double dist = normalX * x + normalY * y + normalZ * static_cast<double>(z) - planeParameter;
            sum += dist * dist;
        }
    });
    const double roughness = sqrt(sum / (nd - 1.0));
    orough = roughness < roughCritical ? static_cast<float>(1.0 - roughness / roughCritical) : 0.0f;
}

// ---- the literal walks of the cells the moment form cannot decide, as a phase of their own ---------------------------------
// Inlined into the cell loop (rounds 2-4) the walks cost the kernel a fifth of its time although one cell in 10^4 takes them:
// three disc_walk bodies and the Jacobi sweeps raise the register count past the six-wavefront budget (88 registers: two
// workgroups per CU instead of three), and a wavefront walks when ANY of its 64 cells asks for it.  Now a cell that needs the
// walks is QUEUED (workgroup vote; the list lives in the LDS the prefix records no longer need) and the queue is walked
// COMPACTED, one queued cell per lane: flat synthetic ground — every cell rank-deficient — fills every lane (1000 x 1000 flat
// cells at 2 cm: 0.15 ms, as before), natural terrain leaves a workgroup in a few hundred with one or two lanes to walk, and
// the registers the walks need beyond the budget are spilled in this phase only.  (A whole wavefront per queued cell, members
// side by side and the ordered sums on v_readlane operands, was tried first: 20 us per cell — 3.4 ms for the flat map.)
// filter_fused_kernel's argument list as a struct (the argument segment's layout): the walking phase — the kernel's cold path — reads
// what it needs from there AGAIN, behind its vote, through a pointer the optimiser cannot see through.  Held in scalar registers from
// the kernel's entry to the walk, the same values cost the 2 cm instantiation (64 registers) twenty v_writelane in EVERY wavefront's
// prologue and a vector register for the spill slots (round 6: 1 081 -> 1 06x vector instructions per wavefront).
struct FusedKernArgs {
    MapGeom g;
    const float* elev;
    FilterLayers L;
    double rN, slopeCritical, roughCritical, invSlopeCritical, invRoughCritical;
    StepShape sN;
    double r2nd;
    int h2nd;
    StepShape s2;
    double stepCritical;
    float critDown;
    int nCritical, kStepFlags, travOnly, tilesX, nTiles;
};
template <int H, int TR, int TC, bool kChain = false>
__device__ __forceinline__ void walk_phase(char* ldsRaw, bool needWalk, float stepOut, int tyIn = 0, int txIn = 0) {
    using Lay = FusedLayout<H, TR, TC>;
#ifdef FPE_NO_WALK  // (measurement builds only: what the kernel costs without its walking phase)
    return;
#endif
    if (!__syncthreads_or(needWalk ? 1 : 0)) return;  // (also: every thread is done with the prefix records)
    typedef const FusedKernArgs __attribute__((address_space(4))) * FusedArgPtr;
    FusedArgPtr ka4 = (FusedArgPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka4));
    const FusedKernArgs* ka = (const FusedKernArgs*)ka4;
    const MapGeom& g = ka->g;
    const FilterLayers& L = ka->L;
    const StepShape& sp = ka->sN;
    const double r = ka->rN, slopeCritical = ka->slopeCritical, roughCritical = ka->roughCritical;
    const int kStep = ka->kStepFlags & 1, travOnly = ka->travOnly;
    int tyW = tyIn, txW = txIn;  // (filter_chain_kernel: the tile is not a function of the workgroup number alone)
    if constexpr (!kChain) xcd_tile(ka->tilesX, ka->nTiles, tyW, txW);
    const int ti0 = tyW * TR, tj0 = txW * TC;
    const DiscLds d = disc_carve(ldsRaw, H, TR, TC);
    int* const count = reinterpret_cast<int*>(ldsRaw + Lay::discBytes);
    uint2* const list = reinterpret_cast<uint2*>(ldsRaw + Lay::discBytes + 16);  // (thread, step value bits); TR * TC entries fit the records' space
    static_assert(2 * Lay::recBytes >= 16 + 8 * static_cast<size_t>(TR) * TC, "the walk list lives where the prefix records were");
    int8_t* const rowW = reinterpret_cast<int8_t*>(list + TR * TC);  // the shape's half-widths (the walk's row loop reads them by index)
    static_assert(2 * Lay::recBytes >= 16 + 8 * static_cast<size_t>(TR) * TC + 64 + 32 * static_cast<size_t>(TR) * TC, "... the shape's half-widths and four doubles per walking lane behind it");
    if (threadIdx.x == 0) *count = 0;
    if (sp.edgeRows == 0ull) disc_tables<TR, TC, H>(d, g, ti0, tj0, r);  // (moments_phase left them out; the barrier below publishes them)
    {   // lane o reads the half-width of row o: ONE byte load per lane through a pointer in vector registers (indexing the argument
        // struct by the lane number made the compiler copy twenty bytes of it to scratch for halos above nine cells)
        const int8_t* rw = reinterpret_cast<const int8_t*>(ka) + offsetof(FusedKernArgs, sN) + offsetof(StepShape, rowW);
        asm volatile("" : "+v"(rw));
        if (threadIdx.x < 2 * H + 1) rowW[threadIdx.x] = rw[threadIdx.x];
    }
    __syncthreads();
    if (needWalk) list[atomicAdd(count, 1)] = make_uint2(threadIdx.x, __float_as_uint(stepOut));
    __syncthreads();
    const int n = *count;
    if (static_cast<int>(threadIdx.x) >= n) return;
    const int waveBase = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) & ~63);
    float ox, oy, oz, os, orough;
    {
        const int tid = static_cast<int>(list[threadIdx.x].x);
        double* const stash = reinterpret_cast<double*>(rowW + 64) + 4 * threadIdx.x;
        normals_cell_exact_rows<H>(d, rowW, sp.edgeRows, tid / TC, tid % TC, ti0, tj0, r, slopeCritical, roughCritical, stash, ox, oy, oz, os, orough);
    }
    // (the entry is read again rather than kept, by an index rebuilt from the wavefront's scalar base and the lane number: nothing
    // of the cell's identity — not even the thread id — stays in a vector register across the walks and the sweeps)
    int lane = static_cast<int>(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)));
    asm volatile("" : "+v"(lane));
    const uint2 e = list[waveBase + lane];
    const int li = static_cast<int>(e.x) / TC, lj = static_cast<int>(e.x) % TC;
    const float step = __uint_as_float(e.y);
    const size_t cell = static_cast<size_t>(ti0 + li) * g.cols + (tj0 + lj);
    if (!travOnly) {
        L.nx[cell] = ox;
        L.ny[cell] = oy;
        L.nz[cell] = oz;
        L.slope[cell] = os;
        L.rough[cell] = orough;
    }
    if (kStep) {
        const float third = 1.0f / 3.0f;
        L.trav[cell] = third * ((os + step) + orough);
    }
}
