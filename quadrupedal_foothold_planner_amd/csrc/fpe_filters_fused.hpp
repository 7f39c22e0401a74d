// fpe_filters_fused.hpp — round 5 of the producer's filters (included by fpe_filters.hpp inside its
// anonymous namespace): NormalVectorsFilter + SlopeFilter + RoughnessFilter by row moments with the halo as a TEMPLATE
// parameter, the StepFilter's second window in the same launch, and a chain that stores what the caller asked for.
//
// What changed against filter_normals_moments_kernel (round 4) and why — all three kernels of the chain are bound by the
// SIMDs' instruction issue (profiles/round4_filters.txt: VALU-active = 4 clocks x VALU instructions per wavefront), so the
// lever is instructions per cell:
//   * H (halo) and T (tile edge) are compile-time: the loop over the disc's rows is unrolled, a row's LDS addresses are the
//     thread's base plus an IMMEDIATE, the row offset is a constant operand of the integer multiply-adds;
//   * a row's members need the iterator's own test only where an offset lies ON the circle (step_shape's edge list: no row
//     at 2 cm / r 0.05, seven of eleven rows at 1 cm) — the other rows take their robust half-width from an SGPR;
//   * the six prefix arrays are two arrays of 16-byte records ({n | sum c << 16, sum c^2, sum z'} and {sum z'^2, sum c z'}):
//     four ds_read_b128 per row instead of twelve scalar reads, two integer subtractions instead of three;
//   * the prefix scans run one wavefront per record field group, lane = tile row (no divergence inside a wavefront);
//   * the workgroup -> tile map is XCD-aware (a contiguous band of tiles per XCD: a tile's halo is its neighbours' interior,
//     read through the same L2 instead of over the fabric — round 4 measured 6.2 x the layer in fabric reads);
//   * kMode 2 ("traversability only", the caller passed no layer buffer): the normals, slope, roughness and step values
//     stay in registers, only step_height (needed across the two windows) and traversability are stored — 20 bytes per
//     cell move instead of 52.
// The f64 arithmetic on the moments is the round-4 kernel's, operation for operation (same prefix sums, same differences,
// same recentring), so the layers are bit-identical to that kernel's (which tests/test_gpu_filters.py and the 30 000-map
// campaign of round 4 pinned against oracle/fpo_filters.cpp); cells the moment form cannot decide still take the literal
// walks (normals_cell_exact).
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
__device__ __forceinline__ double acos_unit(double x) {
    const double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01, pS2 = 2.01212532134862925881e-01, pS3 = -4.00555345006794114027e-02,
                 pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05, qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00,
                 qS3 = -6.88283971605453293030e-01, qS4 = 7.70381505559019352791e-02;
    const bool hiHalf = x >= 0.5;
    const double z = hiHalf ? (1.0 - x) * 0.5 : x * x;
    const double p = z * __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, pS5, pS4), pS3), pS2), pS1), pS0);
    const double q = __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, qS4, qS3), qS2), qS1), 1.0);
    const double rr = p * rcp_refined(q);
    const double s = sqrt_refined(z);
    const double up = 2.0 * __builtin_fma(s, rr, s);                                                // x >= 0.5: 2 asin(sqrt((1 - x) / 2))
    const double low = 1.57079632679489655800e+00 - (x - (6.12323399573676603587e-17 - x * rr));   // x < 0.5: pi/2 - asin(x)
    return hiHalf ? up : low;
}

// Scatter moments of a cell's disc -> float normal, slope and roughness values: the closing arithmetic of
// filter_normals_moments_kernel (same thresholds for handing a cell to the literal walks) with the divisions as products by
// refined reciprocals and acos / sqrt as above — every value within an f64 ulp or two of the round-4 kernel's, far inside the
// float layers' rounding (tests/test_gpu_filters.py: one float ulp, >= 99.99 % of the cells bit-identical to the oracle).
__device__ __forceinline__ void normals_from_moments(const DiscLds& d, const MapGeom& g, int li, int lj, int ti0, int tj0, double r, double slopeCritical,
                                                     double roughCritical, double invSlopeCritical, double invRoughCritical, int N, int Sc, int Scc, int Sv,
                                                     int Svv, int Svc, double Sz, double Szz, double Scz, double Svz, float& ox, float& oy, float& oz, float& os,
                                                     float& orough) {
    const double nd = static_cast<double>(N);
    const double invN = rcp_refined(nd);
    const double Avv = static_cast<double>(N * Svv - Sv * Sv) * invN, Acc = static_cast<double>(N * Scc - Sc * Sc) * invN;
    const double Avc = static_cast<double>(N * Svc - Sv * Sc) * invN;
    const double Avz = Svz - static_cast<double>(Sv) * Sz * invN, Acz = Scz - static_cast<double>(Sc) * Sz * invN;
    const double Azz = fmax(Szz - Sz * Sz * invN, 0.0);
    const double res = g.res, res2 = res * res;
    const double a00 = res2 * Avv, a01 = res2 * Avc, a02 = -(res * Avz), a11 = res2 * Acc, a12 = -(res * Acz), a22 = Azz;
    double ex, ey, ez, eigS, eigL;
#ifdef FPE_DBG_NO_EIG
    ex = a02; ey = a12; ez = a00 + a11 + a22 + a01; eigS = 1.0; eigL = 1.0;
#else
    if (!normal_newton(a00, a01, a02, a11, a12, a22, ex, ey, ez, eigS, eigL)) normal_from_scatter(a00, a01, a02, a11, a12, a22, ex, ey, ez, eigS, eigL);
#endif
    const double tinyC = 1e-6;
#ifdef FPE_DBG_NO_EXACT
    if (false) {
#else
    if (!(eigS > 1e-10 * eigL) || fabs(ex) < tinyC || fabs(ey) < tinyC || fabs(ez) < tinyC) {
#endif
        normals_cell_exact(d, li, lj, ti0, tj0, r, slopeCritical, 1, roughCritical, ox, oy, oz, os, orough);
    } else {
        ox = static_cast<float>(ex);
        oy = static_cast<float>(ey);
        oz = static_cast<float>(ez);
        const double slope = acos_unit(static_cast<double>(oz));  // SlopeFilter reads the float layer
        os = slope < slopeCritical ? static_cast<float>(1.0 - slope * invSlopeCritical) : 0.0f;
        // RoughnessFilter: the plane through the mean with the FLOAT normal: sum of squared distances = n^T A n
        const double nx = ox, ny = oy, nz = oz;
        const double q = nx * (nx * a00 + 2.0 * (ny * a01 + nz * a02)) + ny * (ny * a11 + 2.0 * (nz * a12)) + nz * (nz * a22);
        const double roughness = sqrt_refined(fmax(q, 0.0) * rcp_refined(nd - 1.0));  // (N >= 3 here: fewer members are rank-deficient)
        orough = roughness < roughCritical ? static_cast<float>(1.0 - roughness * invRoughCritical) : 0.0f;
    }
}

// The moment phase of one tile: tables and source tile (disc_setup, one barrier), the prefix records (one barrier), then the
// calling thread's cell.  `live`: the thread's cell is inside the map (every thread takes part in the barriers).
template <int H, int TR, int TC>
__device__ __forceinline__ void moments_phase(char* ldsRaw, const MapGeom& g, const float* __restrict__ elev, int ti0, int tj0, double r,
                                              const StepShape& sp, double slopeCritical, double roughCritical, double invSlopeCritical,
                                              double invRoughCritical, bool live, float& ox, float& oy, float& oz, float& os, float& orough) {
    using Lay = FusedLayout<H, TR, TC>;
    constexpr int WR = Lay::WR, W = Lay::WC, W1 = Lay::W1, D = Lay::D;
    static_assert(W <= 64 && WR <= 64, "a tile row is one wavefront load; the scans run one lane per tile row");
    static_assert(TR * TC >= 256, "four wavefronts scan the four record field groups");
    const DiscLds d = disc_carve(ldsRaw, H, TR, TC);
    MomentA* const PA = reinterpret_cast<MomentA*>(ldsRaw + Lay::discBytes);
    MomentB* const PB = reinterpret_cast<MomentB*>(ldsRaw + Lay::discBytes + Lay::recBytes);
    disc_setup<true, TR, TC, H>(d, g, elev, ti0, tj0, r);
    const float zf = d.tile[H * W + H];
    const double z0 = zf == zf ? static_cast<double>(zf) : 0.0;
    {   // prefix records over the tile's columns: wavefront = field group, lane = tile row
        const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
#ifdef FPE_DBG_NO_SCAN
        if (false) {
#else
        if (lane < WR && grp < 4) {
#endif
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
                double* dst = grp == 1 ? &PA[lane * W1].z : (grp == 2 ? &PB[lane * W1].zz : &PB[lane * W1].cz);
                double acc = 0.0;
                dst[0] = 0.0;
#pragma unroll 4
                for (int c = 0; c < W; ++c) {
                    const float z = src[c];
                    const double zz = z == z ? static_cast<double>(z) - z0 : 0.0;
                    const double term = grp == 1 ? zz : (grp == 2 ? zz * zz : static_cast<double>(c) * zz);
                    acc += term;
                    dst[2 * (c + 1)] = acc;
                }
            }
        }
    }
    __syncthreads();
    const int li = threadIdx.x / TC, lj = threadIdx.x % TC;
    const int i = ti0 + li, j = tj0 + lj;
    const float nanf = __builtin_nanf("");
    ox = oy = oz = os = orough = nanf;
    if (!live || !isfinite(d.tile[(li + H) * W + lj + H])) return;
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
#ifdef FPE_DBG_NO_ROWS
    Sz = d.tile[threadIdx.x]; Szz = Sz * Sz + 1.0; Scz = 0.3 * Sz; Svz = 0.1; AnC = 21u | (21u * cc << 16); ACC = 21u * cc * cc + 50; Svv = 50;
#else
#pragma unroll
#endif
    for (int oo = 0; oo < (
#ifdef FPE_DBG_NO_ROWS
        0
#else
        D
#endif
        ); ++oo) {
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
    const int N = static_cast<int>(AnC & 0xFFFFu), SC = static_cast<int>(AnC >> 16);
    const int Sc = SC - cc * N;
    const int Scc = static_cast<int>(ACC) - 2 * cc * SC + cc * cc * N;
    const int Svc = SvC - cc * Sv;
#ifdef FPE_DBG_NO_FINISH
    ox = static_cast<float>(Sz + Szz); oy = static_cast<float>(Scz + Svz); oz = static_cast<float>(N + Sc + Scc); os = static_cast<float>(Sv + Svv + Svc); orough = ox + oy;
#else
    normals_from_moments(d, g, li, lj, ti0, tj0, r, slopeCritical, roughCritical, invSlopeCritical, invRoughCritical, N, Sc, Scc, Sv, Svv, Svc, Sz, Szz, Scz, Svz, ox, oy, oz, os, orough);
#endif
}
