// fpe_bits.hpp — part two of the kernel translation unit (included at the end of fpe_kernels.hip, inside
// namespace fpe): the BIT-WINDOW kernels.
//
// Per map snapshot and threshold pair the engine keeps four bit planes (BitMap, fpe_device.hpp): D (trav <
// defaultFootholdThreshold_, raw compare), Df (the same for finite cells only), C (finite && trav <
// candidateFootholdThreshold_), F (finite).  A leg's search window — every cell its default disc, centroid
// rectangle (cpp:1615-1750) and spiral candidates' foot discs (cpp:2085-2163) can touch — is then one 32-bit row
// mask per plane and window row: lane s of the leg's group loads rows s, s + G, ... (two 16-byte loads per row),
// and everything the reference decides by comparing traversability values becomes bit arithmetic:
//   * centroid row scan (cpp:1717-1750): popcount of the D row under the rectangle's column mask;
//   * checkDefaultFoothold (cpp:2039-2082): Df bit of every cell of the disc (membership stays the f64 test);
//   * checkCirclePolygonFoothold (cpp:2117-2163) for EVERY candidate at once: per window row
//         P = ~F | (~C & inside)      ("cell does not fail": non-finite, or above threshold and inside the polygon)
//     eroded with the host-proved foot-disc offset table, E = AND_k shift(P[row + da_k], db_k); a candidate is
//     valid iff its E bit is set, so a spiral round is an LDS read and a bit test — no map access at all.
//     The reference rectangle's PNPOLY test is exact in INDEX space: cell centres are monotone in the index, so
//     {i : xlo <= x_i < xhi} is an index interval whose ends are found by evaluating the reference's own f64
//     comparison at the two indices next to a predicted boundary.  Other polygons keep the per-cell PNPOLY test,
//     applied only to candidates that pass the threshold erosion.
// The f32 elevation layer is read for the mean heights only (cpp:2520-2554, unchanged ordered sums).
// Exactness rests on three host-side proofs (bits_supported): the foot-disc offset table (derive_foot_offsets), the
// window half-width (every cell that can be touched lies inside it), and getIndex(submap cell centre) == top-left +
// (row, col) for the centroid result.  When a proof fails the engine launches the direct kernels instead.
#pragma once

namespace {

// ---- bit-plane build: one wavefront ballots 64 columns of a row ---------------------------------------------
__global__ __launch_bounds__(256) void build_bitmap_kernel(const float* __restrict__ trav, int rows, int cols, float thrD,
                                                           float thrC, uint4* __restrict__ words, int strideW, int nw) {
    const int i = blockIdx.y;
    const int j = blockIdx.x * 256 + static_cast<int>(threadIdx.x);
    const bool in = j < cols;
    float v = 0.0f;
    if (in) v = trav[static_cast<size_t>(i) * cols + j];
    const bool fin = in && __builtin_isfinite(v);
    const bool d = in && v < thrD;  // raw compare: NaN -> false, -inf -> true (cpp:1653, 1736)
    const bool c = fin && v < thrC;
    const unsigned long long bD = __ballot(d), bDf = __ballot(d && fin), bC = __ballot(c), bF = __ballot(fin);
    const int lane = static_cast<int>(threadIdx.x) & 63;
    if (lane < 2) {
        const int w = (blockIdx.x * 256 + (static_cast<int>(threadIdx.x) & ~63)) / 32 + lane;
        if (w < nw) {
            uint4 o;
            o.x = static_cast<unsigned>(bD >> (32 * lane));
            o.y = static_cast<unsigned>(bDf >> (32 * lane));
            o.z = static_cast<unsigned>(bC >> (32 * lane));
            o.w = static_cast<unsigned>(bF >> (32 * lane));
            words[bit_group_index(i, w, strideW)] = o;
        }
    }
}

// (bit_group_index: fpe_device.hpp — the tiled plane layout, shared with win_issue)

// Synchronisation of the lanes of a pose in the bit-window kernels.  A pose never spans more than ONE wavefront here
// (8 lanes per leg: half a wavefront; one wavefront per pose), and the LDS operations of a wavefront execute in order:
// the compiler must not reorder across the point, nothing has to be waited for.  (pose_sync<64> of the direct kernels is
// a workgroup barrier — a pose owns four wavefronts there — which also waits for every outstanding global load and
// store of the wavefront: in the one-wavefront-per-pose kernels that serialised the leg's loads with its LDS hand-offs.)
template <int G>
__device__ __forceinline__ void bits_sync() {
#ifdef FPE_BITS_BARRIER_SYNC
    pose_sync<G>();
#else
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#endif
}

// 64-lane kernels: upper bound of a CircleIterator bounding box (cells): two rounds of 64 membership tests
constexpr int kBitsMaxBoxCells = 128;

// ---- window rows ------------------------------------------------------------------------------------------
// A window row is KW 32-bit words per plane (KW = 1: windows of up to 32 columns, the 8-lane kernels; 2 / 3: the
// one-wavefront-per-pose kernels of 1 cm / 0.5 cm maps).
template <int NRL, int KW>
struct WinRows {
    uint32_t D[NRL][KW], Df[NRL][KW], C[NRL][KW], F[NRL][KW];
};
// Per-leg LDS: row masks shared between the lanes of the leg's group — a: Df rows, later P rows, then E rows (row-
// interval erosion); f: F rows for polygons that are not folded into P; h[k]: single-word rows: horizontally eroded P rows,
// one array per distinct row half-width of the disc; multi-word rows: ONE array, the E rows of the nested erosion (h[0]
// is also the E rows of the offset-by-offset erosion, and f / h[0] the scratch of the polygon's row masks).  Arrays hold `rows` window rows of KW words (the 64-lane kernels allocate 2 winH + 1 rows, not 64 * NRL:
// LDS, not registers, bounds their occupancy).  The row arrays double as float scratch of a direct disc pass.
struct LegBits {
    uint32_t* a;
    uint32_t* f;
    uint32_t* h0;  // array k of the eroded rows at h0 + k * hStride (a pointer array indexed at run time would live in scratch)
    int hStride;
    float* hs;  // 96-bit-row kernels only: the visited elevations of a leg's three discs (in-chain heights, heights3_finish)
    int rows;
};
// words (4 bytes) of one leg's LDS: the row arrays, then hs for the 96-bit-row kernels
// widest row (in 32-bit words) of the one-wavefront-per-pose kernels whose mean heights leave the chain (flush_seqrec2)
// (2: the 96-bit-row kernels keep their in-chain sums — TermSum / heights3_finish — as they did until the flush read its
// boxes without divisions and in two load groups per batch; measured then: cfg-5 0.402 ms in-chain, 0.379 ms deferred)
#ifndef FPE_SEQ_DEFER_KW
#define FPE_SEQ_DEFER_KW 3
#endif
constexpr int kSeqDeferMaxKW = FPE_SEQ_DEFER_KW;
__host__ __device__ __forceinline__ int legbits_words(int rows, int kw, int nHW, bool wide) {
    // (multi-word rows use three of the arrays only since the nested erosion; the others stay: shrinking the allocation
    // to three arrays was measured 1.5 % SLOWER on cfg-3, neutral on cfg-5 — kept as measured)
    const int arrays = 2 + (nHW > 0 ? nHW : 1);
    return ((arrays * rows * kw + 3) & ~3) + ((wide && kw > kSeqDeferMaxKW) ? 3 * kBitsMaxBoxCells : 0);
}
__device__ __forceinline__ LegBits make_legbits(unsigned char* base, int rows, int kw, int nHW, bool wide) {
    LegBits lb;
    uint32_t* p = reinterpret_cast<uint32_t*>(base);
    const int n = rows * kw;
    lb.rows = rows;
    lb.a = p;
    lb.f = p + n;
    lb.h0 = p + 2 * n;
    lb.hStride = n;
    const int arrays = 2 + (nHW > 0 ? nHW : 1);
    lb.hs = reinterpret_cast<float*>(p + ((arrays * n + 3) & ~3));
    (void)wide;
    return lb;
}

// The y side of a leg's geometry for one gait cycle (8-lane kernels; see fill_yentry below).
struct YEntry {
    int jc;        // getIndex(centre), column
    int j0d, njd;  // foot-disc box columns (centre disc and default-track disc: same y, same radius)
    int j0r, njr;  // centroid rectangle columns (getSubmap, cpp:1615-1627)
    int jA, jB;    // reference rectangle polygon: the columns j with ylo <= y_j < yhi (rectangle_index_bounds)
    int flags;     // bit 0: y part of getSubmap's success; bit 1: |y| usable (centre_usable)
    double ny;
    double sbaseY;   // submap position.y + (0.5 * sublength.y - 0.5 * res)
    double yA, yB;   // cell_pos(sbaseY, res, (rightCol + 1) >> 1), cell_pos(sbaseY, res, rightCol >> 1)  (cpp:1816)
    double dy2[3];   // (cell_pos(baseY, res, j0d + k) - ny)^2, k = 0..2 (3x3 disc form)
    // window columns (window origin jc - winH, one word) of the centroid rectangle [j0r, j0r + njr) and of the
    // reference rectangle polygon [jA, jB]
    uint32_t rmask, pmask;
};
static_assert(sizeof(YEntry) == 96, "YEntry layout");

// ds_swizzle of a double (bit mode), two dwords
template <int kPattern>
__device__ __forceinline__ double swizzle_f64(double v) {
    const long long bits = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_ds_swizzle(static_cast<int>(bits), kPattern);
    const int hi = __builtin_amdgcn_ds_swizzle(static_cast<int>(bits >> 32), kPattern);
    return __builtin_bit_cast(double, (static_cast<long long>(hi) << 32) | static_cast<unsigned int>(lo));
}

// Value of lane L (compile-time) of every 8-lane group: a ds_swizzle in bit mode.  (-DFPE_BCAST8_DPP: two DPP moves instead —
// a quad broadcast, every quad gets ITS lane L & 3, then the half-row mirror hands the owner quad's value to the group's
// other quad; VALU latency instead of an LDS round trip, two VALU instructions instead of one LDS instruction.  Measured on
// the headline, round 4: 27.5 us either way — the exchange is not on the critical path; the swizzle stays.)
template <int L>
__device__ __forceinline__ int bcast8_dpp(int x) {
#ifndef FPE_BCAST8_DPP
    return __builtin_amdgcn_ds_swizzle(x, 0x18 | (L << 5));
#else
    const int q = __builtin_amdgcn_update_dpp(0, x, (L & 3) * 0x55, 0xF, 0xF, true);
    return __builtin_amdgcn_update_dpp(q, q, 0x141, 0xF, (L < 4) ? 0xA : 0x5, false);
#endif
}
template <int L>
__device__ __forceinline__ double bcast8_dpp_f64(double v) {
    const long long bits = __builtin_bit_cast(long long, v);
    const int lo = bcast8_dpp<L>(static_cast<int>(bits)), hi = bcast8_dpp<L>(static_cast<int>(bits >> 32));
    return __builtin_bit_cast(double, (static_cast<long long>(hi) << 32) | static_cast<unsigned int>(lo));
}

// Multi-word row shifts by 0 <= s < 32 columns: shr: bit j of the result = bit j + s of the row; shl: bit j - s.
template <int KW>
__device__ __forceinline__ void row_shr(const unsigned (&x)[KW], unsigned s, unsigned (&o)[KW]) {
#pragma unroll
    for (int q = 0; q < KW; ++q) o[q] = __builtin_amdgcn_alignbit(q + 1 < KW ? x[q + 1] : 0u, x[q], s);
}
template <int KW>
__device__ __forceinline__ void row_shl(const unsigned (&x)[KW], unsigned s, unsigned (&o)[KW]) {
#pragma unroll
    for (int q = 0; q < KW; ++q) o[q] = s ? __builtin_amdgcn_alignbit(x[q], q > 0 ? x[q - 1] : 0u, 32u - s) : x[q];
}

__device__ __forceinline__ unsigned bits_from(int lo) { return lo >= 32 ? 0u : (lo <= 0 ? ~0u : (~0u << lo)); }
__device__ __forceinline__ unsigned bits_to(int hi) { return hi < 0 ? 0u : (hi >= 31 ? ~0u : ((2u << hi) - 1u)); }
// word `wi` of the mask with bits [lo, hi] set (bit positions over the whole multi-word row)
__device__ __forceinline__ unsigned range_word(int lo, int hi, int wi) { return bits_from(lo - 32 * wi) & bits_to(hi - 32 * wi); }

// Layer cell / plane word group at a 32-bit offset from the (uniform) base pointer: one 32-bit multiply-add instead
// of 64-bit address arithmetic per load.  bits_supported() bounds layers and planes below 2 GiB and 2^24 rows / columns.
__device__ __forceinline__ float load_cell(const float* base, unsigned cell) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + static_cast<size_t>(cell << 2));
}
__device__ __forceinline__ uint4 load_group(const uint4* base, unsigned group) {
    return *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(base) + static_cast<size_t>(group << 4));
}

// Window origin (iw0, jw0) = getIndex(centre) - winH.  Lane `sub` holds window rows sub + G * k.  Rows and word
// groups outside the map are clamped onto the zero padding of the planes.
template <int G, int NRL, int KW>
__device__ __forceinline__ void win_issue(const BitMap& bm, const MapGeom& mg, const Grp<G>& g, int iw0, int jw0,
                                          uint4 (&grp)[NRL][KW + 1]) {
    static_assert(KW + 1 <= kBitPadW, "the planes' zero padding must cover a whole window row");
    int w0 = jw0 >> 5;
    w0 = max(-kBitPadW, min(w0, bm.nw + kBitPadW - (KW + 1)));
#pragma unroll
    for (int k = 0; k < NRL; ++k) {
        int i = iw0 + g.sub + G * k;
        i = max(-1, min(i, mg.rows));
#if FPE_BITS_TILED
        // tile (row group, word) = one 128-byte line holding 8 consecutive rows: the lanes that own rows of one row
        // group read different 16-byte pieces of the SAME line (8-lane kernels: one or two lines per load instruction
        // instead of eight; one-wavefront-per-pose kernels: eight or nine instead of 64)
        const unsigned r1 = static_cast<unsigned>(i + 1);
        const unsigned first = ((__umul24(r1 >> 3, static_cast<unsigned>(bm.strideW)) + static_cast<unsigned>(w0 + kBitPadW)) << 3) + (r1 & 7u);
#pragma unroll
        for (int q = 0; q <= KW; ++q) grp[k][q] = load_group(bm.words, first + 8u * static_cast<unsigned>(q));
#else
        const unsigned first = __umul24(static_cast<unsigned>(i + 1), static_cast<unsigned>(bm.strideW)) + static_cast<unsigned>(w0 + kBitPadW);
#pragma unroll
        for (int q = 0; q <= KW; ++q) grp[k][q] = load_group(bm.words, first + static_cast<unsigned>(q));
#endif
    }
}
template <int NRL, int KW>
__device__ __forceinline__ void win_finish(int jw0, const uint4 (&grp)[NRL][KW + 1], WinRows<NRL, KW>& w) {
    const unsigned sh = static_cast<unsigned>(jw0) & 31u;
#pragma unroll
    for (int k = 0; k < NRL; ++k)
#pragma unroll
        for (int q = 0; q < KW; ++q) {
            w.D[k][q] = __builtin_amdgcn_alignbit(grp[k][q + 1].x, grp[k][q].x, sh);
            w.Df[k][q] = __builtin_amdgcn_alignbit(grp[k][q + 1].y, grp[k][q].y, sh);
            w.C[k][q] = __builtin_amdgcn_alignbit(grp[k][q + 1].z, grp[k][q].z, sh);
            w.F[k][q] = __builtin_amdgcn_alignbit(grp[k][q + 1].w, grp[k][q].w, sh);
        }
}
// Bit (window row ri, window column cj) of a row array in LDS; 0 outside the window.
template <int KW>
__device__ __forceinline__ unsigned win_bit(const uint32_t* rows, int nRows, int ri, int cj) {
    const bool in = static_cast<unsigned>(ri) < static_cast<unsigned>(nRows) && static_cast<unsigned>(cj) < 32u * KW;
    const int r = min(max(ri, 0), nRows - 1), c = min(max(cj, 0), 32 * KW - 1);
    const uint32_t wd = rows[r * KW + (c >> 5)];
    return in ? (wd >> (c & 31)) & 1u : 0u;
}

// checkFootholdUseCentroidMethod's row scan (cpp:1649-1658 whole-region test, cpp:1717-1750 blocked rows) from the
// D rows: lane = window row.  `cnt > (rightCol + 1) * 0.5` (cpp:1743) is 2 * cnt > nj in integers.
template <int G, int NRL, int KW>
__device__ __forceinline__ CentroidScan rows_from_bits(const Submap& s, const WinRows<NRL, KW>& w, const Grp<G>& g, int iw0, int jw0,
                                                       const uint32_t* colMask = nullptr) {  // KW == 1: YEntry::rmask
    static_assert(G * NRL <= 128, "blocked-row masks are kept in two 64-bit words");
    CentroidScan r0;
    const int ni = s.ni, nj = s.nj;
    const int c0 = s.j0 - jw0, c1 = c0 + nj - 1;  // window columns of the rectangle
    unsigned long long blk[2] = {0ull, 0ull};      // bit = window row
    unsigned blk32 = 0u;                           // (windows of up to 32 rows: one 32-bit word)
    bool anyBelow = false;
#pragma unroll
    for (int k = 0; k < NRL; ++k) {
        const int ri = g.sub + G * k;
        const int r = iw0 + ri - s.i0;  // row of the rectangle held by this lane in slot k
        const bool liveRow = s.ok && r >= 0 && r < ni;
        int cnt = 0;
#pragma unroll
        for (int q = 0; q < KW; ++q) cnt += __builtin_popcount(w.D[k][q] & (colMask ? *colMask : range_word(c0, c1, q)));
        anyBelow |= liveRow && cnt > 0;
        const bool blocked = liveRow && 2 * cnt > nj;
        const unsigned long long mk = g.ballot(blocked);
        constexpr int kPerWord = 64 / G;  // ballots of G lanes packed into a 64-bit word
        if constexpr (G * NRL <= 32) blk32 |= static_cast<unsigned>(mk) << (G * k);
        else if constexpr (G == 64) blk[k & 1] |= mk;
        else blk[(k / kPerWord) & 1] |= mk << (G * (k % kPerWord));
    }
    const int off = s.i0 - iw0;
    if constexpr (G * NRL <= 32) {  // the 8-lane shapes: 32-bit shifts and bit scans instead of 64-bit ones
        const unsigned rel = blk32 >> (off & 31);
        const unsigned relIn = (static_cast<unsigned>(off) < 32u) ? rel : 0u;
        r0.minRow = relIn ? __builtin_ctz(relIn) : 0;
        r0.maxRow = relIn ? 31 - __builtin_clz(relIn) : 0;
        r0.whole = s.ok && ni * nj > 0 && !g.any(anyBelow);
        return r0;
    }
    if constexpr (G * NRL <= 64) {  // the whole window in one word
        const unsigned long long rel = blk[0] >> (off & 63);
        const unsigned long long relIn = (static_cast<unsigned>(off) < 64u) ? rel : 0ull;
        r0.minRow = relIn ? __builtin_ctzll(relIn) : 0;
        r0.maxRow = relIn ? 63 - __builtin_clzll(relIn) : 0;
        r0.whole = s.ok && ni * nj > 0 && !g.any(anyBelow);
        return r0;
    }
    // rows relative to the rectangle's first row: a 128-bit shift (the rectangle lies inside the window, and a window
    // of more than 64 rows can hold a rectangle of more than 64)
    unsigned long long relLo = 0ull, relHi = 0ull;
    if (off >= 0 && off < 64) {
        relLo = (blk[0] >> off) | (off ? (blk[1] << (64 - off)) : 0ull);
        relHi = blk[1] >> off;
    } else if (off >= 64 && off < 128) {
        relLo = blk[1] >> (off - 64);
    }
    r0.minRow = relLo ? __builtin_ctzll(relLo) : (relHi ? 64 + __builtin_ctzll(relHi) : 0);
    r0.maxRow = relHi ? 127 - __builtin_clzll(relHi) : (relLo ? 63 - __builtin_clzll(relLo) : 0);
    r0.whole = s.ok && ni * nj > 0 && !g.any(anyBelow);
    return r0;
}

// checkDefaultFoothold (cpp:2039-2082) from the Df rows: valid iff >= 1 cell visited and no visited cell has its Df
// bit set.  The visited cells are the ones disc_issue() enumerated (d.vis / the 3x3 form); boxes it did not
// pipeline (clamped at the map border, or larger than the pipeline) are walked here, membership test included.
template <int G, int KW, bool kMid>
__device__ __forceinline__ bool default_ok_bits(const DevMap& m, const PlanConsts& pc, double cx, double cy, const BBox& bb,
                                                const DiscLoads& d, const uint32_t* rowsDf, int nRows, int iw0, int jw0, const Grp<G>& g) {
    bool any = false, fail = false;
    if (d.pipelined) {
        if (G == 8 && d.mid) {  // wave-uniform: cells 0-3 and 5-8 on the lanes, the middle cell always visited
            const int t = g.sub + (g.sub >= 4 ? 1 : 0);
            const int a = t >= 6 ? 2 : (t >= 3 ? 1 : 0);
            const int ri = bb.i0 - iw0, cj = bb.j0 - jw0;
            fail = (d.vis[0] != 0 && win_bit<KW>(rowsDf, nRows, ri + a, cj + (t - 3 * a)) != 0u) ||
                   win_bit<KW>(rowsDf, nRows, ri + 1, cj + 1) != 0u;
            return !g.any(fail);
        }
        if constexpr (!kMid) {
            const float njInv = rcp_small(bb.nj);
#pragma unroll
            for (int r = 0; r < disc_rounds<G>(); ++r) {
                // wave-uniform: a round past every box of the wavefront (64-bit rows: boxes of <= 64 cells; not worth a test on the 96-bit ones)
                if (KW <= 2 && r > 0 && __ballot(d.vis[r] != 0) == 0ull) continue;
                int a, bq;
                divmod_small(min(r * G + g.sub, 4095), max(bb.nj, 1), njInv, a, bq);
                const bool v = d.vis[r] != 0;
                any |= v;
                fail |= v && win_bit<KW>(rowsDf, nRows, bb.i0 + a - iw0, bb.j0 + bq - jw0) != 0u;
            }
            return g.any(any) && !g.any(fail);
        }
    }
    const int nb = bb.ni * bb.nj;
    const float njInv = rcp_small(bb.nj);
    for (int base = 0; base < nb; base += G) {
        const int t = base + g.sub;
        if (t < nb) {
            int a, bq;
            divmod_small(t, bb.nj, njInv, a, bq);
            const int i = bb.i0 + a, j = bb.j0 + bq;
            if (in_range(i, j, m.g.rows, m.g.cols) && cell_in_disc(m.g, i, j, cx, cy, pc.rf2)) {
                any = true;
                fail |= win_bit<KW>(rowsDf, nRows, i - iw0, j - jw0) != 0u;
            }
        }
    }
    return g.any(any) && !g.any(fail);
}

// Centroid case logic (cpp:1684-1952) given the row scan; the result's foot disc is cell-centred, i.e. the
// host-proved offset table in CircleIterator order, and getIndex(result) is top-left + (newRow, newCol).
// kOneCell: the 3x3-only variants run with a one-cell foot disc (rf < res): the result's height is that cell's.
struct CentroidPendingBits {
    CentroidOut o;
    int needDisc;  // 0/1
    float e0;      // kOneCell: elevation of the result's own cell
    float e[kDiscRounds];
    int vis[kDiscRounds];
};
// yA / yB (optional): the two possible result ordinates cell_pos(s.baseY, res, (rightCol + 1) >> 1) and
// cell_pos(s.baseY, res, rightCol >> 1), precomputed with the y side of the leg's geometry (YEntry).
template <int G, bool kOneCell, bool kHaveY = false, bool kLoad = true>
__device__ __forceinline__ void centroid_begin_bits_impl(const DevMap& m, const PlanConsts& pc, const LegCtx& c, const Submap& s,
                                                         const CentroidScan& sc, float zCentre, const Grp<G>& g, CentroidPendingBits& cp,
                                                         double yA, double yB) {
    CentroidOut& o = cp.o;
    cp.needDisc = 0;
    cp.e0 = 0.0f;
    o.x = 0.0;
    o.y = 0.0;
    o.z = 0.0f;
    o.row = -1;
    o.col = -1;
    o.code = 5;
#pragma unroll
    for (int r = 0; r < kDiscRounds; ++r) {
        cp.vis[r] = 0;
        cp.e[r] = 0.0f;
    }
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
    // floor((a) * 0.5) / ceil((a) * 0.5) of small non-negative integers, as integer arithmetic (exact)
    int newRow, newCol;
    if (minRow == 0 && maxRow != bottomRow) {  // case 1, cpp:1777-1786
        newRow = (maxRow + bottomRow + 1) >> 1;
        newCol = (rightCol + 1) >> 1;
        o.code = 1;
    } else if (minRow != 0 && maxRow != bottomRow) {  // case 2, cpp:1843-1886
        if ((minRow - 0) >= (bottomRow - maxRow)) {
            newRow = (minRow + 1) >> 1;
            o.code = 2;
        } else {
            newRow = (maxRow + bottomRow) >> 1;
            o.code = 3;
        }
        newCol = rightCol >> 1;
    } else if (minRow != 0 && maxRow == bottomRow) {  // case 3, cpp:1944-1952
        newRow = (minRow + 1) >> 1;
        newCol = rightCol >> 1;
        o.code = 4;
    } else {
        return;  // first and last row blocked: no branch taken, result stays (0,0,0)
    }
    o.x = cell_pos(s.baseX, m.g.res, newRow);  // map.getPosition(newIndex) on the SUBMAP (cpp:1816)
    if constexpr (kHaveY) o.y = (o.code == 1) ? yA : yB;
    else o.y = cell_pos(s.baseY, m.g.res, newCol);
    o.row = s.i0 + newRow;
    o.col = s.j0 + newCol;
    if constexpr (!kLoad) {
        // (the caller defers the result's height: flush_seqrec walks the offset table itself)
    } else if constexpr (kOneCell) {
        cp.e0 = m.elev[static_cast<size_t>(o.row) * m.g.cols + o.col];  // a cell of the submap: inside the map
    } else {
        constexpr int kRounds = G >= 64 ? 1 : kDiscRounds;  // nFoot <= 64 fits one 64-lane round (bits_supported)
#pragma unroll
        for (int r = 0; r < kRounds; ++r) {
            const int k = r * G + g.sub;
            if (k < pc.nFoot) {
                const int qi = o.row + c.footDa[k], qj = o.col + c.footDb[k];
                if (in_range(qi, qj, m.g.rows, m.g.cols)) {
                    cp.vis[r] = 1;
                    cp.e[r] = m.elev[static_cast<size_t>(qi) * m.g.cols + qj];
                }
            }
        }
    }
    cp.needDisc = 1;
}
template <int G, bool kOneCell, bool kLoad = true>
__device__ __forceinline__ void centroid_begin_bits(const DevMap& m, const PlanConsts& pc, const LegCtx& c, const Submap& s,
                                                    const CentroidScan& sc, float zCentre, const Grp<G>& g, CentroidPendingBits& cp) {
    centroid_begin_bits_impl<G, kOneCell, false, kLoad>(m, pc, c, s, sc, zCentre, g, cp, 0.0, 0.0);
}
template <int G, bool kOneCell, bool kLoad = true>
__device__ __forceinline__ void centroid_begin_bits(const DevMap& m, const PlanConsts& pc, const LegCtx& c, const Submap& s,
                                                    const CentroidScan& sc, float zCentre, const Grp<G>& g, CentroidPendingBits& cp,
                                                    double yA, double yB) {
    centroid_begin_bits_impl<G, kOneCell, true, kLoad>(m, pc, c, s, sc, zCentre, g, cp, yA, yB);
}
// getFootholdMeanHeight (cpp:2520-2554) of the centroid result from the loads centroid_begin_bits issued.
template <int G, bool kOneCell>
__device__ __forceinline__ float centroid_height_bits(const PlanConsts& pc, const Grp<G>& g, const CentroidPendingBits& cp, float* scratch) {
    if constexpr (kOneCell) {
        const float v = __builtin_isfinite(cp.e0) ? cp.e0 : 0.0f;  // cpp:2532-2537
        const bool inc = v < 10;                                   // cpp:2539
        return finish_mean(inc ? 0.0f + v : 0.0f, v, inc ? 1 : 0, pc.h);
    } else if constexpr (G >= 64) {
        // one round of table entries in CircleIterator (row-major) order: compacted into LDS, summed sequentially
        float sum = 0.0f, last = 0.0f;
        int cnt = 0;
        OrderedSum os{scratch, 0};
        const float v = __builtin_isfinite(cp.e[0]) ? cp.e[0] : 0.0f;
        ordered_push(g, os, cp.vis[0] != 0, v);
        ordered_finish(os, sum, last, cnt);
        return finish_mean(sum, last, cnt, pc.h);
    } else {
        // lanes = table entries in CircleIterator order: G dependent adds per round on swizzled lane values
        float sum = 0.0f, last = 0.0f;
        int cnt = 0;
        float v[kDiscRounds];
        bool anyVis = false;
#pragma unroll
        for (int r = 0; r < kDiscRounds; ++r) {
            v[r] = __builtin_isfinite(cp.e[r]) ? cp.e[r] : 0.0f;
            const bool inc = cp.vis[r] != 0 && v[r] < 10;
            anyVis |= cp.vis[r] != 0;
            cnt += __builtin_popcountll(g.ballot(inc));
            if (r == 0 || __ballot(cp.vis[r] != 0) != 0ull) sum = SeqSum<G>::run(sum, inc ? v[r] : -0.0f);
        }
        if (__ballot(cnt == 0 && g.any(anyVis)) != 0ull) {  // every visited value >= 10: the LAST visited value (cpp:2547-2551)
#pragma unroll
            for (int r = 0; r < kDiscRounds; ++r) {
                const unsigned long long mr = g.ballot(cp.vis[r] != 0);
                const float lv = g.bcast(v[r], mr ? 63 - __builtin_clzll(mr) : 0);
                if (mr) last = lv;
            }
        }
        return finish_mean(sum, last, cnt, pc.h);
    }
}

// 64-lane kernels: the three mean heights of a leg (centre disc, default-track disc, centroid result; cpp:2520-2554)
// in ONE pass over three LDS arrays of terms in CircleIterator order; the f32 division runs once.
// Ordered f32 sum of a disc's visited elevations (getFootholdMeanHeight, cpp:2520-2554) prepared for a serial pass of
// PURE additions: what the reference decides per element is decided here while the elements still sit on different
// lanes — the element's term (the value, or -0.0f when >= 10, cpp:2539: s + (-0.0f) == s for every s) is what gets
// compacted into LDS, the count of summed elements is a ballot popcount, and `last` (the mean's fallback when nothing
// was summed, cpp:2547-2551) is the value of the highest visited lane of the last non-empty round.
struct TermSum {
    float* terms;  // >= (cells of the disc bounding box) floats, 16-byte aligned
    int n;         // elements visited
    int cnt;       // elements < 10
    float last;
};
template <int G>
__device__ __forceinline__ void term_push(const Grp<G>& g, TermSum& ts, bool vis, float v) {
    const unsigned long long mask = g.ballot(vis);
    const int rank = __builtin_popcountll(mask & ((1ull << g.sub) - 1ull));
    const bool inc = v < 10;
    if (vis) ts.terms[ts.n + rank] = inc ? v : -0.0f;
    ts.n += __builtin_popcountll(mask);
    ts.cnt += __builtin_popcountll(g.ballot(vis && inc));
    if (mask) ts.last = g.bcast(v, 63 - __builtin_clzll(mask));
}
template <int G>
__device__ __forceinline__ void push_disc(const Grp<G>& g, TermSum& ts, const DiscLoads& d) {
#pragma unroll
    for (int r = 0; r < disc_rounds<G>(); ++r) {
        const float v = __builtin_isfinite(d.e[r]) ? d.e[r] : 0.0f;  // cpp:2532-2537
        term_push(g, ts, d.vis[r] != 0, v);
    }
}
// The three sums of a leg side by side: lanes 0-15 walk the first array, 16-31 the second, 32-63 the third — the same
// instruction stream for all three.  Eight terms per pass (two 16-byte LDS reads, eight dependent additions); the arrays
// are padded with -0.0f to the common length first (they hold kBitsMaxBoxCells, a multiple of 8, floats each).
template <int G>
__device__ __forceinline__ void heights3_finish(const Grp<G>& g, float* hs, const TermSum& A, const TermSum& B, const TermSum& C, double h,
                                                float& zA, float& zB, float& zC) {
    static_assert(kBitsMaxBoxCells % 8 == 0, "heights3_finish reads whole groups of eight");
    const int nMax = max(A.n, max(B.n, C.n));
    const int padEnd = (nMax + 7) & ~7;
    {
        const int kA = A.n + g.sub, kB = B.n + g.sub, kC = C.n + g.sub;
        if (kA < padEnd) hs[kA] = -0.0f;
        if (kB < padEnd) hs[kBitsMaxBoxCells + kB] = -0.0f;
        if (kC < padEnd) hs[2 * kBitsMaxBoxCells + kC] = -0.0f;
    }
    bits_sync<G>();
    const int d = min(g.sub >> 4, 2);
    const float* p = hs + d * kBitsMaxBoxCells;
    float sum = 0.0f;
    for (int t0 = 0; t0 < padEnd; t0 += 8) {
        const float4 q0 = *reinterpret_cast<const float4*>(p + t0), q1 = *reinterpret_cast<const float4*>(p + t0 + 4);
        sum = sum + q0.x;
        sum = sum + q0.y;
        sum = sum + q0.z;
        sum = sum + q0.w;
        sum = sum + q1.x;
        sum = sum + q1.y;
        sum = sum + q1.z;
        sum = sum + q1.w;
    }
    const int cntBC = d == 1 ? B.cnt : C.cnt;
    const int cnt = d == 0 ? A.cnt : cntBC;
    const float lastBC = d == 1 ? B.last : C.last;
    const float last = d == 0 ? A.last : lastBC;
    const float z = finish_mean(sum, last, cnt, h);
    zA = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, z), 0));
    zB = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, z), 16));
    zC = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, z), 32));
}

// The reference rectangle in index space.  Cell centres x_i = base + res * (-i) are non-increasing in i, so
// {i : lo <= x_i < hi} = [iA, iB] with iA = min{i : x_i < hi}, iB = max{i : x_i >= lo}; each end is found by evaluating
// the reference's own comparison at the two indices next to the boundary predicted by (base - limit) * (1/res).
// Lane q of the group evaluates one predicate (q & 4: y axis, q & 2: lower limit, q & 1: second index); the
// prediction e is read back from the even lanes.
struct IndexRect {
    int iA, iB, jA, jB;
};
template <int G>
__device__ __forceinline__ IndexRect rectangle_index_bounds(const MapGeom& mg, double xlo, double xhi, double ylo, double yhi,
                                                            const Grp<G>& g) {
    // (the limits arrive by value: a select between FIELDS of the leg context would keep the whole struct in scratch)
    const int q = g.sub & 7;
    const bool isY = (q & 4) != 0, isLo = (q & 2) != 0;
    const double base = isY ? mg.baseY : mg.baseX;
    const double lim = isY ? (isLo ? ylo : yhi) : (isLo ? xlo : xhi);
    double qf = floor((base - lim) * mg.rinv);
    qf = fmin(fmax(qf, -1.0e9), 1.0e9);
    const int e = static_cast<int>(qf);
    // upper limit: P(e), P(e + 1) with P(i) = x_i < hi;  lower limit: Q(e + 1), Q(e) with Q(i) = x_i >= lo
    const int t = isLo ? e + 1 - (q & 1) : e + (q & 1);
    const double x = cell_pos(base, mg.res, t);
    const bool pred = isLo ? (x >= lim) : (x < lim);
    const unsigned b = static_cast<unsigned>(g.ballot(pred && g.sub < 8));
    const int eXhi = g.template bcast_c<0>(e), eXlo = g.template bcast_c<2>(e), eYhi = g.template bcast_c<4>(e), eYlo = g.template bcast_c<6>(e);
    IndexRect r;
    r.iA = (b & 1u) ? eXhi : ((b & 2u) ? eXhi + 1 : eXhi + 2);
    r.iB = (b & 4u) ? eXlo + 1 : ((b & 8u) ? eXlo : eXlo - 1);
    r.jA = (b & 16u) ? eYhi : ((b & 32u) ? eYhi + 1 : eYhi + 2);
    r.jB = (b & 64u) ? eYlo + 1 : ((b & 128u) ? eYlo : eYlo - 1);
    return r;
}

// Arbitrary polygons on the multi-word windows.  PNPOLY (Polygon::isInside, cpp:2138) counts, for a cell centre
// (px, py), the edges straddling py whose intersection abscissa lies beyond px; py and therefore the abscissae depend
// on the window COLUMN only (see column_crossings above).  With at most two crossings X0, X1 per column a cell is
// inside iff (px < X0) != (px < X1), and since cell centres px_i are non-increasing in the row index i each
// comparison is a row threshold t(X) = min{i : px_i < X} (found exactly, as in rectangle_index_bounds): column c is
// inside for the rows [min(t0, t1), max(t0, t1)).  Lane = column; writes (lo, hi) per column.  Returns false when
// some column has more than two crossings (non-convex polygon): the per-cell PNPOLY loop is then used.
// The columns [colLo, colHi] only (those a candidate's foot disc can touch), and each column sets its bits in the
// "enters" / "leaves" row arrays itself (see the caller): the interval stays in the lane's registers.
template <int G, int KW>
__device__ __forceinline__ bool window_column_rows(const MapGeom& mg, const LegCtx& c, const Grp<G>& g, int iw0, int jw0, int colLo, int colHi,
                                                   int NR, uint32_t* entersAt, uint32_t* leavesAt) {
    const double ninf = -__builtin_huge_val();
    bool over = false;
    for (int b = colLo + g.sub; b <= colHi; b += G) {
        const double py = cell_pos(mg.baseY, mg.res, jw0 + b);
        double X[2] = {ninf, ninf};
        // the (at most two) edges that straddle py, found with comparisons only; their abscissae afterwards — two division
        // sequences per column pass instead of one per polygon edge (the lanes' columns straddle different edges)
        int n = 0, e0 = 0, e1 = 0;
        for (int i = 0, j = c.nv - 1; i < c.nv; j = i++) {
            const bool cross = (c.vy[i] > py) != (c.vy[j] > py);
            e1 = (cross && n == 1) ? i : e1;
            e0 = (cross && n == 0) ? i : e0;
            n += cross ? 1 : 0;
        }
        over |= n > 2;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (n > u) {
                const int i = u == 0 ? e0 : e1;
                const int j = i == 0 ? c.nv - 1 : i - 1;
                const double vxi = c.vx[i], vyi = c.vy[i], vxj = c.vx[j], vyj = c.vy[j];
                const double ex = vxj - vxi;
                const double t = py - vyi;
                double xi = vxi;
                if (!(ex == 0.0 && fabs(t) <= DBL_MAX)) xi = ex * t / (vyj - vyi) + vxi;  // polygon_inside_fast
                X[u] = xi;
            }
        }
        int t[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            double qf = floor((mg.baseX - X[u]) * mg.rinv);
            qf = fmin(fmax(qf, -1.0e9), 1.0e9);
            const int e = static_cast<int>(qf);
            const bool p0 = cell_pos(mg.baseX, mg.res, e) < X[u], p1 = cell_pos(mg.baseX, mg.res, e + 1) < X[u];
            t[u] = p0 ? e : (p1 ? e + 1 : e + 2);
        }
        const int rl = min(t[0], t[1]) - iw0, rh = max(t[0], t[1]) - iw0;  // window rows [rl, rh)
        const uint32_t bit = 1u << (b & 31);
        const int wq = b >> 5;
        if (rl < rh && rh > 0 && rl < NR) {
            atomicOr(&entersAt[max(rl, 0) * KW + wq], bit);
            if (rh < NR) atomicOr(&leavesAt[rh * KW + wq], bit);
        }
    }
    return !g.any(over);
}

// Inclusive OR-scan over the 64 lanes of a wavefront (lane l gets the OR of lanes 0..l): four shifts inside the 16-lane
// DPP rows, then the last lane of a row into the next row, then lane 31 into the upper half.
__device__ __forceinline__ unsigned wave_or_scan(unsigned v) {
    int x = static_cast<int>(v);
    x |= __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);   // row_shr:1
    x |= __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);   // row_shr:2
    x |= __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);   // row_shr:4
    x |= __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);   // row_shr:8
    x |= __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);  // row_bcast:15 into rows 1 and 3
    x |= __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);  // row_bcast:31 into rows 2 and 3
    return static_cast<unsigned>(x);
}

// Distance (columns) from window column cj to the nearest set bit of a row of KW words; >= 1 << 20 when the row is empty.
template <int KW>
__device__ __forceinline__ int nearest_set_bit_distance(const uint32_t* row, int cj) {
    int best = 1 << 20;
#pragma unroll
    for (int q = 0; q < KW; ++q) {
        const int rel = cj - 32 * q;  // the centre column relative to this word
        const unsigned left = row[q] & bits_to(rel), right = row[q] & bits_from(rel);
        const int dl = left ? rel - (31 - __builtin_clz(left)) : (1 << 20);
        const int dr = right ? __builtin_ctz(right) - rel : (1 << 20);
        best = min(best, min(dl, dr));
    }
    return best;
}
// Minimum of a small non-negative value (< 128; larger values count as 127) over the lanes of a group, by ballots.
template <int G>
__device__ __forceinline__ int group_min7(const Grp<G>& g, int v) {
    v = min(v, 127);
    unsigned long long cand = g.ballot(true);
    int r = 0;
#pragma unroll
    for (int b = 6; b >= 0; --b) {
        const unsigned long long zero = g.ballot(((v >> b) & 1) == 0) & cand;
        if (zero) cand = zero;
        else r |= 1 << b;
    }
    return r;
}

// H_d(x): bit j = AND_{|t| <= d} x bit j + t (zeros beyond the row's words), by doubling: x & x>>1, & >>2, ... then centred
template <int KW>
__device__ __forceinline__ void erode_h(unsigned (&A)[KW], int d) {
    if (d <= 0) return;
    const int L = 2 * d + 1;
    unsigned T[KW];
    int span = 1;
    while (2 * span <= L) {  // A covers columns [j, j + span)
        row_shr<KW>(A, static_cast<unsigned>(span), T);
#pragma unroll
        for (int q = 0; q < KW; ++q) A[q] &= T[q];
        span *= 2;
    }
    if (span < L) {
        row_shr<KW>(A, static_cast<unsigned>(L - span), T);
#pragma unroll
        for (int q = 0; q < KW; ++q) A[q] &= T[q];
    }
    row_shl<KW>(A, static_cast<unsigned>(d), T);  // centre the interval: [j - d, j + d]
#pragma unroll
    for (int q = 0; q < KW; ++q) A[q] = T[q];
}

// checkCandidateFoothold (cpp:2085-2114) on the window's bit rows: first valid cell in SpiralIterator order.
// kOneCellFoot: the caller is a 3x3-only kernel, launched for one-cell foot discs only (launch_plan_bits): the erosion
// is compiled out (its code and live scalars cost the chain of those kernels 1 us of register allocation otherwise).
template <int G, int NRL, int KW, bool kOneCellFoot = false>
__device__ bool spiral_bits(const DevMap& m, const PlanConsts& pc, const SpiralLut& lut, const LutHead& head, const LegCtx& c,
                            const WinRows<NRL, KW>& w, const LegBits& lb, const Grp<G>& g, int iw0, int jw0, int& wi, int& wj,
                            const YEntry* ye = nullptr) {
    const int NR = lb.rows;  // allocated window rows (lanes beyond them hold nothing a search can touch)
    // Every candidate has |di|, |dj| <= nRings.  A centre so far off the map that none of them is inside it — poses that
    // walked off the map, or a feet polygon degenerated by the centroid track's "no case" (0,0,0) results — has no valid
    // candidate; without this test such a leg scans every round with in_range false, in every phase of every remaining
    // cycle (cfg-3: 122 of a pose's 128 searches, 1.3 of its 2.5 M clocks, and the kernel waits for its slowest pose).
    if (c.ici + c.nRings < 0 || c.ici - c.nRings >= m.g.rows || c.icj + c.nRings < 0 || c.icj - c.nRings >= m.g.cols) return false;
    if (G == 8) stamp_any(pc, c.cyc, 11);
    if (G == 64) stamp_any(pc, 0, 14);  // (profiling builds: the LAST search of the pose wins; see scratch/trace_seq.py)
    // generic 8-lane kernels: the first round's table entries are requested here, ahead of the P rows and the erosion
    uint4 tabFirst = make_uint4(0u, 0u, 0u, 0u);
    if constexpr (G == 8 && KW == 1 && !kOneCellFoot) tabFirst = reinterpret_cast<const uint4*>(lut.packed)[g.sub];
    // Window rows sized for the largest search radius a pose may ask for (fpe_set_max_leg_search_radius) are beyond the reach
    // of a leg with the usual radius: when every row such a leg's candidates and their foot discs can touch is held in the
    // lanes' FIRST row (one-wavefront-per-pose kernels with two rows per lane), the second row's share of the P rows, the
    // erosion and the ring skip is not computed at all (cfg-5: half of those stages)
    const int kLim = (G == 64 && NRL > 1 && pc.winH + c.nRings + pc.footReach < G) ? 1 : NRL;
    bool polyFolded = true;  // the polygon test is part of P (rectangle: always; other polygons: see below)
    unsigned Preg[NRL];      // single-word rows: this lane's P rows stay in registers for the erosion
#pragma unroll
    for (int k = 0; k < NRL; ++k) Preg[k] = 0u;
    // (1) per row: P = cells that do NOT fail checkCirclePolygonFoothold's per-cell test (cpp:2132-2138)
    if (c.rect) {
        IndexRect ir = rectangle_index_bounds(m.g, c.xlo, c.xhi, c.ylo, c.yhi, g);
        if (ye) {  // the column interval is chain-independent: taken from the hoisted y side (same evaluation)
            ir.jA = ye->jA;
            ir.jB = ye->jB;
        }
#pragma unroll
        for (int k = 0; k < NRL; ++k) {
            if (k >= kLim) continue;  // (rows no candidate of this leg can touch)
            const int ri = g.sub + G * k;
            const int i = iw0 + ri;
            const bool rowIn = i >= ir.iA && i <= ir.iB;
#pragma unroll
            for (int q = 0; q < KW; ++q) {
                const unsigned inside = rowIn ? ((ye && KW == 1) ? ye->pmask : range_word(ir.jA - jw0, ir.jB - jw0, q)) : 0u;
                if (ri < NR) lb.a[ri * KW + q] = ~w.F[k][q] | (~w.C[k][q] & inside);
                if constexpr (KW == 1) Preg[k] = ~w.F[k][0] | (~w.C[k][0] & inside);
            }
        }
    } else {
        bool folded = false;
        if constexpr (KW > 1) {
            static_assert(G == 64, "the column -> row transposition runs on whole wavefronts");
            // the polygon's row interval per window column (lane = column), then transposed into per-row column
            // masks by ballots over the columns, one window row at a time
            uint32_t* entersAt = lb.f;
            uint32_t* leavesAt = lb.h0;
            for (int idx = g.sub; idx < NR * KW; idx += G) {
                entersAt[idx] = 0u;
                leavesAt[idx] = 0u;
            }
            bits_sync<G>();
            // columns a candidate's foot disc can touch: within nRings + footReach of the centre column (winH) — one pass of
            // the wavefront instead of two for the usual radius on a 96-bit window
            const int reachCols = min(c.nRings + pc.footReach, pc.winH);
            folded = window_column_rows<G, KW>(m.g, c, g, iw0, jw0, max(pc.winH - reachCols, 0), min(pc.winH + reachCols, 32 * KW - 1), NR,
                                               entersAt, leavesAt);
            if (folded) {
                // Column intervals -> row masks without a ballot per row.  Column c is inside for the rows [lo_c, hi_c): it
                // ENTERS at row lo_c and LEAVES at row hi_c.  Each column sets its bit in the "enters" word of its first
                // row and in the "leaves" word of its end row (LDS atomic OR; two scratch row arrays that are free here);
                // an inclusive OR-scan over the rows (lane = row: four row shifts and two row broadcasts per word) then
                // gives, for every row, the columns that have entered and the columns that have left:
                //     inside(row) = entered(row) & ~left(row)
                // — the same set as the per-row comparison i >= lo_c && i < hi_c, by construction.
                bits_sync<G>();
                unsigned inside[NRL][KW];
                // rows a candidate's foot disc can touch: within nRings + footReach rows of the centre row (winH)
                const int reachRows = min(c.nRings + pc.footReach, pc.winH);
                const int rowLo = pc.winH - reachRows, rowHi = min(pc.winH + reachRows + 1, NR);  // NR: allocated rows
                unsigned carryIn[KW], carryOut[KW];
#pragma unroll
                for (int q = 0; q < KW; ++q) carryIn[q] = carryOut[q] = 0u;
#pragma unroll
                for (int k = 0; k < NRL; ++k) {
                    if (k >= kLim) continue;  // (rows no candidate of this leg can touch)
                    const int ri = g.sub + G * k;
#pragma unroll
                    for (int q = 0; q < KW; ++q) {
                        unsigned en = ri < NR ? entersAt[ri * KW + q] : 0u, lv = ri < NR ? leavesAt[ri * KW + q] : 0u;
                        en = wave_or_scan(en) | carryIn[q];
                        lv = wave_or_scan(lv) | carryOut[q];
                        carryIn[q] = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(en), 63));
                        carryOut[q] = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(lv), 63));
                        inside[k][q] = (ri >= rowLo && ri < rowHi) ? (en & ~lv) : 0u;
                    }
                }
#pragma unroll
                for (int k = 0; k < NRL; ++k) {
                    if (k >= kLim) continue;  // (rows no candidate of this leg can touch)
                    const int ri = g.sub + G * k;
#pragma unroll
                    for (int q = 0; q < KW; ++q)
                        if (ri < NR) lb.a[ri * KW + q] = ~w.F[k][q] | (~w.C[k][q] & inside[k][q]);
                }
            }
        }
        polyFolded = folded;
        if (!folded) {
#pragma unroll
            for (int k = 0; k < NRL; ++k) {
                if (k >= kLim) continue;  // (rows no candidate of this leg can touch)
                const int ri = g.sub + G * k;
#pragma unroll
                for (int q = 0; q < KW; ++q) {
                    if (ri >= NR) continue;
                    lb.a[ri * KW + q] = ~w.C[k][q];  // threshold only (C implies F); the polygon is tested per candidate below
                    lb.f[ri * KW + q] = w.F[k][q];
                    if constexpr (KW == 1) Preg[k] = ~w.C[k][0];
                }
            }
        }
    }
    bits_sync<G>();
    if (G == 8) stamp_any(pc, c.cyc, 12);
    if (G == 64) stamp_any(pc, 0, 15);
    // (2) erosion with the foot-disc offset table: E bit (row, col) = AND_k P(row + da_k, col + db_k)
    const uint32_t* E = lb.a;
    if (!kOneCellFoot && pc.nFoot > 1 && pc.nHW > 0) {
        // the two small tables in registers, fetched once (indexed inside the loops below they are a scalar load and a
        // wait per iteration): hwList[4] as one word, hwIdx[16] as two
        uint32_t hwListW;
        unsigned long long hwIdxLo, hwIdxHi;
        __builtin_memcpy(&hwListW, pc.hwList, 4);
        __builtin_memcpy(&hwIdxLo, pc.hwIdx, 8);
        __builtin_memcpy(&hwIdxHi, pc.hwIdx + 8, 8);
        // row-interval form: the disc's row +-a holds the columns [-w(a), w(a)], so
        //   E(row) = AND_a H_w(a)(P(row + a)) & H_w(a)(P(row - a)),   H_w(x) bit j = AND_{|t| <= w} x bit j + t
        // — a handful of shifts per distinct width instead of one shift per offset (45 offsets on a 0.5 cm map)
        if constexpr (KW == 1) {
            // single-word rows (8-lane kernels; measured against the nested form below: cfg-4 -2.5 %): H_w of this lane's
            // rows straight from the registers, H_w(x) = AND_{|t| <= w} x shifted by t, one array per distinct width ...
            for (int hw = 0; hw < pc.nHW; ++hw) {
                const int wdt = static_cast<int>((hwListW >> (8 * hw)) & 0xFFu);
                unsigned acc[NRL];
#pragma unroll
                for (int k = 0; k < NRL; ++k) acc[k] = Preg[k];
                for (int t = 1; t <= wdt; ++t) {
#pragma unroll
                    for (int k = 0; k < NRL; ++k) acc[k] &= (Preg[k] >> t) & (Preg[k] << t);
                }
#pragma unroll
                for (int k = 0; k < NRL; ++k)
                    if (g.sub + G * k < NR) lb.h0[hw * lb.hStride + g.sub + G * k] = acc[k];
            }
            bits_sync<G>();
            // ... then the rows +-a of the array of w(a) (a outermost: the reads of all of this lane's rows are in flight together)
            unsigned e[NRL];
#pragma unroll
            for (int k = 0; k < NRL; ++k) e[k] = ~0u;
            for (int a = 0; a <= pc.footReach; ++a) {
                const int hwOfRow = static_cast<int>(((a < 8 ? hwIdxLo : hwIdxHi) >> (8 * (a & 7))) & 0xFFu);
                const uint32_t* hrow = lb.h0 + hwOfRow * lb.hStride;
#pragma unroll
                for (int k = 0; k < NRL; ++k) {
                    const int ri = g.sub + G * k;
                    e[k] &= hrow[min(max(ri - a, 0), NR - 1)] & hrow[min(max(ri + a, 0), NR - 1)];
                }
            }
#pragma unroll
            for (int k = 0; k < NRL; ++k)
                if (g.sub + G * k < NR) lb.a[g.sub + G * k] = e[k];  // the P rows are dead: E takes their place
        } else {
            // Multi-word rows, vertical first: H_w distributes over AND and H_a(H_b(x)) = H_(a + b)(x) (zero fill included),
            // and a disc's widths do not grow with |a| (derive_foot_offsets checks it), so with V_q = AND of the rows
            // P(row +- a) whose width is the q-th distinct one, w_0 > w_1 > ...:
            //   E = H_w0(V_0) & H_w1(V_1) & ... = H_w(n-1)( ... H_(w1 - w2)( H_(w0 - w1)(V_0) & V_1 ) & V_2 ... )
            // — the rows are ANDed as they are read (no intermediate arrays, no second pass), and the horizontal work is
            // w_0 single steps per row in total instead of one full H_w per distinct width (cfg-5: -5 %).
            unsigned acc[NRL][KW];
#pragma unroll
            for (int k = 0; k < NRL; ++k)
#pragma unroll
                for (int q = 0; q < KW; ++q) acc[k][q] = ~0u;
            int curW = static_cast<int>(hwListW & 0xFFu);  // w(0): hwIdx[0] == 0 by construction
            for (int a = 0; a <= pc.footReach; ++a) {
                const int hwOfRow = static_cast<int>(((a < 8 ? hwIdxLo : hwIdxHi) >> (8 * (a & 7))) & 0xFFu);
                const int wa = static_cast<int>((hwListW >> (8 * hwOfRow)) & 0xFFu);
                if (wa != curW) {  // uniform: the next (narrower) group of rows
#pragma unroll
                    for (int k = 0; k < NRL; ++k) {
                        if (k >= kLim) continue;  // (rows no candidate of this leg can touch)
                        erode_h<KW>(acc[k], curW - wa);
                    }
                    curW = wa;
                }
#pragma unroll
                for (int k = 0; k < NRL; ++k) {
                    if (k >= kLim) continue;
                    const int ri = g.sub + G * k;
                    const uint32_t* up = lb.a + min(max(ri - a, 0), NR - 1) * KW;
                    const uint32_t* dn = lb.a + min(max(ri + a, 0), NR - 1) * KW;
#pragma unroll
                    for (int q = 0; q < KW; ++q) acc[k][q] &= up[q] & dn[q];
                }
            }
#pragma unroll
            for (int k = 0; k < NRL; ++k) {
                if (k >= kLim) continue;
                const int ri = g.sub + G * k;
                erode_h<KW>(acc[k], curW);
#pragma unroll
                for (int q = 0; q < KW; ++q)
                    if (ri < NR) lb.h0[ri * KW + q] = acc[k][q];
            }
            E = lb.h0;
        }
        bits_sync<G>();
    } else if (!kOneCellFoot && pc.nFoot > 1) {
#pragma unroll
        for (int k = 0; k < NRL; ++k) {
            if (k >= kLim) continue;  // (rows no candidate of this leg can touch)
            const int ri = g.sub + G * k;
            unsigned e[KW];
#pragma unroll
            for (int q = 0; q < KW; ++q) e[q] = ~0u;
            for (int f = 0; f < pc.nFoot; ++f) {
                const int da = c.footDa[f], db = c.footDb[f];
                const uint32_t* p = lb.a + min(max(ri + da, 0), NR - 1) * KW;
                // shift the row by db columns (|db| <= footReach < 32): bit j of the result = bit j + db of the row
#pragma unroll
                for (int q = 0; q < KW; ++q) {
                    const unsigned cur = p[q];
                    const unsigned up = q + 1 < KW ? p[q + 1] : 0u, dn = q > 0 ? p[q - 1] : 0u;
                    const unsigned sh = db >= 0 ? __builtin_amdgcn_alignbit(up, cur, static_cast<unsigned>(db))
                                                : __builtin_amdgcn_alignbit(cur, dn, static_cast<unsigned>(32 + db));
                    e[q] &= sh;
                }
            }
#pragma unroll
            for (int q = 0; q < KW; ++q)
                if (ri < NR) lb.h0[ri * KW + q] = e[q];
        }
        bits_sync<G>();
        E = lb.h0;
    }
    if (G == 8) stamp_any(pc, c.cyc, 13);
    if (G == 64) stamp_any(pc, 2, 14);
    // Forms of the candidate scan, chosen per kernel shape by measurement (A/B on the BASELINE configurations): the
    // one-wavefront-per-pose kernels (64- and 96-bit rows) take straight-line rounds of 64 candidates with the ring skip
    // (cfg-5: 0.98 -> 0.76 ms in round 2); the generic 8-lane kernels four packed table entries per lane and round (below); the
    // 3x3-only 8-lane kernels, which come here only for ranks beyond their own first sixteen, the straight-line rounds
    // without the skip.  (The branchy rounds at the end of this function are the round-1 form, kept for A/B builds.)
#ifdef FPE_SKIP_ALL
    constexpr bool kFlatRounds = true;
    constexpr bool kRingSkip = true;
#else
    constexpr bool kFlatRounds = true;
    constexpr bool kRingSkip = KW >= 2;
#endif
    if constexpr (G == 8 && KW == 1 && !kOneCellFoot) {
        // (3) generic 8-lane kernels (the 3x3-only ones evaluate ranks 0-15 in leg_fast8m and come here for the rest; the
        // scan below cost them registers: measured +3 % on the headline): FOUR candidates per lane and round (rank k = 32 * round + 4 * lane + u, one uint4 of packed
        // table entries per lane), so the first valid cell in spiral order is the lowest (lane, u) with a pass bit.  A
        // candidate lies within nRings <= winH rows and columns of the window's centre: its E bit is read without range
        // tests; cells outside the map are cleared from E first (wave-uniform, windows at the map's border only).
        const int M = c.nCand;
        const int rowW = c.ici - iw0, colW = c.icj - jw0;  // the centre inside the window: (winH, winH)
        uint32_t* Ew = const_cast<uint32_t*>(E);
        const bool border = iw0 < 0 || jw0 < 0 || iw0 + NR > m.g.rows || jw0 + 32 > m.g.cols;
        if (__ballot(border) != 0ull) {
            const uint32_t colIn = range_word(-jw0, m.g.cols - 1 - jw0, 0);
#pragma unroll
            for (int k = 0; k < NRL; ++k) {
                if (k >= kLim) continue;  // (rows no candidate of this leg can touch)
                const int ri = g.sub + G * k;
                if (ri < NR) Ew[ri] = static_cast<unsigned>(iw0 + ri) < static_cast<unsigned>(m.g.rows) ? (Ew[ri] & colIn) : 0u;
            }
            bits_sync<G>();
        }
        const uint4* tab = reinterpret_cast<const uint4*>(lut.packed);
        const int nRounds = (M + 31) >> 5;
        uint4 nxt = tabFirst;
        for (int round = 0; round < nRounds; ++round) {
            const uint4 cur = nxt;
            nxt = tab[(round + 1) * G + g.sub];  // (the table is padded by one round)
            const uint32_t wds[4] = {cur.x, cur.y, cur.z, cur.w};
            const int k0 = round * 32 + 4 * g.sub;
            bool ok[4], outer[4];
            bool anyOuterOk = false;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool liveU = k0 + u < M;
                const int di = static_cast<int8_t>(wds[u] & 0xFFu), dj = static_cast<int8_t>((wds[u] >> 8) & 0xFFu);
                const int r = static_cast<int>((wds[u] >> 16) & 0xFFu);
                const uint32_t row = Ew[liveU ? rowW + di : 0];  // (entries beyond this leg's radius may point outside the window)
                ok[u] = liveU & (((row >> ((colW + dj) & 31)) & 1u) != 0u);
                // SpiralIterator::generateRing filters rings nRings-1 and nRings by isInside; the centre cell (ring 0) is
                // pushed unfiltered by the constructor
                outer[u] = (r >= 1) & (r + 1 >= c.nRings);
                anyOuterOk |= ok[u] & outer[u];
            }
            if (__ballot(anyOuterOk) != 0ull) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int di = static_cast<int8_t>(wds[u] & 0xFFu), dj = static_cast<int8_t>((wds[u] >> 8) & 0xFFu);
                    if (ok[u] & outer[u]) ok[u] = cell_in_disc(m.g, c.ici + di, c.icj + dj, c.cx, c.cy, c.R2);
                }
            }
            if (!polyFolded && __ballot(ok[0] | ok[1] | ok[2] | ok[3]) != 0ull) {
                // arbitrary polygon not folded into P: every FINITE cell of the foot disc must lie inside it (cpp:2138)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (!ok[u]) continue;
                    const int i = c.ici + static_cast<int8_t>(wds[u] & 0xFFu), j = c.icj + static_cast<int8_t>((wds[u] >> 8) & 0xFFu);
                    for (int f = 0; f < pc.nFoot; ++f) {
                        const int qi = i + c.footDa[f], qj = j + c.footDb[f];
                        if (win_bit<KW>(lb.f, NR, qi - iw0, qj - jw0) == 0u) continue;
                        if (!polygon_inside_fast(c.vx, c.vy, c.nv, cell_pos(m.g.baseX, m.g.res, qi), cell_pos(m.g.baseY, m.g.res, qj))) {
                            ok[u] = false;
                            break;
                        }
                    }
                }
            }
            const unsigned mask = static_cast<unsigned>(g.ballot(ok[0] | ok[1] | ok[2] | ok[3]));
            if (mask) {
                const uint32_t mine = ok[0] ? wds[0] : (ok[1] ? wds[1] : (ok[2] ? wds[2] : wds[3]));
                const uint32_t win = g.bcast(mine, __builtin_ctz(mask));
                wi = c.ici + static_cast<int8_t>(win & 0xFFu);
                wj = c.icj + static_cast<int8_t>((win >> 8) & 0xFFu);
                return true;
            }
        }
    } else if constexpr (kFlatRounds) {
        // (3) candidates in rank order, lane = rank; lowest set ballot bit = argmin of rank.  Straight-line per round
        // (per-lane `if` chains are compiled into exec-mask branches): lanes beyond the table and cells outside the map carry
        // ok = false through unconditional, clamped evaluations; the disc filter of the outer rings and the per-candidate
        // polygon test of an unfolded polygon sit behind wave-uniform branches.
        const int M = c.nCand;
        // (2b) Where the scan can start.  A candidate needs its E bit, and the iterator's ring of a cell is
        // trunc(sqrt(di^2 + dj^2)) (fpe_host.cpp::build_spiral_table): the nearest E bit of every window row (lane = row)
        // gives the lowest ring rho that holds any E bit at all.  Ranks below ringStart[rho] cannot be valid: the scan
        // starts at the round containing ringStart[rho], and a window without an E bit inside the search radius has no
        // candidate — the searches that used to walk every round to the end (a pose stuck on bad terrain repeats them in
        // every remaining cycle; with one wavefront per pose such poses set the kernel's duration).
        int startBase = 0;
        bool nearHit = false;
        if constexpr (kRingSkip) {
            // the usual search has a pass bit within three rows and columns of the centre (ring <= 4: among the first 49
            // ranks, i.e. in the first round of 64): one ballot spares it the ring computation below
            bool near = false;
#pragma unroll
            for (int k = 0; k < NRL; ++k) {
                if (k >= kLim) continue;  // (rows no candidate of this leg can touch)
                const int ri = g.sub + G * k;
                const int cj = c.icj - jw0;
                uint32_t bits = 0u;
#pragma unroll
                for (int q = 0; q < KW; ++q)
                    bits |= E[min(ri, NR - 1) * KW + q] & range_word(max(cj - 3, -jw0), min(cj + 3, m.g.cols - 1 - jw0), q);
                near |= ri < NR && abs(ri - (c.ici - iw0)) <= 3 && bits != 0u && static_cast<unsigned>(iw0 + ri) < static_cast<unsigned>(m.g.rows);
            }
            nearHit = g.any(near);
        }
        if (kRingSkip && !nearHit) {
            int ringRow = 1 << 20;
#pragma unroll
            for (int k = 0; k < NRL; ++k) {
                if (k >= kLim) continue;  // (rows no candidate of this leg can touch)
                const int ri = g.sub + G * k;
                const int a = abs(ri - (c.ici - iw0));
                uint32_t rowIn[KW];  // the row's E bits on columns inside the map (cells outside it pass every test but are no candidates)
#pragma unroll
                for (int q = 0; q < KW; ++q) rowIn[q] = E[min(ri, NR - 1) * KW + q] & range_word(-jw0, m.g.cols - 1 - jw0, q);
                const bool rowInMap = static_cast<unsigned>(iw0 + ri) < static_cast<unsigned>(m.g.rows);
                const int d = rowInMap ? nearest_set_bit_distance<KW>(rowIn, c.icj - jw0) : (1 << 20);
                const int n2 = a * a + d * d;  // <= 2 * 127^2 when in reach: exact in f32
                int rr = static_cast<int>(__builtin_sqrtf(static_cast<float>(min(n2, 1 << 16))));
                rr = (rr + 1) * (rr + 1) <= n2 ? rr + 1 : rr;  // v_sqrt_f32 is 1 ulp: settle floor(sqrt(n2)) exactly
                rr = rr * rr > n2 ? rr - 1 : rr;
                if (ri < NR && d < (1 << 20)) ringRow = min(ringRow, rr);
            }
            const int rho = group_min7<G>(g, ringRow);
            if (rho > c.nRings) return false;
            if (__ballot(rho >= 2) != 0ull) {  // uniform: the usual search (a pass bit in rings 0-1) needs no table lookup
                const int first = lut.ringStart[min(rho, lut.maxRing)];
                startBase = rho >= 2 ? (first / G) * G : 0;
            }
        }
        int round = startBase / G;
        if (G == 64) stamp_any(pc, 2, 15);
        if (G == 64) stamp_value(pc, 4, 14, round);
        int nDi = 0, nDj = 0, nR = c.nRings;
        if (__ballot(round >= kLutHeadRounds) != 0ull) {  // uniform: a late start reads its first round's entries here
            const int kn = min(startBase + g.sub, M - 1);
            nDi = lut.di[kn];
            nDj = lut.dj[kn];
            nR = lut.ring[kn];
        }
        for (int base = startBase; base < M; base += G, ++round) {
            const int k = base + g.sub;
            const bool live = k < M;
            int di, dj, r;
            if (round < kLutHeadRounds) {  // uniform
                const int e = round == 0 ? head.dij[0] : head.dij[1];
                di = static_cast<int16_t>(e & 0xFFFF);
                dj = e >> 16;
                r = round == 0 ? head.ring[0] : head.ring[1];
            } else {
                di = nDi;
                dj = nDj;
                r = nR;
            }
            if (round + 1 >= kLutHeadRounds) {
                // uniform: the next round's table entries, untouched until then (their latency is this round's work);
                // clamped index instead of a lane-dependent branch
                const int kn = min(k + G, M - 1);
                nDi = lut.di[kn];
                nDj = lut.dj[kn];
                nR = lut.ring[kn];
            }
            const int i = c.ici + di, j = c.icj + dj;
            bool ok = live & in_range(i, j, m.g.rows, m.g.cols);
            // SpiralIterator::generateRing filters rings nRings-1 and nRings by isInside; the centre cell (ring 0) is
            // pushed unfiltered by the constructor
            const bool outer = (r >= 1) & ((r == c.nRings) | (r + 1 == c.nRings));
            if (__ballot(ok & outer) != 0ull) {
                const bool inDisc = cell_in_disc(m.g, i, j, c.cx, c.cy, c.R2);
                ok = ok & (!outer | inDisc);
            }
            ok = ok & (win_bit<KW>(E, NR, i - iw0, j - jw0) != 0u);
            if (!polyFolded && __ballot(ok) != 0ull) {
                if (ok) {
                    // arbitrary polygon not folded into P: every FINITE cell of the foot disc must lie inside it (cpp:2138)
                    for (int f = 0; f < pc.nFoot; ++f) {
                        const int qi = i + c.footDa[f], qj = j + c.footDb[f];
                        if (win_bit<KW>(lb.f, NR, qi - iw0, qj - jw0) == 0u) continue;
                        if (!polygon_inside_fast(c.vx, c.vy, c.nv, cell_pos(m.g.baseX, m.g.res, qi), cell_pos(m.g.baseY, m.g.res, qj))) {
                            ok = false;
                            break;
                        }
                    }
                }
            }
            const unsigned long long mask = g.ballot(ok);
            if (mask) {
                const int l = __builtin_ctzll(mask);
                wi = g.bcast(i, l);
                wj = g.bcast(j, l);
                if (G == 64) stamp_any(pc, 3, 14);
                if (G == 64) stamp_value(pc, 4, 15, round);
                return true;
            }
        }
    } else {
        // (3) candidates in rank order, lane = rank; lowest set ballot bit = argmin of rank
        const int M = c.nCand;
        int round = 0;
        int nDi = 0, nDj = 0, nR = c.nRings;
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
            if (k < M) {
                i = c.ici + di;
                j = c.icj + dj;
                ok = in_range(i, j, m.g.rows, m.g.cols);
                // SpiralIterator::generateRing filters rings nRings-1 and nRings by isInside; the centre cell (ring 0)
                // is pushed unfiltered by the constructor
                if (ok && r >= 1 && (r == c.nRings || r + 1 == c.nRings)) ok = cell_in_disc(m.g, i, j, c.cx, c.cy, c.R2);
                if (ok) ok = win_bit<KW>(E, NR, i - iw0, j - jw0) != 0u;
                if (ok && !polyFolded) {
                    // arbitrary polygon not folded into P: every FINITE cell of the foot disc must lie inside it (cpp:2138)
                    for (int f = 0; f < pc.nFoot; ++f) {
                        const int qi = i + c.footDa[f], qj = j + c.footDb[f];
                        if (win_bit<KW>(lb.f, NR, qi - iw0, qj - jw0) == 0u) continue;
                        if (!polygon_inside_fast(c.vx, c.vy, c.nv, cell_pos(m.g.baseX, m.g.res, qi), cell_pos(m.g.baseY, m.g.res, qj))) {
                            ok = false;
                            break;
                        }
                    }
                }
            }
            const unsigned long long mask = g.ballot(ok);
            if (mask) {
                const int l = __builtin_ctzll(mask);
                wi = g.bcast(i, l);
                wj = g.bcast(j, l);
                return true;
            }
        }
    }
    return false;
}

// One-wavefront-per-pose kernels: what a leg's four output records are made of, staged in LDS by the leg's lane 0 and
// finished for a few cycles at a time by one lane per (cycle, leg): full records side by side instead of eight
// single-lane store instructions per leg — and, since round 3, the MEAN HEIGHTS leave the chain as well.  Nothing a later
// phase reads depends on a height (the feet-polygon centre uses x and y, cpp:2421-2463; records are write-only), so the
// chain only deposits which cells of each CircleIterator bounding box were visited (two 64-bit ballots per disc) and
// where the box lies; flush_seqrec reads those elevations itself and runs the reference's ordered f32 sums
// (cpp:2520-2554), up to 32 units side by side instead of one leg at a time (compaction into LDS, three serial sums,
// three divisions per leg-phase: 15-19 % of a leg's clocks on cfg-3 / cfg-5).
struct SeqRecBase {
    double nomX, nomY, cenX, cenY, defX, defY;
    float nomZ, cenZ, defZ;  // final values of the heights that were NOT deferred (see flags)
    int32_t nomRow, nomCol, cenRow, cenCol;
    uint32_t flags;  // nominal valid | source << 8 | centroid code << 16 | kSeqDefer* << 24
};
struct SeqRec : SeqRecBase {  // kernels that defer the heights (64-bit rows)
    int32_t aI0, aJ0, aNj;  // centre disc: bounding box origin and width (cells in row-major order t = a * nj + b)
    int32_t bI0, bJ0, bNj;  // default-track disc
    uint32_t pad[2];
    unsigned long long visA[2], visB[2];  // bit t of word t / 64: cell t of the box is a member inside the map
};
static_assert(sizeof(SeqRecBase) == 80 && sizeof(SeqRec) == 144 && sizeof(SeqRec) % 16 == 0, "SeqRec layout");
template <int KW>
using SeqRecOf = typename std::conditional<(KW <= kSeqDeferMaxKW), SeqRec, SeqRecBase>::type;
constexpr uint32_t kSeqDeferA = 1u << 24;  // zA = mean height of the centre disc, to be computed by flush_seqrec
constexpr uint32_t kSeqDeferB = 1u << 25;  // zB (default track)
constexpr uint32_t kSeqDeferC = 1u << 26;  // zC = mean height of the cell-centred disc of (cenRow, cenCol) (offset table)
constexpr uint32_t kSeqCIsA = 1u << 27;    // zC = zA (whole region valid: the height at the centre, cpp:1687)

// One swing leg of one phase on the bit window: the three tracks' next positions, the centroid method
// (cpp:1605-1997) and checkFoothold (cpp:2001-2036) around the centroid track's position, the mean heights.
// kDirect: the results stay in registers (LegCommit) for a commit decided by wave ballot (8-lane kernels); otherwise
// they are staged in PoseShared::nxt / valid (one-wavefront-per-pose kernels, legs in sequence).
template <int G, int NRL, int KW, bool kMid, bool kDirect>
__device__ __forceinline__ void leg_phase_bits(const DevMap& m, const BitMap& bm, const PlanConsts& pc, const SpiralLut& lut,
                                               const LutHead& head, PoseShared& sh, const LegBits& lb, const Grp<G>& g, int leg,
                                               const LegStatic& ls, double y0, double adjY, double advance, int cyc, int nCycles,
                                               int b, bool live, const fpe_plan_out& out, LegCommit* lc, SeqRecOf<KW>* recs = nullptr,
                                               int* validOut = nullptr) {
    const float Rf = ls.Rf;
    const int polyKind = ls.polyKind;
    const LegConst& lk = ls.lk;
    const double biasX = ls.biasX, biasY = ls.biasY;
    // next default positions of this leg on the three tracks (cpp:2199-2213, 2270-2284)
    const double Ny = y0 + adjY;                         // cpp:2201
    const double nx0 = (sh.ctr[0] + advance) + biasX;  // cpp:2199, 2414
    const double nx1 = (sh.ctr[1] + advance) + biasX;
    const double nx2 = (sh.ctr[2] + advance) + biasX;
    const double ny = Ny + biasY;                        // identical on the three tracks
    if (polyKind != 0 && g.sub == 0) {  // hexagon vertices from the NOMINAL track's position (build-defined, App. E)
        const double r = static_cast<double>(Rf);
        double* vx = sh.polyX[leg];
        double* vy = sh.polyY[leg];
        const double hx = 0.5 * r;
        const double hy = (0.5 * r) * 0.8660254037844386;
        vx[0] = nx2 + r;   vy[0] = ny;
        vx[1] = nx2 + hx;  vy[1] = ny - hy;
        vx[2] = nx2 - hx;  vy[2] = ny - hy;
        vx[3] = nx2 - r;   vy[3] = ny;
        vx[4] = nx2 - hx;  vy[4] = ny + hy;
        vx[5] = nx2 + hx;  vy[5] = ny + hy;
    }
    if (G == 64 && polyKind != 0) bits_sync<G>();  // the vertices are read by the other lanes of the wavefront
    LegCtx c;
    c.cyc = cyc;
    c.cx = nx1;  // centre from the CENTROID track (cpp:861-862)
    c.cy = ny;
    c.nv = (polyKind == 0) ? 4 : 6;
    {
        const double r = static_cast<double>(Rf);  // getSearchPolygon's rectangle around the NOMINAL track (cpp:2496-2517)
        c.rect = polyKind == 0;
        c.xhi = nx2 + r;
        c.xlo = nx2 - r;
        c.yhi = ny + 0.5 * r;
        c.ylo = ny - 0.5 * r;
    }
    c.vx = sh.polyX[leg];
    c.vy = sh.polyY[leg];
    c.footDa = sh.footDa;
    c.footDb = sh.footDb;
    c.footOff = sh.footOff;
    c.R2 = lk.R2;
    c.nRings = lk.nRings;
    c.nCand = lk.nCand;
    c.ti0 = c.tj0 = 0;
    c.ici = c.icj = 0;

    NominalOut no;
    CentroidOut co;
    float zDefault = static_cast<float>(static_cast<double>(0.0f) + pc.h);  // value when no cell is visited
    float* scratch = reinterpret_cast<float*>(lb.a);
    const bool wantDefault = out.default_next != nullptr;
    // one wavefront per pose: the mean heights are deferred to flush_seqrec (SeqRec); what the chain deposits for them
    // (64-bit-row kernels: 1 cm maps, boxes of <= 25 cells, layers that stay in the L2s.  The 96-bit-row kernels keep the
    // heights in the chain: their boxes hold up to 81 cells of a layer that does not fit the L2s, and a flush that waits for
    // eleven dependent load batches per disc costs more than the chain's overlapped loads — measured on cfg-5: +8 %)
    constexpr bool kDeferH = (G == 64) && !kDirect && KW <= kSeqDeferMaxKW;
    uint32_t deferFlags = 0u;
    unsigned long long visA0 = 0ull, visA1 = 0ull, visB0 = 0ull, visB1 = 0ull;
    int aI0 = 0, aJ0 = 0, aNj = 1, bI0 = 0, bJ0 = 0, bNj = 1;
    if (!ls.radiusOk || !centre_usable(c.cx, c.cy)) {
        nominal_invalid(no, c.cx, c.cy, ls.radiusOk ? 2 : 3);
        co.x = co.y = 0.0; co.z = 0.0f; co.row = co.col = -1; co.code = 6;
        if (wantDefault && centre_usable(nx0, ny)) {  // cpp:2289-2301 (leg search skipped: radius / centre unusable)
            const BBox dbox = circle_bbox_fast(m.g, nx0, ny, pc.rf);
            bool unused;
            zDefault = disc_pass_direct<G, false>(m, pc, nx0, ny, dbox, g, unused, scratch);
        }
    } else {
        // corner lanes: box 0 = centre foot disc, box 1 = centroid rectangle, box 2 = default-track disc,
        // box 3 = getIndex(centre)
        const Box b0{c.cx, c.cy, pc.rf, pc.rf}, b1{c.cx, c.cy, 0.5 * lk.lx, 0.5 * lk.ly};
        const Box b2{nx0, ny, pc.rf, pc.rf};
        Corners<G, 16> cs;
        cs.eval(m.g, g, b0, b1, b2, b0, 0x8u);
        const BBox bb = cs.template bbox<0>(g);
        const BBox rbox = cs.template bbox<1>(g);
        const BBox dbox = cs.template bbox<2>(g);
        c.ici = cs.template get<12>(g);
        c.icj = cs.template get<13>(g);
        const Submap sm = submap_from_corners(m.g, rbox, cs.box_within(1), c.cx, c.cy);
        const int iw0 = c.ici - pc.winH, jw0 = c.icj - pc.winH;
        stamp(pc, cyc, 2);
        // one memory round trip: the window's bit rows and the elevation of the two discs around known centres
        uint4 grp[NRL][KW + 1];
        win_issue<G, NRL, KW>(bm, m.g, g, iw0, jw0, grp);
        DiscLoads dc, dd;
        disc_issue<G, false, kMid, !kDeferH>(m, pc, c.cx, c.cy, bb, g, dc);
        const bool dfltUsable = wantDefault && centre_usable(nx0, ny);
        if (dfltUsable) disc_issue<G, false, kMid, !kDeferH>(m, pc, nx0, ny, dbox, g, dd);
        stamp(pc, cyc, 3);
        WinRows<NRL, KW> w;
        win_finish<NRL, KW>(jw0, grp, w);
#pragma unroll
        for (int k = 0; k < NRL; ++k)
#pragma unroll
            for (int q = 0; q < KW; ++q)
                if (g.sub + G * k < lb.rows) lb.a[(g.sub + G * k) * KW + q] = w.Df[k][q];
        const CentroidScan sc = rows_from_bits<G, NRL, KW>(sm, w, g, iw0, jw0);
        bits_sync<G>();
        stamp(pc, cyc, 4);
        const bool defaultOk = default_ok_bits<G, KW, kMid>(m, pc, c.cx, c.cy, bb, dc, lb.a, lb.rows, iw0, jw0, g);  // cpp:2012
        bits_sync<G>();  // lb doubles as scratch below
        bool unused;
        float zCentre = 0.0f;
        TermSum osA{nullptr, 0, 0, 0.0f}, osB{nullptr, 0, 0, 0.0f}, osC{nullptr, 0, 0, 0.0f};
        if constexpr (kDeferH) {
            // membership of the two discs around known centres as ballots over the bounding boxes' cells (t = round * 64 +
            // lane, row-major: CircleIterator order); a box beyond the two rounds (never with bits_supported's bound on
            // the foot radius) is walked here
            if (dc.pipelined) {
                visA0 = g.ballot(dc.vis[0] != 0);
                visA1 = g.ballot(dc.vis[1] != 0);
                aI0 = bb.i0; aJ0 = bb.j0; aNj = max(bb.nj, 1);
                deferFlags |= kSeqDeferA;
            } else {
                zCentre = disc_pass_direct<G, false>(m, pc, c.cx, c.cy, bb, g, unused, scratch);
            }
            if (dfltUsable) {
                if (dd.pipelined) {
                    visB0 = g.ballot(dd.vis[0] != 0);
                    visB1 = g.ballot(dd.vis[1] != 0);
                    bI0 = dbox.i0; bJ0 = dbox.j0; bNj = max(dbox.nj, 1);
                    deferFlags |= kSeqDeferB;
                } else {
                    zDefault = disc_pass_direct<G, false>(m, pc, nx0, ny, dbox, g, unused, scratch);
                }
            }
        } else if constexpr (G == 64) {
            // the three ordered height sums run side by side at the end of the leg (heights3_finish): here the visited
            // elevations of the two discs around known centres are only compacted into LDS
            osA.terms = lb.hs;
            osB.terms = lb.hs + kBitsMaxBoxCells;
            osC.terms = lb.hs + 2 * kBitsMaxBoxCells;
            push_disc(g, osA, dc);
            if (dfltUsable) push_disc(g, osB, dd);
        } else {
            zCentre = disc_consume<G, false, kMid>(m, pc, c.cx, c.cy, bb, g, dc, unused, scratch);  // cpp:2029
        }
        stamp(pc, cyc, 5);
        constexpr bool kOneCell = kMid;  // the 3x3-only variants are launched for one-cell foot discs
        CentroidPendingBits cp;
        centroid_begin_bits<G, kOneCell, !kDeferH>(m, pc, c, sm, sc, zCentre, g, cp);                        // cpp:818-821
        stamp(pc, cyc, 6);
        if constexpr (G != 64) {
            if (dfltUsable) zDefault = disc_consume<G, false, kMid>(m, pc, nx0, ny, dbox, g, dd, unused, scratch);  // cpp:2289-2301
        }
        stamp(pc, cyc, 7);
        if (defaultOk) {
            no.valid = 1;
            no.source = 0;
            no.row = c.ici;
            no.col = c.icj;
            no.x = c.cx;  // cpp:2016-2017
            no.y = c.cy;
        } else {
            nominal_invalid(no, c.cx, c.cy, 2);
            int wi = 0, wj = 0;
            bits_sync<G>();
#ifdef FPE_TRACE
            const long long tSp0 = __builtin_readcyclecounter();
#endif
            const bool spFound = spiral_bits<G, NRL, KW>(m, pc, lut, head, c, w, lb, g, iw0, jw0, wi, wj);
#ifdef FPE_TRACE
            if (g.sub == 0 && G == 64) {
                sh.pad[0] += static_cast<int>((__builtin_readcyclecounter() - tSp0) >> 4);
                sh.pad[1] += 1;
                sh.pad[2] += spFound ? 0 : 1;
            }
#endif
            if (spFound) {  // cpp:2022
                no.valid = 1;
                no.source = 1;
                no.row = wi;
                no.col = wj;
                no.x = cell_pos(m.g.baseX, m.g.res, wi);  // cpp:2105-2107
                no.y = cell_pos(m.g.baseY, m.g.res, wj);
            }
            bits_sync<G>();
        }
        stamp(pc, cyc, 8);
        if constexpr (kDeferH) {
            if (cp.needDisc != 0) deferFlags |= kSeqDeferC;          // the result's own cell-centred disc (offset table)
            else if (cp.o.code == 0) {                               // whole region valid: the height at the centre (cpp:1687)
                if (deferFlags & kSeqDeferA) deferFlags |= kSeqCIsA;
                else cp.o.z = zCentre;
            }
            stamp(pc, cyc, 11);
            stamp(pc, cyc, 12);
        } else if constexpr (G == 64) {
            if (cp.needDisc != 0) {
                const float v = __builtin_isfinite(cp.e[0]) ? cp.e[0] : 0.0f;
                term_push(g, osC, cp.vis[0] != 0, v);
            }
            stamp(pc, cyc, 11);
            float zB, zC;
            heights3_finish(g, lb.hs, osA, osB, osC, pc.h, zCentre, zB, zC);
            stamp(pc, cyc, 12);
            if (dfltUsable) zDefault = zB;
            if (cp.needDisc != 0) cp.o.z = zC;
            else if (cp.o.code == 0) cp.o.z = zCentre;  // whole region valid: the height at the centre (cpp:1687)
        } else {
            if (cp.needDisc != 0) cp.o.z = centroid_height_bits<G, kOneCell>(pc, g, cp, scratch);
        }
        if (no.valid) no.z = zCentre;  // z at the DEFAULT centre, for a spiral candidate too (cpp:2029)
        co = cp.o;
    }
    if constexpr (kDirect) {
        lc->valid = no.valid;
        lc->v[0][0] = nx0;   lc->v[0][1] = ny;    lc->v[0][2] = static_cast<double>(zDefault);
        lc->v[1][0] = co.x;  lc->v[1][1] = co.y;  lc->v[1][2] = static_cast<double>(co.z);
        lc->v[2][0] = no.x;  lc->v[2][1] = no.y;  lc->v[2][2] = static_cast<double>(no.z);
    }
    if (validOut) *validOut = no.valid;  // (one wavefront per pose: the flag is uniform, the vote stays in registers)
    if (g.sub == 0) {
        if constexpr (!kDirect) {
            if (!validOut) sh.valid[leg] = no.valid;
            if (!recs) {  // (with staged records the commit reads the next positions from the record itself)
                sh.nxt[0][leg][0] = nx0;   sh.nxt[0][leg][1] = ny;    sh.nxt[0][leg][2] = static_cast<double>(zDefault);
                sh.nxt[1][leg][0] = co.x;  sh.nxt[1][leg][1] = co.y;  sh.nxt[1][leg][2] = static_cast<double>(co.z);
                sh.nxt[2][leg][0] = no.x;  sh.nxt[2][leg][1] = no.y;  sh.nxt[2][leg][2] = static_cast<double>(no.z);
            }
        }
        stamp(pc, cyc, 13);
        if (live && recs) {  // staged: flush_seqrec writes the records of a few cycles at a time
            SeqRecOf<KW> r;
            r.nomX = no.x; r.nomY = no.y; r.cenX = co.x; r.cenY = co.y; r.defX = nx0; r.defY = ny;
            r.nomZ = no.z; r.cenZ = co.z; r.defZ = zDefault;
            r.nomRow = no.row; r.nomCol = no.col; r.cenRow = co.row; r.cenCol = co.col;
            r.flags = static_cast<uint32_t>(no.valid) | (static_cast<uint32_t>(no.source) << 8) | (static_cast<uint32_t>(co.code) << 16) | deferFlags;
            if constexpr (kDeferH) {
                r.aI0 = aI0; r.aJ0 = aJ0; r.aNj = aNj;
                r.bI0 = bI0; r.bJ0 = bJ0; r.bNj = bNj;
                r.pad[0] = r.pad[1] = 0u;
                r.visA[0] = visA0; r.visA[1] = visA1;
                r.visB[0] = visB0; r.visB[1] = visB1;
            }
            recs[leg] = r;
        } else if (live) {
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

// ---- 8-lane kernels: the y side of a leg's geometry, hoisted out of the chain ---------------------------------------
// A leg's search centre and boxes have y = (initialPose_[1] + ajustedPose_[1]) + defaultBias.y (cpp:2201, 2411-2418):
// it depends on the gait cycle only, never on earlier results.  Everything derived from it — the column indices of
// the foot-disc box, of the centroid rectangle and of getIndex(centre), the y part of getSubmap's geometry, the
// rectangle polygon's column interval, the squared y distances of the 3x3 disc's columns — is computed for eight
// cycles at a time, one (leg, cycle) entry per lane, with the exact functions; the chain then evaluates x only.

// ---- 3x3-only 8-lane kernels: results and heights leave the chain ----------------------------------------------------
// Nothing a later gait cycle reads depends on a mean height (getPolygonCenter uses x and y only, cpp:2421-2463), and
// the output records are write-only.  The chain therefore only DEPOSITS, per (leg, cycle), the elevations its disc
// loads returned and the few words that identify the results; every eighth cycle the 32 lanes of a pose each take one
// (leg, cycle) unit, run its three ordered height sums (cpp:2520-2554) serially and write its four output records —
// one instruction stream for 32 units instead of one per leg and cycle.
struct Unit {
    float eA[9];  // centre disc (checkFoothold's centre, cpp:2029): elevations in CircleIterator order, [4] = middle cell
    float eB[9];  // default-track disc (cpp:2289-2301)
    float eC;     // centroid result's own cell (one-cell foot disc)
    uint32_t pad0;  // (the eight words below start on a 16-byte boundary: lane 0 deposits them with two 16-byte LDS stores, the three
                    // positions with one 16-byte and one 8-byte store — eleven separate stores before)
    uint32_t visA, visB;  // bit k: cell k visited; bit 31: the height was computed in the chain (direct pass) and is in e[0]
    int nomRow, nomCol;
    uint32_t nomFlags;    // valid | source << 8
    int cenRow, cenCol;
    uint32_t cenCode;     // code | 0x100: the result has a one-cell disc whose elevation is in eC | 0x200: ... to be read by flush_unit
    double cx;    // search centre x (nominal x of a default hit / invalid leg; centroid x of code 0)
    double cenX;  // centroid result x (codes 1-4)
    double defX;  // default track x
    uint32_t pad[2];
};
static_assert(sizeof(Unit) == 144 && sizeof(Unit) % 16 == 0 && offsetof(Unit, visA) == 80 && offsetof(Unit, cx) == 112, "Unit layout");

// Two mean heights side by side — a disc around a known centre (bounding box + membership mask) and, optionally, a
// cell-centred disc (offset table) — with the loads of both in ONE batch per eight cells.  Each sum is the ordered f32 sum
// of getFootholdMeanHeight (cpp:2520-2554): visited cells in CircleIterator (row-major) order, non-finite values count as 0,
// values >= 10 are skipped, finish_mean divides (or falls back on the last visited value).
struct MeanAcc {
    float sum, last;
    int cnt;
};
__device__ __forceinline__ void mean_acc(MeanAcc& a, bool vis, float e) {
    const float v = __builtin_isfinite(e) ? e : 0.0f;  // cpp:2532-2537
    const bool inc = vis && v < 10;                     // cpp:2539
    a.last = vis ? v : a.last;
    a.cnt += inc ? 1 : 0;
    a.sum = a.sum + (inc ? v : -0.0f);  // s + (-0.0f) == s for every s
}
#ifndef FPE_FLUSH_NA
#define FPE_FLUSH_NA 8
#endif
#ifndef FPE_FLUSH_NC
#define FPE_FLUSH_NC 8
#endif
// NA box cells and NC table entries per batch (one dependent round trip per batch)
template <int NA = FPE_FLUSH_NA, int NC = FPE_FLUSH_NC>
__device__ __forceinline__ void seq_mean2(const float* __restrict__ elev, int rows, int cols, int i0, int j0, int nj, unsigned long long v0,
                                          unsigned long long v1, bool wantC, int cRow, int cCol, const int8_t* da, const int8_t* db, int nFoot,
                                          double h, float& zBox, float& zC) {
    static_assert(NC == 8, "the offset table is read eight entries (two 64-bit LDS words) at a time");
    MeanAcc A{0.0f, 0.0f, 0}, C{0.0f, 0.0f, 0};
    const int nC = wantC ? nFoot : 0;
    const int nA = v1 ? 128 - __builtin_clzll(v1) : (v0 ? 64 - __builtin_clzll(v0) : 0);  // one past the last visited cell
    // the box is walked row-major (cell t = a * nj + b): column counter and cell offset advance together, no division
    int qcol = 0;
    unsigned cell = __umul24(static_cast<unsigned>(i0), static_cast<unsigned>(cols)) + static_cast<unsigned>(j0);
    const unsigned rowStep = static_cast<unsigned>(cols - nj + 1);
    for (int t0 = 0, c0 = 0; t0 < nA || c0 < nC; t0 += NA, c0 += NC) {
        // bits t0 .. t0 + NA - 1 of the 128-bit membership mask
        const unsigned long long lo = t0 < 64 ? (v0 >> t0) | (t0 ? v1 << (64 - t0) : 0ull) : (t0 < 128 ? v1 >> (t0 - 64) : 0ull);
        const unsigned ba = static_cast<unsigned>(lo) & ((1u << NA) - 1u);
        unsigned bc = 0u;
        float eA[NA], eC[NC];
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            eA[u] = load_cell(elev, ((ba >> u) & 1u) ? cell : 0u);
            const bool wrap = ++qcol == nj;
            qcol = wrap ? 0 : qcol;
            cell += wrap ? rowStep : 1u;
        }
        // eight table entries as two 64-bit words each (the arrays are 16-byte aligned and hold kMaxFootOffsets entries:
        // c0 is a multiple of 8 below nFoot, or 0; entries past nFoot are masked)
        unsigned long long daW, dbW;
        __builtin_memcpy(&daW, da + (c0 < nC ? c0 : 0), 8);
        __builtin_memcpy(&dbW, db + (c0 < nC ? c0 : 0), 8);
#pragma unroll
        for (int u = 0; u < NC; ++u) {
            const int qi = cRow + static_cast<int8_t>((daW >> (8 * u)) & 0xFFull), qj = cCol + static_cast<int8_t>((dbW >> (8 * u)) & 0xFFull);
            const bool visC = c0 + u < nC && in_range(qi, qj, rows, cols);
            bc |= visC ? (1u << u) : 0u;
            const unsigned cellC = visC ? __umul24(static_cast<unsigned>(qi), static_cast<unsigned>(cols)) + static_cast<unsigned>(qj) : 0u;
            eC[u] = load_cell(elev, cellC);
        }
#pragma unroll
        for (int u = 0; u < NA; ++u) mean_acc(A, ((ba >> u) & 1u) != 0u, eA[u]);
#pragma unroll
        for (int u = 0; u < NC; ++u) mean_acc(C, ((bc >> u) & 1u) != 0u, eC[u]);
    }
    zBox = finish_mean(A.sum, A.last, A.cnt, h);
    zC = finish_mean(C.sum, C.last, C.cnt, h);
}

// The same for a box of up to 32 cells (the generic 8-lane kernels' units), walking the VISITED cells only: a foot disc of
// radius two cells has 13 members in a box of 25 — two batches of eight loads instead of four, i.e. two dependent memory
// round trips less per unit (the register-capped kernel cannot keep more than eight loads in flight: batches of 16 / 25 / 32
// spill and lose, measured).  The members are taken in ascending cell order (lowest set bit first): CircleIterator order.
template <int NA = 8, int NC = 8>
__device__ __forceinline__ void seq_mean2_visited(const float* __restrict__ elev, int rows, int cols, int i0, int j0, int nj, uint32_t vis, bool wantC,
                                                  int cRow, int cCol, const int8_t* da, const int8_t* db, int nFoot, double h, float& zBox, float& zC) {
    static_assert(NC == 8, "the offset table is read eight entries (two 64-bit LDS words) at a time");
    MeanAcc A{0.0f, 0.0f, 0}, C{0.0f, 0.0f, 0};
    const int nC = wantC ? nFoot : 0;
    uint32_t rem = vis;
    const unsigned base = __umul24(static_cast<unsigned>(i0), static_cast<unsigned>(cols)) + static_cast<unsigned>(j0);
    // t / nj for t < 32, 1 <= nj <= 32: floor(t * inv / 2^16) with inv = floor(2^16 / nj) + 1 (error below t / 2^16 < 1 / nj)
    const unsigned inv = static_cast<unsigned>(65536.0f * __builtin_amdgcn_rcpf(static_cast<float>(nj))) + 1u;
    for (int c0 = 0; rem != 0u || c0 < nC; c0 += NC) {
        unsigned ba = 0u, bc = 0u;
        float eA[NA], eC[NC];
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const bool v = rem != 0u;
            const unsigned t = v ? static_cast<unsigned>(__builtin_ctz(rem)) : 0u;
            rem &= rem - 1u;  // (0 stays 0)
            const unsigned a = (t * inv) >> 16, b = t - a * static_cast<unsigned>(nj);
            eA[u] = load_cell(elev, v ? base + __umul24(a, static_cast<unsigned>(cols)) + b : 0u);
            ba |= v ? (1u << u) : 0u;
        }
        unsigned long long daW, dbW;
        __builtin_memcpy(&daW, da + (c0 < nC ? c0 : 0), 8);
        __builtin_memcpy(&dbW, db + (c0 < nC ? c0 : 0), 8);
#pragma unroll
        for (int u = 0; u < NC; ++u) {
            const int qi = cRow + static_cast<int8_t>((daW >> (8 * u)) & 0xFFull), qj = cCol + static_cast<int8_t>((dbW >> (8 * u)) & 0xFFull);
            const bool visC = c0 + u < nC && in_range(qi, qj, rows, cols);
            bc |= visC ? (1u << u) : 0u;
            const unsigned cellC = visC ? __umul24(static_cast<unsigned>(qi), static_cast<unsigned>(cols)) + static_cast<unsigned>(qj) : 0u;
            eC[u] = load_cell(elev, cellC);
        }
#pragma unroll
        for (int u = 0; u < NA; ++u) mean_acc(A, ((ba >> u) & 1u) != 0u, eA[u]);
#pragma unroll
        for (int u = 0; u < NC; ++u) mean_acc(C, ((bc >> u) & 1u) != 0u, eC[u]);
    }
    zBox = finish_mean(A.sum, A.last, A.cnt, h);
    zC = finish_mean(C.sum, C.last, C.cnt, h);
}

template <class Rec>
__device__ __forceinline__ void flush_seqrec(const DevMap& m, const PlanConsts& pc, const int8_t* footDa, const int8_t* footDb, const Rec& rLds,
                                             int b, int cyc, int leg, int nCycles, const fpe_plan_out& out) {
    Rec r;
    __builtin_memcpy(&r, &rLds, sizeof(Rec));
    const size_t o = (static_cast<size_t>(b) * nCycles + cyc) * 4 + leg;
    const uint8_t valid = static_cast<uint8_t>(r.flags & 0xFFu), source = static_cast<uint8_t>((r.flags >> 8) & 0xFFu);
    // (records whose heights are deferred go through flush_seqrec2; the ones that arrive here — -DFPE_SEQ_DEFER_KW=2 builds of
    // the 96-bit-row kernel — carry final values)
    const float zN = r.nomZ, zB = r.defZ, zC = r.cenZ;
    (void)footDa;
    (void)footDb;
    (void)m;
    (void)pc;
    if (out.nominal) {
        fpe_foothold f;
        f.row = r.nomRow; f.col = r.nomCol; f.x = r.nomX; f.y = r.nomY; f.z = zN;
        f.valid = valid; f.source = source;
        f.foot_id = static_cast<uint8_t>(leg); f.gait_cycle_id = static_cast<uint8_t>(cyc);
        store_record<true>(out.nominal + o, f);
    }
    store_selected<true>(out, o, r.nomRow, r.nomCol, zN, valid, source, leg, cyc);
    if (out.centroid) {
        fpe_centroid_foothold cf;
        cf.x = r.cenX; cf.y = r.cenY; cf.z = zC; cf.row = r.cenRow; cf.col = r.cenCol;
        cf.code = static_cast<uint8_t>((r.flags >> 16) & 0xFFu); cf.pad[0] = cf.pad[1] = cf.pad[2] = 0;
        store_record<true>(out.centroid + o, cf);
    }
    if (out.default_next) {
        store_record<true>(out.default_next + o * 3 + 0, r.defX);
        store_record<true>(out.default_next + o * 3 + 1, r.defY);
        store_record<true>(out.default_next + o * 3 + 2, static_cast<double>(zB));
    }
}

// The same with two lanes per unit (64-bit-row kernels, deferred heights): lane half 0 takes the centre disc and the
// centroid result's disc and writes the nominal / selected / centroid records, half 1 the default-track disc and the
// default_next record.  One instruction stream for both halves (the arguments differ per lane, not the code).
__device__ __forceinline__ void flush_seqrec2(const DevMap& m, const PlanConsts& pc, const int8_t* footDa, const int8_t* footDb, const SeqRec& rLds,
                                              int b, int cyc, int leg, int half, int nCycles, const fpe_plan_out& out) {
    SeqRec r;
    __builtin_memcpy(&r, &rLds, sizeof(SeqRec));
    const size_t o = (static_cast<size_t>(b) * nCycles + cyc) * 4 + leg;
    const bool h1 = half != 0;
    const bool defer = (r.flags & (h1 ? kSeqDeferB : kSeqDeferA)) != 0u && (h1 ? out.default_next != nullptr : true);
    const bool wantC = !h1 && (r.flags & kSeqDeferC) != 0u && out.centroid != nullptr;
    float sBox, sC;
#ifndef FPE_FLUSH_NA_SEQ
#define FPE_FLUSH_NA_SEQ 12  // box cells per batch here (measured: 8 -> 12: cfg-3 -1.3 %, cfg-5 -1.5 %; 13, 14 the same; 16 worse on cfg-3)
#endif
    seq_mean2<FPE_FLUSH_NA_SEQ, 8>(m.elev, m.g.rows, m.g.cols, h1 ? r.bI0 : r.aI0, h1 ? r.bJ0 : r.aJ0, max(h1 ? r.bNj : r.aNj, 1), defer ? (h1 ? r.visB[0] : r.visA[0]) : 0ull,
              defer ? (h1 ? r.visB[1] : r.visA[1]) : 0ull, wantC, r.cenRow, r.cenCol, footDa, footDb, pc.nFoot, pc.h, sBox, sC);
    if (h1) {
        if (out.default_next) {
            store_record<true>(out.default_next + o * 3 + 0, r.defX);
            store_record<true>(out.default_next + o * 3 + 1, r.defY);
            store_record<true>(out.default_next + o * 3 + 2, static_cast<double>(defer ? sBox : r.defZ));
        }
        return;
    }
    const uint8_t valid = static_cast<uint8_t>(r.flags & 0xFFu), source = static_cast<uint8_t>((r.flags >> 8) & 0xFFu);
    const float zA = defer ? sBox : r.nomZ;
    const float zC = wantC ? sC : ((r.flags & kSeqCIsA) ? zA : r.cenZ);
    const float zN = defer ? (valid ? zA : 0.0f) : r.nomZ;  // z at the DEFAULT centre, for a spiral candidate too (cpp:2029)
    if (out.nominal) {
        fpe_foothold f;
        f.row = r.nomRow; f.col = r.nomCol; f.x = r.nomX; f.y = r.nomY; f.z = zN;
        f.valid = valid; f.source = source;
        f.foot_id = static_cast<uint8_t>(leg); f.gait_cycle_id = static_cast<uint8_t>(cyc);
        store_record<true>(out.nominal + o, f);
    }
    store_selected<true>(out, o, r.nomRow, r.nomCol, zN, valid, source, leg, cyc);
    if (out.centroid) {
        fpe_centroid_foothold cf;
        cf.x = r.cenX; cf.y = r.cenY; cf.z = zC; cf.row = r.cenRow; cf.col = r.cenCol;
        cf.code = static_cast<uint8_t>((r.flags >> 16) & 0xFFu); cf.pad[0] = cf.pad[1] = cf.pad[2] = 0;
        store_record<true>(out.centroid + o, cf);
    }
}

// 3x3 form: lane s holds cell s + (s >= 4) of the box, every lane the middle cell (disc_issue); else the direct pass.
template <bool kWant>
__device__ __forceinline__ void unit_put_disc(const DevMap& m, const PlanConsts& pc, double cx, double cy, const BBox& bb,
                                              const Grp<8>& g, const DiscLoads& d, float* e, uint32_t& vis, float* scratch) {
    if (!kWant) return;
    if (d.pipelined) {  // wave-uniform: the 3x3 form
        e[g.sub + (g.sub >= 4 ? 1 : 0)] = d.e[0];
        if (g.sub == 0) e[4] = d.eMid;
        const unsigned mk = static_cast<unsigned>(g.ballot(d.vis[0] != 0));
        vis = (mk & 0xFu) | 0x10u | ((mk & 0xF0u) << 1);
    } else {
        bool unused;
        const float z = disc_pass_direct<8, false>(m, pc, cx, cy, bb, g, unused, scratch);
        if (g.sub == 0) e[0] = z;
        vis = 0x80000000u;
    }
}
// getFootholdMeanHeight (cpp:2520-2554) over up to nine deposited cells, in order
__device__ __forceinline__ float unit_mean9(const float* e, uint32_t vis, double h) {
    if (vis & 0x80000000u) return e[0];
    float sum = 0.0f, last = 0.0f;
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) {  // branch-free: an unvisited cell adds -0.0f (s + (-0.0f) == s for every s) and leaves `last`
        const bool visited = ((vis >> k) & 1u) != 0u;
        const float v = __builtin_isfinite(e[k]) ? e[k] : 0.0f;  // cpp:2532-2537
        const bool inc = visited && v < 10;                      // cpp:2539
        last = visited ? v : last;
        cnt += inc ? 1 : 0;
        sum = sum + (inc ? v : -0.0f);
    }
    return finish_mean(sum, last, cnt, h);
}
// One (leg, cycle) unit per lane: heights and the four output records of that unit.
__device__ __forceinline__ void flush_unit(const DevMap& m, const PlanConsts& pc, const Unit& uLds, const YEntry& yeLds, int b, int cyc,
                                           int leg, int nCycles, uint32_t okBits, const fpe_plan_out& out) {
    const MapGeom& mg = m.g;
    // the unit and its y entry in registers by one batch of 16-byte LDS reads (read field by field the reads are
    // interleaved with their uses: a dozen serial round trips)
    Unit u;
    YEntry ye;
    __builtin_memcpy(&u, &uLds, sizeof(Unit));
    __builtin_memcpy(&ye, &yeLds, sizeof(YEntry));
    // the centroid result's own cell, when the chain left its elevation to be read here (issued first: the three
    // height sums below cover the round trip)
    float eC = u.eC;
    if (out.centroid && (u.cenCode & 0x200u)) eC = m.elev[static_cast<size_t>(u.cenRow) * mg.cols + u.cenCol];
    if (leg == 0 && out.cycle_ok) out.cycle_ok[static_cast<size_t>(b) * nCycles + cyc] = static_cast<uint8_t>((okBits >> (cyc & 7)) & 1u);
    const float zA = unit_mean9(u.eA, u.visA, pc.h);
    const float zB = out.default_next ? unit_mean9(u.eB, u.visB, pc.h) : 0.0f;
    const int code = static_cast<int>(u.cenCode & 0xFFu);
    float zC = 0.0f;
    if (u.cenCode & 0x300u) {
        const float v = __builtin_isfinite(eC) ? eC : 0.0f;
        const bool inc = v < 10;
        zC = finish_mean(inc ? 0.0f + v : 0.0f, v, inc ? 1 : 0, pc.h);
    } else if (code == 0) {
        zC = zA;  // whole region valid: the height at the centre (cpp:1687)
    }
    const size_t o = (static_cast<size_t>(b) * nCycles + cyc) * 4 + leg;
    const int valid = static_cast<int>(u.nomFlags & 0xFFu), source = static_cast<int>((u.nomFlags >> 8) & 0xFFu);
    const float zN = valid ? zA : 0.0f;  // z at the DEFAULT centre, for a spiral candidate too (cpp:2029)
    if (out.nominal) {
        fpe_foothold f;
        f.row = u.nomRow;
        f.col = u.nomCol;
        f.x = source == 1 ? cell_pos(mg.baseX, mg.res, u.nomRow) : u.cx;  // cpp:2105-2107 / cpp:2016-2017
        f.y = source == 1 ? cell_pos(mg.baseY, mg.res, u.nomCol) : ye.ny;
        f.z = zN;
        f.valid = static_cast<uint8_t>(valid);
        f.source = static_cast<uint8_t>(source);
        f.foot_id = static_cast<uint8_t>(leg);
        f.gait_cycle_id = static_cast<uint8_t>(cyc);
        store_record<true>(out.nominal + o, f);
    }
    store_selected<true>(out, o, u.nomRow, u.nomCol, zN, valid, source, leg, cyc);
    if (out.centroid) {
        fpe_centroid_foothold cf;
        cf.x = code == 0 ? u.cx : (code <= 4 ? u.cenX : 0.0);
        cf.y = code == 0 ? ye.ny : (code == 1 ? ye.yA : (code <= 4 ? ye.yB : 0.0));
        cf.z = zC; cf.row = u.cenRow; cf.col = u.cenCol;
        cf.code = static_cast<uint8_t>(code); cf.pad[0] = cf.pad[1] = cf.pad[2] = 0;
        store_record<true>(out.centroid + o, cf);
    }
    if (out.default_next) {
        store_record<true>(out.default_next + o * 3 + 0, u.defX);
        store_record<true>(out.default_next + o * 3 + 1, ye.ny);
        store_record<true>(out.default_next + o * 3 + 2, static_cast<double>(zB));
    }
}

// ---- generic 8-lane kernels (boxes of up to 32 cells, foot-disc tables): the same deferral --------------------------
// The chain deposits, per (leg, cycle), the MEMBERSHIP of the two discs around known centres (a 32-bit mask over the
// bounding box's cells in CircleIterator order, with the box's origin) and the words that identify the results; it
// issues no elevation load at all.  Every fourth cycle (the LDS of twelve workgroups per CU holds four cycles of units
// and y entries, not eight) lane (leg, s < 4) of a pose takes the unit of cycle base + s, reads the elevations itself
// (seq_mean2: two groups of eight independent loads per batch) and runs the ordered sums (cpp:2520-2554).
constexpr uint32_t kUgValid = 1u << 8, kUgSrcShift = 9, kUgPreA = 1u << 12, kUgPreB = 1u << 13, kUgCTable = 1u << 14, kUgCIsA = 1u << 15;
struct UnitG {
    double cx;    // search centre x (nominal x of a default hit / invalid leg; centroid x of code 0)
    double cenX;  // centroid result x (codes 1-4)
    double defX;  // default track x
    int aI0, aJ0;
    uint32_t visA;  // centre disc (cpp:2029): bit t = cell t of the box visited; kUgPreA: the f32 height itself (direct pass)
    int bI0, bJ0;
    uint32_t visB;  // default-track disc (cpp:2289-2301), kUgPreB likewise
    int nomRow, nomCol, cenRow, cenCol;
    uint32_t flags;  // centroid code | kUgValid | source << 9 | kUg* | aNj << 16 | bNj << 24
    uint32_t pad[3];
};
static_assert(sizeof(UnitG) == 80 && sizeof(UnitG) % 16 == 0, "UnitG layout");

// Membership of a leg's two foot discs — the centre disc around (cxA, cy) and the default-track disc around (cxB, cy):
// same columns, the y side is shared — with lane = BOX ROW: bit q of the lane's word = cell (i0 + sub, j0 + q) is visited
// (inside the box, the map and the disc; CircleIterator::isInside, the expression of cell_in_disc).  Boxes of up to 8 x 8
// cells; one pass over the columns instead of four rounds of eight cells per disc with a division each.
__device__ __forceinline__ void disc_rows8(const MapGeom& mg, double rf2, double cxA, double cxB, double cy, const BBox& ba, const BBox& bbx,
                                           const Grp<8>& g, uint32_t& rowA, uint32_t& rowB) {
    const int iA = ba.i0 + g.sub, iB = bbx.i0 + g.sub;
    const double dxA = cell_pos(mg.baseX, mg.res, iA) - cxA, dxB = cell_pos(mg.baseX, mg.res, iB) - cxB;
    const double dxA2 = dxA * dxA, dxB2 = dxB * dxB;
    const int j0 = ba.j0, nj = ba.nj;  // (both boxes: YEntry::j0d / njd)
    uint32_t a = 0u, b = 0u;
    for (int q = 0; __ballot(q < nj) != 0ull; ++q) {  // wave-uniform trip count
        const double dy = cell_pos(mg.baseY, mg.res, j0 + q) - cy;
        const double dy2 = dy * dy;
        a |= ((dxA2 + dy2) <= rf2) ? (1u << q) : 0u;
        b |= ((dxB2 + dy2) <= rf2) ? (1u << q) : 0u;
    }
    const int lo = max(0, -j0), hi = min(nj - 1, mg.cols - 1 - j0);  // columns inside the box and the map
    const uint32_t colMask = hi >= lo ? ((2u << hi) - (1u << lo)) : 0u;
    rowA = (g.sub < ba.ni && static_cast<unsigned>(iA) < static_cast<unsigned>(mg.rows)) ? (a & colMask) : 0u;
    rowB = (g.sub < bbx.ni && static_cast<unsigned>(iB) < static_cast<unsigned>(mg.rows)) ? (b & colMask) : 0u;
}
// OR over the eight lanes of a group (DPP: two quad permutations and the half-row mirror)
__device__ __forceinline__ uint32_t or_reduce8(uint32_t v) {
    int x = static_cast<int>(v);
    x |= __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
    x |= __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
    x |= __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, true);  // row_half_mirror
    return static_cast<uint32_t>(x);
}
// The box's 32-bit membership mask in CircleIterator order (cell t = a * nj + b) from the row words
__device__ __forceinline__ uint32_t box_mask_from_rows8(uint32_t row, int nj, const Grp<8>& g) {
    return or_reduce8(row << min(g.sub * nj, 31));  // (rows beyond the box hold 0)
}
// checkFoothold's default test (cpp:2012) on the row words: no visited cell of the centre disc has its Df bit set
__device__ __forceinline__ bool default_ok_rows8(uint32_t rowA, const BBox& bb, const uint32_t* rowsDf, int nRows, int iw0, int jw0, const Grp<8>& g) {
    const int ri = bb.i0 - iw0 + g.sub, cj0 = bb.j0 - jw0;
    const uint32_t df = rowsDf[min(max(ri, 0), nRows - 1)];
    // bit q of `sh` = window column cj0 + q of the row (columns outside the 32-bit window: 0, as win_bit)
    const uint32_t sh = (cj0 >= 32 || cj0 <= -32) ? 0u : (cj0 >= 0 ? df >> cj0 : df << -cj0);
    const bool fail = static_cast<unsigned>(ri) < static_cast<unsigned>(nRows) && (rowA & sh) != 0u;
    return g.any(rowA != 0u) && !g.any(fail);
}
__device__ __forceinline__ void unitg_put_disc(const DevMap& m, const PlanConsts& pc, double cx, double cy, const BBox& bb, const Grp<8>& g,
                                               const DiscLoads& d, uint32_t& vis, bool& pre, float* scratch) {
    if (d.pipelined) {  // wave-uniform
        if (d.mid) {    // 3x3 form: lane s holds cell s + (s >= 4), the middle cell is always visited
            const unsigned mk = static_cast<unsigned>(g.ballot(d.vis[0] != 0));
            vis = (mk & 0xFu) | 0x10u | ((mk & 0xF0u) << 1);
        } else {
            vis = 0u;
#pragma unroll
            for (int r = 0; r < kDiscRounds; ++r) vis |= (static_cast<uint32_t>(g.ballot(d.vis[r] != 0)) & 0xFFu) << (8 * r);
        }
        pre = false;
    } else {
        bool unused;
        vis = __float_as_uint(disc_pass_direct<8, false>(m, pc, cx, cy, bb, g, unused, scratch));
        pre = true;
    }
}
// Two lanes per (leg, cycle) unit: lane half 0 takes the centre disc and the centroid result's disc and writes the
// nominal / selected / centroid records, half 1 the default-track disc, the default_next record and the cycle's
// validity.  One instruction stream for both (the arguments differ per lane, not the code).
__device__ __forceinline__ void flush_unit_g(const DevMap& m, const PlanConsts& pc, const int8_t* footDa, const int8_t* footDb,
                                             const UnitG& uLds, const YEntry& yeLds, int b, int cyc, int leg, int half, int nCycles,
                                             uint32_t okBits, const fpe_plan_out& out) {
    const MapGeom& mg = m.g;
    UnitG u;
    __builtin_memcpy(&u, &uLds, sizeof(UnitG));
    const double ny = yeLds.ny, yA = yeLds.yA, yB = yeLds.yB;
    const bool h1 = half != 0;
    const bool pre = (u.flags & (h1 ? kUgPreB : kUgPreA)) != 0u;
    const uint32_t visW = h1 ? u.visB : u.visA;
    const bool wantBox = !pre && (h1 ? out.default_next != nullptr : true);
    const bool wantC = !h1 && (u.flags & kUgCTable) != 0u && out.centroid != nullptr;
    const int nj = max(static_cast<int>(h1 ? (u.flags >> 24) : ((u.flags >> 16) & 0xFFu)), 1);
    float sBox, sC;
#ifdef FPE_FLUSH_BOX_ORDER
    seq_mean2(m.elev, mg.rows, mg.cols, h1 ? u.bI0 : u.aI0, h1 ? u.bJ0 : u.aJ0, nj, wantBox ? static_cast<unsigned long long>(visW) : 0ull, 0ull,
              wantC, u.cenRow, u.cenCol, footDa, footDb, pc.nFoot, pc.h, sBox, sC);
#else
    seq_mean2_visited(m.elev, mg.rows, mg.cols, h1 ? u.bI0 : u.aI0, h1 ? u.bJ0 : u.aJ0, nj, wantBox ? visW : 0u, wantC, u.cenRow, u.cenCol, footDa,
                      footDb, pc.nFoot, pc.h, sBox, sC);
#endif
    const float zBox = pre ? __uint_as_float(visW) : sBox;
    const size_t o = (static_cast<size_t>(b) * nCycles + cyc) * 4 + leg;
    if (h1) {
        if (leg == 0 && out.cycle_ok) out.cycle_ok[static_cast<size_t>(b) * nCycles + cyc] = static_cast<uint8_t>((okBits >> (cyc & 7)) & 1u);
        if (out.default_next) {
            store_record<true>(out.default_next + o * 3 + 0, u.defX);
            store_record<true>(out.default_next + o * 3 + 1, ny);
            store_record<true>(out.default_next + o * 3 + 2, static_cast<double>(zBox));
        }
        return;
    }
    const float zA = zBox;
    const float zC = wantC ? sC : ((u.flags & kUgCIsA) ? zA : 0.0f);  // code 0, whole region valid: the height at the centre (cpp:1687)
    const int code = static_cast<int>(u.flags & 0xFFu);
    const int valid = (u.flags & kUgValid) ? 1 : 0, source = static_cast<int>((u.flags >> kUgSrcShift) & 3u);
    const float zN = valid ? zA : 0.0f;  // z at the DEFAULT centre, for a spiral candidate too (cpp:2029)
    if (out.nominal) {
        fpe_foothold f;
        f.row = u.nomRow;
        f.col = u.nomCol;
        f.x = source == 1 ? cell_pos(mg.baseX, mg.res, u.nomRow) : u.cx;  // cpp:2105-2107 / cpp:2016-2017
        f.y = source == 1 ? cell_pos(mg.baseY, mg.res, u.nomCol) : ny;
        f.z = zN;
        f.valid = static_cast<uint8_t>(valid);
        f.source = static_cast<uint8_t>(source);
        f.foot_id = static_cast<uint8_t>(leg);
        f.gait_cycle_id = static_cast<uint8_t>(cyc);
        store_record<true>(out.nominal + o, f);
    }
    store_selected<true>(out, o, u.nomRow, u.nomCol, zN, valid, source, leg, cyc);
    if (out.centroid) {
        fpe_centroid_foothold cf;
        cf.x = code == 0 ? u.cx : (code <= 4 ? u.cenX : 0.0);
        cf.y = code == 0 ? ny : (code == 1 ? yA : (code <= 4 ? yB : 0.0));
        cf.z = zC; cf.row = u.cenRow; cf.col = u.cenCol;
        cf.code = static_cast<uint8_t>(code); cf.pad[0] = cf.pad[1] = cf.pad[2] = 0;
        store_record<true>(out.centroid + o, cf);
    }
}

__device__ __forceinline__ void fill_yentry(const MapGeom& mg, const PlanConsts& pc, const LegStatic& ls, double ny, YEntry& e) {
    const double ly = ls.lk.ly;  // centroid rectangle width (cpp:1617)
    const double r = static_cast<double>(ls.Rf);
    int flags = fabs(ny) <= 1e6 ? 2 : 0;
    e.ny = ny;
    // foot-disc box (CircleIterator::findSubmapParameters, y axis), getIndex(centre), centroid rectangle
    // (getSubmapInformation, y axis: corners centre +- 0.5 * ly).
    // Predicted, as in the x pass of the chain (PlanConsts::cornerEps): a corner strictly inside the map whose quotient
    // is farther than cornerEps from an integer has the index -trunc(quotient) and stays within the map, whatever
    // boundPositionToRange's rewrite and the index division do to the last bits.  When any lane of the wavefront is
    // too close to a cell boundary or to the map's edge, the wavefront evaluates the reference's own expressions (a
    // real branch: the if-converted form would pay five divisions per entry).
    const double xs0[5] = {ny + pc.rf, ny - pc.rf, ny, ny + 0.5 * ly, ny - 0.5 * ly};
    int idx[5];
    bool safe = true;
    const double colsD = static_cast<double>(mg.cols);
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const double qf = ((xs0[k] - mg.orgY) - mg.posY) * mg.rinv;
        const double kk = trunc(qf);
        const double fr = fabs(qf - kk);
        safe = safe & (fr > pc.cornerEps) & (fr < 1.0 - pc.cornerEps);
        if (k != 2) safe = safe & (qf < -pc.cornerEps) & (qf > pc.cornerEps - colsD);
        idx[k] = -static_cast<int>(kk);
    }
    bool cornersWithin = true;  // checkIfPositionWithinMap of the centroid rectangle's bounded corners (y axis)
    if (__ballot(!safe) != 0ull) {
        const double tly = bound_axis(xs0[0], mg.orgY, mg.posY, mg.lenY);
        const double bry = bound_axis(xs0[1], mg.orgY, mg.posY, mg.lenY);
        const double tlr = bound_axis(xs0[3], mg.orgY, mg.posY, mg.lenY);
        const double brr = bound_axis(xs0[4], mg.orgY, mg.posY, mg.lenY);
        const double xs[5] = {tly, bry, ny, tlr, brr};
#pragma unroll
        for (int k = 0; k < 5; ++k) idx[k] = index_of(xs[k], mg.orgY, mg.posY, mg.res);
        cornersWithin = within_axis(tlr, mg.orgY, mg.posY, mg.lenY) && within_axis(brr, mg.orgY, mg.posY, mg.lenY);
    }
    e.j0d = idx[0];
    e.njd = idx[1] - idx[0] + 1;
    e.jc = idx[2];
    const int j0r = idx[3];
    const int j1r = idx[4];
    e.j0r = j0r;
    e.njr = j1r - j0r + 1;
    bool okY = cornersWithin && j0r >= 0 && j0r < mg.cols && j1r < mg.cols;  // top-left in range, region fits the buffer (getSubmap)
    const double cornerY = cell_pos(mg.baseY, mg.res, j0r) - (-(0.5 * mg.res));
    const double subLenY = static_cast<double>(e.njr) * mg.res;
    const double subOrgY = 0.5 * subLenY;
    const double subPosY = cornerY - subOrgY;
    okY = okY && within_axis(ny, subOrgY, subPosY, subLenY);
    e.sbaseY = subPosY + (subOrgY - 0.5 * mg.res);
    const int rightCol = e.njr - 1;
    e.yA = cell_pos(e.sbaseY, mg.res, (rightCol + 1) >> 1);
    e.yB = cell_pos(e.sbaseY, mg.res, rightCol >> 1);
    if (okY) flags |= 1;
    e.flags = flags;
    // reference rectangle polygon (getSearchPolygon, cpp:2496-2517): y limits centre -+ 0.5 * r
    {
        const double yhi = ny + 0.5 * r, ylo = ny - 0.5 * r;
        double qh = floor((mg.baseY - yhi) * mg.rinv), ql = floor((mg.baseY - ylo) * mg.rinv);
        qh = fmin(fmax(qh, -1.0e9), 1.0e9);
        ql = fmin(fmax(ql, -1.0e9), 1.0e9);
        const int eh = static_cast<int>(qh), el = static_cast<int>(ql);
        const bool p0 = cell_pos(mg.baseY, mg.res, eh) < yhi, p1 = cell_pos(mg.baseY, mg.res, eh + 1) < yhi;
        const bool q1 = cell_pos(mg.baseY, mg.res, el + 1) >= ylo, q0 = cell_pos(mg.baseY, mg.res, el) >= ylo;
        e.jA = p0 ? eh : (p1 ? eh + 1 : eh + 2);
        e.jB = q1 ? el + 1 : (q0 ? el : el - 1);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double dy = cell_pos(mg.baseY, mg.res, e.j0d + k) - ny;
        e.dy2[k] = dy * dy;
    }
    const int jw0 = e.jc - pc.winH;
    e.rmask = range_word(e.j0r - jw0, e.j0r - jw0 + e.njr - 1, 0);
    e.pmask = range_word(e.jA - jw0, e.jB - jw0, 0);
}

// One swing leg of one phase, 8 lanes per leg, y side from the YEntry.  The x side is ONE lane-transposed pass: lane
// q evaluates the index of one box corner — 0/1 foot disc (cx +- rf), 2/3 centroid rectangle (cx +- lx / 2),
// 4 getIndex(cx), 5/6 default-track disc (nx0 +- rf) — by prediction (PlanConsts::cornerEps); when any lane of the
// wavefront is within rounding distance of a cell boundary, or outside the map, the wavefront evaluates the
// reference's own expressions (corner_quantity) instead.
template <int NRL, bool kMid>
__device__ __forceinline__ void leg_phase_bits8(const DevMap& m, const BitMap& bm, const PlanConsts& pc, const SpiralLut& lut,
                                                const LutHead& head, PoseShared& sh, const LegBits& lb, const Grp<8>& g, int leg,
                                                const LegStatic& ls, const YEntry& ye, double ctr0, double ctr1, double ctr2,
                                                double advance, int cyc, int nCycles, int b, bool live, const fpe_plan_out& out,
                                                LegCommit* lc, typename std::conditional<kMid, Unit, UnitG>::type* unit) {
    constexpr int G = 8, KW = 1;
    // heights and records are deposited in `unit` and finished by flush_unit (3x3-only kernels, every eighth cycle) /
    // flush_unit_g (generic kernels, every fourth cycle)
    constexpr bool kDefer = true;
    uint32_t ugFlags = 0u, ugVisA = 0u, ugVisB = 0u;  // generic kernels: UnitG fields in the making
    int ugAI0 = 0, ugAJ0 = 0, ugANj = 1, ugBI0 = 0, ugBJ0 = 0, ugBNj = 1;
    const float Rf = ls.Rf;
    const int polyKind = ls.polyKind;
    const LegConst& lk = ls.lk;
    const double biasX = ls.biasX;
    // next default positions of this leg on the three tracks (cpp:2199-2213, 2270-2284)
    const double nx0 = (ctr0 + advance) + biasX;  // cpp:2199, 2414
    const double nx1 = (ctr1 + advance) + biasX;
    const double nx2 = (ctr2 + advance) + biasX;
    const double ny = ye.ny;  // (initialPose_[1] + ajustedPose_[1]) + bias.y, identical on the three tracks (cpp:2201)
    if (polyKind != 0 && g.sub == 0) {  // hexagon vertices from the NOMINAL track's position (build-defined, App. E)
        const double r = static_cast<double>(Rf);
        double* vx = sh.polyX[leg];
        double* vy = sh.polyY[leg];
        const double hx = 0.5 * r;
        const double hy = (0.5 * r) * 0.8660254037844386;
        vx[0] = nx2 + r;   vy[0] = ny;
        vx[1] = nx2 + hx;  vy[1] = ny - hy;
        vx[2] = nx2 - hx;  vy[2] = ny - hy;
        vx[3] = nx2 - r;   vy[3] = ny;
        vx[4] = nx2 - hx;  vy[4] = ny + hy;
        vx[5] = nx2 + hx;  vy[5] = ny + hy;
    }
    LegCtx c;
    c.cyc = cyc;
    c.cx = nx1;  // centre from the CENTROID track (cpp:861-862)
    c.cy = ny;
    c.nv = (polyKind == 0) ? 4 : 6;
    {
        const double r = static_cast<double>(Rf);  // getSearchPolygon's rectangle around the NOMINAL track (cpp:2496-2517)
        c.rect = polyKind == 0;
        c.xhi = nx2 + r;
        c.xlo = nx2 - r;
        c.yhi = ny + 0.5 * r;
        c.ylo = ny - 0.5 * r;
    }
    c.vx = sh.polyX[leg];
    c.vy = sh.polyY[leg];
    c.footDa = sh.footDa;
    c.footDb = sh.footDb;
    c.footOff = sh.footOff;
    c.R2 = lk.R2;
    c.nRings = lk.nRings;
    c.nCand = lk.nCand;
    c.ti0 = c.tj0 = 0;
    c.ici = c.icj = 0;

    NominalOut no;
    CentroidOut co;
    float zDefault = static_cast<float>(static_cast<double>(0.0f) + pc.h);  // value when no cell is visited
    float* scratch = reinterpret_cast<float*>(lb.a);
    const bool wantDefault = out.default_next != nullptr;
    const bool usable = (ye.flags & 2) != 0 && fabs(c.cx) <= 1e6;  // centre_usable(c.cx, c.cy)
    if (!ls.radiusOk || !usable) {
        nominal_invalid(no, c.cx, c.cy, ls.radiusOk ? 2 : 3);
        co.x = co.y = 0.0; co.z = 0.0f; co.row = co.col = -1; co.code = 6;
        if (wantDefault && centre_usable(nx0, ny)) {  // cpp:2289-2301 (leg search skipped: radius / centre unusable)
            const BBox dbox = circle_bbox_fast(m.g, nx0, ny, pc.rf);
            bool unused;
            zDefault = disc_pass_direct<G, false>(m, pc, nx0, ny, dbox, g, unused, scratch);
        }
        if constexpr (kMid) {
            if (g.sub == 0) {
                unit->visA = 0x80000000u;  // the nominal leg is invalid: its height is never used
                unit->eA[0] = 0.0f;
                unit->visB = 0x80000000u;
                unit->eB[0] = zDefault;
                unit->eC = 0.0f;
            }
        } else {
            ugFlags = kUgPreA | kUgPreB;
            ugVisA = __float_as_uint(0.0f);
            ugVisB = __float_as_uint(zDefault);
        }
    } else {
        // ---- x side: one corner quantity per lane ----
        const int q = g.sub;
        const double cq = (q == 5 || q == 6) ? nx0 : c.cx;
        const bool rawq = q == 4 || q == 7;
        const double hq = (q == 2 || q == 3) ? 0.5 * lk.lx : (rawq ? 0.0 : pc.rf);
        const bool minus = q == 1 || q == 3 || q == 6;
        const double xq = rawq ? cq : (minus ? cq - hq : cq + hq);
        int idxq;
        bool withinq = true;
        {
            const double n = (xq - m.g.orgX) - m.g.posX;
            const double qf = n * m.g.rinv;
            const double k = trunc(qf);
            const double fr = fabs(qf - k);
            bool safe = fr > pc.cornerEps && fr < 1.0 - pc.cornerEps;
            // strictly inside the map: boundPositionToRange only rewrites the position (no clamp), within stays true
            if (!rawq) safe = safe && qf < -pc.cornerEps && qf > pc.cornerEps - static_cast<double>(m.g.rows);
            idxq = -static_cast<int>(k);
            if (__ballot(!safe) != 0ull) {  // wave-uniform, rare: the reference's own expressions
                const Box bq{cq, 0.0, hq, 0.0};
                const CornerVal cv = corner_quantity(m.g, minus ? 2 : 0, bq, rawq);
                idxq = cv.idx;
                withinq = cv.within;
            }
        }
        constexpr int kKeep = (~(G - 1)) & 0x1F;
        BBox bb, rbox, dbox;
        bb.i0 = __builtin_amdgcn_ds_swizzle(idxq, kKeep | (0 << 5));
        bb.ni = __builtin_amdgcn_ds_swizzle(idxq, kKeep | (1 << 5)) - bb.i0 + 1;
        rbox.i0 = __builtin_amdgcn_ds_swizzle(idxq, kKeep | (2 << 5));
        rbox.ni = __builtin_amdgcn_ds_swizzle(idxq, kKeep | (3 << 5)) - rbox.i0 + 1;
        c.ici = __builtin_amdgcn_ds_swizzle(idxq, kKeep | (4 << 5));
        dbox.i0 = __builtin_amdgcn_ds_swizzle(idxq, kKeep | (5 << 5));
        dbox.ni = __builtin_amdgcn_ds_swizzle(idxq, kKeep | (6 << 5)) - dbox.i0 + 1;
        bb.j0 = dbox.j0 = ye.j0d;
        bb.nj = dbox.nj = ye.njd;
        rbox.j0 = ye.j0r;
        rbox.nj = ye.njr;
        c.icj = ye.jc;
        const unsigned wbits = static_cast<unsigned>(g.ballot(withinq));
        // getSubmapInformation's tail (submap_from_corners), x part here, y part from the entry
        Submap sm;
        sm.i0 = rbox.i0;
        sm.j0 = rbox.j0;
        sm.ni = rbox.ni;
        sm.nj = rbox.nj;
        {
            const bool okX = (wbits & 0xCu) == 0xCu && sm.i0 >= 0 && sm.i0 < m.g.rows && sm.i0 + sm.ni <= m.g.rows;  // (region fits the buffer)
            const double cornerX = cell_pos(m.g.baseX, m.g.res, sm.i0) - (-(0.5 * m.g.res));
            const double subLenX = static_cast<double>(sm.ni) * m.g.res;
            const double subOrgX = 0.5 * subLenX;
            const double subPosX = cornerX - subOrgX;
            sm.ok = okX && (ye.flags & 1) != 0 && within_axis(c.cx, subOrgX, subPosX, subLenX);
            sm.baseX = sm.ok ? subPosX + (subOrgX - 0.5 * m.g.res) : 0.0;
            sm.baseY = sm.ok ? ye.sbaseY : 0.0;
        }
        const int iw0 = c.ici - pc.winH, jw0 = c.icj - pc.winH;
        stamp(pc, cyc, 2);
        // one memory round trip: the window's bit rows and the elevation of the two discs around known centres
        uint4 grp[NRL][KW + 1];
        win_issue<G, NRL, KW>(bm, m.g, g, iw0, jw0, grp);
        DiscLoads dc, dd;
        const bool dfltUsable = wantDefault && fabs(nx0) <= 1e6;
        uint32_t rowA = 0u, rowB = 0u;     // generic kernels: membership of the two discs, lane = box row
        bool rowsA = false, rowsB = false;  // ... when every box of the wavefront has at most 8 x 8 = 32 cells
        if constexpr (kMid) {
            disc_issue<G, false, kMid, kMid>(m, pc, c.cx, c.cy, bb, g, dc, ye.dy2);
            if (dfltUsable) disc_issue<G, false, kMid, kMid>(m, pc, nx0, ny, dbox, g, dd, ye.dy2);
        } else {
            dc.pipelined = dd.pipelined = false;
            dc.mid = dd.mid = false;
            const bool fitA = bb.ni <= 8 && bb.nj <= 8 && bb.ni * bb.nj <= 32;
            const bool fitB = !dfltUsable || (dbox.ni <= 8 && dbox.nj <= 8 && dbox.ni * dbox.nj <= 32);
            rowsA = __ballot(!fitA) == 0ull;
            rowsB = __ballot(!fitB) == 0ull;
            if (rowsA || rowsB) disc_rows8(m.g, pc.rf2, c.cx, nx0, c.cy, bb, dbox, g, rowA, rowB);
        }
        stamp(pc, cyc, 3);
        WinRows<NRL, KW> w;
        win_finish<NRL, KW>(jw0, grp, w);
#pragma unroll
        for (int k = 0; k < NRL; ++k) lb.a[g.sub + G * k] = w.Df[k][0];
        const CentroidScan sc = rows_from_bits<G, NRL, KW>(sm, w, g, iw0, jw0);
        bits_sync<G>();
        stamp(pc, cyc, 4);
        bool defaultOk;
        if constexpr (kMid) {
            defaultOk = default_ok_bits<G, KW, kMid>(m, pc, c.cx, c.cy, bb, dc, lb.a, lb.rows, iw0, jw0, g);  // cpp:2012
        } else {
            defaultOk = rowsA ? default_ok_rows8(rowA, bb, lb.a, lb.rows, iw0, jw0, g)
                              : default_ok_bits<G, KW, kMid>(m, pc, c.cx, c.cy, bb, dc, lb.a, lb.rows, iw0, jw0, g);
        }
        bits_sync<G>();  // lb doubles as scratch below
        bool unused;
        float zCentre = 0.0f;
        if constexpr (kMid) {
            uint32_t visA = 0u, visB = 0u;
            unit_put_disc<true>(m, pc, c.cx, c.cy, bb, g, dc, unit->eA, visA, scratch);
            if (dfltUsable) unit_put_disc<true>(m, pc, nx0, ny, dbox, g, dd, unit->eB, visB, scratch);
            if (g.sub == 0) {
                unit->visA = visA;
                unit->visB = visB;
            }
        } else {
            bool pre;
            if (rowsA) {
                ugVisA = box_mask_from_rows8(rowA, bb.nj, g);
            } else {
                unitg_put_disc(m, pc, c.cx, c.cy, bb, g, dc, ugVisA, pre, scratch);  // (not pipelined: the direct pass)
                if (pre) ugFlags |= kUgPreA;
            }
            ugAI0 = bb.i0; ugAJ0 = bb.j0; ugANj = max(bb.nj, 1);
            if (dfltUsable) {
                if (rowsB) {
                    ugVisB = box_mask_from_rows8(rowB, dbox.nj, g);
                } else {
                    unitg_put_disc(m, pc, nx0, ny, dbox, g, dd, ugVisB, pre, scratch);
                    if (pre) ugFlags |= kUgPreB;
                }
                ugBI0 = dbox.i0; ugBJ0 = dbox.j0; ugBNj = max(dbox.nj, 1);
            } else {
                ugFlags |= kUgPreB;
                ugVisB = __float_as_uint(zDefault);
            }
        }
        (void)unused;
        stamp(pc, cyc, 5);
        constexpr bool kOneCell = kMid;  // the 3x3-only variants are launched for one-cell foot discs
        CentroidPendingBits cp;
        centroid_begin_bits<G, kOneCell, kMid>(m, pc, c, sm, sc, zCentre, g, cp, ye.yA, ye.yB);          // cpp:818-821
        stamp(pc, cyc, 6);
        stamp(pc, cyc, 7);
        if (defaultOk) {
            no.valid = 1;
            no.source = 0;
            no.row = c.ici;
            no.col = c.icj;
            no.x = c.cx;  // cpp:2016-2017
            no.y = c.cy;
            no.z = zCentre;
        } else {
            nominal_invalid(no, c.cx, c.cy, 2);
            int wi = 0, wj = 0;
            bits_sync<G>();
            if (spiral_bits<G, NRL, KW, kMid>(m, pc, lut, head, c, w, lb, g, iw0, jw0, wi, wj, &ye)) {  // cpp:2022
                no.valid = 1;
                no.source = 1;
                no.row = wi;
                no.col = wj;
                no.x = cell_pos(m.g.baseX, m.g.res, wi);  // cpp:2105-2107
                no.y = cell_pos(m.g.baseY, m.g.res, wj);
                no.z = zCentre;  // z at the DEFAULT centre even for a candidate (cpp:2029)
            }
            bits_sync<G>();
        }
        stamp(pc, cyc, 8);
        if constexpr (kMid) {
            if (g.sub == 0) unit->eC = cp.e0;
            cp.o.z = 0.0f;
            if (g.sub == 0) unit->cenCode = static_cast<uint32_t>(cp.o.code) | (cp.needDisc != 0 ? 0x100u : 0u);
        } else {
            cp.o.z = 0.0f;
            if (cp.needDisc != 0) ugFlags |= kUgCTable;   // the result's own cell-centred disc (offset table)
            else if (cp.o.code == 0) ugFlags |= kUgCIsA;  // whole region valid: the height at the centre (cpp:1687)
        }
        co = cp.o;
    }
    lc->valid = no.valid;
    lc->v[0][0] = nx0;   lc->v[0][1] = ny;    lc->v[0][2] = static_cast<double>(zDefault);
    lc->v[1][0] = co.x;  lc->v[1][1] = co.y;  lc->v[1][2] = static_cast<double>(co.z);
    lc->v[2][0] = no.x;  lc->v[2][1] = no.y;  lc->v[2][2] = static_cast<double>(no.z);
    if constexpr (!kMid) {
        if (g.sub == 0) {  // what flush_unit_g needs to rebuild this leg's four records
            UnitG u;
            u.cx = c.cx; u.cenX = co.x; u.defX = nx0;
            u.aI0 = ugAI0; u.aJ0 = ugAJ0; u.visA = ugVisA;
            u.bI0 = ugBI0; u.bJ0 = ugBJ0; u.visB = ugVisB;
            u.nomRow = no.row; u.nomCol = no.col; u.cenRow = co.row; u.cenCol = co.col;
            u.flags = static_cast<uint32_t>(co.code) | (no.valid ? kUgValid : 0u) | (static_cast<uint32_t>(no.source) << kUgSrcShift) | ugFlags |
                      (static_cast<uint32_t>(ugANj) << 16) | (static_cast<uint32_t>(ugBNj) << 24);
            u.pad[0] = u.pad[1] = u.pad[2] = 0u;
            *unit = u;
        }
        return;
    } else if constexpr (kDefer) {
        if (g.sub == 0) {  // what flush_unit needs to rebuild this leg's four records
            unit->nomRow = no.row;
            unit->nomCol = no.col;
            unit->nomFlags = static_cast<uint32_t>(no.valid) | (static_cast<uint32_t>(no.source) << 8);
            unit->cenRow = co.row;
            unit->cenCol = co.col;
            if (!(!ls.radiusOk || !usable)) {
                // (cenCode was written above)
            } else {
                unit->cenCode = static_cast<uint32_t>(co.code);
            }
            unit->cx = c.cx;
            unit->cenX = co.x;
            unit->defX = nx0;
        }
        return;
    }
    if (g.sub == 0 && live) {
        const size_t o = (static_cast<size_t>(b) * nCycles + cyc) * 4 + leg;
        if (out.nominal) store_foothold<true>(out.nominal + o, no, leg, cyc);
        store_selected<true>(out, o, no.row, no.col, no.z, no.valid, no.source, leg, cyc);
        if (out.centroid) {
            fpe_centroid_foothold cf;
            cf.x = co.x; cf.y = co.y; cf.z = co.z; cf.row = co.row; cf.col = co.col;
            cf.code = static_cast<uint8_t>(co.code); cf.pad[0] = cf.pad[1] = cf.pad[2] = 0;
            store_record<true>(out.centroid + o, cf);
        }
        if (out.default_next) {
            store_record<true>(out.default_next + o * 3 + 0, static_cast<double>(nx0));
            store_record<true>(out.default_next + o * 3 + 1, static_cast<double>(ny));
            store_record<true>(out.default_next + o * 3 + 2, static_cast<double>(zDefault));
        }
    }
}

// The common case of the 3x3-only kernels as straight-line code: every swing leg of the wavefront has a usable centre,
// its two foot-disc boxes are unclamped 3x3 boxes (the middle cell is inside the disc whatever the centre,
// PlanConsts::midCellInside) and the default track is wanted and usable.  No LDS hand-offs besides the spiral's pass
// rows: the default check is evaluated by the lanes that OWN the three window rows of the box against the ballot of
// the membership tests; loads are unconditional; the centroid case logic is a chain of selects.  Any other situation
// (map border, unusable centre, missing products) sends the whole wavefront through leg_phase_bits8 for this phase.
// Constants of the fast path held in VECTOR registers for the whole kernel: as kernel arguments they live in scalar
// memory, and with more uniform state than SGPRs the compiler re-fetches them (s_load + wait) inside the cycle loop.
struct HotConsts {
    double rf, rf2, cornerEps, oneMinusEps, drift;
};
// Lane roles of the x pass.  Lane q of a leg group evaluates the index of ONE box corner — 0/1 centre disc (cx -+ rf),
// 2/3 centroid rectangle (cx -+ lx / 2), 4 getIndex(cx), 5/6 default-track disc (nx0 -+ rf); lane 7 evaluates nothing
// — and, before that, the feet-polygon centre of the track its corner belongs to (centroid track on lanes 0-4, default
// track on 5-6, nominal track on 7), so that indices and positions reach the other lanes in ONE exchange.  The
// per-lane constants live in vector registers, computed once: written as selects on q inside the cycle loop they are
// rebuilt every cycle, and a chain of `q == k` tests is compiled into a switch, i.e. into exec-mask branches.
struct LaneRole {
    double hqS;       // signed half extent of the lane's corner: xq = (track position) + hqS
    double qLo, qHi;  // the predicted quotient of a box corner must lie strictly inside the map (raw lanes: unbounded)
};
__device__ __forceinline__ int lane_track(int q) { return (q == 5 || q == 6) ? 0 : (q == 7 ? 2 : 1); }
__device__ __forceinline__ LaneRole make_lane_role(int q, double rf, double lx, double cornerEps, double rowsD) {
    LaneRole r;
    const double inf = __builtin_huge_val();
    const bool raw = q == 4 || q == 7;
    const double h = (q == 2 || q == 3) ? 0.5 * lx : (raw ? 0.0 : rf);
    const bool minus = q == 1 || q == 3 || q == 6;
    r.hqS = in_vgpr(minus ? -h : h);
    r.qLo = in_vgpr(raw ? -inf : cornerEps - rowsD);
    r.qHi = in_vgpr(raw ? inf : -cornerEps);
    return r;
}
// The first two rounds of the candidate scan (ranks 0-15) WITHOUT the LDS: those sixteen cells lie within two rows and
// columns of the centre (rings 0, 1 and the head of ring 2: SpiralLut::fast16), i.e. in FIVE consecutive window rows, and
// a group's eight lanes own eight consecutive rows per slot — so every one of the five rows has its own lane.  That lane
// looks at the five pass bits around the centre column of ITS row and turns each into the bit (1 << rank) of the
// candidate it stands for (rowTab: the rank per column offset, fetched once per kernel); an OR over the group (three
// DPP steps) gives the sixteen candidates' verdicts, the lowest set bit is the first valid cell in SpiralIterator order
// (cpp:2085-2114), and its offset comes out of two packed 64-bit tables.  Before: the pass rows written to the leg's
// LDS, a fence, two dependent LDS reads, two ballots and a ds_bpermute per search — four LDS round trips that the two
// wavefronts of a SIMD cannot hide (stage trace: 1 150 clocks per search, 88 % of the headline's cycles have one).
struct FastRanks {
    uint32_t rowTab;          // this lane's row: five 5-bit ranks by column offset -2..2 (31: none), 0x1FFFFFF when the lane owns none of the five rows
    int slot;                 // which of the lane's NRL rows it is
    unsigned long long di, dj;  // (offset + 2) of rank q in the 4-bit field q
};
template <int NRL>
__device__ __forceinline__ FastRanks load_fast_ranks(const SpiralLut& lut, const Grp<8>& g, int winH) {
    FastRanks fr;
    fr.slot = 0;
    int d = 99;
#pragma unroll
    for (int k = 0; k < NRL; ++k) {
        const int dk = g.sub + 8 * k - winH;  // row offset from the centre row (window row winH)
        const bool mine = dk >= -2 && dk <= 2;
        fr.slot = mine ? k : fr.slot;
        d = mine ? dk : d;
    }
    const uint32_t w = lut.fast16[min(max(d + 2, 0), 4)];
    fr.rowTab = d == 99 ? 0x1FFFFFFu : w;
    fr.di = *reinterpret_cast<const unsigned long long*>(lut.fast16 + 6);
    fr.dj = *reinterpret_cast<const unsigned long long*>(lut.fast16 + 8);
    asm volatile("" : "+v"(fr.di), "+v"(fr.dj));  // (uniform, but kept in vector registers: the chain has no scalar registers to spare)
    return fr;
}

template <int NRL, bool kNoDefault>
__device__ __forceinline__ void leg_fast8m(const DevMap& m, const BitMap& bm, const PlanConsts& pc, const HotConsts& hc, const LaneRole& role,
                                           const FastRanks& fk,
                                           const SpiralLut& lut, const LutHead& head, PoseShared& sh, const LegBits& lb, const Grp<8>& g,
                                           int leg, const LegStatic& ls, const YEntry& yeIn, double myCtr, double advance, int cyc,
                                           int nCycles, int b, bool live, const fpe_plan_out& out, LegCommit* lc, Unit* unit) {
    constexpr int G = 8, KW = 1;
    const LegConst& lk = ls.lk;
    // the entry's scalar fields in ONE batch of LDS reads (scattered reads would each wait for their own round trip);
    // dy2 stays in LDS (lane-dependent index)
    const YEntry& yeLds = yeIn;
    YEntry ye;
    ye.jc = yeLds.jc; ye.j0d = yeLds.j0d; ye.njd = yeLds.njd; ye.j0r = yeLds.j0r;
    ye.njr = yeLds.njr; ye.jA = yeLds.jA; ye.jB = yeLds.jB; ye.flags = yeLds.flags;
    ye.ny = yeLds.ny; ye.sbaseY = yeLds.sbaseY; ye.yA = yeLds.yA; ye.yB = yeLds.yB;
    ye.rmask = yeLds.rmask; ye.pmask = yeLds.pmask;
    // ---- x side: this lane's track position and corner (cpp:2199, 2414; see leg_phase_bits8) ----
    const double nxq = (myCtr + advance) + ls.biasX;
    const double ny = ye.ny;
    // (kNoDefault: the launch writes no default-track product — compile-time, see specialise_products: the default-track disc is
    // neither loaded nor tested; its lanes of the x pass still run, in the same instructions as the others)
    const bool wantDefault = kNoDefault ? false : out.default_next != nullptr;
    const double xq = nxq + role.hqS;  // a - h == a + (-h)
    const double qf = ((xq - m.g.orgX) - m.g.posX) * m.g.rinv;
    const double kq = trunc(qf);
    const double fr = fabs(qf - kq);
    const bool safe = (fr > hc.cornerEps && fr < hc.oneMinusEps && qf < role.qHi && qf > role.qLo) || g.sub == 7;
    const int idxq = -static_cast<int>(kq);
    constexpr int kKeep = (~(G - 1)) & 0x1F;
    const int i0d = bcast8_dpp<0>(idxq);
    const int i1d = bcast8_dpp<1>(idxq);
    const int i0r = bcast8_dpp<2>(idxq);
    const int i1r = bcast8_dpp<3>(idxq);
    const int ici = bcast8_dpp<4>(idxq);
    const int i0f = bcast8_dpp<5>(idxq);
    const int i1f = bcast8_dpp<6>(idxq);
    const double cx = bcast8_dpp_f64<0>(nxq);   // centre from the CENTROID track (cpp:861-862)
    const double nx0 = bcast8_dpp_f64<5>(nxq);  // default track
    const double nx2 = bcast8_dpp_f64<7>(nxq);  // nominal track (search polygon)
    const int j0d = ye.j0d, icj = ye.jc;
    // the window rows are requested before anything else looks at the indices (win_issue clamps whatever it is given;
    // the rare path below discards them): the round trip runs under the box tests, the ballot and the submap arithmetic
    const int iw0 = ici - pc.winH, jw0 = icj - pc.winH;
    uint4 grp[NRL][KW + 1];
    win_issue<G, NRL, KW>(bm, m.g, g, iw0, jw0, grp);
    // both foot-disc boxes: 3x3 and clear of the map's outermost rows / columns (not clamped, inside the map)
    // (bitwise: a short-circuit chain is compiled into exec-mask branches)
    const int lowest = kNoDefault ? min(i0d, j0d) : min(min(i0d, i0f), j0d), lastRow = (kNoDefault ? i0d : max(i0d, i0f)) + 4;
    const bool boxF = kNoDefault ? true : ((i1f - i0f) == 2);
    const bool boxes = ((i1d - i0d) == 2) & boxF & (ye.njd == 3) & (lowest >= 1) & (lastRow <= m.g.rows) & (j0d + 4 <= m.g.cols);
    // (kNoDefault: lanes 5-6 evaluate default-track corners nobody reads: their `safe` / magnitude tests do not count)
    const bool dfltLane = (g.sub == 5) | (g.sub == 6);
    const bool laneOk = kNoDefault ? (dfltLane | (safe & (fabs(nxq) <= 1e6))) : (safe & (fabs(nxq) <= 1e6));
    const bool rare = !ls.radiusOk | ((ye.flags & 2) == 0) | !laneOk | (!kNoDefault & !wantDefault) | !boxes;
    if (__ballot(rare) != 0ull) {  // wave-uniform
        const double ctr0 = swizzle_f64<kKeep | (5 << 5)>(myCtr), ctr1 = swizzle_f64<kKeep | (0 << 5)>(myCtr),
                     ctr2 = swizzle_f64<kKeep | (7 << 5)>(myCtr);
        leg_phase_bits8<NRL, true>(m, bm, pc, lut, head, sh, lb, g, leg, ls, yeIn, ctr0, ctr1, ctr2, advance, cyc, nCycles, b, live, out, lc, unit);
        return;
    }
    // getSubmapInformation's tail, x part (corners strictly inside the map: within); y part from the entry
    Submap sm;
    sm.i0 = i0r;
    sm.j0 = ye.j0r;
    sm.ni = i1r - i0r + 1;
    sm.nj = ye.njr;
    {
        const double cornerX = cell_pos(m.g.baseX, m.g.res, sm.i0) - (-(0.5 * m.g.res));
        const double subLenX = static_cast<double>(sm.ni) * m.g.res;
        const double subOrgX = 0.5 * subLenX;
        const double subPosX = cornerX - subOrgX;
        sm.ok = (ye.flags & 1) != 0 && within_axis(cx, subOrgX, subPosX, subLenX);
        sm.baseX = subPosX + (subOrgX - 0.5 * m.g.res);
        sm.baseY = ye.sbaseY;
    }
    stamp(pc, cyc, 2);
    // ---- same round trip: the elevation of both discs (lane = cell t of the 3x3 boxes) ----
    const int t = g.sub + (g.sub >= 4 ? 1 : 0);
    const int a = t >= 6 ? 2 : (t >= 3 ? 1 : 0);
    const int bq = t - 3 * a;
    const double dy2 = yeLds.dy2[bq];
    const double dxA = cell_pos(m.g.baseX, m.g.res, i0d + a) - cx;
    const double dxB = cell_pos(m.g.baseX, m.g.res, i0f + a) - nx0;
    const bool visA = (dxA * dxA + dy2) <= hc.rf2;  // CircleIterator::isInside (cell_in_disc)
    const bool visB = kNoDefault ? false : (dxB * dxB + dy2) <= hc.rf2;
    // (32-bit cell offsets from the uniform layer base: bits_supported bounds the layer below 2 GiB)
    const unsigned colsU = static_cast<unsigned>(m.g.cols);
    const unsigned laneCell = __umul24(static_cast<unsigned>(a), colsU) + static_cast<unsigned>(bq);
    const unsigned boxA = __umul24(static_cast<unsigned>(i0d), colsU) + static_cast<unsigned>(j0d);
    const unsigned boxB = __umul24(static_cast<unsigned>(i0f), colsU) + static_cast<unsigned>(j0d);
    const float eA = load_cell(m.elev, boxA + laneCell);
    const float eMidA = load_cell(m.elev, boxA + colsU + 1u);
    float eB = 0.0f, eMidB = 0.0f;
    if constexpr (!kNoDefault) {
        eB = load_cell(m.elev, boxB + laneCell);
        eMidB = load_cell(m.elev, boxB + colsU + 1u);
    }
    // In the shadow of that round trip: the rows of the search rectangle, which only a spiral search uses — but most
    // wavefronts have one leg in eight that needs it (88 % of the headline's cycles), and these forty instructions
    // would otherwise sit on the dependent chain behind the default check.  (Moving the candidates' window addresses
    // and the chain-independent unit fields up here as well changed nothing.)
    const bool fastSpiral = __ballot(ls.polyKind != 0 || lk.nRings < 4 || lk.nCand < 16) == 0ull && pc.nFoot <= 1;  // uniform
    int iA = 0, iB = -1;
    if (fastSpiral) {
        const double r = static_cast<double>(ls.Rf);
        const double xhi = nx2 + r, xlo = nx2 - r;  // getSearchPolygon around the NOMINAL track (cpp:2496-2517)
        double qh = floor((m.g.baseX - xhi) * m.g.rinv), ql = floor((m.g.baseX - xlo) * m.g.rinv);
        qh = fmin(fmax(qh, -1.0e9), 1.0e9);
        ql = fmin(fmax(ql, -1.0e9), 1.0e9);
        const int eH = static_cast<int>(qh), eL = static_cast<int>(ql);
        // lane q & 3: 0 P(eH), 1 P(eH + 1) with P(i) = x_i < xhi;  2 Q(eL + 1), 3 Q(eL) with Q(i) = x_i >= xlo
        const bool isLo = (g.sub & 2) != 0;
        const int odd = g.sub & 1;
        const int tLo = eL + 1 - odd, tHi = eH + odd;
        const int tq = isLo ? tLo : tHi;
        const double lim = isLo ? xlo : xhi;
        const double xt = cell_pos(m.g.baseX, m.g.res, tq);
        const bool predLo = xt >= lim, predHi = xt < lim;
        const bool pred = isLo ? predLo : predHi;
        const unsigned pb = static_cast<unsigned>(g.ballot(pred));
        const int iA1 = (pb & 2u) ? eH + 1 : eH + 2, iB1 = (pb & 8u) ? eL : eL - 1;
        iA = (pb & 1u) ? eH : iA1;
        iB = (pb & 4u) ? eL + 1 : iB1;
    }
    stamp(pc, cyc, 3);
    WinRows<NRL, KW> w;
    win_finish<NRL, KW>(jw0, grp, w);
    const CentroidScan sc = rows_from_bits<G, NRL, KW>(sm, w, g, iw0, jw0, &ye.rmask);
    stamp(pc, cyc, 4);
    // ---- checkDefaultFoothold: the lanes owning the box's three window rows test their Df bits under the members ----
    const unsigned mA = static_cast<unsigned>(g.ballot(visA)), mB = kNoDefault ? 0u : static_cast<unsigned>(g.ballot(visB));
    // the nine membership bits in CircleIterator order (the middle cell is always a member)
    const unsigned visA9 = (mA & 0xFu) | 0x10u | ((mA & 0xF0u) << 1), visB9 = kNoDefault ? 0u : ((mB & 0xFu) | 0x10u | ((mB & 0xF0u) << 1));
    bool fail = false;
    {
        const unsigned sh3 = static_cast<unsigned>(j0d - jw0) & 31u;
#pragma unroll
        for (int k = 0; k < NRL; ++k) {
            const int ar = g.sub + G * k - (i0d - iw0);  // row of the box held in slot k
            const unsigned bitsRow = (visA9 >> (3u * (static_cast<unsigned>(ar) & 3u))) & 7u;
            const unsigned sel = static_cast<unsigned>(ar) < 3u ? bitsRow : 0u;
            fail |= (((w.Df[k][0] >> sh3) & 7u) & sel) != 0u;
        }
    }
    const bool defaultOk = !g.any(fail);  // the middle cell is always visited (cpp:2069-2081: at least one cell)
    // ---- deposits for flush_unit: elevations in CircleIterator order ----
    unit->eA[t] = eA;
    unit->eA[4] = eMidA;  // every lane stores the same value
    if constexpr (!kNoDefault) {
        unit->eB[t] = eB;
        unit->eB[4] = eMidB;
    }
    stamp(pc, cyc, 5);
    // ---- centroid method (cpp:1684-1952) as selects ----
    const int bottomRow = sm.ni - 1, rightCol = sm.nj - 1;
    const int minRow = sc.minRow, maxRow = sc.maxRow;
    // (every select below has two ready operands: nested conditionals are compiled into branches)
    const bool top = minRow == 0, bottom = maxRow == bottomRow;
    const bool case1 = top && !bottom;
    const bool case2 = !top && !bottom;
    const bool upper = minRow >= (bottomRow - maxRow);
    const int code23 = upper ? 2 : 3, code51 = bottom ? 5 : 1;
    int code = bottom ? 4 : code23;  // case3 (4) / case2 (2, 3): the first row is not blocked
    code = top ? code51 : code;      // case1 (1) / no case (5)
    code = sc.whole ? 0 : code;
    code = sm.ok ? code : 6;
    const bool useMaxRow = case1 || (case2 && !upper);
    const int rowA = (maxRow + bottomRow + (case1 ? 1 : 0)) >> 1, rowB = (minRow + 1) >> 1;
    const int newRow = useMaxRow ? rowA : rowB;
    const int newCol = (rightCol + (case1 ? 1 : 0)) >> 1;
    const bool whole = code == 0;
    const bool hasCell = static_cast<unsigned>(code - 1) < 4u;
    CentroidOut co;
    co.code = code;
    co.z = 0.0f;
    const double cellX = cell_pos(sm.baseX, m.g.res, newRow);  // cpp:1816
    const double yAB = code == 1 ? ye.yA : ye.yB;
    const double xCell = hasCell ? cellX : 0.0, yCell = hasCell ? yAB : 0.0;
    const int rowCell = hasCell ? sm.i0 + newRow : -1, colCell = hasCell ? sm.j0 + newCol : -1;
    co.x = whole ? cx : xCell;  // cpp:1687
    co.y = whole ? ny : yCell;
    co.row = whole ? ici : rowCell;
    co.col = whole ? icj : colCell;
    stamp(pc, cyc, 7);
    // ---- nominal result: the default foothold, else the spiral search (cpp:2012-2029) ----
    NominalOut no;
    no.valid = 1;
    no.source = 0;
    no.row = ici;
    no.col = icj;
    no.x = cx;  // cpp:2016-2017
    no.y = ny;
    no.z = 0.0f;
    if (!defaultOk) {
        nominal_invalid(no, cx, ny, 2);
        const double r = static_cast<double>(ls.Rf);
        int wi = 0, wj = 0;
        bool found = false, searched = false;
        // The usual search as straight-line code: reference rectangle, one-cell foot disc, and the candidates of the
        // first two rounds (ranks 0-15: rings 0-2, whose cells the iterator does not filter when nRings >= 4).  Same
        // evaluation as spiral_bits: x interval as in rectangle_index_bounds, columns from the y entry, pass rows
        // P = ~F | (~C & inside) in the leg's LDS, lowest set ballot bit = first valid cell in spiral order.
        if (fastSpiral) {
#ifdef FPE_FAST16_LDS
            const int NR = lb.rows;
#pragma unroll
            for (int k = 0; k < NRL; ++k) {
                const int ri = g.sub + G * k;
                const int i = iw0 + ri;
                const unsigned inside = (i >= iA && i <= iB) ? ye.pmask : 0u;
                if (ri < NR) lb.a[ri] = ~w.F[k][0] | (~w.C[k][0] & inside);
            }
            bits_sync<G>();
            const int rowW = ici - iw0, colW = icj - jw0;  // the centre inside the window (winH, winH)
            static_assert(kLutHeadRounds == 2, "both register rounds are evaluated side by side");
            // both rounds' pass bits in flight together, one broadcast of the winning entry
            const int e0 = head.dij[0], e1 = head.dij[1];
            const int di0 = static_cast<int16_t>(e0 & 0xFFFF), dj0 = e0 >> 16, di1 = static_cast<int16_t>(e1 & 0xFFFF), dj1 = e1 >> 16;
            const unsigned bit0 = win_bit<KW>(lb.a, NR, rowW + di0, colW + dj0), bit1 = win_bit<KW>(lb.a, NR, rowW + di1, colW + dj1);
            const bool ok0 = in_range(ici + di0, icj + dj0, m.g.rows, m.g.cols) & (bit0 != 0u);
            const bool ok1 = in_range(ici + di1, icj + dj1, m.g.rows, m.g.cols) & (bit1 != 0u);
            const unsigned m0 = static_cast<unsigned>(g.ballot(ok0)), m1 = static_cast<unsigned>(g.ballot(ok1));
            const bool first = m0 != 0u;
            const unsigned mSel = first ? m0 : m1;
            const int eMine = first ? e0 : e1;
            const int eWin = g.bcast(eMine, __builtin_ctz(mSel | 0x100u) & 7);
            found = (m0 | m1) != 0u;
            wi = ici + static_cast<int16_t>(eWin & 0xFFFF);
            wj = icj + (eWin >> 16);
            searched = lk.nCand <= G * kLutHeadRounds;  // nothing beyond the two rounds
            bits_sync<G>();
#else
            // this lane's row of the five around the centre (FastRanks): pass bits P = ~F | (~C & inside) (cpp:2132-2138)
            unsigned Fs = w.F[0][0], Cs = w.C[0][0];
#pragma unroll
            for (int k = 1; k < NRL; ++k) {
                Fs = fk.slot == k ? w.F[k][0] : Fs;
                Cs = fk.slot == k ? w.C[k][0] : Cs;
            }
            const int i = iw0 + g.sub + G * fk.slot;
            const unsigned inside = (i >= iA && i <= iB) ? ye.pmask : 0u;
            unsigned P = ~Fs | (~Cs & inside);
            // cells outside the map pass every test (their F bit is 0) but are no candidates: windows over the map's edge only
            const bool border = (iw0 < 0) | (jw0 < 0) | (iw0 + G * NRL > m.g.rows) | (jw0 + 32 > m.g.cols);
            if (__ballot(border) != 0ull) {  // wave-uniform, rare
                const unsigned colIn = range_word(-jw0, m.g.cols - 1 - jw0, 0);
                P = static_cast<unsigned>(i) < static_cast<unsigned>(m.g.rows) ? (P & colIn) : 0u;
            }
            const unsigned b5 = P >> static_cast<unsigned>(pc.winH - 2);  // bit c = column offset c - 2 from the centre column (winH)
            unsigned m16 = 0u;
#pragma unroll
            for (int c = 0; c < 5; ++c) m16 |= ((b5 >> c) & 1u) << ((fk.rowTab >> (5 * c)) & 31u);  // (rank 31: not a candidate)
            const unsigned all16 = or_reduce8(m16) & 0xFFFFu;
            found = all16 != 0u;
            const unsigned rank4 = static_cast<unsigned>(__builtin_ctz(all16 | 0x10000u) & 15) * 4u;
            wi = ici + static_cast<int>((fk.di >> rank4) & 7ull) - 2;
            wj = icj + static_cast<int>((fk.dj >> rank4) & 7ull) - 2;
            searched = lk.nCand <= 16;  // nothing beyond the sixteen
#endif
        }
        if (!found && !searched) {  // other polygons, larger foot discs, small search radii, or no hit in the first two rounds
            LegCtx c;
            c.cyc = cyc;
            c.cx = cx;
            c.cy = ny;
            c.nv = ls.polyKind == 0 ? 4 : 6;
            c.rect = ls.polyKind == 0;
            c.xhi = nx2 + r;
            c.xlo = nx2 - r;
            c.yhi = ny + 0.5 * r;
            c.ylo = ny - 0.5 * r;
            c.vx = sh.polyX[leg];
            c.vy = sh.polyY[leg];
            c.footDa = sh.footDa;
            c.footDb = sh.footDb;
            c.footOff = sh.footOff;
            c.R2 = lk.R2;
            c.nRings = lk.nRings;
            c.nCand = lk.nCand;
            c.ti0 = c.tj0 = 0;
            c.ici = ici;
            c.icj = icj;
            if (!c.rect) {  // hexagon vertices from the NOMINAL track's position (build-defined, App. E)
                if (g.sub == 0) {
                    double* vx = sh.polyX[leg];
                    double* vy = sh.polyY[leg];
                    const double hx = 0.5 * r, hy = (0.5 * r) * 0.8660254037844386;
                    vx[0] = nx2 + r;   vy[0] = ny;
                    vx[1] = nx2 + hx;  vy[1] = ny - hy;
                    vx[2] = nx2 - hx;  vy[2] = ny - hy;
                    vx[3] = nx2 - r;   vy[3] = ny;
                    vx[4] = nx2 - hx;  vy[4] = ny + hy;
                    vx[5] = nx2 + hx;  vy[5] = ny + hy;
                }
                bits_sync<G>();
            }
            found = spiral_bits<G, NRL, KW, true>(m, pc, lut, head, c, w, lb, g, iw0, jw0, wi, wj, &yeIn);  // cpp:2022
            bits_sync<G>();
        }
        if (found) {
            no.valid = 1;
            no.source = 1;
            no.row = wi;
            no.col = wj;
            no.x = cell_pos(m.g.baseX, m.g.res, wi);  // cpp:2105-2107
            no.y = cell_pos(m.g.baseY, m.g.res, wj);
        }
    }
    stamp(pc, cyc, 8);
    if (g.sub == 0) {  // what flush_unit needs to rebuild this leg's four records
        unit->visA = visA9;
        unit->visB = visB9;
        unit->nomRow = no.row;
        unit->nomCol = no.col;
        unit->nomFlags = static_cast<uint32_t>(no.valid) | (static_cast<uint32_t>(no.source) << 8);
        unit->cenRow = co.row;
        unit->cenCol = co.col;
        unit->cenCode = static_cast<uint32_t>(code) | (hasCell ? 0x200u : 0u);  // flush_unit reads the result's own cell
        unit->cx = cx;
        unit->cenX = co.x;
        unit->defX = nx0;
    }
    lc->valid = no.valid;
    lc->v[0][0] = nx0;   lc->v[0][1] = ny;    lc->v[0][2] = 0.0;
    lc->v[1][0] = co.x;  lc->v[1][1] = co.y;  lc->v[1][2] = 0.0;
    lc->v[2][0] = no.x;  lc->v[2][1] = no.y;  lc->v[2][2] = 0.0;
}

}  // namespace

// ---- products as a compile-time mask --------------------------------------------------------------------------------
// Which of fpe_plan_out's products a launch writes is a run-time null test per product in the generic instantiation
// (kProd = 0: any combination).  The two shapes that matter are compiled on their own: kProd = 2, ALL seven base products
// (bench.py's headline step, fpe_plan with every array: the tests fold away) and kProd = 1, the NOMINAL track only —
// {nominal, selected, selected_packed, cycle_ok}: the service's response (cpp:1588) and the multi-GPU exchange record —
// where the default-track disc (its loads, membership, deposits and height sums), the centroid result's height and record,
// the stance and the first-cycle gate are not compiled at all.  The engine picks the instantiation from the pointers.
template <int kProd>
__device__ __forceinline__ fpe_plan_out specialise_products(fpe_plan_out out) {
    if constexpr (kProd == 1) {
        out.centroid = nullptr;
        out.default_next = nullptr;
        out.stance = nullptr;
        out.pose_status = nullptr;
    } else if constexpr (kProd == 2) {
        __builtin_assume(out.nominal != nullptr);
        __builtin_assume(out.centroid != nullptr);
        __builtin_assume(out.default_next != nullptr);
        __builtin_assume(out.cycle_ok != nullptr);
        __builtin_assume(out.stance != nullptr);
        __builtin_assume(out.selected != nullptr);
        __builtin_assume(out.pose_status != nullptr);
    }
    return out;
}
__host__ inline int product_shape(const fpe_plan_out& o) {
    if (o.nominal && o.centroid && o.default_next && o.cycle_ok && o.stance && o.selected && o.pose_status) return 2;
    if (!o.centroid && !o.default_next && !o.stance && !o.pose_status) return 1;
    return 0;
}

// ---- chained plan on the bit window: 8 lanes per leg, two poses per wavefront ------------------------------------
#ifndef FPE_BITS_GENERIC_WAVES
#define FPE_BITS_GENERIC_WAVES 3  // measured on cfg-4: 2 -> 1.36 ms, 3 -> 1.25 ms (27 spilled VGPRs), 4 -> 1.46 ms (69 spilled)
#endif
template <int NRL, bool kMid, int kProd>
// (the pose pointer and the counts lead the argument list: scalar arguments at the head of the kernarg segment are
// preloaded into SGPRs at wave launch, -amdgpu-kernarg-preload-count, so the pose loads can be issued at once)
__global__ __launch_bounds__(64, kMid ? 2 : FPE_BITS_GENERIC_WAVES) void plan_bits_kernel(const fpe_pose* __restrict__ poses, int B, int nCycles,
                                                          DevMap mArg, BitMap bm, PlanConsts pc, SpiralLut lut, fpe_plan_out outArg) {
    constexpr int G = 8;
    const fpe_plan_out out = specialise_products<kProd>(outArg);
    constexpr bool kNoDefault = kProd == 1;
    constexpr int NR = G * NRL;
    constexpr int kPoseThreads = 4 * G;
    const int tid = static_cast<int>(threadIdx.x);
    const int slot = tid / kPoseThreads;
    const int leg = (tid / G) & 3;
    // the pose first: its address needs nothing but the preloaded arguments, and everything else waits for it
    int b = blockIdx.x * 2 + slot;
    const bool live = b < B;  // the padding pose of the last block runs the chain on pose B-1, stores nothing
    if (!live) b = B - 1;
    const fpe_pose* pp = poses + b;
    const double x0 = pp->position[0], y0 = pp->position[1], z0 = pp->position[2];
    const int gait = pp->gait;
    const float rOverride = pp->leg_search_radius[leg];
    const int polyKindIn = pp->leg_polygon_kind[leg];
    __builtin_amdgcn_sched_barrier(0);  // (the loads above stay ahead of the kernel-argument fetches below)
    stamp(pc, 1, 11);
    stamp(pc, 6, 14);  // (-DFPE_TRACE_ALL_BLOCKS builds: start / end / hardware id of every workgroup, with the stamp after the cycle loop)
    // the map geometry doubles are operands of vector f64 arithmetic only: parked in VGPRs (see plan_chained_kernel) — in
    // the 3x3-only variants; the generic ones run at their register cap (168 VGPRs at three wavefronts per SIMD), where the
    // twenty registers cost more in spills than the scalar operands do in moves (measured: cfg-4 0.713 -> 0.664 ms without)
    DevMap m = mArg;
    if constexpr (kMid) {
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
    HotConsts hc;
    hc.rf = in_vgpr(pc.rf);
    hc.rf2 = in_vgpr(pc.rf2);
    hc.cornerEps = in_vgpr(pc.cornerEps);
    hc.oneMinusEps = in_vgpr(1.0 - pc.cornerEps);
    hc.drift = kMid ? in_vgpr(pc.drift) : pc.drift;  // (the generic variants run at their register cap: nothing extra parked)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const Grp<G> g(tid);
    const size_t legBytes = 4 * static_cast<size_t>(legbits_words(NR, 1, pc.nHW, false));
    // cycles between two flushes (units and y entries staged in LDS): eight for the 3x3-only kernels, four for the generic ones
    constexpr int kBatch = kMid ? 8 : 4;
    using UnitT = typename std::conditional<kMid, Unit, UnitG>::type;
    const size_t poseBytes = sizeof(PoseShared) + 4 * legBytes + (sizeof(YEntry) + sizeof(UnitT)) * 4 * kBatch;
    unsigned char* base = smem + static_cast<size_t>(slot) * poseBytes;
    PoseShared& sh = *reinterpret_cast<PoseShared*>(base);
    const LegBits lb = make_legbits(base + sizeof(PoseShared) + static_cast<size_t>(leg) * legBytes, NR, 1, pc.nHW, false);
    YEntry* ytab = reinterpret_cast<YEntry*>(base + sizeof(PoseShared) + 4 * legBytes) + leg * kBatch;  // [cycle % kBatch] of this leg
    UnitT* units = reinterpret_cast<UnitT*>(base + sizeof(PoseShared) + 4 * legBytes + sizeof(YEntry) * 4 * kBatch) + leg * kBatch;

    const LutHead head = load_lut_head(lut, g);
#ifdef FPE_TRACE
    asm volatile("" ::"v"(x0), "v"(gait));
    stamp(pc, 3, 11);  // pose arrived
    asm volatile("" ::"v"(head.dij[0]), "v"(head.ring[1]));
    stamp(pc, 3, 12);  // rank-table head arrived
#endif
    LegStatic ls;
    {
        if (__ballot(rOverride > 0.0f) != 0ull) {  // some leg of the wavefront overrides the search radius (build-defined)
            ls = make_leg_static(pc, pp, leg, m.g.res, lut);
        } else {  // the reference's single searchRadius_: constants precomputed on the host
            ls.Rf = pc.searchRadius;
            ls.polyKind = polyKindIn;
            ls.radiusOk = true;
            const double R = static_cast<double>(pc.searchRadius);
            ls.lk.Rf = pc.searchRadius;
            ls.lk.R2 = R * R;
            ls.lk.nRings = pc.defNRings;
            ls.lk.nCand = pc.defNCand;
            ls.lk.lx = static_cast<double>(pc.searchRadius * 2);
            ls.lk.ly = static_cast<double>(pc.searchRadius);
            // (two selects on the leg's bits: a run-time index into the kernel-argument array is a dependent global load)
            const bool odd = (leg & 1) != 0, high = (leg & 2) != 0;
            const double bxLo = odd ? pc.biasX[1] : pc.biasX[0], bxHi = odd ? pc.biasX[3] : pc.biasX[2];
            const double byLo = odd ? pc.biasY[1] : pc.biasY[0], byHi = odd ? pc.biasY[3] : pc.biasY[2];
            ls.biasX = high ? bxHi : bxLo;
            ls.biasY = high ? byHi : byLo;
        }
    }
    // (loaded values parked here: inside the cycle loop the compiler would wait for "all outstanding loads" at their
    // first use in every iteration)
    if constexpr (kMid) {
        ls.biasX = in_vgpr(ls.biasX);
        ls.biasY = in_vgpr(ls.biasY);
    }
    stamp(pc, 3, 13);  // per-leg constants
    if constexpr (kMid) {  // launched for one-cell foot discs only: the table is the single offset (0, 0)
        if (tid % kPoseThreads == 0) {
            sh.footDa[0] = 0;
            sh.footDb[0] = 0;
            sh.footOff[0] = 0;
        }
    } else {
        for (int k = tid % kPoseThreads; k < pc.nFoot; k += kPoseThreads) {
            sh.footDa[k] = pc.footDa[k];
            sh.footDb[k] = pc.footDb[k];
            sh.footOff[k] = 0;
        }
    }
    stamp(pc, 3, 14);  // offset table copied
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
    bits_sync<G>();
    stamp(pc, 1, 12);
    if (out.pose_status) {
        // getGaitCycleSearchGridMap's getSubmap in the first cycle (opt_gate_cycle0), its four corners on four lanes
        const double gx = polygon_center_x(sh.cur[0]) + pc.step, gy = y0 + 0.0;  // cpp:2327-2329
        Submap gs;
        {
            // lane q & 3: 0 top-left x, 1 top-left y, 2 bottom-right x, 3 bottom-right y — predicted as in the x pass
            // of the chain; the reference's own expressions when any lane is near a cell boundary or the map's edge
            const bool isY = (g.sub & 1) != 0, isBR = (g.sub & 2) != 0;
            const double ctr = isY ? gy : gx, halfExt = isY ? 0.5 * pc.isosWid : 0.5 * pc.isosLen;
            const double org = isY ? m.g.orgY : m.g.orgX, pos = isY ? m.g.posY : m.g.posX;
            const double cells = isY ? static_cast<double>(mArg.g.cols) : static_cast<double>(mArg.g.rows);
            const double vq = isBR ? ctr - halfExt : ctr + halfExt;
            const double qf = ((vq - org) - pos) * m.g.rinv;
            const double kq = trunc(qf);
            const double fr = fabs(qf - kq);
            const bool safe = (fr > pc.cornerEps) & (fr < 1.0 - pc.cornerEps) & (qf < -pc.cornerEps) & (qf > pc.cornerEps - cells);
            if (__ballot(!safe) == 0ull) {
                const int idxq = -static_cast<int>(kq);
                constexpr int kKeep = (~(G - 1)) & 0x1F;
                BBox gbb;
                gbb.i0 = __builtin_amdgcn_ds_swizzle(idxq, kKeep | (0 << 5));
                gbb.j0 = __builtin_amdgcn_ds_swizzle(idxq, kKeep | (1 << 5));
                gbb.ni = __builtin_amdgcn_ds_swizzle(idxq, kKeep | (2 << 5)) - gbb.i0 + 1;
                gbb.nj = __builtin_amdgcn_ds_swizzle(idxq, kKeep | (3 << 5)) - gbb.j0 + 1;
                gs = submap_from_corners(m.g, gbb, true, gx, gy);
            } else {
                const Box gb{gx, gy, 0.5 * pc.isosLen, 0.5 * pc.isosWid};
                Corners<G, 8> gc;
                gc.eval(m.g, g, gb, gb, gb, gb, 0x0u);
                gs = submap_from_corners(m.g, gc.template bbox<0>(g), gc.box_within(0), gx, gy);
            }
        }
        if (live && leg == 0 && g.sub == 0)
            out.pose_status[b] = (centre_usable(gx, gy) && gs.ok) ? 0 : static_cast<uint8_t>(FPE_POSE_OPT_SUBMAP_FAILED);
    }
    stamp(pc, 1, 13);

    double adjY = 0.0;  // ajustedPose_[1], cpp:759
    const int nPhases = (gait == 1) ? 4 : 1;
    const double advance = (gait == 1) ? pc.stepQuarter : pc.step;
    // swing order LF,RH,RF,LH (RF_FIRST=false) or RF,LH,LF,RH (build-defined walk)
    const int walkOrder = pc.RF_FIRST ? ((0) | (2 << 2) | (3 << 4) | (1 << 6)) : ((3) | (1 << 2) | (0 << 4) | (2 << 6));
    constexpr int kPoseLanes = 4 * G;
    const unsigned long long poseMask = ((1ull << kPoseLanes) - 1ull) << (slot * kPoseLanes);
    // the track whose feet-polygon centre this lane evaluates: 3x3-only kernels: the track of the lane's corner
    // (LaneRole); generic kernels: lane t evaluates track t
    const int myTrack = kMid ? lane_track(g.sub) : (g.sub < 2 ? g.sub : 2);
    LaneRole role{};
    if constexpr (kMid) role = make_lane_role(g.sub, pc.rf, ls.lk.lx, pc.cornerEps, static_cast<double>(mArg.g.rows));
    FastRanks fr{};
    if constexpr (kMid) fr = load_fast_ranks<NRL>(lut, g, pc.winH);
    uint32_t okBits = 0u;  // cycleOk of the cycles since the last flush (3x3-only kernels: stored by flush_unit)

    // Issue priority, 3x3-only kernels (two wavefronts per SIMD at the headline's batch): the SIMD's arbiter serves the OLDER of
    // its two wavefronts first, so the older one finishes a sixth ahead (49 k against 59 k clocks, profiles/round3_residency.txt)
    // and the younger one runs its tail alone at half the issue rate.  Three eighths into the chain the younger wavefront (odd
    // hardware wave slot = launched second) raises its priority: the lead the older one built is what the younger one builds
    // from there on, and the two finish together.  Measured (round 4, 50-step A/B, six repetitions): headline 26.8 -> 25.5 us,
    // cfg-2 27.2 -> 26.0 us; switching at 2/8: the same, at 4/8: 25.9, at 1/8: 26.0, from the start (the plain reversal round 3
    // tried): 26.7 = no change; handing the priority back near the end or alternating every one / two cycles: 26.1 - 26.3.
    unsigned hwSlot = 0u;
    if constexpr (kMid) {
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwSlot));
        hwSlot &= 15u;
    }
#ifndef FPE_PRIO_SWAP_EIGHTHS
#define FPE_PRIO_SWAP_EIGHTHS 3
#endif
    for (int cyc = 0; cyc < nCycles; ++cyc) {
#ifndef FPE_NO_PRIO_SWAP
        if (kMid && cyc == (nCycles * FPE_PRIO_SWAP_EIGHTHS) / 8 && (hwSlot & 1u)) __builtin_amdgcn_s_setprio(2);
#endif
        if ((cyc & (kBatch - 1)) == 0) {
            // y side of the next kBatch cycles: lane (leg, s) fills the entry of cycle cyc + s.  ajustedPose_[1] is the
            // reference's running sum (cpp:1578): cycle cyc + s has seen s more additions of the drift
            double a = adjY, mine = adjY;
#pragma unroll
            for (int k = 1; k < kBatch; ++k) {
                a += hc.drift;
                if (g.sub == k) mine = a;
            }
            if (kBatch == 8 || g.sub < kBatch) fill_yentry(m.g, pc, ls, (y0 + mine) + ls.biasY, ytab[g.sub]);  // cpp:2201, 2414
            bits_sync<G>();
            stamp(pc, 1, 14);
        }
        const YEntry& ye = ytab[cyc & (kBatch - 1)];
        bool cycleOk = true;
        for (int ph = 0; ph < nPhases; ++ph) {
            const unsigned mask = (gait == 1) ? (1u << ((walkOrder >> (2 * ph)) & 3)) : 0xFu;
            const bool active = (mask >> leg) & 1u;
            stamp(pc, cyc, 0);
            // feet-polygon centres (getPolygonCenter, cpp:2191, 2265): every lane computes ONE track's centre from the
            // committed feet in LDS; the values reach the group's other lanes by swizzle (no LDS hand-off, no barrier)
            const double myCtr = polygon_center_x(sh.cur[myTrack]);
            stamp(pc, cyc, 1);
            // footholdValidation_ (cpp:1323) is a ballot over the pose's lanes; the committed positions go from
            // registers straight to PoseShared::cur (cpp:1332-1576)
            LegCommit lc;
            lc.valid = 1;  // non-swing legs do not vote
            if (active) {
                if constexpr (kMid) {
                    leg_fast8m<NRL, kNoDefault>(m, bm, pc, hc, role, fr, lut, head, sh, lb, g, leg, ls, ye, myCtr, advance, cyc, nCycles, b, live, out, &lc,
                                                units + (cyc & (kBatch - 1)));
                } else {
                    constexpr int kKeep = (~(G - 1)) & 0x1F;
                    const double ctr0 = swizzle_f64<kKeep | (0 << 5)>(myCtr), ctr1 = swizzle_f64<kKeep | (1 << 5)>(myCtr),
                                 ctr2 = swizzle_f64<kKeep | (2 << 5)>(myCtr);
                    leg_phase_bits8<NRL, false>(m, bm, pc, lut, head, sh, lb, g, leg, ls, ye, ctr0, ctr1, ctr2, advance, cyc, nCycles, b, live,
                                                out, &lc, units + (cyc & (kBatch - 1)));
                }
            }
            stamp(pc, cyc, 9);
            const bool phaseOk = (__ballot(lc.valid == 0) & poseMask) == 0ull;
            if (phaseOk && active && g.sub == 0) {
#pragma unroll
                for (int t = 0; t < 3; ++t) {  // x and y only: no later cycle reads a committed z (getPolygonCenter, cpp:2421-2463)
                    sh.cur[t][leg][0] = lc.v[t][0];
                    sh.cur[t][leg][1] = lc.v[t][1];
                }
            }
            bits_sync<G>();
            cycleOk = cycleOk && phaseOk;
            stamp(pc, cyc, 10);
        }
        adjY += hc.drift;  // cpp:1578
        okBits |= (cycleOk ? 1u : 0u) << (cyc & 7);
        if ((cyc & (kBatch - 1)) == kBatch - 1 || cyc == nCycles - 1) {
            // heights, output records and cycle validity of the last (up to) kBatch cycles: lane (leg, s) takes the
            // unit of cycle base + s
            const int c0 = cyc & ~(kBatch - 1);
            stamp(pc, 2, 11);
            if constexpr (kMid) {
                if (live && c0 + g.sub <= cyc) flush_unit(m, pc, units[g.sub], ytab[g.sub], b, c0 + g.sub, leg, nCycles, okBits, out);
            } else {
                const int us = g.sub >> 1;  // two lanes per unit (kBatch * 2 == G)
                if (live && c0 + us <= cyc)
                    flush_unit_g(m, pc, sh.footDa, sh.footDb, units[us], ytab[us], b, c0 + us, leg, g.sub & 1, nCycles, okBits, out);
            }
            okBits = 0u;
            bits_sync<G>();  // the units and the y entries are rewritten next
            stamp(pc, 2, 12);
        }
    }
    stamp(pc, 6, 15);
}

// ---- chained plan on the bit window, sequential-legs form (large windows): one wavefront per pose, lane = window
// row, KW words per row; the swing legs of a phase are searched one after the other (see plan_sequential_kernel) ----
#ifndef FPE_SEQ_WAVES  // wavefronts per SIMD the register allocation aims at (measurement builds: 3 / 5)
#define FPE_SEQ_WAVES 4
#endif
// The kernel's argument list as a struct: HIP lays a kernel's arguments out one after the other, each at its natural alignment —
// a C struct of the same members in the same order — so this is a VIEW of plan_bits_seq_kernel's argument segment, through which a
// leg search can read its constants again (FPE_SEQ_RELOAD_ARGS, below) instead of keeping them in scalar registers across the
// whole chain.  (The kernel keeps its separate arguments: taking this struct as its one argument cost <1, 2> 0.6 %.)  Any change
// of the kernel's signature must be mirrored here; every parity test of cfg-5's kernel fails loudly otherwise.
struct SeqKernArgs {
    DevMap m;
    BitMap bm;
    PlanConsts pc;
    SpiralLut lut;
    const fpe_pose* poses;
    int B, nCycles;
    fpe_plan_out out;
    int recSlots;
};
// FPE_SEQ_RELOAD_ARGS: 0 never, 1 always, 2 (default) the 96-bit-row instantiations only — measured, round 6, A/B in one call, twice:
// cfg-5 (<2, 3>) 0.3060 -> 0.3017 ms and its 32 B of vector scratch gone; cfg-3 (<1, 2>) 0.6075 -> 0.6211 ms although three quarters of
// its leg search's spill reads disappear with it (see the leg loop): the lane reads were never what bound that kernel.
#ifndef FPE_SEQ_RELOAD_ARGS
#define FPE_SEQ_RELOAD_ARGS 2
#endif
#ifndef FPE_SEQ_GROUP16  // 0: always one pose per workgroup (measurement builds)
#define FPE_SEQ_GROUP16 1
#endif
// One pose's chain, from its stance to its last gait cycle: a FUNCTION the kernel calls once per wavefront, not inlined.  Round 6:
// as a callee the body reads everything uniform from the kernel's ARGUMENT SEGMENT (scalar loads through `kaIn`) instead of holding the
// arguments in scalar registers the allocator spills to vector lanes (no spilled scalars in the kernel, 60-150 before: cfg-3 0.6046 ->
// 0.5959 ms), and the kernel can put SIXTEEN poses in one workgroup (one workgroup per CU instead of sixteen: cfg-5 0.3023 -> 0.2959 ms;
// plan_bits_seq_kernel below).  A/B in one call, three repetitions: profiles/round6_seq_floor.txt.
template <int NRL, int KW, int kProd>
__device__ __attribute__((noinline)) void seq_run_pose(const SeqKernArgs __attribute__((address_space(4))) * kaIn, int slotOffIn, int bInV, int tid, unsigned hwidIn,
                                                       const LutHead& head) {
    constexpr int G = 64;
    constexpr int NR = G * NRL;
    // (a function's arguments arrive in VECTOR registers: the uniform ones go back to scalars here, or every address and index derived
    // from them would be vector arithmetic — and the argument-segment pointer could not feed scalar loads at all)
    typedef const SeqKernArgs __attribute__((address_space(4))) * KernArgPtrS;
    const unsigned long long kaBits = reinterpret_cast<unsigned long long>(kaIn);
    const KernArgPtrS kaArg = reinterpret_cast<KernArgPtrS>((static_cast<unsigned long long>(static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(kaBits >> 32)))) << 32) |
                                                              static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(kaBits))));
    const int slotOff = __builtin_amdgcn_readfirstlane(slotOffIn), b = __builtin_amdgcn_readfirstlane(bInV);
    const unsigned hwid = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(hwidIn)));
    const SeqKernArgs* kaG = (const SeqKernArgs*)kaArg;
    const DevMap& m = kaG->m;
    const BitMap& bm = kaG->bm;
    const PlanConsts& pc = kaG->pc;
    const SpiralLut& lut = kaG->lut;
    const fpe_pose* __restrict__ poses = kaG->poses;
    const int nCycles = kaG->nCycles, recSlots = kaG->recSlots;
    const fpe_plan_out out = specialise_products<kProd>(kaG->out);
    // (the workgroup's LDS by its own symbol: a pointer PARAMETER would be a generic one, and every LDS access a flat instruction)
    extern __shared__ __attribute__((aligned(16))) unsigned char smemAll[];
    unsigned char* const smem = smemAll + slotOff;
    const Grp<G> g(tid);
    PoseShared& sh = *reinterpret_cast<PoseShared*>(smem);
#ifdef FPE_TRACE
    if (tid < 4) sh.pad[tid] = 0;
#endif
    // per-leg constants of the pose, computed once (lane = leg) instead of once per leg and phase: a division and a
    // dependent rank-table load each
    LegStatic* lsTab = reinterpret_cast<LegStatic*>(smem + sizeof(PoseShared));
    constexpr size_t kLsBytes = (4 * sizeof(LegStatic) + 15) & ~static_cast<size_t>(15);
    // rows actually allocated: the window's 2 winH + 1 (not 64 * NRL) — LDS bounds the occupancy of these kernels
    const LegBits lb = make_legbits(smem + sizeof(PoseShared) + kLsBytes, min(2 * pc.winH + 1, NR), KW, pc.nHW, true);
    // staged output records: recSlots (a power of two, sized by the launch to keep the LDS within the occupancy budget)
    // cycles of four legs behind the row arrays
    using Rec = SeqRecOf<KW>;
    Rec* recBase = reinterpret_cast<Rec*>(
        smem + ((sizeof(PoseShared) + kLsBytes + 4 * static_cast<size_t>(legbits_words(min(2 * pc.winH + 1, NR), KW, pc.nHW, true)) + 15) & ~static_cast<size_t>(15)));
    const bool live = true;

    const fpe_pose* pp = poses + b;
    const double x0 = pp->position[0], y0 = pp->position[1], z0 = pp->position[2];
    const int gait = pp->gait;
    for (int k = tid; k < pc.nFoot; k += G) {
        sh.footDa[k] = pc.footDa[k];
        sh.footDb[k] = pc.footDb[k];
        sh.footOff[k] = 0;
    }
    // initial stance (cpp:350-378) and first-gait shift (setFirstGait, cpp:2679-2699): lane = leg
    if (tid < 4) {
        const int leg = tid;
        lsTab[leg] = make_leg_static(pc, pp, leg, m.g.res, lut);
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

    const int cycLag = (static_cast<int>(hwid & 3u) * nCycles) / 16;  // launch order of this wavefront on its SIMD (HW_ID.WAVE_ID: 0 oldest .. 3) x a sixteenth of the cycles
#ifdef FPE_TRACE
    long long flushClocks = 0, nFlushes = 0, legClocks = 0, nLegs = 0;
    stamp(pc, 5, 14);  // end of the prologue
#endif
    for (int cyc = 0; cyc < nCycles; ++cyc) {
        {
            // Issue priority by PROGRESS (s_setprio, four levels): the SIMD's arbiter serves the oldest wavefront first, so the
            // four poses of a SIMD finish one after the other and the last one runs alone at a third of the four-wavefront issue
            // rate (profiles/round3_residency.txt: lifetimes 1.2 / 1.4 / 1.6 / 1.9 M clocks by launch order).  A wavefront that
            // is behind gets the higher priority: the four advance together and finish together.  Measured: cfg-3 0.727 -> 0.640 ms,
            // cfg-5 0.375 -> 0.325 ms; the reverse mapping reproduces the default.  (The 8-lane kernels: headline neutral — its two
            // wavefronts per SIMD start and advance together anyway —, cfg-4 +3 %: new workgroups would starve the ones about to
            // finish; not used there.)
            // The levels change where a half, a quarter and an eighth of the cycles remain: wavefronts re-synchronise at every
            // boundary (the one ahead waits at the lower level), and the free run after the last boundary — oldest first again —
            // is the last eighth only.  Measured against four equal quarters: cfg-3 0.644 -> 0.625 ms, cfg-5 0.328 -> 0.325 ms;
            // boundaries per leg search instead of per cycle, later boundaries (1/4, 1/8, 1/16) and a rotating offset that
            // emulates sixteen levels were all slower.
            // (the younger wavefronts of the SIMD keep their level a little longer — cycLag, from the hardware wave slot = launch
            // order, see above the loop: within a level the arbiter serves the oldest first.  cfg-3 0.623 -> 0.608 ms, cfg-5 the
            // same; lags of 1 / 32, 3 / 32 and 4 / 32 of the cycles per slot: less or nothing)
            const int cycEff = max(cyc - cycLag, 0);
            const int rem8 = ((nCycles - cycEff) * 8 + nCycles - 1) / nCycles;  // remaining cycles in eighths, rounded up: 8 .. 1
            const int q = rem8 > 4 ? 0 : (rem8 > 2 ? 1 : (rem8 > 1 ? 2 : 3));
            if (q == 0) __builtin_amdgcn_s_setprio(3);
            else if (q == 1) __builtin_amdgcn_s_setprio(2);
            else if (q == 2) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
        bool cycleOk = true;
        for (int ph = 0; ph < nPhases; ++ph) {
            const unsigned mask = (gait == 1) ? (1u << ((walkOrder >> (2 * ph)) & 3)) : 0xFu;
            stamp(pc, cyc, 0);
            // feet-polygon centres: lane t computes track t (getPolygonCenter, cpp:2191, 2265)
            if (tid < 3) sh.ctr[tid] = polygon_center_x(sh.cur[tid]);
            pose_sync<16>();
            int allValid = 1;  // non-swing legs do not vote
            for (int leg = 0; leg < 4; ++leg) {
                if (!((mask >> leg) & 1u)) continue;
                const LegStatic ls = lsTab[leg];
                stamp(pc, cyc, 1);
#ifdef FPE_TRACE
                const long long tLeg0 = static_cast<long long>(__builtin_readcyclecounter());
#endif
                int legValid = 1;
                constexpr bool kReload = FPE_SEQ_RELOAD_ARGS == 1 || (FPE_SEQ_RELOAD_ARGS == 2 && KW >= 3);
                if constexpr (kReload) {
                // Round 6: the leg search reads the map's geometry, the plan constants, the table and output pointers from the
                // ARGUMENT SEGMENT again (scalar loads through a pointer the optimiser cannot see through: nothing is hoisted out
                // of the chain) instead of holding ~130 scalar registers of them across 128 leg searches.  The register allocator
                // had spilled those to lanes of three vector registers in the prologue and read them back with v_readlane inside
                // the leg search — 226 static lane reads of its 1 197 vector instructions in <1, 2, 0>, 276 of 1 529 in <2, 3, 0>
                // (profiles/round6_seq_floor.txt) — in kernels whose VECTOR unit is what is busy (0.86 of the SIMD's time at four
                // wavefronts).  With the reload 53 / 57 remain, the kernels hold 67 / 69 spilled scalars instead of 142 / 152 and
                // <2, 3, 0> no vector scratch — and the time says what those reads were worth: cfg-5 -1.3 %, cfg-3 +2 % (the scalar
                // loads' waits now sit INSIDE the leg search, in front of its first uses); hence the per-instantiation switch above.
                typedef const SeqKernArgs __attribute__((address_space(4))) * KernArgPtr;
                KernArgPtr ka4 = (KernArgPtr)kaArg;
                asm volatile("" : "+s"(ka4));
                const SeqKernArgs* ka = (const SeqKernArgs*)ka4;
                const fpe_plan_out outL = specialise_products<kProd>(ka->out);
                // (the LDS carve-up likewise: a few scalar operations on two of the constants instead of six held registers)
                const int rowsL = min(2 * ka->pc.winH + 1, NR);
                const LegBits lbL = make_legbits(smem + sizeof(PoseShared) + kLsBytes, rowsL, KW, ka->pc.nHW, true);
                Rec* const recL = reinterpret_cast<Rec*>(
                    smem + ((sizeof(PoseShared) + kLsBytes + 4 * static_cast<size_t>(legbits_words(rowsL, KW, ka->pc.nHW, true)) + 15) & ~static_cast<size_t>(15)));
                leg_phase_bits<G, NRL, KW, false, false>(ka->m, ka->bm, ka->pc, ka->lut, head, sh, lbL, g, leg, ls, y0, adjY, advance, cyc, ka->nCycles, b, live, outL,
                                                         nullptr, recL + 4 * (cyc & (ka->recSlots - 1)), &legValid);
                } else {
                leg_phase_bits<G, NRL, KW, false, false>(m, bm, pc, lut, head, sh, lb, g, leg, ls, y0, adjY, advance, cyc, nCycles, b, live, out, nullptr,
                                                         recBase + 4 * (cyc & (recSlots - 1)), &legValid);
                }
                allValid &= legValid;
                stamp(pc, cyc, 9);
#ifdef FPE_TRACE
                legClocks += static_cast<long long>(__builtin_readcyclecounter()) - tLeg0;
                ++nLegs;
#endif
            }
            pose_sync<16>();
            // footholdValidation_ = AND of the swing legs' flags (cpp:1323); commit or skip (cpp:1332-1576)
            const bool phaseOk = allValid != 0;
            if (phaseOk && tid < 24) {
                // x and y of the three tracks' next positions, straight from the staged record (its first six doubles: nominal,
                // centroid, default track); no later cycle reads a committed z (getPolygonCenter, cpp:2421-2463)
                const int leg = tid / 6, e = tid - leg * 6;
                if ((mask >> leg) & 1u) {
                    const double* rd = reinterpret_cast<const double*>(recBase + 4 * (cyc & (recSlots - 1)) + leg);
                    sh.cur[2 - (e >> 1)][leg][e & 1] = rd[e];
                }
            }
            pose_sync<16>();
            cycleOk = cycleOk && phaseOk;
            stamp(pc, cyc, 10);
        }
        if (tid == 0 && out.cycle_ok) out.cycle_ok[static_cast<size_t>(b) * nCycles + cyc] = cycleOk ? 1 : 0;
        adjY += pc.drift;  // cpp:1578
        {   // the staged records of the last recSlots cycles: lane = (cycle slot, leg)
            const int slot = cyc & (recSlots - 1);
            if (slot == recSlots - 1 || cyc == nCycles - 1) {
#ifdef FPE_TRACE
                const long long tFlush0 = static_cast<long long>(__builtin_readcyclecounter());
#endif
                pose_sync<16>();
                // (the flush reads its constants and pointers from the argument segment as well where the leg loop does)
                constexpr bool kReloadF = FPE_SEQ_RELOAD_ARGS == 1 || (FPE_SEQ_RELOAD_ARGS == 2 && KW >= 3);
                typedef const SeqKernArgs __attribute__((address_space(4))) * KernArgPtr;
                KernArgPtr kf4 = (KernArgPtr)kaArg;
                if constexpr (kReloadF) asm volatile("" : "+s"(kf4));
                const SeqKernArgs* kf = (const SeqKernArgs*)kf4;
                const DevMap& mF = kReloadF ? kf->m : m;
                const PlanConsts& pcF = kReloadF ? kf->pc : pc;
                const fpe_plan_out outF = kReloadF ? specialise_products<kProd>(kf->out) : out;
                const int nCycF = kReloadF ? kf->nCycles : nCycles, slotsF = kReloadF ? kf->recSlots : recSlots;
                Rec* const recF = kReloadF ? reinterpret_cast<Rec*>(smem + ((sizeof(PoseShared) + kLsBytes +
                                                                            4 * static_cast<size_t>(legbits_words(min(2 * pcF.winH + 1, NR), KW, pcF.nHW, true)) + 15) &
                                                                           ~static_cast<size_t>(15)))
                                           : recBase;
                if constexpr (KW <= kSeqDeferMaxKW) {  // deferred heights: two lanes per (cycle, leg) unit
                    const int un = tid >> 1, c = (cyc - slot) + (un >> 2);
                    if (un < 4 * slotsF && c <= cyc) flush_seqrec2(mF, pcF, sh.footDa, sh.footDb, recF[un], b, c, un & 3, tid & 1, nCycF, outF);
                } else {
                    const int s = tid >> 2, c = (cyc - slot) + s;
                    if (tid < 4 * slotsF && c <= cyc) flush_seqrec(mF, pcF, sh.footDa, sh.footDb, recF[tid], b, c, tid & 3, nCycF, outF);
                }
                pose_sync<16>();  // the slots are rewritten next
#ifdef FPE_TRACE
                __builtin_amdgcn_s_waitcnt(0);
                flushClocks += static_cast<long long>(__builtin_readcyclecounter()) - tFlush0;
                ++nFlushes;
#endif
            }
        }
    }
    stamp(pc, 6, 15);
#ifdef FPE_TRACE
    stamp_value(pc, 7, 14, flushClocks);  // clocks inside the flushes (records + deferred heights) and their number
    stamp_value(pc, 7, 15, nFlushes);
    stamp_value(pc, 5, 15, legClocks);    // clocks inside the leg searches of ALL cycles, and their number
    stamp_value(pc, 1, 14, nLegs);
    stamp_value(pc, 6, 13, static_cast<long long>(sh.pad[0]) << 4);  // clocks inside the spiral search,
    stamp_value(pc, 6, 12, sh.pad[1]);                               // searches, and searches without a hit
    stamp_value(pc, 6, 11, sh.pad[2]);
#endif
}

// The kernel: kGroup wavefronts — poses — per workgroup, each runs seq_run_pose on its own slot of the workgroup's LDS.  kGroup 16 (one
// workgroup of 1 024 threads per CU; the launch's choice for batches of at least 64 poses on the 96-bit-row windows) or 1.
template <int NRL, int KW, int kProd, int kGroup>
__global__ __launch_bounds__(64 * kGroup, kGroup == 1 ? FPE_SEQ_WAVES : 1) void plan_bits_seq_kernel(DevMap m, BitMap bm, PlanConsts pc, SpiralLut lut,
                                                                                                const fpe_pose* __restrict__ poses, int B, int nCycles, fpe_plan_out outArg,
                                                                                                int recSlots, int slotBytes) {
    stamp(pc, 6, 14);  // (profiling builds: lifetime of the wavefront, with the stamp after the cycle loop)
    const int tid = static_cast<int>(threadIdx.x) & 63, wv = static_cast<int>(threadIdx.x) >> 6;
    const int b = static_cast<int>(blockIdx.x) * kGroup + wv;
    if (b >= B) return;
    const Grp<64> g(tid);
    const LutHead head = load_lut_head(lut, g);
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    (void)m; (void)bm; (void)poses; (void)nCycles; (void)outArg; (void)recSlots;
    typedef const SeqKernArgs __attribute__((address_space(4))) * KernArgPtr0;
    seq_run_pose<NRL, KW, kProd>((KernArgPtr0)__builtin_amdgcn_kernarg_segment_ptr(), wv * slotBytes, b, tid, hwid, head);
}

// ---- host side of the bit-window path --------------------------------------------------------------------------
size_t bitmap_words(int rows, int cols, int* strideW, int* nw) {
    *nw = (cols + 31) / 32;
    *strideW = *nw + 2 * kBitPadW;
#if FPE_BITS_TILED
    return static_cast<size_t>(bit_row_groups(rows)) * 8 * (*strideW) * 4;  // 4-byte units (4 planes per word group, 8 rows per tile)
#else
    return static_cast<size_t>(rows + 2) * (*strideW) * 4;  // 4-byte units (4 planes per word group)
#endif
}

hipError_t launch_build_bitmap(const float* d_trav, int rows, int cols, float thrDefault, float thrCandidate, uint32_t* d_words,
                               hipStream_t stream) {
    int strideW, nw;
    const size_t units = bitmap_words(rows, cols, &strideW, &nw);
    hipError_t e = hipMemsetAsync(d_words, 0, units * 4, stream);  // padding rows / word groups; recycled buffers are dirty
    if (e != hipSuccess) return e;
    dim3 grid((cols + 255) / 256, rows);
    hipLaunchKernelGGL(build_bitmap_kernel, grid, dim3(256), 0, stream, d_trav, rows, cols, thrDefault, thrCandidate,
                       reinterpret_cast<uint4*>(d_words), strideW, nw);
    return hipGetLastError();
}

// Kernel shape for a window half-width: 8 lanes per leg with 2-4 rows per lane (windows of up to 32 rows / columns);
// one wavefront per pose with 64-bit rows (up to 64 rows) or 96-bit rows (up to 96 columns, 2 rows per lane).
struct BitsShape {
    int lanes;  // 8 or 64; 0 = no instantiation fits
    int nrl, kw;
};
static BitsShape bits_shape(int winH) {
    const int side = 2 * winH + 1;
    if (winH <= 0) return {0, 0, 0};
    if (side <= 16) return {8, 2, 1};
    if (side <= 24) return {8, 3, 1};
    if (side <= 32) return {8, 4, 1};
    if (side <= 64) return {64, 1, 2};
    if (side <= 96) return {64, 2, 3};
    return {0, 0, 0};
}

bool bits_supported(const PlanConsts& pc, const MapGeom& g) {
    if (pc.noBits != 0 || pc.winH <= 0) return false;
    // 32-bit offsets into layers and planes (load_cell / load_group), 24-bit multiplies of rows and strides
    if (g.rows >= (1 << 24) - 2 || g.cols >= (1 << 24) - 64) return false;
    if (static_cast<double>(g.rows) * g.cols * 4.0 >= 2147483648.0 - 65536.0) return false;
    if ((static_cast<double>(g.rows) + 16.0) * ((g.cols + 31) / 32 + 2 * kBitPadW) * 16.0 >= 2147483648.0) return false;
    const BitsShape sp = bits_shape(pc.winH);
    if (sp.lanes == 0) return false;
    if (pc.groupOverride != 0 && pc.groupOverride != (sp.lanes == 8 ? 8 : 65)) return false;
    if (pc.nFoot > (sp.lanes == 8 ? kDiscRounds * 8 : 64)) return false;  // the offset table is walked one entry per lane
    // the per-leg LDS (3 row arrays) doubles as float scratch of a direct disc pass over a CircleIterator
    // bounding box of up to (2 ceil(rf / res) + 2)^2 cells
    const double side = 2.0 * ceil(pc.rf / g.res) + 2.0;
    if (sp.lanes == 64) return side * side <= kBitsMaxBoxCells;  // 64-lane kernels: membership in two rounds of 64 cells (SeqRec::visA / visB)
    return side * side <= legbits_words(8 * sp.nrl, 1, pc.nHW, false);
}

// Which kernel a chained plan with these constants launches (evidence for bench.py / profiles).
void describe_plan_kernel(const PlanConsts& pc, const MapGeom& g, char* buf, size_t n) {
    if (bits_supported(pc, g)) {
        const BitsShape sp = bits_shape(pc.winH);
        const bool mid = mid_variant(pc, g.res) && pc.nFoot == 1;
        if (sp.lanes == 8)
            snprintf(buf, n, "plan_bits_kernel<%d, %s> (8 lanes per leg, %d x %d bit window%s)", sp.nrl, mid ? "true" : "false",
                     2 * pc.winH + 1, 2 * pc.winH + 1, mid ? ", 3x3-only fast path" : "");
        else
            snprintf(buf, n, "plan_bits_seq_kernel<%d, %d> (one wavefront per pose, %d x %d bit window, %d-bit rows)", sp.nrl, sp.kw,
                     2 * pc.winH + 1, 2 * pc.winH + 1, 32 * sp.kw);
        return;
    }
    const int G = plan_group_size(pc);
    if (G == 65) snprintf(buf, n, "plan_sequential_kernel (direct, one wavefront per pose)");
    else snprintf(buf, n, "plan_chained_kernel<%d, %s> (direct)", G, (G == 8 && mid_variant(pc, g.res)) ? "true" : "false");
}

hipError_t launch_plan_bits(const DevMap& m, const BitMap& bm, const PlanConsts& pc, const SpiralLut& lut, const fpe_pose* d_poses,
                            int B, int nCycles, const fpe_plan_out& d_out, hipStream_t stream) {
    const BitsShape sp = bits_shape(pc.winH);
    const bool mid = mid_variant(pc, m.g.res) && pc.nFoot == 1;  // rf < res: the candidate disc is the candidate's own cell
    const dim3 block(64);
    const int prod = product_shape(d_out);
#define FPE_LAUNCH_BITS_P(NRL, MID, PROD)                                                                                    \
    hipLaunchKernelGGL((plan_bits_kernel<NRL, MID, PROD>), dim3((B + 1) / 2), block,                                        \
                       2 * (sizeof(PoseShared) + 16 * legbits_words(8 * NRL, 1, pc.nHW, false) +                                      \
                            (MID ? (sizeof(YEntry) + sizeof(Unit)) * 32 : (sizeof(YEntry) + sizeof(UnitG)) * 16)), stream, d_poses, B,  \
                       nCycles, m, bm, pc, lut, d_out)
#define FPE_LAUNCH_BITS(NRL, MID)                                  \
    do {                                                           \
        if (prod == 2) FPE_LAUNCH_BITS_P(NRL, MID, 2);             \
        else if (prod == 1) FPE_LAUNCH_BITS_P(NRL, MID, 1);        \
        else FPE_LAUNCH_BITS_P(NRL, MID, 0);                       \
    } while (0)
#define FPE_LAUNCH_BITS_SEQ_G(NRL, KW, GRP, GRID, LDS, SLOT)                                                                                      \
    do {                                                                                                                                        \
        if ((LDS) > 64 * 1024) {                                                                                                                \
            const hipError_t ea = hipFuncSetAttribute(prod == 1 ? reinterpret_cast<const void*>(plan_bits_seq_kernel<NRL, KW, 1, GRP>)          \
                                                                : reinterpret_cast<const void*>(plan_bits_seq_kernel<NRL, KW, 0, GRP>),        \
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(LDS));                       \
            if (ea != hipSuccess) return ea;                                                                                                    \
        }                                                                                                                                       \
        if (prod == 1)                                                                                                                          \
            hipLaunchKernelGGL((plan_bits_seq_kernel<NRL, KW, 1, GRP>), dim3(GRID), dim3(64 * GRP), LDS, stream, m, bm, pc, lut, d_poses, B,    \
                               nCycles, d_out, recSlots, static_cast<int>(SLOT));                                                              \
        else                                                                                                                                    \
            hipLaunchKernelGGL((plan_bits_seq_kernel<NRL, KW, 0, GRP>), dim3(GRID), dim3(64 * GRP), LDS, stream, m, bm, pc, lut, d_poses, B,    \
                               nCycles, d_out, recSlots, static_cast<int>(SLOT));                                                              \
    } while (0)
#define FPE_LAUNCH_BITS_SEQ(NRL, KW)                                                                                         \
    do {                                                                                                                     \
        const size_t base = (sizeof(PoseShared) + ((4 * sizeof(LegStatic) + 15) & ~static_cast<size_t>(15)) +                            \
                             4 * legbits_words(2 * pc.winH + 1 < 64 * NRL ? 2 * pc.winH + 1 : 64 * NRL, KW, pc.nHW, true) + 15) &        \
                            ~static_cast<size_t>(15);                                                                                 \
        int recSlots = 8; /* cycles of staged records: as many as keep sixteen poses per CU (10 KiB each) */                       \
        while (recSlots > 1 && base + recSlots * 4 * sizeof(SeqRecOf<KW>) > 10240) recSlots >>= 1;                                   \
        const size_t slot = (base + recSlots * 4 * sizeof(SeqRecOf<KW>) + 15) & ~static_cast<size_t>(15);                            \
        /* (the all-seven shape takes the generic instantiation here: compiled on its own it spills more — cfg-5 +2 %, cfg-3 0) */     \
        /* sixteen poses per workgroup — one workgroup per CU — where it was measured to pay: the 96-bit-row windows (cfg-5 -2 %; the    \
           64-bit-row kernel of cfg-3 +2 %), batches that fill at least four CUs, slots that fit sixteen times into the LDS */            \
        if (FPE_SEQ_GROUP16 && KW >= 3 && B >= 64 && 16 * slot <= 160 * 1024)                                                            \
            FPE_LAUNCH_BITS_SEQ_G(NRL, KW, 16, (B + 15) / 16, 16 * slot, slot);                                                          \
        else                                                                                                                             \
            FPE_LAUNCH_BITS_SEQ_G(NRL, KW, 1, B, slot, slot);                                                                            \
    } while (0)
    if (sp.lanes == 8) {
        if (sp.nrl == 2 && mid) FPE_LAUNCH_BITS(2, true);
        else if (sp.nrl == 2) FPE_LAUNCH_BITS(2, false);
        else if (sp.nrl == 3 && mid) FPE_LAUNCH_BITS(3, true);
        else if (sp.nrl == 3) FPE_LAUNCH_BITS(3, false);
        else if (sp.nrl == 4 && mid) FPE_LAUNCH_BITS(4, true);
        else FPE_LAUNCH_BITS(4, false);
    } else if (sp.lanes == 64) {
        if (sp.kw == 2) FPE_LAUNCH_BITS_SEQ(1, 2);
        else FPE_LAUNCH_BITS_SEQ(2, 3);
    } else {
        return hipErrorInvalidValue;
    }
#undef FPE_LAUNCH_BITS
#undef FPE_LAUNCH_BITS_P
#undef FPE_LAUNCH_BITS_SEQ
#undef FPE_LAUNCH_BITS_SEQ_G
    return hipGetLastError();
}
