// fpe_opt.hpp — part three of the kernel translation unit (included at the end of fpe_kernels.hip, inside namespace
// fpe): the OPT TRACK of globalFootholdPlan.
//
// Reference: cpp:54-88 nloptFunc, cpp:92-148 nloptConstraint1..8, cpp:913-1319 the per-cycle driver, cpp:1485-1568 the
// commit, cpp:2307-2408 getGaitCycleSearchGridMap, cpp:2557-2568 getMapIndex.  One workgroup per pose — one wavefront
// for batches, eight (seven of them only helping the optimiser's search) for a handful of poses — gait cycles in
// sequence (the track's feet chain like the other tracks'); inside a cycle
//   * the gait-cycle submap gaitMap_ = gridmap_.getSubmap(next feet centre, isos_.length x isos_.width) as a MapGeom
//     of its own plus its top-left index in the map (no copy: cells are read from the map's layers);
//   * checkFootholdUseCentroidMethod ON gaitMap_ for the four legs side by side, 16 lanes per leg (lane = row of the
//     leg's rectangle), with traversableBeginRow / traversableEndRow (cpp:1608-1609);
//   * the BUILD-DEFINED optimiser (NLopt is absent and unpinned — include/fpe.h): the literal objective and
//     constraints evaluated on every integer point of the box, lane = point, wave-wide lexicographic minimum of
//     (violation, objective, enumeration order) — the same rule, expression for expression, as
//     oracle/fpo_opt.cpp::solveLattice;
//   * positions and mean heights taken from gaitMap_ (its own geometry: a CircleIterator on a submap is clamped to the
//     submap and measures distances to the SUBMAP's cell centres), the commit with the nominal track's validity.
// Everything is the reference's f64 expression order (-ffp-contract=off); this is a secondary product, written for
// exactness first: the hot path is fpe_bits.hpp.
#pragma once

namespace {

// GridMap::getSubmap(position, length) as geometry only: the submap's own MapGeom (setGeometry(SubmapGeometry): size,
// length = size * res, position = top-left corner - length / 2) and its top-left index.  ok = isSuccess.
struct SubGeom {
    MapGeom g;
    int i0, j0;
    bool ok;
};
__device__ __forceinline__ SubGeom opt_submap(const MapGeom& g, double px, double py, double lx, double ly) {
    SubGeom r;
    const Submap s = submap_info(g, px, py, lx, ly);
    r.ok = s.ok;
    r.i0 = s.i0;
    r.j0 = s.j0;
    const int ni = s.ok ? s.ni : 1, nj = s.ok ? s.nj : 1;
    const double cornerX = cell_pos(g.baseX, g.res, s.i0) - (-(0.5 * g.res));
    const double cornerY = cell_pos(g.baseY, g.res, s.j0) - (-(0.5 * g.res));
    const double subLenX = static_cast<double>(ni) * g.res, subLenY = static_cast<double>(nj) * g.res;
    r.g = make_geom(ni, nj, g.res, cornerX - 0.5 * subLenX, cornerY - 0.5 * subLenY);
    return r;
}

// getFootholdMeanHeight (cpp:2520-2554) on a map given by its geometry `g` whose cell (i, j) is cell (offI + i, offJ + j)
// of the elevation layer.  The 16 lanes of a leg fetch the bounding box's cells side by side (one memory round trip
// instead of one per cell) and leave (value, visited) in the leg's LDS scratch; lane 0 then runs the reference's f32
// sum over them in CircleIterator order (row-major over the bounding box).  Boxes beyond the scratch (kOptBoxCells
// cells) are walked by lane 0 alone, cell by cell.  The result is valid on lane 0 of the group.
constexpr int kOptBoxCells = 128;
__device__ float opt_mean_height(const Grp<16>& g, float* vals, const MapGeom& mg, const float* elev, int ld, int offI, int offJ, double cx,
                                 double cy, double rf, double rf2, double h) {
    float iHeight = 0.0f, meanHeight = 0.0f;
    int n = 0;
    if (!centre_usable(cx, cy)) return static_cast<float>(meanHeight + h);
    const BBox bb = circle_bbox(mg, cx, cy, rf);
    const int nb = (bb.ni > 0 && bb.nj > 0) ? bb.ni * bb.nj : 0;
    if (nb <= kOptBoxCells) {
        const float njInv = rcp_small(max(bb.nj, 1));
        for (int t = g.sub; t < nb; t += 16) {
            int a, b;
            divmod_small(t, bb.nj, njInv, a, b);
            const int i = bb.i0 + a, j = bb.j0 + b;
            const bool vis = in_range(i, j, mg.rows, mg.cols) && cell_in_disc(mg, i, j, cx, cy, rf2);
            float e = 0.0f;
            if (vis) e = elev[static_cast<size_t>(offI + i) * ld + (offJ + j)];
            // (a visited value is stored finite-or-zero as the reference reads it, cpp:2532-2537; NaN marks "not visited")
            vals[t] = vis ? (__builtin_isfinite(e) ? e : 0.0f) : __builtin_nanf("");
        }
        pose_sync<16>();
        if (g.sub == 0) {
            for (int t = 0; t < nb; ++t) {
                const float v = vals[t];
                if (v != v) continue;
                iHeight = v;
                if (iHeight < 10) {  // cpp:2539
                    n++;
                    meanHeight = meanHeight + iHeight;
                }
            }
        }
        pose_sync<16>();
    } else if (g.sub == 0) {
        for (int a = 0; a < bb.ni; ++a)
            for (int b = 0; b < bb.nj; ++b) {
                const int i = bb.i0 + a, j = bb.j0 + b;
                if (!in_range(i, j, mg.rows, mg.cols) || !cell_in_disc(mg, i, j, cx, cy, rf2)) continue;
                const float e = elev[static_cast<size_t>(offI + i) * ld + (offJ + j)];
                iHeight = __builtin_isfinite(e) ? e : 0.0f;
                if (iHeight < 10) {
                    n++;
                    meanHeight = meanHeight + iHeight;
                }
            }
    }
    if (n != 0) meanHeight = meanHeight / n;
    else meanHeight = iHeight;
    return static_cast<float>(meanHeight + h);
}

// nloptFunc, cpp:54-88 — the reference's expression, term for term (abs = std::abs(double): fabs).
// (the two quotients lengthBase/mapResolution and 2*skew/mapResolution are loop invariants of the search: evaluated once
// per call by the host — same operands, same division, same f64 values — OptConsts::lbOverRes / skew2OverRes)
__device__ __forceinline__ double opt_objective(const double (&x)[8], const OptConsts& oc, const int (&nominalIndex)[8],
                                                const int (&centroidIndex)[8], double lfCurrentRow, double rhCurrentRow) {
    const double w1 = oc.w1, w2 = oc.w2, w3 = oc.w3, w4 = oc.w4, wr = oc.wr, wc = oc.wc;
#define abs fabs
#define FPE_LB_OVER_RES oc.lbOverRes       /* lengthBase/mapResolution */
#define FPE_SKEW2_OVER_RES oc.skew2OverRes /* 2*skew/mapResolution */
    return (
            w1*( wr*(abs(x[0]-nominalIndex[0])) + wc*(abs(x[1]-nominalIndex[1])) +
                 wr*(abs(x[2]-nominalIndex[2])) + wc*(abs(x[3]-nominalIndex[3])) +
                 wr*(abs(x[4]-nominalIndex[4])) + wc*(abs(x[5]-nominalIndex[5])) +
                 wr*(abs(x[6]-nominalIndex[6])) + wc*(abs(x[7]-nominalIndex[7])) ) +
            w2*( wr*(abs(x[0]-centroidIndex[0])) + wc*(abs(x[1]-centroidIndex[1])) +
                 wr*(abs(x[2]-centroidIndex[2])) + wc*(abs(x[3]-centroidIndex[3])) +
                 wr*(abs(x[4]-centroidIndex[4])) + wc*(abs(x[5]-centroidIndex[5])) +
                 wr*(abs(x[6]-centroidIndex[6])) + wc*(abs(x[7]-centroidIndex[7])) ) +
            w3*( abs(abs(x[0]-x[2]) - FPE_LB_OVER_RES) +
                 abs(abs(x[4]-x[6]) - FPE_LB_OVER_RES) ) +
            w4*( abs(abs(0.5*abs(x[0]-x[2]) - 0.5*abs(x[4]-x[6])) - FPE_SKEW2_OVER_RES) +
                 abs(abs(0.5*abs(x[4]-x[6]) - 0.5*abs(lfCurrentRow - rhCurrentRow)) - FPE_SKEW2_OVER_RES) )
            );
#undef abs
#undef FPE_LB_OVER_RES
#undef FPE_SKEW2_OVER_RES
}

// nloptConstraint1..8, cpp:92-148, folded into (every value <= ctol, largest value) — solveLattice's key.
__device__ __forceinline__ double opt_violation(const double (&x)[8], const OptConsts& oc, double lfCurrentRow, double rhCurrentRow) {
    const double t1 = oc.t1, t2 = oc.t2, t3 = oc.t3, t4 = oc.t4;
#define abs fabs
    const double c[8] = {
        ( t1 - abs(x[0] - x[2]) ),
        ( abs(x[0] - x[2]) - t2 ),
        ( t1 - abs(x[4] - x[6]) ),
        ( abs(x[4] - x[6]) - t2 ),
        ( t3 - 0.5*abs( abs(x[0] - x[2]) - abs(x[4] - x[6]) ) ),
        ( 0.5*abs( abs(x[0] - x[2]) - abs(x[4] - x[6]) ) - t4 ),
        ( t3 - 0.5*abs( abs(x[4] - x[6]) - abs(lfCurrentRow - rhCurrentRow) ) ),
        ( 0.5*abs( abs(x[4] - x[6]) - abs(lfCurrentRow - rhCurrentRow) ) - t4 ),
    };
#undef abs
    bool feasible = true;
    double resmax = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        feasible = feasible && c[q] <= oc.ctol;
        resmax = c[q] > resmax ? c[q] : resmax;
    }
    return feasible ? 0.0 : resmax;
}

// Wave-wide reductions on 32-bit words by DPP row shifts and row broadcasts (gfx9: row_shr 1 / 2 / 4 / 8 inside a row of 16, then
// row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2 and 3): six VALU operations and a v_readlane where the butterfly of
// ds_bpermute exchanges takes six dependent LDS-crossbar round trips (~0.2 us each way on a lone wavefront — the opt track's
// chain is ONE wavefront's latency, stage after stage).  Lanes without a source keep the operation's identity.
#ifndef FPE_OPT_NO_DPP
#define FPE_DPP_STEP(op, ident, v, ctrl, rmask) v = op(v, static_cast<unsigned>(__builtin_amdgcn_update_dpp(static_cast<int>(ident), static_cast<int>(v), ctrl, rmask, 0xF, false)))
__device__ __forceinline__ unsigned dpp_min_op(unsigned a, unsigned b) { return a < b ? a : b; }
__device__ __forceinline__ unsigned dpp_or_op(unsigned a, unsigned b) { return a | b; }
__device__ __forceinline__ unsigned dpp_add_op(unsigned a, unsigned b) { return a + b; }
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
    FPE_DPP_STEP(dpp_min_op, 0xFFFFFFFFu, v, 0x111, 0xF);
    FPE_DPP_STEP(dpp_min_op, 0xFFFFFFFFu, v, 0x112, 0xF);
    FPE_DPP_STEP(dpp_min_op, 0xFFFFFFFFu, v, 0x114, 0xF);
    FPE_DPP_STEP(dpp_min_op, 0xFFFFFFFFu, v, 0x118, 0xF);
    FPE_DPP_STEP(dpp_min_op, 0xFFFFFFFFu, v, 0x142, 0xA);
    FPE_DPP_STEP(dpp_min_op, 0xFFFFFFFFu, v, 0x143, 0xC);
    return static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(v), 63));
}
__device__ __forceinline__ unsigned wave_or_u32(unsigned v) {
    FPE_DPP_STEP(dpp_or_op, 0u, v, 0x111, 0xF);
    FPE_DPP_STEP(dpp_or_op, 0u, v, 0x112, 0xF);
    FPE_DPP_STEP(dpp_or_op, 0u, v, 0x114, 0xF);
    FPE_DPP_STEP(dpp_or_op, 0u, v, 0x118, 0xF);
    FPE_DPP_STEP(dpp_or_op, 0u, v, 0x142, 0xA);
    FPE_DPP_STEP(dpp_or_op, 0u, v, 0x143, 0xC);
    return static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(v), 63));
}
__device__ __forceinline__ int wave_inclusive_sum(int x) {  // every lane: the sum of the lanes up to and including itself
    unsigned v = static_cast<unsigned>(x);
    FPE_DPP_STEP(dpp_add_op, 0u, v, 0x111, 0xF);
    FPE_DPP_STEP(dpp_add_op, 0u, v, 0x112, 0xF);
    FPE_DPP_STEP(dpp_add_op, 0u, v, 0x114, 0xF);
    FPE_DPP_STEP(dpp_add_op, 0u, v, 0x118, 0xF);
    FPE_DPP_STEP(dpp_add_op, 0u, v, 0x142, 0xA);
    FPE_DPP_STEP(dpp_add_op, 0u, v, 0x143, 0xC);
    return static_cast<int>(v);
}
#undef FPE_DPP_STEP
#else  // the same by ds_bpermute exchanges (measurement builds)
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
    for (int off = 32; off >= 1; off >>= 1) {
        const unsigned o = __shfl_xor(v, off);
        v = o < v ? o : v;
    }
    return v;
}
__device__ __forceinline__ unsigned wave_or_u32(unsigned v) {
    for (int off = 32; off >= 1; off >>= 1) v |= __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ int wave_inclusive_sum(int x) {
    const int lane = static_cast<int>(threadIdx.x) & 63;
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(x, off);
        if (lane >= off) x += o;
    }
    return x;
}
#endif
// wave-wide minimum of a NON-NEGATIVE double (violations, objectives: sums of absolute values; +inf allowed): such doubles order
// like their bit patterns — the high words' minimum, then the low words' among the lanes that hold it
__device__ __forceinline__ double wave_min_nonneg(double v) {
    const unsigned hi = static_cast<unsigned>(__double2hiint(v)), lo = static_cast<unsigned>(__double2loint(v));
    const unsigned mh = wave_min_u32(hi);
    const unsigned ml = wave_min_u32(hi == mh ? lo : 0xFFFFFFFFu);
    return __hiloint2double(static_cast<int>(mh), static_cast<int>(ml));
}
// wave-wide lexicographic minimum of (key, f, t), key and f non-negative: every lane ends with the winner.  Word by word: the
// minimum of a word among the lanes still level on the words before it.
__device__ __forceinline__ void opt_wave_min(double& key, double& f, unsigned& t) {
    const unsigned w[5] = {static_cast<unsigned>(__double2hiint(key)), static_cast<unsigned>(__double2loint(key)),
                           static_cast<unsigned>(__double2hiint(f)), static_cast<unsigned>(__double2loint(f)), t};
    unsigned m[5];
    bool level = true;
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        m[q] = wave_min_u32(level ? w[q] : 0xFFFFFFFFu);
        level = level && w[q] == m[q];
    }
    key = __hiloint2double(static_cast<int>(m[0]), static_cast<int>(m[1]));
    f = __hiloint2double(static_cast<int>(m[2]), static_cast<int>(m[3]));
    t = m[4];
}

// One candidate of the cross-wavefront minimum (W > 1: the waves of a pose's workgroup each search a slice of the box)
struct OptBest {
    double key, f;
    unsigned t, pad;
};
// The optimiser's problem of one cycle as wavefront 0 publishes it to the helper wavefronts (W > 1).
struct OptProblem {
    int pad[2];
    int nIdx[8], cIdx[8], lo[8], up[8];
    double x[8];  // start point with the columns already decided
    double lfRow, rhRow;
};

// Index splits of the row lattice.  Its enumeration runs to kMaxLatticePoints = 2^24 and a pair index (ab, cd) to the same
// when the other pair is a single point — past the range divmod_small is proven for (t < 2^22: the f32 estimate within one
// of the quotient).  Larger indices take the exact division; the test is uniform wherever the index is.
__device__ __forceinline__ void divmod_lattice(int t, int d, float dinv, int& q, int& r) {
    if (t < (1 << 22)) {
        divmod_small(t, d, dinv, q, r);
    } else {
        q = static_cast<int>(static_cast<unsigned>(t) / static_cast<unsigned>(d));
        r = t - q * d;
    }
}

// oracle/fpo_opt.cpp::solveLattice, first part (one wavefront): start x = x0 = centroidIndex (cpp:1180-1183), NLopt's
// precondition (status 1) and the size of the row lattice (status 3) — both known from the bounds alone, before any search:
// -1 when the row search is to run (the helper wavefronts can be told at once), else the final status.
__device__ int opt_solve_check(const OptConsts& oc, const int (&nIdx)[8], const int (&cIdx)[8], const int (&lo)[8], const int (&up)[8],
                               double lfRow, double rhRow, double (&x)[8], double& minf) {
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = cIdx[k];
    minf = 0.0;
    bool bad = false;
#pragma unroll
    for (int k = 0; k < 8; ++k) bad = bad || lo[k] > up[k] || x[k] < lo[k] || x[k] > up[k];
    if (bad) {
        minf = opt_objective(x, oc, nIdx, cIdx, lfRow, rhRow);
        return 1;
    }
    const double points = static_cast<double>(up[0] - lo[0] + 1) * static_cast<double>(up[2] - lo[2] + 1) *
                          static_cast<double>(up[4] - lo[4] + 1) * static_cast<double>(up[6] - lo[6] + 1);
    if (points > static_cast<double>(kMaxLatticePoints)) return 3;
    return -1;
}
// ... the four column variables x[1], x[3], x[5], x[7] (statuses -1 and 3): each the minimiser of the objective over its
// interval with the others where they stand, smallest integer on ties; one after the other.
__device__ void opt_solve_columns(const OptConsts& oc, const int (&nIdx)[8], const int (&cIdx)[8], const int (&lo)[8], const int (&up)[8],
                                  double lfRow, double rhRow, int lane, double (&x)[8]) {
    const double inf = __builtin_huge_val();
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        constexpr int kk[4] = {1, 3, 5, 7};
        const int k = kk[c];
        double key = 0.0, bestF = inf;
        unsigned bestV = 0xFFFFFFFFu;
        for (int v = lo[k] + lane; v <= up[k]; v += 64) {
            x[k] = v;
            const double f = opt_objective(x, oc, nIdx, cIdx, lfRow, rhRow);
            if (f < bestF) {
                bestF = f;
                bestV = static_cast<unsigned>(v - lo[k]);
            }
        }
        opt_wave_min(key, bestF, bestV);
        x[k] = lo[k] + static_cast<int>(bestV);
    }
}

__device__ OptBest opt_solve_rows_plain(const OptConsts& oc, const int (&nIdx)[8], const int (&cIdx)[8], const int (&lo)[8], const int (&up)[8],
                                        double lfRow, double rhRow, int lane, int slice, int nSearch, const double (&x)[8]);
#ifdef FPE_OPT_TRACE  // measurement builds only (profiles/collect_opt_trace.sh): wall-clock stamps of workgroup 0's stages, per gait cycle
__device__ unsigned long long g_optTrace[256][16];
#define FPE_OPT_STAMP(k) do { if (blockIdx.x == 0 && lane == 0 && cyc < 256) g_optTrace[cyc][k] = wall_clock64(); } while (0)
#else
#define FPE_OPT_STAMP(k) do { } while (0)
#endif
constexpr int kOptListCap = 1024;  // surviving points a wavefront lists (LDS, 4 bytes each); more: every point is searched
// LDS of the listed search: the lists (one per searching wavefront) and, for W > 1, what the wavefronts tell each other about
// their share of the Dab values (per lane and (c, d) slot: smallest violation, mask of the values that attain it)
template <int W>
struct OptListLds {
    unsigned list[W][W == 1 ? kOptListCap : kOptListCap / 4];
    double partMin[W == 1 ? 1 : W][W == 1 ? 1 : 2][W == 1 ? 1 : 64];
    unsigned long long partGood[W == 1 ? 1 : W][W == 1 ? 1 : 2][W == 1 ? 1 : 64];
};
// Wavefronts of one workgroup meeting WITHOUT the workgroup barrier (round 5): a counter in LDS every arriving wavefront adds
// one to, and a wait for it to reach a target that only grows (nothing is ever reset: no race on re-use).  The waves of a
// workgroup are resident together, so the wait cannot starve what it waits for; s_sleep keeps the poll off the SIMD's issue
// slots.  Release before the add / acquire after the wait order the LDS stores around it.
__device__ __forceinline__ void opt_arrive(int* cnt, int lane) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void opt_publish(int* seq, int value, int lane) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_store(seq, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void opt_wait_ge(int* cnt, int target) {
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// What a wavefront carries from the row search's first part (the bounds and the rows: everything but the objective) to its
// second (the columns decided).
struct OptRowsState {
    int mode;   // 0: the listed search (count entries in the wavefront's list); 1: every point, (c, d) pairs in the lanes' slots; 2: the plain loop
    int count;
    double minKey;
    bool live[2], feasCD[2];
    double A4n[2], A4c[2], A6n[2], A6c[2], Dcd[2], Ucd[2], hcd[2], V2[2], rmCD[2];
};
// First part of the row search: nothing here reads the column variables, so the helper wavefronts of a W > 1 workgroup run it
// WHILE wavefront 0 decides the columns (round 5; before, they waited for the columns and the whole search followed them).
// slice / nSearch: this wavefront's share (k = slice, slice + nSearch, ...) among the nSearch searching wavefronts;
// meet / meets: their meeting point for the exchange of the Dab shares (nSearch > 1) and how often this wavefront has met there.
template <int W>
__device__ __forceinline__ void opt_rows_prepare(const OptConsts& oc, const int (&nIdx)[8], const int (&cIdx)[8], const int (&lo)[8], const int (&up)[8], double lfRow,
                                 double rhRow, int lane, int slice, int nSearch, OptListLds<W>* ll, int* meet, int& meets, OptRowsState& st, int cyc = 0) {
    const double inf = __builtin_huge_val();
    const int n0 = up[0] - lo[0] + 1, n2 = up[2] - lo[2] + 1, n4 = up[4] - lo[4] + 1, n6 = up[6] - lo[6] + 1;
    const int nAB = n0 * n2, nCD = n4 * n6;
    constexpr int kSlots = 2;
    st.count = 0;
    st.minKey = 0.0;
    if (nCD > 64 * kSlots) {
        st.mode = 2;  // (uniform over the workgroup — every wavefront is given the same problem — like every early return below: nobody is left waiting at the exchange)
        return;
    }
    st.mode = 1;
    const double wr = oc.wr;
    const double L = oc.lbOverRes, K = oc.skew2OverRes;  // lengthBase/mapResolution, 2*skew/mapResolution
    const double t1 = oc.t1, t2 = oc.t2, t3 = oc.t3, t4 = oc.t4, ctol = oc.ctol;
    const double dlr = fabs(lfRow - rhRow), hlr = 0.5 * dlr;  // abs(lfCurrentRow - rhCurrentRow), 0.5*abs(...)
    // this lane's (c, d) pairs
    const float n6Inv = rcp_small(n6), n2Inv = rcp_small(n2);
#pragma unroll
    for (int s = 0; s < kSlots; ++s) {
        const int cd = lane + 64 * s;
        st.live[s] = cd < nCD;
        int c, d;
        divmod_lattice(st.live[s] ? cd : 0, n6, n6Inv, c, d);
        const double y4 = lo[4] + c, y6 = lo[6] + d;
        st.A4n[s] = wr * fabs(y4 - nIdx[4]);
        st.A4c[s] = wr * fabs(y4 - cIdx[4]);
        st.A6n[s] = wr * fabs(y6 - nIdx[6]);
        st.A6c[s] = wr * fabs(y6 - cIdx[6]);
        st.Dcd[s] = fabs(y4 - y6);
        st.Ucd[s] = fabs(st.Dcd[s] - L);
        st.hcd[s] = 0.5 * st.Dcd[s];
        st.V2[s] = fabs(fabs(st.hcd[s] - hlr) - K);
        const double c3 = t1 - st.Dcd[s], c4 = st.Dcd[s] - t2, G = 0.5 * fabs(st.Dcd[s] - dlr), c7 = t3 - G, c8 = G - t4;
        st.feasCD[s] = c3 <= ctol && c4 <= ctol && c7 <= ctol && c8 <= ctol;
        double rm = 0.0;
        rm = c3 > rm ? c3 : rm;
        rm = c4 > rm ? c4 : rm;
        rm = c7 > rm ? c7 : rm;
        rm = c8 > rm ? c8 : rm;
        st.rmCD[s] = rm;
    }
    // ---- the points that can win, listed (round 4) ----
    // The search orders points by (violation, objective, index), and the violation depends on a point only through
    // Dab = |x0 - x2| and Dcd = |x4 - x6|.  With the constraints on, the smallest violation over the box follows from the <= 64
    // values of Dab against this lane's (c, d) pairs (one pass: per-lane minimum with a mask of the Dab values that attain it,
    // then the wave's minimum), and only points that attain it can win.  They are few — the reference's yaml constraint set
    // has no feasible point and the smallest violation is attained along a curve of (Dab, Dcd) combinations: 1.5-4 % of the
    // box — but they are spread over most (a, b) pairs, one to six lanes each (profiles/round4_opt_stage_trace.txt), so the
    // wavefront LISTS them (ballot prefix into LDS: integer work only) and evaluates the objective on the list, 64 points at
    // a time, with the same expressions on the same operands: the same winner, bit for bit.  Longer lists than the cap
    // (feasible problems: the whole band of violation 0), boxes with more than 64 values of Dab or 128 (c, d) pairs search
    // every point.
    if (ll == nullptr || !oc.useConstraints) return;  // (uniform over the workgroup: every wavefront searches the same problem)
    unsigned* list = ll->list[slice];
    constexpr int kCap = W == 1 ? kOptListCap : kOptListCap / 4;
    const int dLo = lo[0] - up[2], dHi = up[0] - lo[2];  // x0 - x2 takes every integer of [dLo, dHi]
    const int dMin = (dLo <= 0 && dHi >= 0) ? 0 : min(abs(dLo), abs(dHi)), dMax = max(abs(dLo), abs(dHi));
    const int nD = dMax - dMin + 1;
    if (!(nD <= 64 && nAB <= 0xFFFF)) return;
    double myMin[kSlots];
    unsigned long long good[kSlots];
#pragma unroll
    for (int s = 0; s < kSlots; ++s) {
        myMin[s] = inf;
        good[s] = 0ull;
    }
    for (int k = slice; k < nD; k += nSearch) {  // (nSearch > 1: this wavefront's share of the values; exchanged below)
        const double Dab = dMin + k;
        const double c1 = t1 - Dab, c2 = Dab - t2;
        const bool feasAB = c1 <= ctol && c2 <= ctol;
        double rmAB = 0.0;
        rmAB = c1 > rmAB ? c1 : rmAB;
        rmAB = c2 > rmAB ? c2 : rmAB;
#pragma unroll
        for (int s = 0; s < kSlots; ++s) {
            const double E = 0.5 * fabs(Dab - st.Dcd[s]), c5 = t3 - E, c6 = E - t4;
            const bool feasible = feasAB && st.feasCD[s] && c5 <= ctol && c6 <= ctol;
            double rm = rmAB;
            rm = st.rmCD[s] > rm ? st.rmCD[s] : rm;
            rm = c5 > rm ? c5 : rm;
            rm = c6 > rm ? c6 : rm;
            const double key = feasible ? 0.0 : rm;
            if (st.live[s]) {
                if (key < myMin[s]) {
                    myMin[s] = key;
                    good[s] = 1ull << k;
                } else if (key == myMin[s]) {
                    good[s] |= 1ull << k;
                }
            }
        }
    }
    if (slice == 0 && nSearch > 1) FPE_OPT_STAMP(14);
    if constexpr (W > 1) {
        if (nSearch > 1) {
#pragma unroll
            for (int s = 0; s < kSlots; ++s) {
                ll->partMin[slice][s][lane] = myMin[s];
                ll->partGood[slice][s][lane] = good[s];
            }
            ++meets;
            opt_arrive(meet, lane);
            opt_wait_ge(meet, nSearch * meets);
#pragma unroll
            for (int s = 0; s < kSlots; ++s) {
                double mn = inf;
                for (int w = 0; w < nSearch; ++w) {
                    const double o = ll->partMin[w][s][lane];
                    mn = o < mn ? o : mn;
                }
                unsigned long long g = 0ull;
                for (int w = 0; w < nSearch; ++w)
                    if (ll->partMin[w][s][lane] == mn) g |= ll->partGood[w][s][lane];
                myMin[s] = mn;
                good[s] = g;
            }
        }
    }
    const double minKey = wave_min_nonneg(myMin[0] < myMin[1] ? myMin[0] : myMin[1]);
#pragma unroll
    for (int s = 0; s < kSlots; ++s)
        if (!(myMin[s] == minKey)) good[s] = 0ull;
    if (slice == 0 && nSearch > 1) FPE_OPT_STAMP(15);
    // list the surviving points in enumeration order: entry = ab | cd << 16.
    // Whether (ab, cd) survives depends on ab only through its Dab value k: the (c, d) pairs of value k are two 64-bit masks
    // (the ballots of bit k of the lanes' `good`), kept by lane k.  A lane then takes an (a, b) pair of the wavefront's share,
    // fetches its value's masks, a prefix sum over the lanes gives it its place in the list, and it writes its entries itself
    // (round 5; before: the pairs one after the other with two ballots and a dependent prefix each — 2.4 of a cycle's 14.8 us)
    const unsigned long long anyMine = good[0] | good[1];
    const unsigned long long anyK = static_cast<unsigned long long>(wave_or_u32(static_cast<unsigned>(anyMine))) |
                                    (static_cast<unsigned long long>(wave_or_u32(static_cast<unsigned>(anyMine >> 32))) << 32);
    unsigned long long mk0 = 0ull, mk1 = 0ull;  // lane k: the (c, d) pairs (slot 0 / slot 1) that survive beside Dab value k
    for (unsigned long long left = anyK; left != 0ull; left &= left - 1ull) {
        const int k = __builtin_ctzll(left);
        const unsigned long long b0 = __ballot(((good[0] >> k) & 1ull) != 0ull), b1 = __ballot(((good[1] >> k) & 1ull) != 0ull);
        if (lane == k) {
            mk0 = b0;
            mk1 = b1;
        }
    }
    int count = 0;
    for (int ab0 = slice; ab0 < nAB && count <= kCap; ab0 += 64 * nSearch) {
        const int ab = ab0 + nSearch * lane;
        const bool mine = ab < nAB;
        int a, b;
        divmod_lattice(mine ? ab : 0, n2, n2Inv, a, b);
        const int k = abs((lo[0] + a) - (lo[2] + b)) - dMin;  // in [0, nD)
        unsigned long long m0 = __shfl(mk0, k), m1 = __shfl(mk1, k);
        if (!mine) m0 = m1 = 0ull;
        const int cnt = __builtin_popcountll(m0) + __builtin_popcountll(m1);
        const int incl = wave_inclusive_sum(cnt);
        int at = count + incl - cnt;
        count += __builtin_amdgcn_readlane(incl, 63);
        for (; m0 != 0ull; m0 &= m0 - 1ull, ++at)
            if (at < kCap) list[at] = static_cast<unsigned>(ab) | (static_cast<unsigned>(__builtin_ctzll(m0)) << 16);
        for (; m1 != 0ull; m1 &= m1 - 1ull, ++at)
            if (at < kCap) list[at] = static_cast<unsigned>(ab) | (static_cast<unsigned>(64 + __builtin_ctzll(m1)) << 16);
    }
    if (count <= kCap) {
        st.mode = 0;
        st.count = count;
        st.minKey = minKey;
    }
}
// Second part, with the columns decided (x[1], x[3], x[5], x[7]): row points (x[0], x[2], x[4], x[6]) = lo + (a, b, c, d),
// enumeration index t = ((a n2 + b) n4 + c) n6 + d.  A searching wavefront takes the (a, b) pairs slice, slice + nSearch, ...,
// its lanes the (c, d) pairs; per lane t only grows, so "strictly better" keeps the first of equals, and the reduction orders
// by (violation, objective, t): the lexicographically first minimum, as the oracle's nested loops find it.
// The objective and the constraints are taken apart by what they depend on (round 4): every f64 operation below is one of the
// reference's expression, on the same operands, in the same order — a point's objective is
//   ((w1 * S1 + w2 * S2) + w3 * (Uab + Ucd)) + w4 * (V1 + V2),   S = ((((((A0 + C1) + A2) + C3) + A4) + C5) + A6) + C7
// (cpp:61-72, left to right) — but the sub-expressions of (x0, x2) alone are evaluated once per (a, b) pair and those of
// (x4, x6) alone once per LANE (a lane keeps its one or two (c, d) pairs for the whole search): 31 operations per point
// instead of ~110, and the values are bit for bit the ones opt_objective / opt_violation return (tests/test_gpu_opt.py).
template <int W>
__device__ __forceinline__ OptBest opt_rows_finish(const OptConsts& oc, const int (&nIdx)[8], const int (&cIdx)[8], const int (&lo)[8], const int (&up)[8], double lfRow,
                                   double rhRow, int lane, int slice, int nSearch, const double (&x)[8], OptListLds<W>* ll, const OptRowsState& st) {
    if (st.mode == 2) return opt_solve_rows_plain(oc, nIdx, cIdx, lo, up, lfRow, rhRow, lane, slice, nSearch, x);
    const double inf = __builtin_huge_val();
    const int n0 = up[0] - lo[0] + 1, n2 = up[2] - lo[2] + 1, n4 = up[4] - lo[4] + 1, n6 = up[6] - lo[6] + 1;
    const int nAB = n0 * n2, nCD = n4 * n6;
    constexpr int kSlots = 2;
    const double w1 = oc.w1, w2 = oc.w2, w3 = oc.w3, w4 = oc.w4, wr = oc.wr, wc = oc.wc;
    const double L = oc.lbOverRes, K = oc.skew2OverRes;
    const double t1 = oc.t1, t2 = oc.t2, t3 = oc.t3, t4 = oc.t4, ctol = oc.ctol;
    const double hlr = 0.5 * fabs(lfRow - rhRow);
    const float n6Inv = rcp_small(n6), n2Inv = rcp_small(n2);
    // column terms (the columns are decided: opt_solve_columns)
    const double C1n = wc * fabs(x[1] - nIdx[1]), C3n = wc * fabs(x[3] - nIdx[3]), C5n = wc * fabs(x[5] - nIdx[5]), C7n = wc * fabs(x[7] - nIdx[7]);
    const double C1c = wc * fabs(x[1] - cIdx[1]), C3c = wc * fabs(x[3] - cIdx[3]), C5c = wc * fabs(x[5] - cIdx[5]), C7c = wc * fabs(x[7] - cIdx[7]);
    if (st.mode == 0) {
        const unsigned* list = ll->list[slice];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        double bestF = inf;
        unsigned bestT = 0xFFFFFFFFu;
        for (int p = lane; p < st.count; p += 64) {
            const unsigned e = list[p];
            const int ab = static_cast<int>(e & 0xFFFFu), cd = static_cast<int>(e >> 16);
            int a, b, c, d;
            divmod_lattice(ab, n2, n2Inv, a, b);
            divmod_lattice(cd, n6, n6Inv, c, d);
            const double y0 = lo[0] + a, y2 = lo[2] + b, y4 = lo[4] + c, y6 = lo[6] + d;
            const double Pn = ((wr * fabs(y0 - nIdx[0]) + C1n) + wr * fabs(y2 - nIdx[2])) + C3n;
            const double Pc = ((wr * fabs(y0 - cIdx[0]) + C1c) + wr * fabs(y2 - cIdx[2])) + C3c;
            const double Dab = fabs(y0 - y2), Uab = fabs(Dab - L), hab = 0.5 * Dab;
            const double a4n = wr * fabs(y4 - nIdx[4]), a4c = wr * fabs(y4 - cIdx[4]), a6n = wr * fabs(y6 - nIdx[6]), a6c = wr * fabs(y6 - cIdx[6]);
            const double dcd = fabs(y4 - y6), ucd = fabs(dcd - L), hcdP = 0.5 * dcd, v2 = fabs(fabs(hcdP - hlr) - K);
            const double S1 = (((Pn + a4n) + C5n) + a6n) + C7n;
            const double S2 = (((Pc + a4c) + C5c) + a6c) + C7c;
            const double V1 = fabs(fabs(hab - hcdP) - K);
            const double f = ((w1 * S1 + w2 * S2) + w3 * (Uab + ucd)) + w4 * (V1 + v2);
            if (f < bestF) {  // (per lane the index only grows: the first of equals stays)
                bestF = f;
                bestT = static_cast<unsigned>(ab) * static_cast<unsigned>(nCD) + static_cast<unsigned>(cd);
            }
        }
        double bestKey = st.minKey;
        opt_wave_min(bestKey, bestF, bestT);
        return OptBest{bestKey, bestF, bestT, 0u};
    }
    double bestKey = inf, bestF = inf;
    unsigned bestT = 0xFFFFFFFFu;
    for (int ab = slice; ab < nAB; ab += nSearch) {
        int a, b;
        divmod_lattice(ab, n2, n2Inv, a, b);
        const double y0 = lo[0] + a, y2 = lo[2] + b;
        const double Pn = ((wr * fabs(y0 - nIdx[0]) + C1n) + wr * fabs(y2 - nIdx[2])) + C3n;
        const double Pc = ((wr * fabs(y0 - cIdx[0]) + C1c) + wr * fabs(y2 - cIdx[2])) + C3c;
        const double Dab = fabs(y0 - y2), Uab = fabs(Dab - L), hab = 0.5 * Dab;
        const double c1 = t1 - Dab, c2 = Dab - t2;
        const bool feasAB = c1 <= ctol && c2 <= ctol;
        double rmAB = 0.0;
        rmAB = c1 > rmAB ? c1 : rmAB;
        rmAB = c2 > rmAB ? c2 : rmAB;
        const unsigned tAB = static_cast<unsigned>(ab) * static_cast<unsigned>(nCD);
#pragma unroll
        for (int s = 0; s < kSlots; ++s) {
            const double S1 = (((Pn + st.A4n[s]) + C5n) + st.A6n[s]) + C7n;
            const double S2 = (((Pc + st.A4c[s]) + C5c) + st.A6c[s]) + C7c;
            const double V1 = fabs(fabs(hab - st.hcd[s]) - K);
            const double f = ((w1 * S1 + w2 * S2) + w3 * (Uab + st.Ucd[s])) + w4 * (V1 + st.V2[s]);
            const double E = 0.5 * fabs(Dab - st.Dcd[s]), c5 = t3 - E, c6 = E - t4;
            const bool feasible = feasAB && st.feasCD[s] && c5 <= ctol && c6 <= ctol;
            double rm = rmAB;  // (the maximum of the eight values and 0: the order of the comparisons does not matter)
            rm = st.rmCD[s] > rm ? st.rmCD[s] : rm;
            rm = c5 > rm ? c5 : rm;
            rm = c6 > rm ? c6 : rm;
            const double key = oc.useConstraints ? (feasible ? 0.0 : rm) : 0.0;
            if (st.live[s] && (key < bestKey || (key == bestKey && f < bestF))) {  // (per lane t only grows: the first of equals stays)
                bestKey = key;
                bestF = f;
                bestT = tAB + static_cast<unsigned>(lane + 64 * s);
            }
        }
    }
    opt_wave_min(bestKey, bestF, bestT);
    return OptBest{bestKey, bestF, bestT, 0u};
}

// Boxes with more than 128 (c, d) pairs: the plain loop over every point with the literal expressions.
__device__ OptBest opt_solve_rows_plain(const OptConsts& oc, const int (&nIdx)[8], const int (&cIdx)[8], const int (&lo)[8], const int (&up)[8],
                                        double lfRow, double rhRow, int lane, int slice, int nSearch, const double (&x)[8]) {
    const double inf = __builtin_huge_val();
    const int n0 = up[0] - lo[0] + 1, n2 = up[2] - lo[2] + 1, n4 = up[4] - lo[4] + 1, n6 = up[6] - lo[6] + 1;
    const int nAB = n0 * n2, nCD = n4 * n6;
    const float n2Inv = rcp_small(n2), n6Inv = rcp_small(n6);
    double bestKey = inf, bestF = inf;
    unsigned bestT = 0xFFFFFFFFu;
    double y[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) y[k] = x[k];
    for (int ab = slice; ab < nAB; ab += nSearch) {
        int a, b;
        divmod_lattice(ab, n2, n2Inv, a, b);
        y[0] = lo[0] + a;
        y[2] = lo[2] + b;
        const unsigned tAB = static_cast<unsigned>(ab) * static_cast<unsigned>(nCD);
        for (int cd = lane; cd < nCD; cd += 64) {
            int c, d;
            divmod_lattice(cd, n6, n6Inv, c, d);
            y[4] = lo[4] + c;
            y[6] = lo[6] + d;
            const double key = oc.useConstraints ? opt_violation(y, oc, lfRow, rhRow) : 0.0;
            const double f = opt_objective(y, oc, nIdx, cIdx, lfRow, rhRow);
            if (key < bestKey || (key == bestKey && f < bestF)) {
                bestKey = key;
                bestF = f;
                bestT = tAB + static_cast<unsigned>(cd);
            }
        }
    }
    opt_wave_min(bestKey, bestF, bestT);
    return OptBest{bestKey, bestF, bestT, 0u};
}

// Third part (wavefront 0): the winner's point.
__device__ int opt_solve_end(const OptBest& best, const int (&lo)[8], const int (&up)[8], double (&x)[8], double& minf) {
    const int n2 = up[2] - lo[2] + 1, n4 = up[4] - lo[4] + 1, n6 = up[6] - lo[6] + 1;
    const int nCD = n4 * n6;
    int ab, cd, a, b, c, d;
    divmod_lattice(static_cast<int>(best.t), nCD, rcp_small(nCD), ab, cd);
    divmod_lattice(ab, n2, rcp_small(n2), a, b);
    divmod_lattice(cd, n6, rcp_small(n6), c, d);
    x[0] = lo[0] + a;
    x[2] = lo[2] + b;
    x[4] = lo[4] + c;
    x[6] = lo[6] + d;
    minf = best.f;
    return best.key > 0.0 ? 2 : 0;
}

struct OptShared {
    double cur[4][3];  // RF,RH,LH,LF_optCurrentPosition_
    float vals[4][kOptBoxCells];  // per-leg scratch of the mean heights
};

}  // namespace

// W wavefronts per pose: 1 for batches (throughput: a pose per wavefront, two per SIMD), 8 for a handful of poses (512 threads
// keep the 256-register budget: 1024 halve it and spill) — the service call plans ONE pose, and the optimiser's box (14 641 row
// points per cycle at 2 cm) would otherwise be walked by one wavefront, cycle after cycle.  Wavefront 0 runs the track and decides
// the columns; the seven helpers poll a sequence number in LDS (s_sleep between polls), analyse and list the row lattice as soon as
// a cycle's bounds are published — beside the column searches —, evaluate their lists once the columns are, and hand back their
// best (round 5; until then they waited at the workgroup barrier and the whole search followed the columns).
struct OptKernArgs {  // opt_track_kernel's argument list as a struct (the argument segment's layout)
    DevMap m;
    PlanConsts pc;
    OptConsts oc;
    const fpe_pose* poses;
    int B, nCycles;
    const uint8_t* cycleOk;
    fpe_opt_out out;
    uint32_t* doneFlag;
    uint32_t doneValue;
};
#ifndef FPE_OPT_RELOAD_ARGS
#define FPE_OPT_RELOAD_ARGS 1
#endif
template <int W>
__global__ __launch_bounds__(64 * W) __attribute__((amdgpu_waves_per_eu(2))) void opt_track_kernel(DevMap m, PlanConsts pc, OptConsts oc, const fpe_pose* __restrict__ poses, int B,
                                                           int nCycles, const uint8_t* __restrict__ cycleOk, fpe_opt_out out, uint32_t* doneFlag, uint32_t doneValue) {
    const int nCyclesArg = nCycles;
    const DevMap& mArg = m;
    const PlanConsts& pcArg = pc;
    const OptConsts& ocArg = oc;
    const fpe_opt_out& outArg = out;
    __shared__ OptShared sh;
    __shared__ OptBest slots[W];
    __shared__ OptProblem probs[2];  // by cycle parity: wavefront 0 may publish cycle g + 1 while a helper still reads cycle g
    __shared__ OptListLds<W> optLists;  // the listed search of opt_solve_rows
    const int b = blockIdx.x;
    if (b >= B) return;
    const int lane = static_cast<int>(threadIdx.x) & 63;
    const int wave = static_cast<int>(threadIdx.x) >> 6;
    // (W > 1) wavefront 0 and its helpers meet through sequence numbers and counters in LDS, never at the workgroup barrier:
    // sync[0] = cycles whose problem (bounds, rows) is published, sync[1] = cycles whose columns are, sync[2] = helpers arrived at the
    // Dab exchange, sync[3] = helpers' results handed back
    __shared__ int sync[4];
    // a cycle's verdict for the helpers — 1: search its rows, 0: nothing to search, -1: the track has stopped for good — one entry per
    // cycle (the track's records count cycles in a byte): wavefront 0 does not wait for anybody in a cycle without a search, and a
    // slot shared by cycle parity could be two cycles ahead of a helper that has not looked yet
    __shared__ signed char runOf[256];
    if constexpr (W > 1) {
        if (threadIdx.x < 4) sync[threadIdx.x] = 0;
        __syncthreads();
        if (wave > 0) {  // helper wavefronts: the row search only
#ifndef FPE_OPT_NO_PRIO
            // wavefront 4 shares its SIMD with wavefront 0, whose column searches are done long before anybody needs them: the helpers
            // go first (they sleep in their polls while wavefront 0 has the critical path), or the seven wait for the one at their exchange
            __builtin_amdgcn_s_setprio(3);
#endif
            int meets = 0;
            for (int cyc = 0; cyc < nCycles; ++cyc) {
                opt_wait_ge(&sync[0], cyc + 1);  // the cycle's problem is published
                if (wave == 1) FPE_OPT_STAMP(10);
                const OptProblem& prob = probs[cyc & 1];
                const int run = runOf[cyc & 255];
                if (run < 0) return;
                if (run == 0) continue;
                int nIdx[8], cIdx[8], lo[8], up[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    nIdx[k] = prob.nIdx[k];
                    cIdx[k] = prob.cIdx[k];
                    lo[k] = prob.lo[k];
                    up[k] = prob.up[k];
                }
                const double lfRowP = prob.lfRow, rhRowP = prob.rhRow;
                OptRowsState st;
                opt_rows_prepare<W>(oc, nIdx, cIdx, lo, up, lfRowP, rhRowP, lane, wave - 1, W - 1, &optLists, &sync[2], meets, st, cyc);  // while wavefront 0 decides the columns
                if (wave == 1) FPE_OPT_STAMP(11);
                opt_wait_ge(&sync[1], cyc + 1);  // the columns are decided
                if (wave == 1) FPE_OPT_STAMP(12);
                double x[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) x[k] = prob.x[k];
                const OptBest mine = opt_rows_finish<W>(oc, nIdx, cIdx, lo, up, lfRowP, rhRowP, lane, wave - 1, W - 1, x, &optLists, st);
                if (lane == 0) slots[wave] = mine;
                opt_arrive(&sync[3], lane);
                if (wave == 1) FPE_OPT_STAMP(13);
            }
            return;
        }
    }
    const Grp<16> g(lane);
    const int leg = lane >> 4;  // RF, RH, LH, LF: 16 lanes each
    const fpe_pose* pp = poses + b;
    const double x0 = pp->position[0], y0 = pp->position[1], z0 = pp->position[2];
    const int gait = pp->gait;
    const float rOverride = pp->leg_search_radius[leg];
    // the nominal track's cycle flags of this pose, fetched once (device-mapped host memory for small calls: a PCIe round
    // trip each): lane l holds cycles l, l + 64, l + 128, l + 192
    unsigned long long okMask[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = lane + 64 * q;
        okMask[q] = __ballot(c < nCycles && cycleOk[static_cast<size_t>(b) * nCycles + c] != 0);
    }
    const float Rf = rOverride > 0.0f ? rOverride : pc.searchRadius;  // searchRadius_ (cpp:1616-1617)
    const double lx = static_cast<double>(Rf * 2), ly = static_cast<double>(Rf);
    const bool odd = (leg & 1) != 0, high = (leg & 2) != 0;
    const double biasX = high ? (odd ? pc.biasX[3] : pc.biasX[2]) : (odd ? pc.biasX[1] : pc.biasX[0]);
    const double biasY = high ? (odd ? pc.biasY[3] : pc.biasY[2]) : (odd ? pc.biasY[1] : pc.biasY[0]);
    // stance (cpp:350-378) and setFirstGait (cpp:582-588, 2679-2699)
    if (g.sub == 0) {
        double sx = (leg == 0 || leg == 3) ? pc.LbHalf : -pc.LbHalf;
        double sy = (leg <= 1) ? pc.WbHalfNeg : pc.WbHalfPos;
        double sz = 0;
        sx += x0;
        sy += y0;
        sz += z0;
        sh.cur[leg][0] = sx - pc.stepHalf;
        sh.cur[leg][1] = sy;
        sh.cur[leg][2] = sz;
    }
    pose_sync<16>();
    double adjY = 0.0;                                  // ajustedPose_[1], cpp:759
    double lfRow = oc.lfRow0, rhRow = oc.rhRow0;        // cpp:36
    int failCycle = 255;
    bool stopped = gait != 0;  // the walk gait (build-defined) has no opt track: zero records
    bool helpersGone = false;  // (W > 1) the helper wavefronts have been told to leave
    int nRuns = 0;             // (W > 1) cycles whose row search ran: the helpers hand back W - 1 results each

    for (int cycIt = 0; cycIt < nCyclesArg; ++cycIt) {
        // Round 6, the one-pose form (W > 1): the cycle reads the map's geometry and the constants from the ARGUMENT SEGMENT again
        // (scalar loads through a laundered pointer) instead of holding them — spilled to vector lanes — across the chain, as the
        // one-wavefront-per-pose plan kernels do: the service call 108.3 -> 106.9 us (A/B in one call, three repetitions).
        constexpr bool kReloadO = FPE_OPT_RELOAD_ARGS != 0 && W > 1;
        typedef const OptKernArgs __attribute__((address_space(4))) * OptArgPtr;
        OptArgPtr ka4 = (OptArgPtr)__builtin_amdgcn_kernarg_segment_ptr();
        if constexpr (kReloadO) asm volatile("" : "+s"(ka4));
        const OptKernArgs* ka = (const OptKernArgs*)ka4;
        const DevMap& m = kReloadO ? ka->m : mArg;
        const PlanConsts& pc = kReloadO ? ka->pc : pcArg;
        const OptConsts& oc = kReloadO ? ka->oc : ocArg;
        const int nCycles = kReloadO ? ka->nCycles : nCyclesArg;
        const fpe_opt_out out = kReloadO ? ka->out : outArg;
        const int cyc = cycIt;
        const size_t oCyc = static_cast<size_t>(b) * nCycles + cyc;
        fpe_opt_cycle rec;
        __builtin_memset(&rec, 0, sizeof(rec));
        fpe_opt_foothold fh;
        __builtin_memset(&fh, 0, sizeof(fh));
        fh.foot_id = static_cast<uint8_t>(leg);
        fh.gait_cycle_id = static_cast<uint8_t>(cyc);
        SubGeom gm;
        gm.ok = false;
        double nextX = 0.0, nextY = 0.0;
        FPE_OPT_STAMP(0);
        if (!stopped) {
            // ---- STEP(1) getGaitCycleSearchGridMap, cpp:2307-2408 ----
            nextX = polygon_center_x(sh.cur) + pc.step;  // cpp:2322-2327
            nextY = y0 + adjY;                           // cpp:2329
            if (centre_usable(nextX, nextY)) gm = opt_submap(m.g, nextX, nextY, pc.isosLen, pc.isosWid);  // cpp:2345
            if (!gm.ok) {  // cpp:2347-2349 -> cpp:931-934: the service handler returns false here
                stopped = true;
                failCycle = cyc;
                rec.gate_failed = 1;
            }
        }
        if constexpr (W > 1) {
            if (stopped && !helpersGone) {
                if (lane == 0) runOf[cyc & 255] = -1;
                opt_publish(&sync[0], cyc + 1, lane);  // the helpers read -1 and return
                helpersGone = true;
            }
        }
        if (!stopped) {
            rec.lf_current_row = lfRow;
            rec.rh_current_row = rhRow;
            rec.gait_top_left[0] = gm.i0;
            rec.gait_top_left[1] = gm.j0;
            rec.gait_size[0] = gm.g.rows;
            rec.gait_size[1] = gm.g.cols;
            // the leg's next default position (getDefaultFootholdNext, cpp:2391-2397, 2411-2418) and its gaitMap_ index
            // (getMapIndex, cpp:965-976)
            const double nx = nextX + biasX, ny = nextY + biasY;
            const int nomI = index_of(nx, gm.g.orgX, gm.g.posX, gm.g.res), nomJ = index_of(ny, gm.g.orgY, gm.g.posY, gm.g.res);
            FPE_OPT_STAMP(1);
            // ---- STEP(3) checkFootholdUseCentroidMethod(gaitMap_, next, result, beginRow, endRow), cpp:1010-1013 ----
            int code = 6, beginRow = 0, endRow = 0;
            double cenX = 0.0, cenY = 0.0;
            Submap ss;
            ss.ok = false;
            if (centre_usable(nx, ny)) ss = submap_info(gm.g, nx, ny, lx, ly);  // cpp:1627 on gaitMap_
            if (ss.ok) {
                const int ni = ss.ni, nj = ss.nj;
                const int bottomRow = ni - 1, rightCol = nj - 1;
                // row scan (cpp:1649-1658 whole-region test, cpp:1717-1750 blocked rows): lane = row, 16 rows per round
                unsigned long long blkLo = 0ull, blkHi = 0ull;
                bool anyBelow = false;
                for (int r0 = 0; r0 < ni; r0 += 16) {
                    const int r = r0 + g.sub;
                    int cnt = 0;
                    if (r < ni) {
                        const float* row = m.trav + static_cast<size_t>(gm.i0 + ss.i0 + r) * m.g.cols + (gm.j0 + ss.j0);
                        for (int c0 = 0; c0 < nj; c0 += 8) {  // eight loads in flight (a plain loop waits for each)
                            float v[8];
#pragma unroll
                            for (int u = 0; u < 8; ++u) v[u] = row[min(c0 + u, nj - 1)];
#pragma unroll
                            for (int u = 0; u < 8; ++u) cnt += (c0 + u < nj && v[u] < pc.thrDefault) ? 1 : 0;  // raw compare: NaN passes (cpp:1653, 1736)
                        }
                    }
                    anyBelow |= cnt > 0;
                    const unsigned long long mk = g.ballot(r < ni && 2 * cnt > nj);  // cpp:1743
                    if (r0 < 64) blkLo |= mk << r0;
                    else if (r0 < 128) blkHi |= mk << (r0 - 64);
                }
                FPE_OPT_STAMP(2);
                const bool whole = ni * nj > 0 && !g.any(anyBelow);
                const int minRow = blkLo ? __builtin_ctzll(blkLo) : (blkHi ? 64 + __builtin_ctzll(blkHi) : 0);
                const int maxRow = blkHi ? 127 - __builtin_clzll(blkHi) : (blkLo ? 63 - __builtin_clzll(blkLo) : 0);
                int newRow = 0, newCol = 0, bandB = 0, bandE = 0;
                bool hasCell = false;
                if (whole) {  // cpp:1684-1693
                    code = 0;
                    bandB = 0;
                    bandE = bottomRow;
                } else if (minRow == 0 && maxRow != bottomRow) {  // case 1, cpp:1777-1795
                    code = 1;
                    newRow = (maxRow + bottomRow + 1) >> 1;
                    newCol = (rightCol + 1) >> 1;
                    bandB = maxRow + 1;
                    bandE = bottomRow;
                    hasCell = true;
                } else if (minRow != 0 && maxRow != bottomRow) {  // case 2, cpp:1843-1890
                    if (minRow >= (bottomRow - maxRow)) {
                        code = 2;
                        newRow = (minRow + 1) >> 1;
                        bandB = 0;
                        bandE = minRow - 1;
                    } else {
                        code = 3;
                        newRow = (maxRow + bottomRow) >> 1;
                        bandB = maxRow + 1;
                        bandE = bottomRow;
                    }
                    newCol = rightCol >> 1;
                    hasCell = true;
                } else if (minRow != 0 && maxRow == bottomRow) {  // case 3, cpp:1944-1961
                    code = 4;
                    newRow = (minRow + 1) >> 1;
                    newCol = rightCol >> 1;
                    bandB = 0;
                    bandE = minRow - 1;
                    hasCell = true;
                } else {
                    code = 5;  // no branch taken: result and rows untouched
                }
                if (code <= 4) {
                    // cpp:1696-1710: the band's rows as rows of gaitMap_, through the position of the rectangle's cell (row, 1);
                    // ONE Position for both conversions (a failed getPosition keeps it; uninitialised before the first: (0,0))
                    double qx = 0.0;
                    if (in_range(bandB, 1, ni, nj)) qx = cell_pos(ss.baseX, gm.g.res, bandB);
                    beginRow = index_of(qx, gm.g.orgX, gm.g.posX, gm.g.res);
                    if (in_range(bandE, 1, ni, nj)) qx = cell_pos(ss.baseX, gm.g.res, bandE);
                    endRow = index_of(qx, gm.g.orgX, gm.g.posX, gm.g.res);
                    cenX = hasCell ? cell_pos(ss.baseX, gm.g.res, newRow) : nx;  // cpp:1816 / cpp:1688-1689
                    cenY = hasCell ? cell_pos(ss.baseY, gm.g.res, newCol) : ny;
                }
            }
            // centroidIndex (cpp:1030-1041): getMapIndex of the result — (0, 0) when it was left untouched
            const int cenI = index_of(cenX, gm.g.orgX, gm.g.posX, gm.g.res), cenJ = index_of(cenY, gm.g.orgY, gm.g.posY, gm.g.res);
            // gather the four legs (lane 16 * leg holds the leg's values; constant lanes: v_readlane, no LDS-crossbar round trip);
            // optimiser order LF, RH, RF, LH
            int nIdx[8], cIdx[8], lo[8], up[8];
            {
                constexpr int order[4] = {3, 1, 0, 2};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int src = 16 * order[k];
                    nIdx[2 * k] = __builtin_amdgcn_readlane(nomI, src);
                    nIdx[2 * k + 1] = __builtin_amdgcn_readlane(nomJ, src);
                    cIdx[2 * k] = __builtin_amdgcn_readlane(cenI, src);
                    cIdx[2 * k + 1] = __builtin_amdgcn_readlane(cenJ, src);
                    lo[2 * k] = __builtin_amdgcn_readlane(beginRow, src);  // xBounds rows, cpp:1067-1076
                    up[2 * k] = __builtin_amdgcn_readlane(endRow, src);
                }
                lo[1] = lo[7] = oc.colLoA;  // cpp:1063-1066
                up[1] = up[7] = oc.colUpA;
                lo[3] = lo[5] = oc.colLoB;
                up[3] = up[5] = oc.colUpB;
#pragma unroll
                for (int l = 0; l < 4; ++l) {
                    rec.traversable_row[0][l] = __builtin_amdgcn_readlane(beginRow, 16 * l);
                    rec.traversable_row[1][l] = __builtin_amdgcn_readlane(endRow, 16 * l);
                    rec.centroid_code[l] = static_cast<uint8_t>(__builtin_amdgcn_readlane(code, 16 * l));
                }
            }
            // ---- STEP(4) the optimiser (build-defined) ----
            double x[8], minf;
            int status = opt_solve_check(oc, nIdx, cIdx, lo, up, lfRow, rhRow, x, minf);
            if constexpr (W > 1) {
                // the bounds and the rows are all the helpers' first part needs: published BEFORE the column searches, which the
                // helpers' Dab analysis and listing then overlap (round 5: 8.3 k of a cycle's 39 k clocks were the columns with seven
                // wavefronts waiting, 18.9 k the row search with wavefront 0 as one of eight searchers)
                OptProblem& prob = probs[cyc & 1];
                if (lane == 0) {
                    runOf[cyc & 255] = status < 0 ? 1 : 0;
                    prob.lfRow = lfRow;
                    prob.rhRow = rhRow;
                }
                if (status < 0 && lane == 0) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        prob.nIdx[k] = nIdx[k];
                        prob.cIdx[k] = cIdx[k];
                        prob.lo[k] = lo[k];
                        prob.up[k] = up[k];
                    }
                }
                opt_publish(&sync[0], cyc + 1, lane);
            }
            FPE_OPT_STAMP(3);
            if (status != 1) opt_solve_columns(oc, nIdx, cIdx, lo, up, lfRow, rhRow, lane, x);
            if (status == 3) minf = opt_objective(x, oc, nIdx, cIdx, lfRow, rhRow);
            if (status < 0) {
                OptBest best;
                if constexpr (W > 1) {
                    OptProblem& prob = probs[cyc & 1];
                    if (lane == 0) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) prob.x[k] = x[k];
                    }
                    opt_publish(&sync[1], cyc + 1, lane);
                    FPE_OPT_STAMP(4);
                    ++nRuns;
                    opt_wait_ge(&sync[3], (W - 1) * nRuns);  // every helper's best is in place
                    FPE_OPT_STAMP(5);
                    OptBest o{__builtin_huge_val(), __builtin_huge_val(), 0xFFFFFFFFu, 0u};
                    if (lane < W - 1) o = slots[1 + lane];
                    opt_wave_min(o.key, o.f, o.t);
                    best = o;
                } else {
                    OptRowsState st;
                    int meets = 0;
                    opt_rows_prepare<W>(oc, nIdx, cIdx, lo, up, lfRow, rhRow, lane, 0, 1, &optLists, nullptr, meets, st);
                    best = opt_rows_finish<W>(oc, nIdx, cIdx, lo, up, lfRow, rhRow, lane, 0, 1, x, &optLists, st);
                }
                status = opt_solve_end(best, lo, up, x, minf);
            }
            FPE_OPT_STAMP(6);
            rec.solver_status = static_cast<uint8_t>(status);
            rec.minf = minf;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                rec.nominal_index[k] = nIdx[k];
                rec.centroid_index[k] = cIdx[k];
                rec.x_lower[k] = lo[k];
                rec.x_upper[k] = up[k];
                rec.x[k] = static_cast<int>(x[k]);
            }
            // ---- STEP(6) positions on gaitMap_ (cpp:1283-1314): ONE Position through the four conversions, in the order
            // LF, RH, RF, LH; a failed getPosition keeps the previous one ----
            double ppx = 0.0, ppy = 0.0;
            double myX = 0.0, myY = 0.0;
            int myI = 0, myJ = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                constexpr int order[4] = {3, 1, 0, 2};
                const int i = static_cast<int>(x[2 * k]), j = static_cast<int>(x[2 * k + 1]);
                if (in_range(i, j, gm.g.rows, gm.g.cols)) {
                    ppx = cell_pos(gm.g.baseX, gm.g.res, i);
                    ppy = cell_pos(gm.g.baseY, gm.g.res, j);
                }
                if (leg == order[k]) {
                    myX = ppx;
                    myY = ppy;
                    myI = i;
                    myJ = j;
                }
            }
            FPE_OPT_STAMP(7);
            const bool commit = ((okMask[(cyc >> 6) & 3] >> (cyc & 63)) & 1ull) != 0ull;  // footholdValidation_ of the NOMINAL track (cpp:1323-1332)
            // (heights feed the footholds' z and nothing the chain reads later — getPolygonCenter's x uses x and y only: a caller
            // that asked for no footholds, the service's gate-only call, skips the elevation round trip; uniform over the workgroup)
            float z = 0.0f;
            if (out.footholds) {
                const float zLane = opt_mean_height(g, sh.vals[leg], gm.g, m.elev, m.g.cols, gm.i0, gm.j0, myX, myY, pc.rf, pc.rf2, pc.h);  // on gaitMap_
                z = __shfl(zLane, 16 * leg);
            }
            fh.x = myX;
            fh.y = myY;
            fh.z = z;
            fh.row = myI;
            fh.col = myJ;
            fh.committed = commit ? 1 : 0;
            rec.committed = commit ? 1 : 0;
            pose_sync<16>();
            if (commit) {  // cpp:1553-1568
                if (g.sub == 0) {
                    sh.cur[leg][0] = myX;
                    sh.cur[leg][1] = myY;
                    sh.cur[leg][2] = static_cast<double>(z);
                }
                const int rowOnGait = index_of(myX, gm.g.orgX, gm.g.posX, gm.g.res);  // gaitMap_.getIndex(...).x()
                lfRow = static_cast<double>(__builtin_amdgcn_readlane(rowOnGait, 16 * 3));
                rhRow = static_cast<double>(__builtin_amdgcn_readlane(rowOnGait, 16 * 1));
            }
            pose_sync<16>();
            adjY += pc.drift;  // cpp:1578
            FPE_OPT_STAMP(8);
        }
        if (lane == 0 && out.cycles) out.cycles[oCyc] = rec;
        if (g.sub == 0 && out.footholds) out.footholds[oCyc * 4 + leg] = fh;
    }
    if (lane == 0 && out.gate_fail_cycle) out.gate_fail_cycle[b] = static_cast<uint8_t>(failCycle);
    if (lane == 0 && out.rows_after) {  // lfCurrentRow / rhCurrentRow as this call leaves them (cpp:1561-1568)
        out.rows_after[2 * static_cast<size_t>(b)] = lfRow;
        out.rows_after[2 * static_cast<size_t>(b) + 1] = rhRow;
    }
    // One-pose service call (round 6): wavefront 0 — the only one that stores products — says so in the caller's host-mapped arena
    // with a system-scope RELEASE store behind its product stores; the host polls that word instead of waiting for the stream's
    // completion signal (fpe_engine.cpp, plan_host: -4 us of a ~100 us call).  (Measured: an additional __threadfence_system, or a
    // relaxed store behind s_waitcnt 0, end at the same time — the ordering is not what the word costs.)
    if (doneFlag && lane == 0) __hip_atomic_store(doneFlag, doneValue, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

hipError_t launch_opt_track(const DevMap& m, const PlanConsts& pc, const OptConsts& oc, const fpe_pose* d_poses, int B, int nCycles,
                            const uint8_t* d_cycleOk, const fpe_opt_out& d_out, hipStream_t stream, uint32_t* doneFlag, uint32_t doneValue) {
    if (B != 1) doneFlag = nullptr;  // (one workgroup per pose: a single word can only speak for one of them)
#ifndef FPE_OPT_W_SMALL
#define FPE_OPT_W_SMALL 8
#endif
    if (B <= 64) hipLaunchKernelGGL(opt_track_kernel<FPE_OPT_W_SMALL>, dim3(B), dim3(64 * FPE_OPT_W_SMALL), 0, stream, m, pc, oc, d_poses, B, nCycles, d_cycleOk, d_out, doneFlag, doneValue);
    else hipLaunchKernelGGL(opt_track_kernel<1>, dim3(B), dim3(64), 0, stream, m, pc, oc, d_poses, B, nCycles, d_cycleOk, d_out, doneFlag, doneValue);
    return hipGetLastError();
}
