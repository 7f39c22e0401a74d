// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see fpo_planner.hpp / fpo_gridmap.hpp).
// The "opt" track of globalFootholdPlan: /root/reference/foothold_planner/src/FootholdPlanner.cpp
//   cpp:28-51    file-scope NLopt globals            cpp:54-88    nloptFunc (objective)
//   cpp:92-148   nloptConstraint1..8                 cpp:913-1319 the per-cycle driver
//   cpp:1485-1570 commit of the opt results          cpp:2307-2408 getGaitCycleSearchGridMap
//   cpp:2557-2568 getMapIndex                        yaml:53-63   nlopt/* parameters
//
// What is restated LITERALLY (every f64 expression in the reference's order, every float -> double promotion, every
// double -> int truncation): the gait-cycle submap, the four next default positions, nominalIndex, the centroid method on
// the gait-cycle submap with traversableBeginRow / traversableEndRow, centroidIndex, the integer bounds xBounds, the
// constraint thresholds t1..t4, the objective and the eight constraints, the conversion of the optimiser's x back to
// positions and heights on the gait-cycle submap, the commit rule, lfCurrentRow / rhCurrentRow.
//
// What is BUILD-DEFINED: the optimiser.  The reference calls nlopt::opt("LN_COBYLA", 8) (yaml:60); NLopt is not in the
// image, is not pinned by the reference (CMakeLists.txt:20) and COBYLA's iterates are implementation-defined down to
// the last ulp — they cannot be reproduced here, and are not attempted.  `solveLattice` below minimises the SAME
// objective under the SAME constraints (same tolerance ctol) over the integer points of the same box — the reference
// truncates the optimiser's x to int before it uses it (cpp:1287-1312) — by exhaustive search, deterministic
// tie-breaks.  The engine runs the identical algorithm on the GPU; their parity is bit-exact.  Their distance to NLopt's
// COBYLA cannot be measured here; their distance to ANOTHER implementation of COBYLA (scipy 1.15's, in the build container
// only) driving this file's literal chain is: tests/golden/make_cobyla_golden.py -> tests/golden/cobyla_vs_lattice.json —
// 1 280 poses x 8 cycles on four synthetic maps: on identical problems the truncated x agrees in all eight variables in 29 %
// of the cycles (rows 35 %, columns 43 %; median L1 distance 2 rows + 1 column; the lattice's objective is lower by a median
// of 2.5, COBYLA's point being truncated after the fact), and the SERVICE GATE verdict — the cycle in which
// getGaitCycleSearchGridMap fails — is the same for 94.8 % of the poses (26 refused by the lattice's feet only, 29 by COBYLA's
// only, 12 by both in different cycles).  That is the weight fpe_set_tuning("service_opt_gate", 2) deserves: right for 19
// requests in 20 against a different COBYLA, unknown against NLopt's.
#include <cmath>
#include <cstdlib>

#include "fpo_planner.hpp"

namespace fpo {

namespace {

// The file-scope globals the objective and the constraints read (cpp:28-51), as one value.
struct NloptGlobals {
    double w1, w2, w3, w4, wr, wc;
    double lengthBase, skew, mapResolution;  // cpp:497-498, 514
    double lfCurrentRow, rhCurrentRow;       // cpp:36, 1564-1568
    double t1, t2, t3, t4;                   // cpp:1156-1159
    int nominalIndex[8], centroidIndex[8];   // order LF, RH, RF, LH x (row, col); cpp:50-51
};

// cpp:54-88.  `abs` is std::abs(double) (using namespace std, <cmath>; cpp:12-18).
double nloptFunc(const double* x, const NloptGlobals& g) {
    using std::abs;
    const int* nominalIndex = g.nominalIndex;
    const int* centroidIndex = g.centroidIndex;
    const double w1 = g.w1, w2 = g.w2, w3 = g.w3, w4 = g.w4, wr = g.wr, wc = g.wc;
    const double lengthBase = g.lengthBase, skew = g.skew, mapResolution = g.mapResolution;
    const double lfCurrentRow = g.lfCurrentRow, rhCurrentRow = g.rhCurrentRow;
    return (
            w1*( wr*(abs(x[0]-nominalIndex[0])) + wc*(abs(x[1]-nominalIndex[1])) +
                 wr*(abs(x[2]-nominalIndex[2])) + wc*(abs(x[3]-nominalIndex[3])) +
                 wr*(abs(x[4]-nominalIndex[4])) + wc*(abs(x[5]-nominalIndex[5])) +
                 wr*(abs(x[6]-nominalIndex[6])) + wc*(abs(x[7]-nominalIndex[7])) ) +
            w2*( wr*(abs(x[0]-centroidIndex[0])) + wc*(abs(x[1]-centroidIndex[1])) +
                 wr*(abs(x[2]-centroidIndex[2])) + wc*(abs(x[3]-centroidIndex[3])) +
                 wr*(abs(x[4]-centroidIndex[4])) + wc*(abs(x[5]-centroidIndex[5])) +
                 wr*(abs(x[6]-centroidIndex[6])) + wc*(abs(x[7]-centroidIndex[7])) ) +
            w3*( abs(abs(x[0]-x[2]) - lengthBase/mapResolution) +
                 abs(abs(x[4]-x[6]) - lengthBase/mapResolution) ) +
            w4*( abs(abs(0.5*abs(x[0]-x[2]) - 0.5*abs(x[4]-x[6])) - 2*skew/mapResolution) +
                 abs(abs(0.5*abs(x[4]-x[6]) - 0.5*abs(lfCurrentRow - rhCurrentRow)) - 2*skew/mapResolution) )
            );
}

// cpp:92-148: constraint k is satisfied when value <= ctol (NLopt's inequality-constraint tolerance, cpp:1166-1173).
double nloptConstraint(int k, const double* x, const NloptGlobals& g) {
    using std::abs;
    const double t1 = g.t1, t2 = g.t2, t3 = g.t3, t4 = g.t4;
    const double lfCurrentRow = g.lfCurrentRow, rhCurrentRow = g.rhCurrentRow;
    switch (k) {
        case 1: return ( t1 - abs(x[0] - x[2]) );
        case 2: return ( abs(x[0] - x[2]) - t2 );
        case 3: return ( t1 - abs(x[4] - x[6]) );
        case 4: return ( abs(x[4] - x[6]) - t2 );
        case 5: return ( t3 - 0.5*abs( abs(x[0] - x[2]) - abs(x[4] - x[6]) ) );
        case 6: return ( 0.5*abs( abs(x[0] - x[2]) - abs(x[4] - x[6]) ) - t4 );
        case 7: return ( t3 - 0.5*abs( abs(x[4] - x[6]) - abs(lfCurrentRow - rhCurrentRow) ) );
        default: return ( 0.5*abs( abs(x[4] - x[6]) - abs(lfCurrentRow - rhCurrentRow) ) - t4 );
    }
}

}  // namespace

// BUILD-DEFINED optimiser (see the header of this file).  Start x = x0 = centroidIndex (cpp:1180-1183).
//   status 1: NLopt's own precondition fails — some lb > ub, or x0 outside [lb, ub] (nlopt_optimize returns
//             NLOPT_INVALID_ARGS, the C++ wrapper throws, the reference swallows it at cpp:1224-1226) — x stays x0;
//   columns:  x[1], x[3], x[5], x[7] enter the objective through separate |.| terms only and no constraint: each is set,
//             in that order, to the integer of its interval that minimises the objective (others held; smallest integer on
//             ties);
//   rows:     every integer point of the box of (x[0], x[2], x[4], x[6]) is evaluated in lexicographic order, x[0]
//             slowest — the literal objective and, when useInequalityConstraits, the eight constraints.  A point is
//             FEASIBLE when every constraint value is <= ctol (NLopt's meaning of the tolerance).  The winner is the
//             first point with the smallest key (violation, objective), violation = 0 for a feasible point and the
//             largest constraint value otherwise: feasible points by objective; when the problem has none — with the
//             yaml values it never has: constraints 1-4 keep both hip distances within [0.9, 1.1] lengthBase = 20..24 rows
//             at 2 cm while 5-6 want them 6.4..9.6 rows apart — the point of least violation (what COBYLA's merit
//             function drives towards), status 2;
//   status 3: more than kMaxLatticePoints row points (x stays x0 with the columns of the previous step).
int solveLattice(const OptParams& op, const int* nominalIndex, const int* centroidIndex, const int* xLower, const int* xUpper,
                 double lengthBase, double skew, double mapResolution, double lfCurrentRow, double rhCurrentRow, double* x,
                 double* minf) {
    NloptGlobals g;
    g.w1 = op.w1; g.w2 = op.w2; g.w3 = op.w3; g.w4 = op.w4; g.wr = op.wr; g.wc = op.wc;
    g.lengthBase = lengthBase; g.skew = skew; g.mapResolution = mapResolution;
    g.lfCurrentRow = lfCurrentRow; g.rhCurrentRow = rhCurrentRow;
    g.t1 = lengthBase * op.hipLowerScale/mapResolution;    // cpp:1156
    g.t2 = lengthBase * op.hipUpperScale/mapResolution;    // cpp:1157
    g.t3 = 2* skew * op.skewLowerScale/mapResolution;      // cpp:1158
    g.t4 = 2* skew * op.skewUpperScale/mapResolution;      // cpp:1159
    for (int k = 0; k < 8; ++k) {
        g.nominalIndex[k] = nominalIndex[k];
        g.centroidIndex[k] = centroidIndex[k];
        x[k] = centroidIndex[k];  // cpp:1180-1183
    }
    *minf = nloptFunc(x, g);
    for (int k = 0; k < 8; ++k)
        if (xLower[k] > xUpper[k] || x[k] < xLower[k] || x[k] > xUpper[k]) return 1;
    static const int cols[4] = {1, 3, 5, 7};
    for (int c = 0; c < 4; ++c) {
        const int k = cols[c];
        double best = 0.0;
        int bestV = xLower[k];
        for (int v = xLower[k]; v <= xUpper[k]; ++v) {
            x[k] = v;
            const double f = nloptFunc(x, g);
            if (v == xLower[k] || f < best) {
                best = f;
                bestV = v;
            }
        }
        x[k] = bestV;
    }
    *minf = nloptFunc(x, g);
    double points = 1.0;
    for (int k = 0; k < 8; k += 2) points *= static_cast<double>(xUpper[k] - xLower[k] + 1);
    if (points > static_cast<double>(kMaxLatticePoints)) return 3;
    bool found = false;
    double bestKey = 0.0, best = 0.0;
    double bx[4] = {x[0], x[2], x[4], x[6]};
    double y[8];
    for (int k = 0; k < 8; ++k) y[k] = x[k];
    for (int a = xLower[0]; a <= xUpper[0]; ++a)
        for (int b = xLower[2]; b <= xUpper[2]; ++b)
            for (int c = xLower[4]; c <= xUpper[4]; ++c)
                for (int d = xLower[6]; d <= xUpper[6]; ++d) {
                    y[0] = a; y[2] = b; y[4] = c; y[6] = d;
                    double key = 0.0;
                    if (op.useInequalityConstraits) {
                        bool feasible = true;
                        double resmax = 0.0;
                        for (int q = 1; q <= 8; ++q) {
                            const double v = nloptConstraint(q, y, g);
                            feasible = feasible && v <= op.ctol;
                            resmax = v > resmax ? v : resmax;
                        }
                        key = feasible ? 0.0 : resmax;
                    }
                    const double f = nloptFunc(y, g);
                    if (!found || key < bestKey || (key == bestKey && f < best)) {
                        found = true;
                        bestKey = key;
                        best = f;
                        bx[0] = a; bx[1] = b; bx[2] = c; bx[3] = d;
                    }
                }
    x[0] = bx[0]; x[2] = bx[1]; x[4] = bx[2]; x[6] = bx[3];
    *minf = best;
    return bestKey > 0.0 ? 2 : 0;
}

// The opt track of one plan_global_footholds call (cpp:913-1319 inside the cycle loop cpp:762-1579), for the trot gait
// of the reference.  cycleOk[g] = footholdValidation_ of cycle g (the NOMINAL track's flags, cpp:1323): the opt track
// commits with the other tracks (cpp:1332, 1485-1568).  The handler returns false in the cycle whose
// getGaitCycleSearchGridMap fails (cpp:920-934): the chain stops there (gateFailCycle).
// forcedX / nForced (test infrastructure for tests/golden/make_cobyla_golden.py): the optimiser's x of the first nForced cycles
// is TAKEN from forcedX[g * 8 ..] (doubles, as an optimiser returns them — the reference truncates them afterwards, cpp:1287)
// instead of computed: lets an optimiser that lives outside this file (scipy's COBYLA) drive the literal chain cycle by cycle.
void planOptTrack(const GridMap& gridmap_, const Params& p, const OptParams& op, const PoseSpec& ps, int nCycles,
                  const uint8_t* cycleOk, OptOutput& out, const double* forcedX, int nForced) {
    const Constants c = makeConstants(p);
    out.cycles.assign((size_t)nCycles, OptCycle());
    out.gateFailCycle = -1;
    out.feetCenterPath.clear();
    out.feetDistance.clear();
    out.cogSpeed.clear();
    const double* initialPose_ = ps.pose;
    // initialize(): cpp:340-341 (float members), cpp:384-394 (doubles holding f32 values / sums of doubles), cpp:497-498
    const float lengthBaseF = p.length;
    const float widthBaseF = p.width + p.l1 * 2;
    const double isosLongEdge = lengthBaseF + p.skew * 2;    // cpp:391 (f32 sum stored in a double)
    const double isosShortEdge = widthBaseF;                  // cpp:392
    const double footSearchRectLength = p.searchRadius * 2;   // cpp:384
    const double footSearchRectWidth = p.searchRadius;        // cpp:385
    const double isosLength = isosLongEdge + footSearchRectLength;  // cpp:393
    const double isosWidth = isosShortEdge + footSearchRectWidth;   // cpp:394
    const double lengthBase = lengthBaseF;                    // cpp:497 (file-scope double)
    const double skew = p.skew;                               // cpp:498
    const double mapResolution = gridmap_.res;                // cpp:514
    const double footSearchRectCol = footSearchRectWidth/mapResolution;  // cpp:529 (double member, hpp:704-705)

    // stance + setFirstGait (cpp:350-378, 582-588, 2679-2699)
    const double sx[4] = {c.LbHalf, -c.LbHalf, -c.LbHalf, c.LbHalf};
    const double sy[4] = {c.WbHalfNeg, c.WbHalfNeg, c.WbHalfPos, c.WbHalfPos};
    Point3 optCurrent[4];
    for (int l = 0; l < 4; ++l) {
        Point3 s;
        s.x = sx[l]; s.y = sy[l]; s.z = 0;
        s.x += initialPose_[0]; s.y += initialPose_[1]; s.z += initialPose_[2];
        optCurrent[l] = s;
        optCurrent[l].x = s.x - c.stepHalf;
    }
    double ajustedPoseY = 0.0;                 // cpp:759
    double lfCurrentRow = op.lfCurrentRow0;    // file-scope globals: whatever the previous call left (0 at node start)
    double rhCurrentRow = op.rhCurrentRow0;
    if (ps.gait != 0) return;                  // the walk gait is build-defined and has no opt track

    for (int gaitCycleIndex = 0; gaitCycleIndex < nCycles; ++gaitCycleIndex) {
        OptCycle& oc = out.cycles[(size_t)gaitCycleIndex];
        oc.lfCurrentRow = lfCurrentRow;
        oc.rhCurrentRow = rhCurrentRow;
        // ---- STEP(1) getGaitCycleSearchGridMap, cpp:2307-2408 ----
        const Point3 feetCenter = getPolygonCenter(optCurrent[RF], optCurrent[RH], optCurrent[LH], optCurrent[LF]);  // cpp:2322
        Point3 nextFeetCenter;
        nextFeetCenter.x = feetCenter.x + p.stepLength;          // cpp:2327
        nextFeetCenter.y = initialPose_[1] + ajustedPoseY;       // cpp:2329
        nextFeetCenter.z = feetCenter.z;
        const Vec2 pc{nextFeetCenter.x, nextFeetCenter.y};
        bool isSuccess = false;
        SubmapInfo gaitInfo;
        GridMap gaitMap_;
        if (std::fabs(pc.x) <= 1e6 && std::fabs(pc.y) <= 1e6)   // oracle-defined guard (centreUsable, fpo_planner.cpp)
            gaitMap_ = gridmap_.getSubmap(pc, {isosLength, isosWidth}, isSuccess, &gaitInfo, true);  // cpp:2345
        if (!isSuccess) {  // cpp:2347-2349 -> cpp:931-934: the service handler returns false
            oc.gateFailed = 1;
            out.gateFailCycle = gaitCycleIndex;
            return;
        }
        oc.gaitTopLeft[0] = gaitInfo.topLeft.i; oc.gaitTopLeft[1] = gaitInfo.topLeft.j;
        oc.gaitSize[0] = gaitMap_.size.i; oc.gaitSize[1] = gaitMap_.size.j;
        // cpp:2391-2397 and again (same expressions) getFootholdSearchGridMap cpp:939-959 -> 2199-2213; the latter also
        // pushes THIS track's feet centre onto centroidFeetCenterPath (cpp:946, 2194-2196)
        out.feetCenterPath.push_back(feetCenter);
        Point3 optNext[4];
        for (int l = 0; l < 4; ++l) {  // getDefaultFootholdNext, cpp:2411-2418
            optNext[l].x = nextFeetCenter.x + c.biasX[l];
            optNext[l].y = nextFeetCenter.y + c.biasY[l];
            optNext[l].z = 0;
        }
        static const int order[4] = {LF, RH, RF, LH};  // cpp:50-51, 967-976
        // nominalIndex, cpp:965-976 (getMapIndex cpp:2557-2568: the index is written even when outside)
        for (int k = 0; k < 4; ++k) {
            Idx2 index;
            gaitMap_.getIndex({optNext[order[k]].x, optNext[order[k]].y}, index);
            oc.nominalIndex[2 * k] = index.i;
            oc.nominalIndex[2 * k + 1] = index.j;
        }
        // ---- STEP(3) the centroid method on the gait-cycle submap, cpp:1003-1013 ----
        CentroidResult centroidOfOpt[4];
        for (int l = 0; l < 4; ++l) {  // RF, RH, LH, LF = columns 0..3 of traversableRow
            const float R = ps.legRadius[l] > 0 ? ps.legRadius[l] : p.searchRadius;
            checkFootholdUseCentroidMethod(gaitMap_, {optNext[l].x, optNext[l].y}, R, p, centroidOfOpt[l], &gridmap_,
                                           &oc.traversableRow[0][l], &oc.traversableRow[1][l]);
            oc.centroidCode[l] = centroidOfOpt[l].code;
        }
        // centroidIndex, cpp:1030-1041 (an untouched result is the point (0,0))
        for (int k = 0; k < 4; ++k) {
            Idx2 index;
            gaitMap_.getIndex({centroidOfOpt[order[k]].x, centroidOfOpt[order[k]].y}, index);
            oc.centroidIndex[2 * k] = index.i;
            oc.centroidIndex[2 * k + 1] = index.j;
        }
        // xBounds, cpp:1057-1076 (Eigen::MatrixXi: every assignment truncates a double toward zero)
        int* lo = oc.xLower;
        int* up = oc.xUpper;
        lo[1] = lo[7] = 0;
        up[1] = up[7] = static_cast<int>(footSearchRectCol);
        lo[3] = lo[5] = static_cast<int>(isosWidth/mapResolution - footSearchRectCol);
        up[3] = up[5] = static_cast<int>(isosWidth/mapResolution);
        lo[0] = oc.traversableRow[0][3]; up[0] = oc.traversableRow[1][3];  // x1: LF
        lo[2] = oc.traversableRow[0][1]; up[2] = oc.traversableRow[1][1];  // x3: RH
        lo[4] = oc.traversableRow[0][0]; up[4] = oc.traversableRow[1][0];  // x5: RF
        lo[6] = oc.traversableRow[0][2]; up[6] = oc.traversableRow[1][2];  // x7: LH
        // ---- STEP(4) the optimiser (build-defined) ----
        if (forcedX && gaitCycleIndex < nForced) {
            for (int k = 0; k < 8; ++k) oc.x[k] = forcedX[(size_t)gaitCycleIndex * 8 + k];
            oc.minf = 0.0;
            oc.solverStatus = 255;  // not this file's optimiser
        } else
        oc.solverStatus = static_cast<uint8_t>(solveLattice(op, oc.nominalIndex, oc.centroidIndex, lo, up, lengthBase, skew,
                                                            mapResolution, lfCurrentRow, rhCurrentRow, oc.x, &oc.minf));
        // ---- STEP(6) positions and heights on the gait-cycle submap, cpp:1283-1314 ----
        Vec2 pp{0, 0};  // ONE Position for the four conversions (uninitialised in the reference; a failed getPosition keeps it)
        Point3 result[4];
        for (int k = 0; k < 4; ++k) {
            Idx2 i;
            i.i = static_cast<int>(oc.x[2 * k]);      // cpp:1287: Index = double -> int
            i.j = static_cast<int>(oc.x[2 * k + 1]);
            gaitMap_.getPosition(i, pp);
            const int l = order[k];
            result[l].z = getFootholdMeanHeight(gaitMap_, pp, p.footRadius, p.h);
            result[l].x = pp.x;
            result[l].y = pp.y;
        }
        for (int l = 0; l < 4; ++l) oc.result[l] = result[l];
        // ---- commit with the other tracks, cpp:1332, 1485-1568 ----
        if (cycleOk[gaitCycleIndex]) {
            getHipDistance(result, out.feetDistance);                        // cpp:1488
            getCogSpeed(result, optCurrent, p.RF_FIRST, out.cogSpeed);       // cpp:1497
            oc.committed = 1;
            for (int l = 0; l < 4; ++l) optCurrent[l] = result[l];           // cpp:1553-1556
            Idx2 ii;
            gaitMap_.getIndex({optCurrent[LF].x, optCurrent[LF].y}, ii);     // cpp:1561-1564
            lfCurrentRow = ii.i;
            gaitMap_.getIndex({optCurrent[RH].x, optCurrent[RH].y}, ii);     // cpp:1565-1568
            rhCurrentRow = ii.i;
        }
        ajustedPoseY += p.lateralDrift;  // cpp:1578
    }
}

}  // namespace fpo
