// ORACLE — TEST INFRASTRUCTURE ONLY (see fpo_gridmap.hpp header).  PARITY UNPINNED: the
// reference has no tests / golden vectors for this path; fidelity is argued by citation.
//
// CPU restatement of the reference's foothold-search hot path
// (/root/reference/foothold_planner/src/FootholdPlanner.cpp, "cpp:" below; header "hpp:").
// Every function cites the reference lines it follows and keeps the reference's float/double
// typing and expression order.  Deliberate divergences:
//   * maps are passed by const reference, not by value (hpp:94-143 deep-copy every layer);
//   * the centroid row scan stays in bounds (cpp:1719-1720 reads one cell past the last column);
//   * no ROS publishing / logging; the "opt" track (cpp:913-1319) is restated in fpo_opt.cpp with a BUILD-DEFINED
//     optimiser in NLopt's place (NLopt is absent and unpinned).
// Build-defined extensions (NOT in the reference): batch of poses, 4-phase walk
// gait, per-leg search radius / polygon kind.  With gait = trot, radius override <= 0 and polygon
// kind 0 the code path is exactly the reference's.
#pragma once
#include <cstdint>
#include <vector>

#include "fpo_gridmap.hpp"

namespace fpo {

enum Leg { RF = 0, RH = 1, LH = 2, LF = 3 };  // foot_id order, cpp:685-699

// ROS params of the path (cpp:248-314) with the reference's types (hpp:609-619, 657-697).
struct Params {
    float footRadius;                  // cpp:255
    float defaultFootholdThreshold;    // cpp:257
    float candidateFootholdThreshold;  // cpp:258
    float searchRadius;                // cpp:261
    float stepLength;                  // cpp:262
    float length, width, l1;           // laikago_kinematics, cpp:285-287
    float skew;                        // isos_.skew, cpp:290
    int RF_FIRST;                      // cpp:264
    double h;                          // h_ = 0.01, cpp:336
    double lateralDrift;               // ajustedPose_[1] += -0.007, cpp:1578
};

// One body pose of the batch.  Reference: initialPose_ (cpp:293-295); the rest is build-defined.
struct PoseSpec {
    double pose[3];
    int gait;            // 0 trot (reference), 1 walk (build-defined)
    float legRadius[4];  // <= 0: use Params::searchRadius
    int legPoly[4];      // 0 reference rectangle (cpp:2496-2517), 1 build-defined hexagon
};

struct Point3 {
    double x = 0, y = 0, z = 0;
};

// One nominal-track leg result (checkFoothold, cpp:2001-2036).
struct LegResult {
    int row = -1, col = -1;  // chosen grid index
    double x = 0, y = 0;
    float z = 0;         // getFootholdMeanHeight returns float (cpp:2520); z stays 0 when invalid
    uint8_t valid = 0;   // footholdValidation
    uint8_t source = 2;  // 0 default-disc hit, 1 spiral candidate, 2 none
};

// One centroid-track leg result (checkFootholdUseCentroidMethod, cpp:1605-1997).
struct CentroidResult {
    double x = 0, y = 0;
    float z = 0;
    int row = -1, col = -1;  // index of (x, y) on the full map, -1 when the result is untouched
    // 0 whole region valid; 1 case 1; 2 case 2 upper band; 3 case 2 lower band; 4 case 3;
    // 5 no case (first and last row blocked, result left (0,0,0)); 6 getSubmap failed (0,0,0)
    uint8_t code = 5;
};

struct Constants {
    double LbHalf, WbHalfNeg, WbHalfPos;  // lengthBase*0.5, -widthBase*0.5, widthBase*0.5
    double biasX[4], biasY[4];            // defaultBias, cpp:403-421
    double stepHalf;                      // double(stepLength_/2) (f32 division), cpp:2693
    double step;                          // double(stepLength_)
    double stepQuarter;                   // double(stepLength_/4) — walk extension only
};
Constants makeConstants(const Params& p);

// --- per-leg functions (names follow the reference) ---------------------------------------------
bool checkDefaultFoothold(const GridMap& map, const Vec2& center, float footRadius, const Params& p);
bool checkCirclePolygonFoothold(const GridMap& map, const Vec2& center, float footRadius,
                                const Polygon& polygon, const Params& p);
bool checkCandidateFoothold(const GridMap& map, const Vec2& spiralCenter, float footRadius,
                            float searchRadius, const Polygon& polygon, const Params& p, LegResult& out);
float getFootholdMeanHeight(const GridMap& map, const Vec2& center, float radius, double h);
void checkFoothold(const GridMap& map, const Vec2& center, float footRadius, float searchRadius,
                   const Polygon& polygon, const Params& p, LegResult& out);
void checkFootholdUseCentroidMethod(const GridMap& map, const Vec2& defaultFoothold, float searchRadius,
                                    const Params& p, CentroidResult& out, const GridMap* heightMap = nullptr,
                                    int* traversableBeginRow = nullptr, int* traversableEndRow = nullptr);
Point3 getPolygonCenter(const Point3& rf, const Point3& rh, const Point3& lh, const Point3& lf);
bool getGaitCycleSearchGridMap(const GridMap& gridmap, const Params& p, const Point3 optCurrent[4], double initialPoseY,
                               double ajustedPoseY);
Polygon getSearchPolygon(const Point3& center, float radius, int kind = 0);
void getHipDistance(const Point3 result[4], std::vector<double>& feetDistance);
void getCogSpeed(const Point3 result[4], const Point3 current[4], int RF_FIRST, std::vector<double>& cogSpeed);

// --- the chained plan for ONE initial pose (globalFootholdPlan, cpp:539-1602) --------------------
struct PlanOutput {
    // arrays sized nCycles*4 (index g*4+leg) except cycleOk (nCycles)
    std::vector<LegResult> nominal;
    std::vector<CentroidResult> centroid;
    std::vector<Point3> defaultNext;  // default-track next positions with z (cpp:774-781)
    std::vector<uint8_t> cycleOk;     // footholdValidation_ per cycle (cpp:1323)
    Point3 stance[4];                 // RF/RH/LH/LF_initialPosition_ (cpp:350-378)
    // getGaitCycleSearchGridMap (cpp:2307-2349) fails in the FIRST gait cycle => the reference's service handler
    // returns false (cpp:920-934).  Later cycles: planOptTrack (the opt track's own chain).
    uint8_t optGate0Failed = 0;
    // Evaluation products of the centroid ([0]) and nominal ([1]) tracks (trot only):
    //   feetCenterPath: one pose per planned cycle = getPolygonCenter of the track's CURRENT feet,
    //                   pushed in getFootholdSearchGridMap whether or not the cycle commits (cpp:2191-2196);
    //   feetDistance / cogSpeed: two entries per COMMITTED cycle (getHipDistance cpp:2571-2584,
    //                   getCogSpeed cpp:2587-2623; called at cpp:1357-1368 and cpp:1422-1432).
    std::vector<Point3> feetCenterPath[2];
    std::vector<double> feetDistance[2];
    std::vector<double> cogSpeed[2];
};
void planGlobalFootholds(const GridMap& map, const Params& p, const PoseSpec& pose, int nCycles,
                         PlanOutput& out);

// --- the opt track (fpo_opt.cpp; SURVEY.md §8(f) N4: cpp:54-148, 913-1319, 1485-1570, 2307-2408, 2557-2568) -----------
// nlopt/* ROS parameters (cpp:297-307, yaml:53-63) and the file-scope constants of cpp:28-51.
struct OptParams {
    double w1, w2, w3, w4, wr, wc;       // cpp:297-303
    int useInequalityConstraits;         // cpp:306 (code default false, yaml true)
    int pad;
    double ctol;                         // cpp:34: 1e-2
    double hipLowerScale, hipUpperScale;    // cpp:48: 0.9, 1.1
    double skewLowerScale, skewUpperScale;  // cpp:49: 0.8, 1.2
    double lfCurrentRow0, rhCurrentRow0;    // cpp:36 at entry of the call: 0 at node start, else what the previous call left
};
constexpr long long kMaxLatticePoints = 1ll << 24;  // build-defined optimiser: row points it will enumerate
struct OptCycle {
    uint8_t gateFailed = 0;    // getGaitCycleSearchGridMap returned false (cpp:931-934): the service returns false
    uint8_t committed = 0;     // the cycle committed (cycleOk)
    uint8_t solverStatus = 0;  // solveLattice
    uint8_t centroidCode[4] = {0, 0, 0, 0};  // checkFootholdUseCentroidMethod on gaitMap_, RF,RH,LH,LF
    int gaitTopLeft[2] = {0, 0}, gaitSize[2] = {0, 0};  // gaitMap_ inside gridmap_
    int nominalIndex[8] = {0}, centroidIndex[8] = {0};  // LF,RH,RF,LH x (row, col) on gaitMap_ (cpp:965-976, 1030-1041)
    int traversableRow[2][4] = {{0}};                   // begin / end row x RF,RH,LH,LF (cpp:1009-1013)
    int xLower[8] = {0}, xUpper[8] = {0};               // xBounds (cpp:1057-1076)
    double x[8] = {0};                                  // the optimiser's result (LF,RH,RF,LH x (row, col))
    double minf = 0;
    double lfCurrentRow = 0, rhCurrentRow = 0;          // the values this cycle's objective used
    Point3 result[4];                                   // RF,RH,LH,LF_footholdResult_opt (cpp:1283-1314)
};
struct OptOutput {
    std::vector<OptCycle> cycles;
    int gateFailCycle = -1;              // first cycle whose gate failed, -1 none
    std::vector<Point3> feetCenterPath;  // this track's feet centres, pushed onto centroidFeetCenterPath (cpp:946)
    std::vector<double> feetDistance, cogSpeed;  // footholdsKPI_.feetDistance_opt / cogSpeed_opt (cpp:1488-1499)
};
int solveLattice(const OptParams& op, const int* nominalIndex, const int* centroidIndex, const int* xLower, const int* xUpper,
                 double lengthBase, double skew, double mapResolution, double lfCurrentRow, double rhCurrentRow, double* x,
                 double* minf);
void planOptTrack(const GridMap& gridmap_, const Params& p, const OptParams& op, const PoseSpec& pose, int nCycles,
                  const uint8_t* cycleOk, OptOutput& out, const double* forcedX = nullptr, int nForced = 0);

// "As-written" cost EMULATION: the reference passes grid_map::GridMap BY
// VALUE (hpp:94,110,124,140,262,288), i.e. deep-copies every layer at each of those call sites.  When
// enabled, the oracle performs an equivalent copy of its two layers at the same sites (results are
// unchanged).  It is an emulation of the reference's copy traffic, not a measurement of the reference.
void setEmulateByValueCopies(bool on);
unsigned long long byValueCopyCount();

}  // namespace fpo
