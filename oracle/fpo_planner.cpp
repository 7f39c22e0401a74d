// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see fpo_planner.hpp / fpo_gridmap.hpp).
// Restates /root/reference/foothold_planner/src/FootholdPlanner.cpp ("cpp:") for the hot path.
#include "fpo_planner.hpp"

#include <cmath>

namespace fpo {

// ---- by-value copy emulation (see fpo_planner.hpp) ----------------------------------------------
static thread_local bool g_emulateCopies = false;
static thread_local unsigned long long g_copyCount = 0;
static thread_local std::vector<float> g_sinkTrav, g_sinkElev;
void setEmulateByValueCopies(bool on) { g_emulateCopies = on; g_copyCount = 0; }
unsigned long long byValueCopyCount() { return g_copyCount; }
static inline void byValueCopy(const GridMap& map) {
    if (!g_emulateCopies) return;
    g_sinkTrav = map.trav;  // deep copy of each layer, as GridMap's copy constructor does
    g_sinkElev = map.elev;
    asm volatile("" ::"r"(g_sinkTrav.data()), "r"(g_sinkElev.data()) : "memory");
    ++g_copyCount;
}

// initialize(), cpp:340-421.  lengthBase/widthBase are FLOAT members (hpp:666-667); widthBase is
// computed in f32 (cpp:341); every later use promotes the float to double first.
Constants makeConstants(const Params& p) {
    Constants c;
    const float lengthBase = p.length;            // cpp:340
    const float widthBase = p.width + p.l1 * 2;   // cpp:341 (f32 arithmetic)
    c.LbHalf = lengthBase * 0.5;                  // cpp:350
    c.WbHalfNeg = -widthBase * 0.5;               // cpp:351
    c.WbHalfPos = widthBase * 0.5;                // cpp:359
    if (p.RF_FIRST) {                             // cpp:403-411
        c.biasX[RF] = 0.5 * lengthBase + p.skew;
        c.biasY[RF] = -0.5 * widthBase;
        c.biasX[RH] = -0.5 * lengthBase - p.skew;
        c.biasY[RH] = -0.5 * widthBase;
        c.biasX[LH] = -0.5 * lengthBase + p.skew;
        c.biasY[LH] = 0.5 * widthBase;
        c.biasX[LF] = 0.5 * lengthBase - p.skew;
        c.biasY[LF] = 0.5 * widthBase;
    } else {                                      // cpp:412-421
        c.biasX[RF] = 0.5 * lengthBase - p.skew;
        c.biasY[RF] = -0.5 * widthBase;
        c.biasX[RH] = -0.5 * lengthBase + p.skew;
        c.biasY[RH] = -0.5 * widthBase;
        c.biasX[LH] = -0.5 * lengthBase - p.skew;
        c.biasY[LH] = 0.5 * widthBase;
        c.biasX[LF] = 0.5 * lengthBase + p.skew;
        c.biasY[LF] = 0.5 * widthBase;
    }
    c.stepHalf = p.stepLength / 2;     // cpp:2693: float / int -> float, then promoted
    c.step = p.stepLength;             // cpp:2199
    c.stepQuarter = p.stepLength / 4;  // walk extension (build-defined)
    return c;
}

// cpp:2039-2082.  false iff the disc is empty or a FINITE cell is below the threshold; NaN cells
// are ignored; `validation = true` closes every iteration (cpp:2078).
bool checkDefaultFoothold(const GridMap& map, const Vec2& center, float footRadius, const Params& p) {
    byValueCopy(map);  // hpp:110 / cpp:2012
    bool validation = false;
    for (CircleIterator it(map, center, footRadius); !it.isPastEnd(); ++it) {
        const float v = map.travAt(*it);
        if (GridMap::isValid(v)) {                        // cpp:2055
            if (v < p.defaultFootholdThreshold) {         // cpp:2057
                validation = false;
                break;                                    // cpp:2066
            }
        }
        validation = true;                                // cpp:2078
    }
    return validation;
}

// cpp:2117-2163.
bool checkCirclePolygonFoothold(const GridMap& map, const Vec2& center, float footRadius,
                                const Polygon& polygon, const Params& p) {
    byValueCopy(map);  // hpp:140 / cpp:2100: one whole-map copy per spiral candidate
    bool validation = false;
    for (CircleIterator it(map, center, footRadius); !it.isPastEnd(); ++it) {
        const float v = map.travAt(*it);
        if (GridMap::isValid(v)) {                        // cpp:2132
            Vec2 cell{0, 0};
            map.getPosition(*it, cell);                   // cpp:2136
            if (v < p.candidateFootholdThreshold || false == polygon.isInside(cell)) {  // cpp:2138
                validation = false;
                break;                                    // cpp:2147
            }
        }
        validation = true;                                // cpp:2159
    }
    return validation;
}

// cpp:2085-2114: first valid cell in SpiralIterator order.
bool checkCandidateFoothold(const GridMap& map, const Vec2& spiralCenter, float footRadius,
                            float searchRadius, const Polygon& polygon, const Params& p, LegResult& out) {
    byValueCopy(map);  // hpp:124 / cpp:2022
    bool validation = false;
    for (SpiralIterator it(map, spiralCenter, searchRadius); !it.isPastEnd(); ++it) {
        Vec2 footCenter{0, 0};
        map.getPosition(*it, footCenter);                 // cpp:2098
        validation = checkCirclePolygonFoothold(map, footCenter, footRadius, polygon, p);  // cpp:2100
        if (validation) {
            Vec2 q{0, 0};
            map.getPosition(*it, q);                      // cpp:2105
            out.x = q.x;
            out.y = q.y;
            out.row = (*it).i;
            out.col = (*it).j;
            break;
        }
    }
    return validation;
}

// Oracle-defined guard (NOT in the reference): a search centre must be finite and of sane
// magnitude.  A NaN centre is reachable once the centroid track has committed its "no case"
// (0,0,0) results (cpp:1777-1944) and the feet polygon degenerates; the reference then feeds NaN to
// getIndexFromPosition's (int) cast — undefined behaviour.  Defined here (and in the engine) as
// "no cell is visited": invalid leg, getSubmap failure, mean height = h.
static bool centreUsable(const Vec2& c) { return std::fabs(c.x) <= 1e6 && std::fabs(c.y) <= 1e6; }

// cpp:2520-2554.  f32 sequential sum in CircleIterator (row-major bbox) order; NaN -> 0.0 and
// counted; values >= 10 skipped; empty count -> last iHeight; "+ h" in f64, returned as float.
float getFootholdMeanHeight(const GridMap& map, const Vec2& center, float radius, double h) {
    byValueCopy(map);  // hpp:262 / cpp:2029, 2292-2301, 1687, 1820
    float iHeight = 0.0, meanHeight = 0.0;
    int i = 0;
    if (!centreUsable(center)) return static_cast<float>(meanHeight + h);
    for (CircleIterator it(map, center, radius); !it.isPastEnd(); ++it) {
        const float e = map.elevAt(*it);
        if (GridMap::isValid(e)) {
            iHeight = e;
        } else {
            iHeight = 0.0;
        }
        if (iHeight < 10) {
            i++;
            meanHeight = meanHeight + iHeight;
        }
    }
    if (i != 0) {
        meanHeight = meanHeight / i;
    } else {
        meanHeight = iHeight;
    }
    return static_cast<float>(meanHeight + h);
}

// cpp:2001-2036.  z is measured at the DEFAULT centre even when a candidate was chosen (cpp:2029).
void checkFoothold(const GridMap& map, const Vec2& center, float footRadius, float searchRadius,
                   const Polygon& polygon, const Params& p, LegResult& out) {
    byValueCopy(map);  // cpp:863-869: std::thread(&checkFoothold, this, gridmap_, ...) copies the argument
    byValueCopy(map);  // ... and checkFoothold receives it by value again (hpp:94)
    out = LegResult();
    if (!centreUsable(center)) {  // oracle-defined, see centreUsable
        out.x = center.x;
        out.y = center.y;
        return;
    }
    bool defaultFootholdIsvalid = checkDefaultFoothold(map, center, footRadius, p);  // cpp:2012
    out.valid = defaultFootholdIsvalid;
    out.x = center.x;                                     // cpp:2016-2017
    out.y = center.y;
    bool candidateFootholdIsvalid = false;
    if (!defaultFootholdIsvalid) {
        candidateFootholdIsvalid = checkCandidateFoothold(map, center, footRadius, searchRadius, polygon, p, out);
        out.valid = candidateFootholdIsvalid;             // cpp:2023
    }
    if (defaultFootholdIsvalid | candidateFootholdIsvalid) {
        out.z = getFootholdMeanHeight(map, center, footRadius, p.h);  // cpp:2029
    }
    if (defaultFootholdIsvalid) {
        Idx2 idx;
        map.getIndex(center, idx);  // reporting only
        out.row = idx.i;
        out.col = idx.j;
        out.source = 0;
    } else if (candidateFootholdIsvalid) {
        out.source = 1;
    } else {
        out.source = 2;
        out.row = out.col = -1;
    }
}

// cpp:1605-1997.  `searchRadius` is searchRadius_ in the reference (cpp:1616-1617).
// `gridmap` is the map the rectangle is cut from — gridmap_ for the centroid track (cpp:818-821), the gait-cycle submap
// gaitMap_ for the opt track (cpp:1010-1013); the result's height is always measured on the member gridmap_
// (`heightMap`, cpp:1687, 1820).  traversableBeginRow / traversableEndRow (cpp:1608-1609): first / last row of the
// traversable band in `gridmap`'s rows; the reference leaves them untouched on the paths that set no result (getSubmap
// failure, no case) — there the caller's storage is uninitialised (Eigen::MatrixXi(2,4), cpp:1009): oracle-defined 0.
void checkFootholdUseCentroidMethod(const GridMap& gridmap, const Vec2& p, float searchRadius,
                                    const Params& prm, CentroidResult& out, const GridMap* heightMapIn,
                                    int* traversableBeginRow, int* traversableEndRow) {
    const GridMap& heightMap = heightMapIn ? *heightMapIn : gridmap;
    out = CentroidResult();
    if (traversableBeginRow) *traversableBeginRow = 0;
    if (traversableEndRow) *traversableEndRow = 0;
    if (!centreUsable(p)) {  // oracle-defined, see centreUsable
        out.code = 6;
        return;
    }
    Vec2 rect;
    rect.x = searchRadius * 2;  // cpp:1616 (float * int -> float)
    rect.y = searchRadius;      // cpp:1617
    bool isSuccess;
    GridMap map = gridmap.getSubmap(p, rect, isSuccess);  // cpp:1627
    if (!isSuccess) {
        out.code = 6;  // cpp:1628-1631: return false, result untouched
        return;
    }
    // cpp:1649-1658: whole-region test, raw `<` (NaN passes, -inf blocks), linear storage order
    bool wholeRegionValid = false;
    const size_t n = map.trav.size();
    for (size_t i = 0; i < n; ++i) {
        if (map.trav[i] < prm.defaultFootholdThreshold) {
            wholeRegionValid = false;
            break;
        }
        wholeRegionValid = true;
    }
    const int topRow = 0, bottomRow = map.size.i - 1, rightCol = map.size.j - 1;  // cpp:1679-1682

    // cpp:1692-1710 (and the same block in every case): the band's first / last row of the rectangle, mapped to rows of
    // `gridmap` through the position of the rectangle's cell (row, 1).  One Position is reused for both conversions
    // (a failed getPosition leaves it as it was; the reference's local is uninitialised before the first: oracle (0,0)).
    auto bandRows = [&](int beginRow, int endRow) {
        if (!traversableBeginRow && !traversableEndRow) return;
        Vec2 q{0, 0};
        Idx2 i2;
        map.getPosition({beginRow, 1}, q);
        gridmap.getIndex(q, i2);
        if (traversableBeginRow) *traversableBeginRow = i2.i;
        map.getPosition({endRow, 1}, q);
        gridmap.getIndex(q, i2);
        if (traversableEndRow) *traversableEndRow = i2.i;
    };
    auto finish = [&](const Vec2& q, uint8_t code) {
        out.z = getFootholdMeanHeight(heightMap, q, prm.footRadius, prm.h);  // on gridmap_ (cpp:1687, 1820)
        out.x = q.x;
        out.y = q.y;
        out.code = code;
        Idx2 idx;
        gridmap.getIndex(q, idx);  // reporting only
        out.row = idx.i;
        out.col = idx.j;
    };

    if (wholeRegionValid) {  // cpp:1684-1689
        finish(p, 0);
        bandRows(topRow, bottomRow);  // cpp:1692-1693
        return;
    }
    // cpp:1717-1750: row scan.  The reference's LineIterator end index (row, size(1)) reads one
    // cell past the last column (UB); the oracle scans columns 0..size(1)-1 only (App. D).
    int minRow = 0, maxRow = 0, k = 0;
    for (int j = 0; j < bottomRow + 1; ++j) {
        int i = 0;
        for (int c = 0; c < map.size.j; ++c)
            if (map.travAt({j, c}) < prm.defaultFootholdThreshold) ++i;  // cpp:1736
        if (i > ((rightCol + 1) * 0.5)) {  // cpp:1743
            if (k == 0) minRow = j;
            maxRow = j;
            ++k;
        }
    }
    Idx2 newIndex;
    uint8_t code;
    if (minRow == topRow && maxRow != bottomRow) {  // case 1, cpp:1777-1786
        newIndex.i = static_cast<int>(std::floor((maxRow + bottomRow + 1) * 0.5));
        newIndex.j = static_cast<int>(std::floor((rightCol + 1) * 0.5));
        code = 1;
        bandRows(maxRow + 1, bottomRow);  // cpp:1794-1795
    } else if (minRow != topRow && maxRow != bottomRow) {  // case 2, cpp:1843-1886
        if ((minRow - topRow) >= (bottomRow - maxRow)) {
            newIndex.i = static_cast<int>(std::ceil(minRow * 0.5));
            newIndex.j = static_cast<int>(std::floor((rightCol + 0) * 0.5));
            code = 2;
            bandRows(topRow, minRow - 1);  // cpp:1858-1859
        } else {
            newIndex.i = static_cast<int>(std::floor((maxRow + bottomRow) * 0.5));
            newIndex.j = static_cast<int>(std::floor((rightCol + 0) * 0.5));
            code = 3;
            bandRows(maxRow + 1, bottomRow);  // cpp:1889-1890
        }
    } else if (minRow != topRow && maxRow == bottomRow) {  // case 3, cpp:1944-1952
        newIndex.i = static_cast<int>(std::ceil(minRow * 0.5));
        newIndex.j = static_cast<int>(std::floor((rightCol + 0) * 0.5));
        code = 4;
        bandRows(topRow, minRow - 1);  // cpp:1960-1961
    } else {
        out.code = 5;  // minRow == topRow && maxRow == bottomRow: no branch, result stays (0,0,0)
        return;
    }
    Vec2 newRegionCentroid{0, 0};
    map.getPosition(newIndex, newRegionCentroid);  // cpp:1816 (on the SUBMAP)
    finish(newRegionCentroid, code);
}

// cpp:2421-2463.
Point3 getPolygonCenter(const Point3& rf, const Point3& rh, const Point3& lh, const Point3& lf) {
    int n = 4;
    double x1, y1, x2, y2, x3 = 0, y3 = 0;
    double sum_x = 0, sum_y = 0, sum_s = 0;
    x1 = rf.x;
    y1 = rf.y;
    x2 = rh.x;
    y2 = rh.y;
    for (int i = 1; i <= n - 2; i++) {
        switch (i) {
            case 1:
                x3 = lh.x;
                y3 = lh.y;
                break;
            case 2:
                x3 = lf.x;
                y3 = lf.y;
                break;
        }
        double s = ((x2 - x1) * (y3 - y1) - (x3 - x1) * (y2 - y1)) / 2.0;
        sum_x += (x1 + x2 + x3) * s;
        sum_y += (y1 + y2 + y3) * s;
        sum_s += s;
        x2 = x3;
        y2 = y3;
    }
    Point3 c;
    c.x = sum_x / sum_s / 3.0;
    c.y = sum_y / sum_s / 3.0;
    c.z = (rf.z + rh.z + lh.z + lf.z) / 4.0;
    return c;
}

// cpp:2307-2349, the part that decides the return value: getSubmap of isos_.length x isos_.width around the
// next feet centre.  isos_ (cpp:384-394, hpp:677-697): longEdge = lengthBase + skew*2 evaluated in f32
// (hpp:666, 683) and stored in a double; footSearchRect_.length = searchRadius_*2 / .width = searchRadius_
// are f32 values stored in doubles (hpp:700-701); isos_.length / .width are sums of doubles.
bool getGaitCycleSearchGridMap(const GridMap& gridmap, const Params& prm, const Point3 cur[4], double initialPoseY,
                               double ajustedPoseY) {
    const Point3 feetCenter = getPolygonCenter(cur[RF], cur[RH], cur[LH], cur[LF]);  // cpp:2322
    Vec2 p;
    p.x = feetCenter.x + prm.stepLength;    // cpp:2327
    p.y = initialPoseY + ajustedPoseY;      // cpp:2329
    if (!centreUsable(p)) return false;     // oracle-defined, see centreUsable
    const float lengthBase = prm.length;                // cpp:340
    const float widthBase = prm.width + prm.l1 * 2;     // cpp:341
    const double longEdge = lengthBase + prm.skew * 2;  // cpp:391 (f32 sum, promoted on assignment)
    const double shortEdge = widthBase;                 // cpp:392
    const double rectLength = prm.searchRadius * 2;     // cpp:384
    const double rectWidth = prm.searchRadius;          // cpp:385
    Vec2 rect;
    rect.x = longEdge + rectLength;  // cpp:393, 2340
    rect.y = shortEdge + rectWidth;  // cpp:394, 2341
    bool isSuccess = false;
    (void)gridmap.getSubmap(p, rect, isSuccess);  // cpp:2345
    return isSuccess;                             // cpp:2347-2349
}

// cpp:2496-2517 (kind 0).  Vertex order LU, RU, RD, LD; `radius` is float, promoted per use.
// kind 1 (build-defined, App. E): flattened hexagon with the same x extent and y half-extent
// 0.5*r*kHexH, all vertices from products of doubles so host and device agree bit for bit.
Polygon getSearchPolygon(const Point3& center, float radius, int kind) {
    Polygon polygon;
    if (kind == 0) {
        polygon.addVertex({center.x + radius, center.y + 0.5 * radius});
        polygon.addVertex({center.x + radius, center.y - 0.5 * radius});
        polygon.addVertex({center.x - radius, center.y - 0.5 * radius});
        polygon.addVertex({center.x - radius, center.y + 0.5 * radius});
    } else {
        const double r = radius;
        const double hx = 0.5 * r;
        const double hy = (0.5 * r) * 0.8660254037844386;
        polygon.addVertex({center.x + r, center.y});
        polygon.addVertex({center.x + hx, center.y - hy});
        polygon.addVertex({center.x - hx, center.y - hy});
        polygon.addVertex({center.x - r, center.y});
        polygon.addVertex({center.x - hx, center.y + hy});
        polygon.addVertex({center.x + hx, center.y + hy});
    }
    return polygon;
}

// getHipDistance, cpp:2571-2584: x distance of each diagonal pair.
void getHipDistance(const Point3 result[4], std::vector<double>& feetDistance) {
    double feetDistance1, feetDistance2;
    feetDistance1 = (result[RF].x - result[LH].x);
    feetDistance.push_back(feetDistance1);
    feetDistance2 = (result[LF].x - result[RH].x);
    feetDistance.push_back(feetDistance2);
}

// getCogSpeed, cpp:2587-2623: the two half-cycle COG displacements over 0.5*gaitCycle_ (gaitCycle_ = 1.0, cpp:332).
void getCogSpeed(const Point3 result[4], const Point3 current[4], int RF_FIRST, std::vector<double>& cogSpeed) {
    const double gaitCycle = 1.0;
    double feetCenter1, feetCenter2, feetCenter3;
    if (RF_FIRST) {
        feetCenter1 = (current[RF].x + current[LH].x) / 2;
        feetCenter2 = (result[LF].x + result[RH].x) / 2;
        feetCenter3 = (result[RF].x + result[LH].x) / 2;
    } else {
        feetCenter1 = (current[LF].x + current[RH].x) / 2;
        feetCenter2 = (result[RF].x + result[LH].x) / 2;
        feetCenter3 = (result[LF].x + result[RH].x) / 2;
    }
    double cogMovedDistance1 = feetCenter2 - feetCenter1;
    double cogMovedDistance2 = feetCenter3 - feetCenter2;
    cogSpeed.push_back(cogMovedDistance1 / (0.5 * gaitCycle));
    cogSpeed.push_back(cogMovedDistance2 / (0.5 * gaitCycle));
}

// globalFootholdPlan, cpp:539-1602, for one initial pose.  Tracks: 0 default, 1 centroid, 2 nominal.
// A gait cycle is a sequence of phases, each with a set of swing legs and a centre advance:
//   trot (reference): ONE phase, all four legs, advance stepLength_ (cpp:762-1579);
//   walk (build-defined): four single-leg phases, advance stepLength_/4 each, swing order
//        LF,RH,RF,LH (RF_FIRST=false) or RF,LH,LF,RH; a phase commits iff its swing leg is valid;
//        cycleOk = AND over the phases; the lateral drift is applied once per cycle.
void planGlobalFootholds(const GridMap& map, const Params& p, const PoseSpec& ps, int nCycles,
                         PlanOutput& out) {
    const Constants c = makeConstants(p);
    const double* pose = ps.pose;
    out.nominal.assign((size_t)nCycles * 4, LegResult());
    out.centroid.assign((size_t)nCycles * 4, CentroidResult());
    out.defaultNext.assign((size_t)nCycles * 4, Point3());
    out.cycleOk.assign((size_t)nCycles, 0);
    for (int k = 0; k < 2; ++k) {  // cpp:606-611, 635-636
        out.feetCenterPath[k].clear();
        out.feetDistance[k].clear();
        out.cogSpeed[k].clear();
    }

    // cpp:350-378: initial stance = hip rectangle + initialPose_
    const double sx[4] = {c.LbHalf, -c.LbHalf, -c.LbHalf, c.LbHalf};
    const double sy[4] = {c.WbHalfNeg, c.WbHalfNeg, c.WbHalfPos, c.WbHalfPos};
    for (int l = 0; l < 4; ++l) {
        Point3 s;
        s.x = sx[l];
        s.y = sy[l];
        s.z = 0;
        s.x += pose[0];
        s.y += pose[1];
        s.z += pose[2];
        out.stance[l] = s;
    }
    // setFirstGait, cpp:2679-2699 (x -= stepLength_/2), for every track (cpp:562-588)
    Point3 cur[3][4];
    for (int t = 0; t < 3; ++t)
        for (int l = 0; l < 4; ++l) {
            cur[t][l] = out.stance[l];
            cur[t][l].x = out.stance[l].x - c.stepHalf;
        }
    double ajustedPoseY = 0.0;  // cpp:759
    // opt track, first cycle (cpp:916-934): its current feet are the shifted stance too (cpp:582-588)
    out.optGate0Failed = (nCycles > 0 && !getGaitCycleSearchGridMap(map, p, cur[0], pose[1], ajustedPoseY)) ? 1 : 0;

    static const int walkOrderLF[4] = {LF, RH, RF, LH};
    static const int walkOrderRF[4] = {RF, LH, LF, RH};
    const int nPhases = ps.gait == 1 ? 4 : 1;
    const double advance = ps.gait == 1 ? c.stepQuarter : c.step;

    for (int g = 0; g < nCycles; ++g) {  // cpp:762
        bool cycleOk = true;
        for (int ph = 0; ph < nPhases; ++ph) {
            unsigned mask = 0xF;
            if (ps.gait == 1) mask = 1u << (p.RF_FIRST ? walkOrderRF[ph] : walkOrderLF[ph]);

            Point3 next[3][4];
            Polygon nominalPoly[4];
            for (int t = 0; t < 3; ++t) {
                // getDefaultFootholds cpp:2265-2284 / getFootholdSearchGridMap cpp:2191-2213
                Point3 C = getPolygonCenter(cur[t][RF], cur[t][RH], cur[t][LH], cur[t][LF]);
                if (t >= 1 && ps.gait == 0) out.feetCenterPath[t - 1].push_back(C);  // cpp:2194-2196
                Point3 N;
                N.x = C.x + advance;              // cpp:2199 / 2270
                N.y = pose[1] + ajustedPoseY;     // cpp:2201 / 2272
                N.z = C.z;
                for (int l = 0; l < 4; ++l) {     // getDefaultFootholdNext, cpp:2411-2418 (z = 0)
                    next[t][l].x = N.x + c.biasX[l];
                    next[t][l].y = N.y + c.biasY[l];
                    next[t][l].z = 0;
                }
            }
            LegResult nom[4];
            CentroidResult cen[4];
            bool phaseOk = true;
            for (int l = 0; l < 4; ++l) {
                if (!(mask & (1u << l))) continue;
                const float R = ps.legRadius[l] > 0 ? ps.legRadius[l] : p.searchRadius;
                // default track height, cpp:2289-2301
                next[0][l].z = getFootholdMeanHeight(map, {next[0][l].x, next[0][l].y}, p.footRadius, p.h);
                // centroid track, cpp:818-821
                checkFootholdUseCentroidMethod(map, {next[1][l].x, next[1][l].y}, R, p, cen[l]);
                // nominal track: centre from the CENTROID track, polygon from the NOMINAL track
                // (cpp:861-869)
                nominalPoly[l] = getSearchPolygon(next[2][l], R, ps.legPoly[l]);  // cpp:2235-2244
                checkFoothold(map, {next[1][l].x, next[1][l].y}, p.footRadius, R, nominalPoly[l], p, nom[l]);
                phaseOk = phaseOk && nom[l].valid;  // cpp:1323
                out.nominal[(size_t)g * 4 + l] = nom[l];
                out.centroid[(size_t)g * 4 + l] = cen[l];
                out.defaultNext[(size_t)g * 4 + l] = next[0][l];
            }
            if (phaseOk && ps.gait == 0) {  // KPIs use the track's current feet BEFORE the commit
                Point3 nomP[4], cenP[4];
                for (int l = 0; l < 4; ++l) {
                    nomP[l].x = nom[l].x; nomP[l].y = nom[l].y; nomP[l].z = nom[l].z;
                    cenP[l].x = cen[l].x; cenP[l].y = cen[l].y; cenP[l].z = cen[l].z;
                }
                getHipDistance(nomP, out.feetDistance[1]);               // cpp:1357
                getCogSpeed(nomP, cur[2], p.RF_FIRST, out.cogSpeed[1]);  // cpp:1366
                getHipDistance(cenP, out.feetDistance[0]);               // cpp:1422
                getCogSpeed(cenP, cur[1], p.RF_FIRST, out.cogSpeed[0]);  // cpp:1430
            }
            if (phaseOk) {  // cpp:1332-1483: commit every track
                for (int l = 0; l < 4; ++l) {
                    if (!(mask & (1u << l))) continue;
                    cur[0][l] = next[0][l];                                   // cpp:1338-1341
                    cur[2][l].x = nom[l].x;                                   // cpp:1413-1416
                    cur[2][l].y = nom[l].y;
                    cur[2][l].z = nom[l].z;
                    cur[1][l].x = cen[l].x;                                   // cpp:1480-1483
                    cur[1][l].y = cen[l].y;
                    cur[1][l].z = cen[l].z;
                }
            }
            cycleOk = cycleOk && phaseOk;  // cpp:1571-1576: nothing advances on failure
        }
        out.cycleOk[g] = cycleOk;
        ajustedPoseY += p.lateralDrift;  // cpp:1578
    }
}

}  // namespace fpo
