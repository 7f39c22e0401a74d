// fpe_host.cpp — host-side logic of the engine (no GPU calls in this file).
#include "fpe_host.hpp"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace fpe {

// ---- SpiralIterator visiting order ---------------------------------------------------------------
// grid_map SpiralIterator::generateRing walks ring d counter-clockwise from offset (d, 0), keeping
// to cells whose truncated Euclidean index distance equals d, and the iterator then pops cells from
// the BACK of the ring vector; so ring d is visited in the reverse of the walk.
// The table holds the unfiltered order; the in-map test and the in-radius test of the last two
// rings only remove entries, they never reorder (fpe_kernels.hip::candidate_search_wave).
namespace {
inline int sgn(int v) { return (v > 0) - (v < 0); }
inline int trunc_norm(int x, int y) {
    return static_cast<int>(std::sqrt(static_cast<double>(x) * x + static_cast<double>(y) * y));
}
}  // namespace

void build_spiral_table(int nRings, SpiralTable& out) {
    out.di.clear();
    out.dj.clear();
    out.ring.clear();
    out.ringStart.assign(1, 0);
    out.maxRing = nRings;
    out.di.push_back(0);
    out.dj.push_back(0);
    out.ring.push_back(0);
    std::vector<std::pair<int, int>> walk;
    for (int d = 1; d <= nRings; ++d) {
        out.ringStart.push_back(static_cast<int32_t>(out.di.size()));
        walk.clear();
        int px = d, py = 0;
        do {
            walk.emplace_back(px, py);
            const int nx = -sgn(py), ny = sgn(px);
            if (nx != 0 && trunc_norm(px + nx, py) == d) {
                px += nx;
            } else if (ny != 0 && trunc_norm(px, py + ny) == d) {
                py += ny;
            } else {
                px += nx;
                py += ny;
            }
        } while (px != d || py != 0);
        for (auto it = walk.rbegin(); it != walk.rend(); ++it) {
            out.di.push_back(static_cast<int16_t>(it->first));
            out.dj.push_back(static_cast<int16_t>(it->second));
            out.ring.push_back(static_cast<uint8_t>(d));
        }
    }
    out.ringStart.push_back(static_cast<int32_t>(out.di.size()));  // end sentinel = ringStart[nRings+1]
}

int spiral_rings(float searchRadius, double resolution) {
    const double R = static_cast<double>(searchRadius);
    return static_cast<int>(static_cast<unsigned int>(std::ceil(R / resolution)));
}

int tile_halfwidth(float maxSearchRadius, float footRadius, double resolution) {
    const double R = static_cast<double>(maxSearchRadius);
    const double rf = static_cast<double>(footRadius);
    const int nRings = static_cast<int>(std::ceil(R / resolution));
    const int nFootBox = static_cast<int>(std::ceil(rf / resolution)) + 1;  // bbox of a disc around any point
    // candidates reach nRings cells from the centre index and their discs nFootBox further; the
    // centroid rectangle reaches ceil(R/res)+1 rows; +1 guards the continuous-centre offset
    return nRings + nFootBox + 1;
}

// The bit-window kernels (fpe_bits.hpp) look every traversability decision up in a window of row masks around
// getIndex(centre).  They are exact when
//  (1) the foot-disc offset table is proved (derive_foot_offsets) and has at most 64 entries (one per lane of the
//      widest kernel; the 8-lane kernels take up to 32: fpe_bits.hpp::bits_supported);
//  (2) every cell a search can touch lies within `H` cells of the centre index.  For a bounded position p and the
//      centre c the index difference is floor-like in (p - c) / res; with rounding errors of at most `slack` cells
//      (positions are rewritten by boundPositionToRange and divided by res in f64) the reach of an extent d is
//      floor(d / res + frac) <= ceil(d / res) cells, one more when d / res is within `slack` of an integer.
//      Extents: the centroid rectangle reaches R rows (lx / 2 = R) and R / 2 columns (cpp:1616-1617); a spiral
//      candidate of an unfiltered ring lies within nRings - 2 cells, one of the two filtered rings has its cell
//      centre within R of the continuous centre (SpiralIterator::isInside), i.e. at most floor(R / res + 1/2) cells
//      from the centre cell, and its foot disc footReach further; a member of the centre's own disc at most
//      floor(rf / res + 1/2) cells;
//  (3) getIndex(cell centre of the centroid submap) is the top-left index plus (row, col): the submap's cell centres
//      differ from the map's by a few ulps of the coordinate magnitude, which must be far below half a cell;
//  (4) index predictions (base - limit) / res fit an int with room to spare.
// Search centres outside the map need no extra room: their clamped boxes hold no disc member and getSubmap fails.
int bits_window_halfwidth(const PlanConsts& c, const MapGeom& g) {
    if (!c.footRobust || c.nFoot < 1 || c.nFoot > 64) return 0;
    const double res = g.res;
    const double R = static_cast<double>(c.maxSearchRadius), rf = c.rf;
    const double mag = std::fabs(g.posX) + g.lenX + std::fabs(g.posY) + g.lenY + R + rf + 1.0;
    const double slack = 64.0 * DBL_EPSILON * mag / res + 1e-12;  // rounding error of an index quotient, in cells
    if (!(slack < 1e-3)) return 0;                                 // (3): also keeps cell centres far from cell edges
    if (!((mag + 2.0e6) / res < 5.0e8)) return 0;                  // (4)
    if (!(R / res < 1.0e4) || !(rf / res < 1.0e3)) return 0;
    auto reach = [&](double d) {  // rows between getIndex(centre) and getIndex(centre +- d): ceil(d / res), one more on a tie
        const double q = d / res;
        int n = static_cast<int>(std::ceil(q));
        if (std::fabs(q - std::round(q)) <= slack) n = static_cast<int>(std::round(q)) + 1;
        return n;
    };
    auto within = [&](double d) {  // largest index offset of a cell CENTRE within d of a point of the centre cell
        return static_cast<int>(std::floor((d + 0.5 * res) / res + slack));
    };
    const int nRings = spiral_rings(c.maxSearchRadius, res);
    const int rectReach = reach(R);
    // a candidate of the two filtered rings has its cell centre within R of the continuous centre, which lies within
    // half a cell of the centre cell's centre; unfiltered rings end at nRings - 2
    const int candReach = std::min(nRings, std::max(nRings - 2, within(R))) + c.footReach;
    const int discReach = within(rf);  // members of the continuous-centre disc
    return std::max(std::max(rectReach, candReach), std::max(discReach, 1));
}

int validate_params(const fpe_params& p) {
    const float fs[] = {p.footRadius, p.defaultFootholdThreshold, p.candidateFootholdThreshold, p.searchRadius,
                        p.stepLength, p.length, p.width, p.l1, p.skew};
    for (float f : fs)
        if (!std::isfinite(f)) return FPE_E_INVALID_ARG;
    if (!(p.footRadius >= 0.0f) || !(p.searchRadius >= 0.0f)) return FPE_E_INVALID_ARG;
    if (!std::isfinite(p.h) || !std::isfinite(p.lateralDrift)) return FPE_E_INVALID_ARG;
    return FPE_OK;
}

// initialize(), cpp:340-421.  lengthBase / widthBase / skew / stepLength_ are float members
// (hpp:657-668, 683, 613); widthBase is computed in f32 (cpp:341); each use below promotes the
// float operand to double exactly where the reference's expression does.
// A cell-centred CircleIterator visits cell q = candidate + (a, b) iff the f64 expression
// dx*dx + dy*dy <= rf*rf holds, with dx = cell_pos(i+a) - cell_pos(i).  In exact
// arithmetic that is (a^2 + b^2) res^2 <= rf^2, independent of the candidate.  The f64 evaluation
// differs from the exact value by at most `tol` (two rounded cell positions of magnitude <= |P|+L
// per axis, then squares and a sum), so when every lattice offset near the circle clears the radius
// by more than tol the visited set is the same offset list for every candidate, and the
// iterator's bounding box (>= half a cell of slack) always contains it.  Otherwise footRobust = 0
// and the kernels walk the literal per-candidate bounding box.
void derive_foot_offsets(float footRadius, const MapGeom& g, PlanConsts& c) {
    const double rf = static_cast<double>(footRadius);
    const double res = g.res;
    const double rf2 = rf * rf;
    const double maxAbs = std::max(std::fabs(g.posX) + g.lenX, std::fabs(g.posY) + g.lenY) + rf + res;
    const double e1 = 4.0 * maxAbs * DBL_EPSILON;                    // |error| of one position difference
    const double tol = 4.0 * (rf + 2.0 * res) * e1 + 1e-9 * rf2 + 8.0 * DBL_EPSILON * rf2;
    const int reach = static_cast<int>(std::ceil(rf / res)) + 1;
    c.nFoot = 0;
    c.footRobust = 1;
    c.footReach = reach;  // bound of the literal bounding-box walk; tightened below when the table is valid
    if (reach > 100) {
        c.footRobust = 0;
        return;
    }
    for (int a = -reach; a <= reach; ++a)
        for (int b = -reach; b <= reach; ++b) {
            const double d2 = (static_cast<double>(a) * a + static_cast<double>(b) * b) * (res * res);
            if (std::fabs(d2 - rf2) <= tol) {
                c.footRobust = 0;  // a lattice point sits on the circle within rounding: stay literal
                c.nFoot = 0;
                return;
            }
            if (d2 < rf2) {
                if (c.nFoot >= kMaxFootOffsets) {
                    c.footRobust = 0;
                    c.nFoot = 0;
                    return;
                }
                c.footDa[c.nFoot] = static_cast<int8_t>(a);
                c.footDb[c.nFoot] = static_cast<int8_t>(b);
                ++c.nFoot;
            }
        }
    int mx = 0;
    for (int k = 0; k < c.nFoot; ++k) mx = std::max(mx, std::max(std::abs(static_cast<int>(c.footDa[k])), std::abs(static_cast<int>(c.footDb[k]))));
    c.footReach = mx;
    // Row-interval form of the table: a disc's row +-a holds the columns [-w(a), w(a)].  Verified entry by entry, so
    // the kernels may erode with one interval per row instead of one shift per offset.
    c.nHW = 0;
    if (mx <= 15) {
        int w[16];
        for (int a = 0; a < 16; ++a) w[a] = -1;
        for (int k = 0; k < c.nFoot; ++k) w[std::abs(static_cast<int>(c.footDa[k]))] = std::max(w[std::abs(static_cast<int>(c.footDa[k]))], std::abs(static_cast<int>(c.footDb[k])));
        int expect = 0;
        bool ok = true;
        for (int a = 0; a <= mx; ++a) {
            if (w[a] < 0) ok = false;
            else expect += (a == 0 ? 1 : 2) * (2 * w[a] + 1);
            if (a > 0 && w[a] > w[a - 1]) ok = false;  // (the kernels' nested erosion takes the widths in non-increasing order)
        }
        if (ok && expect == c.nFoot) {  // |set| matches and every entry lies inside its row interval: the forms are equal
            int n = 0;
            for (int a = 0; a <= mx && ok; ++a) {
                int idx = -1;
                for (int q = 0; q < n; ++q)
                    if (c.hwList[q] == w[a]) idx = q;
                if (idx < 0) {
                    if (n >= kMaxHW) {
                        ok = false;
                        break;
                    }
                    c.hwList[n] = static_cast<int8_t>(w[a]);
                    idx = n++;
                }
                c.hwIdx[a] = static_cast<int8_t>(idx);
            }
            if (ok) c.nHW = n;
        }
    }
}

void derive_constants(const fpe_params& p, const MapGeom& geom, float maxSearchRadius, const Tuning& tuning, PlanConsts& c) {
    const double resolution = geom.res;
    std::memset(&c, 0, sizeof(c));
    c.footRadius = p.footRadius;
    c.thrDefault = p.defaultFootholdThreshold;
    c.thrCandidate = p.candidateFootholdThreshold;
    c.searchRadius = p.searchRadius;
    c.rf = static_cast<double>(p.footRadius);
    c.rf2 = c.rf * c.rf;  // CircleIterator: radiusSquare_ = pow(radius_, 2)
    const float lengthBase = p.length;           // cpp:340
    const float widthBase = p.width + p.l1 * 2;  // cpp:341
    c.LbHalf = lengthBase * 0.5;                 // cpp:350
    c.WbHalfNeg = -widthBase * 0.5;              // cpp:351
    c.WbHalfPos = widthBase * 0.5;               // cpp:359
    const double k = static_cast<double>(p.skew);
    // cpp:403-421: RF_FIRST flips the sign of skew
    c.biasX[0] = p.RF_FIRST ? 0.5 * lengthBase + k : 0.5 * lengthBase - k;     // RF
    c.biasX[1] = p.RF_FIRST ? -0.5 * lengthBase - k : -0.5 * lengthBase + k;   // RH
    c.biasX[2] = p.RF_FIRST ? -0.5 * lengthBase + k : -0.5 * lengthBase - k;   // LH
    c.biasX[3] = p.RF_FIRST ? 0.5 * lengthBase - k : 0.5 * lengthBase + k;     // LF
    c.biasY[0] = -0.5 * widthBase;
    c.biasY[1] = -0.5 * widthBase;
    c.biasY[2] = 0.5 * widthBase;
    c.biasY[3] = 0.5 * widthBase;
    c.stepHalf = p.stepLength / 2;     // cpp:2693 (float / int)
    c.step = p.stepLength;             // cpp:2199
    c.stepQuarter = p.stepLength / 4;  // build-defined walk gait
    c.h = p.h;
    c.drift = p.lateralDrift;
    c.RF_FIRST = p.RF_FIRST ? 1 : 0;
    c.maxSearchRadius = maxSearchRadius;
    c.tileH = tile_halfwidth(maxSearchRadius, p.footRadius, resolution);
    {
        // the per-leg LDS tile (tileW^2 flag bytes + 16 tileW of column crossings, fpe_kernels.hip::tile_total_bytes)
        // doubles as float scratch of the ordered height sums, one float per visited cell of a foot-disc bounding
        // box of up to (2 ceil(rf/res) + 2)^2 cells: a foot radius much larger than the search radius must not
        // overrun it, so the tile grows until the scratch fits
        const double boxSide = 2.0 * std::ceil(c.rf / resolution) + 2.0;
        const double scratch = 4.0 * boxSide * boxSide;
        while (static_cast<double>(2 * c.tileH + 1) * (2 * c.tileH + 1) + 16.0 * (2 * c.tileH + 1) < scratch && c.tileH < 4096) ++c.tileH;
    }
    c.tileW = 2 * c.tileH + 1;
    c.tileWMagic = fastdiv_magic(static_cast<uint32_t>(c.tileW));
    c.groupOverride = tuning.planGroup;
    c.noMidVariant = tuning.noMidVariant;
    c.noBits = tuning.noBits;
    c.trace = nullptr;
#ifdef FPE_TRACE  // profiling builds only (scratch/trace.py); the shipped library never reads the environment per call
    if (const char* tr = std::getenv("FPE_TRACE_PTR")) c.trace = reinterpret_cast<unsigned long long*>(std::strtoull(tr, nullptr, 0));
#endif
    // isos_ (cpp:384-394): longEdge = lengthBase + skew*2 in f32 (hpp:666, 683), promoted on assignment to the
    // double member (hpp:679); footSearchRect_.length = searchRadius_*2 and .width = searchRadius_ are f32 values
    // stored in doubles (hpp:700-701); the sums are f64
    {
        const float longEdgeF = lengthBase + p.skew * 2;
        const double longEdge = longEdgeF, shortEdge = widthBase;
        const double rectLen = p.searchRadius * 2, rectWid = p.searchRadius;
        c.isosLen = longEdge + rectLen;
        c.isosWid = shortEdge + rectWid;
    }
    derive_foot_offsets(p.footRadius, geom, c);
    // Middle cell of an unclamped 3x3 CircleIterator box.  The box rows are i0 = index(cx + r) .. i0 + 2 =
    // index(cx - r) (findSubmapParameters); cell i covers (x_i - res/2, x_i + res/2], so cx + r <= x_m + 1.5 res and
    // cx - r >= x_m - 1.5 res for the middle row's centre x_m, i.e. |cx - x_m| <= 1.5 res - r, and likewise in y.
    // With r >= 0.9 res the middle cell's squared distance is <= 2 (0.6 res)^2 = 0.72 res^2 < 0.81 res^2 <= r^2: an 11 %
    // margin against rounding errors of order 1e-9 m (|coordinates| <= 1e6), so the reference's f64 test
    // dx*dx + dy*dy <= r^2 is true for that cell without evaluating it.
    c.midCellInside = (c.rf >= 0.9 * resolution) ? 1 : 0;
    if (tuning.literalDiscs) {
        c.midCellInside = 0;
        c.footRobust = 0;
        c.footReach = static_cast<int>(std::ceil(c.rf / resolution)) + 1;
    }  // test knob: force the literal bounding-box walk
    c.winH = bits_window_halfwidth(c, geom);
    {
        // bound of the rounding error of an index quotient, in cells (the same bound as `slack` above, doubled)
        const double mag = std::fabs(geom.posX) + geom.lenX + std::fabs(geom.posY) + geom.lenY +
                           static_cast<double>(maxSearchRadius) + c.rf + 1.0;
        c.cornerEps = 128.0 * DBL_EPSILON * mag / resolution + 1e-12;
    }
}

// globalFootholdPlan message bookkeeping: cpp:681-699 (stance entries), cpp:1378-1396 (valid
// cycle), cpp:1574 (invalid cycle: success=false, nothing appended).
void assemble_global_footholds(const fpe_foothold* nominal, const uint8_t* cycleOk, const double* stance,
                               int nCycles, fpe_global_footholds* msg) {
    std::memset(msg, 0, sizeof(*msg));
    msg->gait_cycles = static_cast<uint8_t>(nCycles);
    msg->gait_cycles_succeed = 0;
    msg->success = 0;
    int n = 0;
    for (int l = 0; l < 4; ++l) {
        fpe_msg_foothold& f = msg->footholds[n++];
        f.x = stance[l * 3 + 0];
        f.y = stance[l * 3 + 1];
        f.z = stance[l * 3 + 2];
        f.foot_id = static_cast<uint8_t>(l);
        f.gait_cycle_id = 0;
    }
    for (int g = 0; g < nCycles; ++g) {
        if (cycleOk[g]) {
            msg->gait_cycles_succeed = static_cast<uint8_t>(g + 1);
            msg->success = 1;
            for (int l = 0; l < 4; ++l) {
                const fpe_foothold& s = nominal[g * 4 + l];
                fpe_msg_foothold& f = msg->footholds[n++];
                f.x = s.x;
                f.y = s.y;
                f.z = static_cast<double>(s.z);
                f.foot_id = static_cast<uint8_t>(l);
                f.gait_cycle_id = static_cast<uint8_t>(g);
            }
        } else {
            msg->success = 0;
        }
    }
    msg->n_footholds = n;
}

// centroidGlobalFootholdsMsg_ (cpp:709-727, 1444-1462).  Unlike the nominal message its `success` is set false
// once per call (cpp:711) and true on every committed cycle (cpp:1446) — a failed cycle never clears it again
// (cpp:1574 touches nominalGlobalFootholdsMsg_ only) — and its `gait_cycles` field is never written (stays 0).
void assemble_centroid_footholds(const fpe_centroid_foothold* cen, const uint8_t* cycleOk, const double* stance,
                                 int nCycles, fpe_global_footholds* msg) {
    std::memset(msg, 0, sizeof(*msg));
    int n = 0;
    for (int l = 0; l < 4; ++l) {
        fpe_msg_foothold& f = msg->footholds[n++];
        f.x = stance[l * 3 + 0];
        f.y = stance[l * 3 + 1];
        f.z = stance[l * 3 + 2];
        f.foot_id = static_cast<uint8_t>(l);
        f.gait_cycle_id = 0;
    }
    for (int g = 0; g < nCycles; ++g) {
        if (!cycleOk[g]) continue;
        msg->gait_cycles_succeed = static_cast<uint8_t>(g + 1);  // cpp:1445
        msg->success = 1;                                        // cpp:1446
        for (int l = 0; l < 4; ++l) {
            const fpe_centroid_foothold& s = cen[g * 4 + l];
            fpe_msg_foothold& f = msg->footholds[n++];
            f.x = s.x;
            f.y = s.y;
            f.z = static_cast<double>(s.z);
            f.foot_id = static_cast<uint8_t>(l);
            f.gait_cycle_id = static_cast<uint8_t>(g);
        }
    }
    msg->n_footholds = n;
}

// getPolygonCenter (cpp:2421-2463) on the host, with z (= mean of the four z, cpp:2459-2461).
static void polygon_center_host(const double feet[4][3], double out[3]) {
    const double x1 = feet[0][0], y1 = feet[0][1];
    double x2 = feet[1][0], y2 = feet[1][1];
    double sum_x = 0, sum_y = 0, sum_s = 0;
    for (int i = 1; i <= 2; i++) {
        const double x3 = feet[i + 1][0], y3 = feet[i + 1][1];
        const double s = ((x2 - x1) * (y3 - y1) - (x3 - x1) * (y2 - y1)) / 2.0;
        sum_x += (x1 + x2 + x3) * s;
        sum_y += (y1 + y2 + y3) * s;
        sum_s += s;
        x2 = x3;
        y2 = y3;
    }
    out[0] = sum_x / sum_s / 3.0;
    out[1] = sum_y / sum_s / 3.0;
    out[2] = (feet[0][2] + feet[1][2] + feet[2][2] + feet[3][2]) / 4.0;
}

// Feet-centre path and KPIs of one track, rebuilt from the kernel's per-cycle results: the track's
// current feet start at the stance shifted by -stepLength_/2 (setFirstGait, cpp:2679-2699) and take the
// results of every committed cycle (cpp:1413-1416 / 1480-1483).
void assemble_track_report(const double* resultXYZ /*[nCycles][4][3]*/, const uint8_t* cycleOk, const double* stance,
                           int nCycles, const fpe_params& params, fpe_track_report* rep) {
    std::memset(rep, 0, sizeof(*rep));
    const double stepHalf = static_cast<double>(params.stepLength / 2);
    double cur[4][3];
    for (int l = 0; l < 4; ++l) {
        cur[l][0] = stance[l * 3 + 0] - stepHalf;
        cur[l][1] = stance[l * 3 + 1];
        cur[l][2] = stance[l * 3 + 2];
    }
    enum { RF = 0, RH = 1, LH = 2, LF = 3 };
    const double gaitCycle = 1.0;  // cpp:332
    for (int g = 0; g < nCycles; ++g) {
        polygon_center_host(cur, rep->feet_center_path[rep->n_path++]);
        if (!cycleOk[g]) continue;
        const double(*r)[3] = reinterpret_cast<const double(*)[3]>(resultXYZ + static_cast<size_t>(g) * 12);
        rep->feet_distance[rep->n_kpi] = r[RF][0] - r[LH][0];
        rep->feet_distance[rep->n_kpi + 1] = r[LF][0] - r[RH][0];
        double c1, c2, c3;
        if (params.RF_FIRST) {
            c1 = (cur[RF][0] + cur[LH][0]) / 2;
            c2 = (r[LF][0] + r[RH][0]) / 2;
            c3 = (r[RF][0] + r[LH][0]) / 2;
        } else {
            c1 = (cur[LF][0] + cur[RH][0]) / 2;
            c2 = (r[RF][0] + r[LH][0]) / 2;
            c3 = (r[LF][0] + r[RH][0]) / 2;
        }
        const double d1 = c2 - c1, d2 = c3 - c2;
        rep->cog_speed[rep->n_kpi] = d1 / (0.5 * gaitCycle);
        rep->cog_speed[rep->n_kpi + 1] = d2 / (0.5 * gaitCycle);
        rep->n_kpi += 2;
        for (int l = 0; l < 4; ++l)
            for (int k = 0; k < 3; ++k) cur[l][k] = r[l][k];
    }
}

int derive_opt_constants(const fpe_params& p, const fpe_opt_params& op, const MapGeom& g, const PlanConsts& pc, OptConsts& oc) {
    const double vals[] = {op.w1, op.w2, op.w3, op.w4, op.wr, op.wc, op.ctol, op.hip_lower_scale, op.hip_upper_scale,
                           op.skew_lower_scale, op.skew_upper_scale, op.lf_current_row0, op.rh_current_row0};
    for (double v : vals)
        if (!std::isfinite(v)) return FPE_E_INVALID_ARG;
    oc.w1 = op.w1; oc.w2 = op.w2; oc.w3 = op.w3; oc.w4 = op.w4; oc.wr = op.wr; oc.wc = op.wc;
    oc.ctol = op.ctol;
    const double lengthBase = static_cast<double>(p.length);  // cpp:497: lengthBase = laikagoKinematics_.lengthBase (float)
    const double skew = static_cast<double>(p.skew);          // cpp:498
    const double mapResolution = g.res;                       // cpp:514
    oc.lengthBase = lengthBase;
    oc.skew = skew;
    oc.mapResolution = mapResolution;
    const double hip_lower_scale = op.hip_lower_scale, hip_upper_scale = op.hip_upper_scale;
    const double skew_lower_scale = op.skew_lower_scale, skew_upper_scale = op.skew_upper_scale;
    oc.t1 = lengthBase * hip_lower_scale/mapResolution;   // cpp:1156
    oc.t2 = lengthBase * hip_upper_scale/mapResolution;   // cpp:1157
    oc.t3 = 2* skew * skew_lower_scale/mapResolution;     // cpp:1158
    oc.t4 = 2* skew * skew_upper_scale/mapResolution;     // cpp:1159
    oc.lbOverRes = lengthBase/mapResolution;              // cpp:69-70
    oc.skew2OverRes = 2*skew/mapResolution;               // cpp:71-72
    oc.lfRow0 = op.lf_current_row0;
    oc.rhRow0 = op.rh_current_row0;
    oc.useConstraints = op.use_inequality_constraints ? 1 : 0;
    // footSearchRect_.width = searchRadius_ (an f32 value in a double, cpp:385); .col = width / mapResolution, a double
    // (cpp:529, hpp:704-705); Eigen::MatrixXi assignments truncate toward zero (cpp:1063-1066)
    const double footSearchRectWidth = p.searchRadius;
    const double footSearchRectCol = footSearchRectWidth/mapResolution;
    const double a = footSearchRectCol, b = pc.isosWid/mapResolution - footSearchRectCol, c = pc.isosWid/mapResolution;
    if (!(std::fabs(a) < 2.0e9) || !(std::fabs(b) < 2.0e9) || !(std::fabs(c) < 2.0e9)) return FPE_E_INVALID_ARG;
    oc.colLoA = 0;
    oc.colUpA = static_cast<int>(a);
    oc.colLoB = static_cast<int>(b);
    oc.colUpB = static_cast<int>(c);
    oc.pad = 0;
    return FPE_OK;
}

void assemble_opt_footholds(const fpe_opt_foothold* opt, const uint8_t* cycleOk, const double* stance, int nCycles,
                            fpe_global_footholds* msg) {
    std::memset(msg, 0, sizeof(*msg));
    int n = 0;
    for (int l = 0; l < 4; ++l) {  // cpp:737-755: the initial stance, gait_cycle_id 0
        fpe_msg_foothold& f = msg->footholds[n++];
        f.x = stance[l * 3 + 0];
        f.y = stance[l * 3 + 1];
        f.z = stance[l * 3 + 2];
        f.foot_id = static_cast<uint8_t>(l);
        f.gait_cycle_id = 0;
    }
    for (int g = 0; g < nCycles; ++g) {
        if (!cycleOk[g]) continue;
        msg->gait_cycles_succeed = static_cast<uint8_t>(g + 1);  // cpp:1511
        msg->success = 1;                                        // cpp:1512
        for (int l = 0; l < 4; ++l) {
            const fpe_opt_foothold& s = opt[g * 4 + l];
            fpe_msg_foothold& f = msg->footholds[n++];
            f.x = s.x;
            f.y = s.y;
            f.z = static_cast<double>(s.z);
            f.foot_id = static_cast<uint8_t>(l);
            f.gait_cycle_id = static_cast<uint8_t>(g);
        }
    }
    msg->n_footholds = n;
}

void interleave_centroid_path(fpe_track_report* cen, const fpe_track_report& opt) {
    const int n = cen->n_path < opt.n_path ? cen->n_path : opt.n_path;
    for (int g = n - 1; g >= 0; --g) {
        for (int k = 0; k < 3; ++k) {
            const double c = cen->feet_center_path[g][k];
            cen->feet_center_path[2 * g][k] = c;
            cen->feet_center_path[2 * g + 1][k] = opt.feet_center_path[g][k];
        }
    }
    cen->n_path = 2 * n;
}

}  // namespace fpe

// ---- C ABI: host-only entry points -----------------------------------------------------------------
extern "C" {

int fpe_params_yaml(fpe_params* p) {  // foothold_planner/config/foothold_planner.yaml:10-64
    if (!p) return FPE_E_INVALID_ARG;
    p->footRadius = float(0.02);
    p->defaultFootholdThreshold = float(0.9);
    p->candidateFootholdThreshold = float(0.7);
    p->searchRadius = float(0.1);
    p->stepLength = float(0.18);
    p->length = float(0.4387);
    p->width = float(0.175);
    p->l1 = float(0.037);
    p->skew = float(0.04);
    p->RF_FIRST = 0;
    p->h = 0.01;
    p->lateralDrift = -0.007;
    return FPE_OK;
}

int fpe_params_code_defaults(fpe_params* p) {  // readParameters, cpp:255-290
    if (!p) return FPE_E_INVALID_ARG;
    p->footRadius = float(0.03);
    p->defaultFootholdThreshold = float(0.7);
    p->candidateFootholdThreshold = float(0.7);
    p->searchRadius = float(0.1);
    p->stepLength = float(0.2);
    p->length = float(0.4387);
    p->width = float(0.175);
    p->l1 = float(0.037);
    p->skew = float(0.1);
    p->RF_FIRST = 0;
    p->h = 0.01;
    p->lateralDrift = -0.007;
    return FPE_OK;
}

int fpe_opt_params_yaml(fpe_opt_params* p) {  // foothold_planner.yaml:53-63, FootholdPlanner.cpp:34, 48-49
    if (!p) return FPE_E_INVALID_ARG;
    std::memset(p, 0, sizeof(*p));
    p->w1 = p->w2 = p->w3 = p->w4 = 1.0;
    p->wr = p->wc = 1.0;
    p->use_inequality_constraints = 1;
    p->ctol = 1e-2;
    p->hip_lower_scale = 0.9;
    p->hip_upper_scale = 1.1;
    p->skew_lower_scale = 0.8;
    p->skew_upper_scale = 1.2;
    return FPE_OK;
}

int fpe_opt_params_code_defaults(fpe_opt_params* p) {  // readParameters, cpp:297-307
    const int rc = fpe_opt_params_yaml(p);
    if (rc == FPE_OK) p->use_inequality_constraints = 0;
    return rc;
}

int fpe_spiral_offsets(int32_t n_rings, int32_t* out, int32_t max_cells) {
    if (n_rings < 0 || n_rings > fpe::kMaxRings) return FPE_E_INVALID_ARG;
    fpe::SpiralTable t;
    fpe::build_spiral_table(n_rings, t);
    const int n = static_cast<int>(t.di.size());
    if (out)
        for (int k = 0; k < n && k < max_cells; ++k) {
            out[3 * k + 0] = t.di[k];
            out[3 * k + 1] = t.dj[k];
            out[3 * k + 2] = t.ring[k];
        }
    return n;
}

int fpe_tile_halfwidth(float search_radius, float foot_radius, double resolution) {
    if (!(resolution > 0.0)) return FPE_E_INVALID_ARG;
    return fpe::tile_halfwidth(search_radius, foot_radius, resolution);
}

double fpe_algorithmic_bytes_per_foothold(float search_radius, float foot_radius, double resolution) {
    const double R = static_cast<double>(search_radius), rf = static_cast<double>(foot_radius);
    const int nS = static_cast<int>(std::floor(R / resolution + 0.5));
    const int nF = static_cast<int>(std::floor(rf / resolution));
    const int W = 2 * (nS + nF) + 1;
    int nFoot = 0;  // cells of a cell-centred foot disc
    const int reach = nF + 1;
    for (int a = -reach; a <= reach; ++a)
        for (int b = -reach; b <= reach; ++b)
            if ((static_cast<double>(a * a + b * b)) * resolution * resolution <= rf * rf) ++nFoot;
    return 4.0 * W * W + 8.0 * nFoot + 16.0;
}

}  // extern "C"
