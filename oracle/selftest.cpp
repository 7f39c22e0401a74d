// ORACLE — TEST INFRASTRUCTURE ONLY.  Stand-alone driver for sanitizer runs (tests/test_oracle_sanitize.py):
// builds a hostile random map, runs chained plans (trot, walk, hexagons, poses inside / at the border /
// outside the map) and open-loop searches under -fsanitize=address,undefined.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <random>

#include "fpo_planner.hpp"

using namespace fpo;

int main(int argc, char** argv) {
    const unsigned seed = argc > 1 ? static_cast<unsigned>(std::atoi(argv[1])) : 1u;
    std::mt19937 rng(seed);
    std::uniform_real_distribution<double> u(0.0, 1.0);
    const int rows = 140, cols = 120;
    const double res = 0.02;
    GridMap m;
    m.size = {rows, cols};
    m.res = res;
    m.length = {rows * res, cols * res};
    m.position = {0.3, -0.2};
    m.trav.resize((size_t)rows * cols);
    m.elev.resize((size_t)rows * cols);
    for (size_t k = 0; k < m.trav.size(); ++k) {
        const double r = u(rng);
        m.trav[k] = r < 0.05 ? std::numeric_limits<float>::quiet_NaN()
                  : r < 0.06 ? -std::numeric_limits<float>::infinity()
                  : r < 0.07 ? std::numeric_limits<float>::infinity()
                             : static_cast<float>(0.4 + 0.6 * u(rng));
        m.elev[k] = u(rng) < 0.05 ? std::numeric_limits<float>::quiet_NaN() : (u(rng) < 0.03 ? 12.0f : static_cast<float>(0.2 * u(rng)));
    }
    Params p;
    p.footRadius = 0.02f; p.defaultFootholdThreshold = 0.9f; p.candidateFootholdThreshold = 0.7f; p.searchRadius = 0.1f;
    p.stepLength = 0.18f; p.length = 0.4387f; p.width = 0.175f; p.l1 = 0.037f; p.skew = 0.04f; p.RF_FIRST = 0;
    p.h = 0.01; p.lateralDrift = -0.007;
    long valid = 0, legs = 0;
    PlanOutput out;
    for (int b = 0; b < 60; ++b) {
        PoseSpec ps;
        ps.pose[0] = 0.3 + (u(rng) - 0.5) * 4.0;  // inside, at the border and outside the 2.8 x 2.4 m map
        ps.pose[1] = -0.2 + (u(rng) - 0.5) * 3.6;
        ps.pose[2] = 0.1 * u(rng);
        ps.gait = b % 2;
        for (int l = 0; l < 4; ++l) {
            ps.legRadius[l] = (b % 3 == 0) ? static_cast<float>(0.05 + 0.1 * u(rng)) : 0.0f;
            ps.legPoly[l] = (b % 5 == 0) ? 1 : 0;
        }
        p.RF_FIRST = b % 4 == 3;
        planGlobalFootholds(m, p, ps, 6, out);
        for (const auto& r : out.nominal) { valid += r.valid; ++legs; }
    }
    // open-loop searches with degenerate polygons (0..8 vertices, repeated points)
    for (int q = 0; q < 200; ++q) {
        Polygon poly;
        const int nv = q % 9;
        const double cx = 0.3 + (u(rng) - 0.5) * 3.0, cy = -0.2 + (u(rng) - 0.5) * 2.6;
        for (int v = 0; v < nv; ++v) poly.addVertex({cx + 0.2 * (u(rng) - 0.5), cy + 0.2 * (u(rng) - 0.5)});
        LegResult r;
        checkFoothold(m, {cx, cy}, p.footRadius, 0.1f, poly, p, r);
        valid += r.valid;
        ++legs;
    }
    // non-finite / absurd centres are defined as "no cell visited"
    LegResult r;
    checkFoothold(m, {std::numeric_limits<double>::quiet_NaN(), 0.0}, p.footRadius, 0.1f, Polygon(), p, r);
    if (r.valid) return 2;
    CentroidResult c;
    checkFootholdUseCentroidMethod(m, {1e300, 0.0}, 0.1f, p, c);
    if (c.code != 6) return 3;
    // other geometries: resolutions that are not dyadic fractions, off-origin maps, poses along every edge — a corner
    // bounded onto the far edge can round to the index `size` there (getSubmap must fail, not copy past the layer;
    // the 280-column 4 cm map centred at y = 4.8865 is random-campaign case 502981)
    const double resList[5] = {0.04, 0.03, 0.025, 0.0125, 0.02};
    for (int gcase = 0; gcase < 10; ++gcase) {
        GridMap g;
        const double r2 = resList[gcase % 5];
        const int rw = gcase == 0 ? 213 : 60 + static_cast<int>(u(rng) * 120), cl = gcase == 0 ? 280 : 60 + static_cast<int>(u(rng) * 120);
        g.size = {rw, cl};
        g.res = r2;
        g.length = {rw * r2, cl * r2};
        g.position = gcase == 0 ? Vec2{-4.887825252429465, 4.886504903968108} : Vec2{(u(rng) - 0.5) * 12.0, (u(rng) - 0.5) * 12.0};
        g.trav.resize((size_t)rw * cl);
        g.elev.resize((size_t)rw * cl);
        for (size_t k = 0; k < g.trav.size(); ++k) {
            g.trav[k] = u(rng) < 0.03 ? std::numeric_limits<float>::quiet_NaN() : static_cast<float>(0.3 + 0.7 * u(rng));
            g.elev[k] = static_cast<float>(0.2 * u(rng));
        }
        Params q = p;
        q.searchRadius = static_cast<float>(5.4 * r2);
        q.footRadius = static_cast<float>(r2);
        for (int b = 0; b < 80; ++b) {
            PoseSpec ps;
            const int edge = b % 4;  // walk the poses along the four edges, inside and outside
            const double tpos = u(rng) - 0.5, off = (u(rng) - 0.5) * 0.6;
            ps.pose[0] = g.position.x + (edge < 2 ? (edge == 0 ? 0.5 : -0.5) * g.length.x + off : tpos * g.length.x);
            ps.pose[1] = g.position.y + (edge >= 2 ? (edge == 2 ? 0.5 : -0.5) * g.length.y + off : tpos * g.length.y);
            ps.pose[2] = 0.0;
            ps.gait = b % 2;
            for (int l = 0; l < 4; ++l) {
                ps.legRadius[l] = 0.0f;
                ps.legPoly[l] = (b % 7 == 0) ? 1 : 0;
            }
            planGlobalFootholds(g, q, ps, 4, out);
            for (const auto& rr : out.nominal) { valid += rr.valid; ++legs; }
        }
    }
    std::printf("selftest ok: %ld valid of %ld legs\n", valid, legs);
    return 0;
}
