// upstream_check.cpp — ONE command that turns "parity unpinned" into "pinned on this machine", for whoever has grid_map.
//
// The engine's parity rests on oracle/fpo_gridmap.hpp, a literal restatement of the few grid_map_core routines the
// reference's hot path calls (GridMapMath's getIndexFromPosition / getPositionFromIndex /
// boundPositionToRange / getSubmapInformation, GridMap::setGeometry / getSubmap, Polygon::isInside, CircleIterator,
// SpiralIterator).  grid_map_core is not in the build image and not vendored by the reference, so that restatement is
// argued by citation only.  This program runs the SAME seeded inputs through the REAL library and through the restatement
// and stops at the first difference:
//
//   g++ -std=c++17 -O1 -ffp-contract=off -DFPE_WITH_GRID_MAP -I<repo>/oracle $(pkg-config --cflags eigen3)
//       -I/opt/ros/$ROS_DISTRO/include tests/probe/upstream_check.cpp -L/opt/ros/$ROS_DISTRO/lib -lgrid_map_core -o upstream_check
//   ./upstream_check [cases per section, default 200000]        # exit 0: every case identical; 1: first difference printed
//
//   Without FPE_WITH_GRID_MAP it compares the restatement with itself — a self-test of the harness that
// runs in the build image; against tests/probe/ros_mock (declarations only) the upstream half is PARSED there
// (tests/test_cpu_abi_and_host.py), never linked.  What it covers, with inputs drawn to sit ON the decisive roundings
// (positions on cell borders +- a few ulps, radii that are whole numbers of cells, maps far from the origin, windows hanging
// over the map edge, resolutions that are not dyadic):
//   1. GridMap::getIndex / getPosition / isInside                (cpp:1703.., 2098, 2105, 2136)
//   2. CircleIterator: the visited indices, in order             (cpp:2048, 2126, 2529)
//   3. SpiralIterator: the visited indices, in order             (cpp:2095)
//   4. Polygon::isInside on rectangles and hexagons              (cpp:2138, 2496-2517)
//   5. GridMap::getSubmap: success, size, position, cell values  (cpp:1627, 2345)
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "fpo_gridmap.hpp"

#ifdef FPE_WITH_GRID_MAP
#include <grid_map_core/grid_map_core.hpp>
#endif

namespace {

struct Rng {  // splitmix64: the same stream on every machine
    uint64_t s;
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    double uni() { return static_cast<double>(next() >> 11) * (1.0 / 9007199254740992.0); }
    double uni(double a, double b) { return a + (b - a) * uni(); }
    int below(int n) { return static_cast<int>(next() % static_cast<uint64_t>(n)); }
};

// a double nudged by k ulps
double nudge(double v, int k) {
    for (; k > 0; --k) v = std::nextafter(v, 1e300);
    for (; k < 0; ++k) v = std::nextafter(v, -1e300);
    return v;
}

struct MapCase {
    double lx, ly, res, px, py;
};
MapCase draw_map(Rng& r) {
    static const double kRes[] = {0.02, 0.01, 0.005, 0.03, 0.04, 0.025, 0.1};
    MapCase m;
    m.res = kRes[r.below(7)];
    m.lx = m.res * (4 + r.below(60));
    m.ly = m.res * (4 + r.below(60));
    const int where = r.below(4);
    m.px = where == 0 ? 0.0 : (where == 1 ? r.uni(-5, 5) : (where == 2 ? r.uni(-2000, 2000) : m.res * (r.below(2001) - 1000)));
    m.py = where == 0 ? 0.0 : (where == 1 ? r.uni(-5, 5) : (where == 2 ? r.uni(-2000, 2000) : m.res * (r.below(2001) - 1000)));
    return m;
}
// positions that matter: anywhere around the map, on a cell centre, on a cell border, each +- a few ulps
fpo::Vec2 draw_position(Rng& r, const fpo::GridMap& m) {
    fpo::Vec2 p;
    for (int k = 0; k < 2; ++k) {
        const double len = k ? m.length.y : m.length.x, pos = k ? m.position.y : m.position.x;
        const int n = k ? m.size.j : m.size.i;
        double v;
        switch (r.below(4)) {
            case 0: v = pos + r.uni(-0.7, 0.7) * len; break;
            case 1: v = (pos + 0.5 * len - 0.5 * m.res) - m.res * (r.below(n + 4) - 2); break;  // a cell centre (or just outside)
            case 2: v = (pos + 0.5 * len) - m.res * (r.below(n + 3) - 1); break;                 // a cell border
            default: v = pos + (r.below(2) ? 0.5 : -0.5) * len; break;                           // the map's edge
        }
        (k ? p.y : p.x) = nudge(v, r.below(7) - 3);
    }
    return p;
}
double draw_radius(Rng& r, double res) {
    switch (r.below(3)) {
        case 0: return res * (1 + r.below(12));                        // whole cells: offsets ON the circle
        case 1: return nudge(res * (1 + r.below(12)), r.below(5) - 2);
        default: return r.uni(0.3, 9.0) * res;
    }
}

int g_fail = 0;
#define CHECK(cond, ...)                                  \
    do {                                                  \
        if (!(cond)) {                                    \
            std::printf("DIFFERENCE (%s): ", #cond);      \
            std::printf(__VA_ARGS__);                     \
            std::printf("\n");                            \
            g_fail = 1;                                   \
            return;                                       \
        }                                                 \
    } while (0)

#ifdef FPE_WITH_GRID_MAP
// ---- the real library ----
using UpMap = grid_map::GridMap;
UpMap make_up(const MapCase& c, const std::vector<float>& trav) {
    UpMap m({"traversability"});
    m.setGeometry(grid_map::Length(c.lx, c.ly), c.res, grid_map::Position(c.px, c.py));
    grid_map::Matrix& layer = m["traversability"];
    for (int j = 0; j < m.getSize()(1); ++j)
        for (int i = 0; i < m.getSize()(0); ++i) layer(i, j) = trav[static_cast<size_t>(i) + static_cast<size_t>(j) * m.getSize()(0)];
    return m;
}
bool up_get_index(const UpMap& m, const fpo::Vec2& p, fpo::Idx2& idx) {
    grid_map::Index i;
    const bool ok = m.getIndex(grid_map::Position(p.x, p.y), i);
    idx = {i(0), i(1)};
    return ok;
}
bool up_get_position(const UpMap& m, const fpo::Idx2& idx, fpo::Vec2& p) {
    grid_map::Position q;
    const bool ok = m.getPosition(grid_map::Index(idx.i, idx.j), q);
    if (ok) p = {q.x(), q.y()};
    return ok;
}
std::vector<fpo::Idx2> up_circle(const UpMap& m, const fpo::Vec2& c, double r) {
    std::vector<fpo::Idx2> v;
    for (grid_map::CircleIterator it(m, grid_map::Position(c.x, c.y), r); !it.isPastEnd(); ++it) v.push_back({(*it)(0), (*it)(1)});
    return v;
}
std::vector<fpo::Idx2> up_spiral(UpMap& m, const fpo::Vec2& c, double r) {
    std::vector<fpo::Idx2> v;
    for (grid_map::SpiralIterator it(m, grid_map::Position(c.x, c.y), r); !it.isPastEnd(); ++it) v.push_back({(*it)(0), (*it)(1)});
    return v;
}
bool up_polygon(const std::vector<fpo::Vec2>& vs, const fpo::Vec2& p) {
    grid_map::Polygon poly;
    for (const fpo::Vec2& v : vs) poly.addVertex(grid_map::Position(v.x, v.y));
    return poly.isInside(grid_map::Position(p.x, p.y));
}
struct SubOut {
    bool ok;
    fpo::Idx2 size;
    fpo::Vec2 position;
    std::vector<float> trav;  // column-major
};
SubOut up_submap(const UpMap& m, const fpo::Vec2& p, const fpo::Vec2& len) {
    SubOut o{false, {0, 0}, {0, 0}, {}};
    bool ok = false;
    UpMap s = m.getSubmap(grid_map::Position(p.x, p.y), grid_map::Length(len.x, len.y), ok);
    o.ok = ok;
    if (!ok) return o;
    o.size = {s.getSize()(0), s.getSize()(1)};
    o.position = {s.getPosition().x(), s.getPosition().y()};
    // (a submap of a map whose start index is 0 has start index 0 too; read through the accessor to stay layout-free)
    for (int j = 0; j < o.size.j; ++j)
        for (int i = 0; i < o.size.i; ++i) o.trav.push_back(s.at("traversability", grid_map::Index(i, j)));
    // column-major order: (i, j) at i + j * rows — the loop above is j-outer, i-inner
    return o;
}
#else
// ---- self-test of the harness: the restatement against itself ----
struct UpMap {
    fpo::GridMap g;
};
UpMap make_up(const MapCase& c, const std::vector<float>& trav) {
    UpMap m;
    m.g.setGeometry({c.lx, c.ly}, c.res, {c.px, c.py});
    m.g.trav = trav;
    return m;
}
bool up_get_index(const UpMap& m, const fpo::Vec2& p, fpo::Idx2& idx) { return m.g.getIndex(p, idx); }
bool up_get_position(const UpMap& m, const fpo::Idx2& idx, fpo::Vec2& p) { return m.g.getPosition(idx, p); }
std::vector<fpo::Idx2> up_circle(const UpMap& m, const fpo::Vec2& c, double r) {
    std::vector<fpo::Idx2> v;
    for (fpo::CircleIterator it(m.g, c, r); !it.isPastEnd(); ++it) v.push_back(*it);
    return v;
}
std::vector<fpo::Idx2> up_spiral(UpMap& m, const fpo::Vec2& c, double r) {
    std::vector<fpo::Idx2> v;
    for (fpo::SpiralIterator it(m.g, c, r); !it.isPastEnd(); ++it) v.push_back(*it);
    return v;
}
bool up_polygon(const std::vector<fpo::Vec2>& vs, const fpo::Vec2& p) {
    fpo::Polygon poly;
    for (const fpo::Vec2& v : vs) poly.addVertex(v);
    return poly.isInside(p);
}
struct SubOut {
    bool ok;
    fpo::Idx2 size;
    fpo::Vec2 position;
    std::vector<float> trav;
};
SubOut up_submap(const UpMap& m, const fpo::Vec2& p, const fpo::Vec2& len) {
    SubOut o{false, {0, 0}, {0, 0}, {}};
    fpo::GridMap s = m.g.getSubmap(p, len, o.ok);
    if (!o.ok) return o;
    o.size = s.size;
    o.position = s.position;
    o.trav = s.trav;
    return o;
}
#endif

bool same_bits(double a, double b) { return a == b || (a != a && b != b); }

void one_case(Rng& r, int section) {
    const MapCase c = draw_map(r);
    fpo::GridMap om;
    om.setGeometry({c.lx, c.ly}, c.res, {c.px, c.py});
    om.trav.resize(static_cast<size_t>(om.size.i) * om.size.j);
    for (float& v : om.trav) v = static_cast<float>(r.uni());
    UpMap um = make_up(c, om.trav);
    const char* tag = "map %.17g x %.17g @ %.17g at (%.17g, %.17g)";
#define MAPARGS c.lx, c.ly, c.res, c.px, c.py
    if (section == 1) {
        const fpo::Vec2 p = draw_position(r, om);
        fpo::Idx2 a{0, 0}, b{0, 0};
        const bool oa = om.getIndex(p, a), ob = up_get_index(um, p, b);
        CHECK(oa == ob && a.i == b.i && a.j == b.j, "getIndex(%.17g, %.17g): restatement %d (%d, %d), upstream %d (%d, %d); " "map %.17g x %.17g @ %.17g at (%.17g, %.17g)",
              p.x, p.y, oa, a.i, a.j, ob, b.i, b.j, MAPARGS);
        const fpo::Idx2 q{r.below(om.size.i + 2) - 1, r.below(om.size.j + 2) - 1};
        fpo::Vec2 pa{0, 0}, pb{0, 0};
        const bool ga = om.getPosition(q, pa), gb = up_get_position(um, q, pb);
        CHECK(ga == gb && (!ga || (same_bits(pa.x, pb.x) && same_bits(pa.y, pb.y))), "getPosition(%d, %d): restatement %d (%.17g, %.17g), upstream %d (%.17g, %.17g)",
              q.i, q.j, ga, pa.x, pa.y, gb, pb.x, pb.y);
    } else if (section == 2 || section == 3) {
        const fpo::Vec2 ctr = draw_position(r, om);
        const double rad = draw_radius(r, c.res);
        std::vector<fpo::Idx2> a, b;
        if (section == 2) {
            for (fpo::CircleIterator it(om, ctr, rad); !it.isPastEnd(); ++it) a.push_back(*it);
            b = up_circle(um, ctr, rad);
        } else {
            // (SpiralIterator of 1.6.x dereferences an empty ring when the centre cell lies outside the map: undefined upstream,
            // oracle-defined in the restatement — such centres are not part of the contract)
            fpo::Idx2 ci;
            if (!om.getIndex(ctr, ci)) return;
            for (fpo::SpiralIterator it(om, ctr, rad); !it.isPastEnd(); ++it) a.push_back(*it);
            b = up_spiral(um, ctr, rad);
        }
        CHECK(a.size() == b.size(), "%s around (%.17g, %.17g) r %.17g: %zu cells in the restatement, %zu upstream; " "map %.17g x %.17g @ %.17g at (%.17g, %.17g)",
              section == 2 ? "CircleIterator" : "SpiralIterator", ctr.x, ctr.y, rad, a.size(), b.size(), MAPARGS);
        for (size_t k = 0; k < a.size(); ++k)
            CHECK(a[k].i == b[k].i && a[k].j == b[k].j, "%s around (%.17g, %.17g) r %.17g: cell %zu is (%d, %d) in the restatement, (%d, %d) upstream; " "map %.17g x %.17g @ %.17g at (%.17g, %.17g)",
                  section == 2 ? "CircleIterator" : "SpiralIterator", ctr.x, ctr.y, rad, k, a[k].i, a[k].j, b[k].i, b[k].j, MAPARGS);
    } else if (section == 4) {
        // getSearchPolygon's rectangle (cpp:2496-2517: LU, RU, RD, LD) or a convex hexagon, around a cell centre; points on
        // cell centres (what checkCirclePolygonFoothold tests, cpp:2138), some exactly on an edge
        const fpo::Vec2 ctr = draw_position(r, om);
        const double rad = draw_radius(r, c.res);
        std::vector<fpo::Vec2> vs;
        if (r.below(2)) {
            vs = {{ctr.x + rad, ctr.y + 0.5 * rad}, {ctr.x + rad, ctr.y - 0.5 * rad}, {ctr.x - rad, ctr.y - 0.5 * rad}, {ctr.x - rad, ctr.y + 0.5 * rad}};
        } else {
            for (int k = 0; k < 6; ++k) vs.push_back({ctr.x + rad * std::cos(1.0471975511965976 * k + 0.3), ctr.y + 0.7 * rad * std::sin(1.0471975511965976 * k + 0.3)});
        }
        fpo::Polygon poly;
        for (const fpo::Vec2& v : vs) poly.addVertex(v);
        for (int k = 0; k < 8; ++k) {
            fpo::Vec2 p = draw_position(r, om);
            if (k < 2) p = {vs[0].x, nudge(ctr.y, r.below(5) - 2)};   // on the first edge's x
            if (k == 2) p = {nudge(ctr.x, r.below(5) - 2), vs[0].y};  // level with a vertex
            const bool a = poly.isInside(p), b = up_polygon(vs, p);
            CHECK(a == b, "Polygon::isInside(%.17g, %.17g): restatement %d, upstream %d (polygon around (%.17g, %.17g), r %.17g, %zu vertices)", p.x, p.y, a, b, ctr.x,
                  ctr.y, rad, vs.size());
        }
    } else {
        const fpo::Vec2 p = draw_position(r, om);
        const fpo::Vec2 len{c.res * r.uni(0.5, 30.0), c.res * r.uni(0.5, 30.0)};
        bool oka = false;
        const fpo::GridMap s = om.getSubmap(p, len, oka);
        const SubOut u = up_submap(um, p, len);
        CHECK(oka == u.ok, "getSubmap((%.17g, %.17g), (%.17g, %.17g)): restatement %d, upstream %d; " "map %.17g x %.17g @ %.17g at (%.17g, %.17g)", p.x, p.y, len.x, len.y,
              oka, u.ok, MAPARGS);
        if (!oka) return;
        CHECK(s.size.i == u.size.i && s.size.j == u.size.j && same_bits(s.position.x, u.position.x) && same_bits(s.position.y, u.position.y),
              "getSubmap((%.17g, %.17g), (%.17g, %.17g)): size (%d, %d) at (%.17g, %.17g) in the restatement, (%d, %d) at (%.17g, %.17g) upstream", p.x, p.y, len.x,
              len.y, s.size.i, s.size.j, s.position.x, s.position.y, u.size.i, u.size.j, u.position.x, u.position.y);
        CHECK(s.trav.size() == u.trav.size(), "getSubmap: %zu cells against %zu", s.trav.size(), u.trav.size());
        for (size_t k = 0; k < s.trav.size(); ++k) CHECK(s.trav[k] == u.trav[k], "getSubmap: cell %zu differs (%g against %g)", k, s.trav[k], u.trav[k]);
    }
    (void)tag;
}

}  // namespace

int main(int argc, char** argv) {
    const long n = argc > 1 ? std::atol(argv[1]) : 200000;
    static const char* const names[] = {"", "getIndex / getPosition", "CircleIterator", "SpiralIterator", "Polygon::isInside", "getSubmap"};
    for (int section = 1; section <= 5; ++section) {
        Rng r{0x5EEDull * 1000003ull + static_cast<uint64_t>(section)};
        for (long k = 0; k < n && !g_fail; ++k) one_case(r, section);
        if (g_fail) {
            std::printf("upstream_check: FIRST DIFFERENCE in section %d (%s) — oracle/fpo_gridmap.hpp does not restate this grid_map_core\n", section, names[section]);
            return 1;
        }
        std::printf("section %d (%s): %ld cases identical\n", section, names[section], n);
    }
#ifdef FPE_WITH_GRID_MAP
    std::printf("upstream_check: oracle/fpo_gridmap.hpp restates this grid_map_core on every case — parity PINNED on this machine\n");
#else
    std::printf("upstream_check: harness self-test only (built without FPE_WITH_GRID_MAP): nothing was compared with grid_map_core\n");
#endif
    return 0;
}
