// fpe_gridmath.hpp — grid_map geometry as closed-form per-axis functions, shared by the HIP
// kernels and the host side of the engine.
//
// ASSUMED UPSTREAM SEMANTICS (ANYbotics/grid_map 1.6.x, grid_map_core/src/GridMapMath.cpp; not
// vendored by the reference).  Each function keeps the f64 expression ORDER of
// the upstream routine it stands for, because the reference's chosen indices depend on it; the
// library must be compiled with -ffp-contract=off.  The device map is canonical (start index 0),
// so the circular-buffer wrap of upstream is the identity here.
//
// Unlike the oracle (oracle/fpo_gridmap.hpp), which restates the iterator OBJECTS, this file
// exposes the arithmetic per axis so a wavefront can evaluate many cells at once.
#pragma once
#include <cfloat>
#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define FPE_HD __host__ __device__ __forceinline__
#else
#define FPE_HD inline
#endif

namespace fpe {

// Geometry of the current map snapshot (plain data, passed to kernels by value).
struct MapGeom {
    int32_t rows, cols;  // size(0) spans x, size(1) spans y
    double res;
    double lenX, lenY;    // size * res                       (GridMap::setGeometry)
    double posX, posY;    // map centre
    double orgX, orgY;    // 0.5 * len                        (getVectorToOrigin)
    double baseX, baseY;  // pos + (org - 0.5*res)            (mapPosition + getVectorToFirstCell)
    double rinv;          // fl(1/res): used only to PREDICT a quotient, never to produce one
};

FPE_HD MapGeom make_geom(int rows, int cols, double res, double px, double py) {
    MapGeom g;
    g.rows = rows;
    g.cols = cols;
    g.res = res;
    g.lenX = static_cast<double>(rows) * res;
    g.lenY = static_cast<double>(cols) * res;
    g.posX = px;
    g.posY = py;
    g.orgX = 0.5 * g.lenX;
    g.orgY = 0.5 * g.lenY;
    g.baseX = px + (g.orgX - 0.5 * res);
    g.baseY = py + (g.orgY - 0.5 * res);
    g.rinv = 1.0 / res;
    return g;
}

// getPositionFromIndex, one axis: (mapPosition + offset) + resolution * double(-index).
FPE_HD double cell_pos(double base, double res, int idx) { return base + res * static_cast<double>(-idx); }

// getIndexFromPosition, one axis: -(int)(((position - offset) - mapPosition) / resolution).
FPE_HD int index_of(double x, double org, double pos, double res) {
    return -static_cast<int>(((x - org) - pos) / res);
}

// Same value as index_of, without the f64 division in the common case.  q' = n * fl(1/res) is
// within 3.4e-16*|q| of the correctly rounded q = n / res, so trunc(q') == trunc(q) whenever q' is
// farther than eps = 4.5e-16*|q'| from an integer; otherwise (exact ties such as cell-centre +
// half-cell radii, or NaN) the true division decides.  Bit-identical to index_of by construction.
// (hipcc if-converts the rare branch below, i.e. the division is speculated; measured on MI355X
// that straight-line form is ~2.5% faster for the plan kernel than keeping a real branch.)
FPE_HD int index_of_fast(double x, double org, double pos, double res, double rinv) {
    const double n = (x - org) - pos;
    const double qf = n * rinv;
    double k = trunc(qf);
    const double fr = fabs(qf - k);
    const double eps = fabs(qf) * 4.5e-16 + 1e-290;
    if (__builtin_expect(!(fr > eps && fr < 1.0 - eps), 0)) {
        k = trunc(n / res);
    }
    return -static_cast<int>(k);
}

// checkIfPositionWithinMap, one axis: t = -((position - mapPosition) - offset); 0 <= t < length.
FPE_HD bool within_axis(double x, double org, double pos, double len) {
    const double t = -((x - pos) - org);
    return t >= 0.0 && t < len;
}

// boundPositionToRange, one axis.  The position is rewritten even when nothing is clamped.
FPE_HD double bound_axis(double x, double org, double pos, double len) {
    double s = (x - pos) + org;
    double eps = 10.0 * DBL_EPSILON;
    if (fabs(x) > 1.0) eps *= fabs(x);
    if (s <= 0) {
        s = eps;
    } else if (s >= len) {
        s = len - eps;
    }
    return (s + pos) - org;
}

FPE_HD bool in_range(int i, int j, int rows, int cols) { return i >= 0 && j >= 0 && i < rows && j < cols; }

// CircleIterator::findSubmapParameters: bounding box of the disc (centre c, radius r) in cells.
struct BBox {
    int i0, j0, ni, nj;
};
FPE_HD BBox circle_bbox(const MapGeom& g, double cx, double cy, double r) {
    const double tlx = bound_axis(cx + r, g.orgX, g.posX, g.lenX);
    const double tly = bound_axis(cy + r, g.orgY, g.posY, g.lenY);
    const double brx = bound_axis(cx - r, g.orgX, g.posX, g.lenX);
    const double bry = bound_axis(cy - r, g.orgY, g.posY, g.lenY);
    BBox b;
    b.i0 = index_of(tlx, g.orgX, g.posX, g.res);
    b.j0 = index_of(tly, g.orgY, g.posY, g.res);
    b.ni = index_of(brx, g.orgX, g.posX, g.res) - b.i0 + 1;
    b.nj = index_of(bry, g.orgY, g.posY, g.res) - b.j0 + 1;
    return b;
}

FPE_HD BBox circle_bbox_fast(const MapGeom& g, double cx, double cy, double r) {
    const double tlx = bound_axis(cx + r, g.orgX, g.posX, g.lenX);
    const double tly = bound_axis(cy + r, g.orgY, g.posY, g.lenY);
    const double brx = bound_axis(cx - r, g.orgX, g.posX, g.lenX);
    const double bry = bound_axis(cy - r, g.orgY, g.posY, g.lenY);
    BBox b;
    b.i0 = index_of_fast(tlx, g.orgX, g.posX, g.res, g.rinv);
    b.j0 = index_of_fast(tly, g.orgY, g.posY, g.res, g.rinv);
    b.ni = index_of_fast(brx, g.orgX, g.posX, g.res, g.rinv) - b.i0 + 1;
    b.nj = index_of_fast(bry, g.orgY, g.posY, g.res, g.rinv) - b.j0 + 1;
    return b;
}

// CircleIterator::isInside / SpiralIterator::isInside: squared cell-centre distance <= r^2.
FPE_HD bool cell_in_disc(const MapGeom& g, int i, int j, double cx, double cy, double r2) {
    const double dx = cell_pos(g.baseX, g.res, i) - cx;
    const double dy = cell_pos(g.baseY, g.res, j) - cy;
    return (dx * dx + dy * dy) <= r2;
}

// getSubmapInformation for the centroid rectangle (cpp:1615-1627): top-left index, size and the
// submap's own geometry (its getPosition is used at cpp:1816).  ok=false <=> getSubmap fails.
struct Submap {
    int i0, j0, ni, nj;
    double baseX, baseY;  // submap position + (0.5*sublen - 0.5*res)
    bool ok;
};
FPE_HD Submap submap_info(const MapGeom& g, double px, double py, double lx, double ly) {
    Submap s;
    s.ok = false;
    s.i0 = s.j0 = 0;
    s.ni = s.nj = 0;
    s.baseX = s.baseY = 0.0;
    const double tlx = bound_axis(px - (-0.5 * lx), g.orgX, g.posX, g.lenX);
    const double tly = bound_axis(py - (-0.5 * ly), g.orgY, g.posY, g.lenY);
    s.i0 = index_of_fast(tlx, g.orgX, g.posX, g.res, g.rinv);
    s.j0 = index_of_fast(tly, g.orgY, g.posY, g.res, g.rinv);
    if (!(within_axis(tlx, g.orgX, g.posX, g.lenX) && within_axis(tly, g.orgY, g.posY, g.lenY))) return s;
    const double brx = bound_axis(px + (-0.5 * lx), g.orgX, g.posX, g.lenX);
    const double bry = bound_axis(py + (-0.5 * ly), g.orgY, g.posY, g.lenY);
    const int i1 = index_of_fast(brx, g.orgX, g.posX, g.res, g.rinv);
    const int j1 = index_of_fast(bry, g.orgY, g.posY, g.res, g.rinv);
    if (!(within_axis(brx, g.orgX, g.posX, g.lenX) && within_axis(bry, g.orgY, g.posY, g.lenY))) return s;
    if (!in_range(s.i0, s.j0, g.rows, g.cols)) return s;  // getPositionFromIndex(topLeft) range check
    if (i1 >= g.rows || j1 >= g.cols) return s;            // getBufferRegionsForSubmap: the region must fit the buffer
    const double cornerX = cell_pos(g.baseX, g.res, s.i0) - (-(0.5 * g.res));
    const double cornerY = cell_pos(g.baseY, g.res, s.j0) - (-(0.5 * g.res));
    s.ni = i1 - s.i0 + 1;
    s.nj = j1 - s.j0 + 1;
    const double subLenX = static_cast<double>(s.ni) * g.res;
    const double subLenY = static_cast<double>(s.nj) * g.res;
    const double subOrgX = 0.5 * subLenX, subOrgY = 0.5 * subLenY;
    const double subPosX = cornerX - subOrgX, subPosY = cornerY - subOrgY;
    // last step of getSubmapInformation: the requested position must index inside the submap
    if (!(within_axis(px, subOrgX, subPosX, subLenX) && within_axis(py, subOrgY, subPosY, subLenY))) return s;
    s.baseX = subPosX + (subOrgX - 0.5 * g.res);
    s.baseY = subPosY + (subOrgY - 0.5 * g.res);
    s.ok = true;
    return s;
}

// Polygon::isInside (PNPOLY), vertices in arrays.  cpp:2138.
FPE_HD bool polygon_inside(const double* vx, const double* vy, int n, double px, double py) {
    int cross = 0;
    for (int i = 0, j = n - 1; i < n; j = i++) {
        if (((vy[i] > py) != (vy[j] > py)) &&
            (px < (vx[j] - vx[i]) * (py - vy[i]) / (vy[j] - vy[i]) + vx[i])) {
            cross++;
        }
    }
    return (cross & 1) != 0;
}

// Same result as polygon_inside.  For an edge with vx[j] == vx[i] the literal intersection
// (0 * t) / d + vx[i] is exactly vx[i] (t finite, d != 0 because the edge straddles py), so the
// division is skipped; every edge of the reference's rectangle that can straddle is of that kind.
FPE_HD bool polygon_inside_fast(const double* vx, const double* vy, int n, double px, double py) {
    int cross = 0;
    for (int i = 0, j = n - 1; i < n; j = i++) {
        if ((vy[i] > py) != (vy[j] > py)) {
            const double ex = vx[j] - vx[i];
            const double t = py - vy[i];
            double xi = vx[i];
            if (!(ex == 0.0 && fabs(t) <= DBL_MAX)) {
                // slanted edge: the literal intersection, division included
                xi = ex * t / (vy[j] - vy[i]) + vx[i];
            }
            if (px < xi) cross++;
        }
    }
    return (cross & 1) != 0;
}

// n / d for n*d < 2^32 via one 32x32->64 multiply (magic = floor((2^32-1)/d) + 1; d = 1 wraps
// the magic to 0, which stands for "divide by one").
FPE_HD uint32_t fastdiv_magic(uint32_t d) { return static_cast<uint32_t>(0xFFFFFFFFu / d) + 1u; }
FPE_HD uint32_t fastdiv(uint32_t n, uint32_t magic) {
    return magic == 0u ? n : static_cast<uint32_t>((static_cast<uint64_t>(n) * magic) >> 32);
}

}  // namespace fpe
