// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped engine; only
// tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call it.
//
// PARITY UNPINNED: the reference has no tests/golden vectors for this path and grid_map_core is not
// vendored under /root/reference nor installed in this image.  This file is the
// ONE place that freezes the *assumed upstream semantics* of ANYbotics/grid_map 1.6.x
// (grid_map_core/src/GridMapMath.cpp, GridMap.cpp, Polygon.cpp, iterators/*.cpp), restated
// literally (iterator objects, vectors, operation order of every f64 expression) so that it can be
// audited against upstream later.  Call sites in the reference that rely on each piece are cited.
//
// Conventions: index (i, j) = (row, col); i grows toward -x, j toward -y; index (0,0) is the cell
// with the largest x and y.  Layer storage is column-major f32 (Eigen::MatrixXf): (i,j) at i+j*rows.
// Only start index (0,0) is supported (upstream SpiralIterator ignores the
// circular-buffer start index, so parity is only defined for canonical maps).
#pragma once
#include <cmath>
#include <cfloat>
#include <cstdint>
#include <limits>
#include <vector>

namespace fpo {

struct Vec2 {
    double x, y;
    double& operator[](int k) { return k == 0 ? x : y; }
    double operator[](int k) const { return k == 0 ? x : y; }
};
struct Idx2 {
    int i, j;
    int& operator[](int k) { return k == 0 ? i : j; }
    int operator[](int k) const { return k == 0 ? i : j; }
};

// ---- GridMapMath.cpp ------------------------------------------------------------------------

// getVectorToOrigin: 0.5 * mapLength.
inline Vec2 vectorToOrigin(const Vec2& len) { return {0.5 * len.x, 0.5 * len.y}; }

// getVectorToFirstCell: vectorToOrigin - 0.5 * resolution.
inline Vec2 vectorToFirstCell(const Vec2& len, double res) {
    Vec2 o = vectorToOrigin(len);
    return {o.x - 0.5 * res, o.y - 0.5 * res};
}

inline bool checkIfIndexInRange(const Idx2& idx, const Idx2& size) {
    return idx.i >= 0 && idx.j >= 0 && idx.i < size.i && idx.j < size.j;
}

// getPositionFromIndex (start index 0): position = mapPosition + offset + resolution * (-index).
// Used by reference cpp:2098, 2105, 2136, 1702.., 1816 (GridMap::getPosition) and by every iterator.
inline bool getPositionFromIndex(Vec2& position, const Idx2& index, const Vec2& len, const Vec2& mapPos,
                                 double res, const Idx2& size) {
    if (!checkIfIndexInRange(index, size)) return false;
    Vec2 off = vectorToFirstCell(len, res);
    // Eigen evaluates (mapPosition + offset) + (resolution * indexVector) per coefficient;
    // indexVector = (-I * index).cast<double>().
    position.x = (mapPos.x + off.x) + res * static_cast<double>(-index.i);
    position.y = (mapPos.y + off.y) + res * static_cast<double>(-index.j);
    return true;
}

// checkIfPositionWithinMap: positionTransformed = -I * (position - mapPosition - offset).
inline bool checkIfPositionWithinMap(const Vec2& position, const Vec2& len, const Vec2& mapPos) {
    Vec2 off = vectorToOrigin(len);
    double tx = -((position.x - mapPos.x) - off.x);
    double ty = -((position.y - mapPos.y) - off.y);
    return tx >= 0.0 && ty >= 0.0 && tx < len.x && ty < len.y;
}

// getIndexFromPosition (start index 0): indexVector = (position - offset - mapPosition) / resolution;
// index = -(int)indexVector (C++ truncation toward zero).  The index is written even when the
// position is outside the map (return false).  Reference call sites: cpp:1703.., SpiralIterator ctor.
inline bool getIndexFromPosition(Idx2& index, const Vec2& position, const Vec2& len, const Vec2& mapPos,
                                 double res, const Idx2& /*size*/) {
    Vec2 off = vectorToOrigin(len);
    double vx = ((position.x - off.x) - mapPos.x) / res;
    double vy = ((position.y - off.y) - mapPos.y) / res;
    index.i = -static_cast<int>(vx);
    index.j = -static_cast<int>(vy);
    return checkIfPositionWithinMap(position, len, mapPos);
}

// boundPositionToRange: note the position is ALWAYS rewritten (shift, clamp, shift back), even
// when no clamping happens — the round trip can move it by an ulp.
inline void boundPositionToRange(Vec2& position, const Vec2& len, const Vec2& mapPos) {
    Vec2 o = vectorToOrigin(len);
    double shifted[2] = {(position.x - mapPos.x) + o.x, (position.y - mapPos.y) + o.y};
    const double pos[2] = {position.x, position.y};
    const double l[2] = {len.x, len.y};
    for (int k = 0; k < 2; ++k) {
        double epsilon = 10.0 * std::numeric_limits<double>::epsilon();
        if (std::fabs(pos[k]) > 1.0) epsilon *= std::fabs(pos[k]);
        if (shifted[k] <= 0) {
            shifted[k] = epsilon;
            continue;
        }
        if (shifted[k] >= l[k]) {
            shifted[k] = l[k] - epsilon;
            continue;
        }
    }
    position.x = (shifted[0] + mapPos.x) - o.x;
    position.y = (shifted[1] + mapPos.y) - o.y;
}

// getSubmapInformation (start index 0).  transform = -I, so "top left" = request + 0.5*length.
struct SubmapInfo {
    Idx2 topLeft;   // index of the submap's (0,0) cell in the parent map
    Idx2 size;      // rows, cols
    Vec2 position;  // submap centre position
    Vec2 length;
    Idx2 requestedIndexInSubmap;
};
inline bool getSubmapInformation(SubmapInfo& out, const Vec2& reqPos, const Vec2& reqLen, const Vec2& len,
                                 const Vec2& mapPos, double res, const Idx2& size) {
    Vec2 topLeftPosition = {reqPos.x - (-0.5 * reqLen.x), reqPos.y - (-0.5 * reqLen.y)};
    boundPositionToRange(topLeftPosition, len, mapPos);
    if (!getIndexFromPosition(out.topLeft, topLeftPosition, len, mapPos, res, size)) return false;
    Idx2 topLeftIndex = out.topLeft;

    Vec2 bottomRightPosition = {reqPos.x + (-0.5 * reqLen.x), reqPos.y + (-0.5 * reqLen.y)};
    boundPositionToRange(bottomRightPosition, len, mapPos);
    Idx2 bottomRightIndex;
    if (!getIndexFromPosition(bottomRightIndex, bottomRightPosition, len, mapPos, res, size)) return false;

    Vec2 topLeftCorner;
    if (!getPositionFromIndex(topLeftCorner, out.topLeft, len, mapPos, res, size)) return false;
    // topLeftCorner -= transform * Constant(0.5*resolution)  ==  -= -(0.5*res)
    topLeftCorner.x = topLeftCorner.x - (-(0.5 * res));
    topLeftCorner.y = topLeftCorner.y - (-(0.5 * res));

    out.size = {bottomRightIndex.i - topLeftIndex.i + 1, bottomRightIndex.j - topLeftIndex.j + 1};
    out.length = {static_cast<double>(out.size.i) * res, static_cast<double>(out.size.j) * res};
    Vec2 so = vectorToOrigin(out.length);
    out.position = {topLeftCorner.x - so.x, topLeftCorner.y - so.y};
    if (!getIndexFromPosition(out.requestedIndexInSubmap, reqPos, out.length, out.position, res, out.size))
        return false;
    return true;
}

// ---- GridMap.cpp ------------------------------------------------------------------------------

struct GridMap {
    Idx2 size{0, 0};
    double res = 0.0;
    Vec2 length{0, 0};
    Vec2 position{0, 0};
    std::vector<float> trav;  // "traversability", column-major
    std::vector<float> elev;  // "elevation", column-major (may be empty in submaps)

    // setGeometry(length, resolution, position): size = round(length/res), length = size*res.
    void setGeometry(const Vec2& len, double resolution, const Vec2& pos) {
        size.i = static_cast<int>(std::round(len.x / resolution));
        size.j = static_cast<int>(std::round(len.y / resolution));
        res = resolution;
        length = {static_cast<double>(size.i) * res, static_cast<double>(size.j) * res};
        position = pos;
    }
    float travAt(const Idx2& idx) const { return trav[(size_t)idx.i + (size_t)idx.j * size.i]; }
    float elevAt(const Idx2& idx) const { return elev[(size_t)idx.i + (size_t)idx.j * size.i]; }
    bool getPosition(const Idx2& idx, Vec2& p) const {
        return getPositionFromIndex(p, idx, length, position, res, size);
    }
    bool getIndex(const Vec2& p, Idx2& idx) const {
        return getIndexFromPosition(idx, p, length, position, res, size);
    }
    // GridMap::isValid(index, layer) = isfinite(at(layer, index)).  cpp:2055, 2132, 2532.
    static bool isValid(float v) { return std::isfinite(v); }

    // GridMap::getSubmap(position, length, isSuccess): geometry via getSubmapInformation +
    // setGeometry(SubmapGeometry), data copied cell by cell ("traversability" only is needed by
    // cpp:1650).  cpp:1627.
    // withElevation: also copy the "elevation" layer (upstream copies every layer; the centroid method reads the
    // traversability only, the opt track's gait-cycle submap is asked for heights too, cpp:1290-1314).
    GridMap getSubmap(const Vec2& p, const Vec2& len, bool& isSuccess, SubmapInfo* infoOut = nullptr,
                      bool withElevation = false) const {
        GridMap sub;
        SubmapInfo info;
        isSuccess = getSubmapInformation(info, p, len, length, position, res, size);
        if (!isSuccess) return sub;
        // GridMap::getSubmap then asks getBufferRegionsForSubmap for the region's buffer pieces, which fails when
        // top-left + size reaches past the map (GridMapMath.cpp: `(index + submapBufferSize > bufferSize).any()`).
        // That happens when a corner bounded onto the map's far edge rounds to the index `size` ((eps - len) / res with
        // len / res a hair above the integer, e.g. 280 * 0.04 / 0.04): without the test the copy below would read one
        // row / column past the layer.
        if (info.topLeft.i + info.size.i > size.i || info.topLeft.j + info.size.j > size.j) {
            isSuccess = false;
            return sub;
        }
        sub.setGeometry(info.length, res, info.position);
        sub.trav.resize((size_t)sub.size.i * sub.size.j);
        for (int j = 0; j < sub.size.j; ++j)
            for (int i = 0; i < sub.size.i; ++i)
                sub.trav[(size_t)i + (size_t)j * sub.size.i] = travAt({info.topLeft.i + i, info.topLeft.j + j});
        if (withElevation && !elev.empty()) {
            sub.elev.resize((size_t)sub.size.i * sub.size.j);
            for (int j = 0; j < sub.size.j; ++j)
                for (int i = 0; i < sub.size.i; ++i)
                    sub.elev[(size_t)i + (size_t)j * sub.size.i] = elevAt({info.topLeft.i + i, info.topLeft.j + j});
        }
        if (infoOut) *infoOut = info;
        isSuccess = true;
        return sub;
    }
};

// ---- Polygon.cpp: Polygon::isInside (PNPOLY).  cpp:2138 --------------------------------------
struct Polygon {
    std::vector<Vec2> vertices;
    void addVertex(const Vec2& v) { vertices.push_back(v); }
    bool isInside(const Vec2& point) const {
        int cross = 0;
        const int n = static_cast<int>(vertices.size());
        for (int i = 0, j = n - 1; i < n; j = i++) {
            if (((vertices[i].y > point.y) != (vertices[j].y > point.y)) &&
                (point.x < (vertices[j].x - vertices[i].x) * (point.y - vertices[i].y) /
                                   (vertices[j].y - vertices[i].y) +
                               vertices[i].x)) {
                cross++;
            }
        }
        return (cross % 2) != 0;
    }
};

// ---- iterators/SubmapIterator.cpp + CircleIterator.cpp.  cpp:2048, 2126, 2529 ----------------
// Row-major walk (i outer, j inner) over the bounding box [start, start+size) and the in-circle
// filter `squareNorm <= radius^2` on f64 cell-centre positions.
class CircleIterator {
public:
    CircleIterator(const GridMap& map, const Vec2& center, double radius)
        : map_(map), center_(center), radius_(radius) {
        radiusSquare_ = std::pow(radius_, 2);
        // findSubmapParameters
        Vec2 topLeft = {center.x + radius, center.y + radius};
        Vec2 bottomRight = {center.x - radius, center.y - radius};
        boundPositionToRange(topLeft, map.length, map.position);
        boundPositionToRange(bottomRight, map.length, map.position);
        getIndexFromPosition(start_, topLeft, map.length, map.position, map.res, map.size);
        Idx2 endIndex;
        getIndexFromPosition(endIndex, bottomRight, map.length, map.position, map.res, map.size);
        subSize_ = {endIndex.i - start_.i + 1, endIndex.j - start_.j + 1};  // getSubmapSizeFromCornerIndeces
        sub_ = {0, 0};
        pastEnd_ = !(subSize_.i > 0 && subSize_.j > 0);
        if (!pastEnd_ && !isInside()) ++(*this);
    }
    bool isPastEnd() const { return pastEnd_; }
    Idx2 operator*() const { return {start_.i + sub_.i, start_.j + sub_.j}; }
    CircleIterator& operator++() {
        stepInternal();
        for (; !pastEnd_; stepInternal())
            if (isInside()) break;
        return *this;
    }
    Idx2 bboxStart() const { return start_; }
    Idx2 bboxSize() const { return subSize_; }

private:
    void stepInternal() {  // incrementIndexForSubmap: column (j) first, then next row
        if (pastEnd_) return;
        if (sub_.j + 1 < subSize_.j) {
            sub_.j++;
        } else {
            sub_.i++;
            sub_.j = 0;
        }
        if (!checkIfIndexInRange(sub_, subSize_)) pastEnd_ = true;
    }
    bool isInside() const {
        Vec2 position{0, 0};
        getPositionFromIndex(position, **this, map_.length, map_.position, map_.res, map_.size);
        double dx = position.x - center_.x, dy = position.y - center_.y;
        double squareNorm = dx * dx + dy * dy;
        return squareNorm <= radiusSquare_;
    }
    const GridMap& map_;
    Vec2 center_;
    double radius_, radiusSquare_;
    Idx2 start_, subSize_, sub_;
    bool pastEnd_;
};

// ---- iterators/SpiralIterator.cpp.  cpp:2095 ---------------------------------------------------
// Rings are generated by a walk from offset (d,0) and CONSUMED FROM THE BACK (operator* = back(),
// ++ = pop_back()).  Only rings nRings-1 and nRings apply the in-radius filter.
// Oracle-defined semantics where upstream is undefined: an empty intermediate ring is skipped
// (upstream 1.6.x would dereference back() of an empty vector; can only happen at map borders).
class SpiralIterator {
public:
    SpiralIterator(const GridMap& map, const Vec2& center, double radius)
        : map_(map), center_(center), radius_(radius), distance_(0) {
        radiusSquare_ = radius_ * radius_;
        map.getIndex(center_, indexCenter_);
        nRings_ = static_cast<unsigned int>(std::ceil(radius_ / map.res));
        if (checkIfIndexInRange(indexCenter_, map.size))
            pointsRing_.push_back(indexCenter_);
        else
            while (pointsRing_.empty() && !isPastEnd()) generateRing();
    }
    bool isPastEnd() const { return distance_ == nRings_ && pointsRing_.empty(); }
    Idx2 operator*() const { return pointsRing_.back(); }
    SpiralIterator& operator++() {
        pointsRing_.pop_back();
        while (pointsRing_.empty() && !isPastEnd()) generateRing();
        return *this;
    }
    unsigned int nRings() const { return nRings_; }
    unsigned int currentRing() const { return distance_; }
    Idx2 indexCenter() const { return indexCenter_; }

private:
    static int signum(int v) { return (v > 0) - (v < 0); }
    bool isInside(const Idx2& index) const {
        Vec2 position{0, 0};
        getPositionFromIndex(position, index, map_.length, map_.position, map_.res, map_.size);
        double dx = position.x - center_.x, dy = position.y - center_.y;
        return (dx * dx + dy * dy) <= radiusSquare_;
    }
    static int intNorm(int x, int y) {  // (int) Eigen::Vector2d(x, y).norm()
        return static_cast<int>(std::sqrt(static_cast<double>(x) * x + static_cast<double>(y) * y));
    }
    void generateRing() {
        distance_++;
        const int d = static_cast<int>(distance_);
        Idx2 point{d, 0};
        Idx2 pointInMap, normal;
        do {
            pointInMap.i = point.i + indexCenter_.i;
            pointInMap.j = point.j + indexCenter_.j;
            if (checkIfIndexInRange(pointInMap, map_.size)) {
                if (distance_ == nRings_ || distance_ == nRings_ - 1) {
                    if (isInside(pointInMap)) pointsRing_.push_back(pointInMap);
                } else {
                    pointsRing_.push_back(pointInMap);
                }
            }
            normal.i = -signum(point.j);
            normal.j = signum(point.i);
            if (normal.i != 0 && intNorm(point.i + normal.i, point.j) == d)
                point.i += normal.i;
            else if (normal.j != 0 && intNorm(point.i, point.j + normal.j) == d)
                point.j += normal.j;
            else {
                point.i += normal.i;
                point.j += normal.j;
            }
        } while (point.i != d || point.j != 0);
    }
    const GridMap& map_;
    Vec2 center_;
    Idx2 indexCenter_;
    double radius_, radiusSquare_;
    unsigned int nRings_, distance_;
    std::vector<Idx2> pointsRing_;
};

}  // namespace fpo
