// MOCK (tests/probe/ros_mock/README.md): the accessors of grid_map::GridMap the adapter calls — with storage, so that
// tests/probe/adapter_run.cpp can hand the adapter a map the way gridmapCallback's grid_map does (column-major layers,
// circular-buffer start index)
#pragma once
#include <map>
#include <string>
#include <vector>
namespace grid_map {
struct Vec2i { int v[2]; int operator()(int k) const { return v[k]; } };
struct Vec2d { double v[2]; double x() const { return v[0]; } double y() const { return v[1]; } };
struct Matrix {
    std::vector<float> v;  // column-major, as Eigen::MatrixXf
    const float* data() const { return v.data(); }
    float& operator()(int i, int j);  // (declared for tests/probe/upstream_check.cpp's parse check: never defined, never linked)
};
// What tests/probe/upstream_check.cpp reads of the upstream types (Eigen::Vector2d / Array2d / Array2i upstream): DECLARATIONS
// ONLY, for `g++ -fsyntax-only -DFPE_WITH_GRID_MAP` — the real comparison links the real grid_map_core on the maintainer's machine.
struct Position {
    Position();
    Position(double x, double y);
    double x() const;
    double y() const;
};
struct Length {
    Length(double x, double y);
};
struct Index {
    Index();
    Index(int i, int j);
    int operator()(int k) const;
};
struct GridMap {
    Vec2i size{{0, 0}}, startIndex{{0, 0}};
    Vec2d position{{0.0, 0.0}};
    double resolution = 0.0;
    std::map<std::string, Matrix> layers;
    bool exists(const std::string& name) const { return layers.count(name) != 0; }
    Vec2i getSize() const { return size; }
    double getResolution() const { return resolution; }
    Vec2d getPosition() const { return position; }
    Vec2i getStartIndex() const { return startIndex; }
    const Matrix& operator[](const std::string& name) const { return layers.at(name); }
    // ---- declared for upstream_check.cpp's parse check (see Position above) ----
    GridMap() = default;
    explicit GridMap(const std::vector<std::string>& layerNames);
    void setGeometry(const Length& length, double resolution, const Position& position);
    Matrix& operator[](const std::string& name);
    bool getIndex(const Position& position, Index& index) const;
    bool getPosition(const Index& index, Position& position) const;
    GridMap getSubmap(const Position& position, const Length& length, bool& isSuccess) const;
    float at(const std::string& layer, const Index& index) const;
};
struct CircleIterator {
    CircleIterator(const GridMap& map, const Position& center, double radius);
    bool isPastEnd() const;
    CircleIterator& operator++();
    const Index& operator*() const;
};
struct SpiralIterator {
    SpiralIterator(GridMap& map, const Position& center, double radius);
    bool isPastEnd() const;
    SpiralIterator& operator++();
    const Index& operator*() const;
};
struct Polygon {
    void addVertex(const Position& vertex);
    bool isInside(const Position& point) const;
};
}
