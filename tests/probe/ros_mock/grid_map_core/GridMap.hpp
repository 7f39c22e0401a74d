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
};
}
