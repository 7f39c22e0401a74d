// MOCK (tests/probe/ros_mock/README.md): the accessors of grid_map::GridMap the adapter calls
#pragma once
#include <string>
namespace grid_map {
struct Vec2i { int v[2]; int operator()(int k) const { return v[k]; } };
struct Vec2d { double v[2]; double x() const { return v[0]; } double y() const { return v[1]; } };
struct Matrix { const float* data() const { return nullptr; } };
struct GridMap {
    bool exists(const std::string&) const { return true; }
    Vec2i getSize() const { return {}; }
    double getResolution() const { return 0; }
    Vec2d getPosition() const { return {}; }
    Vec2i getStartIndex() const { return {}; }
    const Matrix& operator[](const std::string&) const { static Matrix m; return m; }
};
}
