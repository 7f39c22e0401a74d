// ORACLE — test infrastructure only (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline).  Never shipped, never
// on the product path.
//
// CPU restatement of the PRODUCER of the path's input: elevation layer -> traversability layer.
// The reference only launches that producer (`launch/mapping.launch:12-13`, `launch/all.launch:21-22`, README.md:29:
// leggedrobotics/traversability_estimation) and subscribes to its output (`FootholdPlanner.cpp:188`); neither the
// package nor its filter configuration is under /root/reference, and its version is not pinned anywhere
// (`package.xml`, `CMakeLists.txt:8-18`).  PARITY UNPINNED: what follows restates, from the published sources of
// that package as recalled (traversability_estimation_filters/src/{SlopeFilter,StepFilter,RoughnessFilter}.cpp,
// grid_map_filters/src/NormalVectorsFilter.cpp (area method), and the default chain of
// traversability_estimation/config/robot_filter_parameter.yaml), the arithmetic of its default filter chain:
//
//   surface normals : per valid cell, the points (x, y, z) of the valid cells in CircleIterator(centre, radius);
//                     mean; covariance NN * NN^T; the unit eigenvector of the smallest eigenvalue, turned to n_z >= 0;
//                     rank-deficient covariance (published: fullPivHouseholderQr().rank() < 3) -> (0, 0, 1).
//   slope           : acos(n_z) against critical_value; 1 - slope / critical below it, 0 at or above.
//   step            : pass 1 step_height = max - min of the valid cells in the first window; pass 2 over the second
//                     window: stepMax, nCells above critical; step = min(stepMax, nCells / nCritical * stepMax).
//   roughness       : sqrt(sum of squared distances to the plane through the mean with the cell's normal / (n - 1)).
//   traversability  : (1/3) * (slope + step + roughness) in float (MathExpressionFilter on float matrices).
//
// Build-defined where the published code leans on Eigen internals that cannot be restated bit for bit: sums run in
// iterator order (Eigen's reductions may re-associate), the symmetric eigenproblem is solved by cyclic Jacobi
// rotations, and the rank test is `lambda_min <= 3 eps lambda_max`.  Layers are float as in grid_map (Matrix =
// Eigen::MatrixXf, column-major); the filters read the float-rounded normals back, as the published filters do.
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

#include "fpo_gridmap.hpp"

namespace fpo {

struct FilterParams {
    double normalRadius;       // NormalVectorsFilter `radius` (0.05)
    double slopeCritical;      // SlopeFilter critical_value [rad] (1.0)
    double stepCritical;       // StepFilter critical_value [m] (0.12)
    double stepFirstRadius;    // first_window_radius (0.08)
    double stepSecondRadius;   // second_window_radius (0.08)
    int32_t stepCriticalCells; // critical_cell_number (4)
    int32_t pad;
    double roughnessCritical;  // RoughnessFilter critical_value (0.05)
    double roughnessRadius;    // estimation_radius (0.05)
};

// Eigen decomposition of a symmetric 3x3 matrix by cyclic Jacobi rotations (rows of `a` above the diagonal are used).
// Returns the eigenvalues in `w` and the eigenvectors in the COLUMNS of `v`.
static void jacobi3(double a[3][3], double w[3], double v[3][3]) {
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) v[r][c] = r == c ? 1.0 : 0.0;
    static const int P[3] = {0, 0, 1}, Q[3] = {1, 2, 2}, R[3] = {2, 1, 0};
    for (int sweep = 0; sweep < 12; ++sweep) {
        const double off = std::fabs(a[0][1]) + std::fabs(a[0][2]) + std::fabs(a[1][2]);
        if (off == 0.0) break;
        for (int k = 0; k < 3; ++k) {
            const int p = P[k], q = Q[k], r = R[k];
            const double apq = a[p][q];
            if (apq == 0.0) continue;
            if (std::fabs(apq) <= 2.3e-18 * (std::fabs(a[p][p]) + std::fabs(a[q][q]))) {  // negligible: no rotation
                a[p][q] = a[q][p] = 0.0;
                continue;
            }
            const double theta = (a[q][q] - a[p][p]) / (2.0 * apq);
            const double t = (theta >= 0.0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
            const double c = 1.0 / std::sqrt(t * t + 1.0);
            const double s = t * c;
            a[p][p] = a[p][p] - t * apq;
            a[q][q] = a[q][q] + t * apq;
            a[p][q] = a[q][p] = 0.0;
            const double arp = a[r][p], arq = a[r][q];
            a[r][p] = a[p][r] = c * arp - s * arq;
            a[r][q] = a[q][r] = s * arp + c * arq;
            for (int m = 0; m < 3; ++m) {
                const double vmp = v[m][p], vmq = v[m][q];
                v[m][p] = c * vmp - s * vmq;
                v[m][q] = s * vmp + c * vmq;
            }
        }
    }
    for (int k = 0; k < 3; ++k) w[k] = a[k][k];
}

struct Layers {
    int rows, cols;
    std::vector<float> nx, ny, nz, slope, stepHeight, step, rough, trav;  // column-major like grid_map::Matrix
    float& at(std::vector<float>& l, const Idx2& i) { return l[(size_t)i.i + (size_t)i.j * rows]; }
};

// Oracle-defined where upstream is undefined: a bounding box whose far corner rounds to the index `size` (see
// GridMap::getSubmap in fpo_gridmap.hpp) would make the published filters read one row / column past the layer; such
// iterator cells are skipped.
static void run_filters(const GridMap& map, const FilterParams& fp, Layers& L) {
    const float nan = std::numeric_limits<float>::quiet_NaN();
    const size_t n = (size_t)map.size.i * map.size.j;
    L.rows = map.size.i;
    L.cols = map.size.j;
    for (std::vector<float>* l : {&L.nx, &L.ny, &L.nz, &L.slope, &L.stepHeight, &L.step, &L.rough, &L.trav}) l->assign(n, nan);
    std::vector<double> px, py, pz;
    // ---- NormalVectorsFilter::computeWithArea + SlopeFilter::update ----
    for (int j = 0; j < map.size.j; ++j)
        for (int i = 0; i < map.size.i; ++i) {  // GridMapIterator (column-major linear index); cells are independent
            const Idx2 idx{i, j};
            if (!GridMap::isValid(map.elevAt(idx))) continue;
            Vec2 center;
            map.getPosition(idx, center);
            px.clear(); py.clear(); pz.clear();
            for (CircleIterator it(map, center, fp.normalRadius); !it.isPastEnd(); ++it) {
                if (!checkIfIndexInRange(*it, map.size)) continue;  // oracle-defined (see run_filters' header note)
                const float z = map.elevAt(*it);
                if (!GridMap::isValid(z)) continue;  // getPosition3 fails on invalid cells
                Vec2 p;
                map.getPosition(*it, p);
                px.push_back(p.x); py.push_back(p.y); pz.push_back(static_cast<double>(z));
            }
            const size_t np = px.size();
            double sx = 0.0, sy = 0.0, sz = 0.0;
            for (size_t k = 0; k < np; ++k) { sx += px[k]; sy += py[k]; sz += pz[k]; }
            const double mx = sx / static_cast<double>(np), my = sy / static_cast<double>(np), mz = sz / static_cast<double>(np);
            double a[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
            for (size_t k = 0; k < np; ++k) {
                const double dx = px[k] - mx, dy = py[k] - my, dz = pz[k] - mz;
                a[0][0] += dx * dx; a[0][1] += dx * dy; a[0][2] += dx * dz;
                a[1][1] += dy * dy; a[1][2] += dy * dz; a[2][2] += dz * dz;
            }
            a[1][0] = a[0][1]; a[2][0] = a[0][2]; a[2][1] = a[1][2];
            double w[3], v[3][3];
            jacobi3(a, w, v);
            int smallest = 0;
            double smallestValue = std::numeric_limits<double>::max(), largestValue = 0.0;
            for (int k = 0; k < 3; ++k) {
                if (w[k] < smallestValue) { smallestValue = w[k]; smallest = k; }
                if (w[k] > largestValue) largestValue = w[k];
            }
            double ex = v[0][smallest], ey = v[1][smallest], ez = v[2][smallest];
            if (!(smallestValue > 3.0 * DBL_EPSILON * largestValue)) { ex = 0.0; ey = 0.0; ez = 1.0; }  // rank < 3: z axis
            if (ez < 0.0) { ex = -ex; ey = -ey; ez = -ez; }  // normal_vector_positive_axis: z
            L.at(L.nx, idx) = static_cast<float>(ex);
            L.at(L.ny, idx) = static_cast<float>(ey);
            L.at(L.nz, idx) = static_cast<float>(ez);
            // SlopeFilter: reads the float layer back
            const double slope = std::acos(static_cast<double>(L.at(L.nz, idx)));
            L.at(L.slope, idx) = slope < fp.slopeCritical ? static_cast<float>(1.0 - slope / fp.slopeCritical) : 0.0f;
        }
    // ---- StepFilter::update, first iteration ----
    for (int j = 0; j < map.size.j; ++j)
        for (int i = 0; i < map.size.i; ++i) {
            const Idx2 idx{i, j};
            if (!GridMap::isValid(map.elevAt(idx))) continue;
            Vec2 center;
            map.getPosition(idx, center);
            double heightMax = 0.0, heightMin = 0.0;
            bool init = false;
            for (CircleIterator it(map, center, fp.stepFirstRadius); !it.isPastEnd(); ++it) {
                if (!checkIfIndexInRange(*it, map.size)) continue;
                const float z = map.elevAt(*it);
                if (!GridMap::isValid(z)) continue;
                const double height = static_cast<double>(z);
                if (!init) { heightMax = height; heightMin = height; init = true; continue; }
                if (height > heightMax) heightMax = height;
                if (height < heightMin) heightMin = height;
            }
            if (init) L.at(L.stepHeight, idx) = static_cast<float>(heightMax - heightMin);
        }
    // ---- StepFilter::update, second iteration ----
    for (int j = 0; j < map.size.j; ++j)
        for (int i = 0; i < map.size.i; ++i) {
            const Idx2 idx{i, j};
            Vec2 center;
            map.getPosition(idx, center);
            int nCells = 0;
            double stepMax = 0.0;
            bool isValid = false;
            for (CircleIterator it(map, center, fp.stepSecondRadius); !it.isPastEnd(); ++it) {
                if (!checkIfIndexInRange(*it, map.size)) continue;
                const float sh = L.at(L.stepHeight, *it);
                if (!GridMap::isValid(sh)) continue;
                isValid = true;
                if (static_cast<double>(sh) > stepMax) stepMax = static_cast<double>(sh);
                if (static_cast<double>(sh) > fp.stepCritical) nCells++;
            }
            if (isValid) {
                const double step = std::min(stepMax, static_cast<double>(nCells) / static_cast<double>(fp.stepCriticalCells) * stepMax);
                L.at(L.step, idx) = step < fp.stepCritical ? static_cast<float>(1.0 - step / fp.stepCritical) : 0.0f;
            }
        }
    // ---- RoughnessFilter::update ----
    for (int j = 0; j < map.size.j; ++j)
        for (int i = 0; i < map.size.i; ++i) {
            const Idx2 idx{i, j};
            if (!GridMap::isValid(L.at(L.nx, idx))) continue;  // "empty cell (hole in the map)"
            Vec2 center;
            map.getPosition(idx, center);
            px.clear(); py.clear(); pz.clear();
            for (CircleIterator it(map, center, fp.roughnessRadius); !it.isPastEnd(); ++it) {
                if (!checkIfIndexInRange(*it, map.size)) continue;
                const float z = map.elevAt(*it);
                if (!GridMap::isValid(z)) continue;
                Vec2 p;
                map.getPosition(*it, p);
                px.push_back(p.x); py.push_back(p.y); pz.push_back(static_cast<double>(z));
            }
            const size_t np = px.size();
            double sx = 0.0, sy = 0.0, sz = 0.0;
            for (size_t k = 0; k < np; ++k) { sx += px[k]; sy += py[k]; sz += pz[k]; }
            const double mx = sx / static_cast<double>(np), my = sy / static_cast<double>(np), mz = sz / static_cast<double>(np);
            const double normalX = L.at(L.nx, idx), normalY = L.at(L.ny, idx), normalZ = L.at(L.nz, idx);
            const double planeParameter = mx * normalX + my * normalY + mz * normalZ;
            double sum = 0.0;
            for (size_t k = 0; k < np; ++k) {
                const double dist = normalX * px[k] + normalY * py[k] + normalZ * pz[k] - planeParameter;
                sum += dist * dist;
            }
            const double roughness = std::sqrt(sum / (static_cast<double>(np) - 1.0));
            L.at(L.rough, idx) = roughness < fp.roughnessCritical ? static_cast<float>(1.0 - roughness / fp.roughnessCritical) : 0.0f;
        }
    // ---- MathExpressionFilter: (1.0 / 3.0) * (slope + step + roughness) on float matrices ----
    const float third = 1.0f / 3.0f;
    for (size_t k = 0; k < n; ++k) L.trav[k] = third * ((L.slope[k] + L.step[k]) + L.rough[k]);
}

}  // namespace fpo

extern "C" {

int fpo_filter_defaults(fpo::FilterParams* fp) {
    if (!fp) return -1;
    std::memset(fp, 0, sizeof(*fp));
    fp->normalRadius = 0.05;
    fp->slopeCritical = 1.0;
    fp->stepCritical = 0.12;
    fp->stepFirstRadius = 0.08;
    fp->stepSecondRadius = 0.08;
    fp->stepCriticalCells = 4;
    fp->roughnessCritical = 0.05;
    fp->roughnessRadius = 0.05;
    return 0;
}

// elevation and the eight output layers (nx, ny, nz, slope, step_height, step, roughness, traversability) are
// column-major rows x cols floats (cell (i, j) at i + j * rows), start index (0, 0).
int fpo_filters(int rows, int cols, double res, double posX, double posY, const float* elevation,
                const fpo::FilterParams* fp, float* out8) {
    if (rows <= 0 || cols <= 0 || !(res > 0.0) || !elevation || !fp || !out8) return -1;
    fpo::GridMap map;
    map.size = {rows, cols};
    map.res = res;
    map.length = {static_cast<double>(rows) * res, static_cast<double>(cols) * res};
    map.position = {posX, posY};
    map.elev.assign(elevation, elevation + (size_t)rows * cols);
    fpo::Layers L;
    fpo::run_filters(map, *fp, L);
    const size_t n = (size_t)rows * cols;
    const std::vector<float>* order[8] = {&L.nx, &L.ny, &L.nz, &L.slope, &L.stepHeight, &L.step, &L.rough, &L.trav};
    for (int l = 0; l < 8; ++l) std::memcpy(out8 + (size_t)l * n, order[l]->data(), n * sizeof(float));
    return 0;
}

}  // extern "C"
