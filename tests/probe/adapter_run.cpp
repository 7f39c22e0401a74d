// tests/probe/adapter_run.cpp — the ROS adapter (csrc/ros_adapter/fpe_ros_adapter.hpp) RUN against the mock ROS / grid_map
// types of tests/probe/ros_mock and the real libfpe.so: a map file in, every call of the adapter on a list of poses, the
// messages out as text (doubles as C99 hex floats).  tests/test_gpu_ros_adapter.py builds this, runs it on the GPU box and
// compares the output with the same service calls made through the Python binding.  Test infrastructure, not product.
#define FPE_WITH_ROS 1
#include "fpe_ros_adapter.hpp"

#include <cstdio>
#include <cstdlib>
#include <fstream>

static void put_msg(FILE* f, const char* tag, const foothold_planner_msgs::GlobalFootholds& m) {
    std::fprintf(f, "%s %d %d %d %zu\n", tag, m.success ? 1 : 0, static_cast<int>(m.gait_cycles), static_cast<int>(m.gait_cycles_succeed), m.footholds.size());
    for (const auto& h : m.footholds)
        std::fprintf(f, " %a %a %a %d %d\n", h.point.x, h.point.y, h.point.z, static_cast<int>(h.foot_id), static_cast<int>(h.gait_cycle_id));
}
static void put_vec(FILE* f, const char* tag, const std::vector<double>& v) {
    std::fprintf(f, "%s %zu", tag, v.size());
    for (double x : v) std::fprintf(f, " %a", x);
    std::fprintf(f, "\n");
}
static void put_path(FILE* f, const char* tag, const nav_msgs::Path& p) {
    std::fprintf(f, "%s %zu", tag, p.poses.size());
    for (const auto& q : p.poses) std::fprintf(f, " %a %a %a", q.pose.position.x, q.pose.position.y, q.pose.position.z);
    std::fprintf(f, "\n");
}

int main(int argc, char** argv) {
    if (argc != 3) return 2;
    // input: int32 rows, cols, start i, start j, gait cycles, n poses; f64 resolution, position x, y; then the two layers
    // (rows * cols f32 each, column-major buffer with the start index applied) and n x 3 f64 initial poses
    std::ifstream in(argv[1], std::ios::binary);
    int32_t hdr[6];
    double geo[3];
    in.read(reinterpret_cast<char*>(hdr), sizeof(hdr));
    in.read(reinterpret_cast<char*>(geo), sizeof(geo));
    grid_map::GridMap map;
    map.size = {{hdr[0], hdr[1]}};
    map.startIndex = {{hdr[2], hdr[3]}};
    map.resolution = geo[0];
    map.position = {{geo[1], geo[2]}};
    const size_t n = static_cast<size_t>(hdr[0]) * hdr[1];
    for (const char* name : {"traversability", "elevation"}) {
        grid_map::Matrix& m = map.layers[name];
        m.v.resize(n);
        in.read(reinterpret_cast<char*>(m.v.data()), static_cast<std::streamsize>(n * sizeof(float)));
    }
    std::vector<double> poses(static_cast<size_t>(hdr[5]) * 3);
    in.read(reinterpret_cast<char*>(poses.data()), static_cast<std::streamsize>(poses.size() * sizeof(double)));
    if (!in) return 3;
    const uint8_t gaitCycles = static_cast<uint8_t>(hdr[4]);

    FILE* f = std::fopen(argv[2], "w");
    if (!f) return 4;
    try {
        fpe_ros::Engine eng(0);
        if (!eng.upload(map)) {
            std::fprintf(f, "upload failed: %s\n", eng.lastError());
            return 5;
        }
        fpe_params params;
        fpe_params_yaml(&params);
        fpe_opt_params optParams;
        fpe_opt_params_yaml(&optParams);
        // the members a node keeps between service calls (never cleared by the reference: cpp:715, 743; cpp:36)
        foothold_planner_msgs::GlobalFootholds centroidMsg, optMsg;
        centroidMsg.gait_cycles = 77;
        optMsg.gait_cycles = 78;
        double lfRh[2] = {0.0, 0.0};
        for (int k = 0; k < hdr[5]; ++k) {
            const double* pose = &poses[static_cast<size_t>(k) * 3];
            foothold_planner_msgs::GlobalFootholds msg;
            std::fprintf(f, "pose %d\n", k);
            const bool ok = eng.plan(params, pose, gaitCycles, msg);
            std::fprintf(f, "plan %d\n", ok ? 1 : 0);
            if (ok) put_msg(f, "msg", msg);
            std::vector<std::array<double, 12>> rows;
            nav_msgs::Path pathN, pathC;
            std::vector<double> csN, fdN, csC, fdC;
            const bool okAll = eng.planAllTracks(params, pose, gaitCycles, msg, centroidMsg, rows, pathN, pathC, csN, fdN, csC, fdC);
            std::fprintf(f, "all %d\n", okAll ? 1 : 0);
            if (okAll) {
                put_msg(f, "msg", msg);
                put_msg(f, "centroid", centroidMsg);
                std::fprintf(f, "rows %zu", rows.size());
                for (const auto& r : rows)
                    for (double x : r) std::fprintf(f, " %a", x);
                std::fprintf(f, "\n");
                put_path(f, "pathN", pathN);
                put_path(f, "pathC", pathC);
                put_vec(f, "csN", csN);
                put_vec(f, "fdN", fdN);
                put_vec(f, "csC", csC);
                put_vec(f, "fdC", fdC);
            }
            std::vector<double> csO, fdO;
            const bool okOpt = eng.planWithOptTrack(params, optParams, lfRh, pose, gaitCycles, msg, optMsg, csO, fdO);
            std::fprintf(f, "opt %d\n", okOpt ? 1 : 0);
            if (okOpt) {
                put_msg(f, "msg", msg);
                put_msg(f, "optmsg", optMsg);
                put_vec(f, "csO", csO);
                put_vec(f, "fdO", fdO);
            }
            std::fprintf(f, "lfrh %a %a\n", lfRh[0], lfRh[1]);
        }
    } catch (const std::exception& e) {
        std::fprintf(f, "exception: %s\n", e.what());
        std::fclose(f);
        return 6;
    }
    std::fclose(f);
    return 0;
}
