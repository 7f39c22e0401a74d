// fpe_ros_adapter.hpp — header-only glue between the reference's ROS node and libfpe.so.
//
// Not linked in this repository (neither this image nor the GPU box has ROS1 / grid_map): compile
// it inside the reference's catkin package with -DFPE_WITH_ROS and link -lfpe.  tests/test_cpu_abi_and_host.py
// parses it (-fsyntax-only) against a minimal mock of the few ROS / grid_map types it touches (tests/probe/ros_mock).
// Thread safety: the service handler may run concurrently with itself under AsyncSpinner(0)
// (foothold_planner_node.cpp:12), so every response / report buffer lives on the CALLER's stack (heap for the
// 33 KB message structs) — the adapter has no mutable members besides the engine handle.  It keeps the
// service name, message types and subscriber of the reference untouched (cpp:188, cpp:219, cpp:237)
// and only replaces (a) the body of gridmapCallback's map copy and (b) the per-cycle loop of
// globalFootholdPlan for the nominal response.  See INTEGRATION.md for the three call sites.
#pragma once
#ifdef FPE_WITH_ROS

#include <foothold_planner_msgs/GlobalFootholds.h>
#include <geometry_msgs/PoseStamped.h>
#include <grid_map_core/GridMap.hpp>
#include <nav_msgs/Path.h>

#include <algorithm>
#include <array>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "fpe.h"

namespace fpe_ros {

class Engine {
public:
    explicit Engine(int device = 0) {
        // the structs passed by pointer have no size field: a library of another layout version must not be driven with them
        if (fpe_abi_version() != FPE_ABI_VERSION)
            throw std::runtime_error("libfpe.so has struct layout version " + std::to_string(fpe_abi_version()) + ", this adapter was compiled against " +
                                     std::to_string(FPE_ABI_VERSION));
        if (fpe_create(device, &h_) != FPE_OK) throw std::runtime_error(std::string("fpe_create: ") + fpe_last_error(nullptr));
    }
    ~Engine() { fpe_destroy(h_); }
    Engine(const Engine&) = delete;
    Engine& operator=(const Engine&) = delete;

    // gridmapCallback (cpp:504-536): hand the two layers to the GPU once per message.  grid_map
    // stores Eigen::MatrixXf (column-major) with a circular-buffer start index; the engine
    // canonicalises on the device, so no host-side copy of the layers is made here.
    bool upload(const grid_map::GridMap& map) {
        if (!map.exists("traversability") || !map.exists("elevation")) return false;
        fpe_map_desc d;
        d.rows = map.getSize()(0);
        d.cols = map.getSize()(1);
        d.resolution = map.getResolution();
        d.position[0] = map.getPosition().x();
        d.position[1] = map.getPosition().y();
        d.start_index[0] = map.getStartIndex()(0);
        d.start_index[1] = map.getStartIndex()(1);
        d.storage_order = 0;  // Eigen column-major
        return fpe_upload_map(h_, &d, map["traversability"].data(), map["elevation"].data()) == FPE_OK;
    }

    // globalFootholdPlan (cpp:539-1602): nominal response for one initial pose.  `params` carries
    // the node's ROS parameters with their member types (readParameters, cpp:248-314).
    bool plan(const fpe_params& params, const double initialPose[3], uint8_t gaitCycles,
              foothold_planner_msgs::GlobalFootholds& msg) {
        std::unique_ptr<fpe_global_footholds> resp(new fpe_global_footholds);  // per call: handlers may run concurrently
        // FPE_E_SERVICE_FALSE included: the reference's handler returns false there too (cpp:566 / cpp:931-934)
        if (fpe_plan_service(h_, &params, initialPose, gaitCycles, resp.get()) != FPE_OK) return false;
        if (gateFailed()) return false;
        fill(*resp, msg, true);  // cpp:591 clears the nominal message
        return true;
    }

    // Same call with every product the node publishes or logs for the nominal and centroid tracks
    // (SURVEY.md §8(f) N2).  `centroidMsg` is APPENDED to, as the reference never clears it between
    // calls (cpp:715); the paths are cleared first (cpp:635-636); the KPI vectors are the members of
    // footholdsKPI_ (hpp:732-745; cleared at cpp:606-611).
    bool planAllTracks(const fpe_params& params, const double initialPose[3], uint8_t gaitCycles,
                       foothold_planner_msgs::GlobalFootholds& msg, foothold_planner_msgs::GlobalFootholds& centroidMsg,
                       std::vector<std::array<double, 12>>& defaultFootholds, nav_msgs::Path& nominalFeetCenterPath,
                       nav_msgs::Path& centroidFeetCenterPath, std::vector<double>& cogSpeedNominal,
                       std::vector<double>& feetDistanceNominal, std::vector<double>& cogSpeedCentroid,
                       std::vector<double>& feetDistanceCentroid) {
        std::vector<double> rows(static_cast<size_t>(1 + gaitCycles) * 12);
        int32_t nRows = 0;
        // per-call buffers (no shared members): two service calls may be in flight at once
        std::unique_ptr<fpe_global_footholds> resp(new fpe_global_footholds), cen(new fpe_global_footholds);
        std::unique_ptr<fpe_track_report> repNominal(new fpe_track_report), repCentroid(new fpe_track_report);
        const fpe_track_report& repNominal_ = *repNominal;
        const fpe_track_report& repCentroid_ = *repCentroid;
        if (fpe_plan_service_report(h_, &params, initialPose, gaitCycles, resp.get(), cen.get(), rows.data(), &nRows,
                                    repNominal.get(), repCentroid.get()) != FPE_OK)
            return false;
        if (gateFailed()) return false;
        fill(*resp, msg, true);
        // centroidGlobalFootholdsMsg_ is never cleared between calls (cpp:715): append; its gait_cycles field is
        // never written by the reference, so the caller's value is kept
        const uint8_t keepGaitCycles = centroidMsg.gait_cycles;
        fill(*cen, centroidMsg, false);
        centroidMsg.gait_cycles = keepGaitCycles;
        defaultFootholds.clear();  // cpp:601
        for (int r = 0; r < nRows; ++r) {
            std::array<double, 12> row;
            std::copy(rows.begin() + r * 12, rows.begin() + (r + 1) * 12, row.begin());
            defaultFootholds.push_back(row);
        }
        fillPath(repNominal_, nominalFeetCenterPath);
        fillPath(repCentroid_, centroidFeetCenterPath);
        cogSpeedNominal.assign(repNominal_.cog_speed, repNominal_.cog_speed + repNominal_.n_kpi);
        feetDistanceNominal.assign(repNominal_.feet_distance, repNominal_.feet_distance + repNominal_.n_kpi);
        cogSpeedCentroid.assign(repCentroid_.cog_speed, repCentroid_.cog_speed + repCentroid_.n_kpi);
        feetDistanceCentroid.assign(repCentroid_.feet_distance, repCentroid_.feet_distance + repCentroid_.n_kpi);
        return true;
    }

    // The same call with the opt track's products as well: global_footholds_opt (cpp:221, 1510-1532;
    // like the centroid message never cleared between calls: appended) and footholdsKPI_.feetDistance_opt / cogSpeed_opt
    // (cpp:1488-1499).  `optParams` are the node's nlopt/* parameters (cpp:297-307); lfCurrentRow / rhCurrentRow are
    // file-scope globals in the reference (cpp:36) that survive from one service call to the next: the CALLER keeps them
    // (`lfRhCurrentRow`, zero at node start) and this function hands back the values the call left.  The optimiser behind
    // it is build-defined (include/fpe.h): the node's NLopt dependency is not needed for this call.
    bool planWithOptTrack(const fpe_params& params, fpe_opt_params optParams, double lfRhCurrentRow[2], const double initialPose[3],
                          uint8_t gaitCycles, foothold_planner_msgs::GlobalFootholds& msg,
                          foothold_planner_msgs::GlobalFootholds& optMsg, std::vector<double>& cogSpeedOpt,
                          std::vector<double>& feetDistanceOpt) {
        optParams.lf_current_row0 = lfRhCurrentRow[0];
        optParams.rh_current_row0 = lfRhCurrentRow[1];
        std::unique_ptr<fpe_global_footholds> resp(new fpe_global_footholds), opt(new fpe_global_footholds);
        std::unique_ptr<fpe_track_report> repOpt(new fpe_track_report);
        std::vector<fpe_opt_cycle> cycles(std::max<size_t>(gaitCycles, 1));
        if (fpe_plan_service_opt(h_, &params, &optParams, initialPose, gaitCycles, resp.get(), nullptr, nullptr, nullptr, nullptr, nullptr,
                                 opt.get(), repOpt.get(), cycles.data()) != FPE_OK)
            return false;  // FPE_E_SERVICE_FALSE: getGaitCycleSearchGridMap failed (cpp:931-934; kinds: fpe_service_gate)
        if (gateFailed()) return false;
        fill(*resp, msg, true);
        const uint8_t keepGaitCycles = optMsg.gait_cycles;  // never written by the reference (cpp:743)
        fill(*opt, optMsg, false);
        optMsg.gait_cycles = keepGaitCycles;
        const fpe_track_report& rep = *repOpt;
        cogSpeedOpt.assign(rep.cog_speed, rep.cog_speed + rep.n_kpi);
        feetDistanceOpt.assign(rep.feet_distance, rep.feet_distance + rep.n_kpi);
        // cpp:1561-1568: lfCurrentRow / rhCurrentRow as the engine's chain left them — gaitMap_.getIndex of the committed LF /
        // RH POSITIONS (not the optimiser's x: the two differ when getPosition failed for a leg and the previous Position was
        // kept, cpp:1284-1312), unchanged by cycles that did not commit
        fpe_service_gate gate;
        if (fpe_last_service_gate(h_, &gate) == FPE_OK && gate.chain_ran) {
            lfRhCurrentRow[0] = gate.lf_current_row;
            lfRhCurrentRow[1] = gate.rh_current_row;
        }
        return true;
    }

    // The handler's gate failed in a way the call did not refuse by itself: under fpe_set_tuning("service_opt_gate", 0 | 1) the
    // engine answers FPE_OK although the opt track's chain stopped at its gate (kind FPE_GATE_BUILD_DEFINED, opt products
    // empty).  The reference's handler returns false at that gate (cpp:931-934): so does this adapter, whatever the mode.
    bool gateFailed() const {
        fpe_service_gate gate;
        return fpe_last_service_gate(h_, &gate) == FPE_OK && gate.fail_kind != FPE_GATE_NONE;
    }

    const char* lastError() const { return fpe_last_error(h_); }
    fpe_handle handle() const { return h_; }

private:
    static void fill(const fpe_global_footholds& r, foothold_planner_msgs::GlobalFootholds& msg, bool clear) {
        msg.success = r.success;
        msg.gait_cycles = r.gait_cycles;
        msg.gait_cycles_succeed = r.gait_cycles_succeed;
        if (clear) msg.footholds.clear();
        for (int k = 0; k < r.n_footholds; ++k) {
            foothold_planner_msgs::Foothold f;
            f.point.x = r.footholds[k].x;
            f.point.y = r.footholds[k].y;
            f.point.z = r.footholds[k].z;
            f.foot_id = r.footholds[k].foot_id;
            f.gait_cycle_id = r.footholds[k].gait_cycle_id;
            msg.footholds.push_back(f);
        }
    }
    static void fillPath(const fpe_track_report& rep, nav_msgs::Path& path) {
        path.poses.clear();
        for (int k = 0; k < rep.n_path; ++k) {
            geometry_msgs::PoseStamped p;  // cpp:2194-2196: only pose.position is set
            p.pose.position.x = rep.feet_center_path[k][0];
            p.pose.position.y = rep.feet_center_path[k][1];
            p.pose.position.z = rep.feet_center_path[k][2];
            path.poses.push_back(p);
        }
    }
    fpe_handle h_ = nullptr;  // the only member: fpe_* entry points are thread-safe per include/fpe.h
};

}  // namespace fpe_ros

#endif  // FPE_WITH_ROS
