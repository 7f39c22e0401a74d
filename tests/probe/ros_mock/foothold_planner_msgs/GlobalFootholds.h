// MOCK (tests/probe/ros_mock/README.md): field names of foothold_planner_msgs/{Foothold,GlobalFootholds}.msg
#pragma once
#include <cstdint>
#include <vector>
namespace geometry_msgs { struct Point { double x = 0, y = 0, z = 0; }; }
namespace foothold_planner_msgs {
struct Foothold { geometry_msgs::Point point; uint8_t foot_id = 0; uint8_t gait_cycle_id = 0; };
struct GlobalFootholds { bool success = false; uint8_t gait_cycles = 0; uint8_t gait_cycles_succeed = 0; std::vector<Foothold> footholds; };
}
