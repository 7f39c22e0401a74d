// MOCK (tests/probe/ros_mock/README.md)
#pragma once
#include <foothold_planner_msgs/GlobalFootholds.h>
namespace geometry_msgs { struct Pose { Point position; }; struct PoseStamped { Pose pose; }; }
