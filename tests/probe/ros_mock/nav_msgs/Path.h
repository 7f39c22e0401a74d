// MOCK (tests/probe/ros_mock/README.md)
#pragma once
#include <geometry_msgs/PoseStamped.h>
#include <vector>
namespace nav_msgs { struct Path { std::vector<geometry_msgs::PoseStamped> poses; }; }
