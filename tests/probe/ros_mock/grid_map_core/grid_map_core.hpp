// MOCK (tests/probe/ros_mock/README.md): upstream's umbrella header, for the parse check of tests/probe/upstream_check.cpp
#pragma once
#include "GridMap.hpp"
