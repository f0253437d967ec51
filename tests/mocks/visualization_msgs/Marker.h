#pragma once
#include <ros/ros.h>
namespace visualization_msgs { struct Marker { std_msgs::Header header; std::string ns; int32_t id, type, action; }; }
