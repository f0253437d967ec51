#pragma once
#include <ros/ros.h>
#include <geometry_msgs/Point32.h>
namespace sensor_msgs {
struct ChannelFloat32 { std::string name; std::vector<float> values; };
struct PointCloud { std_msgs::Header header; std::vector<geometry_msgs::Point32> points; std::vector<ChannelFloat32> channels; };
}
