#pragma once
#include <ros/ros.h>
namespace sensor_msgs {
struct LaserScan {
  std_msgs::Header header;
  float angle_min, angle_max, angle_increment, time_increment, scan_time, range_min, range_max;
  std::vector<float> ranges, intensities;
};
}
