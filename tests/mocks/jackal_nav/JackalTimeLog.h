// generated-message stand-in: fields of the reference's msg/JackalTimeLog.msg
#pragma once
#include <ros/ros.h>
namespace jackal_nav { struct JackalTimeLog { std_msgs::Header header; float dmap_time = 0, pcl_time = 0, obstacle_scan_time = 0, safe_path_time = 0; }; }
