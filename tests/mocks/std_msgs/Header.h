#pragma once
#include <cstdint>
#include <string>
namespace ros { struct Time { static Time now(); double toSec() const; }; }
namespace std_msgs { struct Header { uint32_t seq = 0; ros::Time stamp; std::string frame_id; }; }
