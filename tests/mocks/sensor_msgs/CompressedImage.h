#pragma once
#include <ros/ros.h>
namespace sensor_msgs {
struct CompressedImage { std_msgs::Header header; std::string format; std::vector<uint8_t> data; };
typedef boost::shared_ptr<CompressedImage const> CompressedImageConstPtr;
struct Image { std_msgs::Header header; uint32_t height, width; std::string encoding; uint8_t is_bigendian; uint32_t step; std::vector<uint8_t> data; };
typedef boost::shared_ptr<Image> ImagePtr;
}
