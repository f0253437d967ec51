#pragma once
#include <ros/ros.h>
#include <sensor_msgs/CompressedImage.h>
namespace image_transport {
class Publisher { public: void publish(const sensor_msgs::ImagePtr& msg) const; };
class ImageTransport { public: explicit ImageTransport(const ros::NodeHandle& nh); Publisher advertise(const std::string& topic, uint32_t queue_size, bool latch = false); };
}
