#pragma once
#include <opencv2/opencv.hpp>
#include <sensor_msgs/CompressedImage.h>
namespace cv_bridge {
class CvImage {
 public:
  std_msgs::Header header; std::string encoding; cv::Mat image;
  CvImage(const std_msgs::Header& h, const std::string& enc, const cv::Mat& img);
  sensor_msgs::ImagePtr toImageMsg() const;
};
}
