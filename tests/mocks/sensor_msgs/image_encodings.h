#pragma once
#include <string>
namespace sensor_msgs { namespace image_encodings { extern const std::string MONO8; extern const std::string BGR8; } }
