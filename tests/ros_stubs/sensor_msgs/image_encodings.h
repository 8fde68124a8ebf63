// TEST INFRASTRUCTURE: see ros/ros.h in this directory.
#pragma once
#include <string>
namespace sensor_msgs { namespace image_encodings { const std::string BGR8 = "bgr8"; } }
