// TEST INFRASTRUCTURE: see ros/ros.h in this directory.  Field names and types of sensor_msgs/Image.msg + std_msgs/Header.msg.
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include <ros/ros.h>
namespace std_msgs { struct Header { uint32_t seq = 0; ros::Time stamp; std::string frame_id; }; }
namespace sensor_msgs
{
struct Image { std_msgs::Header header; uint32_t height = 0, width = 0; std::string encoding; uint8_t is_bigendian = 0; uint32_t step = 0; std::vector<uint8_t> data; };
}
