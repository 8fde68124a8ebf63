// TEST INFRASTRUCTURE: see ros/ros.h in this directory.  image_transport::ImageTransport / Publisher as the wrapper uses them.
#pragma once
#include <cstdint>
#include <string>
#include <ros/ros.h>
#include <sensor_msgs/Image.h>
namespace image_transport
{
class Publisher
{
public:
    uint32_t getNumSubscribers() const { return 0; }
    void publish(const sensor_msgs::Image&) const {}
};
class ImageTransport
{
public:
    explicit ImageTransport(const ros::NodeHandle&) {}
    Publisher advertise(const std::string&, uint32_t, bool = false) { return Publisher(); }
};
} // namespace image_transport
