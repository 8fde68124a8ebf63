// TEST INFRASTRUCTURE: the smallest stand-in for the roscpp API that lt-mapper_amd/host/ros/src/removert_main_ros.cpp uses, so that the catkin
// wrapper -- which cannot be built in an image without ROS -- is at least compiled and linked against the host mirror on every CPU test run
// (tests/test_host_cpp.py).  Signatures follow roscpp (ros/node_handle.h, ros/time.h, ros/console.h); nothing here is shipped.
#pragma once
#include <cstdio>
#include <string>
#include <vector>
namespace ros
{
struct Time { double t = 0; static Time now() { return Time(); } };
inline void init(int&, char**, const std::string&) {}
inline void spin() {}
inline void spinOnce() {}
class NodeHandle
{
public:
    bool getParam(const std::string&, bool&) const { return false; }
    bool getParam(const std::string&, int&) const { return false; }
    bool getParam(const std::string&, double&) const { return false; }
    bool getParam(const std::string&, std::string&) const { return false; }
    bool getParam(const std::string&, std::vector<double>&) const { return false; }
};
} // namespace ros
#define ROS_INFO(...) do { std::printf(__VA_ARGS__); std::printf("\n"); } while (0)
#define ROS_FATAL(...) do { std::fprintf(stderr, __VA_ARGS__); std::fprintf(stderr, "\n"); } while (0)
