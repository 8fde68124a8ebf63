// utility.h -- host-side mirror of ltremovert/include/removert/utility.h for the MI355X build.
//
// Same role and names as the reference header, minus ROS/PCL/OpenCV: point type, PCD + pose-file I/O, and the
// small helpers the Session/Removerter mirrors need.  All point arithmetic of the hot path lives behind the C ABI
// (include/ltm.h); nothing in this directory projects, votes or searches neighbours on the CPU.
#pragma once

#include <cstddef>
#include <cstdint>
#include <functional>
#include <fstream>
#include <string>
#include <utility>
#include <vector>

#include "ltm.h"

namespace ltremovert
{

// pcl::PointXYZI as stored on disk and on the device: 16 bytes (utility.h:90)
struct PointType { float x, y, z, intensity; };
using Cloud = std::vector<PointType>;

const float kFlagNoPOINT = 10000.0f;       // utility.h:93
const float kValidDiffUpperBound = 200.0f; // utility.h:94

using Matrix4d = std::vector<double>;      // 16 doubles, row-major (pose text layout, Session.cpp:102-114)

// utility.cpp:28-36
std::vector<double> splitPoseLine(const std::string& _str_line, char _delimiter);
// utility.cpp:222-236
std::pair<int, int> resetRimgSize(const std::pair<float, float> _fov, const float _resize_ratio);

// general 4x4 inverse in double (stands in for Eigen::Matrix4d::inverse(), Session.cpp:110, RosParamServer.cpp:30)
bool inverse4x4(const double* m, double* inv);

// ---- PCD v0.7 (pcl::io::loadPCDFile / savePCDFileBinary, Session.cpp:275, Removerter.cpp:232..1645)
// reads ascii / binary / binary_compressed files with float x y z [intensity] fields (other fields are skipped)
bool loadPCDFile(const std::string& path, Cloud& out, std::string* err = nullptr);
// the LZF stream of a binary_compressed payload into exactly out_len bytes; false on a malformed stream (never reads or writes out of bounds)
bool lzf_decompress(const unsigned char* in, size_t in_len, unsigned char* out, size_t out_len);
// byte-compatible with pcl::io::savePCDFileBinary for PointXYZI.  `octree_layout`: the reference's clouds that come
// out of octreeDownsampling carry width=1,height=n (utility.cpp:217-218); everything else is width=n,height=1.
bool savePCDFileBinary(const std::string& path, const Cloud& cloud, bool octree_layout, std::string* err = nullptr);
bool savePCDFileBinary(const std::string& path, const PointType* pts, size_t n, bool octree_layout, std::string* err = nullptr);
// the same file written piecewise: opens `path`, writes the header for `n` points and leaves `f` positioned at the first point
bool openPCDFileBinary(const std::string& path, size_t n, bool octree_layout, std::ofstream* f, std::string* err = nullptr);
// POINTS of a PCD header without touching the payload (the pipelined loader sizes its device array from these)
bool readPCDPointCount(const std::string& path, size_t* n_points, std::string* err = nullptr);

// pcl::VoxelGrid as used by Session::loadKeyframes (Session.cpp:284-289), including the int32 overflow early-out
// (output = input) that PCL takes for large extents.  Points inside a voxel are summed in input order.
void voxelGridFilter(const Cloud& in, float leaf, Cloud& out);
void voxelGridFilter(Cloud&& in, float leaf, Cloud& out);      // the early-out hands the input over instead of copying it


// runs f(i) for i in [0, n) on up to `threads` host threads (0 = hardware concurrency); exceptions are re-thrown on the caller
void parallelFor(size_t n, const std::function<void(size_t)>& f, unsigned threads = 0);

void fsmkdir(const std::string& _path);    // Removerter.cpp:6-10
std::vector<std::string> listDirectorySorted(const std::string& dir, std::vector<std::string>* names);

// Background writer of the output protocol (Removerter.cpp:1637-1650, 1446-1520): a pool of host threads that
// wait for an asynchronous device->host fetch (ltm_fetch_wait) and write the PCD files while the GPU keeps computing.
class AsyncWriter
{
public:
    explicit AsyncWriter(unsigned threads);
    ~AsyncWriter();
    void submit(std::function<void()> task);
    void drain();                     // returns when every submitted task has run; re-throws the first task error
private:
    struct Impl;
    Impl* impl_;
};

// per-thread switch: ranks other than 0 of a multi-GPU run keep quiet (their logs would only repeat rank 0's)
bool& logQuiet();

// throws std::runtime_error carrying ltm_last_error() when rc != LTM_OK
void ltmCheck(ltm_ctx* ctx, int rc, const char* what);

} // namespace ltremovert
