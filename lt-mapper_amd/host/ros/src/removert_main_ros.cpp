// removert_main_ros.cpp -- ROS1 entry point of the MI355X build: same node name, parameter namespace and launch file as
// ltremovert/src/removert_main.cpp:3-12, so `roslaunch removert run_ltmapper.launch` keeps working.
//
// Not built for real in this repository (the build container has no ROS); tests/test_host_cpp.py compiles and links it against minimal
// stand-ins for the roscpp / image_transport / sensor_msgs declarations it uses (tests/ros_stubs).  It is deliberately thin: it copies the 26
// `removert/` parameters from the ROS parameter server into the YAML subset the ROS-free RosParamServer mirror reads,
// then runs the same Removerter::run() as `ltm_run`.
#include <image_transport/image_transport.h>
#include <ros/ros.h>
#include <sensor_msgs/Image.h>
#include <sensor_msgs/image_encodings.h>

#include <unistd.h>

#include <cstdio>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "removert/Removerter.h"

namespace {
template <class T> void put(ros::NodeHandle& nh, std::ostream& o, const char* key)
{
    T v;
    if (nh.getParam(std::string("removert/") + key, v)) o << "  " << key << ": " << v << "\n";
}
template <> void put<std::string>(ros::NodeHandle& nh, std::ostream& o, const char* key)
{
    std::string v;
    if (nh.getParam(std::string("removert/") + key, v)) o << "  " << key << ": \"" << v << "\"\n";
}
template <class T> void put_list(ros::NodeHandle& nh, std::ostream& o, const char* key)
{
    std::vector<T> v;
    if (!nh.getParam(std::string("removert/") + key, v)) return;
    o << "  " << key << ": [";
    o.precision(17);
    for (size_t i = 0; i < v.size(); ++i) o << (i ? ", " : "") << v[i];
    o << "]\n";
}

// The four image topics of Removerter.cpp:55-71, fed with the device-rendered BGR8 images (ltm_debug_viz_images) instead of
// cv_bridge + cv::applyColorMap.  Images are rendered only while removert/gpu_viz_every > 0.
class RosRemoverter : public ltremovert::Removerter
{
    image_transport::ImageTransport it_;
    image_transport::Publisher pub_[4];

public:
    explicit RosRemoverter(ros::NodeHandle& nh) : it_(nh)
    {
        const char* topic[4] = {"/scan_rimg_single", "/map_rimg_single", "/diff_rimg_single", "/map_rimg_ptidx_single"};
        for (int i = 0; i < 4; ++i) pub_[i] = it_.advertise(topic[i], 10);
    }
    void publishDebugImages(int, size_t, int rows, int cols, const uint8_t* scan_bgr, const uint8_t* map_bgr, const uint8_t* diff_bgr,
                            const uint8_t* ptidx_bgr) override
    {
        const uint8_t* src[4] = {scan_bgr, map_bgr, diff_bgr, ptidx_bgr};
        for (int i = 0; i < 4; ++i) {
            if (pub_[i].getNumSubscribers() == 0) continue;
            sensor_msgs::Image msg;
            msg.header.stamp = ros::Time::now();
            msg.header.frame_id = "removert";
            msg.height = rows; msg.width = cols; msg.encoding = sensor_msgs::image_encodings::BGR8; msg.step = cols * 3;
            msg.data.assign(src[i], src[i] + (size_t)rows * cols * 3);
            pub_[i].publish(msg);
        }
        ros::spinOnce();
    }
};
} // namespace

int main(int argc, char** argv)
{
    ros::init(argc, argv, "removert");
    ros::NodeHandle nh;
    ROS_INFO("\033[1;32m----> Removert Main Started (MI355X build).\033[0m");

    const std::string tmp = "/tmp/removert_params_" + std::to_string(::getpid()) + ".yaml";
    {
        std::ofstream o(tmp);
        o << "removert:\n";
        for (const char* k : {"isScanFileKITTIFormat", "use_keyframe_gap", "use_keyframe_meter", "saveMapPCD", "saveCleanScansPCD", "gpu_use_self_removert", "gpu_skip_hd_knn"})
            put<bool>(nh, o, k);
        for (const char* k : {"num_nn_points_within", "start_idx", "end_idx", "keyframe_gap", "repeat_removert_iter", "num_omp_cores", "gpu_device", "gpu_viz_every"})
            put<int>(nh, o, k);
        for (const char* k : {"rimg_color_min", "rimg_color_max", "sequence_vfov", "sequence_hfov", "dist_nn_points_within", "downsample_voxel_size", "keyframe_meter"})
            put<double>(nh, o, k);
        for (const char* k : {"central_sess_scan_dir", "central_sess_pose_path", "query_sess_scan_dir", "query_sess_pose_path", "save_pcd_directory"})
            put<std::string>(nh, o, k);
        put_list<double>(nh, o, "remove_resolution_list");
        put_list<double>(nh, o, "revert_resolution_list");
        put_list<double>(nh, o, "ExtrinsicLiDARtoPoseBase");
    }
    try {
        RosParamServer::setParamFile(tmp);
        RosRemoverter RMV(nh);
        RMV.run();
    } catch (const std::exception& e) {
        ROS_FATAL("removert: %s", e.what());
        std::remove(tmp.c_str());
        return 1;
    }
    std::remove(tmp.c_str());
    ros::spin();   // the reference node stays alive after run() (removert_main.cpp:11)
    return 0;
}
