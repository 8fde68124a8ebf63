// RosParamServer.h -- mirror of ltremovert/include/removert/RosParamServer.h:12-97 without ROS.
// The same 26 keys under the `removert/` namespace, the same defaults (RosParamServer.cpp:7-59); values come from
// the reference's own config/params_ltmapper.yaml through a small YAML-subset reader instead of the ROS param server.
// A ROS build reads them with nh.param<>() exactly as the reference and fills the same members.
#pragma once
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "removert/utility.h"

class RosParamServer
{
public:
    // loads `removert:` keys from a yaml file; missing keys keep the reference defaults
    static std::map<std::string, std::string> readYamlNamespace(const std::string& path, const std::string& ns);
    static void setParamFile(const std::string& path);       // process-wide, like the ROS parameter server
    static std::string paramFile();

    RosParamServer();

    bool isScanFileKITTIFormat_;
    float rimg_color_min_, rimg_color_max_;
    std::pair<float, float> kRangeColorAxis, kRangeColorAxisForDiff;

    float kVFOV, kHFOV;
    std::pair<float, float> kFOV;
    std::vector<float> remove_resolution_list_, revert_resolution_list_;

    int kNumKnnPointsToCompare;
    float kScanKnnAndMapKnnAvgDiffThreshold;

    std::vector<double> kVecExtrinsicLiDARtoPoseBase;
    ltremovert::Matrix4d kSE3MatExtrinsicLiDARtoPoseBase, kSE3MatExtrinsicPoseBasetoLiDAR;

    float kDownsampleVoxelSize;
    std::string central_sess_scan_dir_, central_sess_pose_path_, query_sess_scan_dir_, query_sess_pose_path_;
    int start_idx_, end_idx_;
    bool use_keyframe_gap_, use_keyframe_meter_;
    int keyframe_gap_;
    float keyframe_gap_meter_;
    int repeat_removert_iter_;
    int kNumOmpCores;
    bool kFlagSaveMapPointcloud, kFlagSaveCleanScans;
    std::string save_pcd_directory_;

    // new, optional (safe defaults = shipped behaviour)
    bool gpu_use_self_removert_;   // removert/gpu_use_self_removert: run selfRemovert() (commented out at Removerter.cpp:1582,1586)
    bool gpu_skip_hd_knn_;         // removert/gpu_skip_hd_knn: skip the visualisation-only HD kNN stage
    int gpu_device_;               // removert/gpu_device
    int gpu_lanes_;                // removert/gpu_lanes (default 2, one GPU only): the independent chains of run() side by side on the context and a lane (include/ltm.h "lanes"); 1 = one after the other
    bool gpu_async_io_;            // removert/gpu_async_io (default true): pipelined loader (decode || H2D) and background output writer; false = the synchronous path
    bool gpu_fetch_chunked_;       // removert/gpu_fetch_chunked (default true): the background writer takes its data through the library's ring of pinned chunks
                                   // (ltm_*_fetch_chunks_begin) instead of one page-locked buffer per output; false = the whole-buffer fetches
    int gpu_viz_every_;            // removert/gpu_viz_every: >0 = emit the four RViz images (Removerter.cpp:580-585) of every N-th source keyframe of each vote pass
};
