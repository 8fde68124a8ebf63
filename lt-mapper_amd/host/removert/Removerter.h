// Removerter.h -- mirror of ltremovert::Removerter (ltremovert/include/removert/Removerter.h:67-201) over the C ABI.
// Same public method names and the same run() script (Removerter.cpp:1653-1678); the bodies are calls into
// libltm_hip.so.  The four RViz images of Removerter.cpp:580-585 come colour-mapped from the device (ltm_debug_viz_images)
// through publishDebugImages(); point-cloud publishers (Removerter.cpp:55-71) are visualisation only and absent.
#pragma once
#include <atomic>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "removert/RosParamServer.h"
#include "removert/Session.h"

namespace ltremovert
{

class Removerter : public RosParamServer
{
private:
    std::shared_ptr<Device> dev_;
    Session central_sess_;
    Session query_sess_;

    std::string updated_scans_save_dir_, updated_strong_scans_save_dir_, pd_scans_save_dir, strong_pd_scans_save_dir, strong_nd_scans_save_dir;
    std::string central_map_static_save_dir_, central_map_dynamic_save_dir_;

    float curr_res_alpha_ = 0.0f;
    std::pair<int, int> curr_rimg_shape_{0, 0};
    CloudPtr union_q_, union_c_;

    void saveMap(const std::string& file, const CloudPtr& cloud, bool octree_layout = true);
    std::pair<CloudPtr, CloudPtr> votePartition(const Session& tgt, const CloudPtr& map, const ScansPtr& scans, const Session& src, float res, int mode);
    int viz_pass_ = 0;
    // background output writer: fetch tickets in flight + the handles that must outlive them
    // the ticket is handed back (ltm_fetch_release, any thread) by the writer task that consumed it last, so that its pinned buffer
    // serves the next fetch; whatever is still there when the outputs are finished (errors, empty sets) is released then
    struct PendingFetch { std::shared_ptr<std::atomic<ltm_fetch*>> ticket; CloudPtr cloud; ScansPtr scans; };
    std::unique_ptr<AsyncWriter> writer_;
    std::vector<PendingFetch> fetches_;
    void finishOutputs();      // waits for every queued file, releases the tickets
    bool bench_mode_ = false;  // runBench(): no output files -- saveMap / saveScans return at once

    // Lanes (include/ltm.h; removert/gpu_lanes, default 2 on one GPU): makeGlobalMap + Steps 1-3 with the independent chains of the reference's run()
    // side by side -- this object drives the central-side halves on its own context, `lane_` (a second Removerter on a lane of the same GPU, run from a
    // second host thread, with its own writer) the query-side / PD / "strong" halves; clouds pass between the two without copies.  Same outputs as
    // runStagesOneLane(), byte for byte.
    std::shared_ptr<Device> lane_dev_;
    std::unique_ptr<Removerter> lane_;
    bool useLanes() const { return gpu_lanes_ >= 2 && dev_->world() == 1 && gpu_viz_every_ <= 0; }      // (the RViz images serialise a pass and number their files per object: one lane)
    void ensureLane();
    void runStages(bool write_outputs);
    void runStagesOneLane(bool write_outputs);
    void runStagesTwoLanes(bool write_outputs);

public:
    Removerter();
    // one rank of a keyframe-sharded multi-GPU run: its own device context plus its Comm endpoint (removert_main.cpp --gpus K)
    explicit Removerter(std::shared_ptr<Device> dev);
    virtual ~Removerter();
    int rank() const { return dev_->rank(); }
    // `ltm_run <yaml> --bench K [--warmup W]`: loads the two sessions once, then times makeGlobalMap + Steps 1-3 K times with the inputs resident
    // on the device and no output files (the same timed region as bench.py's, driven by THIS host); prints one JSON line with the wall time
    // per step and the per-class kernel times / launch counts of ltm_profile_read
    int runBench(int steps, int warmup);

    // pubRangeImg x4 (Removerter.cpp:580-585): called for every gpu_viz_every-th source keyframe of a vote pass with the
    // BGR8 images /scan_rimg_single, /map_rimg_single, /diff_rimg_single, /map_rimg_ptidx_single.  The default writes
    // <save_pcd_directory>viz/<pass>_<kf>_{scan,map,diff,ptidx}.ppm; the ROS wrapper overrides it with image_transport.
    virtual void publishDebugImages(int pass, size_t kf, int rows, int cols, const uint8_t* scan_bgr, const uint8_t* map_bgr,
                                    const uint8_t* diff_bgr, const uint8_t* ptidx_bgr);

    void loadSessionInfo(void);
    void parseKeyframes(void);
    void loadKeyframes(void);
    void precleaningKeyframes(float _radius);

    void makeGlobalMap(Session& _sess);
    void makeGlobalMap();

    std::pair<CloudPtr, CloudPtr> partitionCurrentMap(const Session& _target_sess, const Session& _source_sess, float _res_alpha);
    std::pair<CloudPtr, CloudPtr> partitionCurrentMapForND(const Session& _target_sess, const Session& _source_sess, float _res_alpha);
    std::pair<CloudPtr, CloudPtr> partitionCurrentMapForPD(const Session& _target_sess, const Session& _source_sess, float _res_alpha);

    void removeOnce(const Session& _target_sess, const Session& _source_sess, float _res_alpha);
    void revertOnce(const Session& _target_sess, const Session& _source_sess, float _res_alpha);
    void resetCurrrentMapAsDynamic(const Session& _sess, bool _as_dynamic);
    void resetCurrrentMapAsDynamic(const Session& _sess);
    void resetCurrrentMapAsStatic(const Session& _sess);
    void selfRemovert(const Session& _sess, int _repeat);
    void removeHighDynamicPoints(void);
    void removeHighDynamicPointsOnSessionGroups(void);   // even worlds: one session per rank group, then a swap between rank pairs (Comm.h)
    CloudPtr group_orig_noisy_;                          // set by makeGlobalMap() while this rank's session group is entered

    void filterStrongND(Session& _sess_src, Session& _sess_cleaner);
    void iremoveOnceForND(const Session& _target_sess, const Session& _source_sess, float _res_alpha);
    void filterStrongPD(Session& _sess_src, Session& _sess_cleaner);
    void removeOnceForPD(const Session& _target_sess, const Session& _source_sess, float _res_alpha);
    void detectLowDynamicPoints(void);

    void updateCurrentMap(void);
    void parseStaticScansViaProjection(Session& _sess);
    void parseStaticScansViaProjection(void);
    void parseUpdatedStaticScansViaProjection(void);
    void parseUpdatedStaticScansViaProjection(Session& _sess);
    void parseLDScansViaProjection();
    void parseLDScansViaProjection(Session& _sess);
    void updateScansScanwise(Session& _sess);
    void updateScansScanwise();

    void saveCurrentStaticAndDynamicPointCloudGlobal(const Session& _sess, std::string _postfix);
    void saveAllTypeOfScans();
    void saveUpdatedScans(Session& _sess);
    void saveLDScans(Session& _sess);
    void savePDScans(Session& _sess);
    void saveStrongPDScans(Session& _sess);
    void saveStrongNDScans(Session& _sess);
    void saveScans(Session& _sess, const ScansPtr& _scans, std::string _save_dir, bool octree_layout);

    // ---- the reference's fine-grained methods, kept as thin wrappers over the C ABI (Removerter.h:126-201).  run() does not use
    // them: the batch calls above replace their per-scan loops; they exist so that code written against the class still links.
    // A range image stands in for the cv::Mat pair of map2RangeImg (range CV_32FC1, point index CV_32SC1).
    using RangeImage = ltremovert::RangeImage;
    RangeImage scan2RangeImg(const CloudPtr& _scan, const std::pair<float, float> _fov, const std::pair<int, int> _rimg_size);   // Removerter.cpp:109-156
    std::vector<int> calcDescrepancyAndParseDynamicPointIdxForEachScan(std::pair<int, int> _rimg_shape);                           // :542-593 (central on itself)
    std::vector<int> calcDescrepancyAndParseDynamicPointIdxForEachScan(const Session& _target_sess, const Session& _source_sess, std::pair<int, int> _rimg_shape);
    std::vector<int> calcDescrepancyAndParseDynamicPointIdxForEachScanForND(const Session& _target_sess, const Session& _source_sess, std::pair<int, int> _rimg_shape);   // :485-540
    std::vector<int> calcDescrepancyAndParseDynamicPointIdxForEachScanForPD(const Session& _target_sess, const Session& _source_sess, std::pair<int, int> _rimg_shape);   // :429-482
    std::vector<int> getStaticIdxFromDynamicIdx(const std::vector<int>& _dynamic_point_indexes, int _num_all_points);              // :675-687
    void parsePointcloudSubsetUsingPtIdx(const CloudPtr& _ptcloud_orig, std::vector<int>& _point_indexes, CloudPtr& _ptcloud_to_save);   // :933-946

    // Step 0, then: the fine-grained wrappers must reproduce partitionCurrentMap() (same static / dynamic clouds, byte for byte) and
    // scan2RangeImg must agree with the range of every scan point.  Used by `ltm_run <yaml> --check-wrappers` (tests/test_gpu_cli.py).
    bool checkFineGrainedWrappers();

    void saveKeyframePoses(const Session& _sess);   // <save_pcd_directory>scans_updated_poses.txt: poses matching scans_updated/ (cascade hand-over)
    void run(void);
};

} // namespace ltremovert
