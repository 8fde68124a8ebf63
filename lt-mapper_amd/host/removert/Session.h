// Session.h -- mirror of ltremovert::Session (ltremovert/include/removert/Session.h:9-136) over the C ABI.
// Member and method names follow the reference; clouds are device handles (ltm_cloud / ltm_scanset) instead of
// pcl::PointCloud::Ptr, and the kd-tree / ICP members are gone (the kNN stage is one C-ABI call).
#pragma once
#include <memory>
#include <set>
#include <string>
#include <utility>
#include <vector>

#include "removert/Comm.h"
#include "removert/RosParamServer.h"
#include "removert/utility.h"

namespace ltremovert
{

// one context per process, shared by both sessions and the Removerter (the reference shares nothing but ROS)
struct Device
{
    ltm_ctx* ctx = nullptr;
    std::shared_ptr<Comm> comm;             // null = single GPU; else this rank's endpoint of the keyframe-sharded run (Comm.h)
    explicit Device(const RosParamServer& p, int device_ordinal = -1, std::shared_ptr<Comm> comm_ = nullptr);
    struct LaneOf { Device& parent; };
    explicit Device(LaneOf l);                // a lane of `parent` (ltm_lane_create): same GPU, own stream and pool, for a second host thread
    ~Device();
    int rank() const { return comm ? comm->rank() : 0; }
    int world() const { return comm ? comm->world() : 1; }
    Device(const Device&) = delete;
    Device& operator=(const Device&) = delete;
};

// RAII handles ("boost::shared_ptr<cloud>" of the reference)
struct CloudH
{
    ltm_ctx* ctx = nullptr; ltm_cloud h = 0;
    std::shared_ptr<void> lender;           // a borrowed view (lendCloud) keeps the owner's handle alive
    CloudH() = default; CloudH(ltm_ctx* c, ltm_cloud v) : ctx(c), h(v) {}
    ~CloudH(); size_t size() const; Cloud download() const;
};
// A scan set is either whole (keyframes 0..n) or, in a multi-GPU run, this rank's SHARD: keyframes [kb, kb + numKeyframes()) of
// n_total.  Per-keyframe results stay sharded from stage to stage; Session::gatherScans() assembles the whole set when a stage
// needs every keyframe (merging scans into a global map).
struct ScansH
{
    ltm_ctx* ctx = nullptr; ltm_scanset h = 0;
    bool shard = false; size_t kb = 0, n_total = 0;
    std::shared_ptr<void> lender;           // a borrowed view (lendScans) keeps the owner's handle alive
    ScansH() = default;
    ScansH(ltm_ctx* c, ltm_scanset v) : ctx(c), h(v) {}
    ~ScansH();
    size_t numKeyframes() const;
    std::vector<Cloud> download() const;
};
using CloudPtr = std::shared_ptr<CloudH>;
using ScansPtr = std::shared_ptr<ScansH>;
// lanes (include/ltm.h): a cloud / scan set of one context made usable in another context of the same GPU without a copy.
// lend*: a view; the owner keeps the memory (and is kept alive by the view).  give*: the memory moves, `c` is left empty.  Null in, null out.
CloudPtr lendCloud(const CloudPtr& c, Device& to);
CloudPtr giveCloud(CloudPtr& c, Device& to);
ScansPtr lendScans(const ScansPtr& s, Device& to);
ScansPtr giveScans(ScansPtr& s, Device& to);

// ---- free functions of ltremovert/include/removert/utility.h:128-167 on device clouds (thin wrappers over the C ABI; the batch
// stages of Session / Removerter do not go through them).  Matrices are row-major 4x4 doubles (Matrix4d).
struct RangeImage { int rows = 0, cols = 0; std::vector<float> range; std::vector<int32_t> ptidx; };   // the cv::Mat pair of map2RangeImg
RangeImage map2RangeImg(const CloudPtr& _scan, const std::pair<float, float> _fov, const std::pair<int, int> _rimg_size);        // utility.cpp:92-142
CloudPtr parseProjectedPoints(const CloudPtr& _scan, const std::pair<float, float> _fov, const std::pair<int, int> _rimg_size); // :74-89
void transformGlobalMapToLocal(const CloudPtr& _map_global, const Matrix4d& _base_pose_inverse, const Matrix4d& _base2lidar, CloudPtr& _map_local);   // :64-72
CloudPtr local2global(const CloudPtr& _scan_local, const Matrix4d& _scan_pose, const Matrix4d& _base2lidar);                      // :160-168 (argument name as in the reference, quirk Q7)
CloudPtr global2local(const CloudPtr& _scan_global, const Matrix4d& _scan_pose_inverse, const Matrix4d& _base2lidar);             // :194-202
CloudPtr mergeScansWithinGlobalCoordUtil(const std::vector<CloudPtr>& _scans, const std::vector<Matrix4d>& _scans_poses, const Matrix4d& _lidar2base);   // :170-192
void octreeDownsampling(const CloudPtr& _src, CloudPtr& _to_save, const float _kDownsampleVoxelSize = 0.05f);                      // :204-219
std::set<int> convertIntVecToSet(const std::vector<int>& v);                                                                       // :238-245
template <typename T> std::vector<T> linspace(T a, T b, size_t N)                                                                  // utility.h:158-167, quirk Q5 kept
{
    std::vector<T> xs(N);
    if (N == 0) return xs;
    const T h = (b - a) / static_cast<T>(N - 1);      // integer T: truncating division; N == 1 divides by zero as in the reference
    T val = a;
    for (size_t i = 0; i < N; ++i, val += h) xs[i] = val;
    return xs;
}

class Session : public RosParamServer
{
public:
    const float kReprojectionAlpha = 3.0;   // Session.h:13
    // multi-GPU: clouds below this many points (default 2^24) are voxelised on every rank instead of sharded + all-gathered: the
    // exchange ships the whole output to every rank, which costs more than the sort it saves unless the input is several times
    // larger than the output (makeGlobalMap) (env LTM_VOXEL_SHARD_MIN)
    static size_t kVoxelShardMin;

    explicit Session(std::shared_ptr<Device> dev);

    std::shared_ptr<Device> dev_;
    // The communicator this session's per-keyframe stages shard over: the world's, or -- between enterSessionGroup() and leaveSessionGroup(),
    // i.e. during makeGlobalMap + Step 1 of an even world -- this rank's session group (Comm.h).  kf_begin_ / kf_end_ / poses_local_h_ follow it.
    std::shared_ptr<Comm> group_comm_;
    Comm* comm() const { return group_comm_ ? group_comm_.get() : dev_->comm.get(); }
    int rank() const { return comm() ? comm()->rank() : 0; }
    int world() const { return comm() ? comm()->world() : 1; }
    void enterSessionGroup(std::shared_ptr<Comm> group);
    void leaveSessionGroup();
    std::string sess_type_;
    float kDownsampleVoxelSize;
    std::string scan_dir_, pose_path_;

    std::vector<std::string> scan_names_, scan_paths_;
    std::vector<Matrix4d> scan_poses_, scan_inverse_poses_;
    int num_scans_ = 0;

    int keyframe_gap_ = 1;
    std::vector<std::string> keyframe_names_, keyframe_paths_;
    std::vector<Matrix4d> keyframe_poses_, keyframe_inverse_poses_;

    ltm_poses poses_h_ = 0;                 // keyframe_poses_ + keyframe_inverse_poses_ on the device
    ltm_poses poses_local_h_ = 0;           // multi-GPU: the poses of this rank's keyframe block [kf_begin_, kf_end_)
    size_t kf_begin_ = 0, kf_end_ = 0;      // this rank's block of keyframes (all of them on a single GPU)
    ScansPtr keyframe_scans_, keyframe_scans_static_projected_, keyframe_scans_dynamic_;
    ScansPtr scans_knn_coexist_, scans_knn_diff_;
    ScansPtr keyframe_scans_updated_, keyframe_scans_updated_strong_, keyframe_scans_pd_, keyframe_scans_strong_pd_,
        keyframe_scans_strong_nd_, keyframe_scans_weak_nd_;

    // `mutable`: the reference changes these maps through `const Session&` parameters (Removerter.h:81,103-111,175,181) -- legal there because
    // the members are boost::shared_ptr and only the pointees are written (`*a = *b`); here `*a = *b` is a re-bound device handle, so the
    // members themselves must be writable through a const reference for the signatures to stay verbatim
    mutable CloudPtr map_global_orig_, map_global_curr_, map_global_curr_static_, map_global_curr_dynamic_;
    mutable CloudPtr map_global_updated_, map_global_updated_strong_;
    mutable CloudPtr map_global_nd_, map_global_nd_strong_, map_global_nd_weak_;
    mutable CloudPtr map_global_pd_, map_global_pd_orig_, map_global_pd_strong_, map_global_pd_weak_;

    void loadSessionInfo(std::string _sess_type, std::string _scan_dir, std::string _pose_path);   // Session.cpp:80-118
    void setDownsampleSize(float _voxel_size);
    void clearKeyframes(void);
    void parseKeyframes(int _gap = 1);                                                             // :176-183
    void parseKeyframes(std::pair<int, int> _range, int _gap = 1);                                 // :138-173
    void parseKeyframesInROI(const std::vector<Matrix4d>& _roi_poses, int _gap = 1);               // :230-263
    void loadKeyframes(void);                                                                      // :266-302
    void precleaningKeyframes(float _radius);                                                      // :506-533
    void mergeScansWithinGlobalCoord(void);                                                        // :186-202

    void parseStaticScansViaProjection(void);
    void parseUpdatedStaticScansViaProjection(void);
    void parseUpdatedStrongStaticScansViaProjection(void);
    void parsePDScansViaProjection(void);
    void parseStrongPDScansViaProjection(void);
    void parseWeakNDScansViaProjection(void);
    void parseStrongNDScansViaProjection(void);
    void parseScansViaProjection(const CloudPtr& _map, ScansPtr& _vec_to_store);                   // :348-360

    void updateScansScanwise();                                                                    // :362-380
    void extractLowDynPointsViaKnnDiff(const CloudPtr& _target_map);                               // :393-427
    void extractHighDynPointsViaKnnDiff(const CloudPtr& _target_map);                              // :487-504
    // per-scan forms (Session.cpp:537-607, 610-642), thin wrappers: same rule on one keyframe against the target map of the last
    // extract*ViaKnnDiff call (the reference keeps that map's kd-tree as a member); {coexist, diff} resp. {static, dynamic}
    std::pair<CloudPtr, CloudPtr> partitionLowDynamicPointsOfScanByKnn(int _scan_idx);
    std::pair<CloudPtr, CloudPtr> partitionHighDynamicPointsOfScanByKnn(int _scan_idx);
    void allocateMemory() {}                                                                       // :62-78: nothing to pre-allocate here
    CloudPtr knn_target_map_;
    void constructGlobalNDMap();                                                                   // :430-435
    void removeWeakNDMapPointsHavingStrongNDInNear();                                              // :452-484
    void constructGlobalPDMap();                                                                   // :437-445
    void revertStrongPDMapPointsHavingWeakPDInNear();                                              // :447-450 (empty in the reference)

    // helpers shared with Removerter
    CloudPtr wrap(ltm_cloud h) const { return std::make_shared<CloudH>(dev_->ctx, h); }
    ScansPtr wrap_scans(ltm_scanset h) const { return std::make_shared<ScansH>(dev_->ctx, h); }
    ScansPtr wrap_shard(ltm_scanset h) const;        // result of a per-keyframe stage over [kf_begin_, kf_end_)
    ScansPtr gatherScans(const ScansPtr& scans) const;   // whole scan set on every rank (all-gather-v); identity on a single GPU
    // the scans / poses / keyframe range to hand to a per-keyframe C-ABI stage for `scans` (whole or shard)
    void stageArgs(const ScansPtr& scans, ltm_poses* poses, size_t* kb, size_t* ke) const;
    CloudPtr mergeScansToGlobal(const ScansPtr& scans) const;        // utility.cpp:170-192
    // octreeDownsampling(mergeScansToGlobal(scans), leaf), bit for bit.  Keyframe-sharded run with a rank-local `scans`: the key-range exchange of
    // include/ltm.h (box and histogram combined over the ranks, one all-to-all of the points, all-gather of the centroid lists) instead of
    // gathering the scans and gridding the whole merge on every rank
    CloudPtr mergeVoxel(const ScansPtr& scans, float leaf) const;
    CloudPtr octreeDownsampling(const CloudPtr& src, float leaf) const;   // utility.cpp:204-219
    // the same for several independent clouds (consecutive octreeDownsampling calls of the reference): one ltm_voxel_centroid_batch,
    // i.e. two host round trips for all of them
    std::vector<CloudPtr> octreeDownsamplingBatch(const std::vector<CloudPtr>& src, float leaf) const;
    CloudPtr concat(const std::vector<CloudPtr>& parts) const;
    void uploadPoses();
    void setKeyframeBlock();        // kf_begin_ / kf_end_ / poses_local_h_ of rank() in world()
    // lanes: this session becomes `from` as ITS device sees it -- names and poses copied (poses uploaded to this device), the keyframe scans a borrowed view
    void adoptKeyframes(const Session& from, bool with_scans);
};

} // namespace ltremovert
