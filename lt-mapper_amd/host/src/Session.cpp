// Session.cpp -- mirror of ltremovert/src/Session.cpp over the C ABI.  Host plumbing (directory scan, pose parsing,
// keyframe selection, PCD loading) is restated here; every loop over points is a call into libltm_hip.so.
#include "removert/Session.h"

#include <cmath>
#include <fstream>
#include <iostream>
#include <stdexcept>

namespace ltremovert
{

Device::Device(const RosParamServer& p)
{
    ltm_config cfg{};
    cfg.vfov = p.kVFOV; cfg.hfov = p.kHFOV; cfg.device = p.gpu_device_; cfg.max_kf_batch = 0;
    for (int i = 0; i < 16; ++i) cfg.lidar2base[i] = p.kSE3MatExtrinsicLiDARtoPoseBase[i];
    const int rc = ltm_create(&cfg, &ctx);
    if (rc != LTM_OK) throw std::runtime_error("ltm_create failed (" + std::to_string(rc) + "): no usable MI355X device; there is no CPU fallback");
}
Device::~Device() { ltm_destroy(ctx); }

CloudH::~CloudH() { if (ctx && h) ltm_cloud_free(ctx, h); }
size_t CloudH::size() const { size_t n = 0; ltmCheck(ctx, ltm_cloud_size(ctx, h, &n), "ltm_cloud_size"); return n; }
Cloud CloudH::download() const
{
    Cloud c(size());
    ltmCheck(ctx, ltm_cloud_download(ctx, h, c.data(), c.size(), sizeof(PointType)), "ltm_cloud_download");
    return c;
}
ScansH::~ScansH() { if (ctx && h) ltm_scanset_free(ctx, h); }
size_t ScansH::numKeyframes() const { size_t nk = 0, np = 0; ltmCheck(ctx, ltm_scanset_info(ctx, h, &nk, &np), "ltm_scanset_info"); return nk; }
std::vector<Cloud> ScansH::download() const
{
    size_t nk = 0, np = 0;
    ltmCheck(ctx, ltm_scanset_info(ctx, h, &nk, &np), "ltm_scanset_info");
    std::vector<uint64_t> off(nk + 1);
    ltmCheck(ctx, ltm_scanset_offsets(ctx, h, off.data()), "ltm_scanset_offsets");
    Cloud all(np);
    ltmCheck(ctx, ltm_scanset_download(ctx, h, all.data(), np, sizeof(PointType)), "ltm_scanset_download");
    std::vector<Cloud> out(nk);
    for (size_t k = 0; k < nk; ++k) out[k].assign(all.begin() + off[k], all.begin() + off[k + 1]);
    return out;
}

// ------------------------------------------------------------------ utility.h free functions on device clouds
static float alphaOfShape(ltm_ctx* ctx, const std::pair<float, float>& fov, const std::pair<int, int>& shape)
{
    (void)ctx;
    const float alpha = (float)shape.first / fov.first;
    int r = 0, c = 0;
    ltm_rimg_size(fov.first, fov.second, alpha, &r, &c);
    if (r != shape.first || c != shape.second) throw std::runtime_error("range image shape is not a resolution of this field of view");
    return alpha;
}
RangeImage map2RangeImg(const CloudPtr& _scan, const std::pair<float, float> _fov, const std::pair<int, int> _rimg_size)
{
    RangeImage img;
    img.rows = _rimg_size.first; img.cols = _rimg_size.second;
    img.range.resize((size_t)img.rows * img.cols);
    img.ptidx.resize(img.range.size());
    ltmCheck(_scan->ctx, ltm_debug_range_image(_scan->ctx, _scan->h, nullptr, nullptr, alphaOfShape(_scan->ctx, _fov, _rimg_size), img.range.data(), img.ptidx.data()),
             "ltm_debug_range_image");
    return img;
}
CloudPtr parseProjectedPoints(const CloudPtr& _scan, const std::pair<float, float> _fov, const std::pair<int, int> _rimg_size)
{
    const RangeImage img = map2RangeImg(_scan, _fov, _rimg_size);
    std::vector<int32_t> idx;                              // row-major, `ptidx != 0` (utility.cpp:82: point 0 doubles as "empty", quirk Q3)
    for (int32_t v : img.ptidx) if (v != 0) idx.push_back(v);
    ltm_cloud h = 0;
    ltmCheck(_scan->ctx, ltm_cloud_select(_scan->ctx, _scan->h, idx.data(), idx.size(), &h), "ltm_cloud_select");
    return std::make_shared<CloudH>(_scan->ctx, h);
}
static CloudPtr transformed(const CloudPtr& in, const Matrix4d& first, const Matrix4d& second)
{
    ltm_cloud h = 0;
    ltmCheck(in->ctx, ltm_cloud_transform(in->ctx, in->h, first.data(), second.data(), &h), "ltm_cloud_transform");
    return std::make_shared<CloudH>(in->ctx, h);
}
void transformGlobalMapToLocal(const CloudPtr& _map_global, const Matrix4d& _base_pose_inverse, const Matrix4d& _base2lidar, CloudPtr& _map_local)
{
    _map_local = transformed(_map_global, _base_pose_inverse, _base2lidar);
}
CloudPtr local2global(const CloudPtr& _scan_local, const Matrix4d& _scan_pose, const Matrix4d& _base2lidar) { return transformed(_scan_local, _base2lidar, _scan_pose); }
CloudPtr global2local(const CloudPtr& _scan_global, const Matrix4d& _scan_pose_inverse, const Matrix4d& _base2lidar) { return transformed(_scan_global, _scan_pose_inverse, _base2lidar); }
CloudPtr mergeScansWithinGlobalCoordUtil(const std::vector<CloudPtr>& _scans, const std::vector<Matrix4d>& _scans_poses, const Matrix4d& _lidar2base)
{
    if (_scans.empty()) throw std::runtime_error("mergeScansWithinGlobalCoordUtil: no scans");
    std::vector<CloudPtr> parts;
    std::vector<ltm_cloud> hs;
    for (size_t i = 0; i < _scans.size(); ++i) { parts.push_back(transformed(_scans[i], _lidar2base, _scans_poses.at(i))); hs.push_back(parts.back()->h); }
    ltm_cloud h = 0;
    ltmCheck(_scans[0]->ctx, ltm_cloud_concat(_scans[0]->ctx, hs.data(), hs.size(), &h), "ltm_cloud_concat");
    return std::make_shared<CloudH>(_scans[0]->ctx, h);
}
void octreeDownsampling(const CloudPtr& _src, CloudPtr& _to_save, const float _kDownsampleVoxelSize)
{
    ltm_cloud h = 0;
    ltmCheck(_src->ctx, ltm_voxel_centroid(_src->ctx, _src->h, _kDownsampleVoxelSize, &h), "ltm_voxel_centroid");
    _to_save = std::make_shared<CloudH>(_src->ctx, h);
}
std::set<int> convertIntVecToSet(const std::vector<int>& v) { return std::set<int>(v.begin(), v.end()); }

Session::Session(std::shared_ptr<Device> dev) : dev_(std::move(dev)) { kDownsampleVoxelSize = RosParamServer::kDownsampleVoxelSize; }

void Session::loadSessionInfo(std::string _sess_type, std::string _scan_dir, std::string _pose_path)
{
    sess_type_ = _sess_type; scan_dir_ = _scan_dir; pose_path_ = _pose_path;
    scan_paths_ = listDirectorySorted(_scan_dir, &scan_names_);
    num_scans_ = (int)scan_paths_.size();
    std::cout << "\033[1;32m Total : " << num_scans_ << " scans in the directory.\033[0m" << std::endl;

    std::ifstream pose_file_handle(_pose_path);
    if (!pose_file_handle) throw std::runtime_error("cannot open pose file " + _pose_path);
    std::string strOneLine;
    while (getline(pose_file_handle, strOneLine)) {
        std::vector<double> ith_pose_vec = splitPoseLine(strOneLine, ' ');
        if (ith_pose_vec.empty()) continue;
        if (ith_pose_vec.size() == 12) { ith_pose_vec.insert(ith_pose_vec.end(), {0.0, 0.0, 0.0, 1.0}); }
        if (ith_pose_vec.size() != 16) throw std::runtime_error("pose line with " + std::to_string(ith_pose_vec.size()) + " values in " + _pose_path);
        Matrix4d inv(16);
        if (!inverse4x4(ith_pose_vec.data(), inv.data())) throw std::runtime_error("singular pose in " + _pose_path);
        scan_poses_.emplace_back(ith_pose_vec);
        scan_inverse_poses_.emplace_back(inv);
    }
    // Session.cpp:116-117: an assert compiled out in the reference's Release build -- a mismatch is an error here
    if (scan_paths_.size() != scan_poses_.size())
        throw std::runtime_error("scan count (" + std::to_string(scan_paths_.size()) + ") != pose count (" + std::to_string(scan_poses_.size()) + ") for " + _scan_dir);
}

void Session::setDownsampleSize(float _voxel_size) { kDownsampleVoxelSize = _voxel_size; }

void Session::clearKeyframes(void)
{
    keyframe_names_.clear(); keyframe_paths_.clear(); keyframe_poses_.clear(); keyframe_inverse_poses_.clear();
}

void Session::parseKeyframes(std::pair<int, int> _range, int _gap)
{
    clearKeyframes();
    const int start_idx = _range.first, end_idx = _range.second;
    int num_valid_parsed{0};
    for (int curr_idx = 0; curr_idx < int(scan_paths_.size()); curr_idx++) {
        if (curr_idx > end_idx || curr_idx < start_idx) {
            curr_idx++;          // sic, Session.cpp:149-152 (quirk Q6): out-of-range indices advance twice
            continue;
        }
        if (std::remainder(num_valid_parsed, _gap) != 0) { num_valid_parsed++; continue; }
        keyframe_paths_.emplace_back(scan_paths_.at(curr_idx));
        keyframe_names_.emplace_back(scan_names_.at(curr_idx));
        keyframe_poses_.emplace_back(scan_poses_.at(curr_idx));
        keyframe_inverse_poses_.emplace_back(scan_inverse_poses_.at(curr_idx));
        num_valid_parsed++;
    }
    std::cout << "\033[1;32m Total " << keyframe_paths_.size() << " nodes are used from the index range [" << start_idx << ", " << end_idx << "]"
              << " (every " << _gap << " frames parsed)\033[0m" << std::endl;
}

void Session::parseKeyframes(int _gap)
{
    clearKeyframes();
    parseKeyframes({0, int(scan_paths_.size())}, _gap);
}

static double xyzDist(const Matrix4d& a, const Matrix4d& b)   // Session.cpp:205-212
{
    return std::sqrt((a[3] - b[3]) * (a[3] - b[3]) + (a[7] - b[7]) * (a[7] - b[7]) + (a[11] - b[11]) * (a[11] - b[11]));
}

void Session::parseKeyframesInROI(const std::vector<Matrix4d>& _roi_poses, int _gap)
{
    clearKeyframes();
    const double inplace_thres = 10.0;   // Session.cpp:234
    int num_valid_parsed{0};
    for (int curr_idx = 0; curr_idx < int(scan_paths_.size()); curr_idx++) {
        double nn_dist = 10000000000.0;
        for (const auto& roi : _roi_poses) nn_dist = std::min(nn_dist, xyzDist(scan_poses_.at(curr_idx), roi));
        if (nn_dist > inplace_thres) continue;
        if (std::remainder(num_valid_parsed, _gap) != 0) { num_valid_parsed++; continue; }
        keyframe_paths_.emplace_back(scan_paths_.at(curr_idx));
        keyframe_names_.emplace_back(scan_names_.at(curr_idx));
        keyframe_poses_.emplace_back(scan_poses_.at(curr_idx));
        keyframe_inverse_poses_.emplace_back(scan_inverse_poses_.at(curr_idx));
        num_valid_parsed++;
    }
    std::cout << "\033[1;32m Total " << keyframe_paths_.size() << " keyframes parsed in the map's ROI\033[0m" << std::endl;
}

void Session::uploadPoses()
{
    if (poses_h_) { ltm_poses_free(dev_->ctx, poses_h_); poses_h_ = 0; }
    std::vector<double> p, pi;
    for (size_t k = 0; k < keyframe_poses_.size(); ++k) {
        p.insert(p.end(), keyframe_poses_[k].begin(), keyframe_poses_[k].end());
        pi.insert(pi.end(), keyframe_inverse_poses_[k].begin(), keyframe_inverse_poses_[k].end());
    }
    ltmCheck(dev_->ctx, ltm_poses_create(dev_->ctx, keyframe_poses_.size(), p.data(), pi.data(), &poses_h_), "ltm_poses_create");
}

void Session::loadKeyframes(void)
{
    const int cout_interval{10};
    int cout_counter{0};
    std::cout << std::endl << " ... (display every " << cout_interval << " readings) ..." << std::endl;
    // PCD decode + per-scan VoxelGrid are independent per keyframe: done on all host cores, concatenated in keyframe order
    std::vector<Cloud> per_kf(keyframe_paths_.size());
    std::vector<size_t> raw_sizes(keyframe_paths_.size(), 0);
    parallelFor(keyframe_paths_.size(), [&](size_t k) {
        Cloud points;
        std::string err;
        if (!loadPCDFile(keyframe_paths_[k], points, &err)) throw std::runtime_error(err);
        raw_sizes[k] = points.size();
        voxelGridFilter(points, kDownsampleVoxelSize, per_kf[k]);
    }, (unsigned)std::max(1, kNumOmpCores));
    Cloud all;
    std::vector<uint64_t> offsets(1, 0);
    for (size_t k = 0; k < per_kf.size(); ++k) {
        all.insert(all.end(), per_kf[k].begin(), per_kf[k].end());
        offsets.push_back(all.size());
        if (++cout_counter % cout_interval == 0)
            std::cout << keyframe_paths_[k] << std::endl << "Read a pointcloud with " << raw_sizes[k] << " points (downsampled size: " << per_kf[k].size() << " points)" << std::endl;
        Cloud().swap(per_kf[k]);
    }
    ltm_scanset h = 0;
    ltmCheck(dev_->ctx, ltm_scanset_upload(dev_->ctx, all.data(), sizeof(PointType), offsets.data(), offsets.size() - 1, &h), "ltm_scanset_upload");
    keyframe_scans_ = wrap_scans(h);
    uploadPoses();
}

void Session::precleaningKeyframes(float _radius)
{
    ltm_scanset h = 0;
    ltmCheck(dev_->ctx, ltm_preclean(dev_->ctx, keyframe_scans_->h, _radius, &h), "ltm_preclean");
    keyframe_scans_ = wrap_scans(h);
}

CloudPtr Session::mergeScansToGlobal(const ScansPtr& scans) const
{
    ltm_cloud h = 0;
    ltmCheck(dev_->ctx, ltm_merge_to_global(dev_->ctx, scans->h, poses_h_, &h), "ltm_merge_to_global");
    return wrap(h);
}
CloudPtr Session::octreeDownsampling(const CloudPtr& src, float leaf) const
{
    ltm_cloud h = 0;
    ltmCheck(dev_->ctx, ltm_voxel_centroid(dev_->ctx, src->h, leaf, &h), "ltm_voxel_centroid");
    return wrap(h);
}
CloudPtr Session::concat(const std::vector<CloudPtr>& parts) const
{
    std::vector<ltm_cloud> hs;
    for (const auto& p : parts) if (p) hs.push_back(p->h);
    ltm_cloud h = 0;
    ltmCheck(dev_->ctx, ltm_cloud_concat(dev_->ctx, hs.data(), hs.size(), &h), "ltm_cloud_concat");
    return wrap(h);
}

void Session::mergeScansWithinGlobalCoord(void) { map_global_orig_ = mergeScansToGlobal(keyframe_scans_); }

void Session::parseScansViaProjection(const CloudPtr& _map, ScansPtr& _vec_to_store)
{
    ltm_scanset h = 0;
    ltmCheck(dev_->ctx, ltm_reproject(dev_->ctx, _map->h, poses_h_, 0, keyframe_poses_.size(), kReprojectionAlpha, &h), "ltm_reproject");
    _vec_to_store = wrap_scans(h);
}
void Session::parseStaticScansViaProjection(void) { parseScansViaProjection(map_global_curr_, keyframe_scans_static_projected_); }
void Session::parseUpdatedStaticScansViaProjection(void) { parseScansViaProjection(map_global_updated_, keyframe_scans_updated_); }
void Session::parseUpdatedStrongStaticScansViaProjection(void) { parseScansViaProjection(map_global_updated_strong_, keyframe_scans_updated_strong_); }
void Session::parsePDScansViaProjection(void) { parseScansViaProjection(map_global_pd_orig_, keyframe_scans_pd_); }
void Session::parseStrongPDScansViaProjection(void) { parseScansViaProjection(map_global_pd_strong_, keyframe_scans_strong_pd_); }
void Session::parseWeakNDScansViaProjection(void) { parseScansViaProjection(map_global_nd_weak_, keyframe_scans_weak_nd_); }
void Session::parseStrongNDScansViaProjection(void) { parseScansViaProjection(map_global_nd_strong_, keyframe_scans_strong_nd_); }

void Session::updateScansScanwise()
{
    ltm_scanset merged = 0, voxelised = 0;
    ltmCheck(dev_->ctx, ltm_scanset_zip_concat(dev_->ctx, keyframe_scans_updated_->h, keyframe_scans_weak_nd_->h, keyframe_scans_pd_->h, &merged), "ltm_scanset_zip_concat");
    ScansPtr guard = wrap_scans(merged);
    ltmCheck(dev_->ctx, ltm_voxel_centroid_scanset(dev_->ctx, merged, 0.05f, &voxelised), "ltm_voxel_centroid_scanset");
    keyframe_scans_updated_ = wrap_scans(voxelised);
}

static std::pair<CloudPtr, CloudPtr> knnOneScan(const Session& s, const CloudPtr& target, const ScansPtr& scans, int idx)
{
    if (!target) throw std::runtime_error("no kNN target map yet: call extract*ViaKnnDiff first");
    ltm_scanset co = 0, di = 0;
    ltmCheck(s.dev_->ctx, ltm_knn_partition(s.dev_->ctx, target->h, scans->h, s.poses_h_, (size_t)idx, (size_t)idx + 1, s.kNumKnnPointsToCompare,
                                            s.kScanKnnAndMapKnnAvgDiffThreshold, &co, &di), "ltm_knn_partition");
    const ScansH a(s.dev_->ctx, co), b(s.dev_->ctx, di);          // freed on return
    ltm_cloud ca = 0, cb = 0;
    ltmCheck(s.dev_->ctx, ltm_scanset_keyframe(s.dev_->ctx, co, 0, &ca), "ltm_scanset_keyframe");
    ltmCheck(s.dev_->ctx, ltm_scanset_keyframe(s.dev_->ctx, di, 0, &cb), "ltm_scanset_keyframe");
    return {s.wrap(ca), s.wrap(cb)};
}
std::pair<CloudPtr, CloudPtr> Session::partitionLowDynamicPointsOfScanByKnn(int _scan_idx) { return knnOneScan(*this, knn_target_map_, keyframe_scans_static_projected_, _scan_idx); }
std::pair<CloudPtr, CloudPtr> Session::partitionHighDynamicPointsOfScanByKnn(int _scan_idx) { return knnOneScan(*this, knn_target_map_, keyframe_scans_, _scan_idx); }

void Session::extractLowDynPointsViaKnnDiff(const CloudPtr& _target_map)
{
    knn_target_map_ = _target_map;
    // Session.cpp:395-402 build a 0.4 m octree for an ICP that is disabled (useICPrefinement{false}); no observable effect, not run
    ltm_scanset co = 0, di = 0;
    ltmCheck(dev_->ctx, ltm_knn_partition(dev_->ctx, _target_map->h, keyframe_scans_static_projected_->h, poses_h_, 0, keyframe_poses_.size(),
                                          kNumKnnPointsToCompare, kScanKnnAndMapKnnAvgDiffThreshold, &co, &di), "ltm_knn_partition");
    scans_knn_coexist_ = wrap_scans(co); scans_knn_diff_ = wrap_scans(di);
}

void Session::extractHighDynPointsViaKnnDiff(const CloudPtr& _target_map)
{
    knn_target_map_ = _target_map;
    ltm_scanset di = 0;
    ltmCheck(dev_->ctx, ltm_knn_partition(dev_->ctx, _target_map->h, keyframe_scans_->h, poses_h_, 0, keyframe_poses_.size(),
                                          kNumKnnPointsToCompare, kScanKnnAndMapKnnAvgDiffThreshold, nullptr, &di), "ltm_knn_partition");
    keyframe_scans_dynamic_ = wrap_scans(di);
}

void Session::constructGlobalNDMap() { map_global_nd_ = octreeDownsampling(mergeScansToGlobal(scans_knn_diff_), 0.05f); }
void Session::constructGlobalPDMap()
{
    map_global_pd_ = octreeDownsampling(mergeScansToGlobal(scans_knn_diff_), 0.05f);
    map_global_pd_orig_ = map_global_pd_;
}
void Session::revertStrongPDMapPointsHavingWeakPDInNear() {}

void Session::removeWeakNDMapPointsHavingStrongNDInNear()
{
    if (!map_global_nd_strong_ || map_global_nd_strong_->size() == 0) return;
    ltm_cloud near = 0, far = 0;   // kNumKnnPointsToCompare = 2, threshold 1.0 hard-coded at Session.cpp:468-469
    ltmCheck(dev_->ctx, ltm_knn_split_cloud(dev_->ctx, map_global_nd_strong_->h, map_global_nd_weak_->h, 2, 1.0f, &near, &far), "ltm_knn_split_cloud");
    CloudPtr add = wrap(near);
    map_global_nd_strong_ = concat({map_global_nd_strong_, add});   // += (append)
    map_global_nd_weak_ = wrap(far);                                 // = (remake)
}

} // namespace ltremovert
