// Session.cpp -- mirror of ltremovert/src/Session.cpp over the C ABI.  Host plumbing (directory scan, pose parsing,
// keyframe selection, PCD loading) is restated here; every loop over points is a call into libltm_hip.so.
#include "removert/Session.h"

#include <cstdio>
#include <chrono>
#include <memory>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <stdexcept>
#include <thread>

namespace ltremovert
{

size_t Session::kVoxelShardMin = getenv("LTM_VOXEL_SHARD_MIN") ? (size_t)atoll(getenv("LTM_VOXEL_SHARD_MIN")) : ((size_t)1 << 24);

Device::Device(const RosParamServer& p, int device_ordinal, std::shared_ptr<Comm> comm_) : comm(std::move(comm_))
{
    ltm_config cfg{};
    cfg.vfov = p.kVFOV; cfg.hfov = p.kHFOV; cfg.device = device_ordinal >= 0 ? device_ordinal : p.gpu_device_; cfg.max_kf_batch = 0;
    for (int i = 0; i < 16; ++i) cfg.lidar2base[i] = p.kSE3MatExtrinsicLiDARtoPoseBase[i];
    const int rc = ltm_create(&cfg, &ctx);
    if (rc != LTM_OK) throw std::runtime_error("ltm_create failed (" + std::to_string(rc) + "): no usable MI355X device; there is no CPU fallback");
}
Device::Device(LaneOf l)
{
    const int rc = ltm_lane_create(l.parent.ctx, &ctx);
    if (rc != LTM_OK) throw std::runtime_error("ltm_lane_create failed (" + std::to_string(rc) + ")");
}
Device::~Device() { ltm_destroy(ctx); }

CloudPtr lendCloud(const CloudPtr& c, Device& to)
{
    if (!c) return nullptr;
    ltm_cloud h = 0;
    ltmCheck(c->ctx, ltm_cloud_lend(c->ctx, c->h, to.ctx, &h), "ltm_cloud_lend");
    auto v = std::make_shared<CloudH>(to.ctx, h);
    v->lender = c;
    return v;
}
CloudPtr giveCloud(CloudPtr& c, Device& to)
{
    if (!c) return nullptr;
    ltm_cloud h = 0;
    ltmCheck(c->ctx, ltm_cloud_give(c->ctx, c->h, to.ctx, &h), "ltm_cloud_give");
    c->h = 0;                 // every other holder of this pointer sees an empty handle from now on
    c.reset();
    return std::make_shared<CloudH>(to.ctx, h);
}
ScansPtr lendScans(const ScansPtr& s, Device& to)
{
    if (!s) return nullptr;
    ltm_scanset h = 0;
    ltmCheck(s->ctx, ltm_scanset_lend(s->ctx, s->h, to.ctx, &h), "ltm_scanset_lend");
    auto v = std::make_shared<ScansH>(to.ctx, h);
    v->shard = s->shard; v->kb = s->kb; v->n_total = s->n_total;
    v->lender = s;
    return v;
}
ScansPtr giveScans(ScansPtr& s, Device& to)
{
    if (!s) return nullptr;
    ltm_scanset h = 0;
    ltmCheck(s->ctx, ltm_scanset_give(s->ctx, s->h, to.ctx, &h), "ltm_scanset_give");
    auto v = std::make_shared<ScansH>(to.ctx, h);
    v->shard = s->shard; v->kb = s->kb; v->n_total = s->n_total;
    s->h = 0;
    s.reset();
    return v;
}

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
    // any alpha whose rounded products reproduce both dimensions will do (see Removerter.cpp alphaOfShape)
    const double lo = std::max(((double)shape.first - 0.5) / fov.first, ((double)shape.second - 0.5) / fov.second);
    const double hi = std::min(((double)shape.first + 0.5) / fov.first, ((double)shape.second + 0.5) / fov.second);
    const float cand[3] = {(float)(0.5 * (lo + hi)), (float)shape.second / fov.second, (float)shape.first / fov.first};
    for (float alpha : cand) {
        int r = 0, c = 0;
        ltm_rimg_size(fov.first, fov.second, alpha, &r, &c);
        if (r == shape.first && c == shape.second) return alpha;
    }
    throw std::runtime_error("range image shape is not a resolution of this field of view");
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
    if (!logQuiet()) std::cout << "\033[1;32m Total : " << num_scans_ << " scans in the directory.\033[0m" << std::endl;

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
    if (!logQuiet()) std::cout << "\033[1;32m Total " << keyframe_paths_.size() << " nodes are used from the index range [" << start_idx << ", " << end_idx << "]"
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
    if (!logQuiet()) std::cout << "\033[1;32m Total " << keyframe_paths_.size() << " keyframes parsed in the map's ROI\033[0m" << std::endl;
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
    setKeyframeBlock();
}

// this rank's block of keyframes (Comm.h shardRange) among the ranks of comm(): every per-keyframe loop of the reference runs over it
void Session::adoptKeyframes(const Session& from, bool with_scans)
{
    sess_type_ = from.sess_type_; kDownsampleVoxelSize = from.kDownsampleVoxelSize; keyframe_gap_ = from.keyframe_gap_;
    keyframe_names_ = from.keyframe_names_; keyframe_paths_ = from.keyframe_paths_;
    keyframe_poses_ = from.keyframe_poses_; keyframe_inverse_poses_ = from.keyframe_inverse_poses_;
    uploadPoses();
    keyframe_scans_ = with_scans ? lendScans(from.keyframe_scans_, *dev_) : nullptr;
}
void Session::setKeyframeBlock()
{
    if (poses_local_h_) { ltm_poses_free(dev_->ctx, poses_local_h_); poses_local_h_ = 0; }
    shardRange(keyframe_poses_.size(), rank(), world(), &kf_begin_, &kf_end_);
    if (world() > 1) {
        std::vector<double> p, pi;
        for (size_t k = kf_begin_; k < kf_end_; ++k) {
            p.insert(p.end(), keyframe_poses_[k].begin(), keyframe_poses_[k].end());
            pi.insert(pi.end(), keyframe_inverse_poses_[k].begin(), keyframe_inverse_poses_[k].end());
        }
        ltmCheck(dev_->ctx, ltm_poses_create(dev_->ctx, kf_end_ - kf_begin_, p.data(), pi.data(), &poses_local_h_), "ltm_poses_create");
    }
}

void Session::enterSessionGroup(std::shared_ptr<Comm> group)
{
    group_comm_ = std::move(group);
    setKeyframeBlock();
}
void Session::leaveSessionGroup()
{
    group_comm_.reset();
    setKeyframeBlock();
}

ScansPtr Session::wrap_shard(ltm_scanset h) const
{
    ScansPtr s = wrap_scans(h);
    if (world() > 1) { s->shard = true; s->kb = kf_begin_; s->n_total = keyframe_poses_.size(); }
    return s;
}

void Session::stageArgs(const ScansPtr& scans, ltm_poses* poses, size_t* kb, size_t* ke) const
{
    if (scans->shard) {      // a shard holds exactly this rank's keyframes: address it with the local pose slice
        if (scans->kb != kf_begin_ || scans->numKeyframes() != kf_end_ - kf_begin_) throw std::runtime_error("scan shard and keyframe block disagree");
        *poses = poses_local_h_; *kb = 0; *ke = kf_end_ - kf_begin_;
    } else {
        *poses = poses_h_; *kb = kf_begin_; *ke = kf_end_;
    }
}

// all-gather-v of a sharded scan set: keyframe counts and per-keyframe sizes first (host integers), then the points straight into
// the assembled scan set's device array.  Blocks are contiguous in rank order, so the concatenation is in keyframe order.
ScansPtr Session::gatherScans(const ScansPtr& scans) const
{
    if (!scans->shard) return scans;
    Comm& comm = *this->comm();
    ltm_ctx* ctx = dev_->ctx;
    size_t nk = 0, np = 0;
    ltmCheck(ctx, ltm_scanset_info(ctx, scans->h, &nk, &np), "ltm_scanset_info");
    std::vector<uint64_t> off(nk + 1);
    ltmCheck(ctx, ltm_scanset_offsets(ctx, scans->h, off.data()), "ltm_scanset_offsets");
    // per-keyframe point counts of every rank: one all-gather-v of the count tables
    std::vector<uint64_t> nks, nps;
    comm.allGatherU64(ctx, nk, nks);
    comm.allGatherU64(ctx, np, nps);
    size_t total_kf = 0;
    for (uint64_t v : nks) total_kf += v;
    if (total_kf != scans->n_total) throw std::runtime_error("gatherScans: keyframe blocks do not add up");
    std::vector<uint64_t> cnt_bytes(nks.size());
    for (size_t r = 0; r < nks.size(); ++r) cnt_bytes[r] = nks[r] * sizeof(uint64_t);
    std::vector<uint64_t> my_counts(nk), all_counts(total_kf);
    for (size_t k = 0; k < nk; ++k) my_counts[k] = off[k + 1] - off[k];
    void *d_my = nullptr, *d_all = nullptr;
    ltmCheck(ctx, ltm_buffer_alloc(ctx, std::max<size_t>(nk, 1) * 8, &d_my), "ltm_buffer_alloc");
    ltmCheck(ctx, ltm_buffer_alloc(ctx, std::max<size_t>(total_kf, 1) * 8, &d_all), "ltm_buffer_alloc");
    ltmCheck(ctx, ltm_buffer_copy(ctx, d_my, my_counts.data(), nk * 8, 0), "ltm_buffer_copy");
    comm.allGatherV(ctx, d_my, nk * 8, d_all, cnt_bytes);
    ltmCheck(ctx, ltm_buffer_copy(ctx, all_counts.data(), d_all, total_kf * 8, 1), "ltm_buffer_copy");
    ltmCheck(ctx, ltm_buffer_free(ctx, d_my), "ltm_buffer_free");
    ltmCheck(ctx, ltm_buffer_free(ctx, d_all), "ltm_buffer_free");
    std::vector<uint64_t> full_off(total_kf + 1, 0);
    for (size_t k = 0; k < total_kf; ++k) full_off[k + 1] = full_off[k] + all_counts[k];
    ltm_scanset h = 0;
    ltmCheck(ctx, ltm_scanset_alloc(ctx, full_off.data(), total_kf, &h), "ltm_scanset_alloc");
    ScansPtr full = wrap_scans(h);
    const void *src = nullptr, *dst = nullptr;
    ltmCheck(ctx, ltm_scanset_device_ptr(ctx, scans->h, &src), "ltm_scanset_device_ptr");
    ltmCheck(ctx, ltm_scanset_device_ptr(ctx, h, &dst), "ltm_scanset_device_ptr");
    std::vector<uint64_t> pt_bytes(nps.size());
    for (size_t r = 0; r < nps.size(); ++r) pt_bytes[r] = nps[r] * sizeof(PointType);
    comm.allGatherV(ctx, src, np * sizeof(PointType), const_cast<void*>(dst), pt_bytes);
    ltmCheck(ctx, ltm_synchronize(ctx), "ltm_synchronize");
    return full;
}

void Session::loadKeyframes(void)
{
    const int cout_interval{10};
    if (!logQuiet()) std::cout << std::endl << " ... (display every " << cout_interval << " readings) ..." << std::endl;
    const size_t n_kf = keyframe_paths_.size();
    // The reference reads its scans on one thread (Session.cpp:266-302; num_omp_cores is for its OpenMP regions).  Here one task per file -- and the task
    // is not the read (0.1 ms) but pcl::VoxelGrid's std::sort order on the scan's 54 k points (1.7-2.3 ms of thread time per os1-64 scan on the lot, where
    // the grid really thins the scan; LTM_STEP0_TIMING prints it): ~0.9 s of CPU per 500-keyframe session.  As many threads as the hand-over of the cascade
    // uses for the same sort, up to a cap (hardware threads, at most 32; LTM_LOADER_THREADS overrides): a container's CPU quota is averaged over 100 ms periods, so a
    // burst this short runs wider than its 16 CPUs -- up to a point: measured on the GPU box (profiles/r5_step0_loader_threads.txt) 16 threads leave the feeder
    // waiting for the decode (query session 56 ms), 32 do not (29 ms), 64 and 128 starve the feeder thread's staging copies instead (57-67 ms).
    const char* lt_env = std::getenv("LTM_LOADER_THREADS");
    const unsigned hw_threads = std::max(1u, std::thread::hardware_concurrency());
    const unsigned n_threads = lt_env && std::atoi(lt_env) > 0 ? (unsigned)std::atoi(lt_env)
                                                              : std::max<unsigned>((unsigned)std::max(1, kNumOmpCores), std::min(32u, hw_threads));
    std::vector<Cloud> per_kf(n_kf);
    std::vector<size_t> raw_sizes(n_kf, 0), out_sizes(n_kf, 0);
    std::atomic<uint64_t> decode_us[2] = {{0}, {0}};      // LTM_STEP0_TIMING: thread time inside loadPCDFile / voxelGridFilter, summed over the files
    auto decode = [&](size_t k) {          // Session.cpp:272-292: loadPCDFile + per-scan pcl::VoxelGrid
        Cloud points;
        std::string err;
        const auto td0 = std::chrono::steady_clock::now();
        if (!loadPCDFile(keyframe_paths_[k], points, &err)) throw std::runtime_error(err);
        const auto td1 = std::chrono::steady_clock::now();
        raw_sizes[k] = points.size();
        voxelGridFilter(std::move(points), kDownsampleVoxelSize, per_kf[k]);
        out_sizes[k] = per_kf[k].size();
        decode_us[0] += (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(td1 - td0).count();
        decode_us[1] += (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - td1).count();
    };
    auto report = [&](size_t k) {
        if ((k + 1) % cout_interval == 0 && !logQuiet())
            std::cout << keyframe_paths_[k] << std::endl << "Read a pointcloud with " << raw_sizes[k] << " points (downsampled size: " << out_sizes[k] << " points)" << std::endl;
    };
    ltm_scanset h = 0;
    if (!gpu_async_io_) {
        // synchronous path: decode everything (on all host cores), concatenate in keyframe order, one upload
        parallelFor(n_kf, decode, n_threads);
        Cloud all;
        std::vector<uint64_t> offsets(1, 0);
        for (size_t k = 0; k < n_kf; ++k) {
            all.insert(all.end(), per_kf[k].begin(), per_kf[k].end());
            offsets.push_back(all.size());
            report(k);
            Cloud().swap(per_kf[k]);
        }
        ltmCheck(dev_->ctx, ltm_scanset_upload(dev_->ctx, all.data(), sizeof(PointType), offsets.data(), offsets.size() - 1, &h), "ltm_scanset_upload");
    } else {
        // Pipelined feeder: decode threads fill per_kf[] in any order; this thread hands every keyframe to the device as
        // soon as it AND all earlier ones are ready -- pinned double buffering on the copy stream, so PCD decode, the per-scan
        // VoxelGrid, host packing and the H2D DMA overlap.  The device array is sized from the PCD headers (VoxelGrid never grows a scan).
        using clk = std::chrono::steady_clock;
        const bool detail = std::getenv("LTM_STEP0_TIMING") != nullptr;
        const auto tl0 = clk::now();
        double wait_ms = 0.0, chunk_ms = 0.0;
        std::vector<size_t> header_pts(n_kf, 0);
        parallelFor(n_kf, [&](size_t k) { std::string err; if (!readPCDPointCount(keyframe_paths_[k], &header_pts[k], &err)) throw std::runtime_error(err); }, n_threads);
        const auto tl1 = clk::now();
        size_t capacity = 0;
        for (size_t v : header_pts) capacity += v;
        ltm_upload up = 0;
        ltmCheck(dev_->ctx, ltm_scanset_upload_begin(dev_->ctx, capacity, &up), "ltm_scanset_upload_begin");
        std::vector<char> done(n_kf, 0);
        std::mutex m;
        std::condition_variable cv;
        std::exception_ptr err;
        std::atomic<size_t> next{0};
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < std::min<size_t>(n_threads, std::max<size_t>(n_kf, 1)); ++t)
            pool.emplace_back([&] {
                for (size_t k = next++; k < n_kf; k = next++) {
                    try { decode(k); } catch (...) { std::lock_guard<std::mutex> g(m); if (!err) err = std::current_exception(); next = n_kf; }
                    { std::lock_guard<std::mutex> g(m); done[k] = 1; }
                    cv.notify_all();
                }
            });
        try {
            for (size_t k = 0; k < n_kf; ++k) {
                const auto tw0 = clk::now();
                {
                    std::unique_lock<std::mutex> lk(m);
                    cv.wait(lk, [&] { return done[k] || err; });
                    if (err) break;
                }
                const auto tw1 = clk::now();
                wait_ms += std::chrono::duration<double, std::milli>(tw1 - tw0).count();
                const uint64_t n = out_sizes[k];
                ltmCheck(dev_->ctx, ltm_scanset_upload_chunk(dev_->ctx, up, per_kf[k].data(), sizeof(PointType), &n, 1), "ltm_scanset_upload_chunk");
                chunk_ms += std::chrono::duration<double, std::milli>(clk::now() - tw1).count();
                report(k);
                Cloud().swap(per_kf[k]);
            }
        } catch (...) { std::lock_guard<std::mutex> g(m); if (!err) err = std::current_exception(); next = n_kf; }
        const auto tl2 = clk::now();
        for (auto& th : pool) th.join();
        if (err) { ltm_scanset tmp = 0; (void)ltm_scanset_upload_end(dev_->ctx, up, &tmp); if (tmp) (void)ltm_scanset_free(dev_->ctx, tmp); std::rethrow_exception(err); }
        ltmCheck(dev_->ctx, ltm_scanset_upload_end(dev_->ctx, up, &h), "ltm_scanset_upload_end");
        if (detail) {
            auto msd = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
            std::fprintf(stderr, "[ltm_run] step 0:   %s: headers %.1f ms, feeder loop %.1f ms (waiting for the decode %.1f, in upload_chunk %.1f), join + upload_end %.1f ms, %u threads; "
                                 "thread time per file: loadPCDFile %.0f us, voxelGridFilter %.0f us\n",
                         sess_type_.c_str(), msd(tl0, tl1), msd(tl1, tl2), wait_ms, chunk_ms, msd(tl2, clk::now()), n_threads,
                         (double)decode_us[0].load() / (double)std::max<size_t>(n_kf, 1), (double)decode_us[1].load() / (double)std::max<size_t>(n_kf, 1));
        }
    }
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
    const ScansPtr whole = gatherScans(scans);          // needs every keyframe: the one place where per-keyframe results are exchanged
    ltm_cloud h = 0;
    ltmCheck(dev_->ctx, ltm_merge_to_global(dev_->ctx, whole->h, poses_h_, &h), "ltm_merge_to_global");
    return wrap(h);
}
CloudPtr Session::mergeVoxel(const ScansPtr& scans, float leaf) const
{
    if (world() == 1 || !scans->shard) return octreeDownsampling(mergeScansToGlobal(scans), leaf);
    Comm& comm = *this->comm();
    ltm_ctx* ctx = dev_->ctx;
    const int w = world(), me = rank();
    // this rank's keyframes in the global frame
    ltm_cloud hl = 0;
    ltmCheck(ctx, ltm_merge_to_global(ctx, scans->h, poses_local_h_, &hl), "ltm_merge_to_global");
    CloudPtr loc = wrap(hl);
    // the box of the whole merge: six floats per rank through the byte all-gather, min / max on the host (exact in float)
    float box[6];
    ltmCheck(ctx, ltm_cloud_bbox(ctx, loc->h, box, box + 3), "ltm_cloud_bbox");
    auto gather_host = [&](const void* mine, size_t bytes, std::vector<uint8_t>& all) {
        void *d_my = nullptr, *d_all = nullptr;
        ltmCheck(ctx, ltm_buffer_alloc(ctx, bytes, &d_my), "ltm_buffer_alloc");
        ltmCheck(ctx, ltm_buffer_alloc(ctx, bytes * (size_t)w, &d_all), "ltm_buffer_alloc");
        ltmCheck(ctx, ltm_buffer_copy(ctx, d_my, mine, bytes, 0), "ltm_buffer_copy");
        comm.allGatherV(ctx, d_my, bytes, d_all, std::vector<uint64_t>((size_t)w, bytes));
        all.resize(bytes * (size_t)w);
        ltmCheck(ctx, ltm_buffer_copy(ctx, all.data(), d_all, all.size(), 1), "ltm_buffer_copy");
        ltmCheck(ctx, ltm_buffer_free(ctx, d_my), "ltm_buffer_free");
        ltmCheck(ctx, ltm_buffer_free(ctx, d_all), "ltm_buffer_free");
    };
    std::vector<uint8_t> raw;
    gather_host(box, sizeof box, raw);
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int r = 0; r < w; ++r) {
        float b[6];
        std::memcpy(b, raw.data() + (size_t)r * sizeof b, sizeof b);
        for (int d = 0; d < 3; ++d) { mn[d] = std::min(mn[d], b[d]); mx[d] = std::max(mx[d], b[3 + d]); }
    }
    if (!std::isfinite(mn[0])) { ltm_cloud he = 0; ltmCheck(ctx, ltm_cloud_alloc(ctx, 0, &he), "ltm_cloud_alloc"); return wrap(he); }
    // cuts of the key space: summed 4096-bin histogram, equal weight per rank (the arithmetic of dist.ShardedOps.balanced_cuts)
    std::vector<uint32_t> hist(4096);
    ltmCheck(ctx, ltm_voxel_key_histogram(ctx, loc->h, mn, mx, leaf, hist.data()), "ltm_voxel_key_histogram");
    gather_host(hist.data(), hist.size() * sizeof(uint32_t), raw);
    std::vector<uint64_t> cum(4097, 0);
    for (size_t b = 0; b < 4096; ++b) {
        uint64_t v = 0;
        for (int r = 0; r < w; ++r) { uint32_t x; std::memcpy(&x, raw.data() + ((size_t)r * 4096 + b) * 4, 4); v += x; }
        cum[b + 1] = cum[b] + v;
    }
    std::vector<uint32_t> cuts((size_t)w + 1, 0);
    for (int r = 1; r < w; ++r) {
        const uint64_t want = cum[4096] * (uint64_t)r / (uint64_t)w;
        const uint32_t at = (uint32_t)(std::lower_bound(cum.begin(), cum.end(), want) - cum.begin());
        cuts[(size_t)r] = std::min<uint32_t>(std::max(at, cuts[(size_t)r - 1]), 4096u);
    }
    cuts[(size_t)w] = 4096;
    // every point moves once, to the owner of its key range; arrival order = rank order = keyframe order
    std::vector<ltm_cloud> parts((size_t)w, 0);
    ltmCheck(ctx, ltm_voxel_key_split(ctx, loc->h, mn, mx, leaf, (uint32_t)w, cuts.data(), parts.data()), "ltm_voxel_key_split");
    std::vector<CloudPtr> keep;
    std::vector<uint64_t> send_n((size_t)w), send_bytes((size_t)w), recv_bytes((size_t)w);
    for (int r = 0; r < w; ++r) { keep.push_back(wrap(parts[(size_t)r])); send_n[(size_t)r] = keep.back()->size(); send_bytes[(size_t)r] = send_n[(size_t)r] * sizeof(PointType); }
    gather_host(send_n.data(), send_n.size() * 8, raw);                       // everybody's send table: row r, column me = what rank r sends here
    size_t n_recv = 0;
    for (int r = 0; r < w; ++r) { uint64_t v; std::memcpy(&v, raw.data() + ((size_t)r * (size_t)w + (size_t)me) * 8, 8); recv_bytes[(size_t)r] = v * sizeof(PointType); n_recv += v; }
    CloudPtr send = concat(keep);                                             // the parts back to back, in destination order
    ltm_cloud hr = 0;
    ltmCheck(ctx, ltm_cloud_alloc(ctx, n_recv, &hr), "ltm_cloud_alloc");
    CloudPtr recv = wrap(hr);
    const void *sp = nullptr, *rp = nullptr;
    ltmCheck(ctx, ltm_cloud_device_ptr(ctx, send->h, &sp), "ltm_cloud_device_ptr");
    ltmCheck(ctx, ltm_cloud_device_ptr(ctx, recv->h, &rp), "ltm_cloud_device_ptr");
    comm.allToAllV(ctx, sp, send_bytes, const_cast<void*>(rp), recv_bytes);
    ltmCheck(ctx, ltm_synchronize(ctx), "ltm_synchronize");
    // this rank's key range under the common frame, then the centroid lists of all ranks in rank order
    ltm_cloud hv = 0;
    ltmCheck(ctx, ltm_voxel_centroid_box(ctx, recv->h, mn, mx, leaf, &hv), "ltm_voxel_centroid_box");
    CloudPtr mine = wrap(hv);
    std::vector<uint64_t> sizes;
    comm.allGatherU64(ctx, mine->size(), sizes);
    size_t total = 0;
    for (uint64_t& v : sizes) { total += v; v *= sizeof(PointType); }
    ltm_cloud ho = 0;
    ltmCheck(ctx, ltm_cloud_alloc(ctx, total, &ho), "ltm_cloud_alloc");
    CloudPtr out = wrap(ho);
    const void *ms = nullptr, *od = nullptr;
    ltmCheck(ctx, ltm_cloud_device_ptr(ctx, mine->h, &ms), "ltm_cloud_device_ptr");
    ltmCheck(ctx, ltm_cloud_device_ptr(ctx, out->h, &od), "ltm_cloud_device_ptr");
    comm.allGatherV(ctx, ms, mine->size() * sizeof(PointType), const_cast<void*>(od), sizes);
    ltmCheck(ctx, ltm_synchronize(ctx), "ltm_synchronize");
    return out;
}
CloudPtr Session::octreeDownsampling(const CloudPtr& src, float leaf) const
{
    ltm_ctx* ctx = dev_->ctx;
    ltm_cloud h = 0;
    // Multi-GPU: the (replicated) input's Morton key space is cut into `world` contiguous ranges of equal point count; every rank
    // sorts and reduces its own range and the centroid lists, all-gathered in rank order, ARE the single-GPU output
    // (ltm_voxel_centroid_shard).  Small clouds stay replicated: the exchange would cost more than the sort.
    if (world() > 1 && src->size() >= kVoxelShardMin) {
        ltm_cloud piece = 0;
        ltmCheck(ctx, ltm_voxel_centroid_shard(ctx, src->h, leaf, (uint32_t)rank(), (uint32_t)world(), &piece), "ltm_voxel_centroid_shard");
        CloudPtr mine = wrap(piece);
        const size_t n = mine->size();
        std::vector<uint64_t> sizes;
        comm()->allGatherU64(ctx, n, sizes);
        size_t total = 0;
        for (uint64_t& v : sizes) { total += v; v *= sizeof(PointType); }
        ltmCheck(ctx, ltm_cloud_alloc(ctx, total, &h), "ltm_cloud_alloc");
        CloudPtr out = wrap(h);
        const void *s = nullptr, *d = nullptr;
        ltmCheck(ctx, ltm_cloud_device_ptr(ctx, piece, &s), "ltm_cloud_device_ptr");
        ltmCheck(ctx, ltm_cloud_device_ptr(ctx, h, &d), "ltm_cloud_device_ptr");
        comm()->allGatherV(ctx, s, n * sizeof(PointType), const_cast<void*>(d), sizes);
        ltmCheck(ctx, ltm_synchronize(ctx), "ltm_synchronize");      // `mine` is released on return
        return out;
    }
    ltmCheck(ctx, ltm_voxel_centroid(ctx, src->h, leaf, &h), "ltm_voxel_centroid");
    return wrap(h);
}
std::vector<CloudPtr> Session::octreeDownsamplingBatch(const std::vector<CloudPtr>& src, float leaf) const
{
    std::vector<CloudPtr> out(src.size());
    std::vector<size_t> batch;
    for (size_t i = 0; i < src.size(); ++i) {
        if (world() > 1 && src[i]->size() >= kVoxelShardMin) out[i] = octreeDownsampling(src[i], leaf);      // sharded + all-gathered
        else batch.push_back(i);
    }
    if (!batch.empty()) {
        std::vector<ltm_cloud> in(batch.size()), res(batch.size(), 0);
        std::vector<float> leafs(batch.size(), leaf);
        for (size_t k = 0; k < batch.size(); ++k) in[k] = src[batch[k]]->h;
        ltmCheck(dev_->ctx, ltm_voxel_centroid_batch(dev_->ctx, in.size(), in.data(), leafs.data(), res.data()), "ltm_voxel_centroid_batch");
        for (size_t k = 0; k < batch.size(); ++k) out[batch[k]] = wrap(res[k]);
    }
    return out;
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
    ltm_scanset h = 0;      // Session.cpp:354: the loop over keyframes, here over this rank's block
    ltmCheck(dev_->ctx, ltm_reproject(dev_->ctx, _map->h, poses_h_, kf_begin_, kf_end_, kReprojectionAlpha, &h), "ltm_reproject");
    _vec_to_store = wrap_shard(h);
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
    // the three operands come from reprojections of the same keyframe block, so on a multi-GPU run this is shard-local
    const bool shard = keyframe_scans_updated_->shard;
    keyframe_scans_updated_ = shard ? wrap_shard(voxelised) : wrap_scans(voxelised);
}

static std::pair<CloudPtr, CloudPtr> knnOneScan(const Session& s, const CloudPtr& target, const ScansPtr& scans, int idx)
{
    if (!target) throw std::runtime_error("no kNN target map yet: call extract*ViaKnnDiff first");
    if (scans->shard) throw std::runtime_error("the per-scan kNN wrappers address whole scan sets (single-GPU runs)");
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
    ltm_poses ph = 0; size_t kb = 0, ke = 0;      // Session.cpp:408: the OpenMP loop over keyframes, here over this rank's block
    stageArgs(keyframe_scans_static_projected_, &ph, &kb, &ke);
    ltmCheck(dev_->ctx, ltm_knn_partition(dev_->ctx, _target_map->h, keyframe_scans_static_projected_->h, ph, kb, ke,
                                          kNumKnnPointsToCompare, kScanKnnAndMapKnnAvgDiffThreshold, &co, &di), "ltm_knn_partition");
    scans_knn_coexist_ = wrap_shard(co); scans_knn_diff_ = wrap_shard(di);
}

void Session::extractHighDynPointsViaKnnDiff(const CloudPtr& _target_map)
{
    knn_target_map_ = _target_map;
    ltm_scanset di = 0;
    ltm_poses ph = 0; size_t kb = 0, ke = 0;      // Session.cpp:491
    stageArgs(keyframe_scans_, &ph, &kb, &ke);
    ltmCheck(dev_->ctx, ltm_knn_partition(dev_->ctx, _target_map->h, keyframe_scans_->h, ph, kb, ke,
                                          kNumKnnPointsToCompare, kScanKnnAndMapKnnAvgDiffThreshold, nullptr, &di), "ltm_knn_partition");
    keyframe_scans_dynamic_ = wrap_shard(di);
}

void Session::constructGlobalNDMap() { map_global_nd_ = mergeVoxel(scans_knn_diff_, 0.05f); }
void Session::constructGlobalPDMap()
{
    map_global_pd_ = mergeVoxel(scans_knn_diff_, 0.05f);
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
