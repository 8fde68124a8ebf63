// Removerter.cpp -- mirror of ltremovert/src/Removerter.cpp over the C ABI (include/ltm.h).
#include "removert/Removerter.h"

#include <algorithm>
#include <cstdlib>
#include <sstream>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <stdexcept>
#include <thread>

namespace ltremovert
{

#define LTM_INFO(msg) do { if (!logQuiet()) std::cout << "\033[1;32m" << msg << "\033[0m" << std::endl; } while (0)

static std::shared_ptr<Device> make_device() { RosParamServer p; return std::make_shared<Device>(p); }

Removerter::Removerter() : Removerter(make_device()) {}

Removerter::Removerter(std::shared_ptr<Device> dev) : dev_(std::move(dev)), central_sess_(dev_), query_sess_(dev_)
{
    // Removerter.cpp:26-50 : output directory protocol
    if (save_pcd_directory_.substr(save_pcd_directory_.size() - 1, 1) != std::string("/")) save_pcd_directory_ = save_pcd_directory_ + "/";
    fsmkdir(save_pcd_directory_);
    updated_scans_save_dir_ = save_pcd_directory_ + "scans_updated";                fsmkdir(updated_scans_save_dir_);
    updated_strong_scans_save_dir_ = save_pcd_directory_ + "scans_updated_strong";  fsmkdir(updated_strong_scans_save_dir_);
    pd_scans_save_dir = save_pcd_directory_ + "scans_pd";                            fsmkdir(pd_scans_save_dir);
    strong_pd_scans_save_dir = save_pcd_directory_ + "scans_pd_strong";              fsmkdir(strong_pd_scans_save_dir);
    strong_nd_scans_save_dir = save_pcd_directory_ + "scans_nd_strong";              fsmkdir(strong_nd_scans_save_dir);
    central_map_static_save_dir_ = save_pcd_directory_ + "map_static";               fsmkdir(central_map_static_save_dir_);
    central_map_dynamic_save_dir_ = save_pcd_directory_ + "map_dynamic";             fsmkdir(central_map_dynamic_save_dir_);
}

// After a normal run() there is nothing left to do here.  If run() was left by an exception, writer tasks may still be reading
// fetched data: let them finish and hand the tickets back before the clouds they read from (fetches_) and the context go away.
Removerter::~Removerter()
{
    try { finishOutputs(); } catch (...) {}
}

void Removerter::finishOutputs()
{
    if (lane_) lane_->finishOutputs();      // its files may still be read out of clouds that were handed over to this context
    if (!writer_) return;
    std::exception_ptr e;
    try { writer_->drain(); } catch (...) { e = std::current_exception(); }
    for (PendingFetch& f : fetches_)
        if (ltm_fetch* t = f.ticket->exchange(nullptr)) (void)ltm_fetch_release(dev_->ctx, t);
    fetches_.clear();
    if (e) std::rethrow_exception(e);
}

void Removerter::saveMap(const std::string& file, const CloudPtr& cloud, bool octree_layout)
{
    if (bench_mode_) return;
    if (dev_->rank() != 0) return;      // maps are replicated on every rank: rank 0 writes them
    if (gpu_async_io_) {
        // D2H on the copy stream into pinned memory + the file write on a writer thread; the GPU goes on with the next stage
        if (!writer_) writer_.reset(new AsyncWriter((unsigned)std::max(1, kNumOmpCores)));
        ltm_fetch* t = nullptr;
        ltm_ctx* ctx = dev_->ctx;
        if (gpu_fetch_chunked_) {
            // through the library's ring of pinned chunks: the file is written piece by piece as the copier thread delivers them
            ltmCheck(ctx, ltm_cloud_fetch_chunks_begin(ctx, cloud->h, &t), "ltm_cloud_fetch_chunks_begin");
            auto ticket = std::make_shared<std::atomic<ltm_fetch*>>(t);
            fetches_.push_back(PendingFetch{ticket, cloud, nullptr});
            writer_->submit([t, ticket, ctx, file, octree_layout] {
                std::exception_ptr failure;
                std::ofstream f;
                try {
                    size_t n = 0;
                    std::string err;
                    if (ltm_fetch_info(t, &n, nullptr, nullptr) != LTM_OK) throw std::runtime_error("ltm_fetch_info failed for " + file);
                    if (!openPCDFileBinary(file, n, octree_layout, &f, &err)) throw std::runtime_error(err);
                } catch (...) { failure = std::current_exception(); }
                for (;;) {      // every chunk is taken and handed back even after a failure: the ring must not run dry for the tickets behind
                    const void* pts = nullptr; size_t cnt = 0;
                    const int rc = ltm_fetch_next_chunk(t, &pts, nullptr, &cnt, nullptr, nullptr);
                    if (rc <= 0) { if (rc < 0 && !failure) failure = std::make_exception_ptr(std::runtime_error("ltm_fetch_next_chunk failed for " + file)); break; }
                    if (!failure) f.write(static_cast<const char*>(pts), (std::streamsize)(cnt * sizeof(PointType)));
                    (void)ltm_fetch_chunk_done(t, pts);
                }
                if (!failure && !f) failure = std::make_exception_ptr(std::runtime_error(file + ": write failed"));
                if (failure) std::rethrow_exception(failure);
                if (ltm_fetch* mine = ticket->exchange(nullptr)) (void)ltm_fetch_release(ctx, mine);
            });
            return;
        }
        ltmCheck(dev_->ctx, ltm_cloud_fetch_begin(dev_->ctx, cloud->h, &t), "ltm_cloud_fetch_begin");
        auto ticket = std::make_shared<std::atomic<ltm_fetch*>>(t);
        fetches_.push_back(PendingFetch{ticket, cloud, nullptr});
        writer_->submit([t, ticket, ctx, file, octree_layout] {
            const void* pts = nullptr; size_t n = 0;
            if (ltm_fetch_wait(t, &pts, &n, nullptr, nullptr) != LTM_OK) throw std::runtime_error("ltm_fetch_wait failed for " + file);
            std::string err;
            if (!savePCDFileBinary(file, static_cast<const PointType*>(pts), n, octree_layout, &err)) throw std::runtime_error(err);
            if (ltm_fetch* mine = ticket->exchange(nullptr)) (void)ltm_fetch_release(ctx, mine);      // the pinned buffer goes back to the pool now
        });
        return;
    }
    std::string err;
    if (!savePCDFileBinary(file, cloud->download(), octree_layout, &err)) throw std::runtime_error(err);
}

void Removerter::loadSessionInfo(void)
{
    central_sess_.loadSessionInfo("Central", central_sess_scan_dir_, central_sess_pose_path_);
    query_sess_.loadSessionInfo("Query", query_sess_scan_dir_, query_sess_pose_path_);
    central_sess_.setDownsampleSize(kDownsampleVoxelSize);
    query_sess_.setDownsampleSize(kDownsampleVoxelSize);
}

void Removerter::parseKeyframes(void)
{
    central_sess_.parseKeyframes({start_idx_, end_idx_}, keyframe_gap_);
    query_sess_.parseKeyframesInROI(central_sess_.keyframe_poses_, keyframe_gap_);
}

void Removerter::loadKeyframes(void) { central_sess_.loadKeyframes(); query_sess_.loadKeyframes(); }

void Removerter::precleaningKeyframes(float _radius)
{
    central_sess_.precleaningKeyframes(_radius);
    query_sess_.precleaningKeyframes(_radius);
}

void Removerter::makeGlobalMap(Session& _sess)
{
    _sess.mergeScansWithinGlobalCoord();
    LTM_INFO(" Map pointcloud (having redundant points) have: " << _sess.map_global_orig_->size() << " points.");
    LTM_INFO(" Downsampling leaf size is " << kDownsampleVoxelSize << " m.");
    _sess.map_global_curr_ = _sess.octreeDownsampling(_sess.map_global_orig_, kDownsampleVoxelSize);
    _sess.map_global_orig_.reset();
    if (kFlagSaveMapPointcloud) {
        const std::string name = save_pcd_directory_ + "OriginalNoisy" + _sess.sess_type_ + "MapGlobal.pcd";
        saveMap(name, _sess.map_global_curr_);
        LTM_INFO(" The original pointcloud is saved (global coord): " << name);
    }
}
// both sessions' merges first, then their two grids as ONE batch (two host round trips instead of four) -- the same sequence of C-ABI calls as
// lt-mapper_amd/removerter.py's makeGlobalMap (tests/test_gpu_cli.py compares the two hosts' kernel-class launch counts)
void Removerter::makeGlobalMap(void)
{
    if (std::shared_ptr<Comm> group = dev_->comm ? dev_->comm->sessionGroup() : nullptr) {
        // Session groups (Comm.h): this rank's group merges + grids ITS session only; the group stays entered until removeHighDynamicPoints()
        // has finished the session's Step 1 and the rank pairs have swapped the results.  The map files are written there as well.
        Session& mine = dev_->rank() % 2 == 0 ? central_sess_ : query_sess_;
        mine.enterSessionGroup(group);
        mine.mergeScansWithinGlobalCoord();
        LTM_INFO(" Map pointcloud (having redundant points) have: " << mine.map_global_orig_->size() << " points.");
        LTM_INFO(" Downsampling leaf size is " << kDownsampleVoxelSize << " m.");
        mine.map_global_curr_ = mine.octreeDownsampling(mine.map_global_orig_, kDownsampleVoxelSize);
        mine.map_global_orig_.reset();
        group_orig_noisy_ = mine.map_global_curr_;
        return;
    }
    Session* ss[2] = {&central_sess_, &query_sess_};
    for (Session* s : ss) {
        s->mergeScansWithinGlobalCoord();
        LTM_INFO(" Map pointcloud (having redundant points) have: " << s->map_global_orig_->size() << " points.");
        LTM_INFO(" Downsampling leaf size is " << kDownsampleVoxelSize << " m.");
    }
    auto ds = central_sess_.octreeDownsamplingBatch({central_sess_.map_global_orig_, query_sess_.map_global_orig_}, kDownsampleVoxelSize);
    for (int i = 0; i < 2; ++i) {
        Session& s = *ss[i];
        s.map_global_curr_ = ds[(size_t)i];
        s.map_global_orig_.reset();
        if (kFlagSaveMapPointcloud) {
            const std::string name = save_pcd_directory_ + "OriginalNoisy" + s.sess_type_ + "MapGlobal.pcd";
            saveMap(name, s.map_global_curr_);
            LTM_INFO(" The original pointcloud is saved (global coord): " << name);
        }
    }
}

// Removerter.cpp:801-828 / :771-799 / :740-768 : one visibility vote + index-ascending split
std::pair<CloudPtr, CloudPtr> Removerter::votePartition(const Session& tgt, const CloudPtr& map, const ScansPtr& scans, const Session& src, float res, int mode)
{
    ltm_ctx* ctx = tgt.dev_->ctx;
    if (gpu_viz_every_ > 0 && map->size() > 0 && dev_->world() == 1) {     // Removerter.cpp:580-585, only when asked for: it serialises the pass
        int rows = 0, cols = 0;
        ltm_rimg_size(kVFOV, kHFOV, res, &rows, &cols);
        std::vector<uint8_t> img[4];
        for (auto& v : img) v.resize((size_t)rows * cols * 3);
        size_t n_kf = 0, n_pts = 0;
        ltmCheck(ctx, ltm_scanset_info(ctx, scans->h, &n_kf, &n_pts), "ltm_scanset_info");
        for (size_t kf = 0; kf < n_kf; kf += (size_t)gpu_viz_every_) {
            ltmCheck(ctx, ltm_debug_viz_images(ctx, map->h, scans->h, src.poses_h_, kf, res, mode, kRangeColorAxis.first, kRangeColorAxis.second,
                                               kRangeColorAxisForDiff.first, kRangeColorAxisForDiff.second, img[0].data(), img[1].data(),
                                               img[2].data(), img[3].data()),
                     "ltm_debug_viz_images");
            publishDebugImages(viz_pass_, kf, rows, cols, img[0].data(), img[1].data(), img[2].data(), img[3].data());
        }
    }
    ++viz_pass_;
    ltm_cloud kept = 0, flagged = 0;
    if (src.world() == 1) {      // the ranks that share the SOURCE session's keyframes: the world, or the session's rank group during Step 1
        ltmCheck(ctx, ltm_visibility_partition(ctx, map->h, scans->h, src.poses_h_, res, 0.1f, mode, &kept, &flagged, nullptr),
                 "ltm_visibility_partition");
        return {tgt.wrap(kept), tgt.wrap(flagged)};
    }
    // Multi-GPU (Removerter.cpp:555, the loop over the source keyframes): this rank votes with its block of keyframes into an
    // M-byte mask, the masks are united by an element-wise MAX across the ranks (the std::set union of :589-590), and every rank
    // performs the same deterministic index-ascending split -- no broadcast of the result is needed.
    const size_t M = map->size();
    void* labels = nullptr;
    ltmCheck(ctx, ltm_buffer_alloc(ctx, std::max<size_t>(M, 1), &labels), "ltm_buffer_alloc");
    ltmCheck(ctx, ltm_buffer_fill(ctx, labels, 0, std::max<size_t>(M, 1)), "ltm_buffer_fill");
    ltm_poses ph = 0; size_t kb = 0, ke = 0;
    src.stageArgs(scans, &ph, &kb, &ke);
    if (M) ltmCheck(ctx, ltm_visibility_vote(ctx, map->h, scans->h, ph, kb, ke, res, 0.1f, mode, static_cast<uint8_t*>(labels)), "ltm_visibility_vote");
    src.comm()->allReduceMaxU8(ctx, labels, M);
    ltmCheck(ctx, ltm_partition_by_labels(ctx, map->h, static_cast<const uint8_t*>(labels), &kept, &flagged), "ltm_partition_by_labels");
    ltmCheck(ctx, ltm_buffer_free(ctx, labels), "ltm_buffer_free");
    return {tgt.wrap(kept), tgt.wrap(flagged)};
}

void Removerter::publishDebugImages(int pass, size_t kf, int rows, int cols, const uint8_t* scan_bgr, const uint8_t* map_bgr,
                                    const uint8_t* diff_bgr, const uint8_t* ptidx_bgr)
{
    const std::string dir = save_pcd_directory_ + "viz/";
    fsmkdir(dir);
    const uint8_t* src[4] = {scan_bgr, map_bgr, diff_bgr, ptidx_bgr};
    const char* name[4] = {"scan", "map", "diff", "ptidx"};
    std::vector<uint8_t> rgb((size_t)rows * cols * 3);
    for (int i = 0; i < 4; ++i) {
        for (size_t p = 0; p < (size_t)rows * cols; ++p) { rgb[3 * p] = src[i][3 * p + 2]; rgb[3 * p + 1] = src[i][3 * p + 1]; rgb[3 * p + 2] = src[i][3 * p]; }
        char file[64];
        std::snprintf(file, sizeof file, "%03d_%06zu_%s.ppm", pass, kf, name[i]);
        std::ofstream o(dir + file, std::ios::binary);
        o << "P6\n" << cols << " " << rows << "\n255\n";
        o.write(reinterpret_cast<const char*>(rgb.data()), (std::streamsize)rgb.size());
    }
}

std::pair<CloudPtr, CloudPtr> Removerter::partitionCurrentMap(const Session& _target_sess, const Session& _source_sess, float _res_alpha)
{
    curr_res_alpha_ = _res_alpha;
    curr_rimg_shape_ = resetRimgSize(kFOV, _res_alpha);
    LTM_INFO(" with resolution: x" << _res_alpha << " (" << 1.0 / _res_alpha << " deg/pixel)");
    LTM_INFO(" -- The range image size is: [" << curr_rimg_shape_.first << ", " << curr_rimg_shape_.second << "].");
    LTM_INFO(" -- The number of " << _target_sess.sess_type_ << " map points: " << _target_sess.map_global_curr_->size());
    LTM_INFO(" -- ... starts cleaning ... ");
    auto r = votePartition(_target_sess, _target_sess.map_global_curr_, _source_sess.keyframe_scans_, _source_sess, _res_alpha, 0);
    LTM_INFO(" -- The number of dynamic points: " << r.second->size());
    LTM_INFO(" -- The number of static points: " << r.first->size());
    return r;
}
std::pair<CloudPtr, CloudPtr> Removerter::partitionCurrentMapForND(const Session& _target_sess, const Session& _source_sess, float _res_alpha)
{
    curr_res_alpha_ = _res_alpha;
    curr_rimg_shape_ = resetRimgSize(kFOV, _res_alpha);
    LTM_INFO(" -- The number of " << _target_sess.sess_type_ << " ND map points: " << _target_sess.map_global_nd_->size());
    LTM_INFO(" -- ... starts to clean ambiguous ND ... ");
    return votePartition(_target_sess, _target_sess.map_global_nd_, _source_sess.keyframe_scans_static_projected_, _source_sess, _res_alpha, 1);
}
std::pair<CloudPtr, CloudPtr> Removerter::partitionCurrentMapForPD(const Session& _target_sess, const Session& _source_sess, float _res_alpha)
{
    curr_res_alpha_ = _res_alpha;
    curr_rimg_shape_ = resetRimgSize(kFOV, _res_alpha);
    LTM_INFO(" -- The number of " << _target_sess.sess_type_ << " PD map points: " << _target_sess.map_global_pd_->size());
    LTM_INFO(" -- ... starts to clean non-volume-extending PD ... ");
    return votePartition(_target_sess, _target_sess.map_global_pd_, _source_sess.keyframe_scans_static_projected_, _source_sess, _res_alpha, 0);
}

static CloudPtr append(const Session& s, const CloudPtr& a, const CloudPtr& b) { return a ? s.concat({a, b}) : s.concat({b}); }

void Removerter::removeOnce(const Session& t, const Session& src, float _res_alpha)      // Removerter.cpp:882-905
{
    LTM_INFO("\nSelf-removing starts ");
    auto [static_tt, dynamic_tt] = partitionCurrentMap(t, src, _res_alpha);
    auto ds = t.octreeDownsamplingBatch({static_tt, append(t, t.map_global_curr_dynamic_, dynamic_tt)}, 0.05f);    // :896, :903 -- independent
    t.map_global_curr_static_ = ds[0];
    LTM_INFO(" Current Static pointcloud have: " << t.map_global_curr_static_->size() << " points.");
    t.map_global_curr_ = t.map_global_curr_static_;
    t.map_global_curr_dynamic_ = ds[1];
    LTM_INFO(" Current Dynamic pointcloud have: " << t.map_global_curr_dynamic_->size() << " points.");
}

void Removerter::revertOnce(const Session& t, const Session& src, float _res_alpha)      // Removerter.cpp:908-931
{
    LTM_INFO("\nSelf-reverting starts ");
    auto [static_tt, dynamic_tt] = partitionCurrentMap(t, src, _res_alpha);
    auto ds = t.octreeDownsamplingBatch({dynamic_tt, append(t, t.map_global_curr_static_, static_tt)}, 0.05f);     // :921, :928
    t.map_global_curr_dynamic_ = ds[0];
    LTM_INFO(" Current Dynamic pointcloud have: " << t.map_global_curr_dynamic_->size() << " points.");
    t.map_global_curr_ = t.map_global_curr_dynamic_;
    t.map_global_curr_static_ = ds[1];
    LTM_INFO(" Current Static pointcloud have: " << t.map_global_curr_static_->size() << " points.");
}

void Removerter::resetCurrrentMapAsDynamic(const Session& _sess, bool _as_dynamic)       // Removerter.cpp:714-737
{
    _sess.map_global_curr_ = _as_dynamic ? _sess.map_global_curr_dynamic_ : _sess.map_global_curr_static_;
}
void Removerter::resetCurrrentMapAsDynamic(const Session& _sess) { resetCurrrentMapAsDynamic(_sess, true); }
void Removerter::resetCurrrentMapAsStatic(const Session& _sess) { resetCurrrentMapAsDynamic(_sess, false); }

void Removerter::selfRemovert(const Session& _sess, int _repeat = 1)                     // Removerter.cpp:1378-1393
{
    if (_repeat > 0 && !remove_resolution_list_.empty()) {
        // every scan range image the passes below will ask for (res and 0.95 res of each list entry), in ONE pass over the scans
        std::vector<float> alphas(remove_resolution_list_.begin(), remove_resolution_list_.end());
        for (float _res : remove_resolution_list_) alphas.push_back((float)(0.95 * _res));
        ltm_poses ph = 0; size_t kb = 0, ke = 0;
        _sess.stageArgs(_sess.keyframe_scans_, &ph, &kb, &ke);
        ltm_ctx* ctx = _sess.dev_->ctx;
        ltmCheck(ctx, ltm_scanset_prepare_range_images(ctx, _sess.keyframe_scans_->h, kb, ke, alphas.data(), alphas.size()), "ltm_scanset_prepare_range_images");
    }
    for (float _res : remove_resolution_list_) {
        for (int i = 0; i < _repeat; i++) {
            removeOnce(_sess, _sess, _res);
            resetCurrrentMapAsDynamic(_sess);
            revertOnce(_sess, _sess, 0.95 * _res);
            resetCurrrentMapAsStatic(_sess);
            removeOnce(_sess, _sess, _res);
        }
    }
    saveCurrentStaticAndDynamicPointCloudGlobal(_sess, "_MVM");
}

void Removerter::saveCurrentStaticAndDynamicPointCloudGlobal(const Session& _sess, std::string _postfix)   // Removerter.cpp:318-338
{
    if (!kFlagSaveMapPointcloud) return;
    const std::string r = std::to_string(curr_res_alpha_);
    saveMap(central_map_dynamic_save_dir_ + "/" + _sess.sess_type_ + "DynamicMapMapsideGlobal" + _postfix + "ResX" + r + ".pcd", _sess.map_global_curr_dynamic_);
    saveMap(central_map_static_save_dir_ + "/" + _sess.sess_type_ + "StaticMapMapsideGlobalResX" + _postfix + "ResX" + r + ".pcd", _sess.map_global_curr_static_);   // doubled "ResX": sic (:335)
}

// Step 1 with session groups: the remove / revert passes and the HD kNN map of THIS rank's session on its group, then the swap of the finished
// maps with the partner rank of the other group -- every rank ends with what the one-after-the-other order leaves behind (the scans of the
// HD kNN, only ever merged into the HD map, stay with the group that made them)
void Removerter::removeHighDynamicPointsOnSessionGroups(void)
{
    const bool central = dev_->rank() % 2 == 0;
    Session& mine = central ? central_sess_ : query_sess_;
    Session& other = central ? query_sess_ : central_sess_;
    const bool self = gpu_use_self_removert_ && !remove_resolution_list_.empty();
    if (kFlagSaveMapPointcloud) saveMap(save_pcd_directory_ + "OriginalNoisy" + mine.sess_type_ + "MapGlobal.pcd", group_orig_noisy_);
    if (self) selfRemovert(mine, repeat_removert_iter_);
    else removeOnce(mine, mine, 2.5);
    std::vector<CloudPtr> give = {group_orig_noisy_, mine.map_global_curr_static_, mine.map_global_curr_dynamic_};
    if (!gpu_skip_hd_knn_) {
        mine.extractHighDynPointsViaKnnDiff(mine.map_global_curr_static_);
        give.push_back(mine.mergeVoxel(mine.keyframe_scans_dynamic_, 0.05f));
    }
    mine.leaveSessionGroup();
    group_orig_noisy_.reset();

    ltm_ctx* ctx = dev_->ctx;
    Comm& comm = *dev_->comm;
    std::vector<uint64_t> n_mine, n_theirs;
    for (const CloudPtr& c : give) n_mine.push_back(c->size());
    comm.swapU64WithPeer(ctx, n_mine, n_theirs);
    size_t total = 0;
    for (uint64_t v : n_theirs) total += v;
    CloudPtr send = mine.concat(give);
    ltm_cloud hr = 0;
    ltmCheck(ctx, ltm_cloud_alloc(ctx, total, &hr), "ltm_cloud_alloc");
    CloudPtr recv = other.wrap(hr);
    const void *sp = nullptr, *rp = nullptr;
    ltmCheck(ctx, ltm_cloud_device_ptr(ctx, send->h, &sp), "ltm_cloud_device_ptr");
    ltmCheck(ctx, ltm_cloud_device_ptr(ctx, recv->h, &rp), "ltm_cloud_device_ptr");
    comm.swapWithPeer(ctx, sp, send->size() * sizeof(PointType), const_cast<void*>(rp), total * sizeof(PointType));
    ltmCheck(ctx, ltm_synchronize(ctx), "ltm_synchronize");
    std::vector<CloudPtr> got;
    size_t at = 0;
    for (uint64_t n : n_theirs) {
        ltm_cloud h = 0;
        ltmCheck(ctx, ltm_cloud_from_device(ctx, static_cast<const PointType*>(rp) + at, (size_t)n, &h), "ltm_cloud_from_device");
        got.push_back(other.wrap(h));
        at += (size_t)n;
    }
    other.map_global_curr_static_ = got[1];
    other.map_global_curr_dynamic_ = got[2];
    other.map_global_curr_ = other.map_global_curr_static_;
    // the files the other group's chain would have written on a rank that writes (rank 0 belongs to the central group)
    if (kFlagSaveMapPointcloud) saveMap(save_pcd_directory_ + "OriginalNoisy" + other.sess_type_ + "MapGlobal.pcd", got[0]);
    if (self) saveCurrentStaticAndDynamicPointCloudGlobal(other, "_MVM");
    if (gpu_skip_hd_knn_) return;
    const CloudPtr& hd_c = central ? give[3] : got[3];
    const CloudPtr& hd_q = central ? got[3] : give[3];
    saveMap(save_pcd_directory_ + "central_sess_high_dyn.pcd", hd_c);
    saveMap(save_pcd_directory_ + "query_sess_high_dyn.pcd", hd_q);
    LTM_INFO(" high dynamic maps are saved. ");
}

void Removerter::removeHighDynamicPoints(void)                                     // Removerter.cpp:1580-1604
{
    if (group_orig_noisy_) return removeHighDynamicPointsOnSessionGroups();       // makeGlobalMap() entered this rank's session group
    if (gpu_use_self_removert_ && !remove_resolution_list_.empty()) {
        selfRemovert(central_sess_, repeat_removert_iter_);
        selfRemovert(query_sess_, repeat_removert_iter_);
    } else {
        removeOnce(central_sess_, central_sess_, 2.5);
        removeOnce(query_sess_, query_sess_, 2.5);
    }
    if (gpu_skip_hd_knn_) return;
    central_sess_.extractHighDynPointsViaKnnDiff(central_sess_.map_global_curr_static_);
    query_sess_.extractHighDynPointsViaKnnDiff(query_sess_.map_global_curr_static_);
    // merge + grid of the two sessions' dynamic scans: one batch on a single GPU; keyframe-sharded, the scans are rank-local and each merge is a
    // key-range exchange (Session::mergeVoxel) -- the same split as lt-mapper_amd/removerter.py's mergeVoxelBatch
    std::vector<CloudPtr> hd;
    if (dev_->world() > 1) hd = {central_sess_.mergeVoxel(central_sess_.keyframe_scans_dynamic_, 0.05f), query_sess_.mergeVoxel(query_sess_.keyframe_scans_dynamic_, 0.05f)};
    else hd = central_sess_.octreeDownsamplingBatch({central_sess_.mergeScansToGlobal(central_sess_.keyframe_scans_dynamic_),
                                                    query_sess_.mergeScansToGlobal(query_sess_.keyframe_scans_dynamic_)}, 0.05f);
    saveMap(save_pcd_directory_ + "central_sess_high_dyn.pcd", hd[0]);
    saveMap(save_pcd_directory_ + "query_sess_high_dyn.pcd", hd[1]);
    LTM_INFO(" high dynamic maps are saved. ");
}

void Removerter::iremoveOnceForND(const Session& t, const Session& src, float _res_alpha)   // Removerter.cpp:831-854
{
    LTM_INFO("\nIdentifying Strong/Weak ND points starts ");
    auto [static_tt, dynamic_tt] = partitionCurrentMapForND(t, src, _res_alpha);
    auto ds = t.octreeDownsamplingBatch({static_tt, append(t, t.map_global_nd_weak_, dynamic_tt)}, 0.05f);
    t.map_global_nd_strong_ = ds[0];
    t.map_global_nd_ = t.map_global_nd_strong_;
    t.map_global_nd_weak_ = ds[1];
}
void Removerter::removeOnceForPD(const Session& t, const Session& src, float _res_alpha)    // Removerter.cpp:856-880
{
    LTM_INFO("\nIdentifying Strong/Weak PD points starts ");
    auto [static_tt, dynamic_tt] = partitionCurrentMapForPD(t, src, _res_alpha);
    auto ds = t.octreeDownsamplingBatch({static_tt, append(t, t.map_global_pd_weak_, dynamic_tt)}, 0.05f);
    t.map_global_pd_strong_ = ds[0];
    t.map_global_pd_ = t.map_global_pd_strong_;
    t.map_global_pd_weak_ = ds[1];
}
void Removerter::filterStrongPD(Session& a, Session& b) { const float res = 2.5; for (int i = 0; i < 3; ++i) removeOnceForPD(a, b, res); }   // :1395-1401
void Removerter::filterStrongND(Session& a, Session& b) { const float res = 2.5; for (int i = 0; i < 3; ++i) iremoveOnceForND(a, b, res); }  // :1403-1411

void Removerter::detectLowDynamicPoints(void)                                      // Removerter.cpp:1413-1481
{
    LTM_INFO(" parse low dynamic diff via knn: " << central_sess_.sess_type_ << " to " << query_sess_.sess_type_);
    central_sess_.extractLowDynPointsViaKnnDiff(query_sess_.map_global_curr_static_);
    LTM_INFO(" parse low dynamic diff via knn: " << query_sess_.sess_type_ << " to " << central_sess_.sess_type_);
    query_sess_.extractLowDynPointsViaKnnDiff(central_sess_.map_global_curr_static_);

    central_sess_.constructGlobalNDMap();
    filterStrongND(central_sess_, query_sess_);
    central_sess_.removeWeakNDMapPointsHavingStrongNDInNear();

    query_sess_.constructGlobalPDMap();
    filterStrongPD(query_sess_, central_sess_);
    query_sess_.revertStrongPDMapPointsHavingWeakPDInNear();

    central_sess_.map_global_pd_ = query_sess_.map_global_pd_;
    central_sess_.map_global_pd_orig_ = query_sess_.map_global_pd_orig_;
    central_sess_.map_global_pd_strong_ = query_sess_.map_global_pd_strong_;

    // :1443-1480 merged maps "for visual debug" (they also re-voxelise state that Step 3 reads)
    Session& C = central_sess_; Session& Q = query_sess_;
    const bool has_strong_nd = C.map_global_nd_strong_->size() != 0;
    std::vector<CloudPtr> ds;
    if (dev_->world() > 1) {
        // keyframe-sharded: the four merges of rank-local scans are key-range exchanges, the replicated maps one batch
        ds = {Q.mergeVoxel(Q.scans_knn_coexist_, 0.05f), C.mergeVoxel(C.scans_knn_coexist_, 0.05f), Q.mergeVoxel(Q.scans_knn_diff_, 0.05f), C.mergeVoxel(C.scans_knn_diff_, 0.05f)};
        std::vector<CloudPtr> rest = {C.map_global_nd_weak_, Q.map_global_pd_strong_, Q.map_global_pd_weak_};
        if (has_strong_nd) rest.push_back(C.map_global_nd_strong_);
        for (const CloudPtr& c : C.octreeDownsamplingBatch(rest, 0.05f)) ds.push_back(c);
    } else {
        std::vector<CloudPtr> ins = {Q.mergeScansToGlobal(Q.scans_knn_coexist_), C.mergeScansToGlobal(C.scans_knn_coexist_),
                                     Q.mergeScansToGlobal(Q.scans_knn_diff_), C.mergeScansToGlobal(C.scans_knn_diff_),
                                     C.map_global_nd_weak_, Q.map_global_pd_strong_, Q.map_global_pd_weak_};
        if (has_strong_nd) ins.push_back(C.map_global_nd_strong_);
        ds = C.octreeDownsamplingBatch(ins, 0.05f);        // the eight independent voxel grids of :1445-1476 as one batch
    }
    union_q_ = ds[0];
    saveMap(save_pcd_directory_ + "union_map_queryside.pcd", union_q_);
    union_c_ = ds[1];
    saveMap(save_pcd_directory_ + "union_map_centralside.pcd", union_c_);
    saveMap(save_pcd_directory_ + "pd_map.pcd", ds[2]);
    saveMap(save_pcd_directory_ + "nd_map.pcd", ds[3]);
    LTM_INFO(" Union, PD, and ND map saved ");
    if (has_strong_nd) {
        C.map_global_nd_strong_ = ds[7];
        saveMap(save_pcd_directory_ + "strong_nd_map.pcd", C.map_global_nd_strong_);
    }
    C.map_global_nd_weak_ = ds[4];
    saveMap(save_pcd_directory_ + "weak_nd_map.pcd", C.map_global_nd_weak_);
    LTM_INFO(" Strong/Weak ND map saved ");
    Q.map_global_pd_strong_ = ds[5];
    saveMap(save_pcd_directory_ + "strong_pd_map.pcd", Q.map_global_pd_strong_);
    Q.map_global_pd_weak_ = ds[6];
    saveMap(save_pcd_directory_ + "weak_pd_map.pcd", Q.map_global_pd_weak_);
    LTM_INFO(" Strong/Weak PD map saved ");
}

void Removerter::updateCurrentMap(void)                                            // Removerter.cpp:1483-1524
{
    Session& C = central_sess_;
    // the union maps of :1489-1493 are recomputed from unchanged inputs in the reference: identical to :1445-1451
    CloudPtr updated = C.concat({union_q_, union_c_, C.map_global_nd_weak_});
    LTM_INFO(" -- The number of map points (updating ...): " << updated->size());
    auto ds = C.octreeDownsamplingBatch({C.concat({updated, C.map_global_pd_strong_}), C.concat({updated, C.map_global_pd_orig_})}, 0.05f);
    C.map_global_updated_strong_ = ds[0];
    LTM_INFO(" -- The number of strong map points (updating ...): " << C.map_global_updated_strong_->size());
    C.map_global_updated_ = ds[1];
    LTM_INFO(" -- The number of map points (updating ...): " << C.map_global_updated_->size());
    saveMap(save_pcd_directory_ + "updated_map.pcd", C.map_global_updated_);
    saveMap(save_pcd_directory_ + "updated_map_strong.pcd", C.map_global_updated_strong_);
    LTM_INFO(" -- The updated map is saved ");
}

void Removerter::parseStaticScansViaProjection(Session& _sess)
{
    LTM_INFO(" parse static scans via projection: " << _sess.sess_type_);
    _sess.parseStaticScansViaProjection();
}
void Removerter::parseStaticScansViaProjection(void) { parseStaticScansViaProjection(central_sess_); parseStaticScansViaProjection(query_sess_); }
void Removerter::updateScansScanwise() { updateScansScanwise(central_sess_); }
void Removerter::updateScansScanwise(Session& _sess) { _sess.updateScansScanwise(); LTM_INFO(" final update scans: " << _sess.sess_type_); }
void Removerter::parseUpdatedStaticScansViaProjection() { parseUpdatedStaticScansViaProjection(central_sess_); }
void Removerter::parseUpdatedStaticScansViaProjection(Session& _sess)
{
    _sess.parseUpdatedStaticScansViaProjection();
    _sess.parseUpdatedStrongStaticScansViaProjection();
    LTM_INFO(" parse updated scans via projection: " << _sess.sess_type_);
}
void Removerter::parseLDScansViaProjection() { parseLDScansViaProjection(central_sess_); }
void Removerter::parseLDScansViaProjection(Session& _sess)
{
    _sess.parsePDScansViaProjection();
    _sess.parseStrongPDScansViaProjection();
    _sess.parseWeakNDScansViaProjection();
    if (!_sess.map_global_nd_strong_) { ltm_cloud h = 0; PointType none{}; ltmCheck(dev_->ctx, ltm_cloud_upload(dev_->ctx, &none, 0, 16, &h), "ltm_cloud_upload"); _sess.map_global_nd_strong_ = _sess.wrap(h); }
    _sess.parseStrongNDScansViaProjection();
    LTM_INFO(" parse LD scans via projection: " << _sess.sess_type_);
}

void Removerter::saveAllTypeOfScans() { saveUpdatedScans(central_sess_); saveLDScans(central_sess_); }
void Removerter::saveLDScans(Session& _sess) { savePDScans(_sess); saveStrongPDScans(_sess); saveStrongNDScans(_sess); }
void Removerter::saveUpdatedScans(Session& _sess)
{
    saveScans(_sess, _sess.keyframe_scans_updated_, updated_scans_save_dir_, true);           // went through octreeDownsampling (Session.cpp:375)
    saveScans(_sess, _sess.keyframe_scans_updated_strong_, updated_strong_scans_save_dir_, false);
}
void Removerter::savePDScans(Session& _sess) { saveScans(_sess, _sess.keyframe_scans_pd_, pd_scans_save_dir, false); }
void Removerter::saveStrongPDScans(Session& _sess) { saveScans(_sess, _sess.keyframe_scans_strong_pd_, strong_pd_scans_save_dir, false); }
void Removerter::saveStrongNDScans(Session& _sess) { saveScans(_sess, _sess.keyframe_scans_strong_nd_, strong_nd_scans_save_dir, false); }

void Removerter::saveScans(Session& _sess, const ScansPtr& _scans, std::string _save_dir, bool octree_layout)   // Removerter.cpp:1637-1650
{
    if (bench_mode_) return;
    // one file per keyframe: in a multi-GPU run every rank writes the files of its own keyframe block (no gather needed)
    const size_t first = _scans->shard ? _scans->kb : 0;
    if (!_scans->shard && dev_->rank() != 0) return;
    if (gpu_async_io_) {
        // one asynchronous D2H of the whole scan set; every file is its own writer task on a slice of the pinned buffer
        if (!writer_) writer_.reset(new AsyncWriter((unsigned)std::max(1, kNumOmpCores)));
        ltm_fetch* t = nullptr;
        ltm_ctx* ctx = dev_->ctx;
        const size_t nk = _scans->numKeyframes();
        // chunked: chunks of whole keyframes come out of the library's pinned ring; `pumps` writer tasks take them as they arrive and
        // write the files of their chunk.  (LTM_E_INVALID = a keyframe larger than a chunk: the whole-buffer fetch below instead.)
        if (gpu_fetch_chunked_ && ltm_scanset_fetch_chunks_begin(ctx, _scans->h, &t) == LTM_OK) {
            auto ticket = std::make_shared<std::atomic<ltm_fetch*>>(t);
            fetches_.push_back(PendingFetch{ticket, nullptr, _scans});
            auto names = std::make_shared<std::vector<std::string>>();
            for (size_t idx_scan = 0; idx_scan < nk; ++idx_scan) names->push_back(_save_dir + "/" + _sess.keyframe_names_.at(first + idx_scan));
            const size_t pumps = (size_t)std::max(1, kNumOmpCores);
            auto remaining = std::make_shared<std::atomic<size_t>>(pumps);
            for (size_t p = 0; p < pumps; ++p)
                writer_->submit([t, ticket, remaining, ctx, names, octree_layout] {
                    std::exception_ptr failure;
                    const uint64_t* off = nullptr; size_t n_kf = 0;
                    if (ltm_fetch_info(t, nullptr, &off, &n_kf) != LTM_OK || n_kf != names->size())
                        failure = std::make_exception_ptr(std::runtime_error("ltm_fetch_info failed for " + (names->empty() ? std::string("a scan set") : names->front())));
                    for (;;) {      // chunks are taken and handed back even after a failure (see saveMap)
                        const void* pts = nullptr; size_t first_point = 0, cnt = 0, first_kf = 0, kfs = 0;
                        const int rc = ltm_fetch_next_chunk(t, &pts, &first_point, &cnt, &first_kf, &kfs);
                        if (rc <= 0) { if (rc < 0 && !failure) failure = std::make_exception_ptr(std::runtime_error("ltm_fetch_next_chunk failed")); break; }
                        for (size_t k = first_kf; k < first_kf + kfs && !failure; ++k) {
                            std::string err;
                            if (!savePCDFileBinary((*names)[k], static_cast<const PointType*>(pts) + (off[k] - first_point), (size_t)(off[k + 1] - off[k]), octree_layout, &err))
                                failure = std::make_exception_ptr(std::runtime_error(err));
                        }
                        (void)ltm_fetch_chunk_done(t, pts);
                    }
                    if (failure) std::rethrow_exception(failure);
                    if (remaining->fetch_sub(1) == 1)           // the last pump: the ticket goes back
                        if (ltm_fetch* mine = ticket->exchange(nullptr)) (void)ltm_fetch_release(ctx, mine);
                });
            LTM_INFO(" " << nk << " scans queued for " << _save_dir);
            return;
        }
        ltmCheck(dev_->ctx, ltm_scanset_fetch_begin(dev_->ctx, _scans->h, &t), "ltm_scanset_fetch_begin");
        auto ticket = std::make_shared<std::atomic<ltm_fetch*>>(t);
        fetches_.push_back(PendingFetch{ticket, nullptr, _scans});
        auto remaining = std::make_shared<std::atomic<size_t>>(nk);
        for (size_t idx_scan = 0; idx_scan < nk; ++idx_scan) {
            const std::string file_name = _save_dir + "/" + _sess.keyframe_names_.at(first + idx_scan);
            writer_->submit([t, ticket, remaining, ctx, file_name, idx_scan, octree_layout] {
                const void* pts = nullptr; size_t n = 0; const uint64_t* off = nullptr; size_t n_kf = 0;
                if (ltm_fetch_wait(t, &pts, &n, &off, &n_kf) != LTM_OK || idx_scan >= n_kf) throw std::runtime_error("ltm_fetch_wait failed for " + file_name);
                std::string err;
                if (!savePCDFileBinary(file_name, static_cast<const PointType*>(pts) + off[idx_scan], (size_t)(off[idx_scan + 1] - off[idx_scan]), octree_layout, &err))
                    throw std::runtime_error(err);
                if (remaining->fetch_sub(1) == 1)           // the last file of the set: the pinned buffer goes back to the pool
                    if (ltm_fetch* mine = ticket->exchange(nullptr)) (void)ltm_fetch_release(ctx, mine);
            });
        }
        LTM_INFO(" " << nk << " scans queued for " << _save_dir);
        return;
    }
    const std::vector<Cloud> scans = _scans->download();      // one D2H for the whole scan set
    parallelFor(scans.size(), [&](size_t idx_scan) {           // the files are independent: written from all host cores
        const std::string file_name = _save_dir + "/" + _sess.keyframe_names_.at(first + idx_scan);   // same file name as the input scan
        std::string err;
        if (!savePCDFileBinary(file_name, scans[idx_scan], octree_layout, &err)) throw std::runtime_error(err);
    }, (unsigned)std::max(1, kNumOmpCores));
    LTM_INFO(" " << scans.size() << " scans saved under " << _save_dir);
}

// ------------------------------------------------------------------ fine-grained reference methods (thin wrappers)
// the resolution factor whose image shape is `_rimg_shape` (resetRimgSize inverted; the shapes come from it in the first place)
static float alphaOfShape(float vfov, float hfov, std::pair<int, int> shape)
{
    // any alpha with round(vfov * alpha) == rows and round(hfov * alpha) == cols reproduces the shape (e.g. the revert resolution
    // 0.95 * 2.5 gives 119 x 855, which neither rows / vfov = 2.38 nor an integer ratio reproduces): take the middle of the
    // intersection of the two rounding intervals and verify it with the library's own resetRimgSize
    const double lo = std::max(((double)shape.first - 0.5) / vfov, ((double)shape.second - 0.5) / hfov);
    const double hi = std::min(((double)shape.first + 0.5) / vfov, ((double)shape.second + 0.5) / hfov);
    const float cand[3] = {(float)(0.5 * (lo + hi)), (float)shape.second / hfov, (float)shape.first / vfov};
    for (float alpha : cand) {
        int r = 0, c = 0;
        ltm_rimg_size(vfov, hfov, alpha, &r, &c);
        if (r == shape.first && c == shape.second) return alpha;
    }
    throw std::runtime_error("range image shape is not a resolution of this field of view");
}

Removerter::RangeImage Removerter::scan2RangeImg(const CloudPtr& _scan, const std::pair<float, float> _fov, const std::pair<int, int> _rimg_size)
{
    if (_fov.first != kVFOV || _fov.second != kHFOV) throw std::runtime_error("scan2RangeImg: the field of view is fixed per context (sequence_vfov / sequence_hfov)");
    RangeImage img;
    img.rows = _rimg_size.first; img.cols = _rimg_size.second;
    img.range.resize((size_t)img.rows * img.cols);
    ltmCheck(_scan->ctx, ltm_debug_range_image(_scan->ctx, _scan->h, nullptr, nullptr, alphaOfShape(kVFOV, kHFOV, _rimg_size), img.range.data(), nullptr),
             "ltm_debug_range_image");
    return img;
}

static std::vector<int> votedIndexes(const Session& tgt, const CloudPtr& map, const ScansPtr& scans, const Session& src, float alpha, int mode)
{
    std::vector<uint8_t> labels(map->size());
    ltmCheck(tgt.dev_->ctx, ltm_visibility_partition(tgt.dev_->ctx, map->h, scans->h, src.poses_h_, alpha, 0.1f, mode, nullptr, nullptr, labels.data()),
             "ltm_visibility_partition");
    std::vector<int> idx;                       // ascending and unique, like the std::set round trip of Removerter.cpp:589-590
    for (size_t i = 0; i < labels.size(); ++i) if (labels[i]) idx.push_back((int)i);
    return idx;
}
std::vector<int> Removerter::calcDescrepancyAndParseDynamicPointIdxForEachScan(std::pair<int, int> _rimg_shape)
{
    return calcDescrepancyAndParseDynamicPointIdxForEachScan(central_sess_, central_sess_, _rimg_shape);
}
std::vector<int> Removerter::calcDescrepancyAndParseDynamicPointIdxForEachScan(const Session& _target_sess, const Session& _source_sess, std::pair<int, int> _rimg_shape)
{
    return votedIndexes(_target_sess, _target_sess.map_global_curr_, _source_sess.keyframe_scans_, _source_sess, alphaOfShape(kVFOV, kHFOV, _rimg_shape), 0);
}
std::vector<int> Removerter::calcDescrepancyAndParseDynamicPointIdxForEachScanForND(const Session& _target_sess, const Session& _source_sess, std::pair<int, int> _rimg_shape)
{
    return votedIndexes(_target_sess, _target_sess.map_global_nd_, _source_sess.keyframe_scans_static_projected_, _source_sess, alphaOfShape(kVFOV, kHFOV, _rimg_shape), 1);
}
std::vector<int> Removerter::calcDescrepancyAndParseDynamicPointIdxForEachScanForPD(const Session& _target_sess, const Session& _source_sess, std::pair<int, int> _rimg_shape)
{
    return votedIndexes(_target_sess, _target_sess.map_global_pd_, _source_sess.keyframe_scans_static_projected_, _source_sess, alphaOfShape(kVFOV, kHFOV, _rimg_shape), 0);
}

// complement of the dynamic indexes in linspace<int>(0, N, N) (utility.h:158-167), quirk Q5 included: integer step N/(N-1), i.e.
// {0, 2} for N == 2 and a division by zero for N == 1 (reported as an exception here)
std::vector<int> Removerter::getStaticIdxFromDynamicIdx(const std::vector<int>& _dynamic_point_indexes, int _num_all_points)
{
    if (_num_all_points == 1) throw std::runtime_error("getStaticIdxFromDynamicIdx: linspace<int>(0, 1, 1) divides by zero in the reference");
    std::vector<char> dyn((size_t)std::max(_num_all_points, 0) + 3, 0);
    for (int i : _dynamic_point_indexes) if (i >= 0 && (size_t)i < dyn.size()) dyn[(size_t)i] = 1;
    const int step = _num_all_points == 2 ? 2 : 1;
    std::vector<int> out;
    for (int k = 0, v = 0; k < _num_all_points; ++k, v += step) if (!dyn[(size_t)v]) out.push_back(v);
    return out;
}

void Removerter::parsePointcloudSubsetUsingPtIdx(const CloudPtr& _ptcloud_orig, std::vector<int>& _point_indexes, CloudPtr& _ptcloud_to_save)
{
    ltm_cloud h = 0;
    static_assert(sizeof(int) == sizeof(int32_t), "int32 indices");
    ltmCheck(_ptcloud_orig->ctx, ltm_cloud_select(_ptcloud_orig->ctx, _ptcloud_orig->h, reinterpret_cast<const int32_t*>(_point_indexes.data()),
                                                   _point_indexes.size(), &h), "ltm_cloud_select");
    _ptcloud_to_save = std::make_shared<CloudH>(_ptcloud_orig->ctx, h);
}

bool Removerter::checkFineGrainedWrappers()
{
    loadSessionInfo();
    parseKeyframes();
    loadKeyframes();
    precleaningKeyframes(2.5);
    makeGlobalMap();
    const float alpha = 2.5f;
    const std::pair<int, int> shape = resetRimgSize(kFOV, alpha);
    bool ok = true;
    auto expect = [&](bool cond, const char* what) { if (!cond) { std::fprintf(stderr, "check-wrappers: %s FAILED\n", what); ok = false; } };
    // 1. index form of the vote == the batch partition
    Session& C = central_sess_;
    std::vector<int> dyn = calcDescrepancyAndParseDynamicPointIdxForEachScan(shape);
    std::vector<int> sta = getStaticIdxFromDynamicIdx(dyn, (int)C.map_global_curr_->size());
    CloudPtr dyn_pts, sta_pts;
    parsePointcloudSubsetUsingPtIdx(C.map_global_curr_, dyn, dyn_pts);
    parsePointcloudSubsetUsingPtIdx(C.map_global_curr_, sta, sta_pts);
    auto [static_tt, dynamic_tt] = partitionCurrentMap(C, C, alpha);
    auto same = [](const CloudPtr& a, const CloudPtr& b) {
        const Cloud x = a->download(), y = b->download();
        return x.size() == y.size() && (x.empty() || std::memcmp(x.data(), y.data(), x.size() * sizeof(PointType)) == 0);
    };
    expect(!dyn.empty() && dyn.size() + sta.size() == C.map_global_curr_->size(), "index sets partition the map");
    expect(same(dyn_pts, dynamic_tt), "dynamic subset == partitionCurrentMap().second");
    expect(same(sta_pts, static_tt), "static subset == partitionCurrentMap().first");
    // 2. scan2RangeImg: every pixel is either empty (10000) or the range of some scan point, and the nearest point of the scan is in it
    ltm_cloud h = 0;
    ltmCheck(dev_->ctx, ltm_scanset_keyframe(dev_->ctx, C.keyframe_scans_->h, 0, &h), "ltm_scanset_keyframe");
    CloudPtr scan0 = C.wrap(h);
    const RangeImage img = scan2RangeImg(scan0, kFOV, shape);
    const Cloud pts = scan0->download();
    float rmin = 1e30f, imin = 1e30f;
    for (const PointType& p : pts) rmin = std::min(rmin, std::sqrt(p.x * p.x + p.y * p.y + p.z * p.z));
    size_t filled = 0;
    for (float v : img.range) { if (v < kFlagNoPOINT) { ++filled; imin = std::min(imin, v); } }
    expect(img.rows == shape.first && img.cols == shape.second && filled > 0 && filled <= pts.size(), "scan2RangeImg fills at most one pixel per point");
    expect(std::fabs(imin - rmin) <= 1e-4f * rmin, "the nearest scan point is in the range image");
    // 3. free functions of utility.h: one keyframe through transformGlobalMapToLocal + parseProjectedPoints == that keyframe of the
    //    batch reprojection; two scans through local2global / mergeScansWithinGlobalCoordUtil == the head of the batch merge
    {
        CloudPtr local;
        transformGlobalMapToLocal(C.map_global_curr_, C.keyframe_inverse_poses_.at(0), kSE3MatExtrinsicPoseBasetoLiDAR, local);
        const CloudPtr projected = ltremovert::parseProjectedPoints(local, kFOV, resetRimgSize(kFOV, C.kReprojectionAlpha));
        ScansPtr batch;
        C.parseScansViaProjection(C.map_global_curr_, batch);
        ltm_cloud hk = 0;
        ltmCheck(dev_->ctx, ltm_scanset_keyframe(dev_->ctx, batch->h, 0, &hk), "ltm_scanset_keyframe");
        expect(projected->size() > 0 && same(projected, C.wrap(hk)), "transformGlobalMapToLocal + parseProjectedPoints == parseScansViaProjection[0]");
        std::vector<CloudPtr> two;
        for (size_t k = 0; k < 2; ++k) { ltm_cloud hs = 0; ltmCheck(dev_->ctx, ltm_scanset_keyframe(dev_->ctx, C.keyframe_scans_->h, k, &hs), "ltm_scanset_keyframe"); two.push_back(C.wrap(hs)); }
        const CloudPtr merged2 = mergeScansWithinGlobalCoordUtil(two, {C.keyframe_poses_.at(0), C.keyframe_poses_.at(1)}, kSE3MatExtrinsicLiDARtoPoseBase);
        ltm_cloud hall = 0;
        ltmCheck(dev_->ctx, ltm_merge_to_global(dev_->ctx, C.keyframe_scans_->h, C.poses_h_, &hall), "ltm_merge_to_global");
        const Cloud all = C.wrap(hall)->download(), head = merged2->download();
        expect(head.size() == two[0]->size() + two[1]->size() && head.size() <= all.size() &&
               std::memcmp(head.data(), all.data(), head.size() * sizeof(PointType)) == 0, "mergeScansWithinGlobalCoordUtil == head of mergeScansWithinGlobalCoord");
        expect(same(local2global(two[0], C.keyframe_poses_.at(0), kSE3MatExtrinsicLiDARtoPoseBase), mergeScansWithinGlobalCoordUtil({two[0]}, {C.keyframe_poses_.at(0)}, kSE3MatExtrinsicLiDARtoPoseBase)),
               "local2global == a one-scan merge");
        CloudPtr down;
        ltremovert::octreeDownsampling(merged2, down, 0.4f);
        expect(same(down, C.octreeDownsampling(merged2, 0.4f)), "free octreeDownsampling == Session::octreeDownsampling");
        expect((linspace<int>(0, 5, 5) == std::vector<int>{0, 1, 2, 3, 4}) && (linspace<int>(0, 2, 2) == std::vector<int>{0, 2}), "linspace<int>");
    }
    // 4. per-scan kNN forms == that keyframe of the batch stage
    {
        C.parseStaticScansViaProjection();
        C.extractLowDynPointsViaKnnDiff(query_sess_.map_global_curr_);
        const auto [co1, di1] = C.partitionLowDynamicPointsOfScanByKnn(1);
        ltm_cloud ha = 0, hb = 0;
        ltmCheck(dev_->ctx, ltm_scanset_keyframe(dev_->ctx, C.scans_knn_coexist_->h, 1, &ha), "ltm_scanset_keyframe");
        ltmCheck(dev_->ctx, ltm_scanset_keyframe(dev_->ctx, C.scans_knn_diff_->h, 1, &hb), "ltm_scanset_keyframe");
        expect(co1->size() + di1->size() > 0 && same(co1, C.wrap(ha)) && same(di1, C.wrap(hb)), "partitionLowDynamicPointsOfScanByKnn(1) == extractLowDynPointsViaKnnDiff()[1]");
    }
    // 5. quirk Q5
    expect((getStaticIdxFromDynamicIdx({}, 2) == std::vector<int>{0, 2}), "linspace<int>(0,2,2) == {0,2}");
    bool threw = false;
    try { getStaticIdxFromDynamicIdx({}, 1); } catch (const std::exception&) { threw = true; }
    expect(threw, "N == 1 is reported");
    std::printf("check-wrappers: %s (%zu dynamic of %zu map points)\n", ok ? "OK" : "FAILED", dyn.size(), C.map_global_curr_->size());
    return ok;
}

// Lifelong hand-over (reference README.md:115-118 leaves it to the user): scans_updated/ holds one file per
// central KEYFRAME, while the session's pose file has one line per SCAN, so pointing central_sess_scan_dir at scans_updated/ for
// the next run needs the matching subset of poses.  This writes it next to the scans, in the input format (12 numbers per line,
// round-trip precision); the next run uses start_idx 0, end_idx n-1, keyframe_gap 1.
void Removerter::saveKeyframePoses(const Session& _sess)
{
    if (dev_->rank() != 0) return;
    const std::string file = save_pcd_directory_ + "scans_updated_poses.txt";
    std::ofstream o(file);
    o.precision(17);
    for (const Matrix4d& T : _sess.keyframe_poses_) {
        for (int i = 0; i < 12; ++i) o << (i ? " " : "") << T[i];
        o << "\n";
    }
    if (!o) throw std::runtime_error("cannot write " + file);
    LTM_INFO(" keyframe poses of the central session saved: " << file);
}

int Removerter::runBench(int steps, int warmup)
{
    using clk = std::chrono::steady_clock;
    bench_mode_ = true;
    loadSessionInfo();
    parseKeyframes();
    loadKeyframes();
    precleaningKeyframes(2.5);
    ltm_ctx* ctx = dev_->ctx;
    auto reset = [&](Session& s) {      // everything a run derives from the loaded scans (the reference's allocateMemory(), Session.cpp:17-78)
        s.keyframe_scans_static_projected_.reset(); s.keyframe_scans_dynamic_.reset(); s.scans_knn_coexist_.reset(); s.scans_knn_diff_.reset();
        s.keyframe_scans_updated_.reset(); s.keyframe_scans_updated_strong_.reset(); s.keyframe_scans_pd_.reset(); s.keyframe_scans_strong_pd_.reset();
        s.keyframe_scans_strong_nd_.reset(); s.keyframe_scans_weak_nd_.reset(); s.knn_target_map_.reset();
        s.map_global_orig_.reset(); s.map_global_curr_.reset(); s.map_global_curr_static_.reset(); s.map_global_curr_dynamic_.reset();
        s.map_global_updated_.reset(); s.map_global_updated_strong_.reset(); s.map_global_nd_.reset(); s.map_global_nd_strong_.reset(); s.map_global_nd_weak_.reset();
        s.map_global_pd_.reset(); s.map_global_pd_orig_.reset(); s.map_global_pd_strong_.reset(); s.map_global_pd_weak_.reset();
    };
    double total = 0.0;
    if (useLanes()) ensureLane();
    for (int it = 0; it < warmup + steps; ++it) {
        reset(central_sess_); reset(query_sess_); union_q_.reset(); union_c_.reset();
        ltmCheck(ctx, ltm_clear_caches(ctx), "ltm_clear_caches");      // a step is a whole pass: no scan image survives from the previous one (as in bench.py)
        if (it == warmup && !std::getenv("LTM_BENCH_NO_PROFILE")) {
            ltmCheck(ctx, ltm_profile_enable(ctx, 1), "ltm_profile_enable"); ltmCheck(ctx, ltm_profile_reset(ctx), "ltm_profile_reset");
            if (lane_dev_) { ltmCheck(lane_dev_->ctx, ltm_profile_enable(lane_dev_->ctx, 1), "ltm_profile_enable"); ltmCheck(lane_dev_->ctx, ltm_profile_reset(lane_dev_->ctx), "ltm_profile_reset"); }
        }
        ltmCheck(ctx, ltm_synchronize(ctx), "ltm_synchronize");
        if (lane_) { reset(lane_->central_sess_); reset(lane_->query_sess_); lane_->union_q_.reset(); ltmCheck(lane_dev_->ctx, ltm_clear_caches(lane_dev_->ctx), "ltm_clear_caches"); }
        const auto t0 = clk::now();
        runStages(false);
        ltmCheck(ctx, ltm_synchronize(ctx), "ltm_synchronize");
        if (lane_dev_) ltmCheck(lane_dev_->ctx, ltm_synchronize(lane_dev_->ctx), "ltm_synchronize");
        if (it >= warmup) total += std::chrono::duration<double>(clk::now() - t0).count();
    }
    // kernel classes of the step: this context's and, on two lanes, the lane's (launch counts and units add up; the event times of launches that
    // overlapped count each other's work -- `ltm_run --bench` with removert/gpu_lanes: 1 gives clean class times)
    struct Cls { double ms = 0, units = 0; uint64_t launches = 0; };
    std::vector<std::pair<std::string, Cls>> classes;
    auto collect = [&](ltm_ctx* cx) {
        const char* names[64]; double ms[64], units[64], bytes[64]; uint64_t launches[64];
        const int nc = ltm_profile_read(cx, names, ms, launches, units, bytes, 64);
        for (int i = 0; i < nc && i < 64; ++i) {
            auto it = std::find_if(classes.begin(), classes.end(), [&](const auto& kv) { return kv.first == names[i]; });
            if (it == classes.end()) { classes.emplace_back(names[i], Cls()); it = classes.end() - 1; }
            it->second.ms += ms[i]; it->second.units += units[i]; it->second.launches += launches[i];
        }
    };
    collect(ctx);
    if (lane_dev_) collect(lane_dev_->ctx);
    std::ostringstream js;
    js.precision(17);
    js << "{\"host\": \"lt-mapper_amd/host (C++ mirror of Removerter/Session over the C ABI)\", \"steps\": " << steps << ", \"warmup\": " << warmup
       << ", \"lanes\": " << (useLanes() ? 2 : 1) << ", \"ms_per_step\": " << 1e3 * total / std::max(steps, 1) << ", \"keyframes\": [" << central_sess_.keyframe_names_.size() << ", "
       << query_sess_.keyframe_names_.size() << "], \"classes\": {";
    for (size_t i = 0; i < classes.size(); ++i)
        js << (i ? ", " : "") << "\"" << classes[i].first << "\": {\"ms_per_step\": " << classes[i].second.ms / std::max(steps, 1) << ", \"launches_per_step\": "
           << (double)classes[i].second.launches / std::max(steps, 1) << ", \"units_per_step\": " << classes[i].second.units / std::max(steps, 1) << "}";
    js << "}}";
    std::cout << "[bench] " << js.str() << std::endl;
    return 0;
}


// makeGlobalMap + Steps 1-3 of the reference's run() (Removerter.cpp:1662-1675).  write_outputs: the per-keyframe outputs that are final before
// updateScansScanwise are queued for writing as soon as they exist (asynchronous I/O only); scans_updated is written by the caller.
void Removerter::runStages(bool write_outputs)
{
    if (useLanes()) runStagesTwoLanes(write_outputs);
    else runStagesOneLane(write_outputs);
}

void Removerter::runStagesOneLane(bool write_outputs)
{
    makeGlobalMap();
    // # Step 1: HD noise removal
    removeHighDynamicPoints();
    parseStaticScansViaProjection();
    // # Step 2: LD change detection
    detectLowDynamicPoints();
    // # Step 3: LT-map
    updateCurrentMap();
    parseUpdatedStaticScansViaProjection();
    parseLDScansViaProjection();
    if (write_outputs && gpu_async_io_) {      // four of the five per-keyframe outputs are final here: their fetch + writes overlap the last stage
        saveScans(central_sess_, central_sess_.keyframe_scans_updated_strong_, updated_strong_scans_save_dir_, false);
        saveLDScans(central_sess_);
    }
    updateScansScanwise();
}

namespace {
// main_fn on this thread, lane_fn on a second one; device-side: the lane starts behind what the main context has been given so far, and the main
// context's next work is ordered behind everything the lane submitted (lent clouds may be freed or overwritten from then on)
template <class A, class B>
void forkLanes(ltm_ctx* main_ctx, ltm_ctx* lane_ctx, A&& main_fn, B&& lane_fn)
{
    ltmCheck(main_ctx, ltm_lane_fence(main_ctx, lane_ctx), "ltm_lane_fence");
    std::exception_ptr lane_err;
    std::thread th([&] { try { lane_fn(); } catch (...) { lane_err = std::current_exception(); } });
    try { main_fn(); } catch (...) { th.join(); throw; }
    th.join();
    if (lane_err) std::rethrow_exception(lane_err);
    ltmCheck(lane_ctx, ltm_lane_fence(lane_ctx, main_ctx), "ltm_lane_fence");
}
} // namespace

void Removerter::ensureLane()
{
    if (lane_) return;
    lane_dev_ = std::make_shared<Device>(Device::LaneOf{*dev_});
    lane_.reset(new Removerter(lane_dev_));
}

void Removerter::runStagesTwoLanes(bool write_outputs)
{
    ensureLane();
    Removerter& L = *lane_;
    L.bench_mode_ = bench_mode_;
    Device& M = *dev_; Device& LD = *lane_dev_;
    Session& C = central_sess_; Session& Q = query_sess_;
    Session& Cl = L.central_sess_; Session& Ql = L.query_sess_;      // the lane's views of the two sessions
    Ql.adoptKeyframes(Q, true);
    Cl.adoptKeyframes(C, false);
    const bool self = gpu_use_self_removert_ && !remove_resolution_list_.empty();
    const bool timing = std::getenv("LTM_STAGE_TIMING") != nullptr;      // wall time of the three fork / join stages, on stderr
    auto lap = [&, last = std::chrono::steady_clock::now()](const char* what) mutable {
        if (!timing) return;
        ltm_synchronize(M.ctx); ltm_synchronize(LD.ctx);
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[ltm_run] two lanes: %-28s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(now - last).count());
        last = now;
    };
    lap("views of the inputs");

    // ---- stage A: makeGlobalMap, the Step-1 chain, the HD kNN map and the static reprojection of one session per lane (Removerter.cpp:213-252, :1580-1604, :1527-1538)
    auto chainA = [self](Removerter& R, Session& s) {
        R.makeGlobalMap(s);
        if (self) R.selfRemovert(s, R.repeat_removert_iter_);
        else R.removeOnce(s, s, 2.5);
        if (!R.gpu_skip_hd_knn_) {
            s.extractHighDynPointsViaKnnDiff(s.map_global_curr_static_);
            const CloudPtr hd = s.octreeDownsampling(s.mergeScansToGlobal(s.keyframe_scans_dynamic_), 0.05f);
            R.saveMap(R.save_pcd_directory_ + (s.sess_type_ == "Central" ? "central" : "query") + "_sess_high_dyn.pcd", hd);
        }
        R.parseStaticScansViaProjection(s);
    };
    forkLanes(M.ctx, LD.ctx, [&] { chainA(*this, C); }, [&] { chainA(L, Ql); });
    lap("stage A (Step 1 chains)");
    if (!gpu_skip_hd_knn_) LTM_INFO(" high dynamic maps are saved. ");
    Ql.map_global_curr_.reset(); Ql.knn_target_map_.reset();
    Q.map_global_curr_static_ = giveCloud(Ql.map_global_curr_static_, M);
    Q.map_global_curr_dynamic_ = giveCloud(Ql.map_global_curr_dynamic_, M);
    Q.map_global_curr_ = Q.map_global_curr_static_;
    Q.keyframe_scans_dynamic_ = giveScans(Ql.keyframe_scans_dynamic_, M);

    // ---- stage B: the two directions of the LD kNN diff, the ND against the PD filter, the grids of :1445-1476 (Removerter.cpp:1413-1481)
    Q.keyframe_scans_static_projected_ = lendScans(Ql.keyframe_scans_static_projected_, M);      // source scans of filterStrongND
    Cl.map_global_curr_static_ = lendCloud(C.map_global_curr_static_, LD);                         // kNN target of the query session
    Cl.keyframe_scans_static_projected_ = lendScans(C.keyframe_scans_static_projected_, LD);      // source scans of filterStrongPD
    forkLanes(M.ctx, LD.ctx,
        [&] {
            LTM_INFO(" parse low dynamic diff via knn: " << C.sess_type_ << " to " << Q.sess_type_);
            C.extractLowDynPointsViaKnnDiff(Q.map_global_curr_static_);
            C.constructGlobalNDMap();
            filterStrongND(C, Q);
            C.removeWeakNDMapPointsHavingStrongNDInNear();
            const bool has_strong_nd = C.map_global_nd_strong_->size() != 0;
            std::vector<CloudPtr> ins = {C.mergeScansToGlobal(C.scans_knn_coexist_), C.mergeScansToGlobal(C.scans_knn_diff_), C.map_global_nd_weak_};
            if (has_strong_nd) ins.push_back(C.map_global_nd_strong_);
            const auto ds = C.octreeDownsamplingBatch(ins, 0.05f);
            union_c_ = ds[0];
            saveMap(save_pcd_directory_ + "union_map_centralside.pcd", union_c_);
            saveMap(save_pcd_directory_ + "nd_map.pcd", ds[1]);
            if (has_strong_nd) { C.map_global_nd_strong_ = ds[3]; saveMap(save_pcd_directory_ + "strong_nd_map.pcd", C.map_global_nd_strong_); }
            C.map_global_nd_weak_ = ds[2];
            saveMap(save_pcd_directory_ + "weak_nd_map.pcd", C.map_global_nd_weak_);
        },
        [&] {
            Ql.extractLowDynPointsViaKnnDiff(Cl.map_global_curr_static_);
            Ql.constructGlobalPDMap();
            L.filterStrongPD(Ql, Cl);
            Ql.revertStrongPDMapPointsHavingWeakPDInNear();
            const auto ds = Ql.octreeDownsamplingBatch({Ql.mergeScansToGlobal(Ql.scans_knn_coexist_), Ql.mergeScansToGlobal(Ql.scans_knn_diff_),
                                                        Ql.map_global_pd_strong_, Ql.map_global_pd_weak_}, 0.05f);
            L.union_q_ = ds[0];
            L.saveMap(L.save_pcd_directory_ + "union_map_queryside.pcd", L.union_q_);
            L.saveMap(L.save_pcd_directory_ + "pd_map.pcd", ds[1]);
            Ql.map_global_pd_strong_ = ds[2];
            L.saveMap(L.save_pcd_directory_ + "strong_pd_map.pcd", Ql.map_global_pd_strong_);
            Ql.map_global_pd_weak_ = ds[3];
            L.saveMap(L.save_pcd_directory_ + "weak_pd_map.pcd", Ql.map_global_pd_weak_);
        });
    lap("stage B (low dynamic)");
    LTM_INFO(" Union, PD, and ND map saved ");
    LTM_INFO(" Strong/Weak ND map saved ");
    LTM_INFO(" Strong/Weak PD map saved ");
    // the views go, the query side's results come over
    Q.keyframe_scans_static_projected_.reset(); Q.knn_target_map_.reset(); C.knn_target_map_.reset();
    Cl.map_global_curr_static_.reset(); Cl.keyframe_scans_static_projected_.reset(); Ql.knn_target_map_.reset();
    Q.keyframe_scans_static_projected_ = giveScans(Ql.keyframe_scans_static_projected_, M);
    Q.scans_knn_coexist_ = giveScans(Ql.scans_knn_coexist_, M);
    Q.scans_knn_diff_ = giveScans(Ql.scans_knn_diff_, M);
    const bool pd_is_orig = Ql.map_global_pd_ == Ql.map_global_pd_orig_;
    Q.map_global_pd_orig_ = giveCloud(Ql.map_global_pd_orig_, M);
    if (pd_is_orig) { Ql.map_global_pd_.reset(); Q.map_global_pd_ = Q.map_global_pd_orig_; }
    else Q.map_global_pd_ = giveCloud(Ql.map_global_pd_, M);
    Q.map_global_pd_strong_ = giveCloud(Ql.map_global_pd_strong_, M);
    Q.map_global_pd_weak_ = giveCloud(Ql.map_global_pd_weak_, M);
    union_q_ = giveCloud(L.union_q_, M);
    C.map_global_pd_ = Q.map_global_pd_;                 // Removerter.cpp:1435-1437
    C.map_global_pd_orig_ = Q.map_global_pd_orig_;
    C.map_global_pd_strong_ = Q.map_global_pd_strong_;

    // ---- stage C: updateCurrentMap, then the reprojections of the updated / PD / weak-ND maps and updateScansScanwise (main) beside those of the
    // "strong" maps (lane) (Removerter.cpp:1483-1577)
    updateCurrentMap();
    lap("hand-over + updateCurrentMap");
    if (!C.map_global_nd_strong_) { ltm_cloud h = 0; PointType none{}; ltmCheck(M.ctx, ltm_cloud_upload(M.ctx, &none, 0, 16, &h), "ltm_cloud_upload"); C.map_global_nd_strong_ = C.wrap(h); }
    Cl.map_global_pd_strong_ = lendCloud(C.map_global_pd_strong_, LD);
    Cl.map_global_nd_strong_ = lendCloud(C.map_global_nd_strong_, LD);
    Cl.map_global_updated_strong_ = lendCloud(C.map_global_updated_strong_, LD);
    const bool early = write_outputs && gpu_async_io_;
    forkLanes(M.ctx, LD.ctx,
        [&] {
            C.parseUpdatedStaticScansViaProjection();
            C.parsePDScansViaProjection();
            if (early) savePDScans(C);
            C.parseWeakNDScansViaProjection();
            updateScansScanwise();
        },
        [&] {
            Cl.parseStrongPDScansViaProjection();
            if (early) L.saveStrongPDScans(Cl);
            Cl.parseStrongNDScansViaProjection();
            if (early) L.saveStrongNDScans(Cl);
            Cl.parseUpdatedStrongStaticScansViaProjection();
            if (early) L.saveScans(Cl, Cl.keyframe_scans_updated_strong_, L.updated_strong_scans_save_dir_, false);
        });
    lap("stage C (Step 3)");
    LTM_INFO(" parse updated scans via projection: " << C.sess_type_);
    LTM_INFO(" parse LD scans via projection: " << C.sess_type_);
    Cl.map_global_pd_strong_.reset(); Cl.map_global_nd_strong_.reset(); Cl.map_global_updated_strong_.reset();
    C.keyframe_scans_strong_pd_ = giveScans(Cl.keyframe_scans_strong_pd_, M);
    C.keyframe_scans_strong_nd_ = giveScans(Cl.keyframe_scans_strong_nd_, M);
    C.keyframe_scans_updated_strong_ = giveScans(Cl.keyframe_scans_updated_strong_, M);
    Ql.keyframe_scans_.reset();      // the view of the query session's scans
    if (write_outputs && !gpu_async_io_) { /* the synchronous path writes everything after the stages (run()) */ }
}

void Removerter::run(void)                                                         // Removerter.cpp:1653-1678
{
    using clk = std::chrono::steady_clock;
    const auto t0 = clk::now();
    // # Step 0: Preparations
    const bool t0_detail = std::getenv("LTM_STEP0_TIMING") != nullptr;      // where Step 0 goes, on stderr
    auto lap = [&, last = t0](const char* what) mutable {
        if (!t0_detail) return;
        ltm_synchronize(dev_->ctx);
        const auto now = clk::now();
        std::fprintf(stderr, "[ltm_run] step 0: %-22s %7.1f ms\n", what, std::chrono::duration<double, std::milli>(now - last).count());
        last = now;
    };
    loadSessionInfo(); lap("loadSessionInfo");
    parseKeyframes(); lap("parseKeyframes");
    central_sess_.loadKeyframes(); lap("loadKeyframes central");
    query_sess_.loadKeyframes(); lap("loadKeyframes query");
    precleaningKeyframes(2.5); lap("precleaningKeyframes");
    const auto t1 = clk::now();      // (makeGlobalMap belongs to the stages: on two lanes each session's merge + grid opens its own chain)
    runStages(true);
    ltmCheck(dev_->ctx, ltm_synchronize(dev_->ctx), "ltm_synchronize");
    const auto t2 = clk::now();
    if (gpu_async_io_) saveScans(central_sess_, central_sess_.keyframe_scans_updated_, updated_scans_save_dir_, true);
    else saveAllTypeOfScans();
    finishOutputs();
    saveKeyframePoses(central_sess_);
    ltmCheck(dev_->ctx, ltm_synchronize(dev_->ctx), "ltm_synchronize");
    if (dev_->comm) dev_->comm->barrier();
    const auto t3 = clk::now();
    auto s = [](auto a, auto b) { return std::chrono::duration<double>(b - a).count(); };
    LTM_INFO(" [timing] step0 (load) " << s(t0, t1) << " s, makeGlobalMap + steps 1-3 " << s(t1, t2) << " s (includes map PCD writes), scan writes " << s(t2, t3) << " s");
    // files -> files wall time (SURVEY 8d "T_total"), machine readable
    if (dev_->rank() == 0)
    std::cout << "[timing] ranks " << dev_->world() << " (" << (dev_->comm ? dev_->comm->backend() : "single") << ") T_total " << s(t0, t3) << " s T_step0 " << s(t0, t1) << " s T_steps123 " << s(t1, t2) << " s T_scan_writes " << s(t2, t3)
              << " s keyframes " << central_sess_.keyframe_names_.size() << " " << query_sess_.keyframe_names_.size() << std::endl;
}

} // namespace ltremovert
