/*
 * TEST INFRASTRUCTURE -- C entry points into the REFERENCE-COMPILED code (oracle/_ref/libltm_ref.so).
 *
 * oracle/refshim/Makefile compiles the unmodified reference sources where they lie -- /root/reference/ltremovert/src/{utility,
 * RosParamServer,Session,Removerter}.cpp -- against the stand-in headers of oracle/refshim/include and links them with this file.
 * Every function below only marshals flat arrays into the reference's own types (pcl::PointCloud, cv::Mat, Eigen::Matrix4d,
 * Session, Removerter) and calls the reference function named in its comment; there is no algorithm here.  The one piece of
 * orchestration restated here is the tail of Removerter::removeHighDynamicPoints() for the selfRemovert variant (the reference's own
 * commented-out call sites, Removerter.cpp:1582,1586, enabled): see ref_pipeline_run.
 *
 * Only tests/, tools/ fixture generators and bench.py's cpu_baseline leg may load this library; the product never does.
 */
#include <cstdint>
#include <cstring>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "refshim/cv.h"
#include "refshim/eigen.h"
#include "refshim/pcl.h"
#include "refshim/ros.h"

/* Removerter keeps its two sessions private (Removerter.h:14-15); the in-memory driver below has to fill and read them */
#define private public
#include "removert/Removerter.h"
#undef private

namespace {

typedef pcl::PointCloud<PointType> Cloud;

pcPtr to_cloud(const float* xyzi, size_t n)
{
    pcPtr c(new Cloud());
    c->points.resize(n);
    for (size_t i = 0; i < n; ++i) {
        c->points[i].x = xyzi[4 * i]; c->points[i].y = xyzi[4 * i + 1]; c->points[i].z = xyzi[4 * i + 2]; c->points[i].intensity = xyzi[4 * i + 3];
    }
    c->width = (uint32_t)n; c->height = 1;
    return c;
}

size_t from_cloud(const Cloud& c, float* out, size_t cap)
{
    for (size_t i = 0; i < c.points.size() && i < cap && out; ++i) {
        out[4 * i] = c.points[i].x; out[4 * i + 1] = c.points[i].y; out[4 * i + 2] = c.points[i].z; out[4 * i + 3] = c.points[i].intensity;
    }
    return c.points.size();
}

Eigen::Matrix4d to_mat(const double* rowmajor16)
{
    /* the same conversion the reference uses for pose lines and the extrinsic (Session.cpp:108, RosParamServer.cpp:29) */
    Eigen::Matrix4d m = Eigen::Map<const Eigen::Matrix<double, -1, -1, Eigen::RowMajor>>(rowmajor16, 4, 4);
    return m;
}

void from_mat(const Eigen::Matrix4d& m, double* rowmajor16)
{
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) rowmajor16[4 * r + c] = m(r, c);
}

std::string num(double v)
{
    char b[64];
    std::snprintf(b, sizeof b, "%.17g", v);
    return b;
}

} // namespace

extern "C" {

typedef struct {
    float vfov, hfov;
    int k;
    float knn_thr;
    float voxel;
    double lidar2base[16];
    int use_self_removert;            /* 0: run() as shipped (removeOnce at 2.5); 1: selfRemovert(sess, repeat) for both sessions */
    int n_res; float res_list[8];
    int repeat;
} ref_params;

/* fills the stand-in parameter table (what rosparam load params_ltmapper.yaml would have put under removert/) */
void ref_set_params(const ref_params* p, const char* save_dir)
{
    using refshim::set_param;
    set_param("removert/sequence_vfov", num(p->vfov));
    set_param("removert/sequence_hfov", num(p->hfov));
    set_param("removert/num_nn_points_within", std::to_string(p->k));
    set_param("removert/dist_nn_points_within", num(p->knn_thr));
    set_param("removert/downsample_voxel_size", num(p->voxel));
    std::string e = "[";
    for (int i = 0; i < 16; ++i) e += (i ? ", " : "") + num(p->lidar2base[i]);
    set_param("removert/ExtrinsicLiDARtoPoseBase", e + "]");
    std::string r = "[";
    for (int i = 0; i < p->n_res; ++i) r += (i ? ", " : "") + num(p->res_list[i]);
    set_param("removert/remove_resolution_list", r + "]");
    set_param("removert/repeat_removert_iter", std::to_string(p->repeat));
    set_param("removert/saveMapPCD", "true");
    set_param("removert/save_pcd_directory", save_dir);
    set_param("removert/num_omp_cores", "1");
}

/* ---- scalar numerics: utility.cpp:38-56, 222-236 ---- */
float ref_rad2deg(float r) { return rad2deg(r); }
void ref_rad2deg_array(const float* r, size_t n, float* out) { for (size_t i = 0; i < n; ++i) out[i] = rad2deg(r[i]); }
void ref_cart2sph_array(const float* xyz, size_t n, float* az_el_r)
{
    for (size_t i = 0; i < n; ++i) {
        PointType p; p.x = xyz[3 * i]; p.y = xyz[3 * i + 1]; p.z = xyz[3 * i + 2];
        const SphericalPoint s = cart2sph(p);
        az_el_r[3 * i] = s.az; az_el_r[3 * i + 1] = s.el; az_el_r[3 * i + 2] = s.r;
    }
}
void ref_resetRimgSize(float vfov, float hfov, float alpha, int* rows, int* cols)
{
    const std::pair<int, int> s = resetRimgSize(std::pair<float, float>(vfov, hfov), alpha);
    *rows = s.first; *cols = s.second;
}
/* utility.h:158-167 */
size_t ref_linspace_int(int a, int b, size_t N, int* out)
{
    const std::vector<int> v = linspace<int>(a, b, N);
    for (size_t i = 0; i < v.size(); ++i) out[i] = v[i];
    return v.size();
}
/* utility.cpp:28-36 */
size_t ref_splitPoseLine(const char* line, double* out, size_t cap)
{
    const std::vector<double> v = splitPoseLine(line, ' ');
    for (size_t i = 0; i < v.size() && i < cap; ++i) out[i] = v[i];
    return v.size();
}
/* Eigen::Matrix4d::inverse() as Session.cpp:110 calls it (stand-in Eigen: restated, parity unpinned) */
void ref_inverse4x4(const double* m, double* inv) { from_mat(to_mat(m).inverse(), inv); }

/* ---- transforms: utility.cpp:64-72, 160-202 ---- */
void ref_transformGlobalMapToLocal(const float* pts, size_t n, const double* base_pose_inverse, const double* base2lidar, float* out)
{
    pcPtr in = to_cloud(pts, n), loc(new Cloud());
    transformGlobalMapToLocal(in, to_mat(base_pose_inverse), to_mat(base2lidar), loc);
    from_cloud(*loc, out, n);
}
void ref_local2global(const float* pts, size_t n, const double* pose, const double* lidar2base, float* out)
{
    from_cloud(*local2global(to_cloud(pts, n), to_mat(pose), to_mat(lidar2base)), out, n);
}
void ref_global2local(const float* pts, size_t n, const double* pose_inverse, const double* base2lidar, float* out)
{
    from_cloud(*global2local(to_cloud(pts, n), to_mat(pose_inverse), to_mat(base2lidar)), out, n);
}
void ref_mergeScansWithinGlobalCoordUtil(const float* scans, const uint64_t* off, size_t n_kf, const double* poses, const double* lidar2base, float* out)
{
    std::vector<pcPtr> v; std::vector<Eigen::Matrix4d> ps;
    for (size_t k = 0; k < n_kf; ++k) { v.push_back(to_cloud(scans + 4 * off[k], off[k + 1] - off[k])); ps.push_back(to_mat(poses + 16 * k)); }
    from_cloud(*mergeScansWithinGlobalCoordUtil(v, ps, to_mat(lidar2base)), out, off[n_kf]);
}

/* ---- range images: utility.cpp:92-142 (serial: built without OpenMP, so the documented race of :127-138 cannot occur) ---- */
void ref_map2RangeImg(const float* pts, size_t n, float vfov, float hfov, int rows, int cols, float* rimg, int32_t* ptidx)
{
    auto im = map2RangeImg(to_cloud(pts, n), std::pair<float, float>(vfov, hfov), std::pair<int, int>(rows, cols));
    std::memcpy(rimg, im.first.ptr<float>(), sizeof(float) * rows * cols);
    if (ptidx) std::memcpy(ptidx, im.second.ptr<int>(), sizeof(int) * rows * cols);
}
/* utility.cpp:74-89 */
size_t ref_parseProjectedPoints(const float* pts, size_t n, float vfov, float hfov, int rows, int cols, float* out, size_t cap)
{
    return from_cloud(*parseProjectedPoints(to_cloud(pts, n), std::pair<float, float>(vfov, hfov), std::pair<int, int>(rows, cols)), out, cap);
}
/* utility.cpp:204-219 (over the stand-in pointer octree) */
size_t ref_octreeDownsampling(const float* pts, size_t n, float leaf, float* out, size_t cap)
{
    pcPtr in = to_cloud(pts, n), res(new Cloud());
    octreeDownsampling(in, res, leaf);
    return from_cloud(*res, out, cap);
}

/* ---- stand-in leaves on their own (refshim/pcl.h), for the oracle-vs-second-derivation tests ---- */
size_t ref_leaf_voxel_grid(const float* pts, size_t n, float leaf, float* out, size_t cap)
{
    pcl::VoxelGrid<PointType> g;
    g.setLeafSize(leaf, leaf, leaf);
    pcPtr in = to_cloud(pts, n);
    g.setInputCloud(in);
    Cloud o;
    g.filter(o);
    return from_cloud(o, out, cap);
}
void ref_leaf_knn(const float* target, size_t m, const float* query, size_t q, int k, int32_t* idx, float* sqd)
{
    pcl::KdTreeFLANN<PointType> t;
    t.setInputCloud(to_cloud(target, m));
    std::vector<int> i; std::vector<float> d;
    for (size_t j = 0; j < q; ++j) {
        PointType p; p.x = query[4 * j]; p.y = query[4 * j + 1]; p.z = query[4 * j + 2];
        const int got = t.nearestKSearch(p, k, i, d);
        for (int a = 0; a < k; ++a) { idx[(size_t)k * j + a] = a < got ? i[a] : -1; sqd[(size_t)k * j + a] = a < got ? d[a] : -1.0f; }
    }
}

/* ---- a Removerter instance for its member functions and for the pipeline ---- */
typedef struct ref_rmv ref_rmv;

ref_rmv* ref_rmv_create(const ref_params* p, const char* save_dir, int write_files)
{
    ref_set_params(p, save_dir);
    pcl::io::save_sink().capture = true;
    pcl::io::save_sink().write_files = write_files != 0;
    return reinterpret_cast<ref_rmv*>(new ltremovert::Removerter());       /* ctor: Removerter.cpp:17-73 (creates the 7 output directories) */
}
void ref_rmv_destroy(ref_rmv* h) { delete reinterpret_cast<ltremovert::Removerter*>(h); }

/* Removerter::scan2RangeImg, Removerter.cpp:109-156 */
void ref_scan2RangeImg(ref_rmv* h, const float* pts, size_t n, float vfov, float hfov, int rows, int cols, float* rimg)
{
    auto* R = reinterpret_cast<ltremovert::Removerter*>(h);
    cv::Mat im = R->scan2RangeImg(to_cloud(pts, n), std::pair<float, float>(vfov, hfov), std::pair<int, int>(rows, cols));
    std::memcpy(rimg, im.ptr<float>(), sizeof(float) * rows * cols);
}
/* Removerter::calcDescrepancyAndParseDynamicPointIdx, Removerter.cpp:381-413 */
size_t ref_calcDescrepancy(ref_rmv* h, const float* scan_rimg, const float* diff_rimg, const int32_t* ptidx, int rows, int cols, float thres,
                           int32_t* out, size_t cap)
{
    auto* R = reinterpret_cast<ltremovert::Removerter*>(h);
    cv::Mat s(rows, cols, CV_32FC1), d(rows, cols, CV_32FC1), pi(rows, cols, CV_32SC1);
    std::memcpy(s.ptr<float>(), scan_rimg, sizeof(float) * rows * cols);
    std::memcpy(d.ptr<float>(), diff_rimg, sizeof(float) * rows * cols);
    std::memcpy(pi.ptr<int>(), ptidx, sizeof(int) * rows * cols);
    const std::vector<int> v = R->calcDescrepancyAndParseDynamicPointIdx(s, d, pi, thres);
    for (size_t i = 0; i < v.size() && i < cap; ++i) out[i] = v[i];
    return v.size();
}
/* Removerter::getStaticIdxFromDynamicIdx, Removerter.cpp:675-687 */
size_t ref_getStaticIdxFromDynamicIdx(ref_rmv* h, const int32_t* dyn, size_t n_dyn, int num_all, int32_t* out, size_t cap)
{
    auto* R = reinterpret_cast<ltremovert::Removerter*>(h);
    const std::vector<int> v = R->getStaticIdxFromDynamicIdx(std::vector<int>(dyn, dyn + n_dyn), num_all);
    for (size_t i = 0; i < v.size() && i < cap; ++i) out[i] = v[i];
    return v.size();
}
/* Removerter::parsePointcloudSubsetUsingPtIdx, Removerter.cpp:933-946 */
size_t ref_parsePointcloudSubsetUsingPtIdx(ref_rmv* h, const float* pts, size_t n, const int32_t* idx, size_t n_idx, float* out, size_t cap)
{
    auto* R = reinterpret_cast<ltremovert::Removerter*>(h);
    std::vector<int> v(idx, idx + n_idx);
    pcPtr o(new Cloud());
    R->parsePointcloudSubsetUsingPtIdx(to_cloud(pts, n), v, o);
    return from_cloud(*o, out, cap);
}

/* Session::parseKeyframes({start, end}, gap), Session.cpp:138-183 (quirk Q6 included): the selected scan indices */
size_t ref_parseKeyframes(ref_rmv* h, int n_scans, int start, int end, int gap, int32_t* out, size_t cap)
{
    auto* R = reinterpret_cast<ltremovert::Removerter*>(h);
    ltremovert::Session& s = R->central_sess_;
    s.scan_paths_.clear(); s.scan_names_.clear(); s.scan_poses_.clear(); s.scan_inverse_poses_.clear();
    for (int i = 0; i < n_scans; ++i) {
        s.scan_paths_.push_back(std::to_string(i)); s.scan_names_.push_back(std::to_string(i));
        s.scan_poses_.push_back(Eigen::Matrix4d::Identity()); s.scan_inverse_poses_.push_back(Eigen::Matrix4d::Identity());
    }
    s.parseKeyframes(std::pair<int, int>(start, end), gap);
    for (size_t i = 0; i < s.keyframe_names_.size() && i < cap; ++i) out[i] = std::stoi(s.keyframe_names_[i]);
    return s.keyframe_names_.size();
}

/* Session::precleaningKeyframes, Session.cpp:506-533, on one scan */
size_t ref_precleaning(ref_rmv* h, const float* pts, size_t n, float radius, float* out, size_t cap)
{
    auto* R = reinterpret_cast<ltremovert::Removerter*>(h);
    ltremovert::Session& s = R->central_sess_;
    s.keyframe_scans_.clear();
    s.keyframe_scans_.push_back(to_cloud(pts, n));
    s.precleaningKeyframes(radius);
    const size_t m = from_cloud(*s.keyframe_scans_[0], out, cap);
    s.keyframe_scans_.clear();
    return m;
}

/* Session::removeWeakNDMapPointsHavingStrongNDInNear, Session.cpp:452-484 (k = 2, thr = 1.0 hard-coded there): near[i] = 1 iff weak point i
 * moved to the strong map.  The k-NN label rule (accumulate in double from 0.0, float mean, fabs, strict <) is the reference's text. */
void ref_weakStrongSplit(ref_rmv* h, const float* strong, size_t ns, const float* weak, size_t nw, uint8_t* near)
{
    auto* R = reinterpret_cast<ltremovert::Removerter*>(h);
    ltremovert::Session& s = R->central_sess_;
    *s.map_global_nd_strong_ = *to_cloud(strong, ns);
    *s.map_global_nd_weak_ = *to_cloud(weak, nw);
    s.removeWeakNDMapPointsHavingStrongNDInNear();
    /* the function appends the moved points to strong and rebuilds weak from the others, both in input order: recover the flags
       (identical points are identical queries and go the same way, so matching the next moved point by value is unambiguous) */
    size_t a = ns;
    for (size_t i = 0; i < nw; ++i) {
        const bool moved = a < s.map_global_nd_strong_->points.size() && std::memcmp(&s.map_global_nd_strong_->points[a].x, weak + 4 * i, 12) == 0;
        near[i] = moved ? 1 : 0;
        if (moved) ++a;
    }
    s.map_global_nd_strong_->clear(); s.map_global_nd_weak_->clear();
}

/* ---- the pipeline on in-memory sessions: Removerter::run(), Removerter.cpp:1653-1678, from makeGlobalMap() on ----
 * (loadSessionInfo / parseKeyframes / loadKeyframes / precleaningKeyframes are the file side; the process-level binary
 * oracle/_ref/removert_removert runs those too).  Poses are 4x4 row-major; inverses come from Matrix4d::inverse() as in Session.cpp:110. */
static void fill_session(ltremovert::Session& s, const char* type, float voxel, const float* scans, const uint64_t* off, size_t n_kf, const double* poses)
{
    s.sess_type_ = type;
    s.setDownsampleSize(voxel);
    s.keyframe_scans_.clear(); s.keyframe_poses_.clear(); s.keyframe_inverse_poses_.clear(); s.keyframe_names_.clear();
    for (size_t k = 0; k < n_kf; ++k) {
        s.keyframe_scans_.push_back(to_cloud(scans + 4 * off[k], off[k + 1] - off[k]));
        const Eigen::Matrix4d pose = to_mat(poses + 16 * k);
        s.keyframe_poses_.push_back(pose);
        s.keyframe_inverse_poses_.push_back(pose.inverse());
        char nm[32];
        std::snprintf(nm, sizeof nm, "%06zu.pcd", k);
        s.keyframe_names_.push_back(nm);
    }
}

int ref_pipeline_run(ref_rmv* h, const ref_params* p,
                     const float* c_scans, const uint64_t* c_off, size_t c_nkf, const double* c_poses,
                     const float* q_scans, const uint64_t* q_off, size_t q_nkf, const double* q_poses)
{
    auto* R = reinterpret_cast<ltremovert::Removerter*>(h);
    pcl::io::save_sink().clouds.clear();
    fill_session(R->central_sess_, "Central", p->voxel, c_scans, c_off, c_nkf, c_poses);
    fill_session(R->query_sess_, "Query", p->voxel, q_scans, q_off, q_nkf, q_poses);
    R->makeGlobalMap();
    if (!p->use_self_removert) {
        R->removeHighDynamicPoints();
    } else {
        /* Removerter::removeHighDynamicPoints() with its own commented-out call sites (Removerter.cpp:1582,1586) instead of the two
           removeOnce(…, 2.5) calls; the rest is the text of :1590-1601 */
        ltremovert::Session& C = R->central_sess_; ltremovert::Session& Q = R->query_sess_;
        R->selfRemovert(C, R->repeat_removert_iter_);
        R->selfRemovert(Q, R->repeat_removert_iter_);
        C.extractHighDynPointsViaKnnDiff(C.map_global_curr_static_);
        Q.extractHighDynPointsViaKnnDiff(Q.map_global_curr_static_);
        auto hc = mergeScansWithinGlobalCoordUtil(C.keyframe_scans_dynamic_, C.keyframe_poses_, C.kSE3MatExtrinsicLiDARtoPoseBase);
        auto hq = mergeScansWithinGlobalCoordUtil(Q.keyframe_scans_dynamic_, Q.keyframe_poses_, Q.kSE3MatExtrinsicLiDARtoPoseBase);
        octreeDownsampling(hc, hc, 0.05);
        octreeDownsampling(hq, hq, 0.05);
        pcl::io::savePCDFileBinary(R->save_pcd_directory_ + "central_sess_high_dyn.pcd", *hc);
        pcl::io::savePCDFileBinary(R->save_pcd_directory_ + "query_sess_high_dyn.pcd", *hq);
    }
    R->parseStaticScansViaProjection();
    R->detectLowDynamicPoints();
    R->updateCurrentMap();
    R->parseUpdatedStaticScansViaProjection();
    R->parseLDScansViaProjection();
    R->updateScansScanwise();
    R->saveAllTypeOfScans();
    return 0;
}

/* The vote of one pass over the given keyframes: Removerter::calcDescrepancyAndParseDynamicPointIdxForEachScan (Removerter.cpp:542-593; which = 0),
 * ...ForND (:485-540, diff = map - scan; which = 1), ...ForPD (:429-482; which = 2).  The session's map (map_global_curr_ / _nd_ / _pd_) is `map`,
 * the source scans are `scans` (keyframe_scans_ for which = 0, keyframe_scans_static_projected_ otherwise).  Returns the sorted unique indices. */
size_t ref_vote_dynamic_idx(ref_rmv* h, int which, const float* map, size_t M, const float* scans, const uint64_t* off, size_t n_kf, const double* poses,
                            float res_alpha, int32_t* out, size_t cap)
{
    auto* R = reinterpret_cast<ltremovert::Removerter*>(h);
    ltremovert::Session& s = R->central_sess_;
    fill_session(s, "Central", R->kDownsampleVoxelSize, scans, off, n_kf, poses);
    s.keyframe_scans_static_projected_ = s.keyframe_scans_;
    pcPtr m = to_cloud(map, M);
    *s.map_global_curr_ = *m; *s.map_global_nd_ = *m; *s.map_global_pd_ = *m;
    const std::pair<int, int> shape = resetRimgSize(R->kFOV, res_alpha);
    const std::vector<int> v = which == 0 ? R->calcDescrepancyAndParseDynamicPointIdxForEachScan(s, s, shape)
                             : which == 1 ? R->calcDescrepancyAndParseDynamicPointIdxForEachScanForND(s, s, shape)
                                          : R->calcDescrepancyAndParseDynamicPointIdxForEachScanForPD(s, s, shape);
    for (size_t i = 0; i < v.size() && i < cap; ++i) out[i] = v[i];
    s.keyframe_scans_.clear(); s.keyframe_scans_static_projected_.clear();
    s.map_global_curr_->clear(); s.map_global_nd_->clear(); s.map_global_pd_->clear();
    return v.size();
}

/* a cloud the reference saved during the run: name relative to save_pcd_directory, e.g. "updated_map.pcd", "scans_pd/000003.pcd" */
int ref_saved_cloud(ref_rmv* h, const char* rel, const float** pts, size_t* n, uint32_t* width, uint32_t* height)
{
    auto* R = reinterpret_cast<ltremovert::Removerter*>(h);
    auto& s = pcl::io::save_sink();
    auto it = s.clouds.find(R->save_pcd_directory_ + rel);
    if (it == s.clouds.end()) return -1;
    *pts = it->second.data(); *n = it->second.size() / 4;
    if (width) *width = s.shapes[it->first].first;
    if (height) *height = s.shapes[it->first].second;
    return 0;
}
/* names of everything saved, '\n'-separated, relative to save_pcd_directory */
size_t ref_saved_names(ref_rmv* h, char* buf, size_t cap)
{
    auto* R = reinterpret_cast<ltremovert::Removerter*>(h);
    std::string all;
    for (auto& kv : pcl::io::save_sink().clouds) {
        std::string k = kv.first;
        if (k.compare(0, R->save_pcd_directory_.size(), R->save_pcd_directory_) == 0) k = k.substr(R->save_pcd_directory_.size());
        all += k + "\n";
    }
    if (buf && cap) { std::strncpy(buf, all.c_str(), cap - 1); buf[cap - 1] = 0; }
    return all.size();
}
int ref_empty_saves(void) { return pcl::io::empty_saves(); }

/* session state the reference does not save: which = "static_projected" | "knn_coexist" | "knn_diff" | "scans_dynamic" | "weak_nd" */
int ref_session_scans(ref_rmv* h, int query, const char* which, size_t kf, float* out, size_t cap, size_t* n)
{
    auto* R = reinterpret_cast<ltremovert::Removerter*>(h);
    ltremovert::Session& s = query ? R->query_sess_ : R->central_sess_;
    const std::string w = which;
    const std::vector<pcPtr>* v = w == "static_projected" ? &s.keyframe_scans_static_projected_ : w == "knn_coexist" ? &s.scans_knn_coexist_
        : w == "knn_diff" ? &s.scans_knn_diff_ : w == "scans_dynamic" ? &s.keyframe_scans_dynamic_ : w == "weak_nd" ? &s.keyframe_scans_weak_nd_ : nullptr;
    if (!v || kf >= v->size()) return -1;
    *n = from_cloud(*(*v)[kf], out, cap);
    return 0;
}
/* session maps: which = "curr" | "static" | "dynamic" | "nd" | "nd_strong" | "nd_weak" | "pd" | "pd_orig" | "pd_strong" | "pd_weak" | "updated" | "updated_strong" */
int ref_session_map(ref_rmv* h, int query, const char* which, float* out, size_t cap, size_t* n)
{
    auto* R = reinterpret_cast<ltremovert::Removerter*>(h);
    ltremovert::Session& s = query ? R->query_sess_ : R->central_sess_;
    const std::string w = which;
    pcPtr c = w == "curr" ? s.map_global_curr_ : w == "static" ? s.map_global_curr_static_ : w == "dynamic" ? s.map_global_curr_dynamic_
        : w == "nd" ? s.map_global_nd_ : w == "nd_strong" ? s.map_global_nd_strong_ : w == "nd_weak" ? s.map_global_nd_weak_
        : w == "pd" ? s.map_global_pd_ : w == "pd_orig" ? s.map_global_pd_orig_ : w == "pd_strong" ? s.map_global_pd_strong_
        : w == "pd_weak" ? s.map_global_pd_weak_ : w == "updated" ? s.map_global_updated_ : w == "updated_strong" ? s.map_global_updated_strong_ : pcPtr();
    if (!c) return -1;
    *n = from_cloud(*c, out, cap);
    return 0;
}

} // extern "C"
