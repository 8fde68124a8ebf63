/*
 * ltm_oracle.h -- TEST INFRASTRUCTURE.  CPU oracle for the LT-removert / LT-map hot path.
 *
 * A ROS/PCL/OpenCV-free, single-threaded-by-default restatement of the reference's
 * algorithm (gisbi-kim/lt-mapper, ltremovert/).  Every function cites the reference
 * file:line it follows.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this; the product (lt-mapper_amd/) never does.
 *
 * PARITY STATUS (round 4): the reference ships no tests, golden vectors or fixtures for this path,
 * and its real build needs ROS, PCL 1.10, OpenCV and Eigen, none of which exist in the image.  But everything the reference
 * ITSELF computes is plain C++ over those libraries' containers: oracle/refshim/ compiles the reference's own, unmodified
 * sources (/root/reference/ltremovert/src/{utility,RosParamServer,Session,Removerter,removert_main}.cpp) against stand-in
 * headers into oracle/_ref/ (serial: no OpenMP, the reference's parallel min-update is a documented race), and this oracle
 * is PINNED to that build: tests/test_ref_compiled.py compares them bitwise on the scalar numerics, range images, vote
 * passes (all three forms), the std::set union / complement, the remove / revert / selfRemovert state machine, the k-NN
 * label rule, detectLowDynamicPoints / updateCurrentMap / updateScansScanwise, parseKeyframes (quirk Q6), pre-cleaning and
 * the whole run() -- in memory (1-res, 3-res, extrinsic, full SE(3) poses 50 km from the origin) and as a process, files
 * to files, at configs[0]'s real size; tests/golden/*.npz are outputs of that build.
 * STILL UNPINNED (no source in the container): the library leaves the stand-ins restate -- pcl::transformPointCloud<double>,
 * OctreePointCloudVoxelCentroid, VoxelGrid, KdTreeFLANN / FLANN exact k-NN, ExtractIndices, the PCD reader / writer, Eigen's
 * Matrix4d::inverse(), OpenCV's colour map.  For those the stand-ins are a SECOND, deliberately literal derivation (pointer
 * octree, PCL's index sort, a plain kd-tree) which the oracle must match bitwise -- that is how the in-voxel summation
 * order of pcl::VoxelGrid (std::sort on the leaf index, not input order) was found.  Also pinned: atanf/atan2f against the
 * host glibc (pin_atan2f.c, exhaustive) and the hand-derivable known-answer tests of SURVEY.md Appendix B.
 *
 * C ABI so that tests can drive it through ctypes.  All clouds are packed XYZI float32
 * (16 B / point).  Matrices are 4x4 row-major double.
 */
#ifndef LTM_ORACLE_H
#define LTM_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- scalar numerics (utility.cpp:38-56, 114-125, 222-236) ---- */
float orc_atan2f(float y, float x);
void  orc_atan2f_array(const float* y, const float* x, float* out, size_t n);
float orc_rad2deg(float rad);
void  orc_cart2sph(const float* xyz, float* az_el_r);
void  orc_rimg_size(float vfov, float hfov, float alpha, int* rows, int* cols);
void  orc_pixel(const float* xyz, float vfov, float hfov, int rows, int cols, int* row, int* col, float* range);

/* sensitivity experiments only: move `ppm` per million of all atan2f results by +-1 ulp (hash-chosen); 0 = off (default) */
void  orc_set_atan2f_perturbation(unsigned ppm, uint64_t seed);

/* ---- transforms (utility.cpp:64-72,160-202; PCL transformPointCloud<double>) ---- */
void orc_transform(const double* T, const float* in, float* out, size_t n);
/* Eigen::Matrix4d::inverse() of Session.cpp:110 / RosParamServer.cpp:30: general 4x4 inverse in double in the operation order of
 * Eigen 3.3.7's SSE2 kernel (2x2 block adjugates; restated from knowledge, parity unpinned) */
int  orc_inverse4x4(const double* m, double* inv);
/* variant 0 = the above, 1 = cofactor expansion, 2 = Gauss-Jordan with partial pivoting (sensitivity test only) */
int  orc_inverse4x4_variant(const double* m, double* inv, int variant);

/* ---- range images (utility.cpp:92-142, Removerter.cpp:109-156) ----
 * T1/T2 may be NULL (no transform).  ptidx may be NULL (scan image). */
void orc_range_image(const float* pts, size_t n, const double* T1, const double* T2,
                     float vfov, float hfov, int rows, int cols, float* rimg, int32_t* ptidx);

/* ---- visibility vote (Removerter.cpp:381-413, 429-593) ----
 * mode 0: diff = scan - map (remove / revert / PD); mode 1: diff = map - scan (ND).
 * labels[M] is OR-accumulated (caller zeroes); keyframes [kf_begin, kf_end). */
void orc_vote_labels(const float* map, size_t M, const float* scans, const uint64_t* offsets, size_t n_kf,
                     const double* inv_poses, const double* base2lidar, float vfov, float hfov,
                     float alpha, float diff_thres, int mode, size_t kf_begin, size_t kf_end,
                     int threads, uint8_t* labels);

/* ---- voxel centroid (utility.cpp:204-219; PCL OctreePointCloudVoxelCentroid) ----
 * returns number of voxels; writes at most cap points to out (may be NULL to count). */
size_t orc_voxel_centroid(const float* pts, size_t n, float leaf, float* out, size_t cap);
/* multi-rank tests: grid / Morton keys under the octree frame of a GIVEN box mn_mx = {min x y z, max x y z}; keys returns the depth (< 0: too deep) */
size_t orc_voxel_centroid_box(const float* pts, size_t n, const float* mn_mx, float leaf, float* out, size_t cap);
int orc_voxel_keys_box(const float* pts, size_t n, const float* mn_mx, float leaf, uint64_t* keys);

/* ---- reprojection (utility.cpp:74-89, Session.cpp:348-360) ----
 * out gets at most cap points; out_offsets has (kf_end-kf_begin)+1 entries. returns total points. */
size_t orc_reproject(const float* map, size_t M, const double* inv_poses, const double* base2lidar,
                     float vfov, float hfov, float alpha, size_t kf_begin, size_t kf_end,
                     int threads, float* out, size_t cap, uint64_t* out_offsets);

/* ---- kNN diff (Session.cpp:393-427, 487-504, 537-642) ----
 * For every scan point: coexist[i] = 1/0, and local_out = the point after
 * local2global(B2L sic, pose) -> global2local(inv_pose, B2L).  use_kdtree=0 selects brute force. */
void orc_knn_labels(const float* target, size_t Mt, const float* scans, const uint64_t* offsets, size_t n_kf,
                    const double* poses, const double* inv_poses, const double* base2lidar,
                    int k, float thr, size_t kf_begin, size_t kf_end, int threads, int use_kdtree,
                    uint8_t* coexist, float* local_out);
/* Session.cpp:452-484 (and any cloud-vs-cloud split): near[i]=1 iff mean of k nn sq.dists < thr */
void orc_knn_split(const float* target, size_t Mt, const float* query, size_t Q, int k, float thr,
                   int use_kdtree, uint8_t* near);

/* ---- merge (utility.cpp:170-192) ---- out must hold offsets[n_kf] points */
void orc_merge_to_global(const float* scans, const uint64_t* offsets, size_t n_kf, const double* poses,
                         const double* lidar2base, float* out);

/* ---- pre-clean (Session.cpp:506-533) ---- returns kept count, in-order compaction into out */
size_t orc_preclean(const float* pts, size_t n, float radius, float* out);

/* ---- loader-side pcl::VoxelGrid (Session.cpp:284-289) ---- returns output count; out may be NULL to count */
size_t orc_voxel_grid(const float* pts, size_t n, float leaf, float* out, size_t cap);
/* 0 (default): the in-voxel summation order of PCL's std::sort on the leaf index (== oracle/_ref); 1: input order (what the device
 * form of the cascade hand-over uses; differs in the last bit where a voxel holds >= 3 points) */
void orc_set_voxel_grid_stable(int stable);

/* ---- full pipeline: Removerter::run() Steps 1-3 (Removerter.cpp:1653-1678) on in-memory sessions ---- */
/* ---- RViz images (utility.h:114-127 convertColorMappedImg, utility.cpp:248-256 pubRangeImg, Removerter.cpp:580-585) ----
 * dst8 = saturate_u8(round_half_even(src*a + b)), a = 255*(1/(max-min)), b = -255*min*(1/(max-min)) as cv::MatExpr folds
 * them (double scalars, float arithmetic for float images, double for int32 images), then cv::COLORMAP_JET, BGR order.
 * PARITY UNPINNED: OpenCV 4.2 is not in /root/reference; the JET table is restated from its published construction
 * (256 float samples of the piecewise-linear jet ramps, *255.f, round-half-even); every ramp entry is a rounding tie
 * (127.5 + 4i), so single entries may differ by 1 LSB from the library's table. */
void orc_jet_lut(uint8_t* lut_bgr /* 256*3 */);
void orc_colormap_f32(const float* src, size_t n, float cmin, float cmax, uint8_t* bgr);
void orc_colormap_i32(const int32_t* src, size_t n, float cmin, float cmax, uint8_t* bgr);

typedef struct {
    float vfov, hfov;                 /* sequence_vfov / sequence_hfov */
    int   k;                          /* num_nn_points_within */
    float knn_thr;                    /* dist_nn_points_within */
    float voxel;                      /* downsample_voxel_size */
    double lidar2base[16];            /* ExtrinsicLiDARtoPoseBase */
    int   use_self_removert;          /* 0 = as shipped (removeOnce 2.5), 1 = selfRemovert over res list */
    int   n_res; float res_list[8];   /* remove_resolution_list */
    int   repeat;                     /* repeat_removert_iter */
    int   threads;                    /* OpenMP threads over keyframes (1 = reference-serial semantics) */
    int   skip_hd_knn;                /* 1 = skip the viz-only HD kNN stage */
    int   kf_sample_stride;           /* >1: per-keyframe loops visit every s-th keyframe (baseline sampling only) */
} orc_params;

typedef struct orc_run orc_run;
orc_run* orc_pipeline_run(const orc_params* p,
                          const float* c_scans, const uint64_t* c_offsets, size_t c_nkf, const double* c_poses, const double* c_inv,
                          const float* q_scans, const uint64_t* q_offsets, size_t q_nkf, const double* q_poses, const double* q_inv);
/* named map clouds: e.g. "updated_map", "strong_nd_map", ...; returns 0 if found */
int  orc_run_cloud(const orc_run* r, const char* name, const float** pts, size_t* n);
/* named per-keyframe scan sets: "scans_updated", "scans_pd", ...; offsets has n_kf+1 entries */
int  orc_run_scanset(const orc_run* r, const char* name, const float** pts, const uint64_t** offsets, size_t* n_kf);
/* stage timings in seconds; returns number of stages; names are static strings */
int  orc_run_timings(const orc_run* r, const char** names, double* secs, int cap);
void orc_run_free(orc_run* r);

#ifdef __cplusplus
}
#endif
#endif
