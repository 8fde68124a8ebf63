// ltm_kernels.h -- host-callable launchers of the gfx950 kernels (definitions in ltm_kernels.hip).
// All launchers enqueue on the given stream and return the hipError_t of the launch.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdint.h>

namespace ltm {

struct Geom {            // range-image geometry (utility.cpp:222-236 resetRimgSize + fov)
    float vfov, hfov;
    int rows, cols;
    int fast;            // 1: use the self-checked fast arithmetic forms (ltm_device_math.h), 0: plain IEEE divisions
    float cull_eps_px;   // half-width [pixels] of the band around a pixel-rounding boundary inside which the bounded-error projection
                         // does not trust its pixel (proportional to the image resolution: the angular error is fixed, see geom_for)
    // Elevation of the bounded-error projection when the field of view clamps everything steeper than vfov/2 + 2 deg < 45 deg:
    // atan(t) ~ t * (el_c[0] + u (el_c[1] + u (el_c[2] + u el_c[3]))), u = t^2, fitted on [0, tan(vfov/2 + 2 deg)] when the context is
    // created (fit_elevation_poly in ltm_api_core.cpp) and evaluated at min(t, el_tclamp); el_tclamp = tan(vfov/2 + one pixel) puts
    // every clamped elevation in the MIDDLE of the out-of-image row -1 / R, where its pixel (row 0 / R-1 after the clamp) is certain.
    int el_fit;          // 0: the fit is not good enough for this field of view, the kernels use the generic polynomial on [0, 1]
    float el_c[4];
    float el_tclamp;
};

// 3x4 row-major double (last row of the 4x4 is never used by PCL's se3 transformer)
struct HostMat34 { double m[12]; };

static const uint32_t kNoPointBits = 0x461C4000u;   // bit pattern of kFlagNoPOINT = 10000.0f (utility.h:93)

// ---- fills ----
hipError_t fill_u32(uint32_t* p, uint32_t v, size_t n, hipStream_t s);
hipError_t fill_u64(uint64_t* p, uint64_t v, size_t n, hipStream_t s);

// ---- projection / vote ----
// scan2RangeImg for keyframes [kb, kb+nb): scan_img[(kf-kb)*npx + px] = min range bits
// smax_bits (nb entries, zero-initialised by the caller, may be null): per keyframe the float bits of the largest scan range
hipError_t scan_range_images(const float4* scans, const uint64_t* offsets_dev, size_t kb, size_t nb,
                             uint64_t first_pt, uint64_t n_pts, uint64_t max_kf_pts, Geom g, uint32_t* scan_img, uint32_t* smax_bits, hipStream_t s);
// squared-range bound image of the range-culled vote kernel from finished scan images (see k_scan_qbound)
// scan2RangeImg of keyframes [kb, kb+nb) into n_shapes (<= kMaxScanShapes) image shapes of one field of view in ONE pass over the points; imgs[j] pre-filled
// with the empty range, smax_bits[j] (nullable) zeroed.  Bit-identical to n_shapes calls of scan_range_images.
static constexpr int kMaxScanShapes = 8;
hipError_t scan_range_images_multi(const float4* scans, const uint64_t* offsets_dev, size_t kb, size_t nb, uint64_t n_pts, uint64_t max_kf_pts, Geom g,
                                   int n_shapes, const int* rows, const int* cols, uint32_t* const* imgs, uint32_t* const* smax_bits, hipStream_t s);
hipError_t scan_qbound(const uint32_t* scan_img, size_t n, float thr, float* qbound, hipStream_t s);
// bounds[6*t..] = {min xyz, max xyz} of map points [4096 t, 4096 (t+1))
hipError_t tile_bounds(const float4* map, size_t M, float* bounds, hipStream_t s);
hipError_t subtile_bounds(const float4* map, size_t M, float* bounds, hipStream_t s);   // 4 x 6 floats per 4096-point tile: its 1024-point quarters
// Kernel variants and diagnostics, held by the context (ltm_ctx::kopts, read from the environment at ltm_create) and handed to the launchers: K contexts
// on K host threads (ltm_run --gpus K) share no mutable state.  The non-default values are A/B baselines and timing diagnostics, not product paths.
struct KernelOpts {
    int map_kernel_variant = 2;   // LTM_MAP_KERNEL   exact arg-min image: 0 one global atomic per point, 1 LDS pre-reduction, 2 + workgroup-local arg-min pre-filter
    int vote_cull = 1;            // LTM_VOTE_CULL    1: mode-0 votes use k_vote_map_cull, 0: the exact-image kernel
    int tile_cull = 1;            // LTM_TILE_CULL    whole-tile range cull inside k_vote_map_cull
    int stats_blockmin = 0;       // LTM_STATS_BLOCKMIN  ltm_debug_cull_stats reports the exact-image kernel's survivor counts instead of the vote kernel's
};
// transformGlobalMapToLocal + map2RangeImg: map_img[(kf-kb)*npx+px] = min (range_bits<<32 | idx)
// inv_poses_dev: 12 doubles per keyframe (3x4 row-major).  b2l: 12 doubles, b2l_identity skips the arithmetic.
hipError_t map_range_images(const float4* map, size_t M, const double* inv_poses_dev, const float* approx_poses_dev, size_t kb, size_t nb,
                            HostMat34 b2l, int b2l_identity, Geom g, uint64_t* map_img, hipStream_t s, const KernelOpts& ko);
// occlusion cull of the exact-image kernel on large maps (see ltm_kernels.hip): pair t = tile * nb + keyframe.  flags: n_tiles * nb bytes,
// done: n_tiles * nb zeroed bytes, pos / list: n_tiles * nb uint32, count: one uint32, cmax: nb * rows * ceil(cols/8) uint32, temp: scan_temp_bytes(n_tiles * nb)
hipError_t occlusion_shell_pairs(const float* approx_poses_dev, size_t kb, size_t nb, const float* tile_bounds_dev, size_t n_tiles, Geom g, float r_lo, float r_hi,
                                 const uint64_t* img, int use_cmax, uint32_t* cmax, uint8_t* done, uint8_t* flags, uint32_t* pos, uint32_t* list, uint32_t* count,
                                 void* temp, size_t temp_bytes, hipStream_t s, uint32_t* dirty_rows = nullptr,
                                 const float* sub_bounds_dev = nullptr, uint8_t* submask = nullptr, unsigned long long* sub_stats = nullptr);   // submask[pair]: live 1024-point quarters of the tile (bit q), sub_stats: {quarters of live pairs, quarters left alive}
hipError_t map_range_images_pairs(const float4* map, size_t M, const double* inv_poses_dev, const float* approx_poses_dev, size_t kb, size_t nb,
                                  HostMat34 b2l, int b2l_identity, Geom g, uint64_t* map_img, const uint32_t* pairs, size_t n_pairs, hipStream_t s, const KernelOpts& ko,
                                  const uint8_t* submask = nullptr);
// calcDescrepancyAndParseDynamicPointIdx over nb images; labels[idx] = 1 for flagged points
hipError_t compare_and_flag(const uint32_t* scan_img, const uint64_t* map_img, size_t n_px_total, float thr, int mode,
                            uint8_t* labels, hipStream_t s);
// vote form of the above: mode 0 may cull points that provably cannot be flagged (needs the finished scan images)
// approx_poses_dev: 16 floats per keyframe {A[9], c_hi[3], c_lo[3], ok} with p_local ~= A (p - c) (see xform_approx)
hipError_t vote_map_range_images(const float4* map, size_t M, const double* inv_poses_dev, const float* approx_poses_dev, size_t kb, size_t nb,
                                 HostMat34 b2l, int b2l_identity, Geom g, const float* qbound_img, const float* tile_bounds_dev,
                                 const uint32_t* smax_bits_dev, float thr, int mode, uint64_t* map_img, hipStream_t s, const KernelOpts& ko);
hipError_t count_live_tiles(const float* approx_poses_dev, size_t kb, size_t nb, const float* tile_bounds_dev, size_t n_tiles,
                            const uint32_t* smax_bits_dev, float thr, unsigned long long* live_dev, hipStream_t s);
hipError_t cull_stats(unsigned long long* out2, int reset, hipStream_t s, int which_kernel);   // {survivors, points} since the last reset; 0 vote kernel, 1 exact-image kernel
hipError_t cull_check(const float* xyz_dev, size_t n, const HostMat34* T, const HostMat34* b2l, int b2l_identity, const float* approx_pose_dev,
                      Geom g, unsigned long long* bad_dev, hipStream_t s);
// n probe points on / beside the pixel-rounding boundaries of the image shape `g` (local frame, or moved into the map frame by `pose`): input of cull_check
hipError_t cull_probe_points(Geom g, size_t n, const HostMat34* pose, float* xyz_dev, hipStream_t s);
// generic single image with up to two explicit transforms (debug / parity)
hipError_t single_range_image(const float4* pts, size_t n, const HostMat34* T1, const HostMat34* T2, Geom g,
                              uint64_t* img, hipStream_t s);
hipError_t decode_image(const uint64_t* img, size_t npx, float* rimg, int32_t* ptidx, hipStream_t s);
// RViz images: diff of a scan image (float bits) and a decoded map image; JET colour mapping of float / int32 images (BGR8)
hipError_t viz_diff(const uint32_t* scan_bits, const float* map_r, size_t n, int mode, float* out, hipStream_t s);
hipError_t viz_colormap_f32(const float* src, size_t n, float a, float b, const uint8_t* lut, uint8_t* bgr, hipStream_t s);
hipError_t viz_colormap_i32(const int32_t* src, size_t n, double a, double b, const uint8_t* lut, uint8_t* bgr, hipStream_t s);
hipError_t debug_project(const float* xyz_dev, size_t n, Geom g, float* az_el_r, int32_t* row_col, hipStream_t s);
// exhaustive device self-check of the fast arithmetic forms against their exact definitions for (vfov, hfov):
// counts[0] rad2deg mismatches, counts[1] /vfov mismatches, counts[2] /hfov mismatches over all 2^32 binary32 inputs
hipError_t selfcheck_fast_math(float vfov, float hfov, unsigned long long* counts_dev, hipStream_t s);

// ---- compaction helpers (prefix sums via rocPRIM) ----
size_t scan_temp_bytes(size_t n);
// exclusive scan of (labels[i] != 0) into pos[i] (uint32); temp from scan_temp_bytes(n)
hipError_t exclusive_scan_u8(const uint8_t* labels, uint32_t* pos, size_t n, void* temp, size_t temp_bytes, hipStream_t s);
// exclusive scan of ((uint32)img[i] != 0)
hipError_t exclusive_scan_img_valid(const uint64_t* img, uint32_t* pos, size_t n, void* temp, size_t temp_bytes, hipStream_t s);
hipError_t exclusive_scan_u32(const uint32_t* in, uint32_t* out, size_t n, void* temp, size_t temp_bytes, hipStream_t s);
// kept/flagged split in ascending index order: pos = exclusive scan of labels
hipError_t partition_scatter(const float4* in, const uint8_t* labels, const uint32_t* pos, size_t n,
                             float4* kept, float4* flagged, hipStream_t s);
// reprojection gather: out[pos[i]] = local-frame map point of pixel i (if valid)
hipError_t reproject_gather(const uint64_t* img, const uint32_t* pos, size_t npx, size_t nb, const float4* map,
                            const double* inv_poses_dev, size_t kb, HostMat34 b2l, int b2l_identity,
                            float4* out, hipStream_t s);
// gather arbitrary positions of a u32 array: out[j] = (idx[j] < n ? in[idx[j]] : tail)
hipError_t image_bounds(const uint32_t* pos, const uint64_t* img, size_t npx, size_t nb, uint32_t* out, hipStream_t s);
hipError_t flag_bounds(const uint32_t* pos, const uint8_t* flag, size_t n, const uint64_t* offsets_dev, size_t kf0, uint64_t first, size_t nb,
                       uint32_t* out, hipStream_t s);
hipError_t gather_u64_by_u32(const uint64_t* in, const uint32_t* idx_dev, size_t n, uint64_t* out, hipStream_t s);
hipError_t gather_u32(const uint32_t* in, const uint64_t* idx_dev, size_t m, size_t n, uint32_t tail_value_index_n,
                      uint32_t* out, hipStream_t s);

// ---- transforms ----
// out[i] = xform(second, xform(first, in[i])) with per-keyframe `second` (merge: first=L2B, second=pose[kf])
hipError_t transform_cloud(const float4* in, size_t n, const HostMat34* t1, const HostMat34* t2, float4* out, hipStream_t s);   // one or two rigid transforms of a cloud
hipError_t transform_scans(const float4* in, const uint64_t* offsets_dev, size_t n_kf, uint64_t n_pts,
                           HostMat34 first, int first_identity, const double* per_kf_dev, float4* out, hipStream_t s);
// out[k] = a[k] ++ b[k] ++ c[k] per keyframe; c (and oc) may alias b with an all-zero-length offsets array
hipError_t zip_concat(const float4* a, const uint64_t* oa, const float4* b, const uint64_t* ob, const float4* c, const uint64_t* oc,
                      const uint64_t* out_off_dev, size_t n_kf, uint64_t n, float4* out, hipStream_t s);
hipError_t preclean_flags(const float4* in, uint64_t n, float radius, uint8_t* keep, hipStream_t s);

// ---- voxel centroid ----
// bbox[6] device floats encoded as ordered uint32: minx,miny,minz,maxx,maxy,maxz (init by bbox_init)
hipError_t bbox_init(uint32_t* bbox, hipStream_t s);
hipError_t bbox_reduce(const float4* pts, size_t n, uint32_t* bbox, hipStream_t s);
float      bbox_decode(uint32_t enc);
struct OctreeFrame { double minx, miny, minz, res; unsigned depth; };
hipError_t bbox_reduce_check(const float4* pts, size_t n, OctreeFrame cached_frame, uint32_t* bbox8, hipStream_t s);   // box + "strictly increasing codes under cached_frame" (bbox8[6] = 1 if not)
size_t voxel_heads_starts_temp_bytes(size_t n);
hipError_t voxel_heads_starts(const uint64_t* sorted_keys, size_t n, unsigned shift, uint32_t* starts, void* temp, uint32_t* total_out_dev, hipStream_t s);
hipError_t morton_keys(const float4* pts, size_t n, OctreeFrame f, uint64_t* keys, uint32_t* idx, hipStream_t s);
// scan-set (segmented) forms: one bbox / octree frame per keyframe, composite sort key (kf << shift) | morton
hipError_t bbox_reduce_seg(const float4* pts, const uint64_t* offsets_dev, size_t n_kf, uint64_t n, uint32_t* bbox, hipStream_t s);
hipError_t morton_keys_seg(const float4* pts, const uint64_t* offsets_dev, size_t n_kf, uint64_t n, const OctreeFrame* frames_dev,
                           unsigned shift, uint64_t* keys, uint32_t* idx, hipStream_t s);
// pcl::VoxelGrid of the session loader for every keyframe of a scan set (Session.cpp:284-289): per-keyframe frame computed on the host
struct VoxelGridFrame { float inv; int min_b[3]; int div_b[3]; int passthrough; };
hipError_t voxelgrid_keys_seg(const float4* pts, const uint64_t* offsets_dev, size_t n_kf, uint64_t n, const VoxelGridFrame* frames_dev,
                              uint64_t* keys, uint32_t* idx, hipStream_t s);
hipError_t voxelgrid_centroids(const float4* pts, const uint64_t* sorted_keys, const uint32_t* sorted_idx, const uint32_t* starts,
                               const VoxelGridFrame* frames_dev, size_t n_vox, size_t n, float4* out, hipStream_t s);
size_t sort_temp_bytes(size_t n);
hipError_t scan_total_to(const uint8_t* flags, const uint32_t* pos, size_t n, uint32_t* out_dev, hipStream_t s);
hipError_t sort_pairs_u64(const uint64_t* keys_in, uint64_t* keys_out, const uint32_t* val_in, uint32_t* val_out, size_t n,
                          unsigned end_bit, void* temp, size_t temp_bytes, hipStream_t s);
hipError_t head_flags(const uint64_t* keys, size_t n, uint8_t* heads, hipStream_t s, unsigned shift = 0);   // compares keys >> shift
// Order-preserving compression of the Morton code: bit positions that are a FUNCTION OF MORE SIGNIFICANT BITS for every key the
// cloud can produce never decide a comparison and are left out (fewer radix passes).  With the octree box centred on the data, an
// axis whose extent is well below the cube's side has its second, third ... bits tied to its top bit (z of a 10 m high map in a
// 205 m cube: 4 of 12 bits).  The kept positions are given as up to kMaxKeyRuns runs of consecutive bits: out |= ((code >> src) & mask) << dst.
static constexpr int kMaxKeyRuns = 24;
struct KeyCompress { int n_runs; unsigned bits; unsigned char src[kMaxKeyRuns], dst[kMaxKeyRuns]; uint64_t mask[kMaxKeyRuns]; };
// packed voxel path: (compressed Morton << idx_bits) | index in one word, keys-only sort over the code bits
hipError_t morton_keys_packed(const float4* pts, size_t n, OctreeFrame f, KeyCompress kc, unsigned idx_bits, uint64_t* keys, hipStream_t s);
size_t sort_keys_temp_bytes(size_t n);
hipError_t sort_keys_u64(const uint64_t* keys_in, uint64_t* keys_out, size_t n, unsigned begin_bit, unsigned end_bit, void* temp,
                         size_t temp_bytes, hipStream_t s);
hipError_t voxel_centroids_packed(const float4* pts, const uint64_t* sorted_keys, uint64_t idx_mask, const uint32_t* starts, size_t n_vox,
                                  size_t n, float4* out, hipStream_t s);
hipError_t compact_keys(const uint64_t* keys, const uint8_t* flags, const uint32_t* pos, size_t n, uint64_t* keys_out, hipStream_t s);
// sharded voxel grid: 4096-bin histogram of (key >> shift), range flags, order-preserving compaction of (key, index)
static constexpr int kVoxelKeyBins = 4096;
hipError_t key_histogram(const uint64_t* keys, size_t n, unsigned shift, uint32_t* hist, hipStream_t s);
hipError_t key_range_flags(const uint64_t* keys, size_t n, uint64_t lo, uint64_t hi, uint8_t* flags, hipStream_t s);
hipError_t compact_pairs(const uint64_t* keys, const uint32_t* idx, const uint8_t* flags, const uint32_t* pos, size_t n,
                         uint64_t* keys_out, uint32_t* idx_out, hipStream_t s);
// starts[u] = j for every head j (u = pos[j])
hipError_t segment_starts(const uint8_t* heads, const uint32_t* pos, size_t n, uint32_t* starts, hipStream_t s);
hipError_t voxel_centroids(const float4* pts, const uint32_t* sorted_idx, const uint32_t* starts, size_t n_vox, size_t n,
                           float4* out, hipStream_t s);

// ---- kNN ----
struct KnnGrid { double ox, oy, oz, inv_cell; long long nx, ny, nz; };
hipError_t cell_keys(const float4* pts, size_t n, KnnGrid g, uint64_t* keys, uint32_t* idx, hipStream_t s);
hipError_t gather_points(const float4* in, const uint32_t* idx, size_t n, float4* out, hipStream_t s);
hipError_t gather_u64(const uint64_t* in, const uint32_t* idx, size_t n, uint64_t* out, hipStream_t s);
struct HashEntry { uint64_t key; uint32_t start, end; };
hipError_t hash_build(const uint64_t* sorted_keys, const uint32_t* starts, size_t n_cells, size_t n_pts,
                      HashEntry* table, uint32_t table_mask, hipStream_t s);
// queries in scan sets: g = pose*(first*p) ; label = coexist ; local = b2l*(inv*g)
hipError_t knn_query_scans(const float4* scans, const uint64_t* offsets_dev, size_t kb, size_t ke, uint64_t first_pt, uint64_t n_pts, uint64_t max_kf_pts,
                           const double* poses_dev, const double* inv_poses_dev, HostMat34 b2l, int b2l_identity,
                           const float4* sorted_target, size_t Mt, KnnGrid g, const HashEntry* table, uint32_t table_mask,
                           int k, float thr, float cell2_lo, uint8_t* coexist, float4* local_out, hipStream_t s);
// two-phase form for k <= 4 (see ltm_kernels.hip): 64-byte buckets of quantised cell points decide "certainly coexist", the rest is
// compacted and searched exactly.  buckets: n_buckets x 64 bytes, initialised to 0xff; pos / queue: n_pts uint32 each; count: one uint32
hipError_t knn_bucket_build(const float4* sorted_target, const uint64_t* sorted_keys, const uint32_t* starts, size_t n_cells, size_t n_pts, KnnGrid g,
                            void* buckets, uint32_t n_buckets, hipStream_t s);
// sparse occupancy bitmap of the grid (4 x 4 x 4 cells per hashed 64-bit word): word_mask + 1 zeroed uint64 words (a power of two); lets the
// exact search skip empty cells
hipError_t knn_bitmap_build(const uint64_t* sorted_keys, const uint32_t* starts, size_t n_cells, KnnGrid g, void* occ_words, uint32_t word_mask, hipStream_t s);
// phase 1: local_out for every query, coexist[i] = 1 (certainly coexist) / 0 (certainly diff: outside the grid) / 2 (undecided)
hipError_t knn_two_phase_fast(const float4* scans, const uint64_t* offsets_dev, size_t kb, size_t ke, uint64_t first_pt, uint64_t n_pts, uint64_t max_kf_pts,
                              const double* poses_dev, const double* inv_poses_dev, HostMat34 b2l, int b2l_identity, KnnGrid g, const void* buckets,
                              uint32_t n_buckets, int k, float thr, uint8_t* coexist, float4* local_out, hipStream_t s,
                              float4* gp_undecided = nullptr);      // gp_undecided[i] = the query's global point where coexist[i] == 2: phase 2 starts from it
// phase 2: the undecided queries are compacted (scan + scatter) and searched exactly; *count = their number
hipError_t knn_two_phase_exact(const float4* scans, const uint64_t* offsets_dev, size_t kb, size_t ke, uint64_t first_pt, uint64_t n_pts,
                               const double* poses_dev, HostMat34 b2l, int b2l_identity, const float4* sorted_target, size_t Mt, KnnGrid g,
                               const HashEntry* table, uint32_t table_mask, const void* bitmap, uint32_t bitmap_mask, int k, float thr, float cell2_lo, uint8_t* coexist,
                               uint32_t* pos, uint32_t* queue, uint32_t* count, void* temp, size_t temp_bytes, hipStream_t s);
// phase 2 over a queue sorted by cell id (queue word = cell id << ibits | query index): compaction, [host reads *count], sort + exact search
unsigned knn_sorted_queue_bits(KnnGrid g, uint64_t n_pts, unsigned* ibits_out);
hipError_t knn_two_phase_compact_keyed(const float4* scans, const uint64_t* offsets_dev, size_t kb, size_t ke, uint64_t first_pt, uint64_t n_pts, const double* poses_dev,
                                       HostMat34 b2l, int b2l_identity, KnnGrid g, unsigned ibits, const uint8_t* flags, uint32_t* pos, uint64_t* queue, uint32_t* count,
                                       void* temp, size_t temp_bytes, hipStream_t s, const float4* gp_undecided = nullptr);
hipError_t knn_two_phase_exact_sorted(const float4* scans, const uint64_t* offsets_dev, size_t kb, size_t ke, uint64_t first_pt, const double* poses_dev, HostMat34 b2l,
                                      int b2l_identity, const float4* sorted_target, size_t Mt, KnnGrid g, const HashEntry* table, uint32_t table_mask, const void* bitmap,
                                      uint32_t bitmap_mask, int k, float thr, float cell2_lo, uint8_t* coexist, const uint64_t* queue_in, uint64_t* queue_sorted, uint32_t n_und,
                                      unsigned ibits, unsigned kbits, void* temp, size_t temp_bytes, hipStream_t s, const float4* gp_undecided = nullptr);
hipError_t knn_query_cloud(const float4* query, size_t Q, const float4* sorted_target, size_t Mt, KnnGrid g,
                           const HashEntry* table, uint32_t table_mask, int k, float thr, float cell2_lo,
                           uint8_t* near, hipStream_t s);

} // namespace ltm
