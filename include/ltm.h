/*
 * ltm.h -- C ABI of libltm_hip.so: the MI355X (gfx950) implementation of the LT-removert /
 * LT-map hot path of gisbi-kim/lt-mapper (package `removert`).
 *
 * The reference has no plugin/FFI boundary: Removerter/Session are
 * monolithic C++ classes.  This header is the boundary a maintainer binds instead of the
 * PCL/OpenCV/OpenMP bodies of the functions cited on each entry point below; the host-side
 * mirror of the class surface that calls it lives in lt-mapper_amd/host/ (C++) and
 * the ctypes binding lt-mapper_amd/capi.py (used by the tests and bench.py).  INTEGRATION.md shows the
 * reference-side patch.
 *
 * Conventions
 *  - C linkage, POD only, no torch/PCL/Eigen types.  Every call returns LTM_OK (0) or a
 *    negative LTM_E_* code; the library never throws and never exits.  ltm_last_error()
 *    gives the message for the last failing call on that context.
 *  - A context owns one HIP stream and all device memory reachable through its handles.
 *    Calls on ONE context serialise on a mutex of that context, so a handle may be freed from any
 *    thread; the pipeline of one context is still driven from one host thread (as the reference's
 *    run() is).  Independent chains of a run -- the two sessions' remove / revert passes
 *    (Removerter.cpp:1580-1587), the ND and PD filters (:1395-1411), the reprojections of Step 3
 *    (:1534-1575) -- may run side by side on LANES: further contexts on the same device, one host
 *    thread each, that exchange clouds without copies (section "lanes" below).
 *  - Clouds are XYZI float32, 16 B / point on the device.  Host buffers are described by a
 *    byte stride (16 = packed, 32 = pcl::PointXYZI: xyz+pad, intensity+pad).
 *  - Matrices are 4x4 row-major double (the layout of the pose text files, Session.cpp:102-114).
 *  - The library never keeps a host pointer after a call returns.  Calls are synchronous
 *    with respect to the host unless stated otherwise.
 *  - There is no CPU fallback: without a usable gfx950 device ltm_create() fails.
 */
#ifndef LTM_H
#define LTM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LTM_ABI_VERSION 1

enum {
    LTM_OK = 0,
    LTM_E_INVALID = -1,    /* bad argument / bad handle */
    LTM_E_DEVICE = -2,     /* HIP runtime error, no device */
    LTM_E_NOMEM = -3,      /* device or host allocation failed */
    LTM_E_UNSUPPORTED = -4 /* outside the supported domain (e.g. octree depth > 21) */
};

typedef struct ltm_ctx ltm_ctx;
typedef uint64_t ltm_cloud;   /* device-resident XYZI cloud; 0 is never a valid handle */
typedef uint64_t ltm_scanset; /* N per-keyframe clouds packed in one device array + offsets[N+1] */
typedef uint64_t ltm_poses;   /* N keyframe poses + inverse poses (double, device + host copy) */

/* RosParamServer.cpp:15-33 -- the parameters the hot path reads */
typedef struct {
    float  vfov, hfov;        /* removert/sequence_vfov, sequence_hfov (defaults 50, 360) */
    double lidar2base[16];    /* removert/ExtrinsicLiDARtoPoseBase; its inverse is derived inside */
    int    device;            /* HIP device ordinal (LOCAL_RANK for multi-process runs) */
    int    max_kf_batch;      /* keyframes whose range images are resident at once; 0 = default (512) */
} ltm_config;

int         ltm_abi_version(void);
int         ltm_create(const ltm_config* cfg, ltm_ctx** out);
void        ltm_destroy(ltm_ctx* ctx);
const char* ltm_last_error(const ltm_ctx* ctx);
int         ltm_synchronize(ltm_ctx* ctx);
/* drops derived data the context keeps between calls (finished scan range images, keyed by scan set + image shape).
 * Results never depend on it; benchmarks call it so that every timed run does the work of a fresh run. */
int         ltm_clear_caches(ltm_ctx* ctx);
/* the hipStream_t every kernel of this context is launched on (for events / interop) */
void*       ltm_stream(ltm_ctx* ctx);

/* ------------------------------------------------------------------ clouds ---- */
int ltm_cloud_upload(ltm_ctx*, const void* pts, size_t n, size_t stride_bytes, ltm_cloud* out);
int ltm_cloud_from_device(ltm_ctx*, const void* dev_xyzi, size_t n, ltm_cloud* out);  /* D2D copy of packed float4 */
int ltm_cloud_size(ltm_ctx*, ltm_cloud, size_t* n);
int ltm_cloud_download(ltm_ctx*, ltm_cloud, void* dst, size_t cap_pts, size_t stride_bytes);
int ltm_cloud_device_ptr(ltm_ctx*, ltm_cloud, const void** dev_xyzi);                 /* borrowed, valid until free */
int ltm_cloud_clone(ltm_ctx*, ltm_cloud, ltm_cloud* out);                              /* `*a = *b` deep copies */
int ltm_cloud_concat(ltm_ctx*, const ltm_cloud* in, size_t n, ltm_cloud* out);         /* `*a += *b` (order kept) */
/* pcl::transformPointCloud(cloud, out, Matrix4d) applied once or twice (row-major 4x4, NULL = skip), the float result of the first
 * being the input of the second: local2global = (T1 = lidar->base, T2 = pose), global2local / transformGlobalMapToLocal =
 * (T1 = inverse pose, T2 = base->lidar)  (utility.cpp:64-72, 160-168, 194-202) */
int ltm_cloud_transform(ltm_ctx*, ltm_cloud in, const double* T1, const double* T2, ltm_cloud* out);
/* pcl::ExtractIndices as used by parsePointcloudSubsetUsingPtIdx (Removerter.cpp:933-946): out[j] = in[idx[j]], order kept;
 * an index outside [0, n) is LTM_E_INVALID (undefined behaviour in the reference) */
int ltm_cloud_select(ltm_ctx*, ltm_cloud in, const int32_t* idx_host, size_t n_idx, ltm_cloud* out);
int ltm_cloud_free(ltm_ctx*, ltm_cloud);

/* an UNINITIALISED cloud of n points whose device array (ltm_cloud_device_ptr) a collective fills: the all-gather that assembles a
 * map from per-rank pieces writes straight into it (north_star: "RCCL all-gather over xGMI to assemble the final maps") */
int ltm_cloud_alloc(ltm_ctx*, size_t n, ltm_cloud* out);

/* ---------------------------------------------------------------- scan sets ---- */
/* keyframe_scans_ and friends (Session.h:41-58): offsets[n_kf+1] in points */
int ltm_scanset_upload(ltm_ctx*, const void* pts, size_t stride_bytes, const uint64_t* offsets, size_t n_kf, ltm_scanset* out);
int ltm_scanset_from_device(ltm_ctx*, const void* dev_xyzi, const uint64_t* host_offsets, size_t n_kf, ltm_scanset* out);
int ltm_scanset_info(ltm_ctx*, ltm_scanset, size_t* n_kf, size_t* n_points);
int ltm_scanset_offsets(ltm_ctx*, ltm_scanset, uint64_t* offsets /* n_kf+1 */);
int ltm_scanset_keyframe(ltm_ctx*, ltm_scanset, size_t kf, ltm_cloud* out);            /* one keyframe's scan as a cloud (D2D copy) */
int ltm_scanset_download(ltm_ctx*, ltm_scanset, void* dst, size_t cap_pts, size_t stride_bytes);
int ltm_scanset_device_ptr(ltm_ctx*, ltm_scanset, const void** dev_xyzi);
int ltm_scanset_as_cloud(ltm_ctx*, ltm_scanset, ltm_cloud* out);                       /* flat copy of all points */
/* concatenate per-rank scan sets keyframe-wise (multi-GPU assembly): out has sum(n_kf) keyframes */
int ltm_scanset_concat(ltm_ctx*, const ltm_scanset* in, size_t n, ltm_scanset* out);
/* per-keyframe `a[i] += b[i] (+= c[i])` of Session.cpp:365-371 ; c may be 0 */
int ltm_scanset_zip_concat(ltm_ctx*, ltm_scanset a, ltm_scanset b, ltm_scanset c, ltm_scanset* out);
/* the same for a scan set: keyframe layout given by host offsets[n_kf+1], points uninitialised (ltm_scanset_device_ptr) */
int ltm_scanset_alloc(ltm_ctx*, const uint64_t* offsets, size_t n_kf, ltm_scanset* out);
int ltm_scanset_free(ltm_ctx*, ltm_scanset);

/* ------------------------------------------------------- raw device buffers ---- */
/* Label masks and staging areas of the multi-GPU host: pooled device memory, ordered on the context's stream
 * (ltm_stream) -- a collective enqueued on that stream needs no extra synchronisation.  ltm_buffer_copy returns when the copy
 * has completed; kind: 0 = host->device, 1 = device->host, 2 = device->device. */
int ltm_buffer_alloc(ltm_ctx*, size_t bytes, void** dev);
int ltm_buffer_free(ltm_ctx*, void* dev);
int ltm_buffer_fill(ltm_ctx*, void* dev, int byte_value, size_t bytes);            /* asynchronous on the context's stream */
int ltm_buffer_copy(ltm_ctx*, void* dst, const void* src, size_t bytes, int kind);

/* --------------------------------------------- pipelined feeder / asynchronous fetch ---- */
/* f-1 (Session.cpp:266-302): a scan set assembled from chunks of consecutive keyframes WHILE the host is still
 * decoding the next files.  Each chunk is staged through one of two pinned buffers and copied on a dedicated copy stream, so the
 * host-side packing of chunk i+1 overlaps the DMA of chunk i (and both overlap the decode threads).  capacity_points is an upper
 * bound of the total (the POINTS fields of the PCD headers).  ltm_scanset_upload_chunk returns as soon as `pts` may be reused.
 * Two-stream rule: the device array comes from the context's stream-ordered pool, so ltm_scanset_upload_begin makes the copy stream
 * wait for everything already queued on the compute stream (a recycled block may still be read by queued kernels), and
 * ltm_scanset_upload_end waits for the copies before the scan set is handed to the compute stream.  An upload may therefore be
 * started at any time, not only while the context is idle. */
typedef uint64_t ltm_upload;
int ltm_scanset_upload_begin(ltm_ctx*, size_t capacity_points, ltm_upload* up);
int ltm_scanset_upload_chunk(ltm_ctx*, ltm_upload up, const void* pts, size_t stride_bytes, const uint64_t* kf_sizes, size_t n_kf);
int ltm_scanset_upload_end(ltm_ctx*, ltm_upload up, ltm_scanset* out);

/* f-2 (Removerter.cpp:1637-1650, 1446-1520): asynchronous device->host fetch for the output writer.  *_fetch_begin enqueues the
 * copy of the cloud / scan set as it is after everything already submitted to the context, on the copy stream, into a pinned host
 * buffer owned by the library; the context keeps computing.  ltm_fetch_wait blocks until that copy has completed and may be
 * called from ANY host thread (writer threads) -- it touches only the ticket.  The points are packed XYZI (16 B); for a scan set
 * `offsets` (n_kf + 1 entries, owned by the ticket) delimit the keyframes.  The source handle must stay alive until the wait has
 * returned; ltm_fetch_release recycles the pinned buffer and may also be called from any thread, as soon as the data has been
 * consumed (page-locking new buffers is what costs time, so hand them back early). */
typedef struct ltm_fetch ltm_fetch;
int ltm_cloud_fetch_begin(ltm_ctx*, ltm_cloud, ltm_fetch** out);
int ltm_scanset_fetch_begin(ltm_ctx*, ltm_scanset, ltm_fetch** out);
int ltm_fetch_wait(ltm_fetch*, const void** host_xyzi, size_t* n_points, const uint64_t** offsets, size_t* n_kf);
int ltm_fetch_release(ltm_ctx*, ltm_fetch*);
/* Chunked form of the same: nothing the size of an output is page-locked.  The context keeps a small ring of pinned chunks
 * (8 x 32 MB; LTM_FETCH_SLOTS, LTM_FETCH_CHUNK_MB) and a copier thread that, once the compute stream has reached the point of the
 * *_fetch_chunks_begin call, moves the points chunk by chunk into free ring slots.  Consumers -- any number of threads per ticket
 * -- take chunks with ltm_fetch_next_chunk (blocks; 1 = a chunk, 0 = no more, < 0 = error) and hand every one back with
 * ltm_fetch_chunk_done as soon as they have used it (the copier waits for free slots, the context thread never does).  A cloud
 * comes in consecutive ranges of points; a scan set in chunks of WHOLE keyframes [first_kf, first_kf + n_kf), whose points start
 * at offsets[first_kf] - first_point inside the chunk (LTM_E_INVALID from *_begin if one keyframe is larger than a chunk).
 * ltm_fetch_info gives the totals right away (a writer needs them for its header).
 * Contract of the chunked form (the in-tree writer, host/src/Removerter.cpp, is the model):
 *  - ONE copier thread serves the tickets strictly in the order of their *_begin calls through the shared ring, so tickets must
 *    also be CONSUMED in that order: a ticket nobody drains (or consumers that all block on a later ticket) stalls every later one;
 *  - the device source (cloud / scan set handle) must stay alive and unmodified until ltm_fetch_release has returned: the copier
 *    reads it asynchronously;
 *  - call ltm_fetch_release (any thread) exactly once, after every consumer thread of the ticket has seen ltm_fetch_next_chunk
 *    return 0 or an error and has handed back its chunks: release frees the ticket, a thread still inside ltm_fetch_next_chunk
 *    would touch freed memory. */
int ltm_cloud_fetch_chunks_begin(ltm_ctx*, ltm_cloud, ltm_fetch** out);
int ltm_scanset_fetch_chunks_begin(ltm_ctx*, ltm_scanset, ltm_fetch** out);
int ltm_fetch_info(ltm_fetch*, size_t* n_points, const uint64_t** offsets, size_t* n_kf);
int ltm_fetch_next_chunk(ltm_fetch*, const void** host_xyzi, size_t* first_point, size_t* n_points, size_t* first_kf, size_t* n_kf);
int ltm_fetch_chunk_done(ltm_fetch*, const void* host_xyzi);

/* -------------------------------------------------------------------- poses ---- */
/* keyframe_poses_ / keyframe_inverse_poses_ (Session.cpp:102-114).  inv may be NULL: then the
 * inverse is ltm_inverse4x4 of every pose. */
int ltm_poses_create(ltm_ctx*, size_t n_kf, const double* poses, const double* inv_or_null, ltm_poses* out);
/* `Eigen::Matrix4d::inverse()` as the reference uses it for the poses (Session.cpp:109-110) and the extrinsic
 * (RosParamServer.cpp:29-30): general 4x4 inverse in double with the operation order of Eigen 3.3.7's SSE2 kernel (2x2 block
 * adjugates; restated).  Row-major in and out, needs no context or device.  The ONE inverse of the
 * product: ltm_create uses it for the extrinsic, ltm_poses_create for missing inverses, the C++ host for the pose files.
 * Returns LTM_OK, or LTM_E_INVALID for a null argument or a singular / non-finite matrix (Eigen would return inf / NaN entries). */
int ltm_inverse4x4(const double* m16, double* inv16);
int ltm_poses_free(ltm_ctx*, ltm_poses);

/* ------------------------------------------------- the hot path, one call per stage ---- */

/* Session.cpp:506-533 precleaningKeyframes: drop points with range<radius & |z|<0.5 */
int ltm_preclean(ltm_ctx*, ltm_scanset in, float radius, ltm_scanset* out);

/* utility.cpp:170-192 mergeScansWithinGlobalCoordUtil / Session.cpp:186-202: local -> global concat */
int ltm_merge_to_global(ltm_ctx*, ltm_scanset scans, ltm_poses poses, ltm_cloud* out);

/* utility.cpp:204-219 octreeDownsampling (PCL OctreePointCloudVoxelCentroid): voxel centroids in octree DFS order */
int ltm_voxel_centroid(ltm_ctx*, ltm_cloud in, float leaf, ltm_cloud* out);
/* the same for n independent clouds at once (e.g. the static and the dynamic map after a vote pass, Removerter.cpp:894-903; the eight
 * maps of Removerter.cpp:1445-1476): identical results, but the host reads all bounding boxes and all voxel counts in two round
 * trips for the whole batch instead of two per cloud */
int ltm_voxel_centroid_batch(ltm_ctx*, size_t n, const ltm_cloud* in, const float* leaf, ltm_cloud* out);
/* multi-GPU form of the above: only the voxels of shard `shard` of `n_shards`.  The Morton key space of the
 * octree is cut into n_shards contiguous ranges of about equal point count (a pure function of `in`), so the outputs of
 * shards 0..n_shards-1 concatenated in order ARE ltm_voxel_centroid(in); each rank sorts 1/n_shards of the points. */
int ltm_voxel_centroid_shard(ltm_ctx*, ltm_cloud in, float leaf, uint32_t shard, uint32_t n_shards, ltm_cloud* out);
/* Key-range exchange: the voxel grid of a cloud whose POINTS are spread over the ranks -- the merge of
 * rank-local per-keyframe scans (utility.cpp:170-192 followed by :204-219) -- without gathering the points on every rank.  Every rank:
 *   ltm_cloud_bbox            box of its own points            -> the ranks combine them (min / max): the box of the whole cloud
 *   ltm_voxel_key_histogram   4096-bin histogram of its points' octree keys under the WHOLE cloud's frame (top bits of the compressed
 *                             Morton code)                     -> summed over the ranks; cut into n contiguous bin ranges of equal weight
 *   ltm_voxel_key_split       its points of every range, each in input order -> all-to-all: rank r receives range r from everybody, in
 *                             rank order = keyframe order = the input order of the single-GPU merge restricted to the range
 *   ltm_voxel_centroid_box    the voxel grid of what it received, under the whole cloud's frame (the box is given, not derived)
 * and the outputs of ranks 0..n-1 concatenated ARE ltm_voxel_centroid of the merged cloud: a bin is a prefix of the key, so no voxel
 * straddles a cut, and every voxel sums its points in the order the single-GPU grid does.  hist has 4096 entries; cut_bins n_parts + 1,
 * cut_bins[0] = 0, cut_bins[n_parts] = 4096.  An empty cloud has the box (+inf, -inf) and an all-zero histogram. */
int ltm_cloud_bbox(ltm_ctx*, ltm_cloud in, float mn[3], float mx[3]);
int ltm_voxel_key_histogram(ltm_ctx*, ltm_cloud in, const float mn[3], const float mx[3], float leaf, uint32_t hist[4096]);
int ltm_voxel_key_split(ltm_ctx*, ltm_cloud in, const float mn[3], const float mx[3], float leaf, uint32_t n_parts, const uint32_t* cut_bins, ltm_cloud* parts);
int ltm_voxel_centroid_box(ltm_ctx*, ltm_cloud in, const float mn[3], const float mx[3], float leaf, ltm_cloud* out);
/* the same applied to every keyframe of a scan set (Session.cpp:362-380 updateScansScanwise) */
int ltm_voxel_centroid_scanset(ltm_ctx*, ltm_scanset in, float leaf, ltm_scanset* out);

/* pcl::VoxelGrid as the session loader applies it to every scan it reads (Session.cpp:284-289, leaf = downsample_voxel_size), for all
 * keyframes of a scan set at once: centroids (x, y, z, intensity) per occupied leaf in ascending leaf index; a keyframe whose grid
 * would exceed INT32_MAX cells is returned unchanged (PCL's "leaf size is too small" early-out -- the common case for a raw 0.05 m
 * scan).  The float sums of a leaf run in the order PCL's std::sort on the leaf index leaves its points in (the permutation
 * is computed on host threads, one keyframe per task, by a restatement of libstdc++'s introsort that makes the same element moves without
 * the branch mispredictions -- csrc/ltm_pclsort.h, checked against std::sort itself; bit-identical to
 * the reference's sources compiled against stand-in PCL headers, oracle/_ref -- PCL itself is not available, so "as PCL 1.10 is understood to do it").  LTM_VOXELGRID_ORDER=input sums in input
 * order entirely on the device: faster, but the last bit of a centroid of three or more points may differ.  This is what makes a device-resident cascade hand over the scans the reference would
 * re-load from scans_updated/ (README.md:115-118; Removerter.cpp:1658-1660). */
int ltm_voxel_grid_scanset(ltm_ctx*, ltm_scanset in, float leaf, ltm_scanset* out);
/* The same in two halves, for a caller that has other work for the device meanwhile (the lifelong cascade: the next run's query session does not depend on
 * the re-loaded central scans -- Session.cpp:284-289 of the NEXT process, Removerter.cpp:1653-1660).  _begin enqueues the key kernels, sends the keys to the
 * host on the copy stream and starts the host threads that reproduce std::sort's order; it returns without waiting for either, and whatever is submitted to
 * the context afterwards runs beside them.  _end waits for the order and finishes the grid.  `in` must stay alive and unchanged until _end, which consumes
 * the ticket whatever it returns.  ltm_voxel_grid_scanset is _begin + _end back to back. */
typedef struct ltm_vgs ltm_vgs;
int ltm_voxel_grid_scanset_begin(ltm_ctx*, ltm_scanset in, float leaf, ltm_vgs** ticket);
int ltm_voxel_grid_scanset_end(ltm_ctx*, ltm_vgs* ticket, ltm_scanset* out);

/* Visibility vote, keyframes [kf_begin,kf_end) of `scans`/`poses` against `map`:
 *   scan2RangeImg (Removerter.cpp:109-156) + transformGlobalMapToLocal (utility.cpp:64-72) +
 *   map2RangeImg (utility.cpp:92-142) + calcDescrepancyAndParseDynamicPointIdx (Removerter.cpp:381-413),
 *   i.e. the loop body of calcDescrepancyAndParseDynamicPointIdxForEachScan[ForND|ForPD] (:429-593).
 * mode 0: diff = scan - map (remove / revert / PD);  mode 1: diff = map - scan (ND).
 * labels_dev: device buffer of M bytes; label 1 is OR-ed in for every flagged map point (the std::set
 * union of :589-590).  The caller zeroes it; ranks combine theirs with a MAX all-reduce. */
int ltm_visibility_vote(ltm_ctx*, ltm_cloud map, ltm_scanset scans, ltm_poses poses, size_t kf_begin, size_t kf_end,
                        float res_alpha, float diff_thres, int mode, uint8_t* labels_dev);
/* Optional: scan2RangeImg (Removerter.cpp:109-156) of keyframes [kf_begin,kf_end) of `scans` for ALL the listed resolutions in one pass over the points,
 * kept for the votes that follow (selfRemovert projects every scan at res and 0.95 res for each entry of remove_resolution_list: six image shapes whose
 * spherical coordinates are the same).  ltm_visibility_vote computes what it does not find, one shape at a time; the images are identical either way. */
int ltm_scanset_prepare_range_images(ltm_ctx*, ltm_scanset scans, size_t kf_begin, size_t kf_end, const float* res_alphas, size_t n_alphas);
/* partitionCurrentMap tail (Removerter.cpp:816-824, :675-687, :933-946): index-ascending split */
int ltm_partition_by_labels(ltm_ctx*, ltm_cloud map, const uint8_t* labels_dev, ltm_cloud* kept, ltm_cloud* flagged);
/* vote over all keyframes + partition (single-GPU convenience).  host_labels (M bytes) may be NULL. */
int ltm_visibility_partition(ltm_ctx*, ltm_cloud map, ltm_scanset scans, ltm_poses poses, float res_alpha,
                             float diff_thres, int mode, ltm_cloud* kept, ltm_cloud* flagged, uint8_t* host_labels);

/* parseProjectedPoints / Session::parseScansViaProjection (utility.cpp:74-89, Session.cpp:348-360):
 * out has (kf_end-kf_begin) keyframes of local-frame points, row-major pixel order, ptidx==0 dropped. */
int ltm_reproject(ltm_ctx*, ltm_cloud map, ltm_poses poses, size_t kf_begin, size_t kf_end, float res_alpha, ltm_scanset* out);

/* extractLowDynPointsViaKnnDiff / extractHighDynPointsViaKnnDiff (Session.cpp:393-427, 487-504, 537-642):
 * exact k-NN of every scan point (moved to the global frame) in `target`; coexist iff mean of the k squared
 * distances < thr.  Outputs are in the local frame, input order kept.  Either output may be NULL. */
int ltm_knn_partition(ltm_ctx*, ltm_cloud target, ltm_scanset scans, ltm_poses poses, size_t kf_begin, size_t kf_end,
                      int k, float thr, ltm_scanset* coexist, ltm_scanset* diff);
/* removeWeakNDMapPointsHavingStrongNDInNear (Session.cpp:452-484): split `query` by k-NN distance to `target` */
int ltm_knn_split_cloud(ltm_ctx*, ltm_cloud target, ltm_cloud query, int k, float thr, ltm_cloud* near, ltm_cloud* far);

/* ------------------------------------------------------------------- lanes ---- */
/* The reference runs the stages of Removerter::run() one after the other on one thread; several of them do not depend on each other: the
 * central and the query session's makeGlobalMap + Step-1 chains (Removerter.cpp:213-252, :1580-1587), their HD kNN maps and static reprojections
 * (:1590-1601, :1527-1538), the two directions of the LD kNN diff (:1418-1421), filterStrongND against filterStrongPD (:1395-1411), the grids of
 * :1445-1476 and the six reprojections of :1551-1577.  A LANE is a second context on the parent's device (own stream, own pool, the parent's
 * configuration and create-time self-check) that a second host thread drives; the projection kernels are bound by vector-instruction issue and the
 * grids / kNN stages by launch latency, so the two kinds of work fill each other's gaps.  The library keeps the lanes out of each other's way by
 * itself: the large projection launches of a context and its lanes are chained on the device (one at a time, in the order the hosts submit them) and
 * go to a stream of the lowest priority, so that one lane's partition + grids are dispatched ahead of -- and finish under -- the other lane's
 * projection instead of both lanes projecting together and then idling together.  Results do not depend on the schedule: every kernel sees the same
 * inputs as in the one-lane order.
 *
 *   ltm_lane_create           a context like `parent` (same device, field of view, extrinsic, kernel switches); destroy with ltm_destroy
 *   ltm_cloud_lend / _give    make a cloud of context `from` usable in context `to` WITHOUT copying: `lend` leaves ownership with `from` (the new handle
 *                             is a borrowed view: freeing it releases nothing; `from` must keep its cloud alive and unchanged until the borrower is done
 *                             with it and must order its own later writes / frees after the borrower's reads -- ltm_lane_fence(to, from) or an event);
 *                             `give` moves the memory block into `to`'s pool and invalidates the handle in `from`.  Both make `to`'s stream wait for what
 *                             `from`'s stream has been given so far (the cloud's producer).  Same for scan sets.
 *   ltm_lane_fence(a, b)      everything submitted to b afterwards runs after everything submitted to a so far (device-side; the host does not wait)
 *   ltm_event_record / _wait  the same dependency in two halves, for a consumer on another thread: record on one context, hand the event over by
 *                             whatever means the host language has, wait on the other.  An event may be waited for any number of times.
 * Two contexts are locked in address order by the two-context calls, so they may be issued from either thread. */
typedef struct ltm_event ltm_event;
int  ltm_lane_create(ltm_ctx* parent, ltm_ctx** lane);
int  ltm_lane_fence(ltm_ctx* done_in, ltm_ctx* before_next_of);
int  ltm_event_record(ltm_ctx*, ltm_event** ev);
int  ltm_event_wait(ltm_ctx*, ltm_event* ev);
void ltm_event_destroy(ltm_event* ev);
int  ltm_cloud_lend(ltm_ctx* from, ltm_cloud h, ltm_ctx* to, ltm_cloud* out);
int  ltm_cloud_give(ltm_ctx* from, ltm_cloud h, ltm_ctx* to, ltm_cloud* out);
int  ltm_scanset_lend(ltm_ctx* from, ltm_scanset h, ltm_ctx* to, ltm_scanset* out);
int  ltm_scanset_give(ltm_ctx* from, ltm_scanset h, ltm_ctx* to, ltm_scanset* out);
/* ------------------------------------------------------ parity / debug helpers ---- */
/* one range image of `pts` after optional transforms T1 then T2 (NULL = none); rimg R*C floats,
 * ptidx R*C int32 or NULL.  Host output buffers. */
int ltm_debug_range_image(ltm_ctx*, ltm_cloud pts, const double* T1, const double* T2, float res_alpha,
                          float* rimg, int32_t* ptidx);
/* The four RViz images of one keyframe (Removerter.cpp:580-585 pubRangeImg x4; utility.h:114-127 convertColorMappedImg =
 * 255*(img-min)/(max-min) -> 8 bit -> cv::COLORMAP_JET), computed and colour-mapped on the device so that a ROS host only
 * pays for them when somebody subscribes: scan range image, map range image (both on [range_min, range_max] =
 * rimg_color_min/max), their difference (scan-map for mode 0, map-scan for mode 1, on [diff_min, diff_max] = 0..0.5 in
 * RosParamServer.cpp:12) and the map point-index image (on [0, M]).  Each output is rows*cols*3 bytes BGR8
 * (ltm_rimg_size) in host memory, or NULL to skip it. */
int ltm_debug_viz_images(ltm_ctx*, ltm_cloud map, ltm_scanset scans, ltm_poses poses, size_t kf, float res_alpha, int mode,
                         float range_min, float range_max, float diff_min, float diff_max,
                         uint8_t* scan_bgr, uint8_t* map_bgr, uint8_t* diff_bgr, uint8_t* ptidx_bgr);
/* element-wise device evaluation of the projection arithmetic: out_az_el_r (3n), out_row_col (2n) */
int ltm_debug_project(ltm_ctx*, const float* xyz, size_t n, float res_alpha, float* out_az_el_r, int32_t* out_row_col);
void ltm_rimg_size(float vfov, float hfov, float res_alpha, int* rows, int* cols);   /* utility.cpp:222-236 */
/* result of the create-time exhaustive device self-check of the fast arithmetic forms (rad2deg by multiplication,
 * division by the FOV constants) against plain IEEE division over all 2^32 binary32 inputs:
 * mismatches3 = {rad2deg, /vfov, /hfov}; the fast forms are used only when all three are zero. */
int ltm_debug_selfcheck(ltm_ctx*, uint64_t* mismatches3, int* fast_math_enabled);
/* the elevation polynomial of the bounded-error projection, as fitted when a context with this vertical field of view is created
 * (host arithmetic only, needs no device): atan(t) ~ t (c[0] + u (c[1] + u (c[2] + u c[3]))), u = t^2, on [0, tan(vfov/2 + 2 deg)];
 * *max_err_rad = largest error of its binary32 evaluation.  Returns 1 if the kernels use it for this field of view (vfov/2 + 2 deg
 * <= 45 deg and error <= 1e-6 rad), 0 if they keep the generic polynomial on [0, 1], < 0 on invalid arguments. */
int ltm_debug_elevation_fit(float vfov_deg, float* c4, double* max_err_rad);
/* Host arithmetic only: the point order ltm_voxel_grid_scanset's default (PCL) path gives one keyframe whose points have the leaf indices
 * leaf_idx[0..n) -- the permutation std::sort with pcl::VoxelGrid's leaf-index-only comparator leaves behind (voxel_grid.hpp), computed by
 * lt-mapper_amd/csrc/ltm_pclsort.h (use_std_sort == 0) or by std::sort itself (!= 0); the two must agree (tests/test_abi.py). */
int ltm_debug_pcl_sort_order(const uint32_t* leaf_idx, size_t n, uint32_t* order_out, int use_std_sort, uint32_t* heap_sort_fallbacks /* nullable: how
                             often the restatement went into introsort's heap-sort fallback (0 for std::sort, which does not say) */);
/* the voxel grid's sort key (host arithmetic only, needs no device): for a cloud with bounding box [mn, mx] and leaf size `leaf`, the
 * octree frame (depth, lattice origin) and the mask of the interleaved Morton-code bits (x = bit 3L+2, y = 3L+1, z = 3L of level L)
 * that the radix sort looks at -- the others are functions of more significant bits for every key inside the box and are left out
 * (fewer passes; order and equality of codes are unchanged).  Returns the number of kept bits, or < 0. */
int ltm_debug_voxel_key_bits(const float* mn3, const float* mx3, float leaf, uint64_t* kept_mask, unsigned* depth, double* frame_min3);
/* checks the bounded-error projection that the range-culled vote kernel uses to decide which points need the exact
 * arithmetic: counts points (host xyz, n*3 floats; global frame if inv_pose16 is given, else local) whose exact pixel /
 * range fall outside its candidate set / bounds.  Must be 0. */
int ltm_debug_cull_check(ltm_ctx*, const float* xyz, size_t n, const double* inv_pose16_or_null, float res_alpha, uint64_t* violations);
/* The culled projection kernels (k_vote_map_cull, the pre-filter of k_map_rimg_blockmin: map2RangeImg, utility.cpp:92-142, evaluated exactly only where it can
 * matter) rest on error bounds of a bounded-error projection.  Every range-image shape is validated on the device the first time a context uses it -- 2^20
 * probe points on and beside the pixel-rounding boundaries, in the sensor frame and through one keyframe pose of the call -- and a shape that leaves the
 * bounds is served by the exact kernels from then on (a line on stderr says so).  Counters since ltm_create: shapes checked / shapes that failed. */
int ltm_debug_cull_validation(ltm_ctx*, uint64_t* shapes_checked, uint64_t* shapes_failed);
/* diagnostic counters of the occlusion cull in front of the exact-image kernel on large maps (reprojection, ND votes) since the
 * last reset: (tile, keyframe) pairs seen by culled launches, pairs of the first distance shell, pairs projected in all (the rest was
 * proven hidden and dropped) */
int ltm_debug_occlusion_stats(ltm_ctx*, uint64_t* pairs, uint64_t* first_shell, uint64_t* projected, int reset);
/* voxel grids of clouds since the last reset, and how many of them were recognised as the identity during their bounding-box pass (the
 * input an order-preserving subset of an earlier grid's output, still in octree order and one point per voxel under the frame this
 * call derives: output = copy of the input, bit for bit what the sort + centroid path gives) */
int ltm_debug_voxel_stats(ltm_ctx*, uint64_t* grids, uint64_t* identity_hits, int reset);
/* diagnostic counters of the range-culled vote kernel since the last reset: points tested / points that needed the exact path */
int ltm_debug_cull_stats(ltm_ctx*, uint64_t* survivors, uint64_t* points, int reset);

/* ----------------------------------------------------------- measurement ---- */
/* Per-kernel-class HIP-event timing on the context's stream.  Classes: "vote_map", "vote_scan",
 * "vote_compare", "reproject_map", "knn_query", "voxel", .... */
int ltm_profile_enable(ltm_ctx*, int on);
int ltm_profile_reset(ltm_ctx*);
/* returns number of classes; fills up to cap entries.  units = class-specific work count
 * (point-projections for vote_map), bytes = algorithmic bytes, ms = sum of event durations */
int ltm_profile_read(ltm_ctx*, const char** names, double* ms, uint64_t* launches, double* units, double* bytes, int cap);
/* same class order as ltm_profile_read: the COMPULSORY bytes of each class's launches as this library issues them -- a projection
 * launch (map2RangeImg for a batch of keyframes, utility.cpp:92-142 under the loops Removerter.cpp:555 / Session.cpp:354) reads its map
 * once for the whole batch, where SURVEY.md 8d's figure counts one map read per keyframe; equal to `bytes` for every other class */
int ltm_profile_read_compulsory(ltm_ctx*, double* bytes_c, int cap);

#ifdef __cplusplus
}
#endif
#endif
