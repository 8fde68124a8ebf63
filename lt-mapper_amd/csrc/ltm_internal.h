// ltm_internal.h -- what the translation units of the C ABI (ltm_api_*.cpp) share: error type, the stream-ordered pool, handle tables, the context,
// scratch / profiling / transfer helpers and the entry-point guard.  Internal to libltm_hip.so (include/ltm.h is the interface).  No CPU fallback: every
// stage runs on the device or fails with an error code.  One stage has a host HALF by design: ltm_voxel_grid_scanset's default order -- the permutation
// pcl::VoxelGrid's std::sort leaves the points of a voxel in is a property of libstdc++'s introsort, reproduced on host threads (ltm_pclsort.h, which
// assumes that library's algorithm: checked against std::sort itself in tests/test_abi.py); keys down, point order up, everything else on the device.
#pragma once
#include "ltm.h"
#include "ltm_pclsort.h"
#include "ltm_kernels.h"

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

using namespace ltm;


namespace ltm_detail {

struct Err { int code; std::string msg; };

#define LTM_HIP(expr)                                                                               \
    do {                                                                                            \
        hipError_t e__ = (expr);                                                                    \
        if (e__ != hipSuccess) throw Err{LTM_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e__)}; \
    } while (0)
#define LTM_REQUIRE(cond, msg)                                  \
    do {                                                        \
        if (!(cond)) throw Err{LTM_E_INVALID, std::string(msg)}; \
    } while (0)

// Stream-ordered caching allocator: kernels and copies of a context are issued on ONE stream (c->stream), so a
// block may be handed out again as soon as the host has released it.  The two users of the second (copy) stream order
// themselves explicitly: ltm_scanset_upload_begin makes the copy stream wait for the compute stream before the first DMA
// into a (possibly recycled) block, ltm_scanset_upload_end drains the copy stream; fetches record an event on the compute
// stream and their sources are kept alive by the caller.  hipMalloc/hipFree (which synchronise the device) happen only on
// first use of a size class and at ltm_destroy().
// Lanes: a block that ltm_*_give moved into another context's pool goes HOME when it is freed there (through the home pool's inbox, with an event of
// the freeing stream that the home stream waits for before the block is handed out again).  Without this every pair run would move ~1.5 GB from the
// lane's pool into the main one for good: the lane would hipMalloc -- which waits for the whole device, both lanes' queues included -- some twenty
// times per run and the main pool would grow without bound (measured: profiles/r6_lanes_pool_migration.txt).
struct PoolInbox {
    struct Item { void* p; size_t bytes; hipEvent_t ev; };
    std::mutex mx;
    std::vector<Item> items;
    bool closed = false;            // the home pool is gone: returned blocks are released instead
};
struct Pool {
    std::multimap<size_t, void*> free_blocks;
    std::unordered_map<void*, size_t> live;
    struct Foreign { size_t bytes; std::shared_ptr<PoolInbox> home; };
    std::unordered_map<void*, Foreign> foreign;      // live blocks that belong to another context's pool
    std::shared_ptr<PoolInbox> inbox = std::make_shared<PoolInbox>();
    hipStream_t stream = nullptr;   // the owning context's stream (events of returned blocks)
    size_t bytes_total = 0;
    size_t n_malloc = 0;            // diagnostics (LTM_POOL_STATS=1 prints them when the context is destroyed)
    double malloc_s = 0.0;
    static size_t round_up(size_t b)
    {
        if (b < 512) return 512;
        int e = 63 - __builtin_clzll(b);
        size_t step = (size_t)1 << (e > 3 ? e - 3 : 0);   // 8 size classes per power of two
        return (b + step - 1) / step * step;
    }
    void drain_inbox()
    {
        std::vector<PoolInbox::Item> got;
        { std::lock_guard<std::mutex> lk(inbox->mx); got.swap(inbox->items); }
        for (const PoolInbox::Item& it : got) {
            if (it.ev) { (void)hipStreamWaitEvent(stream, it.ev, 0); (void)hipEventDestroy(it.ev); }
            free_blocks.emplace(it.bytes, it.p);
        }
    }
    void* alloc(size_t bytes)
    {
        if (bytes == 0) bytes = 1;
        const size_t want = round_up(bytes);
        drain_inbox();
        auto it = free_blocks.lower_bound(want);
        if (it != free_blocks.end() && it->first <= want + want / 4) {
            void* p = it->second;
            live[p] = it->first;
            free_blocks.erase(it);
            return p;
        }
        void* p = nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        hipError_t e = hipMalloc(&p, want);
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        malloc_s += dt;
        ++n_malloc;
        static const bool trace = getenv("LTM_POOL_TRACE") != nullptr;
        if (trace) fprintf(stderr, "[ltm] pool: hipMalloc #%zu of %.1f MB took %.2f ms (held %.1f MB)\n", n_malloc, want / 1048576.0, 1e3 * dt, bytes_total / 1048576.0);
        if (e != hipSuccess) {
            release_cached();
            e = hipMalloc(&p, want);
            if (e != hipSuccess) throw Err{LTM_E_NOMEM, "hipMalloc of " + std::to_string(want) + " bytes failed"};
        }
        bytes_total += want;
        live[p] = want;
        return p;
    }
    void free(void* p)
    {
        if (!p) return;
        auto it = live.find(p);
        if (it != live.end()) {
            free_blocks.emplace(it->second, p);
            live.erase(it);
            return;
        }
        auto f = foreign.find(p);
        if (f == foreign.end()) return;
        send_home(p, f->second, true);
        foreign.erase(f);
    }
    void send_home(void* p, const Foreign& f, bool with_event)
    {
        hipEvent_t ev = nullptr;
        if (with_event && (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess || hipEventRecord(ev, stream) != hipSuccess)) {
            if (ev) (void)hipEventDestroy(ev);
            ev = nullptr;
            (void)hipStreamSynchronize(stream);      // no event: the block goes home only when this stream is done with it
        }
        std::unique_lock<std::mutex> lk(f.home->mx);
        if (f.home->closed) { lk.unlock(); if (ev) { (void)hipEventSynchronize(ev); (void)hipEventDestroy(ev); } (void)hipFree(p); return; }
        f.home->items.push_back(PoolInbox::Item{p, f.bytes, ev});
    }
    bool owns(void* p) const { return live.count(p) != 0 || foreign.count(p) != 0; }
    // ltm_*_give: a live block changes contexts (both on one device); its home stays where it was allocated
    bool move_to(void* p, Pool& other)
    {
        auto it = live.find(p);
        if (it != live.end()) {
            other.foreign[p] = Foreign{it->second, inbox};
            live.erase(it);
            return true;
        }
        auto f = foreign.find(p);
        if (f == foreign.end()) return false;
        if (f->second.home == other.inbox) other.live[p] = f->second.bytes;      // it comes home while still in use
        else other.foreign[p] = f->second;
        foreign.erase(f);
        return true;
    }
    void release_cached()
    {
        drain_inbox();
        for (auto& kv : free_blocks) { (void)hipFree(kv.second); bytes_total -= kv.first; }
        free_blocks.clear();
    }
    void release_all()      // the owning context's streams have been drained
    {
        for (auto& kv : foreign) send_home(kv.first, kv.second, false);
        foreign.clear();
        { std::lock_guard<std::mutex> lk(inbox->mx); inbox->closed = true; }
        release_cached();
        for (auto& kv : live) (void)hipFree(kv.first);
        live.clear();
    }
};

struct Cloud {
    float4* d = nullptr; size_t n = 0;
    bool borrowed = false;          // a view of another context's cloud (ltm_cloud_lend): freeing the handle releases nothing
    // the octree frame (and leaf) of the voxel grid this cloud came out of, kept by order-preserving subsets of it (partition outputs, clones):
    // lets the next grid of the cloud test "nothing to do" during its bounding-box pass (k_bbox_reduce_check)
    bool vf_ok = false; OctreeFrame vf{}; float vleaf = 0.0f;
};
struct ScanSet { float4* d = nullptr; size_t n_pts = 0; std::vector<uint64_t> off; uint64_t* off_dev = nullptr; bool borrowed = false; size_t nkf() const { return off.size() - 1; } };
struct Poses { size_t n = 0; std::vector<double> pose, inv; double* pose_dev = nullptr; double* inv_dev = nullptr; float* approx_dev = nullptr; };

// bytes = SURVEY 8(d)'s algorithmic bytes (one map read per keyframe); bytes_c = the compulsory bytes of the launch as this design issues it (a projection
// launch reads its map ONCE for all the keyframes of the batch): equal to `bytes` for every class but the projection kernels
struct ProfClass { double ms = 0; uint64_t launches = 0; double units = 0, bytes = 0, bytes_c = 0; };
// scan2RangeImg depends only on (scan set, image shape, keyframe range): the remove / revert / remove passes of one
// resolution and the three ND / PD filter passes project the same scans again, so finished scan images are kept.
struct ScanImgEntry { uint64_t ss; int rows, cols; size_t kb, nb; uint32_t* buf; uint32_t* smax; size_t bytes; uint64_t stamp; float* qbound; float q_thr; };
struct Pending { int cls; hipEvent_t a, b; };
// a vote launch whose algorithmic bytes depend on the number of (tile, keyframe) workgroups that survive the whole-tile cull:
// counted on the device into slot `slot` of ctx->live_counts, folded into the class totals when the profile is collected
struct PendingLive { int cls; int slot; double max_pts; double image_bytes; double map_pts; double image_bytes_c; };
struct PinnedBlock { void* p; size_t bytes; bool in_use; };
// pipelined scan-set upload: device array of `cap` points filled front to back, two pinned staging buffers in flight
// (fill: bytes gathered in stage[next] and not yet on their way; flushed: points whose DMA has been issued -- round 5: chunks are gathered into 16 MB
// staging buffers before a DMA is issued; one hipMemcpyAsync + event per 54 k-point scan cost ~0.1 ms of fixed overhead each, 55 ms per 500-keyframe session)
struct UploadState { float4* d = nullptr; size_t cap = 0, n = 0; std::vector<uint64_t> off{0}; void* stage[2] = {nullptr, nullptr}; size_t stage_sz[2] = {0, 0}; hipEvent_t ev[2] = {nullptr, nullptr}; bool busy[2] = {false, false}; int next = 0; size_t fill = 0, flushed = 0; };

} // namespace ltm_detail
using namespace ltm_detail;

// ticket of an asynchronous device->host fetch (include/ltm.h)
// Chunked tickets (ltm_*_fetch_chunks_begin): the points travel through a small fixed ring of pinned chunks (FetchRing) filled by
// the context's copier thread and consumed, chunk by chunk, by writer threads -- nothing the size of an output is ever page-locked.
struct FetchChunk { void* host; size_t first_point, n_points, first_kf, n_kf; };
struct FetchRing;
struct ltm_fetch {
    void* host = nullptr; size_t bytes = 0, n_points = 0; std::vector<uint64_t> off; hipEvent_t done = nullptr; int device = 0;
    // chunked form
    bool chunked = false;
    const float4* src = nullptr;            // device source, kept alive by the caller until the ticket is released
    FetchRing* ring = nullptr;
    std::vector<FetchChunk> plan;           // what the copier thread will produce, in order
    std::mutex mx;
    std::condition_variable cv;
    std::deque<FetchChunk> avail;           // produced, not yet handed to a consumer
    bool produced_all = false;
    int error = LTM_OK;
};
struct FetchRing {
    int device = 0;
    size_t slot_bytes = 0;
    std::vector<void*> slots, free_slots;
    std::mutex mx;
    std::condition_variable cv_free, cv_jobs;
    std::deque<ltm_fetch*> jobs;
    bool stop = false;
    std::thread worker;
};

// Lanes (include/ltm.h): the projection kernels are bound by vector-instruction issue, everything else by launch latency.  Two projection launches of two
// lanes running TOGETHER share the vector pipes and end together, after which both lanes' latency-bound stages run together on an idle machine
// (measured, tools/ubench/stream_priority.hip and profiles/r6_lanes_*: "lockstep").  So the heavy launches of a lane family are CHAINED -- each waits,
// on the device, for the previous one of any lane -- and go to a stream of the lowest priority: one heavy kernel at a time fills the machine while the
// other lane's grids / scans / partitions are dispatched ahead of its remaining workgroups.
extern "C" void destroy_ring(FetchRing* r);      // ltm_api_core.cpp (defined among the fetch entry points, inside their extern "C" block)
struct HeavyChain {
    std::mutex mx;
    hipEvent_t last = nullptr;      // end of the most recent heavy launch of the family
    // the family also shares ONE ring of pinned chunks + copier thread for chunked fetches (page-locking 64 MB costs ~30 ms: a lane of a one-shot
    // run would pay it a second time for the few outputs it writes itself)
    std::mutex ring_mx;
    FetchRing* ring = nullptr;
    ~HeavyChain() { if (last) (void)hipEventDestroy(last); destroy_ring(ring); }
};

struct ltm_ctx {
    std::recursive_mutex mx;        // every entry point holds it (guarded): handles may be freed from any thread, lanes exchange clouds under both locks
    ltm_config cfg;
    int device = 0;
    hipStream_t stream = nullptr;
    bool in_lane_family = false;                 // this context has lanes / is one: its heavy launches are chained with theirs
    std::shared_ptr<HeavyChain> heavy;           // shared by a context and its lanes
    // LTM_HEAVY_CHAIN=0: no chaining (A/B).  A lowest-priority stream of their own for the heavy launches was built, measured and removed in round 6: it
    // bought nothing (profiles/r6_lanes_*) and every extra stream is one more hardware queue: with five queues on the four compute pipes one lane's
    // projection stream shared a pipe with the OTHER lane's stream and held up its small launches for the length of a vote (seen with the C++ host,
    // whose loader had created a copy stream first: 170 instead of 157 ms per step)
    int heavy_chain_on = 1;
    size_t heavy_min_blocks = 100000;            // LTM_HEAVY_MIN_BLOCKS: launches below this many workgroups (revert passes on the small dynamic map, ND / PD filters) run unchained
    HostMat34 L2B, B2L;
    int l2b_identity = 1, b2l_identity = 1;
    size_t kf_batch = 512;
    KernelOpts kopts;               // kernel variants / diagnostics of this context (environment, read at ltm_create)
    int fast_math = 0;   // set by the create-time self-check of the fast arithmetic forms for this FOV
    unsigned long long selfcheck[3] = {0, 0, 0};
    Pool pool;
    double pinned_s = 0.0;          // diagnostics (LTM_POOL_STATS): time inside hipHostMalloc, bytes pinned
    size_t pinned_bytes = 0;
    std::mutex pinned_mx;           // the pinned blocks are handed back by writer threads (ltm_fetch_release)
    uint64_t next_handle = 1;
    std::unordered_map<uint64_t, Cloud> clouds;
    std::unordered_map<uint64_t, ScanSet> scansets;
    std::unordered_map<uint64_t, Poses> poses;
    std::string err;
    // profiling
    bool prof_on = false;
    std::vector<std::string> prof_names;
    std::vector<ProfClass> prof;
    std::vector<Pending> pending;
    std::vector<PendingLive> pending_live;
    unsigned long long* live_counts = nullptr;   // device, kLiveSlots entries
    std::vector<hipEvent_t> event_pool;
    double scan_multi_max_density = 2.5;        // mean scan points per pixel of the smallest shape up to which ltm_scanset_prepare_range_images uses the one-pass kernel (scan_images_prepare)
    std::vector<ScanImgEntry> scan_cache;
    uint64_t scan_cache_stamp = 0;
    size_t scan_cache_cap = (size_t)3 << 30;   // bytes
    int occlusion_cull = 1;                     // LTM_OCCLUSION=0: the exact-image kernel runs every (tile, keyframe) pair (A/B switch)
    size_t occlusion_min_pairs = (size_t)1 << 21;   // LTM_OCCLUSION_MIN_PAIRS: smaller launches are not worth the two extra passes (2 M pairs = 4096 tiles x 512 keyframes)
    float occlusion_r_near = 60.0f;             // LTM_OCCLUSION_RNEAR [m]: tiles nearer than this are projected first and serve as occluders
    int occlusion_subtile = 1;                  // 0: the cull looks at whole 4096-point tiles only (round 5's form; the A/B is profiles/r6_ab_occlusion_second_look_at_tile_quarters.txt)
    int occlusion_stats_on = 0;                 // LTM_OCCLUSION_STATS: also count the quarters of live pairs and those left alive (one more host round trip per batch)
    uint64_t occl_quarters = 0, occl_quarters_live = 0;
    uint64_t occl_pairs = 0, occl_near = 0, occl_far_live = 0;      // statistics (LTM_OCCLUSION_STATS): pairs seen, in the first shell, projected in all
    void* occl_scratch = nullptr; size_t occl_scratch_bytes = 0;
    int voxel_key_compress = 1;                 // LTM_VOXEL_KEYBITS=0: sort over all 3*depth Morton bits (A/B switch)
    int voxel_fused_tail = 1;                   // LTM_VOXEL_FUSED_TAIL=0: head flags / scan / segment starts as four kernels (A/B switch)
    int voxel_identity = 1;                     // LTM_VOXEL_IDENTITY=0: never take the "already gridded under this frame" shortcut (A/B switch)
    uint64_t voxel_identity_hits = 0, voxel_calls = 0;
    int knn_two_phase = 1;                      // LTM_KNN_FAST=0: the one-kernel exact search for every query (A/B switch)
    int knn_stats_on = 0;                       // LTM_KNN_STATS=1: count the queries phase 1 leaves undecided (one host round trip per call)
    uint64_t knn_undecided = 0, knn_queries = 0;
    // The culled kernels rest on error bounds of the bounded-error projection that were validated empirically (tools/eps_sweep.py, ltm_debug_cull_check in the
    // tests) for the fields of view and extrinsics that were fuzzed.  Every image shape is therefore checked ON THE DEVICE the first time a context uses
    // it (cull_geometry_ok: 2^20 probe points on and beside the pixel boundaries, local frame and through one real keyframe pose); a shape that fails falls
    // back to the exact kernels for good.  LTM_CULL_SELFCHECK=0 skips it (A/B).
    std::map<std::pair<int, int>, bool> cull_geom_ok;
    uint64_t cull_geoms_checked = 0, cull_geoms_failed = 0;
    int cull_selfcheck = 1;
    float cull_eps_scale = 0.0f;                // LTM_CULL_EPS_SCALE: pixels of distrust per (pixel per degree); 0 = the validated default of geom_for
    float cull_eps_floor = 1.0e-3f;             // LTM_CULL_EPS_FLOOR: the band is never narrower than this [pixels]
    int el_fit = 0;                             // fitted elevation polynomial usable (vfov/2 + 2 deg <= 45 deg and error <= 1e-6 rad)
    float el_c[4] = {1.0f, 0.0f, 0.0f, 0.0f};
    double el_fit_err = 0.0;
    // copy engine side (pipelined loader / asynchronous output fetch): its own stream, pinned staging memory
    void* scratch_pinned = nullptr;             // staging of the small host round trips (d2h / h2d helpers)
    hipStream_t copy_stream = nullptr;
    std::vector<PinnedBlock> pinned;
    std::unordered_map<uint64_t, UploadState> uploads;
    std::vector<struct ltm_vgs*> vgs_open;      // ltm_voxel_grid_scanset_begin tickets not ended yet: joined and released by ltm_destroy at the latest
};

namespace ltm_detail {

struct DevBuf {   // RAII pooled scratch
    ltm_ctx* c; void* p;
    DevBuf(ltm_ctx* c_, size_t bytes) : c(c_), p(c_->pool.alloc(bytes)) {}
    ~DevBuf() { c->pool.free(p); }
    DevBuf(const DevBuf&) = delete; DevBuf& operator=(const DevBuf&) = delete;
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

inline void use_device(ltm_ctx* c) { LTM_HIP(hipSetDevice(c->device)); }

inline int prof_class(ltm_ctx* c, const char* name)
{
    for (size_t i = 0; i < c->prof_names.size(); ++i) if (c->prof_names[i] == name) return (int)i;
    c->prof_names.push_back(name); c->prof.push_back(ProfClass());
    return (int)c->prof_names.size() - 1;
}
inline hipEvent_t get_event(ltm_ctx* c)
{
    if (!c->event_pool.empty()) { hipEvent_t e = c->event_pool.back(); c->event_pool.pop_back(); return e; }
    hipEvent_t e; LTM_HIP(hipEventCreate(&e)); return e;
}
struct ProfScope {   // HIP-event bracket around one kernel class on the context's stream
    ltm_ctx* c; int cls = -1; hipEvent_t a = nullptr;
    ProfScope(ltm_ctx* c_, const char* name, double units, double bytes, double bytes_c = -1.0, bool count_launch = true) : c(c_)
    {
        if (!c->prof_on) return;
        cls = prof_class(c, name);
        if (count_launch) c->prof[cls].launches++; c->prof[cls].units += units; c->prof[cls].bytes += bytes; c->prof[cls].bytes_c += bytes_c < 0.0 ? bytes : bytes_c;
        a = get_event(c);
        LTM_HIP(hipEventRecord(a, c->stream));
    }
    ~ProfScope()
    {
        if (cls < 0) return;
        hipEvent_t b = nullptr;
        if (hipEventCreate(&b) != hipSuccess) return;
        (void)hipEventRecord(b, c->stream);
        c->pending.push_back(Pending{cls, a, b});
    }
};
// One heavy (vector-issue bound) launch of a lane family: see HeavyChain.  Without lanes this is a no-op.  (A separate lowest-priority stream for these
// launches was measured in round 6 and lost: a fifth hardware queue shares a compute pipe with the other lane's stream, NOTEBOOK §6.)
struct HeavyScope {
    ltm_ctx* c; bool chained = false;
    std::unique_lock<std::mutex> turn;      // held from the wait for the predecessor until this launch has become the family's latest: two lanes that get here
                                            // together must not both queue behind the SAME predecessor (they would run side by side -- seen with the C++ host's two threads)
    HeavyScope(ltm_ctx* c_, size_t n_blocks) : c(c_)
    {
        if (!c->in_lane_family || n_blocks < c->heavy_min_blocks) return;
        chained = c->heavy_chain_on != 0;
        if (chained) {
            turn = std::unique_lock<std::mutex>(c->heavy->mx);
            if (c->heavy->last) LTM_HIP(hipStreamWaitEvent(stream(), c->heavy->last, 0));
        }
    }
    hipStream_t stream() const { return c->stream; }
    void done()      // after the launch: the family's next heavy launch continues behind it
    {
        if (!chained) return;
        hipEvent_t e = nullptr;
        LTM_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        const hipError_t rc = hipEventRecord(e, stream());
        if (rc != hipSuccess) { (void)hipEventDestroy(e); LTM_HIP(rc); return; }
        if (c->heavy->last) (void)hipEventDestroy(c->heavy->last);      // waits already enqueued on it keep it alive inside the runtime
        c->heavy->last = e;
        turn.unlock();
    }
};

static constexpr int kLiveSlots = 4096;
inline void prof_collect(ltm_ctx* c)
{
    if (c->pending.empty() && c->pending_live.empty()) return;
    LTM_HIP(hipStreamSynchronize(c->stream));
    if (!c->pending_live.empty()) {
        std::vector<unsigned long long> live(kLiveSlots);
        LTM_HIP(hipMemcpy(live.data(), c->live_counts, sizeof(unsigned long long) * kLiveSlots, hipMemcpyDeviceToHost));
        for (const PendingLive& p : c->pending_live) {
            const double pts = std::min(p.max_pts, (double)live[(size_t)p.slot] * 4096.0);
            c->prof[p.cls].units += pts;
            c->prof[p.cls].bytes += 16.0 * pts + p.image_bytes;
            c->prof[p.cls].bytes_c += 16.0 * std::min(pts, p.map_pts) + p.image_bytes_c;      // the map tiles at most once per launch
        }
        c->pending_live.clear();
    }
    for (Pending& p : c->pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) c->prof[p.cls].ms += ms;
        c->event_pool.push_back(p.a); c->event_pool.push_back(p.b);
    }
    c->pending.clear();
}

inline void sync(ltm_ctx* c) { LTM_HIP(hipStreamSynchronize(c->stream)); }
// Small host<->device transfers (counts, bounding boxes, offset tables: the host round trips between stages) go through a pinned
// scratch buffer of the context: a copy to / from pageable memory makes the runtime stage or pin pages on every call.
static constexpr size_t kSmallCopy = 64 << 10;
inline void* small_scratch(ltm_ctx* c)
{
    if (!c->scratch_pinned && hipHostMalloc(&c->scratch_pinned, kSmallCopy, hipHostMallocDefault) != hipSuccess) c->scratch_pinned = nullptr;
    return c->scratch_pinned;
}
inline void d2h(ltm_ctx* c, void* dst, const void* src, size_t bytes)
{
    if (!bytes) return;
    void* sp = bytes <= kSmallCopy ? small_scratch(c) : nullptr;
    LTM_HIP(hipMemcpyAsync(sp ? sp : dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    sync(c);
    if (sp) memcpy(dst, sp, bytes);
}
inline void h2d(ltm_ctx* c, void* dst, const void* src, size_t bytes)
{
    if (!bytes) return;
    void* sp = bytes <= kSmallCopy ? small_scratch(c) : nullptr;
    if (sp) memcpy(sp, src, bytes);
    LTM_HIP(hipMemcpyAsync(dst, sp ? sp : src, bytes, hipMemcpyHostToDevice, c->stream));
    sync(c);   // the host buffer may be pageable and is not ours to keep (and the scratch is reused by the next small copy)
}
inline void d2d(ltm_ctx* c, void* dst, const void* src, size_t bytes)
{
    if (!bytes) return;
    LTM_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, c->stream));
}

inline Cloud& get_cloud(ltm_ctx* c, ltm_cloud h)
{
    auto it = c->clouds.find(h);
    if (it == c->clouds.end()) throw Err{LTM_E_INVALID, "invalid cloud handle " + std::to_string(h)};
    return it->second;
}
inline ScanSet& get_ss(ltm_ctx* c, ltm_scanset h)
{
    auto it = c->scansets.find(h);
    if (it == c->scansets.end()) throw Err{LTM_E_INVALID, "invalid scanset handle " + std::to_string(h)};
    return it->second;
}
inline Poses& get_poses(ltm_ctx* c, ltm_poses h)
{
    auto it = c->poses.find(h);
    if (it == c->poses.end()) throw Err{LTM_E_INVALID, "invalid poses handle " + std::to_string(h)};
    return it->second;
}
inline ltm_cloud new_cloud(ltm_ctx* c, float4* d, size_t n)
{
    const uint64_t h = c->next_handle++;
    Cloud cl; cl.d = d; cl.n = n;
    c->clouds[h] = cl;
    return h;
}
inline void inherit_frame(ltm_ctx* c, ltm_cloud child, const Cloud& parent)
{
    if (!parent.vf_ok) return;
    Cloud& ch = c->clouds[child];
    ch.vf_ok = true; ch.vf = parent.vf; ch.vleaf = parent.vleaf;
}
inline void set_frame(ltm_ctx* c, ltm_cloud h, const OctreeFrame& f, float leaf, bool ok)
{
    Cloud& cl = c->clouds[h];
    cl.vf_ok = ok; cl.vf = f; cl.vleaf = leaf;
}
inline ltm_cloud alloc_cloud(ltm_ctx* c, size_t n, float4** d)
{
    *d = reinterpret_cast<float4*>(c->pool.alloc(std::max<size_t>(n, 1) * sizeof(float4)));
    return new_cloud(c, *d, n);
}
inline ltm_scanset new_scanset(ltm_ctx* c, float4* d, std::vector<uint64_t> off)
{
    ScanSet s;
    s.d = d; s.n_pts = off.back(); s.off = std::move(off);
    s.off_dev = reinterpret_cast<uint64_t*>(c->pool.alloc(s.off.size() * sizeof(uint64_t)));
    h2d(c, s.off_dev, s.off.data(), s.off.size() * sizeof(uint64_t));
    const uint64_t h = c->next_handle++;
    c->scansets[h] = std::move(s);
    return h;
}

inline bool mat_is_identity(const double* m16)
{
    for (int r = 0; r < 4; ++r) for (int k = 0; k < 4; ++k) if (m16[4 * r + k] != (r == k ? 1.0 : 0.0)) return false;
    return true;
}
inline HostMat34 to34(const double* m16) { HostMat34 h; memcpy(h.m, m16, 12 * sizeof(double)); return h; }



inline hipStream_t copy_stream(ltm_ctx* c)
{
    if (!c->copy_stream) LTM_HIP(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    return c->copy_stream;
}
// Pinned staging for uploads and fetches.  Page-locking is the expensive part (hipHostMalloc pins at ~4.5 GB/s: 0.55 s for the
// 2.5 GB of outputs of a 2x500-keyframe run, on the context thread), so blocks are recycled as soon as a writer thread is done
// with them (ltm_fetch_release may be called there), and once 768 MB are pinned any free block that is large enough is taken
// rather than pinning another one of the ideal size.  Tried instead, on the 2x500-keyframe files -> files run (0.75-0.9 s): pinning
// 3.5x the input ahead of time on a helper thread (the runtime serialises the loader's calls behind the pinning: Step 0 0.23 ->
// 0.6 s), and no pinning at all -- the writer threads copy into ordinary memory themselves -- which frees the context thread of
// the 0.5 s but slows it by as much through the concurrent blocking copies (0.95 s).  What would remove the cost is a small
// fixed ring of pinned chunks that the writers consume chunk-wise; that changes the writer interface and is left for the next round.
inline void* pinned_alloc(ltm_ctx* c, size_t bytes)
{
    if (bytes == 0) bytes = 16;
    {
        std::lock_guard<std::mutex> lk(c->pinned_mx);
        int best = -1;
        for (size_t i = 0; i < c->pinned.size(); ++i)
            if (!c->pinned[i].in_use && c->pinned[i].bytes >= bytes && (best < 0 || c->pinned[i].bytes < c->pinned[(size_t)best].bytes)) best = (int)i;
        if (best >= 0 && (c->pinned[(size_t)best].bytes <= 2 * bytes + (1u << 20) || c->pinned_bytes >= ((size_t)768 << 20))) {
            c->pinned[(size_t)best].in_use = true;
            return c->pinned[(size_t)best].p;
        }
    }
    void* p = nullptr;
    const size_t want = (bytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
    const auto t0 = std::chrono::steady_clock::now();
    if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) throw Err{LTM_E_NOMEM, "hipHostMalloc of " + std::to_string(want) + " bytes failed"};
    std::lock_guard<std::mutex> lk(c->pinned_mx);
    c->pinned_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    c->pinned_bytes += want;
    c->pinned.push_back(PinnedBlock{p, want, true});
    return p;
}
inline void pinned_free(ltm_ctx* c, void* p)
{
    std::lock_guard<std::mutex> lk(c->pinned_mx);
    for (PinnedBlock& b : c->pinned) if (b.p == p) { b.in_use = false; return; }
}
// compute stream -> copy stream ordering: everything submitted so far on the context's stream happens before later copy-stream work
inline void copy_after_compute(ltm_ctx* c)
{
    hipEvent_t e = get_event(c);
    LTM_HIP(hipEventRecord(e, c->stream));
    LTM_HIP(hipStreamWaitEvent(copy_stream(c), e, 0));
    c->event_pool.push_back(e);      // safe to reuse: the wait has captured the recorded state
}

template <class F>
inline int guarded(ltm_ctx* c, F&& f)
{
    if (!c) return LTM_E_INVALID;
    std::lock_guard<std::recursive_mutex> lk(c->mx);
    try {
        use_device(c);
        f();
        return LTM_OK;
    } catch (const Err& e) {
        c->err = e.msg;
        return e.code;
    } catch (const std::bad_alloc&) {
        c->err = "host allocation failed";
        return LTM_E_NOMEM;
    } catch (const std::exception& e) {
        c->err = e.what();
        return LTM_E_INVALID;
    } catch (...) {
        c->err = "unknown error";
        return LTM_E_INVALID;
    }
}

// two contexts, one call: both locks, taken in address order whichever thread calls
struct TwoLocks {
    std::unique_lock<std::recursive_mutex> a, b;
    TwoLocks(ltm_ctx* x, ltm_ctx* y)
    {
        if (x == y) { a = std::unique_lock<std::recursive_mutex>(x->mx); return; }
        ltm_ctx* lo = x < y ? x : y; ltm_ctx* hi = x < y ? y : x;
        a = std::unique_lock<std::recursive_mutex>(lo->mx); b = std::unique_lock<std::recursive_mutex>(hi->mx);
    }
};
// everything submitted to `to` from now on runs after everything submitted to `from` so far
inline void stream_after(ltm_ctx* from, ltm_ctx* to)
{
    if (from == to) return;
    hipEvent_t e = nullptr;
    LTM_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hipError_t rc = hipEventRecord(e, from->stream);
    if (rc == hipSuccess) rc = hipStreamWaitEvent(to->stream, e, 0);
    (void)hipEventDestroy(e);      // the wait has captured the recorded state; the runtime keeps the event alive until then
    LTM_HIP(rc);
}
template <class F>
inline int guarded2(ltm_ctx* from, ltm_ctx* to, F&& f)
{
    if (!from || !to) return LTM_E_INVALID;
    TwoLocks lk(from, to);
    const int rc = guarded(from, [&] {
        LTM_REQUIRE(from->device == to->device, "lanes must share one device");
        f();
    });
    if (rc != LTM_OK) to->err = from->err;
    return rc;
}

// ---- helpers defined in one unit and used by others
bool inverse4x4(const double* m, double* inv);                                   // ltm_api_core.cpp (Eigen::Matrix4d::inverse restated)
void approx_pose(const double* b2l16, const double* inv16, float* out);           // ltm_api_core.cpp
int elevation_fit_for(float vfov, float c4[4], double* err);                     // ltm_api_core.cpp
void pack_from_host(const void* src, size_t n, size_t stride, std::vector<float>& out);
void unpack_to_host(const float* packed, size_t n, size_t stride, void* dst);
Geom geom_for(const ltm_ctx* c, float alpha);                                     // ltm_api_vote.cpp (utility.cpp:222-236 resetRimgSize)
size_t scan_total_u8(ltm_ctx* c, const uint8_t* labels, const uint32_t* pos, size_t n);
void scan_cache_drop(ltm_ctx* c, uint64_t ss_handle);                             // ltm_api_vote.cpp
void do_partition(ltm_ctx* c, const Cloud& map, const uint8_t* labels, ltm_cloud* kept, ltm_cloud* flagged);   // ltm_api_vote.cpp
void bbox_of(ltm_ctx* c, const float4* pts, size_t n, float mn[3], float mx[3]);  // ltm_api_voxel.cpp
void vgs_release_all(ltm_ctx* c);                                                 // ltm_api_voxel.cpp: open ltm_voxel_grid_scanset tickets, at ltm_destroy
void split_by_flag(ltm_ctx* c, const float4* pts, const uint8_t* flag, size_t n, const std::vector<uint64_t>& bounds, const uint64_t* offsets_dev,
                   size_t kf0, uint64_t first, float4** d_set, std::vector<uint64_t>* off_set, float4** d_unset, std::vector<uint64_t>* off_unset);   // ltm_api_knn.cpp

} // namespace ltm_detail
using namespace ltm_detail;
