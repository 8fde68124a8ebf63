// ltm_k_knn.hip -- inter-session kNN change detection (Session.cpp:393-504, 537-642): hash grid, two-phase exact fixed-radius search
// (gfx950 / CDNA4, wave64; part of libltm_hip.so -- shared definitions in ltm_kernels_common.h, launch wrappers declared in ltm_kernels.h)
#include "ltm_kernels_common.h"
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/transform_iterator.hpp>
namespace ltm {

// ---------------------------------------------------------------------------------------- kNN
// Uniform grid with cell edge >= sqrt(k*thr)*(1+1e-3): every neighbour that can make a query "coexist"
// lies in the 27 cells around the query's cell (see DESIGN.md for the exactness argument).
__device__ __forceinline__ bool cell_of(const KnnGrid& g, float x, float y, float z, long long& cx, long long& cy, long long& cz)
{
    cx = (long long)floor(((double)x - g.ox) * g.inv_cell);
    cy = (long long)floor(((double)y - g.oy) * g.inv_cell);
    cz = (long long)floor(((double)z - g.oz) * g.inv_cell);
    return cx >= 0 && cy >= 0 && cz >= 0 && cx < g.nx && cy < g.ny && cz < g.nz;
}
__device__ __forceinline__ uint64_t cell_id(const KnnGrid& g, long long cx, long long cy, long long cz)
{
    return (uint64_t)((cx * g.ny + cy) * g.nz + cz);
}
__global__ void __launch_bounds__(kBlock)
k_cell_keys(const float4* __restrict__ pts, size_t n, KnnGrid g, uint64_t* __restrict__ keys, uint32_t* __restrict__ idx)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 p = pts[i];
    long long cx, cy, cz;
    cell_of(g, p.x, p.y, p.z, cx, cy, cz);
    cx = min(max(cx, 0ll), g.nx - 1); cy = min(max(cy, 0ll), g.ny - 1); cz = min(max(cz, 0ll), g.nz - 1);
    keys[i] = cell_id(g, cx, cy, cz);
    idx[i] = (uint32_t)i;
}
hipError_t cell_keys(const float4* pts, size_t n, KnnGrid g, uint64_t* keys, uint32_t* idx, hipStream_t s)
{
    if (!n) return hipSuccess;
    k_cell_keys<<<dim3(grid_for(n)), dim3(kBlock), 0, s>>>(pts, n, g, keys, idx);
    return hipGetLastError();
}
__global__ void __launch_bounds__(kBlock)
k_gather_points(const float4* __restrict__ in, const uint32_t* __restrict__ idx, size_t n, float4* __restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[idx[i]];
}
hipError_t gather_points(const float4* in, const uint32_t* idx, size_t n, float4* out, hipStream_t s)
{
    if (!n) return hipSuccess;
    k_gather_points<<<dim3(grid_for(n)), dim3(kBlock), 0, s>>>(in, idx, n, out);
    return hipGetLastError();
}
__global__ void __launch_bounds__(kBlock)
k_gather_u64(const uint64_t* __restrict__ in, const uint32_t* __restrict__ idx, size_t n, uint64_t* __restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[idx[i]];
}
hipError_t gather_u64(const uint64_t* in, const uint32_t* idx, size_t n, uint64_t* out, hipStream_t s)
{
    if (!n) return hipSuccess;
    k_gather_u64<<<dim3(grid_for(n)), dim3(kBlock), 0, s>>>(in, idx, n, out);
    return hipGetLastError();
}

// 32-bit mixer (two 32-bit multiplies; a 64-bit finaliser costs ~8 on this ISA and runs up to 27 times per query)
__device__ __forceinline__ uint32_t hash64(uint64_t k)
{
    uint32_t x = (uint32_t)k ^ ((uint32_t)(k >> 32) * 0x9e3779b1u);
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
static constexpr uint64_t kEmptyKey = ~0ull;

__global__ void __launch_bounds__(kBlock)
k_hash_build(const uint64_t* __restrict__ sorted_keys, const uint32_t* __restrict__ starts, size_t n_cells, size_t n_pts,
             HashEntry* __restrict__ table, uint32_t mask)
{
    const size_t u = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= n_cells) return;
    const uint32_t a = starts[u];
    const uint32_t b = (u + 1 < n_cells) ? starts[u + 1] : (uint32_t)n_pts;
    const uint64_t key = sorted_keys[a];
    uint32_t h = hash64(key) & mask;
    while (true) {
        const unsigned long long prev = atomicCAS(reinterpret_cast<unsigned long long*>(&table[h].key), (unsigned long long)kEmptyKey,
                                                  (unsigned long long)key);
        if (prev == kEmptyKey) { table[h].start = a; table[h].end = b; return; }
        h = (h + 1) & mask;
    }
}
hipError_t hash_build(const uint64_t* sorted_keys, const uint32_t* starts, size_t n_cells, size_t n_pts, HashEntry* table,
                      uint32_t table_mask, hipStream_t s)
{
    if (!n_cells) return hipSuccess;
    k_hash_build<<<dim3(grid_for(n_cells)), dim3(kBlock), 0, s>>>(sorted_keys, starts, n_cells, n_pts, table, table_mask);
    return hipGetLastError();
}

// sparse occupancy bitmap of the kNN grid (see k_knn_bitmap_build)
__device__ __forceinline__ uint32_t occ_word_of(uint32_t bx, uint32_t by, uint32_t bz, uint32_t mask)
{
    uint32_t h = bx * 0x9e3779b1u ^ (by * 0x85ebca6bu + 0x165667b1u) ^ (bz * 0xc2b2ae35u);
    h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12;
    return h & mask;
}
__device__ __forceinline__ bool occ_test(const unsigned long long* __restrict__ occ, uint32_t mask, int x, int y, int z)
{
    const unsigned long long w = occ[occ_word_of((uint32_t)x >> 2, (uint32_t)y >> 2, (uint32_t)z >> 2, mask)];
    return ((w >> ((((uint32_t)x & 3u) << 4) | (((uint32_t)y & 3u) << 2) | ((uint32_t)z & 3u))) & 1ull) != 0ull;
}
static constexpr int kMaxK = 16;

// returns the coexist/near predicate of Session.cpp:590-599 for a global-frame query point.
// KT > 0: k is the compile-time constant KT (and the target holds at least KT points): the k best distances live in
// registers and are maintained by a branch-free insertion network.  KT == 0: any k <= kMaxK, indexed array.
template <int KT>
__device__ __forceinline__ bool knn_near(float qx, float qy, float qz, const float4* __restrict__ tgt, size_t Mt, const KnnGrid& g,
                                         const HashEntry* __restrict__ table, uint32_t mask, int k_param, float thr, float cell2_lo,
                                         const unsigned long long* __restrict__ bitmap = nullptr, uint32_t bitmap_mask = 0)
{
    if (Mt == 0) return false;                        // reference: undefined; defined here as "far"
    const int k = KT ? KT : (int)min((size_t)k_param, Mt);      // pcl::KdTreeFLANN::nearestKSearch clamps k
    float best[KT ? KT : kMaxK];
    int cnt = 0;
    if (KT) {
#pragma unroll
        for (int j = 0; j < (KT ? KT : 1); ++j) best[j] = __builtin_inff();
    }
    auto push = [&](float d) {
        if (KT) {
            // sorted insert: slot j takes its left neighbour if d goes before it, else the smaller of itself and d
#pragma unroll
            for (int j = (KT ? KT : 1) - 1; j >= 0; --j) {
                if (j > 0) best[j] = (d < best[j - 1]) ? best[j - 1] : fminf(best[j], d);
                else best[0] = fminf(best[0], d);
            }
            cnt = min(cnt + 1, k);
        } else {
            if (cnt == k && !(d < best[k - 1])) return;
            int j = (cnt < k) ? cnt++ : k - 1;
            while (j > 0 && best[j - 1] > d) { best[j] = best[j - 1]; --j; }
            best[j] = d;
        }
    };
    // float sum = accumulate(begin, end, 0.0): double accumulation in ascending order, narrowed to float
    auto mean_below_thr = [&]() {
        double acc = 0.0;
        if (KT) {
#pragma unroll
            for (int j = 0; j < (KT ? KT : 1); ++j) if (j < cnt) acc = acc + (double)best[j];
        } else {
            for (int j = 0; j < cnt; ++j) acc = acc + (double)best[j];
        }
        return fabsf((float)acc / (float)k_param) < thr;
    };
    if (Mt <= 64 || (size_t)k_param > Mt) {
        for (size_t j = 0; j < Mt; ++j) { const float4 t = tgt[j]; push(sqdist_l2simple(qx, qy, qz, t.x, t.y, t.z)); }
    } else {
        // 32-bit cell coordinates (every axis has < 2^20 cells); a query farther than one cell outside the grid gets the
        // sentinel -2 so that all of its 27 cells fail the range test below
        const double fx = floor(((double)qx - g.ox) * g.inv_cell), fy = floor(((double)qy - g.oy) * g.inv_cell),
                     fz = floor(((double)qz - g.oz) * g.inv_cell);
        const int nx = (int)g.nx, ny = (int)g.ny, nz = (int)g.nz;
        const int cx = (fx >= -1.0 && fx <= (double)nx) ? (int)fx : -2;
        const int cy = (fy >= -1.0 && fy <= (double)ny) ? (int)fy : -2;
        const int cz = (fz >= -1.0 && fz <= (double)nz) ? (int)fz : -2;
        // 27 cells, centre first, then faces, edges, corners.  Early exit: as soon as the current k best already
        // satisfy the predicate the answer is final -- further neighbours can only lower the (monotonically
        // rounded) sum, so "coexist" cannot flip back.  Most queries of a static scene stop after the first cell.
        constexpr int8_t kOrder[27][3] = {
            {0, 0, 0},
            {-1, 0, 0}, {1, 0, 0}, {0, -1, 0}, {0, 1, 0}, {0, 0, -1}, {0, 0, 1},
            {-1, -1, 0}, {-1, 1, 0}, {1, -1, 0}, {1, 1, 0}, {-1, 0, -1}, {-1, 0, 1}, {1, 0, -1}, {1, 0, 1}, {0, -1, -1}, {0, -1, 1}, {0, 1, -1}, {0, 1, 1},
            {-1, -1, -1}, {-1, -1, 1}, {-1, 1, -1}, {-1, 1, 1}, {1, -1, -1}, {1, -1, 1}, {1, 1, -1}, {1, 1, 1}};
        // One cell: its table entry `e` was fetched from the first probe position `h`; continue the linear probe if that slot holds
        // another cell.  Returns true when the answer is final (early exit above).
        auto visit = [&](uint64_t key, uint32_t h, HashEntry e) -> bool {
            while (e.key != key) {
                if (e.key == kEmptyKey) return false;
                h = (h + 1) & mask;
                e = table[h];
            }
            for (uint32_t j = e.start; j < e.end; ++j) { const float4 t = tgt[j]; push(sqdist_l2simple(qx, qy, qz, t.x, t.y, t.z)); }
            return e.start != e.end && cnt == k && best[k - 1] < cell2_lo && mean_below_thr();
        };
        // in the grid and -- when the occupancy bitmap exists -- holding at least one target point (an empty cell has no table entry: the
        // probe would run into an empty slot and contribute nothing)
        auto cell_key = [&](int c, uint64_t& key) -> bool {
            const int x = cx + kOrder[c][0], y = cy + kOrder[c][1], z = cz + kOrder[c][2];
            key = ((uint64_t)(uint32_t)x * (uint32_t)ny + (uint32_t)y) * (uint32_t)nz + (uint32_t)z;   // == cell_id()
            const bool in = !((unsigned)x >= (unsigned)nx || (unsigned)y >= (unsigned)ny || (unsigned)z >= (unsigned)nz);
            if (!bitmap || !in) return in;
            return occ_test(bitmap, bitmap_mask, x, y, z);
        };
        {   // the centre cell alone: most queries of a static scene end here
            uint64_t key;
            if (cell_key(0, key)) { const uint32_t h = hash64(key) & mask; if (visit(key, h, table[h])) return true; }
        }
        // The 26 neighbours in four batches (faces, edges, edges, corners).  The table entries of a batch are requested together
        // before any of them is used: a query with no neighbours (a changed / dynamic point, a few per wavefront, which the whole
        // wavefront waits for) pays 4 dependent memory round trips here instead of 26.
        constexpr int kBatchEnd[4] = {7, 13, 19, 27};
        int c0 = 1;
#pragma unroll
        for (int bt = 0; bt < 4; ++bt) {
            constexpr int kMaxBatch = 8;
            uint64_t keys[kMaxBatch];
            uint32_t hs[kMaxBatch];
            bool in[kMaxBatch];
            HashEntry es[kMaxBatch];
#pragma unroll
            for (int i = 0; i < kMaxBatch; ++i) {
                if (c0 + i >= kBatchEnd[bt]) break;
                in[i] = cell_key(c0 + i, keys[i]);
                hs[i] = in[i] ? (hash64(keys[i]) & mask) : 0u;       // an out-of-grid cell reads slot 0 and ignores it
                es[i] = table[hs[i]];
            }
#pragma unroll
            for (int i = 0; i < kMaxBatch; ++i) {
                if (c0 + i >= kBatchEnd[bt]) break;
                if (in[i] && visit(keys[i], hs[i], es[i])) return true;
            }
            c0 = kBatchEnd[bt];
        }
        // fewer than k neighbours inside the provably-complete radius => the k-th neighbour is >= cell away => "diff"
        if (cnt < k || !(best[k - 1] < cell2_lo)) return false;
    }
    return mean_below_thr();
}

template <bool B2L_IDENTITY, int KT>
__global__ void __launch_bounds__(kBlock)
k_knn_query_scans(const float4* __restrict__ scans, const uint64_t* __restrict__ offsets, size_t kb, size_t ke, uint64_t first_pt,
                  uint64_t n_pts, const double* __restrict__ poses, const double* __restrict__ inv_poses, HostMat34 b2l_h,
                  const float4* __restrict__ tgt, size_t Mt, KnnGrid g, const HashEntry* __restrict__ table, uint32_t mask,
                  int k, float thr, float cell2_lo, uint8_t* __restrict__ coexist, float4* __restrict__ local_out)
{
    // grid = (chunks of the longest keyframe, keyframes): no per-point search for the keyframe
    const size_t kf = kb + blockIdx.y;
    const uint64_t a = offsets[kf], local = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (local >= offsets[kf + 1] - a) return;
    const uint64_t gi = a + local, i = gi - first_pt;
    const float4 p4 = scans[gi];
    float3 p = make_float3(p4.x, p4.y, p4.z);
    // Session.cpp:545 / :618: local2global(scan, pose, kSE3MatExtrinsicPoseBasetoLiDAR)  (sic, quirk Q7)
    if (B2L_IDENTITY) p = xform_identity(p); else p = xform(to_dev(b2l_h), p);
    const float3 gp = xform(load_mat(poses + 12 * kf), p);
    // :603-604 global2local(.., inverse pose, base2lidar)
    float3 l = xform(load_mat(inv_poses + 12 * kf), gp);
    if (B2L_IDENTITY) l = xform_identity(l); else l = xform(to_dev(b2l_h), l);
    local_out[i] = make_float4(l.x, l.y, l.z, p4.w);
    coexist[i] = knn_near<KT>(gp.x, gp.y, gp.z, tgt, Mt, g, table, mask, k, thr, cell2_lo) ? 1 : 0;
}
hipError_t knn_query_scans(const float4* scans, const uint64_t* offsets_dev, size_t kb, size_t ke, uint64_t first_pt, uint64_t n_pts,
                           uint64_t max_kf_pts, const double* poses_dev, const double* inv_poses_dev, HostMat34 b2l, int b2l_identity,
                           const float4* sorted_target, size_t Mt, KnnGrid g, const HashEntry* table, uint32_t table_mask,
                           int k, float thr, float cell2_lo, uint8_t* coexist, float4* local_out, hipStream_t s)
{
    if (!n_pts || !max_kf_pts) return hipSuccess;
    if (k < 1 || k > kMaxK) return hipErrorInvalidValue;
    const int kt = (k <= 4 && Mt >= (size_t)k) ? k : 0;     // register-resident specialisations for the usual k (yaml 2, default 3)
    auto launch = [&](auto b2l_tag, auto kt_tag) {
        for (size_t k0 = kb; k0 < ke; k0 += 65535) {      // gridDim.y limit
            const size_t k1 = std::min(ke, k0 + 65535);
            k_knn_query_scans<decltype(b2l_tag)::value, decltype(kt_tag)::value><<<dim3(grid_for(max_kf_pts), (unsigned)(k1 - k0)), dim3(kBlock), 0, s>>>(
                scans, offsets_dev, k0, k1, first_pt, n_pts, poses_dev, inv_poses_dev, b2l, sorted_target, Mt, g, table, table_mask, k, thr, cell2_lo,
                coexist, local_out);
        }
    };
    auto by_kt = [&](auto b2l_tag) {
        switch (kt) {
        case 1: launch(b2l_tag, std::integral_constant<int, 1>{}); break;
        case 2: launch(b2l_tag, std::integral_constant<int, 2>{}); break;
        case 3: launch(b2l_tag, std::integral_constant<int, 3>{}); break;
        case 4: launch(b2l_tag, std::integral_constant<int, 4>{}); break;
        default: launch(b2l_tag, std::integral_constant<int, 0>{}); break;
        }
    };
    if (b2l_identity) by_kt(std::true_type{}); else by_kt(std::false_type{});
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// Two-phase kNN query (k <= 4).  Round 2's kernel moved 8.5x its algorithmic bytes: every query probed a 16-byte hash entry in one
// cache line and then gathered the ~9 points of its cell from two or three others, three dependent round trips, and the few
// queries of a wavefront that end "diff" (27 cells) held the other lanes back.  Now:
//   phase 1 (k_knn_fast, every query): ONE 64-byte bucket per occupied cell holds up to nine of the cell's points QUANTISED to 16 bit
//     per axis inside the cell (2-choice hashing: the bucket is at one of two places).  From it the query gets an UPPER bound of its
//     squared distance to each of those points; if the k smallest bounds already satisfy the predicate of Session.cpp:590-599 with
//     a margin, the query is "coexist" whatever else is in the target -- the k nearest neighbours can only be nearer, and the
//     sum is rounded monotonically.  Everything else is left UNDECIDED (flag 2).
//   phase 2 (k_knn_slow, the undecided queries, compacted): the exact search of knn_near, on dense wavefronts.
// The quantised points only ever say "certainly coexist"; every other answer comes from the exact arithmetic, so the flags are
// those of the exact kernel (LTM_KNN_FAST=0 runs it alone; tests compare both with the oracle).
// Error budget of the bound: quantisation cell / 2^17 per axis (1.1e-6 m at the yaml cell of 0.1416 m), the query's own offset in its
// cell is computed in double and rounded once to float (1e-8 m), the float evaluation of d~ adds < 1e-7 m, so
// |d - d~| <= sqrt(3) cell / 2^17 + 2e-7 =: slack (computed on the host, 2.1e-6 m at yaml values, with 5 % on top); FLANN's float
// evaluation of d^2 (Sterbenz-exact differences, three roundings) is within 3e-7 relative of the true value.
// ub = (d~ + slack)^2 (1 + 3e-6), test: sum ub < k thr (1 - 1e-5).
struct __attribute__((aligned(64))) KnnBucketRaw { uint32_t w[16]; };      // w[0..1] cell key, then 27 x u16 (x y z of 9 points), u8 count, u8 pad
static_assert(sizeof(KnnBucketRaw) == 64, "");
static constexpr int kBucketPts = 9;

__device__ __forceinline__ uint32_t hash64b(uint64_t k)      // second hash function of the 2-choice table
{
    uint32_t x = (uint32_t)(k >> 32) ^ ((uint32_t)k * 0x85ebca6bu) ^ 0x27d4eb2fu;
    x ^= x >> 15; x *= 0x2c1b3c6du; x ^= x >> 12; x *= 0x297a2d39u; x ^= x >> 15;
    return x;
}
__device__ __forceinline__ uint32_t bucket_of(uint32_t h, uint32_t n_buckets) { return (uint32_t)(((uint64_t)h * n_buckets) >> 32); }

__global__ void __launch_bounds__(kBlock)
k_knn_bucket_build(const float4* __restrict__ tgt, const uint64_t* __restrict__ sorted_keys, const uint32_t* __restrict__ starts, size_t n_cells,
                   size_t n_pts, KnnGrid g, KnnBucketRaw* __restrict__ buckets, uint32_t n_buckets)
{
    const size_t u = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= n_cells) return;
    const uint32_t a = starts[u];
    const uint32_t b = (u + 1 < n_cells) ? starts[u + 1] : (uint32_t)n_pts;
    const uint64_t key = sorted_keys[a];
    uint32_t slot = bucket_of(hash64(key), n_buckets);
    unsigned long long* kp = reinterpret_cast<unsigned long long*>(&buckets[slot].w[0]);
    if (atomicCAS(kp, (unsigned long long)kEmptyKey, (unsigned long long)key) != kEmptyKey) {
        slot = bucket_of(hash64b(key), n_buckets);
        kp = reinterpret_cast<unsigned long long*>(&buckets[slot].w[0]);
        if (atomicCAS(kp, (unsigned long long)kEmptyKey, (unsigned long long)key) != kEmptyKey) return;      // both taken: this cell's queries go to phase 2
    }
    const uint32_t n = b - a, cnt = min(n, (uint32_t)kBucketPts);
    uint16_t q[3 * kBucketPts];
#pragma unroll
    for (int s = 0; s < kBucketPts; ++s) {
        // more points than fit: evenly spread over the cell's run (the target is a voxel-grid output in octree order: spatially spread)
        const uint32_t j = a + ((uint32_t)s < cnt ? (cnt > 1 ? (uint32_t)(((uint64_t)s * (n - 1)) / (cnt - 1)) : 0u) : 0u);
        const float4 p = tgt[j];
        const double t[3] = {((double)p.x - g.ox) * g.inv_cell, ((double)p.y - g.oy) * g.inv_cell, ((double)p.z - g.oz) * g.inv_cell};
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const double fr = t[d] - floor(t[d]);
            q[3 * s + d] = (uint16_t)min(65535, max(0, (int)(fr * 65536.0)));
        }
    }
    uint32_t* w = buckets[slot].w;
#pragma unroll
    for (int i = 0; i < 13; ++i) w[2 + i] = (uint32_t)q[2 * i] | ((uint32_t)q[2 * i + 1] << 16);
    w[15] = (uint32_t)q[26] | (cnt << 16);
}
hipError_t knn_bucket_build(const float4* sorted_target, const uint64_t* sorted_keys, const uint32_t* starts, size_t n_cells, size_t n_pts, KnnGrid g,
                            void* buckets, uint32_t n_buckets, hipStream_t s)
{
    if (!n_cells) return hipSuccess;
    k_knn_bucket_build<<<dim3(grid_for(n_cells)), dim3(kBlock), 0, s>>>(sorted_target, sorted_keys, starts, n_cells, n_pts, g,
                                                                     reinterpret_cast<KnnBucketRaw*>(buckets), n_buckets);
    return hipGetLastError();
}

// Sparse occupancy bitmap of the grid: the cells are grouped in blocks of 4 x 4 x 4, a block's 64 occupancy bits live in ONE 64-bit word
// found by hashing the block coordinates (no keys, no probing: two blocks that share a word see the OR of their bits, i.e. at worst a
// few false "occupied" answers, never a false "empty").  The exact search tests it before it probes the hash table, so the empty
// cells among the 27 -- most of them around the sparse, far-from-the-trajectory points that make up the bulk of the "diff" answers --
// cost a bit test in a word that neighbouring cells and neighbouring lanes share instead of a 64-byte line of the table each.
// Sized at two words per occupied cell / 8 (>= 4 words per occupied block on surfaces), any scene extent.
__global__ void __launch_bounds__(kBlock)
k_knn_bitmap_build(const uint64_t* __restrict__ sorted_keys, const uint32_t* __restrict__ starts, size_t n_cells, KnnGrid g, unsigned long long* __restrict__ occ, uint32_t mask)
{
    const size_t u = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= n_cells) return;
    const uint64_t key = sorted_keys[starts[u]];
    const uint32_t z = (uint32_t)(key % (uint64_t)g.nz), y = (uint32_t)((key / (uint64_t)g.nz) % (uint64_t)g.ny), x = (uint32_t)(key / ((uint64_t)g.nz * (uint64_t)g.ny));
    atomicOr(&occ[occ_word_of(x >> 2, y >> 2, z >> 2, mask)], 1ull << (((x & 3u) << 4) | ((y & 3u) << 2) | (z & 3u)));
}
hipError_t knn_bitmap_build(const uint64_t* sorted_keys, const uint32_t* starts, size_t n_cells, KnnGrid g, void* occ_words, uint32_t word_mask, hipStream_t s)
{
    if (!n_cells) return hipSuccess;
    k_knn_bitmap_build<<<dim3(grid_for(n_cells)), dim3(kBlock), 0, s>>>(sorted_keys, starts, n_cells, g, reinterpret_cast<unsigned long long*>(occ_words), word_mask);
    return hipGetLastError();
}

// phase-1 verdict from the query's own cell: 1 = certainly coexist (bucket), 0 = certainly diff (outside the grid), 2 = undecided
template <int KT>
__device__ __forceinline__ int knn_bucket_coexist(float qx, float qy, float qz, const KnnGrid& g, const KnnBucketRaw* __restrict__ buckets, uint32_t n_buckets,
                                                   float cell_m, float dist_slack, float k_thr_lo)
{
    const double tx = ((double)qx - g.ox) * g.inv_cell, ty = ((double)qy - g.oy) * g.inv_cell, tz = ((double)qz - g.oz) * g.inv_cell;
    const double fx = floor(tx), fy = floor(ty), fz = floor(tz);
    // outside the grid (NaN included): the occupied cells are 1 .. n-2 on every axis, so no target point is within a cell edge of such a
    // query: fewer than k neighbours inside the provably-complete radius, "diff" exactly as knn_near decides it
    if (!(fx >= 0.0 && fy >= 0.0 && fz >= 0.0 && fx < (double)g.nx && fy < (double)g.ny && fz < (double)g.nz)) return 0;
    const uint64_t key = ((uint64_t)(uint32_t)(int)fx * (uint32_t)g.ny + (uint32_t)(int)fy) * (uint32_t)g.nz + (uint32_t)(int)fz;   // == cell_id()
    const float ux = (float)(tx - fx), uy = (float)(ty - fy), uz = (float)(tz - fz);
    const uint4* bp = reinterpret_cast<const uint4*>(buckets + bucket_of(hash64(key), n_buckets));
    uint4 v0 = bp[0];
    if ((((uint64_t)v0.y << 32) | v0.x) != key) {
        bp = reinterpret_cast<const uint4*>(buckets + bucket_of(hash64b(key), n_buckets));
        v0 = bp[0];
        if ((((uint64_t)v0.y << 32) | v0.x) != key) return 2;
    }
    const uint4 v1 = bp[1], v2 = bp[2], v3 = bp[3];
    const uint32_t w[14] = {v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w};
    const uint32_t cnt = w[13] >> 16;
    if (cnt < (uint32_t)KT) return 2;
    float best[KT];
#pragma unroll
    for (int j = 0; j < KT; ++j) best[j] = __builtin_inff();
#pragma unroll
    for (int s = 0; s < kBucketPts; ++s) {
        auto h = [&](int i) { return (float)((i & 1) ? (w[i >> 1] >> 16) : (w[i >> 1] & 0xffffu)); };
        const float dx = ux - (h(3 * s) + 0.5f) * (1.0f / 65536.0f), dy = uy - (h(3 * s + 1) + 0.5f) * (1.0f / 65536.0f),
                    dz = uz - (h(3 * s + 2) + 0.5f) * (1.0f / 65536.0f);
        float d = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));      // squared distance in cell units
        d = ((uint32_t)s < cnt) ? d : __builtin_inff();
#pragma unroll
        for (int j = KT - 1; j >= 0; --j) {
            if (j > 0) best[j] = (d < best[j - 1]) ? best[j - 1] : fminf(best[j], d);
            else best[0] = fminf(best[0], d);
        }
    }
    float sum = 0.0f;
#pragma unroll
    for (int j = 0; j < KT; ++j) {
        const float dm = __builtin_sqrtf(best[j]) * cell_m + dist_slack;         // metres, upper bound of the true distance
        sum += dm * dm * (1.0f + 3.0e-6f);
    }
    return sum < k_thr_lo ? 1 : 2;
}

template <bool B2L_IDENTITY, int KT>
__global__ void __launch_bounds__(kBlock)
k_knn_fast(const float4* __restrict__ scans, const uint64_t* __restrict__ offsets, size_t kb, uint64_t first_pt,
           const double* __restrict__ poses, const double* __restrict__ inv_poses, HostMat34 b2l_h, KnnGrid g,
           const KnnBucketRaw* __restrict__ buckets, uint32_t n_buckets, float cell_m, float dist_slack, float k_thr_lo,
           uint8_t* __restrict__ coexist, float4* __restrict__ local_out, float4* __restrict__ gp_undecided)
{
    const size_t kf = kb + blockIdx.y;
    const uint64_t a = offsets[kf], local = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (local >= offsets[kf + 1] - a) return;
    const uint64_t gi = a + local, i = gi - first_pt;
    const float4 p4 = scans[gi];
    float3 p = make_float3(p4.x, p4.y, p4.z);
    if (B2L_IDENTITY) p = xform_identity(p); else p = xform(to_dev(b2l_h), p);      // Session.cpp:545 / :618 (quirk Q7)
    const float3 gp = xform(load_mat(poses + 12 * kf), p);
    float3 l = xform(load_mat(inv_poses + 12 * kf), gp);                               // :603-604
    if (B2L_IDENTITY) l = xform_identity(l); else l = xform(to_dev(b2l_h), l);
    local_out[i] = make_float4(l.x, l.y, l.z, p4.w);
    const uint8_t f = (uint8_t)knn_bucket_coexist<KT>(gp.x, gp.y, gp.z, g, buckets, n_buckets, cell_m, dist_slack, k_thr_lo);      // 2 = undecided
    coexist[i] = f;
    // phase 2 starts from this point instead of finding the keyframe (nine dependent loads), the scan point and the pose again (round 6)
    if (gp_undecided && f == 2) gp_undecided[i] = make_float4(gp.x, gp.y, gp.z, 0.0f);
}

// queue[pos[i]] = i for the undecided queries (pos = exclusive scan of flag == 2); *count = their number
__global__ void __launch_bounds__(kBlock)
k_knn_queue_scatter(const uint8_t* __restrict__ flag, const uint32_t* __restrict__ pos, uint64_t n, uint32_t* __restrict__ queue, uint32_t* __restrict__ count)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool und = flag[i] == 2;
    if (und) queue[pos[i]] = (uint32_t)i;
    if (i == n - 1) *count = pos[i] + (und ? 1u : 0u);
}

template <bool B2L_IDENTITY, int KT>
__global__ void __launch_bounds__(kBlock)
k_knn_slow(const float4* __restrict__ scans, const uint64_t* __restrict__ offsets, size_t kb, size_t ke, uint64_t first_pt,
           const double* __restrict__ poses, HostMat34 b2l_h, const float4* __restrict__ tgt, size_t Mt, KnnGrid g,
           const HashEntry* __restrict__ table, uint32_t mask, const unsigned long long* __restrict__ bitmap, uint32_t bitmap_mask, int k, float thr, float cell2_lo,
           const uint32_t* __restrict__ queue, const uint32_t* __restrict__ count, uint8_t* __restrict__ coexist)
{
    const uint32_t n = *count;
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < n; q += gridDim.x * blockDim.x) {
        const uint32_t i = queue[q];
        const uint64_t gi = first_pt + i;
        const size_t kf = find_kf(offsets, kb, ke, gi);
        const float4 p4 = scans[gi];
        float3 p = make_float3(p4.x, p4.y, p4.z);
        if (B2L_IDENTITY) p = xform_identity(p); else p = xform(to_dev(b2l_h), p);
        const float3 gp = xform(load_mat(poses + 12 * kf), p);
        coexist[i] = knn_near<KT>(gp.x, gp.y, gp.z, tgt, Mt, g, table, mask, k, thr, cell2_lo, bitmap, bitmap_mask) ? 1 : 0;
    }
}

struct FlagUndecided { __host__ __device__ uint32_t operator()(uint8_t v) const { return v == 2 ? 1u : 0u; } };

hipError_t knn_two_phase_fast(const float4* scans, const uint64_t* offsets_dev, size_t kb, size_t ke, uint64_t first_pt, uint64_t n_pts, uint64_t max_kf_pts,
                              const double* poses_dev, const double* inv_poses_dev, HostMat34 b2l, int b2l_identity, KnnGrid g, const void* buckets,
                              uint32_t n_buckets, int k, float thr, uint8_t* coexist, float4* local_out, hipStream_t s, float4* gp_undecided)
{
    if (!n_pts || !max_kf_pts) return hipSuccess;
    if (k < 1 || k > 4) return hipErrorInvalidValue;
    const float cell_m = (float)(1.0 / g.inv_cell);
    const float dist_slack = (float)((1.7320508075688772 * (1.0 / g.inv_cell) / 131072.0) * 1.05 + 2.0e-7);
    const float k_thr_lo = (float)((double)k * (double)thr * (1.0 - 1.0e-5));
    const KnnBucketRaw* bk = reinterpret_cast<const KnnBucketRaw*>(buckets);
    auto fast = [&](auto b2l_tag, auto kt_tag) {
        for (size_t k0 = kb; k0 < ke; k0 += 65535) {
            const size_t k1 = std::min(ke, k0 + 65535);
            k_knn_fast<decltype(b2l_tag)::value, decltype(kt_tag)::value><<<dim3(grid_for(max_kf_pts), (unsigned)(k1 - k0)), dim3(kBlock), 0, s>>>(
                scans, offsets_dev, k0, first_pt, poses_dev, inv_poses_dev, b2l, g, bk, n_buckets, cell_m, dist_slack, k_thr_lo, coexist, local_out, gp_undecided);
        }
    };
    auto by_kt = [&](auto b2l_tag) {
        switch (k) {
        case 1: fast(b2l_tag, std::integral_constant<int, 1>{}); break;
        case 2: fast(b2l_tag, std::integral_constant<int, 2>{}); break;
        case 3: fast(b2l_tag, std::integral_constant<int, 3>{}); break;
        default: fast(b2l_tag, std::integral_constant<int, 4>{}); break;
        }
    };
    if (b2l_identity) by_kt(std::true_type{}); else by_kt(std::false_type{});
    return hipGetLastError();
}

hipError_t knn_two_phase_exact(const float4* scans, const uint64_t* offsets_dev, size_t kb, size_t ke, uint64_t first_pt, uint64_t n_pts,
                               const double* poses_dev, HostMat34 b2l, int b2l_identity, const float4* sorted_target, size_t Mt, KnnGrid g,
                               const HashEntry* table, uint32_t table_mask, const void* bitmap, uint32_t bitmap_mask, int k, float thr, float cell2_lo, uint8_t* coexist,
                               uint32_t* pos, uint32_t* queue, uint32_t* count, void* temp, size_t temp_bytes, hipStream_t s)
{
    if (!n_pts) return hipSuccess;
    if (k < 1 || k > 4 || Mt < (size_t)k) return hipErrorInvalidValue;
    auto it = rocprim::make_transform_iterator(coexist, FlagUndecided());
    hipError_t e = rocprim::exclusive_scan(temp, temp_bytes, it, pos, 0u, n_pts, rocprim::plus<uint32_t>(), s);
    if (e != hipSuccess) return e;
    k_knn_queue_scatter<<<dim3(grid_for(n_pts)), dim3(kBlock), 0, s>>>(coexist, pos, n_pts, queue, count);
    const unsigned blocks = (unsigned)std::min<uint64_t>(grid_for(n_pts), 4096);
    auto slow = [&](auto b2l_tag, auto kt_tag) {
        k_knn_slow<decltype(b2l_tag)::value, decltype(kt_tag)::value><<<dim3(blocks), dim3(kBlock), 0, s>>>(
            scans, offsets_dev, kb, ke, first_pt, poses_dev, b2l, sorted_target, Mt, g, table, table_mask, reinterpret_cast<const unsigned long long*>(bitmap), bitmap_mask, k, thr,
            cell2_lo, queue, count, coexist);
    };
    auto by_kt = [&](auto b2l_tag) {
        switch (k) {
        case 1: slow(b2l_tag, std::integral_constant<int, 1>{}); break;
        case 2: slow(b2l_tag, std::integral_constant<int, 2>{}); break;
        case 3: slow(b2l_tag, std::integral_constant<int, 3>{}); break;
        default: slow(b2l_tag, std::integral_constant<int, 4>{}); break;
        }
    };
    if (b2l_identity) by_kt(std::true_type{}); else by_kt(std::false_type{});
    return hipGetLastError();
}

// ---- phase 2 over a queue SORTED BY CELL (round 4).  The undecided queries of a scan arrive in ring order, i.e. scattered over the whole
// map: every lane of a wavefront probed its own 27 cells, table entries and point runs in cache lines nobody else wanted (k_knn_slow:
// 7 ms for a quarter of the queries on the lot, 48 ms at KITTI scale).  Here the compaction writes (cell id << index bits | query index),
// a keys-only radix sort over the cell bits brings the queries of one cell -- and of neighbouring cells along z and y -- together, and the
// same exact search then runs on wavefronts whose lanes share their bitmap words, table entries and target points.  Answers are written
// back through the query index, so the order of the queue cannot change any flag.
template <bool B2L_IDENTITY>
__global__ void __launch_bounds__(kBlock)
k_knn_queue_scatter_keyed(const uint8_t* __restrict__ flag, const uint32_t* __restrict__ pos, uint64_t n, const float4* __restrict__ scans,
                          const uint64_t* __restrict__ offsets, size_t kb, size_t ke, uint64_t first_pt, const double* __restrict__ poses, HostMat34 b2l_h, KnnGrid g,
                          unsigned ibits, uint64_t* __restrict__ queue, uint32_t* __restrict__ count, const float4* __restrict__ gp_undecided)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool und = flag[i] == 2;
    if (und) {
        float3 gp;
        if (gp_undecided) { const float4 v = gp_undecided[i]; gp = make_float3(v.x, v.y, v.z); }      // what phase 1 computed for this query
        else {
            const uint64_t gi = first_pt + i;
            const size_t kf = find_kf(offsets, kb, ke, gi);
            const float4 p4 = scans[gi];
            float3 p = make_float3(p4.x, p4.y, p4.z);
            if (B2L_IDENTITY) p = xform_identity(p); else p = xform(to_dev(b2l_h), p);
            gp = xform(load_mat(poses + 12 * kf), p);
        }
        // undecided queries lie inside the grid (phase 1 answered the others): the same cell arithmetic as knn_near / knn_bucket_coexist
        const double fx = floor(((double)gp.x - g.ox) * g.inv_cell), fy = floor(((double)gp.y - g.oy) * g.inv_cell), fz = floor(((double)gp.z - g.oz) * g.inv_cell);
        const uint64_t cx = (uint64_t)min(max((long long)fx, 0ll), g.nx - 1), cy = (uint64_t)min(max((long long)fy, 0ll), g.ny - 1), cz = (uint64_t)min(max((long long)fz, 0ll), g.nz - 1);
        const uint64_t key = (cx * (uint64_t)g.ny + cy) * (uint64_t)g.nz + cz;
        queue[pos[i]] = (key << ibits) | i;
    }
    if (i == n - 1) *count = pos[i] + (und ? 1u : 0u);
}
template <bool B2L_IDENTITY, int KT>
__global__ void __launch_bounds__(kBlock)
k_knn_slow_sorted(const float4* __restrict__ scans, const uint64_t* __restrict__ offsets, size_t kb, size_t ke, uint64_t first_pt,
                  const double* __restrict__ poses, HostMat34 b2l_h, const float4* __restrict__ tgt, size_t Mt, KnnGrid g,
                  const HashEntry* __restrict__ table, uint32_t mask, const unsigned long long* __restrict__ bitmap, uint32_t bitmap_mask, int k, float thr, float cell2_lo,
                  const uint64_t* __restrict__ queue, uint32_t n, uint64_t imask, uint8_t* __restrict__ coexist, const float4* __restrict__ gp_undecided)
{
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < n; q += gridDim.x * blockDim.x) {
        const uint64_t i = queue[q] & imask;
        float3 gp;
        if (gp_undecided) { const float4 v = gp_undecided[i]; gp = make_float3(v.x, v.y, v.z); }
        else {
            const uint64_t gi = first_pt + i;
            const size_t kf = find_kf(offsets, kb, ke, gi);
            const float4 p4 = scans[gi];
            float3 p = make_float3(p4.x, p4.y, p4.z);
            if (B2L_IDENTITY) p = xform_identity(p); else p = xform(to_dev(b2l_h), p);
            gp = xform(load_mat(poses + 12 * kf), p);
        }
        coexist[i] = knn_near<KT>(gp.x, gp.y, gp.z, tgt, Mt, g, table, mask, k, thr, cell2_lo, bitmap, bitmap_mask) ? 1 : 0;
    }
}
// bits of the largest cell id + bits of the largest query index; 0 if they do not fit one word (the caller then keeps the unsorted queue)
unsigned knn_sorted_queue_bits(KnnGrid g, uint64_t n_pts, unsigned* ibits_out)
{
    const unsigned __int128 cells = (unsigned __int128)(uint64_t)g.nx * (uint64_t)g.ny * (uint64_t)g.nz;
    unsigned kbits = 1, ibits = 1;
    while (kbits < 64 && ((unsigned __int128)1 << kbits) < cells) ++kbits;
    while (ibits < 63 && (1ull << ibits) < n_pts) ++ibits;
    if (ibits_out) *ibits_out = ibits;
    return kbits + ibits <= 64 ? kbits : 0u;
}
hipError_t knn_two_phase_compact_keyed(const float4* scans, const uint64_t* offsets_dev, size_t kb, size_t ke, uint64_t first_pt, uint64_t n_pts, const double* poses_dev,
                                       HostMat34 b2l, int b2l_identity, KnnGrid g, unsigned ibits, const uint8_t* flags, uint32_t* pos, uint64_t* queue, uint32_t* count,
                                       void* temp, size_t temp_bytes, hipStream_t s, const float4* gp_undecided)
{
    if (!n_pts) return hipMemsetAsync(count, 0, 4, s);
    auto it = rocprim::make_transform_iterator(flags, FlagUndecided());
    hipError_t e = rocprim::exclusive_scan(temp, temp_bytes, it, pos, 0u, n_pts, rocprim::plus<uint32_t>(), s);
    if (e != hipSuccess) return e;
    if (b2l_identity) k_knn_queue_scatter_keyed<true><<<dim3(grid_for(n_pts)), dim3(kBlock), 0, s>>>(flags, pos, n_pts, scans, offsets_dev, kb, ke, first_pt, poses_dev, b2l, g, ibits, queue, count, gp_undecided);
    else k_knn_queue_scatter_keyed<false><<<dim3(grid_for(n_pts)), dim3(kBlock), 0, s>>>(flags, pos, n_pts, scans, offsets_dev, kb, ke, first_pt, poses_dev, b2l, g, ibits, queue, count, gp_undecided);
    return hipGetLastError();
}
hipError_t knn_two_phase_exact_sorted(const float4* scans, const uint64_t* offsets_dev, size_t kb, size_t ke, uint64_t first_pt, const double* poses_dev, HostMat34 b2l,
                                      int b2l_identity, const float4* sorted_target, size_t Mt, KnnGrid g, const HashEntry* table, uint32_t table_mask, const void* bitmap,
                                      uint32_t bitmap_mask, int k, float thr, float cell2_lo, uint8_t* coexist, const uint64_t* queue_in, uint64_t* queue_sorted, uint32_t n_und,
                                      unsigned ibits, unsigned kbits, void* temp, size_t temp_bytes, hipStream_t s, const float4* gp_undecided)
{
    if (!n_und) return hipSuccess;
    if (k < 1 || k > 4 || Mt < (size_t)k) return hipErrorInvalidValue;
    hipError_t e = rocprim::radix_sort_keys(temp, temp_bytes, queue_in, queue_sorted, (size_t)n_und, ibits, ibits + kbits, s);
    if (e != hipSuccess) return e;
    const unsigned blocks = (unsigned)std::min<uint64_t>(grid_for(n_und), 8192);
    const uint64_t imask = (1ull << ibits) - 1ull;
    auto slow = [&](auto b2l_tag, auto kt_tag) {
        k_knn_slow_sorted<decltype(b2l_tag)::value, decltype(kt_tag)::value><<<dim3(blocks), dim3(kBlock), 0, s>>>(
            scans, offsets_dev, kb, ke, first_pt, poses_dev, b2l, sorted_target, Mt, g, table, table_mask, reinterpret_cast<const unsigned long long*>(bitmap), bitmap_mask, k, thr,
            cell2_lo, queue_sorted, n_und, imask, coexist, gp_undecided);
    };
    auto by_kt = [&](auto b2l_tag) {
        switch (k) {
        case 1: slow(b2l_tag, std::integral_constant<int, 1>{}); break;
        case 2: slow(b2l_tag, std::integral_constant<int, 2>{}); break;
        case 3: slow(b2l_tag, std::integral_constant<int, 3>{}); break;
        default: slow(b2l_tag, std::integral_constant<int, 4>{}); break;
        }
    };
    if (b2l_identity) by_kt(std::true_type{}); else by_kt(std::false_type{});
    return hipGetLastError();
}

template <int KT>
__global__ void __launch_bounds__(kBlock)
k_knn_query_cloud(const float4* __restrict__ query, size_t Q, const float4* __restrict__ tgt, size_t Mt, KnnGrid g,
                  const HashEntry* __restrict__ table, uint32_t mask, int k, float thr, float cell2_lo, uint8_t* __restrict__ near)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Q) return;
    const float4 q = query[i];
    near[i] = knn_near<KT>(q.x, q.y, q.z, tgt, Mt, g, table, mask, k, thr, cell2_lo) ? 1 : 0;
}
hipError_t knn_query_cloud(const float4* query, size_t Q, const float4* sorted_target, size_t Mt, KnnGrid g, const HashEntry* table,
                           uint32_t table_mask, int k, float thr, float cell2_lo, uint8_t* near, hipStream_t s)
{
    if (!Q) return hipSuccess;
    if (k < 1 || k > kMaxK) return hipErrorInvalidValue;
    const int kt = (k <= 4 && Mt >= (size_t)k) ? k : 0;
    auto launch = [&](auto kt_tag) {
        k_knn_query_cloud<decltype(kt_tag)::value><<<dim3(grid_for(Q)), dim3(kBlock), 0, s>>>(query, Q, sorted_target, Mt, g, table, table_mask, k, thr, cell2_lo, near);
    };
    switch (kt) {
    case 1: launch(std::integral_constant<int, 1>{}); break;
    case 2: launch(std::integral_constant<int, 2>{}); break;
    case 3: launch(std::integral_constant<int, 3>{}); break;
    case 4: launch(std::integral_constant<int, 4>{}); break;
    default: launch(std::integral_constant<int, 0>{}); break;
    }
    return hipGetLastError();
}


} // namespace ltm
