// ltm_kernels_common.h -- what the kernel translation units of libltm_hip.so share (internal; the launch wrappers are declared in ltm_kernels.h).
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (bit-exact parity with the reference's non-FMA x86-64 arithmetic; see ltm_device_math.h).
//
// Data layout in HBM: clouds are float4 XYZI arrays (16 B/pt, one coalesced dwordx4 load per lane); a range image is ONE 64-bit word per pixel,
// (range_bits << 32) | point_index, so that the serial reference rule "strictly smaller range wins, lowest index wins ties" (utility.cpp:134-138) is
// a single order on uint64 and a single global_atomic_umin_x2; scan images only need the range (u32).  Images of a whole batch of keyframes are
// resident at once ([kf][row][col]); one launch covers (map tiles) x (keyframes).
#pragma once
#include "ltm_kernels.h"
#include "ltm_device_math.h"

#include <algorithm>
#include <cstring>
#include <type_traits>
#include <hip/hip_runtime.h>

namespace ltm {

static constexpr int kBlock = 256;

__host__ __device__ inline RimgGeom make_geom(Geom g)
{
    RimgGeom r;
    r.vfov = g.vfov; r.hfov = g.hfov;
    r.half_v = g.vfov / 2.0f; r.half_h = g.hfov / 2.0f;
    r.inv_v = 1.0f / g.vfov; r.inv_h = 1.0f / g.hfov;
    r.fast = g.fast != 0;
    r.eps = g.cull_eps_px;
    r.el_c0 = g.el_c[0]; r.el_c1 = g.el_c[1]; r.el_c2 = g.el_c[2]; r.el_c3 = g.el_c[3]; r.el_tclamp = g.el_tclamp; r.el_fit = g.el_fit != 0;
    r.rows = g.rows; r.cols = g.cols;
    r.frows = (float)g.rows; r.fcols = (float)g.cols;
    r.row_max = (float)(g.rows - 1); r.col_max = (float)(g.cols - 1);
    return r;
}

inline unsigned grid_for(size_t n, int block = kBlock)
{
    size_t b = (n + block - 1) / block;
    return (unsigned)(b == 0 ? 1 : b);
}

__device__ __forceinline__ Mat34 load_mat(const double* p)
{
    Mat34 T;
#pragma unroll
    for (int i = 0; i < 12; ++i) T.m[i] = p[i];
    return T;
}
__device__ __forceinline__ Mat34 to_dev(const HostMat34& h)
{
    Mat34 T;
#pragma unroll
    for (int i = 0; i < 12; ++i) T.m[i] = h.m[i];
    return T;
}

// range-min with a relaxed pre-test: the image only ever decreases, so a stale (larger) value read
// can only let a redundant atomic through, never suppress a needed one.
__device__ __forceinline__ void img_min_u64(uint64_t* p, uint64_t v)
{
    const uint64_t cur = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v < cur) atomicMin(reinterpret_cast<unsigned long long*>(p), (unsigned long long)v);
}
__device__ __forceinline__ void img_min_u32(uint32_t* p, uint32_t v)
{
    const uint32_t cur = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v < cur) atomicMin(p, v);
}

// keyframe of global point index gi (binary search over the offset table)
__device__ __forceinline__ size_t find_kf(const uint64_t* __restrict__ offsets, size_t lo, size_t hi, uint64_t gi)
{
    while (hi - lo > 1) {
        const size_t mid = (lo + hi) >> 1;
        if (offsets[mid] <= gi) lo = mid; else hi = mid;
    }
    return lo;
}

} // namespace ltm
