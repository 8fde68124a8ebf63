// ltm_k_stream.hip -- prefix sums, partition / gather / reprojection gather, rigid transforms, zip-concat, pre-clean, RViz images (Removerter.cpp:675-687, 933-946; utility.cpp:64-89, 160-202; Session.cpp:506-533)
// (gfx950 / CDNA4, wave64; part of libltm_hip.so -- shared definitions in ltm_kernels_common.h, launch wrappers declared in ltm_kernels.h)
#include "ltm_kernels_common.h"
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/transform_iterator.hpp>
namespace ltm {

// ------------------------------------------------------------------------------------ scans
struct U8Flag { __host__ __device__ uint32_t operator()(uint8_t v) const { return v ? 1u : 0u; } };
struct ImgValid { __host__ __device__ uint32_t operator()(uint64_t v) const { return ((uint32_t)v) ? 1u : 0u; } };

size_t scan_temp_bytes(size_t n)
{
    size_t bytes = 0;
    uint32_t* d = nullptr;
    (void)rocprim::exclusive_scan(nullptr, bytes, d, d, 0u, n ? n : 1, rocprim::plus<uint32_t>());
    return bytes + 256;
}
hipError_t exclusive_scan_u8(const uint8_t* labels, uint32_t* pos, size_t n, void* temp, size_t temp_bytes, hipStream_t s)
{
    if (!n) return hipSuccess;
    auto it = rocprim::make_transform_iterator(labels, U8Flag());
    return rocprim::exclusive_scan(temp, temp_bytes, it, pos, 0u, n, rocprim::plus<uint32_t>(), s);
}
hipError_t exclusive_scan_img_valid(const uint64_t* img, uint32_t* pos, size_t n, void* temp, size_t temp_bytes, hipStream_t s)
{
    if (!n) return hipSuccess;
    auto it = rocprim::make_transform_iterator(img, ImgValid());
    return rocprim::exclusive_scan(temp, temp_bytes, it, pos, 0u, n, rocprim::plus<uint32_t>(), s);
}
hipError_t exclusive_scan_u32(const uint32_t* in, uint32_t* out, size_t n, void* temp, size_t temp_bytes, hipStream_t s)
{
    if (!n) return hipSuccess;
    return rocprim::exclusive_scan(temp, temp_bytes, in, out, 0u, n, rocprim::plus<uint32_t>(), s);
}

__global__ void __launch_bounds__(kBlock)
k_partition_scatter(const float4* __restrict__ in, const uint8_t* __restrict__ labels, const uint32_t* __restrict__ pos, size_t n,
                    float4* __restrict__ kept, float4* __restrict__ flagged)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 p = in[i];
    const uint32_t f = pos[i];
    if (labels[i]) { if (flagged) flagged[f] = p; }
    else if (kept) kept[i - f] = p;
}
hipError_t partition_scatter(const float4* in, const uint8_t* labels, const uint32_t* pos, size_t n, float4* kept, float4* flagged,
                             hipStream_t s)
{
    if (!n) return hipSuccess;
    k_partition_scatter<<<dim3(grid_for(n)), dim3(kBlock), 0, s>>>(in, labels, pos, n, kept, flagged);
    return hipGetLastError();
}

template <bool B2L_IDENTITY>
__global__ void __launch_bounds__(kBlock)
k_reproject_gather(const uint64_t* __restrict__ img, const uint32_t* __restrict__ pos, size_t npx, size_t total,
                   const float4* __restrict__ map, const double* __restrict__ inv_poses, size_t kb, HostMat34 b2l_h,
                   float4* __restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const uint32_t idx = (uint32_t)img[i];
    if (idx == 0) return;                                   // utility.cpp:82 -- 0 doubles as "no point"
    const size_t kf = kb + i / npx;
    const Mat34 Tinv = load_mat(inv_poses + 12 * kf);
    const float4 p4 = map[idx];
    float3 p = xform(Tinv, make_float3(p4.x, p4.y, p4.z));
    if (B2L_IDENTITY) p = xform_identity(p); else p = xform(to_dev(b2l_h), p);
    out[pos[i]] = make_float4(p.x, p.y, p.z, p4.w);
}
hipError_t reproject_gather(const uint64_t* img, const uint32_t* pos, size_t npx, size_t nb, const float4* map,
                            const double* inv_poses_dev, size_t kb, HostMat34 b2l, int b2l_identity, float4* out, hipStream_t s)
{
    const size_t total = npx * nb;
    if (!total) return hipSuccess;
    if (b2l_identity) k_reproject_gather<true><<<dim3(grid_for(total)), dim3(kBlock), 0, s>>>(img, pos, npx, total, map, inv_poses_dev, kb, b2l, out);
    else k_reproject_gather<false><<<dim3(grid_for(total)), dim3(kBlock), 0, s>>>(img, pos, npx, total, map, inv_poses_dev, kb, b2l, out);
    return hipGetLastError();
}

// out[j] = number of emitted points before image j (j = 0..nb) given the exclusive scan `pos` of the valid-pixel flags of nb images of
// npx pixels: the per-keyframe boundaries and the total of a reprojection in ONE small array (one host round trip instead of three)
__global__ void k_image_bounds(const uint32_t* __restrict__ pos, const uint64_t* __restrict__ img, size_t npx, size_t nb, uint32_t* __restrict__ out)
{
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j > nb) return;
    if (j < nb) out[j] = pos[j * npx];
    else { const size_t last = nb * npx - 1; out[j] = pos[last] + (((uint32_t)img[last]) ? 1u : 0u); }
}
hipError_t image_bounds(const uint32_t* pos, const uint64_t* img, size_t npx, size_t nb, uint32_t* out, hipStream_t s)
{
    if (!nb || !npx) return hipSuccess;
    k_image_bounds<<<dim3(grid_for(nb + 1)), dim3(kBlock), 0, s>>>(pos, img, npx, nb, out);
    return hipGetLastError();
}

// out[j] (j = 0..nb) = number of set flags before point offsets[kf0 + j] - first, given the exclusive scan `pos` of `flag` over n
// points; a boundary at or past n (the last one) yields the total.  Per-keyframe output boundaries + total in one small array.
__global__ void k_flag_bounds(const uint32_t* __restrict__ pos, const uint8_t* __restrict__ flag, size_t n, const uint64_t* __restrict__ offsets,
                              size_t kf0, uint64_t first, size_t nb, uint32_t* __restrict__ out)
{
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j > nb) return;
    const uint64_t at = offsets[kf0 + j] - first;
    out[j] = (at < n) ? pos[at] : (pos[n - 1] + (flag[n - 1] ? 1u : 0u));
}
hipError_t flag_bounds(const uint32_t* pos, const uint8_t* flag, size_t n, const uint64_t* offsets_dev, size_t kf0, uint64_t first, size_t nb,
                       uint32_t* out, hipStream_t s)
{
    if (!n) return hipSuccess;
    k_flag_bounds<<<dim3(grid_for(nb + 1)), dim3(kBlock), 0, s>>>(pos, flag, n, offsets_dev, kf0, first, nb, out);
    return hipGetLastError();
}

__global__ void k_gather_u32(const uint32_t* in, const uint64_t* idx, size_t m, size_t n, uint32_t tail, uint32_t* out)
{
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const uint64_t i = idx[j];
    out[j] = (i < n) ? in[i] : tail;
}
// out[j] = in[idx[j]] on 64-bit words (the host-ordered form of the loader's voxel grid permutes its keys with the point indices)
__global__ void __launch_bounds__(kBlock) k_gather_u64_by_u32(const uint64_t* __restrict__ in, const uint32_t* __restrict__ idx, size_t n, uint64_t* __restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[idx[i]];
}
hipError_t gather_u64_by_u32(const uint64_t* in, const uint32_t* idx_dev, size_t n, uint64_t* out, hipStream_t s)
{
    if (!n) return hipSuccess;
    k_gather_u64_by_u32<<<dim3(grid_for(n)), dim3(kBlock), 0, s>>>(in, idx_dev, n, out);
    return hipGetLastError();
}
hipError_t gather_u32(const uint32_t* in, const uint64_t* idx_dev, size_t m, size_t n, uint32_t tail, uint32_t* out, hipStream_t s)
{
    if (!m) return hipSuccess;
    k_gather_u32<<<dim3(grid_for(m)), dim3(kBlock), 0, s>>>(in, idx_dev, m, n, tail, out);
    return hipGetLastError();
}

// -------------------------------------------------------------------------------- transforms

template <bool FIRST_IDENTITY>
__global__ void __launch_bounds__(kBlock)
k_transform_scans(const float4* __restrict__ in, const uint64_t* __restrict__ offsets, size_t n_kf, uint64_t n_pts,
                  HostMat34 first_h, const double* __restrict__ per_kf, float4* __restrict__ out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pts) return;
    const size_t kf = find_kf(offsets, 0, n_kf, i);
    const float4 p4 = in[i];
    float3 p = make_float3(p4.x, p4.y, p4.z);
    if (FIRST_IDENTITY) p = xform_identity(p); else p = xform(to_dev(first_h), p);
    p = xform(load_mat(per_kf + 12 * kf), p);
    out[i] = make_float4(p.x, p.y, p.z, p4.w);
}
hipError_t transform_scans(const float4* in, const uint64_t* offsets_dev, size_t n_kf, uint64_t n_pts, HostMat34 first,
                           int first_identity, const double* per_kf_dev, float4* out, hipStream_t s)
{
    if (!n_pts) return hipSuccess;
    if (first_identity) k_transform_scans<true><<<dim3(grid_for(n_pts)), dim3(kBlock), 0, s>>>(in, offsets_dev, n_kf, n_pts, first, per_kf_dev, out);
    else k_transform_scans<false><<<dim3(grid_for(n_pts)), dim3(kBlock), 0, s>>>(in, offsets_dev, n_kf, n_pts, first, per_kf_dev, out);
    return hipGetLastError();
}

// pcl::transformPointCloud<PointT, double> applied once or twice to a whole cloud (utility.cpp:64-72, 160-168, 194-202): the float
// result of the first transform is the input of the second; intensity is copied
__global__ void __launch_bounds__(kBlock)
k_transform_cloud(const float4* __restrict__ in, size_t n, int has1, HostMat34 t1, int has2, HostMat34 t2, float4* __restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 p4 = in[i];
    float3 p = make_float3(p4.x, p4.y, p4.z);
    if (has1) p = xform(to_dev(t1), p);
    if (has2) p = xform(to_dev(t2), p);
    out[i] = make_float4(p.x, p.y, p.z, p4.w);
}
hipError_t transform_cloud(const float4* in, size_t n, const HostMat34* t1, const HostMat34* t2, float4* out, hipStream_t s)
{
    if (!n) return hipSuccess;
    HostMat34 z{};
    k_transform_cloud<<<dim3(grid_for(n)), dim3(kBlock), 0, s>>>(in, n, t1 != nullptr, t1 ? *t1 : z, t2 != nullptr, t2 ? *t2 : z, out);
    return hipGetLastError();
}

// per-keyframe concatenation a[k] ++ b[k] ++ c[k] (Session.cpp:365-371) as one gather: out_off = offsets of the result
__global__ void __launch_bounds__(kBlock)
k_zip_concat(const float4* __restrict__ a, const uint64_t* __restrict__ oa, const float4* __restrict__ b, const uint64_t* __restrict__ ob,
             const float4* __restrict__ c, const uint64_t* __restrict__ oc, const uint64_t* __restrict__ out_off, size_t n_kf, uint64_t n,
             float4* __restrict__ out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t k = find_kf(out_off, 0, n_kf, i);
    uint64_t j = i - out_off[k];
    const uint64_t na = oa[k + 1] - oa[k], nb = ob[k + 1] - ob[k];
    if (j < na) { out[i] = a[oa[k] + j]; return; }
    j -= na;
    if (j < nb) { out[i] = b[ob[k] + j]; return; }
    j -= nb;
    out[i] = c[oc[k] + j];
}
hipError_t zip_concat(const float4* a, const uint64_t* oa, const float4* b, const uint64_t* ob, const float4* c, const uint64_t* oc,
                      const uint64_t* out_off_dev, size_t n_kf, uint64_t n, float4* out, hipStream_t s)
{
    if (!n) return hipSuccess;
    k_zip_concat<<<dim3(grid_for(n)), dim3(kBlock), 0, s>>>(a, oa, b, ob, c, oc, out_off_dev, n_kf, n, out);
    return hipGetLastError();
}

// Session.cpp:506-533: drop iff (range < radius) & (z < 0.5) & (-0.5 < z)
__global__ void __launch_bounds__(kBlock)
k_preclean_flags(const float4* __restrict__ in, uint64_t n, float radius, uint8_t* __restrict__ drop)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 p = in[i];
    const float r = __builtin_sqrtf((p.x * p.x + p.y * p.y) + p.z * p.z);
    drop[i] = ((r < radius) & (p.z < 0.5f) & (-0.5f < p.z)) ? 1 : 0;
}
hipError_t preclean_flags(const float4* in, uint64_t n, float radius, uint8_t* drop, hipStream_t s)
{
    if (!n) return hipSuccess;
    k_preclean_flags<<<dim3(grid_for(n)), dim3(kBlock), 0, s>>>(in, n, radius, drop);
    return hipGetLastError();
}

// ---- RViz images: convertColorMappedImg (utility.h:114-127) evaluated on the device
__global__ void __launch_bounds__(kBlock)
k_viz_diff(const uint32_t* __restrict__ scan_bits, const float* __restrict__ map_r, size_t n, int mode, float* __restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float s = u2f(scan_bits[i]), m = map_r[i];
    out[i] = mode == 0 ? s - m : m - s;          // Removerter.cpp:572 (scan - map) / :519 (map - scan)
}
__global__ void __launch_bounds__(kBlock)
k_viz_colormap_f32(const float* __restrict__ src, size_t n, float a, float b, const uint8_t* __restrict__ lut, uint8_t* __restrict__ bgr)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = src[i] * a + b;              // cv::Mat::convertTo on a float image: float multiply-add, not fused
    int q = (v != v) ? 0 : (v <= -1.0f ? 0 : (v >= 256.0f ? 255 : __float2int_rn(v)));   // saturate_cast<uchar>: round half to even
    q = min(max(q, 0), 255);
    bgr[3 * i + 0] = lut[3 * q + 0]; bgr[3 * i + 1] = lut[3 * q + 1]; bgr[3 * i + 2] = lut[3 * q + 2];
}
__global__ void __launch_bounds__(kBlock)
k_viz_colormap_i32(const int32_t* __restrict__ src, size_t n, double a, double b, const uint8_t* __restrict__ lut, uint8_t* __restrict__ bgr)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = (double)src[i] * a + b;     // int32 image: double arithmetic, rounded to int, then saturated to 8 bits
    int q = v <= -1.0 ? 0 : (v >= 256.0 ? 255 : __double2int_rn(v));
    q = min(max(q, 0), 255);
    bgr[3 * i + 0] = lut[3 * q + 0]; bgr[3 * i + 1] = lut[3 * q + 1]; bgr[3 * i + 2] = lut[3 * q + 2];
}
hipError_t viz_diff(const uint32_t* scan_bits, const float* map_r, size_t n, int mode, float* out, hipStream_t s)
{
    if (!n) return hipSuccess;
    k_viz_diff<<<dim3(grid_for(n)), dim3(kBlock), 0, s>>>(scan_bits, map_r, n, mode, out);
    return hipGetLastError();
}
hipError_t viz_colormap_f32(const float* src, size_t n, float a, float b, const uint8_t* lut, uint8_t* bgr, hipStream_t s)
{
    if (!n) return hipSuccess;
    k_viz_colormap_f32<<<dim3(grid_for(n)), dim3(kBlock), 0, s>>>(src, n, a, b, lut, bgr);
    return hipGetLastError();
}
hipError_t viz_colormap_i32(const int32_t* src, size_t n, double a, double b, const uint8_t* lut, uint8_t* bgr, hipStream_t s)
{
    if (!n) return hipSuccess;
    k_viz_colormap_i32<<<dim3(grid_for(n)), dim3(kBlock), 0, s>>>(src, n, a, b, lut, bgr);
    return hipGetLastError();
}


} // namespace ltm
