// ltm_k_projection.hip -- range-image projection, the remove / revert visibility vote, the exact arg-min image and its occlusion cull (utility.cpp:38-142, Removerter.cpp:109-156, 381-593)
// (gfx950 / CDNA4, wave64; part of libltm_hip.so -- shared definitions in ltm_kernels_common.h, launch wrappers declared in ltm_kernels.h)
#include "ltm_kernels_common.h"
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

namespace ltm {

// ------------------------------------------------------------------------------------ fills
__global__ void k_fill_u32(uint32_t* p, uint32_t v, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = v;
}
__global__ void k_fill_u64(uint64_t* p, uint64_t v, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = v;
}
hipError_t fill_u32(uint32_t* p, uint32_t v, size_t n, hipStream_t s)
{
    if (!n) return hipSuccess;
    k_fill_u32<<<dim3((unsigned)std::min<size_t>(grid_for(n), 8192)), dim3(kBlock), 0, s>>>(p, v, n);
    return hipGetLastError();
}
hipError_t fill_u64(uint64_t* p, uint64_t v, size_t n, hipStream_t s)
{
    if (!n) return hipSuccess;
    k_fill_u64<<<dim3((unsigned)std::min<size_t>(grid_for(n), 8192)), dim3(kBlock), 0, s>>>(p, v, n);
    return hipGetLastError();
}

// Removerter.cpp:109-156 scan2RangeImg, all keyframes of a batch in one launch.
__global__ void __launch_bounds__(kBlock)
k_scan_rimg(const float4* __restrict__ scans, const uint64_t* __restrict__ offsets, size_t kb, size_t nb,
            uint64_t first_pt, uint64_t n_pts, Geom gg, uint32_t* __restrict__ img)
{
    // grid = (chunks of the longest keyframe, keyframes): the keyframe comes from blockIdx.y instead of a binary search over
    // the offsets (nine dependent loads per point)
    const size_t lo = kb + blockIdx.y;
    const uint64_t a = offsets[lo], local = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (local >= offsets[lo + 1] - a) return;
    const uint64_t gi = a + local;
    const RimgGeom g = make_geom(gg);
    const float4 p = scans[gi];
    const Sph s = cart2sph(p.x, p.y, p.z);
    const int px = pixel_index(g, s.az, s.el);
    img_min_u32(img + (lo - kb) * (size_t)(g.rows * g.cols) + px, f2u(s.r));
}

// The same for up to kMaxScanShapes image shapes at once (round 6): the remove / revert passes of selfRemovert project every scan into six shapes
// (2.5, 2.375, 2.0, 1.9, 1.5, 1.425 pixels per degree).  The point is read once, the exact atan2f / sqrtf chain and the two divisions by the field of
// view (shape-independent) run once, and each shape costs a multiply, a round, a clamp and its atomic -- bit for bit pixel_row_col's arithmetic.
struct ScanShapes { int n; int rows[kMaxScanShapes], cols[kMaxScanShapes]; uint32_t* img[kMaxScanShapes]; };
__global__ void __launch_bounds__(kBlock)
k_scan_rimg_multi(const float4* __restrict__ scans, const uint64_t* __restrict__ offsets, size_t kb, Geom gg, ScanShapes sh)
{
    const size_t lo = kb + blockIdx.y;
    const uint64_t a = offsets[lo], local = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (local >= offsets[lo + 1] - a) return;
    const RimgGeom g = make_geom(gg);      // rows / cols of gg are not used: only the field of view and the fast-form switch
    const float4 p = scans[a + local];
    const Sph s = cart2sph(p.x, p.y, p.z);
    const float el_deg = g.fast ? rad2deg_fast(s.el) : rad2deg_exact(s.el);
    const float az_deg = g.fast ? rad2deg_fast(s.az) : rad2deg_exact(s.az);
    const float u = 1.0f - div_by_const(el_deg + g.half_v, g.vfov, g.inv_v, g.fast);
    const float w = div_by_const(az_deg + g.half_h, g.hfov, g.inv_h, g.fast);
    const uint32_t rbits = f2u(s.r);
#pragma unroll
    for (int j = 0; j < kMaxScanShapes; ++j) {
        if (j >= sh.n) break;
        const float frows = (float)sh.rows[j], fcols = (float)sh.cols[j];
        float fr = roundf(frows * u), fc = roundf(fcols * w);
        fr = (fr < 0.0f) ? 0.0f : fr;  fr = (frows - 1.0f < fr) ? frows - 1.0f : fr;
        fc = (fc < 0.0f) ? 0.0f : fc;  fc = (fcols - 1.0f < fc) ? fcols - 1.0f : fc;
        const size_t npx = (size_t)sh.rows[j] * (size_t)sh.cols[j];
        img_min_u32(sh.img[j] + (size_t)blockIdx.y * npx + (size_t)((int)fr * sh.cols[j] + (int)fc), rbits);
    }
}

// smax[kf] = float bits of the largest NON-EMPTY (< 10000) pixel of the finished scan image of keyframe kf: a map point farther than
// smax - thr cannot be flagged in mode 0.  grid = (chunks, keyframes); positive floats order like their bit patterns.
__global__ void __launch_bounds__(kBlock)
k_image_max(const uint32_t* __restrict__ img, uint32_t npx, uint32_t* __restrict__ smax)
{
    __shared__ uint32_t sm[kBlock / 64];
    const uint32_t* __restrict__ imgk = img + (size_t)blockIdx.y * npx;
    uint32_t m = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < npx; i += gridDim.x * blockDim.x) {
        const uint32_t v = imgk[i];
        m = max(m, (v < 0x461c4000u /* 10000.0f = empty */) ? v : 0u);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off, 64));
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / 64; ++w) m = max(m, sm[w]);
        m = max(m, sm[0]);
        if (m) atomicMax(smax + blockIdx.y, m);
    }
}

// qbound[px] for the range-culled vote (mode 0, diff = scan - map > thr): a map point of this pixel can only be flagged if its
// exact range r_e < s - thr (+ float rounding of the subtraction, <= 2e-5 for |diff| < 200).  The kernel tests the squared range
// r2 of its approximate projection, r_e^2 >= r2 (1 - 3e-6) (validated bound), so it may drop the point iff
// r2 >= ((s - thr + 1e-3) (1 + 1e-5))^2, rounded up.  Empty pixels (10000) never flag below 9800 m (and farther points take the
// exact path): 0.  Evaluated in double, rounded up to float.
__global__ void __launch_bounds__(kBlock)
k_scan_qbound(const uint32_t* __restrict__ scan_img, size_t n, float thr, float* __restrict__ qbound)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t sb = scan_img[i];
    float q = 0.0f;
    if (sb < kNoPointBits) {
        const double lim = ((double)u2f(sb) - (double)thr + 1.0e-3) * (1.0 + 1.0e-5);
        if (lim > 0.0) {
            const double q2 = lim * lim;
            q = (float)q2;
            if ((double)q < q2) q = u2f(f2u(q) + 1u);      // round up (q > 0)
        }
    }
    qbound[i] = q;
}
hipError_t scan_qbound(const uint32_t* scan_img, size_t n, float thr, float* qbound, hipStream_t s)
{
    if (!n) return hipSuccess;
    k_scan_qbound<<<dim3(grid_for(n)), dim3(kBlock), 0, s>>>(scan_img, n, thr, qbound);
    return hipGetLastError();
}

hipError_t scan_range_images(const float4* scans, const uint64_t* offsets_dev, size_t kb, size_t nb, uint64_t first_pt,
                             uint64_t n_pts, uint64_t max_kf_pts, Geom g, uint32_t* scan_img, uint32_t* smax_bits, hipStream_t s)
{
    if (n_pts && max_kf_pts)
        for (size_t k0 = 0; k0 < nb; k0 += 65535) {       // gridDim.y limit
            const size_t nk = std::min<size_t>(65535, nb - k0);
            k_scan_rimg<<<dim3(grid_for(max_kf_pts), (unsigned)nk), dim3(kBlock), 0, s>>>(scans, offsets_dev, kb + k0, nk, first_pt, n_pts, g,
                                                                                        scan_img + k0 * (size_t)(g.rows * g.cols));
        }
    if (smax_bits && nb) {
        const uint32_t npx = (uint32_t)(g.rows * g.cols);
        k_image_max<<<dim3(std::min<unsigned>(grid_for(npx, kBlock * 8), 64), (unsigned)nb), dim3(kBlock), 0, s>>>(scan_img, npx, smax_bits);
    }
    return hipGetLastError();
}

hipError_t scan_range_images_multi(const float4* scans, const uint64_t* offsets_dev, size_t kb, size_t nb, uint64_t n_pts, uint64_t max_kf_pts, Geom g,
                                   int n_shapes, const int* rows, const int* cols, uint32_t* const* imgs, uint32_t* const* smax_bits, hipStream_t s)
{
    if (n_shapes < 1 || n_shapes > kMaxScanShapes) return hipErrorInvalidValue;
    ScanShapes sh{};
    sh.n = n_shapes;
    for (int j = 0; j < n_shapes; ++j) { sh.rows[j] = rows[j]; sh.cols[j] = cols[j]; sh.img[j] = imgs[j]; }
    if (n_pts && max_kf_pts)
        for (size_t k0 = 0; k0 < nb; k0 += 65535) {       // gridDim.y limit
            const size_t nk = std::min<size_t>(65535, nb - k0);
            ScanShapes part = sh;
            for (int j = 0; j < n_shapes; ++j) part.img[j] = sh.img[j] + k0 * (size_t)rows[j] * (size_t)cols[j];
            k_scan_rimg_multi<<<dim3(grid_for(max_kf_pts), (unsigned)nk), dim3(kBlock), 0, s>>>(scans, offsets_dev, kb + k0, g, part);
        }
    if (nb)
        for (int j = 0; j < n_shapes; ++j) {
            if (!smax_bits || !smax_bits[j]) continue;
            const uint32_t npx = (uint32_t)(rows[j] * cols[j]);
            k_image_max<<<dim3(std::min<unsigned>(grid_for(npx, kBlock * 8), 64), (unsigned)nb), dim3(kBlock), 0, s>>>(imgs[j], npx, smax_bits[j]);
        }
    return hipGetLastError();
}

// axis-aligned bounds of every 4096-point map tile (map frame): 6 floats per tile {min xyz, max xyz}
__global__ void __launch_bounds__(kBlock)
k_tile_bounds(const float4* __restrict__ map, uint32_t M, float* __restrict__ bounds)
{
    __shared__ float smn[3][kBlock / 64], smx[3][kBlock / 64];
    const uint32_t per_block = (uint32_t)kBlock * 16u;
    const uint32_t base = blockIdx.x * per_block;
    const uint32_t nloc = min(per_block, M - base);
    float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (uint32_t li = threadIdx.x; li < nloc; li += kBlock) {
        const float4 p = map[base + li];
        mn[0] = fminf(mn[0], p.x); mn[1] = fminf(mn[1], p.y); mn[2] = fminf(mn[2], p.z);
        mx[0] = fmaxf(mx[0], p.x); mx[1] = fmaxf(mx[1], p.y); mx[2] = fmaxf(mx[2], p.z);
    }
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { mn[d] = fminf(mn[d], __shfl_xor(mn[d], off, 64)); mx[d] = fmaxf(mx[d], __shfl_xor(mx[d], off, 64)); }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int d = 0; d < 3; ++d) { smn[d][wave] = mn[d]; smx[d][wave] = mx[d]; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int d = threadIdx.x;
        float a = smn[d][0], b = smx[d][0];
        for (int w = 1; w < kBlock / 64; ++w) { a = fminf(a, smn[d][w]); b = fmaxf(b, smx[d][w]); }
        bounds[6 * (size_t)blockIdx.x + d] = a; bounds[6 * (size_t)blockIdx.x + 3 + d] = b;
    }
}
// bounds of the four 1024-point quarters of every 4096-point tile (24 floats per tile): the occlusion cull's second look (k_pair_shell_select).
// One wavefront per quarter; a quarter beyond the end of the map gets an inverted box (never live).
__global__ void __launch_bounds__(kBlock)
k_subtile_bounds(const float4* __restrict__ map, uint32_t M, float* __restrict__ bounds)
{
    const uint32_t sub = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    const uint32_t base = sub * 1024u;
    float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (uint32_t li = lane; li < 1024u; li += 64u) {
        if (base + li >= M) break;
        const float4 p = map[base + li];
        mn[0] = fminf(mn[0], p.x); mn[1] = fminf(mn[1], p.y); mn[2] = fminf(mn[2], p.z);
        mx[0] = fmaxf(mx[0], p.x); mx[1] = fmaxf(mx[1], p.y); mx[2] = fmaxf(mx[2], p.z);
    }
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { mn[d] = fminf(mn[d], __shfl_xor(mn[d], off, 64)); mx[d] = fmaxf(mx[d], __shfl_xor(mx[d], off, 64)); }
    if (lane < 3u) { bounds[6 * (size_t)sub + lane] = mn[lane]; bounds[6 * (size_t)sub + 3 + lane] = mx[lane]; }
}
hipError_t subtile_bounds(const float4* map, size_t M, float* bounds, hipStream_t s)
{
    if (!M) return hipSuccess;
    k_subtile_bounds<<<dim3((unsigned)((M + 4095) / 4096)), dim3(kBlock), 0, s>>>(map, (uint32_t)M, bounds);
    return hipGetLastError();
}
hipError_t tile_bounds(const float4* map, size_t M, float* bounds, hipStream_t s)
{
    if (!M) return hipSuccess;
    const size_t per_block = (size_t)kBlock * 16;
    k_tile_bounds<<<dim3((unsigned)((M + per_block - 1) / per_block)), dim3(kBlock), 0, s>>>(map, (uint32_t)M, bounds);
    return hipGetLastError();
}

// utility.cpp:64-72 + :92-142.  grid = (map tiles, keyframes of the batch)
template <bool B2L_IDENTITY>
__global__ void __launch_bounds__(kBlock)
k_map_rimg(const float4* __restrict__ map, size_t M, const double* __restrict__ inv_poses, size_t kb,
           HostMat34 b2l_h, Geom gg, uint64_t* __restrict__ img)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const size_t kf = kb + blockIdx.y;
    const Mat34 Tinv = load_mat(inv_poses + 12 * kf);
    const RimgGeom g = make_geom(gg);
    const float4 p4 = map[i];
    float3 p = xform(Tinv, make_float3(p4.x, p4.y, p4.z));
    if (B2L_IDENTITY) p = xform_identity(p); else p = xform(to_dev(b2l_h), p);
    const Sph s = cart2sph(p.x, p.y, p.z);
    const int px = pixel_index(g, s.az, s.el);
    const uint64_t v = ((uint64_t)f2u(s.r) << 32) | (uint64_t)(uint32_t)i;
    img_min_u64(img + (size_t)blockIdx.y * (size_t)(g.rows * g.cols) + px, v);
}

// Same contract as k_map_rimg, with a per-workgroup LDS pre-reduction.  Map points are stored in octree (Morton)
// order, so the 4096 consecutive points of a workgroup cover a compact patch of the range image and many of
// them share a pixel: they are first min-reduced in a 2048-slot direct-mapped LDS table (slot = low bits of
// row/col, 64-bit ds_min), and only one global atomic per touched pixel leaves the CU.  A point whose slot
// is owned by another pixel goes straight to the global image.  uint64 min is associative and commutative,
// so the final image is identical to the serial reference result.
static constexpr int kLdsSlots = 2048;
static constexpr int kPtsPerThread = 16;
static constexpr uint32_t kEmptyTag = 0xffffffffu;

// XCD-aware workgroup -> (map tile, keyframe) mapping.  Workgroup b runs on XCD b % 8 (observed dispatch rule; used for speed
// only, any other placement is still correct).  Consecutive workgroups of one XCD take the SAME map tile for `kfg`
// consecutive keyframes, so the tile (64 KB) is fetched from HBM / Infinity Cache once and served from that XCD's L2
// for the other kfg-1 keyframes: with ~224 resident workgroups per XCD the live tile set is ~28 x 64 KB << 4 MiB of L2.
static constexpr unsigned kKfPerTile = 8;      // keyframes that reuse one map tile on an XCD (4 / 16 measured no better in round 2)
struct TileKf { uint32_t tile, kfb; bool valid; };
__device__ __forceinline__ TileKf tile_kf_of_block(uint32_t b, uint32_t n_tiles, uint32_t nb, uint32_t kfg)
{
    const uint32_t x = b & 7u, r = b >> 3;
    const uint32_t n_tg = (n_tiles + 7u) >> 3;
    const uint32_t kfl = r % kfg, q = r / kfg;
    const uint32_t tg = q % n_tg, kg = q / n_tg;
    TileKf t;
    // the divisions above run on the vector unit; hand the (uniform) results back to scalar registers so that every address derived
    // from them is a scalar base instead of a per-use v_readfirstlane
    t.tile = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tg * 8u + x));
    t.kfb = (uint32_t)__builtin_amdgcn_readfirstlane((int)(kg * kfg + kfl));
    t.valid = (t.tile < n_tiles) & (t.kfb < nb);
    return t;
}
static inline unsigned tile_kf_grid(size_t n_tiles, size_t nb, unsigned kfg)
{
    const size_t n_tg = (n_tiles + 7) / 8, n_kg = (nb + kfg - 1) / kfg;
    return (unsigned)(n_tg * 8 * kfg * n_kg);
}

template <bool B2L_IDENTITY>
__global__ void __launch_bounds__(kBlock)
k_map_rimg_lds(const float4* __restrict__ map, uint32_t M, const double* __restrict__ inv_poses, uint32_t kb, uint32_t nb, uint32_t kfg,
               HostMat34 b2l_h, Geom gg, uint64_t* __restrict__ img)
{
    __shared__ uint64_t vals[kLdsSlots];
    __shared__ uint32_t tags[kLdsSlots];
    const uint32_t per_block = (uint32_t)(kBlock * kPtsPerThread);
    const TileKf tk = tile_kf_of_block(blockIdx.x, (M + per_block - 1) / per_block, nb, kfg);
    if (!tk.valid) return;
    for (int s = threadIdx.x; s < kLdsSlots; s += kBlock) { tags[s] = kEmptyTag; vals[s] = ~0ull; }
    __syncthreads();
    const RimgGeom g = make_geom(gg);
    const uint32_t npx = (uint32_t)(g.rows * g.cols);
    const uint32_t block_base = tk.tile * per_block;
    const float4* __restrict__ mapb = map + block_base;
    const uint32_t nloc = min(per_block, M - block_base);
    const Mat34 Tinv = load_mat(inv_poses + 12 * (size_t)(kb + tk.kfb));
    uint64_t* __restrict__ imgk = img + (size_t)tk.kfb * npx;
#pragma unroll 2
    for (uint32_t li = threadIdx.x; li < nloc; li += kBlock) {
        const float4 p4 = mapb[li];
        float3 p = xform(Tinv, make_float3(p4.x, p4.y, p4.z));
        if (B2L_IDENTITY) p = xform_identity(p); else p = xform(to_dev(b2l_h), p);
        const Sph s = cart2sph(p.x, p.y, p.z);
        int row, col;
        pixel_row_col(g, s.az, s.el, row, col);
        const uint32_t px = (uint32_t)(row * g.cols + col);
        const uint64_t v = ((uint64_t)f2u(s.r) << 32) | (uint64_t)(block_base + li);
        const int slot = ((row & 15) << 7) | (col & 127);
        uint32_t t = tags[slot];
        if (t == kEmptyTag) {
            const uint32_t old = atomicCAS(&tags[slot], kEmptyTag, px);
            t = (old == kEmptyTag) ? px : old;
        }
        if (t == px) { if (v < vals[slot]) atomicMin(reinterpret_cast<unsigned long long*>(&vals[slot]), (unsigned long long)v); }   // vals only decreases: skip hopeless same-address atomics
        else img_min_u64(imgk + px, v);
    }
    __syncthreads();
    for (int s = threadIdx.x; s < kLdsSlots; s += kBlock) {
        const uint32_t t = tags[s];
        if (t != kEmptyTag) img_min_u64(imgk + t, vals[s]);
    }
}

hipError_t map_range_images(const float4* map, size_t M, const double* inv_poses_dev, const float* approx_poses_dev, size_t kb, size_t nb,
                            HostMat34 b2l, int b2l_identity, Geom g, uint64_t* map_img, hipStream_t s, const KernelOpts& ko);

// ---------------------------------------------------------------------------------------------------------------
// Range-culled vote kernel (mode 0: diff = scan - map).  A map point P can influence the labels only if it could be
// flagged in its own pixel, i.e. fl(scan[px(P)] - r_P) > thr: if it cannot, then (a) it is not flagged itself, and
// (b) removing it from the arg-min competition changes nothing -- whoever wins instead is at least as far, so that
// pixel stays unflagged; and any flagged winner is nearer than P, so P never displaces it.  The scan images are
// complete before this kernel runs, so the test needs no inter-workgroup ordering.
// Phase 1 (every point): exact fp64 transform, then a bounded-error projection (atan2 within 3e-6 rad, native
// sqrt) gives the pixel up to +-Geom::cull_eps_px; the point survives if for ANY candidate pixel the scan range exceeds a
// lower bound of its range by more than thr minus a margin (or if it is in a domain the fast forms do not cover).
// Survivors (typically 10-20 %) are queued in LDS.  Phase 2: survivors get the exact arithmetic and the same LDS
// pre-reduction as k_map_rimg_lds.  Labels are identical to the un-culled path (parity tests); the map image is
// not (culled points are absent), which is why ltm_debug_range_image / reprojection / mode 1 use k_map_rimg_lds.

// rb/cb: the pixel if it is certain; multi: within cull_eps_px of a rounding boundary (candidates r0..r1 x c0..c1, filled by
// cull_expand); r2: squared range of the approximate local point -- the exact range r_e satisfies r_e^2 in r2 * [1 - 3e-6, 1 + 3e-6]
// (validated on the device by ltm_debug_cull_check); unusual: outside the fast forms' domain
struct CullCand { int rb, cb, r0, r1, c0, c1; float rowh, colh, r2; bool multi, unusual; };

// p' = A (p - c): the inverse pose (composed with base->lidar) rewritten around the sensor position c so that the
// subtraction happens between nearby numbers; c is carried as a float-float pair (c_hi, c_lo) and the tiny constant
// -A c_lo is precomputed per keyframe (ap[12..14]) and enters through the first FMA of each row.  Relative error of p'
// <= 5e-7 (binary32 roundings only), i.e. <= 5e-7 rad of direction error at any range.  16 floats per keyframe.
__device__ __forceinline__ float3 xform_approx(const float* __restrict__ ap, float4 p4, bool& ok)
{
    const float dx = p4.x - ap[9], dy = p4.y - ap[10], dz = p4.z - ap[11];
    ok = ap[15] != 0.0f;
    float3 o;
    o.x = __builtin_fmaf(ap[2], dz, __builtin_fmaf(ap[1], dy, __builtin_fmaf(ap[0], dx, ap[12])));
    o.y = __builtin_fmaf(ap[5], dz, __builtin_fmaf(ap[4], dy, __builtin_fmaf(ap[3], dx, ap[13])));
    o.z = __builtin_fmaf(ap[8], dz, __builtin_fmaf(ap[7], dy, __builtin_fmaf(ap[6], dx, ap[14])));
    return o;
}

// steep_clamps: the field of view is narrow enough (vfov/2 < 44 deg) that every elevation beyond +-45 deg clamps into the first /
// last row whatever its value, so the elevation polynomial only ever sees |z| / rxy <= 1 (one v_rsq instead of v_sqrt + v_rcp and
// no octant select); with a wider vertical field of view the steep points take the exact path instead.
// EL3: the field of view is narrow enough for the fitted degree-3 elevation polynomial (Geom::el_fit, two FMAs fewer and ~2x more
// accurate than the generic one on [0, 1]); then elevations beyond the clamp need no special case at all.
// (Tried: packed v_pk_* evaluation of the two polynomials -- no gain on gfx950, where v_pk_fma_f32 issues at half the rate of
// v_fma_f32, tools/ubench/valu_rate.hip.)
template <bool EL3 = false>
__device__ __forceinline__ CullCand cull_candidates(const RimgGeom& g, float3 p, float row_scale, float col_scale, bool steep_clamps)
{
    CullCand cc;
    const float xy2 = __builtin_fmaf(p.x, p.x, p.y * p.y);
    cc.r2 = __builtin_fmaf(p.z, p.z, xy2);
    const float inv_rxy = __builtin_amdgcn_rsqf(xy2);
    const float t_el = fabsf(p.z) * inv_rxy;                                  // tan |elevation|
    // azimuth, reduced to the first octant: min(|x|, |y|) / rxy is its sine (no reciprocal of max(|x|, |y|) needed)
    const float ax = fabsf(p.x), ay = fabsf(p.y);
    const float az_oct = asin_octant_approx(fminf(ax, ay) * inv_rxy);
    float el_abs;
    if (EL3) {
        const float t = fminf(t_el, g.el_tclamp), u = t * t;
        el_abs = t * __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(g.el_c3, u, g.el_c2), u, g.el_c1), u, g.el_c0);
    } else {
        el_abs = atan_unit_approx(fminf(t_el, 1.0f));
    }
    const float el = __builtin_copysignf(el_abs, p.z);
    float az = (ay > ax) ? (1.57079632679f - az_oct) : az_oct;
    az = (p.x < 0.0f) ? (3.14159265359f - az) : az;
    az = __builtin_copysignf(az, p.y);          // az >= 0: one v_bfi instead of compare + select
    // rowf = R*(1 - (el_deg + V/2)/V) = R/2 - el*(R*180/(pi*V)) ; colf = C*((az_deg + H/2)/H) = C/2 + az*(C*180/(pi*H)).
    // rowh = rowf + 0.5 - eps: floor(rowf + 0.5) is the rounded pixel, and with the band half-width eps taken off up front
    //   fract(rowh) < 1 - 2 eps  <=>  fract(rowf + 0.5) in [eps, 1 - eps)  <=>  the pixel is certain, and then floor(rowh) is that pixel
    // (one fract and one compare per axis instead of a two-sided test; the sign of el sits on the uniform factor).
    cc.rowh = __builtin_fmaf(el, -row_scale, 0.5f * g.frows + 0.5f - g.eps);
    cc.colh = __builtin_fmaf(az, col_scale, 0.5f * g.fcols + 0.5f - g.eps);
    const bool certain = fmaxf(__builtin_amdgcn_fractf(cc.rowh), __builtin_amdgcn_fractf(cc.colh)) < 1.0f - 2.0f * g.eps;
    // Outside the fast forms' domain (every such case ends in the exact path):
    //  - the +-180 deg seam -- the SIGN of y picks column 0 or C-1 there, and the approximate y is only good to ~1e-7 of the range,
    //    so x < 0 with |y| <= 1e-6 |x| is undecidable here (y == +-0 included); the same test catches vanishing x and y (the squares
    //    would underflow and 0 * rsq(0) makes a NaN azimuth);
    //  - r >= 8000 m (inf included): the reference's "empty pixel = 10000 m" sentinel arithmetic (diff = 10000 - r,
    //    Removerter.cpp:398-404) flags a map point 9800..9999.9 m from the sensor on an EMPTY scan pixel, which the fast test (empty
    //    pixels never flag) would drop;
    //  - steep elevations when the field of view does not clamp them.
    // A NaN coordinate needs no guard: its range is NaN, `r < rimg` is false in the reference (utility.cpp:134), so the point never
    // wins a pixel -- and here r2 = NaN fails both compares below and the caller's r2 < qbound, so it is dropped, which is the same.
    cc.unusual = (fabsf(p.y) <= __builtin_fmaf(-1.0e-6f, p.x, 1.0e-18f)) | (cc.r2 > 6.4e7f);
    if (!EL3) cc.unusual |= !steep_clamps & (t_el > 1.0f);
    cc.multi = !certain;
    // clamp(floor(v), 0, n-1) == trunc(med3(v, 0, n-1)): the bounds are integers and the clamped value is non-negative
    cc.rb = (int)__builtin_amdgcn_fmed3f(cc.rowh, 0.0f, g.frows - 1.0f);
    cc.cb = (int)__builtin_amdgcn_fmed3f(cc.colh, 0.0f, g.fcols - 1.0f);
    cc.r0 = cc.r1 = cc.rb; cc.c0 = cc.c1 = cc.cb;
    return cc;
}

// lower bound of the exact range from the squared approximate range (native sqrt, 1 ulp): upper bound = r_lo * (1 + 3e-6)
__device__ __forceinline__ float cull_r_lo(float r2) { return __builtin_amdgcn_sqrtf(r2) * (1.0f - 1.5e-6f); }

// candidate pixel rectangle of a point that sits within cull_eps_px of a rounding boundary (rare)
__device__ __forceinline__ void cull_expand(const RimgGeom& g, CullCand& cc)
{
    // rowh / colh carry the -eps shift of cull_candidates: a fraction >= 1 - 2 eps means the unshifted value is within eps of the
    // integer above floor(rowh) -- either pixel floor(rowh) or floor(rowh) + 1
    const float rfl = floorf(cc.rowh), cfl = floorf(cc.colh);
    const float rfr = cc.rowh - rfl, cfr = cc.colh - cfl;
    const int rc = (int)rfl, ccn = (int)cfl;
    const int rmax = g.rows - 1, cmax = g.cols - 1;
    const float lim = 1.0f - 2.0f * g.eps;
    cc.r0 = min(max(rc, 0), rmax);
    cc.r1 = min(max(rc + (rfr >= lim ? 1 : 0), 0), rmax);
    cc.c0 = min(max(ccn, 0), cmax);
    cc.c1 = min(max(ccn + (cfr >= lim ? 1 : 0), 0), cmax);
}

// With a non-identity base->lidar extrinsic the exact path rounds to float between the two transforms (utility.cpp:70-71), i.e.
// at magnitude range + lever arm; relative to a range much smaller than the lever arm that rounding exceeds the validated bounds
// of the approximate projection, so points nearer than lever/8 take the exact path (none with an identity extrinsic).
template <bool B2L_IDENTITY>
__device__ __forceinline__ float cull_min_range(const HostMat34& b2l)
{
    if (B2L_IDENTITY) return 0.0f;
    const float tx = (float)b2l.m[3], ty = (float)b2l.m[7], tz = (float)b2l.m[11];
    return 0.125f * __builtin_sqrtf(tx * tx + ty * ty + tz * tz) + 1.0e-6f;
}

__device__ unsigned long long g_cull_stats[4];   // {survivors, points} of k_vote_map_cull, then of k_map_rimg_blockmin: diagnostic, read by cull_stats()

static constexpr int kCullQueue = 2048;   // survivor queue capacity (~370 of 4096 expected); overflow sends the whole tile down the exact path

// Direct-mapped pixel table of a workgroup: tags[slot] = the pixel that owns the slot (first come), slot = low bits of row / column.
// Returns the slot if `px` owns it, or -1 (then the caller goes to the global image).  A slot never changes owner.
// (Tried: a second chance in the (ROWS/2) x (2 COLS) cut of the coordinates.  It turns ~7 % of misses into ~2 % on octree-ordered
// map tiles, but the extra probe costs as much as the saved global atomics: no change in either kernel.)
template <int ROWS, int COLS>
__device__ __forceinline__ int table_claim(uint32_t* __restrict__ tags, int row, int col, uint32_t px)
{
    const int slot = ((row & (ROWS - 1)) * COLS) | (col & (COLS - 1));
    uint32_t t = tags[slot];
    if (t == kEmptyTag) {
        const uint32_t old = atomicCAS(&tags[slot], kEmptyTag, px);
        t = (old == kEmptyTag) ? px : old;
    }
    return (t == px) ? slot : -1;
}

// exact projection of one map point into image `imgk`, reduced through the workgroup's LDS table
template <bool B2L_IDENTITY, int SLOTS_R, int SLOTS_C>
__device__ __forceinline__ void exact_insert(const float4* __restrict__ map, uint32_t i, const Mat34& Tinv, const HostMat34& b2l_h, const RimgGeom& g,
                                             uint64_t* __restrict__ vals, uint32_t* __restrict__ tags, uint64_t* __restrict__ imgk)
{
    const float4 p4 = map[i];
    float3 p = xform(Tinv, make_float3(p4.x, p4.y, p4.z));
    if (B2L_IDENTITY) p = xform_identity(p); else p = xform(to_dev(b2l_h), p);
    const Sph s = cart2sph(p.x, p.y, p.z);
    int row, col;
    pixel_row_col(g, s.az, s.el, row, col);
    const uint32_t px = (uint32_t)(row * g.cols + col);
    const uint64_t v = ((uint64_t)f2u(s.r) << 32) | (uint64_t)i;
    const int slot = table_claim<SLOTS_R, SLOTS_C>(tags, row, col, px);
    if (slot >= 0) { if (v < vals[slot]) atomicMin(reinterpret_cast<unsigned long long*>(&vals[slot]), (unsigned long long)v); }   // vals only decreases: skip hopeless same-address atomics
    else img_min_u64(imgk + px, v);
}

// Exact range of a map point whose pixel is already certain (single candidate of the bounded-error projection, which
// ltm_debug_cull_check validates to contain the exact pixel): the reference arithmetic up to sqrtf only --
// r = sqrtf((x*x + y*y) + z*z) on the float-rounded fp64 transform -- without the atan2f / rad2deg / pixel chain.
template <bool B2L_IDENTITY>
__device__ __forceinline__ uint32_t exact_range_bits(const float4 p4, const Mat34& Tinv, const HostMat34& b2l_h)
{
    float3 p = xform(Tinv, make_float3(p4.x, p4.y, p4.z));
    if (B2L_IDENTITY) p = xform_identity(p); else p = xform(to_dev(b2l_h), p);
    const float xy = p.x * p.x + p.y * p.y;
    return f2u(__builtin_sqrtf(xy + p.z * p.z));
}

// Whole-tile range cull: no point of a tile can be nearer to the sensor than the distance from the sensor position to the tile's
// bounding box; if that already exceeds the keyframe's longest scan return (minus thr, with margins for a slightly
// non-orthonormal pose and float rounding) nothing there can be flagged and the workgroup is done.
// ap = the keyframe's approximate pose (16 floats), tb = {min xyz, max xyz} of the tile, smax = longest non-empty scan return.
__device__ __forceinline__ bool tile_out_of_reach(const float* __restrict__ ap, const float* __restrict__ tb, float smax, float thr)
{
    float d2 = 0.0f, far2 = 0.0f;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float c = ap[9 + d];                            // sensor position to ~1e-5 m: the margins below are 1e-2
        const float lo = tb[d] - c, hi = c - tb[3 + d];
        const float e = fmaxf(fmaxf(lo, hi), 0.0f);          // distance to the box along this axis
        const float f = fmaxf(fabsf(lo), fabsf(hi));          // distance to its farthest face
        d2 = __builtin_fmaf(e, e, d2);
        far2 = __builtin_fmaf(f, f, far2);
    }
    // exact local range of any point of the tile: |A (p - c)| >= smin * |p - c| >= smin * dist(c, box); ap[15] = lower bound of smin
    const float smin = ap[15];
    const float reach = fmaxf(smax - thr, 0.0f) + 1.0e-2f + smax * 1.0e-3f;
    // far2 guard: beyond ~8.9 km the reference's "empty pixel = 10000" sentinel arithmetic could flag a point; never cull there
    return smin > 0.5f && far2 < 8.0e7f && d2 * smin * smin * 0.996f > reach * reach;
}

// number of (tile, keyframe) workgroups of a k_vote_map_cull launch that survive the whole-tile cull (measurement only: the
// algorithmic bytes of a launch count the map tiles that are actually read)
__global__ void __launch_bounds__(kBlock)
k_count_live_tiles(const float* __restrict__ approx_poses, uint32_t kb, uint32_t nb, const float* __restrict__ tile_bounds, uint32_t n_tiles,
                   const uint32_t* __restrict__ smax_bits, float thr, unsigned long long* __restrict__ live)
{
    const uint32_t tile = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t kfb = blockIdx.y;
    bool alive = false;
    if (tile < n_tiles) alive = !tile_out_of_reach(approx_poses + 16 * (size_t)(kb + kfb), tile_bounds + 6 * (size_t)tile, u2f(smax_bits[kfb]), thr);
    __shared__ uint32_t wsum[kBlock / 64];
    const uint64_t b = __builtin_amdgcn_ballot_w64(alive);
    if ((threadIdx.x & 63u) == 0u) wsum[threadIdx.x >> 6] = (uint32_t)__popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) {         // one atomic per workgroup
        uint32_t t = 0;
        for (int w = 0; w < kBlock / 64; ++w) t += wsum[w];
        if (t) atomicAdd(live, (unsigned long long)t);
    }
}
hipError_t count_live_tiles(const float* approx_poses_dev, size_t kb, size_t nb, const float* tile_bounds_dev, size_t n_tiles,
                            const uint32_t* smax_bits_dev, float thr, unsigned long long* live_dev, hipStream_t s)
{
    if (!nb || !n_tiles) return hipSuccess;
    k_count_live_tiles<<<dim3(grid_for(n_tiles), (unsigned)nb), dim3(kBlock), 0, s>>>(approx_poses_dev, (uint32_t)kb, (uint32_t)nb, tile_bounds_dev,
                                                                                    (uint32_t)n_tiles, smax_bits_dev, thr, live_dev);
    return hipGetLastError();
}

static constexpr int kCullSlots = 512;   // survivors are ~10 % of a workgroup's points: a small LDS table keeps 8 workgroups per CU

template <bool B2L_IDENTITY, bool EL3>     // EL3: fitted elevation polynomial (Geom::el_fit), see cull_candidates
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(8, 8)))
k_vote_map_cull(const float4* __restrict__ map, uint32_t M, const double* __restrict__ inv_poses, const float* __restrict__ approx_poses,
                uint32_t kb, uint32_t nb, uint32_t kfg, HostMat34 b2l_h, Geom gg, const float* __restrict__ qbound_img,
                const float* __restrict__ tile_bounds, const uint32_t* __restrict__ smax_bits, float thr, uint64_t* __restrict__ img)
{
    __shared__ uint64_t vals[kCullSlots];
    __shared__ uint32_t tags[kCullSlots];
    // survivors of phase 1, one word each: tile-local index (12 bits) | row (9) | column (11); row field 511 = pixel not certain,
    // needs the full exact projection (images with >= 511 rows or > 2048 columns mark every survivor that way)
    __shared__ uint32_t queue[kCullQueue];
    __shared__ uint16_t uqueue[kCullQueue];    // the uncertain ones, re-queued densely in phase 2
    __shared__ uint32_t qcount, ucount;
    const uint32_t per_block = (uint32_t)(kBlock * kPtsPerThread);
    const TileKf tk = tile_kf_of_block(blockIdx.x, (M + per_block - 1) / per_block, nb, kfg);
    if (!tk.valid) return;
    if (tile_bounds && tile_out_of_reach(approx_poses + 16 * (size_t)(kb + tk.kfb), tile_bounds + 6 * (size_t)tk.tile, u2f(smax_bits[tk.kfb]), thr)) return;
    for (int s = threadIdx.x; s < kCullSlots; s += kBlock) { tags[s] = kEmptyTag; vals[s] = ~0ull; }
    if (threadIdx.x == 0) { qcount = 0; ucount = 0; }
    __syncthreads();
    const RimgGeom g = make_geom(gg);
    const uint32_t npx = (uint32_t)(g.rows * g.cols);
    const uint32_t block_base = tk.tile * per_block;
    const float4* __restrict__ mapb = map + block_base;
    const uint32_t nloc = min(per_block, M - block_base);
    const uint32_t kf = kb + tk.kfb;
    uint64_t* __restrict__ imgk = img + (size_t)tk.kfb * npx;
    const float* __restrict__ qk = qbound_img + (size_t)tk.kfb * npx;
    // ---- phase 1: who can matter?  (bounded-error arithmetic only).  Four points per lane are in flight at once so the
    // dependent scan-image load of one overlaps the arithmetic of the others.  Only full tiles: the one partial tile at the end
    // of the map takes the exact path as a whole (below), which keeps bounds tests and clamped indices out of this loop.
    const bool full_tile = nloc == per_block;
    if (full_tile) {
        const float* __restrict__ ap = approx_poses + 16 * (size_t)kf;
        const float row_scale = g.frows * (57.29577951308232f / g.vfov), col_scale = g.fcols * (57.29577951308232f / g.hfov);
        const float rmin = cull_min_range<B2L_IDENTITY>(b2l_h), rmin2 = rmin * rmin;
        const bool steep_clamps = g.vfov < 88.0f;
        const bool ok_img = g.rows < 511 && g.cols <= 2048;      // the queue word holds 9 row bits and 11 column bits
        constexpr int kInFlight = 4;
        constexpr bool kPrefetch = true;     // software pipelining: the next group's points are requested before this group's arithmetic
        float4 nxt[kInFlight];
        if (kPrefetch) {
#pragma unroll
            for (int u = 0; u < kInFlight; ++u) nxt[u] = mapb[(uint32_t)u * kBlock + threadIdx.x];
        }
        const uint32_t lane_word = threadIdx.x << 20;            // queue word: tile-local index (12 bits) | row (9) | column (11)
#pragma unroll
        for (uint32_t j0 = 0; j0 < (uint32_t)kPtsPerThread; j0 += kInFlight) {
            float4 pt[kInFlight];
            CullCand cc[kInFlight];
            float q0[kInFlight];
#pragma unroll
            for (int u = 0; u < kInFlight; ++u) {
                const uint32_t li = (j0 + u) * kBlock + threadIdx.x;
                if (kPrefetch) {
                    pt[u] = nxt[u];
                    if (j0 + kInFlight < (uint32_t)kPtsPerThread) nxt[u] = mapb[li + kInFlight * kBlock];
                } else {
                    pt[u] = mapb[li];
                }
            }
#pragma unroll
            for (int u = 0; u < kInFlight; ++u) {
                bool ok;
                const float3 p = xform_approx(ap, pt[u], ok);
                cc[u] = cull_candidates<EL3>(g, p, row_scale, col_scale, steep_clamps);
                // not certain of the pixel (within cull_eps_px of a rounding boundary, ~1 % of the points): straight to the exact path
                cc[u].unusual |= !ok_img | !ok | cc[u].multi | (B2L_IDENTITY ? false : (cc[u].r2 < rmin2));
                q0[u] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(qk) + ((__umul24((uint32_t)cc[u].rb, (uint32_t)g.cols) + (uint32_t)cc[u].cb) << 2));   // uniform base + 32-bit offset
            }
            bool mt[kInFlight];
#pragma unroll
            for (int u = 0; u < kInFlight; ++u) {
                // qbound[px] = an upper bound of the SQUARED range below which a point of that pixel could be flagged (k_scan_qbound):
                // one compare decides; empty pixels hold 0
                mt[u] = cc[u].unusual | (cc[u].r2 < q0[u]);
            }
            // One LDS atomic per wave for the four points of every lane (hand-rolled ballot/mbcnt aggregation: letting the compiler
            // aggregate four separate atomicAdd(&qcount, 1) costs ~40 instructions each and one of them is taken almost always).
            const uint64_t b0 = __builtin_amdgcn_ballot_w64(mt[0]), b1 = __builtin_amdgcn_ballot_w64(mt[1]),
                           b2 = __builtin_amdgcn_ballot_w64(mt[2]), b3 = __builtin_amdgcn_ballot_w64(mt[3]);
            const uint32_t n0 = (uint32_t)__popcll(b0), n1 = (uint32_t)__popcll(b1), n2 = (uint32_t)__popcll(b2), n3 = (uint32_t)__popcll(b3);
            const uint32_t total = n0 + n1 + n2 + n3;
            if (total) {
                uint32_t base = 0;
                if ((threadIdx.x & 63u) == 0u) base = atomicAdd(&qcount, total);
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                // overflow (rare) is decided per wave and group, on the scalar unit: the whole tile takes the exact path below then
                if (base + total <= (uint32_t)kCullQueue) {
                    const uint64_t bal[kInFlight] = {b0, b1, b2, b3};
                    const uint32_t off[kInFlight] = {0u, n0, n0 + n1, n0 + n1 + n2};
#pragma unroll
                    for (int u = 0; u < kInFlight; ++u) {
                        if (!mt[u]) continue;
                        const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal[u] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal[u], 0u));
                        const uint32_t rowq = cc[u].unusual ? 511u : (uint32_t)cc[u].rb;
                        queue[base + off[u] + below] = (((rowq << 11) | (uint32_t)cc[u].cb) | lane_word) | (((j0 + u) * kBlock) << 20);
                    }
                }
            }
        }
    }
    __syncthreads();
    // ---- phase 2: survivors.  Certain pixel: only the exact range is computed; the ~1 % others are re-queued densely at the
    // top of the same array and get the full exact projection afterwards (keeps both loops free of divergence).
    const uint32_t nq_all = full_tile ? qcount : (uint32_t)kCullQueue + 1u;
    if (full_tile && threadIdx.x == 0 && (blockIdx.x & 63u) == 0u) {   // sampled 1/64: same-address atomics from every workgroup would serialise the grid
        atomicAdd(&g_cull_stats[0], (unsigned long long)nq_all);
        atomicAdd(&g_cull_stats[1], (unsigned long long)nloc);
    }
    if (nq_all) {
        const Mat34 Tinv = load_mat(inv_poses + 12 * (size_t)kf);
        if (__builtin_expect(nq_all > (uint32_t)kCullQueue, 0)) {      // queue overflow: a superset is always correct (min is idempotent)
            for (uint32_t li = threadIdx.x; li < nloc; li += kBlock)
                exact_insert<B2L_IDENTITY, 8, 64>(map, block_base + li, Tinv, b2l_h, g, vals, tags, imgk);
        } else {
            for (uint32_t q = threadIdx.x; q < nq_all; q += kBlock) {
                const uint32_t e = queue[q];
                const int row = (int)((e >> 11) & 511u), col = (int)(e & 2047u);
                if (row == 511) { uqueue[atomicAdd(&ucount, 1u)] = (uint16_t)(e >> 20); continue; }
                const uint32_t i = block_base + (e >> 20);
                const uint32_t px = (uint32_t)(row * g.cols + col);
                const uint64_t v = ((uint64_t)exact_range_bits<B2L_IDENTITY>(map[i], Tinv, b2l_h) << 32) | (uint64_t)i;
                const int slot = table_claim<8, 64>(tags, row, col, px);
                if (slot >= 0) { if (v < vals[slot]) atomicMin(reinterpret_cast<unsigned long long*>(&vals[slot]), (unsigned long long)v); }   // vals only decreases: skip hopeless same-address atomics
                else img_min_u64(imgk + px, v);
            }
            __syncthreads();
            const uint32_t nu = ucount;
            for (uint32_t q = threadIdx.x; q < nu; q += kBlock)
                exact_insert<B2L_IDENTITY, 8, 64>(map, block_base + uqueue[q], Tinv, b2l_h, g, vals, tags, imgk);
        }
    }
    __syncthreads();
    if (nq_all)
        for (int s = threadIdx.x; s < kCullSlots; s += kBlock) {
            const uint32_t t = tags[s];
            if (t != kEmptyTag) img_min_u64(imgk + t, vals[s]);
        }
}

hipError_t cull_stats(unsigned long long* out2, int reset, hipStream_t s, int which_kernel)
{
    hipError_t e = hipMemcpyFromSymbolAsync(out2, HIP_SYMBOL(g_cull_stats), 16, 16 * (size_t)(which_kernel ? 1 : 0), hipMemcpyDeviceToHost, s);
    if (e != hipSuccess) return e;
    e = hipStreamSynchronize(s);
    if (e != hipSuccess || !reset) return e;
    const unsigned long long z[4] = {0, 0, 0, 0};
    e = hipMemcpyToSymbolAsync(HIP_SYMBOL(g_cull_stats), z, 32, 0, hipMemcpyHostToDevice, s);
    return e == hipSuccess ? hipStreamSynchronize(s) : e;
}


hipError_t vote_map_range_images(const float4* map, size_t M, const double* inv_poses_dev, const float* approx_poses_dev, size_t kb, size_t nb,
                                 HostMat34 b2l, int b2l_identity, Geom g, const float* qbound_img, const float* tile_bounds_dev,
                                 const uint32_t* smax_bits_dev, float thr, int mode, uint64_t* map_img, hipStream_t s, const KernelOpts& ko)
{
    if (!M || !nb) return hipSuccess;
    if (mode != 0 || !ko.vote_cull || !approx_poses_dev || !qbound_img) return map_range_images(map, M, inv_poses_dev, approx_poses_dev, kb, nb, b2l, b2l_identity, g, map_img, s, ko);
    const size_t per_block = (size_t)kBlock * kPtsPerThread;
    const unsigned kfg = kKfPerTile;
    dim3 grid(tile_kf_grid((M + per_block - 1) / per_block, nb, kfg));
    const float* tb = (ko.tile_cull && smax_bits_dev) ? tile_bounds_dev : nullptr;
    const bool el3 = g.el_fit != 0;
#define LTM_LAUNCH_CULL(ID, E) k_vote_map_cull<ID, E><<<grid, dim3(kBlock), 0, s>>>(map, (uint32_t)M, inv_poses_dev, approx_poses_dev, (uint32_t)kb, (uint32_t)nb, kfg, b2l, g, qbound_img, tb, smax_bits_dev, thr, map_img)
    if (!b2l_identity) { if (el3) LTM_LAUNCH_CULL(false, true); else LTM_LAUNCH_CULL(false, false); }
    else { if (el3) LTM_LAUNCH_CULL(true, true); else LTM_LAUNCH_CULL(true, false); }
#undef LTM_LAUNCH_CULL
    return hipGetLastError();
}

// debug: number of points whose exact pixel is NOT inside the candidate set of the bounded-error projection.
// T (3x4 double) / ap (16 floats) are the exact and the approximate form of the same keyframe transform, or null.
__global__ void __launch_bounds__(kBlock)
k_cull_check(const float* __restrict__ xyz, size_t n, HostMat34 T, HostMat34 b2l, int b2l_identity, const float* __restrict__ ap, Geom gg,
             unsigned long long* __restrict__ bad)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const RimgGeom g = make_geom(gg);
    const float4 p4 = make_float4(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], 0.0f);
    float3 pe = make_float3(p4.x, p4.y, p4.z), pa = pe;
    bool ok = true;
    if (ap) {   // exact side exactly as the vote kernels do it: inverse pose, float store, then base->lidar (utility.cpp:64-72)
        pe = xform(to_dev(T), pe);
        pe = b2l_identity ? xform_identity(pe) : xform(to_dev(b2l), pe);
        pa = xform_approx(ap, p4, ok);
    }
    const float row_scale = g.frows * (57.29577951308232f / g.vfov), col_scale = g.fcols * (57.29577951308232f / g.hfov);
    CullCand cc = g.el_fit ? cull_candidates<true>(g, pa, row_scale, col_scale, g.vfov < 88.0f) : cull_candidates<false>(g, pa, row_scale, col_scale, g.vfov < 88.0f);
    if (cc.unusual || !ok) return;
    const float rmin = cull_min_range<false>(b2l);
    if (ap && !b2l_identity && cc.r2 < rmin * rmin) return;      // these take the exact path in the kernels
    if (cc.multi) cull_expand(g, cc);
    const Sph s = cart2sph(pe.x, pe.y, pe.z);
    int row, col;
    pixel_row_col(g, s.az, s.el, row, col);
    // the vote kernel relies on r_e^2 >= r2 (1 - 3e-6); the exact-image kernel on r_lo <= r_e <= r_lo (1 + 3e-6) with its r_lo
    const double re2 = (double)s.r * (double)s.r;
    const float r_lo = cull_r_lo(cc.r2);
    const bool good = (row == cc.r0 || row == cc.r1) && (col == cc.c0 || col == cc.c1) && (r_lo <= s.r) && (s.r <= r_lo * (1.0f + 3.0e-6f)) &&
                      (re2 >= (double)cc.r2 * (1.0 - 3.0e-6)) && (re2 <= (double)cc.r2 * (1.0 + 3.0e-6));
    if (!good) atomicAdd(bad, 1ull);
}
hipError_t cull_check(const float* xyz_dev, size_t n, const HostMat34* T, const HostMat34* b2l, int b2l_identity, const float* approx_pose_dev,
                      Geom g, unsigned long long* bad_dev, hipStream_t s)
{
    if (!n) return hipSuccess;
    HostMat34 z{};
    k_cull_check<<<dim3(grid_for(n)), dim3(kBlock), 0, s>>>(xyz_dev, n, T ? *T : z, b2l ? *b2l : z, b2l ? b2l_identity : 1,
                                                            T ? approx_pose_dev : nullptr, g, bad_dev);
    return hipGetLastError();
}

// Probe points for the create-on-first-use validation of the bounded-error projection (ltm_api_vote.cpp: cull_geometry_ok): local-frame points that sit
// ON and a hair beside the pixel-rounding boundaries of this image shape -- rows, columns and their crossings -- at ranges from 0.3 m to 200 m, plus
// points in general position; optionally moved into the map frame by `pose` (3x4) so that the approximate inverse transform is exercised too.
__global__ void __launch_bounds__(kBlock)
k_cull_probe_points(Geom g, size_t n, HostMat34 pose, int with_pose, float* __restrict__ xyz)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t h = (uint32_t)i * 0x9e3779b1u + 0x7f4a7c15u;
    auto next = [&]() { h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16; return h; };
    const double jit[8] = {0.0, 1.0e-5, -1.0e-5, 1.0e-4, -1.0e-4, 4.0e-4, -4.0e-4, 2.0e-3};      // pixels off the boundary
    const double ranges[8] = {0.3, 1.0, 3.0, 10.0, 30.0, 60.0, 119.0, 200.0};
    const uint32_t kind = next() & 3u;                     // 0: row boundary, 1: column boundary, 2: both, 3: general position
    double rowf = (double)(next() % (uint32_t)(g.rows * 16 + 1)) / 16.0 - 0.5;
    double colf = (double)(next() % (uint32_t)(g.cols * 16 + 1)) / 16.0 - 0.5;
    if (kind == 0 || kind == 2) rowf = (double)(next() % (uint32_t)(g.rows + 1)) - 0.5 + jit[next() & 7u];
    if (kind == 1 || kind == 2) colf = (double)(next() % (uint32_t)(g.cols + 1)) - 0.5 + jit[next() & 7u];
    // rowf = R (1 - (el + V/2) / V), colf = C (az + H/2) / H  (utility.cpp:122-123)
    const double el = ((double)g.vfov * 0.5 - (double)g.vfov * rowf / (double)g.rows) * (3.14159265358979323846 / 180.0);
    const double az = ((double)g.hfov * colf / (double)g.cols - (double)g.hfov * 0.5) * (3.14159265358979323846 / 180.0);
    const double r = ranges[next() & 7u] * (1.0 + 1.0e-3 * (double)(next() & 1023u));
    double x = r * cos(el) * cos(az), y = r * cos(el) * sin(az), z = r * sin(el);
    if (with_pose) {
        const double X = pose.m[0] * x + pose.m[1] * y + pose.m[2] * z + pose.m[3], Y = pose.m[4] * x + pose.m[5] * y + pose.m[6] * z + pose.m[7],
                     Z = pose.m[8] * x + pose.m[9] * y + pose.m[10] * z + pose.m[11];
        x = X; y = Y; z = Z;
    }
    xyz[3 * i] = (float)x; xyz[3 * i + 1] = (float)y; xyz[3 * i + 2] = (float)z;
}
hipError_t cull_probe_points(Geom g, size_t n, const HostMat34* pose, float* xyz_dev, hipStream_t s)
{
    if (!n) return hipSuccess;
    HostMat34 z{};
    k_cull_probe_points<<<dim3(grid_for(n)), dim3(kBlock), 0, s>>>(g, n, pose ? *pose : z, pose ? 1 : 0, xyz_dev);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// Exact arg-min range image with a workgroup-local pre-filter (reprojection, ND votes, and any caller that needs the
// true image).  Only a point that could be the nearest of its pixel AMONG THE 4096 POINTS OF ITS OWN TILE can be the
// global arg-min, so:
//   phase 1a  bounded-error projection of every point; points whose pixel is certain (one candidate) publish an UPPER
//             bound of their range into a per-pixel LDS min table;
//   phase 1b  a point survives unless its pixel is certain, owns a table slot, and its range LOWER bound exceeds the
//             table's minimum upper bound (then some other point of the tile is strictly nearer in the same exact pixel);
//   phase 2   survivors (a few per pixel) get the exact arithmetic and the usual 64-bit LDS/global min.
// The result is bit-identical to k_map_rimg_lds / the serial reference: discarded points are provably not arg-mins.
static constexpr int kBmSlotsMax = 1024;   // LDS table of the pre-filter: SLOT_ROWS x 64 pixels.  8 rows (8 workgroups per CU instead of 6) were tried: 15.3 ms against 10.3 ms per full-map launch -- the misses of the smaller table cost far more than the occupancy brings
static constexpr int kBmQueue = 2048;     // survivor queue capacity (~11 % of 4096 expected); overflow: the whole tile goes exact
// per-point record / queue entry: tile-local index (12 bits) | row (9) | column (11).  Row field 511 = pixel not certain
// (needs the full exact projection).  Images with >= 511 rows or >= 2048 columns mark every point that way.
static constexpr uint32_t kBmRowUncertain = 511u;
static constexpr int kBmUQueue = 1024;    // dense re-queue of the uncertain survivors (~1 % of the tile); beyond it they are handled in place


// pairs != null: workgroup b processes the (tile, keyframe) pair pairs[b] = tile * nb + keyframe (occlusion-culled launch, see exact_images_occlusion_*)
// (Round 5 tried three restructurings of phase 1 -- a wave-level combine of the lanes of one pixel, a single 64-bit {owner, minimum} table word, and
// a one-pass candidate stream -- all bit-exact, none faster: profiles/r5_ab_blockmin_restructurings.txt)
template <bool B2L_IDENTITY, bool EL3, int SLOT_ROWS = 16>
__global__ void __launch_bounds__(kBlock)
k_map_rimg_blockmin(const float4* __restrict__ map, uint32_t M, const double* __restrict__ inv_poses, const float* __restrict__ approx_poses,
                    uint32_t kb, uint32_t nb, uint32_t kfg, HostMat34 b2l_h, Geom gg, uint64_t* __restrict__ img, const uint32_t* __restrict__ pairs, uint32_t n_pairs,
                    const uint8_t* __restrict__ submask)
{
    constexpr int kBmSlots = SLOT_ROWS * 64;
    static_assert(kBmSlots <= kBmSlotsMax, "");
    __shared__ uint64_t vals[kBmSlots];
    __shared__ uint32_t tags[kBmSlots];
    // the pre-filter's minimum table is only alive in phase 1 and the 64-bit (range | index) table only in phase 2: same LDS
    // (22.6 KB instead of 26.6 KB per workgroup: 7 workgroups per CU instead of 6)
    uint32_t* const amin = reinterpret_cast<uint32_t*>(vals);
    __shared__ uint32_t queue[kBmQueue];
    __shared__ uint16_t uqueue[kBmUQueue];
    __shared__ uint32_t qcount, ucount;
    const uint32_t per_block = (uint32_t)(kBlock * kPtsPerThread);
    TileKf tk;
    uint32_t quarters = 0xfu;
    if (pairs) {
        // the list is sorted by (tile, keyframe); workgroup b runs on XCD b % 8 (see tile_kf_of_block), so XCD x walks the x-th eighth of
        // the list front to back: consecutive workgroups of one XCD share a tile and it is served from that XCD's L2, as in the plain launch
        const uint32_t seg = (n_pairs + 7u) >> 3, at = (blockIdx.x & 7u) * seg + (blockIdx.x >> 3);
        if ((blockIdx.x >> 3) >= seg || at >= n_pairs) return;
        const uint32_t pr = pairs[at];
        tk.tile = (uint32_t)__builtin_amdgcn_readfirstlane((int)(pr / nb)); tk.kfb = (uint32_t)__builtin_amdgcn_readfirstlane((int)(pr % nb)); tk.valid = true;
        if (submask) quarters = (uint32_t)__builtin_amdgcn_readfirstlane((int)submask[pr]);      // quarters of the tile the occlusion cull left alive
    } else {
        tk = tile_kf_of_block(blockIdx.x, (M + per_block - 1) / per_block, nb, kfg);
    }
    if (!tk.valid) return;
    for (int s = threadIdx.x; s < kBmSlots; s += kBlock) { tags[s] = kEmptyTag; amin[s] = 0x7f800000u; }
    if (threadIdx.x == 0) { qcount = 0; ucount = 0; }
    __syncthreads();
    const RimgGeom g = make_geom(gg);
    const uint32_t npx = (uint32_t)(g.rows * g.cols);
    const uint32_t block_base = tk.tile * per_block;
    const float4* __restrict__ mapb = map + block_base;
    const uint32_t nloc = min(per_block, M - block_base);
    const uint32_t kf = kb + tk.kfb;
    uint64_t* __restrict__ imgk = img + (size_t)tk.kfb * npx;
    const float* __restrict__ ap = approx_poses + 16 * (size_t)kf;
    const float row_scale = g.frows * (57.29577951308232f / g.vfov), col_scale = g.fcols * (57.29577951308232f / g.hfov);
    const bool packable = g.rows < (int)kBmRowUncertain && g.cols <= 2048;
    const float rmin = cull_min_range<B2L_IDENTITY>(b2l_h), rmin2 = rmin * rmin;
    const bool steep_clamps = g.vfov < 88.0f;
    // per-lane record of the 16 points: amin slot (bits 20+) | row | col, and the range lower bound -- of the points that own an
    // amin slot; -1 for the others, which no table entry can beat (their slot field is 0: any valid index)
    float rlo[kPtsPerThread];
    uint32_t rec[kPtsPerThread];
    // Only full tiles run the pre-filter: the one partial tile at the end of the map takes the exact path as a whole (phase 2),
    // which keeps bounds tests and clamped indices out of these loops.
    const bool full_tile = nloc == per_block;
    if (full_tile) {
        // ---- phase 1a (four points per lane in flight, as in k_vote_map_cull)
#pragma unroll
        for (int j0 = 0; j0 < kPtsPerThread; j0 += 4) {
            // points (j0 .. j0+3) * 256 + lane are the 1024 consecutive points of quarter j0 / 4: a hidden quarter costs nothing (uniform branch)
            if (!((quarters >> (j0 >> 2)) & 1u)) {
#pragma unroll
                for (int u = 0; u < 4; ++u) { rlo[j0 + u] = -1.0f; rec[j0 + u] = 0u; }
                continue;
            }
            float4 pt[4];
            CullCand cc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) pt[u] = mapb[(uint32_t)(j0 + u) * kBlock + threadIdx.x];
            bool ok = true;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float3 p = xform_approx(ap, pt[u], ok);
                cc[u] = cull_candidates<EL3>(g, p, row_scale, col_scale, steep_clamps);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + u;
                const float r_lo = cull_r_lo(cc[u].r2);
                rlo[j] = -1.0f;
                const bool certain = packable & ok & !cc[u].unusual & !cc[u].multi & !(cc[u].r2 < rmin2);
                rec[j] = ((certain ? (uint32_t)cc[u].rb : kBmRowUncertain) << 11) | (uint32_t)(cc[u].cb & 2047);
                if (!certain) continue;
                const uint32_t px = (uint32_t)(cc[u].rb * g.cols + cc[u].cb);
                const int slot = table_claim<SLOT_ROWS, 64>(tags, cc[u].rb, cc[u].cb, px);
                if (slot < 0) continue;                                    // slot owned by another pixel: survive unconditionally
                rlo[j] = r_lo;
                rec[j] |= (uint32_t)slot << 20;
                // upper bound of the exact range (r_lo = r_approx*(1-1.5e-6)).  amin only ever decreases, so a plain read that is
                // already smaller makes the (same-address, hence serialised) atomic unnecessary for most points of a pixel
                const uint32_t hi = f2u(r_lo * (1.0f + 3.5e-6f));
                if (hi < amin[slot]) atomicMin(&amin[slot], hi);
            }
        }
        __syncthreads();
        // ---- phase 1b: a point survives unless it owns a slot and some point of the tile is provably nearer in the same pixel
        const uint32_t lane_word = threadIdx.x << 20;
#pragma unroll
        for (int j0 = 0; j0 < kPtsPerThread; j0 += 4) {
            // the four table reads go out together and nothing branches on them (the slot index is valid whatever the flags say)
            float am[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                am[u] = u2f(amin[rec[j0 + u] >> 20]);
            }
            bool sv[4];
            const bool quarter_live = ((quarters >> (j0 >> 2)) & 1u) != 0u;
#pragma unroll
            for (int u = 0; u < 4; ++u) sv[u] = quarter_live & !(rlo[j0 + u] > am[u]);
            // one LDS atomic per wave and group of four (see k_vote_map_cull)
            const uint64_t b0 = __builtin_amdgcn_ballot_w64(sv[0]), b1 = __builtin_amdgcn_ballot_w64(sv[1]),
                           b2 = __builtin_amdgcn_ballot_w64(sv[2]), b3 = __builtin_amdgcn_ballot_w64(sv[3]);
            const uint32_t n0 = (uint32_t)__popcll(b0), n1 = (uint32_t)__popcll(b1), n2 = (uint32_t)__popcll(b2), n3 = (uint32_t)__popcll(b3);
            const uint32_t total = n0 + n1 + n2 + n3;
            if (!total) continue;
            uint32_t base = 0;
            if ((threadIdx.x & 63u) == 0u) base = atomicAdd(&qcount, total);
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            if (base + total > (uint32_t)kBmQueue) continue;          // overflow, decided on the scalar unit: the whole tile goes exact in phase 2
            const uint64_t bal[4] = {b0, b1, b2, b3};
            const uint32_t off[4] = {0u, n0, n0 + n1, n0 + n1 + n2};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (!sv[u]) continue;
                const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal[u] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal[u], 0u));
                queue[base + off[u] + below] = ((rec[j0 + u] & 0xfffffu) | lane_word) | ((uint32_t)((j0 + u) * kBlock) << 20);
            }
        }
    }
    __syncthreads();
    for (int s = threadIdx.x; s < kBmSlots; s += kBlock) vals[s] = ~0ull;        // amin is dead from here on
    __syncthreads();
    // ---- phase 2: survivors.  Certain pixel: only the exact range; the others are re-queued densely and get the full exact projection
    const uint32_t nq = full_tile ? qcount : (uint32_t)kBmQueue + 1u;
    if (full_tile && threadIdx.x == 0 && (blockIdx.x & 63u) == 0u) {   // sampled diagnostic
        atomicAdd(&g_cull_stats[2], (unsigned long long)nq);
        atomicAdd(&g_cull_stats[3], (unsigned long long)nloc);
    }
    const Mat34 Tinv = load_mat(inv_poses + 12 * (size_t)kf);
    if (__builtin_expect(nq > (uint32_t)kBmQueue, 0)) {      // queue overflow: a superset is always correct (min is idempotent)
        for (uint32_t li = threadIdx.x; li < nloc; li += kBlock)
            exact_insert<B2L_IDENTITY, SLOT_ROWS, 64>(map, block_base + li, Tinv, b2l_h, g, vals, tags, imgk);
    } else {
        for (uint32_t q = threadIdx.x; q < nq; q += kBlock) {
            const uint32_t e = queue[q];
            const uint32_t li = e >> 20;
            const int row = (int)((e >> 11) & 511u), col = (int)(e & 2047u);
            const uint32_t i = block_base + li;
            if (row == (int)kBmRowUncertain) {
                const uint32_t up = atomicAdd(&ucount, 1u);
                if (up < (uint32_t)kBmUQueue) uqueue[up] = (uint16_t)li;
                else exact_insert<B2L_IDENTITY, SLOT_ROWS, 64>(map, i, Tinv, b2l_h, g, vals, tags, imgk);
                continue;
            }
            const uint32_t px = (uint32_t)(row * g.cols + col);
            const uint64_t v = ((uint64_t)exact_range_bits<B2L_IDENTITY>(map[i], Tinv, b2l_h) << 32) | (uint64_t)i;
            const int slot = table_claim<SLOT_ROWS, 64>(tags, row, col, px);
            if (slot >= 0) { if (v < vals[slot]) atomicMin(reinterpret_cast<unsigned long long*>(&vals[slot]), (unsigned long long)v); }   // vals only decreases: skip hopeless same-address atomics
            else img_min_u64(imgk + px, v);
        }
        __syncthreads();
        const uint32_t nu = min(ucount, (uint32_t)kBmUQueue);
        for (uint32_t q = threadIdx.x; q < nu; q += kBlock)
            exact_insert<B2L_IDENTITY, SLOT_ROWS, 64>(map, block_base + uqueue[q], Tinv, b2l_h, g, vals, tags, imgk);
    }
    __syncthreads();
    for (int s = threadIdx.x; s < kBmSlots; s += kBlock) {
        const uint32_t t = tags[s];
        if (t != kEmptyTag && vals[s] != ~0ull) img_min_u64(imgk + t, vals[s]);
    }
}



hipError_t map_range_images(const float4* map, size_t M, const double* inv_poses_dev, const float* approx_poses_dev, size_t kb, size_t nb,
                            HostMat34 b2l, int b2l_identity, Geom g, uint64_t* map_img, hipStream_t s, const KernelOpts& ko)
{
    if (!M || !nb) return hipSuccess;
    if (ko.map_kernel_variant >= 2 && approx_poses_dev) {
        const size_t per_block = (size_t)kBlock * kPtsPerThread;
        const unsigned kfg = kKfPerTile;
        dim3 grid(tile_kf_grid((M + per_block - 1) / per_block, nb, kfg));
#define LTM_LAUNCH_BM(ID, E) k_map_rimg_blockmin<ID, E><<<grid, dim3(kBlock), 0, s>>>(map, (uint32_t)M, inv_poses_dev, approx_poses_dev, (uint32_t)kb, (uint32_t)nb, kfg, b2l, g, map_img, nullptr, 0u, nullptr)
        const bool el3 = g.el_fit != 0;
        if (!b2l_identity) { if (el3) LTM_LAUNCH_BM(false, true); else LTM_LAUNCH_BM(false, false); }
        else { if (el3) LTM_LAUNCH_BM(true, true); else LTM_LAUNCH_BM(true, false); }
#undef LTM_LAUNCH_BM
        return hipGetLastError();
    }
    if (ko.map_kernel_variant >= 1) {
        const size_t per_block = (size_t)kBlock * kPtsPerThread;
        const unsigned kfg = kKfPerTile;
        dim3 grid(tile_kf_grid((M + per_block - 1) / per_block, nb, kfg));
        if (b2l_identity) k_map_rimg_lds<true><<<grid, dim3(kBlock), 0, s>>>(map, (uint32_t)M, inv_poses_dev, (uint32_t)kb, (uint32_t)nb, kfg, b2l, g, map_img);
        else k_map_rimg_lds<false><<<grid, dim3(kBlock), 0, s>>>(map, (uint32_t)M, inv_poses_dev, (uint32_t)kb, (uint32_t)nb, kfg, b2l, g, map_img);
        return hipGetLastError();
    }
    dim3 grid(grid_for(M), (unsigned)nb);
    if (b2l_identity) k_map_rimg<true><<<grid, dim3(kBlock), 0, s>>>(map, M, inv_poses_dev, kb, b2l, g, map_img);
    else k_map_rimg<false><<<grid, dim3(kBlock), 0, s>>>(map, M, inv_poses_dev, kb, b2l, g, map_img);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// Occlusion cull for the exact arg-min image on LARGE maps (KITTI scale: 45 M points = 11 000 tiles, 2000 keyframes).  There nearly every
// (tile, keyframe) pair is far from the sensor, subtends a pixel or two, and loses to nearer returns -- yet k_map_rimg_blockmin ran
// its ~130 instructions per point for all of them (reproject_map 484 of 887 ms per step).  Now, per batch of keyframes:
//   1. the pairs are visited in SHELLS of sensor-to-tile distance (r_near, 2 r_near, 4 r_near, ...); the first shell is projected as it
//      is (list-driven k_map_rimg_blockmin);
//   2. before every later shell a coarse image holds, per run of 8 pixels of an image row, the LARGEST range of the image built so far
//      (an empty pixel counts as 10000);
//   3. a pair of that shell is dropped iff every coarse block its bounding sphere can touch is covered by returns strictly nearer
//      than the sphere's nearest point: none of its points can then be the arg-min of any pixel (ties need an equal range, excluded
//      by the strict comparison with margins), so the final image is bit-identical; the others are projected and become occluders of
//      the shells behind them (a facade 80 m away hides what a single near / far split at 60 m could not).
// The sphere's pixel rectangle is conservative: centre direction from the float transform (1e-5 rad), angular radius asin(rho / d)
// enlarged, +-1 pixel, rows clamped like the reference clamps elevations; spheres that straddle the +-180 deg seam, reach above 80 deg
// of elevation, are nearer than two radii or touch more than 256 coarse entries are simply kept.
struct SphereRect { int r0, r1, c0, c1; float d_lo; bool cullable; };
__device__ __forceinline__ SphereRect sphere_rect(const RimgGeom& g, const float* __restrict__ ap, const float* __restrict__ tb)
{
    SphereRect o; o.cullable = false; o.r0 = o.r1 = o.c0 = o.c1 = 0; o.d_lo = 0.0f;
    const float smin = ap[15];
    if (!(smin > 0.999f)) return o;                                   // not (nearly) a rigid pose: angles are not preserved
    const float cx = 0.5f * (tb[0] + tb[3]), cy = 0.5f * (tb[1] + tb[4]), cz = 0.5f * (tb[2] + tb[5]);
    const float hx = 0.5f * (tb[3] - tb[0]), hy = 0.5f * (tb[4] - tb[1]), hz = 0.5f * (tb[5] - tb[2]);
    const float rho = __builtin_sqrtf(hx * hx + hy * hy + hz * hz) * 1.001f + 2.0e-3f;
    bool ok;
    const float3 l = xform_approx(ap, make_float4(cx, cy, cz, 0.0f), ok);
    const float d = __builtin_sqrtf(l.x * l.x + l.y * l.y + l.z * l.z);
    if (!ok || !(d > 2.0f * rho) || !(d < 8.0e3f)) return o;         // NaN-safe; beyond 8 km the 10000-sentinel arithmetic starts to matter
    const float ratio = rho / d;                                      // <= 0.5
    const float alpha = asinf(ratio) * 1.01f + 2.0e-4f;
    const float rxy = __builtin_sqrtf(l.x * l.x + l.y * l.y);
    const float el0 = atan2f(l.z, rxy), az0 = atan2f(l.y, l.x);
    if (!(fabsf(el0) + alpha < 1.39f)) return o;                      // 80 deg
    const float sa = sinf(alpha) / cosf(fabsf(el0) + alpha);
    if (!(sa < 0.95f)) return o;
    const float daz = asinf(sa) * 1.01f + 2.0e-4f;
    if (!(az0 - daz > -3.1415f && az0 + daz < 3.1415f)) return o;     // would wrap around the seam
    const float r2d = 57.29577951308232f;
    // same (monotone) pixel mapping as pixel_row_col, evaluated in plain float with a pixel of margin on either side
    const float row_hi_el = g.frows * (1.0f - ((el0 + alpha) * r2d + g.half_v) / g.vfov), row_lo_el = g.frows * (1.0f - ((el0 - alpha) * r2d + g.half_v) / g.vfov);
    const float col_lo = g.fcols * (((az0 - daz) * r2d + g.half_h) / g.hfov), col_hi = g.fcols * (((az0 + daz) * r2d + g.half_h) / g.hfov);
    o.r0 = (int)fminf(fmaxf(floorf(row_hi_el) - 1.0f, 0.0f), g.row_max);
    o.r1 = (int)fminf(fmaxf(ceilf(row_lo_el) + 1.0f, 0.0f), g.row_max);
    o.c0 = (int)fminf(fmaxf(floorf(col_lo) - 1.0f, 0.0f), g.col_max);
    o.c1 = (int)fminf(fmaxf(ceilf(col_hi) + 1.0f, 0.0f), g.col_max);
    o.d_lo = (d - rho) * 0.9995f - 2.0e-3f;                            // below the exact float range of every point of the tile
    o.cullable = o.d_lo > 0.0f;
    return o;
}

// One distance shell of the occlusion-culled launch: flags[tile * nb + kfb] = 1 iff the pair has not been projected yet, the tile's
// bounding box comes within [r_lo, r_hi) of the sensor (pairs that cannot be culled at all count as distance 0) and -- once a coarse
// maximum of the image built so far exists -- the tile is not hidden behind it.  done[] remembers projected / dropped pairs.
__global__ void __launch_bounds__(kBlock)
k_pair_shell_select(const float* __restrict__ approx_poses, uint32_t kb, uint32_t nb, const float* __restrict__ tile_bounds, uint32_t n_tiles, Geom gg,
                    float r_lo, float r_hi, const uint32_t* __restrict__ cmax, uint32_t rbs, uint32_t cbs, uint8_t* __restrict__ done, uint8_t* __restrict__ flags,
                    uint32_t* __restrict__ dirty, uint32_t dw, const float* __restrict__ sub_bounds, uint8_t* __restrict__ submask, unsigned long long* __restrict__ sub_stats)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tiles * nb) return;
    if (done[t]) { flags[t] = 0; return; }
    const uint32_t tile = t / nb, kfb = t % nb;
    const float* ap = approx_poses + 16 * (size_t)(kb + kfb);
    const float* tb = tile_bounds + 6 * (size_t)tile;
    float d2 = 0.0f;
#pragma unroll
    for (int d = 0; d < 3; ++d) { const float c = ap[9 + d]; const float e = fmaxf(fmaxf(tb[d] - c, c - tb[3 + d]), 0.0f); d2 = __builtin_fmaf(e, e, d2); }
    const RimgGeom g = make_geom(gg);
    const SphereRect sr = sphere_rect(g, ap, tb);
    if (!sr.cullable || !(d2 == d2)) d2 = 0.0f;                        // NaN bounds / unusable pose / too near: first shell, never dropped
    if (!(d2 >= r_lo * r_lo && d2 < r_hi * r_hi)) { flags[t] = 0; return; }
    bool live = true;
    if (cmax && sr.cullable) {
        const int cb0 = sr.c0 >> 3, cb1 = sr.c1 >> 3;
        if ((sr.r1 - sr.r0 + 1) * (cb1 - cb0 + 1) <= 256) {
            uint32_t m = 0;
            for (int r = sr.r0; r <= sr.r1; ++r)
                for (int cb = cb0; cb <= cb1; ++cb) m = max(m, cmax[((size_t)kfb * rbs + r) * cbs + cb]);
            live = !(u2f(m) < sr.d_lo);
        }
    }
    // round 6: a second look at the four 1024-point quarters of a tile that stays live -- a quarter's bounding sphere has about half the tile's radius,
    // so its pixel rectangle is a quarter of the area and its nearest point is farther: quarters behind the image built so far are masked out of the
    // projection kernel's phase 1 (the same argument as for the whole tile: every pixel the quarter can touch already holds a strictly nearer return)
    if (submask) {
        uint32_t m4 = 0xfu;
        if (live && cmax && sr.cullable && sub_bounds) {
            m4 = 0u;
            for (int qd = 0; qd < 4; ++qd) {
                const float* sb = sub_bounds + 6 * ((size_t)tile * 4 + qd);
                if (!(sb[0] <= sb[3])) continue;                      // empty quarter (past the end of the map)
                const SphereRect qr = sphere_rect(g, ap, sb);
                bool ql = true;
                if (qr.cullable) {
                    const int qb0 = qr.c0 >> 3, qb1 = qr.c1 >> 3;
                    if ((qr.r1 - qr.r0 + 1) * (qb1 - qb0 + 1) <= 256) {
                        uint32_t m = 0;
                        for (int r = qr.r0; r <= qr.r1; ++r)
                            for (int cb = qb0; cb <= qb1; ++cb) m = max(m, cmax[((size_t)kfb * rbs + r) * cbs + cb]);
                        ql = !(u2f(m) < qr.d_lo);
                    }
                }
                if (ql) m4 |= 1u << qd;
            }
            if (m4 == 0u) live = false;
        }
        submask[t] = (uint8_t)m4;
        if (sub_stats && live) { atomicAdd(&sub_stats[0], 4ull); atomicAdd(&sub_stats[1], (unsigned long long)__popc(m4)); }
    }
    done[t] = 1;
    flags[t] = live ? 1 : 0;
    // rows of keyframe kfb's image that the projection of this pair can lower: the rows of its pixel rectangle (the rectangle the cull itself
    // trusts to contain every pixel the tile can touch), all rows if it has none.  The coarse maximum is recomputed for those rows only.
    if (live && dirty) {
        uint32_t* dk = dirty + (size_t)kfb * dw;
        if (sr.cullable) {
            for (int w = sr.r0 >> 5; w <= sr.r1 >> 5; ++w) {
                const int lo = max(sr.r0, w * 32) & 31, hi = min(sr.r1, w * 32 + 31) & 31;
                const uint32_t m = (hi == 31 ? ~0u : ((1u << (hi + 1)) - 1u)) & ~((1u << lo) - 1u);
                if ((__hip_atomic_load(dk + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & m) != m) atomicOr(dk + w, m);
            }
        } else {
            for (uint32_t w = 0; w < dw; ++w) atomicOr(dk + w, ~0u);
        }
    }
}
// cmax[kfb][row][cb] = largest range bits of the 8 pixels cb*8 .. cb*8+7 of one image row (positive floats order like their bits; empty =
// 10000).  Row-granular on purpose: the rows just above what near facades were mapped at stay empty until far tiles fill them, and an
// 8 x 8 block would let those rows keep every far tile near the horizon alive.
__global__ void __launch_bounds__(kBlock)
k_coarse_max(const uint64_t* __restrict__ img, uint32_t rows, uint32_t cols, uint32_t cbs, uint32_t nb, uint32_t* __restrict__ cmax,
             const uint32_t* __restrict__ dirty, uint32_t dw)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nb * rows * cbs) return;
    const uint32_t cb = t % cbs, r = (t / cbs) % rows, kfb = t / (cbs * rows);
    // round 4: only rows that the previous shell's projections could lower are re-reduced (the pass used to re-read every image of the batch
    // before every shell: 27 ms per step on the KITTI-scale workload, a tenth of it now); an image row nothing was projected into keeps its maximum
    if (dirty && !((dirty[(size_t)kfb * dw + (r >> 5)] >> (r & 31u)) & 1u)) return;
    const uint64_t* __restrict__ im = img + ((size_t)kfb * rows + r) * cols;
    uint32_t m = 0;
    for (uint32_t cc = cb * 8; cc < min(cols, cb * 8 + 8); ++cc) m = max(m, (uint32_t)(im[cc] >> 32));
    cmax[t] = m;
}
struct FlagIs { uint8_t v; __host__ __device__ uint32_t operator()(uint8_t f) const { return f == v ? 1u : 0u; } };
__global__ void __launch_bounds__(kBlock)
k_pair_list_scatter(const uint8_t* __restrict__ flags, uint8_t v, const uint32_t* __restrict__ pos, uint32_t n, uint32_t* __restrict__ list, uint32_t* __restrict__ count)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const bool on = flags[t] == v;
    if (on) list[pos[t]] = t;
    if (t == n - 1) *count = pos[t] + (on ? 1u : 0u);
}

// one shell: [r_lo, r_hi) of sensor-to-tile distance; img = the image built so far (ignored for the first shell: use_cmax = 0)
// dirty = nb x ((rows + 31) / 32) words (or null: every row re-reduced): in, the rows the PREVIOUS shell's projections could lower (all ones before
// the second shell: the first one is not tracked); out, those of this shell's
hipError_t occlusion_shell_pairs(const float* approx_poses_dev, size_t kb, size_t nb, const float* tile_bounds_dev, size_t n_tiles, Geom g, float r_lo, float r_hi,
                                 const uint64_t* img, int use_cmax, uint32_t* cmax, uint8_t* done, uint8_t* flags, uint32_t* pos, uint32_t* list, uint32_t* count,
                                 void* temp, size_t temp_bytes, hipStream_t s, uint32_t* dirty, const float* sub_bounds_dev, uint8_t* submask, unsigned long long* sub_stats)
{
    const uint32_t n = (uint32_t)(n_tiles * nb);
    const uint32_t rbs = (uint32_t)g.rows, cbs = (uint32_t)(g.cols + 7) / 8;
    const uint32_t dw = (rbs + 31u) / 32u;
    if (use_cmax) k_coarse_max<<<dim3(grid_for((size_t)nb * rbs * cbs)), dim3(kBlock), 0, s>>>(img, (uint32_t)g.rows, (uint32_t)g.cols, cbs, (uint32_t)nb, cmax, dirty, dw);
    if (dirty) { hipError_t e0 = hipMemsetAsync(dirty, 0, (size_t)nb * dw * 4, s); if (e0 != hipSuccess) return e0; }
    k_pair_shell_select<<<dim3(grid_for(n)), dim3(kBlock), 0, s>>>(approx_poses_dev, (uint32_t)kb, (uint32_t)nb, tile_bounds_dev, (uint32_t)n_tiles, g, r_lo, r_hi,
                                                                use_cmax ? cmax : nullptr, rbs, cbs, done, flags, dirty, dw, sub_bounds_dev, submask, sub_stats);
    auto it = rocprim::make_transform_iterator(flags, FlagIs{1});
    hipError_t e = rocprim::exclusive_scan(temp, temp_bytes, it, pos, 0u, (size_t)n, rocprim::plus<uint32_t>(), s);
    if (e != hipSuccess) return e;
    k_pair_list_scatter<<<dim3(grid_for(n)), dim3(kBlock), 0, s>>>(flags, 1, pos, n, list, count);
    return hipGetLastError();
}
// k_map_rimg_blockmin over an explicit list of n_pairs (tile * nb + keyframe) pairs
hipError_t map_range_images_pairs(const float4* map, size_t M, const double* inv_poses_dev, const float* approx_poses_dev, size_t kb, size_t nb,
                                  HostMat34 b2l, int b2l_identity, Geom g, uint64_t* map_img, const uint32_t* pairs, size_t n_pairs, hipStream_t s, const KernelOpts& ko,
                                  const uint8_t* submask)
{
    if (!n_pairs) return hipSuccess;
    dim3 grid((unsigned)(((n_pairs + 7) / 8) * 8));
#define LTM_LAUNCH_BMP(ID, E) k_map_rimg_blockmin<ID, E><<<grid, dim3(kBlock), 0, s>>>(map, (uint32_t)M, inv_poses_dev, approx_poses_dev, (uint32_t)kb, (uint32_t)nb, 1u, b2l, g, map_img, pairs, (uint32_t)n_pairs, submask)
    const bool el3 = g.el_fit != 0;
    if (!b2l_identity) { if (el3) LTM_LAUNCH_BMP(false, true); else LTM_LAUNCH_BMP(false, false); }
    else { if (el3) LTM_LAUNCH_BMP(true, true); else LTM_LAUNCH_BMP(true, false); }
#undef LTM_LAUNCH_BMP
    return hipGetLastError();
}

// Removerter.cpp:381-413 (+ the diff of :458 / :515 / :572)
__global__ void __launch_bounds__(kBlock)
k_compare_flag(const uint32_t* __restrict__ scan_img, const uint64_t* __restrict__ map_img, size_t n, float thr, int mode,
               uint8_t* __restrict__ labels)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t mv = map_img[i];
    const float map_r = u2f((uint32_t)(mv >> 32));
    const float scan_r = u2f(scan_img[i]);
    const float diff = (mode == 0) ? (scan_r - map_r) : (map_r - scan_r);
    if (diff < 200.0f /* kValidDiffUpperBound, utility.h:94 */ && diff > thr) labels[(uint32_t)mv] = 1;
}

hipError_t compare_and_flag(const uint32_t* scan_img, const uint64_t* map_img, size_t n, float thr, int mode, uint8_t* labels,
                            hipStream_t s)
{
    if (!n) return hipSuccess;
    k_compare_flag<<<dim3(grid_for(n)), dim3(kBlock), 0, s>>>(scan_img, map_img, n, thr, mode, labels);
    return hipGetLastError();
}

__global__ void __launch_bounds__(kBlock)
k_single_rimg(const float4* __restrict__ pts, size_t n, HostMat34 T1, int has1, HostMat34 T2, int has2, Geom gg,
              uint64_t* __restrict__ img)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const RimgGeom g = make_geom(gg);
    const float4 p4 = pts[i];
    float3 p = make_float3(p4.x, p4.y, p4.z);
    if (has1) p = xform(to_dev(T1), p);
    if (has2) p = xform(to_dev(T2), p);
    const Sph s = cart2sph(p.x, p.y, p.z);
    const int px = pixel_index(g, s.az, s.el);
    img_min_u64(img + px, ((uint64_t)f2u(s.r) << 32) | (uint64_t)(uint32_t)i);
}

hipError_t single_range_image(const float4* pts, size_t n, const HostMat34* T1, const HostMat34* T2, Geom g, uint64_t* img,
                              hipStream_t s)
{
    if (!n) return hipSuccess;
    HostMat34 z{};
    k_single_rimg<<<dim3(grid_for(n)), dim3(kBlock), 0, s>>>(pts, n, T1 ? *T1 : z, T1 != nullptr, T2 ? *T2 : z, T2 != nullptr, g, img);
    return hipGetLastError();
}

__global__ void k_decode_image(const uint64_t* img, size_t npx, float* rimg, int32_t* ptidx)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npx) return;
    const uint64_t v = img[i];
    rimg[i] = u2f((uint32_t)(v >> 32));
    if (ptidx) ptidx[i] = (int32_t)(uint32_t)v;
}
hipError_t decode_image(const uint64_t* img, size_t npx, float* rimg, int32_t* ptidx, hipStream_t s)
{
    k_decode_image<<<dim3(grid_for(npx)), dim3(kBlock), 0, s>>>(img, npx, rimg, ptidx);
    return hipGetLastError();
}

__global__ void k_debug_project(const float* xyz, size_t n, Geom gg, float* sph, int32_t* rc)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const RimgGeom g = make_geom(gg);
    const Sph s = cart2sph(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
    sph[3 * i] = s.az; sph[3 * i + 1] = s.el; sph[3 * i + 2] = s.r;
    const int px = pixel_index(g, s.az, s.el);
    rc[2 * i] = px / g.cols; rc[2 * i + 1] = px % g.cols;
}
hipError_t debug_project(const float* xyz_dev, size_t n, Geom g, float* az_el_r, int32_t* row_col, hipStream_t s)
{
    if (!n) return hipSuccess;
    k_debug_project<<<dim3(grid_for(n)), dim3(kBlock), 0, s>>>(xyz_dev, n, g, az_el_r, row_col);
    return hipGetLastError();
}

__global__ void __launch_bounds__(kBlock)
k_selfcheck(float vfov, float hfov, unsigned long long* __restrict__ counts)
{
    const float inv_v = 1.0f / vfov, inv_h = 1.0f / hfov;
    unsigned long long bad0 = 0, bad1 = 0, bad2 = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
        const float a = u2f((uint32_t)i);
        const float e0 = rad2deg_exact(a), f0 = rad2deg_fast(a);
        const float e1 = a / vfov, f1 = div_by_const(a, vfov, inv_v, true);
        const float e2 = a / hfov, f2 = div_by_const(a, hfov, inv_h, true);
        // NaN payloads are irrelevant (NaN never reaches a pixel index in a defined way): compare as "both NaN"
        bad0 += !((f2u(e0) == f2u(f0)) | ((e0 != e0) & (f0 != f0)));
        bad1 += !((f2u(e1) == f2u(f1)) | ((e1 != e1) & (f1 != f1)));
        bad2 += !((f2u(e2) == f2u(f2)) | ((e2 != e2) & (f2 != f2)));
    }
    if (bad0) atomicAdd(counts + 0, bad0);
    if (bad1) atomicAdd(counts + 1, bad1);
    if (bad2) atomicAdd(counts + 2, bad2);
}
hipError_t selfcheck_fast_math(float vfov, float hfov, unsigned long long* counts_dev, hipStream_t s)
{
    k_selfcheck<<<dim3(8192), dim3(kBlock), 0, s>>>(vfov, hfov, counts_dev);
    return hipGetLastError();
}


} // namespace ltm
