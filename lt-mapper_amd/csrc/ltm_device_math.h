// ltm_device_math.h -- gfx950 device arithmetic for the projection hot path.
//
// Everything here must reproduce, bit for bit, what the reference's generic x86-64 build computes
// (ltremovert/src/utility.cpp:38-56 cart2sph/rad2deg, :114-125 pixel index, :64-72 transform):
// binary32/binary64 IEEE arithmetic, round-to-nearest-even, NO fused multiply-add.  The translation
// unit is compiled with -ffp-contract=off; f32 divide and sqrt are the correctly rounded forms
// (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt).  Explicit __builtin_fmaf is used only
// where a proof of exactness is given next to it.
//
// atan2f follows the algorithm glibc ships for binary32 (flt-32 e_atan2f / s_atanf: 5-interval
// reduction at 7/16, 11/16, 19/16, 39/16, odd degree-11 polynomial split in two halves, hi/lo
// tables), restructured for a SIMT machine: one select-driven straight-line body for every finite
// non-zero argument pair, and a never-inlined slow function for zeros/inf/NaN/extreme ratios.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ltm {

__device__ __forceinline__ uint32_t f2u(float f) { return __float_as_uint(f); }
__device__ __forceinline__ float u2f(uint32_t u) { return __uint_as_float(u); }

// atan(|x|) for finite |x| < 2^25 given as its bit pattern ix (sign stripped); returns >= 0.
__device__ __forceinline__ float atanf_pos_core(float ax, uint32_t ix)
{
    // interval select: id = -1 (<7/16), 0 (<11/16), 1 (<19/16), 2 (<39/16), 3 (otherwise)
    const bool lt0 = ix < 0x3ee00000u;
    const bool lt1 = ix < 0x3f300000u;
    const bool lt2 = ix < 0x3f980000u;
    const bool lt3 = ix < 0x401c0000u;
    // numerator / denominator of the reduced argument (one divide on every path; x/1 is exact)
    //   id0: (2x-1)/(2+x)   id1: (x-1)/(x+1)   id2: (x-1.5)/(1+1.5x)   id3: -1/x   id-1: x/1
    const float two_x = 2.0f * ax;
    const float num = lt0 ? ax : (lt1 ? (two_x - 1.0f) : (lt2 ? (ax - 1.0f) : (lt3 ? (ax - 1.5f) : -1.0f)));
    const float den = lt0 ? 1.0f : (lt1 ? (2.0f + ax) : (lt2 ? (ax + 1.0f) : (lt3 ? (1.0f + 1.5f * ax) : ax)));
    const float hi = lt0 ? 0.0f : (lt1 ? 4.6364760399e-01f : (lt2 ? 7.8539812565e-01f : (lt3 ? 9.8279368877e-01f : 1.5707962513e+00f)));
    const float lo = lt0 ? 0.0f : (lt1 ? 5.0121582440e-09f : (lt2 ? 3.7748947079e-08f : (lt3 ? 3.4473217170e-08f : 7.5497894159e-08f)));
    const float t = num / den;
    const float z = t * t;
    const float w = z * z;
    const float s1 = z * (3.3333334327e-01f + w * (1.4285714924e-01f + w * (9.0908870101e-02f + w * (6.6610731184e-02f +
                     w * (4.9768779427e-02f + w * 1.6285819933e-02f)))));
    const float s2 = w * (-2.0000000298e-01f + w * (-1.1111110449e-01f + w * (-7.6918758452e-02f + w * (-5.8335702866e-02f +
                     w * -3.6531571299e-02f))));
    // id>=0: hi - ((t*(s1+s2) - lo) - t).  With hi = lo = 0 the same expression is -(t*s - t) == t - t*s
    // exactly (negation commutes with rounding), which is the id<0 formula.
    return hi - ((t * (s1 + s2) - lo) - t);
}

// full atanf (used by the slow path and by debug entry points)
__device__ __forceinline__ float atanf_exact(float x)
{
    const uint32_t hx = f2u(x), ix = hx & 0x7fffffffu;
    if (ix >= 0x4c000000u) {
        if (ix > 0x7f800000u) return x + x;
        const float big = 1.5707962513e+00f + 7.5497894159e-08f;
        return (hx >> 31) ? -big : big;
    }
    if (ix < 0x31000000u) return x;
    const float r = atanf_pos_core(u2f(ix), ix);
    return (hx >> 31) ? -r : r;
}

__device__ __noinline__ float atan2f_slow(float y, float x)
{
    const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f;
    const float pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    const uint32_t hx = f2u(x), hy = f2u(y), ix = hx & 0x7fffffffu, iy = hy & 0x7fffffffu;
    if (ix > 0x7f800000u || iy > 0x7f800000u) return x + y;
    if (hx == 0x3f800000u) return atanf_exact(y);
    const int m = (int)((hy >> 31) & 1u) | (int)((hx >> 30) & 2u);
    if (iy == 0) return (m < 2) ? y : ((m == 2) ? pi + tiny : -pi - tiny);
    if (ix == 0) return (hy >> 31) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000u) {
        if (iy == 0x7f800000u)
            return (m == 0) ? pi_o_4 + tiny : (m == 1) ? -pi_o_4 - tiny : (m == 2) ? 3.0f * pi_o_4 + tiny : -3.0f * pi_o_4 - tiny;
        return (m == 0) ? 0.0f : (m == 1) ? -0.0f : (m == 2) ? pi + tiny : -pi - tiny;
    }
    if (iy == 0x7f800000u) return (hy >> 31) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    const int k = ((int)iy - (int)ix) >> 23;
    float z;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
    else if ((hx >> 31) && k < -60) z = 0.0f;
    else z = atanf_exact(u2f(f2u(y / x) & 0x7fffffffu));
    return (m == 0) ? z : (m == 1) ? u2f(f2u(z) ^ 0x80000000u) : (m == 2) ? pi - (z - pi_lo) : (z - pi_lo) - pi;
}

__device__ __forceinline__ float atan2f_exact(float y, float x)
{
    const uint32_t hx = f2u(x), hy = f2u(y), ix = hx & 0x7fffffffu, iy = hy & 0x7fffffffu;
    const int k = ((int)iy - (int)ix) >> 23;
    // fast domain: both finite and non-zero, exponent gap within +-60
    const bool fast = (ix - 1u < 0x7f7fffffu) && (iy - 1u < 0x7f7fffffu) && (k <= 60) && (k >= -60);
    if (__builtin_expect(!fast, 0)) return atan2f_slow(y, x);
    const float q = u2f(f2u(y / x) & 0x7fffffffu);   // |y/x|, finite: |k|<=60 keeps it well inside the range
    const uint32_t iq = f2u(q);
    float z;
    if (iq >= 0x4c000000u) z = 1.5707962513e+00f + 7.5497894159e-08f;
    else if (iq < 0x31000000u) z = q;
    else z = atanf_pos_core(q, iq);
    const float pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    const bool xneg = hx >> 31, yneg = hy >> 31;
    const float zz = z - pi_lo;
    // m = 2*xneg + yneg : 0 -> z ; 1 -> -z ; 2 -> pi-(z-pi_lo) ; 3 -> (z-pi_lo)-pi
    return xneg ? (yneg ? (zz - pi) : (pi - zz)) : (yneg ? -z : z);
}

// utility.cpp:53-56  float rad2deg(float r) { return r * 180.0 / M_PI; }  -> (float)(((double)r*180.0)/pi)
__device__ __forceinline__ float rad2deg_exact(float r)
{
    return (float)(((double)r * 180.0) / 3.14159265358979323846);
}

// Fast form: q = (double)r * K with K = RN64(180 / M_PI_d) = 0x1.ca5dc1a63c1f8p+5.  q is within 3 ulp64 of the
// reference's RN64((r*180.0)/M_PI) (r*180.0 is exact: 24 x 8 significant bits), so RN32(q) equals the reference
// result unless a binary32 rounding midpoint (low 29 mantissa bits == 0x10000000) lies within those few ulp of q;
// that case (about 2e-8 of inputs) and results below the binary32 normal range take the exact division.
// ltm_debug_selfcheck() compares this function with rad2deg_exact over all 2^32 inputs on the device.
__device__ __forceinline__ float rad2deg_fast(float r)
{
    const double q = (double)r * 0x1.ca5dc1a63c1f8p+5;
    const uint64_t b = (uint64_t)__double_as_longlong(q);
    const uint32_t lo = (uint32_t)b & 0x1fffffffu;
    const uint32_t hi = (uint32_t)(b >> 32) & 0x7fffffffu;
    const bool near_mid = (lo - 0x0ffffff8u) <= 16u;            // |low29 - 2^28| <= 8
    const bool tiny = (hi < 0x39b00000u) & ((hi | (uint32_t)b) != 0u);   // 0 < |q| < 2^-100
    if (__builtin_expect(near_mid | tiny, 0)) return rad2deg_exact(r);
    return (float)q;
}

// a / b for a loop-invariant b: q0 = RN(a*y), r = a - q0*b (exact in an FMA), q = RN(q0 + r*y) with y = RN(1/b).
// This is the correctly rounded quotient (Markstein) when b's significand is not all ones and nothing under/overflows;
// outside 2^-60 <= |a| <= 2^60 the plain division is used.  Validity for the context's FOV constants is established
// by ltm_debug_selfcheck() (exhaustive over every binary32 `a`); a context whose constants fail it uses plain division.
__device__ __forceinline__ float div_by_const(float a, float b, float inv_b, bool fast_ok)
{
    const uint32_t ia = f2u(a) & 0x7fffffffu;
    if (__builtin_expect(!fast_ok | (ia - 0x21800000u > 0x3c000000u), 0)) return a / b;   // |a| outside [2^-60, 2^60]
    const float q0 = a * inv_b;
    const float r = __builtin_fmaf(-q0, b, a);
    return __builtin_fmaf(r, inv_b, q0);
}

struct Sph { float az, el, r; };

// utility.cpp:38-51
__device__ __forceinline__ Sph cart2sph(float x, float y, float z)
{
    Sph s;
    const float xx = x * x, yy = y * y, zz = z * z;
    const float xy = xx + yy;
    s.az = atan2f_exact(y, x);
    s.el = atan2f_exact(z, __builtin_sqrtf(xy));
    s.r = __builtin_sqrtf(xy + zz);
    return s;
}

struct RimgGeom {
    float vfov, hfov;   // degrees
    float half_v, half_h;
    float inv_v, inv_h; // RN32(1/vfov), RN32(1/hfov)
    int rows, cols;
    float frows, fcols, row_max, col_max;
    bool fast;          // the fast forms were verified for these constants
    float eps;          // Geom::cull_eps_px
    float el_c0, el_c1, el_c2, el_c3, el_tclamp;   // Geom::el_c / el_tclamp
    bool el_fit;
};

// utility.cpp:114-125
__device__ __forceinline__ void pixel_row_col(const RimgGeom& g, float az, float el, int& row, int& col)
{
    const float el_deg = g.fast ? rad2deg_fast(el) : rad2deg_exact(el);
    const float az_deg = g.fast ? rad2deg_fast(az) : rad2deg_exact(az);
    float fr = roundf(g.frows * (1.0f - div_by_const(el_deg + g.half_v, g.vfov, g.inv_v, g.fast)));
    float fc = roundf(g.fcols * div_by_const(az_deg + g.half_h, g.hfov, g.inv_h, g.fast));
    fr = (fr < 0.0f) ? 0.0f : fr;  fr = (g.row_max < fr) ? g.row_max : fr;
    fc = (fc < 0.0f) ? 0.0f : fc;  fc = (g.col_max < fc) ? g.col_max : fc;
    row = (int)fr; col = (int)fc;
}
__device__ __forceinline__ int pixel_index(const RimgGeom& g, float az, float el)
{
    int row, col;
    pixel_row_col(g, az, el, row, col);
    return row * g.cols + col;
}

// ---- bounded-error fast forms, used ONLY to decide which points need the exact arithmetic (never for a result) ----
// atan(a), 0 <= a <= 1: a*P(a^2), degree-5 minimax fit; |error| <= 1.8e-6 rad including binary32 evaluation
// (fit + bound: DESIGN.md 4.4; checked on the device by ltm_debug_cull_check).
__device__ __forceinline__ float atan_unit_approx(float a)
{
    const float t = a * a;
    float p = -1.171913743e-02f;
    p = __builtin_fmaf(p, t, 5.264735594e-02f);
    p = __builtin_fmaf(p, t, -1.164264902e-01f);
    p = __builtin_fmaf(p, t, 1.935403794e-01f);
    p = __builtin_fmaf(p, t, -3.326228261e-01f);
    p = __builtin_fmaf(p, t, 9.999772310e-01f);
    return a * p;
}
// asin(s), 0 <= s <= 0.70712: s*Q(s^2), degree-5 minimax fit (Lawson iteration, tools-free numpy fit recorded in DESIGN.md 4.1);
// |error| <= 4.0e-7 rad including binary32 evaluation.  Used for the azimuth: with inv_rxy = rsq(x^2 + y^2) already at hand for
// the elevation, min(|x|, |y|) * inv_rxy is the SINE of the octant-reduced azimuth, which saves the v_rcp_f32 (a quarter-rate
// instruction) and the max() that the tangent form min/max needs.
__device__ __forceinline__ float asin_octant_approx(float s)
{
    const float t = s * s;
    float p = 1.1113390326e-01f;
    p = __builtin_fmaf(p, t, -4.4537104666e-02f);
    p = __builtin_fmaf(p, t, 7.1046762168e-02f);
    p = __builtin_fmaf(p, t, 7.0704236627e-02f);
    p = __builtin_fmaf(p, t, 1.6695931554e-01f);
    p = __builtin_fmaf(p, t, 9.9999433756e-01f);
    return s * p;
}

// the same polynomial on two arguments at once (v_pk_mul_f32 / v_pk_fma_f32: one issue slot for both): the azimuth and the
// elevation ratio of one point.  Same operations in the same order as atan_unit_approx, so the same error bound.
typedef float ltm_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ ltm_v2f atan_unit_approx2(ltm_v2f a)
{
    const ltm_v2f t = a * a;
    ltm_v2f p = {-1.171913743e-02f, -1.171913743e-02f};
    p = __builtin_elementwise_fma(p, t, (ltm_v2f){5.264735594e-02f, 5.264735594e-02f});
    p = __builtin_elementwise_fma(p, t, (ltm_v2f){-1.164264902e-01f, -1.164264902e-01f});
    p = __builtin_elementwise_fma(p, t, (ltm_v2f){1.935403794e-01f, 1.935403794e-01f});
    p = __builtin_elementwise_fma(p, t, (ltm_v2f){-3.326228261e-01f, -3.326228261e-01f});
    p = __builtin_elementwise_fma(p, t, (ltm_v2f){9.999772310e-01f, 9.999772310e-01f});
    return a * p;
}

// atan2 for finite, non-zero y and x; |error| <= 3e-6 rad
__device__ __forceinline__ float atan2_approx(float y, float x)
{
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    float r = atan_unit_approx(mn * __builtin_amdgcn_rcpf(mx));
    r = (ay > ax) ? (1.57079632679f - r) : r;
    r = (x < 0.0f) ? (3.14159265359f - r) : r;
    return __builtin_copysignf(r, y);          // r >= 0: one v_bfi instead of compare + select
}

// same for x >= 0 (elevation: second argument is a norm)
__device__ __forceinline__ float atan2_approx_xpos(float y, float x)
{
    const float ay = fabsf(y);
    const float mx = fmaxf(x, ay), mn = fminf(x, ay);
    float r = atan_unit_approx(mn * __builtin_amdgcn_rcpf(mx));
    r = (ay > x) ? (1.57079632679f - r) : r;
    return __builtin_copysignf(r, y);
}

// PCL transformPointCloud<PointXYZI,double>: (float)(((m0*x + m1*y) + m2*z) + m3) per row, double math.
struct Mat34 { double m[12]; };

__device__ __forceinline__ float3 xform(const Mat34& T, float3 p)
{
    const double x = p.x, y = p.y, z = p.z;
    float3 o;
    o.x = (float)(((T.m[0] * x + T.m[1] * y) + T.m[2] * z) + T.m[3]);
    o.y = (float)(((T.m[4] * x + T.m[5] * y) + T.m[6] * z) + T.m[7]);
    o.z = (float)(((T.m[8] * x + T.m[9] * y) + T.m[10] * z) + T.m[11]);
    return o;
}

// An identity matrix still executes ((1*x + 0*y) + 0*z) + 0 in the reference: that is x + 0.0, which maps
// -0 to +0 and leaves everything else (finite) untouched.
__device__ __forceinline__ float3 xform_identity(float3 p)
{
    float3 o;
    o.x = p.x + 0.0f; o.y = p.y + 0.0f; o.z = p.z + 0.0f;
    return o;
}

// FLANN L2_Simple<float>: result = 0; for d: diff = a[d]-b[d]; result += diff*diff
__device__ __forceinline__ float sqdist_l2simple(float qx, float qy, float qz, float tx, float ty, float tz)
{
    const float dx = qx - tx, dy = qy - ty, dz = qz - tz;
    float r = dx * dx;
    r = r + dy * dy;
    r = r + dz * dz;
    return r;
}

} // namespace ltm
