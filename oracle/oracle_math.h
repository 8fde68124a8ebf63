/*
 * oracle_math.h -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * Restatement of the libm routines the reference reaches through
 *   std::atan2(float,float)   ltremovert/src/utility.cpp:46-47  (cart2sph)
 * i.e. glibc's atan2f/atanf.  The reference pins glibc only through its Docker
 * base (docker/Dockerfile:1, osrf/ros:noetic => Ubuntu 20.04 => glibc 2.31), whose
 * sysdeps/ieee754/flt-32/{e_atan2f.c,s_atanf.c} is the fdlibm single-precision
 * algorithm (no x86_64 multiarch/FMA variant exists for atanf/atan2f in that
 * release, so it is plain SSE2 binary32 arithmetic).  The published algorithm is
 * restated below from its description: argument reduction to one of five
 * intervals with the breakpoints 7/16, 11/16, 19/16, 39/16, an odd degree-11
 * polynomial in two interleaved halves, and hi/lo split table constants.
 *
 * Pin: oracle/pin_atan2f.c compares this restatement bit-for-bit against the libm
 * of the machine it runs on (glibc 2.35 in the build container, same source
 * file) over all 2^32 atanf inputs and >= 2^31 atan2f pairs.  See DESIGN.md.
 *
 * Everything here is binary32 mul/add/div with no fused contraction: build with
 * -ffp-contract=off.
 */
#ifndef LTM_ORACLE_MATH_H
#define LTM_ORACLE_MATH_H

#include <stdint.h>
#include <string.h>

static inline uint32_t om_f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float om_u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* Table constants are the published decimal literals; the compiler's decimal->binary32
 * conversion is what the shipped libm contains (e.g. aT[0] parses to 0x3eaaaaab, one ulp
 * above the 0x3eaaaaaa quoted in the algorithm's commentary -- verified against the
 * .rodata of glibc 2.35's libm.so.6 and by the exhaustive pin). */
static const float OM_ATANHI[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
static const float OM_ATANLO[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
static const float OM_AT[11] = {3.3333334327e-01f, -2.0000000298e-01f, 1.4285714924e-01f, -1.1111110449e-01f,
                                9.0908870101e-02f, -7.6918758452e-02f, 6.6610731184e-02f, -5.8335702866e-02f,
                                4.9768779427e-02f, -3.6531571299e-02f, 1.6285819933e-02f};

static inline float om_atanf(float x)
{
    const uint32_t hx = om_f2u(x);
    const uint32_t ix = hx & 0x7fffffffu;
    int id;
    if (ix >= 0x4c000000u) {                 /* |x| >= 2^25 (or NaN): glibc flt-32 threshold */
        if (ix > 0x7f800000u) return x + x;  /* NaN */
        float big = OM_ATANHI[3] + OM_ATANLO[3];
        return (hx >> 31) ? -big : big;
    }
    if (ix < 0x3ee00000u) {                  /* |x| < 7/16 */
        if (ix < 0x31000000u) return x;      /* |x| < 2^-29: atan(x) = x to rounding */
        id = -1;
    } else {
        x = om_u2f(ix);                      /* fabsf */
        if (ix < 0x3f980000u) {              /* |x| < 19/16 */
            if (ix < 0x3f300000u) { id = 0; x = (2.0f * x - 1.0f) / (2.0f + x); }
            else                  { id = 1; x = (x - 1.0f) / (x + 1.0f); }
        } else {
            if (ix < 0x401c0000u) { id = 2; x = (x - 1.5f) / (1.0f + 1.5f * x); }
            else                  { id = 3; x = -1.0f / x; }
        }
    }
    const float z = x * x;
    const float w = z * z;
    const float a0 = OM_AT[0], a1 = OM_AT[1], a2 = OM_AT[2],
                a3 = OM_AT[3], a4 = OM_AT[4], a5 = OM_AT[5],
                a6 = OM_AT[6], a7 = OM_AT[7], a8 = OM_AT[8],
                a9 = OM_AT[9], a10 = OM_AT[10];
    const float s1 = z * (a0 + w * (a2 + w * (a4 + w * (a6 + w * (a8 + w * a10)))));
    const float s2 = w * (a1 + w * (a3 + w * (a5 + w * (a7 + w * a9))));
    if (id < 0) return x - x * (s1 + s2);
    float r = OM_ATANHI[id] - ((x * (s1 + s2) - OM_ATANLO[id]) - x);
    return (hx >> 31) ? -r : r;
}

static inline float om_atan2f(float y, float x)
{
    const float tiny = 1.0e-30f;
    const float pi_o_4 = 7.8539818525e-01f;
    const float pi_o_2 = 1.5707963705e+00f;
    const float pi = 3.1415927410e+00f;
    const float pi_lo = -8.7422776573e-08f;
    const uint32_t hx = om_f2u(x), hy = om_f2u(y);
    const uint32_t ix = hx & 0x7fffffffu, iy = hy & 0x7fffffffu;
    if (ix > 0x7f800000u || iy > 0x7f800000u) return x + y;        /* NaN */
    if (hx == 0x3f800000u) return om_atanf(y);                      /* x == 1 */
    const int m = (int)((hy >> 31) & 1u) | (int)((hx >> 30) & 2u);  /* 2*sign(x)+sign(y) */
    if (iy == 0) {
        switch (m) {
        case 0: case 1: return y;
        case 2: return pi + tiny;
        default: return -pi - tiny;
        }
    }
    if (ix == 0) return (hy >> 31) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000u) {
        if (iy == 0x7f800000u) {
            switch (m) {
            case 0: return pi_o_4 + tiny;
            case 1: return -pi_o_4 - tiny;
            case 2: return 3.0f * pi_o_4 + tiny;
            default: return -3.0f * pi_o_4 - tiny;
            }
        } else {
            switch (m) {
            case 0: return 0.0f;
            case 1: return -0.0f;
            case 2: return pi + tiny;
            default: return -pi - tiny;
            }
        }
    }
    if (iy == 0x7f800000u) return (hy >> 31) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    const int k = ((int)iy - (int)ix) >> 23;
    float z;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;      /* |y/x| > 2^60 */
    else if ((hx >> 31) && k < -60) z = 0.0f;   /* |y|/x < -2^60 */
    else z = om_atanf(om_u2f(om_f2u(y / x) & 0x7fffffffu));
    switch (m) {
    case 0: return z;
    case 1: return om_u2f(om_f2u(z) ^ 0x80000000u);
    case 2: return pi - (z - pi_lo);
    default: return (z - pi_lo) - pi;
    }
}

#endif
