/*
 * pin_atan2f.c -- TEST INFRASTRUCTURE.  Pins oracle_math.h's atanf/atan2f restatement
 * against the libm of this machine (the function the reference actually calls through
 * std::atan2(float,float), ltremovert/src/utility.cpp:46-47).
 *
 *   ./pin_atan2f [log2_pairs=31] [exhaustive_atanf=1]
 *
 * Exit code 0 iff zero bit mismatches (NaN payloads compared as "both NaN").
 * Build: gcc -O2 -ffp-contract=off -fopenmp pin_atan2f.c -lm
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "oracle_math.h"

static inline uint64_t splitmix64(uint64_t *s)
{
    uint64_t z = (*s += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

static int same(float a, float b)
{
    if (a != a && b != b) return 1;
    return om_f2u(a) == om_f2u(b);
}

int main(int argc, char **argv)
{
    int lg = argc > 1 ? atoi(argv[1]) : 31;
    int exhaustive = argc > 2 ? atoi(argv[2]) : 1;
    unsigned long long bad1 = 0, bad2 = 0, n2 = 0;

    if (exhaustive) {
#pragma omp parallel for reduction(+ : bad1) schedule(static)
        for (long long i = 0; i < (1ll << 32); ++i) {
            float x = om_u2f((uint32_t)i);
            if (!same(om_atanf(x), atanf(x))) {
                if (bad1 < 5) fprintf(stderr, "atanf mismatch x=%a got=%a ref=%a\n", x, om_atanf(x), atanf(x));
                bad1++;
            }
        }
        printf("atanf exhaustive 2^32 inputs: %llu mismatches\n", bad1);
    }

    const long long npairs = 1ll << lg;
    const int nchunk = 1024;
#pragma omp parallel for reduction(+ : bad2, n2) schedule(dynamic)
    for (int c = 0; c < nchunk; ++c) {
        uint64_t s = 0x20250224ull * 1315423911ull + (uint64_t)c;
        for (long long i = 0; i < npairs / nchunk; ++i) {
            uint64_t r = splitmix64(&s);
            float y, x;
            switch (i & 3) {
            case 0: /* arbitrary bit patterns */
                y = om_u2f((uint32_t)r); x = om_u2f((uint32_t)(r >> 32)); break;
            case 1: { /* lidar-like magnitudes: +-[2^-10, 2^8) */
                uint32_t a = (uint32_t)r, b = (uint32_t)(r >> 32);
                y = om_u2f((a & 0x807fffffu) | ((117u + (a >> 23) % 18u) << 23));
                x = om_u2f((b & 0x807fffffu) | ((117u + (b >> 23) % 18u) << 23));
                break; }
            case 2: { /* near-equal exponents (ratio ~ 1 .. interval edges) */
                uint32_t a = (uint32_t)r, b = (uint32_t)(r >> 32);
                uint32_t e = 100u + (a >> 24) % 60u;
                y = om_u2f((a & 0x807fffffu) | (e << 23));
                x = om_u2f((b & 0x807fffffu) | ((e + (b >> 29) - 3u) << 23));
                break; }
            default: { /* elevation-like: second arg is a non-negative sqrt */
                uint32_t a = (uint32_t)r, b = (uint32_t)(r >> 32);
                y = om_u2f((a & 0x807fffffu) | ((110u + (a >> 23) % 25u) << 23));
                x = om_u2f((b & 0x007fffffu) | ((110u + (b >> 23) % 25u) << 23));
                break; }
            }
            if (!same(om_atan2f(y, x), atan2f(y, x))) {
                if (bad2 < 5) fprintf(stderr, "atan2f mismatch y=%a x=%a got=%a ref=%a\n", y, x, om_atan2f(y, x), atan2f(y, x));
                bad2++;
            }
            n2++;
        }
    }
    /* structured specials */
    const float sp[] = {0.0f, -0.0f, 1.0f, -1.0f, INFINITY, -INFINITY, NAN, 1e-45f, -1e-45f, 3.4e38f, -3.4e38f,
                        0.4375f, 0.6875f, 1.1875f, 2.4375f, 0x1p-29f, 0x1p34f, 0x1p60f, 0x1p-60f, 0x1p61f, 0x1p-61f};
    const int nsp = (int)(sizeof sp / sizeof sp[0]);
    for (int a = 0; a < nsp; ++a)
        for (int b = 0; b < nsp; ++b) {
            if (!same(om_atan2f(sp[a], sp[b]), atan2f(sp[a], sp[b]))) {
                fprintf(stderr, "atan2f special mismatch y=%a x=%a got=%a ref=%a\n", sp[a], sp[b], om_atan2f(sp[a], sp[b]), atan2f(sp[a], sp[b]));
                bad2++;
            }
            n2++;
        }
    printf("atan2f %llu pairs: %llu mismatches\n", n2, bad2);
    return (bad1 || bad2) ? 1 : 0;
}
