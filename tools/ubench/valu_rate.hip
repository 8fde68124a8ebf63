// valu_rate.hip -- measures the VALU issue rate of gfx950 (wave64 instructions per second over the whole chip) for a few
// instruction kinds, to put SQ_INSTS_VALU-derived rates of the projection kernels against a measured ceiling instead of a
// nominal-clock model.   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(x) x x x x x x x x
template <int KIND>
__global__ void __launch_bounds__(256) k_rate(float* out, int iters)
{
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float b = 1.0001f, c = 0.5f;
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {        // v_fma_f32, 8 independent chains
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                              "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        } else if (KIND == 1) { // v_mul_f32
            REP8(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                              "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
        } else if (KIND == 2) { // v_rcp_f32 (transcendental)
            REP8(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                              "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (KIND == 3) { // v_cndmask / integer mix: v_and_b32
            REP8(asm volatile("v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n"
                              "v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
        } else if (KIND == 5) { // v_pk_fma_f32 on register pairs (two FMAs per lane and instruction)
            typedef float v2 __attribute__((ext_vector_type(2)));
            v2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}; const v2 e = {b, b}, f = {c, c};
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                              "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(e), "v"(f));)
            a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x; a5 = p2.y; a6 = p3.x; a7 = p3.y;
        } else if (KIND == 6) { // one dependent chain per wave: v_fma_f32 latency-bound issue
            REP8(asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                              "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                              : "+v"(a0) : "v"(b), "v"(c));)
        } else if (KIND == 7) { // v_cmp + v_cndmask pairs
            REP8(asm volatile("v_cmp_gt_f32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %4, vcc\n v_cmp_gt_f32 vcc, %1, %4\n v_cndmask_b32 %1, %1, %4, vcc\n"
                              "v_cmp_gt_f32 vcc, %2, %4\n v_cndmask_b32 %2, %2, %4, vcc\n v_cmp_gt_f32 vcc, %3, %4\n v_cndmask_b32 %3, %3, %4, vcc\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");)
        } else {                // v_fma_f64 on register pairs
            double d0 = a0, d1 = a1, d2 = a2, d3 = a3; const double e = 1.0001, f = 0.5;
            REP8(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"
                              "v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(e), "v"(f));)
            a0 = (float)d0; a1 = (float)d1; a2 = (float)d2; a3 = (float)d3;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int KIND> double run(const char* name, float* d, int blocks, int iters)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k_rate<KIND><<<blocks, 256>>>(d, 16);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k_rate<KIND><<<blocks, 256>>>(d, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    const double wave_insts = (double)blocks * 4.0 * (double)iters * 64.0;      // 4 waves per block, 64 instructions per iteration
    const double rate = wave_insts / (ms * 1e-3);
    std::printf("%-12s %8.3f ms  %.4e wave-insts/s  = %.3f per SIMD-cycle at 2.4 GHz (1024 SIMDs)  [1/4 = one wave64 instruction per 4 cycles]\n",
                name, ms, rate, rate / (1024.0 * 2.4e9));
    return rate;
}

int main()
{
    const int blocks = 256 * 8 * 4;      // 8 workgroups of 4 waves per CU resident, 4 rounds
    float* d; hipMalloc(&d, (size_t)blocks * 256 * 4);
    run<0>("v_fma_f32", d, blocks, 4000);
    run<1>("v_mul_f32", d, blocks, 4000);
    run<2>("v_rcp_f32", d, blocks, 1000);
    run<3>("v_and_b32", d, blocks, 4000);
    run<4>("v_fma_f64", d, blocks, 2000);
    run<5>("v_pk_fma_f32", d, blocks, 2000);
    run<6>("fma dep-chain", d, blocks, 2000);
    run<7>("cmp+cndmask", d, blocks, 2000);
    // the same with one workgroup per CU only (4 waves per CU = 1 per SIMD): single-wave issue rate
    run<0>("fma 1w/SIMD", d, 256, 8000);
    run<6>("dep 1w/SIMD", d, 256, 8000);
    hipFree(d);
    return 0;
}
