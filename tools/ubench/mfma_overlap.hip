// mfma_overlap.hip -- does the matrix pipe of gfx950 run BESIDE the vector ALU?  (measured before restructuring anything.)
//
// The dominant kernel, k_vote_map_cull, issues ~98 VALU wave-instructions per point-projection and is bound by VALU issue; ~12 of them are the
// affine transform A (p - c) of phase 1.  v_mfma_f32_16x16x4_f32 could produce the local coordinates of 16 points x 16 keyframes in three
// instructions (one per axis; the x / y / z of a (point, keyframe) pair then sit in the same lane) -- IF the MFMA pipe really runs concurrently
// with the VALU work of the other waves of the SIMD.  This micro-benchmark answers that with the kernel's own instruction budget: per group of four
// (point, keyframe) pairs per lane a wave issues
//     A: 392 VALU instructions                     (98 x 4, today)
//     B: 344 VALU instructions + 3 MFMA 16x16x4    (the transform moved to the matrix pipe)
//     C: 344 VALU instructions                     (the transform for free: the upper bound of what B can reach)
//     D: 3 MFMA 16x16x4 alone
// at 8 waves per SIMD over the whole chip.  t(B) ~ t(C) means the matrix pipe is free beside the VALU; t(B) ~ t(C) + t(D) means it is not.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_overlap.hip -o /tmp/mfma_overlap && /tmp/mfma_overlap
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float v4f __attribute__((ext_vector_type(4)));

#define VALU8(a0, a1, a2, a3, a4, a5, a6, a7, b, c)                                                                                        \
    asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"             \
                 "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"             \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c))

template <int N_VALU8, int N_MFMA>
__global__ void __launch_bounds__(256) k_mix(float* out, int iters)
{
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float b = 1.0001f, c = 0.5f;
    v4f acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
    float ma = a0, mb = a1;
    for (int i = 0; i < iters; ++i) {
        if (N_MFMA >= 1) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ma, mb, acc0, 0, 0, 0);
        if (N_MFMA >= 2) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(mb, ma, acc1, 0, 0, 0);
        if (N_MFMA >= 3) acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(ma, ma, acc2, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < N_VALU8; ++r) VALU8(a0, a1, a2, a3, a4, a5, a6, a7, b, c);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + acc0.x + acc1.y + acc2.z + acc0.w;
}

template <int N_VALU8, int N_MFMA> double run(const char* name, float* d, int blocks, int iters)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k_mix<N_VALU8, N_MFMA><<<blocks, 256>>>(d, 16);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        k_mix<N_VALU8, N_MFMA><<<blocks, 256>>>(d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double waves = (double)blocks * 4.0;
    printf("%-46s %8.3f ms   %.3e VALU wave-instr/s   %.3e MFMA/s\n", name, best, waves * iters * N_VALU8 * 8.0 / (best * 1e-3), waves * iters * N_MFMA / (best * 1e-3));
    return best;
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount * 8;          // 8 workgroups of 4 waves per CU = 8 waves per SIMD
    float* d; hipMalloc(&d, (size_t)blocks * 256 * sizeof(float));
    const int iters = 2000;
    printf("%s, %d CUs, %d workgroups x 256 threads, %d iterations (one iteration = four (point, keyframe) pairs per lane)\n", p.name, p.multiProcessorCount, blocks, iters);
    const double tA = run<49, 0>("A: 392 VALU (today's budget)", d, blocks, iters);
    const double tB = run<43, 3>("B: 344 VALU + 3 MFMA 16x16x4 f32", d, blocks, iters);
    const double tC = run<43, 0>("C: 344 VALU (transform for free)", d, blocks, iters);
    const double tD = run<0, 3>("D: 3 MFMA 16x16x4 f32 alone", d, blocks, iters);
    printf("B / A = %.3f (what moving the affine transform to the matrix pipe can buy at best: C / A = %.3f); B - C = %.3f ms vs D = %.3f ms: the matrix pipe runs %s the VALU\n",
           tB / tA, tC / tA, tB - tC, tD, (tB - tC) < 0.35 * tD ? "BESIDE" : "IN SERIES WITH");
    return 0;
}
