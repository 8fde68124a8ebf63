// micro-benchmark: is packed FP32 (v_pk_fma_f32) faster than scalar v_fma_f32 per element on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int N> __global__ void k_scalar(float* out, float a, float b, int iters)
{
    float x[N];
    for (int i = 0; i < N; ++i) x[i] = threadIdx.x * 1e-3f + i;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < N; ++i) x[i] = __builtin_fmaf(x[i], a, b);
    float s = 0; for (int i = 0; i < N; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int N> __global__ void k_packed(float* out, float a, float b, int iters)
{
    f2 x[N / 2];
    for (int i = 0; i < N / 2; ++i) x[i] = (f2){threadIdx.x * 1e-3f + i, threadIdx.x * 2e-3f + i};
    const f2 av = {a, a}, bv = {b, b};
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < N / 2; ++i) x[i] = __builtin_elementwise_fma(x[i], av, bv);
    float s = 0; for (int i = 0; i < N / 2; ++i) s += x[i].x + x[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main()
{
    float* d; hipMalloc(&d, 256 * 2048 * 4 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4096, blocks = 256 * 8;
    for (int rep = 0; rep < 2; ++rep) {
        float ms;
        hipEventRecord(e0); k_scalar<16><<<blocks, 256>>>(d, 1.0001f, 0.5f, iters); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        double fl = 2.0 * 16 * iters * blocks * 256;
        printf("scalar v_fma_f32 : %.3f ms  %.1f TFLOP/s\n", ms, fl / ms / 1e9);
        hipEventRecord(e0); k_packed<16><<<blocks, 256>>>(d, 1.0001f, 0.5f, iters); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("packed v_pk_fma  : %.3f ms  %.1f TFLOP/s\n", ms, fl / ms / 1e9);
    }
    return 0;
}
