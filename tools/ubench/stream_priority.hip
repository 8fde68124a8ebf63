// stream_priority.hip -- do HIP stream priorities shape how two concurrent kernel chains share one MI355X?
// Two chains of the step's shape ([VALU-bound launch of many short workgroups] -> [a run of small, latency-bound launches]) x P, on two streams:
//   serial      both chains on one stream
//   equal       two streams of equal priority, started together (they fall into lockstep: vote on vote, rest on rest)
//   hi/lo       chain A on a high-priority stream, chain B on a low-priority one
//   offset      equal priority, chain B starts with its small launches
// Prints the completion time of each chain.  Build: make ubench; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void __launch_bounds__(256) k_busy(float* out, int iters)
{
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-6f, c = 0.5f, d = 0.25f;
    for (int i = 0; i < iters; ++i) { a = a * 1.0001f + b; c = c * 0.9999f + a; d = d * 1.0002f + c; b = b * 0.9998f + d; }
    if (a + b + c + d == 12345.678f) out[0] = a;
}
__global__ void __launch_bounds__(256) k_small(const float4* in, float4* out, size_t n)
{
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) { float4 v = in[i]; v.x += 1.0f; out[i] = v; }
}

struct Chain { hipStream_t s; float4 *a, *b; float* o; };

static void issue(const Chain& c, int passes, int busy_blocks, int iters, int n_small, size_t small_n, bool small_first)
{
    for (int p = 0; p < passes; ++p) {
        if (!small_first) k_busy<<<busy_blocks, 256, 0, c.s>>>(c.o, iters);
        for (int j = 0; j < n_small; ++j) k_small<<<(unsigned)((small_n + 255) / 256), 256, 0, c.s>>>((j & 1) ? c.b : c.a, (j & 1) ? c.a : c.b, small_n);
        if (small_first) k_busy<<<busy_blocks, 256, 0, c.s>>>(c.o, iters);
    }
}

int main()
{
    int lo = 0, hi = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    printf("stream priority range: least %d, greatest %d\n", lo, hi);
    const size_t small_n = 1 << 20;      // 16 MB in, 16 MB out: a ~10-20 us kernel
    const int busy_blocks = 200000, iters = 700, n_small = 40, passes = 6;
    hipStream_t s_eq1, s_eq2, s_hi, s_lo;
    CK(hipStreamCreateWithFlags(&s_eq1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s_eq2, hipStreamNonBlocking));
    CK(hipStreamCreateWithPriority(&s_hi, hipStreamNonBlocking, hi));
    CK(hipStreamCreateWithPriority(&s_lo, hipStreamNonBlocking, lo));
    Chain A, B;
    for (Chain* c : {&A, &B}) { CK(hipMalloc(&c->a, small_n * 16)); CK(hipMalloc(&c->b, small_n * 16)); CK(hipMalloc(&c->o, 64)); CK(hipMemset(c->a, 0, small_n * 16)); }
    hipEvent_t t0, ta, tb;
    CK(hipEventCreate(&t0)); CK(hipEventCreate(&ta)); CK(hipEventCreate(&tb));
    auto run = [&](const char* name, hipStream_t sa, hipStream_t sb, bool b_small_first) -> int {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipDeviceSynchronize());
            A.s = sa; B.s = sb;
            CK(hipEventRecord(t0, sa));
            if (sa != sb) CK(hipStreamWaitEvent(sb, t0, 0));
            // interleave the issue so that neither stream's queue runs dry on the host side
            for (int p = 0; p < passes; ++p) { issue(A, 1, busy_blocks, iters, n_small, small_n, false); issue(B, 1, busy_blocks, iters, n_small, small_n, b_small_first); }
            CK(hipEventRecord(ta, sa)); CK(hipEventRecord(tb, sb));
            CK(hipDeviceSynchronize());
            float ma = 0, mb = 0;
            CK(hipEventElapsedTime(&ma, t0, ta)); CK(hipEventElapsedTime(&mb, t0, tb));
            printf("%-28s rep %d: chain A done at %7.2f ms, chain B done at %7.2f ms\n", name, rep, ma, mb);
        }
        return 0;
    };
    // the parts alone
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(t0, s_eq1)); k_busy<<<busy_blocks, 256, 0, s_eq1>>>(A.o, iters); CK(hipEventRecord(ta, s_eq1));
        A.s = s_eq1; for (int j = 0; j < n_small; ++j) k_small<<<(unsigned)((small_n + 255) / 256), 256, 0, s_eq1>>>((j & 1) ? A.b : A.a, (j & 1) ? A.a : A.b, small_n);
        CK(hipEventRecord(tb, s_eq1)); CK(hipDeviceSynchronize());
        float m1 = 0, m2 = 0; CK(hipEventElapsedTime(&m1, t0, ta)); CK(hipEventElapsedTime(&m2, ta, tb));
        printf("alone: busy launch %.2f ms, %d small launches %.2f ms -> one chain of %d passes ~ %.1f ms\n", m1, n_small, m2, passes, passes * (m1 + m2));
    }
    if (run("serial (one stream)", s_eq1, s_eq1, false)) return 1;
    if (run("equal priority, together", s_eq1, s_eq2, false)) return 1;
    if (run("equal priority, B offset", s_eq1, s_eq2, true)) return 1;
    if (run("A high / B low priority", s_hi, s_lo, false)) return 1;
    if (run("A high / B low, B offset", s_hi, s_lo, true)) return 1;
    return 0;
}
