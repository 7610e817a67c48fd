// Micro-benchmark: issue rate of plain vs packed float32 FMA and u32 add on gfx950 (wave64).
// Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, float a, float b, int iters)
{
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7};
    unsigned u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, u4 = u0 + 4, u5 = u0 + 5, u6 = u0 + 6, u7 = u0 + 7;
    const f2 av = {a, a}, bv = {b, b};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                x0 = fmaf(x0, a, b); x1 = fmaf(x1, a, b); x2 = fmaf(x2, a, b); x3 = fmaf(x3, a, b);
                x4 = fmaf(x4, a, b); x5 = fmaf(x5, a, b); x6 = fmaf(x6, a, b); x7 = fmaf(x7, a, b);
            }
        } else if (MODE == 1) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                p0 = __builtin_elementwise_fma(p0, av, bv); p1 = __builtin_elementwise_fma(p1, av, bv);
                p2 = __builtin_elementwise_fma(p2, av, bv); p3 = __builtin_elementwise_fma(p3, av, bv);
                p0 = __builtin_elementwise_fma(p0, av, bv); p1 = __builtin_elementwise_fma(p1, av, bv);
                p2 = __builtin_elementwise_fma(p2, av, bv); p3 = __builtin_elementwise_fma(p3, av, bv);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                u0 = u0 * 3u + u1; u1 += u2 ^ 5u; u2 += u3; u3 += u4 ^ 7u; u4 += u5; u5 += u6 ^ 9u; u6 += u7; u7 += u0;
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x +
                                                 p2.y + p3.x + p3.y + (float)(u0 + u1 + u2 + u3 + u4 + u5 + u6 + u7);
}

template <int MODE> void run(const char* name, int blocks, double ops_per_iter_lane)
{
    float* d; hipMalloc(&d, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    k<MODE><<<blocks, 256>>>(d, 1.0001f, 0.5f, 100);
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(d, 1.0001f, 0.5f, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double inst = (double)blocks * 4 /*waves*/ * iters * 64.0 /* wave-instructions per iter */;
    std::printf("%-10s blocks=%5d  %.3f ms  %.3e wave-instr/s  => %.2f SIMD-cycles per wave-instr at 2.4 GHz (1024 SIMDs), %s\n",
                name, blocks, ms, inst / (ms * 1e-3), 1024.0 * 2.4e9 / (inst / (ms * 1e-3)),
                MODE == 1 ? "2 FMAs per lane per instr" : "1 op per lane per instr");
    (void)ops_per_iter_lane;
    hipFree(d);
}

int main()
{
    for (int blocks : {256, 1024, 2048}) {
        run<0>("fma_f32", blocks, 64);
        run<1>("pk_fma_f32", blocks, 128);
        run<2>("u32 mix", blocks, 64);
    }
    return 0;
}
