// Does ds_bpermute_b32 / ds_read_b32 / ds_write_b64+ds_read_b64 overlap with VALU work of other wavefronts on the same
// SIMD?  Three kernels: VALU only (64 v_fma per iter), LDS only (8 ops per iter), both interleaved.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE, int LDSKIND>  // MODE 0: valu, 1: lds, 2: both
__global__ void __launch_bounds__(256) k(float* out, float a, float b, int iters)
{
    __shared__ unsigned buf[256 * 4];
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    unsigned u = threadIdx.x, addr = ((threadIdx.x + 3) & 63) * 4 + (threadIdx.x & ~63u) * 4;
    buf[threadIdx.x] = u;
    __syncthreads();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (MODE != 1) {
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x1) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x2) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x3) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x4) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x5) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x6) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x7) : "v"(a), "v"(b));
            }
            if (MODE != 0) {
                if (LDSKIND == 0) asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(u) : "v"(addr));
                else if (LDSKIND == 1) asm volatile("ds_read_b32 %0, %1" : "=v"(u) : "v"(addr));
                else asm volatile("ds_write_b32 %1, %0" : "+v"(u) : "v"(addr));
            }
        }
        if (MODE != 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + (float)u;
}

template <int MODE, int LDSKIND> float run()
{
    const int blocks = 2048, iters = 4000;
    float* d; (void)hipMalloc(&d, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE, LDSKIND><<<blocks, 256>>>(d, 1.0001f, 0.5f, 50);
    (void)hipEventRecord(e0);
    k<MODE, LDSKIND><<<blocks, 256>>>(d, 1.0001f, 0.5f, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipFree(d);
    return ms;
}

int main()
{
    const float v = run<0, 0>();
    std::printf("VALU only (64 v_fma/iter/wave): %.3f ms\n", v);
    const char* names[3] = {"ds_bpermute_b32", "ds_read_b32", "ds_write_b32"};
    float l[3] = {run<1, 0>(), run<1, 1>(), run<1, 2>()};
    float m[3] = {run<2, 0>(), run<2, 1>(), run<2, 2>()};
    for (int i = 0; i < 3; ++i)
        std::printf("%-16s 8/iter: alone %.3f ms, with VALU %.3f ms (sum would be %.3f, max %.3f)\n", names[i], l[i], m[i],
                    v + l[i], v > l[i] ? v : l[i]);
    return 0;
}
