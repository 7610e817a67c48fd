// Micro-benchmark: does the ORDER of 2-cycle and 4-cycle vector instructions matter?  Streams of 8 x v_add_u32 (2-cycle
// class) and 8 x v_lshlrev_b32 (4-cycle class) per iteration, interleaved (ABAB...) or grouped (AAAAAAAABBBBBBBB), on
// independent registers, at 5 wavefronts per SIMD.  Ideal by the cost table: (8*2 + 8*4) / 16 = 3 cycles/instruction.
#include <hip/hip_runtime.h>
#include <cstdio>

extern __shared__ unsigned char dyn_lds[];
#define A(r) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r) : "v"(a));
#define B(r) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(r));
#define F(r) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r) : "v"(a));
#define C(r) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(r));

#define KERNEL(NAME, BODY)                                                                          \
    __global__ void __launch_bounds__(256) NAME(unsigned* out, unsigned a, int iters)               \
    {                                                                                               \
        if (a == 0xdeadbeefu) dyn_lds[threadIdx.x] = 1;                                             \
        unsigned r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, \
                 r7 = r0 + 7, s0 = r0 + 8, s1 = r0 + 9, s2 = r0 + 10, s3 = r0 + 11, s4 = r0 + 12, s5 = r0 + 13,  \
                 s6 = r0 + 14, s7 = r0 + 15;                                                        \
        for (int i = 0; i < iters; ++i) { BODY }                                                    \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7; \
    }

KERNEL(k_inter, A(r0) B(s0) A(r1) B(s1) A(r2) B(s2) A(r3) B(s3) A(r4) B(s4) A(r5) B(s5) A(r6) B(s6) A(r7) B(s7))
KERNEL(k_group, A(r0) A(r1) A(r2) A(r3) A(r4) A(r5) A(r6) A(r7) B(s0) B(s1) B(s2) B(s3) B(s4) B(s5) B(s6) B(s7))
KERNEL(k_pairs, A(r0) A(r1) B(s0) B(s1) A(r2) A(r3) B(s2) B(s3) A(r4) A(r5) B(s4) B(s5) A(r6) A(r7) B(s6) B(s7))
KERNEL(k_finter, F(r0) C(s0) F(r1) C(s1) F(r2) C(s2) F(r3) C(s3) F(r4) C(s4) F(r5) C(s5) F(r6) C(s6) F(r7) C(s7))
KERNEL(k_fgroup, F(r0) F(r1) F(r2) F(r3) F(r4) F(r5) F(r6) F(r7) C(s0) C(s1) C(s2) C(s3) C(s4) C(s5) C(s6) C(s7))
KERNEL(k_allA, A(r0) A(s0) A(r1) A(s1) A(r2) A(s2) A(r3) A(s3) A(r4) A(s4) A(r5) A(s5) A(r6) A(s6) A(r7) A(s7))
KERNEL(k_allB, B(r0) B(s0) B(r1) B(s1) B(r2) B(s2) B(r3) B(s3) B(r4) B(s4) B(r5) B(s5) B(r6) B(s6) B(r7) B(s7))

template <typename K> void run(const char* name, K kern, int waves_per_simd)
{
    const int lds = waves_per_simd >= 8 ? 0 : (160 * 1024 / waves_per_simd) - 1024;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int blocks = 256 * waves_per_simd * 8, iters = 2000;
    unsigned* d; (void)hipMalloc(&d, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    kern<<<blocks, 256, lds>>>(d, 3u, 50);
    (void)hipEventRecord(e0);
    kern<<<blocks, 256, lds>>>(d, 3u, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double inst = (double)blocks * 4 * iters * 16.0;
    std::printf("%-8s waves/SIMD=%d  %7.3f ms  %5.2f SIMD-cycles per instruction (at 2.4 GHz)\n", name, waves_per_simd, ms,
                1024.0 * 2.4e9 * (ms * 1e-3) / inst);
    (void)hipFree(d);
}

int main()
{
    for (int w : {1, 2, 5, 8}) {
        run("allA", k_allA, w); run("allB", k_allB, w); run("inter", k_inter, w); run("pairs", k_pairs, w); run("group", k_group, w);
        run("f-inter", k_finter, w); run("f-group", k_fgroup, w);
    }
    return 0;
}
