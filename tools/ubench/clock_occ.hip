// Micro-benchmark: (1) the core clock the chip holds under a VALU-only load (s_memtime vs s_memrealtime),
// (2) SIMD cycles per wave64 instruction of a few kinds at 1, 2, 4 and 8 wavefronts per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>

#define BODY(ASM)                                                                                        \
    unsigned r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, \
             r7 = r0 + 7;                                                                                \
    const unsigned long long c0 = __builtin_readcyclecounter(), t0 = wall_clock64();                   \
    for (int i = 0; i < iters; ++i) {                                                                    \
        _Pragma("unroll") for (int u = 0; u < 8; ++u)                                                    \
        {                                                                                                \
            asm volatile(ASM : "+v"(r0) : "v"(a), "v"(b));                                               \
            asm volatile(ASM : "+v"(r1) : "v"(a), "v"(b));                                               \
            asm volatile(ASM : "+v"(r2) : "v"(a), "v"(b));                                               \
            asm volatile(ASM : "+v"(r3) : "v"(a), "v"(b));                                               \
            asm volatile(ASM : "+v"(r4) : "v"(a), "v"(b));                                               \
            asm volatile(ASM : "+v"(r5) : "v"(a), "v"(b));                                               \
            asm volatile(ASM : "+v"(r6) : "v"(a), "v"(b));                                               \
            asm volatile(ASM : "+v"(r7) : "v"(a), "v"(b));                                               \
        }                                                                                                \
    }                                                                                                    \
    const unsigned long long c1 = __builtin_readcyclecounter(), t1 = wall_clock64();                   \
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;                  \
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = t1 - t0; }

extern __shared__ unsigned char dyn_lds[];
#define DEFK(NAME, ASM)                                                                                  \
    __global__ void __launch_bounds__(256) NAME(unsigned* out, unsigned long long* clk, unsigned a, unsigned b, int iters) \
    { if (a == 0xdeadbeefu) dyn_lds[threadIdx.x] = 1; BODY(ASM) }

DEFK(k_add, "v_add_u32 %0, %0, %1")
DEFK(k_fma, "v_fma_f32 %0, %0, %1, %2")
DEFK(k_dot4, "v_dot4_u32_u8 %0, %1, %2, %0")
DEFK(k_lshl, "v_lshlrev_b32 %0, 1, %0")
DEFK(k_min, "v_min_i32 %0, %0, %1")
DEFK(k_fmamix, "v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]")
DEFK(k_fmamix2, "v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,0]")
// packed float32: operates on register PAIRS; the harness variables are 32-bit, so pairs are formed from %0 and a copy
__global__ void __launch_bounds__(256) k_pkfma(unsigned* out, unsigned long long* clk, unsigned a, unsigned b, int iters)
{
    if (a == 0xdeadbeefu) dyn_lds[threadIdx.x] = 1;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 r0 = {(float)threadIdx.x, 1.f}, r1 = r0 + 1.f, r2 = r0 + 2.f, r3 = r0 + 3.f, r4 = r0 + 4.f, r5 = r0 + 5.f, r6 = r0 + 6.f, r7 = r0 + 7.f;
    const f2 fa = {(float)a, (float)a}, fb = {(float)b * 1e-3f, (float)b * 1e-3f};
    const unsigned long long c0 = __builtin_readcyclecounter(), t0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(r0) : "v"(fb), "v"(fa));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(r1) : "v"(fb), "v"(fa));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(r2) : "v"(fb), "v"(fa));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(r3) : "v"(fb), "v"(fa));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(r4) : "v"(fb), "v"(fa));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(r5) : "v"(fb), "v"(fa));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(r6) : "v"(fb), "v"(fa));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(r7) : "v"(fb), "v"(fa));
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), t1 = wall_clock64();
    const f2 sum = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned)(sum.x + sum.y);
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = t1 - t0; }
}
DEFK(k_mix, "v_dot4_u32_u8 %0, %1, %2, %0\n\tv_add_u32 %0, %0, %1\n\tv_lshlrev_b32 %0, 1, %0")

template <typename K> void run(const char* name, K kern, int waves_per_simd, int ninst)
{
    // one 256-thread block = one wavefront on each SIMD of a CU; LDS sized so that exactly `waves_per_simd` blocks fit
    const int lds = waves_per_simd >= 8 ? 0 : (160 * 1024 / waves_per_simd) - 1024;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int blocks = 256 * waves_per_simd * 4, iters = 2000;
    unsigned* d; (void)hipMalloc(&d, (size_t)blocks * 256 * 4);
    unsigned long long* clk; (void)hipMalloc(&clk, 16);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    kern<<<blocks, 256, lds>>>(d, clk, 3u, 5u, 50);
    (void)hipEventRecord(e0);
    kern<<<blocks, 256, lds>>>(d, clk, 3u, 5u, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double ghz = (double)h[0] / ((double)h[1] / 100e6) / 1e9;   // s_memrealtime ticks at 100 MHz
    const double inst_per_wave = (double)iters * 64.0 * ninst;
    std::printf("%-10s waves/SIMD=%d  %7.3f ms  clock %.3f GHz  %6.2f cycles per wave-instruction per wave (block 0), %6.2f SIMD-cycles per instruction overall\n",
                name, waves_per_simd, ms, ghz, (double)h[0] / inst_per_wave,
                1024.0 * ghz * 1e9 * (ms * 1e-3) / ((double)blocks * 4 * inst_per_wave));
    (void)hipFree(d); (void)hipFree(clk);
}

int main()
{
    for (int w : {1, 2, 4, 5, 8}) {
        run("v_add_u32", k_add, w, 1); run("v_fma_f32", k_fma, w, 1); run("v_dot4", k_dot4, w, 1);
        run("v_fma_mix(h,f,f)", k_fmamix, w, 1); run("v_fma_mix(hh,h,f)", k_fmamix2, w, 1);
        run("v_pk_fma_f32", k_pkfma, w, 1); run("v_lshlrev", k_lshl, w, 1); run("v_min_i32", k_min, w, 1); run("dot+add+shl", k_mix, w, 3);
    }
    return 0;
}
