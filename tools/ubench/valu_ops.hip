// Micro-benchmark: SIMD cycles per wave64 instruction for the instruction kinds k_fused uses (gfx950).
// 8 independent registers per kind, inline asm so the measured instruction is exactly the named one.
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

#define DEFK(NAME, ASM)                                                                                   \
    __global__ void __launch_bounds__(256) NAME(unsigned* out, unsigned a, unsigned b, int iters)         \
    {                                                                                                     \
        unsigned r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6,  \
                 r7 = r0 + 7;                                                                             \
        for (int i = 0; i < iters; ++i) {                                                                 \
            _Pragma("unroll") for (int u = 0; u < 8; ++u)                                                 \
            {                                                                                             \
                asm volatile(ASM(0) : "+v"(r0) : "v"(a), "v"(b));                                         \
                asm volatile(ASM(1) : "+v"(r1) : "v"(a), "v"(b));                                         \
                asm volatile(ASM(2) : "+v"(r2) : "v"(a), "v"(b));                                         \
                asm volatile(ASM(3) : "+v"(r3) : "v"(a), "v"(b));                                         \
                asm volatile(ASM(4) : "+v"(r4) : "v"(a), "v"(b));                                         \
                asm volatile(ASM(5) : "+v"(r5) : "v"(a), "v"(b));                                         \
                asm volatile(ASM(6) : "+v"(r6) : "v"(a), "v"(b));                                         \
                asm volatile(ASM(7) : "+v"(r7) : "v"(a), "v"(b));                                         \
            }                                                                                             \
        }                                                                                                 \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;               \
    }

#define A_ADD(n) "v_add_u32 %0, %0, %1"
#define A_ADD3(n) "v_add3_u32 %0, %0, %1, %2"
#define A_MAD24(n) "v_mad_i32_i24 %0, %0, %1, %2"
#define A_MADU24(n) "v_mad_u32_u24 %0, %0, %1, %2"
#define A_SDWA(n) "v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1"
#define A_DPP(n) "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
#define A_CVTU(n) "v_cvt_f32_u32 %0, %0"
#define A_CVTB(n) "v_cvt_f32_ubyte1 %0, %1"
#define A_SQRT(n) "v_sqrt_f32 %0, %0"
#define A_PKU8(n) "v_cvt_pk_u8_f32 %0, %1, %2, %0"
#define A_MIN3(n) "v_min3_f32 %0, %0, %1, %2"
#define A_FMA(n) "v_fma_f32 %0, %0, %1, %2"
#define A_MULLO(n) "v_mul_lo_u32 %0, %0, %1"
#define A_BPERM(n) "ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)"
#define A_BPERM_NW(n) "ds_bpermute_b32 %0, %1, %0"
#define A_AND(n) "v_and_b32 %0, %0, %1"
#define A_DOT4(n) "v_dot4_u32_u8 %0, %1, %2, %0"
#define A_ALIGNB(n) "v_alignbyte_b32 %0, %0, %1, %2"
#define A_MAD64(n) "v_mad_u64_u32 v[10:11], s[10:11], %0, %1, v[10:11]"

DEFK(k_add, A_ADD) DEFK(k_add3, A_ADD3) DEFK(k_mad24, A_MAD24) DEFK(k_madu24, A_MADU24) DEFK(k_sdwa, A_SDWA)
DEFK(k_dpp, A_DPP) DEFK(k_cvtu, A_CVTU) DEFK(k_cvtb, A_CVTB) DEFK(k_sqrt, A_SQRT) DEFK(k_pku8, A_PKU8)
DEFK(k_min3, A_MIN3) DEFK(k_fma, A_FMA) DEFK(k_mullo, A_MULLO) DEFK(k_bperm, A_BPERM) DEFK(k_bperm_nw, A_BPERM_NW)
DEFK(k_and, A_AND) DEFK(k_dot4, A_DOT4) DEFK(k_alignb, A_ALIGNB)

template <typename K> void run(const char* name, K kern)
{
    const int blocks = 2048, iters = 4000;
    unsigned* d; (void)hipMalloc(&d, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    kern<<<blocks, 256>>>(d, 3u, 5u, 50);
    (void)hipEventRecord(e0);
    kern<<<blocks, 256>>>(d, 3u, 5u, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double inst = (double)blocks * 4 * iters * 64.0;
    std::printf("%-14s %8.3f ms  %6.2f SIMD-cycles per wave-instruction (assuming 2.4 GHz; divide by ~1.1 at the real clock)\n",
                name, ms, 1024.0 * 2.4e9 * (ms * 1e-3) / inst);
    (void)hipFree(d);
}

int main()
{
    run("v_add_u32", k_add); run("v_add3_u32", k_add3); run("v_and_b32", k_and); run("v_mad_i32_i24", k_mad24);
    run("v_mad_u32_u24", k_madu24); run("v_sub_u32_sdwa", k_sdwa); run("v_add_u32_dpp", k_dpp);
    run("v_cvt_f32_u32", k_cvtu); run("v_cvt_f32_ubyte", k_cvtb); run("v_sqrt_f32", k_sqrt);
    run("v_cvt_pk_u8_f32", k_pku8); run("v_min3_f32", k_min3); run("v_fma_f32", k_fma); run("v_mul_lo_u32", k_mullo);
    run("v_dot4_u32_u8", k_dot4); run("v_alignbyte", k_alignb); run("ds_bpermute+wait", k_bperm); run("ds_bpermute", k_bperm_nw);
    return 0;
}
