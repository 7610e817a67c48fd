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

#define B_SUB(n) "v_sub_u32 %0, %0, %1"
#define B_LSHL(n) "v_lshlrev_b32 %0, 1, %0"
#define B_LSHR(n) "v_lshrrev_b32 %0, 1, %0"
#define B_ASHR(n) "v_ashrrev_i32 %0, 1, %0"
#define B_OR(n) "v_or_b32 %0, %0, %1"
#define B_XOR(n) "v_xor_b32 %0, %0, %1"
#define B_MINU(n) "v_min_u32 %0, %0, %1"
#define B_MAXI(n) "v_max_i32 %0, %0, %1"
#define B_MINF(n) "v_min_f32 %0, %0, %1"
#define B_MAXF(n) "v_max_f32 %0, %0, %1"
#define B_ADDF(n) "v_add_f32 %0, %0, %1"
#define B_MULF(n) "v_mul_f32 %0, %0, %1"
#define B_SUBF(n) "v_sub_f32 %0, %0, %1"
#define B_FMAC(n) "v_fmac_f32 %0, %1, %2"
#define B_MOV(n) "v_mov_b32 %0, %1"
#define B_CNDMASK(n) "v_cndmask_b32 %0, %0, %1, vcc"
#define B_CVTI(n) "v_cvt_f32_i32 %0, %0"
#define B_BFE(n) "v_bfe_u32 %0, %0, 3, 8"
#define B_LSHLADD(n) "v_lshl_add_u32 %0, %0, 1, %1"
#define B_ADDLSHL(n) "v_add_lshl_u32 %0, %0, %1, 1"
#define B_ANDOR(n) "v_and_or_b32 %0, %0, %1, %2"
#define B_OR3(n) "v_or3_b32 %0, %0, %1, %2"
#define B_PERM(n) "v_perm_b32 %0, %0, %1, %2"
#define B_MULU24(n) "v_mul_u32_u24 %0, %0, %1"
#define B_MULI24(n) "v_mul_i32_i24 %0, %0, %1"
#define B_ABSDIFF(n) "v_sad_u32 %0, %0, %1, %2"
#define B_MED3(n) "v_med3_i32 %0, %0, %1, %2"
#define B_MAX3(n) "v_max3_f32 %0, %0, %1, %2"
#define B_RCP(n) "v_rcp_f32 %0, %0"
#define B_RSQ(n) "v_rsq_f32 %0, %0"
#define B_CMP(n) "v_cmp_gt_u32 vcc, %0, %1"
#define B_FMAABS(n) "v_fma_f32 %0, -%0, |%1|, %2"
#define B_ADDCO(n) "v_add_co_u32 %0, vcc, %0, %1"
#define B_PKADD16(n) "v_pk_add_u16 %0, %0, %1"
#define B_PKMUL16(n) "v_pk_mul_lo_u16 %0, %0, %1"
#define B_PKFMAF16(n) "v_pk_fma_f16 %0, %0, %1, %2"
#define B_DOT2(n) "v_dot2_u32_u16 %0, %1, %2, %0"
#define B_LSHLOR(n) "v_lshl_or_b32 %0, %0, 1, %1"
#define B_ALIGNBIT(n) "v_alignbit_b32 %0, %0, %1, 3"
#define B_SDWAADD(n) "v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD"
#define B_CVTPKRTZ(n) "v_cvt_pkrtz_f16_f32 %0, %0, %1"
#define B_BCNT(n) "v_bcnt_u32_b32 %0, %0, %1"
#define B_MBCNT(n) "v_mbcnt_lo_u32_b32 %0, %0, %1"
DEFK(kb_sub, B_SUB) DEFK(kb_lshl, B_LSHL) DEFK(kb_lshr, B_LSHR) DEFK(kb_ashr, B_ASHR) DEFK(kb_or, B_OR) DEFK(kb_xor, B_XOR) DEFK(kb_minu, B_MINU) DEFK(kb_maxi, B_MAXI) DEFK(kb_minf, B_MINF) DEFK(kb_maxf, B_MAXF) DEFK(kb_addf, B_ADDF) DEFK(kb_mulf, B_MULF) DEFK(kb_subf, B_SUBF) DEFK(kb_fmac, B_FMAC) DEFK(kb_mov, B_MOV) DEFK(kb_cndmask, B_CNDMASK) DEFK(kb_cvti, B_CVTI) DEFK(kb_bfe, B_BFE) DEFK(kb_lshladd, B_LSHLADD) DEFK(kb_addlshl, B_ADDLSHL) DEFK(kb_andor, B_ANDOR) DEFK(kb_or3, B_OR3) DEFK(kb_perm, B_PERM) DEFK(kb_mulu24, B_MULU24) DEFK(kb_muli24, B_MULI24) DEFK(kb_absdiff, B_ABSDIFF) DEFK(kb_med3, B_MED3) DEFK(kb_max3, B_MAX3) DEFK(kb_rcp, B_RCP) DEFK(kb_rsq, B_RSQ) DEFK(kb_cmp, B_CMP) DEFK(kb_fmaabs, B_FMAABS) DEFK(kb_addco, B_ADDCO) DEFK(kb_pkadd16, B_PKADD16) DEFK(kb_pkmul16, B_PKMUL16) DEFK(kb_pkfmaf16, B_PKFMAF16) DEFK(kb_dot2, B_DOT2) DEFK(kb_lshlor, B_LSHLOR) DEFK(kb_alignbit, B_ALIGNBIT) DEFK(kb_sdwaadd, B_SDWAADD) DEFK(kb_cvtpkrtz, B_CVTPKRTZ) DEFK(kb_bcnt, B_BCNT) DEFK(kb_mbcnt, B_MBCNT) 

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
    run("v_sub_u32", kb_sub);
    run("v_lshlrev_b32", kb_lshl);
    run("v_lshrrev_b32", kb_lshr);
    run("v_ashrrev_i32", kb_ashr);
    run("v_or_b32", kb_or);
    run("v_xor_b32", kb_xor);
    run("v_min_u32", kb_minu);
    run("v_max_i32", kb_maxi);
    run("v_min_f32", kb_minf);
    run("v_max_f32", kb_maxf);
    run("v_add_f32", kb_addf);
    run("v_mul_f32", kb_mulf);
    run("v_sub_f32", kb_subf);
    run("v_fmac_f32", kb_fmac);
    run("v_mov_b32", kb_mov);
    run("v_cndmask_b32", kb_cndmask);
    run("v_cvt_f32_i32", kb_cvti);
    run("v_bfe_u32", kb_bfe);
    run("v_lshl_add_u32", kb_lshladd);
    run("v_add_lshl_u32", kb_addlshl);
    run("v_and_or_b32", kb_andor);
    run("v_or3_b32", kb_or3);
    run("v_perm_b32", kb_perm);
    run("v_mul_u32_u24", kb_mulu24);
    run("v_mul_i32_i24", kb_muli24);
    run("v_sad_u32", kb_absdiff);
    run("v_med3_i32", kb_med3);
    run("v_max3_f32", kb_max3);
    run("v_rcp_f32", kb_rcp);
    run("v_rsq_f32", kb_rsq);
    run("v_cmp_gt_u32", kb_cmp);
    run("v_fma_f32(fmaabs)", kb_fmaabs);
    run("v_add_co_u32", kb_addco);
    run("v_pk_add_u16", kb_pkadd16);
    run("v_pk_mul_lo_u16", kb_pkmul16);
    run("v_pk_fma_f16", kb_pkfmaf16);
    run("v_dot2_u32_u16", kb_dot2);
    run("v_lshl_or_b32", kb_lshlor);
    run("v_alignbit_b32", kb_alignbit);
    run("v_add_u32_sdwa(sdwaadd)", kb_sdwaadd);
    run("v_cvt_pkrtz_f16_f32", kb_cvtpkrtz);
    run("v_bcnt_u32_b32", kb_bcnt);
    run("v_mbcnt_lo_u32_b32", kb_mbcnt);
    return 0;
}
