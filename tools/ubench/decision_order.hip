// Micro-benchmark: the decision stage of k_fused (per pixel: v_sub_f32, v_cvt_f32_u32, v_mul_f32, v_fma_f32, v_sqrt_f32,
// v_fma_f32, v_cvt_f32_ubyte, v_fma_f32, v_fma_f32, v_min3_f32 (x1), v_cvt_pk_u8_f32) for 8 pixels, emitted
//   (a) pixel by pixel (the order the source is written in), or
//   (b) phase by phase across the 8 pixels, float 2-cycle ops interleaved with the conversions (tools/ubench/pair_matrix.hip
//       says those overlap), square roots back to back.
// Same instructions, same dependencies; 5 wavefronts per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
extern __shared__ unsigned char dyn_lds[];

#define DECL unsigned s0=t,s1=t+1,s2=t+2,s3=t+3,s4=t+4,s5=t+5,s6=t+6,s7=t+7, q0=t+8,q1=t+9,q2=t+10,q3=t+11,q4=t+12,q5=t+13,q6=t+14,q7=t+15; \
  float k0,k1,k2,k3,k4,k5,k6,k7, p0,p1,p2,p3,p4,p5,p6,p7, m0=1e30f, m1=1e30f; unsigned lo=0, hi=0;
#define SUBF(s) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(s) : "v"(c1));
#define CVTQ(q) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(q));
#define MULS(k, s) asm volatile("v_mul_f32 %0, %1, %1" : "=v"(k) : "v"(s));
#define FMAK(k, q) asm volatile("v_fma_f32 %0, %1, %2, -%0" : "+v"(k) : "v"(q), "v"(c2));
#define SQRT(k) asm volatile("v_sqrt_f32 %0, %0" : "+v"(k));
#define FMAD(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(k) : "v"(c1), "v"(c2));
#define CVTB(p, n) asm volatile("v_cvt_f32_ubyte" #n " %0, %1" : "=v"(p) : "v"(pv));
#define FMAP(p) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(p) : "v"(c2), "v"(c1));
#define FMAT(p, s, k) asm volatile("v_fma_f32 %0, -%1, %2, %0" : "+v"(p) : "v"(s), "v"(k));
#define MIN3(m, a, b) asm volatile("v_min3_f32 %0, %0, |%1|, |%2|" : "+v"(m) : "v"(a), "v"(b));
#define PKU8(d, p, n) asm volatile("v_cvt_pk_u8_f32 %0, %1, " #n ", %0" : "+v"(d) : "v"(p));

#define PIXEL(s,q,k,p,n,d) SUBF(s) CVTQ(q) MULS(k,s) FMAK(k,q) SQRT(k) FMAD(k) CVTB(p,n) FMAP(p) FMAT(p,s,k) PKU8(d,p,n)

__global__ void __launch_bounds__(256) k_pixelwise(unsigned* out, float c1, float c2, unsigned pv, int iters)
{
    if (pv == 0xdeadbeefu) dyn_lds[threadIdx.x] = 1;
    const unsigned t = threadIdx.x;
    DECL
    for (int i = 0; i < iters; ++i) {
        PIXEL(s0,q0,k0,p0,0,lo) PIXEL(s1,q1,k1,p1,1,lo) MIN3(m0,p0,p1) MIN3(m1,k0,k1)
        PIXEL(s2,q2,k2,p2,2,lo) PIXEL(s3,q3,k3,p3,3,lo) MIN3(m0,p2,p3) MIN3(m1,k2,k3)
        PIXEL(s4,q4,k4,p4,0,hi) PIXEL(s5,q5,k5,p5,1,hi) MIN3(m0,p4,p5) MIN3(m1,k4,k5)
        PIXEL(s6,q6,k6,p6,2,hi) PIXEL(s7,q7,k7,p7,3,hi) MIN3(m0,p6,p7) MIN3(m1,k6,k7)
    }
    out[blockIdx.x * blockDim.x + t] = lo + hi + __float_as_uint(m0) + __float_as_uint(m1) + s0+s1+s2+s3+s4+s5+s6+s7+q0+q1+q2+q3+q4+q5+q6+q7;
}

__global__ void __launch_bounds__(256) k_phased(unsigned* out, float c1, float c2, unsigned pv, int iters)
{
    if (pv == 0xdeadbeefu) dyn_lds[threadIdx.x] = 1;
    const unsigned t = threadIdx.x;
    DECL
    for (int i = 0; i < iters; ++i) {
        // conversions of Q with the float subtractions of S in their shadow
        CVTQ(q0) SUBF(s0) CVTQ(q1) SUBF(s1) CVTQ(q2) SUBF(s2) CVTQ(q3) SUBF(s3) CVTQ(q4) SUBF(s4) CVTQ(q5) SUBF(s5) CVTQ(q6) SUBF(s6) CVTQ(q7) SUBF(s7)
        // byte conversions with S*S and the K fma in their shadow
        CVTB(p0,0) MULS(k0,s0) CVTB(p1,1) MULS(k1,s1) CVTB(p2,2) MULS(k2,s2) CVTB(p3,3) MULS(k3,s3)
        CVTB(p4,0) MULS(k4,s4) CVTB(p5,1) MULS(k5,s5) CVTB(p6,2) MULS(k6,s6) CVTB(p7,3) MULS(k7,s7)
        FMAK(k0,q0) FMAK(k1,q1) FMAK(k2,q2) FMAK(k3,q3) FMAK(k4,q4) FMAK(k5,q5) FMAK(k6,q6) FMAK(k7,q7)
        FMAP(p0) FMAP(p1) FMAP(p2) FMAP(p3) FMAP(p4) FMAP(p5) FMAP(p6) FMAP(p7)
        // square roots back to back
        SQRT(k0) SQRT(k1) SQRT(k2) SQRT(k3) SQRT(k4) SQRT(k5) SQRT(k6) SQRT(k7)
        FMAD(k0) FMAD(k1) FMAD(k2) FMAD(k3) FMAD(k4) FMAD(k5) FMAD(k6) FMAD(k7)
        // final fma with the pack / min3 in between
        FMAT(p0,s0,k0) FMAT(p1,s1,k1) PKU8(lo,p0,0) FMAT(p2,s2,k2) PKU8(lo,p1,1) FMAT(p3,s3,k3) MIN3(m0,p0,p1) FMAT(p4,s4,k4) PKU8(lo,p2,2)
        FMAT(p5,s5,k5) PKU8(lo,p3,3) FMAT(p6,s6,k6) MIN3(m0,p2,p3) FMAT(p7,s7,k7) PKU8(hi,p4,0) PKU8(hi,p5,1) MIN3(m0,p4,p5) PKU8(hi,p6,2) PKU8(hi,p7,3) MIN3(m0,p6,p7)
        MIN3(m1,k0,k1) MIN3(m1,k2,k3) MIN3(m1,k4,k5) MIN3(m1,k6,k7)
    }
    out[blockIdx.x * blockDim.x + t] = lo + hi + __float_as_uint(m0) + __float_as_uint(m1) + s0+s1+s2+s3+s4+s5+s6+s7+q0+q1+q2+q3+q4+q5+q6+q7;
}

template <typename K> void run(const char* name, K kern)
{
    const int wps = 5, lds = (160 * 1024 / wps) - 1024;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int blocks = 256 * wps * 4, iters = 2000;
    unsigned* d; (void)hipMalloc(&d, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    kern<<<blocks, 256, lds>>>(d, 1.5f, 2.5f, 0x11223344u, 20);
    (void)hipEventRecord(e0); kern<<<blocks, 256, lds>>>(d, 1.5f, 2.5f, 0x11223344u, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); (void)hipFree(d);
    std::printf("%-10s %7.3f ms  %6.1f SIMD-cycles per 8-pixel decision (88 instructions; cost-table sum 304)\n", name, ms,
                1024.0 * 2.4e9 * (ms * 1e-3) / ((double)blocks * 4 * iters));
}
int main() { run("pixelwise", k_pixelwise); run("phased", k_phased); run("pixelwise", k_pixelwise); run("phased", k_phased); return 0; }
