// Micro-benchmark: do gfx950's typed buffer loads (buffer_load_format_*, 8-bit formats converted to float by the
// texture-address unit) work, do they need element alignment, and what do they cost next to a plain 8-byte load?
//   hipcc -O3 --offload-arch=gfx950 -o typed_load typed_load.hip && ./typed_load
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ float buf_load_fmt_x(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.format.f32");
__device__ f32x2 buf_load_fmt_xy(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.format.v2f32");
__device__ f32x4 buf_load_fmt_xyzw(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.format.v4f32");

// gfx9 buffer resource, dword 3: DST_SEL_X/Y/Z/W [11:0], NUM_FORMAT [14:12], DATA_FORMAT [18:15]
constexpr int kSelXYZW = 4 | (5 << 3) | (6 << 6) | (7 << 9);
constexpr int kNumUscaled = 2, kNumUint = 4;
constexpr int kFmt8 = 1, kFmt8_8 = 3, kFmt8_8_8_8 = 10;

__device__ __forceinline__ i32x4 make_rsrc(const void* p, int dfmt, int nfmt)
{
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    i32x4 r;
    r.x = (int)(unsigned)a;
    r.y = (int)((a >> 32) & 0xffffu);  // stride 0: raw buffer, offsets in bytes
    r.z = (int)0xffffffffu;            // num_records (bytes)
    r.w = kSelXYZW | (nfmt << 12) | (dfmt << 15);
    return r;
}

// every lane sums its 8 consecutive bytes; mode 0: one 8-byte load + unpack, 1: 8 x format_x (8 bit),
// 2: 4 x format_xy (8_8), 3: 2 x format_xyzw (8_8_8_8)
template <int MODE>
__global__ void __launch_bounds__(256) k_sum8(const unsigned char* __restrict__ src, float* __restrict__ out, size_t n8, int shift)
{
    const i32x4 r1 = make_rsrc(src, kFmt8, kNumUscaled), r2 = make_rsrc(src, kFmt8_8, kNumUscaled),
                r4 = make_rsrc(src, kFmt8_8_8_8, kNumUscaled);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        const int off = (int)(i * 8) + shift;
        float s = 0.f;
        if (MODE == 0) {
            uint2 v;
            __builtin_memcpy(&v, src + off, 8);
#pragma unroll
            for (int c = 0; c < 4; ++c) s += (float)((v.x >> (8 * c)) & 0xffu) * (float)(c + 1);
#pragma unroll
            for (int c = 0; c < 4; ++c) s += (float)((v.y >> (8 * c)) & 0xffu) * (float)(c + 5);
        } else if (MODE == 1) {
#pragma unroll
            for (int c = 0; c < 8; ++c) s += buf_load_fmt_x(r1, off + c, 0, 0) * (float)(c + 1);
        } else if (MODE == 2) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f32x2 v = buf_load_fmt_xy(r2, off + 2 * c, 0, 0);
                s += v.x * (float)(2 * c + 1) + v.y * (float)(2 * c + 2);
            }
        } else {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const f32x4 v = buf_load_fmt_xyzw(r4, off + 4 * c, 0, 0);
                s += v.x * (float)(4 * c + 1) + v.y * (float)(4 * c + 2) + v.z * (float)(4 * c + 3) + v.w * (float)(4 * c + 4);
            }
        }
        out[i] = s;
    }
}

// cache-resident rate: every lane reads `reps` different 8-byte groups of a 4 MiB buffer (L2 hits), no stores in the loop
template <int MODE>
__global__ void __launch_bounds__(256) k_rate(const unsigned char* __restrict__ src, float* __restrict__ out, int reps)
{
    const i32x4 r1 = make_rsrc(src, kFmt8, kNumUscaled), r4 = make_rsrc(src, kFmt8_8_8_8, kNumUscaled);
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    float s = 0.f;
    for (int rep = 0; rep < reps; ++rep) {
        const int off = (int)(((tid + (unsigned)rep * 4099u * 64u) & ((1u << 19) - 1)) * 8u);
        if (MODE == 0) {
            uint2 v;
            __builtin_memcpy(&v, src + off, 8);
            s += __uint_as_float(v.x) + __uint_as_float(v.y);
        } else if (MODE == 1) {
#pragma unroll
            for (int c = 0; c < 8; ++c) s += buf_load_fmt_x(r1, off + c, 0, 0);
        } else {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const f32x4 v = buf_load_fmt_xyzw(r4, off + 4 * c, 0, 0);
                s += (v.x + v.y) + (v.z + v.w);
            }
        }
    }
    out[tid] = s;
}

template <int MODE>
static void rate(const char* name, const unsigned char* d, float* dout)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int blocks = 256 * 8, reps = 4096;
    k_rate<MODE><<<blocks, 256>>>(d, dout, 64);
    (void)hipEventRecord(e0);
    k_rate<MODE><<<blocks, 256>>>(d, dout, reps);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)blocks * 256 * 8 * reps;
    std::printf("%-28s L2-resident: %7.3f ms  %8.1f GB/s of pixels  = %6.2f pixel bytes per CU per clock (2.4 GHz)\n", name, ms,
                bytes / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / 256 / 2.4e9);
}

template <int MODE>
static void run(const char* name, const unsigned char* d, float* dout, const std::vector<unsigned char>& h, size_t n8, int shift)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int blocks = 256 * 20;
    k_sum8<MODE><<<blocks, 256>>>(d, dout, n8, shift);
    (void)hipEventRecord(e0);
    for (int it = 0; it < 5; ++it) k_sum8<MODE><<<blocks, 256>>>(d, dout, n8, shift);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    std::vector<float> o(4096);
    (void)hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (size_t i = 0; i < o.size(); ++i) {
        float want = 0.f;
        for (int c = 0; c < 8; ++c) want += (float)h[i * 8 + shift + c] * (float)(c + 1);
        if (want != o[i]) ++bad;
    }
    std::printf("%-28s shift=%d  %7.3f ms  %7.1f GB/s read  mismatches in first 4096 lanes: %zu\n", name, shift, ms,
                (double)n8 * 8 / (ms * 1e-3) / 1e9, bad);
}

int main()
{
    const size_t n8 = (size_t)1 << 27;  // 1 GiB of bytes
    std::vector<unsigned char> h(n8 * 8 + 64);
    unsigned x = 12345;
    for (size_t i = 0; i < (size_t)1 << 20; ++i) { x = x * 1664525u + 1013904223u; h[i] = (unsigned char)(x >> 24); }
    for (size_t i = (size_t)1 << 20; i < h.size(); ++i) h[i] = h[i & ((1u << 20) - 1)];
    unsigned char* d;
    float* dout;
    (void)hipMalloc(&d, h.size());
    (void)hipMalloc(&dout, n8 * 4);
    (void)hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice);
    rate<0>("8-byte load", d, dout);
    rate<1>("8 x format_x", d, dout);
    rate<3>("2 x format_xyzw", d, dout);
    for (int shift = 0; shift < 4; shift += 3) {
        run<0>("8-byte load + cvt_ubyte", d, dout, h, n8, shift);
        run<1>("8 x format_x (8, uscaled)", d, dout, h, n8, shift);
        run<2>("4 x format_xy (8_8)", d, dout, h, n8, shift);
        run<3>("2 x format_xyzw (8_8_8_8)", d, dout, h, n8, shift);
    }
    return 0;
}
