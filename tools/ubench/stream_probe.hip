// Hand-written ceiling kernels for bench.py (VERDICT r2 "next" 2a): what this box's memory system gives a plain read, a plain
// write and a plain copy of the benchmark's own page batch - 16 B per lane (dwordx4), grid-stride, four accesses in flight per
// lane, non-temporal.  NOT part of the product library: bench.py loads tools/ubench/libstream_probe.so only to print the
// measured ceilings beside the kernel's number.
//   hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o libstream_probe.so stream_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <vector>

typedef unsigned u4v __attribute__((ext_vector_type(4)));

template <int MODE>  // 0 read, 1 write, 2 copy
__global__ void __launch_bounds__(256) k_stream(const u4v* __restrict__ src, u4v* __restrict__ dst, size_t n16)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    u4v acc = {0, 0, 0, 0};
    for (; i + 3 * stride < n16; i += 4 * stride) {
        u4v v0, v1, v2, v3;
        if (MODE != 1) {
            v0 = __builtin_nontemporal_load(src + i); v1 = __builtin_nontemporal_load(src + i + stride);
            v2 = __builtin_nontemporal_load(src + i + 2 * stride); v3 = __builtin_nontemporal_load(src + i + 3 * stride);
        } else {
            v0 = v1 = v2 = v3 = u4v{(unsigned)i, 1u, 2u, 3u};
        }
        if (MODE == 0) {
            acc ^= v0 ^ v1 ^ v2 ^ v3;
        } else {
            __builtin_nontemporal_store(v0, dst + i); __builtin_nontemporal_store(v1, dst + i + stride);
            __builtin_nontemporal_store(v2, dst + i + 2 * stride); __builtin_nontemporal_store(v3, dst + i + 3 * stride);
        }
    }
    for (; i < n16; i += stride) {
        if (MODE == 0) acc ^= src[i];
        else dst[i] = MODE == 1 ? u4v{(unsigned)i, 1u, 2u, 3u} : src[i];
    }
    if (MODE == 0 && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) dst[0] = acc;  // keeps the loads alive
}

// mode 0 read / 1 write / 2 copy of `bytes` (multiple of 16) on the null stream; *ms_out = median of `reps` launches.
// dst is written in modes 1 and 2 (mode 0 needs a valid dst pointer but does not write it).  Returns a hipError_t.
extern "C" int prl_probe_stream(int mode, const void* src, void* dst, size_t bytes, int reps, float* ms_out)
{
    if (mode < 0 || mode > 2 || !src || !dst || !ms_out || reps < 1) return (int)hipErrorInvalidValue;
    hipEvent_t a, b;
    hipError_t e;
    if ((e = hipEventCreate(&a)) != hipSuccess) return (int)e;
    if ((e = hipEventCreate(&b)) != hipSuccess) { (void)hipEventDestroy(a); return (int)e; }
    const size_t n16 = bytes / 16;
    const dim3 grid(256 * 32), block(256);
    std::vector<float> t;
    for (int r = -1; r < reps && e == hipSuccess; ++r) {   // r = -1: warm-up
        (void)hipEventRecord(a, nullptr);
        if (mode == 0) hipLaunchKernelGGL(k_stream<0>, grid, block, 0, nullptr, (const u4v*)src, (u4v*)dst, n16);
        else if (mode == 1) hipLaunchKernelGGL(k_stream<1>, grid, block, 0, nullptr, (const u4v*)src, (u4v*)dst, n16);
        else hipLaunchKernelGGL(k_stream<2>, grid, block, 0, nullptr, (const u4v*)src, (u4v*)dst, n16);
        (void)hipEventRecord(b, nullptr);
        e = hipEventSynchronize(b);
        float ms = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, a, b);
        if (r >= 0) t.push_back(ms);
    }
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    if (e != hipSuccess) return (int)e;
    std::sort(t.begin(), t.end());
    *ms_out = t[t.size() / 2];
    return 0;
}

// ---- counter calibration (moved here from the product library in round 5) -------------------------------------------------
// Streams `bytes` with the access shape k_fused uses (8 B per lane, wave-contiguous) so that rocprofv3's FETCH_SIZE / WRITE_SIZE
// can be calibrated on a known byte count (MI355X_MICROARCH.md: FETCH_SIZE is only calibrated for 16 B/lane streams on gfx950).
// tools/calib_counters.py / tools/calib.sh.
__global__ void __launch_bounds__(256) k_calib_stream8(const uint2* __restrict__ src, uint2* __restrict__ dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint2 v = src[i];
        v.x ^= 0x01010101u;
        dst[i] = v;
    }
}

extern "C" int prl_probe_calib_stream8(const void* d_src, void* d_dst, size_t bytes, void* stream)
{
    hipLaunchKernelGGL(k_calib_stream8, dim3(256 * 16), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const uint2*>(d_src), static_cast<uint2*>(d_dst), bytes / 8);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
