// Feasibility probe for an on-chip (LDS) accumulator of the exact progressive probabilistic Hough transform ("next" 3):
// a page's 180 x numrho accumulator (int16: 4.3 MB for an A4 page) does not fit one CU's LDS, so G CUs would each own a slice of
// the angles and replay the same point sequence; what one of them finds (a cell reaching the threshold) every other one must
// learn before the sequence can go on - one all-to-all exchange per block of points and one per trigger.
//   exchange : G workgroups (one per CU) do N rounds of "publish an 8-byte {round, payload} granule (sc1 store), read all G
//              granules until every tag is current (sc1 loads)": microseconds per round, for G = 8, 16, 27, 32 and 1 or 8 groups
//              at once
//   votes    : ds_add_rtn_u32 into a 150 KB LDS accumulator at pseudo-random cells, all 64 lanes: votes per second per CU
//   hipcc -O3 --offload-arch=gfx950 -o ppht_onchip_probe ppht_onchip_probe.hip && ./ppht_onchip_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); std::exit(1); } } while (0)

// slots: [group][member] 8-byte granules, each on its own 64-byte line
__global__ void __launch_bounds__(256) k_exchange(unsigned long long* slots, int G, int rounds, unsigned long long* cycles_out)
{
    const int group = blockIdx.x / G, me = blockIdx.x % G;
    unsigned long long* mine = slots + ((size_t)group * G + me) * 8;
    const unsigned long long* base = slots + (size_t)group * G * 8;
    __shared__ unsigned s_acc;
    if (threadIdx.x == 0) s_acc = 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 1; r <= rounds; ++r) {
        if (threadIdx.x == 0)
            __hip_atomic_store(mine, ((unsigned long long)r << 32) | (unsigned)(me * 131 + r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // one wavefront polls: lane l reads member l's granule
        if (threadIdx.x < 64) {
            const int l = threadIdx.x;
            bool ok;
            unsigned sum;
            do {
                unsigned long long v = l < G ? __hip_atomic_load(base + (size_t)l * 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((unsigned long long)r << 32);
                ok = (unsigned)(v >> 32) >= (unsigned)r;
                sum = (unsigned)v;
            } while (__ballot(!ok) != 0ull);
            if (l == 0) s_acc += sum;
        }
        __syncthreads();   // the other wavefronts of the CU wait for the exchange, as the votes of the next block would
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cycles_out[blockIdx.x] = (t1 - t0) + (s_acc & 1u);
}

__global__ void __launch_bounds__(256) k_votes(unsigned* out, int iters)
{
    extern __shared__ unsigned acc[];   // 150 KB = 38400 dwords of packed int16 pairs
    const int n = 38400;
    for (int i = threadIdx.x; i < n; i += blockDim.x) acc[i] = 0x40004000u;
    __syncthreads();
    unsigned x = threadIdx.x * 2654435761u + blockIdx.x * 97u + 12345u, key = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            x = x * 1664525u + 1013904223u;
            const unsigned cell = (x >> 8) % (unsigned)(2 * n);
            const unsigned old = atomicAdd(&acc[cell >> 1], (cell & 1u) ? 0x10000u : 1u);   // ds_add_rtn_u32
            key = max(key, (cell & 1u) ? old >> 16 : old & 0xffffu);
        }
    }
    if (key == 0x12345u) out[0] = key;
    out[1 + blockIdx.x * blockDim.x + threadIdx.x] = key;
}

int main()
{
    unsigned long long *slots, *cycles;
    CK(hipMalloc(&slots, 256 * 64 * 8));
    CK(hipMalloc(&cycles, 256 * 8));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int rounds = 20000;
    for (int groups : {1, 8}) {
        for (int G : {8, 16, 27, 32}) {
            if (groups * G > 256) continue;
            CK(hipMemset(slots, 0, 256 * 64 * 8));
            hipLaunchKernelGGL(k_exchange, dim3(groups * G), dim3(256), 0, 0, slots, G, 200, cycles);
            CK(hipDeviceSynchronize());
            CK(hipMemset(slots, 0, 256 * 64 * 8));
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(k_exchange, dim3(groups * G), dim3(256), 0, 0, slots, G, rounds, cycles);
            CK(hipEventRecord(b));
            CK(hipEventSynchronize(b));
            float ms;
            CK(hipEventElapsedTime(&ms, a, b));
            std::printf("{\"probe\": \"exchange\", \"groups\": %d, \"cus_per_group\": %d, \"rounds\": %d, \"us_per_round\": %.3f}\n", groups, G, rounds,
                        ms * 1e3 / rounds);
            std::fflush(stdout);
        }
    }
    unsigned* out;
    CK(hipMalloc(&out, (1 + 256 * 256) * 4));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_votes), hipFuncAttributeMaxDynamicSharedMemorySize, 153600));
    for (int blocks : {1, 256}) {
        const int iters = 20000;
        hipLaunchKernelGGL(k_votes, dim3(blocks), dim3(256), 153600, 0, out, 100);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(k_votes, dim3(blocks), dim3(256), 153600, 0, out, iters);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        const double votes = (double)blocks * 256 * iters * 8;
        std::printf("{\"probe\": \"lds_votes\", \"workgroups\": %d, \"votes\": %.3g, \"ms\": %.3f, \"votes_per_s_per_cu\": %.4g}\n", blocks, votes, ms,
                    votes / blocks / (ms * 1e-3));
    }
    return 0;
}
