// Micro-benchmark (VERDICT r2 "next" 2a/2b/2c): what the memory system gives the headline workload's ACCESS SHAPES,
// with (almost) no arithmetic.  256 x 4096 x 4096 u8 pages in, the same out, like bench.py's batch.
//   flat   : hand-written read-only / write-only / copy kernels, 16 B per lane, grid-stride, 4 accesses in flight
//   walk   : k_fused's shape - one wavefront walks down a strip of 64*VEC columns, per output row it fetches the
//            entering row (y+30), the leaving row (y) and the compared row (y+15) and stores one mask row;
//            VEC = 8 or 16 bytes per lane, DEPTH rows of software prefetch, 1..3 global streams,
//            ring 0 = every stream from global memory (what k_fused does), 1 = leaving + compared rows from an LDS
//            ring filled by the entering stream, 2 = the same ring in registers (unrolled RINGN rows)
//   hipcc -O3 --offload-arch=gfx950 -o stream_ceiling stream_ceiling.hip && ./stream_ceiling [pages]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));
template <int VEC> struct Vec;
template <> struct Vec<8> { typedef u2v T; };
template <> struct Vec<16> { typedef u4v T; };

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); std::exit(1); } } while (0)

// ---- flat kernels -------------------------------------------------------------------------------------------------
template <int MODE, bool NT>  // 0 read, 1 write, 2 copy
__global__ void __launch_bounds__(256) k_flat(const u4v* __restrict__ src, u4v* __restrict__ dst, size_t n16)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    u4v acc = {0, 0, 0, 0};
    for (; i + 3 * stride < n16; i += 4 * stride) {
        u4v v0, v1, v2, v3;
        if (MODE != 1) {
            if (NT) {
                v0 = __builtin_nontemporal_load(src + i); v1 = __builtin_nontemporal_load(src + i + stride);
                v2 = __builtin_nontemporal_load(src + i + 2 * stride); v3 = __builtin_nontemporal_load(src + i + 3 * stride);
            } else {
                v0 = src[i]; v1 = src[i + stride]; v2 = src[i + 2 * stride]; v3 = src[i + 3 * stride];
            }
        } else {
            v0 = v1 = v2 = v3 = u4v{(unsigned)i, 1u, 2u, 3u};
        }
        if (MODE == 0) {
            acc ^= v0 ^ v1 ^ v2 ^ v3;
        } else if (NT) {
            __builtin_nontemporal_store(v0, dst + i); __builtin_nontemporal_store(v1, dst + i + stride);
            __builtin_nontemporal_store(v2, dst + i + 2 * stride); __builtin_nontemporal_store(v3, dst + i + 3 * stride);
        } else {
            dst[i] = v0; dst[i + stride] = v1; dst[i + 2 * stride] = v2; dst[i + 3 * stride] = v3;
        }
    }
    for (; i < n16; i += stride) {
        if (MODE == 0) acc ^= src[i];
        else dst[i] = MODE == 1 ? u4v{(unsigned)i, 1u, 2u, 3u} : src[i];
    }
    if (MODE == 0 && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) dst[0] = acc;  // keeps the loads alive
}

// ---- row walk -----------------------------------------------------------------------------------------------------
constexpr int kWin = 30;  // w - 1

template <typename T> __device__ __forceinline__ T vld(const uint8_t* p) { T v; __builtin_memcpy(&v, p, sizeof(T)); return v; }
template <typename T> __device__ __forceinline__ void vst_nt(uint8_t* p, T v) { __builtin_nontemporal_store(v, reinterpret_cast<T*>(p)); }

struct WalkArgs {
    const uint8_t* in;
    uint8_t* out;
    int W, H;
    size_t page_stride;
    int rps, n_strips, n_segs;
    unsigned total;
};

template <int VEC, int DEPTH, int NS, int RING, int RINGN>
__global__ void __launch_bounds__(64) k_walk(WalkArgs a)
{
    typedef typename Vec<VEC>::T T;
    const unsigned nb = gridDim.x;
    const unsigned wid = __builtin_amdgcn_readfirstlane((blockIdx.x & 7u) * (nb >> 3) + (blockIdx.x >> 3));
    if (wid >= a.total) return;
    const int lane = threadIdx.x;
    const int per_page = a.n_strips * a.n_segs;
    const int page = (int)(wid / (unsigned)per_page);
    const int rem = (int)(wid - (unsigned)page * (unsigned)per_page);
    const int seg = rem / a.n_strips, strip = rem - seg * a.n_strips;
    const int col = strip * 64 * VEC + lane * VEC;
    const int ys = seg * a.rps, ye = min(ys + a.rps, a.H);
    const uint8_t* base = a.in + (size_t)page * a.page_stride + col;
    uint8_t* obase = a.out + (size_t)page * a.page_stride + col;
    const size_t step = (size_t)a.W;
    auto ld = [&](int row) -> T { return vld<T>(base + (size_t)min(row, a.H - 1) * step); };

    T state = T(0);
    if constexpr (RING == 0) {
        // warm-up: w-1 rows of the entering stream
#pragma unroll 4
        for (int r = 0; r < kWin; ++r) state += ld(ys + r);
        T qe[DEPTH], ql[DEPTH], qc[DEPTH];
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) {
            qe[k] = ld(ys + kWin + k);
            if (NS >= 2) ql[k] = ld(ys + k);
            if (NS >= 3) qc[k] = ld(ys + kWin / 2 + k);
        }
        for (int y = ys; y < ye; y += DEPTH) {
#pragma unroll
            for (int k = 0; k < DEPTH; ++k) {
                const int row = y + k;
                if (row >= ye) break;
                const T e = qe[k], l = NS >= 2 ? ql[k] : e ^ T(0x01010101u), c = NS >= 3 ? qc[k] : l + T(0x00010001u);
                qe[k] = ld(row + kWin + DEPTH);
                if (NS >= 2) ql[k] = ld(row + DEPTH);
                if (NS >= 3) qc[k] = ld(row + kWin / 2 + DEPTH);
                state += e - l;
                vst_nt<T>(obase + (size_t)row * step, (state ^ c) & T(0x80808080u));
            }
        }
    } else if constexpr (RING == 1) {
        static_assert(kWin % DEPTH == 0, "DEPTH must divide the window height");
        __shared__ T ring[RINGN][64];
        T qe[DEPTH];
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) qe[k] = ld(ys + k);
        for (int r = 0; r < kWin; r += DEPTH) {  // rows ys .. ys+29 into the ring and the state
#pragma unroll
            for (int k = 0; k < DEPTH; ++k) {
                const T e = qe[k];
                qe[k] = ld(ys + r + k + DEPTH);
                ring[(ys + r + k) % RINGN][lane] = e;
                state += e;
            }
        }
        for (int y = ys; y < ye; y += DEPTH) {
#pragma unroll
            for (int k = 0; k < DEPTH; ++k) {
                const int row = y + k;
                const T e = qe[k];
                qe[k] = ld(row + kWin + DEPTH);
                const T l = ring[row % RINGN][lane], c = ring[(row + kWin / 2) % RINGN][lane];
                ring[(row + kWin) % RINGN][lane] = e;
                state += e - l;
                if (row < ye) vst_nt<T>(obase + (size_t)row * step, (state ^ c) & T(0x80808080u));
            }
        }
    } else {
        // register ring: RINGN rows, unrolled; rps is a multiple of RINGN, so row % RINGN is a compile-time constant
        constexpr int PF = RINGN - kWin - 1;  // rows of prefetch the ring has room for
        static_assert(PF >= 1, "ring too small");
        T ring[RINGN];
#pragma unroll
        for (int k = 0; k < kWin + PF; ++k) {
            ring[k] = ld(ys + k);
            if (k < kWin) state += ring[k];
        }
        for (int y = ys; y < ye; y += RINGN) {
#pragma unroll
            for (int k = 0; k < RINGN; ++k) {
                const int row = y + k;
                const T e = ring[(k + kWin) % RINGN], l = ring[k], c = ring[(k + kWin / 2) % RINGN];
                state += e - l;
                if (row < ye) vst_nt<T>(obase + (size_t)row * step, (state ^ c) & T(0x80808080u));
                ring[(k + kWin + PF) % RINGN] = ld(row + kWin + PF);  // == slot k: the leaving row's slot is free now
            }
        }
    }
}

struct Timer {
    hipEvent_t a, b;
    Timer() { CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); }
    template <typename F> float run(F f, int reps = 7)
    {
        std::vector<float> t;
        f();
        CK(hipDeviceSynchronize());
        for (int i = 0; i < reps; ++i) {
            CK(hipEventRecord(a));
            f();
            CK(hipEventRecord(b));
            CK(hipEventSynchronize(b));
            float ms;
            CK(hipEventElapsedTime(&ms, a, b));
            t.push_back(ms);
        }
        std::sort(t.begin(), t.end());
        return t[t.size() / 2];
    }
};

template <int VEC, int DEPTH, int NS, int RING, int RINGN>
void run_walk(Timer& tm, const uint8_t* in, uint8_t* out, int pages, int W, int H, int rps, const char* name)
{
    WalkArgs a;
    a.in = in; a.out = out; a.W = W; a.H = H; a.page_stride = (size_t)W * H;
    a.rps = rps;
    a.n_strips = W / (64 * VEC);
    a.n_segs = (H + rps - 1) / rps;
    a.total = (unsigned)pages * a.n_strips * a.n_segs;
    const unsigned grid = (a.total + 7u) & ~7u;
    const float ms = tm.run([&] { hipLaunchKernelGGL((k_walk<VEC, DEPTH, NS, RING, RINGN>), dim3(grid), dim3(64), 0, 0, a); });
    const double alg = 2.0 * pages * (double)W * H;
    std::printf("{\"kernel\": \"walk\", \"name\": \"%s\", \"vec\": %d, \"depth\": %d, \"streams\": %d, \"ring\": %d, \"ringn\": %d, \"rps\": %d, "
                "\"ms\": %.4f, \"alg_GBs\": %.1f, \"frac_of_8TBs\": %.3f}\n",
                name, VEC, DEPTH, NS, RING, RINGN, rps, ms, alg / ms * 1e-6, alg / ms * 1e-6 / 8000.0);
    std::fflush(stdout);
}

int main(int argc, char** argv)
{
    const int pages = argc > 1 ? std::atoi(argv[1]) : 256;
    const int W = 4096, H = 4096;
    const size_t bytes = (size_t)pages * W * H;
    uint8_t *in, *out;
    CK(hipMalloc(&in, bytes + 4096 * 64));
    CK(hipMalloc(&out, bytes + 4096 * 64));
    CK(hipMemset(in, 0x5a, bytes + 4096 * 64));
    CK(hipMemset(out, 0, bytes + 4096 * 64));
    Timer tm;
    const size_t n16 = bytes / 16;
    auto flat = [&](const char* name, auto kern, int blocks_per_cu, double moved) {
        const float ms = tm.run([&] { hipLaunchKernelGGL(kern, dim3(256 * blocks_per_cu), dim3(256), 0, 0, (const u4v*)in, (u4v*)out, n16); });
        std::printf("{\"kernel\": \"flat\", \"name\": \"%s\", \"blocks_per_cu\": %d, \"ms\": %.4f, \"GBs\": %.1f, \"frac_of_8TBs\": %.3f}\n", name,
                    blocks_per_cu, ms, moved / ms * 1e-6, moved / ms * 1e-6 / 8000.0);
        std::fflush(stdout);
    };
    for (int bpc : {8, 16, 32}) {
        flat("read16", k_flat<0, false>, bpc, (double)bytes);
        flat("read16_nt", k_flat<0, true>, bpc, (double)bytes);
        flat("write16", k_flat<1, false>, bpc, (double)bytes);
        flat("write16_nt", k_flat<1, true>, bpc, (double)bytes);
        flat("copy16", k_flat<2, false>, bpc, 2.0 * bytes);
        flat("copy16_nt", k_flat<2, true>, bpc, 2.0 * bytes);
    }
    // k_fused's shape, every stream from global memory
    run_walk<8, 1, 3, 0, 32>(tm, in, out, pages, W, H, 128, "3 global streams, 8 B/lane, 1 row ahead (k_fused probe 3)");
    run_walk<8, 2, 3, 0, 32>(tm, in, out, pages, W, H, 128, "3 global streams, 8 B/lane, 2 rows ahead");
    run_walk<8, 4, 3, 0, 32>(tm, in, out, pages, W, H, 128, "3 global streams, 8 B/lane, 4 rows ahead");
    run_walk<16, 1, 3, 0, 32>(tm, in, out, pages, W, H, 128, "3 global streams, 16 B/lane, 1 row ahead");
    run_walk<16, 2, 3, 0, 32>(tm, in, out, pages, W, H, 128, "3 global streams, 16 B/lane, 2 rows ahead");
    run_walk<16, 4, 3, 0, 32>(tm, in, out, pages, W, H, 128, "3 global streams, 16 B/lane, 4 rows ahead");
    run_walk<8, 2, 2, 0, 32>(tm, in, out, pages, W, H, 128, "2 global streams (entering, leaving), 8 B/lane, 2 ahead");
    run_walk<16, 2, 2, 0, 32>(tm, in, out, pages, W, H, 128, "2 global streams (entering, leaving), 16 B/lane, 2 ahead");
    // the streaming skeleton: one stream in, one out
    run_walk<8, 1, 1, 0, 32>(tm, in, out, pages, W, H, 128, "1 global stream, 8 B/lane, 1 ahead (k_fused probe 2)");
    run_walk<8, 4, 1, 0, 32>(tm, in, out, pages, W, H, 128, "1 global stream, 8 B/lane, 4 ahead");
    run_walk<16, 1, 1, 0, 32>(tm, in, out, pages, W, H, 128, "1 global stream, 16 B/lane, 1 ahead");
    run_walk<16, 2, 1, 0, 32>(tm, in, out, pages, W, H, 128, "1 global stream, 16 B/lane, 2 ahead");
    run_walk<16, 4, 1, 0, 32>(tm, in, out, pages, W, H, 128, "1 global stream, 16 B/lane, 4 ahead");
    run_walk<16, 4, 1, 0, 32>(tm, in, out, pages, W, H, 512, "1 global stream, 16 B/lane, 4 ahead, 512-row segments");
    // LDS ring
    run_walk<8, 2, 1, 1, 32>(tm, in, out, pages, W, H, 128, "LDS ring 32 rows, 8 B/lane, 2 ahead");
    run_walk<8, 5, 1, 1, 32>(tm, in, out, pages, W, H, 128, "LDS ring 32 rows, 8 B/lane, 5 ahead");
    run_walk<8, 5, 1, 1, 32>(tm, in, out, pages, W, H, 512, "LDS ring 32 rows, 8 B/lane, 5 ahead, 512-row segments");
    run_walk<16, 2, 1, 1, 32>(tm, in, out, pages, W, H, 128, "LDS ring 32 rows, 16 B/lane, 2 ahead");
    run_walk<16, 5, 1, 1, 32>(tm, in, out, pages, W, H, 512, "LDS ring 32 rows, 16 B/lane, 5 ahead, 512-row segments");
    // register ring
    run_walk<8, 1, 1, 2, 32>(tm, in, out, pages, W, H, 128, "register ring 32 rows, 8 B/lane");
    run_walk<8, 1, 1, 2, 36>(tm, in, out, pages, W, H, 144, "register ring 36 rows, 8 B/lane");
    run_walk<8, 1, 1, 2, 36>(tm, in, out, pages, W, H, 576, "register ring 36 rows, 8 B/lane, 576-row segments");
    run_walk<8, 1, 1, 2, 40>(tm, in, out, pages, W, H, 160, "register ring 40 rows, 8 B/lane");
    run_walk<16, 1, 1, 2, 36>(tm, in, out, pages, W, H, 144, "register ring 36 rows, 16 B/lane");
    CK(hipFree(in));
    CK(hipFree(out));
    return 0;
}
