// Micro-benchmark: does the ORDER in which wavefronts write a page batch matter?  k_fused / k_morph_bits wavefronts
// walk down a strip (512 B or 1 KiB per row at a 4 KiB row pitch); a fill kernel writes the same bytes linearly.
//   hipcc -O3 --offload-arch=gfx950 -o store_pattern store_pattern.hip && ./store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>

typedef unsigned u2v __attribute__((ext_vector_type(2)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));

// linear: wavefront w writes bytes [w * chunk * rows, ...) as `rows` consecutive pieces of `chunk` bytes
template <int BPL, bool NT>
__global__ void __launch_bounds__(64) k_linear(unsigned char* dst, int rows)
{
    const size_t base = ((size_t)blockIdx.x * rows) * (64 * BPL) + (size_t)threadIdx.x * BPL;
    for (int r = 0; r < rows; ++r) {
        unsigned char* p = dst + base + (size_t)r * 64 * BPL;
        if (BPL == 8) {
            u2v v = {(unsigned)r, blockIdx.x};
            if (NT) __builtin_nontemporal_store(v, reinterpret_cast<u2v*>(p)); else *reinterpret_cast<u2v*>(p) = v;
        } else {
            u4v v = {(unsigned)r, blockIdx.x, 1u, 2u};
            if (NT) __builtin_nontemporal_store(v, reinterpret_cast<u4v*>(p)); else *reinterpret_cast<u4v*>(p) = v;
        }
    }
}

// strips: pages of `pitch` bytes per row; wavefront = (page, segment of `rows` rows, strip of 64*BPL bytes), strips
// fastest - the mapping k_fused uses (without its XCD renumbering when XCD == false)
template <int BPL, bool NT, bool XCD>
__global__ void __launch_bounds__(64) k_strips(unsigned char* dst, int pitch, int height, int rows)
{
    const unsigned nb = gridDim.x;
    const unsigned wid = XCD ? (blockIdx.x & 7u) * (nb >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const int n_strips = pitch / (64 * BPL), n_segs = height / rows;
    const int per_page = n_strips * n_segs;
    const int page = wid / per_page, rem = wid - page * per_page, seg = rem / n_strips, strip = rem - seg * n_strips;
    unsigned char* p = dst + (size_t)page * pitch * height + (size_t)seg * rows * pitch + (size_t)strip * 64 * BPL + (size_t)threadIdx.x * BPL;
    for (int r = 0; r < rows; ++r, p += pitch) {
        if (BPL == 8) {
            u2v v = {(unsigned)r, wid};
            if (NT) __builtin_nontemporal_store(v, reinterpret_cast<u2v*>(p)); else *reinterpret_cast<u2v*>(p) = v;
        } else {
            u4v v = {(unsigned)r, wid, 1u, 2u};
            if (NT) __builtin_nontemporal_store(v, reinterpret_cast<u4v*>(p)); else *reinterpret_cast<u4v*>(p) = v;
        }
    }
}

template <typename F> static void timeit(const char* name, size_t bytes, F launch)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    launch();
    (void)hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) launch();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    std::printf("%-52s %7.3f ms  %7.1f GB/s written\n", name, ms, (double)bytes / (ms * 1e-3) / 1e9);
}

int main()
{
    const int pitch = 4096, height = 4096, pages = 256, rows = 128;
    const size_t bytes = (size_t)pages * pitch * height;
    unsigned char* d;
    (void)hipMalloc(&d, bytes);
    timeit("hipMemsetAsync", bytes, [&] { (void)hipMemsetAsync(d, 1, bytes, 0); });
    timeit("linear 8 B/lane", bytes, [&] { k_linear<8, false><<<bytes / (512 * rows), 64>>>(d, rows); });
    timeit("linear 8 B/lane nt", bytes, [&] { k_linear<8, true><<<bytes / (512 * rows), 64>>>(d, rows); });
    timeit("linear 16 B/lane nt", bytes, [&] { k_linear<16, true><<<bytes / (1024 * rows), 64>>>(d, rows); });
    timeit("strips 512 B x 128 rows", bytes, [&] { k_strips<8, false, false><<<bytes / (512 * rows), 64>>>(d, pitch, height, rows); });
    timeit("strips 512 B x 128 rows nt", bytes, [&] { k_strips<8, true, false><<<bytes / (512 * rows), 64>>>(d, pitch, height, rows); });
    timeit("strips 512 B x 128 rows nt, XCD-contiguous", bytes, [&] { k_strips<8, true, true><<<bytes / (512 * rows), 64>>>(d, pitch, height, rows); });
    timeit("strips 1 KiB x 128 rows nt", bytes, [&] { k_strips<16, true, false><<<bytes / (1024 * rows), 64>>>(d, pitch, height, rows); });
    timeit("strips 1 KiB x 128 rows nt, XCD-contiguous", bytes, [&] { k_strips<16, true, true><<<bytes / (1024 * rows), 64>>>(d, pitch, height, rows); });
    timeit("strips 1 KiB x 32 rows nt, XCD-contiguous", bytes, [&] { k_strips<16, true, true><<<bytes / (1024 * 32), 64>>>(d, pitch, height, 32); });
    return 0;
}
