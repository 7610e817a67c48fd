// host_batch.hip — SURVEY.md §8b / §8e: the page list of a host caller, sharded over the GPUs of the node.
//
// A PRLib user holds pages as cv::Mat in host memory and calls prl::binarizeSauvola once per page
// (samples/binarizations/binarizeSauvola_sample.cpp:48-53).  Pages are independent (src/binarizations/binarizeSauvola.cpp:32-134
// touches only its two Mats), so a list of n pages splits into contiguous blocks, one per device (prl_hip_page_range: the
// same split prlib_amd/dist.py uses for one-process-per-GPU runs), and inside a device into chunks that alternate between
// TWO streams with their own pinned and device buffers:
//
//     worker thread of device d:   stage chunk k (host memcpy into pinned, `host_copy_threads` threads)
//                                  stream k%2:  H2D  ->  binarize (enqueue only)                       [GPU works on k]
//                                  chunk k-1:   prl_hip_finish -> D2H -> copy out to the caller's pages [host works on k-1]
//
// so the host copies of one chunk overlap the DMA and the kernels of the other.  No collective, no peer traffic: each
// device only ever sees its own block.  Results land in the caller's buffers in the caller's order.
// End to end this path is bound by host memory copies and PCIe, not by the kernels.
#include <algorithm>
#include <atomic>
#include <cstring>
#include <thread>
#include <vector>

#include "prl_internal.h"

namespace prl_hip {
namespace {

void copy_pages(int n, size_t row_bytes, int rows, const uint8_t* const* src, size_t src_step, uint8_t* dst_packed, bool to_packed,
                uint8_t* const* dst_pages, size_t dst_step, int threads)
{
    // pages [0, n) between the caller's strided pages and a packed pinned buffer, rows split over `threads`
    const size_t page_bytes = row_bytes * (size_t)rows;
    const long long total_rows = (long long)n * rows;
    auto work = [&](long long r0, long long r1) {
        for (long long r = r0; r < r1; ++r) {
            const int pg = (int)(r / rows), y = (int)(r % rows);
            uint8_t* packed = dst_packed + (size_t)pg * page_bytes + (size_t)y * row_bytes;
            if (to_packed) std::memcpy(packed, src[pg] + (size_t)y * src_step, row_bytes);
            else std::memcpy(dst_pages[pg] + (size_t)y * dst_step, packed, row_bytes);
        }
    };
    const int t = (int)std::max<long long>(1, std::min<long long>(threads, total_rows / 256));
    if (t == 1) { work(0, total_rows); return; }
    std::vector<std::thread> pool;
    const long long per = (total_rows + t - 1) / t;
    for (int i = 1; i < t; ++i) pool.emplace_back(work, std::min(total_rows, i * per), std::min(total_rows, (i + 1) * per));
    work(0, std::min(total_rows, per));
    for (auto& th : pool) th.join();
}

struct Buf {
    hipStream_t stream = nullptr;
    uint8_t *pin_in = nullptr, *pin_out = nullptr, *d_in = nullptr, *d_out = nullptr;
    hipEvent_t done = nullptr;
    int first = 0, count = 0;  // chunk in flight
};

int device_worker(int dev, const prl_binarize_params* p, const prl_binarize_geometry& g, int first, int count,
                  const uint8_t* const* src, size_t src_step, int width, int height, uint8_t* const* dst, size_t dst_step,
                  int copy_threads)
{
    if (count == 0) return PRL_OK;
    int st = prl_hip_set_device(dev);
    if (st != PRL_OK) return st;
    PRL_HIP_CHECK(hipSetDevice(dev));
    const EnvKnobs& knobs = env_knobs();
    const size_t in_page = (size_t)width * height, out_page = (size_t)g.out_w * g.out_h;
    const size_t in_pitch_page = (in_page + 255) / 256 * 256, out_pitch_page = (out_page + 255) / 256 * 256;
    const int chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)count, (knobs.host_chunk_mb << 20) / (in_pitch_page + out_pitch_page)));
    Buf b[2];
    auto cleanup = [&]() {
        for (auto& x : b) {
            if (x.stream) (void)hipStreamSynchronize(x.stream);
            if (x.pin_in) (void)hipHostFree(x.pin_in);
            if (x.pin_out) (void)hipHostFree(x.pin_out);
            if (x.d_in) (void)hipFree(x.d_in);
            if (x.d_out) (void)hipFree(x.d_out);
            if (x.done) (void)hipEventDestroy(x.done);
            if (x.stream) (void)hipStreamDestroy(x.stream);
        }
    };
    auto fail = [&](int code) { cleanup(); return code; };
#define HB_CHECK(expr)                                                                   \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess) {                                                          \
            set_error_detail(std::string(#expr) + ": " + hipGetErrorString(e_));         \
            return fail(e_ == hipErrorOutOfMemory ? PRL_ERR_NOMEM : PRL_ERR_HIP);       \
        }                                                                                \
    } while (0)
    const int n_buf = count > chunk ? 2 : 1;
    for (int i = 0; i < n_buf; ++i) {
        HB_CHECK(hipStreamCreateWithFlags(&b[i].stream, hipStreamNonBlocking));
        HB_CHECK(hipHostMalloc(reinterpret_cast<void**>(&b[i].pin_in), in_page * (size_t)chunk, hipHostMallocDefault));
        HB_CHECK(hipHostMalloc(reinterpret_cast<void**>(&b[i].pin_out), out_page * (size_t)chunk, hipHostMallocDefault));
        HB_CHECK(hipMalloc(reinterpret_cast<void**>(&b[i].d_in), in_pitch_page * (size_t)chunk));
        HB_CHECK(hipMalloc(reinterpret_cast<void**>(&b[i].d_out), out_pitch_page * (size_t)chunk));
        HB_CHECK(hipEventCreateWithFlags(&b[i].done, hipEventDisableTiming));
    }
    DeferredScope deferred;  // binarize only enqueues; this worker finishes each chunk itself
    // second half of a chunk's life: flags / literal redo, download, copy out
    auto drain = [&](Buf& x) -> int {
        if (x.count == 0) return PRL_OK;
        int s2 = prl_hip_finish(x.stream);
        if (s2 != PRL_OK) return s2;
        if (hipMemcpy2DAsync(x.pin_out, out_page, x.d_out, out_pitch_page, out_page, (size_t)x.count, hipMemcpyDeviceToHost, x.stream) != hipSuccess ||
            hipStreamSynchronize(x.stream) != hipSuccess) {
            set_error_detail("host batch: download failed");
            return PRL_ERR_HIP;
        }
        copy_pages(x.count, (size_t)g.out_w, g.out_h, nullptr, 0, x.pin_out, false, dst + x.first, dst_step, copy_threads);
        x.count = 0;
        return PRL_OK;
    };
    int k = 0;
    for (int off = 0; off < count; off += chunk, ++k) {
        Buf& x = b[k % n_buf];
        st = drain(x);  // the chunk that used this buffer two steps ago
        if (st != PRL_OK) return fail(st);
        const int cnt = std::min(chunk, count - off);
        x.first = first + off;
        x.count = cnt;
        copy_pages(cnt, (size_t)width, height, src + x.first, src_step, x.pin_in, true, nullptr, 0, copy_threads);
        HB_CHECK(hipMemcpy2DAsync(x.d_in, in_pitch_page, x.pin_in, in_page, in_page, (size_t)cnt, hipMemcpyHostToDevice, x.stream));
        st = prl_hip_binarize_batch_device(p, cnt, x.d_in, in_pitch_page, (size_t)width, width, height, x.d_out, out_pitch_page,
                                           (size_t)g.out_w, x.stream);
        if (st != PRL_OK) return fail(st);
        if (n_buf == 2) {  // while the GPU works on this chunk, finish the other one
            st = drain(b[(k + 1) % 2]);
            if (st != PRL_OK) return fail(st);
        }
    }
    for (int i = 0; i < n_buf; ++i) {
        st = drain(b[i]);
        if (st != PRL_OK) return fail(st);
    }
#undef HB_CHECK
    cleanup();
    return PRL_OK;
}

}  // namespace
}  // namespace prl_hip

using namespace prl_hip;

extern "C" {

// Contiguous block of a list of n_items owned by part `part` of `n_parts` (block sizes differ by at most one).
int prl_hip_page_range(int n_items, int n_parts, int part, int* first, int* count)
{
    if (!first || !count || n_items < 0 || n_parts < 1 || part < 0 || part >= n_parts) return PRL_ERR_BAD_ARG;
    const int base = n_items / n_parts, extra = n_items % n_parts;
    *first = part * base + std::min(part, extra);
    *count = base + (part < extra ? 1 : 0);
    return PRL_OK;
}

int prl_hip_binarize_batch_host(const prl_binarize_params* p, int n_pages, const uint8_t* const* src, size_t src_step, int width,
                                int height, uint8_t* const* dst, size_t dst_step, int n_devices)
{
    prl_binarize_geometry g;
    int st = prl_hip_binarize_geometry(p, width, height, &g);
    if (st != PRL_OK) return st;
    if (n_pages < 0 || n_devices < 0) return PRL_ERR_BAD_ARG;
    if (n_pages == 0) return PRL_OK;
    if (!src || !dst || src_step < (size_t)width || dst_step < (size_t)g.out_w) return PRL_ERR_BAD_ARG;
    for (int i = 0; i < n_pages; ++i)
        if (!src[i] || !dst[i]) return PRL_ERR_BAD_ARG;
    int visible = 0;
    st = prl_hip_device_count(&visible);
    if (st != PRL_OK) return st;
    if (visible <= 0) {
        set_error_detail("no HIP device");
        return PRL_ERR_NO_DEVICE;
    }
    const int devs = std::min(n_pages, n_devices == 0 ? visible : std::min(n_devices, visible));
    // host threads copying pages in / out of pinned memory, per device worker: the knob, bounded by the cores there are
    const unsigned hw = std::thread::hardware_concurrency();
    const int copy_threads = std::max(1, std::min(env_knobs().host_copy_threads, hw ? (int)(hw / (unsigned)devs) : 1));
    std::vector<int> status((size_t)devs, PRL_OK);
    std::vector<std::string> detail((size_t)devs);
    std::vector<std::thread> workers;
    for (int d = 0; d < devs; ++d) {
        int first = 0, count = 0;
        prl_hip_page_range(n_pages, devs, d, &first, &count);
        workers.emplace_back([=, &status, &detail]() {
            status[(size_t)d] = device_worker(d, p, g, first, count, src, src_step, width, height, dst, dst_step, copy_threads);
            if (status[(size_t)d] != PRL_OK) detail[(size_t)d] = prl_hip_last_error_detail();
        });
    }
    for (auto& w : workers) w.join();
    for (int d = 0; d < devs; ++d)
        if (status[(size_t)d] != PRL_OK) {
            set_error_detail("device " + std::to_string(d) + ": " + detail[(size_t)d]);
            return status[(size_t)d];
        }
    return PRL_OK;
}

}  // extern "C"
