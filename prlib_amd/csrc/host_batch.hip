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
#include <chrono>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
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


// ---- the config-5 chain on a host page list ------------------------------------------------------------------------------
// One worker thread per device; its block of pages goes through the device in chunks of `chain_host_pages` pages (big: the
// angle search of deskew wants hundreds of pages per pass, and 288 GB of HBM hold two chunks of A4 colour scans with room
// to spare).  Three things run at once: the upload of chunk k+1 (helper thread: the caller's pageable pages -> two small
// pinned slots -> device), the chain on chunk k (prl_hip_chain_pages_device, which overlaps its own passes), and the
// download of chunk k-1 (helper thread).  Pinned memory is only the bounce slots: pinning gigabytes costs more than the job.
struct PinSlots {
    uint8_t* p[2] = {nullptr, nullptr};
    uint8_t* d[2] = {nullptr, nullptr};   // device-side packing area of the download direction
    hipEvent_t ev[2] = {nullptr, nullptr};
    hipStream_t stream = nullptr;
    size_t bytes = 0;
    int init(size_t slot_bytes, bool device_side = false)
    {
        bytes = slot_bytes;
        if (device_side)
            for (int i = 0; i < 2; ++i) PRL_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d[i]), slot_bytes));
        PRL_HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        for (int i = 0; i < 2; ++i) {
            PRL_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&p[i]), slot_bytes, hipHostMallocDefault));
            PRL_HIP_CHECK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
        }
        return PRL_OK;
    }
    ~PinSlots()
    {
        if (stream) (void)hipStreamSynchronize(stream);
        for (int i = 0; i < 2; ++i) {
            if (p[i]) (void)hipHostFree(p[i]);
            if (d[i]) (void)hipFree(d[i]);
            if (ev[i]) (void)hipEventDestroy(ev[i]);
        }
        if (stream) (void)hipStreamDestroy(stream);
    }
};

// pages [first, first + cnt) of the caller -> d_in (pages in_pitch apart, rows packed)
int upload_pages(int dev, PinSlots& ps, int cnt, size_t row_bytes, int rows, const uint8_t* const* src, size_t src_step,
                 uint8_t* d_in, size_t in_pitch, int threads)
{
    PRL_HIP_CHECK(hipSetDevice(dev));
    const size_t page = row_bytes * (size_t)rows;
    const int per = (int)std::max<size_t>(1, ps.bytes / page);
    int k = 0;
    for (int off = 0; off < cnt; off += per, ++k) {
        const int n = std::min(per, cnt - off), slot = k & 1;
        PRL_HIP_CHECK(hipEventSynchronize(ps.ev[slot]));  // the DMA that last read this slot
        copy_pages(n, row_bytes, rows, src + off, src_step, ps.p[slot], true, nullptr, 0, threads);
        PRL_HIP_CHECK(hipMemcpy2DAsync(d_in + (size_t)off * in_pitch, in_pitch, ps.p[slot], page, page, (size_t)n, hipMemcpyHostToDevice,
                                       ps.stream));
        PRL_HIP_CHECK(hipEventRecord(ps.ev[slot], ps.stream));
    }
    PRL_HIP_CHECK(hipStreamSynchronize(ps.stream));
    return PRL_OK;
}

// results of cnt pages (sizes in wh, device pages out_pitch apart with rows of dev_step bytes) -> the caller's pages
int download_pages(int dev, PinSlots& ps, int cnt, const int32_t* wh, const uint8_t* d_out, size_t out_pitch, size_t dev_step,
                   uint8_t* const* dst, size_t dst_step, int threads)
{
    PRL_HIP_CHECK(hipSetDevice(dev));
    int k = 0;
    double t_dma = 0, t_copy = 0;
    auto now = [] { return std::chrono::steady_clock::now(); };
    struct Log { double *a, *b; ~Log() { if (env_knobs().debug) std::fprintf(stderr, "[prl chain host] download: dma %.3f s, copy out %.3f s\n", *a, *b); } } log{&t_dma, &t_copy};
    for (int i = 0; i < cnt;) {  // as many whole pages as fit a slot
        const int slot = k++ & 1;
        int j = i;
        size_t used = 0;
        PRL_HIP_CHECK(hipEventSynchronize(ps.ev[slot]));
        const auto t0 = now();
        while (j < cnt) {
            const size_t bytes = (size_t)wh[2 * j] * (size_t)wh[2 * j + 1];
            if (bytes > ps.bytes) return PRL_ERR_BAD_ARG;
            if (used + bytes > ps.bytes) break;
            // rows packed on the device first: a strided device -> host copy of a page took 30 ms, the two steps take 0.3
            PRL_HIP_CHECK(hipMemcpy2DAsync(ps.d[slot] + used, (size_t)wh[2 * j], d_out + (size_t)j * out_pitch, dev_step, (size_t)wh[2 * j],
                                           (size_t)wh[2 * j + 1], hipMemcpyDeviceToDevice, ps.stream));
            used += bytes;
            ++j;
        }
        if (used) PRL_HIP_CHECK(hipMemcpyAsync(ps.p[slot], ps.d[slot], used, hipMemcpyDeviceToHost, ps.stream));
        PRL_HIP_CHECK(hipStreamSynchronize(ps.stream));
        const auto t1 = now();
        size_t at = 0;
        for (int q = i; q < j; ++q) {
            uint8_t* one = dst[q];
            copy_pages(1, (size_t)wh[2 * q], wh[2 * q + 1], nullptr, 0, ps.p[slot] + at, false, &one, dst_step, threads);
            at += (size_t)wh[2 * q] * (size_t)wh[2 * q + 1];
        }
        t_dma += std::chrono::duration<double>(t1 - t0).count();
        t_copy += std::chrono::duration<double>(now() - t1).count();
        i = j;
    }
    return PRL_OK;
}

int chain_worker(int dev, const prl_chain_params* cp, int channels, int first, int count, const uint8_t* const* src, size_t src_step,
                 int width, int height, uint8_t* const* dst, size_t dst_step, int32_t* out_wh, double* angles, int max_w, int max_h,
                 int copy_threads)
{
    if (count == 0) return PRL_OK;
    int st = prl_hip_set_device(dev);
    if (st != PRL_OK) return st;
    PRL_HIP_CHECK(hipSetDevice(dev));
    const size_t row_bytes = (size_t)width * channels;
    const size_t in_pitch = (row_bytes * (size_t)height + 255) / 256 * 256, out_pitch = ((size_t)max_w * max_h + 255) / 256 * 256;
    // Device chunks: the whole block when its pages and results fit the budget (PCIe moves an A4 page in well under a
    // millisecond, the chain takes ten: nothing to hide), else two buffers of a whole number of chain passes each (a chunk cut
    // in the middle of a pass would leave the angle search of deskew a few pages, which it handles at a single page's latency).
    int chunk = env_knobs().chain_host_pages;
    if (chunk <= 0) {
        const size_t budget = env_knobs().chain_host_mb << 20, per = in_pitch + out_pitch;
        if ((size_t)count * per <= budget) chunk = count;
        else {
            int pass = 1;
            st = chain_pass_layout(cp, count, channels, width, height, &pass, nullptr, nullptr);
            if (st != PRL_OK) return st;
            const int fit = (int)std::max<size_t>(1, budget / (2 * per));
            chunk = fit >= pass ? fit / pass * pass : fit;
        }
    }
    chunk = std::max(1, std::min(count, chunk));
    const int n_chunks = (count + chunk - 1) / chunk, n_buf = n_chunks > 1 ? 2 : 1;
    DeviceCtx* ctx = device_ctx(dev);
    std::lock_guard<std::mutex> host_lk(ctx->host_mu);
    struct { uint8_t *in[2], *out[2]; } bufs{};
    for (int i = 0; i < n_buf; ++i) {
        if ((st = ensure_buffer(&ctx->host_buf[i], &ctx->host_buf_bytes[i], in_pitch * (size_t)chunk)) != PRL_OK) return st;
        if ((st = ensure_buffer(&ctx->host_buf[2 + i], &ctx->host_buf_bytes[2 + i], out_pitch * (size_t)chunk)) != PRL_OK) return st;
        bufs.in[i] = static_cast<uint8_t*>(ctx->host_buf[i]);
        bufs.out[i] = static_cast<uint8_t*>(ctx->host_buf[2 + i]);
    }
    const size_t slot = std::max<size_t>({env_knobs().host_chunk_mb << 20, row_bytes * (size_t)height, (size_t)max_w * max_h});
    for (int i = 0; i < 2; ++i) {
        PinSlots* have = static_cast<PinSlots*>(ctx->host_slots[i]);
        if (!have || have->bytes < slot) {
            std::unique_ptr<PinSlots> fresh(new PinSlots());
            if ((st = fresh->init(slot, i == 1)) != PRL_OK) return st;
            delete have;
            ctx->host_slots[i] = fresh.release();
        }
    }
    PinSlots& up = *static_cast<PinSlots*>(ctx->host_slots[0]);
    PinSlots& down = *static_cast<PinSlots*>(ctx->host_slots[1]);
    // one stream per device, kept across calls: the binarizer's workspaces are per stream, and growing them (a device-wide
    // synchronisation) in the middle of a chain would wait for the angle search of the next pass
    {
        std::lock_guard<std::mutex> lk(ctx->streams_mu);
        if (!ctx->host_run) PRL_HIP_CHECK(hipStreamCreateWithFlags(&ctx->host_run, hipStreamNonBlocking));
    }
    hipStream_t run = ctx->host_run;
    struct Helper {   // a thread that reports a status and is always joined
        std::thread th;
        int st = PRL_OK;
        std::string detail;
        int join() { if (th.joinable()) th.join(); return st; }
        ~Helper() { join(); }
    } uploader, downloader;
    const int th2 = std::max(1, copy_threads / 2);
    auto start_upload = [&](int c) {
        const int off = c * chunk, cnt = std::min(chunk, count - off);
        uploader.st = PRL_OK;
        uploader.th = std::thread([&, c, off, cnt] {
            const auto t0 = std::chrono::steady_clock::now();
            struct Log { decltype(t0) t; int c; ~Log() { if (env_knobs().debug) std::fprintf(stderr, "[prl chain host] upload of chunk %d: %.3f s\n", c, std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count()); } } log{t0, c};
            uploader.st = upload_pages(dev, up, cnt, row_bytes, height, src + first + off, src_step, bufs.in[c % n_buf], in_pitch, th2);
            if (uploader.st != PRL_OK) uploader.detail = prl_hip_last_error_detail();
        });
    };
    start_upload(0);
    int result = PRL_OK;
    for (int c = 0; c < n_chunks && result == PRL_OK; ++c) {
        const int off = c * chunk, cnt = std::min(chunk, count - off);
        if (uploader.join() != PRL_OK) { set_error_detail("upload: " + uploader.detail); result = uploader.st; break; }
        if (c + 1 < n_chunks) start_upload(c + 1);   // in[(c+1) % 2]: last read by the chain on chunk c-1, which has finished
        st = prl_hip_chain_pages_device(cp, cnt, channels, bufs.in[c % n_buf], in_pitch, row_bytes, width, height, bufs.out[c % n_buf],
                                        out_pitch, (size_t)max_w, out_wh + 2 * (size_t)(first + off), angles ? angles + first + off : nullptr, run);
        if (st == PRL_OK && hipStreamSynchronize(run) != hipSuccess) { set_error_detail("chain: stream"); st = PRL_ERR_HIP; }
        if (st != PRL_OK) { result = st; break; }
        if (downloader.join() != PRL_OK) { set_error_detail("download: " + downloader.detail); result = downloader.st; break; }
        downloader.st = PRL_OK;
        downloader.th = std::thread([&, c, off, cnt] {   // out[c % 2] is written again by the chain on chunk c+2, after the join above
            const auto t0 = std::chrono::steady_clock::now();
            struct Log { decltype(t0) t; int c; ~Log() { if (env_knobs().debug) std::fprintf(stderr, "[prl chain host] download of chunk %d: %.3f s\n", c, std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count()); } } log{t0, c};
            downloader.st = download_pages(dev, down, cnt, out_wh + 2 * (size_t)(first + off), bufs.out[c % n_buf], out_pitch, (size_t)max_w,
                                           dst + first + off, dst_step, th2);
            if (downloader.st != PRL_OK) downloader.detail = prl_hip_last_error_detail();
        });
    }
    (void)uploader.join();
    if (downloader.join() != PRL_OK && result == PRL_OK) { set_error_detail("download: " + downloader.detail); result = downloader.st; }
    (void)hipStreamSynchronize(run);
    return result;
}

}  // namespace

void host_slots_free(DeviceCtx* ctx)
{
    for (int i = 0; i < 2; ++i) {
        delete static_cast<PinSlots*>(ctx->host_slots[i]);
        ctx->host_slots[i] = nullptr;
    }
}
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
    const int real = visible;
    visible = std::max(visible, env_knobs().fake_devices);   // tests on a one-GPU box: several workers, all on the real device(s)
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
            status[(size_t)d] = device_worker(d % real, p, g, first, count, src, src_step, width, height, dst, dst_step, copy_threads);
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

/*
 * The chain (prl_hip_chain_pages_device: [deskew] -> [denoise] -> [backgroundNormalization] -> gray -> binarize -> [thinning])
 * on a list of host pages of one size, sharded over the GPUs of the node like prl_hip_binarize_batch_host.  dst[i] must have
 * room for the largest result (prl_hip_chain_max_out_size) at dst_step >= that width; out_wh (2 ints per page) receives
 * each page's result size, angles (optional) the deskew angle in degrees.
 */
int prl_hip_chain_batch_host(const prl_chain_params* cp, int n_pages, int channels, const uint8_t* const* src, size_t src_step,
                             int width, int height, uint8_t* const* dst, size_t dst_step, int32_t* out_wh, double* angles,
                             int n_devices)
{
    if (!cp) return PRL_ERR_BAD_ARG;
    int max_w = 0, max_h = 0;
    int st = prl_hip_chain_max_out_size(cp, width, height, &max_w, &max_h);
    if (st != PRL_OK) return st;
    if (channels != 1 && channels != 3 && channels != 4) return PRL_ERR_BAD_CHANNELS;
    if (n_pages < 0 || n_devices < 0) return PRL_ERR_BAD_ARG;
    if (n_pages == 0) return PRL_OK;
    if (!src || !dst || !out_wh || src_step < (size_t)width * channels || dst_step < (size_t)max_w) return PRL_ERR_BAD_ARG;
    for (int i = 0; i < n_pages; ++i)
        if (!src[i] || !dst[i]) return PRL_ERR_BAD_ARG;
    int visible = 0;
    st = prl_hip_device_count(&visible);
    if (st != PRL_OK) return st;
    if (visible <= 0) {
        set_error_detail("no HIP device");
        return PRL_ERR_NO_DEVICE;
    }
    const int real = visible;
    visible = std::max(visible, env_knobs().fake_devices);
    const int devs = std::min(n_pages, n_devices == 0 ? visible : std::min(n_devices, visible));
    const unsigned hw = std::thread::hardware_concurrency();
    const int copy_threads = std::max(1, std::min(env_knobs().host_copy_threads, hw ? (int)(hw / (unsigned)devs) : 1));
    std::vector<int> status((size_t)devs, PRL_OK);
    std::vector<std::string> detail((size_t)devs);
    std::vector<std::thread> workers;
    for (int d = 0; d < devs; ++d) {
        int first = 0, count = 0;
        prl_hip_page_range(n_pages, devs, d, &first, &count);
        workers.emplace_back([=, &status, &detail]() {
            status[(size_t)d] = chain_worker(d % real, cp, channels, first, count, src, src_step, width, height, dst, dst_step, out_wh, angles,
                                             max_w, max_h, copy_threads);
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
