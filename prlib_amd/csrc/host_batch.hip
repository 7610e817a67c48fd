// host_batch.hip — SURVEY.md §8b / §8e: the page list of a host caller, sharded over the GPUs of the node.
//
// A PRLib user holds pages as cv::Mat in host memory and calls prl::binarizeSauvola once per page
// (samples/binarizations/binarizeSauvola_sample.cpp:48-53).  Pages are independent (src/binarizations/binarizeSauvola.cpp:32-134
// touches only its two Mats), so a list of n pages splits into contiguous blocks, one per device (prl_hip_page_range: the
// same split prlib_amd/dist.py uses for one-process-per-GPU runs), and inside a device into chunks that run through a
// three-stage pipeline (device_worker): upload of chunk k+1, kernels of chunk k, download of chunk k-1, on three streams
// with four chunk slots that the library keeps between calls.  Pageable pages go through pinned bounce slots (copied by a
// persistent pool of host threads); pages in pinned memory (prl_hip_alloc_host / prl_hip_host_register, or any
// hipHostMalloc'ed buffer a cv::Mat header points into) are read and written by the DMA engines directly.
// No collective, no peer traffic: each device only ever sees its own block.  Results land in the caller's buffers in the
// caller's order.  End to end this path is bound by PCIe (and, for pageable pages, host memory copies), not by the kernels.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include <unistd.h>

#include "prl_internal.h"
#include "prl/work_pool.h"

namespace prl_hip {
namespace {

// ---- the pool of copy threads (prl/work_pool.h), created on first use: PRL_HIP_HOST_COPY_THREADS, default half of the cores, at most 32
WorkPool& copy_pool()
{
    static WorkPool* pool = [] {
        const unsigned hc = std::max(1u, std::thread::hardware_concurrency());
        int n = env_knobs().host_copy_threads > 0 ? env_knobs().host_copy_threads : (int)std::min(32u, std::max(2u, hc / 2));
        return new WorkPool(std::max(1, std::min(n, (int)hc)));   // never destroyed: no joins during process teardown
    }();
    return *pool;
}

// pages [0, n) between the caller's strided pages and a packed buffer, in tasks of about a megabyte
void copy_pages(int n, size_t row_bytes, int rows, const uint8_t* const* src, size_t src_step, uint8_t* dst_packed, bool to_packed,
                uint8_t* const* dst_pages, size_t dst_step, int /*threads*/)
{
    const size_t page_bytes = row_bytes * (size_t)rows;
    const int rows_per_task = (int)std::max<size_t>(1, ((size_t)1 << 20) / std::max<size_t>(1, row_bytes));
    const int tasks_per_page = (rows + rows_per_task - 1) / rows_per_task;
    copy_pool().parallel_for(n * tasks_per_page, [&](int t) {
        const int pg = t / tasks_per_page, y0 = (t % tasks_per_page) * rows_per_task, y1 = std::min(rows, y0 + rows_per_task);
        uint8_t* packed = dst_packed + (size_t)pg * page_bytes;
        if (to_packed) {
            if (src_step == row_bytes) std::memcpy(packed + (size_t)y0 * row_bytes, src[pg] + (size_t)y0 * row_bytes, row_bytes * (size_t)(y1 - y0));
            else for (int y = y0; y < y1; ++y) std::memcpy(packed + (size_t)y * row_bytes, src[pg] + (size_t)y * src_step, row_bytes);
        } else {
            if (dst_step == row_bytes) std::memcpy(dst_pages[pg] + (size_t)y0 * row_bytes, packed + (size_t)y0 * row_bytes, row_bytes * (size_t)(y1 - y0));
            else for (int y = y0; y < y1; ++y) std::memcpy(dst_pages[pg] + (size_t)y * dst_step, packed + (size_t)y * row_bytes, row_bytes);
        }
    });
}

// Is every page of the list in pinned (hipHostMalloc / hipHostRegister) memory?  Then the DMA engines read and write the
// caller's pages directly and no byte is copied by the CPU.
bool pages_pinned(int n, const uint8_t* const* pages, size_t last_byte)
{
    for (int i = 0; i < n; ++i)
        if (!host_range_pinned(pages[i], last_byte + 1)) return false;
    return true;
}

// ---- prl_hip_binarize_batch_host: per-device resources, kept between calls (DeviceCtx::host_bin, guarded by host_mu) -------
// Three streams (upload, kernels, download) and kSlots chunk slots, each with its device pages and - for callers whose pages are
// pageable memory - pinned bounce buffers.  Allocating these per call cost more than the transfers (and left the
// per-stream workspaces of two short-lived streams behind on every call).
struct HostBin {
    static constexpr int kSlots = 4;
    hipStream_t up = nullptr, run = nullptr, down = nullptr;
    struct Slot {
        uint8_t *pin_in = nullptr, *pin_out = nullptr, *d_in = nullptr, *d_out = nullptr;
        hipEvent_t ev_up = nullptr, ev_down = nullptr;
    } slot[kSlots];
    size_t pin_in_bytes = 0, pin_out_bytes = 0, d_in_bytes = 0, d_out_bytes = 0;   // per slot

    int ensure(size_t d_in_need, size_t d_out_need, size_t pin_in_need, size_t pin_out_need)
    {
        if (!up) {
            PRL_HIP_CHECK(hipStreamCreateWithFlags(&up, hipStreamNonBlocking));
            PRL_HIP_CHECK(hipStreamCreateWithFlags(&run, hipStreamNonBlocking));
            PRL_HIP_CHECK(hipStreamCreateWithFlags(&down, hipStreamNonBlocking));
            for (auto& s : slot) {
                PRL_HIP_CHECK(hipEventCreateWithFlags(&s.ev_up, hipEventDisableTiming));
                PRL_HIP_CHECK(hipEventCreateWithFlags(&s.ev_down, hipEventDisableTiming));
            }
        }
        auto grow_dev = [&](uint8_t* Slot::*member, size_t* have, size_t need) -> int {
            if (*have >= need) return PRL_OK;
            PRL_HIP_CHECK(hipDeviceSynchronize());
            for (auto& s : slot) {
                if (s.*member) PRL_HIP_CHECK(hipFree(s.*member));
                s.*member = nullptr;
            }
            *have = 0;
            for (auto& s : slot) {
                hipError_t e = hipMalloc(reinterpret_cast<void**>(&(s.*member)), need);
                if (e != hipSuccess) { set_error_detail(std::string("host batch: hipMalloc: ") + hipGetErrorString(e)); return e == hipErrorOutOfMemory ? PRL_ERR_NOMEM : PRL_ERR_HIP; }
            }
            *have = need;
            return PRL_OK;
        };
        auto grow_pin = [&](uint8_t* Slot::*member, size_t* have, size_t need) -> int {
            if (*have >= need) return PRL_OK;
            PRL_HIP_CHECK(hipDeviceSynchronize());
            for (auto& s : slot) {
                if (s.*member) PRL_HIP_CHECK(hipHostFree(s.*member));
                s.*member = nullptr;
            }
            *have = 0;
            for (auto& s : slot) {
                hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&(s.*member)), need, hipHostMallocDefault);
                if (e != hipSuccess) { set_error_detail(std::string("host batch: hipHostMalloc: ") + hipGetErrorString(e)); return e == hipErrorOutOfMemory ? PRL_ERR_NOMEM : PRL_ERR_HIP; }
            }
            *have = need;
            return PRL_OK;
        };
        int st;
        if ((st = grow_dev(&Slot::d_in, &d_in_bytes, d_in_need)) != PRL_OK) return st;
        if ((st = grow_dev(&Slot::d_out, &d_out_bytes, d_out_need)) != PRL_OK) return st;
        if ((st = grow_pin(&Slot::pin_in, &pin_in_bytes, pin_in_need)) != PRL_OK) return st;
        if ((st = grow_pin(&Slot::pin_out, &pin_out_bytes, pin_out_need)) != PRL_OK) return st;
        return PRL_OK;
    }
    ~HostBin()
    {
        for (hipStream_t s : {up, run, down})
            if (s) (void)hipStreamSynchronize(s);
        // the binarizer's per-stream workspace of `run` goes with the stream (prl_hip_release_workspace's job otherwise)
        for (auto& s : slot) {
            if (s.pin_in) (void)hipHostFree(s.pin_in);
            if (s.pin_out) (void)hipHostFree(s.pin_out);
            if (s.d_in) (void)hipFree(s.d_in);
            if (s.d_out) (void)hipFree(s.d_out);
            if (s.ev_up) (void)hipEventDestroy(s.ev_up);
            if (s.ev_down) (void)hipEventDestroy(s.ev_down);
        }
        for (hipStream_t s : {up, run, down})
            if (s) (void)hipStreamDestroy(s);
    }
};

// One device's block of the page list, as a three-stage pipeline over chunks of pages:
//     uploader thread   chunk k+1:  [pageable: caller pages -> pinned slot (pool threads)]  H2D on `up`
//     this thread       chunk k  :  binarize on `run` (enqueue), prl_hip_finish (flags, literal redo of an overflowing page)
//     downloader thread chunk k-1:  D2H on `down`  [pageable: pinned slot -> caller pages (pool threads)]
// so both DMA directions and both host copies are busy at once.  Pinned caller pages are read and written by the DMA engines
// directly (no bounce buffers, no CPU copies).
int device_worker(int dev, const prl_binarize_params* p, const prl_binarize_geometry& g, int first, int count,
                  const uint8_t* const* src, size_t src_step, int width, int height, uint8_t* const* dst, size_t dst_step,
                  int /*copy_threads*/)
{
    if (count == 0) return PRL_OK;
    int st = prl_hip_set_device(dev);
    if (st != PRL_OK) return st;
    PRL_HIP_CHECK(hipSetDevice(dev));
    const EnvKnobs& knobs = env_knobs();
    const size_t in_page = (size_t)width * height, out_page = (size_t)g.out_w * g.out_h;
    const size_t in_pitch_page = (in_page + 255) / 256 * 256, out_pitch_page = (out_page + 255) / 256 * 256;
    const int chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)count, (knobs.host_chunk_mb << 20) / (in_pitch_page + out_pitch_page)));
    const int n_chunks = (count + chunk - 1) / chunk;
    const bool pinned_in = pages_pinned(count, src + first, (size_t)(height - 1) * src_step + (size_t)width - 1);
    const bool pinned_out = pages_pinned(count, dst + first, (size_t)(g.out_h - 1) * dst_step + (size_t)g.out_w - 1);

    DeviceCtx* ctx = device_ctx(dev);
    std::lock_guard<std::mutex> host_lk(ctx->host_mu);   // one host-list call at a time per device
    if (!ctx->host_bin) ctx->host_bin = new HostBin();
    HostBin& hb = *static_cast<HostBin*>(ctx->host_bin);
    st = hb.ensure(in_pitch_page * (size_t)chunk, out_pitch_page * (size_t)chunk, pinned_in ? 0 : in_page * (size_t)chunk,
                   pinned_out ? 0 : out_page * (size_t)chunk);
    if (st != PRL_OK) return st;

    struct Clock {   // PRL_HIP_DEBUG: where each of the three threads spends the call
        double wait = 0, copy = 0, dma = 0, run = 0;
        static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    } clk_up, clk_run, clk_down;
    const bool dbg = knobs.debug;
    struct Pipe {
        std::mutex mu;
        std::condition_variable cv;
        int uploaded = 0, ran = 0, freed = 0;   // chunks that have passed each stage
        int err = PRL_OK;
        std::string detail;
        void fail(int code, const std::string& d) { { std::lock_guard<std::mutex> lk(mu); if (err == PRL_OK) { err = code; detail = d; } } cv.notify_all(); }
    } pipe;
    auto chunk_first = [&](int k) { return first + k * chunk; };
    auto chunk_count = [&](int k) { return std::min(chunk, count - k * chunk); };

    auto upload_fn = [&] {
        if (hipSetDevice(dev) != hipSuccess) { pipe.fail(PRL_ERR_NO_DEVICE, "uploader: hipSetDevice"); return; }
        for (int k = 0; k < n_chunks; ++k) {
            double t0 = dbg ? Clock::now() : 0;
            {
                std::unique_lock<std::mutex> lk(pipe.mu);
                pipe.cv.wait(lk, [&] { return pipe.err != PRL_OK || k - pipe.freed < HostBin::kSlots; });
                if (pipe.err != PRL_OK) return;
            }
            if (dbg) { const double t1 = Clock::now(); clk_up.wait += t1 - t0; t0 = t1; }
            HostBin::Slot& s = hb.slot[k % HostBin::kSlots];
            const int f = chunk_first(k), cnt = chunk_count(k);
            hipError_t e = hipSuccess;
            if (pinned_in) {
                for (int i = 0; i < cnt && e == hipSuccess; ++i) {
                    // (a page with dense rows is ONE transfer: the 2-D form moved a 4096-row page in 37 ms, row by row)
                    if (src_step == (size_t)width) e = hipMemcpyAsync(s.d_in + (size_t)i * in_pitch_page, src[f + i], in_page, hipMemcpyHostToDevice, hb.up);
                    else e = hipMemcpy2DAsync(s.d_in + (size_t)i * in_pitch_page, (size_t)width, src[f + i], src_step, (size_t)width, (size_t)height,
                                              hipMemcpyHostToDevice, hb.up);
                }
            } else {
                copy_pages(cnt, (size_t)width, height, src + f, src_step, s.pin_in, true, nullptr, 0, 0);
                if (dbg) { const double t1 = Clock::now(); clk_up.copy += t1 - t0; t0 = t1; }
                e = hipMemcpy2DAsync(s.d_in, in_pitch_page, s.pin_in, in_page, in_page, (size_t)cnt, hipMemcpyHostToDevice, hb.up);
            }
            if (dbg) { const double t1 = Clock::now(); clk_up.dma += t1 - t0; t0 = t1; }
            if (e == hipSuccess) e = hipEventRecord(s.ev_up, hb.up);
            if (e != hipSuccess) { pipe.fail(PRL_ERR_HIP, std::string("host batch upload: ") + hipGetErrorString(e)); return; }
            { std::lock_guard<std::mutex> lk(pipe.mu); pipe.uploaded = k + 1; }
            pipe.cv.notify_all();
        }
    };
    auto download_fn = [&] {
        if (hipSetDevice(dev) != hipSuccess) { pipe.fail(PRL_ERR_NO_DEVICE, "downloader: hipSetDevice"); return; }
        auto finalize = [&](int k) -> bool {   // chunk k's D2H has been enqueued: wait for it, hand the pages over, free the slot
            HostBin::Slot& s = hb.slot[k % HostBin::kSlots];
            double t0 = dbg ? Clock::now() : 0;
            if (hipEventSynchronize(s.ev_down) != hipSuccess) { pipe.fail(PRL_ERR_HIP, "host batch: download failed"); return false; }
            if (dbg) { const double t1 = Clock::now(); clk_down.dma += t1 - t0; t0 = t1; }
            if (!pinned_out) copy_pages(chunk_count(k), (size_t)g.out_w, g.out_h, nullptr, 0, s.pin_out, false, dst + chunk_first(k), dst_step, 0);
            if (dbg) clk_down.copy += Clock::now() - t0;
            { std::lock_guard<std::mutex> lk(pipe.mu); pipe.freed = k + 1; }
            pipe.cv.notify_all();
            return true;
        };
        for (int k = 0; k < n_chunks; ++k) {
            const double tw = dbg ? Clock::now() : 0;
            {
                std::unique_lock<std::mutex> lk(pipe.mu);
                pipe.cv.wait(lk, [&] { return pipe.err != PRL_OK || pipe.ran > k; });
                if (pipe.err != PRL_OK) return;
            }
            if (dbg) clk_down.wait += Clock::now() - tw;
            HostBin::Slot& s = hb.slot[k % HostBin::kSlots];
            const int f = chunk_first(k), cnt = chunk_count(k);
            hipError_t e = hipSuccess;
            if (pinned_out) {
                for (int i = 0; i < cnt && e == hipSuccess; ++i) {
                    if (dst_step == (size_t)g.out_w) e = hipMemcpyAsync(dst[f + i], s.d_out + (size_t)i * out_pitch_page, out_page, hipMemcpyDeviceToHost, hb.down);
                    else e = hipMemcpy2DAsync(dst[f + i], dst_step, s.d_out + (size_t)i * out_pitch_page, (size_t)g.out_w, (size_t)g.out_w,
                                              (size_t)g.out_h, hipMemcpyDeviceToHost, hb.down);
                }
            } else {
                e = hipMemcpy2DAsync(s.pin_out, out_page, s.d_out, out_pitch_page, out_page, (size_t)cnt, hipMemcpyDeviceToHost, hb.down);
            }
            if (e == hipSuccess) e = hipEventRecord(s.ev_down, hb.down);
            if (e != hipSuccess) { pipe.fail(PRL_ERR_HIP, std::string("host batch download: ") + hipGetErrorString(e)); return; }
            if (k > 0 && !finalize(k - 1)) return;   // the copy-out of chunk k-1 runs beside the DMA of chunk k
        }
        (void)finalize(n_chunks - 1);
    };
    // (thread creation can fail - resource exhaustion; a joinable std::thread must not be destroyed, so the first one is
    // told to stop and joined before the error goes back to the caller)
    std::thread uploader, downloader;
    try {
        uploader = std::thread(upload_fn);
        downloader = std::thread(download_fn);
    } catch (const std::system_error& e) {
        pipe.fail(PRL_ERR_NOMEM, std::string("host batch: cannot start the copy threads: ") + e.what());
        if (uploader.joinable()) uploader.join();
        for (hipStream_t q : {hb.up, hb.run, hb.down}) (void)hipStreamSynchronize(q);
        set_error_detail(pipe.detail);
        return PRL_ERR_NOMEM;
    }

    {
        DeferredScope deferred;  // binarize only enqueues; prl_hip_finish below closes every chunk
        for (int k = 0; k < n_chunks; ++k) {
            double t0 = dbg ? Clock::now() : 0;
            {
                std::unique_lock<std::mutex> lk(pipe.mu);
                pipe.cv.wait(lk, [&] { return pipe.err != PRL_OK || pipe.uploaded > k; });
                if (pipe.err != PRL_OK) break;
            }
            if (dbg) { const double t1 = Clock::now(); clk_run.wait += t1 - t0; t0 = t1; }
            HostBin::Slot& s = hb.slot[k % HostBin::kSlots];
            int s2 = hipStreamWaitEvent(hb.run, s.ev_up, 0) == hipSuccess ? PRL_OK : PRL_ERR_HIP;
            if (s2 == PRL_OK)
                s2 = prl_hip_binarize_batch_device(p, chunk_count(k), s.d_in, in_pitch_page, (size_t)width, width, height, s.d_out, out_pitch_page,
                                                   (size_t)g.out_w, hb.run);
            if (s2 == PRL_OK) s2 = prl_hip_finish(hb.run);   // the chunk's masks are final (and the stream idle) when this returns
            if (s2 != PRL_OK) { pipe.fail(s2, prl_hip_last_error_detail()); break; }
            if (dbg) clk_run.run += Clock::now() - t0;
            { std::lock_guard<std::mutex> lk(pipe.mu); pipe.ran = k + 1; }
            pipe.cv.notify_all();
        }
    }
    uploader.join();
    downloader.join();
    if (dbg)
        std::fprintf(stderr, "[prl host batch] device %d, %d pages in %d chunks (%s in, %s out), %d copy threads: upload thread waited %.3f s for a slot, "
                     "copied %.3f, enqueued %.3f; kernel thread waited %.3f for uploads, ran %.3f; download thread waited %.3f for kernels, "
                     "%.3f for the DMA, copied %.3f\n", dev, count, n_chunks, pinned_in ? "pinned" : "pageable", pinned_out ? "pinned" : "pageable",
                     copy_pool().threads(), clk_up.wait, clk_up.copy, clk_up.dma, clk_run.wait, clk_run.run, clk_down.wait, clk_down.dma, clk_down.copy);
    for (hipStream_t q : {hb.up, hb.run, hb.down}) (void)hipStreamSynchronize(q);   // nothing of this call stays in flight
    if (pipe.err != PRL_OK) {
        (void)prl_hip_finish(hb.run);   // drop what the failed call left pending on the kept stream
        set_error_detail(pipe.detail);
        return pipe.err;
    }
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
    delete static_cast<HostBin*>(ctx->host_bin);
    ctx->host_bin = nullptr;
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
#ifdef PRL_TEST_HOOKS
    visible = std::max(visible, env_knobs().fake_devices);   // tests on a one-GPU box: several workers, all on the real device(s)
#endif
    const int devs = std::min(n_pages, n_devices == 0 ? visible : std::min(n_devices, visible));
    const int copy_threads = 0;   // (page copies run on the process-wide WorkPool, shared by the device workers)
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

/* Pinned host memory for a caller's pages (a cv::Mat header over it: cv::Mat(rows, cols, CV_8UC1, ptr, step)): the host-list
 * entries then move such pages by DMA straight from / to the caller's memory. */
int prl_hip_alloc_host(size_t bytes, void** out)
{
    if (!out || bytes == 0) return PRL_ERR_BAD_ARG;
    int dev;
    int st = current_device(&dev);
    if (st != PRL_OK) return st;
    hipError_t e = hipHostMalloc(out, bytes, hipHostMallocDefault);
    if (e != hipSuccess) {
        set_error_detail(std::string("hipHostMalloc: ") + hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? PRL_ERR_NOMEM : PRL_ERR_HIP;
    }
    return PRL_OK;
}

int prl_hip_free_host(void* p)
{
    if (!p) return PRL_OK;
    PRL_HIP_CHECK(hipHostFree(p));
    return PRL_OK;
}

/* Pin / unpin memory the caller already owns (page-locks it: costs about as much as copying it once, pays from the second
 * call on the same buffers on). */
int prl_hip_host_register(void* p, size_t bytes)
{
    if (!p || bytes == 0) return PRL_ERR_BAD_ARG;
    int dev;
    int st = current_device(&dev);
    if (st != PRL_OK) return st;
    PRL_HIP_CHECK(hipHostRegister(p, bytes, hipHostRegisterDefault));
    return PRL_OK;
}

int prl_hip_host_unregister(void* p)
{
    if (!p) return PRL_ERR_BAD_ARG;
    PRL_HIP_CHECK(hipHostUnregister(p));
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
#ifdef PRL_TEST_HOOKS
    visible = std::max(visible, env_knobs().fake_devices);
#endif
    const int devs = std::min(n_pages, n_devices == 0 ? visible : std::min(n_devices, visible));
    const int copy_threads = 0;   // (page copies run on the process-wide WorkPool, shared by the device workers)
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
