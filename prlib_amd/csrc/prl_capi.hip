// prl_capi.hip — the extern "C" surface declared in include/prl_hip.h: argument validation with the
// reference's semantics, per-device workspace, and dispatch to the kernels.
//
// There is deliberately no CPU implementation behind any compute entry point: without a usable
// gfx950 device they return PRL_ERR_NO_DEVICE.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <vector>
#include <unistd.h>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <atomic>

#include "prl_internal.h"

namespace prl_hip {

int init_globals_run(PageGlobals* d_globals, int n_pages, hipStream_t stream, void* fused_counters = nullptr);

namespace {

thread_local std::string t_error_detail;
thread_local int t_device = -1;  // -1: follow hipGetDevice()

struct LastCall {
    int device = -1;
    hipStream_t stream = nullptr;
    uint64_t seq = 0;       // which call of that stream's workspace
    bool valid = false;
};
thread_local LastCall t_last;

std::mutex g_ctx_mu;
std::map<int, std::unique_ptr<DeviceCtx>> g_ctx;
std::atomic<int> g_exec_mode{-1};  // -1: take PRL_HIP_MODE
std::atomic<bool> g_profiling{false};
std::atomic<bool> g_deferred{false};  // prl_hip_set_deferred_completion
std::atomic<int> g_literal_budget{-2};  // prl_hip_set_literal_page_budget; -2: take PRL_HIP_LITERAL_PAGE_BUDGET (default -1 = no limit)

int literal_budget()
{
    int b = g_literal_budget.load(std::memory_order_relaxed);
    if (b == -2) {
        const char* e = std::getenv("PRL_HIP_LITERAL_PAGE_BUDGET");
        b = e && *e ? std::max(-1, std::atoi(e)) : -1;
        g_literal_budget.store(b, std::memory_order_relaxed);
    }
    return b;
}
thread_local bool t_force_deferred = false;  // library-internal callers that call prl_hip_finish themselves (host batch)

}  // namespace

void set_error_detail(const std::string& s) { t_error_detail = s; }

const EnvKnobs& env_knobs()
{
    static const EnvKnobs knobs = [] {
        EnvKnobs k;
        auto geti = [](const char* name, long long dflt) { const char* e = std::getenv(name); return e ? std::atoll(e) : dflt; };
        // what a user sets: diagnostics, the execution mode, memory budgets, host copy threads
        k.debug = std::getenv("PRL_HIP_DEBUG") != nullptr;
        const char* m = std::getenv("PRL_HIP_MODE");
        k.literal_mode = (m && std::strcmp(m, "literal") == 0) ? 1 : 0;
        k.literal_scratch_mb = (size_t)std::max(64ll, geti("PRL_HIP_LITERAL_SCRATCH_MB", 8192));
        k.deskew_work_mb = (size_t)std::max(1ll, geti("PRL_HIP_DESKEW_WORK_MB", 24576));
        k.chain_work_mb = (size_t)std::max(16ll, geti("PRL_HIP_CHAIN_WORK_MB", 49152));
        k.chain_host_mb = (size_t)std::max(16ll, geti("PRL_HIP_CHAIN_HOST_MB", 65536));
        k.host_chunk_mb = (size_t)std::max(1ll, geti("PRL_HIP_HOST_CHUNK_MB", 128));
        k.host_copy_threads = (int)std::max(0ll, std::min(64ll, geti("PRL_HIP_HOST_COPY_THREADS", 0)));
#ifdef PRL_TEST_HOOKS
        // kernel-selection / schedule knobs of the experiments and tests: only libprlib_hip_testhooks.so (make hooks) reads them
        auto is0 = [](const char* name) { const char* e = std::getenv(name); return e && e[0] == '0'; };
        k.fused_wpb = (int)std::max(1ll, std::min(4ll, geti("PRL_HIP_WPB", 1)));
        k.flt = !is0("PRL_HIP_FLT");
        k.nt_store = !is0("PRL_HIP_NT");
        k.rows_per_seg = (int)geti("PRL_HIP_ROWS_PER_SEG", 0);
        if (k.rows_per_seg) k.rows_per_seg = std::max(4, k.rows_per_seg);
        k.tiers = !is0("PRL_HIP_TIERS");
        k.ext_strip = !is0("PRL_HIP_EXT_STRIP");
        k.fused_qint = (int)geti("PRL_HIP_FUSED_QINT", 2);
        k.ragged_uo = !is0("PRL_HIP_RAGGED_UO");
        k.wolf_side = !is0("PRL_HIP_WOLF_SIDE");
        k.wolf_tier_max = (int)std::max(32ll, std::min(512ll, geti("PRL_HIP_WOLF_TIER_MAX", 128)));
        k.byte_mask = std::getenv("PRL_HIP_BYTE_MASK") != nullptr;
        k.morph_rps = (int)geti("PRL_MORPH_RPS", 0);
        if (k.morph_rps) k.morph_rps = std::max(8, k.morph_rps);
        k.morph_wpb = (int)std::max(1ll, std::min(4ll, geti("PRL_MORPH_WPB", 1)));
        k.thin_rps = (int)geti("PRL_THIN_RPS", 0);
        if (k.thin_rps) k.thin_rps = std::max(4, k.thin_rps);
        k.thin_wpb = (int)std::max(1ll, std::min(4ll, geti("PRL_THIN_WPB", 4)));
        k.nlm_xl = (int)geti("PRL_NLM_XL", 3);
        k.nlm_glut = (int)geti("PRL_NLM_GLUT", 0);
        k.ppht_mw = (int)geti("PRL_HIP_PPHT_MW", -1);
        k.ppht_prio = (int)geti("PRL_HIP_PPHT_PRIO", 3);
        k.ppht_group = (int)geti("PRL_HIP_PPHT_GROUP", -1);
        k.ppht_group_g = (int)std::max(1ll, std::min(32ll, geti("PRL_HIP_PPHT_GROUP_G", 1)));
        k.ppht_group_xcd = (int)geti("PRL_HIP_PPHT_GROUP_XCD", 0) ? 1 : 0;
        k.ppht_group_spin_ms = (int)std::max(1ll, geti("PRL_HIP_PPHT_GROUP_SPIN_MS", 10000));
        k.ppht_group_kill = (int)geti("PRL_HIP_PPHT_GROUP_KILL", -1);
        k.ppht_group_cus = (int)std::max(0ll, geti("PRL_HIP_PPHT_GROUP_CUS", 0));
        k.chain_host_pages = (int)std::max(0ll, geti("PRL_HIP_CHAIN_HOST_PAGES", 0));
        k.fake_devices = (int)std::max(0ll, geti("PRL_HIP_FAKE_DEVICES", 0));
        k.chain_pass = (int)std::max(0ll, geti("PRL_HIP_CHAIN_PASS", 0));
        k.chain_first_pass = (int)std::max(0ll, geti("PRL_HIP_CHAIN_FIRST_PASS", 0));
        k.chain_overlap = (int)geti("PRL_HIP_CHAIN_OVERLAP", 2);
        k.segmax_cap = (unsigned)std::max(64ll, std::min(1ll << 20, geti("PRL_HIP_SEGMAX_CAP", 1 << 20)));
#endif
        return k;
    }();
    return knobs;
}


// ---- per-(device, stream) workspace of the binarizers -------------------------------------------------------------
// Calls on different streams of one device share nothing (so they overlap); calls on one stream are ordered by the
// stream.  The per-page flags of a PRL_MODE_AUTO call (queue overflow -> the page must be redone literally; counters
// for prl_hip_last_stats) are copied into a pinned slot and looked at LATER: at once in the default mode, lazily
// (next call that needs the slot, prl_hip_last_stats, prl_hip_finish) with prl_hip_set_deferred_completion(1).
constexpr int kSlots = 4;

struct PendingCall {
    int slot = -1;
    bool has_flags = false;   // (literal-mode calls only hold their slot for the page tables)
    uint64_t seq = 0;
    prl_binarize_params params{};
    int width = 0, height = 0, n_pages = 0;
    PageSet src{};            // base / stride form (table == nullptr) ...
    PageSetOut dst{};
    std::vector<const uint8_t*> src_tab;  // ... or host copies of the page tables
    std::vector<uint8_t*> dst_tab;
    uint64_t pixels = 0;
};

struct CallStats {
    uint64_t seq = 0, pixels = 0, refined = 0, exact = 0, literal_pages = 0, wolf_candidates = 0, exact_sweep_pages = 0;
};

struct StreamWs {
    std::mutex mu;            // enqueue / resolve on this stream's workspace
    hipStream_t stream = nullptr;
    void* small = nullptr; size_t small_bytes = 0;      // globals, tables, fused work area
    void* mask = nullptr; size_t mask_bytes = 0;        // thresholded masks waiting for the morphology pass
    void* scratch = nullptr; size_t scratch_bytes = 0;  // literal pipeline
    void* pinned = nullptr; size_t slot_bytes = 0;      // kSlots x [src table][dst table][PageGlobals]
    hipEvent_t ev[kSlots] = {nullptr, nullptr, nullptr, nullptr};
    int next_slot = 0;
    uint64_t seq = 0;
    std::deque<PendingCall> pending;
    CallStats last;           // the most recent call whose numbers are known
    hipEvent_t prof_start = nullptr, prof_stop = nullptr;
    hipEvent_t call_start = nullptr, call_stop = nullptr;   // around everything the call enqueues (prl_hip_last_call_ms)
    bool prof_valid = false;
    WolfSide wolf;            // Wolf-Jolion's side stream and its events, created on the first Wolf-Jolion call
    // The last call's final kernel left the first `clean_pages` PageGlobals and the counter block of `small` in their
    // initial state (FusedParams::ep_host): a following call with the same page count needs no k_init_globals.
    int clean_pages = 0;
};
constexpr int kEpilogueMaxPages = 256;  // beyond that a launch or two per call no longer matter and one workgroup copying flags would

int ws_grow(void** p, size_t* cur, size_t need, hipStream_t stream, size_t floor_bytes = 0)
{
    if (*cur >= need) return PRL_OK;
    if (*p) {
        PRL_HIP_CHECK(hipStreamSynchronize(stream));  // kernels of this stream may still use the old block
        PRL_HIP_CHECK(hipFree(*p));
        *p = nullptr;
        *cur = 0;
    }
    need = std::max(need, floor_bytes);
    PRL_HIP_CHECK(hipMalloc(p, need));
    *cur = need;
    return PRL_OK;
}

StreamWs* stream_ws(DeviceCtx* ctx, hipStream_t stream)
{
    std::lock_guard<std::mutex> lk(ctx->streams_mu);
    auto& p = ctx->streams[stream];
    if (!p) {
        p.reset(new StreamWs());
        p->stream = stream;
    }
    return p.get();
}

void ws_free(StreamWs* ws)
{
    if (ws->small) (void)hipFree(ws->small);
    if (ws->mask) (void)hipFree(ws->mask);
    if (ws->scratch) (void)hipFree(ws->scratch);
    if (ws->pinned) (void)hipHostFree(ws->pinned);
    for (auto& e : ws->ev) if (e) (void)hipEventDestroy(e);
    if (ws->prof_start) (void)hipEventDestroy(ws->prof_start);
    if (ws->prof_stop) (void)hipEventDestroy(ws->prof_stop);
    if (ws->call_start) (void)hipEventDestroy(ws->call_start);
    if (ws->call_stop) (void)hipEventDestroy(ws->call_stop);
    if (ws->wolf.stream) {
        (void)hipStreamSynchronize(ws->wolf.stream);
        (void)hipStreamDestroy(ws->wolf.stream);
        for (hipEvent_t e : {ws->wolf.ev_fork, ws->wolf.ev_min, ws->wolf.ev_a, ws->wolf.ev_coeff}) if (e) (void)hipEventDestroy(e);
        ws->wolf = WolfSide{};
    }
    ws->small = ws->mask = ws->scratch = ws->pinned = nullptr;
    ws->small_bytes = ws->mask_bytes = ws->scratch_bytes = ws->slot_bytes = 0;
    for (auto& e : ws->ev) e = nullptr;
    ws->prof_start = ws->prof_stop = ws->call_start = ws->call_stop = nullptr;
    ws->prof_valid = false;
    ws->pending.clear();
    ws->next_slot = 0;
    ws->clean_pages = 0;
}

DeviceCtx::~DeviceCtx() = default;  // (process teardown: the driver reclaims device memory)

int current_device(int* dev)
{
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        set_error_detail(std::string("no HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "count = 0"));
        (void)hipGetLastError();
        return PRL_ERR_NO_DEVICE;
    }
    int d = t_device;
    if (d < 0) {
        if (hipGetDevice(&d) != hipSuccess) return PRL_ERR_NO_DEVICE;
    } else if (d >= count) {
        set_error_detail("device index out of range");
        return PRL_ERR_NO_DEVICE;
    }
    if (hipSetDevice(d) != hipSuccess) return PRL_ERR_NO_DEVICE;
    *dev = d;
    return PRL_OK;
}

DeviceCtx* device_ctx(int dev)
{
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    auto& p = g_ctx[dev];
    if (!p) {
        p.reset(new DeviceCtx());
        p->device = dev;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess) p->cu_count = prop.multiProcessorCount;
    }
    return p.get();
}

int ensure_buffer(void** buf, size_t* have, size_t bytes)
{
    if (*have >= bytes) return PRL_OK;
    if (*buf) {
        PRL_HIP_CHECK(hipDeviceSynchronize());
        PRL_HIP_CHECK(hipFree(*buf));
        *buf = nullptr;
        *have = 0;
    }
    PRL_HIP_CHECK(hipMalloc(buf, bytes));
    *have = bytes;
    return PRL_OK;
}

int ensure_scratch(DeviceCtx* ctx, size_t bytes)
{
    if (ctx->scratch_bytes >= bytes) return PRL_OK;
    if (ctx->scratch) {
        PRL_HIP_CHECK(hipDeviceSynchronize());
        PRL_HIP_CHECK(hipFree(ctx->scratch));
        ctx->scratch = nullptr;
        ctx->scratch_bytes = 0;
    }
    PRL_HIP_CHECK(hipMalloc(&ctx->scratch, bytes));
    ctx->scratch_bytes = bytes;
    return PRL_OK;
}

int ensure_mask(DeviceCtx* ctx, size_t bytes)
{
    if (ctx->mask_bytes >= bytes) return PRL_OK;
    if (ctx->mask) {
        PRL_HIP_CHECK(hipDeviceSynchronize());
        PRL_HIP_CHECK(hipFree(ctx->mask));
        ctx->mask = nullptr;
        ctx->mask_bytes = 0;
    }
    PRL_HIP_CHECK(hipMalloc(&ctx->mask, bytes));
    ctx->mask_bytes = bytes;
    return PRL_OK;
}

int ensure_small(DeviceCtx* ctx, size_t bytes)
{
    if (ctx->small_bytes >= bytes) return PRL_OK;
    if (ctx->small) {
        PRL_HIP_CHECK(hipDeviceSynchronize());
        PRL_HIP_CHECK(hipFree(ctx->small));
        ctx->small = nullptr;
        ctx->small_bytes = 0;
    }
    ctx->lut_small[0] = ctx->lut_small[1] = nullptr;  // cached NL-means tables lived in the old block
    bytes = std::max<size_t>(bytes, 1 << 20);
    PRL_HIP_CHECK(hipMalloc(&ctx->small, bytes));
    ctx->small_bytes = bytes;
    return PRL_OK;
}

int ensure_stage(DeviceCtx* ctx, size_t bytes)
{
    if (ctx->stage_bytes >= bytes) return PRL_OK;
    if (ctx->stage) {
        PRL_HIP_CHECK(hipDeviceSynchronize());
        PRL_HIP_CHECK(hipFree(ctx->stage));
        ctx->stage = nullptr;
        ctx->stage_bytes = 0;
    }
    PRL_HIP_CHECK(hipMalloc(&ctx->stage, bytes));
    ctx->stage_bytes = bytes;
    return PRL_OK;
}

int stage_acquire(DeviceCtx* ctx, hipStream_t stream)
{
    if (!ctx->stage_use) {
        PRL_HIP_CHECK(hipEventCreateWithFlags(&ctx->stage_use, hipEventDisableTiming));
        return PRL_OK;
    }
    PRL_HIP_CHECK(hipStreamWaitEvent(stream, ctx->stage_use, 0));
    return PRL_OK;
}

int stage_release(DeviceCtx* ctx, hipStream_t stream)
{
    if (!ctx->stage_use) PRL_HIP_CHECK(hipEventCreateWithFlags(&ctx->stage_use, hipEventDisableTiming));
    PRL_HIP_CHECK(hipEventRecord(ctx->stage_use, stream));
    return PRL_OK;
}

int ensure_stage_pinned(DeviceCtx* ctx, size_t bytes)
{
    if (ctx->stage_pinned_bytes >= bytes) return PRL_OK;
    if (ctx->stage_pinned) {
        PRL_HIP_CHECK(hipDeviceSynchronize());
        PRL_HIP_CHECK(hipHostFree(ctx->stage_pinned));
        ctx->stage_pinned = nullptr;
        ctx->stage_pinned_bytes = 0;
    }
    PRL_HIP_CHECK(hipHostMalloc(&ctx->stage_pinned, bytes, hipHostMallocDefault));
    ctx->stage_pinned_bytes = bytes;
    return PRL_OK;
}

// The copies run in bands of rows so that the CPU's memcpy of one band overlaps the DMA of the previous one; the rows
// of a band are split over a few persistent helper threads (one thread's memcpy moves ~12-15 GB/s, the DMA engine
// several times that: the host entry was bound by the single-threaded bounce copy).
namespace {
constexpr size_t kStageBand = (size_t)2 << 20;  // ~2 MiB per band
int rows_per_band(size_t row_bytes, int rows) { return (int)std::max<size_t>(1, std::min<size_t>((size_t)rows, kStageBand / std::max<size_t>(row_bytes, 1))); }

class CopyPool {
public:
    static CopyPool& get()
    {
        static CopyPool* pool = new CopyPool();  // never destroyed: no joins during process teardown
        return *pool;
    }
    // rows [0, n) of `row_bytes` bytes from src (row step src_step) to dst (row step dst_step)
    void copy_rows(uint8_t* dst, size_t dst_step, const uint8_t* src, size_t src_step, size_t row_bytes, int n)
    {
        // (after a fork() the child has the pool object but not its threads: copy on the calling thread then)
        const bool usable = !workers_.empty() && getpid() == owner_pid_;
        const int parts = (!usable || row_bytes * (size_t)n < ((size_t)256 << 10)) ? 1 : (int)workers_.size() + 1;
        if (parts == 1) {
            run(dst, dst_step, src, src_step, row_bytes, 0, n);
            return;
        }
        std::unique_lock<std::mutex> call(call_mu_);  // one caller at a time posts jobs
        const int per = (n + parts - 1) / parts;
        {
            std::lock_guard<std::mutex> lk(mu_);
            for (int t = 0; t + 1 < parts; ++t) {
                Job j{dst, dst_step, src, src_step, row_bytes, std::min(n, (t + 1) * per), std::min(n, (t + 2) * per)};
                if (j.r0 < j.r1) {
                    jobs_.push_back(j);
                    ++pending_;
                }
            }
        }
        cv_.notify_all();
        run(dst, dst_step, src, src_step, row_bytes, 0, std::min(n, per));
        std::unique_lock<std::mutex> lk(mu_);
        done_.wait(lk, [&] { return pending_ == 0; });
    }

private:
    struct Job {
        uint8_t* dst; size_t dst_step; const uint8_t* src; size_t src_step; size_t row_bytes; int r0, r1;
    };
    static void run(uint8_t* dst, size_t dst_step, const uint8_t* src, size_t src_step, size_t row_bytes, int r0, int r1)
    {
        if (r1 <= r0) return;
        if (src_step == row_bytes && dst_step == row_bytes) {
            std::memcpy(dst + (size_t)r0 * row_bytes, src + (size_t)r0 * row_bytes, row_bytes * (size_t)(r1 - r0));
        } else {
            for (int y = r0; y < r1; ++y) std::memcpy(dst + (size_t)y * dst_step, src + (size_t)y * src_step, row_bytes);
        }
    }
    CopyPool()
    {
        owner_pid_ = getpid();
        int n = 3;
        if (const char* e = std::getenv("PRL_HIP_COPY_THREADS")) n = std::max(0, std::min(15, std::atoi(e) - 1));
        const unsigned hc = std::thread::hardware_concurrency();
        if (hc > 0 && (unsigned)n + 1 > hc) n = (int)hc - 1;
        for (int i = 0; i < n; ++i) {
            workers_.emplace_back([this] {
                for (;;) {
                    Job j;
                    {
                        std::unique_lock<std::mutex> lk(mu_);
                        cv_.wait(lk, [&] { return !jobs_.empty(); });
                        j = jobs_.back();
                        jobs_.pop_back();
                    }
                    run(j.dst, j.dst_step, j.src, j.src_step, j.row_bytes, j.r0, j.r1);
                    {
                        std::lock_guard<std::mutex> lk(mu_);
                        if (--pending_ == 0) done_.notify_all();
                    }
                }
            });
            workers_.back().detach();
        }
    }
    std::vector<std::thread> workers_;
    std::vector<Job> jobs_;
    std::mutex mu_, call_mu_;
    std::condition_variable cv_, done_;
    int pending_ = 0;
    pid_t owner_pid_ = 0;
};
}  // namespace

// Is [p, p + bytes) pinned host memory (hipHostMalloc / prl_hip_alloc_host / hipHostRegister)?  Then the DMA engines can
// read / write it directly.  An ordinary malloc'ed pointer is an "invalid value" to the runtime: not an error here.
bool host_range_pinned(const void* p, size_t bytes)
{
    if (!p || bytes == 0) return false;
    for (const uint8_t* q : {static_cast<const uint8_t*>(p), static_cast<const uint8_t*>(p) + bytes - 1}) {
        hipPointerAttribute_t at{};
        if (hipPointerGetAttributes(&at, q) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        if (at.type != hipMemoryTypeHost) return false;
    }
    // both ends are pinned; they must also belong to ONE allocation (two pinned blocks with pageable memory between them
    // would pass the test above).  The runtime knows the extent of the allocation a pointer lies in ...
    void* base = nullptr;
    size_t size = 0;
    if (hipMemGetAddressRange(reinterpret_cast<hipDeviceptr_t*>(&base), &size, const_cast<void*>(p)) == hipSuccess && base && size) {
        const uint8_t* b = static_cast<const uint8_t*>(base);
        return b <= static_cast<const uint8_t*>(p) && static_cast<const uint8_t*>(p) + bytes <= b + size;
    }
    (void)hipGetLastError();
    // ... and where it does not say (registered memory on some stacks), the range is walked in 2 MiB steps
    for (size_t off = (size_t)2 << 20; off < bytes; off += (size_t)2 << 20) {
        hipPointerAttribute_t at{};
        if (hipPointerGetAttributes(&at, static_cast<const uint8_t*>(p) + off) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        if (at.type != hipMemoryTypeHost) return false;
    }
    return true;
}

int stage_upload(DeviceCtx* ctx, size_t pin_off, const uint8_t* src, size_t src_step, size_t row_bytes, int rows,
                 uint8_t* d_dst, hipStream_t stream)
{
    uint8_t* pin = static_cast<uint8_t*>(ctx->stage_pinned) + pin_off;
    const int band = rows_per_band(row_bytes, rows);
    {   // the staging area may still be read by a chain call enqueued on another (non-blocking) stream
        const int st = stage_acquire(ctx, stream);
        if (st != PRL_OK) return st;
    }
    // a dense page in pinned memory (a cv::Mat over prl_hip_alloc_host memory): one DMA straight from the caller's pixels,
    // no bounce copy.  (The *_host entries return only after their stream has drained, so the source outlives the copy.)
    if (src_step == row_bytes && host_range_pinned(src, row_bytes * (size_t)rows)) {
        PRL_HIP_CHECK(hipMemcpyAsync(d_dst, src, row_bytes * (size_t)rows, hipMemcpyHostToDevice, stream));
        return PRL_OK;
    }
    for (int y0 = 0; y0 < rows; y0 += band) {
        const int n = std::min(band, rows - y0);
        uint8_t* p = pin + (size_t)y0 * row_bytes;
        CopyPool::get().copy_rows(p, row_bytes, src + (size_t)y0 * src_step, src_step, row_bytes, n);
        PRL_HIP_CHECK(hipMemcpyAsync(d_dst + (size_t)y0 * row_bytes, p, row_bytes * (size_t)n, hipMemcpyHostToDevice, stream));
    }
    return PRL_OK;
}

int stage_download(DeviceCtx* ctx, size_t pin_off, const uint8_t* d_src, size_t row_bytes, int rows, uint8_t* dst,
                   size_t dst_step, hipStream_t stream)
{
    if (dst_step == row_bytes && host_range_pinned(dst, row_bytes * (size_t)rows)) {   // dense pinned destination: one DMA, no bounce
        PRL_HIP_CHECK(hipMemcpyAsync(dst, d_src, row_bytes * (size_t)rows, hipMemcpyDeviceToHost, stream));
        PRL_HIP_CHECK(hipStreamSynchronize(stream));
        return PRL_OK;
    }
    uint8_t* pin = static_cast<uint8_t*>(ctx->stage_pinned) + pin_off;
    const int band = rows_per_band(row_bytes, rows);
    const int n_bands = (rows + band - 1) / band;
    // one event per band: the host copies band k out of the bounce buffer while bands k+1.. are still in flight
    while ((int)ctx->stage_events.size() < n_bands) {
        hipEvent_t e;
        PRL_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->stage_events.push_back(e);
    }
    for (int k = 0; k < n_bands; ++k) {
        const int y0 = k * band, n = std::min(band, rows - y0);
        PRL_HIP_CHECK(hipMemcpyAsync(pin + (size_t)y0 * row_bytes, d_src + (size_t)y0 * row_bytes, row_bytes * (size_t)n,
                                     hipMemcpyDeviceToHost, stream));
        PRL_HIP_CHECK(hipEventRecord(ctx->stage_events[(size_t)k], stream));
    }
    for (int k = 0; k < n_bands; ++k) {
        const int y0 = k * band, n = std::min(band, rows - y0);
        PRL_HIP_CHECK(hipEventSynchronize(ctx->stage_events[(size_t)k]));
        const uint8_t* p = pin + (size_t)y0 * row_bytes;
        CopyPool::get().copy_rows(dst + (size_t)y0 * dst_step, dst_step, p, row_bytes, row_bytes, n);
    }
    return PRL_OK;
}

int ensure_pinned(DeviceCtx* ctx, size_t bytes)
{
    if (ctx->pinned_bytes >= bytes) return PRL_OK;
    if (ctx->pinned) {
        PRL_HIP_CHECK(hipDeviceSynchronize());
        PRL_HIP_CHECK(hipHostFree(ctx->pinned));
        ctx->pinned = nullptr;
        ctx->pinned_bytes = 0;
    }
    bytes = std::max<size_t>(bytes, 1 << 16);
    PRL_HIP_CHECK(hipHostMalloc(&ctx->pinned, bytes, hipHostMallocDefault));
    ctx->pinned_bytes = bytes;
    return PRL_OK;
}

namespace {

int exec_mode()
{
    int m = g_exec_mode.load(std::memory_order_relaxed);
    if (m < 0) {
        m = env_knobs().literal_mode ? PRL_MODE_LITERAL : PRL_MODE_AUTO;
        g_exec_mode.store(m, std::memory_order_relaxed);
    }
    return m;
}

size_t literal_scratch_budget()
{
    // bytes of float64 integral scratch the literal pipeline may hold at once (default 8 GiB)
    return env_knobs().literal_scratch_mb << 20;
}

// Argument checks in the reference's order: empty (binarizeSauvola.cpp:38-41), window (:43-47).
int geometry_impl(const prl_binarize_params* p, int width, int height, prl_binarize_geometry* g)
{
    if (!p || !g) return PRL_ERR_BAD_ARG;
    std::memset(g, 0, sizeof(*g));
    if (width <= 0 || height <= 0) return PRL_ERR_EMPTY;
    if (!((p->window_size > 1) && ((p->window_size % 2) == 1))) return PRL_ERR_BAD_WINDOW;
    if (p->method < PRL_SAUVOLA || p->method > PRL_FENG) return PRL_ERR_BAD_ARG;
    const int w = std::min(p->window_size, std::min(width, height));  // :57
    g->w = w;
    g->half = w / 2;                                                  // :65
    g->padded_w = width + 2 * g->half;
    g->padded_h = height + 2 * g->half;
    if (p->method == PRL_SAUVOLA || p->method == PRL_NIBLACK) {      // rect from padded size, :66
        g->out_w = g->padded_w - w;
        g->out_h = g->padded_h - w;
    } else {                                                          // rect before padding, WolfJolion.cpp:69
        g->out_w = width - w;
        g->out_h = height - w;
    }
    if (g->out_w <= 0 || g->out_h <= 0) return PRL_ERR_EMPTY_RECT;
    return PRL_OK;
}

ThrParams make_thr_params(const prl_binarize_params* p, const prl_binarize_geometry& g, int width,
                          int height)
{
    ThrParams tp{};
    tp.method = p->method;
    tp.w = g.w;
    tp.half = g.half;
    tp.width = width;
    tp.height = height;
    tp.pw = g.padded_w;
    tp.ph = g.padded_h;
    tp.ow = g.out_w;
    tp.oh = g.out_h;
    const int wSqr = g.w * g.w;                       // binarizeSauvola.cpp:58
    tp.f = 1.0 / static_cast<double>(wSqr);           // :59
    tp.k = p->k;
    const double R = 128;
    const double RBack = 1.0 / R;                     // :61-62
    tp.a = (p->k * RBack);                            // :117
    tp.b = (1.0 - p->k);                              // :117
    tp.c1 = 1.0 - p->feng_alpha1;                     // binarizeFeng.cpp:133
    tp.k2 = p->feng_k2;
    tp.gamma = p->feng_gamma;
    return tp;
}

size_t r256(size_t v) { return (v + 255) / 256 * 256; }

struct SlotLayout {
    size_t table_bytes, globals_bytes, total;
};
SlotLayout slot_layout(int n_pages)
{
    SlotLayout l;
    l.table_bytes = r256(sizeof(void*) * (size_t)n_pages);
    l.globals_bytes = r256(sizeof(PageGlobals) * (size_t)n_pages);
    l.total = 2 * l.table_bytes + l.globals_bytes;
    return l;
}

// The flagged pages of a PRL_MODE_AUTO call (a queue overflowed: pathological inputs only - more than 2^21 pixels of a call inside
// the float32 decision band, or 2^17 within ~1e-6 of their threshold), redone by the literal pipeline (and its morphology pass)
// from the caller's own source pages into the caller's destination pages, as ONE batch: page-pointer tables on the device,
// chunks sized by the literal scratch budget.  (Round 3 redid them one by one - a launch sequence and a workspace check per page.)
int redo_pages_literal(StreamWs* ws, const PendingCall& pc, const std::vector<int>& idx)
{
    const int n = (int)idx.size();
    if (n == 0) return PRL_OK;
    prl_binarize_geometry g;
    int st = geometry_impl(&pc.params, pc.width, pc.height, &g);
    if (st != PRL_OK) return st;
    const ThrParams tp = make_thr_params(&pc.params, g, pc.width, pc.height);
    const int morph = pc.params.morph_iterations;
    const size_t lit = r256(literal_scratch_per_page(tp));
    const size_t mask_step = ((size_t)g.out_w + 63) / 64 * 64, mask_page = r256(mask_step * (size_t)g.out_h);
    const bool large = morph != 0 && std::abs(morph) > kMorphMaxFusedRadius;
    const size_t per_page = lit + (morph != 0 ? mask_page * (large ? 2 : 1) : 0);
    const int chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)n, literal_scratch_budget() / per_page));
    const size_t tab_bytes = r256(sizeof(void*) * (size_t)n), glob_bytes = r256(sizeof(PageGlobals) * (size_t)n);
    // [literal integral planes x chunk][byte masks x chunk (morph != 0)][second mask buffer (large radii)][PageGlobals x n][src table][dst table]
    st = ws_grow(&ws->scratch, &ws->scratch_bytes, per_page * (size_t)chunk + glob_bytes + 2 * tab_bytes, ws->stream);
    if (st != PRL_OK) return st;
    auto* base = static_cast<uint8_t*>(ws->scratch);
    uint8_t* d_mask = base + lit * (size_t)chunk;
    uint8_t* d_mask2 = d_mask + mask_page * (size_t)chunk;
    uint8_t* tail = base + per_page * (size_t)chunk;
    auto* d_g = reinterpret_cast<PageGlobals*>(tail);
    auto** d_src_tab = reinterpret_cast<const uint8_t**>(tail + glob_bytes);
    auto** d_dst_tab = reinterpret_cast<uint8_t**>(tail + glob_bytes + tab_bytes);
    std::vector<const uint8_t*> h_src((size_t)n);
    std::vector<uint8_t*> h_dst((size_t)n);
    for (int j = 0; j < n; ++j) {
        const size_t i = (size_t)idx[(size_t)j];
        h_src[(size_t)j] = pc.src_tab.empty() ? pc.src.base + i * pc.src.page_stride : pc.src_tab[i];
        h_dst[(size_t)j] = pc.dst_tab.empty() ? pc.dst.base + i * pc.dst.page_stride : pc.dst_tab[i];
    }
    PRL_HIP_CHECK(hipMemcpyAsync(d_src_tab, h_src.data(), sizeof(void*) * (size_t)n, hipMemcpyHostToDevice, ws->stream));
    PRL_HIP_CHECK(hipMemcpyAsync(d_dst_tab, h_dst.data(), sizeof(void*) * (size_t)n, hipMemcpyHostToDevice, ws->stream));
    PRL_HIP_CHECK(hipStreamSynchronize(ws->stream));   // (the host vectors go out of scope; pageable memory)
    st = init_globals_run(d_g, n, ws->stream);
    if (st != PRL_OK) return st;
    for (int c0 = 0; c0 < n; c0 += chunk) {
        const int cnt = std::min(chunk, n - c0);
        PageSet src{};
        src.table = d_src_tab + c0;
        src.step = pc.src.step;
        PageSetOut out{};
        out.table = d_dst_tab + c0;
        out.step = pc.dst.step;
        if (tp.method == PRL_WOLFJOLION || tp.method == PRL_FENG) {
            st = page_min_run(tp, src, cnt, d_g + c0, ws->stream);
            if (st != PRL_OK) return st;
        }
        PageSetOut thr = out;
        if (morph != 0) {
            thr = PageSetOut{};
            thr.base = d_mask;
            thr.page_stride = mask_page;
            thr.step = mask_step;
        }
        st = literal_run(tp, src, 0, cnt, thr, ws->scratch, d_g + c0, ws->stream);
        if (st != PRL_OK) return st;
        if (morph == 0) continue;
        PageSet msrc{};
        msrc.base = d_mask;
        msrc.page_stride = mask_page;
        msrc.step = mask_step;
        if (!large) st = morph_binary_run(morph, msrc, cnt, g.out_w, g.out_h, out, ws->stream);
        else st = morph_large_run(morph, msrc, cnt, g.out_w, g.out_h, out, d_mask2, mask_step, ws->stream);
        if (st != PRL_OK) return st;
    }
    return PRL_OK;
}

// The second chance of pages whose REFINE QUEUE overflowed (bit 0 of worklist_overflow: more pixels inside the float32 decision band
// than the queue holds - pages of stripes whose levels sit near their own threshold): the exact sweep (k_fused_exact: integer sums
// for every strip, the float64 interval test inline, no queue) over those pages as one batch, ties through the usual fix-up, then
// the morphology pass.  `still` receives the pages that overflowed the fix-up list on the way (true ties by the 10^5): those go to
// the literal pipeline.  Own workspace layout inside ws->scratch; synchronises the stream (it reads the flags back).
int redo_pages_exact(StreamWs* ws, const PendingCall& pc, const std::vector<int>& idx, std::vector<int>* still, CallStats* cs)
{
    const int n = (int)idx.size();
    if (n == 0) return PRL_OK;
    prl_binarize_geometry g;
    int st = geometry_impl(&pc.params, pc.width, pc.height, &g);
    if (st != PRL_OK) return st;
    const ThrParams tp = make_thr_params(&pc.params, g, pc.width, pc.height);
    const int morph = pc.params.morph_iterations;
    const size_t mask_step = ((size_t)g.out_w + 63) / 64 * 64, mask_page = r256(mask_step * (size_t)g.out_h);
    const bool large = morph != 0 && std::abs(morph) > kMorphMaxFusedRadius;
    const int chunk = std::min(n, std::max(1, fused_max_pages(tp)));
    const size_t masks = morph != 0 ? mask_page * (size_t)chunk * (large ? 2 : 1) : 0;
    const size_t fused_bytes = r256(fused_small_bytes(chunk));
    const size_t tab_bytes = r256(sizeof(void*) * (size_t)n), glob_bytes = r256(sizeof(PageGlobals) * (size_t)n);
    // [byte masks x chunk (morph != 0)][second mask buffer (large radii)][fused work area][PageGlobals x n][src table][dst table]
    st = ws_grow(&ws->scratch, &ws->scratch_bytes, masks + fused_bytes + glob_bytes + 2 * tab_bytes, ws->stream);
    if (st != PRL_OK) return st;
    auto* base = static_cast<uint8_t*>(ws->scratch);
    uint8_t* d_mask = base;
    uint8_t* d_mask2 = d_mask + mask_page * (size_t)chunk;
    uint8_t* d_fused = base + masks;
    auto* d_g = reinterpret_cast<PageGlobals*>(d_fused + fused_bytes);
    auto** d_src_tab = reinterpret_cast<const uint8_t**>(d_fused + fused_bytes + glob_bytes);
    auto** d_dst_tab = reinterpret_cast<uint8_t**>(d_fused + fused_bytes + glob_bytes + tab_bytes);
    std::vector<const uint8_t*> h_src((size_t)n);
    std::vector<uint8_t*> h_dst((size_t)n);
    for (int j = 0; j < n; ++j) {
        const size_t i = (size_t)idx[(size_t)j];
        h_src[(size_t)j] = pc.src_tab.empty() ? pc.src.base + i * pc.src.page_stride : pc.src_tab[i];
        h_dst[(size_t)j] = pc.dst_tab.empty() ? pc.dst.base + i * pc.dst.page_stride : pc.dst_tab[i];
    }
    PRL_HIP_CHECK(hipMemcpyAsync(d_src_tab, h_src.data(), sizeof(void*) * (size_t)n, hipMemcpyHostToDevice, ws->stream));
    PRL_HIP_CHECK(hipMemcpyAsync(d_dst_tab, h_dst.data(), sizeof(void*) * (size_t)n, hipMemcpyHostToDevice, ws->stream));
    PRL_HIP_CHECK(hipStreamSynchronize(ws->stream));   // (the host vectors go out of scope; pageable memory)
    for (int c0 = 0; c0 < n; c0 += chunk) {
        const int cnt = std::min(chunk, n - c0);
        st = init_globals_run(d_g + c0, cnt, ws->stream, d_fused);   // (+ the counter block)
        if (st != PRL_OK) return st;
        PageSet src{};
        src.table = d_src_tab + c0;
        src.step = pc.src.step;
        PageSetOut out{};
        out.table = d_dst_tab + c0;
        out.step = pc.dst.step;
        PageSetOut thr = out;
        if (morph != 0) {
            thr = PageSetOut{};
            thr.base = d_mask;
            thr.page_stride = mask_page;
            thr.step = mask_step;
        }
        const WolfSide* wolf_side = (tp.method == PRL_WOLFJOLION && ws->wolf.stream) ? &ws->wolf : nullptr;
        st = fused_run(tp, src, cnt, thr, d_fused, d_g + c0, ws->stream, nullptr, nullptr, false, true, nullptr, wolf_side, true);
        if (st != PRL_OK) return st;
        if (morph == 0) continue;
        PageSet msrc{};
        msrc.base = d_mask;
        msrc.page_stride = mask_page;
        msrc.step = mask_step;
        if (!large) st = morph_binary_run(morph, msrc, cnt, g.out_w, g.out_h, out, ws->stream);
        else st = morph_large_run(morph, msrc, cnt, g.out_w, g.out_h, out, d_mask2, mask_step, ws->stream);
        if (st != PRL_OK) return st;
    }
    std::vector<PageGlobals> hg((size_t)n);
    PRL_HIP_CHECK(hipMemcpyAsync(hg.data(), d_g, sizeof(PageGlobals) * (size_t)n, hipMemcpyDeviceToHost, ws->stream));
    PRL_HIP_CHECK(hipStreamSynchronize(ws->stream));
    for (int j = 0; j < n; ++j) {
        cs->refined += hg[(size_t)j].n_refined;
        cs->exact += hg[(size_t)j].n_exact;
        if (hg[(size_t)j].worklist_overflow) still->push_back(idx[(size_t)j]);
    }
    return PRL_OK;
}

// Look at the flags of the oldest pending call (waits for that call's work): statistics, literal redo of overflow pages.
int resolve_front(StreamWs* ws)
{
    PendingCall pc = std::move(ws->pending.front());
    ws->pending.pop_front();
    PRL_HIP_CHECK(hipEventSynchronize(ws->ev[pc.slot]));
    if (!pc.has_flags) return PRL_OK;
    const SlotLayout l = slot_layout(pc.n_pages);
    const auto* hg = reinterpret_cast<const PageGlobals*>(static_cast<uint8_t*>(ws->pinned) + (size_t)pc.slot * ws->slot_bytes +
                                                          2 * l.table_bytes);
    CallStats cs;
    cs.seq = pc.seq;
    cs.pixels = pc.pixels;
    // Flagged pages.  Bit 0 alone: the refine queue overflowed - the exact sweep redoes the page (a few times a page's usual
    // cost, not bounded by the literal budget).  Bit 1 (then, or after that sweep): the fix-up list overflowed - the literal pipeline.
    std::vector<int> flagged, second;
    for (int i = 0; i < pc.n_pages; ++i) {
        cs.refined += hg[(size_t)i].n_refined;
        cs.exact += hg[(size_t)i].n_exact;
        cs.wolf_candidates += hg[(size_t)i].n_cand;
        if (hg[(size_t)i].worklist_overflow & 2u) flagged.push_back(i);
        else if (hg[(size_t)i].worklist_overflow) second.push_back(i);
    }
    cs.exact_sweep_pages = second.size();
    ws->last = cs;
    if (!second.empty()) {
        ws->clean_pages = 0;   // (the stream's state is no longer what the last call's epilogue left)
        const int st2 = redo_pages_exact(ws, pc, second, &flagged, &cs);
        if (st2 != PRL_OK) return st2;
        std::sort(flagged.begin(), flagged.end());
    }
    cs.literal_pages = flagged.size();
    ws->last = cs;
    const int budget = literal_budget();
    if (budget >= 0 && (int)flagged.size() > budget) {   // the caller's cost bound: report, do not redo
        set_error_detail(std::to_string(flagged.size()) + " of " + std::to_string(pc.n_pages) + " pages need the literal pipeline, the budget is " +
                         std::to_string(budget) + " (first: page " + std::to_string(flagged.front()) + ")");
        return PRL_ERR_LITERAL_BUDGET;
    }
    return redo_pages_literal(ws, pc, flagged);
}

int resolve_all(StreamWs* ws)
{
    int st = PRL_OK;
    while (!ws->pending.empty()) {
        const int s2 = resolve_front(ws);
        if (st == PRL_OK) st = s2;
    }
    return st;
}

// A pinned slot for this call: [src table][dst table][flags].  Slots are recycled in order; a slot still referenced
// by an unresolved call is resolved first.
int take_slot(StreamWs* ws, int n_pages, int* slot)
{
    const SlotLayout l = slot_layout(n_pages);
    if (ws->slot_bytes < l.total) {
        int st = resolve_all(ws);
        if (st != PRL_OK) return st;
        if (ws->pinned) {
            PRL_HIP_CHECK(hipStreamSynchronize(ws->stream));
            PRL_HIP_CHECK(hipHostFree(ws->pinned));
            ws->pinned = nullptr;
            ws->slot_bytes = 0;
        }
        const size_t bytes = std::max<size_t>(l.total, 1 << 14);
        PRL_HIP_CHECK(hipHostMalloc(&ws->pinned, bytes * kSlots, hipHostMallocDefault));
        ws->slot_bytes = bytes;
    }
    while ((int)ws->pending.size() >= kSlots) {
        int st = resolve_front(ws);
        if (st != PRL_OK) return st;
    }
    *slot = ws->next_slot;
    ws->next_slot = (ws->next_slot + 1) % kSlots;
    if (!ws->ev[*slot]) PRL_HIP_CHECK(hipEventCreateWithFlags(&ws->ev[*slot], hipEventDisableTiming));
    return PRL_OK;
}

int binarize_common(const prl_binarize_params* p, int n_pages, PageSet src, int width, int height,
                    PageSetOut dst, const uint8_t* const* h_src_tab, uint8_t* const* h_dst_tab,
                    hipStream_t stream)
{
    prl_binarize_geometry g;
    int st = geometry_impl(p, width, height, &g);
    if (st != PRL_OK) return st;
    if (n_pages < 0) return PRL_ERR_BAD_ARG;
    if (n_pages == 0) return PRL_OK;
    if ((!src.base && !h_src_tab) || (!dst.base && !h_dst_tab)) return PRL_ERR_BAD_ARG;
    if (src.step < (size_t)width || dst.step < (size_t)g.out_w) return PRL_ERR_BAD_ARG;
    const int morph = p->morph_iterations;

    int dev;
    st = current_device(&dev);
    if (st != PRL_OK) return st;
    DeviceCtx* ctx = device_ctx(dev);
    StreamWs* ws = stream_ws(ctx, stream);
    std::lock_guard<std::mutex> lk(ws->mu);

    const ThrParams tp = make_thr_params(p, g, width, height);

    // small device area: [PageGlobals x n][src table][dst table][fused work area]
    const SlotLayout sl = slot_layout(n_pages);
    const size_t fused_bytes = fused_small_bytes(n_pages);
    const void* small_before = ws->small;
    st = ws_grow(&ws->small, &ws->small_bytes, sl.globals_bytes + 2 * sl.table_bytes + fused_bytes, stream, 1 << 20);
    if (st != PRL_OK) return st;
    const bool was_clean = ws->small == small_before && ws->clean_pages == n_pages;
    ws->clean_pages = 0;   // unknown until this call's own epilogue is enqueued
    auto* small = static_cast<uint8_t*>(ws->small);
    auto* d_globals = reinterpret_cast<PageGlobals*>(small);
    auto* d_src_tab = reinterpret_cast<const uint8_t**>(small + sl.globals_bytes);
    auto* d_dst_tab = reinterpret_cast<uint8_t**>(small + sl.globals_bytes + sl.table_bytes);
    void* d_fused = small + sl.globals_bytes + 2 * sl.table_bytes;

    const bool use_fused = exec_mode() == PRL_MODE_AUTO && fused_supports(tp);
    const bool need_slot = use_fused || h_src_tab || h_dst_tab;
    int slot = -1;
    uint8_t* pin = nullptr;
    if (need_slot) {
        st = take_slot(ws, n_pages, &slot);
        if (st != PRL_OK) return st;
        pin = static_cast<uint8_t*>(ws->pinned) + (size_t)slot * ws->slot_bytes;
    }
    PendingCall pc;
    pc.slot = slot;
    pc.has_flags = use_fused;
    pc.seq = ++ws->seq;
    pc.params = *p;
    pc.width = width;
    pc.height = height;
    pc.n_pages = n_pages;
    pc.src = src;
    pc.dst = dst;
    pc.pixels = (uint64_t)g.out_w * g.out_h * (uint64_t)n_pages;
    if (h_src_tab) {
        pc.src_tab.assign(h_src_tab, h_src_tab + n_pages);
        std::memcpy(pin, h_src_tab, sizeof(void*) * (size_t)n_pages);
        PRL_HIP_CHECK(hipMemcpyAsync(d_src_tab, pin, sizeof(void*) * (size_t)n_pages, hipMemcpyHostToDevice, stream));
        src.table = d_src_tab;
    }
    if (h_dst_tab) {
        pc.dst_tab.assign(h_dst_tab, h_dst_tab + n_pages);
        std::memcpy(pin + sl.table_bytes, h_dst_tab, sizeof(void*) * (size_t)n_pages);
        PRL_HIP_CHECK(hipMemcpyAsync(d_dst_tab, pin + sl.table_bytes, sizeof(void*) * (size_t)n_pages, hipMemcpyHostToDevice, stream));
        dst.table = d_dst_tab;
    }

    const size_t mask_step = ((size_t)g.out_w + 63) / 64 * 64;
    const size_t mask_page = mask_step * (size_t)g.out_h;
    const size_t literal_per_page = literal_scratch_per_page(tp);
    size_t literal_pages_per_chunk = 0;
    if (!use_fused) {
        literal_pages_per_chunk = std::max<size_t>(1, literal_scratch_budget() / literal_per_page);
        literal_pages_per_chunk = std::min<size_t>(literal_pages_per_chunk, (size_t)n_pages);
        st = ws_grow(&ws->scratch, &ws->scratch_bytes, literal_per_page * literal_pages_per_chunk, stream);
        if (st != PRL_OK) return st;
    }
    // With morphology the thresholded mask goes to a workspace buffer first, then the morph kernel writes dst.  When the
    // fused kernel runs and the radius allows it, that buffer is a BIT plane (1/8 B per pixel written and re-read
    // instead of 1 B).
    const bool bit_mask = use_fused && morph != 0 && std::abs(morph) <= morph_bits_max_radius() && !env_knobs().byte_mask;
    const size_t bit_step = ((size_t)g.out_w + 127) / 128 * 16;
    const size_t bit_page = (bit_step * (size_t)g.out_h + 255) / 256 * 256;
    const bool large_morph = morph != 0 && std::abs(morph) > kMorphMaxFusedRadius;
    PageSetOut thr_dst = dst;
    if (morph != 0) {
        const size_t first = bit_mask ? bit_page * (size_t)n_pages : mask_page * (size_t)n_pages;
        st = ws_grow(&ws->mask, &ws->mask_bytes, first + (large_morph ? mask_page * (size_t)n_pages : 0), stream);
        if (st != PRL_OK) return st;
        thr_dst = PageSetOut{};
        thr_dst.base = static_cast<uint8_t*>(ws->mask);
        thr_dst.page_stride = bit_mask ? bit_page : mask_page;
        thr_dst.step = bit_mask ? bit_step : mask_step;
    }

    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    ws->prof_valid = false;
    if (g_profiling.load()) {
        if (!ws->prof_start) {
            PRL_HIP_CHECK(hipEventCreate(&ws->prof_start));
            PRL_HIP_CHECK(hipEventCreate(&ws->prof_stop));
            PRL_HIP_CHECK(hipEventCreate(&ws->call_start));
            PRL_HIP_CHECK(hipEventCreate(&ws->call_stop));
        }
        ev0 = ws->prof_start;
        ev1 = ws->prof_stop;
        PRL_HIP_CHECK(hipEventRecord(ws->call_start, stream));   // everything this call enqueues lies between call_start and call_stop
    }
    // small batches: the fused pipeline's last kernel delivers the flags and re-initialises globals and counters itself
    const bool epilogue = use_fused && n_pages <= kEpilogueMaxPages;
    if (!(epilogue && was_clean)) {
        st = init_globals_run(d_globals, n_pages, stream, use_fused ? d_fused : nullptr);  // (+ the fused pipeline's counter block)
        if (st != PRL_OK) return st;
    }

    if (use_fused) {
        // threshold sweep, float64 interval test of what it left open, literal fix-up of what THAT left open: all enqueued,
        // the last two find their queues on the device and do nothing when they are empty.  Pages whose fix-up queue
        // overflowed are flagged; the flags travel to the pinned slot and are looked at in resolve_front().
        auto* h_globals = reinterpret_cast<PageGlobals*>(pin + 2 * sl.table_bytes);
        const WolfSide* wolf_side = nullptr;
        if (tp.method == PRL_WOLFJOLION && env_knobs().wolf_side) {
            if (!ws->wolf.stream) {
                // built in a local and published only when the stream and all four events exist: a half-made side would be
                // taken for a whole one by the next call (stream set, events null)
                WolfSide side{};
                hipError_t e = hipStreamCreateWithFlags(&side.stream, hipStreamNonBlocking);
                for (hipEvent_t* ev : {&side.ev_fork, &side.ev_min, &side.ev_a, &side.ev_coeff})
                    if (e == hipSuccess) e = hipEventCreateWithFlags(ev, hipEventDisableTiming);
                if (e != hipSuccess) {
                    for (hipEvent_t ev : {side.ev_fork, side.ev_min, side.ev_a, side.ev_coeff}) if (ev) (void)hipEventDestroy(ev);
                    if (side.stream) (void)hipStreamDestroy(side.stream);
                    PRL_HIP_CHECK(e);
                }
                ws->wolf = side;
            }
            wolf_side = &ws->wolf;
        }
        st = fused_run(tp, src, n_pages, thr_dst, d_fused, d_globals, stream, ev0, ev1, bit_mask, true, epilogue ? h_globals : nullptr, wolf_side);
        if (st != PRL_OK) return st;
        if (epilogue) ws->clean_pages = n_pages;
        else PRL_HIP_CHECK(hipMemcpyAsync(h_globals, d_globals, sizeof(PageGlobals) * (size_t)n_pages, hipMemcpyDeviceToHost, stream));
    } else {
        if (tp.method == PRL_WOLFJOLION || tp.method == PRL_FENG) {
            st = page_min_run(tp, src, n_pages, d_globals, stream);
            if (st != PRL_OK) return st;
        }
        if (ev0) PRL_HIP_CHECK(hipEventRecord(ev0, stream));
        for (int first = 0; first < n_pages; first += (int)literal_pages_per_chunk) {
            const int cnt = std::min<int>((int)literal_pages_per_chunk, n_pages - first);
            st = literal_run(tp, src, first, cnt, thr_dst, ws->scratch, d_globals, stream);
            if (st != PRL_OK) return st;
        }
        if (ev1) PRL_HIP_CHECK(hipEventRecord(ev1, stream));
        ws->last = CallStats{pc.seq, pc.pixels, 0, 0, (uint64_t)n_pages};
    }

    if (morph != 0) {
        PageSet msrc{};
        msrc.base = thr_dst.base;
        msrc.page_stride = thr_dst.page_stride;
        msrc.step = thr_dst.step;
        if (bit_mask) {
            st = morph_bitplane_run(morph, msrc, n_pages, g.out_w, g.out_h, dst, stream);
        } else if (!large_morph) {
            st = morph_binary_run(morph, msrc, n_pages, g.out_w, g.out_h, dst, stream);
        } else {  // large radii: chained single-operator passes through one more page-sized buffer
            st = morph_large_run(morph, msrc, n_pages, g.out_w, g.out_h, dst,
                                 static_cast<uint8_t*>(ws->mask) + mask_page * (size_t)n_pages, mask_step, stream);
        }
        if (st != PRL_OK) return st;
    }
    if (ev0) PRL_HIP_CHECK(hipEventRecord(ws->call_stop, stream));
    ws->prof_valid = (ev0 != nullptr);
    t_last.device = dev;
    t_last.stream = stream;
    t_last.seq = pc.seq;
    t_last.valid = true;
    if (need_slot) {
        PRL_HIP_CHECK(hipEventRecord(ws->ev[slot], stream));
        ws->pending.push_back(std::move(pc));
        if (!g_deferred.load() && !t_force_deferred) return resolve_all(ws);  // default: the call is complete (one wait) when it returns
    }
    return PRL_OK;
}

}  // namespace
}  // namespace prl_hip

namespace prl_hip {
DeferredScope::DeferredScope() : prev_(t_force_deferred) { t_force_deferred = true; }
DeferredScope::~DeferredScope() { t_force_deferred = prev_; }
}  // namespace prl_hip

using namespace prl_hip;

namespace {
constexpr int kMaxPagesPerLaunch = 32768;

// Pages one binarize_common call takes: several kernels keep the page index in a grid dimension limited to 65535, and
// Wolf-Jolion stores one maximum per wavefront of a call (fused_max_pages).  Longer batches go in chunks.
int pages_per_call(const prl_binarize_params* p, int width, int height)
{
    prl_binarize_geometry g;
    if (prl_hip::geometry_impl(p, width, height, &g) != PRL_OK) return kMaxPagesPerLaunch;  // the error surfaces in binarize_common
    const ThrParams tp = prl_hip::make_thr_params(p, g, width, height);
    return std::max(1, std::min(kMaxPagesPerLaunch, fused_max_pages(tp)));
}
}

extern "C" {

int prl_hip_abi_version(void) { return PRL_HIP_ABI_VERSION; }

const char* prl_hip_strerror(int status)
{
    switch (status) {
    case PRL_OK: return "ok";
    case PRL_ERR_EMPTY: return "Input image for binarization is empty";
    case PRL_ERR_BAD_WINDOW:
        return "Window size must satisfy the following condition: ( (windowSize > 1) && ((windowSize % 2) == 1) ) ";
    case PRL_ERR_BAD_CHANNELS: return "unsupported channel count";
    case PRL_ERR_EMPTY_RECT: return "processing rectangle is empty (image not larger than the window)";
    case PRL_ERR_BAD_ARG: return "bad argument";
    case PRL_ERR_NO_DEVICE: return "no usable HIP device (gfx950 required); there is no CPU fallback";
    case PRL_ERR_HIP: return "HIP runtime error";
    case PRL_ERR_NOMEM: return "out of memory";
    case PRL_ERR_LITERAL_BUDGET: return "literal-page budget exceeded: the flagged pages are unfinished";
    default: return "unknown status";
    }
}

const char* prl_hip_last_error_detail(void) { return t_error_detail.c_str(); }

int prl_hip_set_literal_page_budget(int max_pages)
{
    g_literal_budget.store(max_pages < 0 ? -1 : max_pages, std::memory_order_relaxed);
    return PRL_OK;
}

int prl_hip_get_literal_page_budget(void) { return literal_budget(); }

int prl_hip_device_count(int* count)
{
    if (!count) return PRL_ERR_BAD_ARG;
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) {
        (void)hipGetLastError();
        c = 0;
    }
    *count = c;
    return PRL_OK;
}

int prl_hip_set_device(int device)
{
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess || device < 0 || device >= c) {
        (void)hipGetLastError();
        return PRL_ERR_NO_DEVICE;
    }
    t_device = device;
    return PRL_OK;
}

int prl_hip_set_exec_mode(int mode)
{
    if (mode != PRL_MODE_AUTO && mode != PRL_MODE_LITERAL) return PRL_ERR_BAD_ARG;
    g_exec_mode.store(mode);
    return PRL_OK;
}

int prl_hip_get_exec_mode(void) { return exec_mode(); }

int prl_hip_set_profiling(int enabled)
{
    g_profiling.store(enabled != 0);
    return PRL_OK;
}

int prl_hip_last_kernel_ms(float* ms)
{
    if (!ms) return PRL_ERR_BAD_ARG;
    *ms = 0.0f;
    if (!t_last.valid) return PRL_ERR_BAD_ARG;
    PRL_HIP_CHECK(hipSetDevice(t_last.device));
    StreamWs* ws = stream_ws(device_ctx(t_last.device), t_last.stream);
    std::lock_guard<std::mutex> lk(ws->mu);
    if (!ws->prof_valid) return PRL_ERR_BAD_ARG;
    PRL_HIP_CHECK(hipEventSynchronize(ws->prof_stop));
    PRL_HIP_CHECK(hipEventElapsedTime(ms, ws->prof_start, ws->prof_stop));
    return PRL_OK;
}

int prl_hip_last_call_ms(float* ms)
{
    if (!ms) return PRL_ERR_BAD_ARG;
    *ms = 0.0f;
    if (!t_last.valid) return PRL_ERR_BAD_ARG;
    PRL_HIP_CHECK(hipSetDevice(t_last.device));
    StreamWs* ws = stream_ws(device_ctx(t_last.device), t_last.stream);
    std::lock_guard<std::mutex> lk(ws->mu);
    if (!ws->prof_valid) return PRL_ERR_BAD_ARG;
    PRL_HIP_CHECK(hipEventSynchronize(ws->call_stop));
    PRL_HIP_CHECK(hipEventElapsedTime(ms, ws->call_start, ws->call_stop));
    return PRL_OK;
}

int prl_hip_set_deferred_completion(int enabled)
{
    g_deferred.store(enabled != 0);
    return PRL_OK;
}

int prl_hip_finish(void* stream)
{
    int dev;
    int st = current_device(&dev);
    if (st != PRL_OK) return st;
    hipStream_t hs = static_cast<hipStream_t>(stream);
    StreamWs* ws = stream_ws(device_ctx(dev), hs);
    {
        std::lock_guard<std::mutex> lk(ws->mu);
        st = resolve_all(ws);
    }
    PRL_HIP_CHECK(hipStreamSynchronize(hs));
    return st;
}

int prl_hip_release_workspace(void)
{
    int dev;
    int st = current_device(&dev);
    if (st != PRL_OK) return st;
    DeviceCtx* ctx = device_ctx(dev);
    std::lock_guard<std::mutex> hlk(ctx->host_mu);   // lock order everywhere: host_mu, stage_mu, mu, ppht_mu
    std::lock_guard<std::mutex> slk(ctx->stage_mu);
    std::lock_guard<std::mutex> lk(ctx->mu);
    std::lock_guard<std::mutex> plk(ctx->ppht_mu);
    std::lock_guard<std::mutex> mlk(ctx->streams_mu);
    for (auto& kv : ctx->streams) {
        std::lock_guard<std::mutex> wl(kv.second->mu);
        (void)resolve_all(kv.second.get());
    }
    PRL_HIP_CHECK(hipDeviceSynchronize());
    for (int i = 0; i < DeviceCtx::kPphtBufs; ++i) {
        if (ctx->ppht_buf[i]) PRL_HIP_CHECK(hipFree(ctx->ppht_buf[i]));
        ctx->ppht_buf[i] = nullptr;
        ctx->ppht_bytes[i] = 0;
    }
    ctx->ppht_rnd_n = 0;
    for (int i = 0; i < 4; ++i) {
        if (ctx->host_buf[i]) PRL_HIP_CHECK(hipFree(ctx->host_buf[i]));
        ctx->host_buf[i] = nullptr;
        ctx->host_buf_bytes[i] = 0;
    }
    host_slots_free(ctx);
    if (ctx->chain_planes) PRL_HIP_CHECK(hipFree(ctx->chain_planes));
    ctx->chain_planes = nullptr;
    ctx->chain_planes_bytes = 0;
    for (auto& kv : ctx->streams) {
        std::lock_guard<std::mutex> wl(kv.second->mu);
        ws_free(kv.second.get());
    }
    if (ctx->scratch) PRL_HIP_CHECK(hipFree(ctx->scratch));
    ctx->scratch = nullptr;
    ctx->scratch_bytes = 0;
    if (ctx->mask) PRL_HIP_CHECK(hipFree(ctx->mask));
    ctx->mask = nullptr;
    ctx->mask_bytes = 0;
    if (ctx->small) PRL_HIP_CHECK(hipFree(ctx->small));
    ctx->small = nullptr;
    ctx->small_bytes = 0;
    ctx->lut_small[0] = ctx->lut_small[1] = nullptr;
    if (ctx->pinned) PRL_HIP_CHECK(hipHostFree(ctx->pinned));
    ctx->pinned = nullptr;
    ctx->pinned_bytes = 0;
    if (ctx->stage) PRL_HIP_CHECK(hipFree(ctx->stage));
    ctx->stage = nullptr;
    ctx->stage_bytes = 0;
    if (ctx->stage_pinned) PRL_HIP_CHECK(hipHostFree(ctx->stage_pinned));
    ctx->stage_pinned = nullptr;
    ctx->stage_pinned_bytes = 0;
    return PRL_OK;
}

int prl_hip_last_stats(prl_binarize_stats* out)
{
    if (!out) return PRL_ERR_BAD_ARG;
    std::memset(out, 0, sizeof(*out));
    if (!t_last.valid) return PRL_OK;
    PRL_HIP_CHECK(hipSetDevice(t_last.device));
    StreamWs* ws = stream_ws(device_ctx(t_last.device), t_last.stream);
    std::lock_guard<std::mutex> lk(ws->mu);
    // the numbers of this thread's last call live in its pinned flag slot: look at it now if nobody has yet
    while (!ws->pending.empty() && ws->pending.front().seq <= t_last.seq) {
        const int st = resolve_front(ws);
        if (st != PRL_OK) return st;
    }
    if (ws->last.seq != t_last.seq) return PRL_OK;  // a later call on this stream has replaced them
    out->pixels = ws->last.pixels;
    out->refined_pixels = ws->last.refined;
    out->exact_pixels = ws->last.exact;
    out->literal_pages = ws->last.literal_pages;
    out->exact_sweep_pages = ws->last.exact_sweep_pages;
    out->wolf_candidates = ws->last.wolf_candidates;
    return PRL_OK;
}

int prl_hip_default_params(int method, prl_binarize_params* out)
{
    if (!out || method < PRL_SAUVOLA || method > PRL_FENG) return PRL_ERR_BAD_ARG;
    std::memset(out, 0, sizeof(*out));
    out->method = method;
    // binarizeSauvola.h:43-47, binarizeNiblack.h:43-47, binarizeWolfJolion.h:43-47
    out->window_size = 101;
    out->k = 0.01;
    out->morph_iterations = 2;
    if (method == PRL_NICK) {  // binarizeNICK.h:43-47
        out->window_size = 21;
        out->k = -0.01;
        out->morph_iterations = 0;
    }
    if (method == PRL_FENG) {  // binarizeFeng.h:46-53
        out->window_size = 21;
        out->k = 0.0;
        out->morph_iterations = 2;
    }
    out->feng_alpha1 = 0.75;
    out->feng_k1 = 0.2;
    out->feng_k2 = 0.03;
    out->feng_gamma = 2.0;
    return PRL_OK;
}

int prl_hip_binarize_geometry(const prl_binarize_params* p, int width, int height,
                              prl_binarize_geometry* out)
{
    return geometry_impl(p, width, height, out);
}

int prl_hip_binarize_batch_device(const prl_binarize_params* p, int n_pages, const uint8_t* d_src,
                                  size_t src_page_stride, size_t src_step, int width, int height,
                                  uint8_t* d_dst, size_t dst_page_stride, size_t dst_step, void* stream)
{
    PageSet s{};
    s.base = d_src;
    s.page_stride = src_page_stride;
    s.step = src_step;
    PageSetOut d{};
    d.base = d_dst;
    d.page_stride = dst_page_stride;
    d.step = dst_step;
    // (prl_hip_last_stats then describes the last chunk)
    const int per_call = pages_per_call(p, width, height);
    for (int first = 0; first < n_pages || first == 0; first += per_call) {
        const int cnt = std::min(per_call, n_pages - first);
        PageSet sc = s;
        PageSetOut dc = d;
        sc.base = d_src ? d_src + (size_t)first * src_page_stride : nullptr;
        dc.base = d_dst ? d_dst + (size_t)first * dst_page_stride : nullptr;
        const int st = binarize_common(p, cnt, sc, width, height, dc, nullptr, nullptr, static_cast<hipStream_t>(stream));
        if (st != PRL_OK || n_pages <= 0) return st;
    }
    return PRL_OK;
}

int prl_hip_binarize_pages_device(const prl_binarize_params* p, int n_pages,
                                  const uint8_t* const* d_src_pages, size_t src_step, int width,
                                  int height, uint8_t* const* d_dst_pages, size_t dst_step, void* stream)
{
    if (n_pages > 0 && (!d_src_pages || !d_dst_pages)) return PRL_ERR_BAD_ARG;
    PageSet s{};
    s.step = src_step;
    PageSetOut d{};
    d.step = dst_step;
    const int per_call = pages_per_call(p, width, height);
    for (int first = 0; first < n_pages || first == 0; first += per_call) {
        const int cnt = std::min(per_call, n_pages - first);
        const int st = binarize_common(p, cnt, s, width, height, d, d_src_pages ? d_src_pages + first : nullptr,
                                       d_dst_pages ? d_dst_pages + first : nullptr, static_cast<hipStream_t>(stream));
        if (st != PRL_OK || n_pages <= 0) return st;
    }
    return PRL_OK;
}

int prl_hip_binarize_host(const prl_binarize_params* p, const uint8_t* src, size_t src_step, int width,
                          int height, uint8_t* dst, size_t dst_step, uint8_t* padded_out,
                          size_t padded_step)
{
    prl_binarize_geometry g;
    int st = geometry_impl(p, width, height, &g);
    if (st != PRL_OK) return st;
    if (!src || !dst || src_step < (size_t)width || dst_step < (size_t)g.out_w) return PRL_ERR_BAD_ARG;
    if (padded_out && padded_step < (size_t)g.padded_w) return PRL_ERR_BAD_ARG;
    int dev;
    st = current_device(&dev);
    if (st != PRL_OK) return st;

    // rows packed tightly on the device (the kernels take any step >= width); 256-byte aligned page starts
    const size_t in_pitch = (size_t)width, out_pitch = (size_t)g.out_w;
    const size_t in_bytes = (in_pitch * (size_t)height + 255) / 256 * 256, out_bytes = out_pitch * (size_t)g.out_h;
    DeviceCtx* ctx = device_ctx(dev);
    std::lock_guard<std::mutex> slk(ctx->stage_mu);  // cached device + pinned staging: no allocation per page
    st = ensure_stage(ctx, in_bytes + out_bytes);
    if (st != PRL_OK) return st;
    st = ensure_stage_pinned(ctx, in_bytes + out_bytes);
    if (st != PRL_OK) return st;
    uint8_t* d_in = static_cast<uint8_t*>(ctx->stage);
    uint8_t* d_out = d_in + in_bytes;
    hipStream_t stream = nullptr;
    DrainOnExit drain_guard{stream};   // (direct DMA from the caller's pinned page: see prl_internal.h)
    st = stage_upload(ctx, 0, src, src_step, (size_t)width, height, d_in, stream);
    if (st != PRL_OK) return st;
    st = prl_hip_binarize_batch_device(p, 1, d_in, in_bytes, in_pitch, width, height, d_out, out_bytes, out_pitch, stream);
    if (st != PRL_OK) return st;
    st = prl_hip_finish(stream);  // (deferred-completion mode: the page is final before it is fetched)
    if (st != PRL_OK) return st;
    st = stage_download(ctx, in_bytes, d_out, (size_t)g.out_w, g.out_h, dst, dst_step, stream);
    if (st != PRL_OK) return st;
    if (padded_out) {
        // cv::copyMakeBorder(in, in, h, h, h, h, BORDER_REPLICATE) side effect on the caller's Mat
        // (binarizeSauvola.cpp:65): pure data movement of the caller's own host pixels.
        for (int y = 0; y < g.padded_h; ++y) {
            const int sy = std::min(std::max(y - g.half, 0), height - 1);
            const uint8_t* s = src + (size_t)sy * src_step;
            uint8_t* d = padded_out + (size_t)y * padded_step;
            std::memset(d, s[0], (size_t)g.half);
            std::memcpy(d + g.half, s, (size_t)width);
            std::memset(d + g.half + width, s[width - 1], (size_t)g.half);
        }
    }
    return PRL_OK;
}

int prl_hip_morph_batch_device(int morph_iterations, int n_pages, const uint8_t* d_src,
                               size_t src_page_stride, size_t src_step, int width, int height,
                               uint8_t* d_dst, size_t dst_page_stride, size_t dst_step, void* stream)
{
    if (width <= 0 || height <= 0) return PRL_ERR_EMPTY;
    if (n_pages < 0 || !d_src || !d_dst || d_src == d_dst) return PRL_ERR_BAD_ARG;
    if (src_step < (size_t)width || dst_step < (size_t)width) return PRL_ERR_BAD_ARG;
    if (n_pages == 0) return PRL_OK;
    if (n_pages > kMaxPagesPerLaunch) {  // the page index sits in a grid dimension limited to 65535
        for (int first = 0; first < n_pages; first += kMaxPagesPerLaunch) {
            const int st2 = prl_hip_morph_batch_device(morph_iterations, std::min(kMaxPagesPerLaunch, n_pages - first),
                                                       d_src + (size_t)first * src_page_stride, src_page_stride, src_step, width,
                                                       height, d_dst + (size_t)first * dst_page_stride, dst_page_stride, dst_step,
                                                       stream);
            if (st2 != PRL_OK) return st2;
        }
        return PRL_OK;
    }
    int dev;
    int st = current_device(&dev);
    if (st != PRL_OK) return st;
    if (morph_iterations == 0) {
        for (int i = 0; i < n_pages; ++i)
            PRL_HIP_CHECK(hipMemcpy2DAsync(d_dst + (size_t)i * dst_page_stride, dst_step,
                                           d_src + (size_t)i * src_page_stride, src_step, (size_t)width,
                                           (size_t)height, hipMemcpyDeviceToDevice,
                                           static_cast<hipStream_t>(stream)));
        return PRL_OK;
    }
    PageSet s{};
    s.base = d_src;
    s.page_stride = src_page_stride;
    s.step = src_step;
    PageSetOut d{};
    d.base = d_dst;
    d.page_stride = dst_page_stride;
    d.step = dst_step;
    if (std::abs(morph_iterations) <= kMorphMaxFusedRadius)
        return morph_run(morph_iterations, s, n_pages, width, height, d, static_cast<hipStream_t>(stream));
    DeviceCtx* ctx = device_ctx(dev);
    std::lock_guard<std::mutex> lk(ctx->mu);
    const size_t tstep = ((size_t)width + 63) / 64 * 64;
    st = ensure_scratch(ctx, tstep * (size_t)height * (size_t)n_pages);
    if (st != PRL_OK) return st;
    hipStream_t hs = static_cast<hipStream_t>(stream);
    if (ctx->last_use) PRL_HIP_CHECK(hipStreamWaitEvent(hs, ctx->last_use, 0));
    else PRL_HIP_CHECK(hipEventCreateWithFlags(&ctx->last_use, hipEventDisableTiming));
    st = morph_large_run(morph_iterations, s, n_pages, width, height, d, static_cast<uint8_t*>(ctx->scratch), tstep, hs);
    if (st != PRL_OK) return st;
    PRL_HIP_CHECK(hipEventRecord(ctx->last_use, hs));
    return PRL_OK;
}

}  // extern "C"
