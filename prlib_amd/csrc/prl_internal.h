// prl_internal.h — shared host-side declarations of libprlib_hip.so (not part of the public ABI).
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <deque>
#include <map>
#include <memory>
#include <condition_variable>
#include <mutex>
#include <vector>
#include <string>

#include "../../include/prl_hip.h"

namespace prl_hip {

// ---- error plumbing -----------------------------------------------------------------------
void set_error_detail(const std::string& s);

#define PRL_HIP_CHECK(expr)                                                                       \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            ::prl_hip::set_error_detail(std::string(#expr) + ": " + hipGetErrorString(_e));       \
            return (_e == hipErrorOutOfMemory) ? PRL_ERR_NOMEM : PRL_ERR_HIP;                     \
        }                                                                                         \
    } while (0)

struct StreamWs;  // per-(device, stream) workspace of the binarizers (prl_capi.hip)

// The *_host entries may start a DMA straight from / into the CALLER's pinned pixels (stage_upload / stage_download).  Whatever
// way such an entry returns - also early, on an error of a later step - nothing may still be reading or writing that memory:
// the stream is drained at scope exit (a no-op after the success path's own wait).
struct DrainOnExit {
    hipStream_t s;
    ~DrainOnExit() { (void)hipStreamSynchronize(s); }
};

// ---- per-device context: cached scratch memory ------------------------------------------------
struct DeviceCtx {
    std::mutex streams_mu;  // guards `streams` only (never held across device work: deskew holds `mu` for seconds)
    std::map<hipStream_t, std::unique_ptr<StreamWs>> streams;  // the entries have their own locks
    ~DeviceCtx();
    std::mutex mu;          // one binarize/denoise call at a time per device (scratch is shared)
    int device = -1;
    void* scratch = nullptr;   // literal pipeline: float64 integral planes
    size_t scratch_bytes = 0;
    void* mask = nullptr;      // thresholded masks waiting for the morphology pass
    size_t mask_bytes = 0;
    void* small = nullptr;  // counters / work lists / per-page globals
    size_t small_bytes = 0;
    std::mutex stage_mu;    // one *_host call at a time per device (taken before `mu`)
    void* stage = nullptr;  // device staging of the *_host entry points (page in + page out)
    size_t stage_bytes = 0;
    void* stage_pinned = nullptr;  // pinned host bounce buffer of the *_host entry points (same lock)
    size_t stage_pinned_bytes = 0;
    std::vector<hipEvent_t> stage_events;  // one per download band (same lock)
    // NL-means weight tables already resident in `small` (slot 0 / 1): rebuilt only when (channels, h) or `small` change
    int lut_channels[2] = {0, 0};
    float lut_h[2] = {0.f, 0.f};
    int lut_n[2] = {0, 0};
    const void* lut_small[2] = {nullptr, nullptr};
    void* pinned = nullptr; // pinned host staging for tiny transfers
    size_t pinned_bytes = 0;
    int cu_count = 0;
    hipEvent_t prof_start = nullptr, prof_stop = nullptr;  // prl_hip_set_profiling
    bool prof_valid = false;
    hipEvent_t last_use = nullptr;  // recorded after each call; the next call's stream waits on it
    hipEvent_t stage_use = nullptr; // same for the staging area (`stage`): recorded by its last user (guarded by stage_mu)
    // The angle search of prl::deskew has its own workspace, lock and stream: in the chain it runs for the next pass
    // while the other stages of the current one use `scratch` / `mask` / `small` (glue.hip).
    std::mutex ppht_mu;
    // fixed part (masks, lists' layout ...), point / segment lists, gray pages, the group kernel's workspace (ppht_group.hip),
    // accumulators in device memory (k_ppht_mw, for the pages the group kernel did not take), cv::RNG's output
    static constexpr int kPphtBufs = 6;
    void* ppht_buf[kPphtBufs] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t ppht_bytes[kPphtBufs] = {0, 0, 0, 0, 0, 0};
    size_t ppht_rnd_n = 0;          // numbers of cv::RNG(-1)'s sequence resident in ppht_buf[5]
    hipStream_t host_run = nullptr; // stream of prl_hip_chain_batch_host's device work (kept: its per-stream workspaces persist)
    // page buffers and pinned bounce slots of prl_hip_chain_batch_host, kept between calls (allocating and freeing tens of
    // gigabytes per call cost two seconds); one such call at a time per device
    std::mutex host_mu;
    void* host_buf[4] = {nullptr, nullptr, nullptr, nullptr};  // in 0 / 1, out 0 / 1
    size_t host_buf_bytes[4] = {0, 0, 0, 0};
    void* host_slots[2] = {nullptr, nullptr};   // PinSlots of host_batch.hip (replaced when they grow, never freed at exit)
    void* host_bin = nullptr;                   // HostBin of host_batch.hip: streams and chunk slots of prl_hip_binarize_batch_host (same lock)
    void* chain_planes = nullptr;   // Lab planes of a whole chain pass (split mode of glue.hip; guarded by stage_mu)
    size_t chain_planes_bytes = 0;
    hipStream_t side = nullptr;     // created on first use (non-blocking)
    hipEvent_t side_ev = nullptr;
};

// Knobs of the PRL_* environment variables, read ONCE (first use) instead of on every call.  The product library reads the
// user-facing ones (PRL_HIP_DEBUG, PRL_HIP_MODE, the memory budgets *_MB, the copy-thread counts); the kernel-selection and
// schedule knobs below keep their defaults unless the library is built with -DPRL_TEST_HOOKS (libprlib_hip_testhooks.so,
// `make hooks`: what tests/ and tools/ load for A/B runs and for forcing rarely taken paths).
struct EnvKnobs {
    int fused_wpb = 1;            // PRL_HIP_WPB          wavefronts per workgroup of k_fused (1..4)
    bool flt = true;              // PRL_HIP_FLT=0        forces the integer sum pipeline everywhere
    bool nt_store = true;         // PRL_HIP_NT=0         plain instead of non-temporal mask stores
    int rows_per_seg = 0;         // PRL_HIP_ROWS_PER_SEG (0 = chosen from the batch size)
    bool tiers = true;            // PRL_HIP_TIERS=0      one segment length for the whole call (A/B of the tiered schedule)
    bool ext_strip = true;        // PRL_HIP_EXT_STRIP=0  no extended last strip (binarize_fused.hip strip_layout)
    int fused_qint = 2;           // PRL_HIP_FUSED_QINT=0 windows of 33..129 columns stay on k_fused's integer loop (1: k_fused_q for interior strips only, 2: border strips too)
    bool ragged_uo = true;        // PRL_HIP_RAGGED_UO=0  outputs per strip always a multiple of 8
    bool wolf_side = true;        // PRL_HIP_WOLF_SIDE=0  Wolf-Jolion on one stream, the threshold sweep after the literal devianceMax (round-3 schedule)
    int wolf_tier_max = 128;      // PRL_HIP_WOLF_TIER_MAX longest row segment of a tiered Wolf-Jolion call (profiles/r03/wolf_tier_max.txt)
    bool debug = false;           // PRL_HIP_DEBUG
    bool byte_mask = false;       // PRL_HIP_BYTE_MASK    byte instead of bit-plane hand-off to the morphology pass
    int morph_rps = 0, morph_wpb = 1;   // PRL_MORPH_RPS, PRL_MORPH_WPB
    int thin_rps = 0, thin_wpb = 4;     // PRL_THIN_RPS, PRL_THIN_WPB
    int nlm_xl = 3;               // PRL_NLM_XL
    int nlm_glut = 0;             // PRL_NLM_GLUT   bit 0 / 1: L / ab plane read the weight table from memory instead of LDS
    size_t literal_scratch_mb = 8192;   // PRL_HIP_LITERAL_SCRATCH_MB
    size_t deskew_work_mb = 24576;      // PRL_HIP_DESKEW_WORK_MB
    int ppht_mw = -1;                   // PRL_HIP_PPHT_MW   1 / 0: always / never three wavefronts per page (default: by batch size)
    int ppht_prio = 3;                  // PRL_HIP_PPHT_PRIO=0   k_ppht does not raise its wavefront priority
    int ppht_group = -1;                // PRL_HIP_PPHT_GROUP    1 / 0: always (where a page qualifies) / never the on-chip group kernel (ppht_group.hip)
    int ppht_group_g = 1;               // PRL_HIP_PPHT_GROUP_G  at least this many workgroups per page
    int ppht_group_xcd = 0;             // PRL_HIP_PPHT_GROUP_XCD=1   a group's members on one XCD (workgroups b, b + 8, ...) instead of consecutive ids: 24 instead of 28 groups of 9, no faster exchange
    int ppht_group_spin_ms = 10000;     // PRL_HIP_PPHT_GROUP_SPIN_MS how long a member waits for its group before the group gives up
    int ppht_group_kill = -1;           // PRL_HIP_PPHT_GROUP_KILL=n  (tests) member 1 of group 0 falls silent after n exchanges
    int ppht_group_cus = 0;             // PRL_HIP_PPHT_GROUP_CUS     workgroups of the group kernel at most (0: one per CU)
    int chain_pass = 0, chain_first_pass = 0;   // PRL_HIP_CHAIN_PASS / PRL_HIP_CHAIN_FIRST_PASS   pages per pass of the chain with deskew (0: start at 192 with denoise, then follow the measured search / NL-means times; else 256)
    int chain_overlap = 2;              // PRL_HIP_CHAIN_OVERLAP   0: passes one after the other; 1: the search of the next pass beside all stages
                                        //                         of this one; 2: beside its NL-means kernels only (head / body / tail, glue.hip)
    size_t chain_work_mb = 49152;       // PRL_HIP_CHAIN_WORK_MB
    size_t host_chunk_mb = 128;         // PRL_HIP_HOST_CHUNK_MB   pages staged per buffer of prl_hip_binarize_batch_host
    int chain_host_pages = 0;           // PRL_HIP_CHAIN_HOST_PAGES   pages per device chunk of prl_hip_chain_batch_host (0: from the budget)
    size_t chain_host_mb = 65536;       // PRL_HIP_CHAIN_HOST_MB      device memory for the page buffers of prl_hip_chain_batch_host
    int fake_devices = 0;               // PRL_HIP_FAKE_DEVICES   (tests) logical devices of the *_batch_host entries, mapped onto the real ones
    int host_copy_threads = 0;          // PRL_HIP_HOST_COPY_THREADS  threads of the pool that copies pageable pages in / out of pinned memory (0: half of the cores, at most 32)
    unsigned segmax_cap = 1u << 20;     // PRL_HIP_SEGMAX_CAP   wavefronts per Wolf-Jolion call (tests shrink it)
    int literal_mode = 0;         // PRL_HIP_MODE=literal
};
const EnvKnobs& env_knobs();

// While alive, binarize calls of this thread only enqueue (as with prl_hip_set_deferred_completion(1)); the owner calls
// prl_hip_finish(stream) itself.
class DeferredScope {
public:
    DeferredScope();
    ~DeferredScope();
private:
    bool prev_;
};

int current_device(int* dev);             // validates that a gfx950 device is usable
DeviceCtx* device_ctx(int dev);
int ensure_scratch(DeviceCtx* ctx, size_t bytes);
int ensure_mask(DeviceCtx* ctx, size_t bytes);
int ensure_small(DeviceCtx* ctx, size_t bytes);
int ensure_pinned(DeviceCtx* ctx, size_t bytes);
int ensure_stage(DeviceCtx* ctx, size_t bytes);  // caller holds stage_mu
// The staging area is shared by every stream of the device: a user makes its stream wait for the previous user's work
// (stage_acquire) before the first write and records its own work at the end (stage_release).  Caller holds stage_mu.
int stage_acquire(DeviceCtx* ctx, hipStream_t stream);
int stage_release(DeviceCtx* ctx, hipStream_t stream);
struct StageRelease {   // records the area's new last use on every exit of the scope, error exits included
    DeviceCtx* c; hipStream_t s;
    ~StageRelease() { (void)stage_release(c, s); }
};
// Host image <-> device staging through the cached pinned bounce buffer (caller holds stage_mu).  hipMemcpy2D from
// pageable memory runs at ~1 GB/s on this stack; row memcpy into pinned memory + one DMA is an order faster.
// `pin_off`: byte offset inside the bounce buffer (so an upload and a download can share it).
int stage_upload(DeviceCtx* ctx, size_t pin_off, const uint8_t* src, size_t src_step, size_t row_bytes, int rows,
                 uint8_t* d_dst, hipStream_t stream);                      // rows packed tightly on the device
int stage_download(DeviceCtx* ctx, size_t pin_off, const uint8_t* d_src, size_t row_bytes, int rows, uint8_t* dst,
                   size_t dst_step, hipStream_t stream);                  // synchronises `stream`
int ensure_stage_pinned(DeviceCtx* ctx, size_t bytes);
bool host_range_pinned(const void* p, size_t bytes);   // pinned host memory: the DMA engines may use it directly

// ---- page addressing: contiguous batch or table of page pointers ---------------------------------
struct PageSet {
    const uint8_t* base = nullptr;   // page i at base + i*page_stride ...
    size_t page_stride = 0;
    const uint8_t* const* table = nullptr;  // ... unless a DEVICE array of page pointers is given
    size_t step = 0;
    __host__ __device__ const uint8_t* page(int i) const {
        return table ? table[i] : base + (size_t)i * page_stride;
    }
};
struct PageSetOut {
    uint8_t* base = nullptr;
    size_t page_stride = 0;
    uint8_t* const* table = nullptr;
    size_t step = 0;
    __host__ __device__ uint8_t* page(int i) const {
        return table ? table[i] : base + (size_t)i * page_stride;
    }
};

// Threshold constants shared by the literal and fused kernels (host-prepared, passed by value).
struct ThrParams {
    int method;
    int w;         // effective window
    int half;
    int width, height;      // unpadded page
    int pw, ph;             // padded page
    int ow, oh;             // output
    double f;      // 1.0 / (double)(w*w)                  binarizeSauvola.cpp:58-59
    double k;
    double a, b;   // Sauvola: k*(1/128), 1-k             binarizeSauvola.cpp:117
    double c1;     // Feng: 1 - alpha1                    binarizeFeng.cpp:133
    double k2;     // Feng
    double gamma;  // Feng
};

// Per-page globals living in device memory (Wolf-Jolion / Feng reductions, fix-up bookkeeping).
struct PageGlobals {
    int imin;                       // min over the page (cv::minMaxLoc(imageInput), binarizeWolfJolion.cpp:116)
    int smax_found;                 // any non-NaN deviation seen
    unsigned long long smax_bits;   // bit pattern of max deviation (>= +0, so integer order == value order)
    double coeff;                   // k / devianceMax             binarizeWolfJolion.cpp:121
    unsigned int n_refined;         // pixels decided by the float64 interval test
    unsigned int n_exact;           // pixels sent to the absolute-integral fix-up
    unsigned int worklist_overflow; // bit 0: the refine queue overflowed (too many pixels inside the float32 band: the exact sweep redoes the page); bit 1: the fix-up list or Wolf-Jolion's candidate list overflowed (the literal pipeline redoes it)
    unsigned int v32max_bits;       // Wolf fused sweep A: float32 bits of the page's largest variance estimate
    unsigned int n_cand;            // Wolf: candidate pixels for the literal devianceMax (sweep B; statistics)
    unsigned int need_literal;      // Wolf: a pixel of this page reached the literal fix-up -> the literal devianceMax is computed (lazily)
    unsigned long long kmax_bits;   // Wolf: exact integer maximum of K = w^2 Q - S^2 over the output region (sweep B)
    double coeff_rel;               // Wolf: |k / devianceMax_literal - coeff| <= coeff_rel |coeff| (coeff = k / (f sqrt(Kmax)) until need_literal)
    unsigned int cand_overflow;     // Wolf: the candidate list overflowed: the literal devianceMax is not available for this page
    unsigned int reserved0;
};

// ---- literal pipeline (binarize_literal.hip) ---------------------------------------------------
size_t literal_scratch_per_page(const ThrParams& tp);
int literal_run(const ThrParams& tp, const PageSet& src, int first_page, int n_pages,
                const PageSetOut& dst, void* scratch, PageGlobals* d_globals, hipStream_t stream);

// ---- fused pipeline (binarize_fused.hip) -------------------------------------------------------
// Counter block at the start of the fused work area: words 0..63 = list lengths and the epilogue's arrival counter, then
// one counter per refine-queue bucket on its own 128-byte line (kRefBuckets of them): every queued pixel costs a device-scope
// atomic, and 4 * 10^5 of them on ONE address serialised (2.8 ns each: the threshold sweep of 256 real 4K scans took 4.27 ms
// instead of 3.18) - a wavefront adds to the bucket its id selects.
constexpr int kRefBuckets = 256;
constexpr int kRefCounterStride = 32;                                             // words
constexpr int kFusedCounterWords = 64 + kRefBuckets * kRefCounterStride;
constexpr size_t kFusedCounterBytes = ((size_t)kFusedCounterWords * 4 + 255) / 256 * 256;
// Wolf-Jolion's side stream (per workspace): the literal devianceMax - candidate sweep, absolute corner sums of the
// candidates, k / devianceMax - and the page minimum of the border bands run here, beside the two big sweeps on the caller's
// stream (see fused_run).
struct WolfSide {
    hipStream_t stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_min = nullptr, ev_a = nullptr, ev_coeff = nullptr;
};
size_t fused_small_bytes(int n_pages);
// bit_out: dst is a bit plane (rows of dst.step bytes, 1 bit per output pixel) instead of a 0/255 byte mask
int fused_run(const ThrParams& tp, const PageSet& src, int n_pages, const PageSetOut& dst,
              void* small, PageGlobals* d_globals, hipStream_t stream, hipEvent_t ev_start,
              hipEvent_t ev_stop, bool bit_out = false, bool counters_zeroed = false,
              PageGlobals* host_globals = nullptr, const WolfSide* wolf_side = nullptr,   // host_globals (pinned, n_pages entries): see FusedParams::ep_host
              bool exact = false);   // exact: the second chance of flagged pages (k_fused_exact: the float64 interval test inline, no queue)
bool fused_supports(const ThrParams& tp);
int fused_max_pages(const ThrParams& tp);  // pages one fused_run call can take (Wolf-Jolion: per-wavefront maxima storage)

// ---- thinning (thin.hip): the C entry plus the option to thin cv::bitwise_not of the source (chain glue) ------
int thin_batch_device(int method, int n_pages, const uint8_t* d_src, size_t src_page_stride, size_t src_step, int width,
                      int height, uint8_t* d_dst, size_t dst_page_stride, size_t dst_step, void* stream, bool invert_input);

// ---- prl::denoise in three parts (nlm.hip; the chain separates its streaming parts from its compute part) ----------------
size_t denoise_plane_bytes(int width, int height);   // per page: L, ab, L', ab'
int denoise_convert_in(DeviceCtx* ctx, int cnt, int channels, const uint8_t* src, size_t src_page_stride, size_t src_step, int width,
                       int height, uint8_t* planes, hipStream_t s);
int denoise_nlm(DeviceCtx* ctx, int cnt, float strength, uint8_t* planes, int width, int height, hipStream_t s);
int denoise_convert_out(DeviceCtx* ctx, int cnt, int channels, const uint8_t* planes, int width, int height, uint8_t* dst,
                        size_t dst_page_stride, size_t dst_step, hipStream_t s);

// ---- deskew (deskew.hip): one pass over `cnt` pages, as its two halves (the chain overlaps them) or in one go ------
int deskew_pages_per_pass(int n_pages, int width, int height);
struct DeskewPlan {                   // what the angle search of a pass hands to its rotation
    std::vector<unsigned char> warp;  // one WarpPage record per page (deskew.hip)
    std::vector<int32_t> wh;          // result size per page
    std::vector<double> angles;       // findAngle's degrees per page
    int max_ow = 0, max_oh = 0;
};
// gray -> Otsu -> HoughLinesP -> vote on `hs` (synchronises it); own workspace, takes ctx->ppht_mu only
// Lets a caller order its own work behind the START of the search's long kernel (the Hough transform): deskew_find records
// `ev` on the search's stream right before it launches that kernel - the event completes when the streaming prelude (gray, Otsu,
// point lists) is through - and then calls launched().  wait() returns true once that has happened, false when the search
// ended without getting there (finish() is called by whoever ran deskew_find).
struct SearchStart {
    hipEvent_t ev = nullptr;
    bool prefer_mw = false;   // this search runs beside compute-bound work that needs the CUs (the chain's NL-means) and would hide behind it: take k_ppht_mw (accumulators in device memory, no LDS) instead of the group kernel
    std::mutex mu;
    std::condition_variable cv;
    bool recorded = false, done = false;
    void reset() { std::lock_guard<std::mutex> lk(mu); recorded = done = false; }
    void launched() { { std::lock_guard<std::mutex> lk(mu); recorded = true; } cv.notify_all(); }
    void finish() { { std::lock_guard<std::mutex> lk(mu); done = true; } cv.notify_all(); }
    bool wait() { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return recorded || done; }); return recorded; }
};
int deskew_find(DeviceCtx* ctx, int cnt, int channels, const uint8_t* src, size_t src_page_stride, size_t src_step, int width,
                int height, DeskewPlan* plan, hipStream_t hs, SearchStart* start = nullptr);
// dark pixels (<= the page's Otsu threshold) per page = the points HoughLinesP will visit; takes ctx->ppht_mu, synchronises hs
int deskew_ink_census(DeviceCtx* ctx, int n_pages, int channels, const uint8_t* src, size_t src_page_stride, size_t src_step, int width,
                      int height, std::vector<unsigned>* points, hipStream_t hs);
// prl::rotate of every page by its angle (copy where none was found); takes ctx->mu
int deskew_apply(DeviceCtx* ctx, const DeskewPlan& plan, int cnt, int channels, const uint8_t* src, size_t src_page_stride,
                 size_t src_step, int width, int height, uint8_t* dst, size_t dst_page_stride, size_t dst_step, hipStream_t hs);
int deskew_pages(DeviceCtx* ctx, int cnt, int channels, const uint8_t* src, size_t src_page_stride, size_t src_step, int width,
                 int height, uint8_t* dst, size_t dst_page_stride, size_t dst_step, int32_t* out_wh, double* angles, hipStream_t hs);
// ppht_group.hip: HoughLinesP's second stage with the accumulator in LDS, a group of workgroups per page.  Device pointers are
// those of ppht_pages' workspace (deskew.hip); the host arrays must stay alive until the stream has been synchronised.
struct PphtGroupIn {
    int n_pages = 0, width = 0, height = 0, threshold = 0, line_length = 0, line_gap = 0;
    const uint8_t* d_mask = nullptr; size_t mask_page = 0;          // byte masks (k_dark_mask)
    const unsigned* d_nz = nullptr; const unsigned long long* d_nzoff = nullptr; const unsigned* d_count = nullptr;
    const float* d_ttab = nullptr; const float* h_ttab = nullptr;
    int* d_lines = nullptr; const unsigned long long* d_lnoff = nullptr; const unsigned* d_cap = nullptr; unsigned* d_nlines = nullptr;
    const unsigned* h_count = nullptr; const unsigned long long* h_nzoff = nullptr;
    std::vector<int> page_list;          // pages to process, heaviest first
    int cu_limit = 0;                    // workgroups at most (0: one per CU)
    hipEvent_t ev[2] = {nullptr, nullptr};   // optional: recorded before / after the group kernel (diagnostics)
    unsigned* status_out = nullptr;      // host, n_pages: 1 = finished by the group kernel (valid after the stream sync)
    unsigned long long* prof_out = nullptr;   // host, 16 per page (optional)
    int geometry_out[4] = {0, 0, 0, 0};  // members per group, groups, workgroups, LDS bytes
    std::vector<unsigned char> keep;     // host tables the asynchronous copies read
};
bool ppht_group_eligible(int width, int height, int threshold);
int ppht_group_run(DeviceCtx* ctx, PphtGroupIn& in, hipStream_t stream);
// glue.hip: pages per pass of the chain and its workspace bytes per page (host_batch.hip sizes device chunks in whole passes)
int chain_pass_layout(const prl_chain_params* cp, int n_pages, int channels, int width, int height, int* pass_pages,
                      size_t* per_page_out, size_t* desk_page_out);
int ensure_buffer(void** buf, size_t* have, size_t bytes);
void host_slots_free(DeviceCtx* ctx);  // host_batch.hip: the pinned bounce slots of prl_hip_chain_batch_host (caller holds host_mu)  // grow-only device buffer (synchronises the device when it grows)

// ---- morphology (morph.hip) ------------------------------------------------------------------
int morph_run(int iterations, const PageSet& src, int n_pages, int width, int height,
              const PageSetOut& dst, hipStream_t stream);
int morph_binary_run(int iterations, const PageSet& src, int n_pages, int width, int height,
                     const PageSetOut& dst, hipStream_t stream);
int morph_large_run(int iterations, const PageSet& src, int n_pages, int width, int height, const PageSetOut& dst,
                    uint8_t* tmp, size_t tmp_step, hipStream_t stream);
int morph_bits_max_radius();
// source = bit plane (rows of bits.step bytes, 1 bit per pixel, bit j of byte i = pixel 8 i + j)
int morph_bitplane_run(int iterations, const PageSet& bits, int n_pages, int width, int height, const PageSetOut& dst,
                       hipStream_t stream);
int pack_mask_run(const uint8_t* src, size_t src_step, int width, int height, uint8_t* bits, size_t bit_step,
                  hipStream_t stream);
constexpr int kMorphMaxFusedRadius = 8;  // morph_run / morph_binary_run handle |iterations| up to this

// ---- page reductions (binarize_literal.hip) ---------------------------------------------------
int page_min_run(const ThrParams& tp, const PageSet& src, int n_pages, PageGlobals* d_globals,
                 hipStream_t stream);
int wolf_coeff_run(const ThrParams& tp, PageGlobals* d_globals, int first_page, int n_pages, hipStream_t stream);

}  // namespace prl_hip
