/*
 * prl_hip.h — C ABI of the MI355X (gfx950) implementation of PRLib's local-adaptive
 * binarization hot path and its NL-means pre-stage.
 *
 * This is the drop-in boundary: everything above it (the prl::binarize*(cv::Mat&, cv::Mat&, ...)
 * wrappers in prlib_amd/csrc/prl/, the ctypes binding in prlib_amd/_capi.py) only marshals
 * pointers, sizes and strides.  No C++ or torch types appear in any signature.
 *
 * Reference interfaces replaced (paths relative to the PRLib tree):
 *   prl::binarizeSauvola     src/binarizations/binarizeSauvola.h:43-47     .cpp:32-134
 *   prl::binarizeNiblack     src/binarizations/binarizeNiblack.h:43-47     .cpp:32-127
 *   prl::binarizeWolfJolion  src/binarizations/binarizeWolfJolion.h:43-47  .cpp:33-148
 *   prl::binarizeNICK        src/binarizations/binarizeNICK.h:43-47        .cpp:33-144
 *   prl::binarizeFeng        src/binarizations/binarizeFeng.h:46-53        .cpp:31-164
 *   prl::denoise             src/denoise/denoiseNLM.h:32                   .cpp:29-32
 *
 * Conventions
 *   - Images are 8-bit, row-major, `step` bytes between row starts (step >= width*channels),
 *     exactly cv::Mat's (data, step, rows, cols) view.
 *   - `*_device` entry points take DEVICE pointers and a hipStream_t (passed as void*; NULL = the
 *     null stream); they enqueue work and return without synchronising unless stated.
 *   - `*_host` entry points take HOST pointers, stage through the device and return when the
 *     result is in the caller's buffer.
 *   - All functions return PRL_OK (0) or a prl_status error; prl_hip_strerror() explains it and
 *     prl_hip_last_error_detail() carries the HIP runtime message for PRL_ERR_HIP.
 *   - Threading: every entry point may be called from any thread.  Calls that name different (device, stream) pairs share
 *     no workspace and overlap; calls on one stream - from one thread or several - are serialised by that stream's
 *     workspace lock and ordered by the stream.  The *_host entries and the stages that use the per-device staging area
 *     (denoise, deskew, backgroundNormalization, the chain) take a per-device lock for their duration.  Process-wide
 *     switches (exec mode, deferred completion, profiling, literal-page budget) are atomics; prl_hip_set_device, the error
 *     detail and prl_hip_last_stats are per thread.  Held by tests/test_concurrency_gpu.py (four threads on their own
 *     streams, two threads on one, 2 000 mixed calls with the device's free memory back at its baseline after
 *     prl_hip_release_workspace).
 *   - The library never falls back to a CPU implementation.  Without a usable gfx950 device every
 *     compute entry point fails with PRL_ERR_NO_DEVICE.
 */
#ifndef PRL_HIP_H_
#define PRL_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PRL_HIP_ABI_VERSION 4   /* 2: prl_chain_params grew deskew / background_normalization; 3: prl_hip_last_call_ms, prl_binarize_stats.wolf_candidates; 4: prl_hip_find_angle_*, prl_deskew_stats */

typedef enum prl_status {
    PRL_OK = 0,
    PRL_ERR_EMPTY = 1,       /* empty input image: reference throws std::invalid_argument (binarizeSauvola.cpp:38-41) */
    PRL_ERR_BAD_WINDOW = 2,  /* !(windowSize > 1 && windowSize odd): std::invalid_argument (binarizeSauvola.cpp:43-47) */
    PRL_ERR_BAD_CHANNELS = 3,/* channel count the reference's cvtColor / NLM would reject */
    PRL_ERR_EMPTY_RECT = 4,  /* processing rectangle has no pixels (reference: cv::Exception from the ROI/filter2D) */
    PRL_ERR_BAD_ARG = 5,     /* null pointer, step < row bytes, negative count, unknown method ... */
    PRL_ERR_NO_DEVICE = 6,   /* no gfx950 device / HIP runtime unusable */
    PRL_ERR_HIP = 7,         /* a HIP call failed; see prl_hip_last_error_detail() */
    PRL_ERR_NOMEM = 8,       /* device or host allocation failed */
    PRL_ERR_LITERAL_BUDGET = 9 /* more pages of the call need the literal redo than prl_hip_set_literal_page_budget() allows:
                                  the masks of those pages are UNFINISHED (every other page is complete); see INTEGRATION.md §3 */
} prl_status;

/* The five local-adaptive binarizers of src/binarizations named by the north star. */
typedef enum prl_method {
    PRL_SAUVOLA = 0,     /* T = m * (1 + k*(s/128 - 1))            binarizeSauvola.cpp:115-118   */
    PRL_NIBLACK = 1,     /* T = m + k*s                             binarizeNiblack.cpp:108       */
    PRL_WOLFJOLION = 2,  /* T = m + (k*s/max(s) - k)*(m - min(I))   binarizeWolfJolion.cpp:115-130 */
    PRL_NICK = 3,        /* T = m + k*sqrt(m*m + s*s)               binarizeNICK.cpp:121-126      */
    PRL_FENG = 4         /* as written in binarizeFeng.cpp:111-142 (Rs aliases s)                */
} prl_method;

/*
 * Parameters of one binarization call; field meaning = the reference's default arguments.
 *   Sauvola / Niblack / WolfJolion: window_size=101, k=0.01, morph_iterations=2
 *   NICK:                            window_size=21,  k=-0.01, morph_iterations=0
 *   Feng: window_size=21, feng_alpha1=0.75, feng_k1=0.2, feng_k2=0.03, feng_gamma=2.0, morph=2
 * prl_hip_default_params() fills these.
 */
typedef struct prl_binarize_params {
    int32_t method;            /* prl_method */
    int32_t window_size;       /* windowSize: must be > 1 and odd (checked before clamping to min(W,H)) */
    double  k;                 /* thresholdCoefficient (unused by Feng) */
    int32_t morph_iterations;  /* >0: dilate^n then erode^n; <0: erode^n then dilate^n; 0: none */
    int32_t reserved0;
    double  feng_alpha1;
    double  feng_k1;           /* dead in the reference (binarizeFeng.cpp:128); carried for signature parity */
    double  feng_k2;
    double  feng_gamma;
} prl_binarize_params;

/* Geometry derived from (params, W, H) exactly as the reference derives it. */
typedef struct prl_binarize_geometry {
    int32_t w;        /* effective window = min(windowSize, min(W,H))      binarizeSauvola.cpp:57 */
    int32_t half;     /* w/2 = replicate padding on each side              binarizeSauvola.cpp:65 */
    int32_t padded_w; /* W + 2*half : size the caller's input Mat ends up with */
    int32_t padded_h; /* H + 2*half */
    int32_t out_w;    /* Sauvola/Niblack: W+2*half-w ; Wolf/NICK/Feng: W-w  (binarizeSauvola.cpp:66 vs binarizeWolfJolion.cpp:69) */
    int32_t out_h;
} prl_binarize_geometry;

/* Execution mode of the binarizers (process-wide; default PRL_MODE_AUTO). */
typedef enum prl_exec_mode {
    PRL_MODE_AUTO = 0,     /* fused sliding-window kernel + exact fix-up of the few undecided pixels */
    PRL_MODE_LITERAL = 1   /* materialised float64 integral images, one literal evaluation per pixel */
} prl_exec_mode;

/* Counters of the last binarize call on the calling thread (for tests and the bench report). */
typedef struct prl_binarize_stats {
    uint64_t pixels;            /* output pixels produced */
    uint64_t refined_pixels;    /* decided by the in-kernel float64 interval test instead of the float32 one */
    uint64_t exact_pixels;      /* decided by the absolute-integral literal evaluation (fix-up kernel) */
    uint64_t literal_pages;     /* pages that ran the full literal pipeline */
    uint64_t wolf_candidates;   /* Wolf-Jolion: pixels whose deviation was evaluated literally to find devianceMax (a lower bound: a wavefront stops counting once its page's list is full) */
    uint64_t exact_sweep_pages; /* pages whose refine queue overflowed and which the exact sweep redid (the float64 interval test inline; was reserved[0]) */
    uint64_t reserved[2];
} prl_binarize_stats;

/* ---- library / device ------------------------------------------------------------------- */

int         prl_hip_abi_version(void);
const char* prl_hip_strerror(int status);
const char* prl_hip_last_error_detail(void);           /* thread-local, never NULL */
int         prl_hip_device_count(int* count);          /* number of visible HIP devices */
/* Device used by subsequent calls of this thread (thread-local, sticky).  Each call then makes it the thread's current HIP
 * device, exactly as hipSetDevice(device) would, and leaves it so: a caller that juggles several devices on one thread sets
 * its own device again afterwards.  Without this call the library follows hipGetDevice(). */
int         prl_hip_set_device(int device);
int         prl_hip_set_exec_mode(int mode);           /* prl_exec_mode */
int         prl_hip_get_exec_mode(void);
int         prl_hip_last_stats(prl_binarize_stats* out);
/* Cost bound for hostile input (process-wide; INTEGRATION.md §3 "What a hostile input costs").  A page whose queues of
 * undecided pixels overflow is redone by the literal pipeline (~50 x the fast path's time per pixel); an exact two-level
 * periodic pattern makes every page of a call do that.  With a budget >= 0 a call that would redo more than `max_pages`
 * pages does not: it returns PRL_ERR_LITERAL_BUDGET (prl_hip_last_stats().literal_pages says how many needed it) and the
 * caller decides where that request runs.  -1 (default; env PRL_HIP_LITERAL_PAGE_BUDGET): no limit.  Results are never
 * approximated: a page is either bit-exact or reported unfinished. */
int         prl_hip_set_literal_page_budget(int max_pages);
int         prl_hip_get_literal_page_budget(void);
int         prl_hip_release_workspace(void);           /* free cached device scratch of the current device */
/* Measurement aid: when enabled, HIP events bracket the dominant kernel of each binarize call
 * (k_fused in PRL_MODE_AUTO, the whole integral+threshold chain in PRL_MODE_LITERAL) on the stream it
 * is launched on; prl_hip_last_kernel_ms() waits for them and returns the elapsed milliseconds. */
int         prl_hip_set_profiling(int enabled);
int         prl_hip_last_kernel_ms(float* ms);
/* Same switch, the WHOLE call: events around everything the last binarize call of this thread enqueued on its stream (every
 * sweep of Wolf-Jolion, interval refinement, literal fix-up, morphology pass, flag copy) - what one prl::binarize*() costs
 * on the device.  Calls that were split into page chunks report their last chunk. */
int         prl_hip_last_call_ms(float* ms);
/* Deferred completion of prl_hip_binarize_*_device (process-wide switch, default off): see the comment there. */
int         prl_hip_set_deferred_completion(int enabled);
/* Completes every binarize call enqueued on `stream` of the current device (flag check, literal redo of overflowing
 * pages) and waits for the stream.  Cheap when nothing is pending. */
int         prl_hip_finish(void* stream);

/* ---- binarizers -------------------------------------------------------------------------- */

int prl_hip_default_params(int method, prl_binarize_params* out);

/* Validation + geometry; returns the status the reference's argument checks imply. */
int prl_hip_binarize_geometry(const prl_binarize_params* p, int width, int height,
                              prl_binarize_geometry* out);

/*
 * Binarize n_pages single-channel pages of equal size that already live in device memory.
 *   d_src        first page; page i starts at d_src + i*src_page_stride
 *   d_dst        first output page (out_w x out_h, see prl_hip_binarize_geometry); values {0,255}
 *   stream       hipStream_t or NULL
 * Everything is enqueued on `stream` (threshold sweep, interval refinement, literal fix-up - the last two find their
 * work lists on the device and do nothing when they are empty - and the morphology pass).  One thing needs the host:
 * a page with more than 2^17 pixels within ~1e-6 of their threshold (pathological input) is flagged and redone by the
 * literal pipeline.  Default: the call waits for its own work once, looks at the flags and returns with the result
 * complete.  With prl_hip_set_deferred_completion(1) it returns right after enqueuing; the flags are looked at later
 * (a later call that needs the slot, prl_hip_last_stats, prl_hip_finish), so the caller must call
 * prl_hip_finish(stream) before consuming the masks.  Calls on different streams of a device use separate
 * workspaces and overlap.
 * Replaces the body of prl::binarize{Sauvola,Niblack,WolfJolion,NICK,Feng} after cvtColor.
 */
int prl_hip_binarize_batch_device(const prl_binarize_params* p, int n_pages,
                                  const uint8_t* d_src, size_t src_page_stride, size_t src_step,
                                  int width, int height,
                                  uint8_t* d_dst, size_t dst_page_stride, size_t dst_step,
                                  void* stream);

/* Same, pages addressed through host arrays of device pointers (pages need not be contiguous). */
int prl_hip_binarize_pages_device(const prl_binarize_params* p, int n_pages,
                                  const uint8_t* const* d_src_pages, size_t src_step,
                                  int width, int height,
                                  uint8_t* const* d_dst_pages, size_t dst_step,
                                  void* stream);

/*
 * One page from/to host memory (what the cv::Mat wrapper calls).  `src` is 1-channel.
 * If padded_out != NULL it receives the replicate-padded gray image (padded_w x padded_h) the
 * reference leaves in the caller's input Mat (binarizeSauvola.cpp:65).
 */
int prl_hip_binarize_host(const prl_binarize_params* p,
                          const uint8_t* src, size_t src_step, int width, int height,
                          uint8_t* dst, size_t dst_step,
                          uint8_t* padded_out, size_t padded_step);

/*
 * A list of n_pages equal-size 1-channel pages in HOST memory (what a caller holding n cv::Mat has,
 * samples/binarizations/binarizeSauvola_sample.cpp:48-53), sharded over the devices of the node: contiguous blocks
 * (prl_hip_page_range); per device a three-stage pipeline (upload / kernels / download of neighbouring chunks of pages on
 * three streams, chunk slots kept between calls), results in the caller's buffers in the caller's order.
 * n_devices: 0 = every visible device.  No collective; returns when all pages are done.
 */
int prl_hip_binarize_batch_host(const prl_binarize_params* p, int n_pages, const uint8_t* const* src, size_t src_step,
                                int width, int height, uint8_t* const* dst, size_t dst_step, int n_devices);

/*
 * Pinned host memory for a caller's pages.  The reference's callers hold pages as cv::Mat (pageable memory,
 * samples/binarizations/binarizeSauvola_sample.cpp:48); a Mat header over memory from prl_hip_alloc_host -
 * cv::Mat(rows, cols, CV_8UC1, ptr, step) - or over memory pinned in place with prl_hip_host_register is moved by the
 * DMA engines directly: prl_hip_binarize_batch_host detects such pages (hipPointerGetAttributes) and skips its bounce
 * buffers and the two CPU copies per page.  Pageable pages keep working (bounce path).
 * prl_hip_host_register page-locks `bytes` at `p` (costs about one copy of them; pays from the second call on).
 */
int prl_hip_alloc_host(size_t bytes, void** out);
int prl_hip_free_host(void* p);
int prl_hip_host_register(void* p, size_t bytes);
int prl_hip_host_unregister(void* p);

/* The block of a list of n_items that part `part` of `n_parts` owns (sizes differ by at most one): the split used by
 * prl_hip_binarize_batch_host over devices and by one-process-per-GPU launchers over ranks. */
int prl_hip_page_range(int n_items, int n_parts, int part, int* first, int* count);

/* (2n+1)x(2n+1) rectangular closing (n>0) / opening (n<0) with out-of-image pixels ignored:
 * the cv::dilate/cv::erode pair at binarizeSauvola.cpp:125-134.  In place is NOT allowed. */
int prl_hip_morph_batch_device(int morph_iterations, int n_pages,
                               const uint8_t* d_src, size_t src_page_stride, size_t src_step,
                               int width, int height,
                               uint8_t* d_dst, size_t dst_page_stride, size_t dst_step,
                               void* stream);

/* ---- NL-means denoise (prl::denoise -> cv::fastNlMeansDenoisingColored) -------------------- */

/*
 * Non-local means on interleaved 8-bit planes, template 7x7, search 21x21, integer SSD and
 * fixed-point weights as in OpenCV's FastNlMeansDenoisingInvoker.
 *   channels 1: the L plane (h = strength); channels 2: the ab planes (h = 3);
 *   channels 3: treated as three jointly weighted planes (cv::fastNlMeansDenoising on 8UC3).
 */
int prl_hip_nlm_planes_device(int n_pages, int channels, float h,
                              const uint8_t* d_src, size_t src_page_stride, size_t src_step,
                              int width, int height,
                              uint8_t* d_dst, size_t dst_page_stride, size_t dst_step,
                              void* stream);

/*
 * prl::denoise on BGR (channels=3) or BGRA (channels=4) device pages:
 * LBGR->Lab, NLM(L, strength), NLM(ab, 3), Lab->LBGR  (denoiseNLM.cpp:31).
 */
int prl_hip_denoise_batch_device(int n_pages, int channels, float strength,
                                 const uint8_t* d_src, size_t src_page_stride, size_t src_step,
                                 int width, int height,
                                 uint8_t* d_dst, size_t dst_page_stride, size_t dst_step,
                                 void* stream);

int prl_hip_denoise_host(int channels, float strength,
                         const uint8_t* src, size_t src_step, int width, int height,
                         uint8_t* dst, size_t dst_step);

/* ---- thinning (prl::thinZhangSuen / prl::thinGuoHall) --------------------------- */

typedef enum prl_thin_method {
    PRL_THIN_ZHANGSUEN = 0,  /* src/thinning/thinZhangSuen.cpp:15-108 */
    PRL_THIN_GUOHALL = 1     /* src/thinning/thinGuoHall.cpp:15-107   */
} prl_thin_method;

/*
 * Iterative thinning of 1-channel 8-bit pages: foreground = pixels with bit 0 set (the reference's `&= 1`),
 * output 0 / 255.  Replaces the body of prl::thinZhangSuen / prl::thinGuoHall after cvtColor.  d_src == d_dst is
 * allowed.  Synchronises the stream every few passes to read the convergence flags.
 */
int prl_hip_thin_batch_device(int method, int n_pages, const uint8_t* d_src, size_t src_page_stride, size_t src_step,
                              int width, int height, uint8_t* d_dst, size_t dst_page_stride, size_t dst_step,
                              void* stream);

int prl_hip_thin_host(int method, const uint8_t* src, size_t src_step, int width, int height,
                      uint8_t* dst, size_t dst_step);

/* ---- channel adapters and the device-resident chain ---------------------- */

/*
 * cv::cvtColor(src, dst, cv::COLOR_BGR2GRAY) on 8-bit BGR (channels = 3) / BGRA (4) pages already in device
 * memory: the first call of every binarizer for a colour input (src/binarizations/binarizeSauvola.cpp:51, same
 * line in the other four; src/thinning/thinZhangSuen.cpp:78).  14-bit fixed point, bit-exact.
 */
int prl_hip_bgr2gray_batch_device(int n_pages, int channels, const uint8_t* d_src, size_t src_page_stride,
                                  size_t src_step, int width, int height, uint8_t* d_dst, size_t dst_page_stride,
                                  size_t dst_step, void* stream);

/* cv::cvtColor(COLOR_GRAY2BGR / GRAY2BGRA): what a caller needs in front of prl::denoise, which only accepts
 * 3/4-channel input (src/denoise/denoiseNLM.cpp:31 -> fastNlMeansDenoisingColored). */
int prl_hip_gray2bgr_batch_device(int n_pages, int channels, const uint8_t* d_src, size_t src_page_stride,
                                  size_t src_step, int width, int height, uint8_t* d_dst, size_t dst_page_stride,
                                  size_t dst_step, void* stream);

/* cv::bitwise_not on 1-channel pages (d_src == d_dst allowed): the binarizers emit white = background while
 * prl::thinZhangSuen thins white (src/thinning/thinZhangSuen.cpp:85). */
int prl_hip_invert_batch_device(int n_pages, const uint8_t* d_src, size_t src_page_stride, size_t src_step,
                                int width, int height, uint8_t* d_dst, size_t dst_page_stride, size_t dst_step,
                                void* stream);

#define PRL_CHAIN_NO_THINNING (-1)

/* One page through prl::deskew -> prl::denoise -> prl::backgroundNormalization -> cvtColor -> prl::binarize* ->
 * bitwise_not -> prl::thin* (BASELINE config 5; every stage but the binarizer optional).  Not a function of the
 * reference: a caller writes these calls one after the other; here the intermediates stay in device memory. */
typedef struct prl_chain_params {
    int denoise;                   /* != 0: prl::denoise(denoise_strength); needs a 3/4-channel input */
    float denoise_strength;        /* src/denoise/denoiseNLM.h:32 default 5.5 */
    prl_binarize_params binarize;  /* which binarizer and its arguments */
    int thin;                      /* PRL_CHAIN_NO_THINNING, PRL_THIN_ZHANGSUEN or PRL_THIN_GUOHALL */
    int deskew;                    /* != 0: prl::deskew first (per-page result sizes: prl_hip_chain_pages_device) */
    int background_normalization;  /* != 0: prl::backgroundNormalization after the denoise stage (4 channels become 3) */
} prl_chain_params;

void prl_hip_default_chain_params(prl_chain_params* out);

/* d_dst: out_w x out_h bytes per page (prl_hip_binarize_geometry): the binarizer's mask, or, with thinning, the
 * skeleton of the dark strokes (white on black).  Synchronises `stream` where the stages do.  params->deskew must be 0
 * here (a deskewed page has its own size). */
int prl_hip_chain_batch_device(const prl_chain_params* params, int n_pages, int channels, const uint8_t* d_src,
                               size_t src_page_stride, size_t src_step, int width, int height, uint8_t* d_dst,
                               size_t dst_page_stride, size_t dst_step, void* stream);

/* Largest result a page of width x height can have through the chain (with deskew: a max(width,height) square in, so
 * the binarizer's geometry of that). */
int prl_hip_chain_max_out_size(const prl_chain_params* params, int width, int height, int* out_w, int* out_h);

/* The chain with per-page result sizes (needed as soon as params->deskew is set): page i's result is
 * out_wh[2i] x out_wh[2i+1] bytes (host array), written at dst_step bytes per row into a d_dst page with room for
 * prl_hip_chain_max_out_size; angles (host, optional) receives findAngle's degrees per page.  Synchronises. */
int prl_hip_chain_pages_device(const prl_chain_params* params, int n_pages, int channels, const uint8_t* d_src,
                               size_t src_page_stride, size_t src_step, int width, int height, uint8_t* d_dst,
                               size_t dst_page_stride, size_t dst_step, int32_t* out_wh, double* angles, void* stream);

/* The same chain on a list of HOST pages of one size (the cv::Mats of a caller that loops over prl::deskew, prl::denoise,
 * prl::backgroundNormalization, prl::binarizeSauvola, prl::thinZhangSuen page by page, BASELINE config 5), sharded over the
 * first n_devices GPUs (0 = all visible) in contiguous blocks (prl_hip_page_range), no collective.  Per device: a worker
 * thread, chunks of PRL_HIP_CHAIN_HOST_PAGES pages, the upload of chunk k+1 and the download of chunk k-1 overlapped with the
 * chain on chunk k.  dst[i]: room for prl_hip_chain_max_out_size at dst_step >= that width; out_wh[2i], out_wh[2i+1]: the
 * size of page i's result; angles: optional.  Returns when every page is in the caller's memory. */
int prl_hip_chain_batch_host(const prl_chain_params* params, int n_pages, int channels, const uint8_t* const* src,
                             size_t src_step, int width, int height, uint8_t* const* dst, size_t dst_step, int32_t* out_wh,
                             double* angles, int n_devices);

/* ---- background normalisation (prl::backgroundNormalization) ------------------------------ */

/*
 * prl::backgroundNormalization(const cv::Mat&, cv::Mat&) (src/backgroundNormalization.cpp:36-61) =
 * Leptonica's pixBackgroundNormSimple(pixs, NULL, NULL) between prl::opencvToLeptonica / prl::leptonicaToOpenCV
 * (src/formatConvert.cpp:38-218): channels 1 -> 1-channel result, 3 or 4 -> 3-channel result (the fourth byte is
 * dropped by the reference's converter).  d_dst rows hold width * prl_hip_bgnorm_out_channels(channels) bytes.
 * Enqueues on `stream`, no synchronisation.
 */
int prl_hip_bgnorm_out_channels(int channels);
int prl_hip_bgnorm_batch_device(int n_pages, int channels, const uint8_t* d_src, size_t src_page_stride, size_t src_step,
                                int width, int height, uint8_t* d_dst, size_t dst_page_stride, size_t dst_step, void* stream);
int prl_hip_bgnorm_host(int channels, const uint8_t* src, size_t src_step, int width, int height, uint8_t* dst,
                        size_t dst_step);

/* ---- local-variance binarizers ------------------------------------------------------------- */

/*
 * prl::binarizeByLocalVariances(in, out, varianceThresholdCoeff = 0.125, minResultVariance = 25, gamma = 2.0)
 * (with_filters != 0; src/binarizations/binarizeByLocalVariances.cpp:13-145) and
 * prl::binarizeByLocalVariancesWithoutFilters(in, out, varianceThresholdCoeff = 0.125, minResultVariance = 10)
 * (with_filters == 0; :148-292, gamma ignored) on 8-bit 3-CHANNEL pages (the reference reads three variance planes);
 * d_dst: width x height bytes, 0 / 255.  Enqueues on `stream`.  The filtered variant evaluates float32 log / exp / pow
 * (cv::log, cv::exp in the reference): identical results across libraries are not defined, see DESIGN.md.
 */
int prl_hip_binarize_lv_batch_device(int n_pages, int with_filters, double coeff, int min_result_variance, double gamma,
                                     const uint8_t* d_src, size_t src_page_stride, size_t src_step, int width, int height,
                                     uint8_t* d_dst, size_t dst_page_stride, size_t dst_step, void* stream);
int prl_hip_binarize_lv_host(int with_filters, double coeff, int min_result_variance, double gamma, const uint8_t* src,
                             size_t src_step, int width, int height, uint8_t* dst, size_t dst_step);

/* ---- deskew / rotate (prl::deskew, prl::rotate) ----------------------------------------------- */

/* Size of prl::rotate's result (src/rotate.cpp:35-72): transposed for 90 / 270 degrees, unchanged for 180, else a square
 * of side max(width, height). */
int prl_hip_rotate_out_size(int width, int height, double angle, int* out_w, int* out_h);

/* prl::rotate(input, output, angles[i]) per page; d_dst pages need room for the largest result, rows of dst_step bytes.
 * 1..4 channels.  In place is not allowed. */
int prl_hip_rotate_batch_device(int n_pages, int channels, const double* angles, const uint8_t* d_src, size_t src_page_stride,
                                size_t src_step, int width, int height, uint8_t* d_dst, size_t dst_page_stride, size_t dst_step,
                                void* stream);

/* cv::HoughLinesP(image, lines, 1, CV_PI/180, threshold, line_length, line_gap) on one 1-channel device page, the call
 * of prl::findAngle (src/deskew/deskew.cpp:148); segments as (x0, y0, x1, y1) into the host array `lines` (4*cap ints),
 * *n_lines = number found (may exceed cap).  Synchronises. */
int prl_hip_houghp_device(const uint8_t* d_image, size_t step, int width, int height, int threshold, int line_length,
                          int line_gap, int32_t* lines, int cap, int* n_lines, void* stream);

/* Diagnostics of the HoughLinesP searches (prl::deskew, prl::findAngle, the chain's deskew stage, prl_hip_houghp_device)
 * accumulated process-wide since the last reset - the chain searches on a helper thread, so these are not per thread.
 * The point and segment lists are sized per page from the page's ink before the search: `segment_capacity` is the room the
 * segment lists had, `min_page_headroom` the smallest (capacity - segments found) of any page.  A page that needed more
 * room than it had makes its call fail with PRL_ERR_NOMEM ("HoughLinesP: segment list overflow"): nothing is clipped
 * silently, and a negative headroom is only ever seen together with that error. */
typedef struct prl_deskew_stats {
    uint64_t pages;              /* pages searched */
    uint64_t points;             /* non-zero pixels handed to HoughLinesP */
    uint64_t segments;           /* segments found */
    uint64_t segment_capacity;   /* room of the segment lists */
    uint64_t max_page_points;
    uint64_t max_page_segments;
    int64_t  min_page_headroom;  /* min over pages of capacity - segments */
    uint64_t reserved;
} prl_deskew_stats;
int prl_hip_last_deskew_stats(prl_deskew_stats* out);
int prl_hip_reset_deskew_stats(void);

/*
 * prl::findAngle (src/deskew/deskew.h:62, src/deskew/deskew.cpp:139-205) on 1-channel pages (the thresholded page prl::deskew
 * hands it, :226; any 8-bit page is accepted, its points are the pixels != 255 as after the reference's bitwise_not):
 * HoughLinesP(~page, 1, CV_PI/180, 100, width/8.f, 20), atan2 per segment, first-fit clusters of 0.01 rad, the most
 * populated cluster's first angle in degrees; 0.0 when no segment was found.  angles (host, one per page) is required,
 * n_segments (host, optional) receives the number of segments HoughLinesP found.  Synchronises.
 */
int prl_hip_find_angle_batch_device(int n_pages, const uint8_t* d_image, size_t page_stride, size_t step, int width, int height,
                                    double* angles, int32_t* n_segments, void* stream);
int prl_hip_find_angle_host(const uint8_t* src, size_t src_step, int width, int height, double* angle, int32_t* n_segments);

/*
 * prl::deskew (src/deskew/deskew.cpp:208-251) on n_pages device pages of 1, 3 or 4 channels: gray -> Otsu -> findAngle
 * (HoughLinesP + angle vote) -> prl::rotate.  Page i's result is out_wh[2i] x out_wh[2i+1] pixels (host array):
 * max(width,height)^2 when an angle was found, width x height otherwise.  d_dst pages need room for max(width,height)
 * rows of dst_step >= max(width,height) * channels bytes.  angles (host, optional) receives findAngle's degrees.
 * The orientation step (:238, Leptonica) is a no-op for the page the reference hands it; see DESIGN.md.  Synchronises.
 */
int prl_hip_deskew_batch_device(int n_pages, int channels, const uint8_t* d_src, size_t src_page_stride, size_t src_step,
                                int width, int height, uint8_t* d_dst, size_t dst_page_stride, size_t dst_step,
                                int32_t* out_wh, double* angles, void* stream);

/* Host-image forms of the two (what the cv::Mat wrappers call): prl_hip_rotate_host's dst holds the size
 * prl_hip_rotate_out_size reports; prl_hip_deskew_host's dst has room for max(width,height)^2 pixels and *out_w x *out_h
 * tells which part was written. */
int prl_hip_rotate_host(int channels, double angle, const uint8_t* src, size_t src_step, int width, int height, uint8_t* dst,
                        size_t dst_step);
int prl_hip_deskew_host(int channels, const uint8_t* src, size_t src_step, int width, int height, uint8_t* dst, size_t dst_step,
                        int* out_w, int* out_h, double* angle);

#ifdef __cplusplus
}
#endif
#endif /* PRL_HIP_H_ */
