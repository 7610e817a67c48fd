// binarize_fused.hip — the fast path: one pass over the page, no integral image in memory.
//
// What the reference computes per output pixel (src/binarizations/binarizeSauvola.cpp:72-122 and the
// same lines of Niblack/NICK/Feng) depends on the page only through two window sums over the
// replicate-padded image P:
//     S = sum of P   over rows y+1..y+w-1, cols x+1..x+w-1   ((w-1)x(w-1) window)
//     Q = sum of P*P over the same window
// plus last-bit rounding noise of the float64 4-tap filter on ABSOLUTE integral values.  The kernel
// therefore
//   1. keeps exact integer S and Q with a sliding window: each lane owns 8 adjacent columns and
//      carries their vertical (w-1)-row sums in registers while the wavefront walks down the rows;
//      the horizontal (w-1)-column sum is a wave-level prefix scan (DPP) + one cross-lane fetch;
//   2. evaluates the threshold in float32 with a proven error bound eps1 (fused_bounds below) and
//      decides every pixel whose margin |T - (p - 0.5)| exceeds it;
//   3. re-evaluates the few remaining pixels in float64 with a per-pixel interval that also covers
//      the literal sequence's rounding noise;
//   4. queues what is still undecided (|T - (p-0.5)| ~< 1e-6) for k_corner_partial / k_fixup_final, which
//      rebuild the absolute integral corners from the page and run the literal float64 sequence.
// Steps 2-4 make the output bit-identical to the literal pipeline (binarize_literal.hip) while
// >99.9999 % of pixels only pay for step 2.
//
// Memory: 1 B/px read + 1 B/px written to HBM (the algorithmic 2 B/px); window halos, the "leaving" row of
// the sliding window and the compared-pixel row are re-read through L2 / Infinity Cache (measured 20 GB of
// fabric traffic per 8.6 GB algorithmic).  Roofline by the metric: HBM (0.335 of 8 TB/s on the driver's run, 0.365 on the
// fastest box).  The row loop is balanced against four pipes, none saturated: per 512-column wavefront-row of an interior
// strip 149 vector instructions (117 two-cycle, 24 four-cycle, 8 v_sqrt_f32: ~450 SIMD cycles of issue; edge strips 199,
// ~650), 42 scalar ones, 14 ds_bpermute, 6 vector-memory instructions, against 588 measured cycles (GRBM_GUI_ACTIVE: the
// chip runs this kernel at ~1.8 GHz under its power cap): vector issue ~0.84, vector memory 0.65-0.98, LDS 0.57 of the time
// (tools/isa_budget.py reads the instruction mix from the compiler's output; costs in profiles/r01/valu_issue_costs.txt).
// Perfect overlap would end at ~0.43 of the HBM roofline; see DESIGN.md 4.1.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "prl_device_math.h"
#include "prl_internal.h"

namespace prl_hip {

namespace {

constexpr int kWave = 64;
constexpr int CPL = 8;                 // columns per lane
constexpr int SW = kWave * CPL;        // padded columns per wavefront strip
constexpr unsigned kRefBucketCap = 1u << 13;                      // entries per refine-queue bucket (a wavefront queues into bucket wid % kRefBuckets)
constexpr unsigned kRefineCap = kRefBuckets * kRefBucketCap;      // pixels the float32 test may leave undecided per call (2^21)
constexpr unsigned kWorkCap = 1u << 17;    // pixels the float64 interval test may leave undecided (more: literal page); also Wolf-Jolion's candidates
constexpr unsigned kPageMajorMin = 64;     // from this many queued pixels on, their corner sums are built page by page (k_corner_rows)
constexpr int kPageMajorMaxW = 8192;       // ... for pages a workgroup of k_corner_rows covers with 32 columns per thread
constexpr int kRowChunks = 16;             // workgroups per page of k_corner_rows (each: 2 or 4 wavefronts taking rows in turn)
constexpr size_t kSegmaxCap = 1u << 20;    // wavefronts per call whose sweep-A maxima can be kept (Wolf)

struct RefItem {   // undecided after the float32 test: the window sums travel with the pixel
    int page, y, x;
    unsigned S, Q, p;   // p bit 31 (kRefApprox): Q is the float32 pipeline's sum, within FusedParams::flt_dq of the exact one
};
constexpr unsigned kRefApprox = 0x80000000u;
struct WorkItem {  // undecided after the float64 interval test
    int page, y, x;
    int pad;
};
struct CornerAcc {   // absolute integral corners of a queued pixel, accumulated by k_corner_partial
    unsigned long long a[8];  // [0..3] sums of P over top-left, top-right, bottom-left, bottom-right; [4..7] of P*P
};

// pseudo-methods of the two extra Wolf-Jolion sweeps (max deviation search)
constexpr int kWolfMax = 100;      // sweep A: float32 variance maximum per page and per wavefront
constexpr int kWolfCollect = 101;  // sweep B: queue every pixel whose variance could be the literal maximum

constexpr unsigned kSBias = 0x4B000000u;  // float 2^23: window sums S <= 255 * 256^2 < 2^23 ride in its mantissa
constexpr float kZ = 1073741824.0f;  // 2^30: every float32 threshold quantity is carried times Z (exact scaling)

struct PageK {  // per-page constants of the float32 test
    float c1;    // Wolf: f * k / devianceMax ; Wolf sweep B: candidate threshold on K~
    float imin;  // Wolf: Z * cv::minMaxLoc(imageInput) minimum
    float p0;    // P2 = fma(p, Z, p0) = Z (p - 0.5) [Feng: minus Z (k2*Imin - Imin)]
    float eps1;  // Z * decision margin (Wolf: page dependent through k / devianceMax)
};

struct FusedParams {
    ThrParams tp;
    int uo;            // useful output columns per strip (multiple of 8)
    int n_strips;
    // Row segments in TIERS: the first tier has the longest segments (few (w-1)-row warm-ups), the last the shortest
    // (a short tail when the chip drains); workgroups are dispatched in index order, so tier 0's wavefronts start first.
    // Tier k: `segs` segments of `rows` rows per page and strip, starting at output row `row0`; `waves` = pages * strips *
    // segs of them, numbered from `first` (the canonical wavefront id; Wolf-Jolion's per-wavefront maxima are indexed by it).
    struct Tier { int rows, segs, row0; unsigned waves, first; } tier[8];
    int n_tiers;
    int lane_off;      // (w-1) / 8
    int ext;           // the last strip of a row is an extended one (strip_layout)
    unsigned total_waves;
    unsigned xcd_waves;  // wavefront slots per XCD share of the grid (sum over tiers of ceil(waves / 8))
    float w2f;         // (float)(w*w), exact
    float c0, c1;      // method constants in float32, pre-multiplied by f and Z (see eval32)
    float eps1;        // Z * float32 decision margin (covers float32 evaluation + literal rounding noise)
    float vthr32;      // floor on K~ = w^2 Q - S^2 (= variance floor / f^2) below which the float32 test is not trusted
    double vthr;       // same floor for the float64 interval test
    double Em, Eq;     // |m_literal - f*S| <= Em, |q_literal - f*Q| <= Eq
    unsigned ref_cap, wl_cap;   // (ref_cap: total capacity of the refine queue, kRefBuckets x kRefBucketCap; wl_cap: fix-up / candidate lists)
    unsigned cand_page_cap;     // Wolf-Jolion sweep B: candidates one page may queue before it is marked cand_overflow
    int need_p0;       // T may be negative: mask bytes of p == 0 pixels must be cleared explicitly
    float es_max;      // Wolf: bound on |s_literal - s*| for v* >= vthr (enters eps1 scaled by |k/devianceMax|)
    float rho;         // Wolf: relative error bound of the float32 variance v~
    float ev2;         // Wolf: 2 * Ev (literal variance noise)
    float* segmax;     // Wolf: per-wavefront maximum of v~ (sweep A -> sweep B)
    int nt_store;      // non-temporal mask stores (on unless PRL_HIP_NT=0)
    int bit_out;       // the mask is written as a bit plane (1 bit per pixel) for the bit-domain morphology pass
    int flt;           // interior strips run the float32 pipeline (typed loads, float sums): w - 1 <= 30, see strip_loop_f
    // Epilogue of the call (small batches: a launch costs ~4 us, which is what the flag copy and the next call's
    // k_init_globals cost each): the last workgroup of the last kernel (k_corner_partial<true>) writes the per-page flags
    // straight into the caller's pinned slot and leaves globals and counters in their initial state.  ep_host == null: off.
    int flt_a;         // Wolf sweep A runs the float32 loop for this (wide) window too: no decision there, only K~ (flt_a_usable)
    float kabs;        // ... whose K~ then carries an ABSOLUTE error bound (w^2 times the rounding of the wide Q sums), in K units
    double flt_dq;     // float32 pipeline: |Q~ - Q| <= flt_dq (absolute, flt_usable's delta) for the sums queued pixels carry
    PageGlobals* ep_host;
    PageGlobals* ep_dev;
    unsigned* ep_counters;
    int ep_pages;
};
constexpr int kEpArrive = 60;  // counter word counting the workgroups of the last kernel that are done
constexpr int kCntPageMajor = 5;   // counter word: number of pages k_corner_rows has to walk (0: k_corner_partial does the work)

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// inclusive prefix sum over the 64 lanes: 4 row_shr steps inside each 16-lane row, then the two
// row broadcasts (DPP, no LDS traffic)
__device__ __forceinline__ unsigned wave_scan_incl(unsigned v)
{
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);  // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);  // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);  // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);  // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1,3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2,3
    return (unsigned)x;
}

// value held by lane + 1 (0 beyond the wavefront): DPP wave_shl:1
__device__ __forceinline__ unsigned lane_up1(unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, true);
}

__device__ __forceinline__ unsigned byte_of(uint2 v, int c)
{
    const unsigned w = c < 4 ? v.x : v.y;
    return (w >> (8 * (c & 3))) & 0xffu;
}

// ---- float32 evaluation: returns tn = Z (T~ - (p - 0.5)) and K~ = w^2 Q - S^2 (variance / f^2) ----------
// With m = f S, v = f^2 K, s = f sqrt(K) every threshold is a polynomial in S and sqrt(K) (or sqrt(Q)); the
// constants carry the powers of f and the scale Z = 2^30, so no multiply is spent on them (P2 = Z (p - 0.5)):
//   SAUVOLA  c0 = Z a f^2, c1 = Z b f            T = S (c0 sqrtK + c1)            a = k/128, b = 1-k
//   NIBLACK  c0 = Z k f,   c1 = Z f              T = c0 sqrtK + c1 S
//   NICK     c0 = Z k sqrt(f), c1 = Z f          T = c0 sqrtQ + c1 S              (m*m + s*s == q)
//   FENG     c0 = Z (1 + (1-alpha1)) f           T = c0 S + Z c3 (c3 folded into P2 per page)   (s > 0)
//   WOLF     c0 = k, c1 = Z f, pk.c1 = f k/max(s), pk.imin = Z Imin
//                                                T = c1 S + (pk.c1 sqrtK - c0)(c1 S - pk.imin)
template <int METHOD>
__device__ __forceinline__ float eval32f(const FusedParams& fp, float Sf, float Qf, float P2, const PageK& pk, float* k_out)
{
    // returns the NEGATED margin tn = Z (T~ - (p - 0.5)): white <=> tn < 0 <=> sign bit set, which pack_signs() turns
    // into the 0xFF mask byte; the settled test only uses |tn|
    if (METHOD == PRL_NICK) {
        // T only needs sqrt(Q); the "variance is not tiny" guard of the settled test rides on Q itself: v* >= q* / (1 + R)
        // always ((w-1)^2 pixels divided by w^2), so Q above (1 + R) vthr / f implies v* > vthr (fp.vthr32 is that floor for NICK).
        // Two instructions per pixel and two registers less than K~ (NICK sat at 98 VGPRs = 4 wavefronts per SIMD).
        *k_out = Qf;
        return fmaf(Sf, fp.c1, fmaf(__builtin_amdgcn_sqrtf(Qf), fp.c0, -P2));
    }
    const float K = fmaf(fp.w2f, Qf, -(Sf * Sf));
    *k_out = K;
    if (METHOD == PRL_SAUVOLA) {
        const float d = fmaf(__builtin_amdgcn_sqrtf(K), fp.c0, fp.c1);
        return fmaf(Sf, d, -P2);
    } else if (METHOD == PRL_NIBLACK) {
        return fmaf(Sf, fp.c1, fmaf(__builtin_amdgcn_sqrtf(K), fp.c0, -P2));
    } else if (METHOD == PRL_WOLFJOLION) {
        const float d = fmaf(__builtin_amdgcn_sqrtf(K), pk.c1, -fp.c0);
        const float e = fmaf(Sf, fp.c1, -pk.imin);
        return fmaf(d, e, Sf * fp.c1) - P2;
    } else if (METHOD == PRL_FENG) {
        return fmaf(Sf, fp.c0, -P2);
    } else {  // Wolf sweeps: only K~ is used
        return 0.0f;
    }
}

// mask bytes of four pixels from the sign bits of their (negated) margins: v_perm_b32 selector 9 / 11 = the sign of
// the low / high source replicated into a byte, 12 = 0x00.  Two permutes and an OR per four pixels instead of four
// v_cvt_pk_u8_f32 (all 4-cycle instructions; the OR is a 2-cycle one).
__device__ __forceinline__ unsigned pack_signs(float t0, float t1, float t2, float t3)
{
    const unsigned a = __builtin_amdgcn_perm(__float_as_uint(t1), __float_as_uint(t0), 0x0c0c0b09u);
    const unsigned b = __builtin_amdgcn_perm(__float_as_uint(t3), __float_as_uint(t2), 0x0b090c0cu);
    return a | b;
}

template <int METHOD, bool WIDE = false>
__device__ __forceinline__ float eval32(const FusedParams& fp, unsigned S, unsigned Q, float P2, const PageK& pk,
                                        float* k_out)
{
    // S arrives as 0x4B000000 + S (the horizontal sum carries that bias), whose bit pattern is the float 2^23 + S for
    // S < 2^23: one 2-cycle v_sub_f32 instead of the 4-cycle v_cvt_f32_u32 (profiles/r01/valu_issue_costs.txt)
    // (WIDE: windows wider than 181, where S can reach 2^24: plain conversion, no bias)
    const float Sf = WIDE ? (float)S : __uint_as_float(S) - 8388608.0f, Qf = (float)Q;
    return eval32f<METHOD>(fp, Sf, Qf, P2, pk, k_out);
}

// ---- float64 interval evaluation: 255 / 0 when provably decided, 2 otherwise -----------------
// T* is the threshold in exact arithmetic from the exact window sums; the literal sequence differs
// from it by at most ET (propagated from Em, Eq, the host-side bounds on the 4-tap rounding noise).
template <int METHOD>
// eq_extra: additional uncertainty of q = f Q (the float32 pipeline's Q~ is within flt_dq of the exact sum: f flt_dq; else 0)
// coeff_rel (Wolf-Jolion): the literal k / devianceMax lies within coeff_rel |coeff| of `coeff` (k_wolf_interval)
__device__ __forceinline__ unsigned refine64(const FusedParams& fp, unsigned S, unsigned Q, unsigned p, double imin,
                                             double coeff, double eq_extra = 0.0, double coeff_rel = 0.0)
{
    if (p == 0) return 0;  // 0 > T8 is false for every T8 (also for NaN -> 0)
    const ThrParams& tp = fp.tp;
    const double tiny = 8.9e-16;  // 8 ulp: float64 evaluation noise of the few operations below
    const double m = (double)S * tp.f;
    const double q = (double)Q * tp.f;
    const double v = q - m * m;
    const double Eq = fp.Eq + eq_extra;
    const double Ev = Eq + 2.0 * m * fp.Em + fp.Em * fp.Em + tiny * (q + m * m);
    if (!(v > 4.0 * Ev)) return 2;  // literal sqrt may be NaN / arbitrarily far
    const double s = sqrt(v);
    // |sqrt(v') - sqrt(v)| for |v' - v| <= Ev < v / 4 is at most Ev / sqrt(v - Ev) <= Ev / (0.866 s); the bound only has to hold,
    // so it takes the hardware's reciprocal square root as it comes (v_rsq_f64: within 5.3e-8 = 2^-24.2 of 1 / sqrt(v) over 6.7e7
    // arguments, tools/ubench/rsq64.hip, profiles/r06/v_rsq_f64_accuracy.json) with the slack in the constant - 1.16 leaves 0.36 %
    // over 1.001 / 0.866 = 1.1559 - instead of a second correctly rounded square root and a division: a third of
    // this function's instructions, which is what bounds k_fused_exact on a page whose every pixel comes here
    const double Es = 1.16 * Ev * __builtin_amdgcn_rsq(v) + tiny * s;
    double T, ET;
    if (METHOD == PRL_SAUVOLA) {
        const double d = s * tp.a + tp.b;
        const double Ed = fabs(tp.a) * Es + tiny * (fabs(tp.a) * s + fabs(tp.b));
        T = m * d;
        ET = m * Ed + fabs(d) * fp.Em + fp.Em * Ed + tiny * fabs(T);
    } else if (METHOD == PRL_NIBLACK) {
        T = s * tp.k + m;
        ET = fabs(tp.k) * Es + fp.Em + tiny * (fabs(T) + m);
    } else if (METHOD == PRL_NICK) {
        // literal: C = fl(fl(m*m) + fl(s*s)) = q_literal up to a few ulp
        const double EC = Eq + tiny * q;
        const double c = sqrt(q);
        const double Ec = 1.16 * EC * __builtin_amdgcn_rsq(q) + tiny * c;   // (q >= v > 4 Ev >= 4 Eq and tiny q is nothing: q - EC > 0.75 q as above)
        T = m + c * tp.k;
        ET = fp.Em + fabs(tp.k) * Ec + tiny * (fabs(T) + m);
    } else if (METHOD == PRL_WOLFJOLION) {
        if (!(fabs(coeff) < 1e300) || !(coeff_rel < 1.0)) return 2;  // devianceMax == 0 / no finite deviation / no usable bound: literal decides
        const double d = s * coeff + (-tp.k);
        const double Ed = fabs(coeff) * (Es + coeff_rel * (s + Es)) + tiny * (fabs(coeff) * s + fabs(tp.k));
        const double e = m - imin;
        const double Ee = fp.Em + tiny * (m + imin);
        const double gg = d * e;
        const double Eg = fabs(d) * Ee + fabs(e) * Ed + Ed * Ee + tiny * fabs(gg);
        T = m + gg;
        ET = fp.Em + Eg + tiny * (fabs(T) + m);
    } else {  // FENG with s > 0 (guaranteed by v > 4 Ev): r = r2 = c2 = 1
        const double c3 = (tp.k2 * imin + (-imin)) + 0.0;
        const double g = 1.0 + tp.c1;
        T = g * m + c3;
        ET = fabs(g) * fp.Em + tiny * (fabs(T) + fabs(g) * m + fabs(c3));
    }
    ET = 2.0 * ET + 1e-12;
    if (!(fabs(T) < 1e9)) return 2;
    const double P = (double)p - 0.5;
    if (T < P - ET) return 255;
    if (T > P + ET) return 0;
    return 2;
}

// one decided pixel (k_refine / k_fixup_final): byte mask, or one bit of the bit plane (other threads may own
// neighbouring bits of the word: atomics)
__device__ __forceinline__ void store_decision(const PageSetOut& dst, int bit_out, int page, int y, int x, unsigned r)
{
    uint8_t* row = dst.page(page) + (size_t)y * dst.step;
    if (!bit_out) {
        row[x] = (uint8_t)r;
    } else {
        unsigned* w = reinterpret_cast<unsigned*>(row) + (x >> 5);
        const unsigned m = 1u << (x & 31);
        if (r) atomicOr(w, m);
        else atomicAnd(w, ~m);
    }
}

// queue a pixel the float32 test left open: bucket = wavefront id mod kRefBuckets (see kRefBuckets in prl_internal.h)
// -> true when the bucket is full: the page is flagged (the literal pipeline redoes it) and the caller stops queueing its pixels
__device__ __forceinline__ bool ref_push(RefItem* __restrict__ rl, unsigned* __restrict__ counters, PageGlobals* __restrict__ g,
                                         unsigned wid, const RefItem& it)
{
    const unsigned b = wid & (unsigned)(kRefBuckets - 1);
    const unsigned idx = atomicAdd(&counters[64 + kRefCounterStride * b], 1u);
    if (idx < kRefBucketCap) {
        rl[(size_t)b * kRefBucketCap + idx] = it;
        return false;
    }
    atomicOr(&g[it.page].worklist_overflow, 1u);
    return true;
}

typedef const uint8_t __attribute__((address_space(1)))* gcptr;  // known-global pointers: global_load, not flat_load
typedef uint8_t __attribute__((address_space(1)))* gptr;

__device__ __forceinline__ uint2 gload8(gcptr p)
{
    uint2 v;
    __builtin_memcpy(&v, (const uint8_t*)p, 8);
    return v;
}

// ---- raw buffer access (strip_loop_f) -----------------------------------------------------------------------------------
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wundefined-internal"
__device__ f32x4 buf_load_fmt_xyzw(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.format.v4f32");
typedef int i32x2 __attribute__((ext_vector_type(2)));
__device__ void buf_store_x2(i32x2 data, i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.v2i32");
__device__ i32x2 buf_load_x2(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v2i32");
__device__ void buf_store_u8(unsigned char data, i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.i8");
#pragma clang diagnostic pop

// Raw buffer resource over the first `limit` bytes of ONE row of an output page (the range check of a raw buffer covers the
// scalar offset too, so the row address goes into the base: two scalar instructions a row).  The check works per dword: an
// 8-byte store whose second dword starts at `limit` writes its first dword and drops the second - the partial store of a
// ragged strip's last lane (4 valid bytes) without a divergent branch.
__device__ __forceinline__ i32x4 clip_rsrc(unsigned long long a, int limit)
{
    i32x4 r;
    r.x = (int)(unsigned)a;
    r.y = (int)((a >> 32) & 0xffffu);
    r.z = limit;
    r.w = (4 | (5 << 3) | (6 << 6) | (7 << 9)) | (7 << 12) | (4 << 15);   // 32-bit data, untyped access
    return r;
}


// Per-lane constants of the replicate clamp for an 8-byte row fetch: the fetch address is clamped into the row
// (colc) and the wanted bytes are picked out of the fetched ones afterwards (edge strips only).  Wanted byte c is
// image column clamp(col + c, 0, W-1) = fetched byte clamp(col + c, 0, W-1) - colc, a fixed byte permutation per
// lane: two v_perm_b32 per fetch (the first version shifted and filled through 64-bit arithmetic, ~20 instructions).
struct EdgeFix {
    int colc;        // clamped fetch column
    unsigned selx;   // v_perm_b32 selectors for wanted bytes 0..3 and 4..7 (0..3 = low dword, 4..7 = high dword)
    unsigned sely;
};

__device__ __forceinline__ EdgeFix make_edge(int col, int W)
{
    EdgeFix e;
    e.colc = clampi(col, 0, W - 8);
    e.selx = e.sely = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const unsigned idx = (unsigned)(clampi(col + c, 0, W - 1) - e.colc);  // 0..7
        if (c < 4) e.selx |= idx << (8 * c);
        else e.sely |= idx << (8 * (c - 4));
    }
    return e;
}

__device__ __forceinline__ uint2 apply_edge(uint2 v, const EdgeFix& e)
{
    return make_uint2(__builtin_amdgcn_perm(v.y, v.x, e.selx), __builtin_amdgcn_perm(v.y, v.x, e.sely));
}

// the lane that straddles the row end: its nv = 1..7 mask bytes as dword + halfword + byte
__device__ __forceinline__ void store_tail(gptr o, unsigned lo, unsigned hi, int nv)
{
    unsigned long long bytes = ((unsigned long long)hi << 32) | lo;
    if (nv & 4) {
        const unsigned d4 = (unsigned)bytes;
        __builtin_memcpy((uint8_t*)o, &d4, 4);
        o += 4;
        bytes >>= 32;
    }
    if (nv & 2) {
        const unsigned short d2 = (unsigned short)bytes;
        __builtin_memcpy((uint8_t*)o, &d2, 2);
        o += 2;
        bytes >>= 16;
    }
    if (nv & 1) *o = (uint8_t)bytes;
}

// One wavefront: a strip of SW padded columns x a segment of output rows.
//   EDGE  : the strip touches the left/right page border (replicate clamp, partial stores)
//           the leaving row and the compared pixels are never re-read from memory
// EXACT (k_fused_exact, the second chance of a page whose queue overflowed): what the float32 test leaves open is decided HERE by the
// float64 interval test on the exact sums this loop holds (refine64, what k_refine would do with a queued pixel) and the mask byte is
// set before it is stored; only what that leaves open too - true ties - goes to the fix-up list.  Nothing is queued.
struct ExactOut {
    WorkItem* wl = nullptr;      // the fix-up list (counters[1] entries), its corner accumulators and arrival counts
    CornerAcc* acc = nullptr;
    unsigned* done = nullptr;
};
template <int METHOD, int SH, bool EDGE, bool WIDE, bool EXACT = false>
__device__ __forceinline__ void strip_loop(gcptr img, gptr out, size_t istep, size_t ostep,
                                           const FusedParams& fp, int page, int xs, int ys, int ye, int lane,
                                           const PageK& pk, unsigned wid, PageGlobals* __restrict__ g,
                                           RefItem* __restrict__ rl, WorkItem* __restrict__ cand,
                                           unsigned* __restrict__ counters, bool last_ext = false, const ExactOut& xo = ExactOut{})
{
    constexpr bool SWEEP = (METHOD == kWolfMax || METHOD == kWolfCollect);
    // WIDE: w - 1 > 181, S does not fit the mantissa trick (eval32)
    const ThrParams& tp = fp.tp;
    const int W = tp.width, H = tp.height, h = tp.half, w = tp.w;
    const int col0 = xs + 1 - h + CPL * lane;  // image column of this lane's sub-column 0
    const int x0 = xs + CPL * lane;            // first output column of this lane
    // Extended last strip (EDGE, wide windows - strip_layout()): lane 63 and every column right of it is the replicated
    // border column, so the lanes whose far edge lies beyond the wavefront have outputs too.  Their far lane is clamped to
    // lane 63 - any padding lane holds the same in-lane prefixes sub x V - and the totals of the lanes that do not exist,
    // 8 V each (V = the border column's sums), are added to W: n0 / n1 missing lanes' worth for the two far positions.
    const bool ext = EDGE && last_ext;
    const bool lane_has_out = ((CPL * lane < fp.uo) || ext) && (x0 < tp.ow);
    const bool full8 = lane_has_out && (x0 + CPL <= tp.ow);
    const EdgeFix ew = EDGE ? make_edge(col0, W) : EdgeFix{col0, 0x03020100u, 0x07060504u};
    const EdgeFix ep = make_edge(x0, W);  // lanes without output fetch a clamped (ignored) location
    const int far_addr0 = min(lane + fp.lane_off, 63) * 4;  // ds_bpermute byte address of the lane holding E(j0+w-1)
    const int far_addr1 = min(lane + fp.lane_off + 1, 63) * 4;
    const unsigned n0 = ext ? 8u * (unsigned)max(lane + fp.lane_off - 63, 0) : 0u;
    const unsigned n1 = ext ? 8u * (unsigned)max(lane + fp.lane_off + 1 - 63, 0) : 0u;

    auto load_win = [&](int padded_row) -> uint2 {
        const size_t ro = (size_t)clampi(padded_row - h, 0, H - 1) * istep;  // wave-uniform
        uint2 v = gload8(img + ro + ew.colc);   // (through a buffer resource like strip_loop_f's: w=101 +2.5 %, w=51 +8.5 % - slower here)
        if (EDGE) v = apply_edge(v, ew);
        return v;
    };

    unsigned VS[CPL], VQ[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) VS[c] = VQ[c] = 0;
    // Wolf sweep A also carries cv::minMaxLoc's page minimum: bytewise running minimum of every window row this
    // wavefront fetches (even and odd bytes as packed 16-bit lanes); the rows and columns no sweep fetches are
    // covered by k_page_min_border
    typedef unsigned short us2v __attribute__((ext_vector_type(2)));
    us2v pm_e = {255, 255}, pm_o = {255, 255};
    auto track_min = [&](uint2 v) {
        if (METHOD != kWolfMax) return;
        pm_e = __builtin_elementwise_min(pm_e, __builtin_bit_cast(us2v, v.x & 0x00ff00ffu));
        pm_o = __builtin_elementwise_min(pm_o, __builtin_bit_cast(us2v, (v.x >> 8) & 0x00ff00ffu));
        pm_e = __builtin_elementwise_min(pm_e, __builtin_bit_cast(us2v, v.y & 0x00ff00ffu));
        pm_o = __builtin_elementwise_min(pm_o, __builtin_bit_cast(us2v, (v.y >> 8) & 0x00ff00ffu));
    };

    // warm-up: vertical sums over padded rows ys+1 .. ys+w-1
    // (8 rows in flight: a lone wavefront of a small batch waits for each fetch - 4 A4 pages, w=101: 0.090 -> 0.081 ms; 256
    // pages: -2..-3 %, profiles/r03/warm_unroll.txt)
#pragma unroll 8
    for (int pr = ys + 1; pr <= ys + w - 1; ++pr) {
        const uint2 v = load_win(pr);
        track_min(v);
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const unsigned b = byte_of(v, c);
            VS[c] += b;
            VQ[c] += b * b;
        }
    }

    float vmax_lane = 0.0f;  // Wolf sweep A
    bool page_flagged = false;   // (wave-uniform) a push of this wavefront found its queue bucket full
    [[maybe_unused]] unsigned x_refined = 0;   // EXACT: pixels this lane decided by the interval test (statistics, one atomic per lane at the end)
    // EXACT: the page's constants of the interval test, once per wavefront (read per pixel they were three dependent loads each)
    [[maybe_unused]] double x_imin = 0.0, x_coeff = 0.0, x_crel = 0.0;
    if constexpr (EXACT) {
        x_imin = (double)g[page].imin;
        x_coeff = g[page].coeff;
        if (METHOD == PRL_WOLFJOLION) x_crel = g[page].coeff_rel;
    }
    uint2 vnew_n = load_win(ys + w);  // entering row of the first iteration, fetched one iteration ahead
#pragma unroll 1
    for (int y = ys; y < ye; ++y) {
        uint2 pv, vold;
        const uint2 vnew = vnew_n;
        track_min(vnew);
        if (!SWEEP) pv = gload8(img + (size_t)y * istep + ep.colc);
        vnew_n = load_win(y + 1 + w);
        vold = load_win(y + 1);
        if (EDGE && !SWEEP) pv = apply_edge(pv, ep);

        // horizontal window sums: S(j0) = E(j0+w-1) - E(j0) with E the exclusive prefix of the column sums.
        // No wavefront-wide scan is needed: with RAW in-lane prefixes the difference between this lane and the lane
        // that holds the far edge is E_far[sub] - E_own[c] + W, W = totals of the lanes in between (`lane_off` of
        // them, one more for the columns whose far edge lies a lane further), gathered by DPP wave_shl:1 steps.
        unsigned ES[CPL], EQ[CPL];
        unsigned tot_s, tot_q;
        {
            unsigned accs = 0, accq = 0;
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                ES[c] = accs;
                EQ[c] = accq;
                accs += VS[c];
                accq += VQ[c];
            }
            tot_s = accs;
            tot_q = accq;
        }
        unsigned w0s = 0, w0q = 0, w1s = tot_s, w1q = tot_q;  // sums over lanes [lane, lane+j) and [lane, lane+j]
#define PRL_W_STEP()                      \
    do {                                  \
        w0s = w1s;                        \
        w0q = w1q;                        \
        w1s = tot_s + lane_up1(w1s);      \
        w1q = tot_q + lane_up1(w1q);      \
    } while (0)
        switch (fp.lane_off) {  // wave-uniform; straight-line code for the common window sizes (w <= 41)
        case 0: break;
        case 1: PRL_W_STEP(); break;
        case 2: PRL_W_STEP(); PRL_W_STEP(); break;
        case 3: PRL_W_STEP(); PRL_W_STEP(); PRL_W_STEP(); break;
        case 4: PRL_W_STEP(); PRL_W_STEP(); PRL_W_STEP(); PRL_W_STEP(); break;
        default: {  // wide windows: one wavefront scan of the lane totals, W = difference of its values at the two lanes
            const unsigned ps = wave_scan_incl(tot_s) - tot_s, pq = wave_scan_incl(tot_q) - tot_q;
            w0s = (unsigned)__builtin_amdgcn_ds_bpermute(far_addr0, (int)ps) - ps;
            w1s = (unsigned)__builtin_amdgcn_ds_bpermute(far_addr1, (int)ps) - ps;
            w0q = (unsigned)__builtin_amdgcn_ds_bpermute(far_addr0, (int)pq) - pq;
            w1q = (unsigned)__builtin_amdgcn_ds_bpermute(far_addr1, (int)pq) - pq;
            if (ext) {  // wave-uniform; both factors below 2^24, sums modulo 2^32 like every partial sum here
                const unsigned vs = (unsigned)__builtin_amdgcn_readlane((int)VS[CPL - 1], 63);
                const unsigned vq = (unsigned)__builtin_amdgcn_readlane((int)VQ[CPL - 1], 63);
                w0s += __umul24(n0, vs);
                w1s += __umul24(n1, vs);
                w0q += __umul24(n0, vq);
                w1q += __umul24(n1, vq);
            }
        }
        }
#undef PRL_W_STEP
        const unsigned sbias = WIDE ? 0u : kSBias;
        const unsigned w0sb = w0s + sbias, w1sb = w1s + sbias;  // Ssum comes out as kSBias + S (see eval32)
        // all exchanges first, then the integer arithmetic in one run (integer 2-cycle instructions issue at 4 cycles
        // next to 4-cycle ones, profiles/r01/valu_issue_costs.txt): 3.86 -> 3.82 ms
        unsigned Ssum[CPL], Qsum[CPL];
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const int sub = (c + SH) & 7;
            const int addr = (c + SH) >= 8 ? far_addr1 : far_addr0;
            Ssum[c] = (unsigned)__builtin_amdgcn_ds_bpermute(addr, (int)ES[sub]);
        }
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const int sub = (c + SH) & 7;
            const int addr = (c + SH) >= 8 ? far_addr1 : far_addr0;
            Qsum[c] = (unsigned)__builtin_amdgcn_ds_bpermute(addr, (int)EQ[sub]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const bool far1 = (c + SH) >= 8;
            Ssum[c] = (Ssum[c] - ES[c]) + (far1 ? w1sb : w0sb);
        }
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const bool far1 = (c + SH) >= 8;
            Qsum[c] = (Qsum[c] - EQ[c]) + (far1 ? w1q : w0q);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (METHOD == kWolfMax) {
            // Wolf sweep A: running maximum of the float32 variance over the wavefront's valid pixels
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                float v32;
                (void)eval32<METHOD, WIDE>(fp, Ssum[c], Qsum[c], 0.0f, pk, &v32);
                if (lane_has_out && (x0 + c < tp.ow)) vmax_lane = fmaxf(vmax_lane, v32);
            }
        } else if (METHOD == kWolfCollect) {
            // Wolf sweep B: queue the pixels whose literal deviation could be the page maximum
            float vm = 0.0f;
            float vv[CPL];
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                (void)eval32<METHOD, WIDE>(fp, Ssum[c], Qsum[c], 0.0f, pk, &vv[c]);
                if (lane_has_out && (x0 + c < tp.ow)) vm = fmaxf(vm, vv[c]);
            }
            if (__ballot(vm >= pk.c1) != 0ull) {
                // the EXACT maximum of K = w^2 Q - S^2 (u64: S^2 < 2^48, w^2 Q < 2^49) bounds the literal devianceMax to a few
                // parts in 10^9 (k_wolf_interval) - all the sweeps and k_refine need; the candidates themselves are only
                // evaluated literally when a pixel of the page reaches the literal fix-up (k_wolf_literal)
                unsigned long long kbest = 0ull;
                unsigned mine = 0u;   // bit c: pixel c of this lane is a candidate
#pragma unroll
                for (int c = 0; c < CPL; ++c) {
                    if (lane_has_out && (x0 + c < tp.ow) && vv[c] >= pk.c1) {
                        const unsigned long long s64 = (unsigned long long)(Ssum[c] - sbias);
                        const unsigned long long k64 = (unsigned long long)(unsigned)(w * w) * (unsigned long long)Qsum[c] - s64 * s64;
                        kbest = k64 > kbest ? k64 : kbest;
                        mine |= 1u << c;
                    }
                }
                // The list slots of a wavefront-row are taken with ONE atomic on the page's counter and one on the list's (a
                // flat page makes every pixel a candidate: two same-address atomics per pixel serialised 8.7 million times
                // per A4 page).  A page may queue fp.cand_page_cap candidates; past that it is marked cand_overflow - the
                // literal pipeline redoes it if it ever needs its literal maximum - and this wavefront stops queueing
                // (page_flagged), so that one blank page neither stalls the side stream nor eats the other pages' room.
                if (!page_flagged) {
                    const unsigned cnt = (unsigned)__popc(mine);
                    unsigned incl = cnt;
#pragma unroll
                    for (int d = 1; d < kWave; d <<= 1) {
                        const unsigned t = (unsigned)__shfl_up((int)incl, d, kWave);
                        if (lane >= d) incl += t;
                    }
                    const unsigned total = (unsigned)__shfl((int)incl, kWave - 1, kWave);
                    unsigned base = 0xffffffffu;
                    if (lane == 0) {
                        const unsigned before = atomicAdd(&g[page].n_cand, total);
                        if (before + total <= fp.cand_page_cap) base = atomicAdd(&counters[2], total);
                    }
                    base = (unsigned)__shfl((int)base, 0, kWave);
                    if (base == 0xffffffffu || base + total > fp.wl_cap) {
                        if (lane == 0) atomicOr(&g[page].cand_overflow, 1u);
                        page_flagged = true;
                    }
                    if (base != 0xffffffffu) {   // (slots below the list's end are always written: the counter has moved past them)
                        unsigned idx = base + incl - cnt;
#pragma unroll
                        for (int c = 0; c < CPL; ++c) {
                            if ((mine >> c) & 1u) {
                                if (idx < fp.wl_cap) {
                                    WorkItem it;
                                    it.page = page;
                                    it.y = y;
                                    it.x = x0 + c;
                                    it.pad = 0;
                                    cand[idx] = it;
                                }
                                ++idx;
                            }
                        }
                    }
                }
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) {
                    const unsigned long long o = ((unsigned long long)(unsigned)__shfl_xor((int)(kbest >> 32), d, kWave) << 32) |
                                                 (unsigned)__shfl_xor((int)(unsigned)kbest, d, kWave);
                    kbest = o > kbest ? o : kbest;
                }
                if (lane == 0 && kbest) atomicMax(&g[page].kmax_bits, kbest);
            }
        } else {
            // float32 decision; tmin/vmin track the smallest margin / variance of the lane's 8 pixels.
            // Mask bytes: the sign of the negated margin (pack_signs); unsettled pixels are overwritten by
            // k_refine/k_fixup.
            float tn[CPL];
            [[maybe_unused]] float xv[CPL];   // EXACT: the float32 variances, kept for the per-pixel test below
            float tmin = 3.0e38f, vmin = 3.0e38f;
    #pragma unroll
            for (int c = 0; c < CPL; ++c) {
                const unsigned p = byte_of(pv, c);
                const float P2 = fmaf((float)p, kZ, pk.p0);
                float v32;
                tn[c] = eval32<METHOD, WIDE>(fp, Ssum[c], Qsum[c], P2, pk, &v32);
                tmin = fminf(tmin, fabsf(tn[c]));
                vmin = fminf(vmin, v32);
                if constexpr (EXACT) xv[c] = v32;
            }
            unsigned lo = pack_signs(tn[0], tn[1], tn[2], tn[3]), hi = pack_signs(tn[4], tn[5], tn[6], tn[7]);
            if (fp.need_p0) {
                // p == 0 can never exceed T8: clear those bytes (only needed when T may be negative)
                const unsigned nzl = (((pv.x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | pv.x) & 0x80808080u;
                const unsigned nzh = (((pv.y & 0x7f7f7f7fu) + 0x7f7f7f7fu) | pv.y) & 0x80808080u;
                lo &= (nzl >> 7) * 255u;
                hi &= (nzh >> 7) * 255u;
            }

            // rare: some pixel of this lane is not settled by the float32 test -> queue it for k_refine
            const bool unsure = lane_has_out && !((tmin > pk.eps1) && (vmin > fp.vthr32));
            if constexpr (EXACT) {
                // The lane's eight pixels in straight-line code (the queueing loop below picks pixel c out of the register arrays
                // with select chains and evaluates it again: 25 instructions per pixel that a page with every pixel open pays
                // 8.7 million times); margins and variances are the ones just computed.
                if (__ballot(unsure) != 0ull && unsure) {
    #pragma unroll
                    for (int c = 0; c < CPL; ++c) {
                        const unsigned p = byte_of(pv, c);
                        const bool open = (x0 + c < tp.ow) && p != 0 && !((fabsf(tn[c]) > pk.eps1) && (xv[c] > fp.vthr32));
                        if (open) {
                            const unsigned r = refine64<METHOD>(fp, Ssum[c] - sbias, Qsum[c], p, x_imin, x_coeff, 0.0, x_crel);
                            if (r != 2) {
                                if (c < 4) lo = (lo & ~(0xffu << (8 * c))) | (r << (8 * c));
                                else hi = (hi & ~(0xffu << (8 * (c - 4)))) | (r << (8 * (c - 4)));
                                ++x_refined;
                            } else {   // a true tie: absolute corners and the literal sequence decide (fix-up list, as k_refine fills it)
                                atomicAdd(&g[page].n_exact, 1u);
                                if (METHOD == PRL_WOLFJOLION) atomicOr(&g[page].need_literal, 1u);
                                const unsigned idx = atomicAdd(&counters[1], 1u);
                                if (idx < fp.wl_cap) {
                                    WorkItem wi;
                                    wi.page = page;
                                    wi.y = y;
                                    wi.x = x0 + c;
                                    wi.pad = 0;
                                    xo.wl[idx] = wi;
#pragma unroll
                                    for (int k = 0; k < 8; ++k) xo.acc[idx].a[k] = 0ull;
                                    xo.done[idx] = 0u;
                                } else {
                                    atomicOr(&g[page].worklist_overflow, 2u);
                                }
                            }
                        }
                    }
                }
            } else if (__ballot(unsure) != 0ull && !page_flagged) {   // (a flagged page is redone literally: no point in queueing more of it)
                bool full = false;
                if (unsure) {
    #pragma unroll 1
                    for (int c = 0; c < CPL; ++c) {
                        if (x0 + c >= tp.ow) break;
                        const unsigned p = (c < 4 ? pv.x >> (8 * c) : pv.y >> (8 * (c - 4))) & 0xffu;
                        if (p == 0) continue;  // 0 > T8 is false whatever T is
                        const unsigned S = c == 0 ? Ssum[0] : c == 1 ? Ssum[1] : c == 2 ? Ssum[2] : c == 3 ? Ssum[3]
                                         : c == 4 ? Ssum[4] : c == 5 ? Ssum[5] : c == 6 ? Ssum[6] : Ssum[7];
                        const unsigned Q = c == 0 ? Qsum[0] : c == 1 ? Qsum[1] : c == 2 ? Qsum[2] : c == 3 ? Qsum[3]
                                         : c == 4 ? Qsum[4] : c == 5 ? Qsum[5] : c == 6 ? Qsum[6] : Qsum[7];
                        float v32;
                        const float t = eval32<METHOD, WIDE>(fp, S, Q, fmaf((float)p, kZ, pk.p0), pk, &v32);
                        if ((fabsf(t) > pk.eps1) && (v32 > fp.vthr32)) continue;
                        RefItem it;
                        it.page = page;
                        it.y = y;
                        it.x = x0 + c;
                        it.S = S - sbias;
                        it.Q = Q;
                        it.p = p;
                        full |= ref_push(rl, counters, g, wid, it);
                    }
                }
                page_flagged = __ballot(full) != 0ull;
            }

            if (fp.bit_out) {
                // bit plane: this lane's 8 pixels are one whole byte (x0 is a multiple of 8); 0x00 / 0xFF bytes -> bits
                if (lane_has_out) {
                    unsigned b = (((lo & 0x01010101u) * 0x01020408u) >> 24) | ((((hi & 0x01010101u) * 0x01020408u) >> 20) & 0xf0u);
                    if (EDGE && !full8) b &= (1u << (tp.ow - x0)) - 1u;  // pixels past the row end stay 0
                    out[(size_t)y * ostep + (x0 >> 3)] = (uint8_t)b;
                }
            } else if (full8) {  // store 8 mask bytes
                if (fp.nt_store) {
                    typedef unsigned u2v __attribute__((ext_vector_type(2)));
                    u2v o = {lo, hi};
                    __builtin_nontemporal_store(o, reinterpret_cast<u2v*>((uint8_t*)(out + (size_t)y * ostep + x0)));
                } else {
                    uint2 o = make_uint2(lo, hi);
                    __builtin_memcpy((uint8_t*)(out + (size_t)y * ostep + x0), &o, 8);
                }
            } else if (EDGE && lane_has_out) {
                store_tail(out + (size_t)y * ostep + x0, lo, hi, tp.ow - x0);
            }
        }

        // slide the window one row down
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const int nbv = (int)byte_of(vnew, c), obv = (int)byte_of(vold, c);
            const int d = nbv - obv, sm = nbv + obv;
            VS[c] += (unsigned)d;
            VQ[c] += (unsigned)(d * sm);
        }
    }
    if constexpr (EXACT) {
        if (x_refined) atomicAdd(&g[page].n_refined, x_refined);
    }
    if (METHOD == kWolfMax) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) vmax_lane = fmaxf(vmax_lane, __shfl_xor(vmax_lane, d, kWave));
        unsigned pm = min(min((unsigned)pm_e.x, (unsigned)pm_e.y), min((unsigned)pm_o.x, (unsigned)pm_o.y));
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) pm = min(pm, (unsigned)__shfl_xor((int)pm, d, kWave));
        if (lane == 0) {
            fp.segmax[wid] = vmax_lane;
            atomicMax(&g[page].v32max_bits, __float_as_uint(vmax_lane));  // v~ >= 0: bit order == value order
            atomicMin(&g[page].imin, (int)pm);
        }
    }
}


// ---- float32 strip loop: interior strips of the threshold sweeps, w - 1 <= 30 -------------------------------------
// The vector ALU is what bounds k_fused, and a third of its instructions only unpack bytes (SDWA)
// or convert integers to float.  gfx950's typed buffer loads do both in the texture-address unit:
// buffer_load_format_xyzw with an 8_8_8_8 USCALED resource returns four pixels as four floats (any byte alignment,
// 7.7 TB/s of pixels from L2 - tools/ubench/typed_load.hip).  With float pixels the column sums, their in-lane
// prefixes and the exchange run on 2-cycle float instructions (v_fma_f32 instead of v_mad_i32_i24, no conversions).
// Exactness: pixels, column sums (<= 30 * 65025), in-lane prefixes and lane totals (<= 8 * 30 * 65025 < 2^24) and
// every S quantity are exact integers in float32.  Only the Q sums of several lanes (the W chain) and the final
// Q = (E_far - E_own) + W can pass 2^24 and round: at most `flt_delta` units in all, and only when Q itself is at
// least 2^24 - 8 (w-1) 65025 - which fused_bounds() turns into the relative error `cq u` of Q~ that widens eps1.
// Queued pixels carry no sums: k_refine rebuilds S and Q exactly from the page when fp.flt is set.
// gfx9 buffer resource over one page: raw (stride 0, byte offsets), no range limit, 8_8_8_8 USCALED -> x, y, z, w
__device__ __forceinline__ i32x4 page_rsrc(gcptr page)
{
    const unsigned long long a = (unsigned long long)(const uint8_t*)page;
    i32x4 r;
    r.x = (int)(unsigned)a;
    r.y = (int)((a >> 32) & 0xffffu);
    r.z = (int)0xffffffffu;
    r.w = (4 | (5 << 3) | (6 << 6) | (7 << 9)) | (2 << 12) | (10 << 15);
    return r;
}

struct F8 {
    float v[CPL];
};

// 8 consecutive pixels of a row as floats: lane byte offset in a VGPR, row byte offset wave-uniform (SGPR)
__device__ __forceinline__ F8 tload8(const i32x4& rsrc, int col, int row_off)
{
    const f32x4 a = buf_load_fmt_xyzw(rsrc, col, row_off, 0), b = buf_load_fmt_xyzw(rsrc, col + 4, row_off, 0);
    F8 r;
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
    r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    return r;
}

__device__ __forceinline__ float lane_up1f(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true));
}

__device__ __forceinline__ float bpermf(int addr, float v)
{
    return __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(v)));
}

// LO: fp.lane_off as a compile-time constant; FAST: byte mask through non-temporal stores and no p == 0 fix-up (the usual
// call) - both only remove wave-uniform branches from the row loop (~8 taken branches per row otherwise).
// EDGE: the strip touches the left / right page border.  Typed loads cannot replicate a border column, so the window rows
// come as packed bytes from a clamped address, are put in place by strip_loop's two v_perm_b32 and converted when the
// slide uses them (16 conversions a row more than an interior strip, still a quarter fewer vector instructions than the
// integer loop: 2 of the 9 strips of a 4096-column page, 2 of the 6 of an A4 page); partial stores at the row end.
// QINT (k_fused_q, windows of 33 .. 129 columns - VERDICT r5 "next" 7): pixels, the column sums VS / VQ (<= 128 x 65025 < 2^23) and
// every S quantity (a wavefront scan value is at most 64 x 8 x 128 x 255 < 2^24) stay exact in float32; VQ lives as 2^23 + VQ, whose
// bits are the integer 0x4B000000 + VQ (no conversion), and the HORIZONTAL Q sums - in-lane prefixes, the W chain, the window sum,
// which pass 2^24 - run on integers like strip_loop's, the known multiple of 0x4B000000 taken off once per row.  S and (float)Q
// reach eval32f with exactly the values the integer loop hands eval32: same decisions, same margins, exact sums in the queue.
// What is saved is the byte unpacking of the two window rows (46 SDWA instructions a row): 713 issue cycles a row against 826-875.
template <int METHOD, int SH, int LO, bool FAST, bool EDGE, bool QINT = false>
__device__ __forceinline__ void strip_loop_f(gcptr img, gptr out, size_t istep, size_t ostep, const FusedParams& fp,
                                             int page, int xs, int ys, int ye, int lane, const PageK& pk, unsigned wid,
                                             PageGlobals* __restrict__ g, RefItem* __restrict__ rl,
                                             unsigned* __restrict__ counters, bool last_ext = false)
{
    static_assert(!QINT || LO == 4, "QINT: the wavefront-scan form of the W chain");
    // QINT && EDGE && last_ext: the extended last strip (strip_layout, strip_loop's `ext`): lane 63 and everything right of it is the
    // replicated border column; the far lane is clamped to 63 and the totals of the lanes that do not exist - 8 V each - are added to W
    const bool ext = QINT && EDGE && last_ext;
    constexpr bool SWEEP_A = METHOD == kWolfMax;  // Wolf-Jolion's variance-maximum sweep: sums and K~ only, no decision
    const ThrParams& tp = fp.tp;
    const int H = tp.height, h = tp.half, w = tp.w;
    const int col0 = xs + 1 - h + CPL * lane;  // image column of this lane's sub-column 0 (interior: no clamp)
    const int x0 = xs + CPL * lane;            // first output column of this lane
    // interior strips with uo a multiple of 8 (FAST): every lane below uo has its 8 outputs.  Otherwise the last lane with
    // outputs may hold fewer: the row ends (EDGE), or uo is not a multiple of 8 (strip_layout: a ragged uo that saves a strip)
    const int xlim = ext ? tp.ow : min(xs + fp.uo, tp.ow);
    const bool lane_has_out = FAST ? CPL * lane < fp.uo : x0 < xlim;
    const bool full8 = FAST ? lane_has_out : x0 + CPL <= xlim;
    // LO 0..3: the number of whole lanes between a window's two edges, W gathered by that many DPP steps.  LO == 4 (sweep A
    // with wide windows): fp.lane_off lanes, W by doubling (see below)
    constexpr bool SCAN = LO == 4;
    const int loff = SCAN ? fp.lane_off : LO;
    const int far_addr0 = QINT ? min(lane + loff, 63) * 4 : (lane + loff) * 4, far_addr1 = QINT ? min(lane + loff + 1, 63) * 4 : far_addr0 + 4;
    [[maybe_unused]] const unsigned xn0 = ext ? 8u * (unsigned)max(lane + loff - 63, 0) : 0u, xn1 = ext ? 8u * (unsigned)max(lane + loff + 1 - 63, 0) : 0u;
    const i32x4 rsrc = page_rsrc(img);
    const int step = (int)istep;
    const EdgeFix ew = EDGE ? make_edge(col0, tp.width) : EdgeFix{0, 0u, 0u};
    const EdgeFix ep = EDGE ? make_edge(x0, tp.width) : EdgeFix{0, 0u, 0u};  // lanes without output fetch a clamped (ignored) location
    // ragged strip whose last lane keeps 4 bytes: range-clipped buffer stores instead of a divergent partial store (wave-uniform)
    const bool clip4 = !QINT && !FAST && !EDGE && !fp.bit_out && fp.nt_store && (fp.uo & 7) == 4;   // (QINT: uo is a multiple of 8)
    unsigned long long orow = (unsigned long long)(uint8_t*)out + (unsigned long long)ys * ostep;   // row y of the output page (clip4)
    auto to_f8 = [](uint2 b) -> F8 {
        F8 r;
#pragma unroll
        for (int c = 0; c < CPL; ++c) r.v[c] = (float)byte_of(b, c);
        return r;
    };
    const i32x4 prsrc = clip_rsrc((unsigned long long)(const uint8_t*)img, -1);   // the page as untyped bytes, no range limit
    auto bload8 = [&](int col, int off) -> uint2 {   // 8 bytes at column `col` (per lane) of the row at byte offset `off` (scalar)
        const i32x2 v = buf_load_x2(prsrc, col, off, 0);
        return make_uint2((unsigned)v.x, (unsigned)v.y);
    };
    auto edge_row = [&](int off) -> uint2 { return apply_edge(bload8(ew.colc, off), ew); };  // bytes in place (EDGE)

    auto load_win = [&](int padded_row) -> F8 {
        const int off = clampi(padded_row - h, 0, H - 1) * step;
        if constexpr (EDGE) return to_f8(edge_row(off));
        else return tload8(rsrc, col0, off);
    };

    float VS[CPL], VQ[CPL];
    // QINT: the column sums of squares live as 2^23 + VQ (exact below 2^24: w - 1 <= 128), whose bit pattern is the integer
    // 0x4B000000 + VQ - no conversion instruction; every horizontal Q sum then carries a known multiple of 0x4B000000, and a window
    // sum exactly (w - 1) of them (mod 2^32), taken off once per row (qbias)
    constexpr float kVQ0 = QINT ? 8388608.0f : 0.0f;
    [[maybe_unused]] const unsigned qbias = (unsigned)(w - 1) * 0x4B000000u;
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        VS[c] = 0.0f;
        VQ[c] = kVQ0;
    }
    float pmin = 255.0f, vmax_lane = 0.0f;  // sweep A: running page minimum (see strip_loop) and variance maximum
    auto track_min = [&](const F8& v) {
        if (!SWEEP_A) return;
        pmin = fminf(fminf(pmin, fminf(v.v[0], v.v[1])), fminf(fminf(v.v[2], v.v[3]), fminf(fminf(v.v[4], v.v[5]), fminf(v.v[6], v.v[7]))));
    };
#pragma unroll 4
    for (int pr = ys + 1; pr <= ys + w - 1; ++pr) {
        const F8 v = load_win(pr);
        track_min(v);
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            VS[c] += v.v[c];
            VQ[c] = fmaf(v.v[c], v.v[c], VQ[c]);
        }
    }

    // Loads are issued where their destination registers have just died, one iteration ahead of their use: the
    // compared pixels of the next row right after this row's decision, the next entering and leaving rows right
    // after the slide.  No second set of registers, a whole iteration of latency hiding.
    // Row byte offsets advance by one row step per iteration; the replicate clamp is a min / max on the offset itself
    // (the entering row can only run past the last page row, the leaving row only start above the first): 5 scalar
    // instructions per iteration instead of 13 - scalar instructions cost issue slots like vector ones here.
    const int off_last = (H - 1) * step;
    int off_new = min((ys + w - h) * step, off_last);   // padded row ys + w
    int off_old = max((ys + 1 - h) * step, 0);          // padded row ys + 1
    F8 vnew, vold;           // interior: the two window rows as floats (typed loads)
    uint2 bnew = make_uint2(0u, 0u), bold = bnew;  // EDGE: as packed bytes, converted where the slide uses them
    if constexpr (EDGE) {
        bnew = edge_row(off_new);
        bold = edge_row(off_old);
    } else {
        vnew = tload8(rsrc, col0, off_new);
        vold = tload8(rsrc, col0, off_old);
    }
    int off_old_raw = (ys + 1 - h) * step;
    uint2 pvb = make_uint2(0u, 0u);
    // the compared pixels and the mask bytes go through raw buffer resources as well: the per-lane part of the address is
    // the loop-invariant column, the row travels in a scalar register - no 64-bit vector address arithmetic per row
    // (headline -0.7 %, 256 A4 pages NICK w=21 -4.9 %, profiles/r03/buffer_path_ab.txt)
    const int pv_col = EDGE ? ep.colc : x0;
    bool page_flagged = false;   // (wave-uniform) a push of this wavefront found its queue bucket full
    int pv_off = ys * step;
    if (!SWEEP_A) pvb = bload8(pv_col, pv_off);
#pragma unroll 1
    for (int y = ys; y < ye; ++y) {
        if constexpr (EDGE) {
            if (SWEEP_A) track_min(to_f8(bnew));
        } else {
            track_min(vnew);
        }
        float Ssum[CPL], Qsum[CPL];
        float ES[CPL], EQ[CPL], tot_s, tot_q;
        {
            float accs = VS[0], accq = VQ[0];  // (not 0 + VS[0]: the compiler keeps a float add of +0)
            ES[0] = EQ[0] = 0.0f;
#pragma unroll
            for (int c = 1; c < CPL; ++c) {
                ES[c] = accs;
                EQ[c] = accq;
                accs += VS[c];
                accq += VQ[c];
            }
            tot_s = accs;
            tot_q = accq;
        }
        [[maybe_unused]] unsigned EQi[CPL], Qi[CPL], tot_qi = 0u, w0qi = 0u, w1qi = 0u;
        if constexpr (QINT) {
            unsigned acc = 0u;
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                EQi[c] = acc;
                acc += __float_as_uint(VQ[c]);
            }
            tot_qi = acc;
        }
        float w0s = 0.0f, w0q = 0.0f, w1s = tot_s, w1q = tot_q;
#define PRL_W_STEP()                      \
    do {                                  \
        w0s = w1s;                        \
        w0q = w1q;                        \
        w1s = tot_s + lane_up1f(w1s);     \
        w1q = tot_q + lane_up1f(w1q);     \
    } while (0)
        if (!SCAN) {
            if (LO >= 1) PRL_W_STEP();  // w - 1 <= 30: at most 3 steps
            if (LO >= 2) PRL_W_STEP();
            if (LO >= 3) PRL_W_STEP();
        } else if constexpr (QINT) {
            // strip_loop's form: one wavefront scan of the lane totals (DPP), W = the difference of its values at the two lanes -
            // 4 ds_bpermute and no wave-uniform branches, where the doubling form below takes 10 and five branches on loff (this
            // loop has a decision behind it: its LDS pipe is not idle).  S: a scan value is at most 64 lanes x 8 columns x 128 rows
            // x 255 < 2^24 - exact.
            auto scan_f = [](float v) -> float {
                int x = __float_as_int(v);
                x = __float_as_int(__int_as_float(x) + __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false)));
                x = __float_as_int(__int_as_float(x) + __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false)));
                x = __float_as_int(__int_as_float(x) + __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false)));
                x = __float_as_int(__int_as_float(x) + __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false)));
                x = __float_as_int(__int_as_float(x) + __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false)));
                x = __float_as_int(__int_as_float(x) + __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false)));
                return __int_as_float(x);
            };
            const float ps = scan_f(tot_s) - tot_s;
            const unsigned pq = wave_scan_incl(tot_qi) - tot_qi, pqb = pq + qbias;
            w0s = bpermf(far_addr0, ps) - ps;
            w1s = bpermf(far_addr1, ps) - ps;
            w0qi = (unsigned)__builtin_amdgcn_ds_bpermute(far_addr0, (int)pq) - pqb;
            w1qi = (unsigned)__builtin_amdgcn_ds_bpermute(far_addr1, (int)pq) - pqb;
            if (ext) {   // wave-uniform.  n V: below 2^24 in float; the biased VQ brings its own share of qbias (8 per missing lane)
                const float vs = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(VS[CPL - 1]), 63));
                const unsigned vq = (unsigned)__builtin_amdgcn_readlane(__float_as_int(VQ[CPL - 1]), 63);
                w0s = fmaf((float)xn0, vs, w0s);
                w1s = fmaf((float)xn1, vs, w1s);
                w0qi += xn0 * vq;
                w1qi += xn1 * vq;
            }
        } else {
            // W over `loff` (5..16) lanes starting at this one: sums of 2, 4, 8 (16) consecutive lanes by doubling - the
            // operand of lane + 2^k comes through ds_bpermute (the LDS pipe has room, the vector ALU has not) - then the
            // binary digits of loff pick the pieces.  Every S quantity stays exact (<= 17 lanes x 8 columns x 128 rows x 255
            // < 2^24); the Q roundings are bounded by flt_a_usable() and only widen sweep B's candidate band.
            const int l4 = lane * 4;
            const float s2 = tot_s + lane_up1f(tot_s), q2 = tot_q + lane_up1f(tot_q);
            const float s4 = s2 + bpermf(l4 + 8, s2), q4 = q2 + bpermf(l4 + 8, q2);
            const float s8 = s4 + bpermf(l4 + 16, s4), q8 = q4 + bpermf(l4 + 16, q4);
            int off = 0;
            w0s = w0q = 0.0f;
            if (loff & 16) {
                w0s = s8 + bpermf(l4 + 32, s8);
                w0q = q8 + bpermf(l4 + 32, q8);
                off = 16;
            }
            if (loff & 8) {
                w0s += off ? bpermf(l4 + 4 * off, s8) : s8;
                w0q += off ? bpermf(l4 + 4 * off, q8) : q8;
                off += 8;
            }
            if (loff & 4) {
                w0s += off ? bpermf(l4 + 4 * off, s4) : s4;
                w0q += off ? bpermf(l4 + 4 * off, q4) : q4;
                off += 4;
            }
            if (loff & 2) {
                w0s += off ? bpermf(l4 + 4 * off, s2) : s2;
                w0q += off ? bpermf(l4 + 4 * off, q2) : q2;
                off += 2;
            }
            if (loff & 1) {
                w0s += off ? bpermf(l4 + 4 * off, tot_s) : tot_s;
                w0q += off ? bpermf(l4 + 4 * off, tot_q) : tot_q;
                off += 1;
            }
            w1s = w0s + bpermf(l4 + 4 * off, tot_s);
            w1q = w0q + bpermf(l4 + 4 * off, tot_q);
        }
#undef PRL_W_STEP
#pragma unroll
        for (int c = 0; c < CPL; ++c) Ssum[c] = bpermf((c + SH) >= 8 ? far_addr1 : far_addr0, ES[(c + SH) & 7]);
        if constexpr (QINT) {
#pragma unroll
            for (int c = 0; c < CPL; ++c) Qi[c] = (unsigned)__builtin_amdgcn_ds_bpermute((c + SH) >= 8 ? far_addr1 : far_addr0, (int)EQi[(c + SH) & 7]);
#pragma unroll
            for (int c = 0; c < CPL; ++c) Ssum[c] = (Ssum[c] - ES[c]) + ((c + SH) >= 8 ? w1s : w0s);
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                Qi[c] = (Qi[c] - EQi[c]) + ((c + SH) >= 8 ? w1qi : w0qi);
                Qsum[c] = (float)Qi[c];   // (the integer loop's conversion: the same rounding above 2^24)
            }
        } else {
#pragma unroll
        for (int c = 0; c < CPL; ++c) Qsum[c] = bpermf((c + SH) >= 8 ? far_addr1 : far_addr0, EQ[(c + SH) & 7]);
#pragma unroll
        for (int c = 0; c < CPL; ++c) Ssum[c] = (Ssum[c] - ES[c]) + ((c + SH) >= 8 ? w1s : w0s);
#pragma unroll
        for (int c = 0; c < CPL; ++c) Qsum[c] = (Qsum[c] - EQ[c]) + ((c + SH) >= 8 ? w1q : w0q);
        }

        if constexpr (SWEEP_A) {
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                const float K = fmaf(fp.w2f, Qsum[c], -(Ssum[c] * Ssum[c]));
                if (lane_has_out && (!EDGE || x0 + c < tp.ow)) vmax_lane = fmaxf(vmax_lane, K);  // (Wolf-Jolion: uo is a multiple of 8)
            }
        } else {
        // the compared pixels come as packed bytes (one 8-byte load, 8 v_cvt_f32_ubyte): the kernel leans on the
        // vector-memory pipe, and one load instruction less is worth more than eight conversions (3.48 -> 3.39 ms;
        // fetching the leaving or both window rows this way too: 3.80 ms)
        if constexpr (EDGE) pvb = apply_edge(pvb, ep);
        F8 pv;
#pragma unroll
        for (int c = 0; c < CPL; ++c) pv.v[c] = (float)byte_of(pvb, c);
        float tn[CPL];
        float tmin = 3.0e38f, vmin = 3.0e38f;
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const float P2 = fmaf(pv.v[c], kZ, pk.p0);
            float v32;
            tn[c] = eval32f<METHOD>(fp, Ssum[c], Qsum[c], P2, pk, &v32);
            tmin = fminf(tmin, fabsf(tn[c]));
            vmin = fminf(vmin, v32);
        }
        unsigned lo = pack_signs(tn[0], tn[1], tn[2], tn[3]), hi = pack_signs(tn[4], tn[5], tn[6], tn[7]);

        if (!FAST && fp.need_p0) {
            // p == 0 can never exceed T8: clear those bytes (only needed when T may be negative); on the packed bytes
            const unsigned nzl = (((pvb.x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | pvb.x) & 0x80808080u;
            const unsigned nzh = (((pvb.y & 0x7f7f7f7fu) + 0x7f7f7f7fu) | pvb.y) & 0x80808080u;
            lo &= (nzl >> 7) * 255u;
            hi &= (nzh >> 7) * 255u;
        }
        const F8 pv_cur = pv;

        // rare: some pixel of this lane is not settled by the float32 test -> queue it (k_refine rebuilds its sums)
        const bool unsure = lane_has_out && !((tmin > pk.eps1) && (vmin > fp.vthr32));
        if (__ballot(unsure) != 0ull && !page_flagged) {   // (a flagged page is redone literally: no point in queueing more of it)
            bool full = false;
            if (unsure) {
                unsigned qm = 0u;   // this lane's pixels to queue, one bit each
#pragma unroll
                for (int c = 0; c < CPL; ++c) {
                    if (!FAST && x0 + c >= xlim) continue;
                    if (pv_cur.v[c] == 0.0f) continue;  // 0 > T8 is false whatever T is
                    float v32;
                    const float t = eval32f<METHOD>(fp, Ssum[c], Qsum[c], fmaf(pv_cur.v[c], kZ, pk.p0), pk, &v32);
                    if (!((fabsf(t) > pk.eps1) && (v32 > fp.vthr32))) qm |= 1u << c;
                }
                // the sums travel with the pixel: S is exact, Q within fp.flt_dq of the exact sum - k_refine's interval test
                // takes that uncertainty first and only rebuilds the sums of what it leaves open.  (One push site, values
                // picked by select chains: eight unrolled pushes cost the whole kernel 5-7 registers.)
#pragma unroll 1
                while (qm) {
                    const int c = __builtin_ctz(qm);
                    qm &= qm - 1u;
                    const float Sv = c == 0 ? Ssum[0] : c == 1 ? Ssum[1] : c == 2 ? Ssum[2] : c == 3 ? Ssum[3]
                                   : c == 4 ? Ssum[4] : c == 5 ? Ssum[5] : c == 6 ? Ssum[6] : Ssum[7];
                    const float Qv = c == 0 ? Qsum[0] : c == 1 ? Qsum[1] : c == 2 ? Qsum[2] : c == 3 ? Qsum[3]
                                   : c == 4 ? Qsum[4] : c == 5 ? Qsum[5] : c == 6 ? Qsum[6] : Qsum[7];
                    RefItem it;
                    it.page = page;
                    it.y = y;
                    it.x = x0 + c;
                    it.S = (unsigned)Sv;
                    it.Q = (unsigned)Qv;
                    it.p = byte_of(pvb, c) | kRefApprox;
                    if constexpr (QINT) {   // exact sums, as the integer loop queues them
                        it.Q = c == 0 ? Qi[0] : c == 1 ? Qi[1] : c == 2 ? Qi[2] : c == 3 ? Qi[3] : c == 4 ? Qi[4] : c == 5 ? Qi[5] : c == 6 ? Qi[6] : Qi[7];
                        it.p = byte_of(pvb, c);
                    }
                    full |= ref_push(rl, counters, g, wid, it);
                }
            }
            page_flagged = __ballot(full) != 0ull;
        }

        if (lane_has_out) {
            if (!FAST && fp.bit_out) {
                unsigned b = (((lo & 0x01010101u) * 0x01020408u) >> 24) | ((((hi & 0x01010101u) * 0x01020408u) >> 20) & 0xf0u);
                if (EDGE && !full8) b &= (1u << (tp.ow - x0)) - 1u;  // pixels past the row end stay 0
                if constexpr (QINT) buf_store_u8((unsigned char)b, clip_rsrc(orow, -1), x0 >> 3, 0, 0);   // (no 64-bit address per row)
                else out[(size_t)y * ostep + (x0 >> 3)] = (uint8_t)b;
            } else if (!FAST && clip4) {
                i32x2 o = {(int)lo, (int)hi};
                buf_store_x2(o, clip_rsrc(orow, xlim), x0, 0, 2);   // aux 2: non-temporal
            } else if (!FAST && !full8) {
                // (interior strip: only a ragged uo gets here and the byte count is the same for every strip and row - scalar
                // branches inside store_tail; a per-lane count costs ~15 instructions a row more)
                store_tail(out + (size_t)y * ostep + x0, lo, hi, EDGE ? xlim - x0 : fp.uo & 7);
            } else if (FAST || fp.nt_store) {
                i32x2 o = {(int)lo, (int)hi};
                buf_store_x2(o, clip_rsrc(orow, -1), x0, 0, 2);   // (no range limit)
            } else {
                uint2 o = make_uint2(lo, hi);
                __builtin_memcpy((uint8_t*)(out + (size_t)y * ostep + x0), &o, 8);
            }
        }

        orow += ostep;
        pv_off += step;  // row y + 1 <= H - 1 exists for every output row
        pvb = bload8(pv_col, pv_off);
        }  // !SWEEP_A

        // slide the window one row down: new^2 - old^2 = (new - old)(new + old), one exact fma
        if constexpr (EDGE) {
            vnew = to_f8(bnew);
            vold = to_f8(bold);
        }
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const float d = vnew.v[c] - vold.v[c], sm = vnew.v[c] + vold.v[c];
            VS[c] += d;
            VQ[c] = fmaf(d, sm, VQ[c]);
        }
        off_new = min(off_new + step, off_last);
        off_old_raw += step;
        off_old = max(off_old_raw, 0);
        if constexpr (EDGE) {
            bnew = edge_row(off_new);
            bold = edge_row(off_old);
        } else {
            vnew = tload8(rsrc, col0, off_new);
            vold = tload8(rsrc, col0, off_old);  // (a non-temporal hint on this last use of the row measured 2 % slower)
        }
    }
    if (SWEEP_A) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            vmax_lane = fmaxf(vmax_lane, __shfl_xor(vmax_lane, d, kWave));
            pmin = fminf(pmin, __shfl_xor(pmin, d, kWave));
        }
        if (lane == 0) {
            fp.segmax[wid] = vmax_lane;
            atomicMax(&g[page].v32max_bits, __float_as_uint(vmax_lane));  // v~ >= 0: bit order == value order
            atomicMin(&g[page].imin, (int)pmin);
        }
    }
}

template <int METHOD, int SH, bool WIDE>
__global__ void __launch_bounds__(256) k_fused(PageSet src, PageSetOut dst, FusedParams fp,
                                              PageGlobals* __restrict__ g, RefItem* __restrict__ rl,
                                              WorkItem* __restrict__ cand, unsigned* __restrict__ counters)
{
    const ThrParams& tp = fp.tp;
    const int lane = threadIdx.x & (kWave - 1);
    const unsigned wpb = blockDim.x >> 6;  // wavefronts per block
    // XCD-aware block order: hardware deals blocks round-robin over the 8 XCDs, so blocks b and b+8
    // share an L2.  Give each XCD a contiguous range of logical wavefronts (= neighbouring strips and
    // segments of the same pages) so halo re-reads hit that XCD's L2.  Speed only, never correctness.
    // XCD x = blockIdx & 7 takes the x-th eighth of every tier, tier 0 first (its time index is blockIdx >> 3).
    const unsigned xcd = blockIdx.x & 7u;
    const unsigned wv = threadIdx.x >> 6;
    unsigned u = __builtin_amdgcn_readfirstlane((blockIdx.x >> 3) * wpb + wv);  // this wavefront's slot in its XCD's share
    unsigned wid = 0;
    int trows = 0, tsegs = 1, trow0 = 0;
    bool found = false;
    for (int k = 0; k < fp.n_tiers; ++k) {
        const unsigned n = fp.tier[k].waves;
        const unsigned lo = (unsigned)(((unsigned long long)n * xcd) >> 3), hi = (unsigned)(((unsigned long long)n * (xcd + 1u)) >> 3);
        if (u < hi - lo) {
            wid = lo + u;
            trows = fp.tier[k].rows; tsegs = fp.tier[k].segs; trow0 = fp.tier[k].row0;
            found = true;
            u = fp.tier[k].first;  // (reused below: the canonical id offset)
            break;
        }
        u -= hi - lo;
    }
    if (!found) return;
    const int per_page = fp.n_strips * tsegs;
    const int page = (int)(wid / (unsigned)per_page);
    const int rem = (int)(wid - (unsigned)page * (unsigned)per_page);
    const int seg = rem / fp.n_strips;
    const int strip = rem - seg * fp.n_strips;
    wid += u;  // canonical wavefront id over all tiers
    // A page whose queue has overflowed is redone whole by k_fused_exact (resolve_front): the strips of it that have not started
    // yet have nothing to add (adversarial pages flag themselves within the first 3 % of their strips).  Threshold sweep only -
    // the Wolf sweeps' maxima are needed by the redo.  A stale read only costs the time it would have saved.
    if (METHOD < kWolfMax && (__hip_atomic_load(&g[page].worklist_overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1u)) return;

    gcptr img = (gcptr)src.page(page);
    gptr out = (gptr)dst.page(page);

    const int xs = strip * fp.uo;          // first output column of the strip
    const int ys = trow0 + seg * trows;    // first output row of the segment
    const int ye = min(ys + trows, tp.oh);

    PageK pk;
    pk.c1 = fp.c1;
    pk.imin = 0.0f;
    pk.p0 = -0.5f * kZ;
    pk.eps1 = fp.eps1;
    if (METHOD == PRL_FENG) {
        const double imin = (double)g[page].imin;
        const double c3 = (tp.k2 * imin + (-imin)) + 0.0;  // binarizeFeng.cpp:137 with r2 = c2 = 1
        pk.p0 = (float)((-0.5 - c3) * (double)kZ);
    } else if (METHOD == PRL_WOLFJOLION) {
        pk.imin = (float)g[page].imin * kZ;
        // The sweep takes k / devianceMax from sweep A's float32 variance maximum (the exact maximum and the literal one are
        // being worked out on the side stream meanwhile) and widens its margin by what that estimate can be off.  K~max is
        // within rho of the exact-arithmetic maximum Kmax (every K~ is within rho of its K), every literal variance within Ev
        // of f^2 K, so devianceMax = f sqrt(Kmax) (1 +- delta), delta <= rho/2 + Ev / (2 f^2 Kmax) + roundings; T moves by
        // |dc| s |m - Imin| <= |k| (s / devianceMax) delta 255.  k_refine decides what this leaves open (coefficient interval
        // from the exact Kmax), the literal fix-up what THAT leaves open (literal devianceMax).
        const float kmax = __uint_as_float(g[page].v32max_bits);
        const float klow = kmax / (1.0f + fp.rho) - fp.kabs;
        const float fl = (float)tp.f;
        if (klow > fp.vthr32 && klow > 64.0f * fp.ev2) {
            const float c = (float)tp.k / (__builtin_amdgcn_sqrtf(kmax) * fl);
            const float delta = 0.51f * fp.rho + (0.26f * fp.ev2 + 0.51f * fp.kabs) / klow + 4.8e-7f;   // (+ 8 u: sqrt, product, quotient, c * f)
            const float ac = fabsf(c) * (1.0f + delta);
            pk.c1 = c * fl;
            // (|T_literal - T*| grows with |coeff| through the sqrt noise: es_max)
            pk.eps1 = fp.eps1 + kZ * (2.02f * 255.0f * ac * fp.es_max + 1.02f * 255.0f * fabsf((float)tp.k) * delta);
        } else {  // (no deviation to speak of on this page: nothing is settled here, k_refine / the literal pipeline decide)
            pk.c1 = 0.0f;
            pk.eps1 = __builtin_inff();
        }
    } else if (METHOD == kWolfCollect) {
        // a pixel can only carry the literal maximum if K~ >= (1-rho) (Kmax/(1+rho) - 2 Ev / f^2)
        const float vmax = __uint_as_float(g[page].v32max_bits);
        pk.c1 = (1.0f - fp.rho) * (vmax / (1.0f + fp.rho) - fp.ev2 - 2.0f * fp.kabs) * 0.999999f;   // (kabs: sweep A's absolute error, once for the maximum, once for the segment maxima)
        if (!(fp.segmax[wid] >= pk.c1)) return;            // nothing in this wavefront's segment qualifies
    }

    // interior strip: every lane's 8-byte window fetch and the whole 512-column output span lie inside
    // the page, so no clamp, no partial store
    const int first_col = xs + 1 - tp.half;
    const bool interior = (first_col >= 0) && (first_col + SW <= tp.width) && (xs + fp.uo <= tp.ow);
    constexpr bool kFloatOk = METHOD != kWolfCollect;  // (sweep B revisits few segments and queues exact candidates: integer)
    if (METHOD == kWolfMax && !WIDE && !fp.flt && fp.flt_a && !(fp.ext && strip == fp.n_strips - 1)) {
        // sweep A with a wide window: the float32 loop in its doubling form (the extended last strip stays on the integer loop)
        if (!interior) strip_loop_f<METHOD, SH, 4, false, true>(img, out, src.step, dst.step, fp, page, xs, ys, ye, lane, pk, wid, g, rl, counters);
        else strip_loop_f<METHOD, SH, 4, false, false>(img, out, src.step, dst.step, fp, page, xs, ys, ye, lane, pk, wid, g, rl, counters);
    } else if (kFloatOk && !WIDE && fp.flt) {
        // (wave-uniform dispatch, once per wavefront: the row loop itself is branch-free in the usual configuration)
        const bool fast = !fp.bit_out && fp.nt_store && !fp.need_p0 && !(fp.uo & 7);
#define PRL_FLT_LOOP(LOV)                                                                                                          \
    do {                                                                                                                           \
        if (!interior) strip_loop_f<METHOD, SH, LOV, false, true>(img, out, src.step, dst.step, fp, page, xs, ys, ye, lane, pk, wid, g, rl, counters);  \
        else if (fast) strip_loop_f<METHOD, SH, LOV, true, false>(img, out, src.step, dst.step, fp, page, xs, ys, ye, lane, pk, wid, g, rl, counters);  \
        else strip_loop_f<METHOD, SH, LOV, false, false>(img, out, src.step, dst.step, fp, page, xs, ys, ye, lane, pk, wid, g, rl, counters);      \
    } while (0)
        switch (fp.lane_off) {
        case 0: PRL_FLT_LOOP(0); break;
        case 1: PRL_FLT_LOOP(1); break;
        case 2: PRL_FLT_LOOP(2); break;
        default: PRL_FLT_LOOP(3); break;
        }
#undef PRL_FLT_LOOP
    } else if (interior)
        strip_loop<METHOD, SH, false, WIDE>(img, out, src.step, dst.step, fp, page, xs, ys, ye, lane, pk, wid, g, rl, cand, counters);
    else
        strip_loop<METHOD, SH, true, WIDE>(img, out, src.step, dst.step, fp, page, xs, ys, ye, lane, pk, wid, g, rl, cand, counters,
                                           fp.ext && strip == fp.n_strips - 1);
}

// k_fused's wavefront -> (page, strip, row segment) mapping for the kernels that share it (k_fused_exact, k_fused_q; k_fused keeps
// its own copy inline - its code is frozen): XCD x = blockIdx & 7 takes the x-th eighth of every tier of segments, tier 0 first.
struct WaveJob {
    bool found;
    int page, strip, ys, ye;
    unsigned wid;   // canonical wavefront id over all tiers
};
__device__ __forceinline__ WaveJob find_wave_job(const FusedParams& fp)
{
    WaveJob j{};
    const unsigned wpb = blockDim.x >> 6;
    const unsigned xcd = blockIdx.x & 7u;
    const unsigned wv = threadIdx.x >> 6;
    unsigned u = __builtin_amdgcn_readfirstlane((blockIdx.x >> 3) * wpb + wv);
    unsigned wid = 0;
    int trows = 0, tsegs = 1, trow0 = 0;
    for (int k = 0; k < fp.n_tiers; ++k) {
        const unsigned n = fp.tier[k].waves;
        const unsigned lo = (unsigned)(((unsigned long long)n * xcd) >> 3), hi = (unsigned)(((unsigned long long)n * (xcd + 1u)) >> 3);
        if (u < hi - lo) {
            wid = lo + u;
            trows = fp.tier[k].rows; tsegs = fp.tier[k].segs; trow0 = fp.tier[k].row0;
            j.found = true;
            u = fp.tier[k].first;
            break;
        }
        u -= hi - lo;
    }
    if (!j.found) return j;
    const int per_page = fp.n_strips * tsegs;
    j.page = (int)(wid / (unsigned)per_page);
    const int rem = (int)(wid - (unsigned)j.page * (unsigned)per_page);
    const int seg = rem / fp.n_strips;
    j.strip = rem - seg * fp.n_strips;
    j.wid = wid + u;
    j.ys = trow0 + seg * trows;
    j.ye = min(j.ys + trows, fp.tp.oh);
    return j;
}

// ---- the second chance of a flagged page ("next" 2) ------------------------------------------------------------------
// A page is flagged when the refine queue overflowed: more pixels inside the float32 decision band than the queue holds (pages
// of stripes whose levels sit on their own threshold; DESIGN.md 6, worst_case.adversarial).  Those pixels are not undecidable -
// they are 1e-4 from their threshold and the float64 interval test settles them - there were only too many to queue.  This
// kernel is k_fused's integer loop with that test INLINE (strip_loop<..., EXACT>): same wavefront -> strip / segment mapping,
// exact integer sums for every strip (no float32 loop), no queue; true ties go to the fix-up list.  The literal pipeline
// (48 bytes per pixel) remains for pages that overflow even that list (2^17 ties).
template <int METHOD, int SH, bool WIDE>
__global__ void __launch_bounds__(256) k_fused_exact(PageSet src, PageSetOut dst, FusedParams fp, PageGlobals* __restrict__ g,
                                                    unsigned* __restrict__ counters, WorkItem* __restrict__ wl,
                                                    CornerAcc* __restrict__ acc, unsigned* __restrict__ done)
{
    const ThrParams& tp = fp.tp;
    const int lane = threadIdx.x & (kWave - 1);
    const WaveJob job = find_wave_job(fp);
    if (!job.found) return;
    const int page = job.page, strip = job.strip, ys = job.ys, ye = job.ye;
    const unsigned wid = job.wid;
    gcptr img = (gcptr)src.page(page);
    gptr out = (gptr)dst.page(page);
    const int xs = strip * fp.uo;
    PageK pk;
    pk.c1 = fp.c1;
    pk.imin = 0.0f;
    pk.p0 = -0.5f * kZ;
    pk.eps1 = fp.eps1;
    if (METHOD == PRL_FENG) {
        const double imin = (double)g[page].imin;
        const double c3 = (tp.k2 * imin + (-imin)) + 0.0;
        pk.p0 = (float)((-0.5 - c3) * (double)kZ);
    } else if (METHOD == PRL_WOLFJOLION) {
        // (the interval coefficient of k_wolf_interval is known by now - the host made this launch wait for it: the float32 test
        // uses it with the same margin arithmetic as k_fused uses sweep A's estimate; what it leaves open refine64 decides with
        // the coefficient's own bound)
        pk.imin = (float)g[page].imin * kZ;
        const float kmax = __uint_as_float(g[page].v32max_bits);
        const float klow = kmax / (1.0f + fp.rho) - fp.kabs;
        const float fl = (float)tp.f;
        if (klow > fp.vthr32 && klow > 64.0f * fp.ev2) {
            const float c = (float)tp.k / (__builtin_amdgcn_sqrtf(kmax) * fl);
            const float delta = 0.51f * fp.rho + (0.26f * fp.ev2 + 0.51f * fp.kabs) / klow + 4.8e-7f;
            const float ac = fabsf(c) * (1.0f + delta);
            pk.c1 = c * fl;
            pk.eps1 = fp.eps1 + kZ * (2.02f * 255.0f * ac * fp.es_max + 1.02f * 255.0f * fabsf((float)tp.k) * delta);
        } else {
            pk.c1 = 0.0f;
            pk.eps1 = __builtin_inff();
        }
    }
    ExactOut xo;
    xo.wl = wl; xo.acc = acc; xo.done = done;
    const int first_col = xs + 1 - tp.half;
    const bool interior = (first_col >= 0) && (first_col + SW <= tp.width) && (xs + fp.uo <= tp.ow);
    if (interior)
        strip_loop<METHOD, SH, false, WIDE, true>(img, out, src.step, dst.step, fp, page, xs, ys, ye, lane, pk, wid, g, nullptr, nullptr, counters, false, xo);
    else
        strip_loop<METHOD, SH, true, WIDE, true>(img, out, src.step, dst.step, fp, page, xs, ys, ye, lane, pk, wid, g, nullptr, nullptr, counters,
                                                 fp.ext && strip == fp.n_strips - 1, xo);
}

// ---- the threshold sweep for windows of 33 .. 129 columns with float window rows ("next" 7; strip_loop_f<..., QINT>) -----
// k_fused's wavefront -> strip / segment mapping; interior strips take the float loop with integer horizontal Q sums, the border
// strips the float loop's border form (the extended last one too).  Every method's threshold sweep.  Its own
// kernel name: the instantiations of k_fused keep theirs (tests/test_frozen_loop.py).
template <int METHOD, int SH>
__global__ void __launch_bounds__(256) k_fused_q(PageSet src, PageSetOut dst, FusedParams fp, PageGlobals* __restrict__ g,
                                                RefItem* __restrict__ rl, WorkItem* __restrict__ cand, unsigned* __restrict__ counters,
                                                int edge_float)
{
    const ThrParams& tp = fp.tp;
    const int lane = threadIdx.x & (kWave - 1);
    const WaveJob job = find_wave_job(fp);
    if (!job.found) return;
    const int page = job.page, strip = job.strip, ys = job.ys, ye = job.ye;
    const unsigned wid = job.wid;
    if (__hip_atomic_load(&g[page].worklist_overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1u) return;   // (as k_fused)
    gcptr img = (gcptr)src.page(page);
    gptr out = (gptr)dst.page(page);
    const int xs = strip * fp.uo;
    PageK pk;
    pk.c1 = fp.c1;
    pk.imin = 0.0f;
    pk.p0 = -0.5f * kZ;
    pk.eps1 = fp.eps1;
    if (METHOD == PRL_FENG) {
        const double imin = (double)g[page].imin;
        const double c3 = (tp.k2 * imin + (-imin)) + 0.0;  // binarizeFeng.cpp:137 with r2 = c2 = 1 (as k_fused)
        pk.p0 = (float)((-0.5 - c3) * (double)kZ);
    } else if (METHOD == PRL_WOLFJOLION) {   // (k_fused's arithmetic: the coefficient from sweep A's float32 maximum, the margin widened by what it can be off)
        pk.imin = (float)g[page].imin * kZ;
        const float kmax = __uint_as_float(g[page].v32max_bits);
        const float klow = kmax / (1.0f + fp.rho) - fp.kabs;
        const float fl = (float)tp.f;
        if (klow > fp.vthr32 && klow > 64.0f * fp.ev2) {
            const float c = (float)tp.k / (__builtin_amdgcn_sqrtf(kmax) * fl);
            const float delta = 0.51f * fp.rho + (0.26f * fp.ev2 + 0.51f * fp.kabs) / klow + 4.8e-7f;
            const float ac = fabsf(c) * (1.0f + delta);
            pk.c1 = c * fl;
            pk.eps1 = fp.eps1 + kZ * (2.02f * 255.0f * ac * fp.es_max + 1.02f * 255.0f * fabsf((float)tp.k) * delta);
        } else {
            pk.c1 = 0.0f;
            pk.eps1 = __builtin_inff();
        }
    }
    const int first_col = xs + 1 - tp.half;
    const bool interior = (first_col >= 0) && (first_col + SW <= tp.width) && (xs + fp.uo <= tp.ow);
    const bool fast = !fp.bit_out && fp.nt_store && !fp.need_p0 && !(fp.uo & 7);   // (as k_fused: wave-uniform, once per wavefront)
    if (interior) {
        if (fast) strip_loop_f<METHOD, SH, 4, true, false, true>(img, out, src.step, dst.step, fp, page, xs, ys, ye, lane, pk, wid, g, rl, counters);
        else strip_loop_f<METHOD, SH, 4, false, false, true>(img, out, src.step, dst.step, fp, page, xs, ys, ye, lane, pk, wid, g, rl, counters);
    } else if (!edge_float) {
        strip_loop<METHOD, SH, true, false>(img, out, src.step, dst.step, fp, page, xs, ys, ye, lane, pk, wid, g, rl, cand, counters,
                                            fp.ext && strip == fp.n_strips - 1);
    } else {
        strip_loop_f<METHOD, SH, 4, false, true, true>(img, out, src.step, dst.step, fp, page, xs, ys, ye, lane, pk, wid, g, rl, counters,
                                                       fp.ext && strip == fp.n_strips - 1);
    }
}

// ---- second stage: float64 interval test of the queued pixels ------------------------------------------------------
// One thread per queued pixel, on the window sums that travel with it: exact ones from the integer pipeline; from the
// float32 pipeline an exact S and a Q~ within flt_dq of the exact sum, which enters the interval as extra uncertainty of q
// (round 4: the first version rebuilt the sums of EVERY queued pixel, a wavefront each - 0.85 ms for the 3.9 10^5 pixels a
// batch of real scans queues).  Only what that test leaves open has its sums rebuilt exactly (a wavefront per pixel).
// The queue is kRefBuckets lists with their own counters (see ref_push).
template <int METHOD>
__global__ void __launch_bounds__(256) k_refine(PageSet src, PageSetOut dst, FusedParams fp, PageGlobals* __restrict__ g,
                                               const RefItem* __restrict__ rl, WorkItem* __restrict__ wl,
                                               unsigned* __restrict__ counters, CornerAcc* __restrict__ acc,
                                               unsigned* __restrict__ done)
{
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x, nthreads = gridDim.x * blockDim.x;
    const int lane = threadIdx.x & 63;
    const ThrParams& tp = fp.tp;
    const double eq_approx = fp.flt_dq * tp.f;
    // Wavefront W serves bucket W mod kRefBuckets, as the (W / kRefBuckets)-th of the wavefronts that do (the launch has a
    // multiple of kRefBuckets wavefronts): one counter load per wavefront, an empty bucket - the rule for small calls - ends it.
    const unsigned wave_id = tid >> 6, n_waves = nthreads >> 6;
    const unsigned per_bucket = n_waves / (unsigned)kRefBuckets;   // >= 1
    if (wave_id >= per_bucket * (unsigned)kRefBuckets) return;
    {
        const unsigned b = wave_id & (unsigned)(kRefBuckets - 1);
        const unsigned n = min(counters[64 + kRefCounterStride * b], kRefBucketCap);
        // (whole wavefronts go round together: the rebuild below is a wavefront's job; i - lane is wave-uniform)
        for (unsigned i = (wave_id / (unsigned)kRefBuckets) * kWave + lane; i - lane < n; i += per_bucket * kWave) {
            const bool valid = i < n;
            RefItem it = rl[(size_t)b * kRefBucketCap + (valid ? i : 0u)];
            const bool approx = (it.p & kRefApprox) != 0u && fp.flt_dq > 0.0;   // (w <= 21: the float32 loop's sums are exact, nothing to rebuild)
            it.p &= 0xffu;
            unsigned r = 2;
            double imin = 0.0, coeff = 0.0, crel = 0.0;
            // (a page that is flagged already will be redone whole - by the exact sweep or the literal pipeline: what the queue still
            // holds of it is moot; adversarial pages left 2^21 entries here, each with a wavefront's rebuild of its sums: 4.7 ms)
            const bool moot = valid && g[it.page].worklist_overflow != 0u;
            if (valid && !moot) {
                imin = (double)g[it.page].imin;
                coeff = g[it.page].coeff;
                if (METHOD == PRL_WOLFJOLION) crel = g[it.page].coeff_rel;
                // one thread per pixel: float64 interval test on the sums the sweep sent along (the float32 pipeline's Q~ with
                // its rounding bound as extra uncertainty of q)
                r = refine64<METHOD>(fp, it.S, it.Q, it.p, imin, coeff, approx ? eq_approx : 0.0, crel);
            }
            // what that leaves open and has an approximate Q: the wavefront rebuilds S and Q exactly from the page - padded rows
            // y+1 .. y+w-1, columns x+1 .. x+w-1 of the replicate-padded page, exact in u32 - one pixel at a
            // time, and the owner lane repeats the test with exact sums
            // (Feng's threshold depends on S only - exact in both loops - and on the variance through the s > 0 guard alone:
            // when the guard held with the approximate Q, exact sums would repeat the same undecided answer - its ties)
            bool rebuild = valid && !moot && r == 2 && approx;
            if (METHOD == PRL_FENG && rebuild) {
                const double mm = (double)it.S * tp.f, vv = (double)it.Q * tp.f - mm * mm;
                rebuild = !(vv > 8.0 * (fp.Eq + eq_approx + 2.0 * mm * fp.Em + fp.Em * fp.Em) + 1e-9);
            }
            unsigned long long todo = __ballot(rebuild);
            while (todo) {
                const int owner = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                const int pg_i = __shfl(it.page, owner, kWave), yy = __shfl(it.y, owner, kWave), xx = __shfl(it.x, owner, kWave);
                const uint8_t* pg = src.page(pg_i);
                unsigned S = 0, Q = 0;
                for (int pc = xx + 1 + lane; pc <= xx + tp.w - 1; pc += 64) {
                    const uint8_t* col = pg + clampi(pc - tp.half, 0, tp.width - 1);
                    for (int pr = yy + 1; pr <= yy + tp.w - 1; ++pr) {
                        const unsigned bb = col[(size_t)clampi(pr - tp.half, 0, tp.height - 1) * src.step];
                        S += bb;
                        Q += bb * bb;
                    }
                }
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) {
                    S += __shfl_xor(S, d, kWave);
                    Q += __shfl_xor(Q, d, kWave);
                }
                if (lane == owner) r = refine64<METHOD>(fp, S, Q, it.p, imin, coeff, 0.0, crel);
            }
            // slots of the fix-up list: one atomic per wavefront (a same-address atomic with return costs ~20 ns under contention:
            // Feng's 21 000 ties on 256 pages took k_refine 0.39 ms with one per pixel)
            const bool to_fixup = valid && !moot && r == 2;
            const unsigned long long fm = __ballot(to_fixup);
            unsigned slot_base = 0;
            if (fm) {
                const int leader = __ffsll((long long)fm) - 1;
                if (lane == leader) slot_base = atomicAdd(&counters[1], (unsigned)__popcll(fm));
                slot_base = (unsigned)__shfl((int)slot_base, leader, kWave);
            }
            if (!valid || moot) continue;
            if (r != 2) {
                store_decision(dst, fp.bit_out, it.page, it.y, it.x, r);
                atomicAdd(&g[it.page].n_refined, 1u);
            } else {
                atomicAdd(&g[it.page].n_exact, 1u);
                if (METHOD == PRL_WOLFJOLION) atomicOr(&g[it.page].need_literal, 1u);   // the fix-up evaluates with the LITERAL devianceMax (k_wolf_literal)
                const unsigned idx = slot_base + (unsigned)__popcll(fm & ((1ull << lane) - 1ull));
                if (idx < fp.wl_cap) {
                    WorkItem w;
                    w.page = it.page;
                    w.y = it.y;
                    w.x = it.x;
                    w.pad = 0;
                    wl[idx] = w;
#pragma unroll
                    for (int k = 0; k < 8; ++k) acc[idx].a[k] = 0ull;  // (instead of a 1 MB memset per call: nothing is queued as a rule)
                    done[idx] = 0u;
                } else {
                    atomicOr(&g[it.page].worklist_overflow, 2u);
                }
            }
        }
    }
}

// ---- fix-up: literal evaluation of the queued pixels from absolute integral corners -----------
// One workgroup per queued pixel.  The four absolute corners of cv::integral are rebuilt from the
// page: II'(Y,X) = sum of the replicate-padded image over rows <= Y, cols <= X (exact integers), then
// the literal float64 sequence of prl_device_math.h decides.  Rare by construction (see header).
// ---- absolute integral corners of queued pixels ----------------------------------------------------------
// cv::integral's value at padded (Y, X) is the sum of the replicate-padded page over rows <= Y, cols <= X.
// For a queued output pixel (y, x) the literal 4-tap filter reads the corners (y, x), (y, x+w-1),
// (y+w-1, x), (y+w-1, x+w-1) of both integrals.  They are rebuilt exactly from the page: the padded
// region [0..y+w-1] x [0..x+w-1] is cut into (top|bottom) x (left|right); every page row is read once
// (16 workgroups per pixel share the rows, dword loads + V_DOT4_U32_U8 for sum and sum of squares) and
// weighted by how many padded rows/columns replicate it.  Integer arithmetic: exact.
constexpr int kSplit = 64;  // workgroups per queued pixel: the row loop of a wavefront is a chain of dependent loads, so the
                            // parallelism has to come from the grid (16 -> 64: Wolf-Jolion's candidate pass 0.35 -> see DESIGN 4.2)

// number of padded indices i in [lo, hi] that replicate-clamp to page index r (page size n, padding h)
__device__ __forceinline__ int pad_count(int lo, int hi, int r, int n, int h)
{
    int a = r + h, b = r + h;          // i - h == r
    if (r == 0) a = 0;                 // i < h clamps to 0
    if (r == n - 1) b = 0x3fffffff;    // i > h + n - 1 clamps to n - 1
    a = max(a, lo);
    b = min(b, hi);
    return b >= a ? b - a + 1 : 0;
}

// sum and sum of squares of page[row][ca..cb] (inclusive, ca <= cb) over the lanes of one wavefront
__device__ __forceinline__ void row_range_sums(const uint8_t* row, int ca, int cb, int lane, unsigned* s_out,
                                               unsigned* q_out)
{
    unsigned s = 0, q = 0;
    const int nbytes = cb - ca + 1;
    // 16 bytes per lane and step (the loop is a chain of dependent-latency loads: fewer, wider loads), dwords and
    // single bytes for the tail
    const int n16 = nbytes / 16;
    for (int t = lane; t < n16; t += kWave) {
        uint4 v;
        __builtin_memcpy(&v, row + ca + 16 * t, 16);
        s = __builtin_amdgcn_udot4(v.x, 0x01010101u, s, false);
        q = __builtin_amdgcn_udot4(v.x, v.x, q, false);
        s = __builtin_amdgcn_udot4(v.y, 0x01010101u, s, false);
        q = __builtin_amdgcn_udot4(v.y, v.y, q, false);
        s = __builtin_amdgcn_udot4(v.z, 0x01010101u, s, false);
        q = __builtin_amdgcn_udot4(v.z, v.z, q, false);
        s = __builtin_amdgcn_udot4(v.w, 0x01010101u, s, false);
        q = __builtin_amdgcn_udot4(v.w, v.w, q, false);
    }
    const int rest = n16 * 16 + lane;  // the last nbytes % 16 (< 16) bytes, one per lane
    if (rest < nbytes) {
        const unsigned b = row[ca + rest];
        s += b;
        q += b * b;
    }
    *s_out = s;
    *q_out = q;
}

// FINAL: the workgroup that delivers the last of a pixel's kSplit partial sums also runs the literal evaluation
// (k_fixup_final's body) - one launch less per call; `done` counts the arrivals per pixel (zeroed by k_refine when it
// queues the pixel).
__device__ __forceinline__ void literal_mq(const CornerAcc& c, double f, double* m, double* q);

template <bool FINAL>
__global__ void __launch_bounds__(256) k_corner_partial(PageSet src, FusedParams fp, const WorkItem* __restrict__ items,
                                                       const unsigned* __restrict__ counters, int which,
                                                       CornerAcc* __restrict__ acc, PageSetOut dst,
                                                       const PageGlobals* __restrict__ g, unsigned* __restrict__ done)
{
    const ThrParams& tp = fp.tp;
    // (Wolf-Jolion's candidates, which == 2: only when some pixel reached the literal fix-up - counters[1] - and then only
    // the candidates of the pages concerned; the rule is an immediate return)
    if (!FINAL && counters[1] == 0u) return;
    const unsigned n_list = min(counters[which], fp.wl_cap);
    const unsigned n = counters[kCntPageMajor] != 0u ? 0u : n_list;   // (non-zero: k_corner_rows has built the sums of this list page by page)
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x >> 6;
    const int W = tp.width, H = tp.height, h = tp.half;
    for (unsigned it = blockIdx.y; it < n; it += gridDim.y) {
        const WorkItem wi = items[it];
        if (!FINAL && !g[wi.page].need_literal) continue;   // (uniform over the workgroup)
        const uint8_t* img = src.page(wi.page);
        const int Y0 = wi.y, X0 = wi.x, Y1 = wi.y + tp.w - 1, X1 = wi.x + tp.w - 1;
        // page rows that appear in padded rows [0..Y1], split over the kSplit workgroups of this pixel
        const int r_last = clampi(Y1 - h, 0, H - 1);
        const int per = (r_last + 1 + kSplit - 1) / kSplit;
        const int r_begin = blockIdx.x * per, r_end = min(r_begin + per, r_last + 1);
        // page columns of the left part [0..X0] and the right part [X0+1..X1] (interior + replicated edges)
        const int lca = 0, lcb = clampi(X0 - h, 0, W - 1);
        const int rca = clampi(X0 + 1 - h, 0, W - 1), rcb = clampi(X1 - h, 0, W - 1);
        // multiplicities of the two edge columns inside each part (interior columns count once)
        const int l_m0 = pad_count(0, X0, 0, W, h), l_mW = pad_count(0, X0, W - 1, W, h);
        const int r_m0 = pad_count(X0 + 1, X1, 0, W, h), r_mW = pad_count(X0 + 1, X1, W - 1, W, h);
        unsigned long long a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int r = r_begin + wv; r < r_end; r += 4) {
            const uint8_t* row = img + (size_t)r * src.step;
            unsigned sl, ql, sr, qr;
            row_range_sums(row, lca, lcb, lane, &sl, &ql);
            row_range_sums(row, rca, rcb, lane, &sr, &qr);
            if (lane == 0) {  // replicated edge columns: add the missing (multiplicity - 1) copies
                const unsigned e0 = row[0], eW = row[W - 1];
                // a part contains page column 0 (resp. W-1) once through its range iff its multiplicity > 0
                if (l_m0 > 0) { sl += (unsigned)(l_m0 - 1) * e0; ql += (unsigned)(l_m0 - 1) * e0 * e0; }
                if (l_mW > 0 && lcb == W - 1) { sl += (unsigned)(l_mW - 1) * eW; ql += (unsigned)(l_mW - 1) * eW * eW; }
                if (r_m0 > 0 && rca == 0) { sr += (unsigned)(r_m0 - 1) * e0; qr += (unsigned)(r_m0 - 1) * e0 * e0; }
                if (r_mW > 0) { sr += (unsigned)(r_mW - 1) * eW; qr += (unsigned)(r_mW - 1) * eW * eW; }
            }
            const unsigned long long ct = (unsigned long long)pad_count(0, Y0, r, H, h);
            const unsigned long long cb2 = (unsigned long long)pad_count(Y0 + 1, Y1, r, H, h);
            a[0] += ct * sl;  a[1] += ct * sr;  a[2] += cb2 * sl;  a[3] += cb2 * sr;
            a[4] += ct * ql;  a[5] += ct * qr;  a[6] += cb2 * ql;  a[7] += cb2 * qr;
        }
        // one atomic per workgroup and sum (256 wavefronts of a pixel adding to the same 8 words serialised)
        __shared__ unsigned long long part[4][8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            unsigned long long v = a[k];
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, kWave);
            if (lane == 0) part[wv][k] = v;
        }
        __syncthreads();
        if (threadIdx.x < 8) {
            const unsigned long long v = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
            if (v != 0) atomicAdd(&acc[it].a[threadIdx.x], v);
        }
        if constexpr (FINAL) {
            __threadfence();   // this workgroup's sums are visible before its arrival is
            __syncthreads();
            if (threadIdx.x == 0 && atomicAdd(&done[it], 1u) == gridDim.x - 1) {
                __threadfence();
                CornerAcc c;
#pragma unroll
                for (int k = 0; k < 8; ++k) c.a[k] = atomicAdd(&acc[it].a[k], 0ull);  // (read at the L2, past this CU's caches)
                double m, q;
                literal_mq(c, tp.f, &m, &q);
                const double s = dev_from(m, q);
                const PageGlobals& pg = g[wi.page];
                const double T = threshold_literal(tp, m, s, (double)pg.imin, pg.coeff);
                const unsigned p = img[(size_t)wi.y * src.step + wi.x];
                store_decision(dst, fp.bit_out, wi.page, wi.y, wi.x, decide_literal(p, T));
            }
        }
        __syncthreads();
    }
    if constexpr (FINAL) {
        if (fp.ep_host) {   // (uniform over the grid)
            // Empty queue (the rule): no workgroup touches the globals or needs the counters to be anything but zero, so
            // workgroup (0, 0) can run the epilogue without waiting for anybody.  Otherwise the last one to finish runs it
            // (1024 arrivals on one word cost ~50 us: acceptable for the rare page that reaches the literal fix-up).
            __shared__ unsigned s_last;
            if (n_list == 0) {   // (not `n`: with a non-empty list the other workgroups still read the counters this epilogue resets)
                if (threadIdx.x == 0) s_last = (blockIdx.x == 0 && blockIdx.y == 0) ? 1u : 0u;
            } else {
                __threadfence();
                __syncthreads();
                if (threadIdx.x == 0) s_last = atomicAdd(&fp.ep_counters[kEpArrive], 1u) == gridDim.x * gridDim.y - 1 ? 1u : 0u;
            }
            __syncthreads();
            if (s_last) {   // every other workgroup has finished: nothing reads the counters or the globals any more
                for (int i = threadIdx.x; i < fp.ep_pages; i += blockDim.x) {
                    fp.ep_host[i] = fp.ep_dev[i];
                    PageGlobals z;
                    z.imin = 255; z.smax_found = 0; z.smax_bits = 0ull; z.coeff = 0.0;
                    z.n_refined = 0; z.n_exact = 0; z.worklist_overflow = 0; z.v32max_bits = 0; z.n_cand = 0; z.need_literal = 0; z.kmax_bits = 0ull; z.coeff_rel = 0.0; z.cand_overflow = 0; z.reserved0 = 0;
                    fp.ep_dev[i] = z;   // = k_init_globals
                }
                if (threadIdx.x < 64) fp.ep_counters[threadIdx.x] = 0u;
                for (int b = threadIdx.x; b < kRefBuckets; b += blockDim.x) fp.ep_counters[64 + kRefCounterStride * b] = 0u;
            }
        }
    }
}

__device__ __forceinline__ void literal_mq(const CornerAcc& c, double f, double* m, double* q)
{
    const unsigned long long* t = c.a;
    const double A = (double)t[0], B = (double)(t[0] + t[1]), C = (double)(t[0] + t[2]),
                 D = (double)(t[0] + t[1] + t[2] + t[3]);
    const double AQ = (double)t[4], BQ = (double)(t[4] + t[5]), CQ = (double)(t[4] + t[6]),
                 DQ = (double)(t[4] + t[5] + t[6] + t[7]);
    *m = box4_literal(A, B, C, D, f);
    *q = box4_literal(AQ, BQ, CQ, DQ, f);
}

// Wolf-Jolion, devianceMax (binarizeWolfJolion.cpp:118-121) in two precisions.
// k_wolf_interval: from the exact integer maximum Kmax of K = w^2 Q - S^2 (sweep B).  The exact-arithmetic variance maximum is
// v* = f^2 Kmax; every literal variance is within Ev of its exact value and the literal sqrt is correctly rounded, so
// devianceMax_literal lies in [sqrt(v* - Ev), sqrt(v* + Ev)] (1 +- 2^-52): coeff = k / sqrt(v*) and a relative bound
// coeff_rel on its distance from the literal k / devianceMax - all the threshold sweep's margin and k_refine's intervals need.
__global__ void k_wolf_interval(FusedParams fp, PageGlobals* __restrict__ g, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    PageGlobals& pg = g[i];
    const double f = fp.tp.f;
    const double v = (double)pg.kmax_bits * f * f;          // (Kmax < 2^49: exact conversion; two roundings + f's own)
    const double Ev = fp.Eq + 2.0 * 255.0 * fp.Em + fp.Em * fp.Em + 8.9e-16 * 2.0 * 65025.0 + 4e-15 * v;
    if (v > 64.0 * Ev) {
        pg.coeff = fp.tp.k / sqrt(v);
        pg.coeff_rel = 0.51 * Ev / (v - Ev) + 1e-14;
    } else {   // no deviation to speak of (or no candidate at all): nothing can be said about the literal maximum
        pg.coeff = 0.0;
        pg.coeff_rel = 2.0;   // refine64 refuses; the pixels go to the literal fix-up, which computes the literal maximum
    }
}

// k_wolf_literal: the LITERAL maximum, for the pages that need it (a pixel reached the literal fix-up: need_literal) - the
// literal deviation of every candidate of sweep B from its absolute integral corners (k_corner_partial<false>); their maximum
// is exactly cv::minMaxLoc(localDevianceValues)'s devianceMax.  Two small kernels behind k_refine, which return at once when
// no pixel of the call reached the fix-up (counters[1] == 0).
__global__ void __launch_bounds__(256) k_wolf_final(FusedParams fp, PageGlobals* __restrict__ g,
                                                   const WorkItem* __restrict__ cand, const CornerAcc* __restrict__ acc,
                                                   const unsigned* __restrict__ counters)
{
    if (counters[1] == 0u) return;
    const unsigned n = min(counters[2], fp.wl_cap);
    for (unsigned it = blockIdx.x * blockDim.x + threadIdx.x; it < n; it += gridDim.x * blockDim.x) {
        if (!g[cand[it].page].need_literal) continue;
        double m, q;
        literal_mq(acc[it], fp.tp.f, &m, &q);
        const double s = dev_from(m, q);
        if (s == s) {  // NaN never wins minMaxLoc
            atomicMax(&g[cand[it].page].smax_bits, (unsigned long long)__double_as_longlong(s) & 0x7fffffffffffffffull);
            atomicOr(&g[cand[it].page].smax_found, 1);
        }
    }
}

__global__ void k_wolf_literal_coeff(FusedParams fp, PageGlobals* __restrict__ g, int n, const unsigned* __restrict__ counters)
{
    if (counters[1] == 0u) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    PageGlobals& pg = g[i];
    if (!pg.need_literal) return;
    if (pg.cand_overflow) {   // the candidate list does not hold every candidate of this page: the literal pipeline redoes it
        pg.worklist_overflow = 2u;   // (bit 1: nothing short of the literal pipeline helps)
        return;
    }
    const double smax = pg.smax_found ? __longlong_as_double((long long)pg.smax_bits)
                                      : -1.7976931348623157e308;  // minMaxLoc's initial -DBL_MAX
    pg.coeff = fp.tp.k / smax;   // double coeff = k / devianceMax  - binarizeWolfJolion.cpp:121 (IEEE division on the device)
}

// ---- corner sums, page by page (round 4) ---------------------------------------------------------------------------------
// k_corner_partial rebuilds the absolute integral corners of ONE queued pixel from scratch: every page row above it is read
// (up to the whole page, 8 MB on average for a 4K page).  Fine for the handful of pixels a call queues as a rule; but Feng's
// threshold (1 + (1 - alpha1)) m + c3 meets p - 0.5 EXACTLY for about 5 pixels in 10^6 (rational coincidences of the integer
// window sum: 21 153 pixels on 256 synthetic 4K pages, 2 508 on the real scans: 28 - 750 ms), and a Wolf-Jolion page that
// needs its literal devianceMax has up to thousands of candidates.  From kPageMajorMin queued pixels on the work is organised
// by PAGE instead: the pixels are grouped by page (k_group_items), and a workgroup walks a chunk of a page's rows ONCE - row
// prefix sums of P and P^2 in LDS - while each of its threads owns one queued pixel of that page and adds the row's
// contribution to its eight sums (the same multiplicity arithmetic as k_corner_partial, range sums as differences of two
// prefix values).  One pass over a page serves all its pixels.
struct GroupArrays {
    unsigned* sidx;     // item indices grouped by page
    unsigned* pstart;   // [n_pages + 1] first position of each page's items in sidx
    unsigned* pcur;     // [n_pages] scatter cursors
    unsigned* plist;    // pages that have items
};

// one workgroup of 1024 threads; `which`: 1 = the fix-up list, 2 = Wolf-Jolion's candidates (only those of pages with need_literal)
__global__ void __launch_bounds__(1024) k_group_items(const WorkItem* __restrict__ items, unsigned* __restrict__ counters, int which,
                                                      const PageGlobals* __restrict__ g, int n_pages, int width, unsigned cap, GroupArrays ga)
{
    __shared__ unsigned s_cnt[1024], s_nz[1024];
    const unsigned t = threadIdx.x;
    const unsigned n = min(counters[which], cap);
    const bool lazy = which == 2;
    if ((lazy && counters[1] == 0u) || n < kPageMajorMin || width > kPageMajorMaxW) {   // (uniform)
        if (t == 0) counters[kCntPageMajor] = 0u;
        return;
    }
    for (unsigned p = t; p < (unsigned)n_pages; p += 1024u) ga.pcur[p] = 0u;
    __syncthreads();
    for (unsigned i = t; i < n; i += 1024u) {
        const int page = items[i].page;
        if (lazy && !g[page].need_literal) continue;
        atomicAdd(&ga.pcur[page], 1u);
    }
    __threadfence();
    __syncthreads();
    // exclusive prefix over the pages (each thread a contiguous run of pages) + compaction of the pages that have items
    const unsigned run = ((unsigned)n_pages + 1023u) / 1024u, p0 = t * run, p1 = min(p0 + run, (unsigned)n_pages);
    unsigned c = 0, z = 0;
    for (unsigned p = p0; p < p1; ++p) {
        const unsigned k = atomicAdd(&ga.pcur[p], 0u);
        c += k;
        z += k != 0u;
    }
    s_cnt[t] = c;
    s_nz[t] = z;
    __syncthreads();
    for (unsigned d = 1; d < 1024u; d <<= 1) {
        const unsigned a = t >= d ? s_cnt[t - d] : 0u, b = t >= d ? s_nz[t - d] : 0u;
        __syncthreads();
        s_cnt[t] += a;
        s_nz[t] += b;
        __syncthreads();
    }
    unsigned pos = s_cnt[t] - c, zi = s_nz[t] - z;
    for (unsigned p = p0; p < p1; ++p) {
        const unsigned k = atomicAdd(&ga.pcur[p], 0u);
        ga.pstart[p] = pos;
        if (k) ga.plist[zi++] = p;
        pos += k;
    }
    if (t == 1023u) {
        ga.pstart[n_pages] = s_cnt[1023];
        counters[kCntPageMajor] = s_nz[1023];
    }
    __threadfence();
    __syncthreads();
    for (unsigned p = p0; p < p1; ++p) ga.pcur[p] = ga.pstart[p];   // scatter cursors
    __threadfence();
    __syncthreads();
    for (unsigned i = t; i < n; i += 1024u) {
        const int page = items[i].page;
        if (lazy && !g[page].need_literal) continue;
        ga.sidx[atomicAdd(&ga.pcur[page], 1u)] = i;
    }
}

// grid (kRowChunks, page slots).  A workgroup walks one chunk of a page's rows; each of its NW wavefronts takes every NW-th row
// of the chunk ON ITS OWN - the row's BPL bytes per lane in registers, dword sums, one DPP scan, the row's dwords and the
// dword-granular prefixes of P and P*P in the wavefront's own piece of LDS - no workgroup barrier per row (the first version,
// one row per workgroup and three barriers, spent 3.4 us a row waiting).  Lane L owns the L-th queued pixel of the page (64 at
// a time) and adds the row's contribution to its eight sums (k_corner_partial's multiplicity arithmetic; a range sum is the
// difference of two prefixes, a prefix at a column = its dword's entry minus the bytes behind the column, two v_dot4).  The
// wavefronts' partial sums meet in LDS once per group of pixels, then one atomicAdd per sum.
// BPL = 64 (rows up to 4096 bytes, NW = 4) or 128 (up to 8192, NW = 2): 55 KB of LDS either way.
template <bool FINAL, int BPL>
__global__ void __launch_bounds__(256) k_corner_rows(PageSet src, FusedParams fp, const WorkItem* __restrict__ items, GroupArrays ga,
                                                    const unsigned* __restrict__ counters, CornerAcc* __restrict__ acc, PageSetOut dst,
                                                    const PageGlobals* __restrict__ g, unsigned* __restrict__ done)
{
    constexpr int DPL = BPL / 4;                 // dwords per lane
    constexpr int JW = kWave * DPL;              // dwords of LDS row per wavefront (before padding)
#define PJ(j) ((j) + ((j) >> 4))
    constexpr int JP = JW + (JW >> 4) + 1;
    extern __shared__ unsigned s_row[];          // per wavefront: raw[JP], ps4[JP], pq4[JP]; reused for the final reduction
    const ThrParams& tp = fp.tp;
    const unsigned n_slots = counters[kCntPageMajor];
    if (n_slots == 0u) return;
    const int W = tp.width, H = tp.height, h = tp.half;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, NW = blockDim.x >> 6;
    unsigned* raw = s_row + (size_t)wv * 3 * JP;
    unsigned* ps4 = raw + JP;
    unsigned* pq4 = raw + 2 * JP;
    const int c0 = BPL * lane, j0 = DPL * lane;
    const bool whole = c0 + BPL <= W;
    const int per = (H + kRowChunks - 1) / kRowChunks;
    const int r_begin = blockIdx.x * per, r_stop = min(r_begin + per, H);
    auto prefix_at = [&](int c, unsigned* s_out, unsigned* q_out) {   // sums of P and P*P over columns 0 .. c of the row in LDS
        const int j = c >> 2, rr = c & 3;
        const unsigned x = raw[PJ(j)];
        const unsigned xh = rr == 3 ? 0u : (x & (0xffffffffu << (8 * (rr + 1))));
        *s_out = ps4[PJ(j)] - __builtin_amdgcn_udot4(xh, 0x01010101u, 0u, false);
        *q_out = pq4[PJ(j)] - __builtin_amdgcn_udot4(xh, xh, 0u, false);
    };
    for (unsigned slot = blockIdx.y; slot < n_slots; slot += gridDim.y) {
        const int page = (int)ga.plist[slot];
        const uint8_t* img = src.page(page);
        const unsigned i0 = ga.pstart[page], i1 = ga.pstart[page + 1];
        for (unsigned g0 = i0; g0 < i1; g0 += (unsigned)kWave) {
            const bool has = g0 + (unsigned)lane < i1;
            const unsigned idx = has ? ga.sidx[g0 + lane] : 0u;
            const WorkItem wi = items[idx];
            // constants of this lane's pixel (k_corner_partial's arithmetic)
            const int Y0 = wi.y, X0 = wi.x, Y1 = wi.y + tp.w - 1, X1 = wi.x + tp.w - 1;
            const int r_last = clampi(Y1 - h, 0, H - 1);
            const int lcb = clampi(X0 - h, 0, W - 1);
            const int rca = clampi(X0 + 1 - h, 0, W - 1), rcb = clampi(X1 - h, 0, W - 1);
            const int l_m0 = pad_count(0, X0, 0, W, h), l_mW = pad_count(0, X0, W - 1, W, h);
            const int r_m0 = pad_count(X0 + 1, X1, 0, W, h), r_mW = pad_count(X0 + 1, X1, W - 1, W, h);
            unsigned long long a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            unsigned nx[DPL];   // the next row of this wavefront, fetched while the current one is worked on
            auto fetch = [&](int r) {
                const uint8_t* row = img + (size_t)r * src.step;
                if (whole) {
#pragma unroll
                    for (int k = 0; k < DPL; k += 4) {
                        uint4 v;
                        __builtin_memcpy(&v, row + c0 + 4 * k, 16);
                        nx[k] = v.x; nx[k + 1] = v.y; nx[k + 2] = v.z; nx[k + 3] = v.w;
                    }
                } else {   // the lane over the row's end (and the lanes beyond it): dwords while they fit, the last bytes one by one, zeros after
#pragma unroll
                    for (int k = 0; k < DPL; ++k) {
                        const int c = c0 + 4 * k;
                        unsigned v = 0u;
                        if (c + 4 <= W) {
                            __builtin_memcpy(&v, row + c, 4);
                        } else {
#pragma unroll
                            for (int b = 0; b < 4; ++b)
                                if (c + b < W) v |= (unsigned)row[c + b] << (8 * b);
                        }
                        nx[k] = v;
                    }
                }
            };
            if (r_begin + wv < r_stop) fetch(r_begin + wv);
            for (int r = r_begin + wv; r < r_stop; r += NW) {
                unsigned d[DPL], dsum[DPL], dsq[DPL], s = 0, q = 0;
#pragma unroll
                for (int k = 0; k < DPL; ++k) d[k] = nx[k];
                if (r + NW < r_stop) fetch(r + NW);
#pragma unroll
                for (int k = 0; k < DPL; ++k) {
                    dsum[k] = __builtin_amdgcn_udot4(d[k], 0x01010101u, 0u, false);
                    dsq[k] = __builtin_amdgcn_udot4(d[k], d[k], 0u, false);
                    s += dsum[k];
                    q += dsq[k];
                }
                unsigned so = wave_scan_incl(s) - s, qo = wave_scan_incl(q) - q;
#pragma unroll
                for (int k = 0; k < DPL; ++k) {
                    so += dsum[k];
                    qo += dsq[k];
                    raw[PJ(j0 + k)] = d[k];
                    ps4[PJ(j0 + k)] = so;
                    pq4[PJ(j0 + k)] = qo;
                }
                __builtin_amdgcn_wave_barrier();   // (one wavefront: LDS operations complete in order; this only pins the schedule)
                if (has && r <= r_last) {
                    const unsigned e0 = raw[0] & 0xffu, eW = (raw[PJ((W - 1) >> 2)] >> (8 * ((W - 1) & 3))) & 0xffu;
                    const unsigned e0q = e0 * e0, eWq = eW * eW;
                    unsigned sl, ql, sr, qr;
                    prefix_at(lcb, &sl, &ql);                                                        // columns 0 .. lcb
                    prefix_at(rcb, &sr, &qr);                                                        // columns rca .. rcb
                    if (rca > 0) {
                        unsigned s0, q0;
                        prefix_at(rca - 1, &s0, &q0);
                        sr -= s0;
                        qr -= q0;
                    }
                    if (l_m0 > 0) { sl += (unsigned)(l_m0 - 1) * e0; ql += (unsigned)(l_m0 - 1) * e0q; }
                    if (l_mW > 0 && lcb == W - 1) { sl += (unsigned)(l_mW - 1) * eW; ql += (unsigned)(l_mW - 1) * eWq; }
                    if (r_m0 > 0 && rca == 0) { sr += (unsigned)(r_m0 - 1) * e0; qr += (unsigned)(r_m0 - 1) * e0q; }
                    if (r_mW > 0) { sr += (unsigned)(r_mW - 1) * eW; qr += (unsigned)(r_mW - 1) * eWq; }
                    const unsigned long long ct = (unsigned long long)pad_count(0, Y0, r, H, h);
                    const unsigned long long cb2 = (unsigned long long)pad_count(Y0 + 1, Y1, r, H, h);
                    a[0] += ct * sl;  a[1] += ct * sr;  a[2] += cb2 * sl;  a[3] += cb2 * sr;
                    a[4] += ct * ql;  a[5] += ct * qr;  a[6] += cb2 * ql;  a[7] += cb2 * qr;
                }
                __builtin_amdgcn_wave_barrier();
            }
            // the wavefronts' partial sums meet in LDS (their row areas are free now), wavefront 0 adds them up
            __syncthreads();
            auto* red = reinterpret_cast<unsigned long long*>(s_row);   // [NW][8][64]
#pragma unroll
            for (int k = 0; k < 8; ++k) red[((size_t)wv * 8 + k) * kWave + lane] = a[k];
            __syncthreads();
            if (wv == 0 && has) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    unsigned long long v = 0;
                    for (int ww = 0; ww < NW; ++ww) v += red[((size_t)ww * 8 + k) * kWave + lane];
                    if (v != 0ull) atomicAdd(&acc[idx].a[k], v);
                }
                if constexpr (FINAL) {
                    __threadfence();   // this workgroup's sums are visible before its arrival is
                    if (atomicAdd(&done[idx], 1u) == gridDim.x - 1) {   // the last of the page's row chunks: the literal evaluation
                        __threadfence();
                        CornerAcc c;
#pragma unroll
                        for (int k = 0; k < 8; ++k) c.a[k] = atomicAdd(&acc[idx].a[k], 0ull);
                        double m, q2;
                        literal_mq(c, tp.f, &m, &q2);
                        const double sdev = dev_from(m, q2);
                        const PageGlobals& pg = g[wi.page];
                        const double T = threshold_literal(tp, m, sdev, (double)pg.imin, pg.coeff);
                        const unsigned p = img[(size_t)wi.y * src.step + wi.x];
                        store_decision(dst, fp.bit_out, wi.page, wi.y, wi.x, decide_literal(p, T));
                    }
                }
            }
            __syncthreads();   // (the row areas are written again by the next group)
        }
    }
}
#undef PJ

// Page minimum of the part of the page no sweep-A wavefront fetches: the sweeps stop h rows above the bottom and may
// stop short of the right border, so the last `band` rows and columns are reduced here (band = w is generous).
__global__ void __launch_bounds__(256) k_page_min_border(PageSet src, int width, int height, int band, PageGlobals* g)
{
    // work items = 16-byte pieces: the bottom `band` rows over the whole width, then the right `band` columns of the
    // rows above; a piece that would run past its region is pulled back inside (re-reading bytes does not change a minimum)
    const int page = blockIdx.y;
    const uint8_t* img = src.page(page);
    const int y0 = max(0, height - band), x0 = max(0, width - band);
    const int cpr_b = (width + 15) / 16, cpr_r = (width - x0 + 15) / 16;
    const int n_bottom = (height - y0) * cpr_b, n_right = y0 * cpr_r;
    unsigned mn = 255;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_bottom + n_right; i += gridDim.x * blockDim.x) {
        int y, x;
        if (i < n_bottom) {
            y = y0 + i / cpr_b;
            x = min((i % cpr_b) * 16, width - 16);
        } else {
            const int j = i - n_bottom;
            y = j / cpr_r;
            x = min(x0 + (j % cpr_r) * 16, width - 16);
        }
        uint4 q;
        __builtin_memcpy(&q, img + (size_t)y * src.step + x, 16);
        const unsigned d[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            mn = min(mn, min(d[k] & 0xffu, (d[k] >> 8) & 0xffu));
            mn = min(mn, min((d[k] >> 16) & 0xffu, d[k] >> 24));
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) mn = min(mn, (unsigned)__shfl_xor((int)mn, d, kWave));
    if ((threadIdx.x & (kWave - 1)) == 0 && mn < 255u) atomicMin(&g[page].imin, (int)mn);
}

template <int METHOD>
int launch_sweep(int sh, hipStream_t stream, const PageSet& src, const PageSetOut& dst, const FusedParams& fp,
                 PageGlobals* g, RefItem* rl, WorkItem* cand, unsigned* cnt)
{
    if constexpr (METHOD == PRL_SAUVOLA || METHOD == PRL_NIBLACK || METHOD == PRL_NICK || METHOD == PRL_WOLFJOLION || METHOD == PRL_FENG) {
        // windows of 33 .. 129 columns: float window rows, integer horizontal Q sums (k_fused_q); typed loads address a page with
        // 32-bit offsets
        const int n1 = fp.tp.w - 1;
        if (env_knobs().fused_qint && env_knobs().flt && !fp.flt && n1 > 30 && n1 <= 128 && !(n1 & 1) &&
            (unsigned long long)src.step * (unsigned long long)fp.tp.height < 0x7fffffffull) {
            const dim3 grid(8u * fp.xcd_waves), block(64);
            const int ef = env_knobs().fused_qint >= 2 ? 1 : 0;
            switch (sh) {
            case 0: hipLaunchKernelGGL((k_fused_q<METHOD, 0>), grid, block, 0, stream, src, dst, fp, g, rl, cand, cnt, ef); break;
            case 2: hipLaunchKernelGGL((k_fused_q<METHOD, 2>), grid, block, 0, stream, src, dst, fp, g, rl, cand, cnt, ef); break;
            case 4: hipLaunchKernelGGL((k_fused_q<METHOD, 4>), grid, block, 0, stream, src, dst, fp, g, rl, cand, cnt, ef); break;
            case 6: hipLaunchKernelGGL((k_fused_q<METHOD, 6>), grid, block, 0, stream, src, dst, fp, g, rl, cand, cnt, ef); break;
            default: return PRL_ERR_BAD_ARG;
            }
            PRL_HIP_CHECK(hipGetLastError());
            return PRL_OK;
        }
    }
    // wavefronts are independent; one per workgroup schedules best (256 x 4K pages: 4 per workgroup 4.38 ms, 2: 4.18,
    // 1: 4.13 - a finished wavefront's slot is refilled at once instead of when its whole workgroup has drained)
    unsigned wpb = 1u;
    wpb = (unsigned)env_knobs().fused_wpb;
    if (fp.total_waves > 0x7fffff00u) wpb = std::max(wpb, 4u);  // grid.x is limited to 2^31 - 1 workgroups
    const unsigned blocks = 8u * ((fp.xcd_waves + wpb - 1) / wpb);   // each XCD: its share of every tier
    const dim3 grid(blocks), block(64 * wpb);
    const bool wide = fp.tp.w - 1 > 181;  // S no longer fits the mantissa of 2^23 (eval32)
#define PRL_LAUNCH_FUSED(SHV)                                                                                    \
    do {                                                                                                         \
        if (wide)                                                                                                \
            hipLaunchKernelGGL((k_fused<METHOD, SHV, true>), grid, block, 0, stream, src, dst, fp, g, rl, cand, cnt);    \
        else                                                                                                     \
            hipLaunchKernelGGL((k_fused<METHOD, SHV, false>), grid, block, 0, stream, src, dst, fp, g, rl, cand, cnt);   \
    } while (0)
    switch (sh) {
    case 0: PRL_LAUNCH_FUSED(0); break;
    case 2: PRL_LAUNCH_FUSED(2); break;
    case 4: PRL_LAUNCH_FUSED(4); break;
    case 6: PRL_LAUNCH_FUSED(6); break;
    default: return PRL_ERR_BAD_ARG;
    }
#undef PRL_LAUNCH_FUSED
    PRL_HIP_CHECK(hipGetLastError());
    return PRL_OK;
}

// the threshold sweep of a flagged page's second chance (k_fused_exact)
template <int METHOD>
int launch_sweep_exact(int sh, hipStream_t stream, const PageSet& src, const PageSetOut& dst, const FusedParams& fp, PageGlobals* g,
                       unsigned* cnt, WorkItem* wl, CornerAcc* acc, unsigned* done)
{
    const unsigned blocks = 8u * fp.xcd_waves;
    const dim3 grid(blocks), block(64);
    const bool wide = fp.tp.w - 1 > 181;
#define PRL_LAUNCH_EXACT(SHV)                                                                                                  \
    do {                                                                                                                       \
        if (wide) hipLaunchKernelGGL((k_fused_exact<METHOD, SHV, true>), grid, block, 0, stream, src, dst, fp, g, cnt, wl, acc, done);   \
        else hipLaunchKernelGGL((k_fused_exact<METHOD, SHV, false>), grid, block, 0, stream, src, dst, fp, g, cnt, wl, acc, done);       \
    } while (0)
    switch (sh) {
    case 0: PRL_LAUNCH_EXACT(0); break;
    case 2: PRL_LAUNCH_EXACT(2); break;
    case 4: PRL_LAUNCH_EXACT(4); break;
    case 6: PRL_LAUNCH_EXACT(6); break;
    default: return PRL_ERR_BAD_ARG;
    }
#undef PRL_LAUNCH_EXACT
    PRL_HIP_CHECK(hipGetLastError());
    return PRL_OK;
}

template <int METHOD>
int launch_fused(int sh, hipStream_t stream, const PageSet& src, const PageSetOut& dst, const FusedParams& fp,
                 PageGlobals* g, RefItem* rl, WorkItem* wl, WorkItem* cand, CornerAcc* acc, unsigned* cnt,
                 hipEvent_t ev_start, hipEvent_t ev_stop, int n_pages, const GroupArrays& ga,
                 hipEvent_t before_refine = nullptr, CornerAcc* cacc = nullptr, bool exact = false)
{
    unsigned* done = reinterpret_cast<unsigned*>(fp.segmax + kSegmaxCap);  // arrivals per queued pixel (see fused_small_bytes)
    if (ev_start) PRL_HIP_CHECK(hipEventRecord(ev_start, stream));
    int st;
    if (exact) {
        // (Wolf-Jolion: the inline interval test needs k / devianceMax with its bound, which the side stream is still working out)
        if (before_refine) PRL_HIP_CHECK(hipStreamWaitEvent(stream, before_refine, 0));
        st = launch_sweep_exact<METHOD>(sh, stream, src, dst, fp, g, cnt, wl, acc, done);
    } else {
        st = launch_sweep<METHOD>(sh, stream, src, dst, fp, g, rl, cand, cnt);
    }
    if (st != PRL_OK) return st;
    if (ev_stop) PRL_HIP_CHECK(hipEventRecord(ev_stop, stream));
    if (before_refine) PRL_HIP_CHECK(hipStreamWaitEvent(stream, before_refine, 0));   // (Wolf-Jolion: the literal k / devianceMax of the side stream)
    // (float pipeline: a wavefront per queued pixel, a few microseconds each; big batches queue ~10^4 of them: 4096 wavefronts
    // instead of 1024 took k_refine from 0.14 to 0.05 ms on 256 A4 pages (Niblack w=31); small calls keep the cheaper launch)
    const unsigned refine_blocks = fp.flt ? (fp.total_waves > 20000u ? 1024u : 256u) : 64u;
    hipLaunchKernelGGL((k_refine<METHOD>), dim3(refine_blocks), dim3(256), 0, stream, src, dst, fp, g, rl, wl, cnt, acc, done);
    PRL_HIP_CHECK(hipGetLastError());
    // From kPageMajorMin queued pixels on the corner sums are built page by page (k_group_items + k_corner_rows, which return at
    // once otherwise); calls of a few pages skip the two launches (a near-empty launch costs ~5 us, a single-page call 33).
    const bool page_major = n_pages >= 8;
    // k_corner_rows: 64 bytes of a row per lane and 4 wavefronts (rows up to 4096 bytes) or 128 and 2; per wavefront the row's dwords
    // and two dword-granular prefixes with one word of padding per 16
    const bool rows_wide = fp.tp.width > 4096;
    const int rows_jw = 64 * (rows_wide ? 32 : 16), rows_nw = rows_wide ? 2 : 4;
    const size_t rows_lds = (size_t)rows_nw * 3 * (size_t)(rows_jw + (rows_jw >> 4) + 1) * sizeof(unsigned);
    const dim3 rows_grid(kRowChunks, (unsigned)std::min(n_pages, 512)), rows_block(64 * rows_nw);
    if (METHOD == PRL_WOLFJOLION) {
        // the literal devianceMax of the pages whose pixels reached the fix-up list (none, as a rule: immediate returns)
        if (page_major) {
            hipLaunchKernelGGL(k_group_items, dim3(1), dim3(1024), 0, stream, cand, cnt, 2, g, n_pages, fp.tp.width, fp.wl_cap, ga);
            if (rows_wide)
                hipLaunchKernelGGL((k_corner_rows<false, 128>), rows_grid, rows_block, rows_lds, stream, src, fp, cand, ga, cnt, cacc, dst, g,
                                   static_cast<unsigned*>(nullptr));
            else
                hipLaunchKernelGGL((k_corner_rows<false, 64>), rows_grid, rows_block, rows_lds, stream, src, fp, cand, ga, cnt, cacc, dst, g,
                                   static_cast<unsigned*>(nullptr));
        }
        hipLaunchKernelGGL(k_corner_partial<false>, dim3(kSplit, 128), dim3(256), 0, stream, src, fp, cand, cnt, 2, cacc, dst, g,
                           static_cast<unsigned*>(nullptr));
        hipLaunchKernelGGL(k_wolf_final, dim3(16), dim3(256), 0, stream, fp, g, cand, cacc, cnt);
        hipLaunchKernelGGL(k_wolf_literal_coeff, dim3((n_pages + 63) / 64), dim3(64), 0, stream, fp, g, n_pages, cnt);
        PRL_HIP_CHECK(hipGetLastError());
    }
    // literal fix-up of what k_refine queued: the kernel reads the queue length on the device and does nothing when it is
    // empty (the usual case), so no host round trip decides whether it runs; k_refine zeroed the accumulators it uses; the
    // workgroup that delivers a pixel's last partial sum evaluates the pixel (no separate k_fixup_final launch)
    if (page_major) {
        hipLaunchKernelGGL(k_group_items, dim3(1), dim3(1024), 0, stream, wl, cnt, 1, g, n_pages, fp.tp.width, fp.wl_cap, ga);
        if (rows_wide) hipLaunchKernelGGL((k_corner_rows<true, 128>), rows_grid, rows_block, rows_lds, stream, src, fp, wl, ga, cnt, acc, dst, g, done);
        else hipLaunchKernelGGL((k_corner_rows<true, 64>), rows_grid, rows_block, rows_lds, stream, src, fp, wl, ga, cnt, acc, dst, g, done);
    }
    hipLaunchKernelGGL(k_corner_partial<true>, dim3(kSplit, 16), dim3(256), 0, stream, src, fp, wl, cnt, 1, acc, dst, g, done);  // (1024 workgroups: an empty queue is the rule, and its launch should cost little)
    PRL_HIP_CHECK(hipGetLastError());
    return PRL_OK;
}

}  // namespace

// ---- host: error bounds (documented in DESIGN.md "Decision margins") ---------------------------
// Exact-arithmetic quantities: m* = f S, q* = f Q, v* = q* - m*^2, s* = sqrt(v*), T* = T(m*, s*).
//   Em, Eq   : |m_literal - m*|, |q_literal - q*| — 4 products and 3 sums of values <= f*IImax
//              (resp. f*IQmax), each rounded once in float64, plus the rounding of f itself.
//   vthr     : below this variance the literal sqrt amplifies Ev too much for a page-wide constant;
//              such pixels take the float64 interval test.  Note v* >= m*^2 (2w-1)/(w-1)^2 always
//              (the (w-1)^2 window divided by w^2), so only near-black windows get there.
//   E1       : float32 evaluation error of eval32 (u = 2^-24), first order, worst-case magnitudes
//              m <= 255, s <= 255; kappa is the relative error of s~ after the q - m^2 cancellation,
//              bounded through m*^2 <= R v*, R = (w-1)^2/(2w-1).
//   Elit     : |T_literal - T*| for v* >= vthr.
//   eps1     : 2 (E1 + Elit) + 1e-6.
struct FusedBounds {
    double Em, Eq, vthr, E1, Elit, eps1, kappa;
    double Ev, Es;   // worst-case literal noise of v and (for v* >= vthr) of s
    double rho;      // relative error bound of the float32 variance v~
};

// cq: |Q~ - Q| <= cq u Q for the float32 value of the window sum of squares (1 = one conversion of the exact integer;
// the float32 pipeline's chain of lane sums is looser, see flt_cq)
static FusedBounds fused_bounds(const ThrParams& tp, double cq = 1.0)
{
    FusedBounds b{};
    const double u = std::ldexp(1.0, -24), e64 = std::ldexp(1.0, -53);
    const double n1 = tp.w - 1.0, R = n1 * n1 / (2.0 * tp.w - 1.0);
    const double M = 255.0, SM = 255.0;
    const double IImax = 255.0 * tp.pw * (double)tp.ph, IQmax = 65025.0 * tp.pw * (double)tp.ph;
    b.Em = 16.0 * e64 * tp.f * IImax;
    b.Eq = 16.0 * e64 * tp.f * IQmax;
    const double Ev = b.Eq + 2.0 * M * b.Em + b.Em * b.Em;
    b.vthr = std::fmax(1e-2, 64.0 * Ev);
    const double Es = 1.01 * Ev / std::sqrt(b.vthr - Ev);
    // K~ = fma(w^2, Q~, -S^2~): |K~ - K| <= u (cq w^2 Q + S^2) + u K, i.e. relative to v* = f^2 K at most
    // rho = (cq + 1)(1 + R) u; sqrt halves it and adds 2u (v_sqrt_f32 is 1 ulp): kappa = ((cq + 1)(1 + R)/2 + 2) u
    // (cq = 1: rho = (2 + 2R) u, kappa = (3 + R) u).
    b.kappa = ((cq + 1.0) * (1.0 + R) * 0.5 + 2.0) * u * 1.1;
    b.Ev = Ev;
    b.Es = Es;
    b.rho = (cq + 1.0) * (1.0 + R) * u * 1.1;
    const double k = std::fabs(tp.k);
    switch (tp.method) {
    case PRL_SAUVOLA: {
        const double a = std::fabs(tp.a), bb = std::fabs(tp.b), Dmax = a * SM + bb;
        b.E1 = M * (a * SM * (b.kappa + u) + bb * u + u * Dmax) + u * (256 + M * Dmax);
        b.Elit = M * a * Es + Dmax * b.Em;
        break;
    }
    case PRL_NIBLACK:
        b.E1 = k * SM * (b.kappa + u) + M * u + 2 * u * (256 + M + k * SM);
        b.Elit = k * Es + b.Em;
        break;
    case PRL_NICK: {
        const double Ec = 1.01 * (b.Eq + 1e-10) / std::sqrt(b.vthr - b.Eq - 1e-10);
        b.E1 = k * SM * (0.5 * cq + 3.0) * u + M * u + 2 * u * (256 + M + k * SM);  // sqrt(Q~): cq u / 2 + 2u, + 1u product
        b.Elit = b.Em + k * Ec;
        break;
    }
    case PRL_WOLFJOLION:
        // T = m + (s c - k)(m - Imin), |c| s <= |k| (1 + tiny) because s <= max(s); the part of Elit that
        // scales with |c| = |k / devianceMax| is added per page in the kernel (FusedParams::es_max).
        b.E1 = M * k * (b.kappa + 14 * u) + 1276 * u;
        b.Elit = (2.0 + 2.0 * k) * b.Em;
        break;
    case PRL_FENG: {
        const double gcoef = std::fabs(1.0 + tp.c1), c3 = 255.0 * (std::fabs(tp.k2) + 1.0);
        b.E1 = gcoef * M * u + 2 * u * (256 + gcoef * M + c3) + c3 * u;
        b.Elit = gcoef * b.Em + 1e-10;
        break;
    }
    default:
        b.E1 = b.Elit = 1e30;
    }
    b.eps1 = 2.0 * (b.E1 + b.Elit) + 1e-6;
    return b;
}

extern "C" int prl_hip_internal_fused_bounds(const prl_binarize_params* p, int width, int height, double* out8)
{
    // test hook (not in the public header): the decision margins for (params, page size)
    prl_binarize_geometry g;
    int st = prl_hip_binarize_geometry(p, width, height, &g);
    if (st != PRL_OK) return st;
    ThrParams tp{};
    tp.method = p->method;
    tp.w = g.w;
    tp.pw = g.padded_w;
    tp.ph = g.padded_h;
    tp.f = 1.0 / (double)(g.w * g.w);
    tp.k = p->k;
    tp.a = p->k * (1.0 / 128.0);
    tp.b = 1.0 - p->k;
    tp.c1 = 1.0 - p->feng_alpha1;
    tp.k2 = p->feng_k2;
    const FusedBounds b = fused_bounds(tp);
    out8[0] = b.Em; out8[1] = b.Eq; out8[2] = b.vthr; out8[3] = b.E1;
    out8[4] = b.Elit; out8[5] = b.eps1; out8[6] = b.kappa; out8[7] = 0;
    return PRL_OK;
}

bool fused_supports(const ThrParams& tp)
{
    if (((tp.w - 1) & 1) != 0) return false;                 // even (clamped) window: rare, literal
    if (tp.w - 1 > 256 || tp.w < 3) return false;            // window sums must stay exact in u32 / float32
    if (tp.width < 16) return false;                         // the 8-byte row fetch needs a row to clamp into
    if (!std::isfinite(tp.k) || std::fabs(tp.k) > 1e3) return false;
    if (tp.method == PRL_FENG && !(tp.gamma > 0.0)) return false;
    if (tp.method == PRL_FENG && (!std::isfinite(tp.c1) || !std::isfinite(tp.k2) ||
                                  std::fabs(tp.c1) > 1e3 || std::fabs(tp.k2) > 1e3)) return false;
    return true;
}

// The float32 pipeline (strip_loop_f): is it usable for this call, and how loose is its Q~?  Sums of j+1 lane totals
// (the W chain) and the final window sum round once each when they reach 2^24 (half an ulp of their largest
// possible value); any such partial sum covers the window plus at most 8 columns outside it, so a rounding can
// only happen when Q >= 2^24 - 8 (w-1) 65025 =: Qmin, which turns the absolute bound into a relative one.
static bool flt_usable(const ThrParams& tp, size_t src_step, double* cq, double* delta_out = nullptr, double* qmin_out = nullptr)
{
    if (!env_knobs().flt) return false;
    const int n1 = tp.w - 1;
    if (n1 > 30) return false;
    if ((unsigned long long)src_step * (unsigned long long)tp.height >= 0x7fffffffull) return false;  // 32-bit buffer offsets
    auto half_ulp = [](double x) { return x < 16777216.0 ? 0.0 : std::ldexp(1.0, (int)std::floor(std::log2(x)) - 24); };
    const double col = (double)n1 * 65025.0;  // largest column sum of squares
    double delta = half_ulp((double)n1 * col);  // the final sum
    for (int j = 1; j <= n1 / 8; ++j) delta += half_ulp((j + 1) * 8.0 * col);
    const double qmin = 16777216.0 - 8.0 * col;
    if (delta > 0.0 && !(qmin > 0.0)) return false;
    *cq = std::fmax(1.0, delta > 0.0 ? delta / (qmin * std::ldexp(1.0, -24)) : 1.0);
    if (delta_out) *delta_out = delta;
    if (qmin_out) *qmin_out = qmin;
    return true;
}

// Wolf-Jolion's sweep A with windows wider than 31 on the float32 loop (LO == 4 form).  There is no decision in sweep A, only
// K~ = fma(w^2, Q~, -S^2) for the variance maximum, so an ABSOLUTE bound on |Q~ - Q| is enough: it widens the band in which
// sweep B looks for the exact maximum and the margin of the threshold sweep's estimated coefficient.  S: every partial sum
// covers at most 17 lanes x 8 columns x (w-1) rows of 255 < 2^24 for w - 1 <= 128: exact.  Q: each addition rounds by at most
// half an ulp of the largest value its result can take; the bound below follows the additions of the loop one by one.
static bool flt_a_usable(const ThrParams& tp, size_t src_step, double* dq_out)
{
    if (!env_knobs().flt) return false;
    const int n1 = tp.w - 1;
    if (n1 <= 30 || n1 > 128 || (n1 & 1)) return false;
    if ((unsigned long long)src_step * (unsigned long long)tp.height >= 0x7fffffffull) return false;  // 32-bit buffer offsets
    auto half_ulp = [](double x) { return x < 16777216.0 ? 0.0 : std::ldexp(1.0, (int)std::floor(std::log2(x)) - 24); };
    const double col = (double)n1 * 65025.0;   // largest column sum of squares (exact: < 2^24)
    double e_tot = 0.0;                          // error of an in-lane prefix / of a lane total
    for (int j = 2; j <= 8; ++j) e_tot += half_ulp(j * col);
    const double T = 8.0 * col;
    const int L = n1 / 8;
    double e2 = 2 * e_tot + half_ulp(2 * T), e4 = 2 * e2 + half_ulp(4 * T), e8 = 2 * e4 + half_ulp(8 * T), e16 = 2 * e8 + half_ulp(16 * T);
    double ew = 0.0;   // error of w1 (w0 is one piece less)
    int lanes = 0;
    for (int bit = 16; bit >= 1; bit >>= 1)
        if (L & bit) {
            ew += bit == 16 ? e16 : bit == 8 ? e8 : bit == 4 ? e4 : bit == 2 ? e2 : e_tot;
            lanes += bit;
            ew += half_ulp(lanes * T);
        }
    ew += e_tot + half_ulp((lanes + 1) * T);
    // Q = (E_far - E_own) + W
    const double dq = 2 * e_tot + half_ulp(T) + ew + half_ulp((lanes + 2) * T);
    *dq_out = dq;
    return true;
}

extern "C" int prl_hip_internal_flt_a_q_error(int w, double* dq)
{
    // test hook (not in the public header): flt_a_usable()'s absolute bound on |Q~ - Q| for sweep A's doubling form; 0 when unused for w
    ThrParams tp{};
    tp.w = w;
    tp.height = 1;
    return flt_a_usable(tp, 1, dq) ? 1 : 0;
}

extern "C" int prl_hip_internal_flt_q_error(int w, double* delta_qmin_cq)
{
    // test hook (not in the public header): the float32 pipeline's bound on |Q~ - Q| for window w -
    // [0] = delta (absolute), [1] = Qmin (no rounding below it), [2] = cq; returns 0 when the pipeline is not used for w
    ThrParams tp{};
    tp.w = w;
    tp.height = 1;
    return flt_usable(tp, 1, &delta_qmin_cq[2], &delta_qmin_cq[0], &delta_qmin_cq[1]) ? 1 : 0;
}

size_t fused_small_bytes(int n_pages)
{
    // [counters][refine list: kRefBuckets x kRefBucketCap][fix-up list][Wolf candidate list][corner sums][Wolf per-wavefront maxima][arrival counters]
    // [... arrival counters][corner sums of Wolf-Jolion's candidates]
    // [... corner sums of Wolf-Jolion's candidates][k_group_items: item indices by page, page starts / cursors / list]
    return kFusedCounterBytes + sizeof(RefItem) * (size_t)kRefineCap + 2 * sizeof(WorkItem) * (size_t)kWorkCap +
           sizeof(CornerAcc) * (size_t)kWorkCap + sizeof(float) * kSegmaxCap + sizeof(unsigned) * (size_t)kWorkCap +
           sizeof(CornerAcc) * (size_t)kWorkCap + sizeof(unsigned) * ((size_t)kWorkCap + 3 * (size_t)std::max(n_pages, 1) + 4);
}

// Strips of a row: uo output columns each, fetched with w - 1 halo columns; uo = 512 - (w - 1) rounded down to a multiple of 8,
// which keeps a lane's 8 outputs whole (8-byte mask stores, one byte of the bit plane).  Two ways to save the last strip:
//  * ragged uo (float32 pipeline, byte masks, not Wolf-Jolion whose sweeps count on whole lanes): the unrounded 512 - (w - 1)
//    when that takes a strip less - the last lane of every strip then stores 2, 4 or 6 bytes.  A4 rows at NICK's default
//    w = 21 (2459 outputs): 5 strips of 492 instead of 6 of 488.
//  * extended last strip (wide windows, the wavefront-scan form of the horizontal sums): when everything from the last lane
//    of what would be the last but one strip onwards is right-hand padding (copies of the border column), that strip takes the
//    rest of the row as well: the totals of the lanes beyond the wavefront are known without fetching them (strip_loop, `ext`).
//    Up to 512 - h - 7 outputs instead of 512 - (w - 1); an A4 row (2479 outputs) at the default w = 101 takes 6 strips instead
//    of 7, a 4096-column row 10 instead of 11.
static int strip_layout(const ThrParams& tp, bool flt, bool bit_out, int* uo_out, int* ext)
{
    int uo = ((SW - (tp.w - 1)) / 8) * 8;
    *ext = 0;
    *uo_out = uo;
    if (uo <= 0) return 0;
    int n = (tp.ow + uo - 1) / uo;
    if (flt && !bit_out && tp.method != PRL_WOLFJOLION && env_knobs().ragged_uo) {
        const int uo_r = SW - (tp.w - 1), n_r = (tp.ow + uo_r - 1) / uo_r;
        if (n_r < n) {
            n = n_r;
            uo = *uo_out = uo_r;
        }
    }
    // (the extended strip multiplies the border column's sums with __umul24: both factors must stay below 2^24 - true for every
    // window fused_supports() admits today (256 * 65025 < 2^24), and tied to it here should that limit ever move)
    if ((long long)(tp.w - 1) * 65025ll >= (1ll << 24)) return n;
    if (!env_knobs().ext_strip || (tp.w - 1) / 8 < 5 || n < 2) return n;
    const int xs = (n - 2) * uo;  // first output column of the strip that would take over
    if (xs + 1 - tp.half + 8 * 63 >= tp.width - 1 && tp.ow - xs <= SW) {
        *ext = 1;
        --n;
    }
    return n;
}

extern "C" int prl_hip_internal_strip_layout(int method, int w, int width, int ow, int bit_out, int* uo_ext)
{
    // test hook (not in the public header): strips per row; uo_ext[0] = outputs per strip, [1] = the last strip is an extended one
    ThrParams tp{};
    tp.method = method;
    tp.w = w;
    tp.half = w / 2;
    tp.width = width;
    tp.height = 1;
    tp.ow = ow;
    double cq;
    return strip_layout(tp, flt_usable(tp, (size_t)width, &cq), bit_out != 0, &uo_ext[0], &uo_ext[1]);
}

// Pages one fused_run call may take.  Wolf-Jolion keeps one float per wavefront of the call (sweep A -> sweep B), kSegmaxCap
// of them: a call's wavefronts = pages x strips x segments; fused_run lengthens its segments until they fit, this only keeps
// the segments of a chunk from becoming much longer than 128 rows.
int fused_max_pages(const ThrParams& tp)
{
    if (tp.method != PRL_WOLFJOLION) return 0x7fffffff;
    const int uo = ((SW - (tp.w - 1)) / 8) * 8;
    if (uo <= 0) return 0x7fffffff;  // not a fused configuration
    const long long n_strips = (tp.ow + uo - 1) / uo, n_segs = (tp.oh + 127) / 128;
    return (int)std::max<long long>(1, (long long)env_knobs().segmax_cap / (n_strips * n_segs));
}

// The whole pipeline of one call: threshold sweep, k_refine, literal fix-up (the last two find their queues on the device).
int fused_run(const ThrParams& tp, const PageSet& src, int n_pages, const PageSetOut& dst, void* small,
              PageGlobals* d_globals, hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop, bool bit_out,
              bool counters_zeroed, PageGlobals* host_globals, const WolfSide* wolf_side, bool exact)
{
    FusedParams fp{};
    fp.tp = tp;
    fp.bit_out = bit_out ? 1 : 0;
    // non-temporal mask stores: the output stream is never re-read, and keeping it out of L2 leaves the cache to the
    // window rows that ARE re-read (the leaving row, the compared-pixel row): 4.31-4.47 -> 4.15 ms on 256 x 4K pages
    fp.nt_store = env_knobs().nt_store ? 1 : 0;
    double cq = 1.0, dq = 0.0;
    fp.flt = !exact && flt_usable(tp, src.step, &cq, &dq) ? 1 : 0;   // (exact: the integer loop for every strip - exact sums)
    fp.flt_dq = fp.flt ? dq : 0.0;
    fp.n_strips = strip_layout(tp, fp.flt != 0, bit_out, &fp.uo, &fp.ext);
    // Row segments.  Long segments amortise the (w-1)-row warm-up, short ones fill the chip and keep the tail short when it
    // drains; workgroups start in index order, so the segments come in TIERS of decreasing length (guided scheduling): each tier
    // takes about half of the rows that are left, in segments sized for ~two rounds of the chip's wavefront slots, down to a
    // floor that keeps a segment's warm-up (a warm-up row costs 0.1-0.25 of a full one) a fraction of its own work.
    // Measured on MI355X (profiles/r03/tiers.txt); before (one size for all, 128 rows): 4K pages, w=31, 256 pages - 64..192 rows
    // within 1 %, 512: +14 % (tail); A4, w=101: 128..512 rows within 3 % (what the longer segments save in warm-up rows the
    // tail gives back, profiles/r03/rps_w101.txt).  Small batches (the chip is never full): one size, halved while far from full:
    // 32 pages - 128: 0.54 ms, 64: 0.48, 32: 0.49, 16: 0.55; 8 pages - 0.168 / 0.125 / 0.127 / 0.131; one page - 0.120 / 0.066 /
    // 0.044 / 0.031 (a single page has 9 x 32 wavefronts at 128 rows for 5120 wavefront slots)
    const long long PS = (long long)n_pages * fp.n_strips;   // page-strips
    const long long slots = 5120;                             // 256 CUs x 20 wavefronts
    auto waves_at = [&](int r) { return PS * ((tp.oh + r - 1) / r); };
    // (shortest segment of a small batch: one A4 page, Niblack w=101 - 32 rows 0.067 ms, 16 rows 0.056, 8 rows 0.054; one 4096^2
    // page, w=15 - 16 / 8 / 4 rows all 0.030-0.033 ms)
    int min_rps = 8;
    while (min_rps < (tp.w - 1) / 8) min_rps *= 2;
    int floor_rps = 32;                                       // tiers: shortest segment = pow2ceil(w - 1) in [32, 128]
    while (floor_rps < tp.w - 1 && floor_rps < 128) floor_rps *= 2;
    floor_rps = std::max(floor_rps, min_rps);
    const unsigned long long wave_cap = tp.method == PRL_WOLFJOLION ? env_knobs().segmax_cap : 0xfffffff0ull;
    auto single_tier = [&](int rps) {
        fp.n_tiers = 1;
        fp.tier[0].rows = rps; fp.tier[0].segs = (tp.oh + rps - 1) / rps; fp.tier[0].row0 = 0;
    };
    // (tiers pay from ~600 rows of work per wavefront slot on: 256 x 4K pages -2 %, 256 A4 pages w=21 -3.5 %, 1024 A4 pages -4 %;
    // 64 x 4K pages -2 %, 32 x 4K pages +3 % - there the single size below, tuned for small batches, stays)
    const bool tiers_on = env_knobs().tiers && !env_knobs().rows_per_seg && PS * tp.oh >= 600 * slots;
    if (!tiers_on) {
        int rps = 128;
        if (waves_at(128) < 40000 && tp.w - 1 <= 64) rps = 64;   // (64 A4 pages, w=101: 128 rows 0.78 ms, 64 rows 0.80)
        // small batches: halve while the chip is far from full (wide windows stop earlier, their warm-up rows weigh more:
        // 4 A4 pages, w=101 - 32 rows (2630 wavefronts) 0.078 ms, 16 rows 0.091; 2 pages - 0.082 / 0.070, profiles/r03/small_batches_floor8.txt)
        const long long fill = tp.w - 1 > 64 ? 2048 : 4096;
        while (rps > min_rps && waves_at(rps) < fill) rps /= 2;
        // (32 x 4K pages: 40 .. 72 rows per segment all measure 0.445-0.478 ms, whole or fractional rounds of the chip's wavefront
        // slots alike - profiles/r03/strong_proxy.txt; a wavefront lives ~160 us of the kernel's 450: ramp-up and drain, not the tail)
        if (env_knobs().rows_per_seg) rps = env_knobs().rows_per_seg;  // tuning knob
        while ((unsigned long long)waves_at(rps) > wave_cap && rps < tp.oh) rps *= 2;  // (Wolf: one sweep-A maximum per wavefront)
        single_tier(rps);
    } else {
        for (int fl = floor_rps;; fl *= 2) {
            // (Wolf-Jolion: sweep B revisits the few segments that hold a candidate for the variance maximum, one wavefront
            // each, and lasts as long as its longest segment takes a lone wavefront - shorter first tiers there: 256 A4 pages,
            // w=31: sweep B 295 -> 99 us, the call 3.97 -> 3.82 ms; with wide windows the extra warm-up rows of the two other
            // sweeps cost what sweep B saves, so the cap stays two floors up - profiles/r03/wolf_tier_max.txt)
            int n = 0, rows_left = tp.oh, row0 = 0;
            int prev = std::max(fl, tp.method == PRL_WOLFJOLION ? std::max(env_knobs().wolf_tier_max, 2 * floor_rps) : 512);
            constexpr int kMaxTiers = (int)(sizeof(fp.tier) / sizeof(fp.tier[0]));
            while (rows_left > 0) {
                const double want = (double)PS * rows_left / (2.0 * (double)slots);
                int R = fl;
                while (R * 2 <= want && R * 2 <= prev) R *= 2;
                R = std::min(R, std::max(prev, fl));
                int segs;
                if (R <= fl || n == kMaxTiers - 1 || rows_left <= R) segs = (rows_left + R - 1) / R;   // last tier: all that is left
                else segs = std::max(1, rows_left / R / 2);
                if (n > 0 && fp.tier[n - 1].rows == R) fp.tier[n - 1].segs += segs;   // same length as the tier before: one tier
                else { fp.tier[n].rows = R; fp.tier[n].segs = segs; fp.tier[n].row0 = row0; ++n; }
                row0 += segs * R;
                rows_left -= segs * R;
                prev = R;
            }
            fp.n_tiers = n;
            unsigned long long tw = 0;
            for (int k = 0; k < n; ++k) tw += (unsigned long long)PS * fp.tier[k].segs;
            if (tw <= wave_cap || fl >= tp.oh) break;
        }
    }
    {
        unsigned long long tw = 0, xw = 0;
        for (int k = 0; k < fp.n_tiers; ++k) {
            const unsigned long long nw = (unsigned long long)PS * fp.tier[k].segs;
            if (nw > 0xfffffff0ull) return PRL_ERR_BAD_ARG;
            fp.tier[k].waves = (unsigned)nw;
            fp.tier[k].first = (unsigned)tw;
            tw += nw;
            xw += (nw + 7) / 8;
        }
        if (tw > 0xfffffff0ull || xw > 0x0ffffff0ull) return PRL_ERR_BAD_ARG;
        fp.total_waves = (unsigned)tw;
        fp.xcd_waves = (unsigned)xw;
    }
    fp.lane_off = (tp.w - 1) / 8;
    const FusedBounds b = fused_bounds(tp, fp.flt ? cq : 1.0);  // margins of the threshold sweep
    const FusedBounds b1 = fused_bounds(tp);                    // integer-pipeline margins (Wolf-Jolion's sweep B, literal noise terms)
    const double Z = (double)kZ, f = tp.f;
    fp.w2f = (float)(tp.w * tp.w);
    fp.Em = b.Em;
    fp.Eq = b.Eq;
    fp.vthr = b.vthr;
    fp.vthr32 = (float)(b.vthr / (f * f) * (1.0 + 2.0 * b.rho + 1e-5));
    if (tp.method == PRL_NICK) {   // a floor on Q~ instead (eval32f): Q > (1 + R) vthr / f  =>  v* >= q* / (1 + R) > vthr
        const double n1 = tp.w - 1.0, R = n1 * n1 / (2.0 * tp.w - 1.0);
        fp.vthr32 = (float)((1.0 + R) * b.vthr / f * (1.0 + 2.0 * (fp.flt ? cq : 1.0) * std::ldexp(1.0, -24) + 1e-5));
    }
    fp.eps1 = (float)(b.eps1 * 1.01 * Z);
    fp.ref_cap = kRefineCap;
    fp.wl_cap = kWorkCap;
    // a page's share of the candidate list: the whole list for a single page, never less than a quarter of it in a batch
    // (real scans queue ~400 candidates a page, profiles/r04/real_scans.jsonl; a flat page would queue every pixel)
    // (a page's share of the candidate list: a quarter for small calls; for batches an n-th, but never below what the reference's
    // own scans need (~10^4) - four flat pages of a big batch no longer fill the list for every document page behind them)
    fp.cand_page_cap = n_pages <= 1 ? kWorkCap : std::max(kWorkCap / (unsigned)std::max(4, n_pages), std::min(kWorkCap / 4u, 10000u));
    // need_p0 = 0 only where T > -0.5 for every window, so a black pixel can never come out white in the float32 sign test (the
    // literal clamps T8 at 0).  Beyond T >= 0 two families qualify: a small negative k of Niblack / NICK - s <= sqrt(q) and
    // q <= 255 m (Q = sum P^2 <= 255 sum P), so T >= m - |k| sqrt(255 m) >= -k^2 255 / 4, above -0.45 for |k| < 0.084 (NICK's header
    // default -0.01) - and Feng when (1 + (1 - alpha1)) r + k2 >= 1 with r = ((w-1)/w)^2: m >= r Imin, so T = g m + (k2 - 1) Imin >= 0.
    const bool small_neg_k = tp.k * tp.k * 63.75 < 0.45;
    switch (tp.method) {
    case PRL_SAUVOLA:
        fp.c0 = (float)(Z * tp.a * f * f);
        fp.c1 = (float)(Z * tp.b * f);
        fp.need_p0 = !(tp.a >= 0.0 && tp.b >= 0.0);
        break;
    case PRL_NIBLACK: fp.c0 = (float)(Z * tp.k * f); fp.c1 = (float)(Z * f); fp.need_p0 = !(tp.k >= 0.0 || small_neg_k); break;
    case PRL_NICK: fp.c0 = (float)(Z * tp.k * std::sqrt(f)); fp.c1 = (float)(Z * f); fp.need_p0 = !(tp.k >= 0.0 || small_neg_k); break;
    case PRL_FENG: {
        const double g1 = 1.0 + tp.c1, r = (tp.w - 1.0) * (tp.w - 1.0) * f;
        fp.c0 = (float)(Z * g1 * f);
        fp.need_p0 = !(g1 >= 0.0 && (g1 * r + tp.k2 - 1.0 >= 1e-6 || tp.k2 >= 1.0));
        break;
    }
    case PRL_WOLFJOLION:
        fp.c0 = (float)tp.k;
        fp.c1 = (float)(Z * f);
        // T = m + (s c - k)(m - Imin) with s c in [0, k (1 + 1e-5)]: for 0 <= k <= 1 that is >= (1 - k) m + k Imin - 3e-3 > -0.5, so a
        // black pixel can never come out white (m < Imin only by the (2w-1)/w^2 of the short window, where both factors are negative)
        fp.need_p0 = !(tp.k >= 0.0 && tp.k <= 1.0);
        break;
    default: return PRL_ERR_BAD_ARG;
    }
    fp.es_max = (float)(b.Es * 1.01);
    fp.rho = (float)(std::fmax(b1.rho, b.rho) * 1.01);  // sweep A runs the float32 pipeline on interior strips, sweep B the integer one
    fp.ev2 = (float)(2.0 * b1.Ev * 1.01 / (f * f));  // in K units
    {
        double dqa = 0.0;
        fp.flt_a = (tp.method == PRL_WOLFJOLION && !fp.flt && flt_a_usable(tp, src.step, &dqa)) ? 1 : 0;
        // |K~ - K| <= w^2 |Q~ - Q| (+ the relative part, rho: S^2 and the fma round as in the integer pipeline's conversion)
        fp.kabs = fp.flt_a ? (float)((double)(tp.w * tp.w) * dqa * 1.01) : 0.0f;
    }

    // [1] fix-up-list length, [2] Wolf candidate-list length, [5] pages of the page-major corner kernel, [60] epilogue arrivals;
    // words 64 ...: the refine queue's bucket counters (kRefCounterStride apart)
    auto* cnt = static_cast<unsigned*>(small);
    if (host_globals) {
        fp.ep_host = host_globals;
        fp.ep_dev = d_globals;
        fp.ep_counters = cnt;
        fp.ep_pages = n_pages;
    }
    auto* rl = reinterpret_cast<RefItem*>(static_cast<uint8_t*>(small) + kFusedCounterBytes);
    auto* wl = reinterpret_cast<WorkItem*>(static_cast<uint8_t*>(small) + kFusedCounterBytes + sizeof(RefItem) * (size_t)kRefineCap);
    auto* cand = wl + kWorkCap;
    auto* acc = reinterpret_cast<CornerAcc*>(cand + kWorkCap);
    fp.segmax = reinterpret_cast<float*>(acc + kWorkCap);
    auto* cacc = reinterpret_cast<CornerAcc*>(reinterpret_cast<unsigned*>(fp.segmax + kSegmaxCap) + kWorkCap);   // (behind the arrival counters)
    GroupArrays ga;
    ga.sidx = reinterpret_cast<unsigned*>(cacc + kWorkCap);
    ga.pstart = ga.sidx + kWorkCap;
    ga.pcur = ga.pstart + n_pages + 1;
    ga.plist = ga.pcur + n_pages;
    if (!counters_zeroed) PRL_HIP_CHECK(hipMemsetAsync(cnt, 0, kFusedCounterBytes, stream));

    if (tp.method == PRL_FENG) {
        int st = page_min_run(tp, src, n_pages, d_globals, stream);
        if (st != PRL_OK) return st;
    }
    const int sh = (tp.w - 1) & 7;
    if (tp.method == PRL_WOLFJOLION) {
        // devianceMax first (binarizeWolfJolion.cpp:118-121): sweep A finds the float32 variance maximum,
        // sweep B revisits only the wavefront segments that can hold the literal maximum and queues their
        // candidate pixels, k_wolf_exact evaluates those literally, k_wolf_coeff forms k / devianceMax.
        if (fp.total_waves > env_knobs().segmax_cap) {  // (unreachable through the C ABI: its page chunks follow fused_max_pages)
            set_error_detail("Wolf-Jolion: more wavefronts in one call than per-wavefront maxima slots");
            return PRL_ERR_BAD_ARG;
        }
        // (cv::minMaxLoc(imageInput) rides on sweep A - every window row a wavefront fetches goes into a running
        // minimum - plus a small kernel for the bottom rows / right columns the sweeps never fetch; Feng, which has no
        // sweep, uses k_page_min)
        // Schedule (round 4).  On the caller's stream: sweep A (float32 variance maximum per page and per wavefront), the
        // threshold sweep - its coefficient comes from sweep A's maximum with a margin for the difference (k_fused) -, k_refine
        // and the fix-up.  On the workspace's side stream, beside the two sweeps: the page minimum of the border bands, sweep B
        // (revisits the wavefronts that can hold the maximum: EXACT integer maximum of K, and the candidate pixels) and
        // k_wolf_interval (k / devianceMax with a bound of a few 10^-9 on its distance from the literal one), for which
        // k_refine waits.  The literal devianceMax itself - absolute integral corners of every candidate, the part that cost
        // 0.2-13 ms - is only computed when a pixel reaches the literal fix-up, and then for that page only (k_corner_partial
        // <false>, k_wolf_final, k_wolf_literal_coeff: they return at once otherwise).  256 A4 pages at the header defaults:
        // profiles/r04/wolf_schedule_ab.txt.  Without a side stream (PRL_HIP_WOLF_SIDE=0, hooks build): the same kernels in
        // order on one stream.
        hipStream_t ss = wolf_side ? wolf_side->stream : stream;
        auto border_min = [&](hipStream_t q) -> int {
            const int band = std::min(std::max(tp.w + 8, 16), std::max(tp.width, tp.height));
            for (int first = 0; first < n_pages; first += 32768) {  // grid.y limit
                PageSet part = src;
                if (part.table) part.table += first; else part.base += (size_t)first * part.page_stride;
                hipLaunchKernelGGL(k_page_min_border, dim3(16, std::min(32768, n_pages - first)), dim3(256), 0, q, part,
                                   tp.width, tp.height, band, d_globals + first);
            }
            PRL_HIP_CHECK(hipGetLastError());
            return PRL_OK;
        };
        // Between the fork and the join the side stream reads the caller's pages: an error return in that span first waits
        // for it (the API must not hand an error back while its kernels still run on the caller's memory).
        auto forked = [&]() -> int {
            int st;
            if (wolf_side) {
                PRL_HIP_CHECK(hipEventRecord(wolf_side->ev_fork, stream));          // (globals and counters are initialised)
                PRL_HIP_CHECK(hipStreamWaitEvent(ss, wolf_side->ev_fork, 0));
            }
            PRL_HIP_CHECK(hipMemsetAsync(cacc, 0, sizeof(CornerAcc) * (size_t)kWorkCap, ss));   // (only the lazy literal path uses it)
            st = border_min(ss);
            if (st != PRL_OK) return st;
            if (wolf_side) PRL_HIP_CHECK(hipEventRecord(wolf_side->ev_min, ss));
            st = launch_sweep<kWolfMax>(sh, stream, src, dst, fp, d_globals, rl, cand, cnt);
            if (st != PRL_OK) return st;
            if (wolf_side) {
                PRL_HIP_CHECK(hipEventRecord(wolf_side->ev_a, stream));
                PRL_HIP_CHECK(hipStreamWaitEvent(ss, wolf_side->ev_a, 0));
            }
            st = launch_sweep<kWolfCollect>(sh, ss, src, dst, fp, d_globals, rl, cand, cnt);
            if (st != PRL_OK) return st;
            hipLaunchKernelGGL(k_wolf_interval, dim3((n_pages + 63) / 64), dim3(64), 0, ss, fp, d_globals, n_pages);
            PRL_HIP_CHECK(hipGetLastError());
            if (env_knobs().debug) {
                unsigned hc[4] = {0, 0, 0, 0};
                (void)hipMemcpyAsync(hc, cnt, sizeof(hc), hipMemcpyDeviceToHost, ss);
                (void)hipStreamSynchronize(ss);
                std::fprintf(stderr, "[prl_hip] Wolf-Jolion: %u maximum-deviation candidates on %d pages (cap %u)\n", hc[2], n_pages, fp.wl_cap);
            }
            hipEvent_t before_refine = nullptr;
            if (wolf_side) {
                PRL_HIP_CHECK(hipEventRecord(wolf_side->ev_coeff, ss));
                PRL_HIP_CHECK(hipStreamWaitEvent(stream, wolf_side->ev_min, 0));   // the threshold sweep needs the whole page minimum
                before_refine = wolf_side->ev_coeff;
            }
            return launch_fused<PRL_WOLFJOLION>(sh, stream, src, dst, fp, d_globals, rl, wl, cand, acc, cnt, ev_start, ev_stop, n_pages, ga,
                                                before_refine, cacc, exact);
        };
        const int st = forked();
        if (st != PRL_OK && wolf_side) (void)hipStreamSynchronize(wolf_side->stream);
        return st;
    }
    switch (tp.method) {
    case PRL_SAUVOLA: return launch_fused<PRL_SAUVOLA>(sh, stream, src, dst, fp, d_globals, rl, wl, cand, acc, cnt, ev_start, ev_stop, n_pages, ga, nullptr, nullptr, exact);
    case PRL_NIBLACK: return launch_fused<PRL_NIBLACK>(sh, stream, src, dst, fp, d_globals, rl, wl, cand, acc, cnt, ev_start, ev_stop, n_pages, ga, nullptr, nullptr, exact);
    case PRL_NICK: return launch_fused<PRL_NICK>(sh, stream, src, dst, fp, d_globals, rl, wl, cand, acc, cnt, ev_start, ev_stop, n_pages, ga, nullptr, nullptr, exact);
    case PRL_FENG: return launch_fused<PRL_FENG>(sh, stream, src, dst, fp, d_globals, rl, wl, cand, acc, cnt, ev_start, ev_stop, n_pages, ga, nullptr, nullptr, exact);
    default: return PRL_ERR_BAD_ARG;
    }
}

}  // namespace prl_hip
