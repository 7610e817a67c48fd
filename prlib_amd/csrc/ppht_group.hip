// ppht_group.hip — cv::HoughLinesP's progressive probabilistic Hough transform with the accumulator ON CHIP: a group of G
// workgroups (one per CU) per page, each holding the cells of its share of the 180 angles in LDS.
//
// Reference: src/deskew/deskew.cpp:148 (cv::HoughLinesP(input, lines, 1, CV_PI/180, 100, width/8.f, 20)); OpenCV's
// HoughLinesProbabilistic [upstream] as restated in the test oracle (oracle/, deskew file).  Same segments, in the same order, as
// k_ppht / k_ppht_mw (deskew.hip), which keep the accumulator in device memory and are bound by the rate of scattered
// read-modify-writes one CU gets through its miss path (0.6 us per point: DESIGN.md 4.4).
//
// What makes the transform sequential is only this: a point votes into the cells the earlier points left, and when a cell reaches
// the threshold a line is walked that erases points and takes their votes back.  Everything else is laid out so that a group of
// workgroups can replay it side by side and has to talk ONCE per block of points:
//   * the visiting order does not depend on the data (cv::RNG draws idx = next() % count and the list swap nz[idx] = nz[count-1]
//     happens whatever the point does).  k_order_link / k_order_resolve compute it for the whole page in parallel before the
//     transform starts: per list position the steps that write it form a linked list (atomicExch), and the point visited at step t
//     is found by following "who wrote this position last before t" back to an original entry.  The transform reads order[] as a
//     stream;
//   * the accumulator rows are COMPACT: for angle n only r in [rmin(n), rmax(n)] can occur (the projections of the page's
//     corners), 115 (W + H) cells in all instead of 180 (2 (W + H) + 1), as int16 with a bias, two to a dword: 1.37 MB for an A4
//     page = the LDS of 9 CUs.  Member g of a group owns the angles n = g (mod G): every member holds 1/G of the cells and casts
//     1/G of the votes;
//   * every member keeps a PRIVATE copy of the page's point mask (1 bit per pixel, in device memory, touched by that workgroup
//     only), replays the line walks on it and erases the same pixels: nothing about the mask is ever exchanged;
//   * the only exchange: after the votes of a block of up to 256 points every member publishes {sequence, first point of the
//     block whose vote reached the threshold in one of MY cells, its count and angle} as one 8-byte granule (agent-scope store)
//     and polls the G granules of its group (agent-scope loads).  No cell at the threshold anywhere: the block stands.
//     Otherwise the earliest such point wins (largest count, then smallest angle: OpenCV's scan order), every member takes back
//     the votes of the younger points, walks the line, erases it, takes back the votes of its pixels if it is long enough
//     (each member its own angles), strikes the erased pixels from the points it has already fetched (a point lies on the walk
//     or not: arithmetic, no re-read) and goes on behind the winning point.
// Votes within a block are exact although 64 points vote in one instruction: the cells of an angle belong to one wavefront, a
// wavefront casts the votes of 64 points for one angle with one ds_add_rtn, and where a returned count reaches the threshold
// the points that hit the same cell in that instruction are ranked in visiting order (the hardware's order among them is not
// defined; the set of returned values is).
//
// Safety: workgroups of a group wait for each other, so all of them must be resident: the launch is cooperative and sized by the
// occupancy query.  Every spin is bounded by a wall-clock budget (s_memrealtime); a member that runs out raises the group's abort
// word and every member leaves; pages not marked done are redone by k_ppht_mw in the same process (deskew.hip).
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "prl_internal.h"

namespace prl_hip {
namespace {

constexpr int kNumAngle = 180;
#ifndef PRL_GRP_THREADS
#define PRL_GRP_THREADS 1024
#endif
constexpr int kGrpThreads = PRL_GRP_THREADS;   // one workgroup per CU (its LDS is full): all its wavefronts work on the page
constexpr int kGrpWaves = kGrpThreads / 64;
constexpr int kFetch = 256;             // points fetched per round (the first kFetch threads)
constexpr int kRing = 1024;             // fetched points waiting for their turn (entries of 4 bytes)
constexpr int kMaxSub = 4;              // a block is up to kMaxSub x 64 points
constexpr int kU = kGrpWaves >= 16 ? 2 : kGrpWaves >= 8 ? 3 : 5;   // angles of a wavefront voted together
constexpr int kWalkLoads = 4;           // chunks of 64 steps a wavefront reads per round trip of the first walk
constexpr int kPostWave = 2;             // posts the next line's exchange while wavefronts 0 and 1 walk the current line
constexpr int kKeep = 32;               // chunks per direction whose points the first walk leaves for the second
constexpr int kEraseWave0 = kGrpWaves > 4 ? kFetch / 64 : 1;   // the wavefronts that erase (wavefront 0 polls the mailboxes; with many wavefronts those that fetch mask bits are spared too)
constexpr int kEraseWaves = kGrpWaves - kEraseWave0;
constexpr int kMaxA = 180;              // angles per member (a small page is one member's)
constexpr int kMaxG = 32;               // members per group
constexpr unsigned kCellBias = 0x4000u; // a cell holds count + bias: neither half of a dword ever borrows from the other
constexpr int kMboxSlots = 4;
constexpr int kGranStride = 8;          // granules 64 bytes apart
constexpr unsigned kAborted = 0xffffffffu;
constexpr int kMaxSide = 8000;          // |count| <= 2 max(W, H) < bias

struct GrpAngle {
    float c, s;   // the trig table's entries of this angle
    int base;     // cell index of r = 0 in this member's accumulator (may be negative: r starts at rmin)
    int n;        // the angle
};

struct GrpArgs {
    int width, height, threshold, line_length, line_gap;
    int G, n_groups, xcd_aligned, n_list;
    int rowwords;                       // dwords per row of a bit mask
    size_t mask_words;                  // dwords per bit mask (multiple of 4)
    const unsigned* mask0;              // the pages' point masks, read only: page i at mask0 + i * mask_words
    unsigned* pmask;                    // private copies: workgroup b at pmask + b * mask_words
    const unsigned* order;              // visiting order (points x | y << 16), page i at order + nz_off[i]
    const unsigned long long* nz_off;
    const unsigned* count;
    const GrpAngle* tab;                // [G][kMaxA]
    const int* tab_n;                   // [G] angles of member g
    const int* tab_dwords;              // [G] accumulator dwords of member g
    const float* ttab;                  // kNumAngle x {cos, sin}
    const int4* ltab;                   // kNumAngle x {xflag, dx0, dy0, 0}: the walk of a line of that angle
    int* lines; const unsigned long long* lines_off; const unsigned* lines_cap; unsigned* n_lines;
    unsigned long long* mbox;           // [n_groups][kMboxSlots][kMaxG] granules, zeroed
    unsigned* abort_word;               // [n_groups], 64 bytes apart, zeroed
    unsigned* queue;                    // next entry of page_list, zeroed
    const int* page_list;               // pages to process, heaviest first
    unsigned* status;                   // per page: 1 = done
    unsigned long long spin_budget;     // s_memrealtime ticks (100 MHz) a member waits for its group before it gives up
    int kill_group, kill_after;         // tests: member 1 of this group stops answering after that many exchanges (-1: never)
    unsigned long long* prof;           // optional: per page 12 counters (exchanges, blocks, triggers, good lines, walk rounds, ...)
};

__device__ __forceinline__ int cv_round_f(float v) { return __float2int_rn(v); }
__device__ __forceinline__ unsigned uni(unsigned v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ unsigned long long uni64(unsigned long long v)
{
    return (unsigned long long)uni((unsigned)v) | ((unsigned long long)uni((unsigned)(v >> 32)) << 32);
}

__device__ __forceinline__ void step_pixel(int xflag, unsigned x0, unsigned y0, int dx, int dy, unsigned s, int* j1, int* i1)
{
    const int x = (int)(x0 + s * (unsigned)dx), y = (int)(y0 + s * (unsigned)dy);  // wraps like the reference's repeated adds
    if (xflag) { *j1 = x; *i1 = y >> 16; }
    else { *j1 = x >> 16; *i1 = y; }
}

__device__ __forceinline__ unsigned wave_max_u32(unsigned v)
{
    int x = (int)(v ^ 0x80000000u);
    x = max(x, __builtin_amdgcn_update_dpp(INT_MIN, x, 0x111, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(INT_MIN, x, 0x112, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(INT_MIN, x, 0x114, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(INT_MIN, x, 0x118, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(INT_MIN, x, 0x142, 0xa, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(INT_MIN, x, 0x143, 0xc, 0xf, false));
    return (unsigned)__builtin_amdgcn_readlane(x, 63) ^ 0x80000000u;
}

__device__ __forceinline__ unsigned load_sc1(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// ---- the visiting order ----------------------------------------------------------------------------------------------------------

// Step t of HoughLinesProbabilistic's main loop draws idx_t = rnd[t] % (N - t), visits list[idx_t] and moves list[N - t - 1] there.
// Per list position the steps that write it, as a linked list in arbitrary order.
__global__ void __launch_bounds__(256) k_order_link(const unsigned* __restrict__ rnd, const unsigned* __restrict__ count,
                                                    const unsigned long long* __restrict__ nz_off, unsigned* __restrict__ head,
                                                    unsigned* __restrict__ next)
{
    const int page = blockIdx.y;
    const unsigned N = count[page];
    const unsigned t = blockIdx.x * 256u + threadIdx.x;
    if (t >= N) return;
    const unsigned long long off = nz_off[page];
    const unsigned idx = rnd[t] % (N - t);
    next[off + t] = atomicExch(&head[off + idx], t);
}

// The point visited at step t.  Position p = idx_t holds its original entry unless an earlier step wrote it; the last such step
// tau moved the content of position N - tau - 1 there, which again is original unless a step before tau wrote it, and so on.
__global__ void __launch_bounds__(256) k_order_resolve(const unsigned* __restrict__ rnd, const unsigned* __restrict__ count,
                                                       const unsigned long long* __restrict__ nz_off, const unsigned* __restrict__ head,
                                                       const unsigned* __restrict__ next, const unsigned* __restrict__ nz,
                                                       unsigned* __restrict__ order)
{
    const int page = blockIdx.y;
    const unsigned N = count[page];
    const unsigned t = blockIdx.x * 256u + threadIdx.x;
    if (t >= N) return;
    const unsigned long long off = nz_off[page];
    const unsigned* hd = head + off;
    const unsigned* nx = next + off;
    unsigned p = rnd[t] % (N - t), bound = t;
    for (;;) {
        unsigned best = 0xffffffffu;   // the last step before `bound` that wrote position p
        for (unsigned e = hd[p]; e != 0xffffffffu; e = nx[e])
            if (e < bound && (best == 0xffffffffu || e > best)) best = e;
        if (best == 0xffffffffu) break;
        bound = best;
        p = N - best - 1;
    }
    order[off + t] = nz[off + p];
}

// byte mask (k_dark_mask of deskew.hip) -> 1 bit per pixel, rows of `rowwords` dwords; one wavefront per row
__global__ void __launch_bounds__(64) k_pack_bits(int width, int height, const uint8_t* __restrict__ mask, size_t mask_page,
                                                  unsigned* __restrict__ bits, size_t mask_words, int rowwords)
{
    const int page = blockIdx.y, y = blockIdx.x, lane = threadIdx.x;
    const uint8_t* m = mask + (size_t)page * mask_page + (size_t)y * width;
    unsigned* out = bits + (size_t)page * mask_words + (size_t)y * rowwords;
    for (int x0 = 0; x0 < width; x0 += 64) {
        const int x = x0 + lane;
        const unsigned long long b = __ballot(x < width && m[x] != 0);
        if (lane == 0) {
            out[x0 >> 5] = (unsigned)b;
            if ((x0 >> 5) + 1 < rowwords) out[(x0 >> 5) + 1] = (unsigned)(b >> 32);
        }
    }
}

// ---- the transform ---------------------------------------------------------------------------------------------------------------

__global__ void __launch_bounds__(kGrpThreads) k_ppht_group(GrpArgs a)
{
    extern __shared__ unsigned acc[];                   // this member's cells, two to a dword
    __shared__ unsigned ring[kRing];                    // x | y << 16 | live << 31
    __shared__ unsigned s_key[2][kMaxSub * 64];         // per point of the block: count * 256 + 255 - angle where a cell reached the threshold
    __shared__ unsigned s_any[2], s_x[2], s_post[2], s_hit, s_end[2];
    __shared__ unsigned long long s_B[2][2][kKeep], s_set2[kGrpWaves];   // s_B[line parity][direction][chunk]
    __shared__ GrpAngle s_ang[kMaxA];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int G = a.G;
    int group, member;
    if (a.xcd_aligned) {   // blocks b and b + 8 share an XCD (observed placement, speed only): a group's members sit on one
        const int slot = blockIdx.x >> 3;
        group = (blockIdx.x & 7) + 8 * (slot / G);
        member = slot % G;
    } else {
        group = blockIdx.x / G;
        member = blockIdx.x % G;
    }
    if (group >= a.n_groups) return;
    const int W = a.width, H = a.height, thr = a.threshold, rowwords = a.rowwords;
    unsigned* pm = a.pmask + (size_t)blockIdx.x * a.mask_words;
    unsigned long long* mbox = a.mbox + (size_t)group * kMboxSlots * kMaxG * kGranStride;
    unsigned* abort_word = a.abort_word + (size_t)group * 16;
    const int A = a.tab_n[member], acc_dwords = a.tab_dwords[member];
    if (tid < A) s_ang[tid] = a.tab[member * kMaxA + tid];
    __syncthreads();
    // this wavefront's first kU angles stay in registers (a member of a full-size group has no more than that per wavefront)
    float rc[kU], rs[kU];
    int rb[kU], rn[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
        const GrpAngle g = s_ang[min(wv + u * kGrpWaves, A - 1)];
        rc[u] = g.c; rs[u] = g.s; rb[u] = g.base; rn[u] = g.n;
    }
    unsigned seq = 0;
    int kill_left = (group == a.kill_group && member == 1) ? a.kill_after : -1;

    // One exchange: every member publishes a payload (post: wavefront 0 / lane 0's value), everybody gets the maximum over the group
    // (collect; kAborted: the group gave up; ends with a workgroup barrier).  A posted exchange may be abandoned - nobody collects
    // it - as long as every member decides so: the transform posts the NEXT line of a block before it walks the current one.
    unsigned long long dbg_polls = 0, dbg_prehit = 0, dbg_pollcyc = 0, dbg_barcyc = 0;
    unsigned long long pre_v = 0;   // wavefront 0: the group's granules as read ahead of collect() (prefetch_poll)
    unsigned pre_seq = 0, pre_ab = 0;
    auto post = [&](unsigned payload, int pw) {   // wavefront pw publishes (its lane 0's value)
        ++seq;
        if (wv == pw) {
            if (lane == 0) {
                s_post[seq & 1] = payload;
                if (G > 1)
                    __hip_atomic_store(mbox + ((size_t)(seq & (kMboxSlots - 1)) * kMaxG + member) * kGranStride,
                                       ((unsigned long long)seq << 32) | payload, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    };
    auto prefetch_poll = [&]() {   // (wavefront 0) reads the granules of the exchange posted last without waiting for them
        if (wv == 0 && G > 1) {
            const unsigned long long* box = mbox + (size_t)(seq & (kMboxSlots - 1)) * kMaxG * kGranStride;
            pre_v = (unsigned long long)seq << 32;
            if (lane < G) pre_v = __hip_atomic_load(box + lane * kGranStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            pre_ab = 0;
            if (lane == 63) pre_ab = load_sc1(abort_word);
            pre_seq = seq;
        }
    };
    auto collect = [&](bool post_was_mine) -> unsigned {   // post_was_mine: wavefront 0 posted, with no barrier since (its value is `posted0`)
        const unsigned long long tc0 = a.prof ? __builtin_readcyclecounter() : 0;
        if (wv == 0) {
            unsigned res = 0;
            if (kill_left == 0) res = kAborted;   // (tests) this member leaves without a word: the others run out of patience, raise the abort word and leave too
            else if (G == 1) res = s_post[seq & 1];
            else {
                if (kill_left > 0) --kill_left;
                const unsigned long long* box = mbox + (size_t)(seq & (kMboxSlots - 1)) * kMaxG * kGranStride;
                unsigned long long t0 = 0;
                unsigned spins = 0;
                bool have = pre_seq == seq;
                const bool had = have;
                for (;;) {
                    ++dbg_polls;
                    unsigned long long v = (unsigned long long)seq << 32;
                    unsigned ab = 0;
                    if (have) { v = pre_v; ab = pre_ab; have = false; }
                    else {
                        if (lane < G) v = __hip_atomic_load(box + lane * kGranStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (lane == 63) ab = load_sc1(abort_word);
                    }
                    if (__ballot(ab != 0)) { res = kAborted; break; }
                    if (!__ballot((unsigned)(v >> 32) != seq)) { res = wave_max_u32((unsigned)v); if (had && spins == 0) ++dbg_prehit; break; }
                    if ((++spins & 63u) == 0) {
                        const unsigned long long now = __builtin_amdgcn_s_memrealtime();
                        if (t0 == 0) t0 = now;
                        else if (now - t0 > a.spin_budget) {
                            if (lane == 0) __hip_atomic_store(abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            res = kAborted;
                            break;
                        }
                    }
                }
            }
            if (lane == 0) s_x[seq & 1] = res;
        }
        const unsigned long long tc1 = a.prof ? __builtin_readcyclecounter() : 0;
        __syncthreads();
        if (a.prof) { dbg_pollcyc += tc1 - tc0; dbg_barcyc += __builtin_readcyclecounter() - tc1; }
        return s_x[seq & 1];
    };
    auto exchange = [&](unsigned payload) -> unsigned {
        post(payload, 0);
        return collect(true);
    };

    for (;;) {
        // ---- next page of the queue (the leader pops, everybody learns it) ----
        unsigned pop = 0;
        if (member == 0 && tid == 0) pop = atomicAdd(a.queue, 1u) + 1u;
        const unsigned got = exchange(pop);
        if (got == kAborted || got - 1u >= (unsigned)a.n_list) return;
        const int page = a.page_list[got - 1u];
        const unsigned N = a.count[page];
        const unsigned* order = a.order + a.nz_off[page];
        const unsigned lines_cap = a.lines_cap[page];
        const unsigned long long lines_off = a.lines_off[page];

        // ---- per page: empty cells, a private copy of the point mask ----
        for (int i = tid; i < acc_dwords; i += kGrpThreads) acc[i] = kCellBias | (kCellBias << 16);
        {
            const uint4* src = reinterpret_cast<const uint4*>(a.mask0 + (size_t)page * a.mask_words);
            uint4* dst = reinterpret_cast<uint4*>(pm);
            for (size_t i = tid; i < a.mask_words / 4; i += kGrpThreads) dst[i] = src[i];
        }
        if (tid < kMaxSub * 64) s_key[0][tid] = 0;
        if (tid == 0) s_any[0] = 0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        unsigned cursor = 0, head = 0, head_vis = 0, next_order = 0, n_lines = 0, blk = 0;
        int S = kMaxSub;
        bool pend = false, o_valid = false, pend_stale = false, stale_line = false, prev_trig = false, late_erase = false;
        int wb = 0;   // which half of s_B the line being walked uses
        unsigned pg_pt = 0, pg_word = 0, o_reg = 0;
        // the line walked last (whose erasures may still be on their way when the next mask bits are fetched)
        int l_xflag = 0, l_dx0 = 0, l_dy0 = 0, l_j = 0, l_i = 0, l_e0 = 0, l_e1 = 0;
        unsigned l_x0 = 0, l_y0 = 0;
        auto on_last_line = [&](unsigned ent) -> bool {
            const int px = (int)(ent & 0x7fffu), py = (int)((ent >> 16) & 0x7fffu);
            if (l_xflag) {
                const int s = (px - l_j) * l_dx0;
                return s >= -l_e1 && s <= l_e0 && ((int)(l_y0 + (unsigned)s * (unsigned)l_dy0) >> 16) == py;
            }
            const int s = (py - l_i) * l_dy0;
            return s >= -l_e1 && s <= l_e0 && ((int)(l_x0 + (unsigned)s * (unsigned)l_dx0) >> 16) == px;
        };
        // The erasure of a line too short to count is put off until the next line is being walked (or the block ends): the
        // wavefronts that erase have nothing else to do then.  It erases the LAST line (l_*), whose points the first walk left in
        // s_B[buf]; only lines within kKeep chunks per direction are put off.
        auto erase_last_line = [&](int buf) {
            if (wv < kEraseWave0) return;
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const int dx = d ? -l_dx0 : l_dx0, dy = d ? -l_dy0 : l_dy0;
                const unsigned end = (unsigned)(d ? l_e1 : l_e0);
                for (unsigned myc = (unsigned)(wv - kEraseWave0); myc <= (end >> 6); myc += kEraseWaves) {
                    const unsigned st = myc * 64u + (unsigned)lane;
                    if (st <= end && !(d == 1 && st == 0) && ((uni64(s_B[buf][d][myc]) >> lane) & 1ull)) {
                        int j1, i1;
                        step_pixel(l_xflag, l_x0, l_y0, dx, dy, st, &j1, &i1);
                        atomicAnd(pm + (size_t)i1 * rowwords + (j1 >> 5), ~(1u << (j1 & 31)));
                    }
                }
            }
        };
        auto flush_pend = [&]() {   // the fetched chunk joins the ring (visible to the others after the next barrier)
            if (tid < kFetch && head + tid < N) {
                unsigned ent = pg_pt | (((pg_word >> (pg_pt & 31u)) & 1u) << 31);
                if (pend_stale && (ent >> 31) && on_last_line(ent)) ent &= 0x7fffffffu;
                ring[(head + tid) & (kRing - 1)] = ent;
            }
            head = min(N, head + (unsigned)kFetch);
            pend = false;
        };
        unsigned n_xchg = 0, n_blocks = 0, n_trig = 0, n_rounds = 0;
        bool aborted = false;
        // (diagnostics: shader-clock cycles per phase, only when the caller asked for them)
        unsigned long long ph[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, ph_t = a.prof ? __builtin_readcyclecounter() : 0;
        auto mark = [&](int k) {
            if (a.prof) {
                const unsigned long long now = __builtin_readcyclecounter();
                ph[k] += now - ph_t;
                ph_t = now;
            }
        };

        unsigned long long rounds_left = 2ull * N + 64;   // every round retires a point or (at most thrice) waits for the first fetch
        while (cursor < N) {
            if (rounds_left-- == 0) {   // (cannot happen; a bounded loop is the last line of defence on a shared machine)
                if (tid == 0) __hip_atomic_store(abort_word, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                aborted = true;
                break;
            }
            const int kb = (int)(blk & 1u);
            // ---- keep the ring filled: the points of a chunk arrive one round after their list entries, their mask bits one round later ----
            if (pend) flush_pend();
            if (o_valid && head + kFetch - cursor <= (unsigned)kRing) {
                pg_pt = o_reg;
                pg_word = 0;
                if (tid < kFetch && head + tid < N) pg_word = load_sc1(pm + (size_t)(pg_pt >> 16) * rowwords + ((pg_pt & 0xffffu) >> 5));
                pend = true;
                pend_stale = stale_line;   // (this fetch may overtake the last line's erasures)
                stale_line = false;
                o_valid = false;
            }
            if (!o_valid && next_order < N) {
                o_reg = (tid < kFetch && next_order + tid < N) ? order[next_order + tid] : 0u;
                next_order += kFetch;
                o_valid = true;
            }
            if (tid < kMaxSub * 64) s_key[kb ^ 1][tid] = 0;
            if (tid == 0) s_any[kb ^ 1] = 0;
            const unsigned win = min((unsigned)S * 64u, head_vis - cursor);
            if (win == 0) {   // (nothing fetched yet: start of the page, or a line ended on the last fetched point)
                __syncthreads();
                head_vis = head;
                mark(0);
                continue;
            }
            const int nsub = (int)((win + 63u) >> 6);
            ++n_blocks;
            ++blk;
            mark(0);

            // ---- votes of the block: lane = point; a wavefront owns its angles' cells, so that the votes of successive sub-blocks
            // reach a cell in visiting order; all of a wavefront's votes of a block (up to kMaxSub x kU) are in flight together ----
            unsigned trig_e = 0, trig_w = 0;
            unsigned ent[kMaxSub];
            bool live[kMaxSub];
#pragma unroll
            for (int sub = 0; sub < kMaxSub; ++sub) {
                const unsigned e = cursor + (unsigned)sub * 64u + (unsigned)lane;
                ent[sub] = sub < nsub ? ring[e & (kRing - 1)] : 0u;
                live[sub] = sub < nsub && e < cursor + win && (ent[sub] >> 31) != 0;
            }
            auto votes = [&](bool undo) {
                for (int u0 = 0; wv + u0 * kGrpWaves < A; u0 += kU) {
                    float c[kU], s[kU];
                    int b[kU], n[kU];
#pragma unroll
                    for (int u = 0; u < kU; ++u) {
                        if (u0 == 0) { c[u] = rc[u]; s[u] = rs[u]; b[u] = rb[u]; n[u] = rn[u]; }
                        else {
                            const GrpAngle g = s_ang[min(wv + (u0 + u) * kGrpWaves, A - 1)];
                            c[u] = g.c; s[u] = g.s; b[u] = g.base; n[u] = g.n;
                        }
                    }
                    int ci[kMaxSub][kU];
                    unsigned old[kMaxSub][kU];
#pragma unroll
                    for (int sub = 0; sub < kMaxSub; ++sub) {
#pragma unroll
                        for (int u = 0; u < kU; ++u) { ci[sub][u] = 0; old[sub][u] = 0; }
                        if (sub >= nsub || (undo && (unsigned)(sub * 64 + 63) <= trig_w)) continue;
                        const float fx = (float)(ent[sub] & 0x7fffu), fy = (float)((ent[sub] >> 16) & 0x7fffu);
                        const bool lv = undo ? (live[sub] && cursor + (unsigned)sub * 64u + (unsigned)lane > trig_e) : live[sub];
#pragma unroll
                        for (int u = 0; u < kU; ++u) {
                            if (wv + (u0 + u) * kGrpWaves >= A) continue;
                            ci[sub][u] = b[u] + cv_round_f(fx * c[u] + fy * s[u]);
                            if (lv) {
                                unsigned* cell = acc + (ci[sub][u] >> 1);
                                const unsigned inc = (ci[sub][u] & 1) ? 0x10000u : 1u;
                                if (undo) atomicSub(cell, inc);
                                else old[sub][u] = atomicAdd(cell, inc);
                            }
                        }
                    }
                    if (undo) continue;
#pragma unroll
                    for (int sub = 0; sub < kMaxSub; ++sub) {
                        if (sub >= nsub) continue;
#pragma unroll
                        for (int u = 0; u < kU; ++u) {
                            if (wv + (u0 + u) * kGrpWaves >= A) continue;
                            const int cnt = (int)((old[sub][u] >> ((ci[sub][u] & 1) * 16)) & 0xffffu) - (int)kCellBias + 1;
                            const bool mine = live[sub] && wv + (u0 + u) * kGrpWaves < A;
                            unsigned long long m = __ballot(mine && cnt >= thr);
                            if (m) {
                                // A cell reached the threshold.  The points that hit one cell in this instruction were served in an
                                // order of the hardware's choosing; the smallest returned count belongs, in visiting order, to the
                                // first of them.
                                while (m) {
                                    const int l = __ffsll((long long)m) - 1;
                                    const int X = __builtin_amdgcn_readlane(ci[sub][u], l);
                                    const unsigned long long D = __ballot(mine && ci[sub][u] == X);
                                    m &= ~D;
                                    int cmin = INT_MAX;
                                    for (unsigned long long d = D; d; d &= d - 1)
                                        cmin = min(cmin, __builtin_amdgcn_readlane(cnt, __ffsll((long long)d) - 1));
                                    if ((D >> lane) & 1ull) {
                                        const int T = cmin + __popcll(D & ((1ull << lane) - 1ull));
                                        if (T >= thr) atomicMax(&s_key[kb][sub * 64 + lane], (unsigned)(T * 256 + 255 - n[u]));
                                    }
                                }
                                if (lane == 0) s_any[kb] = 1;
                            }
                        }
                    }
                }
            };
            votes(false);
            // Erasures are never waited for where they are issued.  The wavefronts that erase drain theirs here and before the barrier
            // behind every line's strike; so at any time only the LAST line's erasures can be on their way, and whoever reads mask bits
            // that could be among them (the next fetch, the next first walk) strikes that line's pixels himself.
            if (wv >= kEraseWave0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            stale_line = false;
            mark(1);
            // ---- the lines of this block.  After a line that was too short to count (no votes taken back) the rest of the block
            // stands as it was voted, unless the walk erased one of its points: the next line is the next point with a key. ----
            unsigned resume = 0;
            bool block_done = false;
            auto next_key = [&](int pw) -> unsigned {   // (wavefront pw) the first point at or behind `resume` whose vote reached the threshold in one of my cells
                unsigned payload = 0;
                if (wv == pw && s_any[kb]) {
                    for (int sub = (int)(resume >> 6); sub < nsub && !payload; ++sub) {
                        const unsigned k = s_key[kb][sub * 64 + lane];
                        const unsigned long long bb = __ballot(k != 0 && (unsigned)(sub * 64 + lane) >= resume);
                        if (bb) {
                            const int p = __ffsll((long long)bb) - 1;
                            payload = ((1023u - (unsigned)(sub * 64 + p)) << 22) | (unsigned)__builtin_amdgcn_readlane((int)k, p);
                        }
                    }
                }
                return payload;
            };
            post(next_key(0), 0);
            for (;;) {
            const unsigned res = collect(true);
            mark(2);
            ++n_xchg;
            if (res == kAborted) { aborted = true; break; }
            if (res == 0) {   // every vote of the block stands
                if (late_erase) { erase_last_line(wb); late_erase = false; }
                cursor += win;
                S = min(2 * S, kMaxSub);
                prev_trig = false;
                block_done = true;
                break;
            }

            // ---- a line: the earliest point whose vote reached the threshold, the first angle with the largest count ----
            ++n_trig;
            trig_w = 1023u - (res >> 22);
            const int max_n = 255 - (int)(res & 255u);
            trig_e = cursor + trig_w;
            if (tid == 0) s_hit = 0;   // (set after the walks, behind their barriers)
            // on the assumption that this line is too short to count and erases no point of the block, the next line of the block
            // is known already: its exchange travels while this line is walked (abandoned if the assumption fails)
            resume = trig_w + 1u;
            post(next_key(kPostWave), kPostWave);
            if (late_erase) {   // (the line before this one; the walk below strikes its pixels from what it reads: stale_line)
                erase_last_line(wb);
                late_erase = false;
            }
            wb ^= 1;
            if (pend) {   // the chunk in flight read its mask bits before this line is erased: into the ring with it, corrected below
                flush_pend();
                __syncthreads();
            }
            mark(3);
            const unsigned tq = uni(ring[trig_e & (kRing - 1)]);
            const int j = (int)(tq & 0x7fffu), i = (int)((tq >> 16) & 0x7fffu);
            const int4 lt = a.ltab[max_n];
            const int xflag = lt.x, dx0 = lt.y, dy0 = lt.z;
            unsigned x0 = (unsigned)j, y0 = (unsigned)i;
            if (xflag) y0 = (y0 << 16) + (1u << 15);
            else x0 = (x0 << 16) + (1u << 15);
            // first walk (read only): wavefront d walks direction d, kWalkLoads x 64 steps per round trip, and finds on its own (ballots,
            // no shared memory) the step at which the walk leaves the page or has seen more than line_gap steps in a row without a
            // point, and the last point before it.  Which steps held a point is left in s_B for the second walk.
            unsigned end_step[2] = {0, 0};
            if (wv < 2) {
                const int dir = wv;
                const int dx = dir ? -dx0 : dx0, dy = dir ? -dy0 : dy0;
                unsigned end = 0;
                int gap = 0;
                bool stop = false;
                for (unsigned base = 0, r = 0; !stop && base < (1u << 17); base += kWalkLoads * 64, ++r) {
                    ++n_rounds;
                    bool inb[kWalkLoads];
                    unsigned w1[kWalkLoads], pix[kWalkLoads];
                    int bitpos[kWalkLoads];
#pragma unroll
                    for (int h = 0; h < kWalkLoads; ++h) {
                        int j1, i1;
                        step_pixel(xflag, x0, y0, dx, dy, base + (unsigned)(h * 64 + lane), &j1, &i1);
                        inb[h] = j1 >= 0 && j1 < W && i1 >= 0 && i1 < H;
                        bitpos[h] = j1 & 31;
                        pix[h] = (unsigned)j1 | ((unsigned)i1 << 16);
                        w1[h] = 0;
                        if (inb[h]) w1[h] = load_sc1(pm + (size_t)i1 * rowwords + (j1 >> 5));
                    }
#pragma unroll
                    for (int h = 0; h < kWalkLoads; ++h) {
                        if (stop) continue;
                        bool set = inb[h] && ((w1[h] >> bitpos[h]) & 1u);
                        if (stale_line && set) set = !on_last_line(pix[h]);   // (that line's erasures may not have landed yet)
                        const unsigned long long bs = __ballot(set);
                        const unsigned chunk = r * kWalkLoads + (unsigned)h;
                        if (lane == 0 && chunk < (unsigned)kKeep) s_B[wb][dir][chunk] = bs;
                        // pointless steps right before mine
                        const unsigned long long below = bs & ((1ull << lane) - 1ull);
                        const int run = below ? lane - 1 - (63 - __clzll((long long)below)) : lane + gap;
                        const unsigned long long bv = __ballot(!inb[h] || (!set && run + 1 > a.line_gap));
                        unsigned long long upto = bs;   // the points before the walk's end
                        if (bv) {
                            stop = true;
                            upto &= (1ull << (__ffsll((long long)bv) - 1)) - 1ull;
                        }
                        if (upto) end = base + (unsigned)(h * 64 + 63 - __clzll((long long)upto));
                        gap = bs ? __clzll((long long)bs) : gap + 64;
                    }
                }
                if (lane == 0) s_end[dir] = end;
            }
            __syncthreads();
            end_step[0] = uni(s_end[0]);
            end_step[1] = uni(s_end[1]);
            mark(4);
            prefetch_poll();
            int ex[2], ey[2];
            step_pixel(xflag, x0, y0, dx0, dy0, end_step[0], &ex[0], &ey[0]);
            step_pixel(xflag, x0, y0, -dx0, -dy0, end_step[1], &ex[1], &ey[1]);
            const bool good_line = abs(ex[1] - ex[0]) >= a.line_length || abs(ey[1] - ey[0]) >= a.line_length;
            // the fetched points that lay on the erased stretch are gone
            l_xflag = xflag; l_dx0 = dx0; l_dy0 = dy0; l_j = j; l_i = i; l_e0 = (int)end_step[0]; l_e1 = (int)end_step[1];
            l_x0 = x0; l_y0 = y0;
            bool hit_block = false;   // a point of this block behind the line's was erased: its vote (and what followed it) does not stand
            for (unsigned e = trig_e + 1u + (unsigned)tid; e < head; e += kGrpThreads) {
                const unsigned en = ring[e & (kRing - 1)];
                if ((en >> 31) && on_last_line(en)) {
                    ring[e & (kRing - 1)] = en & 0x7fffffffu;
                    hit_block = hit_block || e < cursor + win;
                }
            }
            if (__ballot(hit_block) && lane == 0) s_hit = 1;
            if (good_line) {
                if (member == 0 && tid == 0 && n_lines < lines_cap) {
                    int* ln = a.lines + (lines_off + n_lines) * 4;
                    ln[0] = ex[0]; ln[1] = ey[0]; ln[2] = ex[1]; ln[3] = ey[1];
                }
                ++n_lines;
            }
            if (wv >= kEraseWave0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the line before this one: long landed)
            __syncthreads();     // the ring as corrected; the flag
            stale_line = true;   // from here on this line's erasures are on their way
            const bool restart = good_line || s_hit != 0;
            mark(6);
            // second walk: erase the pixels up to the line's ends (this member's copy; wavefronts kEraseWave0 and up, one step per
            // lane); a good line takes their votes back (every wavefront its angles).  Which steps held a point is known from the
            // first walk's ballots (kKeep chunks per direction); beyond them the mask is read again.
            late_erase = !restart && (end_step[0] >> 6) < (unsigned)kKeep && (end_step[1] >> 6) < (unsigned)kKeep;
            for (int d = 0; d < 2 && !late_erase; ++d) {
                const int dx = d ? -dx0 : dx0, dy = d ? -dy0 : dy0;
                const unsigned last_chunk = end_step[d] >> 6;
                for (unsigned ch0 = 0; ch0 <= last_chunk; ch0 += kEraseWaves) {
                    const unsigned nch = min((unsigned)kEraseWaves, last_chunk + 1 - ch0);
                    const bool cached = ch0 + nch <= (unsigned)kKeep;
                    if (wv >= kEraseWave0) {
                        const unsigned myc = ch0 + (unsigned)(wv - kEraseWave0);
                        const unsigned st = myc * 64u + (unsigned)lane;
                        const bool in_range = myc <= last_chunk && st <= end_step[d] && !(d == 1 && st == 0);   // (step 0 is erased by d = 0)
                        bool set = false;
                        int j1, i1;
                        step_pixel(xflag, x0, y0, dx, dy, st, &j1, &i1);
                        unsigned* wp = pm + (size_t)i1 * rowwords + (j1 >> 5);
                        if (cached) set = in_range && ((uni64(s_B[wb][d][min(myc, (unsigned)kKeep - 1)]) >> lane) & 1ull);
                        else if (in_range) set = (load_sc1(wp) >> (j1 & 31)) & 1u;
                        if (set) atomicAnd(wp, ~(1u << (j1 & 31)));
                        if (!cached) {
                            const unsigned long long bb = __ballot(set);
                            if (lane == 0) s_set2[wv - kEraseWave0] = bb;
                        }
                    }
                    if (!cached) __syncthreads();
                    if (good_line) {
                        for (unsigned cc = 0; cc < nch; ++cc) {
                            unsigned long long bits;
                            if (cached) {
                                bits = uni64(s_B[wb][d][ch0 + cc]);
                                const long long room = (long long)end_step[d] - (long long)((ch0 + cc) * 64u);   // steps of this chunk up to the end
                                if (room < 63) bits &= (2ull << room) - 1ull;
                                if (d == 1 && ch0 + cc == 0) bits &= ~1ull;
                            } else {
                                bits = uni64(s_set2[cc]);
                            }
                            if (!bits) continue;
                            int jq, iq;
                            step_pixel(xflag, x0, y0, dx, dy, (ch0 + cc) * 64u + (unsigned)lane, &jq, &iq);
                            const bool on = (bits >> lane) & 1ull;
                            const float fx = (float)jq, fy = (float)iq;
                            for (int u0 = 0; wv + u0 * kGrpWaves < A; u0 += kU) {
#pragma unroll
                                for (int u = 0; u < kU; ++u) {
                                    float c, s;
                                    int b;
                                    if (u0 == 0) { c = rc[u]; s = rs[u]; b = rb[u]; }
                                    else {
                                        const GrpAngle g = s_ang[min(wv + (u0 + u) * kGrpWaves, A - 1)];
                                        c = g.c; s = g.s; b = g.base;
                                    }
                                    const int ci = b + cv_round_f(fx * c + fy * s);
                                    if (on && wv + (u0 + u) * kGrpWaves < A) atomicSub(acc + (ci >> 1), (ci & 1) ? 0x10000u : 1u);
                                }
                            }
                        }
                    }
                    if (!cached) __syncthreads();
                }
            }
            mark(5);
            if (restart) {   // the votes of the younger points no longer stand as they were cast: take them back, vote again behind the line
                votes(true);
                cursor = trig_e + 1u;
                S = prev_trig ? 1 : kMaxSub;   // lines in quick succession: short blocks (fewer votes to take back)
                prev_trig = true;
                __syncthreads();
                break;
            }
            if (resume >= win) {   // (the exchange posted for the rest of the block is abandoned: there is no rest)
                if (late_erase) { erase_last_line(wb); late_erase = false; }
                cursor += win;
                prev_trig = true;
                block_done = true;
                break;
            }
            }   // lines of this block
            head_vis = head;
            if (aborted) break;
            (void)block_done;
        }
        if (aborted) return;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (member == 0 && tid == 0) {
            a.n_lines[page] = n_lines;
            if (a.prof) {
                unsigned long long* pr = a.prof + (size_t)page * 16;
                pr[0] = n_xchg; pr[1] = n_blocks; pr[2] = n_trig; pr[3] = n_lines; pr[4] = n_rounds;
                for (int k = 0; k < 7; ++k) pr[5 + k] = ph[k];
                pr[12] = dbg_polls; pr[13] = dbg_prehit; pr[14] = dbg_pollcyc; pr[15] = dbg_barcyc;
            }
            __hip_atomic_store(a.status + page, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

size_t r256(size_t v) { return (v + 255) / 256 * 256; }

}  // namespace

// Geometry of the group kernel for W x H pages: members per group (0: the page does not qualify), angle tables.
struct GroupPlan {
    int G = 0;
    std::vector<GrpAngle> tab;     // [G][kMaxA]
    std::vector<int> tab_n, tab_dwords;
    size_t lds_bytes = 0;          // dynamic LDS of the launch
};

static GroupPlan plan_group(int width, int height, int threshold, const float* ttab, size_t lds_budget, int min_g)
{
    GroupPlan gp;
    if (std::max(width, height) > kMaxSide || threshold < 1) return gp;
    // r = cvRound(x cos + y sin) over the page: between the projections of two opposite corners (float32 products and sum are
    // monotone in x and y), widened by one cell against the float32 roundings
    int rmin[kNumAngle], len[kNumAngle];
    for (int n = 0; n < kNumAngle; ++n) {
        const double c = ttab[2 * n], s = ttab[2 * n + 1];
        const double x_lo = c >= 0 ? 0 : width - 1, x_hi = c >= 0 ? width - 1 : 0;
        const double y_lo = s >= 0 ? 0 : height - 1, y_hi = s >= 0 ? height - 1 : 0;
        rmin[n] = (int)std::floor(x_lo * c + y_lo * s) - 1;
        const int rmax = (int)std::ceil(x_hi * c + y_hi * s) + 1;
        len[n] = rmax - rmin[n] + 1;
    }
    for (int G = std::max(1, min_g); G <= kMaxG; ++G) {
        if ((kNumAngle + G - 1) / G > kMaxA) continue;
        size_t worst = 0;
        for (int g = 0; g < G; ++g) {
            size_t cells = 0;
            for (int n = g; n < kNumAngle; n += G) cells += (size_t)len[n];
            worst = std::max(worst, (cells + 1) / 2 * 4);
        }
        if (worst > lds_budget) continue;
        gp.G = G;
        gp.lds_bytes = worst;
        gp.tab.assign((size_t)G * kMaxA, GrpAngle{0.f, 0.f, 0, 0});
        gp.tab_n.assign((size_t)G, 0);
        gp.tab_dwords.assign((size_t)G, 0);
        for (int g = 0; g < G; ++g) {
            int cells = 0, k = 0;
            for (int n = g; n < kNumAngle; n += G, ++k) {
                gp.tab[(size_t)g * kMaxA + k] = GrpAngle{ttab[2 * n], ttab[2 * n + 1], cells - rmin[n], n};
                cells += len[n];
            }
            gp.tab_n[(size_t)g] = k;
            gp.tab_dwords[(size_t)g] = (cells + 1) / 2;
        }
        return gp;
    }
    return gp;
}

bool ppht_group_eligible(int width, int height, int threshold)
{
    return std::max(width, height) <= kMaxSide && threshold >= 16 && threshold < 16000;   // (a tiny threshold makes every point a line: one exchange per point)
}

/*
 * Runs the group kernel over the pages of `page_list` (indices into the pass; heaviest first).  The caller (ppht_pages, deskew.hip)
 * has produced the byte masks, the point lists, counts and offsets and the segment lists' layout; status[i] = 1 for every page that was
 * finished here - the caller redoes the others.  Enqueues on `stream`, does not synchronise.
 */
int ppht_group_run(DeviceCtx* ctx, PphtGroupIn& in, hipStream_t stream)
{
    const EnvKnobs& knobs = env_knobs();
    const int n_list = (int)in.page_list.size();
    if (n_list == 0) return PRL_OK;
    const int W = in.width, H = in.height;
    // static LDS of the kernel + a margin: what is left of the CU's 160 KB for the cells
    hipFuncAttributes fa{};
    PRL_HIP_CHECK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k_ppht_group)));
    int max_lds = 0;
    PRL_HIP_CHECK(hipDeviceGetAttribute(&max_lds, hipDeviceAttributeMaxSharedMemoryPerBlock, ctx->device));
    max_lds = std::max(max_lds, 160 * 1024);   // gfx950: 160 KB per CU, all of it available to one workgroup
    const size_t lds_budget = (size_t)max_lds - fa.sharedSizeBytes - 256;
    GroupPlan gp = plan_group(W, H, in.threshold, in.h_ttab, lds_budget, knobs.ppht_group_g);
    if (gp.G == 0) return PRL_ERR_BAD_ARG;
    const int G = gp.G;
    PRL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ppht_group), hipFuncAttributeMaxDynamicSharedMemorySize, (int)gp.lds_bytes));
    int per_cu = 0;
    PRL_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_ppht_group, kGrpThreads, gp.lds_bytes));
    const int cus = std::max(1, ctx->cu_count);
    int max_blocks = std::min(per_cu, 1) * cus;   // one workgroup per CU: a member's speed is its CU's LDS
    if (in.cu_limit > 0) max_blocks = std::min(max_blocks, in.cu_limit);
    if (knobs.ppht_group_cus > 0) max_blocks = std::min(max_blocks, knobs.ppht_group_cus);
    if (max_blocks < G) return PRL_ERR_BAD_ARG;
    const bool xcd = knobs.ppht_group_xcd != 0 && max_blocks >= 8 * G;
    int n_groups, grid;
    if (xcd) {
        const int per_xcd = max_blocks / (8 * G);   // groups per XCD
        const int rounds = std::min(per_xcd, (n_list + 7) / 8);
        n_groups = std::min(n_list, 8 * rounds);
        grid = 8 * G * rounds;
    } else {
        n_groups = std::min(n_list, max_blocks / G);
        grid = n_groups * G;
    }

    const int rowwords = (W + 31) / 32;
    const size_t mask_words = ((size_t)rowwords * H + 3) / 4 * 4;
    const int n_pages = in.n_pages;
    size_t nz_total = 0;
    unsigned max_n = 0;
    for (int i = 0; i < n_pages; ++i) {
        nz_total = std::max<size_t>(nz_total, (size_t)in.h_nzoff[(size_t)i] + (in.h_count[(size_t)i] + 63) / 64 * 64);
        max_n = std::max(max_n, in.h_count[(size_t)i]);
    }
    // workspace: bit masks (shared + private), linked lists, order, random numbers, tables, mailboxes
    const size_t b_mask0 = r256(mask_words * 4 * (size_t)n_pages), b_pmask = r256(mask_words * 4 * (size_t)grid);
    const size_t b_list = r256(nz_total * 4 + 256);
    const size_t b_tab = r256(gp.tab.size() * sizeof(GrpAngle)), b_tabn = r256((size_t)G * 4), b_ltab = r256(kNumAngle * sizeof(int4));
    const size_t b_mbox = r256((size_t)n_groups * kMboxSlots * kMaxG * kGranStride * 8), b_abort = r256((size_t)n_groups * 64 + 64);
    const size_t b_pl = r256((size_t)n_list * 4), b_status = r256((size_t)n_pages * 4), b_prof = r256((size_t)n_pages * 128);
    const size_t total = b_mask0 + b_pmask + 3 * b_list + b_tab + 2 * b_tabn + b_ltab + b_mbox + b_abort + b_pl + b_status + b_prof;
    int st = ensure_buffer(&ctx->ppht_buf[3], &ctx->ppht_bytes[3], total);
    if (st != PRL_OK) return st;
    // cv::RNG(-1)'s output: the same sequence for every page of every call (HoughLinesProbabilistic seeds it anew each time), kept
    // on the device and extended when a page has more points than any before
    if (ctx->ppht_rnd_n < max_n) {
        const size_t want = std::max<size_t>({(size_t)max_n, ctx->ppht_rnd_n * 2, (size_t)1 << 20});
        std::vector<uint32_t> rnd(want);
        uint64_t s = ~0ull;
        for (size_t i = 0; i < want; ++i) {
            s = (uint64_t)(uint32_t)s * 4164903690u + (s >> 32);
            rnd[i] = (uint32_t)s;
        }
        ctx->ppht_rnd_n = 0;
        st = ensure_buffer(&ctx->ppht_buf[5], &ctx->ppht_bytes[5], want * 4);
        if (st != PRL_OK) return st;
        PRL_HIP_CHECK(hipMemcpy(ctx->ppht_buf[5], rnd.data(), want * 4, hipMemcpyHostToDevice));
        ctx->ppht_rnd_n = want;
    }
    uint8_t* w = static_cast<uint8_t*>(ctx->ppht_buf[3]);
    auto take = [&](size_t bytes) { uint8_t* p = w; w += bytes; return p; };
    // [mbox | abort words | queue + status] first: one memset
    unsigned long long* d_mbox = reinterpret_cast<unsigned long long*>(take(b_mbox));
    unsigned* d_abort = reinterpret_cast<unsigned*>(take(b_abort));   // its last 64 bytes hold the queue counter
    unsigned* d_queue = d_abort + (size_t)n_groups * 16;
    unsigned* d_status = reinterpret_cast<unsigned*>(take(b_status));
    unsigned long long* d_prof = reinterpret_cast<unsigned long long*>(take(b_prof));
    const size_t b_zero = b_mbox + b_abort + b_status + b_prof;
    unsigned* d_head = reinterpret_cast<unsigned*>(take(b_list));
    unsigned* d_next = reinterpret_cast<unsigned*>(take(b_list));
    unsigned* d_order = reinterpret_cast<unsigned*>(take(b_list));
    const unsigned* d_rnd = static_cast<const unsigned*>(ctx->ppht_buf[5]);
    GrpAngle* d_tab = reinterpret_cast<GrpAngle*>(take(b_tab));
    int* d_tabn = reinterpret_cast<int*>(take(b_tabn));
    int* d_tabdw = reinterpret_cast<int*>(take(b_tabn));
    int* d_pl = reinterpret_cast<int*>(take(b_pl));
    int4* d_ltab = reinterpret_cast<int4*>(take(b_ltab));
    unsigned* d_mask0 = reinterpret_cast<unsigned*>(take(b_mask0));
    unsigned* d_pmask = reinterpret_cast<unsigned*>(take(b_pmask));

    // the small host tables, in a block the caller keeps until it has synchronised the stream
    in.keep.resize(gp.tab.size() * sizeof(GrpAngle) + (size_t)G * 8 + (size_t)n_list * 4 + kNumAngle * sizeof(int4));
    unsigned char* k_tab = in.keep.data();
    unsigned char* k_tabn = k_tab + gp.tab.size() * sizeof(GrpAngle);
    unsigned char* k_tabdw = k_tabn + (size_t)G * 4;
    unsigned char* k_pl = k_tabdw + (size_t)G * 4;
    std::memcpy(k_tab, gp.tab.data(), gp.tab.size() * sizeof(GrpAngle));
    std::memcpy(k_tabn, gp.tab_n.data(), (size_t)G * 4);
    std::memcpy(k_tabdw, gp.tab_dwords.data(), (size_t)G * 4);
    std::memcpy(k_pl, in.page_list.data(), (size_t)n_list * 4);
    // how a line of angle n is walked (HoughLinesProbabilistic: a = -sin, b = cos; the longer component steps by one pixel, the other in
    // 16.16 fixed point)
    int4* k_ltab = reinterpret_cast<int4*>(k_pl + (size_t)n_list * 4);
    for (int n = 0; n < kNumAngle; ++n) {
        const float fa = -in.h_ttab[2 * n + 1], fb = in.h_ttab[2 * n];
        int4 lt{0, 0, 0, 0};
        if (std::fabs((double)fa) > std::fabs((double)fb)) {
            lt.x = 1;
            lt.y = fa > 0 ? 1 : -1;
            lt.z = (int)std::lrint((double)(fb * 65536.f) / std::fabs((double)fa));
        } else {
            lt.y = (int)std::lrint((double)(fa * 65536.f) / std::fabs((double)fb));
            lt.z = fb > 0 ? 1 : -1;
        }
        std::memcpy(k_ltab + n, &lt, sizeof(lt));
    }
    PRL_HIP_CHECK(hipMemcpyAsync(d_ltab, k_ltab, kNumAngle * sizeof(int4), hipMemcpyHostToDevice, stream));
    PRL_HIP_CHECK(hipMemcpyAsync(d_tab, k_tab, gp.tab.size() * sizeof(GrpAngle), hipMemcpyHostToDevice, stream));
    PRL_HIP_CHECK(hipMemcpyAsync(d_tabn, k_tabn, (size_t)G * 4, hipMemcpyHostToDevice, stream));
    PRL_HIP_CHECK(hipMemcpyAsync(d_tabdw, k_tabdw, (size_t)G * 4, hipMemcpyHostToDevice, stream));
    PRL_HIP_CHECK(hipMemcpyAsync(d_pl, k_pl, (size_t)n_list * 4, hipMemcpyHostToDevice, stream));
    PRL_HIP_CHECK(hipMemsetAsync(d_mbox, 0, b_zero, stream));
    PRL_HIP_CHECK(hipMemsetAsync(d_head, 0xff, nz_total * 4, stream));
    if (max_n) {
        const dim3 og((max_n + 255) / 256, (unsigned)n_pages);
        hipLaunchKernelGGL(k_order_link, og, dim3(256), 0, stream, d_rnd, in.d_count, in.d_nzoff, d_head, d_next);
        hipLaunchKernelGGL(k_order_resolve, og, dim3(256), 0, stream, d_rnd, in.d_count, in.d_nzoff, d_head, d_next, in.d_nz, d_order);
    }
    hipLaunchKernelGGL(k_pack_bits, dim3((unsigned)H, (unsigned)n_pages), dim3(64), 0, stream, W, H, in.d_mask, in.mask_page, d_mask0,
                       mask_words, rowwords);
    PRL_HIP_CHECK(hipGetLastError());
    if (in.ev[0]) PRL_HIP_CHECK(hipEventRecord(in.ev[0], stream));

    GrpArgs a{};
    a.width = W; a.height = H; a.threshold = in.threshold; a.line_length = in.line_length; a.line_gap = in.line_gap;
    a.G = G; a.n_groups = n_groups; a.xcd_aligned = xcd ? 1 : 0; a.n_list = n_list;
    a.rowwords = rowwords; a.mask_words = mask_words; a.mask0 = d_mask0; a.pmask = d_pmask;
    a.order = d_order; a.nz_off = in.d_nzoff; a.count = in.d_count;
    a.tab = d_tab; a.tab_n = d_tabn; a.tab_dwords = d_tabdw; a.ttab = in.d_ttab; a.ltab = d_ltab;
    a.lines = in.d_lines; a.lines_off = in.d_lnoff; a.lines_cap = in.d_cap; a.n_lines = in.d_nlines;
    a.mbox = d_mbox; a.abort_word = d_abort; a.queue = d_queue; a.page_list = d_pl; a.status = d_status;
    a.spin_budget = (unsigned long long)std::max(1, knobs.ppht_group_spin_ms) * 100000ull;
    a.kill_group = knobs.ppht_group_kill >= 0 ? 0 : -1;
    a.kill_after = knobs.ppht_group_kill;
    a.prof = d_prof;
    void* kargs[] = {&a};
    hipError_t le = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(k_ppht_group), dim3((unsigned)grid), dim3(kGrpThreads), kargs,
                                               (unsigned)gp.lds_bytes, stream);
    if (le != hipSuccess) {   // (too large for the device as it is now, or no cooperative launches: the caller's kernel takes the pages)
        (void)hipGetLastError();
        set_error_detail(std::string("k_ppht_group: ") + hipGetErrorString(le));
        return PRL_ERR_HIP;
    }
    if (in.ev[1]) PRL_HIP_CHECK(hipEventRecord(in.ev[1], stream));
    if (in.status_out) PRL_HIP_CHECK(hipMemcpyAsync(in.status_out, d_status, (size_t)n_pages * 4, hipMemcpyDeviceToHost, stream));
    if (in.prof_out) PRL_HIP_CHECK(hipMemcpyAsync(in.prof_out, d_prof, (size_t)n_pages * 128, hipMemcpyDeviceToHost, stream));
    in.geometry_out[0] = G; in.geometry_out[1] = n_groups; in.geometry_out[2] = grid; in.geometry_out[3] = (int)gp.lds_bytes;
    return PRL_OK;
}

}  // namespace prl_hip
