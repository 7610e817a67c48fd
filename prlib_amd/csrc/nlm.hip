// nlm.hip — prl::denoise (src/denoise/denoiseNLM.cpp:29-32) = cv::fastNlMeansDenoisingColored(in, out,
// h = strength) with OpenCV's defaults hColor = 3, template 7x7, search 21x21.
//
// [upstream] semantics restated in SURVEY.md Appendix C (the CPU checker restates them independently):
//   LBGR -> Lab (8-bit fixed point), NLM on the L plane with h, NLM on the interleaved ab planes with
//   h = 3, Lab -> LBGR.  The NLM core is all-integer (SSD over the 7x7 template for each of the 441
//   offsets, `>> 6` binning, host-built weight LUT, int32 accumulation, rounding division), so the
//   device result is bit-identical to the CPU restatement given the same LUT.
//
// Kernel shape (k_nlm_y for 1 and 2 channels, k_nlm for 3).  NL-means is ALU-bound by construction (441 offsets x
// 49 taps per pixel and channel against 2 B/px of traffic), so the design minimises instructions and LDS cycles
// per (pixel, offset):
//   - a workgroup owns a 64x32 output tile (4 wavefronts x 8 rows); it stages the tile plus its 13-pixel
//     reflect-101 halo in LDS EXPANDED to one 8-byte element per pixel position (the 8 bytes that start there),
//     plus SB(pos) = sum over the 7x7 template around pos of E^2 (int32) and the non-zero part of the LUT;
//   - SSD = SA + SB - 2*AB with AB = sum over the template of E(p+t)*E(q+t).  A thread owns one column and walks
//     down 8 output rows per offset; the 7 horizontally adjacent template taps of the other patch are ONE
//     conflict-free ds_read_b64, the own side stays in registers, V_DOT4_U32_U8 accumulates a template row in 2
//     instructions per channel, the 14 row dots are chained through the dot accumulator (prefix sums) so the
//     vertical 7-row window is one subtraction per output, and 2-3 offsets are processed together so that
//     independent chains interleave.  ~37 VALU instructions per (wavefront row, offset) for one channel.
// Roofline: bound by integer VALU issue (and, before the expanded layout, by LDS cycles), not HBM; bench reports
// the HBM fraction because the metric asks for it (algorithmic 2 B/px per plane byte) and states the ALU bound.
#include <algorithm>
#include <cmath>
#include <climits>
#include <type_traits>
#include <vector>

#include "prl_internal.h"

namespace prl_hip {

namespace {

constexpr int kT = 7, kS = 21, kTH = 3, kSH = 10, kBorder = kTH + kSH;  // 13
constexpr int TILE_W = 64, ROWS = 8, WAVES = 4, TILE_H = ROWS * WAVES;  // 8 rows: 94 VGPRs, 5 waves/SIMD (16 rows: 178 VGPRs, 1.3x slower)   // 64 x 32 outputs per workgroup
constexpr int EXT_W = TILE_W + 2 * kBorder, EXT_H = TILE_H + 2 * kBorder; // 90 x 90 staged pixels
constexpr int SB_W = TILE_W + 2 * kSH, SB_H = TILE_H + 2 * kSH;           // 84 x 84 template energies
constexpr int kLutMax = 1024;  // non-zero LUT entries kept in LDS (h <= ~13 for 1 channel); else global

__device__ __forceinline__ int reflect101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}

struct NlmParams {
    int width, height;
    int n_lut;              // number of leading LUT entries that may be non-zero; lut[n_lut] == 0
    const int* lut;         // device copy of almost_dist2weight_ (n_lut + 1 entries)
};

template <int CH>
struct Nb {  // the 7-pixel horizontal neighbourhood of one position: 7*CH bytes in NB dwords
    static constexpr int NB = (7 * CH + 3) / 4;
    unsigned d[NB];
};

// Neighbourhood at an arbitrary byte address of ONE copy: NB+1 aligned dword reads + byte funnel shifts.
template <int CH>
__device__ __forceinline__ Nb<CH> lds_nb(const unsigned* base, unsigned sh)
{
    Nb<CH> v;
    unsigned w[Nb<CH>::NB + 1];
#pragma unroll
    for (int k = 0; k <= Nb<CH>::NB; ++k) w[k] = base[k];
#pragma unroll
    for (int k = 0; k < Nb<CH>::NB; ++k) v.d[k] = __builtin_amdgcn_alignbyte(w[k + 1], w[k], sh);
    return v;
}

// sum over the 7*CH bytes of a.b ; `a` has its padding bytes (beyond 7*CH) zeroed
template <int CH>
__device__ __forceinline__ unsigned dot_nb(const Nb<CH>& a, const Nb<CH>& b, unsigned acc)
{
#pragma unroll
    for (int k = 0; k < Nb<CH>::NB; ++k) acc = __builtin_amdgcn_udot4(a.d[k], b.d[k], acc, false);
    return acc;
}

// ---- k_nlm_y (1 and 2 channels: the two planes prl::denoise filters) ------------------------------------------
// The kernel is LDS-bound (SQ_LDS_IDX_ACTIVE ~ 93 % of the CU cycles in the first version), so the staged tile is
// kept EXPANDED: element P of a row is the 8 bytes starting at pixel P (u64).  A lane's neighbourhood is then one
// 8-byte-aligned ds_read_b64 (two for the 14-byte ab neighbourhood: elements P and P+4), consecutive lanes read
// consecutive elements = all 64 banks exactly once for every offset, and no funnel shifts are needed.  (Byte-
// shifted copies read with ds_read2_b32 gave a 2-way bank conflict for 3 of 4 alignments.)
// LUT_LDS: the non-zero part of the weight table fits the LDS copy (always, for prl::denoise's h range); the
// other instantiation reads it from memory.  A run-time choice inside the loop made the compiler issue both loads.
template <int CH>
struct YGeo {
    static constexpr int NE = CH;                              // u64 elements per neighbourhood (8 / 16 bytes >= 7 / 14)
    static constexpr int YP = EXT_W + (CH == 2 ? 4 : 0);       // elements per staged row
    static constexpr int RAWP = (EXT_W * CH + 24) / 4 * 4;     // raw bytes per row, zero padded (elements overrun the row)
    static constexpr int SB_BYTES = SB_H * SB_W * 4, RAW_BYTES = EXT_H * RAWP;
    static constexpr int SBRAW_WORDS = (SB_BYTES > RAW_BYTES ? SB_BYTES : RAW_BYTES) / 4 + 4;
    // last dword of a neighbourhood: keep the bytes below 7*CH
    static constexpr unsigned kLastMask = CH == 1 ? 0x00ffffffu : 0x0000ffffu;
};

template <int CH>
struct NbY {
    uint2 e[YGeo<CH>::NE];
};

template <int CH>
__device__ __forceinline__ NbY<CH> y_nb(const uint2* p)
{
    NbY<CH> v;
    v.e[0] = p[0];
    if (CH == 2) v.e[1] = p[4];
    return v;
}

// XL layout: element P is the 4 bytes starting at pixel P; a neighbourhood is elements P and P + 4 (1 channel) or
// P, P + 2, P + 4, P + 6 (2 channels), read with ds_read2_b32.  Half the LDS footprint of the 8-byte elements - 4
// workgroups per CU instead of 2 - for the same LDS bytes per neighbourhood.
template <int CH>
__device__ __forceinline__ NbY<CH> y_nb(const unsigned* p)
{
    NbY<CH> v;
#pragma unroll
    for (int k = 0; k < CH; ++k) v.e[k] = make_uint2(p[(8 * k) / CH], p[(8 * k + 4) / CH]);
    return v;
}

template <int CH>
__device__ __forceinline__ NbY<CH> y_mask(NbY<CH> v)
{
    v.e[CH - 1].y &= YGeo<CH>::kLastMask;
    return v;
}

template <int CH>
__device__ __forceinline__ unsigned y_dot(const NbY<CH>& a, const NbY<CH>& b, unsigned acc)
{
#pragma unroll
    for (int k = 0; k < CH; ++k) {
        acc = __builtin_amdgcn_udot4(a.e[k].x, b.e[k].x, acc, false);
        acc = __builtin_amdgcn_udot4(a.e[k].y, b.e[k].y, acc, false);
    }
    return acc;
}

// a - b kept as its own v_sub_u32: the empty asm hides the value, so the compiler cannot re-form a - 2b as
// shift-left + add3.  (A hand-written "v_sub_u32" in the asm is NOT safe here: gfx950 needs wait states between a
// v_dot4 write and a different VALU read of that register, and the hazard pass does not look inside inline asm.)
__device__ __forceinline__ unsigned sub_keep(unsigned a, unsigned b)
{
    unsigned d = a - b;
    asm("" : "+v"(d));
    return d;
}

constexpr int kLutMaxXL = 639;  // the 4-byte-element variant keeps a shorter LDS table (40 KB per workgroup in all)

template <int CH, bool LUT_LDS, bool XL = false>
__device__ __forceinline__ void nlm_y_body(const PageSet& src, const PageSetOut& dst, const NlmParams& np)
{
    using G = YGeo<CH>;
    using Elt = typename std::conditional<XL, unsigned, uint2>::type;
    constexpr int YP = XL ? EXT_W : G::YP, RAWP = G::RAWP;  // XL: the last element read is 83 + 6
    constexpr int kLutN = XL ? kLutMaxXL : kLutMax;
    __shared__ __attribute__((aligned(16))) Elt ytile[EXT_H * YP];
    __shared__ __attribute__((aligned(16))) unsigned sbraw[G::SBRAW_WORDS];   // raw bytes while staging, then SB
    __shared__ int lut_s[LUT_LDS ? kLutN + 1 : 1];
    unsigned char* raw = reinterpret_cast<unsigned char*>(sbraw);
    int* sb = reinterpret_cast<int*>(sbraw);

    const int page = blockIdx.z;
    const uint8_t* __restrict__ img = src.page(page);
    uint8_t* __restrict__ out = dst.page(page);
    const int W = np.width, H = np.height;
    const int x0 = blockIdx.x * TILE_W, y0 = blockIdx.y * TILE_H;

    // raw tile with its reflect-101 halo (copyMakeBorder(BORDER_DEFAULT) by 13), zero padded rows.  Every fetch of the
    // tile is issued before the first one is used: ONE memory round trip per workgroup.  (In the config-5 chain this kernel
    // runs beside the angle search of the next pass, which keeps the memory system saturated with scattered atomics;
    // a round trip then takes tens of microseconds, and the eight dependent batches the plain loop compiled to doubled the
    // kernel's time.)
    {
        using Px = typename std::conditional<CH == 1, uint8_t, unsigned short>::type;   // 2 channels: aligned 16-bit pixels
        static_assert(CH <= 2, "k_nlm_y filters the L and the ab plane");
        constexpr int kIt = (EXT_H * EXT_W + 255) / 256;
        Px px[kIt];
#pragma unroll
        for (int it = 0; it < kIt; ++it) {
            const int i = (int)threadIdx.x + it * 256;
            px[it] = 0;
            if (i < EXT_H * EXT_W) {
                const int r = i / EXT_W, c = i - r * EXT_W;
                const int sy = reflect101(y0 - kBorder + r, H), sx = reflect101(x0 - kBorder + c, W);
                px[it] = *reinterpret_cast<const Px*>(img + (size_t)sy * src.step + (size_t)sx * CH);
            }
        }
        for (int i = threadIdx.x; i < EXT_H * (RAWP - EXT_W * CH); i += blockDim.x) {
            const int r = i / (RAWP - EXT_W * CH), c = i - r * (RAWP - EXT_W * CH);
            raw[r * RAWP + EXT_W * CH + c] = 0;
        }
        if (LUT_LDS)
            for (int i = threadIdx.x; i <= kLutN; i += blockDim.x) lut_s[i] = (i < np.n_lut) ? np.lut[i] : 0;
#pragma unroll
        for (int it = 0; it < kIt; ++it) {
            const int i = (int)threadIdx.x + it * 256;
            if (i < EXT_H * EXT_W) {
                const int r = i / EXT_W, c = i - r * EXT_W;
                *reinterpret_cast<Px*>(raw + r * RAWP + c * CH) = px[it];
            }
        }
    }
    __syncthreads();

    // expand: element (r, k) = raw bytes [k*CH, k*CH + 8) of row r (XL: 4 bytes)
    for (int i = threadIdx.x; i < EXT_H * YP; i += blockDim.x) {
        const int r = i / YP, k = i - r * YP;
        const unsigned P = (unsigned)(k * CH);
        const unsigned* q = reinterpret_cast<const unsigned*>(raw + r * RAWP + (P & ~3u));
        const unsigned w0 = q[0], w1 = q[1], w2 = q[2];
        const uint2 e8 = make_uint2(__builtin_amdgcn_alignbyte(w1, w0, P & 3u), __builtin_amdgcn_alignbyte(w2, w1, P & 3u));
        if constexpr (XL) ytile[i] = e8.x;
        else ytile[i] = e8;
    }
    __syncthreads();

    // SB(r, c) = sum over the 7x7 template centred at staged position (r + 3, c + 3) of E^2: one thread per
    // column and third of the rows, a horizontal dot per row and a sliding vertical sum of seven of them
    {
        constexpr int CHR = (SB_H + 2) / 3;  // rows per thread
        if (threadIdx.x < 3 * SB_W) {
            const int c = threadIdx.x % SB_W, rb = (threadIdx.x / SB_W) * CHR;
            unsigned ring[kT];
            unsigned acc = 0;
#pragma unroll
            for (int i = 0; i < CHR + kT - 1; ++i) {
                const int rr = min(rb + i, EXT_H - 1);  // rows past the tile are never stored
                const NbY<CH> a = y_nb<CH>(ytile + rr * YP + c);
                const unsigned hd = y_dot<CH>(y_mask<CH>(a), a, 0u);
                acc += hd;
                if (i >= kT) acc -= ring[i % kT];
                ring[i % kT] = hd;
                if (i >= kT - 1 && rb + i - (kT - 1) < SB_H) sb[(rb + i - (kT - 1)) * SB_W + c] = (int)acc;
            }
        }
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int oy0 = wv * ROWS;  // first output row of this wavefront inside the tile
    // staged coordinates: output (oy, lane) sits at staged (oy + 13, lane + 13); its template row ty starts at
    // staged column lane + 10 = element index lane + 10.
    const Elt* abase = ytile + (oy0 + kSH) * YP + (lane + kSH);
    const int* sa_base = sb + (oy0 + kSH) * SB_W + (lane + kSH);

    // own side, once: template energies and the 14 neighbourhood rows (padding bytes zeroed) stay in registers
    int SA[ROWS];
#pragma unroll
    for (int i = 0; i < ROWS; ++i) SA[i] = sa_base[i * SB_W];
    NbY<CH> A[ROWS + kT - 1];
#pragma unroll
    for (int r = 0; r < ROWS + kT - 1; ++r) A[r] = y_mask<CH>(y_nb<CH>(abase + r * YP));

    unsigned est[ROWS][CH], wsum[ROWS];
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
        wsum[i] = 0;
#pragma unroll
        for (int k = 0; k < CH; ++k) est[i][k] = 0;
    }

    // NO offsets at a time: their dot chains are independent, which keeps the vector ALU fed at the 2-3
    // wavefronts per SIMD the LDS footprint allows
    const unsigned dmax = (unsigned)np.n_lut * 64u;
    auto offsets = [&](int o, auto no_tag) {
        constexpr int NO = decltype(no_tag)::value;
        const Elt* bbase[NO];
        const int* sb_o[NO];
#pragma unroll
        for (int j = 0; j < NO; ++j) {
            const int dy = (o + j) / kS - kSH, dx = (o + j) - ((o + j) / kS) * kS - kSH;
            bbase[j] = abase + dy * YP + dx;
            sb_o[j] = sa_base + dy * SB_W + dx;
        }
        // P[r] = sum of the row dots 0..r, chained through the dot accumulator; AB(i) = P[i+6] - P[i-1]
        unsigned P[NO][ROWS + kT - 1];
        unsigned cen[NO][ROWS + kT - 1];
        unsigned acc[NO];
#pragma unroll
        for (int j = 0; j < NO; ++j) acc[j] = 0;
#pragma unroll
        for (int r = 0; r < ROWS + kT - 1; ++r) {
#pragma unroll
            for (int j = 0; j < NO; ++j) {
                const NbY<CH> b = y_nb<CH>(bbase[j] + r * YP);
                acc[j] = y_dot<CH>(A[r], b, acc[j]);
                P[j][r] = acc[j];
                // centre pixel of the other patch's row r: bytes 3*CH .. 4*CH-1 of its neighbourhood
                cen[j][r] = CH == 1 ? b.e[0].x : b.e[0].y;
            }
            if (r >= kT - 1) {
                const int i = r - (kT - 1);  // output row; its centre row was fetched at step i + 3 = r - 3
#pragma unroll
                for (int j = 0; j < NO; ++j) {
                    const unsigned AB = i == 0 ? P[j][r] : P[j][r] - P[j][i - 1];
                    // D = SA + SB - 2 AB >= 0.  Two subtractions and shift-right + mask instead of shift-left forms:
                    // on gfx950 v_lshlrev / v_add3 / v_lshl_add issue in 4 cycles, v_sub / v_lshrrev / v_and in 2
                    // (profiles/r01/valu_issue_costs.txt)
                    unsigned D = (unsigned)(SA[i] + sb_o[j][i * SB_W]);
                    D = sub_keep(D, AB);
                    D = sub_keep(D, AB);
                    D = min(D, dmax);            // entries from n_lut on are zero
                    // byte offset of entry D >> 6 (almost_template_window_size_sq_bin_shift_)
                    const unsigned boff = (D >> 4) & ~3u;
                    const unsigned wgt = LUT_LDS ? *reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(lut_s) + boff)
                                                 : *reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(np.lut) + boff);
                    wsum[i] += wgt;
                    const unsigned cw = cen[j][r - 3];
                    if (CH == 1) {
                        est[i][0] += __umul24(wgt, cw >> 24);           // wgt <= 19096, byte: 24-bit mad
                    } else {
                        est[i][0] += __umul24(wgt, (cw >> 16) & 0xffu);
                        est[i][CH - 1] += __umul24(wgt, cw >> 24);
                    }
                }
            }
        }
    };
    // XL layout: GROUPS of offsets that share their LDS reads.  The neighbourhoods of offsets dx, dx + S, dx + 2S (S = 4
    // elements for one channel, 2 for two) overlap in whole dwords - elements P, P+4 | P+4, P+8 | P+8, P+12 - so a group of
    // three fetches 4 (6) dwords per template row instead of 6 (12): the kernel is bound by LDS cycles, and these reads are
    // 60 % (75 %) of them.  The three dot chains are independent, as before.
    auto group = [&](int dy, int dx, auto ng_tag) {
        constexpr int NG = decltype(ng_tag)::value;
        constexpr int S = 4 / CH, NE = CH == 1 ? NG + 1 : NG + 3, NR = ROWS + kT - 1;
        const unsigned* bb = reinterpret_cast<const unsigned*>(abase) + dy * YP + dx;
        const int* sbb = sa_base + dy * SB_W + dx;
        unsigned P[NG][NR], cen[NG][NR], acc[NG];
#pragma unroll
        for (int j = 0; j < NG; ++j) acc[j] = 0;
        // Software pipeline (the compiler on its own issues every LDS read right before its first use and waits for it):
        // the elements of template row r+1 and the SB of this row's output are requested before the dots of row r, and a
        // weight read from the table is only accumulated after the dots of the following row.  sched_barrier pins the order.
        unsigned nxt[NE], wpend[NG], cpend[NG];
        int sbn[NG];
#pragma unroll
        for (int k = 0; k < NE; ++k) nxt[k] = bb[S * k];
#pragma unroll
        for (int r = 0; r <= NR; ++r) {
            unsigned e[NE];
            int sbv[NG];
            if (r < NR) {
#pragma unroll
                for (int k = 0; k < NE; ++k) e[k] = nxt[k];
#pragma unroll
                for (int j = 0; j < NG; ++j) sbv[j] = sbn[j];
                if (r + 1 < NR) {
#pragma unroll
                    for (int k = 0; k < NE; ++k) nxt[k] = bb[(r + 1) * YP + S * k];
                    if (r + 1 >= kT - 1) {
#pragma unroll
                        for (int j = 0; j < NG; ++j) sbn[j] = sbb[j * S + (r + 1 - (kT - 1)) * SB_W];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < NG; ++j) {
                    NbY<CH> b;
                    b.e[0] = make_uint2(e[j], e[j + 1]);
                    if constexpr (CH == 2) b.e[1] = make_uint2(e[j + 2], e[j + 3]);
                    acc[j] = y_dot<CH>(A[r], b, acc[j]);
                    P[j][r] = acc[j];
                    cen[j][r] = CH == 1 ? b.e[0].x : b.e[0].y;
                }
            }
            if (r >= kT) {  // the weights requested one row ago
                const int i = r - kT;
#pragma unroll
                for (int j = 0; j < NG; ++j) {
                    wsum[i] += wpend[j];
                    if (CH == 1) {
                        est[i][0] += __umul24(wpend[j], cpend[j] >> 24);
                    } else {
                        est[i][0] += __umul24(wpend[j], (cpend[j] >> 16) & 0xffu);
                        est[i][CH - 1] += __umul24(wpend[j], cpend[j] >> 24);
                    }
                }
            }
            if (r >= kT - 1 && r < NR) {
                const int i = r - (kT - 1);
#pragma unroll
                for (int j = 0; j < NG; ++j) {
                    const unsigned AB = i == 0 ? P[j][r] : P[j][r] - P[j][i - 1];
                    unsigned D = (unsigned)(SA[i] + sbv[j]);
                    D = sub_keep(D, AB);
                    D = sub_keep(D, AB);
                    D = min(D, dmax);
                    const unsigned boff = (D >> 4) & ~3u;
                    wpend[j] = LUT_LDS ? *reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(lut_s) + boff)
                                       : *reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(np.lut) + boff);
                    cpend[j] = cen[j][r - 3];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    if constexpr (XL) {
#pragma unroll 1
        for (int dy = -kSH; dy <= kSH; ++dy) {
            if constexpr (CH == 1) {
                // dx index 0..20: {0,4,8} {1,5,9} {2,6,10} {3,7,11} {12,16,20} | {13,17} {14,18} {15,19}
#pragma unroll 1
                for (int g = 0; g < 5; ++g) group(dy, (g < 4 ? g : 12) - kSH, std::integral_constant<int, 3>{});
#pragma unroll 1
                for (int g = 0; g < 3; ++g) group(dy, 13 + g - kSH, std::integral_constant<int, 2>{});
            } else {
                // pairs {0,2} {1,3} {4,6} {5,7} ... {16,18} {17,19} | {20}   (triples need ~150 registers: spills at 168)
#pragma unroll 1
                for (int g = 0; g < 10; ++g) group(dy, (g >> 1) * 4 + (g & 1) - kSH, std::integral_constant<int, 2>{});
                group(dy, 20 - kSH, std::integral_constant<int, 1>{});
            }
        }
    } else {
        // offsets in flight per wavefront, 8-byte elements (2 wavefronts per SIMD): 3 / 2 measured best
        constexpr int kPair = CH == 1 ? 3 : 2;
#pragma unroll 1
        for (int o = 0; o + kPair <= kS * kS; o += kPair) offsets(o, std::integral_constant<int, kPair>{});
#pragma unroll 1
        for (int o = (kS * kS) / kPair * kPair; o < kS * kS; ++o) offsets(o, std::integral_constant<int, 1>{});
    }

    // divByWeightsSum + saturate_cast<uchar>
    const int gx = x0 + lane;
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
        const int gy = y0 + oy0 + i;
        if (gx < W && gy < H) {
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                const unsigned v = (est[i][k] + wsum[i] / 2u) / wsum[i];
                out[(size_t)gy * dst.step + (size_t)gx * CH + k] = (uint8_t)(v > 255u ? 255u : v);
            }
        }
    }
}

// The 8-byte-element variant runs 2 wavefronts per SIMD (LDS), the 4-byte one (XL) 3-4: its two-channel instantiation keeps
// three offsets in flight and needs ~150 registers - pinned to 3 wavefronts per SIMD (168) instead of the 241 the scheduler
// takes when left alone.
template <int CH, bool LUT_LDS, bool XL = false>
__global__ void __launch_bounds__(256) k_nlm_y(PageSet src, PageSetOut dst, NlmParams np)
{
    nlm_y_body<CH, LUT_LDS, false>(src, dst, np);
}
template <int CH, bool LUT_LDS>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) k_nlm_y_xl(PageSet src, PageSetOut dst, NlmParams np)
{
    nlm_y_body<CH, LUT_LDS, true>(src, dst, np);
}

// ---- generic kernel (3 interleaved channels; fastNlMeansDenoising on a colour image, not on the prl::denoise
// path): one byte tile, neighbourhoods by aligned dword reads + funnel shifts --------------------------------
template <int CH, bool LUT_LDS>
__global__ void __launch_bounds__(256) k_nlm(PageSet src, PageSetOut dst, NlmParams np)
{
    constexpr int NB = Nb<CH>::NB;
    constexpr int PITCH = (EXT_W * CH + 16 + 3) / 4 * 4;  // bytes per staged row (+16: the dword reads overrun the row)
    __shared__ __attribute__((aligned(16))) unsigned char tile[EXT_H * PITCH + 32];
    __shared__ int sb[SB_H * SB_W];
    __shared__ int lut_s[LUT_LDS ? kLutMax + 1 : 1];

    const int page = blockIdx.z;
    const uint8_t* __restrict__ img = src.page(page);
    uint8_t* __restrict__ out = dst.page(page);
    const int W = np.width, H = np.height;
    const int x0 = blockIdx.x * TILE_W, y0 = blockIdx.y * TILE_H;

    for (int i = threadIdx.x; i < EXT_H * EXT_W; i += blockDim.x) {
        const int r = i / EXT_W, c = i - r * EXT_W;
        const int sy = reflect101(y0 - kBorder + r, H), sx = reflect101(x0 - kBorder + c, W);
        const uint8_t* s = img + (size_t)sy * src.step + (size_t)sx * CH;
#pragma unroll
        for (int k = 0; k < CH; ++k) tile[r * PITCH + c * CH + k] = s[k];
    }
    for (int i = threadIdx.x; i < EXT_H * (PITCH - EXT_W * CH); i += blockDim.x) {
        const int r = i / (PITCH - EXT_W * CH), c = i - r * (PITCH - EXT_W * CH);
        tile[r * PITCH + EXT_W * CH + c] = 0;
    }
    for (int i = threadIdx.x; i < 32; i += blockDim.x) tile[EXT_H * PITCH + i] = 0;
    if (LUT_LDS)
        for (int i = threadIdx.x; i <= kLutMax; i += blockDim.x) lut_s[i] = (i < np.n_lut) ? np.lut[i] : 0;
    __syncthreads();

    constexpr unsigned kPadBytes = NB * 4 - 7 * CH;
    constexpr unsigned kLastMask = kPadBytes == 0 ? 0xffffffffu : (0xffffffffu >> (8 * kPadBytes));
    {
        constexpr int CHR = (SB_H + 2) / 3;
        if (threadIdx.x < 3 * SB_W) {
            const int c = threadIdx.x % SB_W, rb = (threadIdx.x / SB_W) * CHR;
            const unsigned char* p = tile + c * CH;
            const unsigned sh = (unsigned)(p - tile) & 3u;
            const unsigned* pw = reinterpret_cast<const unsigned*>(p - sh);
            unsigned ring[kT];
            unsigned acc = 0;
#pragma unroll
            for (int i = 0; i < CHR + kT - 1; ++i) {
                const int rr = min(rb + i, EXT_H - 1);
                const Nb<CH> a = lds_nb<CH>(pw + rr * (PITCH / 4), sh);
                Nb<CH> m = a;
                m.d[NB - 1] &= kLastMask;
                const unsigned hd = dot_nb<CH>(m, a, 0u);
                acc += hd;
                if (i >= kT) acc -= ring[i % kT];
                ring[i % kT] = hd;
                if (i >= kT - 1 && rb + i - (kT - 1) < SB_H) sb[(rb + i - (kT - 1)) * SB_W + c] = (int)acc;
            }
        }
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int oy0 = wv * ROWS;
    const int aoff = (oy0 + kSH) * PITCH + (lane + kSH) * CH;  // byte offset of template row 0 of output row 0
    const int* sa_base = sb + (oy0 + kSH) * SB_W + (lane + kSH);
    const unsigned ash = (unsigned)aoff & 3u;
    const unsigned* aword = reinterpret_cast<const unsigned*>(tile + (aoff - (int)ash));

    int SA[ROWS];
#pragma unroll
    for (int i = 0; i < ROWS; ++i) SA[i] = sa_base[i * SB_W];
    unsigned est[ROWS][CH], wsum[ROWS];
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
        wsum[i] = 0;
#pragma unroll
        for (int k = 0; k < CH; ++k) est[i][k] = 0;
    }

#pragma unroll 1
    for (int o = 0; o < kS * kS; ++o) {
        const int dy = o / kS - kSH, dx = o - (o / kS) * kS - kSH;
        const int boff = aoff + dy * PITCH + dx * CH;
        const unsigned bsh = (unsigned)boff & 3u;
        const unsigned* bword = reinterpret_cast<const unsigned*>(tile + (boff - (int)bsh));
        const int* sb_o = sa_base + dy * SB_W + dx;
        unsigned ring[kT];
        unsigned qring[4][CH];
        unsigned AB = 0;
#pragma unroll
        for (int r = 0; r < ROWS + kT - 1; ++r) {
            Nb<CH> a = lds_nb<CH>(aword + r * (PITCH / 4), ash);
            a.d[NB - 1] &= kLastMask;
            const Nb<CH> b = lds_nb<CH>(bword + r * (PITCH / 4), bsh);
            const unsigned hd = dot_nb<CH>(a, b, 0u);
            AB += hd;
            if (r >= kT) AB -= ring[r % kT];
            ring[r % kT] = hd;
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                const int byte = 3 * CH + k;
                qring[r % 4][k] = (b.d[byte / 4] >> (8 * (byte % 4))) & 0xffu;
            }
            if (r >= kT - 1) {
                const int i = r - (kT - 1);
                const int D = SA[i] + sb_o[i * SB_W] - 2 * (int)AB;
                int idx = D >> 6;
                idx = min(idx, np.n_lut);
                const unsigned wgt = (unsigned)(LUT_LDS ? lut_s[idx] : np.lut[idx]);
                wsum[i] += wgt;
#pragma unroll
                for (int k = 0; k < CH; ++k) est[i][k] += __umul24(wgt, qring[(r - 3) % 4][k]);
            }
        }
    }

    const int gx = x0 + lane;
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
        const int gy = y0 + oy0 + i;
        if (gx < W && gy < H) {
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                const unsigned v = (est[i][k] + wsum[i] / 2u) / wsum[i];
                out[(size_t)gy * dst.step + (size_t)gx * CH + k] = (uint8_t)(v > 255u ? 255u : v);
            }
        }
    }
}

// ---- 8-bit LBGR <-> Lab (cv::cvtColor COLOR_LBGR2Lab / COLOR_Lab2LBGR) [upstream, SURVEY.md Appendix C]
constexpr int kLabShift = 12, kGammaShift = 3, kLabShift2 = kLabShift + kGammaShift;
constexpr int kCbrtTabSize = 256 * 3 / 2 * (1 << kGammaShift);

struct LabTables {
    int fwd[9];
    float inv[9];
    const unsigned short* cbrt_tab;  // device, kCbrtTabSize entries
};

__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }
__device__ __forceinline__ unsigned char sat8(int v) { return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

// BGR(A) -> L plane + interleaved ab plane (the mixChannels split is fused into the conversion)
__global__ void __launch_bounds__(256) k_lbgr2lab(PageSet src, int channels, int width, int height,
                                                 LabTables lt, uint8_t* __restrict__ lpl,
                                                 uint8_t* __restrict__ abpl, size_t plane_px)
{
    const int page = blockIdx.z;
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= width) return;
    const uint8_t* s = src.page(page) + (size_t)y * src.step + (size_t)x * channels;
    const int R = s[0] << kGammaShift, G = s[1] << kGammaShift, B = s[2] << kGammaShift;
    const int* C = lt.fwd;
    const int fX = lt.cbrt_tab[descale(R * C[0] + G * C[1] + B * C[2], kLabShift)];
    const int fY = lt.cbrt_tab[descale(R * C[3] + G * C[4] + B * C[5], kLabShift)];
    const int fZ = lt.cbrt_tab[descale(R * C[6] + G * C[7] + B * C[8], kLabShift)];
    const int Lscale = (116 * 255 + 50) / 100;
    const int Lshift = -((16 * 255 * (1 << kLabShift2) + 50) / 100);
    const size_t i = (size_t)page * plane_px + (size_t)y * width + x;
    lpl[i] = sat8(descale(Lscale * fY + Lshift, kLabShift2));
    abpl[2 * i] = sat8(descale(500 * (fX - fY) + 128 * (1 << kLabShift2), kLabShift2));
    abpl[2 * i + 1] = sat8(descale(200 * (fY - fZ) + 128 * (1 << kLabShift2), kLabShift2));
}

__device__ __forceinline__ float clip01(float v) { return v < 0.f ? 0.f : (v > 1.f ? 1.f : v); }

// L plane + ab plane -> BGR(A); float path of Lab2RGB_f with gamma disabled, one rounding per operation
__global__ void __launch_bounds__(256) k_lab2lbgr(const uint8_t* __restrict__ lpl,
                                                 const uint8_t* __restrict__ abpl, size_t plane_px,
                                                 int channels, int width, int height, LabTables lt,
                                                 PageSetOut dst)
{
    const int page = blockIdx.z;
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= width) return;
    const size_t i = (size_t)page * plane_px + (size_t)y * width + x;
    const float lThresh = 0.008856f * 903.3f;
    const float fThresh = 7.787f * 0.008856f + 16.0f / 116.0f;
    const float li = lpl[i] * (100.f / 255.f);
    const float ai = (float)((int)abpl[2 * i] - 128);
    const float bi = (float)((int)abpl[2 * i + 1] - 128);
    float Y, fy;
    if (li <= lThresh) {
        Y = li / 903.3f;
        fy = 7.787f * Y + 16.0f / 116.0f;
    } else {
        fy = (li + 16.0f) / 116.0f;
        Y = fy * fy * fy;
    }
    float fx = ai / 500.0f + fy, fz = fy - bi / 200.0f;
    fx = (fx <= fThresh) ? (fx - 16.0f / 116.0f) / 7.787f : fx * fx * fx;
    fz = (fz <= fThresh) ? (fz - 16.0f / 116.0f) / 7.787f : fz * fz * fz;
    const float* C = lt.inv;
    float c0 = C[0] * fx + C[1] * Y + C[2] * fz;
    float c1 = C[3] * fx + C[4] * Y + C[5] * fz;
    float c2 = C[6] * fx + C[7] * Y + C[8] * fz;
    c0 = clip01(c0);
    c1 = clip01(c1);
    c2 = clip01(c2);
    uint8_t* d = dst.page(page) + (size_t)y * dst.step + (size_t)x * channels;
    d[0] = sat8((int)rintf(c0 * 255.f));
    d[1] = sat8((int)rintf(c1 * 255.f));
    d[2] = sat8((int)rintf(c2 * 255.f));
    if (channels == 4) d[3] = 255;
}

// ---- host side -----------------------------------------------------------------------------------

// almost_dist2weight_ of FastNlMeansDenoisingInvoker<..., DistSquared, int> (host, once per call)
std::vector<int> build_weights(int channels, float h)
{
    const int max_estimate_sum_value = kS * kS * 255;
    const int fixed_point_mult = INT_MAX / max_estimate_sum_value;  // 19096
    const int tw_sq = kT * kT;
    int bin_shift = 0;
    while ((1 << bin_shift) < tw_sq) ++bin_shift;
    const double mult = ((double)(1 << bin_shift)) / tw_sq;
    const int max_dist = 255 * 255 * channels;
    const int almost_max_dist = (int)(max_dist / mult + 1);
    const float hh = h * h * channels;  // float arithmetic, as upstream
    std::vector<int> lut((size_t)almost_max_dist);
    for (int i = 0; i < almost_max_dist; ++i) {
        const double dist = i * mult;
        double w = std::exp(-dist / hh);
        if (w != w) w = 1.0;
        int weight = (int)std::nearbyint(fixed_point_mult * w);
        if (weight < 0.001 * fixed_point_mult) weight = 0;
        lut[(size_t)i] = weight;
    }
    return lut;
}

struct HostLab {
    std::vector<unsigned short> cbrt_tab;
    int fwd[9];
    float inv[9];
};

const HostLab& host_lab()
{
    static HostLab t = [] {
        HostLab h;
        static const float sRGB2XYZ_D65[9] = {0.412453f, 0.357580f, 0.180423f, 0.212671f, 0.715160f,
                                              0.072169f, 0.019334f, 0.119193f, 0.950227f};
        static const float XYZ2sRGB_D65[9] = {3.240479f, -1.53715f, -0.498535f, -0.969256f, 1.875991f,
                                              0.041556f, 0.055648f, -0.204043f, 1.057311f};
        static const float D65[3] = {0.950456f, 1.f, 1.088754f};
        h.cbrt_tab.resize(kCbrtTabSize);
        for (int i = 0; i < kCbrtTabSize; ++i) {
            const float x = i * (1.f / (255.f * (1 << kGammaShift)));
            const float v = (1 << kLabShift2) * (x < 0.008856f ? x * 7.787f + 0.13793103448275862f : cbrtf(x));
            const int iv = (int)std::nearbyint((double)v);
            h.cbrt_tab[(size_t)i] = (unsigned short)(iv < 0 ? 0 : (iv > 65535 ? 65535 : iv));
        }
        const float scale[3] = {(1 << kLabShift) / D65[0], (float)(1 << kLabShift), (1 << kLabShift) / D65[2]};
        for (int i = 0; i < 3; ++i) {  // blueIdx = 0
            h.fwd[i * 3 + 2] = (int)std::nearbyint((double)(sRGB2XYZ_D65[i * 3] * scale[i]));
            h.fwd[i * 3 + 1] = (int)std::nearbyint((double)(sRGB2XYZ_D65[i * 3 + 1] * scale[i]));
            h.fwd[i * 3 + 0] = (int)std::nearbyint((double)(sRGB2XYZ_D65[i * 3 + 2] * scale[i]));
            h.inv[i + 6] = XYZ2sRGB_D65[i] * D65[i];
            h.inv[i + 3] = XYZ2sRGB_D65[i + 3] * D65[i];
            h.inv[i + 0] = XYZ2sRGB_D65[i + 6] * D65[i];
        }
        return h;
    }();
    return t;
}

// Device copies of the LUT(s) / cbrt table live in the small workspace: [lutA 512 KiB][lutB 512 KiB][cbrt 8 KiB]
constexpr size_t kLutSlot = 512 * 1024;

int upload_lut(DeviceCtx* ctx, int slot, int channels, float h, hipStream_t stream, NlmParams* np)
{
    int* dcached = reinterpret_cast<int*>(static_cast<uint8_t*>(ctx->small) + (size_t)slot * kLutSlot);
    if (ctx->lut_small[slot] == ctx->small && ctx->lut_channels[slot] == channels && ctx->lut_h[slot] == h) {
        np->n_lut = ctx->lut_n[slot];  // same table as the previous call: still in the workspace (building it costs
        np->lut = dcached;             // 50-100 thousand exp() calls on the host, a visible share of a single page)
        return PRL_OK;
    }
    std::vector<int> lut = build_weights(channels, h);
    int n = (int)lut.size();
    while (n > 0 && lut[(size_t)n - 1] == 0) --n;  // weights are non-increasing: trailing zeros collapse
    lut.resize((size_t)n + 1);
    lut[(size_t)n] = 0;
    if ((size_t)(n + 1) * sizeof(int) > kLutSlot) {
        set_error_detail("NLM weight table too large for the workspace (h too large)");
        return PRL_ERR_BAD_ARG;
    }
    int* d = reinterpret_cast<int*>(static_cast<uint8_t*>(ctx->small) + (size_t)slot * kLutSlot);
    // pageable source: hipMemcpyAsync stages it before returning, so `lut` may die after this call
    PRL_HIP_CHECK(hipMemcpyAsync(d, lut.data(), (size_t)(n + 1) * sizeof(int), hipMemcpyHostToDevice, stream));
    PRL_HIP_CHECK(hipStreamSynchronize(stream));
    np->n_lut = n;
    np->lut = d;
    ctx->lut_small[slot] = ctx->small;
    ctx->lut_channels[slot] = channels;
    ctx->lut_h[slot] = h;
    ctx->lut_n[slot] = n;
    return PRL_OK;
}

template <int CH>
int launch_nlm(const PageSet& src_all, const PageSetOut& dst_all, int n_pages, const NlmParams& np, hipStream_t stream)
{
    const bool lds_lut = np.n_lut <= kLutMax;
    if ((np.height + TILE_H - 1) / TILE_H > 65535) return PRL_ERR_BAD_ARG;  // grid.y
    for (int first = 0; first < n_pages; first += 65535) {  // grid.z holds at most 65535 pages per launch
        const int cnt = std::min(65535, n_pages - first);
        PageSet src = src_all;
        PageSetOut dst = dst_all;
        if (src.table) src.table += first; else src.base += (size_t)first * src.page_stride;
        if (dst.table) dst.table += first; else dst.base += (size_t)first * dst.page_stride;
        const dim3 grid((np.width + TILE_W - 1) / TILE_W, (np.height + TILE_H - 1) / TILE_H, cnt);
        if (CH <= 2) {
            constexpr int C = CH <= 2 ? CH : 1;
            const int xl_env = env_knobs().nlm_xl;
            const bool xl = np.n_lut <= kLutMaxXL && ((xl_env >> (C - 1)) & 1);
            const bool glut = (env_knobs().nlm_glut >> (C - 1)) & 1;  // weight table from memory (L1) instead of LDS
            if (xl && glut) hipLaunchKernelGGL((k_nlm_y_xl<C, false>), grid, dim3(256), 0, stream, src, dst, np);
            else if (xl) hipLaunchKernelGGL((k_nlm_y_xl<C, true>), grid, dim3(256), 0, stream, src, dst, np);
            else if (lds_lut) hipLaunchKernelGGL((k_nlm_y<C, true>), grid, dim3(256), 0, stream, src, dst, np);
            else hipLaunchKernelGGL((k_nlm_y<C, false>), grid, dim3(256), 0, stream, src, dst, np);
        } else {
            constexpr int C = CH > 2 ? CH : 3;
            if (lds_lut) hipLaunchKernelGGL((k_nlm<C, true>), grid, dim3(256), 0, stream, src, dst, np);
            else hipLaunchKernelGGL((k_nlm<C, false>), grid, dim3(256), 0, stream, src, dst, np);
        }
        PRL_HIP_CHECK(hipGetLastError());
    }
    return PRL_OK;
}

int nlm_planes_locked(DeviceCtx* ctx, int slot, int n_pages, int channels, float h, const PageSet& src, int width,
                      int height, const PageSetOut& dst, hipStream_t stream)
{
    NlmParams np{};
    np.width = width;
    np.height = height;
    int st = upload_lut(ctx, slot, channels, h, stream, &np);
    if (st != PRL_OK) return st;
    switch (channels) {
    case 1: return launch_nlm<1>(src, dst, n_pages, np, stream);
    case 2: return launch_nlm<2>(src, dst, n_pages, np, stream);
    case 3: return launch_nlm<3>(src, dst, n_pages, np, stream);
    default: return PRL_ERR_BAD_CHANNELS;
    }
}

}  // namespace
}  // namespace prl_hip

namespace prl_hip {
// prl::denoise in its three parts, on `cnt` pages whose planes live at `planes` (denoise_plane_bytes per page: L, ab, L', ab' -
// 1 + 2 + 1 + 2 bytes per pixel, all pages' L first, then all ab, ...).  The chain runs the first and the last part (streaming
// kernels) away from the angle search of the next pass and only the middle one beside it (glue.hip).
size_t denoise_plane_bytes(int width, int height) { return 6 * (size_t)width * height; }

static int lab_tables(DeviceCtx* ctx, hipStream_t s, LabTables* lt)
{
    const size_t cbrt_off = 2 * kLutSlot;
    int st = ensure_small(ctx, cbrt_off + 64 * 1024);
    if (st != PRL_OK) return st;
    const HostLab& hl = host_lab();
    auto* d_cbrt = reinterpret_cast<unsigned short*>(static_cast<uint8_t*>(ctx->small) + cbrt_off);
    PRL_HIP_CHECK(hipMemcpyAsync(d_cbrt, hl.cbrt_tab.data(), hl.cbrt_tab.size() * sizeof(unsigned short), hipMemcpyHostToDevice, s));
    std::copy(hl.fwd, hl.fwd + 9, lt->fwd);
    std::copy(hl.inv, hl.inv + 9, lt->inv);
    lt->cbrt_tab = d_cbrt;
    return PRL_OK;
}

static int convert_in_locked(DeviceCtx* ctx, int cnt, int channels, const uint8_t* src, size_t src_page_stride, size_t src_step, int width,
                             int height, uint8_t* planes, hipStream_t s)
{
    if (ctx->last_use) PRL_HIP_CHECK(hipStreamWaitEvent(s, ctx->last_use, 0));
    else PRL_HIP_CHECK(hipEventCreateWithFlags(&ctx->last_use, hipEventDisableTiming));
    LabTables lt{};
    int st = lab_tables(ctx, s, &lt);
    if (st != PRL_OK) return st;
    const size_t px = (size_t)width * height;
    uint8_t* L = planes;
    uint8_t* AB = L + px * (size_t)cnt;
    PageSet ps{};
    ps.base = src; ps.page_stride = src_page_stride; ps.step = src_step;
    const dim3 grid((width + 255) / 256, height, cnt);
    hipLaunchKernelGGL(k_lbgr2lab, grid, dim3(256), 0, s, ps, channels, width, height, lt, L, AB, px);
    PRL_HIP_CHECK(hipGetLastError());
    PRL_HIP_CHECK(hipEventRecord(ctx->last_use, s));
    return PRL_OK;
}

static int nlm_locked(DeviceCtx* ctx, int cnt, float strength, uint8_t* planes, int width, int height, hipStream_t s)
{
    if (ctx->last_use) PRL_HIP_CHECK(hipStreamWaitEvent(s, ctx->last_use, 0));
    else PRL_HIP_CHECK(hipEventCreateWithFlags(&ctx->last_use, hipEventDisableTiming));
    const size_t px = (size_t)width * height;
    uint8_t* L = planes;
    uint8_t* AB = L + px * (size_t)cnt;
    uint8_t* L2 = AB + 2 * px * (size_t)cnt;
    uint8_t* AB2 = L2 + px * (size_t)cnt;
    PageSet sl{}, sab{};
    sl.base = L; sl.page_stride = px; sl.step = (size_t)width;
    sab.base = AB; sab.page_stride = 2 * px; sab.step = 2 * (size_t)width;
    PageSetOut dl{}, dab{};
    dl.base = L2; dl.page_stride = px; dl.step = (size_t)width;
    dab.base = AB2; dab.page_stride = 2 * px; dab.step = 2 * (size_t)width;
    int st = nlm_planes_locked(ctx, 0, cnt, 1, strength, sl, width, height, dl, s);
    if (st != PRL_OK) return st;
    st = nlm_planes_locked(ctx, 1, cnt, 2, 3.0f, sab, width, height, dab, s);  // hForColorComponents = 3
    if (st != PRL_OK) return st;
    PRL_HIP_CHECK(hipEventRecord(ctx->last_use, s));
    return PRL_OK;
}

static int convert_out_locked(DeviceCtx* ctx, int cnt, int channels, const uint8_t* planes, int width, int height, uint8_t* dst,
                              size_t dst_page_stride, size_t dst_step, hipStream_t s)
{
    if (ctx->last_use) PRL_HIP_CHECK(hipStreamWaitEvent(s, ctx->last_use, 0));
    else PRL_HIP_CHECK(hipEventCreateWithFlags(&ctx->last_use, hipEventDisableTiming));
    LabTables lt{};
    int st = lab_tables(ctx, s, &lt);
    if (st != PRL_OK) return st;
    const size_t px = (size_t)width * height;
    const uint8_t* L2 = planes + 3 * px * (size_t)cnt;
    const uint8_t* AB2 = L2 + px * (size_t)cnt;
    PageSetOut pd{};
    pd.base = dst; pd.page_stride = dst_page_stride; pd.step = dst_step;
    const dim3 grid((width + 255) / 256, height, cnt);
    hipLaunchKernelGGL(k_lab2lbgr, grid, dim3(256), 0, s, L2, AB2, px, channels, width, height, lt, pd);
    PRL_HIP_CHECK(hipGetLastError());
    PRL_HIP_CHECK(hipEventRecord(ctx->last_use, s));
    return PRL_OK;
}

int denoise_convert_in(DeviceCtx* ctx, int cnt, int channels, const uint8_t* src, size_t src_page_stride, size_t src_step, int width,
                       int height, uint8_t* planes, hipStream_t s)
{
    std::lock_guard<std::mutex> lk(ctx->mu);
    return convert_in_locked(ctx, cnt, channels, src, src_page_stride, src_step, width, height, planes, s);
}
int denoise_nlm(DeviceCtx* ctx, int cnt, float strength, uint8_t* planes, int width, int height, hipStream_t s)
{
    std::lock_guard<std::mutex> lk(ctx->mu);
    return nlm_locked(ctx, cnt, strength, planes, width, height, s);
}
int denoise_convert_out(DeviceCtx* ctx, int cnt, int channels, const uint8_t* planes, int width, int height, uint8_t* dst,
                        size_t dst_page_stride, size_t dst_step, hipStream_t s)
{
    std::lock_guard<std::mutex> lk(ctx->mu);
    return convert_out_locked(ctx, cnt, channels, planes, width, height, dst, dst_page_stride, dst_step, s);
}
// the whole stage on pages [0, cnt) with the planes in the shared scratch area: one lock over all three parts
int denoise_all_locked(DeviceCtx* ctx, int cnt, int channels, float strength, const uint8_t* src, size_t src_page_stride, size_t src_step,
                       int width, int height, uint8_t* dst, size_t dst_page_stride, size_t dst_step, uint8_t* planes, hipStream_t s)
{
    int st = convert_in_locked(ctx, cnt, channels, src, src_page_stride, src_step, width, height, planes, s);
    if (st == PRL_OK) st = nlm_locked(ctx, cnt, strength, planes, width, height, s);
    if (st == PRL_OK) st = convert_out_locked(ctx, cnt, channels, planes, width, height, dst, dst_page_stride, dst_step, s);
    return st;
}
}  // namespace prl_hip

using namespace prl_hip;

extern "C" {

int prl_hip_nlm_planes_device(int n_pages, int channels, float h, const uint8_t* d_src, size_t src_page_stride,
                              size_t src_step, int width, int height, uint8_t* d_dst, size_t dst_page_stride,
                              size_t dst_step, void* stream)
{
    if (width <= 0 || height <= 0) return PRL_ERR_EMPTY;
    if (channels < 1 || channels > 3) return PRL_ERR_BAD_CHANNELS;
    if (n_pages < 0 || !d_src || !d_dst || d_src == d_dst) return PRL_ERR_BAD_ARG;
    if (src_step < (size_t)width * channels || dst_step < (size_t)width * channels) return PRL_ERR_BAD_ARG;
    if (n_pages == 0) return PRL_OK;
    int dev;
    int st = current_device(&dev);
    if (st != PRL_OK) return st;
    DeviceCtx* ctx = device_ctx(dev);
    std::lock_guard<std::mutex> lk(ctx->mu);
    st = ensure_small(ctx, 2 * kLutSlot + 64 * 1024);
    if (st != PRL_OK) return st;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (ctx->last_use) PRL_HIP_CHECK(hipStreamWaitEvent(s, ctx->last_use, 0));
    else PRL_HIP_CHECK(hipEventCreateWithFlags(&ctx->last_use, hipEventDisableTiming));
    PageSet ps{};
    ps.base = d_src;
    ps.page_stride = src_page_stride;
    ps.step = src_step;
    PageSetOut pd{};
    pd.base = d_dst;
    pd.page_stride = dst_page_stride;
    pd.step = dst_step;
    st = nlm_planes_locked(ctx, 0, n_pages, channels, h, ps, width, height, pd, s);
    if (st != PRL_OK) return st;
    PRL_HIP_CHECK(hipEventRecord(ctx->last_use, s));
    return PRL_OK;
}

int prl_hip_denoise_batch_device(int n_pages, int channels, float strength, const uint8_t* d_src,
                                 size_t src_page_stride, size_t src_step, int width, int height, uint8_t* d_dst,
                                 size_t dst_page_stride, size_t dst_step, void* stream)
{
    if (width <= 0 || height <= 0) return PRL_ERR_EMPTY;
    // "Type of input image should be CV_8UC3 or CV_8UC4!" [upstream fastNlMeansDenoisingColored]
    if (channels != 3 && channels != 4) return PRL_ERR_BAD_CHANNELS;
    if (n_pages < 0 || !d_src || !d_dst) return PRL_ERR_BAD_ARG;
    if (src_step < (size_t)width * channels || dst_step < (size_t)width * channels) return PRL_ERR_BAD_ARG;
    if (n_pages == 0) return PRL_OK;
    if (height > 65535) return PRL_ERR_BAD_ARG;  // the colour conversions use one grid row per image row
    int dev;
    int st = current_device(&dev);
    if (st != PRL_OK) return st;
    DeviceCtx* ctx = device_ctx(dev);
    // planes: L, ab, L', ab'  (1 + 2 + 1 + 2 bytes per pixel), processed in page chunks of bounded size
    const size_t px = (size_t)width * height;
    const size_t budget = (size_t)2 << 30;
    int chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_pages, budget / (6 * px)));
    chunk = std::min(chunk, 65535);  // grid.z of the per-page kernels
    std::lock_guard<std::mutex> lk(ctx->mu);
    st = ensure_scratch(ctx, 6 * px * (size_t)chunk);
    if (st != PRL_OK) return st;
    auto* base = static_cast<uint8_t*>(ctx->scratch);
    for (int first = 0; first < n_pages; first += chunk) {
        const int cnt = std::min(chunk, n_pages - first);
        st = denoise_all_locked(ctx, cnt, channels, strength, d_src + (size_t)first * src_page_stride, src_page_stride, src_step, width, height,
                                d_dst + (size_t)first * dst_page_stride, dst_page_stride, dst_step, base, static_cast<hipStream_t>(stream));
        if (st != PRL_OK) return st;
    }
    return PRL_OK;
}

int prl_hip_denoise_host(int channels, float strength, const uint8_t* src, size_t src_step, int width, int height,
                         uint8_t* dst, size_t dst_step)
{
    if (width <= 0 || height <= 0) return PRL_ERR_EMPTY;
    if (channels != 3 && channels != 4) return PRL_ERR_BAD_CHANNELS;
    if (!src || !dst || src_step < (size_t)width * channels || dst_step < (size_t)width * channels) return PRL_ERR_BAD_ARG;
    int dev;
    int st = current_device(&dev);
    if (st != PRL_OK) return st;
    const size_t row = (size_t)width * channels;
    const size_t bytes = (row * (size_t)height + 255) / 256 * 256;
    DeviceCtx* ctx = device_ctx(dev);
    std::lock_guard<std::mutex> slk(ctx->stage_mu);  // cached device + pinned staging (lock order: stage_mu, then mu)
    st = ensure_stage(ctx, 2 * bytes);
    if (st != PRL_OK) return st;
    st = ensure_stage_pinned(ctx, 2 * bytes);
    if (st != PRL_OK) return st;
    uint8_t* d_in = static_cast<uint8_t*>(ctx->stage);
    uint8_t* d_out = d_in + bytes;
    DrainOnExit drain_guard{nullptr};   // (direct DMA from the caller's pinned page: see prl_internal.h)
    st = stage_upload(ctx, 0, src, src_step, row, height, d_in, nullptr);
    if (st != PRL_OK) return st;
    st = prl_hip_denoise_batch_device(1, channels, strength, d_in, bytes, row, width, height, d_out, bytes, row, nullptr);
    if (st != PRL_OK) return st;
    return stage_download(ctx, bytes, d_out, row, height, dst, dst_step, nullptr);
}

}  // extern "C"
