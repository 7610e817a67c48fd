// morph.hip — the cv::dilate / cv::erode pair at the end of every binarizer
// (src/binarizations/binarizeSauvola.cpp:125-134):
//   n > 0 : dilate(n) then erode(n)   (closing)
//   n < 0 : erode(|n|) then dilate(|n|) (opening)
// [upstream] cv::dilate(src, dst, Mat(), Point(-1,-1), n) = 3x3 rectangle iterated n times, which
// OpenCV folds into one (2n+1)x(2n+1) rectangle; the default border value makes out-of-image pixels
// neutral (ignored) for both operations.
//
// Binary masks (the pipeline's own threshold output) go through k_morph_bits: both operators on bit-packed rows held
// in registers, one read and one write of the mask (2 B/px, HBM-bound; see the kernel's comment).  k_morph_stream /
// k_morph_binary are the older byte-dword forms kept for radius 5..8 and unaligned sources; k_morph / k_rect are the
// byte-wise max/min kernels behind the public prl_hip_morph_batch_device entry (any 8-bit image).
#include "prl_internal.h"

namespace prl_hip {

namespace {

constexpr int TW = 64;
constexpr int TH = 32;
constexpr int kMaxN = 8;  // halo 2n <= 16 ; larger n falls back to iterated launches of n<=8

template <bool TAKE_MAX>
__device__ __forceinline__ unsigned char mm(unsigned char a, unsigned char b)
{
    return TAKE_MAX ? (a > b ? a : b) : (a < b ? a : b);
}

// dst(rows x cols) = horizontal rect of radius n over src; src has cols + 2n columns.
template <bool TAKE_MAX>
__device__ __forceinline__ void row_pass(const unsigned char* src, int src_pitch, unsigned char* dst,
                                         int dst_pitch, int rows, int cols, int n)
{
    for (int i = threadIdx.x; i < rows * cols; i += blockDim.x) {
        const int r = i / cols, c = i - r * cols;
        const unsigned char* s = src + r * src_pitch + c;
        unsigned char v = s[0];
        for (int j = 1; j <= 2 * n; ++j) v = mm<TAKE_MAX>(v, s[j]);
        dst[r * dst_pitch + c] = v;
    }
}

// dst(rows x cols) = vertical rect of radius n over src; src has rows + 2n rows.
template <bool TAKE_MAX>
__device__ __forceinline__ void col_pass(const unsigned char* src, int src_pitch, unsigned char* dst,
                                         int dst_pitch, int rows, int cols, int n)
{
    for (int i = threadIdx.x; i < rows * cols; i += blockDim.x) {
        const int r = i / cols, c = i - r * cols;
        const unsigned char* s = src + r * src_pitch + c;
        unsigned char v = s[0];
        for (int j = 1; j <= 2 * n; ++j) v = mm<TAKE_MAX>(v, s[j * src_pitch]);
        dst[r * dst_pitch + c] = v;
    }
}

// FIRST_MAX = true : closing (dilate, erode) ; false : opening (erode, dilate)
template <bool FIRST_MAX>
__global__ void __launch_bounds__(256) k_morph(PageSet src, PageSetOut dst, int width, int height, int n)
{
    constexpr int PITCH = TW + 4 * kMaxN + 4;  // bytes per LDS row
    __shared__ unsigned char bufA[(TH + 4 * kMaxN) * PITCH];
    __shared__ unsigned char bufB[(TH + 4 * kMaxN) * PITCH];

    const int page = blockIdx.z;
    const uint8_t* in = src.page(page);
    uint8_t* out = dst.page(page);
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const int h2 = 2 * n;

    // stage source tile with halo 2n; out-of-image pixels take the first operator's neutral value
    const unsigned char neutral1 = FIRST_MAX ? 0 : 255;
    const int sw = TW + 2 * h2, sh = TH + 2 * h2;
    for (int i = threadIdx.x; i < sw * sh; i += blockDim.x) {
        const int r = i / sw, c = i - r * sw;
        const int gy = y0 - h2 + r, gx = x0 - h2 + c;
        unsigned char v = neutral1;
        if (gy >= 0 && gy < height && gx >= 0 && gx < width) v = in[(size_t)gy * src.step + gx];
        bufA[r * PITCH + c] = v;
    }
    __syncthreads();

    // first operator on the tile + halo n
    const int mw = TW + 2 * n, mh = TH + 2 * n;
    row_pass<FIRST_MAX>(bufA, PITCH, bufB, PITCH, sh, mw, n);  // sh rows, mw cols
    __syncthreads();
    col_pass<FIRST_MAX>(bufB, PITCH, bufA, PITCH, mh, mw, n);  // mh rows, mw cols
    __syncthreads();

    // positions outside the image do not exist for the second operator: make them neutral for it
    const unsigned char neutral2 = FIRST_MAX ? 255 : 0;
    for (int i = threadIdx.x; i < mw * mh; i += blockDim.x) {
        const int r = i / mw, c = i - r * mw;
        const int gy = y0 - n + r, gx = x0 - n + c;
        if (gy < 0 || gy >= height || gx < 0 || gx >= width) bufA[r * PITCH + c] = neutral2;
    }
    __syncthreads();

    row_pass<!FIRST_MAX>(bufA, PITCH, bufB, PITCH, mh, TW, n);
    __syncthreads();
    col_pass<!FIRST_MAX>(bufB, PITCH, bufA, PITCH, TH, TW, n);
    __syncthreads();

    for (int i = threadIdx.x; i < TW * TH; i += blockDim.x) {
        const int r = i / TW, c = i - r * TW;
        const int gy = y0 + r, gx = x0 + c;
        if (gy < height && gx < width) out[(size_t)gy * dst.step + gx] = bufA[r * PITCH + c];
    }
}


// One rectangle operator (max or min) of radius n <= kMaxN on any 8-bit image; out-of-image pixels ignored.
// Rectangles compose (radius a then radius b = radius a+b, also with the ignore-outside rule), which is how
// radii above kMaxN are built: cv::dilate/erode with iterations > 8.
template <bool TAKE_MAX>
__global__ void __launch_bounds__(256) k_rect(PageSet src, PageSetOut dst, int width, int height, int n)
{
    constexpr int PITCH = TW + 2 * kMaxN + 4;
    __shared__ unsigned char bufA[(TH + 2 * kMaxN) * PITCH];
    __shared__ unsigned char bufB[(TH + 2 * kMaxN) * PITCH];
    const int page = blockIdx.z;
    const uint8_t* in = src.page(page);
    uint8_t* out = dst.page(page);
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const unsigned char neutral = TAKE_MAX ? 0 : 255;
    const int sw = TW + 2 * n, sh = TH + 2 * n;
    for (int i = threadIdx.x; i < sw * sh; i += blockDim.x) {
        const int r = i / sw, c = i - r * sw;
        const int gy = y0 - n + r, gx = x0 - n + c;
        unsigned char v = neutral;
        if (gy >= 0 && gy < height && gx >= 0 && gx < width) v = in[(size_t)gy * src.step + gx];
        bufA[r * PITCH + c] = v;
    }
    __syncthreads();
    row_pass<TAKE_MAX>(bufA, PITCH, bufB, PITCH, sh, TW, n);
    __syncthreads();
    col_pass<TAKE_MAX>(bufB, PITCH, bufA, PITCH, TH, TW, n);
    __syncthreads();
    for (int i = threadIdx.x; i < TW * TH; i += blockDim.x) {
        const int r = i / TW, c = i - r * TW;
        const int gy = y0 + r, gx = x0 + c;
        if (gy < height && gx < width) out[(size_t)gy * dst.step + gx] = bufA[r * PITCH + c];
    }
}

// ---- binary masks: 4 pixels per operation ------------------------------------------------------------
// The binarizers only ever emit 0 / 255, and for such bytes max == bitwise OR and min == bitwise AND, so a
// dword carries 4 pixels through every step.  The horizontal pass ORs/ANDs 2n+1 dwords read from LDS at
// consecutive BYTE offsets (unaligned ds_read_b32), the vertical pass 2n+1 row-aligned dwords.
constexpr int BND = 64;   // staged dwords per row = one wavefront; output tile width = 256 - 2*halo pixels
constexpr int BTH = 32;   // output tile height

template <bool IS_OR> __device__ __forceinline__ unsigned comb(unsigned a, unsigned b) { return IS_OR ? (a | b) : (a & b); }

// One wavefront per row, lane = dword.  Horizontal rectangle: dwords 2..61 (the outer two dwords on each
// side are halo that is never consumed).
template <bool IS_OR>
__device__ __forceinline__ void brow_pass(const unsigned char* src, unsigned char* dst, int pitch, int r_begin, int r_end,
                                          int n, int lane, int wv)
{
    if (lane < 2 || lane >= BND - 2) return;
    for (int r = r_begin + wv; r < r_end; r += 4) {
        // five ALIGNED dwords around this lane's pixels; the 2n shifted views are funnel shifts of them
        // (unaligned ds_read is legal on gfx950 but measured several times slower than this)
        const unsigned* p = reinterpret_cast<const unsigned*>(src + r * pitch + 4 * lane);
        const unsigned p2 = p[-2], p1 = p[-1], c0 = p[0], n1 = p[1], n2 = p[2];
        const unsigned long long R = ((unsigned long long)n1 << 32) | c0, R2 = ((unsigned long long)n2 << 32) | n1;
        const unsigned long long L = ((unsigned long long)c0 << 32) | p1, L2 = ((unsigned long long)p1 << 32) | p2;
        unsigned v = c0;
        for (int k = 1; k <= n; ++k) {
            const unsigned right = k <= 4 ? (unsigned)(R >> (8 * k)) : (unsigned)(R2 >> (8 * (k - 4)));
            const unsigned left = k <= 4 ? (unsigned)(L >> (32 - 8 * k)) : (unsigned)(L2 >> (32 - 8 * (k - 4)));
            v = comb<IS_OR>(v, comb<IS_OR>(left, right));
        }
        *reinterpret_cast<unsigned*>(dst + r * pitch + 4 * lane) = v;
    }
}

// Vertical rectangle: output rows r_begin..r_end-1, each from source rows r-n..r+n.
template <bool IS_OR>
__device__ __forceinline__ void bcol_pass(const unsigned char* src, unsigned char* dst, int pitch, int r_begin, int r_end,
                                          int n, int lane, int wv)
{
    for (int r = r_begin + wv; r < r_end; r += 4) {
        const unsigned char* p = src + (r - n) * pitch + 4 * lane;
        unsigned v = *reinterpret_cast<const unsigned*>(p);
        for (int k = 1; k <= 2 * n; ++k) v = comb<IS_OR>(v, *reinterpret_cast<const unsigned*>(p + k * pitch));
        *reinterpret_cast<unsigned*>(dst + r * pitch + 4 * lane) = v;
    }
}

template <bool FIRST_OR>
__global__ void __launch_bounds__(256) k_morph_binary(PageSet src, PageSetOut dst, int width, int height, int n, int btw)
{
    constexpr int PITCH = BND * 4 + 16;             // bytes per staged row (+16: bank spread)
    constexpr int MAXROWS = BTH + 4 * kMaxN;        // 64
    __shared__ __attribute__((aligned(16))) unsigned char bufA[MAXROWS * PITCH];
    __shared__ __attribute__((aligned(16))) unsigned char bufB[MAXROWS * PITCH];

    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int page = blockIdx.z;
    const uint8_t* in = src.page(page);
    uint8_t* out = dst.page(page);
    const int h4 = (BND * 4 - btw) / 2;             // horizontal halo in pixels: >= 2n + 8, multiple of 4
    const int x0 = blockIdx.x * btw, y0 = blockIdx.y * BTH;
    const int rows = BTH + 4 * n;                   // staged rows: y0 - 2n .. y0 + BTH + 2n - 1
    const unsigned neutral1 = FIRST_OR ? 0u : 0xffffffffu, neutral2 = ~neutral1;

    // stage: dword loads where the dword lies inside the page row, bytes (or the neutral value) elsewhere
    const bool aligned_src = (((size_t)in | src.step) & 3) == 0;
    const int gx = x0 - h4 + 4 * lane;
    for (int r = wv; r < rows; r += 4) {
        const int gy = y0 - 2 * n + r;
        unsigned v = neutral1;
        if (gy >= 0 && gy < height) {
            const uint8_t* row = in + (size_t)gy * src.step;
            if (aligned_src && gx >= 0 && gx + 4 <= width) {
                v = *reinterpret_cast<const unsigned*>(row + gx);
            } else {
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    if (gx + b >= 0 && gx + b < width)
                        v = (v & ~(0xffu << (8 * b))) | ((unsigned)row[gx + b] << (8 * b));
            }
        }
        *reinterpret_cast<unsigned*>(bufA + r * PITCH + 4 * lane) = v;
    }
    __syncthreads();

    // first operator: rows (all staged rows), then columns (rows n .. rows-n-1)
    brow_pass<FIRST_OR>(bufA, bufB, PITCH, 0, rows, n, lane, wv);
    __syncthreads();
    bcol_pass<FIRST_OR>(bufB, bufA, PITCH, n, rows - n, n, lane, wv);
    __syncthreads();

    // pixels outside the page do not exist for the second operator: make them neutral for it
    const bool touches_border = (x0 - h4 < 0) || (x0 + btw + h4 > width) || (y0 - n < 0) || (y0 + BTH + n > height);
    if (touches_border) {
        for (int r = n + wv; r < rows - n; r += 4) {
            const int gy = y0 - 2 * n + r;
            unsigned* p = reinterpret_cast<unsigned*>(bufA + r * PITCH + 4 * lane);
            if (gy < 0 || gy >= height) {
                *p = neutral2;
            } else if (gx < 0 || gx + 4 > width) {
                unsigned v = *p;
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    if (gx + b < 0 || gx + b >= width) v = (v & ~(0xffu << (8 * b))) | (neutral2 & (0xffu << (8 * b)));
                *p = v;
            }
        }
        __syncthreads();
    }

    // second operator: rows n .. rows-n-1, then columns -> tile rows 2n .. 2n+BTH-1
    brow_pass<!FIRST_OR>(bufA, bufB, PITCH, n, rows - n, n, lane, wv);
    __syncthreads();
    bcol_pass<!FIRST_OR>(bufB, bufA, PITCH, 2 * n, 2 * n + BTH, n, lane, wv);
    __syncthreads();

    const bool aligned_dst = (((size_t)out | dst.step) & 3) == 0;
    const int ox = gx;  // this lane's pixels; inside the output tile iff h4 <= 4*lane < h4 + btw
    if (4 * lane >= h4 && 4 * lane < h4 + btw && ox < width) {
        for (int r = wv; r < BTH; r += 4) {
            const int gy = y0 + r;
            if (gy >= height) break;
            const unsigned v = *reinterpret_cast<const unsigned*>(bufA + (2 * n + r) * PITCH + 4 * lane);
            uint8_t* o = out + (size_t)gy * dst.step + ox;
            if (aligned_dst && ox + 4 <= width) {
                *reinterpret_cast<unsigned*>(o) = v;
            } else {
                for (int b = 0; b < 4 && ox + b < width; ++b) o[b] = (uint8_t)(v >> (8 * b));
            }
        }
    }
}


// ---- binary masks, streaming form (the default path) ---------------------------------------------------
// One wavefront walks down a strip of 64 dwords (256 pixels); no workgroup tiles, no barriers.  Per row:
//   horizontal rectangle from the lane's dword and its neighbours' (DPP wave shifts + funnel shifts),
//   vertical rectangle = OR/AND over the last 2n+1 horizontally filtered rows kept in a per-wavefront LDS ring;
// the second operator runs on the first one's rows n rows later, the output appears 2n rows behind the
// fetch.  Each page row is fetched once per strip (halo: 2 or 4 dwords per side).
extern __shared__ unsigned morph_ring_lds[];

template <bool IS_OR>
__device__ __forceinline__ unsigned hop(unsigned c0, int n)
{
    // neighbours' dwords: lane-1 (wave_shr:1) and lane+1 (wave_shl:1); a second hop for n > 4
    const unsigned p1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)c0, 0x138, 0xf, 0xf, true);
    const unsigned n1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)c0, 0x130, 0xf, 0xf, true);
    unsigned p2 = 0, n2 = 0;
    if (n > 4) {
        p2 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)p1, 0x138, 0xf, 0xf, true);
        n2 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)n1, 0x130, 0xf, 0xf, true);
    }
    // shifted views: v_alignbyte_b32 (one 4-cycle op each) instead of 64-bit shifts
    unsigned v = c0;
    const int m = n < 4 ? n : 4;
    for (int k = 1; k <= m; ++k) {
        const unsigned right = k == 4 ? n1 : __builtin_amdgcn_alignbyte(n1, c0, (unsigned)k);
        const unsigned left = k == 4 ? p1 : __builtin_amdgcn_alignbyte(c0, p1, (unsigned)(4 - k));
        v = comb<IS_OR>(v, comb<IS_OR>(left, right));
    }
    for (int k = 5; k <= n; ++k) {
        const unsigned right = k == 8 ? n2 : __builtin_amdgcn_alignbyte(n2, n1, (unsigned)(k - 4));
        const unsigned left = k == 8 ? p2 : __builtin_amdgcn_alignbyte(p1, p2, (unsigned)(8 - k));
        v = comb<IS_OR>(v, comb<IS_OR>(left, right));
    }
    return v;
}

template <bool FIRST_OR>
__global__ void __launch_bounds__(256) k_morph_stream(PageSet src, PageSetOut dst, int width, int height, int n,
                                                     int n_strips, int n_segs, int rows_per_seg, unsigned total_waves)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned wid = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + wv);
    if (wid >= total_waves) return;
    const int per_page = n_strips * n_segs;
    const int page = (int)(wid / (unsigned)per_page);
    const int rem = (int)(wid - (unsigned)page * (unsigned)per_page);
    const int seg = rem / n_strips, strip = rem - seg * n_strips;

    const int K = 2 * n + 1;                       // rows of each vertical rectangle
    const int hd = 2 * ((n + 3) / 4);              // halo dwords per side consumed by the two horizontal passes
    const int useful = 64 - 2 * hd;                // output dwords per strip
    unsigned* ring1 = morph_ring_lds + (size_t)wv * 2 * K * 64;
    unsigned* ring2 = ring1 + K * 64;

    const uint8_t* in = src.page(page);
    uint8_t* out = dst.page(page);
    const int gx = (strip * useful - hd + lane) * 4;            // first pixel of this lane's dword
    const int ys = seg * rows_per_seg, ye = min(ys + rows_per_seg, height);
    const unsigned neutral1 = FIRST_OR ? 0u : 0xffffffffu, neutral2 = ~neutral1;
    // bytes of this dword that lie inside the page row
    unsigned inside = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b)
        if (gx + b >= 0 && gx + b < width) inside |= 0xffu << (8 * b);
    const bool whole = inside == 0xffffffffu;
    const bool lane_out = lane >= hd && lane < 64 - hd && gx < width;
    const bool aligned_dst = (((size_t)out | dst.step) & 3) == 0;

    // fetch of page row r (source pointer and step are 4-byte aligned: checked by the host)
    auto fetch = [&](int r) -> unsigned {
        unsigned v = neutral1;
        if (r >= 0 && r < height && inside) {
            const uint8_t* row = in + (size_t)r * src.step;
            if (whole) {
                v = *reinterpret_cast<const unsigned*>(row + gx);
            } else {
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    if ((inside >> (8 * b)) & 1u) v = (v & ~(0xffu << (8 * b))) | ((unsigned)row[gx + b] << (8 * b));
            }
        }
        return v;
    };

    int s1 = 0, s2 = 0;  // ring slots written next
    unsigned va = fetch(ys - 2 * n), vb = fetch(ys - 2 * n + 1);  // two rows in flight ahead of the consumer
    for (int r = ys - 2 * n; r < ye + 2 * n; ++r) {
        const unsigned v = va;
        va = vb;
        vb = fetch(r + 2);
        ring1[s1 * 64 + lane] = hop<FIRST_OR>(v, n);
        s1 = (s1 + 1 == K) ? 0 : s1 + 1;
        const int rc = r - n;                      // centre row of the first operator's vertical window
        if (rc < ys - n) continue;                 // ring 1 not full yet
        unsigned v1 = ring1[lane];
        for (int k = 1; k < K; ++k) v1 = comb<FIRST_OR>(v1, ring1[k * 64 + lane]);
        // pixels outside the page do not exist for the second operator
        if (rc < 0 || rc >= height) v1 = neutral2;
        else v1 = (v1 & inside) | (neutral2 & ~inside);
        ring2[s2 * 64 + lane] = hop<!FIRST_OR>(v1, n);
        s2 = (s2 + 1 == K) ? 0 : s2 + 1;
        const int ro = r - 2 * n;                  // output row
        if (ro < ys) continue;                     // ring 2 not full yet
        unsigned o = ring2[lane];
        for (int k = 1; k < K; ++k) o = comb<!FIRST_OR>(o, ring2[k * 64 + lane]);
        if (lane_out) {
            uint8_t* op = out + (size_t)ro * dst.step + gx;
            if (aligned_dst && whole) {
                *reinterpret_cast<unsigned*>(op) = o;
            } else {
                for (int b = 0; b < 4 && gx + b < width; ++b) op[b] = (uint8_t)(o >> (8 * b));
            }
        }
    }
}

// ---- bit-domain streaming kernel (binary masks, radius <= 4: the reference's default is 2) --------------------
// The two rectangle operators run on BITS: horizontal neighbours by shifts against the words of lane -+ 1 (DPP),
// vertical windows as OR / AND over a (2n+1)-deep register ring, no LDS; ~0.3 vector instructions per pixel, so the
// kernel runs at the speed of its 2 B/px of traffic (the byte-per-lane-dword version above spends ~10x the ALU work
// and two LDS rings per row).
// Lane layout: a wavefront covers 2048 pixels of a row as 128 chunks of 16; lane L owns chunk L (low 16 bits of its
// word) and chunk 64 + L (high 16 bits), so each of the two 16-byte loads / stores per lane is one contiguous 1 KiB
// (whole cache lines) across the wavefront.  (One 32-pixel word per lane needs two stores of alternate 16-byte pieces
// per lane: the write path then took 190 of 330 us.)  Bitwise work is the same for both halves at once; only the
// horizontal shifts mask the bits that would cross from one half into the other, and lane 0 / 63 take their missing
// neighbour half from the other end of the wavefront (DPP wave_ror / wave_rol + half swap).
// The destination is the caller's buffer and may have any alignment: output chunks are the 16-byte-ALIGNED chunks of
// the destination row, chunk c = the top A bits of source chunk c-1 and the low 16-A bits of chunk c (A = row address
// mod 16); strips advance by 125 chunks (2000 px) so that consecutive strips tile the row for every A.
constexpr int kBitsMaxN = 4;
constexpr int kBitsAdvance = 2000;  // output pixels per strip: chunks 2 .. 126

__device__ __forceinline__ unsigned pack_nibble(unsigned x)  // bytes 0x00 / 0xFF -> 4 bits (byte j -> bit j)
{
    return ((x & 0x01010101u) * 0x01020408u) >> 24;
}

__device__ __forceinline__ unsigned pack16(uint4 a)
{
    return pack_nibble(a.x) | (pack_nibble(a.y) << 4) | (pack_nibble(a.z) << 8) | (pack_nibble(a.w) << 12);
}

__device__ __forceinline__ unsigned unpack_nibble(unsigned nib)  // 4 bits -> 4 bytes of 0x00 / 0xFF
{
    const unsigned ones = (nib * 0x00204081u) & 0x01010101u;  // copies 7 bits apart: bit j lands on bit 8j
    return (ones << 8) - ones;                                 // * 255
}

__device__ __forceinline__ uint4 unpack16(unsigned h)
{
    return make_uint4(unpack_nibble(h & 15u), unpack_nibble((h >> 4) & 15u), unpack_nibble((h >> 8) & 15u),
                      unpack_nibble((h >> 12) & 15u));
}

// words of the previous / next chunk for both halves of `c`
__device__ __forceinline__ void neighbours(unsigned c, int lane, unsigned* prev, unsigned* next)
{
    unsigned p = (unsigned)__builtin_amdgcn_update_dpp(0, (int)c, 0x13C, 0xf, 0xf, false);  // wave_ror:1 = lane - 1
    unsigned n = (unsigned)__builtin_amdgcn_update_dpp(0, (int)c, 0x134, 0xf, 0xf, false);  // wave_rol:1 = lane + 1
    // chunk 64 (high half of lane 0) follows chunk 63 (low half of lane 63)
    if (lane == 0) p = (p << 16) | (p >> 16);
    if (lane == 63) n = (n << 16) | (n >> 16);
    *prev = p;
    *next = n;
}

// The two 16-bit halves of a word are independent chunks: packed 16-bit shifts (v_pk_lshlrev_b16 / v_pk_lshrrev_b16)
// move both at once and nothing crosses between them, so a funnel shift against the neighbour chunk is two shifts and
// an OR (the 32-bit shifts needed two masks on top).
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_shl(unsigned x, unsigned k)
{
    const us2 v = __builtin_bit_cast(us2, x) << (us2)((unsigned short)k);
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ unsigned pk_shr(unsigned x, unsigned k)
{
    const us2 v = __builtin_bit_cast(us2, x) >> (us2)((unsigned short)k);
    return __builtin_bit_cast(unsigned, v);
}

// both 16-bit halves shifted towards higher x by k (pixel x takes pixel x - k), 0 <= k <= 15
__device__ __forceinline__ unsigned shift_up(unsigned c, unsigned prev, unsigned k)
{
    return k == 0 ? c : (pk_shl(c, k) | pk_shr(prev, 16u - k));  // (a 16-bit shift by 16 would wrap to 0)
}

// both halves shifted towards lower x by k (pixel x takes pixel x + k), 1 <= k <= 15
__device__ __forceinline__ unsigned shift_down(unsigned c, unsigned next, unsigned k)
{
    return pk_shr(c, k) | pk_shl(next, 16u - k);
}

template <bool IS_OR, int N>
__device__ __forceinline__ unsigned hop_bits(unsigned c, int lane)
{
    unsigned prev, next;
    neighbours(c, lane, &prev, &next);
    unsigned v = c;
#pragma unroll
    for (int k = 1; k <= N; ++k) {
        if (IS_OR) {
            v |= pk_shl(c, k) | pk_shr(prev, 16u - k);
            v |= pk_shr(c, k) | pk_shl(next, 16u - k);
        } else {
            v &= pk_shl(c, k) | pk_shr(prev, 16u - k);
            v &= pk_shr(c, k) | pk_shl(next, 16u - k);
        }
    }
    return v;
}

// BITSRC: the source is already a bit plane (1 bit per pixel, rows of src.step bytes, written by the fused threshold
// kernel): a chunk is one 16-bit load instead of 16 bytes.
template <int N, bool FIRST_OR, bool BITSRC>
__global__ void __launch_bounds__(256) k_morph_bits(PageSet src, PageSetOut dst, int width, int height, int n_strips,
                                                   int n_segs, int rows_per_seg, unsigned total_waves)
{
    constexpr int K = 2 * N + 1;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned wid = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + wv);
    if (wid >= total_waves) return;
    const int per_page = n_strips * n_segs;
    const int page = (int)(wid / (unsigned)per_page);
    const int rem = (int)(wid - (unsigned)page * (unsigned)per_page);
    const int seg = rem / n_strips, strip = rem - seg * n_strips;

    typedef const uint8_t __attribute__((address_space(1)))* mgcptr;  // known-global pointers: global_load / global_store
    typedef uint8_t __attribute__((address_space(1)))* mgptr;
    mgcptr in = (mgcptr)src.page(page);
    mgptr out = (mgptr)dst.page(page);
    const int base_px = strip * kBitsAdvance - 32;                 // source pixel of chunk 0
    const int gx0 = base_px + 16 * lane, gx1 = gx0 + 1024;         // first pixel of this lane's two chunks
    const int ys = seg * rows_per_seg, ye = min(ys + rows_per_seg, height);
    const unsigned neutral1 = FIRST_OR ? 0u : 0xffffffffu, neutral2 = ~neutral1;
    // bits of the two halves that are pixels of the page row
    auto inside16 = [&](int gx) -> unsigned {
        if (gx + 16 <= 0 || gx >= width) return 0u;
        const int b0 = max(0, -gx), b1 = min(16, width - gx);  // [b0, b1)
        return (((1u << (b1 - b0)) - 1u) << b0) & 0xffffu;
    };
    const unsigned in0 = inside16(gx0), in1 = inside16(gx1);
    const unsigned inside = in0 | (in1 << 16);

    // (source rows are 16-byte aligned and padded to whole chunks - checked by the host: the pipeline's own mask
    // buffer - so an edge chunk is fetched whole and masked)
    // which of this lane's two output chunks are stored whole / partly: fixed for the wavefront when every row of the
    // destination has the same address mod 16 (row step a multiple of 16: the usual case), else redone per row
    const bool fixed_a = (dst.step & 15u) == 0;
    unsigned A = (unsigned)((size_t)(uint8_t*)out & 15u);
    bool full0 = false, full1 = false, rag0 = false, rag1 = false;
    auto classify = [&](unsigned a) {
        const int p0 = base_px + 16 * lane - (int)a, p1 = p0 + 1024;
        const bool s0 = lane >= 2 && p0 + 16 > 0 && p0 < width, s1 = lane <= 62 && p1 + 16 > 0 && p1 < width;
        full0 = s0 && p0 >= 0 && p0 + 16 <= width;
        full1 = s1 && p1 >= 0 && p1 + 16 <= width;
        rag0 = s0 && !full0;
        rag1 = s1 && !full1;
    };
    classify(A);

    unsigned ring1[K], ring2[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        ring1[k] = neutral1;
        ring2[k] = neutral2;
    }
    // Rows are taken four at a time: all their loads are issued back to back in straight-line code (clamped row and
    // column addresses instead of branches, the fetched words masked afterwards), so a wavefront waits for memory once
    // per four rows.  With one row per iteration the compiler had to put `s_waitcnt vmcnt(0)` right behind each load -
    // the counter is shared with the (conditional) stores of the previous row, whose write acknowledgements then sat
    // on every row's critical path.
    constexpr int RB = 4;
    const int boff0 = in0 ? (gx0 >> 3) : 0, boff1 = in1 ? (gx1 >> 3) : 0;   // BITSRC: byte offsets of the two 16-bit words
    const int poff0 = in0 ? gx0 : 0, poff1 = in1 ? gx1 : 0;                 // byte masks: offsets of the two 16-byte chunks
#pragma unroll 1
    for (int r0 = ys - 2 * N; r0 < ye + 2 * N; r0 += RB) {
        unsigned raw[RB];
        if (BITSRC) {
            unsigned lo[RB], hi[RB];
#pragma unroll
            for (int k = 0; k < RB; ++k) {
                mgcptr row = in + (size_t)min(max(r0 + k, 0), height - 1) * src.step;
                lo[k] = *reinterpret_cast<const unsigned short*>((const uint8_t*)(row + boff0));
                hi[k] = *reinterpret_cast<const unsigned short*>((const uint8_t*)(row + boff1));
            }
#pragma unroll
            for (int k = 0; k < RB; ++k) raw[k] = lo[k] | (hi[k] << 16);
        } else {
            uint4 qa[RB], qb[RB];
#pragma unroll
            for (int k = 0; k < RB; ++k) {
                mgcptr row = in + (size_t)min(max(r0 + k, 0), height - 1) * src.step;
                qa[k] = *reinterpret_cast<const uint4*>((const uint8_t*)(row + poff0));
                qb[k] = *reinterpret_cast<const uint4*>((const uint8_t*)(row + poff1));
            }
#pragma unroll
            for (int k = 0; k < RB; ++k) raw[k] = pack16(qa[k]) | (pack16(qb[k]) << 16);
        }
#pragma unroll
      for (int kk = 0; kk < RB; ++kk) {
        const int r = r0 + kk;
        if (r >= ye + 2 * N) break;  // wave-uniform
        const unsigned v = (r < 0 || r >= height) ? neutral1 : ((raw[kk] & inside) | (neutral1 & ~inside));
#pragma unroll
        for (int k = K - 1; k > 0; --k) ring1[k] = ring1[k - 1];
        ring1[0] = hop_bits<FIRST_OR, N>(v, lane);
        const int rc = r - N;  // centre row of the first operator's vertical window (complete once r >= ys)
        if (r < ys) continue;  // wave-uniform
        unsigned v1 = ring1[0];
#pragma unroll
        for (int k = 1; k < K; ++k) v1 = comb<FIRST_OR>(v1, ring1[k]);
        // pixels outside the page do not exist for the second operator
        v1 = (rc < 0 || rc >= height) ? neutral2 : ((v1 & inside) | (neutral2 & ~inside));
#pragma unroll
        for (int k = K - 1; k > 0; --k) ring2[k] = ring2[k - 1];
        ring2[0] = hop_bits<!FIRST_OR, N>(v1, lane);
        const int ro = r - 2 * N;  // output row (its window is complete once ro >= ys)
        if (ro < ys) continue;
        unsigned o = ring2[0];
#pragma unroll
        for (int k = 1; k < K; ++k) o = comb<!FIRST_OR>(o, ring2[k]);

        // destination-aligned chunks: chunk c covers pixels [base_px + 16 c - A, + 16)
        mgptr orow = out + (size_t)ro * dst.step;
        if (!fixed_a) {
            A = (unsigned)((size_t)(uint8_t*)orow & 15u);  // wave-uniform
            classify(A);
        }
        unsigned oprev, onext;
        neighbours(o, lane, &oprev, &onext);
        const unsigned bits = shift_up(o, oprev, A);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const bool full = half ? full1 : full0, ragged = half ? rag1 : rag0;
            if (!(full || ragged)) continue;
            const int px = base_px + 16 * (half * 64 + lane) - (int)A;
            // (a 256-entry LDS table for this expansion - two ds_read_b64 instead of ~20 vector instructions - measured the same)
            const uint4 d = unpack16((bits >> (16 * half)) & 0xffffu);
            if (full) {
                typedef unsigned u4v __attribute__((ext_vector_type(4)));
                const u4v dv = {d.x, d.y, d.z, d.w};
                __builtin_nontemporal_store(dv, reinterpret_cast<u4v*>((uint8_t*)(orow + px)));  // streamed out, never re-read
            } else {  // ragged ends of the row: dwords that lie inside, then bytes
                const unsigned dw[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int p4 = px + 4 * q;
                    if (p4 >= 0 && p4 + 4 <= width) {
                        *reinterpret_cast<unsigned*>((uint8_t*)(orow + p4)) = dw[q];
                    } else {
#pragma unroll
                        for (int b = 0; b < 4; ++b)
                            if (p4 + b >= 0 && p4 + b < width) orow[p4 + b] = (uint8_t)(dw[q] >> (8 * b));
                    }
                }
            }
        }
      }
    }
}

// launches k_morph_bits; PRL_ERR_BAD_ARG when the grid would not fit (the caller falls back)
int launch_morph_bits(int iterations, bool bitsrc, const PageSet& src, int n_pages, int width, int height,
                      const PageSetOut& dst, hipStream_t stream)
{
    const int n = iterations > 0 ? iterations : -iterations;
    const int n_strips = (width + 15 + kBitsAdvance - 1) / kBitsAdvance;
    int rps = 64;  // 256 x 4K pages: 64 rows per segment 1.18 ms, 128: 1.23, 256: 1.24, 512: 1.31
    // small batches: fill the chip first (one 4K page, closing 2: 0.082 -> 0.069 ms per call at 8 rows per segment)
    while (rps > 8 && (long long)n_pages * n_strips * ((height + rps - 1) / rps) < 16384) rps /= 2;
    if (env_knobs().morph_rps) rps = env_knobs().morph_rps;  // tuning knob
    const int n_segs = (height + rps - 1) / rps;
    const unsigned long long tw = (unsigned long long)n_pages * n_strips * n_segs;
    if (tw >= 0xfffffff0ull) return PRL_ERR_BAD_ARG;
    // independent wavefronts: one per workgroup refills a finished slot at once (see launch_sweep in binarize_fused.hip)
    const unsigned wpb = (unsigned)env_knobs().morph_wpb;
    if ((tw + wpb - 1) / wpb > 0x7fffff00ull) return PRL_ERR_BAD_ARG;  // grid.x limit: the caller falls back
    const dim3 grid((unsigned)((tw + wpb - 1) / wpb)), block(64 * wpb);
#define PRL_LAUNCH_BITS2(NV, OR1, BS)                                                                                  \
    hipLaunchKernelGGL((k_morph_bits<NV, OR1, BS>), grid, block, 0, stream, src, dst, width, height, n_strips, n_segs, \
                       rps, (unsigned)tw)
#define PRL_LAUNCH_BITS(NV)                                          \
    do {                                                             \
        if (iterations > 0) {                                        \
            if (bitsrc) PRL_LAUNCH_BITS2(NV, true, true);            \
            else PRL_LAUNCH_BITS2(NV, true, false);                  \
        } else {                                                     \
            if (bitsrc) PRL_LAUNCH_BITS2(NV, false, true);           \
            else PRL_LAUNCH_BITS2(NV, false, false);                 \
        }                                                            \
    } while (0)
    switch (n) {
    case 1: PRL_LAUNCH_BITS(1); break;
    case 2: PRL_LAUNCH_BITS(2); break;
    case 3: PRL_LAUNCH_BITS(3); break;
    default: PRL_LAUNCH_BITS(4); break;
    }
#undef PRL_LAUNCH_BITS
#undef PRL_LAUNCH_BITS2
    PRL_HIP_CHECK(hipGetLastError());
    return PRL_OK;
}

// bytes 0 / 255 -> bit plane, one thread per 8 pixels (overflow pages redone by the literal pipeline)
__global__ void __launch_bounds__(256) k_pack_mask(const uint8_t* __restrict__ src, size_t src_step, int width, int height,
                                                  uint8_t* __restrict__ bits, size_t bit_step)
{
    const int y = blockIdx.y, xb = blockIdx.x * 256 + threadIdx.x;  // byte of the bit row
    if (xb * 8 >= width) return;
    const uint8_t* s = src + (size_t)y * src_step + (size_t)xb * 8;
    unsigned b = 0;
    for (int i = 0; i < 8 && xb * 8 + i < width; ++i) b |= (unsigned)(s[i] & 1u) << i;
    bits[(size_t)y * bit_step + xb] = (uint8_t)b;
}

}  // namespace

// The rectangle of n iterations equals n applications of the 3x3 one, and a closing/opening with
// radius n cannot be split into smaller closings — so radii above kMaxN are rejected here and
// handled by the caller (PRL_ERR_BAD_ARG); the reference's defaults are n in {0, 2}.
// Radius above kMaxN: chain single-operator passes of radius <= kMaxN, ping-ponging between `tmp` and `dst`
// so that the last pass lands in dst.  `tmp` must hold n_pages pages of height x tmp_step bytes.
int morph_large_run(int iterations, const PageSet& src, int n_pages, int width, int height, const PageSetOut& dst,
                    uint8_t* tmp, size_t tmp_step, hipStream_t stream)
{
    const int n = iterations > 0 ? iterations : -iterations;
    const int per_op = (n + kMaxN - 1) / kMaxN, passes = 2 * per_op;
    const dim3 grid((width + TW - 1) / TW, (height + TH - 1) / TH, n_pages);
    PageSetOut t{};
    t.base = tmp;
    t.page_stride = tmp_step * (size_t)height;
    t.step = tmp_step;
    PageSet cur = src;
    for (int i = 0; i < passes; ++i) {
        const bool first_op = i < per_op;
        const int idx = first_op ? i : i - per_op;
        const int r = (idx == per_op - 1) ? n - kMaxN * (per_op - 1) : kMaxN;
        const bool take_max = (iterations > 0) == first_op;          // closing: max then min; opening: min then max
        const PageSetOut& o = ((passes - 1 - i) % 2 == 0) ? dst : t;  // last pass -> dst
        if (take_max) hipLaunchKernelGGL(k_rect<true>, grid, dim3(256), 0, stream, cur, o, width, height, r);
        else hipLaunchKernelGGL(k_rect<false>, grid, dim3(256), 0, stream, cur, o, width, height, r);
        PRL_HIP_CHECK(hipGetLastError());
        cur = PageSet{};
        cur.base = o.base;
        cur.page_stride = o.page_stride;
        cur.table = o.table;
        cur.step = o.step;
    }
    return PRL_OK;
}

// binary (0/255) masks: the pipeline's own threshold output
int morph_binary_run(int iterations, const PageSet& src, int n_pages, int width, int height,
                     const PageSetOut& dst, hipStream_t stream)
{
    const int n = iterations > 0 ? iterations : -iterations;
    if (n == 0 || n > kMaxN) {
        set_error_detail("radius above " + std::to_string(kMaxN) + " goes through morph_large_run");
        return PRL_ERR_BAD_ARG;
    }
    if (n <= kBitsMaxN && (((size_t)src.base | src.page_stride | src.step) & 15) == 0 && !src.table &&
        src.step >= (size_t)((width + 15) / 16) * 16) {
        // bit-domain streaming kernel: needs 16-byte aligned source rows padded to whole 16-pixel chunks (the
        // pipeline's own mask buffer always is)
        if (launch_morph_bits(iterations, false, src, n_pages, width, height, dst, stream) == PRL_OK) return PRL_OK;
    }
    if ((((size_t)src.base | src.page_stride | src.step) & 3) == 0 && !src.table) {
        // streaming kernel: needs 4-byte aligned source rows (the pipeline's own mask buffer always is)
        const int hd = 2 * ((n + 3) / 4), useful = (64 - 2 * hd) * 4;
        const int n_strips = (width + useful - 1) / useful;
        int rps = 128;  // short segments: many wavefronts, each with two row fetches in flight
        while (rps > 32 && (long long)n_pages * n_strips * ((height + rps - 1) / rps) < 32768) rps /= 2;
        const int n_segs = (height + rps - 1) / rps;
        const unsigned long long tw = (unsigned long long)n_pages * n_strips * n_segs;
        if (tw < 0xfffffff0ull) {
            const unsigned blocks = (unsigned)((tw + 3) / 4);
            const size_t lds = (size_t)4 * 2 * (2 * n + 1) * 64 * sizeof(unsigned);
            if (iterations > 0)
                hipLaunchKernelGGL(k_morph_stream<true>, dim3(blocks), dim3(256), lds, stream, src, dst, width, height, n,
                                   n_strips, n_segs, rps, (unsigned)tw);
            else
                hipLaunchKernelGGL(k_morph_stream<false>, dim3(blocks), dim3(256), lds, stream, src, dst, width, height, n,
                                   n_strips, n_segs, rps, (unsigned)tw);
            PRL_HIP_CHECK(hipGetLastError());
            return PRL_OK;
        }
    }
    const int h4 = ((2 * n + 3) / 4) * 4 + 8;  // halo: the window (2n) rounded to dwords + 2 guard dwords
    const int btw = BND * 4 - 2 * h4;          // output pixels per tile row (232 for n <= 2)
    const dim3 grid((width + btw - 1) / btw, (height + BTH - 1) / BTH, n_pages);
    if (iterations > 0)
        hipLaunchKernelGGL(k_morph_binary<true>, grid, dim3(256), 0, stream, src, dst, width, height, n, btw);
    else
        hipLaunchKernelGGL(k_morph_binary<false>, grid, dim3(256), 0, stream, src, dst, width, height, n, btw);
    PRL_HIP_CHECK(hipGetLastError());
    return PRL_OK;
}

// binary masks kept as bit planes (rows of src.step bytes, 1 bit per pixel): radius <= morph_bits_max_radius()
int morph_bits_max_radius() { return kBitsMaxN; }

int morph_bitplane_run(int iterations, const PageSet& bits, int n_pages, int width, int height, const PageSetOut& dst,
                       hipStream_t stream)
{
    const int n = iterations > 0 ? iterations : -iterations;
    if (n == 0 || n > kBitsMaxN || bits.table || (((size_t)bits.base | bits.page_stride | bits.step) & 1) != 0 ||
        bits.step < (size_t)((width + 15) / 16) * 2)
        return PRL_ERR_BAD_ARG;
    return launch_morph_bits(iterations, true, bits, n_pages, width, height, dst, stream);
}

int pack_mask_run(const uint8_t* src, size_t src_step, int width, int height, uint8_t* bits, size_t bit_step,
                  hipStream_t stream)
{
    const dim3 grid((unsigned)(((width + 7) / 8 + 255) / 256), (unsigned)height);
    hipLaunchKernelGGL(k_pack_mask, grid, dim3(256), 0, stream, src, src_step, width, height, bits, bit_step);
    PRL_HIP_CHECK(hipGetLastError());
    return PRL_OK;
}

// any 8-bit image (max / min): the public prl_hip_morph_batch_device entry
int morph_run(int iterations, const PageSet& src, int n_pages, int width, int height,
              const PageSetOut& dst, hipStream_t stream)
{
    const int n = iterations > 0 ? iterations : -iterations;
    if (n == 0 || n > kMaxN) {
        set_error_detail("radius above " + std::to_string(kMaxN) + " goes through morph_large_run");
        return PRL_ERR_BAD_ARG;
    }
    const dim3 grid((width + TW - 1) / TW, (height + TH - 1) / TH, n_pages);
    if (iterations > 0)
        hipLaunchKernelGGL(k_morph<true>, grid, dim3(256), 0, stream, src, dst, width, height, n);
    else
        hipLaunchKernelGGL(k_morph<false>, grid, dim3(256), 0, stream, src, dst, width, height, n);
    PRL_HIP_CHECK(hipGetLastError());
    return PRL_OK;
}

}  // namespace prl_hip
