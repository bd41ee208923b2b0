// pyrdown.hip — Frame::ComputeImagePyramid's cv::pyrDown chain on packed device pyramids.
//
// Replaces reference src/Frame.cpp:74-81 (cv::pyrDown per level; OpenCV 2.4 8-bit semantics:
// separable [1 4 6 4 1], BORDER_REFLECT_101, (sum+128)>>8, output ((w+1)/2,(h+1)/2)).
// Integer arithmetic, bit-exact. HBM-bound: each thread owns a strip of 4 output columns (one dword
// store per output row) and walks down PD_ROWS output rows with a sliding window of horizontally
// filtered input rows (v_dot4_u32_u8), each read once as one 16-byte load, vertical pass on packed
// u16; odd sizes take a byte-wise reflected path. Two kernels share that strip task: one launch per
// level (batches), and one launch per pyramid with the intermediate levels in LDS (a few images).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace dsdtm {

__device__ __forceinline__ int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = (p < 0) ? -p : 2 * len - 2 - p;
    return p;
}

struct PyrDownArgs {
    uint8_t* pyr;
    size_t pyr_pitch;
    int n_images;
    int sw, sh, sstride;
    size_t soff;
    int dw, dh, dstride;
    size_t doff;
};

// Each thread owns a strip of 4 output columns and walks down PD_ROWS output rows with a sliding
// window of horizontally filtered rows: every input row of the strip is loaded (one 16-byte load)
// and filtered ONCE and reused by the 2-3 output rows it contributes to.
//
// The first version spent ~150 VALU instructions per output row on byte extraction and scalar
// multiply-adds and was issue-bound (2.2 TB/s). Now: the horizontal [1 4 6 4 1] of one output is
// ONE v_dot4_u32_u8 of the 4-byte window at its column with the constant (1,4,6,4), the fifth tap as
// the accumulator operand (window assembled with v_alignbyte); the filtered rows are kept as packed
// u16 pairs (<= 4080 each) and the vertical pass runs on v_pk_* (<= 65280 + 128 fits 16 bits);
// ~45 instructions per output row. Threads are numbered linearly over (strip, row chunk), so waves
// are full whatever the level width (640 px = 80 strips used to leave the second block column at 16
// of 64 lanes).
constexpr int PD_ROWS = 4;
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ u16x2 pk(uint32_t lo, uint32_t hi) {
    u16x2 v; v.x = (unsigned short)lo; v.y = (unsigned short)hi; return v;
}

// horizontal [1 4 6 4 1] of a loaded input row for the 4 output columns x4..x4+3 (aligned fast path):
// hA = (h0, h1), hB = (h2, h3).
// e0..e3 = columns xs-2 .. xs+13 with xs = 2*x4 - 2. Interior strips load exactly that (base = xs-2). The left-border
// strip (xs = -2) needs columns -4..-1 mirrored to 4..1:
//   SHIFTED = false: it loaded columns 0..15 (base 0) and every word moves up by one (4 selects + 1 perm per row);
//   SHIFTED = true:  it loaded from base -4 like everybody else — the four bytes before the row start are the end of
//                    the previous row, valid memory wherever the caller uses this form — and only e0 is replaced
//                    (1 select + 1 perm). The kernel is bound by instruction issue as much as by HBM; this is the
//                    form of every task that does not touch the top or bottom border.
template <bool SHIFTED>
__device__ __forceinline__ void pd_filter(const uint4 d, bool left, bool right, u16x2& hA, u16x2& hB) {
    const uint32_t d0 = d.x, d1 = d.y, d2 = d.z, d3 = d.w;
    uint32_t e0, e1, e2, e3;
    if constexpr (SHIFTED) {
        e0 = left ? __builtin_amdgcn_perm(d2, d1, 0x01020304u) : d0;                 // bytes [col4, col3, col2, col1]
        e1 = d1; e2 = d2; e3 = d3;
    } else {
        e0 = left ? __builtin_amdgcn_perm(d1, d0, 0x01020304u) : d0;
        e1 = left ? d0 : d1;
        e2 = left ? d1 : d2;
        e3 = left ? d2 : d3;
    }
    // output o: taps at bytes 2+2o .. 6+2o of (e0..e3)
    const uint32_t k = 0x04060401u;                                   // weights of taps 0..3; tap 4 has weight 1
    const uint32_t w0 = __builtin_amdgcn_alignbyte(e1, e0, 2);        // bytes 2..5
    const uint32_t w2 = __builtin_amdgcn_alignbyte(e2, e1, 2);        // bytes 6..9
    const uint32_t t0 = (e1 >> 16) & 0xffu;                           // byte 6
    const uint32_t t1 = e2 & 0xffu;                                   // byte 8
    const uint32_t t2 = (e2 >> 16) & 0xffu;                           // byte 10
    const uint32_t t3 = right ? t2 : (e3 & 0xffu);                    // byte 12; column sw -> sw-2 (BORDER_REFLECT_101)
    const uint32_t h0 = __builtin_amdgcn_udot4(w0, k, t0, false);
    const uint32_t h1 = __builtin_amdgcn_udot4(e1, k, t1, false);
    const uint32_t h2 = __builtin_amdgcn_udot4(w2, k, t2, false);
    const uint32_t h3 = __builtin_amdgcn_udot4(e2, k, t3, false);
    hA = pk(h0, h1);
    hB = pk(h2, h3);
}
// vertical [1 4 6 4 1] + rounding of one packed pair: ((r0 + r4) + 4 (r1 + r3) + 6 r2 + 128) >> 8
__device__ __forceinline__ u16x2 pd_vert(u16x2 r0, u16x2 r1, u16x2 r2, u16x2 r3, u16x2 r4) {
    const u16x2 four = {4, 4}, six = {6, 6}, half = {128, 128}, eight = {8, 8};
    u16x2 v = (r1 + r3) * four + (r0 + r4);
    v = r2 * six + v;
    return (v + half) >> eight;
}
// the four output bytes of a strip from the two packed pairs: ONE v_perm (bytes 0 and 2 of each pair)
__device__ __forceinline__ uint32_t pd_pack(u16x2 oA, u16x2 oB) {
    return __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, oB), __builtin_bit_cast(uint32_t, oA), 0x06040200u);
}

// 16 bytes of a source row: from HBM as one global_load_dwordx4, from LDS as two 8-byte reads. Strip bases are only
// 4-byte aligned (2 * x4 - 4 is 4 mod 8 for every strip but the left one), so the LDS reads go through a type that
// says so: the compiler may still pick ds_read_b64 where the hardware's unaligned DS access allows it, or two
// ds_read2_b32, but never assumes an 8-byte alignment that does not hold. The 4th dword of the right-border strip
// starts at the row end: in HBM it lies inside the pyramid (a source level is never the last level), in LDS inside
// the band buffer's padding; it is not used.
struct __attribute__((packed, aligned(4))) PdU32x2 { uint32_t x, y; };
template <bool SRC_LDS>
__device__ __forceinline__ uint4 pd_load16(const uint8_t* __restrict__ p) {
    if constexpr (SRC_LDS) {
        const PdU32x2* __restrict__ q = (const PdU32x2*)p;
        const PdU32x2 a = q[0], b = q[1];
        return make_uint4(a.x, a.y, b.x, b.y);
    } else {
        const uint32_t* __restrict__ p32 = (const uint32_t*)p;
        return make_uint4(p32[0], p32[1], p32[2], p32[3]);
    }
}

// filter + store of one task whose 2*ROWS + 3 source rows are in `raw`. No control flow around the arithmetic — rows
// behind d_hi (a ragged last chunk; never in the interior form) are computed from valid rows and not stored: with an
// early exit per row the compiler sinks the row loads into the conditional blocks, and the point of `raw` is that
// all of them are in flight at once.
template <bool SHIFTED, bool FULL, bool DST_LDS, int ROWS>
__device__ __forceinline__ void pd_rows(const uint4* raw, bool left, bool right, int y0, int x4, int d_lo, int d_hi,
                                        uint8_t* __restrict__ ldst, int lstride, uint8_t* __restrict__ gp, int gstride,
                                        int own_lo, int own_hi) {
    u16x2 wA[5], wB[5];                              // horizontally filtered rows 2y-2 .. 2y+2
    pd_filter<SHIFTED>(raw[0], left, right, wA[0], wB[0]);
    pd_filter<SHIFTED>(raw[1], left, right, wA[1], wB[1]);
    pd_filter<SHIFTED>(raw[2], left, right, wA[2], wB[2]);
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int y = y0 + r;
        // window slot of input row 2y+k-2 is (2r + k) % 5
        pd_filter<SHIFTED>(raw[2 * r + 3], left, right, wA[(2 * r + 3) % 5], wB[(2 * r + 3) % 5]);
        pd_filter<SHIFTED>(raw[2 * r + 4], left, right, wA[(2 * r + 4) % 5], wB[(2 * r + 4) % 5]);
        const u16x2 oA = pd_vert(wA[(2 * r) % 5], wA[(2 * r + 1) % 5], wA[(2 * r + 2) % 5], wA[(2 * r + 3) % 5], wA[(2 * r + 4) % 5]);
        const u16x2 oB = pd_vert(wB[(2 * r) % 5], wB[(2 * r + 1) % 5], wB[(2 * r + 2) % 5], wB[(2 * r + 3) % 5], wB[(2 * r + 4) % 5]);
        const uint32_t packed = pd_pack(oA, oB);
        if constexpr (DST_LDS) { if (FULL || y < d_hi) *(uint32_t*)(ldst + (ptrdiff_t)(y - d_lo) * lstride + x4) = packed; }
        if (y >= own_lo && y < own_hi) *(uint32_t*)(gp + (ptrdiff_t)r * gstride) = packed;      // own_hi <= d_hi
    }
}

// One strip task of the aligned fast path: output rows y0 .. y0+ROWS-1 (below d_hi) x output columns x4 .. x4+3.
// All 2*ROWS + 3 source rows are in flight before the first one is filtered (the sliding-window loop otherwise waits
// for two rows per output row, and ~2 loads per wave in flight is short of what keeps HBM busy). Row addressing:
// away from the top and bottom of the level the rows of a task are consecutive — one address and compile-time
// multiples of the stride; only tasks that touch a border reflect (BORDER_REFLECT_101, one reflection: sh >= 4).
// (The first version called a general reflect101() with its while loop for every row: 19 divergent loops per task,
// a quarter of the kernel's instructions.) `src` points at row s_row0 of the source level; `last` is the last source
// row index (before reflection) the caller's row range needs.
template <bool SRC_LDS, bool DST_LDS, int ROWS>
__device__ __forceinline__ void pd_task(const uint8_t* __restrict__ src, int sstride, int s_row0, int sw, int sh, int y0, int x4,
                                        int d_lo, int d_hi, int last, uint8_t* __restrict__ ldst, int lstride,
                                        uint8_t* __restrict__ gdst, int gstride, int own_lo, int own_hi) {
    const int xs = 2 * x4 - 2;                       // first input column needed (x4 is a multiple of 4)
    // Interior strips read columns (xs-2)..(xs+13) and use bytes 2..12; the left-border strip (xs = -2) reads columns
    // 0..15 and mirrors; the right-border strip of an even-width level needs column sw -> sw-2. Keeping the border
    // lanes on this path matters: one lane on a byte-wise path stalls its whole wave.
    const bool left = (x4 == 0);
    const bool right = (xs + 10 >= sw);
    const int base = left ? 0 : (xs - 2);            // multiple of 4 (of 8 from the second strip on)
    const int t0 = 2 * y0 - 2;
    uint4 raw[2 * ROWS + 3];
    uint8_t* __restrict__ gp = gdst + (ptrdiff_t)y0 * gstride + x4;
    if (t0 >= (SRC_LDS ? 0 : 1) && t0 + 2 * ROWS + 2 <= min(last, sh - 1)) {
        // (HBM source: the shifted form of the left strip reads 4 bytes before its rows, hence not row 0)
        constexpr bool SH = !SRC_LDS;
        const uint8_t* __restrict__ p = src + (ptrdiff_t)(t0 - s_row0) * sstride + (SH ? xs - 2 : base);
#pragma unroll
        for (int k = 0; k < 2 * ROWS + 3; ++k) raw[k] = pd_load16<SRC_LDS>(p + (ptrdiff_t)k * sstride);
        pd_rows<SH, true, DST_LDS, ROWS>(raw, left, right, y0, x4, d_lo, d_hi, ldst, lstride, gp, gstride, own_lo, own_hi);
    } else {
#pragma unroll
        for (int k = 0; k < 2 * ROWS + 3; ++k) {
            int r = min(t0 + k, last);               // rows behind the caller's range: any valid row (their outputs are not stored)
            r = r < 0 ? -r : r;
            r = r >= sh ? 2 * sh - 2 - r : r;
            raw[k] = pd_load16<SRC_LDS>(src + (ptrdiff_t)(r - s_row0) * sstride + base);
        }
        pd_rows<false, false, DST_LDS, ROWS>(raw, left, right, y0, x4, d_lo, d_hi, ldst, lstride, gp, gstride, own_lo, own_hi);
    }
}

// XCD-aware block numbering. Workgroups go to the 8 XCDs round-robin by their linear id, and each XCD has its own
// L2: with the plain numbering the blocks of ONE image — whose row chunks share 3 halo rows with their
// neighbours — are spread over all eight L2s and every halo row is fetched from HBM by two of them. The
// remapped id gives each XCD a contiguous range of (image, block) pairs, so an image's blocks meet in one L2.
// SA_PD_XCD=0 restores the plain numbering (A/B).
__device__ __forceinline__ void pd_block(int& img, int& bx) {
    const unsigned nb = gridDim.x, total = gridDim.x * gridDim.z;
    unsigned lb = blockIdx.z * nb + blockIdx.x;
    const unsigned q = total / 8u;
    if (lb < q * 8u) lb = (lb % 8u) * q + lb / 8u;
    img = (int)(lb / nb);
    bx = (int)(lb % nb);
}

__global__ __launch_bounds__(256) void pyrdown_kernel(const PyrDownArgs a, int tx, int ty) {
    int img, bx;
    pd_block(img, bx);
    // linear numbering over (strip, row chunk): consecutive lanes = consecutive strips of a row chunk
    const int gid = bx * blockDim.x + threadIdx.x;
    if (gid >= tx * ty) return;
    const int y0 = (gid / tx) * PD_ROWS;                                // first output row of this thread
    const int x4 = (gid % tx) * 4;                                      // first of 4 output columns
    const uint8_t* __restrict__ src = a.pyr + (size_t)img * a.pyr_pitch + a.soff;
    uint8_t* __restrict__ dst = a.pyr + (size_t)img * a.pyr_pitch + a.doff;
    const int xs = 2 * x4 - 2;                       // first input column needed (x4 is a multiple of 4)

    // Fast path (pd_task): the 11 input columns xs..xs+10 of a row come from ONE 16-byte load.
    const bool right = (xs + 10 >= a.sw);             // only column xs+10 == sw can be outside
    const bool fast = ((a.sstride & 3) == 0) && ((((size_t)src) & 3) == 0) && ((a.sw & 1) == 0) && a.sw >= 16 && a.sh >= 4 &&
                      (x4 + 3 < a.dw) && (!right || xs + 10 == a.sw) && (((size_t)(dst + x4)) & 3) == 0 && (a.dstride & 3) == 0;
    if (fast) {
        pd_task<false, false, PD_ROWS>(src, a.sstride, 0, a.sw, a.sh, y0, x4, 0, a.dh, 2 * a.dh, nullptr, 0, dst, a.dstride, 0, a.dh);
        return;
    }
    // generic path (odd widths, unaligned buffers, ragged right edge): byte-wise BORDER_REFLECT_101
    const int wk[5] = {1, 4, 6, 4, 1};
    for (int r = 0; r < PD_ROWS; ++r) {
        const int y = y0 + r;
        if (y >= a.dh) break;
        int acc[4] = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const uint8_t* __restrict__ row = src + (size_t)reflect101(2 * y + k - 2, a.sh) * a.sstride;
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                const int xo = x4 + o;
                if (xo >= a.dw) continue;
                int h = 0;
#pragma unroll
                for (int j = 0; j < 5; ++j) h += wk[j] * (int)row[reflect101(2 * xo + j - 2, a.sw)];
                acc[o] += wk[k] * h;
            }
        }
        for (int o = 0; o < 4 && x4 + o < a.dw; ++o) dst[(size_t)y * a.dstride + x4 + o] = (uint8_t)((acc[o] + 128) >> 8);
    }
}

// (Round 2 also built a "streaming" variant: every lane loads exactly its own 16 input bytes per row — lanes 16
// bytes apart, the wide coalesced pattern — produces 8 outputs per row and takes the 3 halo bytes of the horizontal
// filter from the neighbouring lanes with two DPP wave shifts instead of from overlapping loads. Bit-exact, and
// 5–8 % SLOWER than the kernel above on the same box (0.252–0.265 vs 0.243–0.246 ms for 2048 pyramids): the
// overlapping bytes are L1 hits and were never the limit. Removed.)

// ---------------------------------------------------------------------------------------------
// One launch per pyramid: every level from level 0 in ONE kernel (Frame::ComputeImagePyramid, src/Frame.cpp:74-81).
//
// A workgroup owns a BAND of rows of the coarsest level (full width) and everything above it: it reads the level-0
// rows the band depends on from HBM, filters them into level 1 (kept in LDS, owned rows also written out), level 1
// into level 2 from LDS, and so on — levels 1.. are never read back from HBM, so a 4-level 640x480 pyramid moves
// 307 200 + 100 800 bytes instead of the 504 000 of three separate launches, and there is one launch instead of three.
// Rows a band needs beyond its own (the [1 4 6 4 1] halo, 2 rows per side and level) are recomputed: a band of b rows
// of level K needs 2^K b + 3 (2^K - 1) rows of level 0 (141 for b = 15, K = 3: 1.175x, and the neighbouring band's
// rows are L2 hits with the XCD-aware block numbering). Every stage is the per-level kernel's strip code (4 output
// columns x PD_ROWS output rows per thread, dot4 + packed u16), so the bytes are the same bytes.
struct PyrFusedArgs {
    uint8_t* pyr;
    size_t pyr_pitch;
    int band, n_bands;                      // rows of the coarsest level per workgroup
    int w[DSDTM_MAX_LEVELS], h[DSDTM_MAX_LEVELS], stride[DSDTM_MAX_LEVELS];
    unsigned long long off[DSDTM_MAX_LEVELS];
    int lds_off[DSDTM_MAX_LEVELS];          // band buffer of level l (1 <= l <= K-1) in dynamic LDS: byte offset
    int lds_stride[DSDTM_MAX_LEVELS];       //   and row stride
};

// rows [d_lo, d_hi) of a level from rows of the level below. `src` points at row s_row0 of the source level.
// ROWS = output rows per thread: 8 for the level-0 stage (every input row loaded and filtered once per 2..3 outputs,
// as in the per-level kernel), fewer for the stages that read LDS — those are short, and what matters is that a
// workgroup gets through them quickly (more, shorter tasks), because it issues no HBM loads meanwhile.
constexpr int PF_ROWS_G = 8, PF_ROWS_L = 2;   // output rows per task: level-0 stage (from HBM) / LDS stages
template <bool SRC_LDS, bool DST_LDS, int ROWS>
__device__ __forceinline__ void pf_stage(const uint8_t* __restrict__ src, int sstride, int s_row0, int sw, int sh,
                                         int d_lo, int d_hi, int dw, uint8_t* __restrict__ ldst, int lstride,
                                         uint8_t* __restrict__ gdst, int gstride, int own_lo, int own_hi) {
    const int tx = dw >> 2;
    const int chunks = (d_hi - d_lo + ROWS - 1) / ROWS;
    for (int task = threadIdx.x; task < tx * chunks; task += blockDim.x) {
        const int ch = task / tx;
        pd_task<SRC_LDS, DST_LDS, ROWS>(src, sstride, s_row0, sw, sh, d_lo + ch * ROWS, (task - ch * tx) * 4, d_lo, d_hi, 2 * d_hi,
                                        ldst, lstride, gdst, gstride, own_lo, own_hi);
    }
}

// NT = threads per workgroup, RG = output rows per task of the level-0 stage. Batches: 256 threads, 8 rows (every input row
// loaded once per 2..3 outputs — HBM traffic). A single frame (dsdtm_track_frame: 15 bands on 15 of 256 compute units) is
// bound by how long ONE workgroup takes for its chain of stages, not by traffic: 1024 threads and short tasks.
template <int K, int NT = 256, int RG = PF_ROWS_G>     // K = number of downsampling steps (levels - 1), 2..4
__global__ __launch_bounds__(NT) void pyrdown_fused_kernel(const PyrFusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t pf_lds[];
    int img, band;
    pd_block(img, band);
    // row ranges per level: [lo, hi) computed by this workgroup, [olo, ohi) written to HBM by it
    int lo[K + 1], hi[K + 1], olo[K + 1], ohi[K + 1];
    lo[K] = band * a.band; hi[K] = min(a.h[K], lo[K] + a.band);
    olo[K] = lo[K]; ohi[K] = hi[K];
#pragma unroll
    for (int l = K - 1; l >= 0; --l) {
        lo[l] = max(0, 2 * lo[l + 1] - 2); hi[l] = min(a.h[l], 2 * hi[l + 1] + 1);
        olo[l] = 2 * olo[l + 1]; ohi[l] = (band == a.n_bands - 1) ? a.h[l] : min(a.h[l], 2 * ohi[l + 1]);
    }
    uint8_t* __restrict__ base = a.pyr + (size_t)img * a.pyr_pitch;
    // level 0 (HBM) -> level 1 (LDS + owned rows)
    pf_stage<false, true, RG>(base + a.off[0], a.stride[0], 0, a.w[0], a.h[0], lo[1], hi[1], a.w[1],
                          pf_lds + a.lds_off[1], a.lds_stride[1], base + a.off[1], a.stride[1], olo[1], ohi[1]);
    __syncthreads();
#pragma unroll
    for (int l = 2; l < K; ++l) {
        pf_stage<true, true, PF_ROWS_L>(pf_lds + a.lds_off[l - 1], a.lds_stride[l - 1], lo[l - 1], a.w[l - 1], a.h[l - 1], lo[l], hi[l], a.w[l],
                             pf_lds + a.lds_off[l], a.lds_stride[l], base + a.off[l], a.stride[l], olo[l], ohi[l]);
        __syncthreads();
    }
    pf_stage<true, false, PF_ROWS_L>(pf_lds + a.lds_off[K - 1], a.lds_stride[K - 1], lo[K - 1], a.w[K - 1], a.h[K - 1], lo[K], hi[K], a.w[K],
                          nullptr, 0, base + a.off[K], a.stride[K], olo[K], ohi[K]);
}

// Can the whole pyramid go through the fused kernel? (every source level on the aligned fast path of the strip code)
static bool pyrdown_fused_ok(const uint8_t* pyr, size_t pyr_pitch, int levels, const int* w, const int* h, const int* stride,
                             const size_t* off) {
    if (levels < 3 || levels > 5) return false;
    if (((size_t)pyr & 3) || (pyr_pitch & 3)) return false;
    for (int l = 0; l < levels; ++l) {
        if ((stride[l] & 3) || (off[l] & 3)) return false;
        if (l < levels - 1 && ((w[l] & 7) || w[l] < 16)) return false;      // even width, whole strips of 4 outputs
        if (h[l] < (l < levels - 1 ? 4 : 1)) return false;                  // one reflection at the top and bottom
    }
    return true;
}

hipError_t pyrdown_fused_launch(uint8_t* pyr, size_t pyr_pitch, int n_images, int levels, const int* w, const int* h,
                                const int* stride, const size_t* off, int band, hipStream_t stream, bool* launched) {
    *launched = false;
    if (n_images <= 0) { *launched = true; return hipSuccess; }
    if (!pyrdown_fused_ok(pyr, pyr_pitch, levels, w, h, stride, off)) return hipSuccess;
    const int K = levels - 1;
    PyrFusedArgs a;
    a.pyr = pyr; a.pyr_pitch = pyr_pitch;
    for (int l = 0; l < DSDTM_MAX_LEVELS; ++l) { a.w[l] = a.h[l] = a.stride[l] = 0; a.off[l] = 0; a.lds_off[l] = 0; a.lds_stride[l] = 0; }
    for (int l = 0; l < levels; ++l) { a.w[l] = w[l]; a.h[l] = h[l]; a.stride[l] = stride[l]; a.off[l] = off[l]; }
    // band height: few bands per image for batches (little halo), many for a single frame (parallelism)
    if (band <= 0) {
        const int want_groups = 2048;
        int nb = (want_groups + n_images - 1) / n_images;
        nb = nb < 4 ? 4 : nb;
        band = (h[K] + nb - 1) / nb;
        const int min_band = (long long)n_images * h[K] <= 256 ? 1 : 2;      // a single frame: one workgroup per row of the coarsest level
        if (band < min_band) band = min_band;
    }
    if (band > h[K]) band = h[K];
    size_t lds = 0;
    for (;;) {
        a.band = band; a.n_bands = (h[K] + band - 1) / band;
        lds = 0;
        for (int l = 1; l < K; ++l) {
            int rows = 0;
            for (int b = 0; b < a.n_bands; ++b) {                       // widest band of this level
                int lo = b * band, hi = lo + band < h[K] ? lo + band : h[K];
                for (int m = K - 1; m >= l; --m) {
                    lo = 2 * lo - 2 > 0 ? 2 * lo - 2 : 0;
                    hi = 2 * hi + 1 < h[m] ? 2 * hi + 1 : h[m];
                }
                rows = hi - lo > rows ? hi - lo : rows;
            }
            a.lds_off[l] = (int)lds; a.lds_stride[l] = w[l];
            lds += ((size_t)rows * w[l] + 16 + 15) / 16 * 16;           // + the right-border strip's unused fourth dword
        }
        if (lds <= 64 * 1024 || band <= 2) break;
        band = (band + 1) / 2;                                          // smaller bands until the band buffers fit
    }
    if (lds > 64 * 1024) return hipSuccess;
    for (int i0 = 0; i0 < n_images; i0 += 65535) {
        const int nz = (n_images - i0 < 65535) ? n_images - i0 : 65535;
        PyrFusedArgs b = a;
        b.pyr = pyr + (size_t)i0 * pyr_pitch;
        const dim3 grid((unsigned)a.n_bands, 1u, (unsigned)nz);
        if (K == 4 && (long long)a.n_bands * nz <= 256) {      // fewer workgroups than compute units: what counts is one workgroup's chain
            // (one 640x480 frame, 5 levels, same box: 256 threads x 8 rows 10.8 us; 1024 x 2: 9.9; 1024 x 4: 9.9; 512 x 2: 10.8;
            //  1024 x 1: 10.3; 1024 x 2 with bands of one row: 9.4)
            hipLaunchKernelGGL((pyrdown_fused_kernel<4, 1024, 2>), grid, dim3(1024), lds, stream, b);
            continue;
        }
        if (K == 2) hipLaunchKernelGGL(pyrdown_fused_kernel<2>, grid, dim3(256), lds, stream, b);
        else if (K == 3) hipLaunchKernelGGL(pyrdown_fused_kernel<3>, grid, dim3(256), lds, stream, b);
        else hipLaunchKernelGGL(pyrdown_fused_kernel<4>, grid, dim3(256), lds, stream, b);
    }
    *launched = true;
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// A new frame's level 0 (and a second, small range: what Run reads) from HOST-MAPPED pinned memory to the device, by a
// kernel on the compute stream instead of a copy operation in front of it. For one 640x480 frame the copy engine is not
// slower at moving the bytes (8 us) — what costs is the hand-over from the copy engine to the compute queue: the first
// kernel behind a copy starts 7-8 us after the copy's end (rocprofv3 --memory-copy-trace, profiles/r06_track_frame.txt),
// while a kernel behind a kernel starts at once. Every lane moves 16 bytes; 75 workgroups keep 300 KB of reads in
// flight over the link.
typedef unsigned int ingest_u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void ingest_kernel(const ingest_u32x4* __restrict__ src, ingest_u32x4* __restrict__ dst, unsigned n16,
                                                     const uint8_t* __restrict__ src_tail, uint8_t* __restrict__ dst_tail, unsigned n_tail,
                                                     const ingest_u32x4* __restrict__ src2, ingest_u32x4* __restrict__ dst2, unsigned n16_2) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i < n16) dst[i] = __builtin_nontemporal_load(src + i);
    else if (i - n16 < n16_2) dst2[i - n16] = __builtin_nontemporal_load(src2 + (i - n16));
    if (i < n_tail) dst_tail[i] = src_tail[i];
}

hipError_t ingest_launch(const void* src, void* dst, size_t bytes, const void* src2, void* dst2, size_t bytes2, hipStream_t stream) {
    // src, dst 16-byte aligned; bytes2 a multiple of 16 (the caller's pinned block and device scratch: 256-byte sections)
    if ((((size_t)src | (size_t)dst | (size_t)src2 | (size_t)dst2) & 15) || (bytes2 & 15) || bytes >= (1ull << 35) || bytes2 >= (1ull << 35))
        return hipErrorInvalidValue;
    const unsigned n16 = (unsigned)(bytes / 16), n_tail = (unsigned)(bytes & 15), n16_2 = (unsigned)(bytes2 / 16);
    const unsigned total = n16 + n16_2 > n_tail ? n16 + n16_2 : n_tail;
    if (!total) return hipSuccess;
    hipLaunchKernelGGL(ingest_kernel, dim3((total + 255u) / 256u), dim3(256), 0, stream, (const ingest_u32x4*)src, (ingest_u32x4*)dst, n16,
                       (const uint8_t*)src + (size_t)n16 * 16, (uint8_t*)dst + (size_t)n16 * 16, n_tail, (const ingest_u32x4*)src2, (ingest_u32x4*)dst2, n16_2);
    return hipGetLastError();
}

hipError_t pyrdown_launch(uint8_t* pyr, size_t pyr_pitch, int n_images, int sw, int sh, int sstride,
                          size_t soff, int dstride, size_t doff, hipStream_t stream) {
    if (n_images <= 0) return hipSuccess;
    PyrDownArgs a;
    a.pyr = pyr; a.pyr_pitch = pyr_pitch; a.n_images = n_images;
    a.sw = sw; a.sh = sh; a.sstride = sstride; a.soff = soff;
    a.dw = (sw + 1) / 2; a.dh = (sh + 1) / 2; a.dstride = dstride; a.doff = doff;
    const int ty = (a.dh + PD_ROWS - 1) / PD_ROWS;          // row chunks of PD_ROWS output rows per thread
    const int tx = (a.dw + 3) / 4;                          // strips of 4 output columns
    // gridDim.z is limited to 65535 images per launch
    for (int i0 = 0; i0 < n_images; i0 += 65535) {
        const int nz = (n_images - i0 < 65535) ? n_images - i0 : 65535;
        PyrDownArgs b = a;
        b.pyr = pyr + (size_t)i0 * pyr_pitch;
        const dim3 grid((unsigned)((tx * ty + 255) / 256), 1u, (unsigned)nz);
        hipLaunchKernelGGL(pyrdown_kernel, grid, dim3(256), 0, stream, b, tx, ty);
    }
    return hipGetLastError();
}

}  // namespace dsdtm
