// pyrdown.hip — Frame::ComputeImagePyramid's cv::pyrDown chain on packed device pyramids.
//
// Replaces reference src/Frame.cpp:74-81 (cv::pyrDown per level; OpenCV 2.4 8-bit semantics:
// separable [1 4 6 4 1], BORDER_REFLECT_101, (sum+128)>>8, output ((w+1)/2,(h+1)/2)).
// Integer arithmetic, bit-exact. HBM-bound: each thread owns a strip of 4 output columns (one dword
// store per output row) and walks down 8 output rows with a sliding window of horizontally
// filtered input rows (v_dot4_u32_u8), each read once as one 16-byte load, vertical pass on packed
// u16; odd sizes take a byte-wise reflected path.
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
// ~50 instructions per output row. Threads are numbered linearly over (strip, row chunk), so waves
// are full whatever the level width (640 px = 80 strips used to leave the second block column at 16
// of 64 lanes).
#ifndef SA_PD_ROWS
#define SA_PD_ROWS 8
#endif
constexpr int PD_ROWS = SA_PD_ROWS;
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ u16x2 pk(uint32_t lo, uint32_t hi) {
    u16x2 v; v.x = (unsigned short)lo; v.y = (unsigned short)hi; return v;
}

// one 16-byte load of an input row at `base` (a multiple of 4). The 4th dword of the right-border strip starts at
// the row end: it lies inside the pyramid (a source level is never the last level) and is not used.
__device__ __forceinline__ uint4 pd_load(const uint8_t* __restrict__ src, int sstride, int row, int base) {
    const uint32_t* __restrict__ p32 = (const uint32_t*)(src + (size_t)row * sstride + base);
    return make_uint4(p32[0], p32[1], p32[2], p32[3]);
}
// horizontal [1 4 6 4 1] of a loaded input row for the 4 output columns x4..x4+3 (aligned fast path):
// hA = (h0, h1), hB = (h2, h3)
__device__ __forceinline__ void pd_filter(const uint4 d, bool left, bool right, u16x2& hA, u16x2& hB) {
    const uint32_t d0 = d.x, d1 = d.y, d2 = d.z, d3 = d.w;
    // e0..e3 = columns xs-2 .. xs+13 with xs = 2*x4 - 2; interior strips load exactly that
    // (base = xs-2); the left-border strip (xs = -2) loads columns 0..15 and mirrors -4..-1 -> 4..1
    const uint32_t e0 = left ? __builtin_amdgcn_perm(d1, d0, 0x01020304u) : d0;   // bytes [col4, col3, col2, col1]
    const uint32_t e1 = left ? d0 : d1;
    const uint32_t e2 = left ? d1 : d2;
    const uint32_t e3 = left ? d2 : d3;
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
__device__ __forceinline__ void pd_hrow(const uint8_t* __restrict__ src, int sstride, int row, int base, bool left,
                                        bool right, u16x2& hA, u16x2& hB) {
    pd_filter(pd_load(src, sstride, row, base), left, right, hA, hB);
}

// vertical [1 4 6 4 1] + rounding of one packed pair: ((r0 + r4) + 4 (r1 + r3) + 6 r2 + 128) >> 8
__device__ __forceinline__ u16x2 pd_vert(u16x2 r0, u16x2 r1, u16x2 r2, u16x2 r3, u16x2 r4) {
    const u16x2 four = {4, 4}, six = {6, 6}, half = {128, 128}, eight = {8, 8};
    u16x2 v = (r1 + r3) * four + (r0 + r4);
    v = r2 * six + v;
    return (v + half) >> eight;
}

// XCD-aware block numbering. Workgroups go to the 8 XCDs round-robin by their linear id, and each XCD has its own
// L2: with the plain numbering the blocks of ONE image — whose row chunks share 3 halo rows with their
// neighbours — are spread over all eight L2s and every halo row is fetched from HBM by two of them. The
// remapped id gives each XCD a contiguous range of (image, block) pairs, so an image's blocks meet in one L2.
// SA_PD_XCD=0 restores the plain numbering (A/B).
// all input rows of a thread's chunk loaded before the filtering starts (see the fast path)
#ifndef SA_PD_PREFETCH
#define SA_PD_PREFETCH 1
#endif
#ifndef SA_PD_XCD
#define SA_PD_XCD 1
#endif
__device__ __forceinline__ void pd_block(int& img, int& bx) {
    const unsigned nb = gridDim.x, total = gridDim.x * gridDim.z;
    unsigned lb = blockIdx.z * nb + blockIdx.x;
#if SA_PD_XCD
    const unsigned q = total / 8u;
    if (lb < q * 8u) lb = (lb % 8u) * q + lb / 8u;
#endif
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

    // Fast path: the 11 input columns xs..xs+10 of a row come from ONE 16-byte load. Interior threads
    // read columns (xs-2)..(xs+13) and use bytes 2..12; the left-border thread (xs = -2) reads
    // columns 0..15 and mirrors; the right-border thread of an even-width image needs column
    // sw -> sw-2. Keeping the border lanes on this path matters: one lane on the byte-wise path
    // stalls its whole wave.
    const bool left = (x4 == 0);
    const bool right = (xs + 10 >= a.sw);             // only column xs+10 == sw can be outside
    const bool fast = ((a.sstride & 3) == 0) && ((((size_t)src) & 3) == 0) && ((a.sw & 1) == 0) && a.sw >= 16 &&
                      (x4 + 3 < a.dw) && (!right || xs + 10 == a.sw) && (((size_t)(dst + x4)) & 3) == 0 && (a.dstride & 3) == 0;
    if (fast) {
        const int base = left ? 0 : (xs - 2);         // multiple of 4
        u16x2 wA[5], wB[5];                           // horizontally filtered rows 2y-2 .. 2y+2
#if SA_PD_PREFETCH
        // all 2*PD_ROWS + 3 input rows of the chunk in flight before the first one is filtered: the sliding-window
        // loop below otherwise waits for two rows per output row, and ~2 loads per wave in flight at 7 waves per
        // SIMD is short of what keeps HBM busy (Little: ~47 KB per CU at 6 TB/s and 2 us)
        uint4 raw[2 * PD_ROWS + 3];
#pragma unroll
        for (int k = 0; k < 2 * PD_ROWS + 3; ++k)
            raw[k] = pd_load(src, a.sstride, reflect101(min(2 * y0 - 2 + k, 2 * a.dh), a.sh), base);
        pd_filter(raw[0], left, right, wA[0], wB[0]);
        pd_filter(raw[1], left, right, wA[1], wB[1]);
        pd_filter(raw[2], left, right, wA[2], wB[2]);
#pragma unroll
        for (int r = 0; r < PD_ROWS; ++r) {
            const int y = y0 + r;
            if (y >= a.dh) break;
            pd_filter(raw[2 * r + 3], left, right, wA[(2 * r + 3) % 5], wB[(2 * r + 3) % 5]);
            pd_filter(raw[2 * r + 4], left, right, wA[(2 * r + 4) % 5], wB[(2 * r + 4) % 5]);
            const u16x2 oA = pd_vert(wA[(2 * r) % 5], wA[(2 * r + 1) % 5], wA[(2 * r + 2) % 5], wA[(2 * r + 3) % 5], wA[(2 * r + 4) % 5]);
            const u16x2 oB = pd_vert(wB[(2 * r) % 5], wB[(2 * r + 1) % 5], wB[(2 * r + 2) % 5], wB[(2 * r + 3) % 5], wB[(2 * r + 4) % 5]);
            const uint32_t packed = (uint32_t)oA.x | ((uint32_t)oA.y << 8) | ((uint32_t)oB.x << 16) | ((uint32_t)oB.y << 24);
            *(uint32_t*)(dst + (size_t)y * a.dstride + x4) = packed;
        }
        return;
#endif
        pd_hrow(src, a.sstride, reflect101(2 * y0 - 2, a.sh), base, left, right, wA[0], wB[0]);
        pd_hrow(src, a.sstride, reflect101(2 * y0 - 1, a.sh), base, left, right, wA[1], wB[1]);
        pd_hrow(src, a.sstride, reflect101(2 * y0, a.sh), base, left, right, wA[2], wB[2]);
#pragma unroll
        for (int r = 0; r < PD_ROWS; ++r) {
            const int y = y0 + r;
            if (y >= a.dh) break;
            // window slot of input row 2y+k-2 is (2r + k) % 5
            pd_hrow(src, a.sstride, reflect101(2 * y + 1, a.sh), base, left, right, wA[(2 * r + 3) % 5], wB[(2 * r + 3) % 5]);
            pd_hrow(src, a.sstride, reflect101(2 * y + 2, a.sh), base, left, right, wA[(2 * r + 4) % 5], wB[(2 * r + 4) % 5]);
            const u16x2 oA = pd_vert(wA[(2 * r) % 5], wA[(2 * r + 1) % 5], wA[(2 * r + 2) % 5], wA[(2 * r + 3) % 5], wA[(2 * r + 4) % 5]);
            const u16x2 oB = pd_vert(wB[(2 * r) % 5], wB[(2 * r + 1) % 5], wB[(2 * r + 2) % 5], wB[(2 * r + 3) % 5], wB[(2 * r + 4) % 5]);
            const uint32_t packed = (uint32_t)oA.x | ((uint32_t)oA.y << 8) | ((uint32_t)oB.x << 16) | ((uint32_t)oB.y << 24);
            *(uint32_t*)(dst + (size_t)y * a.dstride + x4) = packed;
        }
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
