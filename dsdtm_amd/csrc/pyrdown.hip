// pyrdown.hip — Frame::ComputeImagePyramid's cv::pyrDown chain on packed device pyramids.
//
// Replaces reference src/Frame.cpp:74-81 (cv::pyrDown per level; OpenCV 2.4 8-bit semantics:
// separable [1 4 6 4 1], BORDER_REFLECT_101, (sum+128)>>8, output ((w+1)/2,(h+1)/2)).
// Integer arithmetic, bit-exact. HBM-bound: each thread owns a strip of 4 output columns (one dword
// store per output row) and walks down 8 output rows with a sliding window of horizontally
// filtered input rows, each read once as one aligned 16-byte load; odd sizes take a byte-wise
// reflected path.
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
// window of horizontally filtered rows: every input row of the strip is loaded (one aligned 16-byte
// load) and filtered ONCE and reused by the 2-3 output rows it contributes to.
constexpr int PD_ROWS = 8;

// horizontal [1 4 6 4 1] of input row `row` for the 4 output columns x4..x4+3 (aligned fast path)
__device__ __forceinline__ void pd_hrow(const uint8_t* __restrict__ src, int sstride, int row, int base, bool left,
                                        bool right, int* h) {
    const uint32_t* __restrict__ p32 = (const uint32_t*)(src + (size_t)row * sstride + base);
    const uint32_t d0 = p32[0], d1 = p32[1], d2 = p32[2];
    // the 4th dword is only needed by interior/left threads; for the right-border thread it would
    // start past the row end, so it is not read there
    const uint32_t d3 = right ? 0u : p32[3];
    int px[16];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        px[j] = (d0 >> (8 * j)) & 0xff; px[4 + j] = (d1 >> (8 * j)) & 0xff;
        px[8 + j] = (d2 >> (8 * j)) & 0xff; px[12 + j] = (d3 >> (8 * j)) & 0xff;
    }
    int w[11];                                // w[j] = pixel at column xs + j (reflected)
#pragma unroll
    for (int j = 0; j < 11; ++j) w[j] = left ? ((j < 2) ? px[2 - j] : px[j - 2]) : px[j + 2];
    if (right) w[10] = w[8];                  // column sw -> sw-2 (BORDER_REFLECT_101)
#pragma unroll
    for (int o = 0; o < 4; ++o) h[o] = w[2 * o] + w[2 * o + 4] + 4 * (w[2 * o + 1] + w[2 * o + 3]) + 6 * w[2 * o + 2];
}

__global__ __launch_bounds__(256) void pyrdown_kernel(const PyrDownArgs a) {
    const int img = blockIdx.z;
    const int y0 = (blockIdx.y * blockDim.y + threadIdx.y) * PD_ROWS;   // first output row of this thread
    const int x4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;         // first of 4 output columns
    if (y0 >= a.dh || x4 >= a.dw) return;
    const uint8_t* __restrict__ src = a.pyr + (size_t)img * a.pyr_pitch + a.soff;
    uint8_t* __restrict__ dst = a.pyr + (size_t)img * a.pyr_pitch + a.doff;
    const int xs = 2 * x4 - 2;                       // first input column needed (x4 is a multiple of 4)

    // Fast path: the 11 input columns xs..xs+10 of a row come from ONE aligned 16-byte load. Interior
    // threads read columns (xs-2)..(xs+13) and use bytes 2..12; the left-border thread (xs = -2) reads
    // columns 0..15 and mirrors columns -2,-1 -> 2,1; the right-border thread of an even-width image
    // needs column sw -> sw-2. Keeping the border lanes on this path matters: one lane on the
    // byte-wise path stalls its whole wave.
    const bool left = (x4 == 0);
    const bool right = (xs + 10 >= a.sw);             // only column xs+10 == sw can be outside
    const bool fast = ((a.sstride & 3) == 0) && ((((size_t)src) & 3) == 0) && ((a.sw & 1) == 0) && a.sw >= 16 &&
                      (x4 + 3 < a.dw) && (!right || xs + 10 == a.sw) && (((size_t)(dst + x4)) & 3) == 0 && (a.dstride & 3) == 0;
    if (fast) {
        const int base = left ? 0 : (xs - 2);         // multiple of 4
        int win[5][4];                                // horizontally filtered rows 2y-2 .. 2y+2
        pd_hrow(src, a.sstride, reflect101(2 * y0 - 2, a.sh), base, left, right, win[0]);
        pd_hrow(src, a.sstride, reflect101(2 * y0 - 1, a.sh), base, left, right, win[1]);
        pd_hrow(src, a.sstride, reflect101(2 * y0, a.sh), base, left, right, win[2]);
#pragma unroll
        for (int r = 0; r < PD_ROWS; ++r) {
            const int y = y0 + r;
            if (y >= a.dh) break;
            // window slot of input row 2y+k-2 is (2r + k) % 5
            pd_hrow(src, a.sstride, reflect101(2 * y + 1, a.sh), base, left, right, win[(2 * r + 3) % 5]);
            pd_hrow(src, a.sstride, reflect101(2 * y + 2, a.sh), base, left, right, win[(2 * r + 4) % 5]);
            uint32_t packed = 0;
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                const int v = win[(2 * r) % 5][o] + win[(2 * r + 4) % 5][o] + 4 * (win[(2 * r + 1) % 5][o] + win[(2 * r + 3) % 5][o]) +
                              6 * win[(2 * r + 2) % 5][o];
                packed |= (uint32_t)((v + 128) >> 8) << (8 * o);
            }
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

hipError_t pyrdown_launch(uint8_t* pyr, size_t pyr_pitch, int n_images, int sw, int sh, int sstride,
                          size_t soff, int dstride, size_t doff, hipStream_t stream) {
    if (n_images <= 0) return hipSuccess;
    PyrDownArgs a;
    a.pyr = pyr; a.pyr_pitch = pyr_pitch; a.n_images = n_images;
    a.sw = sw; a.sh = sh; a.sstride = sstride; a.soff = soff;
    a.dw = (sw + 1) / 2; a.dh = (sh + 1) / 2; a.dstride = dstride; a.doff = doff;
    const dim3 block(64, 4);
    const int tx = (a.dw + 3) / 4;
    // gridDim.z is limited to 65535 images per launch
    for (int i0 = 0; i0 < n_images; i0 += 65535) {
        const int nz = (n_images - i0 < 65535) ? n_images - i0 : 65535;
        PyrDownArgs b = a;
        b.pyr = pyr + (size_t)i0 * pyr_pitch;
        const int ty = (a.dh + PD_ROWS - 1) / PD_ROWS;      // row chunks of PD_ROWS output rows per thread
        const dim3 grid((unsigned)((tx + 63) / 64), (unsigned)((ty + 3) / 4), (unsigned)nz);
        hipLaunchKernelGGL(pyrdown_kernel, grid, block, 0, stream, b);
    }
    return hipGetLastError();
}

}  // namespace dsdtm
