// pyrdown.hip — Frame::ComputeImagePyramid's cv::pyrDown chain on packed device pyramids.
//
// Replaces reference src/Frame.cpp:74-81 (cv::pyrDown per level; OpenCV 2.4 8-bit semantics:
// separable [1 4 6 4 1], BORDER_REFLECT_101, (sum+128)>>8, output ((w+1)/2,(h+1)/2)).
// Integer arithmetic, bit-exact. HBM-bound: each thread produces 4 horizontally adjacent output
// pixels (one dword store) from a 5-row x 12-byte input window read as aligned dwords; border
// threads take a byte-wise reflected path.
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

__global__ __launch_bounds__(256) void pyrdown_kernel(const PyrDownArgs a) {
    const int img = blockIdx.z;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    const int x4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;     // first of 4 output columns
    if (y >= a.dh || x4 >= a.dw) return;
    const uint8_t* __restrict__ src = a.pyr + (size_t)img * a.pyr_pitch + a.soff;
    uint8_t* __restrict__ dst = a.pyr + (size_t)img * a.pyr_pitch + a.doff;

    int rows[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) rows[k] = reflect101(2 * y + k - 2, a.sh);

    int acc[4] = {0, 0, 0, 0};
    const int xs = 2 * x4 - 2;                       // first input column needed (x4 is a multiple of 4)
    const int wk[5] = {1, 4, 6, 4, 1};
    // Fast path: the 11 input columns xs..xs+10 of each of the 5 rows come from ONE aligned 16-byte
    // load. Interior threads read columns (xs-2)..(xs+13) and use bytes 2..12; the left-border thread
    // (xs = -2) reads columns 0..15 and mirrors columns -2,-1 -> 2,1; the right-border thread of an
    // even-width image needs column sw -> sw-2 (BORDER_REFLECT_101). Keeping the border lanes on this
    // path matters: one lane on the byte-wise path stalls its whole wave.
    const bool aligned_ok = ((a.sstride & 3) == 0) && ((((size_t)src) & 3) == 0) && ((a.sw & 1) == 0) && a.sw >= 16 &&
                            (x4 + 3 < a.dw);
    const bool left = (x4 == 0);
    const bool right = (xs + 10 >= a.sw);             // only column xs+10 == sw can be outside
    if (aligned_ok && (!right || xs + 10 == a.sw)) {
        const int base = left ? 0 : (xs - 2);         // multiple of 4
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const uint32_t* __restrict__ p32 = (const uint32_t*)(src + (size_t)rows[k] * a.sstride + base);
            const uint32_t d0 = p32[0], d1 = p32[1], d2 = p32[2];
            // the 4th dword is only needed by interior/left threads (columns base+12..); for the
            // right-border thread it would start past the row end, so it is not read there
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
            if (right) w[10] = w[8];                  // column sw -> sw-2
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                int h = 0;
#pragma unroll
                for (int j = 0; j < 5; ++j) h += wk[j] * w[2 * o + j];
                acc[o] += wk[k] * h;
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const uint8_t* __restrict__ row = src + (size_t)rows[k] * a.sstride;
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
    }
    uint8_t out[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) out[o] = (uint8_t)((acc[o] + 128) >> 8);
    uint8_t* d = dst + (size_t)y * a.dstride + x4;
    if (x4 + 3 < a.dw && (((size_t)d) & 3) == 0) {
        *(uint32_t*)d = (uint32_t)out[0] | ((uint32_t)out[1] << 8) | ((uint32_t)out[2] << 16) | ((uint32_t)out[3] << 24);
    } else {
        for (int o = 0; o < 4 && x4 + o < a.dw; ++o) d[o] = out[o];
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
        const dim3 grid((unsigned)((tx + 63) / 64), (unsigned)((a.dh + 3) / 4), (unsigned)nz);
        hipLaunchKernelGGL(pyrdown_kernel, grid, block, 0, stream, b);
    }
    return hipGetLastError();
}

}  // namespace dsdtm
