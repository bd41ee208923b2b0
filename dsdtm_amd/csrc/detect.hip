// detect.hip — the per-cell part of Feature_detector::detect on packed device pyramids.
//
// Replaces reference src/Feature_detection.cpp:75-108 with the vendored Thirdparty/fast it calls:
// fast_corner_detect_10_sse2 (faster_corner_10_sse.cpp), fast_corner_score_10 (fast_10_score.cpp),
// fast_nonmax_3x3 (nonmax_3x3.cpp), then Feature_detector::shiTomasiScore (:157-198) and the
// best-corner-per-grid-cell selection over all pyramid levels (:94-107). The order-dependent rest of
// detect() (sort, mask discs, Max_fts cap, :110-150) is bookkeeping over <= grid_cols*grid_rows
// corners and stays on the host (dsdtm_amd/feature_detection.py, like SearchLocalPoints).
//
// MI355X design: two dense, bandwidth-bound stencil passes over the u8 pyramid instead of the CPU's
// corner lists. Pass 1 writes a u8 score map (0 = no corner; the FAST score is the closed form
// max over the 16 arcs of 10 contiguous ring pixels of the smallest |difference| on the arc, minus 1 —
// what the generated decision trees compute; sliding-window minima by doubling, no branches). Pass 2
// does the 3x3 non-maximum suppression on the map ("suppressed iff a neighbouring corner scores >="),
// the Shi-Tomasi score of the survivors and a 64-bit atomicMax per grid cell whose key orders by
// (score, then first in the reference's level/raster order), which reproduces "the first corner with
// the strictly greater score wins" of the sequential loop.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_math.h"
#include "kernels.h"

namespace dsdtm {

__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }

// (one launch covers all levels: blockIdx.z = level, the grid has level 0's extent and the blocks
// outside a smaller level leave at once)
__global__ __launch_bounds__(256) void fast_score_kernel(const DetectArgs a) {
    const int level = blockIdx.z;
    const LevelGeom lg = a.lv[level];
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= lg.w || y >= lg.h) return;
    uint8_t* __restrict__ out = a.score + lg.off + (size_t)y * lg.stride + x;
    // faster_corner_10_sse.cpp:23-186: rows 3..h-4, columns 3..w-4
    if (x < 3 || x >= lg.w - 3 || y < 3 || y >= lg.h - 3) { *out = 0; return; }
    const uint8_t* __restrict__ p = a.pyr + lg.off + (size_t)y * lg.stride + x;
    const int st = lg.stride;
    const int c = *p;
    int d[16];   // ring in the order of fast_10_score.cpp:3146-3163
    d[0] = p[3 * st];       d[1] = p[3 * st + 1];   d[2] = p[2 * st + 2];   d[3] = p[st + 3];
    d[4] = p[3];            d[5] = p[-st + 3];      d[6] = p[-2 * st + 2];  d[7] = p[-3 * st + 1];
    d[8] = p[-3 * st];      d[9] = p[-3 * st - 1];  d[10] = p[-2 * st - 2]; d[11] = p[-st - 3];
    d[12] = p[-3];          d[13] = p[st - 3];      d[14] = p[2 * st - 2];  d[15] = p[3 * st - 1];
    // min and max over every window of 10 contiguous ring pixels (circular), by doubling: 2, 4, 8, 8+2
    int mn2[16], mx2[16], mn4[16], mx4[16], mn8[16], mx8[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { mn2[i] = imin(d[i], d[(i + 1) & 15]); mx2[i] = imax(d[i], d[(i + 1) & 15]); }
#pragma unroll
    for (int i = 0; i < 16; ++i) { mn4[i] = imin(mn2[i], mn2[(i + 2) & 15]); mx4[i] = imax(mx2[i], mx2[(i + 2) & 15]); }
#pragma unroll
    for (int i = 0; i < 16; ++i) { mn8[i] = imin(mn4[i], mn4[(i + 4) & 15]); mx8[i] = imax(mx4[i], mx4[(i + 4) & 15]); }
    int best_min = 0, worst_max = 255;   // brighter arcs: max over arcs of (min - c); darker: max of (c - max)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        best_min = imax(best_min, imin(mn8[i], mn2[(i + 8) & 15]));
        worst_max = imin(worst_max, imax(mx8[i], mx2[(i + 8) & 15]));
    }
    const int margin = imax(best_min - c, c - worst_max);     // corner at barrier b iff margin > b
    *out = margin > a.barrier ? (uint8_t)(margin - 1) : (uint8_t)0;      // fast_10_score.cpp: largest such b
}

// Feature_detector::shiTomasiScore (:157-198) for the corner (u, v), computed by the WHOLE wave: lane =
// one pixel of the 8x8 box (the reference's 64-iteration loop), three integer wave sums. The gradient
// sums are integers < 2^24 (exact in float in any order); the rest follows the reference's float
// expression without contraction. u, v are wave-uniform.
#pragma clang fp contract(off)
__device__ __forceinline__ float shi_tomasi_wave(const uint8_t* __restrict__ img, int w, int h, int stride, int u, int v, int lane) {
    const int x_min = u - 4, x_max = u + 4, y_min = v - 4, y_max = v + 4;
    if (x_min < 1 || x_max >= w - 1 || y_min < 1 || y_max >= h - 1) return 0.0f;   // :173
    const uint8_t* __restrict__ p = img + (size_t)stride * (y_min + (lane >> 3)) + x_min + (lane & 7);
    const int dx = (int)p[1] - (int)p[-1];
    const int dy = (int)p[stride] - (int)p[-stride];
    const int sxx = wave_sum_i32(dx * dx), syy = wave_sum_i32(dy * dy), sxy = wave_sum_i32(dx * dy);
    const float dXX = (float)sxx / 128.0f, dYY = (float)syy / 128.0f, dXY = (float)sxy / 128.0f;   // / (2.0 * box_area), exact
    const float tr = dXX + dYY;
    const float disc = tr * tr - 4 * (dXX * dYY - dXY * dXY);
    return 0.5f * (tr - sqrtf(disc));
}

__global__ __launch_bounds__(256) void fast_select_kernel(const DetectArgs a) {
    const int level = blockIdx.z;
    const LevelGeom lg = a.lv[level];
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    const int lane = threadIdx.x & 63;
    if (y >= lg.h || (int)(blockIdx.x * blockDim.x) >= lg.w) return;       // whole block outside the level (block-uniform)
    bool cand = false;
    int k = 0;
    if (x >= 3 && x < lg.w - 3 && y >= 3 && y < lg.h - 3) {
        const uint8_t* __restrict__ sm = a.score + lg.off + (size_t)y * lg.stride + x;
        const int s = *sm;
        if (s) {
            // nonmax_3x3.cpp:47-106: suppressed iff a neighbouring corner scores >= (ties suppress both)
            const int st = lg.stride;
            const int n0 = sm[-st - 1], n1 = sm[-st], n2 = sm[-st + 1], n3 = sm[-1], n4 = sm[1], n5 = sm[st - 1], n6 = sm[st], n7 = sm[st + 1];
            if (imax(imax(imax(n0, n1), imax(n2, n3)), imax(imax(n4, n5), imax(n6, n7))) < s) {
                if (a.keep) a.keep[lg.off + (size_t)y * lg.stride + x] = 1;
                const int scale = 1 << level;
                k = ((y * scale) / a.cell_size) * a.grid_cols + (x * scale) / a.cell_size;      // :97-98
                cand = k >= 0 && k < a.grid_cols * a.grid_rows && !(a.occupied && a.occupied[k]);   // :100
            }
        }
    }
    // the survivors of this wave's 64 pixels, one after the other, each scored by all 64 lanes
    unsigned long long todo = __ballot(cand);
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int cx = __builtin_amdgcn_readlane(x, src);
        const float sc = shi_tomasi_wave(a.pyr + lg.off, lg.w, lg.h, lg.stride, cx, y, lane);   // :103
        if (lane == src && sc > a.detection_threshold) {                                        // :104 vs the initial score (:74)
            // max score wins; among equal scores the first in (level, row, column) order, as the sequential
            // loop's strict '>' leaves it
            const unsigned order = ((unsigned)level << 28) | ((unsigned)y << 14) | (unsigned)x;
            const unsigned long long key = ((unsigned long long)__float_as_uint(sc) << 32) | (unsigned long long)(0xffffffffu - order);
            atomicMax(a.cell_key + k, key);
        }
    }
}

hipError_t detect_launch(const DetectArgs& a, int levels, hipStream_t stream) {
    const dim3 grid((unsigned)((a.lv[0].w + 255) / 256), (unsigned)a.lv[0].h, (unsigned)levels);
    hipLaunchKernelGGL(fast_score_kernel, grid, dim3(256), 0, stream, a);
    hipLaunchKernelGGL(fast_select_kernel, grid, dim3(256), 0, stream, a);
    return hipGetLastError();
}

}  // namespace dsdtm
