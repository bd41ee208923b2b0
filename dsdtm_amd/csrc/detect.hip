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

// FAST-10 score of one pixel from its centre and ring (order of fast_10_score.cpp:3146-3163): min and max over
// every window of 10 contiguous ring pixels (circular) by doubling — 2, 4, 8, 8 + 2 — then
// margin = max(best window minimum - c, c - best window maximum): a corner at barrier b iff margin > b, and the
// score (fast_10_score.cpp: the largest such b) is margin - 1. T = int (one pixel) or a packed pair of u16 (two
// pixels per instruction: v_pk_min_u16 / v_pk_max_u16).
template <typename T, typename MinF, typename MaxF>
__device__ __forceinline__ void fast10_extrema(const T* d, T& best_min, T& worst_max, MinF mn, MaxF mx) {
    T mn2[16], mx2[16], mn4[16], mx4[16], mn8[16], mx8[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { mn2[i] = mn(d[i], d[(i + 1) & 15]); mx2[i] = mx(d[i], d[(i + 1) & 15]); }
#pragma unroll
    for (int i = 0; i < 16; ++i) { mn4[i] = mn(mn2[i], mn2[(i + 2) & 15]); mx4[i] = mx(mx2[i], mx2[(i + 2) & 15]); }
#pragma unroll
    for (int i = 0; i < 16; ++i) { mn8[i] = mn(mn4[i], mn4[(i + 4) & 15]); mx8[i] = mx(mx4[i], mx4[(i + 4) & 15]); }
    best_min = mn(mn8[0], mn2[8]);
    worst_max = mx(mx8[0], mx2[8]);
#pragma unroll
    for (int i = 1; i < 16; ++i) {
        best_min = mx(best_min, mn(mn8[i], mn2[(i + 8) & 15]));
        worst_max = mn(worst_max, mx(mx8[i], mx2[(i + 8) & 15]));
    }
}

// ---- one thread per pixel: levels whose width is not a multiple of 4 ---------------------------------------------
// (grid: x = pixel columns of level 0 / 256, y = rows of level 0, z = frame * levels + level; blocks outside a smaller
// level leave at once)
__global__ __launch_bounds__(256) void fast_score_kernel(const DetectArgs a, unsigned level_mask) {
    const int level = blockIdx.z % a.levels, frame = blockIdx.z / a.levels;
    if (!((level_mask >> level) & 1u)) return;
    const LevelGeom lg = a.lv[level];
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= lg.w || y >= lg.h) return;
    uint8_t* __restrict__ out = a.score + (size_t)frame * a.pyr_pitch + lg.off + (size_t)y * lg.stride + x;
    // faster_corner_10_sse.cpp:23-186: rows 3..h-4, columns 3..w-4
    if (x < 3 || x >= lg.w - 3 || y < 3 || y >= lg.h - 3) { *out = 0; return; }
    const uint8_t* __restrict__ p = a.pyr + (size_t)frame * a.pyr_pitch + lg.off + (size_t)y * lg.stride + x;
    const int st = lg.stride;
    const int c = *p;
    int d[16];   // ring in the order of fast_10_score.cpp:3146-3163
    d[0] = p[3 * st];       d[1] = p[3 * st + 1];   d[2] = p[2 * st + 2];   d[3] = p[st + 3];
    d[4] = p[3];            d[5] = p[-st + 3];      d[6] = p[-2 * st + 2];  d[7] = p[-3 * st + 1];
    d[8] = p[-3 * st];      d[9] = p[-3 * st - 1];  d[10] = p[-2 * st - 2]; d[11] = p[-st - 3];
    d[12] = p[-3];          d[13] = p[st - 3];      d[14] = p[2 * st - 2];  d[15] = p[3 * st - 1];
    int best_min, worst_max;
    fast10_extrema<int>(d, best_min, worst_max, [](int u, int v) { return imin(u, v); }, [](int u, int v) { return imax(u, v); });
    const int margin = imax(best_min - c, c - worst_max);
    *out = margin > a.barrier ? (uint8_t)(margin - 1) : (uint8_t)0;
}

// ---- one thread per strip of 4 pixels x FS_ROWS rows: levels whose width and stride are multiples of 4 ------------
// A thread walks down FS_ROWS output rows with a sliding window of the seven image rows a ring touches; every row
// of the strip is fetched ONCE per thread (three aligned dwords: columns x - 4 .. x + 7 cover the ring columns x - 3
// .. x + 6 of the four pixels) instead of 17 byte loads per pixel, the four scores leave as one dword, and the
// min / max network runs on packed u16 pairs (two pixels per instruction). The ring bytes of a pixel pair come out
// of the row registers with ONE v_perm each (all selectors are compile-time constants). Tasks (strip, row chunk) are
// numbered linearly per level, so waves are full at every level width.
// XCD-aware block numbering. Workgroups go to the 8 XCDs round-robin by their linear id, each XCD has its own L2, and the
// row chunks of one level share 6 of their 10 image rows (score pass) / 2 of their 6 score rows (select pass) with their
// neighbours: with the plain numbering those rows are fetched from HBM by several L2s (measured: 2.0x the pyramid in the score
// pass). Here the blocks of ONE (frame, level) row of the grid that land on the same XCD take a CONTIGUOUS range of that row's
// block numbers — the remap stays inside the row, so every XCD still gets an eighth of every level (remapping the whole grid
// gave some XCDs the large levels and others the blocks that leave at once: traffic 2.57x -> 1.64x but 6 % slower).
// grid = (blocks, levels, frames).
// `na` = the blocks of this row that have work (the grid is as wide as the largest level needs; a smaller level's blocks
// beyond `na` leave at once): the remap runs among those — over the whole row it would put a small level's few active blocks on
// one or two XCDs (measured: 2x slower).
__device__ __forceinline__ int det_block_x(unsigned na) {
    if (blockIdx.x >= na) return (int)blockIdx.x;
    const unsigned off = (gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) & 7u;   // linear id of this row's first block, mod 8
    const unsigned r = (blockIdx.x + off) & 7u;                                         // this block's XCD
    unsigned start = 0;                                                                 // active blocks of this row on the XCDs before r
    for (unsigned k = 0; k < r; ++k) {
        const unsigned x0 = (k - off) & 7u;                                             // first block of the row on XCD k
        start += x0 < na ? (na - x0 + 7u) / 8u : 0u;
    }
    const unsigned x0 = (r - off) & 7u;
    return (int)(start + (blockIdx.x - x0) / 8u);
}

constexpr int FS_ROWS = 4;
typedef uint16_t fs_u16x2 __attribute__((ext_vector_type(2)));
struct FsRow { uint32_t d0, d1, d2; };          // columns x-4..x-1, x..x+3, x+4..x+7

// pixels (I, I + 1) of the strip, column offset DX: bytes 4 + I + DX and 5 + I + DX of the row, as a u16 pair
template <int I, int DX>
__device__ __forceinline__ fs_u16x2 fs_pair(const FsRow& r) {
    constexpr int B = 4 + I + DX;                 // 1 .. 9
    static_assert(B >= 1 && B + 1 <= 10, "ring column inside the three dwords");
    uint32_t v;
    if constexpr (B + 1 <= 7) v = __builtin_amdgcn_perm(r.d1, r.d0, 0x0c000c00u | (uint32_t)B | ((uint32_t)(B + 1) << 16));
    else v = __builtin_amdgcn_perm(r.d2, r.d1, 0x0c000c00u | (uint32_t)(B - 4) | ((uint32_t)(B - 3) << 16));
    return __builtin_bit_cast(fs_u16x2, v);
}
__device__ __forceinline__ fs_u16x2 fs_min(fs_u16x2 u, fs_u16x2 v) { return __builtin_elementwise_min(u, v); }
__device__ __forceinline__ fs_u16x2 fs_max(fs_u16x2 u, fs_u16x2 v) { return __builtin_elementwise_max(u, v); }

// scores of pixels (I, I + 1) of the strip in the low bytes of the two halves; w[k] = image row y - 3 + k
template <int I>
__device__ __forceinline__ uint32_t fs_score_pair(const FsRow* w, int barrier) {
    fs_u16x2 d[16];
    d[0] = fs_pair<I, 0>(w[6]);   d[1] = fs_pair<I, 1>(w[6]);   d[2] = fs_pair<I, 2>(w[5]);   d[3] = fs_pair<I, 3>(w[4]);
    d[4] = fs_pair<I, 3>(w[3]);   d[5] = fs_pair<I, 3>(w[2]);   d[6] = fs_pair<I, 2>(w[1]);   d[7] = fs_pair<I, 1>(w[0]);
    d[8] = fs_pair<I, 0>(w[0]);   d[9] = fs_pair<I, -1>(w[0]);  d[10] = fs_pair<I, -2>(w[1]); d[11] = fs_pair<I, -3>(w[2]);
    d[12] = fs_pair<I, -3>(w[3]); d[13] = fs_pair<I, -3>(w[4]); d[14] = fs_pair<I, -2>(w[5]); d[15] = fs_pair<I, -1>(w[6]);
    const fs_u16x2 c = fs_pair<I, 0>(w[3]);
    fs_u16x2 best_min, worst_max;
    fast10_extrema<fs_u16x2>(d, best_min, worst_max, fs_min, fs_max);
    // margin = max(best_min - c, c - worst_max) on unsigned halves: saturating differences (a negative one is never the maximum
    // unless both are, and then the margin is <= 0 and the pixel is no corner)
    const fs_u16x2 zero = {0, 0};
    const fs_u16x2 up = fs_max(best_min, c) - c, dn = c - fs_min(worst_max, c);
    const fs_u16x2 margin = fs_max(up, dn);
    const fs_u16x2 one = {1, 1};
    const fs_u16x2 bar = {(uint16_t)barrier, (uint16_t)barrier};
    const fs_u16x2 sc = fs_max(margin, one) - one;                     // margin - 1, 0 for margin 0
    const fs_u16x2 keep = (margin > bar) ? sc : zero;
    return __builtin_bit_cast(uint32_t, keep);
}

__global__ __launch_bounds__(256) void fast_score_strip_kernel(const DetectArgs a, unsigned level_mask) {
    const int level = blockIdx.y, frame = blockIdx.z;
    if (!((level_mask >> level) & 1u)) return;
    const LevelGeom lg = a.lv[level];
    const int strips = lg.w >> 2, chunks = (lg.h + FS_ROWS - 1) / FS_ROWS;
    const int bx = det_block_x((unsigned)(strips * chunks + (int)blockDim.x - 1) / blockDim.x);
    const int task = bx * (int)blockDim.x + (int)threadIdx.x;
    if (bx * (int)blockDim.x >= strips * chunks) return;               // block-uniform
    if (task >= strips * chunks) return;
    const int chunk = task / strips, q = task - chunk * strips;         // q: dword (strip) index inside a row
    const int y0 = chunk * FS_ROWS;
    const uint32_t* __restrict__ img = (const uint32_t*)(a.pyr + (size_t)frame * a.pyr_pitch);
    uint32_t* __restrict__ out = (uint32_t*)(a.score + (size_t)frame * a.pyr_pitch);
    const uint32_t last_dw = (uint32_t)(a.pyr_pitch >> 2) - 1u;
    const uint32_t sdw = (uint32_t)lg.stride >> 2, odw = lg.off >> 2;
    // row y of the strip; rows outside the level and the dwords beside the row ends are clamped to harmless
    // addresses — they only feed pixels whose score is forced to 0 below
    auto load_row = [&](int y) {
        const int yc = imin(imax(y, 0), lg.h - 1);
        const uint32_t rb = odw + (uint32_t)yc * sdw + (uint32_t)q;
        FsRow r;
        r.d0 = img[q > 0 ? rb - 1u : rb];
        r.d1 = img[rb];
        r.d2 = img[rb + 1u < last_dw ? rb + 1u : last_dw];
        return r;
    };
    FsRow w[7 + FS_ROWS - 1];
#pragma unroll
    for (int k = 0; k < 7 + FS_ROWS - 1; ++k) w[k] = load_row(y0 - 3 + k);          // all rows of the chunk in flight
    const int x = q << 2;
    // valid columns: 3 <= x + i < w - 3 (faster_corner_10_sse.cpp:23-186)
    uint32_t colmask = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) if (x + i >= 3 && x + i < lg.w - 3) colmask |= 0xffu << (8 * i);
#pragma unroll
    for (int r = 0; r < FS_ROWS; ++r) {
        const int y = y0 + r;
        const uint32_t p01 = fs_score_pair<0>(w + r, a.barrier), p23 = fs_score_pair<2>(w + r, a.barrier);
        // bytes: p01 = [s0, 0, s1, 0], p23 = [s2, 0, s3, 0] -> [s0, s1, s2, s3]
        uint32_t v = __builtin_amdgcn_perm(p23, p01, 0x06040200u) & colmask;
        if (y < 3 || y >= lg.h - 3) v = 0;
        if (y < lg.h) out[odw + (uint32_t)y * sdw + (uint32_t)q] = v;
    }
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

// PX = 4: one thread per dword of the score map — almost every dword is zero and leaves at once (batches: the pass is
// bound by the map's bytes). PX = 1: one thread per pixel — a wave then holds a quarter of the candidates, and the
// serial scoring loop below, which sets the duration of a one-frame launch, is a quarter as long (live tracker).
// (tasks numbered linearly per level; grid y = level, z = frame)
template <int PX>
__global__ __launch_bounds__(256) void fast_select_kernel(const DetectArgs a) {
    const int level = blockIdx.y, frame = blockIdx.z;
    const LevelGeom lg = a.lv[level];
    const int lane = threadIdx.x & 63;
    const int rowdw = (lg.w + PX - 1) / PX;                               // tasks per row (PX = 4: the dwords that hold its pixels)
    const int task = blockIdx.x * blockDim.x + threadIdx.x;
    if ((int)(blockIdx.x * blockDim.x) >= rowdw * lg.h) return;          // whole block outside the level (block-uniform)
    const int y = task / rowdw, q = task - y * rowdw;
    const uint8_t* __restrict__ smap = a.score + (size_t)frame * a.pyr_pitch + lg.off;
    const uint8_t* __restrict__ img = a.pyr + (size_t)frame * a.pyr_pitch + lg.off;
    const int cells = a.grid_cols * a.grid_rows;
    unsigned cand = 0;                                                   // bit i: pixel PX q + i is a surviving corner in a free cell
    int kcell[PX];
#pragma unroll
    for (int i = 0; i < PX; ++i) kcell[i] = 0;
    if (task < rowdw * lg.h && y >= 3 && y < lg.h - 3) {
        const uint8_t* __restrict__ srow = smap + (size_t)y * lg.stride + PX * q;
        uint32_t four = 0;                                               // the task's scores (byte-wise where the row is not dword-aligned)
        if (PX == 4 && ((lg.stride | lg.off) & 3) == 0) four = *(const uint32_t*)srow;
        else {
#pragma unroll
            for (int i = 0; i < PX; ++i) if (PX * q + i < lg.w) four |= (uint32_t)srow[i] << (8 * i);
        }
        if (four) {
#pragma unroll
            for (int i = 0; i < PX; ++i) {
                const int x = PX * q + i, s = (four >> (8 * i)) & 0xff;
                if (!s || x < 3 || x >= lg.w - 3) continue;
                // nonmax_3x3.cpp:47-106: suppressed iff a neighbouring corner scores >= (ties suppress both)
                const uint8_t* __restrict__ sm = srow + i;
                const int st = lg.stride;
                const int n0 = sm[-st - 1], n1 = sm[-st], n2 = sm[-st + 1], n3 = sm[-1], n4 = sm[1], n5 = sm[st - 1], n6 = sm[st], n7 = sm[st + 1];
                if (imax(imax(imax(n0, n1), imax(n2, n3)), imax(imax(n4, n5), imax(n6, n7))) < s) {
                    if (a.keep) a.keep[lg.off + (size_t)y * lg.stride + x] = 1;
                    const int scale = 1 << level;
                    const int k = ((y * scale) / a.cell_size) * a.grid_cols + (x * scale) / a.cell_size;       // :97-98
                    if (k >= 0 && k < cells && !(a.occupied && a.occupied[(size_t)frame * cells + k])) {     // :100
                        cand |= 1u << i; kcell[i] = k;
                    }
                }
            }
        }
    }
    // the survivors of this wave's pixels, one after the other, each scored by all 64 lanes
#pragma unroll
    for (int i = 0; i < PX; ++i) {
        unsigned long long todo = __ballot((cand >> i) & 1u);
        while (todo) {
            const int src = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const int cx = __builtin_amdgcn_readlane(PX * q + i, src), cy = __builtin_amdgcn_readlane(y, src);
            const float sc = shi_tomasi_wave(img, lg.w, lg.h, lg.stride, cx, cy, lane);                        // :103
            if (lane == src && sc > a.detection_threshold) {                                                    // :104 vs the initial score (:74)
                // max score wins; among equal scores the first in (level, row, column) order, as the sequential
                // loop's strict '>' leaves it
                const unsigned order = ((unsigned)level << 28) | ((unsigned)cy << 14) | (unsigned)cx;
                const unsigned long long key = ((unsigned long long)__float_as_uint(sc) << 32) | (unsigned long long)(0xffffffffu - order);
                atomicMax(a.cell_key + (size_t)frame * cells + kcell[i], key);
            }
        }
    }
}

// ---- batches: one thread per dword column x SEL_ROWS rows, four survivors scored per round ---------------------------
// fast_select_kernel<4> above is bound by LATENCY per wave, not by bytes or instructions: a wave reads its score dwords,
// then (dependent) the neighbours of the non-zero ones, then scores its ~2.5 survivors one after the other, each a round
// of loads -> wave sums -> atomic; 1.5 M waves of ~3.8 us for 256 frames (0.71 ms). Here a thread owns 4 x SEL_ROWS pixels:
// the SEL_ROWS + 2 score rows it needs are fetched up front as 8-byte words (no data-dependent second round: the 3x3
// non-maximum test runs on registers), and the survivors of a wave's 1024 pixels are scored FOUR per round — one per
// 16-lane DPP row, four pixels of the 8x8 Shi-Tomasi box per lane, row-local integer sums. Same decisions, same keys.
constexpr int SEL_ROWS = 4;
typedef uint32_t __attribute__((aligned(1))) SelU32;
struct __attribute__((packed, aligned(1))) SelU64 { uint32_t lo, hi; };
__device__ __forceinline__ unsigned long long sel_load64(const uint8_t* p) {
    const SelU64 w = *(const SelU64*)p;
    return (unsigned long long)w.lo | ((unsigned long long)w.hi << 32);
}
__device__ __forceinline__ int row_sum_i32(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false);   // row_ror:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x124, 0xf, 0xf, false);   // row_ror:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x122, 0xf, 0xf, false);   // row_ror:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x121, 0xf, 0xf, false);   // row_ror:1
    return v;
}

__global__ __launch_bounds__(256) void fast_select_rows_kernel(const DetectArgs a) {
    const int level = blockIdx.y, frame = blockIdx.z;
    const LevelGeom lg = a.lv[level];
    const int lane = threadIdx.x & 63;
    const int rowdw = lg.w >> 2;                                          // the launcher guarantees w, stride, off multiples of 4
    const int chunks = (lg.h + SEL_ROWS - 1) / SEL_ROWS;
    const int bx = det_block_x((unsigned)(rowdw * chunks + (int)blockDim.x - 1) / blockDim.x);
    const int task = bx * (int)blockDim.x + (int)threadIdx.x;
    if (bx * (int)blockDim.x >= rowdw * chunks) return;                  // whole block outside the level (block-uniform)
    const bool live = task < rowdw * chunks;
    const int chunk = live ? task / rowdw : 0, q = live ? task - chunk * rowdw : 0;
    const int y0 = chunk * SEL_ROWS;
    const uint8_t* __restrict__ smap = a.score + (size_t)frame * a.pyr_pitch + lg.off;
    const uint8_t* __restrict__ img = a.pyr + (size_t)frame * a.pyr_pitch + lg.off;
    const int cells = a.grid_cols * a.grid_rows;
    // score rows y0 - 1 .. y0 + SEL_ROWS, bytes 4 q - 1 .. 4 q + 4 of each in bits 0..47. Candidate rows are 3 .. h - 4, so
    // only rows 2 .. h - 3 are ever needed: rows outside read as 0 (and the 8-byte fetch of a row never leaves the level)
    unsigned long long rw[SEL_ROWS + 2];
#pragma unroll
    for (int k = 0; k < SEL_ROWS + 2; ++k) {
        const int y = y0 - 1 + k;
        rw[k] = 0ull;
        if (live && y >= 2 && y <= lg.h - 3) {
            const uint8_t* rp = smap + (size_t)y * lg.stride + 4 * q;
            rw[k] = q > 0 ? sel_load64(rp - 1) : (sel_load64(rp) << 8);
        }
    }
    // The 3x3 non-maximum test (nonmax_3x3.cpp:47-106: suppressed iff a neighbouring corner scores >=, ties suppress both) on
    // packed u16 pairs, two pixels per instruction: of every fetched row the byte pairs (0,1) (1,2) (2,3) (3,4) (4,5) by one
    // v_perm each; pixels (0,1) of a row are centred on pair (1,2), pixels (2,3) on pair (3,4). Per row in its three roles —
    // above / below: the three-wide maxima T, T2; centre: the two-sided maxima H, H2 — then neighbours = max(T_up, T_down, H_mid)
    // and "survives" = saturating (centre - neighbours) != 0, which also says centre > 0. (~120 instead of ~400 instructions per
    // thread: this pass had become bound by VALU issue once its latency was out of the way.)
    fs_u16x2 T[SEL_ROWS + 2], T2[SEL_ROWS + 2], Hm[SEL_ROWS + 2], Hm2[SEL_ROWS + 2], C1[SEL_ROWS + 2], C2[SEL_ROWS + 2];
#pragma unroll
    for (int k = 0; k < SEL_ROWS + 2; ++k) {
        const uint32_t lo = (uint32_t)rw[k], hi = (uint32_t)(rw[k] >> 32);
        const fs_u16x2 p01 = __builtin_bit_cast(fs_u16x2, __builtin_amdgcn_perm(hi, lo, 0x0c010c00u));
        const fs_u16x2 q12 = __builtin_bit_cast(fs_u16x2, __builtin_amdgcn_perm(hi, lo, 0x0c020c01u));
        const fs_u16x2 p23 = __builtin_bit_cast(fs_u16x2, __builtin_amdgcn_perm(hi, lo, 0x0c030c02u));
        const fs_u16x2 q34 = __builtin_bit_cast(fs_u16x2, __builtin_amdgcn_perm(hi, lo, 0x0c040c03u));
        const fs_u16x2 p45 = __builtin_bit_cast(fs_u16x2, __builtin_amdgcn_perm(hi, lo, 0x0c050c04u));
        T[k] = fs_max(fs_max(p01, q12), p23); T2[k] = fs_max(fs_max(p23, q34), p45);
        Hm[k] = fs_max(p01, p23); Hm2[k] = fs_max(p23, p45);
        C1[k] = q12; C2[k] = q34;
    }
    unsigned colmask = 0;                                                 // valid columns: 3 <= x < w - 3
#pragma unroll
    for (int i = 0; i < 4; ++i) if (4 * q + i >= 3 && 4 * q + i < lg.w - 3) colmask |= 1u << i;
    unsigned cand = 0;                                                    // bit 4 j + i: pixel (4 q + i, y0 + j) survives the 3x3 test
#pragma unroll
    for (int j = 0; j < SEL_ROWS; ++j) {
        const int y = y0 + j;
        const fs_u16x2 nA = fs_max(fs_max(T[j], T[j + 2]), Hm[j + 1]), nB = fs_max(fs_max(T2[j], T2[j + 2]), Hm2[j + 1]);
        const uint32_t dA = __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(C1[j + 1], nA));
        const uint32_t dB = __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(C2[j + 1], nB));
        unsigned bits = ((dA & 0xffffu) ? 1u : 0u) | ((dA >> 16) ? 2u : 0u) | ((dB & 0xffffu) ? 4u : 0u) | ((dB >> 16) ? 8u : 0u);
        bits &= (y >= 3 && y < lg.h - 3) ? colmask : 0u;
        cand |= bits << (4 * j);
    }
    if (a.keep) {                                                         // diagnostic map of the survivors (dsdtm_debug_fast10)
        for (unsigned m = cand; m; m &= m - 1) {
            const int b = __ffs((int)m) - 1;
            a.keep[lg.off + (size_t)(y0 + (b >> 2)) * lg.stride + 4 * q + (b & 3)] = 1;
        }
    }
    // the survivors of this wave's pixels, four per round: DPP row g of the wave scores the g-th pending lane's lowest one
    const int row = lane >> 4, l = lane & 15;
    const int br = l >> 1, bc = (l & 1) * 4;                              // this lane's four pixels of an 8x8 box: row br, columns bc .. bc + 3
    for (;;) {
        unsigned long long pend = __ballot(cand != 0u);
        if (!pend) break;
        int src[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            src[g] = pend ? __ffsll((long long)pend) - 1 : -1;
            pend &= pend - 1;                                             // (0 & anything stays 0)
        }
        const int my = row == 0 ? src[0] : row == 1 ? src[1] : row == 2 ? src[2] : src[3];
        const int b = cand ? __ffs((int)cand) - 1 : 0;
        const int ox = 4 * q + (b & 3), oy = y0 + (b >> 2);               // this lane's own lowest survivor
        const int cx = __builtin_amdgcn_ds_bpermute((my < 0 ? lane : my) << 2, ox);
        const int cy = __builtin_amdgcn_ds_bpermute((my < 0 ? lane : my) << 2, oy);
        if (lane == src[0] || lane == src[1] || lane == src[2] || lane == src[3]) cand &= cand - 1u;
        if (my < 0) continue;
        const int scale = 1 << level;
        const int k = ((cy * scale) / a.cell_size) * a.grid_cols + (cx * scale) / a.cell_size;       // :97-98
        if (k < 0 || k >= cells || (a.occupied && a.occupied[(size_t)frame * cells + k])) continue;  // :100 (row-uniform)
        // Feature_detector::shiTomasiScore (:157-198) by the 16 lanes of the row
        float scf = 0.0f;
        {
#pragma clang fp contract(off)
            const int x_min = cx - 4, x_max = cx + 4, y_min = cy - 4, y_max = cy + 4;
            if (!(x_min < 1 || x_max >= lg.w - 1 || y_min < 1 || y_max >= lg.h - 1)) {                  // :173
                const uint8_t* __restrict__ p = img + (size_t)lg.stride * (y_min + br) + x_min + bc;
                const unsigned long long m = sel_load64(p - 1);           // columns bc - 1 .. bc + 4 (+ 2 spare bytes, inside the level)
                const uint32_t u4 = *(const SelU32*)(p - lg.stride), d4 = *(const SelU32*)(p + lg.stride);
                int sxx = 0, syy = 0, sxy = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int dx = (int)((m >> (8 * (i + 2))) & 0xff) - (int)((m >> (8 * i)) & 0xff);
                    const int dy = (int)((d4 >> (8 * i)) & 0xff) - (int)((u4 >> (8 * i)) & 0xff);
                    sxx += dx * dx; syy += dy * dy; sxy += dx * dy;
                }
                sxx = row_sum_i32(sxx); syy = row_sum_i32(syy); sxy = row_sum_i32(sxy);
                const float dXX = (float)sxx / 128.0f, dYY = (float)syy / 128.0f, dXY = (float)sxy / 128.0f;   // / (2.0 * box_area), exact
                const float tr = dXX + dYY;
                const float disc = tr * tr - 4 * (dXX * dYY - dXY * dXY);
                scf = 0.5f * (tr - sqrtf(disc));
            }
        }
        if (l == 0 && scf > a.detection_threshold) {                                                       // :104 vs the initial score (:74)
            // max score wins; among equal scores the first in (level, row, column) order, as the sequential loop's strict '>' leaves it
            const unsigned order = ((unsigned)level << 28) | ((unsigned)cy << 14) | (unsigned)cx;
            const unsigned long long key = ((unsigned long long)__float_as_uint(scf) << 32) | (unsigned long long)(0xffffffffu - order);
            atomicMax(a.cell_key + (size_t)frame * cells + k, key);
        }
    }
}

// keys -> the four per-cell arrays the reference's `corners` vector holds (batch entry; the single-frame entries decode on the host)
__global__ __launch_bounds__(256) void detect_decode_kernel(const DetectArgs a) {
    const size_t n = (size_t)a.n_frames * a.grid_cols * a.grid_rows;
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const unsigned long long key = a.cell_key[k];
    if (!key) { a.cell_score[k] = a.detection_threshold; a.cell_x[k] = 0; a.cell_y[k] = 0; a.cell_level[k] = 0; return; }   // :74
    const uint32_t order = 0xffffffffu - (uint32_t)key;
    const int L = (int)(order >> 28), y = (int)((order >> 14) & 0x3fffu), x = (int)(order & 0x3fffu);
    a.cell_score[k] = __uint_as_float((uint32_t)(key >> 32));
    a.cell_x[k] = x << L; a.cell_y[k] = y << L; a.cell_level[k] = L;                                  // :106
}

hipError_t detect_launch(const DetectArgs& a, hipStream_t stream) {
    if (a.n_frames <= 0) return hipSuccess;
    // levels whose rows are whole dwords take the strip kernel, the others (odd widths) one thread per pixel
    unsigned strip_mask = 0, pixel_mask = 0;
    int strip_tasks = 0;
    for (int l = 0; l < a.levels; ++l) {
        const bool strip = ((a.lv[l].w | a.lv[l].stride | (int)a.lv[l].off) & 3) == 0 && a.lv[l].w >= 16 && (a.pyr_pitch & 3) == 0;
        if (strip) { strip_mask |= 1u << l; const int t = (a.lv[l].w >> 2) * ((a.lv[l].h + FS_ROWS - 1) / FS_ROWS); strip_tasks = t > strip_tasks ? t : strip_tasks; }
        else pixel_mask |= 1u << l;
    }
    if (strip_mask)
        hipLaunchKernelGGL(fast_score_strip_kernel, dim3((unsigned)((strip_tasks + 255) / 256), (unsigned)a.levels, (unsigned)a.n_frames),
                           dim3(256), 0, stream, a, strip_mask);
    if (pixel_mask)
        hipLaunchKernelGGL(fast_score_kernel, dim3((unsigned)((a.lv[0].w + 255) / 256), (unsigned)a.lv[0].h, (unsigned)(a.levels * a.n_frames)),
                           dim3(256), 0, stream, a, pixel_mask);
    const int px = a.n_frames >= 8 ? 4 : 1;
    if (px == 4 && pixel_mask == 0) {
        // batches whose levels are all whole dwords wide: 4 x SEL_ROWS pixels per thread, four survivors scored per round
        int sel_tasks = 0;
        for (int l = 0; l < a.levels; ++l) { const int t = (a.lv[l].w >> 2) * ((a.lv[l].h + SEL_ROWS - 1) / SEL_ROWS); sel_tasks = t > sel_tasks ? t : sel_tasks; }
        hipLaunchKernelGGL(fast_select_rows_kernel, dim3((unsigned)((sel_tasks + 255) / 256), (unsigned)a.levels, (unsigned)a.n_frames), dim3(256), 0, stream, a);
    } else {
        int sel_tasks = 0;
        for (int l = 0; l < a.levels; ++l) { const int t = ((a.lv[l].w + px - 1) / px) * a.lv[l].h; sel_tasks = t > sel_tasks ? t : sel_tasks; }
        const dim3 sel_grid((unsigned)((sel_tasks + 255) / 256), (unsigned)a.levels, (unsigned)a.n_frames);
        if (px == 4) hipLaunchKernelGGL(fast_select_kernel<4>, sel_grid, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL(fast_select_kernel<1>, sel_grid, dim3(256), 0, stream, a);
    }
    if (a.cell_score) {
        const size_t n = (size_t)a.n_frames * a.grid_cols * a.grid_rows;
        hipLaunchKernelGGL(detect_decode_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a);
    }
    return hipGetLastError();
}

}  // namespace dsdtm
