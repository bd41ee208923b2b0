// track.hip — the two kernels dsdtm_track_frame adds between Run, FindMatchDirect and the pose refinement so that one
// tracked frame (reference src/Tracking.cpp:199-256) is ONE submission:
//
//   track_match_kernel      UpdateLocalMap's ReprojectPoint for every local map point with the pose Run just produced
//                           (src/Feature_alignment.cpp:54-69, Frame::World2Pixel src/Frame.cpp:318-323, Camera::IsInImage
//                           src/Camera.cpp:187-193), MapPoint::Get_ClosetObs (src/MapPoint.cpp:133-174) and the reference-
//                           pixel test of FindMatchDirect (:135-140) — reproject_point(), lane = map point — in front of
//                           phase 1 of the fused FindMatchDirect kernel (match.hip / match_body.h), which then runs for that
//                           point; block 0 also forwards Run's results to the caller's pinned block.
//   track_replay_kernel     the order-dependent part of SearchLocalPoints / ReprojectCell (:71-121) over the results of
//                           that kernel, ONE workgroup: cells in index order, candidates by found count (stable), bad /
//                           masked candidates skipped, first success per cell, a disc of radius cell_size around every
//                           success that suppresses later candidates, stop after max_matches cells. Then the features the
//                           matches become (px as cv::Point2f, level, bearing of Frame::Add_Feature src/Frame.cpp:83-92) as
//                           the columns the pose-refinement kernel reads.
//
// The walk is sequential in the reference, but a candidate's fate depends only on EARLIER candidates that are close to it:
// the earlier candidates of its own cell, and the earlier candidates whose disc (centre = their refined pixel, known for
// every converged candidate before any decision is taken) covers its reprojected pixel. So the replay is a dataflow
// evaluation of the same recurrence: every live candidate lists its possible blockers once, then decides as soon as they
// have decided — rounds = the depth of the dependency chains (tens), not the number of candidates (hundreds). A candidate
// with more possible blockers than a thread keeps (dense worlds) switches the workgroup to scanning all earlier candidates
// each round: slower, the same decisions.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

#pragma clang fp contract(off)      // the reprojection follows the host layer's operation order (dsdtm_host.hpp), no FMA;
                                    // file scope: also the FP64 chain of warp_body.h and the float sums of align2d_body.h
#include "match_body.h"

namespace dsdtm {

namespace {

// cvRound (OpenCV 2.4, SSE2 cvtsd2si): round half to even; callers have checked that v is finite and small
__device__ __forceinline__ int cv_round(double v) { return (int)rint(v); }

__device__ __forceinline__ bool in_image(int width, int height, double x, double y, int boundary, int level) {   // src/Camera.cpp:187-193
    if (!(fabs(x) < 1e9) || !(fabs(y) < 1e9)) return false;       // (NaN / infinity / beyond any image: cvRound is undefined there)
    const int xr = cv_round(x), yr = cv_round(y);
    return xr >= boundary && xr < width / (1 << level) - boundary && yr >= boundary && yr < height / (1 << level) - boundary;
}

// mOw = -R^T t in the host layer's order (dsdtm_host.hpp Frame::Get_CameraCnt)
__device__ __forceinline__ void camera_centre(const double* __restrict__ m, double& c0, double& c1, double& c2) {
    c0 = -(m[0] * m[3] + m[4] * m[7] + m[8] * m[11]);
    c1 = -(m[1] * m[3] + m[5] * m[7] + m[9] * m[11]);
    c2 = -(m[2] * m[3] + m[6] * m[7] + m[10] * m[11]);
}

}  // namespace

// ReprojectPoint + Get_ClosetObs + the reference-pixel test for map point i with the pose m (3x4, LDS): writes the
// candidate columns of point i (thread = map point)
__device__ __forceinline__ void reproject_point(const TrackArgs& a, int i, const double* m, bool lost) {
    const double P0 = a.mp_world[3 * (size_t)i], P1 = a.mp_world[3 * (size_t)i + 1], P2 = a.mp_world[3 * (size_t)i + 2];
    const bool bad = a.mp_bad[i] != 0;                             // UpdateLocalMap skips bad points (src/Tracking.cpp:288)
    const int o_lo = a.obs_offset[i], o_hi = a.obs_offset[i + 1];
    a.pw[3 * (size_t)i] = P0; a.pw[3 * (size_t)i + 1] = P1; a.pw[3 * (size_t)i + 2] = P2;
    // Frame::World2Pixel (src/Frame.cpp:318-323) -> Camera::Camera2Pixel (src/Camera.cpp:167-171)
    const double x = m[0] * P0 + m[1] * P1 + m[2] * P2 + m[3], y = m[4] * P0 + m[5] * P1 + m[6] * P2 + m[7],
                 z = m[8] * P0 + m[9] * P1 + m[10] * P2 + m[11];
    const double u = (double)a.fx * x / z + (double)a.cx, v = (double)a.fy * y / z + (double)a.cy;
    const bool in_grid = !lost && !bad && in_image(a.width, a.height, u, v, 8, 0);           // ReprojectPoint (:54-69)
    int cell = -1, best = -1, jbest = -1;
    if (in_grid) {
        cell = (int)(v / a.cell_size) * a.grid_cols + (int)(u / a.cell_size);             // :65
        // MapPoint::Get_ClosetObs (src/MapPoint.cpp:133-174)
        double c0, c1, c2;
        camera_centre(m, c0, c1, c2);
        double v0 = c0 - P0, v1 = c1 - P1, v2 = c2 - P2;
        const double nv = sqrt(v0 * v0 + v1 * v1 + v2 * v2);
        v0 /= nv; v1 /= nv; v2 /= nv;
        double best_cos = 0.0;
        int first = -1, jfirst = -1;
        for (int j = o_lo; j < o_hi; ++j) {
            const int k = a.obs_kf[j];
            if (first < 0) { first = k; jfirst = j; }
            double k0, k1, k2;
            camera_centre(a.T_kf_w + 12 * (size_t)k, k0, k1, k2);
            double r0 = k0 - P0, r1 = k1 - P1, r2 = k2 - P2;
            const double nr = sqrt(r0 * r0 + r1 * r1 + r2 * r2);
            r0 /= nr; r1 /= nr; r2 /= nr;
            const double cs = r0 * v0 + r1 * v1 + r2 * v2;
            if (cs > best_cos) { best_cos = cs; best = k; jbest = j; }
        }
        if (best < 0) { best = first; jbest = jfirst; }
        if (best_cos < 0.5) { best = -1; jbest = -1; }
        if (jbest >= 0) {                                          // FindMatchDirect :138-140: the reference pixel 5 px inside its level
            const int lv = a.obs_level[jbest];
            const float rx = a.obs_px[2 * (size_t)jbest], ry = a.obs_px[2 * (size_t)jbest + 1];
            const float s = (float)(1 << (lv < 0 ? 0 : (lv > 30 ? 30 : lv)));
            if (lv < 0 || lv >= a.levels || !in_image(a.width, a.height, (double)(rx / s), (double)(ry / s), 5, lv)) { best = -1; jbest = -1; }
        }
    }
    a.cell[i] = cell;
    a.px0[2 * (size_t)i] = u; a.px0[2 * (size_t)i + 1] = v;
    a.px[2 * (size_t)i] = u; a.px[2 * (size_t)i + 1] = v;          // in/out of FindMatchDirect
    a.cand_kf[i] = best;                                           // -1: rejected by FindMatchDirect (search level -1, not converged)
    a.cand_frame[i] = 0;
    if (jbest >= 0) {
        a.ref_px[2 * (size_t)i] = a.obs_px[2 * (size_t)jbest]; a.ref_px[2 * (size_t)i + 1] = a.obs_px[2 * (size_t)jbest + 1];
        a.ref_level[i] = a.obs_level[jbest];
        a.ref_bearing[3 * (size_t)i] = a.obs_bearing[3 * (size_t)jbest];
        a.ref_bearing[3 * (size_t)i + 1] = a.obs_bearing[3 * (size_t)jbest + 1];
        a.ref_bearing[3 * (size_t)i + 2] = a.obs_bearing[3 * (size_t)jbest + 2];
    } else {
        a.ref_px[2 * (size_t)i] = 0.0f; a.ref_px[2 * (size_t)i + 1] = 0.0f; a.ref_level[i] = 0;
        a.ref_bearing[3 * (size_t)i] = 0.0; a.ref_bearing[3 * (size_t)i + 1] = 0.0; a.ref_bearing[3 * (size_t)i + 2] = 1.0;
    }
    // the mask test of ReprojectCell (:96) reads the ROUNDED reprojected pixel; the caller's mask at the start of the search
    uint8_t blocked = 0;
    if (in_grid && a.mask) blocked = a.mask[(size_t)cv_round(v) * a.mask_stride + cv_round(u)] != 255 ? 1 : 0;
    a.init_blocked[i] = blocked;
}

// One kernel from Run's pose to the FindMatchDirect results of every local map point: the fused FindMatchDirect kernel
// (match.hip: 16 candidates per 256-thread group) whose phase-1 lanes first reproject their map point — the candidate columns
// a lane writes are the ones it reads back in the next statement. (Round 6 first ran the reprojection as a kernel of its own,
// thread = point: 7.4 us in front of a 13.7-us kernel, most of it the fixed cost of one more dependent launch.)
__global__ __launch_bounds__(256) void track_match_kernel(const TrackArgs t, const WarpKernelArgs a, const A2DKernelArgs b) {
    __shared__ MatchShared<MATCH_G> sh;
    __shared__ double s_T[12];
    __shared__ int s_lost;
    unsigned lb = blockIdx.x;
    {
        const unsigned q = gridDim.x / 8u;                         // XCD-aware block numbering, as match_kernel
        if (!a.no_xcd && lb < q * 8u) lb = (lb % 8u) * q + lb / 8u;
    }
    const int cb = (int)lb * MATCH_G;
    const int tid = threadIdx.x;
    const int nb = t.n_points - cb < MATCH_G ? t.n_points - cb : MATCH_G;
    if (tid < 12) s_T[tid] = t.T_run[tid];
    if (tid == 12) s_lost = (t.n_tracked[0] < t.min_tracked) ? 1 : 0;
    if (lb == 0 && tid >= 64 && tid - 64 < t.run_out_n16)          // Run's pose, count, statistics -> the caller's pinned block
        ((uint4*)t.run_out_host)[tid - 64] = ((const uint4*)t.run_out_dev)[tid - 64];
    __syncthreads();
    if (lb == 0 && tid < 12) t.T_opt[tid] = s_T[tid];              // the refinement's in/out pose starts from Run's
    if (tid < nb) {
        reproject_point(t, cb + tid, s_T, s_lost != 0);
        sh.c[tid] = warp_candidate(a, cb + tid);
        sh.sl[tid] = a.search_level[cb + tid];                     // written by warp_candidate (this thread)
    }
    __syncthreads();
    match_rounds<MATCH_G>(a, b, sh, cb, nb, tid);
}

// ---------------------------------------------------------------------------------------------------------------------
// the replay
// ---------------------------------------------------------------------------------------------------------------------
namespace {
constexpr int RP_THREADS = 1024;
enum : uint8_t { ST_UNKNOWN = 0, ST_ACCEPTED = 1, ST_REJ_FREE = 2, ST_REJ_TAKEN = 3, ST_DEAD = 4 };

__device__ __forceinline__ uint32_t pack_xy(int x, int y) { return (uint32_t)(uint16_t)(int16_t)x | ((uint32_t)(uint16_t)(int16_t)y << 16); }
__device__ __forceinline__ int unpack_x(uint32_t v) { return (int)(int16_t)(v & 0xffffu); }
__device__ __forceinline__ int unpack_y(uint32_t v) { return (int)(int16_t)(v >> 16); }

// Dynamic LDS of the replay workgroup, carved by ONE function for the host (size) and the device (pointers)
struct ReplayLds {
    unsigned long long* tmp;   // [mpad] keys in bin order
    uint32_t* hist;      // [cells + 1] counts, then bin starts (exclusive scan); [cells] = n_in
    uint32_t* cur;       // [cells] next free place of a bin
    uint32_t* wsum;      // [RP_THREADS / 64]
    uint32_t* P;         // [mpad] by rank: rounded reprojected pixel (x | y << 16)
    uint32_t* Q;         // [mpad] by rank: rounded refined pixel = disc centre (a dead candidate's: nowhere)
    uint16_t* idx;       // [mpad] candidate (map point) of rank r
    uint16_t* rcell;     // [mpad] its grid cell
    int16_t* hw;         // [radius + 1]
    int8_t* sl;          // [mpad] by rank: search level
    uint8_t* state;      // [mpad] by rank
    size_t bytes;
};
__host__ __device__ inline ReplayLds replay_layout(uint8_t* base, int n_points, int cells, int radius) {
    const size_t mpad = ((size_t)n_points + 63) / 64 * 64;
    ReplayLds L;
    size_t o = 0;
    auto take = [&](size_t bytes, size_t align) { o = (o + align - 1) / align * align; const size_t at = o; o += bytes; return base + at; };
    L.tmp = (unsigned long long*)take(mpad * 8, 8);
    L.hist = (uint32_t*)take(((size_t)cells + 1) * 4, 4);
    L.cur = (uint32_t*)take((size_t)cells * 4, 4);
    L.wsum = (uint32_t*)take((RP_THREADS / 64) * 4, 4);
    L.P = (uint32_t*)take(mpad * 4, 4);
    L.Q = (uint32_t*)take(mpad * 4, 4);
    L.idx = (uint16_t*)take(mpad * 2, 2);
    L.rcell = (uint16_t*)take(mpad * 2, 2);
    L.hw = (int16_t*)take(((size_t)radius + 1) * 2, 2);
    L.sl = (int8_t*)take(mpad, 1);
    L.state = take(mpad, 1);
    L.bytes = (o + 15) / 16 * 16;
    return L;
}
}  // namespace

// half-widths of the rows cv::circle(img, c, r, v, -1) paints (OpenCV 2.4 drawing.cpp Circle(): midpoint circle filled by
// horizontal spans; a row can be painted by several spans, the widest counts) — hw[|dy|], dy = -r..r; radius <= 127
void track_disc_half_widths(int radius, int8_t* hw) {
    for (int i = 0; i <= radius; ++i) hw[i] = -1;
    int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
    while (dx >= dy) {
        if (hw[dy] < dx) hw[dy] = (int8_t)dx;
        if (hw[dx] < dy) hw[dx] = (int8_t)dy;
        dy += 1;
        err += plus;
        plus += 2;
        const int mask = (err <= 0) - 1;
        err -= minus & mask;
        dx += mask;
        minus -= mask & 2;
    }
}

size_t track_replay_lds_bytes(int n_points, int n_cells, int radius) { return replay_layout(nullptr, n_points, n_cells, radius).bytes; }

template <int EPT, int NT = RP_THREADS>
__global__ __launch_bounds__(NT) void track_replay_kernel(const TrackArgs a) {
    static_assert(NT % 64 == 0 && NT <= RP_THREADS, "threads of the replay workgroup");
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    __shared__ int s_overflow, s_n_in;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int M = a.n_points, cells = a.grid_cols * a.grid_rows, R = a.cell_size, cols = a.grid_cols;
    const ReplayLds L = replay_layout(lds_raw, M, cells, R);
    for (int c = tid; c <= cells; c += NT) L.hist[c] = 0u;
    if (tid == 0) s_overflow = 0;
    if (tid <= R) L.hw[tid] = (int16_t)a.disc_hw[tid];          // cv::circle's row half-widths, tabulated by the host (track_disc_half_widths)
    __syncthreads();

    // ---- 1. everything a candidate brings, read ONCE and coalesced (thread = candidate); the grid's cell lists in walk order:
    //         counting sort by cell, inside a cell by (found descending, list index) ----
    unsigned long long key[EPT];
    int kcell[EPT];
    uint32_t cp[EPT], cq[EPT];          // rounded reprojected / refined pixel of the thread's candidates
    int csl[EPT];                       // search level; bit 8: live
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int i = tid + e * NT;
        kcell[e] = -1;
        key[e] = ~0ull;
        cp[e] = cq[e] = 0u; csl[e] = 0;
        if (i < M) {
            const int c = a.cell[i];
            const int f = a.mp_found[i];
            // ReprojectCell: IsBad (:93), mask (:96), FindMatchDirect false (:101-104) — none of them has a side effect, so a
            // candidate that fails any of them is simply not there
            const bool ok = a.converged[i] != 0 && a.mp_bad[i] == 0 && a.init_blocked[i] == 0;
            const double u0 = a.px0[2 * (size_t)i], v0 = a.px0[2 * (size_t)i + 1];
            const double u1 = a.px[2 * (size_t)i], v1 = a.px[2 * (size_t)i + 1];
            const int sl = a.search_level[i];
            if (c >= 0 && c < cells) {
                kcell[e] = c;
                key[e] = ((unsigned long long)(uint32_t)(0x7fffffffll - (long long)f) << 16) | (unsigned long long)i;   // :88,:123-126 (stable: list order)
                atomicAdd(&L.hist[c], 1u);
                const bool live = ok && fabs(u1) < 30000.0 && fabs(v1) < 30000.0;      // (a converged pixel sits in or near the image)
                const int px_ = cv_round(u0), py_ = cv_round(v0);
                const int qx_ = live ? cv_round(u1) : px_, qy_ = live ? cv_round(v1) : py_;
                cp[e] = pack_xy(px_, py_);
                cq[e] = live ? pack_xy(qx_, qy_) : 0x7fff7fffu;          // (a dead candidate's disc is nowhere: no state test in the scans)
                csl[e] = (sl & 0xff) | (live ? 0x100 : 0);
                // the neighbourhood scan below looks two cells around a candidate: valid while a disc centre (the refined pixel)
                // stays within one cell size of its candidate's reprojected pixel — otherwise every earlier candidate is scanned
                const int ddx = qx_ - px_, ddy = qy_ - py_;
                if ((ddx < 0 ? -ddx : ddx) > R || (ddy < 0 ? -ddy : ddy) > R) s_overflow = 1;
            }
        }
    }
    __syncthreads();
    {   // exclusive scan of the cell counts: consecutive cells per thread, wave scan, wave totals
        const int cpt = (cells + NT - 1) / NT;     // <= 16 (cells <= 4096, NT >= 256)
        uint32_t c[16], sum = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) c[q] = 0u;
        for (int q = 0; q < cpt; ++q) { const int cc = tid * cpt + q; c[q] = cc < cells ? L.hist[cc] : 0u; sum += c[q]; }
        uint32_t inc = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t up = (uint32_t)__shfl_up((int)inc, d); if (lane >= d) inc += up; }
        if (lane == 63) L.wsum[wave] = inc;
        __syncthreads();
        uint32_t base = inc - sum;
        for (int w = 0; w < wave; ++w) base += L.wsum[w];
        for (int q = 0; q < cpt; ++q) { const int cc = tid * cpt + q; if (cc < cells) { L.hist[cc] = base; L.cur[cc] = base; } base += c[q]; }
        if (tid == NT - 1) { L.hist[cells] = base; s_n_in = (int)base; }
    }
    __syncthreads();
    const int n_in = s_n_in;
#pragma unroll
    for (int e = 0; e < EPT; ++e)
        if (kcell[e] >= 0) L.tmp[atomicAdd(&L.cur[kcell[e]], 1u)] = ((unsigned long long)kcell[e] << 48) | key[e];
    __syncthreads();
    // a key's place inside its cell = the number of smaller keys in the cell's range of `tmp` (cells hold a handful of candidates):
    // every candidate counts for itself and parks what the walk needs of it at its place (rank)
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        if (kcell[e] < 0) continue;
        const int c = kcell[e];
        const uint32_t s0 = L.hist[c], e0 = L.hist[c + 1];
        const unsigned long long mine = ((unsigned long long)c << 48) | key[e];
        uint32_t rank = 0;
        for (uint32_t q = s0; q < e0; ++q) rank += L.tmp[q] < mine ? 1u : 0u;
        const uint32_t r = s0 + rank;
        L.idx[r] = (uint16_t)(tid + e * NT); L.rcell[r] = (uint16_t)c;
        L.P[r] = cp[e]; L.Q[r] = cq[e];
        L.sl[r] = (int8_t)(csl[e] & 0xff);
        L.state[r] = (csl[e] & 0x100) ? ST_UNKNOWN : ST_DEAD;
    }
    __syncthreads();
    // from here on thread t owns places t * EPT .. t * EPT + EPT - 1 (contiguous: the final prefix count is a plain scan)
    int my_idx[EPT];
    bool live[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int r = tid * EPT + e;
        live[e] = false; my_idx[e] = 0;
        if (r < n_in) { my_idx[e] = (int)L.idx[r]; live[e] = L.state[r] == ST_UNKNOWN; }
    }
    const bool full_scan = s_overflow != 0;
    // ---- 2. possible blockers of a live candidate: the previous live candidate of its cell (pred), and the live EARLIER
    //         candidates whose disc covers its reprojected pixel — they sit at most two cells away, i.e. (ranks are in cell
    //         order) in three contiguous rank ranges: cells cx-2..cx+2 of the rows cy-2, cy-1 and cy, cut at the candidate itself.
    //         One 64-bit mask per range; a range of more than 64 ranks switches the workgroup to the full scan ----
    int pred[EPT];
    uint32_t lo[EPT][3];
    unsigned long long bm[EPT][3];
    int ovf = 0;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int r = tid * EPT + e;
        pred[e] = -1;
#pragma unroll
        for (int k = 0; k < 3; ++k) { lo[e][k] = 0; bm[e][k] = 0ull; }
        if (r < n_in && live[e]) {
            const int c = L.rcell[r];
            for (int rr = r - 1; rr >= (int)L.hist[c]; --rr)
                if (L.state[rr] == ST_UNKNOWN) { pred[e] = rr; break; }
            if (!full_scan) {
                const int x = unpack_x(L.P[r]), y = unpack_y(L.P[r]);
                const int cx = c % cols, cy = c / cols;
                const int x0 = max(cx - 2, 0), x1 = min(cx + 2, cols - 1);
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int row = cy - 2 + k;
                    if (row < 0) continue;
                    const uint32_t l0 = L.hist[row * cols + x0];
                    uint32_t h0 = L.hist[row * cols + x1 + 1];
                    if (h0 > (uint32_t)r) h0 = (uint32_t)r;
                    if (h0 <= l0) continue;
                    lo[e][k] = l0;
                    if (h0 - l0 > 64u) { ovf = 1; continue; }
                    unsigned long long m = 0ull;
                    for (uint32_t rr = l0; rr < h0; rr += 4) {            // four at a time: their loads do not wait for each other
                        uint32_t q[4];
                        int ady[4];
                        int16_t hwv[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) q[j] = L.Q[min(rr + (uint32_t)j, h0 - 1u)];
#pragma unroll
                        for (int j = 0; j < 4; ++j) { const int dy = y - unpack_y(q[j]); ady[j] = dy < 0 ? -dy : dy; hwv[j] = L.hw[min(ady[j], R)]; }
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int dx = x - unpack_x(q[j]), adx = dx < 0 ? -dx : dx;
                            if (rr + (uint32_t)j < h0 && ady[j] <= R && adx <= (int)hwv[j]) m |= 1ull << (rr + (uint32_t)j - l0);
                        }
                    }
                    bm[e][k] = m;
                }
            }
        }
    }
    const bool scan_all = __syncthreads_or(ovf) != 0 || full_scan;
    // up to RP_BL blockers as a register list (polled in ONE batch of independent LDS reads per step); a candidate with more keeps its masks
    constexpr int RP_BL = 8;
    uint16_t bl[EPT][RP_BL];
    int nbl[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        nbl[e] = 0;
#pragma unroll
        for (int j = 0; j < RP_BL; ++j) bl[e][j] = 0;
        const int total_bits = __popcll(bm[e][0]) + __popcll(bm[e][1]) + __popcll(bm[e][2]);
        if (total_bits <= RP_BL) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                unsigned long long mm = bm[e][k];
                while (mm) {
                    const int j = __builtin_ctzll(mm);
                    mm &= mm - 1ull;
                    const uint16_t v = (uint16_t)(lo[e][k] + (uint32_t)j);
#pragma unroll
                    for (int q = 0; q < RP_BL; ++q) if (q == nbl[e]) bl[e][q] = v;
                    nbl[e]++;
                }
                bm[e][k] = 0ull;
            }
        } else nbl[e] = -1;                                        // masks
    }
    // ---- 3. the recurrence, evaluated as its inputs become known ----
    // No barrier per round: a decision is one byte in LDS, visible to every wave of the workgroup as soon as it is written, and a
    // candidate only ever waits for candidates of LOWER rank — the lowest undecided one can always decide, so every wave's loop
    // ends (the iteration bound is a safety net that reports instead of hanging). The chains run along and across the cell rows:
    // tens of dependent steps, each one batch of LDS reads. (Measured flat: an s_sleep between polls; no blocker reads while the
    // predecessor is open; a workgroup of 512 / 256 threads is 11 / 35 us slower.)
    int stuck = 0;
    {
        volatile uint8_t* const vs = L.state;
        bool undecided[EPT];
#pragma unroll
        for (int e = 0; e < EPT; ++e) { const int r = tid * EPT + e; undecided[e] = r < n_in && live[e]; }
        unsigned spins = 0;
        for (;;) {
            bool left = false;
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                if (!undecided[e]) continue;
                const int r = tid * EPT + e;
                // one batch: the predecessor's state and the listed blockers' (a dead rank 0.. read for unused slots is harmless)
                const uint8_t sp = pred[e] < 0 ? (uint8_t)ST_REJ_FREE : vs[pred[e]];
                uint8_t sb[RP_BL];
#pragma unroll
                for (int j = 0; j < RP_BL; ++j) sb[j] = vs[bl[e][j]];
                uint8_t ns = ST_UNKNOWN;
                if (sp == ST_ACCEPTED || sp == ST_REJ_TAKEN) ns = ST_REJ_TAKEN;      // :115 the cell already has its match
                else if (sp != ST_UNKNOWN) {
                    bool any_acc = false, any_unk = false;
                    if (scan_all) {
                        const int x = unpack_x(L.P[r]), y = unpack_y(L.P[r]);
                        for (int rr = 0; rr < r; ++rr) {
                            const uint8_t st_ = vs[rr];
                            if (st_ != ST_ACCEPTED && st_ != ST_UNKNOWN) continue;
                            const uint32_t q = L.Q[rr];
                            const int dy = y - unpack_y(q), ady = dy < 0 ? -dy : dy;
                            if (ady > R) continue;
                            const int dx = x - unpack_x(q), adx = dx < 0 ? -dx : dx;
                            if (adx > (int)L.hw[ady]) continue;
                            any_acc |= st_ == ST_ACCEPTED; any_unk |= st_ == ST_UNKNOWN;
                        }
                    } else if (nbl[e] >= 0) {
#pragma unroll
                        for (int j = 0; j < RP_BL; ++j)
                            if (j < nbl[e]) { any_acc |= sb[j] == ST_ACCEPTED; any_unk |= sb[j] == ST_UNKNOWN; }
                    } else {
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            unsigned long long mm = bm[e][k];
                            while (mm) {
                                const int j = __builtin_ctzll(mm);
                                mm &= mm - 1ull;
                                const uint8_t st_ = vs[lo[e][k] + j];
                                any_acc |= st_ == ST_ACCEPTED;
                                if (st_ != ST_UNKNOWN) bm[e][k] &= ~(1ull << j);     // decided: not polled again
                            }
                            any_unk |= bm[e][k] != 0ull;
                        }
                    }
                    if (any_acc) ns = ST_REJ_FREE;                                   // :96 masked by an earlier match's disc (:111)
                    else if (!any_unk) ns = ST_ACCEPTED;
                }
                if (ns != ST_UNKNOWN) { vs[r] = ns; undecided[e] = false; }
                else left = true;
            }
            if (!__ballot(left)) break;
            if (++spins > (1u << 16)) { stuck = 1; break; }                          // (never: see above)
        }
    }
    const bool unsettled = __syncthreads_or(stuck) != 0;
    // ---- 4. the matches in walk order, at most max_matches (:80); the features they become ----
    uint32_t cnt = 0;
#pragma unroll
    for (int e = 0; e < EPT; ++e) { const int r = tid * EPT + e; if (r < n_in && L.state[r] == ST_ACCEPTED) cnt++; }
    uint32_t inc = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t up = (uint32_t)__shfl_up((int)inc, d); if (lane >= d) inc += up; }
    if (lane == 63) L.wsum[wave] = inc;
    __syncthreads();
    uint32_t k = inc - cnt, total = 0;
    for (int w = 0; w < NT / 64; ++w) { if (w < wave) k += L.wsum[w]; total += L.wsum[w]; }
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int r = tid * EPT + e;
        if (!(r < n_in && L.state[r] == ST_ACCEPTED)) continue;
        if (k < (uint32_t)a.max_matches) {
            const int i = my_idx[e];
            const float fx_ = (float)a.px[2 * (size_t)i], fy_ = (float)a.px[2 * (size_t)i + 1];     // Feature(px as cv::Point2f, :108)
            const int lvl = (int)L.sl[r];
            dsdtm_track_match mo;
            mo.cell = (int)L.rcell[r]; mo.point = i; mo.px[0] = fx_; mo.px[1] = fy_; mo.level = lvl;
            a.matches[k] = mo;
            // Frame::Add_Feature (src/Frame.cpp:83-92): mNormal = Pixel2Camera(cv::Point2f, 1.0f) in float, normalised in double
            const float one = 1.0f;
            const double bx = (double)((one * (fx_ - a.cx)) / a.fx), by = (double)((one * (fy_ - a.cy)) / a.fy);
            const double n = sqrt(bx * bx + by * by + 1.0 * 1.0);
            a.po_bearing[3 * (size_t)k] = bx / n; a.po_bearing[3 * (size_t)k + 1] = by / n; a.po_bearing[3 * (size_t)k + 2] = 1.0 / n;
            a.po_world[3 * (size_t)k] = a.pw[3 * (size_t)i]; a.po_world[3 * (size_t)k + 1] = a.pw[3 * (size_t)i + 1];
            a.po_world[3 * (size_t)k + 2] = a.pw[3 * (size_t)i + 2];
            a.po_level[k] = lvl; a.po_use[k] = 1;
        }
        k++;
    }
    if (tid == 0) {
        const int nm = (int)(total < (uint32_t)a.max_matches ? total : (uint32_t)a.max_matches);
        a.po_n[0] = nm;
        a.counts[0] = n_in; a.counts[1] = nm; a.counts[2] = unsettled ? 2 : (scan_all ? 1 : 0);   // 0 masks, 1 full scan, 2 did not settle (a bug)
    }
}

hipError_t track_match_launch(const TrackArgs& t, const WarpKernelArgs& wa, const A2DKernelArgs& aa, hipStream_t stream) {
    if (wa.m != t.n_points || aa.m != t.n_points || t.run_out_n16 < 0 || t.run_out_n16 > 192) return hipErrorInvalidValue;
    // (at least one workgroup: block 0 also forwards Run's results and seeds the refinement's pose)
    const int blocks = t.n_points > 0 ? (t.n_points + MATCH_G - 1) / MATCH_G : 1;
    hipLaunchKernelGGL(track_match_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, t, wa, aa);
    return hipGetLastError();
}

template <auto Kernel>
static hipError_t replay_optin(size_t bytes) {
    // more than 64 KB of dynamic LDS is an opt-in per kernel and device; harmless to repeat
    return hipFuncSetAttribute((const void*)Kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

hipError_t track_replay_launch(const TrackArgs& a, hipStream_t stream) {
    const size_t lds = track_replay_lds_bytes(a.n_points, a.grid_cols * a.grid_rows, a.cell_size);
    if (lds > 160 * 1024 - 256) return hipErrorInvalidValue;
    const int ept = (a.n_points + RP_THREADS - 1) / RP_THREADS;
    hipError_t e = hipSuccess;
    if (ept <= 1) {
        if (lds > 48 * 1024) e = replay_optin<track_replay_kernel<1>>(lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(track_replay_kernel<1>, dim3(1), dim3(RP_THREADS), lds, stream, a);
    } else if (ept <= 2) {
        if (lds > 48 * 1024) e = replay_optin<track_replay_kernel<2>>(lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(track_replay_kernel<2>, dim3(1), dim3(RP_THREADS), lds, stream, a);
    } else if (ept <= 4) {
        if (lds > 48 * 1024) e = replay_optin<track_replay_kernel<4>>(lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(track_replay_kernel<4>, dim3(1), dim3(RP_THREADS), lds, stream, a);
    } else return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace dsdtm
