// match_body.h — phases 2 and 3 of the fused FindMatchDirect kernel (match.hip: match_kernel; track.hip: track_match_kernel, which
// runs the reprojection of dsdtm_track_frame in front of phase 1): a 256-thread group's candidates, whose phase-1 records are in
// LDS, MATCH_G at a time through the sampling of the bordered patches and Align2D on them. Include behind warp_body.h and
// align2d_body.h, in a translation unit with `#pragma clang fp contract(off)` at file scope.
#pragma once
#include "warp_body.h"
#include "align2d_body.h"

namespace dsdtm {

constexpr int MATCH_G = 16;         // candidates per Align2D round of a 256-thread group (= features per group of align2d_rows_kernel<4>)

template <int CH>
struct MatchShared {
    WarpCand c[CH];
    int sl[CH];                                                          // search level of the group's candidates (-1: rejected)
    __attribute__((aligned(16))) uint8_t pb[MATCH_G * 100];
    __attribute__((aligned(16))) uint8_t pp[MATCH_G * 64];
    __attribute__((aligned(16))) float prod[MATCH_G][192];
};

// candidates cb .. cb + nb - 1 (records in sh.c / sh.sl, written before a barrier)
template <int CH>
__device__ __forceinline__ void match_rounds(const WarpKernelArgs& a, const A2DKernelArgs& b, MatchShared<CH>& sh, int cb, int nb, int tid) {
    constexpr int PPL = 4, LPF = 64 / PPL, FPW = 64 / LPF;
    const int lane = tid & 63;
    const int slot = (tid >> 6) * FPW + lane / LPF, l = lane % LPF;
    for (int sub = 0; sub < CH && sub < nb; sub += MATCH_G) {
        const int c0 = cb + sub;
        const int ng = nb - sub < MATCH_G ? nb - sub : MATCH_G;
        warp_samples<256>(sh.c + sub, ng, tid, sh.pb, sh.pp);
        __syncthreads();
        // ---- Align2DGaussNewton (:318-417) on the patches in LDS; candidate = slot of the round ----
        const int f = c0 + slot;
        const bool exists = slot < ng;
        const int lvl = exists ? sh.sl[sub + slot] : -1;
        const int fr = (exists && b.frame) ? b.frame[f] : 0;
        const bool valid = exists && !(lvl < 0 || lvl >= b.levels || fr < 0 || (b.frame && fr >= b.n_frames));
        if (exists && !valid && l == 0) b.converged[f] = 0;             // rejected candidate: "not converged", pixel untouched
        const LevelGeom lg = b.lv[valid ? lvl : 0];
        const uint8_t* __restrict__ img = b.cur_pyr + (size_t)fr * b.pyr_pitch + lg.off;
        const double lscale = (b.px_level0 && valid) ? (double)(1 << lvl) : 1.0;
        float u, v;
        bool converged;
        align2d_rows_feature<PPL>(valid, img, lg, lg.stride * lg.h, (const uint8_t*)(sh.pb + (exists ? slot : 0) * 100),
                                  (const uint8_t*)(sh.pp + (exists ? slot : 0) * 64), sh.prod[slot],
                                  valid ? b.px_xy[2 * (size_t)f] : 0.0, valid ? b.px_xy[2 * (size_t)f + 1] : 0.0, lscale, b.max_iters, lane, u, v, converged);
        if (valid && l == 0) {
            b.px_xy[2 * (size_t)f] = (double)u * lscale;                // :414 always written back (:154-156 back to level 0)
            b.px_xy[2 * (size_t)f + 1] = (double)v * lscale;
            b.converged[f] = converged ? 1 : 0;
        }
        if (CH > MATCH_G) __syncthreads();                              // the next round overwrites the patches
    }
}

}  // namespace dsdtm
