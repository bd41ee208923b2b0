// match.hip — Feature_Alignment::FindMatchDirect for M candidates in ONE launch: the warp prelude (SolveAffineMatrix,
// GetBestSearchLevel, WarpAffine, GetPatchNoBoarder; reference src/Feature_alignment.cpp:160-275) and Align2DGaussNewton
// (:318-417) on the warped patches, which never leave LDS.
//
// Rounds 1-4 ran two kernels — warp_kernel (warp.hip) wrote every candidate's 10x10 and 8x8 patches to HBM, align2d_rows_kernel
// (align2d.hip) read them back: 2 x 164 B of the 433 B a candidate moves, 2.3x the algorithmic bytes of the call, and a kernel
// boundary in the middle of a 20-us call. Here a 256-thread group carries 16 candidates through three phases:
//   1. lane = candidate (16 lanes): the FP64 chain in the reference's operation order (warp_body.h: warp_candidate);
//   2. thread = sample: the group's 1600 samples of the bordered patches into LDS (warp_body.h: warp_samples);
//   3. four candidates per wavefront, one per 16-lane DPP row: Align2D on the LDS patches (align2d_body.h), the level of the
//      current frame chosen by the search level phase 1 found.
// Same device functions, same arithmetic, same bits as the two-kernel path (which stays: dsdtm_warp_patches and
// dsdtm_align2d_batch are entry points of their own, and DSDTM_FMD_SPLIT=1 runs FindMatchDirect through them for A/B).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

#pragma clang fp contract(off)      // file scope: the FP64 chain of warp_body.h and the float sums of align2d_body.h

#include "warp_body.h"
#include "align2d_body.h"

namespace dsdtm {

constexpr int MATCH_G = 16;         // candidates per 256-thread group (= features per group of align2d_rows_kernel<4>)

__global__ __launch_bounds__(256) void match_kernel(const WarpKernelArgs a, const A2DKernelArgs b) {
    __shared__ WarpCand s_c[MATCH_G];
    __shared__ int s_sl[MATCH_G];                                       // search level of the group's candidates (-1: rejected)
    __shared__ __attribute__((aligned(16))) uint8_t s_pb[MATCH_G * 100];
    __shared__ __attribute__((aligned(16))) uint8_t s_pp[MATCH_G * 64];
    __shared__ __attribute__((aligned(16))) float s_prod[MATCH_G][192];
    const int c0 = (int)blockIdx.x * MATCH_G;
    const int tid = threadIdx.x;
    const int ng = a.m - c0 < MATCH_G ? a.m - c0 : MATCH_G;
    if (tid < ng) {
        s_c[tid] = warp_candidate(a, c0 + tid);
        s_sl[tid] = a.search_level[c0 + tid];                           // written by warp_candidate (this thread)
    }
    __syncthreads();
    warp_samples<256>(s_c, ng, tid, s_pb, s_pp);
    __syncthreads();
    // ---- Align2DGaussNewton (:318-417) on the patches in LDS; candidate = slot of the group ----
    constexpr int PPL = 4, LPF = 64 / PPL, FPW = 64 / LPF;
    const int lane = tid & 63;
    const int slot = (tid >> 6) * FPW + lane / LPF, l = lane % LPF;
    const int f = c0 + slot;
    const bool exists = slot < ng;
    const int lvl = exists ? s_sl[slot] : -1;
    const int fr = (exists && b.frame) ? b.frame[f] : 0;
    const bool valid = exists && !(lvl < 0 || lvl >= b.levels || fr < 0 || (b.frame && fr >= b.n_frames));
    if (exists && !valid && l == 0) b.converged[f] = 0;                 // rejected candidate: "not converged", pixel untouched
    const LevelGeom lg = b.lv[valid ? lvl : 0];
    const uint8_t* __restrict__ img = b.cur_pyr + (size_t)fr * b.pyr_pitch + lg.off;
    const double lscale = (b.px_level0 && valid) ? (double)(1 << lvl) : 1.0;
    float u, v;
    bool converged;
    align2d_rows_feature<PPL>(valid, img, lg, lg.stride * lg.h, (const uint8_t*)(s_pb + (exists ? slot : 0) * 100),
                              (const uint8_t*)(s_pp + (exists ? slot : 0) * 64), s_prod[slot],
                              valid ? b.px_xy[2 * (size_t)f] : 0.0, valid ? b.px_xy[2 * (size_t)f + 1] : 0.0, lscale, b.max_iters, lane, u, v, converged);
    if (valid && l == 0) {
        b.px_xy[2 * (size_t)f] = (double)u * lscale;                    // :414 always written back (:154-156 back to level 0)
        b.px_xy[2 * (size_t)f + 1] = (double)v * lscale;
        b.converged[f] = converged ? 1 : 0;
    }
}

hipError_t match_launch(const WarpKernelArgs& wa, const A2DKernelArgs& aa, hipStream_t stream) {
    if (wa.m <= 0) return hipSuccess;
    hipLaunchKernelGGL(match_kernel, dim3((unsigned)((wa.m + MATCH_G - 1) / MATCH_G)), dim3(256), 0, stream, wa, aa);
    return hipGetLastError();
}

}  // namespace dsdtm
