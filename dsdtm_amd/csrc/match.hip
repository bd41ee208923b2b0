// match.hip — Feature_Alignment::FindMatchDirect for M candidates in ONE launch: the warp prelude (SolveAffineMatrix,
// GetBestSearchLevel, WarpAffine, GetPatchNoBoarder; reference src/Feature_alignment.cpp:160-275) and Align2DGaussNewton
// (:318-417) on the warped patches, which never leave LDS.
//
// Rounds 1-4 ran two kernels — warp_kernel (warp.hip) wrote every candidate's 10x10 and 8x8 patches to HBM, align2d_rows_kernel
// (align2d.hip) read them back: 2 x 164 B of the 433 B a candidate moves, 2.3x the algorithmic bytes of the call, and a kernel
// boundary in the middle of a 20-us call. Here a 256-thread group carries its candidates through three phases:
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

#include "match_body.h"

namespace dsdtm {

// CH = candidates per workgroup (a multiple of MATCH_G, at most 64): phase 1 runs once for all of them — CH lanes of the first
// wavefront, one FP64 chain each — then phases 2 and 3 go through them MATCH_G at a time (match_body.h). Measured
// (tools/fmd_bench.py, 51 200 candidates / one frame's 816): CH = 16: 62.4 / 17.2 us; 32: 65.3 / 26.2 us; 64: 70.7 / 43.4 us (the
// two-kernel path: 64.8 / 18.2 us) — the rounds, not the chain, are what a workgroup spends its time on, and amortising the chain
// only serialises them. CH = 16 is the product shape; the others stay behind DSDTM_MATCH_GROUP for the record.
template <int CH>
__global__ __launch_bounds__(256) void match_kernel(const WarpKernelArgs a, const A2DKernelArgs b) {
    static_assert(CH % MATCH_G == 0 && CH <= 64, "candidates per workgroup");
    __shared__ MatchShared<CH> sh;
    // XCD-aware block numbering (workgroups go to the 8 XCDs round-robin, each XCD has its own L2): every XCD takes a contiguous
    // range of candidate groups, i.e. of current frames / keyframes, instead of every eighth group of every frame — each
    // frame's images are then fetched by one L2 instead of by all eight. The work per group is uniform, so the ranges balance.
    unsigned lb = blockIdx.x;
    {
        const unsigned q = gridDim.x / 8u;
        if (!a.no_xcd && lb < q * 8u) lb = (lb % 8u) * q + lb / 8u;
    }
    const int cb = (int)lb * CH;
    const int tid = threadIdx.x;
    const int nb = a.m - cb < CH ? a.m - cb : CH;
    if (tid < nb) {
        sh.c[tid] = warp_candidate(a, cb + tid);
        sh.sl[tid] = a.search_level[cb + tid];                          // written by warp_candidate (this thread)
    }
    __syncthreads();
    match_rounds<CH>(a, b, sh, cb, nb, tid);
}

hipError_t match_launch(const WarpKernelArgs& wa, const A2DKernelArgs& aa, hipStream_t stream) {
    if (wa.m <= 0) return hipSuccess;
    const int ch = options().match_group ? options().match_group : 16;
    const dim3 grid((unsigned)((wa.m + ch - 1) / ch));
    switch (ch) {
        case 16: hipLaunchKernelGGL(match_kernel<16>, grid, dim3(256), 0, stream, wa, aa); break;
#ifdef DSDTM_DIAG                                     // 32 and 64 candidates per group: measured slower, kept for A/B only
        case 32: hipLaunchKernelGGL(match_kernel<32>, grid, dim3(256), 0, stream, wa, aa); break;
        case 64: hipLaunchKernelGGL(match_kernel<64>, grid, dim3(256), 0, stream, wa, aa); break;
#endif
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace dsdtm
