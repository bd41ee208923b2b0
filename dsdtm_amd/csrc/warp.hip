// warp.hip — warp prelude of Feature_Alignment::FindMatchDirect for M candidates at once.
//
// Replaces reference src/Feature_alignment.cpp:160-190 (SolveAffineMatrix), :192-204
// (GetBestSearchLevel), :206-259 (WarpAffine), :261-275 (GetPatchNoBoarder), including the
// integer-division quirk W1 (`1/(1<<lvl)` is 0 for lvl>=1, :231) and the float->u8 truncation.
// Groups of 2 or 64 candidates per 128-thread workgroup: the 2x2 affine is evaluated in FP64 ONCE per candidate
// (lane = candidate), then the group's 10x10 bordered patches are sampled thread = sample (warp_kernel below).
//
// The outputs are BYTES (float sample positions truncated to u8) and a search level decided by a
// threshold on det(A): a last-bit difference in the FP64 pose chain can flip either. So this file
// evaluates the chain in the reference's own operation order — Sophus' SE3 from a rotation matrix,
// inverse, product and action on a point (Eigen quaternion normalisation as coeffs / norm), no FMA
// contraction, no reciprocals, IEEE division and sqrt — instead of the solver-oriented shortcuts of
// device_math.h (series normalisation, contracted products), and its results are bit-identical to
// the line-by-line CPU restatement (tests/test_align2d_gpu.py::test_warp_patches_match_oracle).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

#pragma clang fp contract(off)      // file scope: every function below, helpers included (warp_body.h too)

#include "warp_body.h"

namespace dsdtm {

// G candidates per 128-thread group (2 for one frame's search, 64 for batches), two phases:
//  1. lane = candidate (G of the 128 lanes): the FP64 chain — SolveAffineMatrix, GetBestSearchLevel, the inverse affine —
//     ONCE per candidate. (Round 1-3: one group per candidate with all 128 lanes evaluating the same chain: 2 waves x ~2 k
//     dependent FP64 instructions per candidate, 96 % of the kernel; 51 200 candidates took 213 us.)
//  2. thread = sample: the group's G x 100 samples of the 10x10 bordered patches in one strided loop (consecutive
//     threads write consecutive bytes of patch_border), four byte gathers each.
// Same arithmetic, same operation order, same bytes as before (tests/test_align2d_gpu.py::test_warp_patches_match_oracle).
template <int G>
__global__ __launch_bounds__(128) void warp_kernel(const WarpKernelArgs a) {
    __shared__ WarpCand s_c[G];
    // the group's output bytes, staged: patch_border (G x 100) and patch (G x 64) are contiguous in memory across the
    // candidates of a group, so they leave as whole dwords instead of one byte store per sample (+ 0.64 per sample)
    __shared__ __attribute__((aligned(16))) uint8_t s_pb[G * 100];
    __shared__ __attribute__((aligned(16))) uint8_t s_pp[G * 64];
    const int c0 = (int)blockIdx.x * G;
    const int tid = threadIdx.x;
    const int ng = a.m - c0 < G ? a.m - c0 : G;          // candidates of this group
    if (tid < ng) {
        s_c[tid] = warp_candidate(a, c0 + tid);
    }
    __syncthreads();
    // ---- WarpAffine, per-sample part (:217-259) + GetPatchNoBoarder (:261-275) ----
    // A sample is two dependent steps — where to read (float arithmetic on the candidate's record), then the bilinear
    // blend of what was read — and every candidate's patch lies somewhere else in its keyframe: each read is an L2 / HBM
    // round trip. UNROLL samples per thread are prepared and their loads issued before the first blend, so a thread has
    // 2 x UNROLL loads in flight instead of 2 (64 candidates per group: 50 dependent round trips per thread -> 10).
    warp_samples<128>(s_c, ng, tid, s_pb, s_pp);
    __syncthreads();
    // (c0 * 100 and c0 * 64 are multiples of 4 for every G used — 2 and 64 — and the scratch sections are 256-byte aligned)
    {
        uint32_t* __restrict__ dst = (uint32_t*)(a.patch_border + (size_t)c0 * 100);
        const uint32_t* src = (const uint32_t*)s_pb;
        for (int i = tid; i < ng * 25; i += 128) dst[i] = src[i];
        uint32_t* __restrict__ dst2 = (uint32_t*)(a.patch + (size_t)c0 * 64);
        const uint32_t* src2 = (const uint32_t*)s_pp;
        for (int i = tid; i < ng * 16; i += 128) dst2[i] = src2[i];
    }
}

hipError_t warp_launch(const WarpKernelArgs& args, hipStream_t stream) {
    if (args.m <= 0) return hipSuccess;
    // few candidates (one frame's search: ~800): groups of 2 — the call is bound by the latency of one candidate's FP64
    // chain, so the sampling loop behind it is kept to two rounds and the groups spread over every compute unit;
    // batches: groups of 64 (one full wave of phase-1 lanes)
    const int g = options().warp_group ? options().warp_group : (args.m < 8192 ? 2 : 16);
    const dim3 grid((unsigned)((args.m + g - 1) / g));
    switch (g) {
        case 2: hipLaunchKernelGGL(warp_kernel<2>, grid, dim3(128), 0, stream, args); break;
        case 16: hipLaunchKernelGGL(warp_kernel<16>, grid, dim3(128), 0, stream, args); break;
#ifdef DSDTM_DIAG                                     // group sizes that were measured and not kept (A/B)
        case 8: hipLaunchKernelGGL(warp_kernel<8>, grid, dim3(128), 0, stream, args); break;
        case 32: hipLaunchKernelGGL(warp_kernel<32>, grid, dim3(128), 0, stream, args); break;
        case 64: hipLaunchKernelGGL(warp_kernel<64>, grid, dim3(128), 0, stream, args); break;
#endif
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace dsdtm
