// align2d.hip — Feature_Alignment::Align2DGaussNewton, one wavefront per feature (gfx950).
//
// Replaces reference src/Feature_alignment.cpp:318-417. lane = one pixel of the 8x8 patch:
// the lane owns its reference intensity and the gradients derived from the 10x10 bordered
// patch (:330-343) and samples the current image bilinearly each iteration (:381-392). The 3x3
// inverse, the update and the convergence test (:345, :395-411) are evaluated redundantly by all lanes.
// float32 throughout, as the reference (Matrix3f, float u,v) — including its float/double mixing
// in the bilinear weights (:373-376).
//
// The three Jres sums (:389-391) are float sums over the 64 pixels in raster order, and the result
// feeds a threshold (du^2 + dv^2 < 0.03^2, :400) that decides "matched or not" for SearchLocalPoints:
// a differently associated sum can flip that bit for a candidate sitting on the threshold. So the
// sums run in the REFERENCE'S ORDER: every lane parks its three products in LDS and lanes 0..2 each
// fold one of the sums sequentially (64 dependent float subtractions, the three chains side by side
// on three lanes) — pixels and flags are bit-identical to the CPU restatement. (H needs no such care:
// its entries are sums of multiples of 1/4 below 2^22, exact in float in any order.) The DPP tree
// version is kept as a template parameter for the cost comparison (DSDTM_A2D_TREE=1, tools/kernels.py).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "kernels.h"
#include "align2d_body.h"

namespace dsdtm {


// wavefront sum; every lane of row 3 (lanes 48..63) holds the total, broadcast from lane 63
__device__ __forceinline__ float wave_sum_f32(float v) {
    v += dpp_f32<0x128, 0xf>(v);  // row_ror:8
    v += dpp_f32<0x124, 0xf>(v);  // row_ror:4
    v += dpp_f32<0x122, 0xf>(v);  // row_ror:2
    v += dpp_f32<0x121, 0xf>(v);  // row_ror:1
    v += dpp_f32<0x142, 0xa>(v);  // row_bcast:15
    v += dpp_f32<0x143, 0xc>(v);  // row_bcast:31
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// Jres[k] -= res * J[k] over the pixels q = 0..63 in order, starting from 0 (:389-391): lane k < 3 folds chain k.
// `prod` is this wave's 3 x 64 floats of LDS; all lanes return the three sums.
__device__ __forceinline__ void jres_in_reference_order(float* prod, int lane, float p0, float p1, float p2,
                                                        float& j0, float& j1, float& j2) {
#pragma clang fp contract(off)
    prod[lane] = p0; prod[64 + lane] = p1; prod[128 + lane] = p2;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    float acc = 0.0f;
    if (lane < 3) {
        const float4* src = (const float4*)(prod + 64 * lane);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float4 v = src[q];
            acc = acc - v.x; acc = acc - v.y; acc = acc - v.z; acc = acc - v.w;
        }
    }
    j0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), 0));
    j1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), 1));
    j2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), 2));
    __builtin_amdgcn_wave_barrier();      // the next iteration's stores come after every lane's loads
}

template <bool TREE>
__global__ __launch_bounds__(256) void align2d_kernel(const A2DKernelArgs a) {
    // no FMA contraction: the reference build has none (CMakeLists.txt:5-8, SSE only) and the
    // 0.03^2 convergence threshold (:400) is compared on float values
#pragma clang fp contract(off)
    __shared__ __attribute__((aligned(16))) float s_prod[4][192];
    const int lane = threadIdx.x & 63;
    const int f = blockIdx.x * 4 + (threadIdx.x >> 6);   // wave-uniform
    float* const prod = s_prod[threadIdx.x >> 6];
    if (f >= a.m) return;
    const int lvl = a.level[f];
    const int fr = a.frame ? a.frame[f] : 0;
    if (lvl < 0 || lvl >= a.levels || fr < 0 || (a.frame && fr >= a.n_frames)) {   // invalid level / frame: report "not converged"
        if (lane == 0) a.converged[f] = 0;
        return;
    }
    const LevelGeom lg = a.lv[lvl];
    const uint8_t* __restrict__ img = a.cur_pyr + (size_t)fr * a.pyr_pitch + lg.off;
    const int img_size = lg.stride * lg.h;

    const int r = lane >> 3, c = lane & 7;
    const uint8_t* __restrict__ bp = a.patch_border + (size_t)f * 100 + (r + 1) * 10 + (c + 1);
    // :336-337  0.5*(it[1]-it[-1]) is exact in float
    const float dx = 0.5f * (float)((int)bp[1] - (int)bp[-1]);
    const float dy = 0.5f * (float)((int)bp[10] - (int)bp[-10]);
    const float ref = (float)a.patch[(size_t)f * 64 + lane];

    // H = sum J J^T, J = [dx, dy, 1]  (:341)
    const float h00 = wave_sum_f32(dx * dx);
    const float h01 = wave_sum_f32(dx * dy);
    const float h02 = wave_sum_f32(dx);
    const float h11 = wave_sum_f32(dy * dy);
    const float h12 = wave_sum_f32(dy);
    const float h22 = 64.0f;
    // Matrix3f::inverse() (:345): Eigen cofactor formula, no conditioning check (quirk A2)
    const float m00 = h00, m01 = h01, m02 = h02, m10 = h01, m11 = h11, m12 = h12, m20 = h02, m21 = h12, m22 = h22;
    const float c00 = m11 * m22 - m12 * m21;
    const float c10 = m21 * m02 - m22 * m01;   // cofactor_3x3<1,0>
    const float c20 = m01 * m12 - m02 * m11;   // cofactor_3x3<2,0>
    const float det = c00 * m00 + (c10 * m10 + c20 * m20);
    const float invdet = 1.0f / det;
    const float i00 = c00 * invdet, i01 = c10 * invdet, i02 = c20 * invdet;
    const float i10 = (m12 * m20 - m10 * m22) * invdet;   // cofactor<0,1>
    const float i11 = (m22 * m00 - m20 * m02) * invdet;   // cofactor<1,1>
    const float i12 = (m02 * m10 - m00 * m12) * invdet;   // cofactor<2,1>
    const float i20 = (m10 * m21 - m11 * m20) * invdet;   // cofactor<0,2>
    const float i21 = (m20 * m01 - m21 * m00) * invdet;   // cofactor<1,2>
    const float i22 = (m00 * m11 - m01 * m10) * invdet;   // cofactor<2,2>

    // (px_level0: the caller's pixel is in level-0 units; the division by 2^level is exact in double, :150)
    const double lscale = a.px_level0 ? (double)(1 << lvl) : 1.0;
    float u = (float)(a.px_xy[2 * (size_t)f] / lscale);
    float v = (float)(a.px_xy[2 * (size_t)f + 1] / lscale);
    float mean_diff = 0.0f;
    const float min_update_squared = (float)(0.03 * 0.03);
    bool converged = false;
    for (int it = 0; it < a.max_iters; ++it) {
        if (u != u || v != v) break;                                         // :368 isnan
        const float fu = floorf(u), fv = floorf(v);
        // compare as floats: the int conversion of a huge float would be undefined
        if (fu < 4.0f || fv < 4.0f || fu > (float)(lg.w - 4) || fv > (float)(lg.h - 4)) break;   // :367-368
        const int u_r = (int)fu, v_r = (int)fv;
        const float sx = u - (float)u_r, sy = v - (float)v_r;
        const float wTL = (float)((1.0 - (double)sx) * (1.0 - (double)sy));  // :373 (double arithmetic)
        const float wTR = sx * (1.0f - sy);                                   // :374 (float arithmetic)
        const float wBL = (float)((1.0 - (double)sx) * (double)sy);          // :375
        const float wBR = sx * sy;                                            // :376
        const int o = (v_r + r - 4) * lg.stride + (u_r + c - 4);             // :383
        // quirk A3: offsets past the level image read as 0 (undefined in the reference)
        const float p00 = (o < img_size) ? (float)img[o] : 0.0f;
        const float p01 = (o + 1 < img_size) ? (float)img[o + 1] : 0.0f;
        const float p10 = (o + lg.stride < img_size) ? (float)img[o + lg.stride] : 0.0f;
        const float p11 = (o + lg.stride + 1 < img_size) ? (float)img[o + lg.stride + 1] : 0.0f;
        const float search = wTL * p00 + wTR * p01 + wBL * p10 + wBR * p11;  // :386
        const float res = search - ref + mean_diff;                          // :387
        float j0, j1, j2;                                                    // :389-391
        if (TREE) { j0 = -wave_sum_f32(res * dx); j1 = -wave_sum_f32(res * dy); j2 = -wave_sum_f32(res); }
        else jres_in_reference_order(prod, lane, res * dx, res * dy, res, j0, j1, j2);
        const float up0 = (i00 * j0 + i01 * j1) + i02 * j2;                  // :395
        const float up1 = (i10 * j0 + i11 * j1) + i12 * j2;
        const float up2 = (i20 * j0 + i21 * j1) + i22 * j2;
        u += up0;
        v += up1;
        mean_diff += up2;
        if (up0 * up0 + up1 * up1 < min_update_squared) { converged = true; break; }   // :400
    }
    if (lane == 0) {
        a.px_xy[2 * (size_t)f] = (double)u * lscale;                         // :414 always written back (:154-156 back to level 0)
        a.px_xy[2 * (size_t)f + 1] = (double)v * lscale;
        a.converged[f] = converged ? 1 : 0;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The product kernel: SEVERAL features per wavefront — 64 / PPL lanes per feature, PPL pixels of a patch row per lane
// (PPL = 4: four features per wave, one per 16-lane DPP row; PPL = 8: eight features per wave, lane = patch row).
// One wavefront per feature (rounds 1-3, the template above, now the DSDTM_A2D_TREE diagnostic only) carried ONE feature
// through an iteration's chain — footprint loads -> 64 bilinear samples -> 64 sequential float subtractions -> update — with
// 16 byte gathers per pixel row, and issued the sequential part (which only three lanes need) once per feature. Here a wave
// carries 4 or 8 features through the same chain at once (the reductions of H are group-local DPP sums, the 12 or 24
// sequential Jres chains run side by side), a lane fetches the two footprint rows of its pixels with two wide loads, and a
// launch needs a quarter / an eighth of the waves. Per pixel the arithmetic is the old kernel's, expression for expression;
// the Jres sums run in the reference's raster order (:389-391) as before: pixels and flags stay bit-identical to the CPU
// restatement. Four features per wave is the product shape (70 VGPRs, 7 waves per SIMD); eight (DSDTM_A2D_GROUP=8: 114
// VGPRs, 4 waves per SIMD) was measured 7 % slower — the kernel lives on the waves a compute unit holds.
template <int PPL>
__global__ __launch_bounds__(256) void align2d_rows_kernel(const A2DKernelArgs a) {
    // no FMA contraction: the reference build has none (CMakeLists.txt:5-8, SSE only) and the
    // 0.03^2 convergence threshold (:400) is compared on float values
#pragma clang fp contract(off)
    constexpr int LPF = 64 / PPL;                  // lanes per feature
    constexpr int FPW = 64 / LPF;                  // features per wave
    constexpr int FPB = 4 * FPW;                   // features per 256-thread group
    __shared__ __attribute__((aligned(16))) float s_prod[FPB][192];     // per feature: 3 x 64 products (Jres chains)
    __shared__ LevelGeom s_lv[DSDTM_MAX_LEVELS];
    if (threadIdx.x < DSDTM_MAX_LEVELS) s_lv[threadIdx.x] = a.lv[threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int grp = lane / LPF, l = lane % LPF;                         // feature of the wave, lane of the feature
    const int slot = (threadIdx.x >> 6) * FPW + grp;                    // feature of the workgroup
    const int f = blockIdx.x * FPB + slot;
    float* const prod = s_prod[slot];
    const bool exists = f < a.m;
    const int lvl = exists ? a.level[f] : -1;
    const int fr = (exists && a.frame) ? a.frame[f] : 0;
    const bool valid = exists && !(lvl < 0 || lvl >= a.levels || fr < 0 || (a.frame && fr >= a.n_frames));
    if (exists && !valid && l == 0) a.converged[f] = 0;                 // invalid level / frame: report "not converged"
    const LevelGeom lg = s_lv[valid ? lvl : 0];
    const uint8_t* __restrict__ img = a.cur_pyr + (size_t)fr * a.pyr_pitch + lg.off;
    const int img_size = lg.stride * lg.h;

    const double lscale = (a.px_level0 && valid) ? (double)(1 << lvl) : 1.0;
    float u, v;
    bool converged;
    align2d_rows_feature<PPL>(valid, img, lg, img_size, a.patch_border + (size_t)(valid ? f : 0) * 100, a.patch + (size_t)(valid ? f : 0) * 64, prod,
                              valid ? a.px_xy[2 * (size_t)f] : 0.0, valid ? a.px_xy[2 * (size_t)f + 1] : 0.0, lscale, a.max_iters, lane, u, v, converged);
    if (valid && l == 0) {
        a.px_xy[2 * (size_t)f] = (double)u * lscale;                         // :414 always written back (:154-156 back to level 0)
        a.px_xy[2 * (size_t)f + 1] = (double)v * lscale;
        a.converged[f] = converged ? 1 : 0;
    }
}

hipError_t align2d_launch(const A2DKernelArgs& args, hipStream_t stream) {
    if (args.m <= 0) return hipSuccess;
#ifdef DSDTM_DIAG
    // a2d_tree (diagnostic build, cost comparison only): DPP tree sums instead of the reference's order; a2d_group 8: 114 VGPRs
    if (options().a2d_tree) { hipLaunchKernelGGL(align2d_kernel<true>, dim3((unsigned)((args.m + 3) / 4)), dim3(256), 0, stream, args); return hipGetLastError(); }
    if (options().a2d_group == 8) { hipLaunchKernelGGL(align2d_rows_kernel<8>, dim3((unsigned)((args.m + 31) / 32)), dim3(256), 0, stream, args); return hipGetLastError(); }
#endif
    hipLaunchKernelGGL(align2d_rows_kernel<4>, dim3((unsigned)((args.m + 15) / 16)), dim3(256), 0, stream, args);
    return hipGetLastError();
}

}  // namespace dsdtm
