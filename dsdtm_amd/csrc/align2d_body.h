// align2d_body.h — Feature_Alignment::Align2DGaussNewton for the features of one wavefront, several per wave (the body of
// align2d.hip's align2d_rows_kernel; also run by match.hip's fused FindMatchDirect kernel on patches that never left LDS).
// Reference: src/Feature_alignment.cpp:318-417. No FMA contraction (function-scope pragma below).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace dsdtm {

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f32(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}

typedef uint32_t __attribute__((aligned(1))) U32u;
typedef uint16_t __attribute__((aligned(1))) U16a;
struct __attribute__((packed, aligned(1))) U64u { uint32_t lo, hi; };
__device__ __forceinline__ unsigned long long load64u(const uint8_t* p) {
    const U64u w = *(const U64u*)p;
    return (unsigned long long)w.lo | ((unsigned long long)w.hi << 32);
}
// up to 12 bytes of a row: byte k of (lo, hi)
struct RowBytes {
    unsigned long long lo;
    uint32_t hi;
    __device__ __forceinline__ int at(int k) const { return k < 8 ? (int)((lo >> (8 * k)) & 0xff) : (int)((hi >> (8 * (k - 8))) & 0xff); }
};
// N bytes at p (N = 4 or 8: one load; N = 5..6: 4 + 2; N = 9..10: 8 + 2), nothing read beyond p + N (rounded up to even)
template <int N>
__device__ __forceinline__ RowBytes load_row_bytes(const uint8_t* p) {
    RowBytes r;
    r.hi = 0;
    if constexpr (N <= 4) r.lo = *(const U32u*)p;
    else if constexpr (N <= 6) r.lo = (unsigned long long)*(const U32u*)p | ((unsigned long long)*(const U16a*)(p + 4) << 32);
    else if constexpr (N <= 8) r.lo = load64u(p);
    else { r.lo = load64u(p); r.hi = *(const U16a*)(p + 8); }
    return r;
}

// sum over the LPF = 16 or 8 lanes of a feature, result in every one of them (integers below 2^22 in float: exact in any order)
template <int LPF>
__device__ __forceinline__ float group_sum_f32(float v) {
    if constexpr (LPF == 16) {
        v += dpp_f32<0x128, 0xf>(v);  // row_ror:8
        v += dpp_f32<0x124, 0xf>(v);  // row_ror:4
        v += dpp_f32<0x122, 0xf>(v);  // row_ror:2
        v += dpp_f32<0x121, 0xf>(v);  // row_ror:1
    } else {
        v += dpp_f32<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
        v += dpp_f32<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
        v += dpp_f32<0x141, 0xf>(v);  // row_half_mirror: lane i <-> 7 - i of every 8
    }
    return v;
}
__device__ __forceinline__ float lane_bcast_f32(float v, int src_lane) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane << 2, __float_as_int(v)));
}

// One feature per 64 / PPL lanes. `valid`: the feature exists and names a level / frame inside the batch (group-uniform);
// img / lg / img_size: its level of the current frame; pb100 / pp64: its bordered and plain reference patches (global memory or
// LDS: PB is the pointer type); prod: the feature's 192 floats of LDS; (px_u, px_v) / lscale: the start pixel on the level.
// Returns the refined pixel (level units) in u, v and the reference's bool in `converged` — every lane of the feature.
template <int PPL, typename PB>
__device__ __forceinline__ void align2d_rows_feature(bool valid, const uint8_t* __restrict__ img, const LevelGeom lg, int img_size,
                                                     PB pb100, PB pp64, float* prod, double px_u, double px_v, double lscale,
                                                     int max_iters, int lane, float& u_out, float& v_out, bool& conv_out) {
#pragma clang fp contract(off)
    constexpr int LPF = 64 / PPL;                  // lanes per feature
    const int grp = lane / LPF, l = lane % LPF;    // feature of the wave, lane of the feature
    // this lane's pixels: q = PPL l .. PPL l + PPL - 1 of the 8x8 patch in raster order -> row r, columns c0 .. c0 + PPL - 1
    const int r = (PPL * l) >> 3, c0 = (PPL * l) & 7;
    float dx[PPL], dy[PPL], ref[PPL];
    {
        // bordered-patch rows r, r + 1, r + 2 around the pixels (10x10, the patch sits at (1, 1)):
        // gradients :336-337, 0.5 * (it[1] - it[-1]) is exact in float
        RowBytes up, mid, dn, pw;
        up.lo = mid.lo = dn.lo = pw.lo = 0ull; up.hi = mid.hi = dn.hi = pw.hi = 0;
        if (valid) {
            const PB bp = pb100 + r * 10 + c0;
            up = load_row_bytes<PPL>(bp + 1);                           // row r,     columns c0 + 1 .. c0 + PPL
            mid = load_row_bytes<PPL + 2>(bp + 10);                     // row r + 1, columns c0 .. c0 + PPL + 1
            dn = load_row_bytes<PPL>(bp + 21);                          // row r + 2, columns c0 + 1 .. c0 + PPL
            pw = load_row_bytes<PPL>(pp64 + PPL * l);
        }
#pragma unroll
        for (int i = 0; i < PPL; ++i) {
            dx[i] = 0.5f * (float)(mid.at(i + 2) - mid.at(i));
            dy[i] = 0.5f * (float)(dn.at(i) - up.at(i));
            ref[i] = (float)pw.at(i);
        }
    }
    // H = sum J J^T, J = [dx, dy, 1]  (:341) over the feature's 64 pixels: PPL per lane, then the lanes of the feature
    float s00 = 0.0f, s01 = 0.0f, s02 = 0.0f, s11 = 0.0f, s12 = 0.0f;
#pragma unroll
    for (int i = 0; i < PPL; ++i) { s00 += dx[i] * dx[i]; s01 += dx[i] * dy[i]; s02 += dx[i]; s11 += dy[i] * dy[i]; s12 += dy[i]; }
    const float h00 = group_sum_f32<LPF>(s00), h01 = group_sum_f32<LPF>(s01), h02 = group_sum_f32<LPF>(s02),
                h11 = group_sum_f32<LPF>(s11), h12 = group_sum_f32<LPF>(s12);
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
    float u = valid ? (float)(px_u / lscale) : 0.0f;
    float v = valid ? (float)(px_v / lscale) : 0.0f;
    float mean_diff = 0.0f;
    const float min_update_squared = (float)(0.03 * 0.03);
    bool converged = false;
    bool active = valid;                                               // group-uniform: the features of a wave end on their own
    for (int it = 0; it < max_iters; ++it) {
        if (active && (u != u || v != v)) active = false;                                        // :368 isnan
        const float fu = floorf(u), fv = floorf(v);
        // compare as floats: the int conversion of a huge float would be undefined
        if (active && (fu < 4.0f || fv < 4.0f || fu > (float)(lg.w - 4) || fv > (float)(lg.h - 4))) active = false;   // :367-368
        if (__builtin_amdgcn_ballot_w64(active) == 0ull) break;
        float pr[3][PPL];
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int i = 0; i < PPL; ++i) pr[k][i] = 0.0f;
        if (active) {
            const int u_r = (int)fu, v_r = (int)fv;
            const float sx = u - (float)u_r, sy = v - (float)v_r;
            const float wTL = (float)((1.0 - (double)sx) * (1.0 - (double)sy));  // :373 (double arithmetic)
            const float wTR = sx * (1.0f - sy);                                   // :374 (float arithmetic)
            const float wBL = (float)((1.0 - (double)sx) * (double)sy);          // :375
            const float wBR = sx * sy;                                            // :376
            const int o = (v_r + r - 4) * lg.stride + (u_r + c0 - 4);            // :383, this lane's first pixel
            // the two footprint rows of the lane's pixels: bytes o .. o + PPL and o + stride .. o + stride + PPL
            RowBytes b0, b1;
            constexpr int SPAN = PPL + 1 <= 8 ? 8 : 10;                           // bytes the fast path reads per row
            if (o + lg.stride + SPAN <= img_size) {
                b0 = load_row_bytes<SPAN>(img + o);
                b1 = load_row_bytes<SPAN>(img + o + lg.stride);
            } else {
                // quirk A3: offsets past the level image read as 0 (undefined in the reference)
                b0.lo = b1.lo = 0ull; b0.hi = b1.hi = 0;
#pragma unroll
                for (int k = 0; k < PPL + 1; ++k) {
                    const uint32_t x0 = (o + k < img_size) ? (uint32_t)img[o + k] : 0u;
                    const uint32_t x1 = (o + lg.stride + k < img_size) ? (uint32_t)img[o + lg.stride + k] : 0u;
                    if (k < 8) { b0.lo |= (unsigned long long)x0 << (8 * k); b1.lo |= (unsigned long long)x1 << (8 * k); }
                    else { b0.hi |= x0 << (8 * (k - 8)); b1.hi |= x1 << (8 * (k - 8)); }
                }
            }
#pragma unroll
            for (int i = 0; i < PPL; ++i) {
                const float p00 = (float)b0.at(i), p01 = (float)b0.at(i + 1);
                const float p10 = (float)b1.at(i), p11 = (float)b1.at(i + 1);
                const float search = wTL * p00 + wTR * p01 + wBL * p10 + wBR * p11;  // :386
                const float res = search - ref[i] + mean_diff;                       // :387
                pr[0][i] = res * dx[i]; pr[1][i] = res * dy[i]; pr[2][i] = res;
            }
        }
        // Jres[k] -= res * J[k] over the pixels q = 0..63 in order, starting from 0 (:389-391): lane k < 3 of the feature
        // folds chain k over the feature's 64 products (this lane's are q = PPL l .. PPL l + PPL - 1)
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int i = 0; i < PPL; i += 4)
                *(float4*)(prod + 64 * k + PPL * l + i) = make_float4(pr[k][i], pr[k][i + 1], pr[k][i + 2], pr[k][i + 3]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        float acc = 0.0f;
        if (l < 3 && active) {
            const float4* src = (const float4*)(prod + 64 * l);
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const float4 w = src[q];
                acc = acc - w.x; acc = acc - w.y; acc = acc - w.z; acc = acc - w.w;
            }
        }
        const float j0 = lane_bcast_f32(acc, grp * LPF), j1 = lane_bcast_f32(acc, grp * LPF + 1), j2 = lane_bcast_f32(acc, grp * LPF + 2);
        __builtin_amdgcn_wave_barrier();      // the next iteration's stores come after every lane's loads
        if (active) {
            const float up0 = (i00 * j0 + i01 * j1) + i02 * j2;                  // :395
            const float up1 = (i10 * j0 + i11 * j1) + i12 * j2;
            const float up2 = (i20 * j0 + i21 * j1) + i22 * j2;
            u += up0;
            v += up1;
            mean_diff += up2;
            if (up0 * up0 + up1 * up1 < min_update_squared) { converged = true; active = false; }   // :400
        }
    }
    u_out = u; v_out = v; conv_out = converged;
}

}  // namespace dsdtm
