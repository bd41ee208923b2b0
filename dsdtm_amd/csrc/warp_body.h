// warp_body.h — the device pieces of the warp prelude (warp.hip: warp_kernel; match.hip: the fused FindMatchDirect kernel).
// Include ONLY from a translation unit with `#pragma clang fp contract(off)` at file scope: the FP64 chain below follows the
// reference's operation order without contraction (see warp.hip's header).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

#pragma clang fp contract(off)      // (file scope, from here on: see above)

namespace dsdtm {


namespace {
struct SE3x { double qw, qx, qy, qz, tx, ty, tz; };

// Eigen QuaternionBase::normalize(): coeffs /= norm()
__device__ __forceinline__ void xq_normalize(SE3x& T) {
    const double n = sqrt(T.qw * T.qw + T.qx * T.qx + T.qy * T.qy + T.qz * T.qz);
    T.qw /= n; T.qx /= n; T.qy /= n; T.qz /= n;
}
// Eigen QuaternionBase::_transformVector: uv = 2 * vec x v; v + w*uv + vec x uv
__device__ __forceinline__ void xq_rotate(const SE3x& T, double vx, double vy, double vz, double& ox, double& oy, double& oz) {
    double ux = T.qy * vz - T.qz * vy;
    double uy = T.qz * vx - T.qx * vz;
    double uz = T.qx * vy - T.qy * vx;
    ux += ux; uy += uy; uz += uz;
    const double cx = T.qy * uz - T.qz * uy;
    const double cy = T.qz * ux - T.qx * uz;
    const double cz = T.qx * uy - T.qy * ux;
    ox = vx + T.qw * ux + cx;
    oy = vy + T.qw * uy + cy;
    oz = vz + T.qw * uz + cz;
}
// SO3(Matrix3d) -> Eigen rotation-matrix-to-quaternion, then normalise; translation copied
__device__ __forceinline__ SE3x xse3_from_rt(const double* __restrict__ T) {
    const double m00 = T[0], m01 = T[1], m02 = T[2];
    const double m10 = T[4], m11 = T[5], m12 = T[6];
    const double m20 = T[8], m21 = T[9], m22 = T[10];
    SE3x o;
    double t = m00 + m11 + m22;
    if (t > 0.0) {
        t = sqrt(t + 1.0);
        o.qw = 0.5 * t;
        t = 0.5 / t;
        o.qx = (m21 - m12) * t;
        o.qy = (m02 - m20) * t;
        o.qz = (m10 - m01) * t;
    } else {
        // i = argmax of the diagonal as Eigen picks it: i = 0; if (m11 > m00) i = 1; if (m22 > m[i][i]) i = 2
        const bool i1 = m11 > m00;
        const bool i2 = m22 > (i1 ? m11 : m00);
        if (i2) {                                   // i = 2, j = 0, k = 1
            t = sqrt(m22 - m00 - m11 + 1.0);
            o.qz = 0.5 * t;
            t = 0.5 / t;
            o.qw = (m10 - m01) * t;
            o.qx = (m02 + m20) * t;
            o.qy = (m12 + m21) * t;
        } else if (i1) {                            // i = 1, j = 2, k = 0
            t = sqrt(m11 - m22 - m00 + 1.0);
            o.qy = 0.5 * t;
            t = 0.5 / t;
            o.qw = (m02 - m20) * t;
            o.qz = (m21 + m12) * t;
            o.qx = (m01 + m10) * t;
        } else {                                    // i = 0, j = 1, k = 2
            t = sqrt(m00 - m11 - m22 + 1.0);
            o.qx = 0.5 * t;
            t = 0.5 / t;
            o.qw = (m21 - m12) * t;
            o.qy = (m10 + m01) * t;
            o.qz = (m20 + m02) * t;
        }
    }
    xq_normalize(o);
    o.tx = T[3]; o.ty = T[7]; o.tz = T[11];
    return o;
}
// SE3::inverse: so3_.inverse() (conjugate); translation = so3_inv * (translation * -1.)
__device__ __forceinline__ SE3x xse3_inverse(const SE3x& a) {
    SE3x r;
    r.qw = a.qw; r.qx = -a.qx; r.qy = -a.qy; r.qz = -a.qz;
    xq_rotate(r, a.tx * -1., a.ty * -1., a.tz * -1., r.tx, r.ty, r.tz);
    return r;
}
// SE3::operator*=: translation += so3 * other.translation; quaternion product, normalize()
__device__ __forceinline__ SE3x xse3_mul(const SE3x& a, const SE3x& b) {
    SE3x r;
    double rx, ry, rz;
    xq_rotate(a, b.tx, b.ty, b.tz, rx, ry, rz);
    r.tx = a.tx + rx; r.ty = a.ty + ry; r.tz = a.tz + rz;
    r.qw = a.qw * b.qw - a.qx * b.qx - a.qy * b.qy - a.qz * b.qz;
    r.qx = a.qw * b.qx + a.qx * b.qw + a.qy * b.qz - a.qz * b.qy;
    r.qy = a.qw * b.qy + a.qy * b.qw + a.qz * b.qx - a.qx * b.qz;
    r.qz = a.qw * b.qz + a.qz * b.qw + a.qx * b.qy - a.qy * b.qx;
    xq_normalize(r);
    return r;
}
}  // namespace

__device__ __forceinline__ void cam2px(const WarpKernelArgs& a, double x, double y, double z, double& u, double& v) {
    u = (double)a.fx * x / z + (double)a.cx;          // src/Camera.cpp:167-171
    v = (double)a.fy * y / z + (double)a.cy;
}

// What phase 2 needs of a candidate (phase 1 -> LDS): the inverse affine in float (:209-213), the reference pixel on its
// level (:215-216), quirk W1's integer scale, and where to sample.
typedef uint16_t __attribute__((aligned(1))) U16u;      // a 16-bit load at any byte address

struct WarpCand {
    float Ai0, Ai1, Ai2, Ai3;
    float refx, refy;
    int int_scale;
    int ok;                // 0: rejected (zero patches)
    const uint8_t* img;    // the reference level of the candidate's keyframe (phase 2 chases no pointer and indexes no table)
    int w, h, stride, pad;
};


// Phase 1 for candidate c: SolveAffineMatrix, GetBestSearchLevel, the per-candidate part of WarpAffine (the FP64 chain, ONCE
// per candidate); writes search_level[c] (and affine[c]) and returns what the sampling phase needs.
__device__ __forceinline__ WarpCand warp_candidate(const WarpKernelArgs& a, int c) {
        WarpCand wc;
        wc.Ai0 = wc.Ai1 = wc.Ai2 = wc.Ai3 = wc.refx = wc.refy = 0.0f; wc.int_scale = 0; wc.ok = 0;
        wc.img = nullptr; wc.w = wc.h = wc.stride = wc.pad = 0;
        const int k = a.cand_kf[c];
        const int tLevel = a.ref_level[c];
        const int fr = a.cand_frame ? a.cand_frame[c] : 0;
        // a candidate that names a keyframe, level or current frame outside the batch is rejected, not dereferenced
        const bool ok = !(k < 0 || k >= a.n_kf || tLevel < 0 || tLevel >= a.levels || fr < 0 || (a.cand_frame && fr >= a.n_frames));
        if (!ok) {
            a.search_level[c] = -1;
        } else {
            // ---- SolveAffineMatrix (:160-190) ----
            const SE3x Tcur = xse3_from_rt(a.cand_frame ? a.T_cur_w_arr + 12 * (size_t)fr : a.T_cur_w);
            const SE3x Tkf = xse3_from_rt(a.T_kf_w + 12 * (size_t)k);
            const SE3x Tki = xse3_inverse(Tkf);
            const double* P = a.p_world + 3 * (size_t)c;
            const double* nb = a.ref_bearing + 3 * (size_t)c;
            const double d0 = Tki.tx - P[0], d1 = Tki.ty - P[1], d2 = Tki.tz - P[2];
            const double dist = sqrt(d0 * d0 + d1 * d1 + d2 * d2);
            const double rp0 = dist * nb[0], rp1 = dist * nb[1], rp2 = dist * nb[2];          // :167
            const float rx = a.ref_px[2 * (size_t)c], ry = a.ref_px[2 * (size_t)c + 1];
            const int HPL = 5;
            const double pxU0 = (double)(rx + (float)(HPL * (1 << tLevel))), pxU1 = (double)ry;   // :171
            const double pxV0 = (double)rx, pxV1 = (double)(ry + (float)(HPL * (1 << tLevel)));  // :172
            // Pixel2Camera(Vector2d, 1.0f) (src/Camera.cpp:180-185), normalise, rescale to the ref depth
            double U0 = 1.0f * (pxU0 - (double)a.cx) / (double)a.fx, U1 = 1.0f * (pxU1 - (double)a.cy) / (double)a.fy, U2 = 1.0;
            double V0 = 1.0f * (pxV0 - (double)a.cx) / (double)a.fx, V1 = 1.0f * (pxV1 - (double)a.cy) / (double)a.fy, V2 = 1.0;
            {
                const double nU = sqrt(U0 * U0 + U1 * U1 + U2 * U2);
                U0 /= nU; U1 /= nU; U2 /= nU;
                const double nV = sqrt(V0 * V0 + V1 * V1 + V2 * V2);
                V0 /= nV; V1 /= nV; V2 /= nV;
                const double sU = rp2 / U2, sV = rp2 / V2;
                U0 *= sU; U1 *= sU; U2 *= sU;
                V0 *= sV; V1 *= sV; V2 *= sV;
            }
            const SE3x Tc2r = xse3_mul(Tcur, Tki);                                             // :181
            double q0x, q0y, q0z, qUx, qUy, qUz, qVx, qVy, qVz;
            xq_rotate(Tc2r, rp0, rp1, rp2, q0x, q0y, q0z); q0x += Tc2r.tx; q0y += Tc2r.ty; q0z += Tc2r.tz;
            xq_rotate(Tc2r, U0, U1, U2, qUx, qUy, qUz);    qUx += Tc2r.tx; qUy += Tc2r.ty; qUz += Tc2r.tz;
            xq_rotate(Tc2r, V0, V1, V2, qVx, qVy, qVz);    qVx += Tc2r.tx; qVy += Tc2r.ty; qVz += Tc2r.tz;
            double c0u, c0v, cUu, cUv, cVu, cVv;
            cam2px(a, q0x, q0y, q0z, c0u, c0v);
            cam2px(a, qUx, qUy, qUz, cUu, cUv);
            cam2px(a, qVx, qVy, qVz, cVu, cVv);
            const double A00 = (cUu - c0u) / HPL, A10 = (cUv - c0v) / HPL;                     // :186
            const double A01 = (cVu - c0u) / HPL, A11 = (cVv - c0v) / HPL;                     // :187
            // ---- GetBestSearchLevel (:192-204) ----
            int sl = 0;
            double D = A00 * A11 - A01 * A10;
            while (D > 3.0 && sl < a.max_search_level) { sl++; D = D * 0.25; }
            if (a.affine) {
                double* o = a.affine + 4 * (size_t)c;
                o[0] = A00; o[1] = A01; o[2] = A10; o[3] = A11;
            }
            a.search_level[c] = sl;
            // ---- WarpAffine, per-candidate part (:206-216, :231) ----
            const double det = A00 * A11 - A01 * A10;
            const double invdet = 1.0 / det;
            wc.Ai0 = (float)(A11 * invdet); wc.Ai1 = (float)(-A01 * invdet);
            wc.Ai2 = (float)(-A10 * invdet); wc.Ai3 = (float)(A00 * invdet);
            wc.refx = rx / (float)(1 << tLevel); wc.refy = ry / (float)(1 << tLevel);          // :215-216
            wc.int_scale = 1 / (1 << sl);                                                      // :231 quirk W1
            wc.ok = 1;
            const LevelGeom lg = a.lv[tLevel];
            wc.img = (a.kf_ptrs ? a.kf_ptrs[k] : a.kf_pyr + (size_t)k * a.kf_pitch) + lg.off;
            wc.w = lg.w; wc.h = lg.h; wc.stride = lg.stride;
        }
        return wc;
}

// Phase 2 for a group of `ng` candidates whose records are in s_c: the group's ng x 100 samples of the 10x10 bordered
// patches, thread = sample (NT threads), into s_pb (ng x 100 bytes) and s_pp (ng x 64: GetPatchNoBoarder).
template <int NT>
__device__ __forceinline__ void warp_samples(const WarpCand* s_c, int ng, int tid, uint8_t* s_pb, uint8_t* s_pp) {
    constexpr int UNROLL = 5;
    typedef const __attribute__((address_space(1))) uint8_t* GlobalU8;
    typedef const __attribute__((address_space(1))) U16u* GlobalU16;
    struct Sample { float w00, w01, w10, w11; uint32_t r0, r1; int mode; };   // mode 0: outside (0), 1: two 16-bit rows, 2: last row
    const int n_samples = ng * 100;
    for (int s0 = tid; s0 < n_samples; s0 += NT * UNROLL) {
        Sample sm[UNROLL];
#pragma unroll
        for (int uu = 0; uu < UNROLL; ++uu) {
            const int s = s0 + uu * NT;
            Sample& q = sm[uu];
            q.mode = 0; q.r0 = q.r1 = 0u; q.w00 = q.w01 = q.w10 = q.w11 = 0.0f;
            if (s >= n_samples) continue;
            const int cl = s / 100, j = s - cl * 100;
            const WarpCand wc = s_c[cl];
            if (!wc.ok) continue;
            struct { int w, h, stride; } lg = {wc.w, wc.h, wc.stride};
            GlobalU8 img = (GlobalU8)wc.img;
            const int ix = (j % 10) - 5, iy = (j / 10) - 5;
            const float gx = (wc.Ai0 * (float)ix + wc.Ai1 * (float)iy) * (float)wc.int_scale;
            const float gy = (wc.Ai2 * (float)ix + wc.Ai3 * (float)iy) * (float)wc.int_scale;
            const float wx = gx + wc.refx, wy = gy + wc.refy;                                   // :232
            if (!(wx != wx) && !(wy != wy) && !(wx < 0) && !(wy < 0) && !(wx > (float)(lg.w - 1)) && !(wy > (float)(lg.h - 1))) {
                const int fx_ = (int)floor((double)wx), fy_ = (int)floor((double)wy);
                const float sx = wx - (float)fx_, sy = wy - (float)fy_;
                const float omx = 1.0f - sx, omy = 1.0f - sy;
                q.w00 = omx * omy;
                q.w01 = omx * sy;                                                               // :242
                q.w10 = sx * omy;                                                               // :243
                q.w11 = 1.0f - q.w00 - q.w01 - q.w10;                                           // :244
                const int sz = lg.stride * lg.h, o = lg.stride * fy_ + fx_;
                if (o + lg.stride + 1 < sz) {
                    // both rows inside the level (everywhere but its last row): the two horizontal neighbours of a row in ONE
                    // 16-bit load (unaligned global loads are native on gfx950)
                    q.mode = 1;
                    q.r0 = *(GlobalU16)(img + o);
                    q.r1 = *(GlobalU16)(img + o + lg.stride);
                } else {
                    q.mode = 2;                                                                 // o + stride + 1 >= sz: p11 = 0
                    q.r0 = (uint32_t)img[o] | ((o + 1 < sz) ? (uint32_t)img[o + 1] << 8 : 0u);
                    q.r1 = (o + lg.stride < sz) ? (uint32_t)img[o + lg.stride] : 0u;
                }
            }
        }
#pragma unroll
        for (int uu = 0; uu < UNROLL; ++uu) {
            const int s = s0 + uu * NT;
            if (s >= n_samples) continue;
            const Sample& q = sm[uu];
            const int cl = s / 100, j = s - cl * 100;
            uint8_t outv = 0;
            if (q.mode) {
                const float p00 = (float)(q.r0 & 0xff), p10 = (float)((q.r0 >> 8) & 0xff);
                const float p01 = (float)(q.r1 & 0xff), p11 = (float)((q.r1 >> 8) & 0xff);
                const float val = q.w00 * p00 + q.w01 * p01 + q.w10 * p10 + q.w11 * p11;        // :254
                outv = (uint8_t)(int)val;                                                       // truncation
            }
            s_pb[s] = outv;
            const int r = j / 10, cc = j % 10;
            if (r >= 1 && r <= 8 && cc >= 1 && cc <= 8) s_pp[cl * 64 + (r - 1) * 8 + (cc - 1)] = outv;
        }
    }
}

}  // namespace dsdtm
