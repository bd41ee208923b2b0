// device_math.h — FP64 building blocks shared by the gfx950 kernels.
//
// SE(3) arithmetic follows Sophus' non-templated SE3/SO3 (unit quaternion + translation)
// because that is what the reference composes poses with (reference
// src/Sprase_ImageAlign.cpp:43,57,254,335); the 6x6 solve follows Eigen 3.2's
// LDLT (diagonal pivoting, rank cutoff, pseudo-inverse of D) used at :318, so that
// degenerate systems (no visible patch, rank-deficient H) behave as in the reference.
// Everything is written for static register indexing: no runtime-indexed private arrays.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dsdtm {

struct SE3d {
    double qw, qx, qy, qz;
    double tx, ty, tz;
};

__device__ __forceinline__ void quat_normalize(SE3d& T) {
    // Eigen divides the four coefficients by sqrt(|q|^2). The solver wave is the serial part of
    // every Gauss-Newton iteration, so its dependent chain is kept short: every quaternion
    // normalised on this path is a product of unit quaternions, |q|^2 = 1 + e with |e| ~ 1e-16, and
    // 1/sqrt(1+e) = 1 - e/2 + 3e^2/8 to better than double precision for |e| < 1e-5 (<= 1 ulp per
    // coefficient vs the division); anything else takes the exact path.
    const double n2 = T.qw * T.qw + T.qx * T.qx + T.qy * T.qy + T.qz * T.qz;
    const double e = n2 - 1.0;
    double rn;
    if (fabs(e) < 1e-5) rn = 1.0 + e * (-0.5 + 0.375 * e);
    else rn = 1.0 / sqrt(n2);
    T.qw *= rn; T.qx *= rn; T.qy *= rn; T.qz *= rn;
}

__device__ __forceinline__ void quat_rotate(const SE3d& T, double vx, double vy, double vz,
                                            double& ox, double& oy, double& oz) {
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

// [R|t] 3x4 row-major -> SE3 (Eigen rotation-matrix -> quaternion, then normalise)
__device__ inline SE3d se3_from_rt(const double* __restrict__ T) {
    const double m00 = T[0], m01 = T[1], m02 = T[2];
    const double m10 = T[4], m11 = T[5], m12 = T[6];
    const double m20 = T[8], m21 = T[9], m22 = T[10];
    SE3d o;
    double t = m00 + m11 + m22;
    if (t > 0.0) {
        t = sqrt(t + 1.0);
        o.qw = 0.5 * t;
        t = 0.5 / t;
        o.qx = (m21 - m12) * t;
        o.qy = (m02 - m20) * t;
        o.qz = (m10 - m01) * t;
    } else if (m00 >= m11 && m00 >= m22) {          // i = 0, j = 1, k = 2
        t = sqrt(m00 - m11 - m22 + 1.0);
        o.qx = 0.5 * t;
        t = 0.5 / t;
        o.qw = (m21 - m12) * t;
        o.qy = (m10 + m01) * t;
        o.qz = (m20 + m02) * t;
    } else if (m11 > m00 && m11 >= m22) {           // i = 1, j = 2, k = 0
        t = sqrt(m11 - m22 - m00 + 1.0);
        o.qy = 0.5 * t;
        t = 0.5 / t;
        o.qw = (m02 - m20) * t;
        o.qz = (m21 + m12) * t;
        o.qx = (m01 + m10) * t;
    } else {                                        // i = 2, j = 0, k = 1
        t = sqrt(m22 - m00 - m11 + 1.0);
        o.qz = 0.5 * t;
        t = 0.5 / t;
        o.qw = (m10 - m01) * t;
        o.qx = (m02 + m20) * t;
        o.qy = (m12 + m21) * t;
    }
    quat_normalize(o);
    o.tx = T[3]; o.ty = T[7]; o.tz = T[11];
    return o;
}

// quaternion -> rotation matrix (Eigen toRotationMatrix), row-major R[9]
__device__ __forceinline__ void quat_to_matrix(const SE3d& q, double* R) {
    const double tx = 2.0 * q.qx, ty = 2.0 * q.qy, tz = 2.0 * q.qz;
    const double twx = tx * q.qw, twy = ty * q.qw, twz = tz * q.qw;
    const double txx = tx * q.qx, txy = ty * q.qx, txz = tz * q.qx;
    const double tyy = ty * q.qy, tyz = tz * q.qy, tzz = tz * q.qz;
    R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz;         R[2] = txz + twy;
    R[3] = txy + twz;         R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;         R[7] = tyz + twx;         R[8] = 1.0 - (txx + tyy);
}

__device__ inline SE3d se3_mul(const SE3d& a, const SE3d& b) {
    SE3d r;
    double rx, ry, rz;
    quat_rotate(a, b.tx, b.ty, b.tz, rx, ry, rz);
    r.tx = a.tx + rx; r.ty = a.ty + ry; r.tz = a.tz + rz;
    r.qw = a.qw * b.qw - a.qx * b.qx - a.qy * b.qy - a.qz * b.qz;
    r.qx = a.qw * b.qx + a.qx * b.qw + a.qy * b.qz - a.qz * b.qy;
    r.qy = a.qw * b.qy + a.qy * b.qw + a.qz * b.qx - a.qx * b.qz;
    r.qz = a.qw * b.qz + a.qz * b.qw + a.qx * b.qy - a.qy * b.qx;
    quat_normalize(r);
    return r;
}

__device__ inline SE3d se3_inverse(const SE3d& a) {
    SE3d r;
    r.qw = a.qw; r.qx = -a.qx; r.qy = -a.qy; r.qz = -a.qz;
    quat_rotate(r, -a.tx, -a.ty, -a.tz, r.tx, r.ty, r.tz);
    return r;
}

// SE3::exp([upsilon, omega]) — Sophus: q = (cos(th/2), sin(th/2)/th * omega), t = V * upsilon with
// V = I + (1-cos th)/th^2 * Om + (th - sin th)/th^3 * Om^2.
// Gauss-Newton steps are small rotations, and for small th every factor above is a short power
// series in th^2 (used for th^2 < 0.01) — no sqrt, no division, no trig call on the solver's
// dependent chain:
//   cos(h)            = sum (-1)^k h^2k / (2k)!          h = th/2
//   sin(h)/th         = 1/2 * sum (-1)^k h^2k / (2k+1)!
//   (1-cos th)/th^2   = 2 (sin(h)/th)^2
//   (th-sin th)/th^3  = sum (-1)^k th^2k / (2k+3)!
// (truncation < 1e-19 relative; <= 2 ulp vs libm). Larger angles use the closed forms with one
// sincos(th/2). Sophus' own th < 1e-10 branch (V = R) is kept.
// SMALL = the caller knows theta^2 < 0.01 (a Gauss-Newton step that was accepted on the series path): the closed-form arm
// with its sincos — ~150 instructions and two dozen 64-bit constants that the compiler would otherwise materialise in
// registers ahead of the caller's loops — is not compiled in. Same arithmetic, same bits.
template <bool SMALL = false>
__device__ __forceinline__ SE3d se3_exp_impl(const double* x) {
    const double ux = x[0], uy = x[1], uz = x[2];
    const double wx = x[3], wy = x[4], wz = x[5];
    const double theta_sq = wx * wx + wy * wy + wz * wz;
    double ch, imag_factor, a, b;
    if (SMALL || theta_sq < 0.01) {      // |omega| < 0.1 rad: every Gauss-Newton step in practice
        const double h2 = 0.25 * theta_sq;
        // Horner in h2 / theta_sq, six terms each: truncation < 1e-19 for theta_sq < 0.01
        ch = 1.0 + h2 * (-1.0 / 2 + h2 * (1.0 / 24 + h2 * (-1.0 / 720 + h2 * (1.0 / 40320 + h2 * (-1.0 / 3628800)))));
        const double sinc = 1.0 + h2 * (-1.0 / 6 + h2 * (1.0 / 120 + h2 * (-1.0 / 5040 + h2 * (1.0 / 362880 +
                            h2 * (-1.0 / 39916800)))));
        imag_factor = 0.5 * sinc;
        a = 2.0 * imag_factor * imag_factor;
        const double t2 = theta_sq;
        b = 1.0 / 6 + t2 * (-1.0 / 120 + t2 * (1.0 / 5040 + t2 * (-1.0 / 362880 + t2 * (1.0 / 39916800 +
            t2 * (-1.0 / 6227020800.0)))));
        if (theta_sq < 1e-20) {          // Sophus: theta < SMALL_EPS
            const double theta_po4 = theta_sq * theta_sq;
            imag_factor = 0.5 - 0.0208333 * theta_sq + 0.000260417 * theta_po4;
        }
    } else {
        const double theta = sqrt(theta_sq);
        double sh;
        sincos(0.5 * theta, &sh, &ch);
        const double inv_theta = 1.0 / theta;
        imag_factor = sh * inv_theta;
        a = 2.0 * imag_factor * imag_factor;
        b = (theta - 2.0 * sh * ch) * (inv_theta * inv_theta * inv_theta);
    }
    SE3d o;
    o.qw = ch;
    o.qx = imag_factor * wx;
    o.qy = imag_factor * wy;
    o.qz = imag_factor * wz;
    quat_normalize(o);
    double V[9];
    if (theta_sq < 1e-20) {
        quat_to_matrix(o, V);            // Sophus: V = so3.matrix()
    } else {
        // Omega = hat(omega); Omega^2 written out as the matrix product
        const double o00 = -wz * wz - wy * wy, o01 = wy * wx, o02 = wz * wx;
        const double o10 = wx * wy, o11 = -wz * wz - wx * wx, o12 = wz * wy;
        const double o20 = wx * wz, o21 = wy * wz, o22 = -wy * wy - wx * wx;
        V[0] = 1.0 + b * o00;          V[1] = a * (-wz) + b * o01;   V[2] = a * wy + b * o02;
        V[3] = a * wz + b * o10;       V[4] = 1.0 + b * o11;          V[5] = a * (-wx) + b * o12;
        V[6] = a * (-wy) + b * o20;    V[7] = a * wx + b * o21;       V[8] = 1.0 + b * o22;
    }
    o.tx = V[0] * ux + V[1] * uy + V[2] * uz;
    o.ty = V[3] * ux + V[4] * uy + V[5] * uz;
    o.tz = V[6] * ux + V[7] * uy + V[8] * uz;
    return o;
}
__device__ inline SE3d se3_exp(const double* x) { return se3_exp_impl<false>(x); }
__device__ __forceinline__ SE3d se3_exp_small(const double* x) { return se3_exp_impl<true>(x); }
// out of line (by value: nothing of the caller goes through memory): for callers that only take it on a rare path
__device__ __attribute__((noinline)) SE3d se3_exp_call(double x0, double x1, double x2, double x3, double x4, double x5) {
    const double x[6] = {x0, x1, x2, x3, x4, x5};
    return se3_exp_impl<false>(x);
}

// exp of a small twist x = [upsilon, omega] (theta^2 = |omega|^2 < 0.01) in MATRIX form:
//   dR = I + a W + b W^2,   dt = (I + b W + c W^2) upsilon,   W = hat(omega), W^2 = omega omega^T - theta^2 I
//   a = sin th / th, b = (1 - cos th) / th^2, c = (th - sin th) / th^3 as power series in theta^2 (six terms:
//   truncation < 1e-19). The same group element as se3_exp(x) — Rodrigues' formula instead of the quaternion —
// without sqrt, division or normalisation: the short way from the Gauss-Newton step to the rotation matrix and
// translation the next residual pass needs (solver_step publishes R_state * dR, t_state + R_state * dt and
// brings its quaternion state up to date afterwards, see sparse_align.hip).
__device__ __forceinline__ void se3_exp_matrix_small(const double* x, double theta_sq, double* dR, double* dt) {
    const double ux = x[0], uy = x[1], uz = x[2];
    const double wx = x[3], wy = x[4], wz = x[5];
    const double t2 = theta_sq;
    const double a = 1.0 + t2 * (-1.0 / 6 + t2 * (1.0 / 120 + t2 * (-1.0 / 5040 + t2 * (1.0 / 362880 + t2 * (-1.0 / 39916800)))));
    const double b = 0.5 + t2 * (-1.0 / 24 + t2 * (1.0 / 720 + t2 * (-1.0 / 40320 + t2 * (1.0 / 3628800 + t2 * (-1.0 / 479001600.0)))));
    const double c = 1.0 / 6 + t2 * (-1.0 / 120 + t2 * (1.0 / 5040 + t2 * (-1.0 / 362880 + t2 * (1.0 / 39916800 +
                     t2 * (-1.0 / 6227020800.0)))));
    const double xy = wx * wy, xz = wx * wz, yz = wy * wz;
    const double dxx = wx * wx - t2, dyy = wy * wy - t2, dzz = wz * wz - t2;
    const double awx = a * wx, awy = a * wy, awz = a * wz;
    dR[0] = 1.0 + b * dxx;   dR[1] = b * xy - awz;    dR[2] = b * xz + awy;
    dR[3] = b * xy + awz;    dR[4] = 1.0 + b * dyy;   dR[5] = b * yz - awx;
    dR[6] = b * xz - awy;    dR[7] = b * yz + awx;    dR[8] = 1.0 + b * dzz;
    const double bwx = b * wx, bwy = b * wy, bwz = b * wz;
    const double v00 = 1.0 + c * dxx, v01 = c * xy - bwz, v02 = c * xz + bwy;
    const double v10 = c * xy + bwz, v11 = 1.0 + c * dyy, v12 = c * yz - bwx;
    const double v20 = c * xz - bwy, v21 = c * yz + bwx, v22 = 1.0 + c * dzz;
    dt[0] = v00 * ux + v01 * uy + v02 * uz;
    dt[1] = v10 * ux + v11 * uy + v12 * uz;
    dt[2] = v20 * ux + v21 * uy + v22 * uz;
}

// ---------------------------------------------------------------------------------------
// Eigen 3.2 LDLT<Matrix<double,6,6>,Lower>::compute + solve, fully unrolled so that every
// matrix index is a compile-time constant (the matrix stays in VGPRs). The pivot row is a
// wave-uniform value; `if (piv == P)` selects the statically indexed swap.
// H is given as its 21 upper-triangular entries, row-major: (0,0),(0,1)..(0,5),(1,1)...
// ---------------------------------------------------------------------------------------
// lower-triangular packed storage: only (i >= j) is ever addressed
#define DSDTM_M(i, j) m[((i) * ((i) + 1)) / 2 + (j)]

template <int K, int P>
__device__ __forceinline__ void ldlt_swap(double* m) {
    // symmetric transposition K <-> P (P > K) on the lower triangle
#pragma unroll
    for (int j = 0; j < K; ++j) { double t = DSDTM_M(K, j); DSDTM_M(K, j) = DSDTM_M(P, j); DSDTM_M(P, j) = t; }
#pragma unroll
    for (int i = P + 1; i < 6; ++i) { double t = DSDTM_M(i, K); DSDTM_M(i, K) = DSDTM_M(i, P); DSDTM_M(i, P) = t; }
    { double t = DSDTM_M(K, K); DSDTM_M(K, K) = DSDTM_M(P, P); DSDTM_M(P, P) = t; }
#pragma unroll
    for (int i = K + 1; i < P; ++i) { double t = DSDTM_M(i, K); DSDTM_M(i, K) = DSDTM_M(P, i); DSDTM_M(P, i) = t; }
}

template <int K, int P>
__device__ __forceinline__ void ldlt_pivot_case(double* m, double* d, int piv) {
    if constexpr (P < 6) {
        const bool sw = (piv == P);
        if (sw) ldlt_swap<K, P>(m);
        // builds P*b incrementally; written as selects so that d[] keeps static indices (an
        // if-cascade over the pivot is turned into a runtime-indexed scratch array by LLVM)
        const double dk = d[K], dp = d[P];
        d[K] = sw ? dp : dk;
        d[P] = sw ? dk : dp;
    }
}

// One elimination step; returns the chosen pivot row (K itself once the rank cutoff hit).
template <int K>
__device__ __forceinline__ int ldlt_step(double* m, double* d, double* rD, double& cutoff, bool& done) {
    if (done) return K;
    int piv = K;
    double big = fabs(DSDTM_M(K, K));
#pragma unroll
    for (int i = K + 1; i < 6; ++i) {
        const double a = fabs(DSDTM_M(i, i));
        if (a > big) { big = a; piv = i; }
    }
    if (K == 0) cutoff = fabs(2.220446049250313e-16 * big);
    if (big < cutoff) { done = true; return K; }   // not full rank: remaining transpositions are identity
    piv = __builtin_amdgcn_readfirstlane(piv);     // wave-uniform by construction
    ldlt_pivot_case<K, K + 1>(m, d, piv);
    ldlt_pivot_case<K, K + 2>(m, d, piv);
    ldlt_pivot_case<K, K + 3>(m, d, piv);
    ldlt_pivot_case<K, K + 4>(m, d, piv);
    ldlt_pivot_case<K, K + 5>(m, d, piv);
    if (K > 0) {
        double temp[6];
#pragma unroll
        for (int j = 0; j < K; ++j) temp[j] = DSDTM_M(j, j) * DSDTM_M(K, j);
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < K; ++j) s += DSDTM_M(K, j) * temp[j];
        DSDTM_M(K, K) -= s;
#pragma unroll
        for (int i = K + 1; i < 6; ++i) {
            double a = 0.0;
#pragma unroll
            for (int j = 0; j < K; ++j) a += DSDTM_M(i, j) * temp[j];
            DSDTM_M(i, K) -= a;
        }
    }
    // A21 /= D_k as one reciprocal + multiplies; the reciprocal is reused by the D^+ step of the solve
    rD[K] = 1.0 / DSDTM_M(K, K);
    if (K < 5 && fabs(DSDTM_M(K, K)) > cutoff) {
#pragma unroll
        for (int i = K + 1; i < 6; ++i) DSDTM_M(i, K) *= rD[K];
    }
    return piv;
}

template <int K>
__device__ __forceinline__ void ldlt_unswap(double* d, int trk) {
#pragma unroll
    for (int p = K + 1; p < 6; ++p) {
        const bool sw = (trk == p);
        const double dk = d[K], dp = d[p];
        d[K] = sw ? dp : dk;
        d[p] = sw ? dk : dp;
    }
}

// Factorisation of H, kept so that later right-hand sides reuse it: within a pyramid level H only
// changes when the set of visible patches does, so most Gauss-Newton iterations skip ldlt6_factor
// and only run ldlt6_apply on the cached factors (identical factors => identical x).
//   m[21]   packed lower triangle: L below the diagonal, D on it
//   dinv[6] 1/D_i
//   tr0..4  transpositions (pivot rows) of steps 0..4
//   dmask   bit i set when |D_i| > tolerance (the pseudo-inverse keeps that component)
// Plain arrays and scalars (not a struct) so that every element stays in a statically indexed VGPR.
__device__ __forceinline__ void ldlt6_factor(const double* Hu, double* m, double* dinv, int& tr0, int& tr1, int& tr2,
                                             int& tr3, int& tr4, unsigned& dmask) {
    {   // lower triangle from the 21 upper-triangular entries (H is symmetric)
        int q = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = i; j < 6; ++j) { DSDTM_M(j, i) = Hu[q]; ++q; }
    }
    double dummy[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};   // the right-hand side is permuted in ldlt6_apply
    double cutoff = 0.0;
    bool done = false;
#pragma unroll
    for (int i = 0; i < 6; ++i) dinv[i] = 0.0;          // 1/D_k where the factorisation reached step k
    tr0 = ldlt_step<0>(m, dummy, dinv, cutoff, done);
    tr1 = ldlt_step<1>(m, dummy, dinv, cutoff, done);
    tr2 = ldlt_step<2>(m, dummy, dinv, cutoff, done);
    tr3 = ldlt_step<3>(m, dummy, dinv, cutoff, done);
    tr4 = ldlt_step<4>(m, dummy, dinv, cutoff, done);
    (void)ldlt_step<5>(m, dummy, dinv, cutoff, done);
    if (done) {   // rank cutoff hit (rare, wave-uniform): reciprocals of the steps that were skipped
#pragma unroll
        for (int i = 0; i < 6; ++i)
            if (dinv[i] == 0.0) dinv[i] = 1.0 / DSDTM_M(i, i);
    }
    // pseudo-inverse of D (Eigen 3.2 LDLT::solve): components with |D_i| <= tolerance are zeroed
    double maxd = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) maxd = fmax(maxd, fabs(DSDTM_M(i, i)));
    double tol = maxd * 2.220446049250313e-16;
    tol = fmax(tol, 1.0 / 1.7976931348623157e308);
    unsigned mask = 0u;
#pragma unroll
    for (int i = 0; i < 6; ++i)
        if (fabs(DSDTM_M(i, i)) > tol) mask |= 1u << i;
    dmask = mask;
}

// x = H^+ b from the cached factors: dst = P b; L^-1; D^+; L^-T; P^T.
__device__ __forceinline__ void ldlt6_apply(const double* m, const double* dinv, int tr0, int tr1, int tr2, int tr3,
                                            int tr4, unsigned dmask, const double* b, double* x) {
    double d[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) d[i] = b[i];
    // P b: swap d[K] <-> d[tr_K] for K = 0..4 (selects keep d[] statically indexed)
    ldlt_unswap<0>(d, tr0);
    ldlt_unswap<1>(d, tr1);
    ldlt_unswap<2>(d, tr2);
    ldlt_unswap<3>(d, tr3);
    ldlt_unswap<4>(d, tr4);
    // L^-1
#pragma unroll
    for (int i = 1; i < 6; ++i) {
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < i; ++j) s += DSDTM_M(i, j) * d[j];
        d[i] -= s;
    }
    // D^+
#pragma unroll
    for (int i = 0; i < 6; ++i) d[i] = ((dmask >> i) & 1u) ? d[i] * dinv[i] : 0.0;
    // L^-T
#pragma unroll
    for (int i = 4; i >= 0; --i) {
        double s = 0.0;
#pragma unroll
        for (int j = i + 1; j < 6; ++j) s += DSDTM_M(j, i) * d[j];
        d[i] -= s;
    }
    // P^T: the same transpositions in reverse order
    ldlt_unswap<4>(d, tr4);
    ldlt_unswap<3>(d, tr3);
    ldlt_unswap<2>(d, tr2);
    ldlt_unswap<1>(d, tr1);
    ldlt_unswap<0>(d, tr0);
#pragma unroll
    for (int i = 0; i < 6; ++i) x[i] = d[i];
}

// H^-1 of a positive definite 6x6 spread over 36 lanes: lane 6 i + j holds element (i, j), and six in-place
// Gauss-Jordan steps (pivot k: row k scaled by p = 1/a_kk, every other row eliminated, column k replaced) turn H
// into its inverse with ~20 instructions per step per lane — one reciprocal (v_rcp_f64 + two Newton steps), two
// lane permutes per operand half for a_ik and a_kj, three multiply-adds — instead of the ~650 wave-uniform
// instructions of the pivoted LDLT + six substitutions (ldlt6_factor / ldlt6_apply: transposition cascades, six
// IEEE divisions). Elimination without pivoting is backward stable for symmetric positive definite matrices
// (growth factor 1), which is what sum J J^T is whenever it has full rank; the result equals H.ldlt().solve(e_j) to
// cond(H) * eps (measured <= 1e-10 relative on matrices of condition 1e6, tests/test_sparse_align_gpu.py).
// Returns false (wave-uniform) when a pivot is not safely positive — rank-deficient, indefinite by rounding,
// zero or non-finite H: the caller then runs the general pivoted code, which reproduces Eigen's rank cutoff and
// pseudo-inverse. All 64 lanes must call it (lanes >= 36 carry no element).
__device__ __forceinline__ double rcp_f64_newton(double a) {
    double r = __builtin_amdgcn_rcp(a);
    double e = __builtin_fma(-a, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-a, r, 1.0);
    return __builtin_fma(r, e, r);
}
__device__ __forceinline__ double lane_read_f64(double v, int src_lane) {      // ds_bpermute on both halves
    const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2hiint(v));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double bcast_f64(double v, int src_lane) {          // wave-uniform lane index
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src_lane),
                            __builtin_amdgcn_readlane(__double2loint(v), src_lane));
}
__device__ __forceinline__ bool gj6_invert_lanes(double& a, int lane) {
    const int l36 = lane < 36 ? lane : 35;
    const int i = l36 / 6, j = l36 - 6 * i;
    double dmax = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) dmax = fmax(dmax, fabs(bcast_f64(a, 7 * k)));
    // a pivot must stay well above the rounding noise of the entries it was eliminated from
    const double floor_ = dmax * 1e-13;
    bool ok = dmax > 0.0 && dmax < 1.7976931348623157e308;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const double akk = bcast_f64(a, 7 * k);
        ok = ok && (akk > floor_);
        const double p = rcp_f64_newton(akk);
        const double aik = lane_read_f64(a, 6 * i + k);       // same row, pivot column
        const double akj = lane_read_f64(a, 6 * k + j);       // pivot row, same column
        const double upd = a - (aik * p) * akj;
        a = (i == k) ? ((j == k) ? p : a * p) : ((j == k) ? -(a * p) : upd);
    }
    return __builtin_amdgcn_readfirstlane((int)ok) != 0;
}

__device__ inline void ldlt6_solve(const double* Hu, const double* b, double* x) {
    double m[21], dinv[6];
    int tr0, tr1, tr2, tr3, tr4;
    unsigned dmask;
    ldlt6_factor(Hu, m, dinv, tr0, tr1, tr2, tr3, tr4, dmask);
    ldlt6_apply(m, dinv, tr0, tr1, tr2, tr3, tr4, dmask, b, x);
}
#undef DSDTM_M

// ---------------------------------------------------------------------------------------
// Wavefront (64 lanes) sum of a double with DPP moves on the two 32-bit halves:
// row_ror 8/4/2/1 leaves every lane of a 16-lane row holding the row sum, then
// row_bcast15 (rows 1,3) and row_bcast31 (rows 2,3) fold the rows: lane 63 holds the total.
// ---------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_sum_to_lane63(double v) {
    v += dpp_f64<0x128, 0xf>(v);  // row_ror:8
    v += dpp_f64<0x124, 0xf>(v);  // row_ror:4
    v += dpp_f64<0x122, 0xf>(v);  // row_ror:2
    v += dpp_f64<0x121, 0xf>(v);  // row_ror:1
    v += dpp_f64<0x142, 0xa>(v);  // row_bcast:15 -> rows 1 and 3
    v += dpp_f64<0x143, 0xc>(v);  // row_bcast:31 -> rows 2 and 3
    return v;
}

// Wavefront sum of an int, result broadcast to every lane (wave-uniform, returned through an SGPR).
__device__ __forceinline__ int wave_sum_i32(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false);  // row_ror:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x124, 0xf, 0xf, false);  // row_ror:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x122, 0xf, 0xf, false);  // row_ror:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x121, 0xf, 0xf, false);  // row_ror:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31
    return __builtin_amdgcn_readlane(v, 63);
}

// row rotations read every lane of the row, so no "old" value is needed: mov_dpp saves the
// v_mov that update_dpp needs to preload its old operand (one per DPP move)
template <int CTRL>
__device__ __forceinline__ double dpp_rot_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// Sum over the 16 lanes of each DPP row only; every lane of a row ends up holding its row's sum.
__device__ __forceinline__ double row_sum16(double v) {
    v += dpp_rot_f64<0x128>(v);  // row_ror:8
    v += dpp_rot_f64<0x124>(v);  // row_ror:4
    v += dpp_rot_f64<0x122>(v);  // row_ror:2
    v += dpp_rot_f64<0x121>(v);  // row_ror:1
    return v;
}

// EIGHT values summed over the 16 lanes of each DPP row in one butterfly that halves the number of live values
// at every step (a "transpose-reduce"): lanes pair up by lane bit 0, 1, 3 and 2 in turn; at each of the first
// three steps a lane keeps one half of its values, sends the other half to its partner and adds what it
// receives. 8 + 4 + 2 + 1 = 15 additions... per lane 4 + 2 + 1 + 1 = 8 instead of the 8 x 4 = 32 of eight
// separate row_sum16 calls, and 18 DPP moves instead of 64 (FP64 has no DPP form: every exchanged double costs
// two v_mov_dpp). Returns, in EVERY lane, the row total of value index
//     row_reduce8_index(lane) = 4 * bit0 + 2 * bit1 + bit3        (lanes L and L ^ 4 hold the same total)
// Fixed pairing order: deterministic, independent of timing.
template <int CTRL>
__device__ __forceinline__ double dpp_perm_f64(double v) {      // quad_perm / row_ror: every lane has a source lane
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double dpp_xor4_f64(double v) {      // lane L <- lane L ^ 4: row_shl:4 into banks 0,2 and row_shr:4 into banks 1,3
    const int lo = __double2loint(v), hi = __double2hiint(v);
    int l2 = __builtin_amdgcn_update_dpp(0, lo, 0x104, 0xf, 0x5, false);
    l2 = __builtin_amdgcn_update_dpp(l2, lo, 0x114, 0xf, 0xa, false);
    int h2 = __builtin_amdgcn_update_dpp(0, hi, 0x104, 0xf, 0x5, false);
    h2 = __builtin_amdgcn_update_dpp(h2, hi, 0x114, 0xf, 0xa, false);
    return __hiloint2double(h2, l2);
}
__device__ __forceinline__ int row_reduce8_index(int lane) { return ((lane & 1) << 2) | (lane & 2) | ((lane >> 3) & 1); }
__device__ __forceinline__ double row_reduce8(const double* v, int lane) {
    const bool b0 = (lane & 1) != 0, b1 = (lane & 2) != 0, b3 = (lane & 8) != 0;
    double w[4], u[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const double keep = b0 ? v[4 + j] : v[j], send = b0 ? v[j] : v[4 + j];
        w[j] = keep + dpp_perm_f64<0xB1>(send);           // quad_perm [1,0,3,2]: lane ^ 1
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const double keep = b1 ? w[2 + j] : w[j], send = b1 ? w[j] : w[2 + j];
        u[j] = keep + dpp_perm_f64<0x4E>(send);           // quad_perm [2,3,0,1]: lane ^ 2
    }
    const double keep = b3 ? u[1] : u[0], send = b3 ? u[0] : u[1];
    double t = keep + dpp_perm_f64<0x128>(send);          // row_ror:8: lane ^ 8
    t += dpp_xor4_f64(t);
    return t;
}

// reference implementation of the same sum through the LDS crossbar (used by the self-test)
__device__ __forceinline__ double wave_sum_shfl(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

}  // namespace dsdtm
