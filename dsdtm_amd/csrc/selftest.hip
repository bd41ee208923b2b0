// selftest.hip — device-side unit checks of the FP64 building blocks (wave reduction by DPP,
// pivoted LDLT, SE(3) exp/mul) so that tests can compare them one by one with the CPU oracle.
// Not part of the hot path; reached only through dsdtm_debug_selftest().
#include <hip/hip_runtime.h>

#include "device_math.h"
#include "kernels.h"

namespace dsdtm {

// per case: in[0..20] H upper triangle, in[21..26] b, in[27..32] xi  -> out[0..5] x = H^+ b,
// out[6..12] exp(xi) (qw,qx,qy,qz,tx,ty,tz), out[13..19] exp(xi)*exp(b) , out[20] DPP wave sum of
// (lane+1)*in[27], out[21] shuffle wave sum of the same, out[22..33] [R|t] of exp(xi) via from_rt(to_rt),
// out[34..69] H^+ by columns from ldlt6_solve on the unit vectors, out[70..105] the same from gj6_invert_lanes
// (untouched where it declines), out[106] 1 where it accepted the matrix, out[107..118] dR (9) and dt (3) of
// se3_exp_matrix_small(xi) (only for |omega|^2 < 0.01), out[119] max |row_reduce8 - row_sum16| over 8 values of all lanes
constexpr int SELFTEST_OUT = 120;
__global__ void selftest_kernel(const double* __restrict__ in, double* __restrict__ out, int n_cases) {
    const int c = blockIdx.x;
    if (c >= n_cases) return;
    const int lane = threadIdx.x;
    const double* p = in + (size_t)c * 33;
    double* o = out + (size_t)c * SELFTEST_OUT;
    __shared__ double s_h[21], s_hinv[36];
    double H[21], b[6], xi[6], x[6];
    for (int i = 0; i < 21; ++i) H[i] = p[i];
    for (int i = 0; i < 6; ++i) { b[i] = p[21 + i]; xi[i] = p[27 + i]; }
    ldlt6_solve(H, b, x);
    const SE3d E = se3_exp(xi);
    const SE3d E2 = se3_mul(E, se3_exp(b));
    const double term = (double)(lane + 1) * p[27];
    const double s_dpp = wave_sum_to_lane63(term);
    const double s_shf = wave_sum_shfl(term);
    double R[9];
    quat_to_matrix(E, R);
    double T[12] = {R[0], R[1], R[2], E.tx, R[3], R[4], R[5], E.ty, R[6], R[7], R[8], E.tz};
    const SE3d Eb = se3_from_rt(T);
    double R2[9];
    quat_to_matrix(Eb, R2);
    // H^+ twice: the general pivoted code on the unit vectors, and the sorted-diagonal fast path
    if (lane < 21) s_h[lane] = p[lane];
    if (lane < 36) s_hinv[lane] = -12345.0;
    __syncthreads();
    bool fast_ok;
    {
        const int l36 = lane < 36 ? lane : 35;
        const int gi = l36 / 6, gj = l36 - 6 * gi;
        const int glo = gi < gj ? gi : gj, ghi = gi < gj ? gj : gi;
        double el = s_h[glo * 6 - (glo * (glo - 1)) / 2 + (ghi - glo)];
        fast_ok = gj6_invert_lanes(el, lane);
        if (fast_ok && lane < 36) s_hinv[gj * 6 + gi] = el;
    }
    __syncthreads();
    double dRm[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, dtm[3] = {0, 0, 0};
    const double th2 = xi[3] * xi[3] + xi[4] * xi[4] + xi[5] * xi[5];
    if (th2 < 0.01) se3_exp_matrix_small(xi, th2, dRm, dtm);
    if (lane < 6) {
        double e[6], colv[6];
        for (int i = 0; i < 6; ++i) e[i] = (i == lane) ? 1.0 : 0.0;
        ldlt6_solve(H, e, colv);
        for (int i = 0; i < 6; ++i) o[34 + lane * 6 + i] = colv[i];
    }
    if (lane < 36) o[70 + lane] = s_hinv[lane];
    // row_reduce8 against eight row_sum16 calls on lane-dependent values
    double rr_err;
    {
        double v8[8], want = 0.0;
        const int idx = row_reduce8_index(lane);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            v8[q] = p[q] * (double)(lane + 1) + p[21 + (q % 6)] * (double)(q + 1);
            const double s16 = row_sum16(v8[q]);
            want = (q == idx) ? s16 : want;
        }
        const double got = row_reduce8(v8, lane);
        rr_err = fabs(got - want) / fmax(1e-300, fabs(want));
#pragma unroll
        for (int o_ = 32; o_ > 0; o_ >>= 1) rr_err = fmax(rr_err, __shfl_xor(rr_err, o_, 64));
    }
    if (lane == 63) {
        o[106] = fast_ok ? 1.0 : 0.0;
        for (int i = 0; i < 9; ++i) o[107 + i] = dRm[i];
        for (int i = 0; i < 3; ++i) o[116 + i] = dtm[i];
        o[119] = rr_err;
        for (int i = 0; i < 6; ++i) o[i] = x[i];
        o[6] = E.qw; o[7] = E.qx; o[8] = E.qy; o[9] = E.qz; o[10] = E.tx; o[11] = E.ty; o[12] = E.tz;
        o[13] = E2.qw; o[14] = E2.qx; o[15] = E2.qy; o[16] = E2.qz; o[17] = E2.tx; o[18] = E2.ty; o[19] = E2.tz;
        o[20] = s_dpp;
        o[21] = s_shf;
        o[22] = R2[0]; o[23] = R2[1]; o[24] = R2[2]; o[25] = Eb.tx;
        o[26] = R2[3]; o[27] = R2[4]; o[28] = R2[5]; o[29] = Eb.ty;
        o[30] = R2[6]; o[31] = R2[7]; o[32] = R2[8]; o[33] = Eb.tz;
    }
}

hipError_t selftest_launch(const double* in, double* out, int n_cases, hipStream_t stream) {
    if (n_cases <= 0) return hipSuccess;
    hipLaunchKernelGGL(selftest_kernel, dim3((unsigned)n_cases), dim3(64), 0, stream, in, out, n_cases);
    return hipGetLastError();
}

}  // namespace dsdtm
