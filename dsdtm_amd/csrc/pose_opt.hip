// pose_opt.hip — Optimizer::PoseOptimization (reference src/Optimizer.cpp:20-101): pose-only refinement
// of one frame against its map-point observations. SURVEY.md §8(f)3.
//
// What the reference runs is a Ceres problem with ONE free 6-parameter block [t, log R] (the map
// points are set constant, :60-61), FullBA_Problem residuals (include/Optimizer.h:129-216),
// CauchyLoss(1.0), PoseLocalParameterization (include/Optimizer.h:219-252) and the default
// trust-region Levenberg-Marquardt minimiser, at most 100 iterations (:68-72). The minimiser is
// restated from Ceres 1.13 (DESIGN.md §3.6 lists what is the reference's and what is Ceres'); the damped
// step is taken from the 6x6 normal equations instead of a QR of the stacked Jacobian.
//
// Mapping: one wavefront per frame (problem). Lane l evaluates features l, l+64, ... — residual,
// 2x6 Jacobian, Cauchy weight — and accumulates its share of J^T J (21), J^T r (6) and the cost in
// VGPRs; one DPP reduction per value per evaluation (fixed order: deterministic). The trust-region
// logic, the 6x6 Cholesky and the SE(3) algebra are wave-uniform and run redundantly on all lanes
// (the cost of one lane). Every candidate is evaluated WITH its Jacobian, so an accepted step — the
// common case — needs no second pass over the features. Nothing is staged: a frame's feature columns
// (<= ~60 B per feature) stay in L1/L2 across the <= 100 iterations.
// Roofline: the kernel is FP64-latency bound on one wave per problem (five divisions per feature and
// evaluation); throughput comes from problems in flight, not from bandwidth.
#include <hip/hip_runtime.h>
#include <float.h>
#include <stdlib.h>

#include "device_math.h"
#include "kernels.h"

namespace dsdtm {

namespace {

// The libm calls of this kernel (sincos, atan, log) sit in functions that are NOT inlined: inlined into the
// iteration loop, their ~60 polynomial coefficients are hoisted out of it as loop invariants and held in
// registers for the whole kernel (296 VGPRs, one wave per SIMD); out of line the kernel needs 2/3 of that.
// Scalar in, scalar out: arguments and results travel in registers, nothing goes through the stack.
// (The few-frames instantiation — one frame of the live tracker, one wave per SIMD — has the whole register file
// to itself: there LAT = true inlines them, which saves the call sequences on the iteration's dependent chain.)
__device__ __attribute__((noinline)) double log_out_of_line(double v) { return log(v); }
__device__ __attribute__((noinline)) double atan_out_of_line(double v) { return atan(v); }
struct SinCos { double s, c; };
__device__ __attribute__((noinline)) SinCos sincos_out_of_line(double v) {
    SinCos r;
    sincos(v, &r.s, &r.c);
    return r;
}
template <bool LAT> __device__ __forceinline__ double po_log(double v) { if constexpr (LAT) return log(v); else return log_out_of_line(v); }
template <bool LAT> __device__ __forceinline__ double po_atan(double v) { if constexpr (LAT) return atan(v); else return atan_out_of_line(v); }
template <bool LAT> __device__ __forceinline__ SinCos po_sincos(double v) {
    if constexpr (LAT) { SinCos r; sincos(v, &r.s, &r.c); return r; } else return sincos_out_of_line(v);
}
// x = [t, w] -> SE3(SO3::exp(w), t) (include/Optimizer.h:147). The quaternion part of se3_exp
// (device_math.h) on its own: same series / closed forms, same normalisation, no V matrix.
template <bool LAT>
__device__ __forceinline__ SE3d pose_of(const double* x) {
    const double wx = x[3], wy = x[4], wz = x[5];
    const double theta_sq = wx * wx + wy * wy + wz * wz;
    double ch, imag_factor;
    if (theta_sq < 0.01) {
        const double h2 = 0.25 * theta_sq;
        ch = 1.0 + h2 * (-1.0 / 2 + h2 * (1.0 / 24 + h2 * (-1.0 / 720 + h2 * (1.0 / 40320 + h2 * (-1.0 / 3628800)))));
        const double sinc = 1.0 + h2 * (-1.0 / 6 + h2 * (1.0 / 120 + h2 * (-1.0 / 5040 + h2 * (1.0 / 362880 +
                            h2 * (-1.0 / 39916800)))));
        imag_factor = 0.5 * sinc;
        if (theta_sq < 1e-20) {          // Sophus: theta < SMALL_EPS
            const double theta_po4 = theta_sq * theta_sq;
            imag_factor = 0.5 - 0.0208333 * theta_sq + 0.000260417 * theta_po4;
        }
    } else {
        const double theta = sqrt(theta_sq);
        const SinCos sc = po_sincos<LAT>(0.5 * theta);
        ch = sc.c;
        imag_factor = sc.s * (1.0 / theta);
    }
    SE3d T;
    T.qw = ch; T.qx = imag_factor * wx; T.qy = imag_factor * wy; T.qz = imag_factor * wz;
    quat_normalize(T);
    T.tx = x[0]; T.ty = x[1]; T.tz = x[2];
    return T;
}

// Sophus SO3::log (atan form) of a unit quaternion
template <bool LAT>
__device__ __forceinline__ void so3_log(const SE3d& q, double* w) {
    const double n = sqrt(q.qx * q.qx + q.qy * q.qy + q.qz * q.qz);
    const double qw = q.qw;
    double f;
    if (n < 1e-10) f = 2. / qw - 2. * (n * n) / (qw * (qw * qw));
    else f = 2 * po_atan<LAT>(n / qw) / n;
    w[0] = f * q.qx; w[1] = f * q.qy; w[2] = f * q.qz;
}

// PoseLocalParameterization::Plus (include/Optimizer.h:222-236)
template <bool LAT>
__device__ __forceinline__ void pose_plus(const SE3d& To /* pose_of(x) */, const double* d, double* out) {
    const SE3d Td = pose_of<LAT>(d);
    const SE3d Tn = se3_mul(Td, To);
    out[0] = Tn.tx; out[1] = Tn.ty; out[2] = Tn.tz;
    so3_log<LAT>(Tn, out + 3);
}


__device__ __forceinline__ bool finite6(const double* v) {
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 6; ++i) ok = ok && isfinite(v[i]);
    return ok;
}

struct Frame {
    const double* bearing;
    const double* pw;
    const int32_t* level;
    const uint8_t* use;
    int n;
};

// FullBA_Problem::Evaluate of one feature at pose T (include/Optimizer.h:139-197); false when unused
__device__ __forceinline__ bool block_residual(const Frame& f, int i, const SE3d& T, double& r0, double& r1,
                                               double& px, double& py, double& pz) {
    if (i >= f.n || !f.use[i]) return false;
    const double b0 = f.bearing[3 * (size_t)i], b1 = f.bearing[3 * (size_t)i + 1], b2 = f.bearing[3 * (size_t)i + 2];
    const double X = f.pw[3 * (size_t)i], Y = f.pw[3 * (size_t)i + 1], Z = f.pw[3 * (size_t)i + 2];
    double rx, ry, rz;
    quat_rotate(T, X, Y, Z, rx, ry, rz);
    px = rx + T.tx; py = ry + T.ty; pz = rz + T.tz;
    const double inv_scale = __hiloint2double((1023 - (f.level[i] & 31)) << 20, 0);   // 1 / (1 << level), exact
    r0 = (b0 / b2 - px / pz) * inv_scale;
    r1 = (b1 / b2 - py / pz) * inv_scale;
    return true;
}

// A lane's feature held in registers (few-frames instantiation with at most one feature per lane): the columns are
// read from memory ONCE — so the host entry point can hand them over in pinned host memory, no copy — and the two
// quotients that do not depend on the pose are formed once (the same IEEE divisions, the same bits).
struct FeatRegs {
    bool use;
    double bx, by;          // b0 / b2, b1 / b2
    double X, Y, Z;
    double inv_scale;
};
__device__ __forceinline__ FeatRegs load_feature(const Frame& f, int i) {
    FeatRegs r;
    r.use = i < f.n && f.use[i];
    r.bx = r.by = r.X = r.Y = r.Z = 0.0; r.inv_scale = 1.0;
    if (r.use) {
        const double b0 = f.bearing[3 * (size_t)i], b1 = f.bearing[3 * (size_t)i + 1], b2 = f.bearing[3 * (size_t)i + 2];
        r.bx = b0 / b2; r.by = b1 / b2;
        r.X = f.pw[3 * (size_t)i]; r.Y = f.pw[3 * (size_t)i + 1]; r.Z = f.pw[3 * (size_t)i + 2];
        r.inv_scale = __hiloint2double((1023 - (f.level[i] & 31)) << 20, 0);
    }
    return r;
}
__device__ __forceinline__ void block_residual_regs(const FeatRegs& c, const SE3d& T, double& r0, double& r1,
                                                    double& px, double& py, double& pz) {
    double rx, ry, rz;
    quat_rotate(T, c.X, c.Y, c.Z, rx, ry, rz);
    px = rx + T.tx; py = ry + T.ty; pz = rz + T.tz;
    r0 = (c.bx - px / pz) * c.inv_scale;
    r1 = (c.by - py / pz) * c.inv_scale;
}

// Program evaluation at pose T: cost = sum 1/2 rho(|r|^2) (wave-uniform); H = J^T J (upper triangle,
// row-major) and g = J^T r of the loss-corrected blocks go to hg[0..20], hg[21..26] in LDS (the solver part
// reads what it needs, when it needs it; the 27 totals are not live in registers across the next
// evaluation). ok = false when anything was not finite.
#define PO_PART_STRIDE 29
// NW waves per frame: one for batches (the wave-uniform solver part is executed once per frame), four for a
// few frames (the live tracker's single frame: the features spread over 256 lanes, every wave repeats the
// solver part on the same totals and so takes the same decisions — no broadcast).
// GPW frames per wavefront (batches, round 6): 1, or 4 — a frame then owns a 16-lane ROW of the wave: its features stride over 16
// lanes (four times the evaluations per lane), its totals are summed over the row, and the solver part below — two thirds of an
// iteration's instructions, executed redundantly by all 64 lanes for ONE frame before — runs for four frames at once, each
// row uniform in itself. Rows that take different branches (a rejected step, a frame that has converged) are serialised by
// the hardware like any divergent code; nothing crosses a row.
template <int NW, int FPL, int GPW = 1>      // FPL: features per lane held in registers (0: read from memory at every evaluation)
__device__ void evaluate(const Frame& f, const FeatRegs* fr, int tid, const SE3d& T /* pose_of(x) */, double& cost, double* hg,
                         double* part, double* red, bool& ok) {
    constexpr bool LAT = NW > 1;
    constexpr bool CACHED = FPL > 0;
    constexpr int LPF = 64 / GPW;                           // lanes per frame
    static_assert(GPW == 1 || (NW == 1 && FPL == 0), "several frames per wave: one wave per workgroup, features from memory");
    const int lane = tid & 63, wave = tid >> 6;
    double c = 0.0, h[21], gg[6];
    bool bad = false;
#pragma unroll
    for (int k = 0; k < 21; ++k) h[k] = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) gg[k] = 0.0;
#pragma unroll
    for (int i = (GPW == 1 ? tid : (lane & (LPF - 1))), j = 0; CACHED ? j < FPL : i < f.n; i += (GPW == 1 ? NW * 64 : LPF), ++j) {
        double r0, r1, px, py, pz;
        if constexpr (CACHED) {
            if (!fr[j].use) continue;
            block_residual_regs(fr[j], T, r0, r1, px, py, pz);
        } else {
            if (!block_residual(f, i, T, r0, r1, px, py, pz)) continue;
        }
        const double z_inv = 1.0 / pz;
        const double z_inv2 = z_inv * z_inv;
        double J0[6], J1[6];
        J0[0] = -z_inv; J0[1] = 0.0; J0[2] = px * z_inv2; J0[3] = py * J0[2]; J0[4] = -(1.0 + px * J0[2]); J0[5] = py * z_inv;
        J1[0] = 0.0; J1[1] = -z_inv; J1[2] = py * z_inv2; J1[3] = 1.0 + py * J1[2]; J1[4] = -px * J1[2]; J1[5] = -px * z_inv;
        bad = bad || !(isfinite(r0) && isfinite(r1) && finite6(J0) && finite6(J1));
        const double s = r0 * r0 + r1 * r1;
        const double sum = 1.0 + s;                         // CauchyLoss(1.0): b = c = 1
        const double inv = 1.0 / sum;
        c += 0.5 * po_log<LAT>(sum);
        const double sq = sqrt(fmax(inv, DBL_MIN));         // Corrector, alpha = 0 (rho'' < 0)
#pragma unroll
        for (int k = 0; k < 6; ++k) { J0[k] *= sq; J1[k] *= sq; }
        r0 *= sq; r1 *= sq;
        int q = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            gg[a] += J0[a] * r0 + J1[a] * r1;
#pragma unroll
            for (int b = a; b < 6; ++b) { h[q] += J0[a] * J0[b] + J1[a] * J1[b]; ++q; }
        }
    }
    // Cross-lane totals through LDS: every lane parks its 28 partials (row stride 29 doubles: the column
    // reads below are conflict-free); in every wave lane k (k < 28) adds column k over the wave's lanes 0..31
    // and lane 32 + k over its lanes 32..63, in lane order; thread k < 28 then folds the 2 NW half sums in
    // fixed order. The order is fixed, so runs repeat bit for bit.
    double* row = part + tid * PO_PART_STRIDE;
#pragma unroll
    for (int k = 0; k < 21; ++k) row[k] = h[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) row[21 + k] = gg[k];
    row[27] = c;
    if constexpr (GPW > 1) {
        // the row's 28 totals: lane gl of the row adds columns gl and gl + 16 over the row's 16 lanes, in lane order
        const int g = lane / LPF, gl = lane & (LPF - 1);
        const unsigned long long bad_row = (__ballot(bad) >> (g * LPF)) & ((1ull << LPF) - 1ull);
        __syncthreads();                                    // (one wave per workgroup: orders the LDS writes and reads)
        for (int col = gl; col < 28; col += LPF) {
            double acc = 0.0;
            const double* src = part + (size_t)(g * LPF) * PO_PART_STRIDE + col;
#pragma unroll
            for (int l = 0; l < LPF; ++l) acc += src[l * PO_PART_STRIDE];
            hg[col] = acc;                                  // hg: this row's totals
        }
        __syncthreads();
        cost = hg[27];
        ok = bad_row == 0ull && isfinite(cost);
        (void)red; (void)wave;
        return;
    }
    const unsigned long long bad_wave = __ballot(bad);
    __syncthreads();                                        // orders LDS writes and reads
    const int col = lane & 31, half = lane >> 5;
    if (col < 28) {
        double acc = 0.0;
        const double* src = part + (wave * 64 + half * 32) * PO_PART_STRIDE + col;
#pragma unroll 8
        for (int l = 0; l < 32; ++l) acc += src[l * PO_PART_STRIDE];
        red[(wave * 2 + half) * 32 + col] = acc;
    }
    if (lane == 0) red[(2 * NW) * 32 + wave] = bad_wave ? 1.0 : 0.0;
    __syncthreads();
    if (tid < 28) {
        double acc = red[tid];
#pragma unroll
        for (int k = 1; k < 2 * NW; ++k) acc += red[k * 32 + tid];
        hg[tid] = acc;                                      // hg[0..20] H, hg[21..26] g, hg[27] cost
    }
    __syncthreads();
    cost = hg[27];
    bool any_bad = false;
#pragma unroll
    for (int w = 0; w < NW; ++w) any_bad = any_bad || red[(2 * NW) * 32 + w] != 0.0;
    ok = !any_bad && isfinite(cost);
    __syncthreads();                                        // red / part are free again
}

#define PO_U(i, j) ((i) * 6 - ((i) * ((i) - 1)) / 2 + ((j) - (i)))   /* packed upper index, i <= j */

// Cholesky solve of the SPD system M y = v; M as packed upper triangle. false on a non-positive pivot.
// One reciprocal per pivot, reused by the column scaling and both substitutions (6 divisions instead of 27
// on the iteration's dependent chain).
__device__ __forceinline__ bool chol6_solve(const double* M, const double* v, double* y) {
    double L[21];   // lower factor, packed by rows: L(i,j) at i(i+1)/2 + j
    double rl[6];   // 1 / L(j,j)
#define PO_L(i, j) L[((i) * ((i) + 1)) / 2 + (j)]
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double d = M[PO_U(j, j)];
#pragma unroll
        for (int k = 0; k < j; ++k) d -= PO_L(j, k) * PO_L(j, k);
        ok = ok && (d > 0.0);
        const double ljj = sqrt(d);
        PO_L(j, j) = ljj;
        rl[j] = 1.0 / ljj;
#pragma unroll
        for (int i = j + 1; i < 6; ++i) {
            double s = M[PO_U(j, i)];
#pragma unroll
            for (int k = 0; k < j; ++k) s -= PO_L(i, k) * PO_L(j, k);
            PO_L(i, j) = s * rl[j];
        }
    }
    double z[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double s = v[i];
#pragma unroll
        for (int k = 0; k < i; ++k) s -= PO_L(i, k) * z[k];
        z[i] = s * rl[i];
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double s = z[i];
#pragma unroll
        for (int k = i + 1; k < 6; ++k) s -= PO_L(k, i) * y[k];
        y[i] = s * rl[i];
    }
#undef PO_L
    return ok;
}

__device__ __forceinline__ double norm6(const double* v) {
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) s += v[i] * v[i];
    return sqrt(s);
}

// max |x - Plus(x, -g)|, used by exactly one test: gmax <= 1e-10. To first order x - Plus(x, -g) is g itself (the
// rotation part through a left Jacobian whose singular values stay within [1/2, 2] for |w| < pi), so a gradient with
// an entry above 1e-6 cannot pass that test and the exact value is not needed: the few-frames instantiation then
// returns the gradient's own max norm and saves one Plus (exp, product, log with its atan) per accepted step.
template <bool LAT>
__device__ __forceinline__ double gradient_max_norm(const SE3d& Tx, const double* x, const double* hg) {
    double ng[6], xp[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) ng[k] = -hg[21 + k];
    if constexpr (LAT) {
        double m = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) m = fmax(m, fabs(ng[k]));
        if (m > 1e-6) return m;
    }
    pose_plus<LAT>(Tx, ng, xp);
    double m = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) m = fmax(m, fabs(x[k] - xp[k]));
    return m;
}

}  // namespace

constexpr int PO_WAVES_PER_EU = 2;
// The whole refinement of one frame by NW waves (FPL > 0: a lane's features in registers). A device function so that ONE
// kernel can carry two instantiations and pick by the frame's live feature count (pose_opt_auto_kernel below).
template <int NW, int FPL, int GPW = 1>
__device__ __forceinline__ void pose_opt_body(const PoseOptArgs& a) {
    constexpr bool LAT = NW > 1;
    constexpr bool CACHED = FPL > 0;
    constexpr int LPF = 64 / GPW;                           // lanes per frame (GPW frames per wavefront, see evaluate)
    constexpr unsigned long long ROWMASK = GPW == 1 ? ~0ull : ((1ull << (LPF & 63)) - 1ull);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int grp = GPW == 1 ? 0 : lane / LPF, gl = lane & (LPF - 1);
    const int frame = GPW == 1 ? (int)blockIdx.x : (int)blockIdx.x * GPW + grp;
    if (frame >= a.n_frames) return;
    const size_t base = (size_t)frame * a.max_features;
    Frame f;
    f.bearing = a.bearing + base * 3;
    f.pw = a.p_world + base * 3;
    f.level = a.level + base;
    f.use = a.use + base;
    f.n = a.n_features ? a.n_features[frame] : a.max_features;
    if (f.n > a.max_features) f.n = a.max_features;
    if (a.only_hi > 0 && !(f.n > a.only_lo && f.n <= a.only_hi)) return;     // dsdtm_track_frame: another instantiation takes this count
    double* Tio = a.T_cur_w + (size_t)frame * 12;

    // residual blocks (src/Optimizer.cpp:45-65)
    int n_blocks = 0;
    if constexpr (!CACHED) {
        for (int b0 = 0; b0 < f.n; b0 += LPF) {
            const int i = b0 + gl;
            const unsigned long long m = __ballot(i < f.n && f.use[i]);
            n_blocks += __popcll((m >> (grp * LPF)) & ROWMASK);
        }
    }

    // parameter block (src/Optimizer.cpp:35-37)
    const SE3d T0 = se3_from_rt(Tio);
    double x[6] = {T0.tx, T0.ty, T0.tz, 0.0, 0.0, 0.0};
    so3_log<LAT>(T0, x + 3);
    SE3d Tx = pose_of<LAT>(x);
    FeatRegs fr[CACHED ? FPL : 1];     // CACHED: this lane's features tid, tid + NW*64, .. (the launcher guarantees n <= FPL * NW * 64)
    if constexpr (CACHED) {
#pragma unroll
        for (int j = 0; j < FPL; ++j) fr[j] = load_feature(f, tid + j * NW * 64);
    } else fr[0].use = false;              // the pose every evaluation of x uses; recomputed from x only when x changes

    int termination = DSDTM_PO_MAX_ITERATIONS, iterations = 0, successful = 0;
    double x_cost = 0.0, initial_cost = 0.0;
    __shared__ double s_hg_all[GPW][2][28];     // per frame of the wave: H (21), g (6), cost of the accepted point [cur] and of the candidate [cur ^ 1]
    double (*s_hg)[28] = s_hg_all[grp];
    __shared__ double s_part[NW * 64 * PO_PART_STRIDE];   // per-lane partials of one evaluation
    __shared__ double s_red[(2 * NW + 1) * 32];           // half-wave sums, then the waves' not-finite flags
    if constexpr (CACHED) {                                // residual blocks = used features over the NW waves
        int mine = 0;
#pragma unroll
        for (int j = 0; j < FPL; ++j) mine += __popcll(__ballot(fr[j].use));
        if (lane == 0) s_red[tid >> 6] = (double)mine;
        __syncthreads();
#pragma unroll
        for (int w = 0; w < NW; ++w) n_blocks += (int)s_red[w];
        __syncthreads();
    }
    int cur = 0;
    bool ok;
    if (n_blocks == 0) {
        termination = DSDTM_PO_NO_RESIDUALS;
    } else {
        evaluate<NW, FPL, GPW>(f, fr, tid, Tx, x_cost, s_hg[0], s_part, s_red, ok);
        if (!ok) {
            termination = DSDTM_PO_EVALUATION_FAILED;
            x_cost = 0.0;
        } else {
            initial_cost = x_cost;
            double scale[6], diagonal[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) scale[k] = 1.0 / (1.0 + sqrt(s_hg[0][PO_U(k, k)]));   // Jacobi scaling, fixed
            double x_norm = norm6(x);
            double gmax = gradient_max_norm<LAT>(Tx, x, s_hg[0]);
            double radius = 1e4, decrease_factor = 2.0;
            bool reuse_diagonal = false;
            int invalid_steps = 0, it = 0;
            for (;;) {
                if (it >= a.max_iterations) { termination = DSDTM_PO_MAX_ITERATIONS; break; }
                if (gmax <= 1e-10) { termination = DSDTM_PO_GRADIENT_TOLERANCE; break; }
                if (radius <= 1e-32) { termination = DSDTM_PO_MIN_RADIUS; break; }
                ++it;
                // scaled system: As = S H S, gs = S g
                double As[21], gs[6], M[21];
                const double* hg = s_hg[cur];
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    gs[i] = scale[i] * hg[21 + i];
#pragma unroll
                    for (int j = i; j < 6; ++j) As[PO_U(i, j)] = (scale[i] * hg[PO_U(i, j)]) * scale[j];
                }
                if (!reuse_diagonal) {
#pragma unroll
                    for (int k = 0; k < 6; ++k) diagonal[k] = fmin(fmax(As[PO_U(k, k)], 1e-6), 1e32);
                }
#pragma unroll
                for (int k = 0; k < 21; ++k) M[k] = As[k];
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const double lm = sqrt(diagonal[k] / radius);
                    M[PO_U(k, k)] += lm * lm;
                }
                double y[6], step[6];
                bool solved = chol6_solve(M, gs, y);
                solved = solved && finite6(y);
#pragma unroll
                for (int k = 0; k < 6; ++k) step[k] = -y[k];
                double lin = 0.0, quad = 0.0;
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    lin += gs[i] * step[i];
                    double s = 0.0;
#pragma unroll
                    for (int j = 0; j < 6; ++j) s += As[i <= j ? PO_U(i, j) : PO_U(j, i)] * step[j];
                    quad += step[i] * s;
                }
                const double model_cost_change = -(lin + quad / 2.0);
                reuse_diagonal = true;
                if (!(solved && model_cost_change > 0.0)) {
                    if (++invalid_steps >= 5) { termination = DSDTM_PO_INVALID_STEPS; break; }
                    radius = radius / decrease_factor;
                    decrease_factor *= 2.0;
                    continue;
                }
                invalid_steps = 0;
                double delta[6], cand[6], cand_cost;
#pragma unroll
                for (int k = 0; k < 6; ++k) delta[k] = step[k] * scale[k];
                pose_plus<LAT>(Tx, delta, cand);
                bool cand_ok = finite6(cand);
                const SE3d Tc = pose_of<LAT>(cand);
                if (cand_ok) evaluate<NW, FPL, GPW>(f, fr, tid, Tc, cand_cost, s_hg[cur ^ 1], s_part, s_red, cand_ok);
                if (!cand_ok) cand_cost = DBL_MAX;
                double diff[6];
#pragma unroll
                for (int k = 0; k < 6; ++k) diff[k] = x[k] - cand[k];
                if (norm6(diff) <= 1e-8 * (x_norm + 1e-8)) { termination = DSDTM_PO_PARAMETER_TOLERANCE; break; }
                const double cost_change = x_cost - cand_cost;
                if (fabs(cost_change) <= 1e-6 * x_cost) { termination = DSDTM_PO_FUNCTION_TOLERANCE; break; }
                const double relative_decrease = cost_change / model_cost_change;
                if (relative_decrease > 1e-3) {
#pragma unroll
                    for (int k = 0; k < 6; ++k) x[k] = cand[k];
                    cur ^= 1;                                  // the candidate's H, g become the accepted ones
                    x_cost = cand_cost;
                    x_norm = norm6(x);
                    Tx = Tc;
                    gmax = gradient_max_norm<LAT>(Tx, x, s_hg[cur]);
                    ++successful;
                    const double t = 2.0 * relative_decrease - 1.0;
                    radius = radius / fmax(1.0 / 3.0, 1.0 - t * t * t);
                    radius = fmin(1e16, radius);
                    decrease_factor = 2.0;
                    reuse_diagonal = false;
                } else {
                    radius = radius / decrease_factor;
                    decrease_factor *= 2.0;
                    reuse_diagonal = true;
                }
            }
            iterations = it;
        }
    }

    // Set_Pose(SE3(SO3::exp(x.tail<3>()), x.head<3>())) (src/Optimizer.cpp:78)
    const SE3d Tf = Tx;
    double R[9];
    quat_to_matrix(Tf, R);
    if constexpr (CACHED) {
        // GetReprojectReidual (src/Optimizer.cpp:297-317) from the features in registers, in residual-block order
        // (= feature order): round j holds features j*NW*64 + tid; a wave's lanes write behind the blocks of the
        // rounds and waves before them
        const int wave = tid >> 6;
        unsigned long long m[FPL];
#pragma unroll
        for (int j = 0; j < FPL; ++j) {
            m[j] = __ballot(fr[j].use);
            if (lane == 0) s_red[j * NW + wave] = (double)__popcll(m[j]);
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < FPL; ++j) {
            int before = 0;
            for (int k = 0; k < j * NW + wave; ++k) before += (int)s_red[k];
            if (fr[j].use) {
                double r0, r1, px, py, pz;
                block_residual_regs(fr[j], Tf, r0, r1, px, py, pz);
                (a.residual_norm + base)[before + __popcll(m[j] & ((1ull << lane) - 1ull))] = sqrt(r0 * r0 + r1 * r1);
            }
        }
    }
    if (tid >= 64) return;                                  // results: the first wave (all waves hold the same state)
    if (gl == 0) {
        Tio[0] = R[0]; Tio[1] = R[1]; Tio[2] = R[2];  Tio[3] = Tf.tx;
        Tio[4] = R[3]; Tio[5] = R[4]; Tio[6] = R[5];  Tio[7] = Tf.ty;
        Tio[8] = R[6]; Tio[9] = R[7]; Tio[10] = R[8]; Tio[11] = Tf.tz;
        if (a.T_mirror) {
            double* Tm = a.T_mirror + (size_t)frame * 12;
            Tm[0] = R[0]; Tm[1] = R[1]; Tm[2] = R[2];  Tm[3] = Tf.tx;
            Tm[4] = R[3]; Tm[5] = R[4]; Tm[6] = R[5];  Tm[7] = Tf.ty;
            Tm[8] = R[6]; Tm[9] = R[7]; Tm[10] = R[8]; Tm[11] = Tf.tz;
        }
        dsdtm_pose_opt_summary& sm = a.summary[frame];
        sm.iterations = iterations;
        sm.successful_steps = successful;
        sm.termination = termination;
        sm.n_residual_blocks = n_blocks;
        sm.initial_cost = initial_cost;
        sm.final_cost = x_cost;
#pragma unroll
        for (int k = 0; k < 6; ++k) sm.x[k] = x[k];
    }
    if constexpr (!CACHED) {
        // GetReprojectReidual (src/Optimizer.cpp:297-317): raw residual norms, in residual-block order
        double* rn = a.residual_norm + base;
        int done = 0;
        for (int b0 = 0; b0 < f.n; b0 += LPF) {
            const int i = b0 + gl;
            double r0 = 0.0, r1 = 0.0, px, py, pz;
            const bool u = block_residual(f, i, Tf, r0, r1, px, py, pz);
            const unsigned long long mw = __ballot(u);
            const unsigned long long m = (mw >> (grp * LPF)) & ROWMASK;
            if (u) rn[done + __popcll(m & ((1ull << gl) - 1ull))] = sqrt(r0 * r0 + r1 * r1);
            done += __popcll(m);
        }
    }
}

template <int NW, int FPL>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(NW == 1 ? PO_WAVES_PER_EU : 1, NW == 1 ? PO_WAVES_PER_EU : 2)))
void pose_opt_kernel(PoseOptArgs a) { pose_opt_body<NW, FPL>(a); }

// Batches: GPW frames per wavefront (16 or 32 lanes each), one wave per workgroup
template <int GPW>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(PO_WAVES_PER_EU, PO_WAVES_PER_EU)))
void pose_opt_rows_kernel(PoseOptArgs a) { pose_opt_body<1, 0, GPW>(a); }

// dsdtm_track_frame: ONE frame whose feature count (<= 256) is only known on the device. dsdtm_pose_optimization picks the
// instantiation on the host — one wave up to 64 features, four waves with the features in registers up to 256 — and the
// summation order, hence the last bits of the result, follows that choice; here the same choice is made by the kernel, so
// that the one-call frame gives the four-call chain's pose bit for bit (a second launch that only reads the count and
// returns cost 4.7 us of a 200-us frame).
__global__ __launch_bounds__(256) void pose_opt_auto_kernel(PoseOptArgs a) {
    const int n = a.n_features ? a.n_features[blockIdx.x] : a.max_features;
    if (n <= 64) {
        if (threadIdx.x >= 64) return;
        pose_opt_body<1, 0>(a);
    } else pose_opt_body<4, 1>(a);
}

hipError_t pose_opt_launch(const PoseOptArgs& args, hipStream_t stream) {
    if (args.n_frames <= 0) return hipSuccess;
    if (args.force_variant == 3) {          // by the device-side count (max_features <= 256)
        if (args.max_features > 256) return hipErrorInvalidValue;
        hipLaunchKernelGGL(pose_opt_auto_kernel, dim3((unsigned)args.n_frames), dim3(256), 0, stream, args);
        return hipGetLastError();
    }
    // a few frames (the live tracker refines one): latency counts, four waves share a frame's features;
    // batches: one wave per frame, the solver part is not repeated
    const bool no_cache = options().po_no_cache != 0;                          // diagnostic (A/B)
    if (args.force_variant == 4 || (args.force_variant == 0 && args.n_frames > 32 && options().po_rows)) {
        // batches: four frames per wavefront (round 6: the solver part of an iteration, two thirds of its instructions, runs for
        // four frames at once)
        // — as many frames per wave as still leave every wave slot of the GPU (256 CUs x 4 SIMDs x 2 waves) a wave: a wave's
        // chain gets longer with every frame it carries, which only pays while the machine stays full (4096 x 200: one frame per
        // wave 0.706 ms, two 0.541 ms, four 0.648 ms — half the slots empty; 16 384 x 200: 2.37 / - / 1.67 ms; two frames per wave at ONE
        // wave per SIMD, which features in LDS would need: 0.868 ms)
        const int slots = 256 * 4 * PO_WAVES_PER_EU;
        const int gpw = args.force_variant == 4 ? 4 : (args.n_frames >= 4 * slots ? 4 : (args.n_frames >= 2 * slots ? 2 : 1));
        if (gpw == 4) { hipLaunchKernelGGL(pose_opt_rows_kernel<4>, dim3((unsigned)((args.n_frames + 3) / 4)), dim3(64), 0, stream, args); return hipGetLastError(); }
        if (gpw == 2) { hipLaunchKernelGGL(pose_opt_rows_kernel<2>, dim3((unsigned)((args.n_frames + 1) / 2)), dim3(64), 0, stream, args); return hipGetLastError(); }
    }
    if (args.force_variant == 1) { hipLaunchKernelGGL((pose_opt_kernel<1, 0>), dim3((unsigned)args.n_frames), dim3(64), 0, stream, args); return hipGetLastError(); }
    if (args.force_variant == 2) { hipLaunchKernelGGL((pose_opt_kernel<4, 1>), dim3((unsigned)args.n_frames), dim3(256), 0, stream, args); return hipGetLastError(); }
    if (args.n_frames <= 32 && args.max_features > 64 && args.max_features <= 256 && !no_cache)
        hipLaunchKernelGGL((pose_opt_kernel<4, 1>), dim3((unsigned)args.n_frames), dim3(256), 0, stream, args);
    else if (args.n_frames <= 32 && args.max_features > 256 && args.max_features <= 512 && !no_cache)
        hipLaunchKernelGGL((pose_opt_kernel<4, 2>), dim3((unsigned)args.n_frames), dim3(256), 0, stream, args);
    else if (args.n_frames <= 32 && args.max_features > 64)
        hipLaunchKernelGGL((pose_opt_kernel<4, 0>), dim3((unsigned)args.n_frames), dim3(256), 0, stream, args);
    else
        hipLaunchKernelGGL((pose_opt_kernel<1, 0>), dim3((unsigned)args.n_frames), dim3(64), 0, stream, args);
    return hipGetLastError();
}

}  // namespace dsdtm
