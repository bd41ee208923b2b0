/*
 * pose_opt_oracle.c — CPU restatement of Optimizer::PoseOptimization (src/Optimizer.cpp:20-101).
 *
 * TEST INFRASTRUCTURE ONLY (see dsdtm_oracle.h): the checker for dsdtm_pose_optimization.
 *
 * PARITY UNPINNED. The reference holds no expected outputs for this function
 * (Test/test_Optimizer.cpp:128 only calls it on an external dataset), and the solver it calls —
 * Ceres Solver — is not under /root/reference: README.md:7 links the repository without a version
 * (the reference was written in 11/2017, when Ceres 1.13.0 was current). What is restated here:
 *
 *   the reference's own part, line by line
 *     - parameter block [t, log R] of the frame pose (src/Optimizer.cpp:35-37)
 *     - one residual block per feature with Mpt && !IsBad && mbInitial (:47-65), map points constant (:60-61)
 *     - FullBA_Problem::Evaluate: residual and 2x6 Jacobian (include/Optimizer.h:139-197). NOTE the
 *       Jacobian is NOT divided by 1 << level although the residual is (:162 vs :176-189): for
 *       features found on coarser levels the model over-predicts by that factor. Kept.
 *     - PoseLocalParameterization::Plus: T_new = SE3(exp(d_w), d_t) * T_old, parameters re-extracted with
 *       so3().log() (:222-236); ComputeJacobian = identity (:238-244)
 *     - CauchyLoss(1.0) (src/Optimizer.cpp:33), options: DENSE_SCHUR, max_num_iterations = 100 (:68-72;
 *       the tIterations argument is never read)
 *     - Set_Pose(SE3(SO3::exp(x.tail), x.head)) (:78); residual norms of GetReprojectReidual (:297-317)
 *   Ceres 1.13 (published algorithm, restated from its documentation and sources as remembered; the
 *   behaviours that differ between Ceres versions are marked VERSION below)
 *     - program reduction: constant blocks are removed, so the problem is one 6-parameter block; with no
 *       e-blocks left DENSE_SCHUR is replaced by DENSE_QR
 *     - ResidualBlock::Evaluate + Corrector: cost = 1/2 rho(s), s = |r|^2; Cauchy has rho'' < 0, so the
 *       correction is r *= sqrt(rho'), J *= sqrt(rho') (corrector.cc, alpha = 0 branch)
 *     - TrustRegionMinimizer (trust_region_minimizer.cc): Jacobi scaling 1/(1+sqrt(colnorm^2)) fixed at
 *       iteration 0; gradient test on |x - Plus(x,-g)|_inf <= 1e-10; step validity by model decrease;
 *       parameter tolerance 1e-8, function tolerance 1e-6 tested on the CANDIDATE before it is accepted
 *       (VERSION: since 1.12 a converged run returns without taking that last candidate; <= 1.11 took it);
 *       acceptance rho > min_relative_decrease = 1e-3
 *     - LevenbergMarquardtStrategy: D^2 = clamp(diag(J^T J), 1e-6, 1e32) / radius, recomputed after an
 *       accepted step only; radius /= max(1/3, 1 - (2 rho - 1)^3) on acceptance, /= 2, 4, 8.. on rejection;
 *       initial radius 1e4, max 1e16, min 1e-32
 *     - DenseQRSolver: least squares of [J; D] y = [r; 0] by Householder QR, step = -y
 *       (VERSION: 1.13 calls Eigen's colPivHouseholderQr here, later versions householderQr/LAPACK; for
 *       the full-rank systems of this problem they agree to rounding)
 *   Sophus (non-templated): SO3::exp / SO3::log (atan form), SE3 product — as in dsdtm_oracle.c.
 *
 * linear_solver = 0: the Householder-QR form above (closest to what Ceres runs).
 * linear_solver = 1: the same minimiser with the step from the 6x6 normal equations
 *   (S A S + D^2) y = S g by Cholesky, model decrease in its quadratic form — exactly what the HIP
 *   kernel computes (one reduction per iteration). tests/test_pose_opt_cpu.py holds the two forms
 *   together (same iterations, same decisions, poses within 1e-9).
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "dsdtm_oracle.h"
#include "mutants.h"     /* MUT(x): 0 in the faithful build (see dsdtm_oracle.c) */

#define SMALL_EPS 1e-10

/* Sophus SO3::exp (so3.cpp expAndTheta) through the SE3 restatement: translation part unused */
static void so3_exp(const double w[3], double q[4]) {
    double x[6] = {0.0, 0.0, 0.0, w[0], w[1], w[2]};
    oracle_se3 e;
    oracle_se3_exp(x, &e);
    memcpy(q, e.q, sizeof(e.q));
}

/* Sophus SO3::logAndTheta (so3.cpp): atan-based log of the unit quaternion */
void oracle_so3_log(const double q[4], double w[3]) {
    const double n = sqrt(q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const double qw = q[0];
    const double squared_w = qw * qw;
    double two_atan_nbyw_by_n;
    if (n < SMALL_EPS) {
        two_atan_nbyw_by_n = 2. / qw - 2. * (n * n) / (qw * squared_w);
    } else {
        if (fabs(qw) < SMALL_EPS) {
            if (qw > 0) two_atan_nbyw_by_n = M_PI / n;
            else two_atan_nbyw_by_n = -M_PI / n;
        }
        two_atan_nbyw_by_n = 2 * atan(n / qw) / n;   /* not in an else: overwrites the branch above, as in Sophus */
    }
    w[0] = two_atan_nbyw_by_n * q[1];
    w[1] = two_atan_nbyw_by_n * q[2];
    w[2] = two_atan_nbyw_by_n * q[3];
}

/* x = [t, w] -> SE3(SO3::exp(w), t) (include/Optimizer.h:147) */
static void pose_of(const double x[6], oracle_se3* T) {
    so3_exp(x + 3, T->q);
    T->t[0] = x[0]; T->t[1] = x[1]; T->t[2] = x[2];
}

/* PoseLocalParameterization::Plus (include/Optimizer.h:222-236) */
void oracle_pose_plus(const double x[6], const double delta[6], double out[6]) {
    if (MUT(MUT_P3_PLUS)) { for (int k = 0; k < 6; ++k) out[k] = x[k] + delta[k]; return; }   /* "fixed": plain vector sum */
    oracle_se3 To, Td, Tn;
    pose_of(x, &To);
    pose_of(delta, &Td);
    oracle_se3_mul(&Td, &To, &Tn);
    out[0] = Tn.t[0]; out[1] = Tn.t[1]; out[2] = Tn.t[2];
    oracle_so3_log(Tn.q, out + 3);
}

typedef struct {
    int n;                 /* residual blocks */
    const double* obs;     /* n x 2 */
    const double* pw;      /* n x 3 */
    const double* inv;     /* n: 1 << level as double */
} blocks_t;

static int finite_all(const double* v, int n) {
    for (int i = 0; i < n; ++i)
        if (!isfinite(v[i])) return 0;
    return 1;
}

/* FullBA_Problem::Evaluate for block i at pose T (include/Optimizer.h:139-197) */
static void block_eval(const blocks_t* b, int i, const oracle_se3* T, double r[2], double J[12]) {
    double p[3];
    oracle_se3_act(T, b->pw + 3 * i, p);
    const double pred0 = p[0] / p[2], pred1 = p[1] / p[2];
    r[0] = (b->obs[2 * i] - pred0) / b->inv[i];
    r[1] = (b->obs[2 * i + 1] - pred1) / b->inv[i];
    if (J) {
        const double x = p[0], y = p[1];
        const double z_inv = 1.0 / p[2];
        const double z_inv2 = z_inv * z_inv;
        J[0] = -z_inv;
        J[1] = 0.0;
        J[2] = x * z_inv2;
        J[3] = y * J[2];
        J[4] = -(1.0 + x * J[2]);
        J[5] = y * z_inv;
        J[6] = 0.0;
        J[7] = -z_inv;
        J[8] = y * z_inv2;
        J[9] = 1.0 + y * J[8];
        J[10] = -x * J[8];
        J[11] = -x * z_inv;
        if (MUT(MUT_P1_JSCALE)) for (int k = 0; k < 12; ++k) J[k] /= b->inv[i];   /* "fixed": scaled like the residual */
    }
}

/* Program evaluation: cost = sum 1/2 rho(|r|^2); with rj != NULL also the corrected residuals (2n),
 * corrected Jacobian (2n x 6 row-major) and gradient J^T r. Returns 0 when something is not finite. */
static int evaluate(const blocks_t* b, const double x[6], double* cost, double* res, double* jac, double g[6],
                    double Hn[36] /* J^T J of the corrected, unscaled blocks */) {
    oracle_se3 T;
    pose_of(x, &T);
    double c = 0.0;
    if (g) memset(g, 0, 6 * sizeof(double));
    if (Hn) memset(Hn, 0, 36 * sizeof(double));
    for (int i = 0; i < b->n; ++i) {
        double r[2], J[12];
        block_eval(b, i, &T, r, jac ? J : NULL);
        if (!finite_all(r, 2) || (jac && !finite_all(J, 12))) return 0;
        const double s = r[0] * r[0] + r[1] * r[1];
        /* CauchyLoss(1.0)::Evaluate: b_ = 1, c_ = 1 */
        const double sum = 1.0 + s * 1.0;
        const double inv = 1.0 / sum;
        const double rho0 = 1.0 * log(sum);
        double rho1 = inv > DBL_MIN ? inv : DBL_MIN;
        if (MUT(MUT_P2_NOLOSS)) { c += 0.5 * s - 0.5 * rho0; rho1 = 1.0; }   /* "fixed": TrivialLoss (rho = s) */
        c += 0.5 * rho0;
        if (jac) {
            const double sq = sqrt(rho1);            /* Corrector, alpha = 0 (rho'' < 0) */
            for (int k = 0; k < 12; ++k) jac[(size_t)i * 12 + k] = J[k] * sq;
            res[2 * i] = r[0] * sq;
            res[2 * i + 1] = r[1] * sq;
            for (int k = 0; k < 6; ++k)
                g[k] += jac[(size_t)i * 12 + k] * res[2 * i] + jac[(size_t)i * 12 + 6 + k] * res[2 * i + 1];
            if (Hn)
                for (int a = 0; a < 6; ++a)
                    for (int c2 = 0; c2 < 6; ++c2)
                        Hn[a * 6 + c2] += jac[(size_t)i * 12 + a] * jac[(size_t)i * 12 + c2] +
                                          jac[(size_t)i * 12 + 6 + a] * jac[(size_t)i * 12 + 6 + c2];
        }
    }
    *cost = c;
    return 1;
}

/* Eigen HouseholderQR (unblocked) least squares of the m x 6 system A y = rhs; A, rhs overwritten */
static void householder_ls(double* A, double* rhs, int m, double y[6]) {
    for (int k = 0; k < 6; ++k) {
        double tail = 0.0;
        for (int i = k + 1; i < m; ++i) tail += A[(size_t)i * 6 + k] * A[(size_t)i * 6 + k];
        const double c0 = A[(size_t)k * 6 + k];
        double tau, beta;
        if (tail == 0.0) {
            tau = 0.0; beta = c0;
            for (int i = k + 1; i < m; ++i) A[(size_t)i * 6 + k] = 0.0;
        } else {
            beta = sqrt(c0 * c0 + tail);
            if (c0 >= 0.0) beta = -beta;
            for (int i = k + 1; i < m; ++i) A[(size_t)i * 6 + k] /= (c0 - beta);
            tau = (beta - c0) / beta;
        }
        A[(size_t)k * 6 + k] = beta;
        for (int j = k + 1; j <= 6; ++j) {          /* j == 6: the right-hand side */
            double w = 0.0;
            for (int i = k + 1; i < m; ++i) w += A[(size_t)i * 6 + k] * (j < 6 ? A[(size_t)i * 6 + j] : rhs[i]);
            w += (j < 6 ? A[(size_t)k * 6 + j] : rhs[k]);
            if (j < 6) A[(size_t)k * 6 + j] -= tau * w; else rhs[k] -= tau * w;
            for (int i = k + 1; i < m; ++i) {
                if (j < 6) A[(size_t)i * 6 + j] -= tau * w * A[(size_t)i * 6 + k];
                else rhs[i] -= tau * w * A[(size_t)i * 6 + k];
            }
        }
    }
    for (int i = 5; i >= 0; --i) {                  /* R y = (Q^T rhs)[0..5] */
        double s = rhs[i];
        for (int j = i + 1; j < 6; ++j) s -= A[(size_t)i * 6 + j] * y[j];
        y[i] = s / A[(size_t)i * 6 + i];
    }
}

/* Cholesky solve of the SPD 6x6 system M y = v (row-major full matrix); returns 0 on a non-positive pivot.
 * One reciprocal per pivot, as the kernel does it. */
int oracle_chol6_solve(const double M[36], const double v[6], double y[6]) {
    double L[36], rl[6];
    memset(L, 0, sizeof(L));
    for (int j = 0; j < 6; ++j) {
        double d = M[j * 6 + j];
        for (int k = 0; k < j; ++k) d -= L[j * 6 + k] * L[j * 6 + k];
        if (!(d > 0.0)) return 0;
        const double ljj = sqrt(d);
        L[j * 6 + j] = ljj;
        rl[j] = 1.0 / ljj;
        for (int i = j + 1; i < 6; ++i) {
            double s = M[i * 6 + j];
            for (int k = 0; k < j; ++k) s -= L[i * 6 + k] * L[j * 6 + k];
            L[i * 6 + j] = s * rl[j];
        }
    }
    double z[6];
    for (int i = 0; i < 6; ++i) {
        double s = v[i];
        for (int k = 0; k < i; ++k) s -= L[i * 6 + k] * z[k];
        z[i] = s * rl[i];
    }
    for (int i = 5; i >= 0; --i) {
        double s = z[i];
        for (int k = i + 1; k < 6; ++k) s -= L[k * 6 + i] * y[k];
        y[i] = s * rl[i];
    }
    return 1;
}

static double norm6(const double v[6]) {
    double s = 0.0;
    for (int i = 0; i < 6; ++i) s += v[i] * v[i];
    return sqrt(s);
}

int oracle_pose_optimization(const double* bearing, const double* p_world, const int32_t* level,
                             const uint8_t* use, int n_features, double T_cur_w[12],
                             const dsdtm_pose_opt_params* prm, int linear_solver,
                             double* residual_norm, dsdtm_pose_opt_summary* summary,
                             double* trace, int trace_cap) {
    if (!T_cur_w || !prm || !summary || n_features < 0) return DSDTM_ERR_INVALID;
    if (n_features > 0 && (!bearing || !p_world || !level || !use || !residual_norm)) return DSDTM_ERR_INVALID;
    memset(summary, 0, sizeof(*summary));

    /* residual blocks in feature order (src/Optimizer.cpp:45-65) */
    int n = 0;
    for (int i = 0; i < n_features; ++i) n += (use[i] != 0 || MUT(MUT_P4_ALLFEAT));
    double* obs = (double*)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1) * 2);
    double* pw = (double*)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1) * 3);
    double* inv = (double*)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    const int m = 2 * n + 6;
    double* res = (double*)malloc(sizeof(double) * (size_t)m);
    double* jac = (double*)malloc(sizeof(double) * (size_t)m * 6);
    double* lhs = (double*)malloc(sizeof(double) * (size_t)m * 6);
    double* rhs = (double*)malloc(sizeof(double) * (size_t)m);
    for (int i = 0, k = 0; i < n_features; ++i) {
        if (!use[i] && !MUT(MUT_P4_ALLFEAT)) continue;
        obs[2 * k] = bearing[3 * i] / bearing[3 * i + 2];       /* include/Optimizer.h:160 */
        obs[2 * k + 1] = bearing[3 * i + 1] / bearing[3 * i + 2];
        if (MUT(MUT_P5_PIXELS)) { obs[2 * k] = bearing[3 * i]; obs[2 * k + 1] = bearing[3 * i + 1]; }
        pw[3 * k] = p_world[3 * i]; pw[3 * k + 1] = p_world[3 * i + 1]; pw[3 * k + 2] = p_world[3 * i + 2];
        inv[k] = (double)(1 << level[i]);
        ++k;
    }
    blocks_t B = {n, obs, pw, inv};
    summary->n_residual_blocks = n;

    /* parameter block (src/Optimizer.cpp:35-37) */
    oracle_se3 T0;
    oracle_se3_from_rt(T_cur_w, &T0);
    double x[6] = {T0.t[0], T0.t[1], T0.t[2], 0, 0, 0};
    oracle_so3_log(T0.q, x + 3);

    int termination = DSDTM_PO_MAX_ITERATIONS;
    int iterations = 0, successful = 0;
    double x_cost = 0.0, initial_cost = 0.0;
    double g[6], scale[6], diagonal[6], Hn[36];

    if (n == 0) {
        termination = DSDTM_PO_NO_RESIDUALS;          /* Ceres: nothing to optimise, parameters untouched */
    } else if (!evaluate(&B, x, &x_cost, res, jac, g, Hn)) {
        termination = DSDTM_PO_EVALUATION_FAILED;     /* "Residual and Jacobian evaluation failed" */
    } else {
        initial_cost = x_cost;
        /* Jacobi scaling, fixed at iteration 0; columns scaled after every evaluation */
        for (int k = 0; k < 6; ++k) {
            double s = 0.0;
            for (int i = 0; i < 2 * n; ++i) s += jac[(size_t)i * 6 + k] * jac[(size_t)i * 6 + k];
            scale[k] = 1.0 / (1.0 + sqrt(s));
        }
        for (int i = 0; i < 2 * n; ++i)
            for (int k = 0; k < 6; ++k) jac[(size_t)i * 6 + k] *= scale[k];
        double x_norm = norm6(x);
        double gradient_max_norm;
        {
            double ng[6], xp[6];
            for (int k = 0; k < 6; ++k) ng[k] = -g[k];
            oracle_pose_plus(x, ng, xp);
            gradient_max_norm = 0.0;
            for (int k = 0; k < 6; ++k) gradient_max_norm = fmax(gradient_max_norm, fabs(x[k] - xp[k]));
        }
        double radius = 1e4, decrease_factor = 2.0;
        int reuse_diagonal = 0, invalid_steps = 0;
        int it = 0;                                    /* index of the last finished iteration */
        for (;;) {
            /* FinalizeIterationAndCheckIfMinimizerCanContinue */
            if (trace && it < trace_cap) {
                trace[4 * it] = x_cost; trace[4 * it + 1] = radius; trace[4 * it + 2] = gradient_max_norm;
                trace[4 * it + 3] = (double)successful;
            }
            if (it >= (MUT(MUT_P6_ITERS) ? 10 : prm->max_iterations)) { termination = DSDTM_PO_MAX_ITERATIONS; break; }
            if (gradient_max_norm <= 1e-10) { termination = DSDTM_PO_GRADIENT_TOLERANCE; break; }
            if (radius <= 1e-32) { termination = DSDTM_PO_MIN_RADIUS; break; }
            ++it;
            /* LevenbergMarquardtStrategy::ComputeStep */
            double As[36], gs[6];
            if (linear_solver == 1) {                  /* the kernel's form: scale the reduced 6x6, not the rows */
                for (int a = 0; a < 6; ++a) {
                    gs[a] = scale[a] * g[a];
                    for (int c = 0; c < 6; ++c) As[a * 6 + c] = (scale[a] * Hn[a * 6 + c]) * scale[c];
                }
            }
            if (!reuse_diagonal) {
                for (int k = 0; k < 6; ++k) {
                    double s = 0.0;
                    if (linear_solver == 1) s = As[k * 6 + k];
                    else for (int i = 0; i < 2 * n; ++i) s += jac[(size_t)i * 6 + k] * jac[(size_t)i * 6 + k];
                    diagonal[k] = fmin(fmax(s, 1e-6), 1e32);
                }
            }
            double lm[6], step[6], y[6];
            for (int k = 0; k < 6; ++k) lm[k] = sqrt(diagonal[k] / radius);
            int solved = 1;
            double model_cost_change;
            if (linear_solver == 0) {
                memcpy(lhs, jac, sizeof(double) * (size_t)2 * n * 6);
                memcpy(rhs, res, sizeof(double) * (size_t)2 * n);
                for (int k = 0; k < 6; ++k) {
                    for (int j = 0; j < 6; ++j) lhs[(size_t)(2 * n + k) * 6 + j] = (j == k) ? lm[k] : 0.0;
                    rhs[2 * n + k] = 0.0;
                }
                householder_ls(lhs, rhs, m, y);
                solved = finite_all(y, 6);
                for (int k = 0; k < 6; ++k) step[k] = -y[k];
                /* model_residuals = J step; model_cost_change = -model_residuals . (residuals + model_residuals / 2) */
                double acc = 0.0;
                for (int i = 0; i < 2 * n; ++i) {
                    double mr = 0.0;
                    for (int k = 0; k < 6; ++k) mr += jac[(size_t)i * 6 + k] * step[k];
                    acc += mr * (res[i] + mr / 2.0);
                }
                model_cost_change = -acc;
            } else {
                double M[36];
                memcpy(M, As, sizeof(M));
                for (int k = 0; k < 6; ++k) M[k * 6 + k] += lm[k] * lm[k];
                solved = oracle_chol6_solve(M, gs, y) && finite_all(y, 6);
                for (int k = 0; k < 6; ++k) step[k] = -y[k];
                /* the same quantity in its quadratic form: -(gs . step + step^T A step / 2) */
                double lin = 0.0, quad = 0.0;
                for (int a = 0; a < 6; ++a) {
                    lin += gs[a] * step[a];
                    double s = 0.0;
                    for (int c = 0; c < 6; ++c) s += As[a * 6 + c] * step[c];
                    quad += step[a] * s;
                }
                model_cost_change = -(lin + quad / 2.0);
            }
            reuse_diagonal = 1;
            const int step_is_valid = solved && (model_cost_change > 0.0);
            if (!step_is_valid) {
                /* HandleInvalidStep */
                if (++invalid_steps >= 5) { termination = DSDTM_PO_INVALID_STEPS; break; }
                radius = radius / decrease_factor;     /* StepIsInvalid = StepRejected(0) */
                decrease_factor *= 2.0;
                continue;
            }
            invalid_steps = 0;
            double delta[6], cand[6], cand_cost;
            for (int k = 0; k < 6; ++k) delta[k] = step[k] * scale[k];
            oracle_pose_plus(x, delta, cand);
            if (!finite_all(cand, 6) || !evaluate(&B, cand, &cand_cost, NULL, NULL, NULL, NULL)) cand_cost = DBL_MAX;
            /* ParameterToleranceReached */
            double diff[6];
            for (int k = 0; k < 6; ++k) diff[k] = x[k] - cand[k];
            if (norm6(diff) <= 1e-8 * (x_norm + 1e-8)) { termination = DSDTM_PO_PARAMETER_TOLERANCE; break; }
            /* FunctionToleranceReached */
            const double cost_change = x_cost - cand_cost;
            if (fabs(cost_change) <= 1e-6 * x_cost) { termination = DSDTM_PO_FUNCTION_TOLERANCE; break; }
            /* IsStepSuccessful */
            const double relative_decrease = cost_change / model_cost_change;
            if (relative_decrease > 1e-3) {
                /* HandleSuccessfulStep */
                memcpy(x, cand, sizeof(x));
                x_norm = norm6(x);
                if (!evaluate(&B, x, &x_cost, res, jac, g, Hn)) { termination = DSDTM_PO_EVALUATION_FAILED; break; }
                for (int i = 0; i < 2 * n; ++i)
                    for (int k = 0; k < 6; ++k) jac[(size_t)i * 6 + k] *= scale[k];
                double ng[6], xp[6];
                for (int k = 0; k < 6; ++k) ng[k] = -g[k];
                oracle_pose_plus(x, ng, xp);
                gradient_max_norm = 0.0;
                for (int k = 0; k < 6; ++k) gradient_max_norm = fmax(gradient_max_norm, fabs(x[k] - xp[k]));
                ++successful;
                radius = radius / fmax(1.0 / 3.0, 1.0 - pow(2.0 * relative_decrease - 1.0, 3));
                radius = fmin(1e16, radius);
                decrease_factor = 2.0;
                reuse_diagonal = 0;
            } else {
                /* HandleUnsuccessfulStep */
                radius = radius / decrease_factor;
                decrease_factor *= 2.0;
                reuse_diagonal = 1;
            }
        }
        iterations = it;
    }

    /* Set_Pose(SE3(SO3::exp(x.tail<3>()), x.head<3>())) (src/Optimizer.cpp:78) */
    oracle_se3 Tf;
    pose_of(x, &Tf);
    oracle_se3_to_rt(&Tf, T_cur_w);
    /* GetReprojectReidual (src/Optimizer.cpp:297-317): raw residual norms at the final parameters */
    for (int i = 0; i < n; ++i) {
        double r[2];
        block_eval(&B, i, &Tf, r, NULL);
        residual_norm[i] = sqrt(r[0] * r[0] + r[1] * r[1]);
    }
    summary->iterations = iterations;
    summary->successful_steps = successful;
    summary->termination = termination;
    summary->initial_cost = initial_cost;
    summary->final_cost = x_cost;
    memcpy(summary->x, x, sizeof(x));
    free(obs); free(pw); free(inv); free(res); free(jac); free(lhs); free(rhs);
    return DSDTM_OK;
}
