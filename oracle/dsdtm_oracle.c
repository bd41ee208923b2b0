/*
 * dsdtm_oracle.c — CPU restatement (plain C, double/float exactly where the reference
 * uses them) of DSDTM's sparse photometric alignment path.
 *
 * TEST INFRASTRUCTURE ONLY — see dsdtm_oracle.h. PARITY UNPINNED by the reference (it
 * holds no golden vectors for this path), except the FAST-10 detector part at the end of this
 * file, which is pinned to a build of the reference's own vendored sources; every function
 * cites the reference lines it follows so the restatement can be audited by reading.
 *
 * Build: see oracle/Makefile (-O2, no -ffast-math, -ffp-contract=off: the reference is
 * built with -msse..-mssse3 only (CMakeLists.txt:5-8), i.e. without FMA contraction).
 */
#include "dsdtm_oracle.h"

#include <math.h>
#include <float.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ===================================================================================
 * Mutation switches (oracle/mutants.h): with -DORACLE_MUTANTS (make mutants -> liboracle_mut.so) every quirk
 * of SURVEY.md §8.1 can be "fixed" one at a time at run time, so that the parity suite can show it notices
 * (tests/test_mutants_*.py). In the faithful build MUT(x) is the constant 0 and none of this exists.
 * =================================================================================== */
#include "mutants.h"
#ifdef ORACLE_MUTANTS
int oracle_mutant_id = MUT_NONE;
void oracle_set_mutant(int id) { oracle_mutant_id = id; }
int oracle_get_mutant(void) { return oracle_mutant_id; }
#endif

/* ===================================================================================
 * Sophus (non-templated) SE3 / SO3 and the Eigen quaternion operations they use.
 * Third-party, not under /root/reference; restated from the published implementation
 * (sophus/so3.cpp, sophus/se3.cpp of the strasdat/Sophus "a621ff" lineage that SVO and
 * DSDTM link as libSophus.so; CMakeLists.txt:27-29). Call sites on the path:
 * src/Sprase_ImageAlign.cpp:43,57,254,335; src/Frame.cpp:171-173;
 * src/Feature_alignment.cpp:181-184.
 * =================================================================================== */
#define SOPHUS_SMALL_EPS 1e-10

static void quat_normalize(double q[4]) {
    /* Eigen QuaternionBase::normalize(): coeffs /= norm() */
    double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}

static void quat_mul(const double a[4], const double b[4], double o[4]) {
    /* Eigen quat_product<..., double>: (w,x,y,z) Hamilton product */
    double w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    double x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    double y = a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3];
    double z = a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1];
    o[0] = w; o[1] = x; o[2] = y; o[3] = z;
}

static void quat_rotate(const double q[4], const double v[3], double o[3]) {
    /* Eigen QuaternionBase::_transformVector: uv = 2 * vec x v; v + w*uv + vec x uv */
    double ux = q[2] * v[2] - q[3] * v[1];
    double uy = q[3] * v[0] - q[1] * v[2];
    double uz = q[1] * v[1] - q[2] * v[0];
    ux += ux; uy += uy; uz += uz;
    double cx = q[2] * uz - q[3] * uy;
    double cy = q[3] * ux - q[1] * uz;
    double cz = q[1] * uy - q[2] * ux;
    o[0] = v[0] + q[0] * ux + cx;
    o[1] = v[1] + q[0] * uy + cy;
    o[2] = v[2] + q[0] * uz + cz;
}

static void quat_from_matrix(const double m[9], double q[4]) {
    /* Eigen quaternionbase_assign_impl<Other,3,3>::run */
    double t = m[0] + m[4] + m[8];
    if (t > 0.0) {
        t = sqrt(t + 1.0);
        q[0] = 0.5 * t;
        t = 0.5 / t;
        q[1] = (m[7] - m[5]) * t;
        q[2] = (m[2] - m[6]) * t;
        q[3] = (m[3] - m[1]) * t;
    } else {
        int i = 0;
        if (m[4] > m[0]) i = 1;
        if (m[8] > m[i * 3 + i]) i = 2;
        int j = (i + 1) % 3;
        int k = (j + 1) % 3;
        t = sqrt(m[i * 3 + i] - m[j * 3 + j] - m[k * 3 + k] + 1.0);
        q[1 + i] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (m[k * 3 + j] - m[j * 3 + k]) * t;
        q[1 + j] = (m[j * 3 + i] + m[i * 3 + j]) * t;
        q[1 + k] = (m[k * 3 + i] + m[i * 3 + k]) * t;
    }
}

static void quat_to_matrix(const double q[4], double R[9]) {
    /* Eigen QuaternionBase::toRotationMatrix */
    const double tx = 2.0 * q[1], ty = 2.0 * q[2], tz = 2.0 * q[3];
    const double twx = tx * q[0], twy = ty * q[0], twz = tz * q[0];
    const double txx = tx * q[1], txy = ty * q[1], txz = tz * q[1];
    const double tyy = ty * q[2], tyz = tz * q[2], tzz = tz * q[3];
    R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz;         R[2] = txz + twy;
    R[3] = txy + twz;         R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;         R[7] = tyz + twx;         R[8] = 1.0 - (txx + tyy);
}

void oracle_se3_from_rt(const double T[12], oracle_se3* out) {
    double R[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
    quat_from_matrix(R, out->q);
    quat_normalize(out->q); /* SO3(const Matrix3d&) -> unit_quaternion_(R); SO3 ctor normalises */
    out->t[0] = T[3]; out->t[1] = T[7]; out->t[2] = T[11];
}

void oracle_se3_to_rt(const oracle_se3* in, double T[12]) {
    double R[9];
    quat_to_matrix(in->q, R);
    T[0] = R[0]; T[1] = R[1]; T[2] = R[2];  T[3] = in->t[0];
    T[4] = R[3]; T[5] = R[4]; T[6] = R[5];  T[7] = in->t[1];
    T[8] = R[6]; T[9] = R[7]; T[10] = R[8]; T[11] = in->t[2];
}

void oracle_se3_act(const oracle_se3* a, const double p[3], double out[3]) {
    /* SE3::operator*(Vector3d): so3_*xyz + translation_ */
    double r[3];
    quat_rotate(a->q, p, r);
    out[0] = r[0] + a->t[0]; out[1] = r[1] + a->t[1]; out[2] = r[2] + a->t[2];
}

void oracle_se3_mul(const oracle_se3* a, const oracle_se3* b, oracle_se3* out) {
    /* SE3::operator*=: translation_ += so3_*(other.translation_); so3_ *= other.so3_
     * SO3::operator*=: unit_quaternion_ *= other.unit_quaternion_; normalize() */
    oracle_se3 r;
    double rt[3];
    quat_rotate(a->q, b->t, rt);
    r.t[0] = a->t[0] + rt[0]; r.t[1] = a->t[1] + rt[1]; r.t[2] = a->t[2] + rt[2];
    quat_mul(a->q, b->q, r.q);
    quat_normalize(r.q);
    *out = r;
}

void oracle_se3_inverse(const oracle_se3* a, oracle_se3* out) {
    /* SE3::inverse: ret.so3_ = so3_.inverse() (conjugate); ret.translation_ = ret.so3_*(translation_*-1.) */
    oracle_se3 r;
    r.q[0] = a->q[0]; r.q[1] = -a->q[1]; r.q[2] = -a->q[2]; r.q[3] = -a->q[3];
    double nt[3] = {a->t[0] * -1., a->t[1] * -1., a->t[2] * -1.};
    quat_rotate(r.q, nt, r.t);
    *out = r;
}

void oracle_se3_exp(const double x[6], oracle_se3* out) {
    /* SE3::exp(Vector6d update): upsilon = head<3>, omega = tail<3>;
     * SO3::expAndTheta; V = I + (1-cos)/th^2 * Om + (th-sin)/th^3 * Om^2 */
    const double ux = x[0], uy = x[1], uz = x[2];
    const double wx = x[3], wy = x[4], wz = x[5];
    double theta = sqrt(wx * wx + wy * wy + wz * wz);
    double half_theta = 0.5 * theta;
    double imag_factor;
    double real_factor = cos(half_theta);
    if (theta < SOPHUS_SMALL_EPS) {
        double theta_sq = theta * theta;
        double theta_po4 = theta_sq * theta_sq;
        imag_factor = 0.5 - 0.0208333 * theta_sq + 0.000260417 * theta_po4;
    } else {
        double sin_half_theta = sin(half_theta);
        imag_factor = sin_half_theta / theta;
    }
    out->q[0] = real_factor;
    out->q[1] = imag_factor * wx;
    out->q[2] = imag_factor * wy;
    out->q[3] = imag_factor * wz;
    quat_normalize(out->q); /* SO3(Quaterniond) normalises */

    /* Omega = hat(omega), Omega_sq = Omega*Omega */
    const double Om[9] = {0.0, -wz, wy, wz, 0.0, -wx, -wy, wx, 0.0};
    double Om2[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0.0;
            for (int k = 0; k < 3; ++k) s += Om[i * 3 + k] * Om[k * 3 + j];
            Om2[i * 3 + j] = s;
        }
    double V[9];
    if (theta < SOPHUS_SMALL_EPS) {
        quat_to_matrix(out->q, V); /* V = so3.matrix() */
    } else {
        double theta_sq = theta * theta;
        double a = (1.0 - cos(theta)) / theta_sq;
        double b = (theta - sin(theta)) / (theta_sq * theta);
        for (int i = 0; i < 9; ++i) V[i] = ((i % 4 == 0) ? 1.0 : 0.0) + a * Om[i] + b * Om2[i];
    }
    out->t[0] = V[0] * ux + V[1] * uy + V[2] * uz;
    out->t[1] = V[3] * ux + V[4] * uy + V[5] * uz;
    out->t[2] = V[6] * ux + V[7] * uy + V[8] * uz;
}

/* ===================================================================================
 * Eigen 3.2.0 LDLT<Matrix<double,6,6>, Lower> (Eigen/src/Cholesky/LDLT.h):
 * ldlt_inplace<Lower>::unblocked with diagonal pivoting and the rank cutoff, then
 * solve with the pseudo-inverse of D. Call site: src/Sprase_ImageAlign.cpp:318.
 * =================================================================================== */
void oracle_ldlt6_solve(const double Hin[36], const double b[6], double x[6]) {
    enum { N = 6 };
    double m[N][N];
    int tr[N];
    double temp[N];
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) m[i][j] = Hin[i * N + j];

    double cutoff = 0.0;
    for (int k = 0; k < N; ++k) {
        /* largest |diagonal| in the trailing corner (first maximum wins, as maxCoeff) */
        int piv = k;
        double big = fabs(m[k][k]);
        for (int i = k + 1; i < N; ++i) {
            double a = fabs(m[i][i]);
            if (a > big) { big = a; piv = i; }
        }
        if (k == 0) cutoff = fabs(DBL_EPSILON * big);
        if (big < cutoff) { /* not full rank: finish early */
            for (int i = k; i < N; ++i) tr[i] = i;
            break;
        }
        tr[k] = piv;
        if (k != piv) {
            /* symmetric transposition touching the lower triangle only */
            for (int j = 0; j < k; ++j) { double t = m[k][j]; m[k][j] = m[piv][j]; m[piv][j] = t; }
            for (int i = piv + 1; i < N; ++i) { double t = m[i][k]; m[i][k] = m[i][piv]; m[i][piv] = t; }
            { double t = m[k][k]; m[k][k] = m[piv][piv]; m[piv][piv] = t; }
            for (int i = k + 1; i < piv; ++i) { double t = m[i][k]; m[i][k] = m[piv][i]; m[piv][i] = t; }
        }
        int rs = N - k - 1;
        if (k > 0) {
            /* temp.head(k) = D.head(k) * A10^T ; A(k,k) -= A10 * temp ; A21 -= A20 * temp */
            for (int j = 0; j < k; ++j) temp[j] = m[j][j] * m[k][j];
            double s = 0.0;
            for (int j = 0; j < k; ++j) s += m[k][j] * temp[j];
            m[k][k] -= s;
            for (int i = k + 1; i < N; ++i) {
                double a = 0.0;
                for (int j = 0; j < k; ++j) a += m[i][j] * temp[j];
                m[i][k] -= a;
            }
        }
        if (rs > 0 && fabs(m[k][k]) > cutoff)
            for (int i = k + 1; i < N; ++i) m[i][k] /= m[k][k];
    }

    /* solve: dst = P b; L^-1; D^+ ; L^-T ; P^T */
    double d[N];
    for (int i = 0; i < N; ++i) d[i] = b[i];
    for (int k = 0; k < N; ++k) if (tr[k] != k) { double t = d[k]; d[k] = d[tr[k]]; d[tr[k]] = t; }
    for (int i = 0; i < N; ++i) {
        double s = 0.0;
        for (int j = 0; j < i; ++j) s += m[i][j] * d[j];
        d[i] -= s;
    }
    double maxd = 0.0;
    for (int i = 0; i < N; ++i) if (fabs(m[i][i]) > maxd) maxd = fabs(m[i][i]);
    double tol = maxd * DBL_EPSILON;
    if (1.0 / DBL_MAX > tol) tol = 1.0 / DBL_MAX;
    for (int i = 0; i < N; ++i) {
        if (fabs(m[i][i]) > tol) d[i] /= m[i][i];
        else d[i] = 0.0;
    }
    for (int i = N - 1; i >= 0; --i) {
        double s = 0.0;
        for (int j = i + 1; j < N; ++j) s += m[j][i] * d[j];
        d[i] -= s;
    }
    for (int k = N - 1; k >= 0; --k) if (tr[k] != k) { double t = d[k]; d[k] = d[tr[k]]; d[tr[k]] = t; }
    for (int i = 0; i < N; ++i) x[i] = d[i];
}

/* ===================================================================================
 * Sprase_ImgAlign
 * =================================================================================== */

/* GetJocabianBA — src/Sprase_ImageAlign.cpp:169-193 */
void oracle_jacobian_ba(const double p[3], double J[12]) {
    const double x = p[0];
    const double y = p[1];
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
}

typedef struct {
    int n;              /* patches that passed the reference-side checks           */
    double* ref_patch;  /* n x 16   mRefPatch       (include/Sprase_ImageAlign.h:61) */
    double* jac;        /* (n*16) x 6 mJocabianPatch (:62)                           */
    double* normals;    /* n x 3    mRefNormals (= bearing * depth) (:63)            */
} level_cache;

/* GetJocabianMat — src/Sprase_ImageAlign.cpp:62-166 */
static void get_jacobian_mat(const dsdtm_pyramid* ref, const dsdtm_camera* cam,
                             const float* px_xy, const double* bearing, const double* p_world,
                             const uint8_t* initial, int n_features,
                             const double ref_cnt[3], const oracle_se3* T_ref, int level, level_cache* c) {
    (void)T_ref;
    const uint8_t* img = ref->data[level];
    const int cols = ref->width[level], rows = ref->height[level];
    const float tScale = (float)(1.0 / (1 << level));          /* :65 */
    const int tRefStep = ref->stride[level];                   /* :66 */
    const int boarder = (int)(0.5 * 4 + 1);                    /* :67, mHalf_PatchSize = 4 */
    const float tFocalth = MUT(MUT_Q1_FX) ? cam->fx : cam->f;  /* :70 */
    const int po = MUT(MUT_Q4_SHIFT) ? 1 : 2;                  /* patch spans -po .. 3-po from floor(px): -2..+1 (:143) */

    double* pts = (double*)malloc(sizeof(double) * 2 * (size_t)(n_features > 0 ? n_features : 1));
    double* depth_pts = (double*)malloc(sizeof(double) * 3 * (size_t)(n_features > 0 ? n_features : 1));
    int tNum = 0;
    for (int i = 0; i < n_features; ++i) {                     /* :84-103 */
        if (!initial[i] && !MUT(MUT_Q3_INITIAL)) continue;
        double px = (double)px_xy[2 * i] * tScale;             /* float mpx -> double, * float scale */
        double py = (double)px_xy[2 * i + 1] * tScale;
        const double* P = p_world + 3 * i;
        int is_zero = (P[0] == 0.0 && P[1] == 0.0 && P[2] == 0.0); /* isZero(0) */
        if (MUT(MUT_Q3_ZERO)) is_zero = 0;
        if (is_zero || px - boarder < 0 || py - boarder < 0 ||
            px + boarder >= cols || py + boarder >= rows)
            continue;
        pts[2 * tNum] = px; pts[2 * tNum + 1] = py;
        depth_pts[3 * tNum] = P[0]; depth_pts[3 * tNum + 1] = P[1]; depth_pts[3 * tNum + 2] = P[2];
        c->normals[3 * tNum] = bearing[3 * i];
        c->normals[3 * tNum + 1] = bearing[3 * i + 1];
        c->normals[3 * tNum + 2] = bearing[3 * i + 2];
        tNum++;
    }
    c->n = tNum;

    for (int j = 0; j < tNum; ++j) {
        /* :117-119  X = bearing * || P_w - C_ref || */
        double dx_ = depth_pts[3 * j] - ref_cnt[0];
        double dy_ = depth_pts[3 * j + 1] - ref_cnt[1];
        double dz_ = depth_pts[3 * j + 2] - ref_cnt[2];
        double depth = sqrt(dx_ * dx_ + dy_ * dy_ + dz_ * dz_);
        c->normals[3 * j] *= depth; c->normals[3 * j + 1] *= depth; c->normals[3 * j + 2] *= depth;
        if (MUT(MUT_Q6_TREF)) oracle_se3_act(T_ref, depth_pts + 3 * j, c->normals + 3 * j);   /* "fixed": X = T_ref * P_w */

        /* :123-132 bilinear coefficients */
        const double u = pts[2 * j], v = pts[2 * j + 1];
        const int fu = (int)floor(u), fv = (int)floor(v);
        const double su = u - fu, sv = v - fv;
        const double omx = 1.0 - su, omy = 1.0 - sv;
        const double w00 = omx * omy, w01 = su * omy, w10 = omx * sv, w11 = su * sv;

        double Jt[12];
        oracle_jacobian_ba(c->normals + 3 * j, Jt);            /* :138 */

        int tNum1 = 0;
        for (int i = 0; i < 4; ++i) {                          /* :141-162 */
            const uint8_t* it = img + (fv - po + i) * tRefStep + (fu - po);
            for (int k = 0; k < 4; ++k, ++it, ++tNum1) {
                c->ref_patch[j * 16 + tNum1] =
                    w00 * it[0] + w01 * it[1] + w10 * it[tRefStep] + w11 * it[tRefStep + 1];
                double dx = 0.5 * ((w00 * it[1] + w01 * it[2] + w10 * it[tRefStep + 1] + w11 * it[tRefStep + 2]) -
                                   (w00 * it[-1] + w01 * it[0] + w10 * it[tRefStep - 1] + w11 * it[tRefStep]));
                double dy = 0.5 * ((w00 * it[tRefStep] + w01 * it[tRefStep + 1] + w10 * it[2 * tRefStep] + w11 * it[2 * tRefStep + 1]) -
                                   (w00 * it[-tRefStep] + w01 * it[-tRefStep + 1] + w10 * it[0] + w11 * it[1]));
                if (MUT(MUT_Q5_RAWGRAD)) {                     /* "fixed": central differences of the raw pixels */
                    dx = 0.5 * ((double)it[1] - (double)it[-1]);
                    dy = 0.5 * ((double)it[tRefStep] - (double)it[-tRefStep]);
                }
                double* Jrow = c->jac + ((size_t)j * 16 + tNum1) * 6;
                for (int q = 0; q < 6; ++q)                    /* :160 */
                    Jrow[q] = (dx * Jt[q] + dy * Jt[6 + q]) * tFocalth * tScale;
            }
        }
    }
    free(pts);
    free(depth_pts);
}

/* ComputeResiduals — src/Sprase_ImageAlign.cpp:240-299 (linearSystem == true) */
static double compute_residuals(const oracle_se3* T, const dsdtm_pyramid* cur, const dsdtm_camera* cam,
                                int level, const level_cache* c, double H[36], double JRes[6], int* tnPts) {
    const uint8_t* img = cur->data[level];
    const int cols = cur->width[level], rows = cur->height[level];
    const float tScale = (float)(1.0 / (1 << level));          /* :244 */
    const int mnboarder = MUT(MUT_Q3_TIGHT) ? 2 : 4 - 1;       /* :245 */
    const int po = MUT(MUT_Q4_SHIFT) ? 1 : 2;                  /* :278 */
    const int tStep = cur->stride[level];                      /* :272 */
    double chi2 = 0.0;
    int tResNum = 0;
    *tnPts = 0;
    for (int n = 0; n < c->n; ++n) {
        double p[3];
        oracle_se3_act(T, c->normals + 3 * n, p);              /* :254 */
        /* Camera2Pixel (src/Camera.cpp:167-171): float intrinsics promoted */
        const double u = ((double)cam->fx * p[0] / p[2] + (double)cam->cx) * tScale;
        const double v = ((double)cam->fy * p[1] / p[2] + (double)cam->cy) * tScale;
        const int u_i = (int)floor(u);
        const int v_i = (int)floor(v);
        if (u_i < 0 || v_i < 0 || u_i - mnboarder < 0 || v_i - mnboarder < 0 ||
            u_i + mnboarder >= cols || v_i + mnboarder >= rows)
            continue;                                          /* :262 */
        if (MUT(MUT_Q3_ZTEST) && !(p[2] > 0.0)) continue;      /* "fixed": points behind the camera are not visible */
        /* NaN pixel coordinates: floor(NaN)->int is UB in the reference; treat as not visible */
        if (!(u == u) || !(v == v)) continue;
        const double su = u - u_i, sv = v - v_i;
        const double tl = (1.0 - su) * (1.0 - sv);
        const double trw = su * (1.0 - sv);
        const double bl = (1.0 - su) * sv;
        const double br = su * sv;
        const int tPtnum = n * 16;
        int tNum = 0;
        for (int i = 0; i < 4; ++i) {
            /* :278 — rows indexed with `cols`, the +1 row neighbour with `step` (quirk Q7) */
            const uint8_t* it = img + (v_i + i - po) * cols + u_i - po;
            for (int j = 0; j < 4; ++j, ++it, ++tNum) {
                double tCurPx = tl * it[0] + trw * it[1] + bl * it[tStep] + br * it[tStep + 1];
                double res = -(c->ref_patch[tPtnum + tNum] - tCurPx);
                chi2 += res * res;
                tResNum++;
                const double* J = c->jac + ((size_t)tPtnum + tNum) * 6;
                for (int a = 0; a < 6; ++a) {
                    for (int b = 0; b < 6; ++b) H[a * 6 + b] += J[a] * J[b];   /* :290 */
                    JRes[a] += J[a] * res;                                      /* :291 */
                }
            }
        }
        (*tnPts)++;
    }
    if (MUT(MUT_Q9_SUM)) return chi2;                          /* "fixed": not normalised by the visible pixel count */
    if (MUT(MUT_Q11_ZERO) && tResNum == 0) return 0.0;         /* "fixed": no 0/0 when nothing is visible */
    return chi2 / tResNum;                                     /* :298 (0/0 -> NaN) */
}

/* GaussNewtonSolver — src/Sprase_ImageAlign.cpp:301-344 */
static void gauss_newton(oracle_se3* T, const dsdtm_pyramid* cur, const dsdtm_camera* cam, int level,
                         const level_cache* c, int max_iters, int* tnPts, dsdtm_align_stats* stats) {
    int stop = 0;
    const double eps = MUT(MUT_Q9_EPS) ? 1e-6 : 1e-8;
    double chi2 = 0.0;
    oracle_se3 Told = *T;
    int iters = 0, exit_code = 0;
    for (int i = 0; i < max_iters; ++i) {
        double H[36], JRes[6], x[6];
        memset(H, 0, sizeof H);
        memset(JRes, 0, sizeof JRes);
        double chi2New = compute_residuals(T, cur, cam, level, c, H, JRes, tnPts);
        iters++;
        oracle_ldlt6_solve(H, JRes, x);                        /* :318 */
        if (isnan(x[0])) stop = 1;                             /* :321-326 */
        if ((i > 0 && (MUT(MUT_Q9_GE) ? chi2New >= chi2 : chi2New > chi2)) || stop) {   /* :328-332 */
            *T = Told;
            exit_code = stop ? 3 : 1;
            break;
        }
        oracle_se3 dT, Tnew;
        oracle_se3_exp(x, &dT);
        if (MUT(MUT_Q8_LEFT)) oracle_se3_mul(&dT, T, &Tnew);   /* "fixed": left-multiplied update */
        else
        oracle_se3_mul(T, &dT, &Tnew);                         /* :335 right-multiply */
        Told = *T;
        *T = Tnew;
        chi2 = chi2New;
        double mx = 0.0;
        for (int a = 0; a < 6; ++a) if (fabs(x[a]) > mx) mx = fabs(x[a]);
        if (mx <= eps) { exit_code = 2; break; }               /* :341 */
    }
    if (stats) {
        stats->iters[level] = iters;
        stats->n_ref[level] = c->n;
        stats->n_vis[level] = *tnPts;
        stats->exit_code[level] = exit_code;
        stats->chi2[level] = chi2;
    }
}

/* Run — src/Sprase_ImageAlign.cpp:29-60 */
int oracle_sparse_align(const dsdtm_pyramid* ref, const dsdtm_pyramid* cur, const dsdtm_camera* cam,
                        const float* px_xy, const double* bearing, const double* p_world,
                        const uint8_t* initial, int n_features,
                        const double T_ref_w[12], double T_cur_w[12],
                        const dsdtm_align_params* prm, int* n_tracked, dsdtm_align_stats* stats) {
    if (!ref || !cur || !cam || !T_ref_w || !T_cur_w || !prm || !n_tracked || n_features < 0)
        return DSDTM_ERR_INVALID;
    if (prm->max_level > ref->levels || prm->max_level > cur->levels || prm->min_level < 0 ||
        prm->max_level > DSDTM_MAX_LEVELS)
        return DSDTM_ERR_INVALID;
    if (stats) memset(stats, 0, sizeof *stats);
    *n_tracked = 0;
    if (MUT(MUT_Q10_MINFTS)) {                                 /* "fixed": only initialised features count */
        int ni = 0;
        for (int i = 0; i < n_features; ++i) ni += initial[i] ? 1 : 0;
        if (ni < prm->min_fts) return DSDTM_OK;
    }
    if (n_features < prm->min_fts) return DSDTM_OK;            /* :34-38 "Too few features" */

    oracle_se3 Tc, Tr, TrInv, T;
    oracle_se3_from_rt(T_cur_w, &Tc);
    oracle_se3_from_rt(T_ref_w, &Tr);
    oracle_se3_inverse(&Tr, &TrInv);
    oracle_se3_mul(&Tc, &TrInv, &T);                           /* :43 */
    /* Frame::Set_Pose (src/Frame.cpp:167-174): mOw = T_cw.inverse().translation() */
    const double ref_cnt[3] = {TrInv.t[0], TrInv.t[1], TrInv.t[2]};

    level_cache c;
    size_t nf = (size_t)(n_features > 0 ? n_features : 1);
    c.ref_patch = (double*)malloc(sizeof(double) * 16 * nf);
    c.jac = (double*)malloc(sizeof(double) * 96 * nf);
    c.normals = (double*)malloc(sizeof(double) * 3 * nf);
    if (!c.ref_patch || !c.jac || !c.normals) { free(c.ref_patch); free(c.jac); free(c.normals); return DSDTM_ERR_NOMEM; }

    int mnPts = 0;
    for (int lvl = prm->max_level - 1; lvl >= prm->min_level; --lvl) {      /* :45-55 */
        get_jacobian_mat(ref, cam, px_xy, bearing, p_world, initial, n_features, ref_cnt, &Tr, lvl, &c);
        int lvlPts = 0;
        gauss_newton(&T, cur, cam, lvl, &c, prm->max_iters, &lvlPts, stats);
        if (!(MUT(MUT_Q10_COARSE) && lvl != prm->max_level - 1)) mnPts = lvlPts;   /* "fixed": the coarsest level's count */
    }
    oracle_se3 Tout;
    oracle_se3_mul(&T, &Tr, &Tout);                            /* :57 */
    oracle_se3_to_rt(&Tout, T_cur_w);
    *n_tracked = mnPts;                                        /* :59 */
    free(c.ref_patch); free(c.jac); free(c.normals);
    return DSDTM_OK;
}

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

double oracle_sparse_align_batch_timed(const dsdtm_batch_desc* b, const dsdtm_camera* cam,
                                       const dsdtm_align_params* prm, int n_threads) {
    (void)n_threads;
    double t0 = now_s();
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads > 0 ? n_threads : 1)
#endif
    for (int i = 0; i < b->n_pairs; ++i) {
        dsdtm_pyramid ref, cur;
        ref.levels = cur.levels = b->levels;
        for (int l = 0; l < b->levels; ++l) {
            ref.data[l] = b->ref_pyr + (size_t)i * b->pyr_pitch + b->level_offset[l];
            cur.data[l] = b->cur_pyr + (size_t)i * b->pyr_pitch + b->level_offset[l];
            ref.width[l] = cur.width[l] = b->width[l];
            ref.height[l] = cur.height[l] = b->height[l];
            ref.stride[l] = cur.stride[l] = b->stride[l];
        }
        int nf = b->n_features ? b->n_features[i] : b->max_features;
        size_t fo = (size_t)i * b->max_features;
        int nt = 0;
        oracle_sparse_align(&ref, &cur, cam, b->px_xy + 2 * fo, b->bearing + 3 * fo, b->p_world + 3 * fo,
                            b->initial + fo, nf, b->T_ref_w + 12 * (size_t)i, b->T_cur_w + 12 * (size_t)i,
                            prm, &nt, b->stats ? b->stats + i : NULL);
        b->n_tracked[i] = nt;
    }
    return now_s() - t0;
}

/* ===================================================================================
 * Feature_Alignment::Align2DGaussNewton — src/Feature_alignment.cpp:318-417
 * float32 throughout (Matrix3f, float u,v), with the reference's float/double mixing.
 * =================================================================================== */
static void mat3f_inverse(const float m[9], float inv[9]) {
    /* Eigen 3.2 compute_inverse<Matrix3f,Matrix3f,3>: cofactors + 1/det */
#define M(i, j) m[(i) * 3 + (j)]
#define COF(i, j) (M(((i) + 1) % 3, ((j) + 1) % 3) * M(((i) + 2) % 3, ((j) + 2) % 3) - \
                   M(((i) + 1) % 3, ((j) + 2) % 3) * M(((i) + 2) % 3, ((j) + 1) % 3))
    float c0 = COF(0, 0), c1 = COF(1, 0), c2 = COF(2, 0);
    /* det = cofactors_col0.cwiseProduct(matrix.col(0)).sum(); redux unroller: p0 + (p1 + p2) */
    float p0 = c0 * M(0, 0), p1 = c1 * M(1, 0), p2 = c2 * M(2, 0);
    float det = p0 + (p1 + p2);
    float invdet = 1.0f / det;
    inv[0] = c0 * invdet; inv[1] = c1 * invdet; inv[2] = c2 * invdet;
    inv[3] = COF(0, 1) * invdet; inv[4] = COF(1, 1) * invdet; inv[5] = COF(2, 1) * invdet;
    inv[6] = COF(0, 2) * invdet; inv[7] = COF(1, 2) * invdet; inv[8] = COF(2, 2) * invdet;
#undef COF
#undef M
}

/* Quirk A3: the bounds test (:367-368) admits u_r == cols-4 / v_r == rows-4, for which the
 * bilinear footprint reaches column `cols` (the next row's first byte) and row `rows`
 * (outside the cv::Mat: undefined in the reference). Defined here, and in the HIP kernel,
 * as: a byte whose offset is >= stride*height reads as 0. */
static inline float a2d_px(const uint8_t* img, long off, long size) {
    return (off >= 0 && off < size) ? (float)img[off] : 0.0f;
}

#ifdef ORACLE_MUTANTS
static int align2d_in_double(const uint8_t* img, int width, int height, int stride,
                             const uint8_t* border, const uint8_t* patch, int max_iters, double px[2]);
#endif

int oracle_align2d(const uint8_t* img, int width, int height, int stride,
                   const uint8_t* border, const uint8_t* patch, int max_iters, double px[2]) {
    enum { HP = 4, PS = 8, LPS = 10 };
#ifdef ORACLE_MUTANTS
    if (MUT(MUT_A1_DOUBLE)) return align2d_in_double(img, width, height, stride, border, patch, max_iters, px);
#endif
    const long img_size = (long)stride * height;
    float H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, Hinv[9];
    float dxs[PS * PS], dys[PS * PS];
    int n = 0;
    for (int l = 0; l < PS; ++l) {                             /* :330-343 */
        const uint8_t* it = border + (l + 1) * LPS + 1;
        for (int i = 0; i < PS; ++i, ++it, ++n) {
            float J[3];
            J[0] = (float)(0.5 * (it[1] - it[-1]));
            J[1] = (float)(0.5 * (it[LPS] - it[-LPS]));
            J[2] = 1.0f;
            dxs[n] = J[0];
            dys[n] = J[1];
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) H[a * 3 + b] += J[a] * J[b];
        }
    }
    mat3f_inverse(H, Hinv);                                    /* :345 */
    if (MUT(MUT_A2_CHECK) && !(fabsf(Hinv[0]) < 1e30f)) return 0;   /* "fixed": a singular H is refused before px is touched */
    float mean_diff = 0.0f;
    float u = (float)px[0];
    float v = (float)px[1];
    const float min_update_squared = (float)(0.03 * 0.03);     /* :352 */
    int converged = 0;
    for (int it_n = 0; it_n < max_iters; ++it_n) {
        /* :365-369; floor(NaN)->int is UB in the reference, the isnan test is what matters */
        if (isnan(u) || isnan(v)) break;
        int u_r = (int)floor(u);
        int v_r = (int)floor(v);
        if (u_r < HP || v_r < HP || u_r > width - HP || v_r > height - HP) break;
        if (MUT(MUT_A3_STRICT) && (u_r >= width - HP || v_r >= height - HP)) break;   /* "fixed": footprint inside the image */
        float subpix_x = u - u_r;
        float subpix_y = v - v_r;
        float wTL = (float)((1.0 - subpix_x) * (1.0 - subpix_y));  /* double arithmetic, :373 */
        float wTR = subpix_x * (1 - subpix_y);                       /* float arithmetic,  :374 */
        float wBL = (float)((1.0 - subpix_x) * subpix_y);          /* double arithmetic, :375 */
        float wBR = subpix_x * subpix_y;                             /* :376 */
        float Jres[3] = {0, 0, 0};
        const uint8_t* it_ref = patch;
        int q = 0;
        for (int j = 0; j < PS; ++j) {
            long it = (long)(v_r + j - HP) * stride + u_r - HP;             /* :383 */
            float row[3] = {0, 0, 0};                                       /* only used by MUT_A1_ROWSUMS */
            for (int k = 0; k < PS; ++k, ++it, ++it_ref, ++q) {
                float tSearchPx = wTL * a2d_px(img, it, img_size) + wTR * a2d_px(img, it + 1, img_size) +
                                  wBL * a2d_px(img, it + stride, img_size) + wBR * a2d_px(img, it + stride + 1, img_size);
                float tRes = tSearchPx - *it_ref + (MUT(MUT_A4_NOMEAN) ? 0.0f : mean_diff);
                if (MUT(MUT_A1_ROWSUMS)) {       /* "parallelised": per-row partial sums, then the rows (float is not associative) */
                    row[0] -= tRes * dxs[q]; row[1] -= tRes * dys[q]; row[2] -= tRes;
                    continue;
                }
                Jres[0] -= tRes * dxs[q];
                Jres[1] -= tRes * dys[q];
                Jres[2] -= tRes;
            }
            if (MUT(MUT_A1_ROWSUMS)) { Jres[0] += row[0]; Jres[1] += row[1]; Jres[2] += row[2]; }
        }
        float upd[3];
        for (int a = 0; a < 3; ++a)                            /* :395 Hinv*Jres */
            upd[a] = (Hinv[a * 3] * Jres[0] + Hinv[a * 3 + 1] * Jres[1]) + Hinv[a * 3 + 2] * Jres[2];
        u += upd[0];
        v += upd[1];
        mean_diff += upd[2];
        if (upd[0] * upd[0] + upd[1] * upd[1] < min_update_squared) { converged = 1; break; }
    }
    if (MUT(MUT_A4_NOWRITE) && !converged) return converged;   /* "fixed": px untouched on failure */
    px[0] = u;                                                 /* :414 written back always */
    px[1] = v;
    return converged;
}

#ifdef ORACLE_MUTANTS
/* MUT_A1_DOUBLE: the same algorithm with every float of the reference promoted to double (quirk A1 "fixed") */
static int align2d_in_double(const uint8_t* img, int width, int height, int stride,
                             const uint8_t* border, const uint8_t* patch, int max_iters, double px[2]) {
    enum { HP = 4, PS = 8, LPS = 10 };
    const long img_size = (long)stride * height;
    double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, Hinv[9];
    double dxs[PS * PS], dys[PS * PS];
    int n = 0;
    for (int l = 0; l < PS; ++l) {
        const uint8_t* it = border + (l + 1) * LPS + 1;
        for (int i = 0; i < PS; ++i, ++it, ++n) {
            double J[3] = {0.5 * (it[1] - it[-1]), 0.5 * (it[LPS] - it[-LPS]), 1.0};
            dxs[n] = J[0];
            dys[n] = J[1];
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) H[a * 3 + b] += J[a] * J[b];
        }
    }
    {   /* cofactor inverse, as mat3f_inverse */
        double c0 = H[4] * H[8] - H[5] * H[7], c1 = H[7] * H[2] - H[8] * H[1], c2 = H[1] * H[5] - H[2] * H[4];
        double invdet = 1.0 / (c0 * H[0] + (c1 * H[3] + c2 * H[6]));
        Hinv[0] = c0 * invdet; Hinv[1] = c1 * invdet; Hinv[2] = c2 * invdet;
        Hinv[3] = (H[5] * H[6] - H[3] * H[8]) * invdet; Hinv[4] = (H[8] * H[0] - H[6] * H[2]) * invdet; Hinv[5] = (H[2] * H[3] - H[0] * H[5]) * invdet;
        Hinv[6] = (H[3] * H[7] - H[4] * H[6]) * invdet; Hinv[7] = (H[6] * H[1] - H[7] * H[0]) * invdet; Hinv[8] = (H[0] * H[4] - H[1] * H[3]) * invdet;
    }
    double mean_diff = 0.0, u = px[0], v = px[1];
    int converged = 0;
    for (int it_n = 0; it_n < max_iters; ++it_n) {
        if (isnan(u) || isnan(v)) break;
        int u_r = (int)floor(u), v_r = (int)floor(v);
        if (u_r < HP || v_r < HP || u_r > width - HP || v_r > height - HP) break;
        double sx = u - u_r, sy = v - v_r;
        double wTL = (1.0 - sx) * (1.0 - sy), wTR = sx * (1.0 - sy), wBL = (1.0 - sx) * sy, wBR = sx * sy;
        double Jres[3] = {0, 0, 0};
        const uint8_t* it_ref = patch;
        int q = 0;
        for (int j = 0; j < PS; ++j) {
            long it = (long)(v_r + j - HP) * stride + u_r - HP;
            for (int k = 0; k < PS; ++k, ++it, ++it_ref, ++q) {
                double s = wTL * a2d_px(img, it, img_size) + wTR * a2d_px(img, it + 1, img_size) +
                           wBL * a2d_px(img, it + stride, img_size) + wBR * a2d_px(img, it + stride + 1, img_size);
                double r = s - *it_ref + mean_diff;
                Jres[0] -= r * dxs[q]; Jres[1] -= r * dys[q]; Jres[2] -= r;
            }
        }
        double upd[3];
        for (int a = 0; a < 3; ++a) upd[a] = (Hinv[a * 3] * Jres[0] + Hinv[a * 3 + 1] * Jres[1]) + Hinv[a * 3 + 2] * Jres[2];
        u += upd[0]; v += upd[1]; mean_diff += upd[2];
        if (upd[0] * upd[0] + upd[1] * upd[1] < 0.03 * 0.03) { converged = 1; break; }
    }
    px[0] = u; px[1] = v;
    return converged;
}
#endif

int oracle_align2d_batch(const dsdtm_pyramid* cur, const uint8_t* patch_border, const uint8_t* patch,
                         const int32_t* level, double* px_xy, uint8_t* converged, int max_iters, int m) {
    if (!cur || !patch_border || !patch || !level || !px_xy || !converged) return DSDTM_ERR_INVALID;
    for (int i = 0; i < m; ++i) {
        int l = level[i];
        if (l < 0 || l >= cur->levels) return DSDTM_ERR_INVALID;
        converged[i] = (uint8_t)oracle_align2d(cur->data[l], cur->width[l], cur->height[l], cur->stride[l],
                                               patch_border + (size_t)i * 100, patch + (size_t)i * 64,
                                               max_iters, px_xy + 2 * (size_t)i);
    }
    return DSDTM_OK;
}

/* ===================================================================================
 * cv::pyrDown for CV_8UC1 (OpenCV 2.4.13 modules/imgproc/src/pyramids.cpp, pyrDown_<
 * FixPtCast<uchar,8>, ...>): separable [1 4 6 4 1], BORDER_REFLECT_101, (s+128)>>8.
 * Call site: src/Frame.cpp:79.
 * =================================================================================== */
static int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = MUT(MUT_PD_BORDER) ? -p - 1 : -p;
        else p = MUT(MUT_PD_BORDER) ? 2 * len - 1 - p : 2 * len - 2 - p;
    }
    return p;
}

void oracle_pyrdown(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride) {
    const int dw = (w + 1) / 2, dh = (h + 1) / 2;
    int* rowbuf = (int*)malloc(sizeof(int) * (size_t)dw * 5);
    for (int y = 0; y < dh; ++y) {
        for (int k = 0; k < 5; ++k) {
            int sy = reflect101(2 * y + k - 2, h);
            const uint8_t* s = src + (size_t)sy * sstride;
            int* r = rowbuf + (size_t)k * dw;
            for (int x = 0; x < dw; ++x) {
                int x0 = reflect101(2 * x - 2, w), x1 = reflect101(2 * x - 1, w), x2 = reflect101(2 * x, w);
                int x3 = reflect101(2 * x + 1, w), x4 = reflect101(2 * x + 2, w);
                r[x] = s[x2] * 6 + (s[x1] + s[x3]) * 4 + s[x0] + s[x4];
            }
        }
        for (int x = 0; x < dw; ++x) {
            int v = rowbuf[2 * dw + x] * 6 + (rowbuf[dw + x] + rowbuf[3 * dw + x]) * 4 + rowbuf[x] + rowbuf[4 * dw + x];
            dst[(size_t)y * dstride + x] = (uint8_t)((v + (MUT(MUT_PD_ROUND) ? 127 : 128)) >> 8);
        }
    }
    free(rowbuf);
}

/* ===================================================================================
 * Warp prelude of FindMatchDirect — src/Feature_alignment.cpp:160-275
 * =================================================================================== */
static void pixel2camera_d(const dsdtm_camera* cam, double px, double py, float depth, double out[3]) {
    /* Camera::Pixel2Camera(const Eigen::Vector2d&, const float&) — src/Camera.cpp:180-185 */
    out[0] = depth * (px - cam->cx) / cam->fx;
    out[1] = depth * (py - cam->cy) / cam->fy;
    out[2] = depth;
}

static void camera2pixel(const dsdtm_camera* cam, const double p[3], double out[2]) {
    out[0] = cam->fx * p[0] / p[2] + cam->cx;                  /* src/Camera.cpp:167-171 */
    out[1] = cam->fy * p[1] / p[2] + cam->cy;
}

static void normalize3(double v[3]) {
    double n = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    v[0] /= n; v[1] /= n; v[2] /= n;
}

int oracle_warp_patches(const dsdtm_pyramid* kf_pyr, int n_kf, const dsdtm_camera* cam,
                        const double* T_kf_w, const double T_cur_w[12],
                        const int32_t* cand_kf, const float* ref_px, const int32_t* ref_level,
                        const double* ref_bearing, const double* p_world,
                        int max_search_level, int m,
                        double* affine, int32_t* search_level, uint8_t* patch_border, uint8_t* patch) {
    if (!kf_pyr || !cam || !T_kf_w || !T_cur_w) return DSDTM_ERR_INVALID;
    oracle_se3 Tcur;
    oracle_se3_from_rt(T_cur_w, &Tcur);
    for (int c = 0; c < m; ++c) {
        const int k = cand_kf[c];
        if (k < 0 || k >= n_kf) return DSDTM_ERR_INVALID;
        const int tLevel = ref_level[c];
        if (tLevel < 0 || tLevel >= kf_pyr[k].levels) return DSDTM_ERR_INVALID;
        oracle_se3 Tkf, TkfInv, T_c2r;
        oracle_se3_from_rt(T_kf_w + 12 * (size_t)k, &Tkf);
        oracle_se3_inverse(&Tkf, &TkfInv);
        /* --- SolveAffineMatrix :160-190 --- */
        const int Half_PatchLarger = 4 + 1;
        const double* P = p_world + 3 * (size_t)c;
        const double* nb = ref_bearing + 3 * (size_t)c;
        double dx_ = TkfInv.t[0] - P[0], dy_ = TkfInv.t[1] - P[1], dz_ = TkfInv.t[2] - P[2];
        double dist = sqrt(dx_ * dx_ + dy_ * dy_ + dz_ * dz_);
        double tRefPoint[3] = {dist * nb[0], dist * nb[1], dist * nb[2]};           /* :167 */
        const float rx = ref_px[2 * c], ry = ref_px[2 * c + 1];
        /* Eigen::Vector2d tRefPxU(tRefPx.x + Half_PatchLarger*(1<<tLevel), tRefPx.y): float + int -> float -> double */
        double pxU[2] = {(double)(rx + (float)(Half_PatchLarger * (1 << tLevel))), (double)ry};
        double pxV[2] = {(double)rx, (double)(ry + (float)(Half_PatchLarger * (1 << tLevel)))};
        double pU[3], pV[3];
        pixel2camera_d(cam, pxU[0], pxU[1], 1.0f, pU);
        pixel2camera_d(cam, pxV[0], pxV[1], 1.0f, pV);
        normalize3(pU);
        normalize3(pV);
        double sU = tRefPoint[2] / pU[2], sV = tRefPoint[2] / pV[2];
        for (int a = 0; a < 3; ++a) { pU[a] *= sU; pV[a] *= sV; }
        oracle_se3_mul(&Tcur, &TkfInv, &T_c2r);                                      /* :181 */
        double q0[3], qU[3], qV[3], c0[2], cU[2], cV[2];
        oracle_se3_act(&T_c2r, tRefPoint, q0);
        oracle_se3_act(&T_c2r, pU, qU);
        oracle_se3_act(&T_c2r, pV, qV);
        camera2pixel(cam, q0, c0);
        camera2pixel(cam, qU, cU);
        camera2pixel(cam, qV, cV);
        double A[4]; /* row-major 2x2; col(0) = (cU-c0)/5, col(1) = (cV-c0)/5 */
        A[0] = (cU[0] - c0[0]) / Half_PatchLarger; A[2] = (cU[1] - c0[1]) / Half_PatchLarger;
        A[1] = (cV[0] - c0[0]) / Half_PatchLarger; A[3] = (cV[1] - c0[1]) / Half_PatchLarger;
        if (affine) memcpy(affine + 4 * (size_t)c, A, sizeof A);
        /* --- GetBestSearchLevel :192-204 --- */
        int tSearch_Level = 0;
        double D = A[0] * A[3] - A[1] * A[2];
        while (D > (MUT(MUT_A13_DET) ? 2.0 : 3.0) && tSearch_Level < max_search_level) { tSearch_Level++; D = D * 0.25; }
        search_level[c] = tSearch_Level;
        /* --- WarpAffine :206-259 --- */
        /* Matrix2d::inverse(): Eigen compute_inverse size 2: invdet = 1/det; [d -b; -c a]*invdet */
        double det = A[0] * A[3] - A[1] * A[2];
        double invdet = 1.0 / det;
        float Ai[4] = {(float)(A[3] * invdet), (float)(-A[1] * invdet),
                       (float)(-A[2] * invdet), (float)(A[0] * invdet)};
        const dsdtm_pyramid* pyr = &kf_pyr[k];
        const uint8_t* img = pyr->data[tLevel];
        const int cols = pyr->width[tLevel], rows = pyr->height[tLevel], tStep = pyr->stride[tLevel];
        float refx = rx / (float)(1 << tLevel), refy = ry / (float)(1 << tLevel);   /* :215-216 */
        if (MUT(MUT_W2_REFLEVEL)) { refx = rx; refy = ry; }    /* "fixed" the other way: level-0 pixel on the level image */
        const int int_scale = 1 / (1 << tSearch_Level);       /* :231 integer division bug W1 */
        uint8_t* out = patch_border + 100 * (size_t)c;
        int j = 0;
        for (int iy = -5; iy < 5; ++iy) {
            for (int ix = -5; ix < 5; ++ix, ++j) {
                /* tWrapMat = tA_r2c*tWrapMat*(1/(1<<lvl)): (A*g) then * float(int) */
                const float wscale = MUT(MUT_W1_FLOATDIV) ? 1.0f / (float)(1 << tSearch_Level) : (float)int_scale;
                float gx = (Ai[0] * (float)ix + Ai[1] * (float)iy) * wscale;
                float gy = (Ai[2] * (float)ix + Ai[3] * (float)iy) * wscale;
                float wx = gx + refx, wy = gy + refy;          /* :232 */
                int fx_ = (int)floor((double)wx), fy_ = (int)floor((double)wy);   /* Eigenfloor(double) */
                float sx = wx - (float)fx_, sy = wy - (float)fy_;
                float omx = 1.0f - sx, omy = 1.0f - sy;
                float w00 = omx * omy;
                float w01 = omx * sy;                          /* :242 */
                float w10 = sx * omy;                          /* :243 */
                float w11 = 1.0f - w00 - w01 - w10;            /* :244 */
                if (!(wx == wx) || !(wy == wy) || wx < 0 || wy < 0 || wx > cols - 1 || wy > rows - 1) {
                    out[j] = 0;                                /* :249-250 (NaN: UB in ref; 0 here) */
                } else {
                    /* wx == cols-1 / wy == rows-1 pass the test with a zero weight on the neighbour
                     * that lies outside; read it as 0 (same value, no out-of-bounds access) */
                    const long sz = (long)tStep * rows, o = (long)tStep * fy_ + fx_;
                    float p00 = img[o], p01 = (o + tStep < sz) ? img[o + tStep] : 0.0f;
                    float p10 = (o + 1 < sz) ? img[o + 1] : 0.0f, p11 = (o + tStep + 1 < sz) ? img[o + tStep + 1] : 0.0f;
                    float val = w00 * p00 + w01 * p01 + w10 * p10 + w11 * p11;
                    out[j] = MUT(MUT_W2_ROUND) ? (uint8_t)(val + 0.5f) : (uint8_t)val;   /* float -> uchar truncation :254 */
                }
            }
        }
        /* --- GetPatchNoBoarder :261-275 --- */
        uint8_t* pn = patch + 64 * (size_t)c;
        for (int i = 1; i < 9; ++i)
            for (int jj = 0; jj < 8; ++jj) pn[(i - 1) * 8 + jj] = out[i * 10 + 1 + jj];
    }
    return DSDTM_OK;
}


/* ===================================================================================
 * Feature_detector::detect — reference src/Feature_detection.cpp:69-154 and the vendored
 * Thirdparty/fast it calls (:79-92). SURVEY §8(f)4.
 *
 * The FAST part is a restatement of the *published definition* the vendored, mechanically
 * generated sources implement (fast_10.cpp / faster_corner_10_sse.cpp: a pixel is a corner at
 * barrier b iff 10 contiguous pixels of the 16-pixel Bresenham circle of radius 3 are all
 * > p + b or all < p - b; fast_10_score.cpp: the largest such b; nonmax_3x3.cpp: suppressed iff
 * one of the 8 neighbours is a corner with score >= its own). Unlike the rest of this file it IS
 * pinned: Thirdparty/fast has no external dependency, `make -C oracle ref` compiles the reference's
 * own sources into oracle/_ref/libfast_ref.so, and tests/test_detector_cpu.py holds this
 * restatement to it corner by corner (and to committed fixtures where the reference is absent).
 * =================================================================================== */
static const int FAST_RING[16][2] = {   /* (dx, dy) in the order of fast_10_score.cpp:3146-3163 */
    {0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1}, {2, -2}, {1, -3},
    {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

/* max over the 16 arcs of 10 contiguous ring pixels of the smallest difference on the arc, for the
 * brighter (+) and the darker (-) polarity: the pixel is a corner at barrier b iff that value > b */
static int fast10_margin(const uint8_t* p, int stride) {
    int d[16];
    const int c = *p;
    for (int i = 0; i < 16; ++i) d[i] = (int)p[FAST_RING[i][1] * stride + FAST_RING[i][0]] - c;
    int best = -256;
    for (int s = 0; s < 16; ++s) {
        int mb = 256, md = 256;
        for (int k = 0; k < 10; ++k) {
            const int v = d[(s + k) & 15];
            if (v < mb) mb = v;
            if (-v < md) md = -v;
        }
        if (mb > best) best = mb;
        if (md > best) best = md;
    }
    return best;
}

void oracle_fast10(const uint8_t* img, int w, int h, int stride, int barrier, uint8_t* score, uint8_t* keep) {
    memset(score, 0, (size_t)w * h);
    memset(keep, 0, (size_t)w * h);
    /* faster_corner_10_sse.cpp:187-203: plain detector below 22 columns, nothing below 7 rows; both
     * scan rows 3..h-4 and columns 3..w-4 */
    if (h < 7 || w < 7) return;
    for (int y = 3; y < h - 3; ++y)
        for (int x = 3; x < w - 3; ++x) {
            const int m = fast10_margin(img + (size_t)y * stride + x, stride);
            if (m > barrier) score[(size_t)y * w + x] = (uint8_t)(MUT(MUT_D_SCORE) ? m : m - 1);     /* fast_10_score.cpp:22-3140 */
        }
    /* nonmax_3x3.cpp:47-106: every comparison is `neighbour score >= own score` */
    for (int y = 3; y < h - 3; ++y)
        for (int x = 3; x < w - 3; ++x) {
            const int s = score[(size_t)y * w + x];
            if (!s) continue;
            int sup = 0;
            for (int dy = -1; dy <= 1 && !sup; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    if (!dx && !dy) continue;
                    const int n = score[(size_t)(y + dy) * w + (x + dx)];
                    if (n && (MUT(MUT_D_NMS_TIE) ? n > s : n >= s)) { sup = 1; break; }
                }
            keep[(size_t)y * w + x] = (uint8_t)!sup;
        }
}

int oracle_fast10_list(const uint8_t* img, int w, int h, int stride, int barrier, int32_t* out, int cap) {
    uint8_t* score = (uint8_t*)malloc((size_t)w * h);
    uint8_t* keep = (uint8_t*)malloc((size_t)w * h);
    oracle_fast10(img, w, h, stride, barrier, score, keep);
    int n = 0;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            if (!score[(size_t)y * w + x]) continue;
            if (n < cap) {
                out[4 * n] = x; out[4 * n + 1] = y; out[4 * n + 2] = score[(size_t)y * w + x];
                out[4 * n + 3] = keep[(size_t)y * w + x];
            }
            ++n;
        }
    free(score); free(keep);
    return n;
}

/* Feature_detector::shiTomasiScore, src/Feature_detection.cpp:157-198. The gradient sums are exact
 * in float (integers < 2^24); the divisions by 2.0*box_area are double divisions by 128 (exact);
 * the last line mixes float operands with the double constant 0.5 and calls the float overload of
 * sqrt that <cmath> puts in scope for a float argument (assumption: the reference translation unit
 * cannot be built here — OpenCV). */
float oracle_shi_tomasi(const uint8_t* img, int w, int h, int stride, int u, int v) {
    float dXX = 0.0f, dYY = 0.0f, dXY = 0.0f;
    const int halfbox_size = 4, box_size = MUT(MUT_D_BOX) ? 9 : 8, box_area = 64;
    const int x_min = u - halfbox_size, x_max = u + halfbox_size, y_min = v - halfbox_size, y_max = v + halfbox_size;
    if (x_min < 1 || x_max >= w - 1 || y_min < 1 || y_max >= h - 1) return 0.0f;      /* :173 */
    for (int y = y_min; y < (MUT(MUT_D_BOX) ? y_max + 1 : y_max); ++y) {
        const uint8_t* l = img + (size_t)stride * y + x_min - 1;
        const uint8_t* r = img + (size_t)stride * y + x_min + 1;
        const uint8_t* t = img + (size_t)stride * (y - 1) + x_min;
        const uint8_t* b = img + (size_t)stride * (y + 1) + x_min;
        for (int x = 0; x < box_size; ++x, ++l, ++r, ++t, ++b) {
            const float dx = (float)(*r - *l);
            const float dy = (float)(*b - *t);
            dXX += dx * dx; dYY += dy * dy; dXY += dx * dy;
        }
    }
    dXX = (float)(dXX / (2.0 * box_area));
    dYY = (float)(dYY / (2.0 * box_area));
    dXY = (float)(dXY / (2.0 * box_area));
    const float tr = dXX + dYY;
    const float disc = tr * tr - 4 * (dXX * dYY - dXY * dXY);
    return (float)(0.5 * (tr - sqrtf(disc)));
}

void oracle_detect_cells(const dsdtm_pyramid* pyr, int levels, int cell_size, int grid_cols, int grid_rows,
                         const uint8_t* grid_occupied, double detection_threshold, int barrier,
                         float* cell_score, int32_t* cell_x, int32_t* cell_y, int32_t* cell_level) {
    const int G = grid_cols * grid_rows;
    for (int k = 0; k < G; ++k) { cell_score[k] = (float)detection_threshold; cell_x[k] = 0; cell_y[k] = 0; cell_level[k] = 0; }   /* :74 */
    for (int L = 0; L < levels; ++L) {                                              /* :76 */
        const int scale = 1 << L, w = pyr->width[L], h = pyr->height[L], st = pyr->stride[L];
        const uint8_t* img = pyr->data[L];
        uint8_t* score = (uint8_t*)malloc((size_t)w * h);
        uint8_t* keep = (uint8_t*)malloc((size_t)w * h);
        oracle_fast10(img, w, h, st, barrier, score, keep);                         /* :79-92 */
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                if (!keep[(size_t)y * w + x]) continue;
                const int k = ((y * scale) / cell_size) * grid_cols + (x * scale) / cell_size;      /* :97-98 */
                if (k < 0 || k >= G) continue;
                if (grid_occupied && grid_occupied[k]) continue;                    /* :100 */
                const float sc = oracle_shi_tomasi(img, w, h, st, x, y);            /* :103 */
                if (MUT(MUT_D_CELLMAX) ? sc >= cell_score[k] : sc > cell_score[k]) {    /* :104-107 */
                    cell_score[k] = sc; cell_x[k] = x * scale; cell_y[k] = y * scale; cell_level[k] = L;
                }
            }
        free(score); free(keep);
    }
}
