/*
 * dsdtm_oracle.h — CPU restatement of the DSDTM sparse photometric alignment path.
 *
 * TEST INFRASTRUCTURE ONLY. Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and
 * only as the checker. The product (dsdtm_amd/csrc, include/dsdtm_amd.h) never links,
 * loads or calls it.
 *
 * PARITY UNPINNED by the reference: gaochq/DSDTM ships no golden vectors, known-answer
 * values or fixtures for this path, and the reference itself cannot
 * be built here (needs OpenCV 2.4, Eigen 3.2, Sophus, Ceres, glog, Pangolin, Boost —
 * none present; see oracle/Makefile `_ref` note). This restatement follows the reference
 * source line by line (citations at each function) and is cross-checked by an
 * independent numpy restatement and analytic ground truth in tests/.
 * EXCEPTION (pinned): the FAST-10 part of the feature detector (oracle_fast10*) — the reference
 * vendors Thirdparty/fast, which builds from its own sources (`make -C oracle ref` ->
 * oracle/_ref/libfast_ref.so); the restatement is held to that build and to vectors it produced
 * (tests/golden/fast_reference.npz, incl. the reference test's known answer of 167 corners).
 *
 * Third-party arithmetic restated from its published algorithm (not under /root/reference):
 *   Sophus (non-templated SE3/SO3; README.md:6 says "Sophus 1.0.0", version unpinned)
 *   Eigen 3.2.0 (LDLT with diagonal pivoting + pseudo-inverse solve; Matrix3f::inverse
 *                by cofactors; Quaternion product / _transformVector / toRotationMatrix)
 *   OpenCV 2.4.13 cv::pyrDown for CV_8UC1.
 *   Ceres Solver (version not stated; 1.13 semantics): trust-region LM of PoseOptimization, pose_opt_oracle.c.
 */
#ifndef DSDTM_ORACLE_H
#define DSDTM_ORACLE_H

#include "../include/dsdtm_amd.h" /* POD types only */

#ifdef __cplusplus
extern "C" {
#endif

/* Sprase_ImgAlign::Run — same argument meaning as dsdtm_sparse_align (no ctx).
 * Returns DSDTM_OK / DSDTM_ERR_INVALID. */
int oracle_sparse_align(const dsdtm_pyramid* ref, const dsdtm_pyramid* cur,
                        const dsdtm_camera* cam,
                        const float* px_xy, const double* bearing, const double* p_world,
                        const uint8_t* initial, int n_features,
                        const double T_ref_w[12], double T_cur_w[12],
                        const dsdtm_align_params* params,
                        int* n_tracked, dsdtm_align_stats* stats);

/* Feature_Alignment::Align2DGaussNewton for one feature. img/width/height/stride = the
 * cv::Mat. Returns the reference's bool. px is written back always (:414). */
int oracle_align2d(const uint8_t* img, int width, int height, int stride,
                   const uint8_t* patch_border, const uint8_t* patch,
                   int max_iters, double px[2]);

/* Batch form with the dsdtm_align2d_batch argument layout. */
int oracle_align2d_batch(const dsdtm_pyramid* cur, const uint8_t* patch_border,
                         const uint8_t* patch, const int32_t* level, double* px_xy,
                         uint8_t* converged, int max_iters, int m);

/* cv::pyrDown (8UC1) — one level. dst is ((w+1)/2) x ((h+1)/2). */
void oracle_pyrdown(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride);

/* Warp prelude, same layout as dsdtm_warp_patches. */
int oracle_warp_patches(const dsdtm_pyramid* kf_pyr, int n_kf, const dsdtm_camera* cam,
                        const double* T_kf_w, const double T_cur_w[12],
                        const int32_t* cand_kf, const float* ref_px, const int32_t* ref_level,
                        const double* ref_bearing, const double* p_world,
                        int max_search_level, int m,
                        double* affine, int32_t* search_level,
                        uint8_t* patch_border, uint8_t* patch);

/* --- Feature_detector::detect (src/Feature_detection.cpp:69-154), SURVEY §8(f)4 ---------------- */
/* FAST-10 corners of one 8-bit image as the reference obtains them (fast_corner_detect_10_sse2 +
 * fast_corner_score_10 + fast_nonmax_3x3, Thirdparty/fast): score[y*w+x] = 0 where (x,y) is not a
 * corner at `barrier`, else the corner score (largest barrier at which it still is one, >= barrier);
 * keep[y*w+x] = 1 where the corner survives the 3x3 non-maximum suppression. */
void oracle_fast10(const uint8_t* img, int w, int h, int stride, int barrier, uint8_t* score, uint8_t* keep);
/* the same as a list in the reference's raster order: (x, y, score, is_nonmax) quadruples; returns the count */
int oracle_fast10_list(const uint8_t* img, int w, int h, int stride, int barrier, int32_t* out, int cap);
/* Feature_detector::shiTomasiScore (:157-198) */
float oracle_shi_tomasi(const uint8_t* img, int w, int h, int stride, int u, int v);
/* The per-cell part of detect() (:75-104): best Shi-Tomasi corner of every grid cell over all pyramid
 * levels; cell_score[k] stays `detection_threshold` (x = y = level = 0) where no corner beats it. */
void oracle_detect_cells(const dsdtm_pyramid* pyr, int levels, int cell_size, int grid_cols, int grid_rows,
                         const uint8_t* grid_occupied, double detection_threshold, int barrier,
                         float* cell_score, int32_t* cell_x, int32_t* cell_y, int32_t* cell_level);

/* --- Optimizer::PoseOptimization (src/Optimizer.cpp:20-101), SURVEY §8(f)3: see pose_opt_oracle.c ---- */
/* Argument meaning of dsdtm_pose_optimization. linear_solver: 0 = Householder QR of [J; D] (what Ceres'
 * DENSE_QR does), 1 = Cholesky of the 6x6 normal equations (what the HIP kernel does).
 * trace (optional, 4 doubles per iteration incl. iteration 0, at most trace_cap iterations):
 * cost, trust-region radius, gradient max norm, successful steps so far. */
int oracle_pose_optimization(const double* bearing, const double* p_world, const int32_t* level,
                             const uint8_t* use, int n_features, double T_cur_w[12],
                             const dsdtm_pose_opt_params* prm, int linear_solver,
                             double* residual_norm, dsdtm_pose_opt_summary* summary,
                             double* trace, int trace_cap);
void oracle_so3_log(const double q[4], double w[3]);
void oracle_pose_plus(const double x[6], const double delta[6], double out[6]);
int oracle_chol6_solve(const double M[36], const double v[6], double y[6]);

/* --- building blocks exported for unit tests ------------------------------------ */
/* SE3 as Sophus stores it: unit quaternion (w,x,y,z) + translation. */
typedef struct oracle_se3 { double q[4]; double t[3]; } oracle_se3;
void oracle_se3_from_rt(const double T[12], oracle_se3* out);
void oracle_se3_to_rt(const oracle_se3* in, double T[12]);
void oracle_se3_exp(const double x[6], oracle_se3* out);
void oracle_se3_mul(const oracle_se3* a, const oracle_se3* b, oracle_se3* out);
void oracle_se3_inverse(const oracle_se3* a, oracle_se3* out);
void oracle_se3_act(const oracle_se3* a, const double p[3], double out[3]);
/* Eigen 3.2 LDLT solve of a 6x6 (row-major, full symmetric) system. */
void oracle_ldlt6_solve(const double H[36], const double b[6], double x[6]);
/* GetJocabianBA (src/Sprase_ImageAlign.cpp:169-193): 2x6 row-major. */
void oracle_jacobian_ba(const double p[3], double J[12]);
/* timing helper for the CPU baseline: runs oracle_sparse_align `reps` times over
 * `n_pairs` packed pairs (layout of dsdtm_batch_desc, host pointers), returns seconds. */
double oracle_sparse_align_batch_timed(const dsdtm_batch_desc* batch, const dsdtm_camera* cam,
                                       const dsdtm_align_params* params, int n_threads);

#ifdef __cplusplus
}
#endif
#endif
