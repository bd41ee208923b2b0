/*
 * dsdtm_amd.h — C ABI of the MI355X (gfx950) sparse photometric alignment path.
 *
 * This is the drop-in boundary for ONE hot path of gaochq/DSDTM:
 *   - DSDTM::Sprase_ImgAlign::Run            (reference src/Sprase_ImageAlign.cpp:29-60)
 *   - DSDTM::Feature_Alignment::Align2DGaussNewton (reference src/Feature_alignment.cpp:318-417)
 *   - the producers/consumers either side of it that SURVEY.md §8(f) marks "next":
 *     Frame::ComputeImagePyramid (src/Frame.cpp:74-81) and the warp prelude of
 *     Feature_Alignment::FindMatchDirect (src/Feature_alignment.cpp:128-275).
 *
 * The reference has no FFI layer: the boundary there is two C++ classes used by
 * Tracking through raw pointers (include/Tracking.h:149-150).  A maintainer binds these
 * entry points from thin adapter classes that keep the reference signatures
 * (INTEGRATION.md shows the adapter).  Everything here is POD: plain pointers and
 * sizes, caller-owned memory, `int` status returns, no exceptions, no globals.
 * Thread-compatible, not thread-safe: one dsdtm_ctx per calling thread (the
 * reference classes are only ever called from the tracking thread).
 *
 * All "host" entry points take host pointers, stage through pinned memory and run
 * the HIP kernels; all "_device" entry points take device pointers that are already
 * resident in HBM and only enqueue kernels on the given hipStream_t.
 * There is NO CPU fallback: without a usable gfx950 device every compute entry
 * point returns DSDTM_ERR_NO_DEVICE.
 */
#ifndef DSDTM_AMD_H
#define DSDTM_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSDTM_MAX_LEVELS 8

/* status codes (the reference has no error codes: Run returns 0 on "too few
 * features", src/Sprase_ImageAlign.cpp:34-38; that case is status DSDTM_OK with
 * *n_tracked == 0 and the pose untouched) */
enum {
    DSDTM_OK = 0,
    DSDTM_ERR_NO_DEVICE = -1, /* no HIP device / wrong arch / HIP runtime missing */
    DSDTM_ERR_INVALID = -2,   /* bad argument (null pointer, bad level range, ...) */
    DSDTM_ERR_HIP = -3,       /* a HIP call failed; see dsdtm_last_error()        */
    DSDTM_ERR_NOMEM = -4
};

/* Pinhole intrinsics. The reference stores them as *float* members and promotes
 * at use (include/Camera.h:138-142); `f` is the single focal `Camera.f` used for
 * the Jacobian scale (src/Sprase_ImageAlign.cpp:70,160) while fx,fy project
 * (src/Camera.cpp:167-171). width/height are the level-0 image size
 * (Camera::IsInImage, src/Camera.cpp:187-193). */
typedef struct dsdtm_camera {
    float fx, fy, cx, cy, f;
    int width, height;
} dsdtm_camera;

/* One image pyramid = Frame::mvImg_Pyr (include/Frame.h, src/Frame.cpp:74-81):
 * u8 single channel, level 0 first. stride in bytes (cv::Mat::step). */
typedef struct dsdtm_pyramid {
    int levels;
    const uint8_t* data[DSDTM_MAX_LEVELS];
    int width[DSDTM_MAX_LEVELS];
    int height[DSDTM_MAX_LEVELS];
    int stride[DSDTM_MAX_LEVELS];
} dsdtm_pyramid;

/* Constructor arguments of Sprase_ImgAlign(int tMaxLevel,int tMinLevel,int tMaxIterators)
 * (src/Sprase_ImageAlign.cpp:10-15) + Config "Camera.Min_fts" (:14).
 * Levels max_level-1 ... min_level are processed coarse to fine (:45). */
typedef struct dsdtm_align_params {
    int max_level;
    int min_level;
    int max_iters;
    int min_fts;
} dsdtm_align_params;

/* Optional per-alignment diagnostics (the reference only prints; these make the
 * executed-iteration counts and the accept/revert decisions testable). Index = pyramid level. */
typedef struct dsdtm_align_stats {
    int32_t iters[DSDTM_MAX_LEVELS];    /* executed ComputeResiduals calls at that level          */
    int32_t n_ref[DSDTM_MAX_LEVELS];    /* patches that passed the reference-side checks (:86-100) */
    int32_t n_vis[DSDTM_MAX_LEVELS];    /* visible patches in the last ComputeResiduals call       */
    int32_t exit_code[DSDTM_MAX_LEVELS];/* 0 cap reached, 1 chi2 increased -> revert, 2 |x|<=eps, 3 NaN guard */
    double chi2[DSDTM_MAX_LEVELS];      /* chi2 of the last accepted step at that level            */
} dsdtm_align_stats;

typedef struct dsdtm_ctx dsdtm_ctx;

/* ---- context ------------------------------------------------------------------- */
/* device = HIP device ordinal. Fails with DSDTM_ERR_NO_DEVICE when none is usable. */
int dsdtm_create(int device, dsdtm_ctx** out);
void dsdtm_destroy(dsdtm_ctx* ctx);
/* Last error text of this context (or of the failed dsdtm_create when ctx==NULL). */
const char* dsdtm_last_error(const dsdtm_ctx* ctx);
/* Library version / build info: "dsdtm_amd <ver> gfx950 ..." */
const char* dsdtm_version(void);
/* Number of visible HIP devices, or a negative status. Does not create a context. */
int dsdtm_device_count(void);

/* ---- Sprase_ImgAlign::Run ------------------------------------------------------ */
/*
 * Replaces: int Sprase_ImgAlign::Run(FramePtr tCurFrame, FramePtr tRefFrame)
 *           (include/Sprase_ImageAlign.h:29, src/Sprase_ImageAlign.cpp:29-60), with its
 *           callees GetJocabianMat :62-166, GetJocabianBA :169-193, ComputeResiduals
 *           :240-299, GaussNewtonSolver :301-344.
 *
 * ref/cur     : pyramids of tRefFrame / tCurFrame (host memory)
 * px_xy       : n_features x 2 float  — Feature::mpx        (include/Feature.h:19); any order (results do not
 *               depend on it beyond summation rounding; a spatially coherent order gathers ~1.5 % faster)
 * bearing     : n_features x 3 double — Feature::mNormal    (include/Feature.h:24)
 * p_world     : n_features x 3 double — Feature::Mpt->Get_Pose() (src/MapPoint.cpp:38-43);
 *               ignored (may be anything) where initial[i]==0
 * initial     : n_features x u8       — Feature::mbInitial  (include/Feature.h:23)
 * T_ref_w     : tRefFrame->Get_Pose() as 3x4 row-major [R|t] (world -> ref camera)
 * T_cur_w     : in: tCurFrame->Get_Pose() (the seed); out: the pose Run passes to
 *               tCurFrame->Set_Pose (:57). Untouched when *n_tracked==0 by the Min_fts rule.
 * n_tracked   : Run's return value (:59)
 * stats       : optional (may be NULL)
 *
 * Row strides (quirk Q7): the reference indexes the CURRENT image's base row with `cols` and its +1 row with
 * `step` (src/Sprase_ImageAlign.cpp:272,278,281) — the same thing for the continuous cv::Mat every producer of the reference
 * hands over (imread, pyrDown). This library indexes every row with `stride`: a pyramid with padded rows (stride > width) is
 * read CORRECTLY here, where the reference itself would read the wrong pixels; for stride == width the two are identical.
 * Host pyramids are repacked to stride == width on upload; the device entries accept padded rows as described.
 */
int dsdtm_sparse_align(dsdtm_ctx* ctx,
                       const dsdtm_pyramid* ref, const dsdtm_pyramid* cur,
                       const dsdtm_camera* cam,
                       const float* px_xy, const double* bearing, const double* p_world,
                       const uint8_t* initial, int n_features,
                       const double T_ref_w[12], double T_cur_w[12],
                       const dsdtm_align_params* params,
                       int* n_tracked, dsdtm_align_stats* stats);

/*
 * Batched, device-resident form: n_pairs independent frame pairs that share camera,
 * pyramid geometry and parameters (BASELINE config 4: embarrassingly parallel pairs).
 * Every pointer is a DEVICE pointer. Pyramids are packed: pair i's reference pyramid
 * starts at ref_pyr + i*pyr_pitch, level l at + level_offset[l], rows `stride[l]` bytes
 * apart. pyr_pitch and every level_offset must be multiples of 4 and each pyramid
 * allocation must extend to a multiple of 4 bytes (the kernels fetch aligned dwords).
 * Features are padded to `max_features` per pair; n_features[i] (device, may be NULL
 * => all pairs use max_features) gives the live count of pair i.
 */
typedef struct dsdtm_batch_desc {
    int n_pairs;
    int max_features;
    int levels;
    int width[DSDTM_MAX_LEVELS];
    int height[DSDTM_MAX_LEVELS];
    int stride[DSDTM_MAX_LEVELS];
    size_t level_offset[DSDTM_MAX_LEVELS];
    size_t pyr_pitch;
    const uint8_t* ref_pyr;   /* n_pairs * pyr_pitch                    */
    const uint8_t* cur_pyr;   /* n_pairs * pyr_pitch                    */
    const float* px_xy;       /* n_pairs * max_features * 2             */
    const double* bearing;    /* n_pairs * max_features * 3             */
    const double* p_world;    /* n_pairs * max_features * 3             */
    const uint8_t* initial;   /* n_pairs * max_features                 */
    const int32_t* n_features;/* n_pairs, or NULL                       */
    const double* T_ref_w;    /* n_pairs * 12                           */
    double* T_cur_w;          /* n_pairs * 12, in: seed, out: result    */
    int32_t* n_tracked;       /* n_pairs                                */
    dsdtm_align_stats* stats; /* n_pairs, or NULL                       */
} dsdtm_batch_desc;

/* ---- Feature_detector::detect (per-cell part) ------------------------------------- */
/*
 * Replaces: the image part of void Feature_detector::detect(Frame*, const double detection_threshold,
 *           const bool) (include/Feature_detection.h:57, src/Feature_detection.cpp:75-108) with the
 *           vendored Thirdparty/fast calls it makes (fast_corner_detect_10_sse2, fast_corner_score_10,
 *           fast_nonmax_3x3) and Feature_detector::shiTomasiScore (:157-198): for every cell of the
 *           detector's grid, the FAST-10 corner (non-maximum suppressed, any pyramid level) with the
 *           best Shi-Tomasi score above detection_threshold, skipping occupied cells.
 * cell_score[k] = detection_threshold and cell_x/y/level[k] = 0 where no corner qualifies — exactly the
 * `corners` vector the reference then sorts and walks (:110-150; that order-dependent rest — mask
 * discs, Max_fts cap — is host bookkeeping over <= grid_cols*grid_rows entries, see
 * dsdtm_amd/feature_detection.py and INTEGRATION.md). cell_x/cell_y are level-0 pixels (x * 2^level).
 */
typedef struct dsdtm_detect_params {
    int32_t cell_size;            /* Camera.CellSize (src/Feature_detection.cpp:12) */
    int32_t grid_cols, grid_rows; /* ceil(width / cell_size), ceil(height / cell_size) (:18-19) */
    int32_t levels;               /* Camera.MaxPyraLevels (:13); <= the pyramid's level count */
    int32_t barrier;              /* FAST barrier, 20 in the reference (:82,:91) */
    float detection_threshold;    /* >= 0 */
} dsdtm_detect_params;
int dsdtm_detect_cells(dsdtm_ctx* ctx, const dsdtm_pyramid* host_pyramid, const uint8_t* grid_occupied,
                       const dsdtm_detect_params* params, float* cell_score, int32_t* cell_x,
                       int32_t* cell_y, int32_t* cell_level);

/* ---- Frames that stay on the device ---------------------------------------------- */
/*
 * Replaces: the image part of DSDTM::Frame (include/Frame.h: mvImg_Pyr, built once per frame by
 *           Frame::ComputeImagePyramid, src/Frame.cpp:74-81) for callers that keep frames between
 *           Run calls. In tracking every frame is the current frame of one Run and the reference
 *           frame of the next (src/Tracking.cpp:204,224): with a dsdtm_frame its pyramid crosses
 *           PCIe once instead of twice, and dsdtm_frame_create_from_image sends level 0 only and
 *           builds the other levels with the library's bit-exact pyrDown.
 * A frame belongs to the context that created it and is immutable. dsdtm_frame_destroy: with the frame's own, live context
 * its buffer goes to that context's pool and serves the next frame of the same size (a tracker creates and destroys one
 * frame per image: no allocation, no device-wide wait per frame); with ctx == NULL, another context, or after the context
 * is gone (the pointer is compared, never dereferenced) the buffer is freed, which waits for the device's pending work.
 * Destroy a frame only when no call that uses it is still running (every entry that takes a dsdtm_frame returns after its
 * stream has drained). dsdtm_sparse_align_frames is dsdtm_sparse_align with the two host
 * pyramids replaced by frames (same results, bit for bit: it runs the same kernel).
 */
typedef struct dsdtm_frame dsdtm_frame;
int dsdtm_frame_create(dsdtm_ctx* ctx, const dsdtm_pyramid* host_pyramid, dsdtm_frame** out);
int dsdtm_frame_create_from_image(dsdtm_ctx* ctx, const uint8_t* level0, int width, int height, int stride,
                                  int levels, dsdtm_frame** out);
void dsdtm_frame_destroy(dsdtm_ctx* ctx, dsdtm_frame* frame);
int dsdtm_sparse_align_frames(dsdtm_ctx* ctx, const dsdtm_frame* ref, const dsdtm_frame* cur,
                              const dsdtm_camera* cam, const float* px_xy, const double* bearing,
                              const double* p_world, const uint8_t* initial, int n_features,
                              const double T_ref_w[12], double T_cur_w[12],
                              const dsdtm_align_params* params, int* n_tracked, dsdtm_align_stats* stats);
/*
 * Replaces: bool Feature_Alignment::FindMatchDirect(const MapPoint*, const FramePtr, Eigen::Vector2d&, int&)
 *           (src/Feature_alignment.cpp:128-158, after Get_ClosetObs and the IsInImage test :135-140) for
 *           M candidates at once on device-resident frames: SolveAffineMatrix / GetBestSearchLevel /
 *           WarpAffine / GetPatchNoBoarder (:142-148), px / 2^level (:150), Align2DGaussNewton(.., 10, ..)
 *           (:152), px * 2^level (:154-156). One call; the warped patches never leave the device.
 * kf          : the keyframes' frames (same pyramid geometry as cur)
 * cand_kf ... p_world, max_search_level : as dsdtm_warp_patches
 * px_xy       : M x 2 double, in: the candidate's reprojected pixel (level 0), out: the refined pixel
 *               (level 0), written back also when not converged (:414)
 * search_level, converged : M outputs
 */
int dsdtm_match_candidates_frames(dsdtm_ctx* ctx, const dsdtm_frame* cur, const dsdtm_frame* const* kf, int n_kf,
                                  const dsdtm_camera* cam, const double* T_kf_w, const double T_cur_w[12],
                                  const int32_t* cand_kf, const float* ref_px, const int32_t* ref_level,
                                  const double* ref_bearing, const double* p_world, int max_search_level,
                                  int max_iters, int m, double* px_xy, int32_t* search_level, uint8_t* converged);

/* The same for the candidates of n_frames CURRENT frames at once (independent sequences batch their reprojection search
 * as they batch Run): packed DEVICE pyramids (current frames at cur_pyr + f * pyr_pitch, keyframes at kf_pyr + k *
 * pyr_pitch, one geometry), device arrays throughout, nothing copied, asynchronous on hip_stream. cand_frame[i] is the
 * current frame of candidate i (T_cur_w: n_frames x 12), cand_kf[i] its reference keyframe; a candidate whose
 * cand_frame is outside [0, n_frames), cand_kf outside [0, n_kf) or ref_level outside the pyramid is rejected on the
 * device (search_level -1, converged 0, pixel untouched) — never dereferenced. scratch:
 * dsdtm_match_candidates_scratch_bytes(m) bytes, 16-byte aligned (kept in the signature: since library 0.5 the warp prelude and
 * Align2D run as ONE kernel and the warped patches stay on chip; only the two-kernel diagnostic path writes them here). px_xy in/out in
 * level-0 pixels, as above. Candidates of one current frame should be contiguous in the arrays (as a tracker produces them): the
 * kernel gives each of the GPU's eight L2 caches a contiguous range of candidates, so that a frame's images are fetched once. Replaces FindMatchDirect (src/Feature_alignment.cpp:128-158) per candidate. */
int dsdtm_match_candidates_batch_device(dsdtm_ctx* ctx, const uint8_t* cur_pyr, int n_frames, const uint8_t* kf_pyr, int n_kf,
                                        size_t pyr_pitch, int levels, const int* width, const int* height, const int* stride,
                                        const size_t* level_offset, const dsdtm_camera* cam, const double* T_kf_w,
                                        const double* T_cur_w, const int32_t* cand_frame, const int32_t* cand_kf,
                                        const float* ref_px, const int32_t* ref_level, const double* ref_bearing,
                                        const double* p_world, int max_search_level, int max_iters, int m,
                                        uint8_t* scratch, double* px_xy, int32_t* search_level, uint8_t* converged,
                                        void* hip_stream);
size_t dsdtm_match_candidates_scratch_bytes(int m);

/* dsdtm_detect_cells on a device-resident frame (keyframe creation detects on the frame that was just tracked) */
int dsdtm_detect_cells_frame(dsdtm_ctx* ctx, const dsdtm_frame* frame, const uint8_t* grid_occupied,
                             const dsdtm_detect_params* params, float* cell_score, int32_t* cell_x,
                             int32_t* cell_y, int32_t* cell_level);

/* The same for n_frames packed DEVICE pyramids at once (frame f at pyr + f * pyr_pitch, geometry as in
 * dsdtm_pyrdown_batch_device): independent sequences — the config-4 style of use — batch their keyframes' detector
 * work as they batch Run, pyramids, Align2D and the pose refinement. Every pointer is a device pointer; nothing is
 * copied; asynchronous on hip_stream. grid_occupied: n_frames * cells bytes or NULL. score_scratch: n_frames * pyr_pitch
 * bytes, key_scratch: n_frames * cells 64-bit words (both overwritten). Outputs: n_frames * cells entries each, the
 * meaning of dsdtm_detect_cells' outputs. pyr, score_scratch and pyr_pitch must be 4-byte aligned and
 * n_frames * params->levels <= 65535 per call (DSDTM_ERR_INVALID otherwise: split the batch).
 * Replaces the image part of src/Feature_detection.cpp:69-154 per frame. */
int dsdtm_detect_cells_batch_device(dsdtm_ctx* ctx, const uint8_t* pyr, size_t pyr_pitch, int n_frames, int levels,
                                    const int* width, const int* height, const int* stride, const size_t* level_offset,
                                    const uint8_t* grid_occupied, const dsdtm_detect_params* params,
                                    uint8_t* score_scratch, unsigned long long* key_scratch,
                                    float* cell_score, int32_t* cell_x, int32_t* cell_y, int32_t* cell_level,
                                    void* hip_stream);


/* ---- Optimizer::PoseOptimization ------------------------------ */
/*
 * Replaces: static void Optimizer::PoseOptimization(FramePtr tCurFrame, int tIterations = 100)
 *           (include/Optimizer.h:29, src/Optimizer.cpp:20-101; call site src/Tracking.cpp:236, right after
 *           SearchLocalPoints) — the numerical part: the Ceres solve of the 6-parameter pose block
 *           [t, log R] over the frame's map-point observations with FullBA_Problem residuals
 *           (include/Optimizer.h:129-216; map points held constant, src/Optimizer.cpp:60-61),
 *           ceres::CauchyLoss(1.0) (:33), PoseLocalParameterization (include/Optimizer.h:219-252),
 *           Solver::Options defaults + max_num_iterations = 100 (:68-72), then Set_Pose (:78) and the
 *           per-block residual norms of GetReprojectReidual (src/Optimizer.cpp:297-317).
 *           The EraseFound walk over those norms (:80-92) is host bookkeeping on MapPoint objects
 *           (dsdtm_amd/optimizer.py, dsdtm_amd/host/dsdtm_host.hpp).
 * Ceres is not under /root/reference (README.md:7 links its repository, no version): the solver is a
 * restatement of Ceres 1.13's TrustRegionMinimizer + LevenbergMarquardtStrategy with the dense
 * linear solve done on the 6x6 normal equations — see DESIGN.md §3.6.
 *
 * bearing   : N x 3 doubles  Feature::mNormal  (observation = (n0/n2, n1/n2), include/Optimizer.h:160)
 * p_world   : N x 3 doubles  Feature::Mpt->Get_Pose()
 * level     : N              Feature::mlevel   (residual is divided by 1 << level, include/Optimizer.h:162)
 * use       : N x u8         1 where the reference adds a residual block: Mpt != NULL && !Mpt->IsBad()
 *                            && mbInitial (src/Optimizer.cpp:47-55)
 * T_cur_w   : [R|t] 3x4 row-major; in: tCurFrame->Get_Pose(), out: the pose handed to Set_Pose
 * residual_norm : N doubles; the first summary->n_residual_blocks entries are the norms in residual-
 *                 block order (the order of the features with use != 0), the rest is untouched
 */
typedef struct dsdtm_pose_opt_params {
    int32_t max_iterations;   /* 100 (src/Optimizer.cpp:72; the tIterations argument is ignored there) */
    int32_t reserved;
} dsdtm_pose_opt_params;
enum dsdtm_pose_opt_termination {
    DSDTM_PO_FUNCTION_TOLERANCE = 0, /* |cost change| <= 1e-6 * cost                      */
    DSDTM_PO_PARAMETER_TOLERANCE = 1, /* |step| <= 1e-8 * (|x| + 1e-8)                     */
    DSDTM_PO_GRADIENT_TOLERANCE = 2,  /* max |x - Plus(x, -g)| <= 1e-10                    */
    DSDTM_PO_MAX_ITERATIONS = 3,
    DSDTM_PO_MIN_RADIUS = 4,          /* trust-region radius <= 1e-32                      */
    DSDTM_PO_INVALID_STEPS = 5,       /* 5 consecutive steps without model decrease        */
    DSDTM_PO_NO_RESIDUALS = 6,        /* no feature with use != 0: pose only re-normalised */
    DSDTM_PO_EVALUATION_FAILED = 7    /* non-finite residual/Jacobian at the start         */
};
typedef struct dsdtm_pose_opt_summary {
    int32_t iterations;        /* trust-region iterations run (Ceres: summary.iterations.size() - 1) */
    int32_t successful_steps;
    int32_t termination;       /* dsdtm_pose_opt_termination */
    int32_t n_residual_blocks;
    double initial_cost, final_cost;   /* 1/2 sum rho(|r_i|^2) */
    double x[6];               /* final parameter block [t, log R] */
} dsdtm_pose_opt_summary;
int dsdtm_pose_optimization(dsdtm_ctx* ctx, const double* bearing, const double* p_world,
                            const int32_t* level, const uint8_t* use, int n_features,
                            double T_cur_w[12], const dsdtm_pose_opt_params* params,
                            double* residual_norm, dsdtm_pose_opt_summary* summary);
/* n_frames independent problems (e.g. one per tracked sequence) in one launch. All pointers are DEVICE
 * pointers; frame i's feature columns start at element i * max_features of each array (bearing and
 * p_world: i * max_features * 3), its pose at T_cur_w + 12 * i, its norms at residual_norm +
 * i * max_features, its summary at summary + i. Asynchronous on `hip_stream`. */
int dsdtm_pose_optimization_batch_device(dsdtm_ctx* ctx, int n_frames, int max_features,
                                         const int32_t* n_features, const double* bearing,
                                         const double* p_world, const int32_t* level, const uint8_t* use,
                                         double* T_cur_w, const dsdtm_pose_opt_params* params,
                                         double* residual_norm, dsdtm_pose_opt_summary* summary,
                                         void* hip_stream);

/* ---- One tracked frame in ONE submission (src/Tracking.cpp:199-256) ----------------- */
/*
 * Replaces, for a tracker that keeps its frames on the device, the chain Tracking runs per frame:
 *   Frame construction (src/Frame.cpp:35-41 -> ComputeImagePyramid :74-81)           new frame: level-0 upload + pyramid
 *   TrackWithLastFrame  (src/Tracking.cpp:199-217): Set_Pose(last pose); Sprase_ImgAlign::Run(cur, last)
 *   UpdateLocalMap      (:258-312): Feature_Alignment::ResetGrid; ReprojectPoint for every local map point
 *   TrackWithLocalMap   (:219-256): SearchLocalPoints (src/Feature_alignment.cpp:71-158: FindMatchDirect for every
 *                       candidate, the cell walk with its order-dependent rules, see below) and
 *                       Optimizer::PoseOptimization (src/Optimizer.cpp:20-79) on the features the search created.
 * Through the entries above that is four synchronous calls with host work between them (0.28 ms wall for 0.18 ms of
 * kernels); here everything is enqueued back to back on the context's stream — level 0 and Run's inputs into HBM, pyramid, Run,
 * reprojection + closest observation + FindMatchDirect for all points, the replay of the cell walk ON THE DEVICE, pose
 * refinement — and the host waits ONCE, for a few hundred bytes of results in pinned memory.
 *
 * The device replay follows src/Feature_alignment.cpp:71-121 exactly: cells in index order (:75; mCellOrder is shuffled
 * but unused), per cell the candidates by Get_FoundNums() descending, stable in ReprojectPoint order (:88, :123-126), bad
 * points skipped (:93), the mask test on the ROUNDED reprojected pixel (:96), first success wins the cell (:115), a success
 * paints a disc of radius cell_size at the rounded REFINED pixel (:111, cv::circle's filled midpoint circle) which can
 * suppress candidates of later cells, stop after max_matches matched cells (:80). The caller's list of map points is
 * the order ReprojectPoint would be called in (UpdateLocalMap walks the local keyframes' map points, :283-299); points the
 * reference never projects (NULL, already projected for this frame) are simply not in the list; bad ones may be (mp_bad).
 * Because the local map is handed over BEFORE Run, it is the caller's choice of local keyframes that is fixed early
 * (the reference picks them with the pose Run produced, :263); visibility (ReprojectPoint's image test) and the closest
 * observation (MapPoint::Get_ClosetObs, src/MapPoint.cpp:133-174) are evaluated on the device with the pose Run produced.
 *
 * Observations are flattened by the caller: point i has observations obs_offset[i] .. obs_offset[i+1]-1 in the iteration
 * order of its mObservations map; observation j names keyframe obs_kf[j] (index into kf[]) and carries the observing
 * feature's mpx (obs_px), mlevel (obs_level) and mNormal (obs_bearing).
 * Limits: n_points <= 4096, grid cells <= 4096, cell_size <= 127, max_matches <= 256, n_kf <= 4096 (DSDTM_ERR_INVALID beyond).
 * The image: rows of `width` bytes, `stride` bytes apart (stride >= width), in host memory or in the memory of the context's
 * device. Level 0 enters through a kernel on the compute stream, not through a copy operation in front of it: a contiguous,
 * 16-byte aligned image in pinned host memory (hipHostMalloc / hipHostRegister) or in device memory is read straight from
 * where it is; any other host image is staged through the context's pinned block first (a pinned one at an odd address, and a
 * strided or odd device image, go through the copy engine). Same results on every path.
 */
typedef struct dsdtm_track_desc {
    /* the new frame (src/Frame.cpp:35-41) */
    const uint8_t* image; int32_t width, height, stride, levels;
    /* Run(cur, ref): the last frame and its features, as dsdtm_sparse_align_frames */
    const dsdtm_frame* ref;
    const float* ref_px_xy; const double* ref_bearing; const double* ref_p_world; const uint8_t* ref_initial;
    int32_t n_ref_features;
    const double* T_ref_w;            /* 12 */
    const double* T_seed;             /* 12: the pose Tracking seeds the new frame with (:201) */
    dsdtm_align_params align;
    int32_t min_tracked;              /* Tracking: Run's count below this (20, :208) => Lost: search and refinement are skipped */
    /* the local map (UpdateLocalMap) */
    const dsdtm_frame* const* kf; int32_t n_kf;
    const double* T_kf_w;             /* n_kf x 12 */
    int32_t n_points;
    const double* mp_world;           /* n_points x 3  MapPoint::Get_Pose()      */
    const int32_t* mp_found;          /* n_points      MapPoint::Get_FoundNums() */
    const uint8_t* mp_bad;            /* n_points      MapPoint::IsBad()         */
    const int32_t* obs_offset;        /* n_points + 1 */
    const int32_t* obs_kf; const float* obs_px; const int32_t* obs_level; const double* obs_bearing;
    const uint8_t* mask; int32_t mask_stride;   /* Frame::mImgMask at the start of the search (255 = free), or NULL = all free */
    int32_t cell_size;                /* Camera.CellSize      */
    int32_t max_pyr_levels;           /* Camera.MaxPyraLevels (search level cap = this - 3, src/Feature_alignment.cpp:144) */
    int32_t max_matches;              /* 200 (src/Feature_alignment.cpp:80) */
    int32_t align2d_iters;            /* 10  (:152) */
    dsdtm_pose_opt_params pose_opt;
} dsdtm_track_desc;
typedef struct dsdtm_track_match {
    int32_t cell;                     /* grid cell of the candidate (index order = creation order of the features)   */
    int32_t point;                    /* index into the caller's map-point list: Feature::Mpt; IncreaseFound() is the caller's to apply (:106) */
    float px[2];                      /* Feature::mpx: the refined pixel as cv::Point2f (:108)                        */
    int32_t level;                    /* Feature::mlevel: the search level                                            */
} dsdtm_track_match;
typedef struct dsdtm_track_result {
    dsdtm_frame* frame;               /* the new frame, resident on the device: the caller's to destroy */
    double T_run[12];                 /* the pose Run handed to Set_Pose (:57); the seed when Run returned 0 by the Min_fts rule */
    int32_t n_tracked;                /* Run's return value */
    int32_t lost;                     /* 1: n_tracked < min_tracked — nothing below was computed (n_matches 0, T_opt = T_run) */
    dsdtm_align_stats stats;
    int32_t n_in_grid;                /* points ReprojectPoint put into the grid */
    int32_t n_matches;
    int32_t replay_full_scan;         /* diagnostic: 1 = a candidate had more possible blockers than the replay keeps per thread and the
                                         workgroup scanned all earlier candidates each round instead (same decisions, slower) */
    int32_t reserved;
    double T_opt[12];                 /* the pose PoseOptimization hands to Set_Pose (src/Optimizer.cpp:78) */
    dsdtm_pose_opt_summary summary;
} dsdtm_track_result;
/* matches: max_matches entries; residual_norm: max_matches doubles (the norms of GetReprojectReidual in feature = match
 * order, for the EraseFound walk of src/Optimizer.cpp:80-92, which stays the caller's). The mask discs of the matches
 * (:111) are the caller's to paint into its Frame::mImgMask from the match list if it keeps one. */
int dsdtm_track_frame(dsdtm_ctx* ctx, const dsdtm_camera* cam, const dsdtm_track_desc* desc,
                      dsdtm_track_result* result, dsdtm_track_match* matches, double* residual_norm);

/* Enqueues the alignment of all pairs on `hip_stream` (a hipStream_t, NULL = default
 * stream). Asynchronous: results are valid after the stream is synchronised. */
int dsdtm_sparse_align_batch_device(dsdtm_ctx* ctx, const dsdtm_batch_desc* batch,
                                    const dsdtm_camera* cam, const dsdtm_align_params* params,
                                    void* hip_stream);

/* The same batch from HOST memory over several devices (BASELINE config 4: 8192 independent pairs over the 8 GPUs
 * of a node). Every pointer of `host_batch` is a HOST pointer. The pairs are cut into n_ctx
 * contiguous blocks — context g takes pairs [lo, hi) of dsdtm_shard_range(n_pairs, n_ctx, g, &lo, &hi), ceil(P/G)
 * each, the last ones possibly shorter or empty — and every block is processed by its own host thread on its
 * context's device: upload, one launch, download of T_cur_w / n_tracked / stats, synchronise. No collective, no
 * peer traffic; pairs are independent, so the results do not depend on n_ctx (bit for bit). The contexts may sit on
 * different devices (one per GPU: the intended use) or share one. Returns the first non-zero status of any shard
 * (dsdtm_last_error of that context has the message). The reference has no counterpart: it runs this path on one
 * thread of one process (src/System.cpp:39-55). Pairs whose chain structure allows it can share frames: the
 * device-resident entry point above accepts cur_pyr == ref_pyr + pyr_pitch (frame k is `cur` of pair k - 1 and
 * `ref` of pair k), so a sequence of P + 1 frames is uploaded and built once. */
int dsdtm_sparse_align_batch_sharded(dsdtm_ctx* const* ctx, int n_ctx, const dsdtm_batch_desc* host_batch,
                                     const dsdtm_camera* cam, const dsdtm_align_params* params);
/* Pairs [*lo, *hi) of shard `shard` of `n_shards` over `n_pairs` pairs: contiguous blocks of ceil(n_pairs / n_shards). */
void dsdtm_shard_range(int n_pairs, int n_shards, int shard, int* lo, int* hi);

/* The same batch STREAMED from host memory — the form a host-fed pipeline wants (the reference feeds frames from host
 * memory: src/Tracking.cpp:45-57 -> src/Frame.cpp:35-41,74-81 -> src/Sprase_ImageAlign.cpp:29-60). Only LEVEL 0 of every
 * frame crosses the link (24 % fewer bytes than a 4-level pyramid); levels 1.. are built on the device by the bit-exact
 * pyrDown (Frame::ComputeImagePyramid), and a CHAINED sequence (cur_image == NULL: n_pairs + 1 frames at ref_image,
 * frame i + 1 is the current frame of pair i — a tracked sequence) uploads and builds every frame once instead of twice.
 * Per shard (contiguous pairs per context, as dsdtm_shard_range) the pairs run in chunks of `chunk_pairs` (<= 0: 128):
 * upload of chunk j + 1 on one of two copy streams overlaps pyramids + alignment + result download of chunk j. Pin the
 * host arrays (hipHostMalloc / hipHostRegister) for the copies to run at link speed and asynchronously. Results are
 * those of dsdtm_sparse_align_batch_sharded on the same frames, bit for bit. Returns when every shard is done. */
typedef struct dsdtm_stream_desc {
    int32_t n_pairs, max_features, levels;   /* levels of the pyramids built on the device (1..DSDTM_MAX_LEVELS) */
    int32_t width, height;                   /* level 0 */
    int32_t row_stride;                      /* bytes between image rows on the host (== width: whole images are copied at once) */
    size_t image_pitch;                      /* bytes between two frames on the host */
    const uint8_t* ref_image;                /* host: pair i's reference frame at ref_image + i * image_pitch */
    const uint8_t* cur_image;                /* host, or NULL = chained (n_pairs + 1 frames at ref_image) */
    const float* px_xy; const double* bearing; const double* p_world; const uint8_t* initial;   /* host, as dsdtm_batch_desc */
    const int32_t* n_features;               /* host, n_pairs, or NULL */
    const double* T_ref_w;                   /* host, n_pairs * 12 */
    double* T_cur_w;                         /* host, n_pairs * 12, in: seed, out: result */
    int32_t* n_tracked;                      /* host, n_pairs */
    dsdtm_align_stats* stats;                /* host, n_pairs, or NULL */
} dsdtm_stream_desc;
int dsdtm_sparse_align_batch_streamed(dsdtm_ctx* const* ctx, int n_ctx, const dsdtm_stream_desc* host_frames, int chunk_pairs,
                                      const dsdtm_camera* cam, const dsdtm_align_params* params);

/* Result check for callers of the asynchronous batch entry point: waits for `hip_stream` and settles the launches
 * issued on it through this context since the last check. Shapes that spread one pair over several compute units
 * (few pairs of more than 448 features; batches of 1025..2048 features) wait for partner workgroups, i.e. for the
 * GPU's dispatch; should such a wait run out (a foreign load on the device), the launch is re-seeded from a copy of
 * its input poses and re-run HERE on the one-compute-unit kernels, and the caller still gets DSDTM_OK with the
 * results of the reference's Run (which cannot fail for scheduling reasons, src/Sprase_ImageAlign.cpp:29-60) — so
 * results are final once this function has returned, not before. DSDTM_ERR_HIP remains for a hand-over that failed
 * inside one workgroup (a broken protocol; never observed) and for a re-run that failed as well. Timeout words,
 * counters and re-run records are per context (host-mapped words read without a copy): a check never synchronises
 * the device or touches another context's state. At most 64 multi-compute-unit launches of a context are kept
 * unsettled; the 65th first waits for the oldest and settles it on the stream it was launched on — so the device
 * buffers a launch names (T_cur_w, n_tracked, stats) and its stream must stay alive until that stream's check, and
 * launches of one stream that have not been checked yet must name DISTINCT output buffers (a re-run is queued behind the
 * stream's later work: were a later launch to reuse the pose buffer, the re-run would overwrite its results — such a case
 * is reported as DSDTM_ERR_HIP by the check instead of being re-run). On devices that are not one 8-XCD / 256-CU partition, and inside
 * a stream capture, those shapes run the one-compute-unit kernels from the start. The reference has no counterpart
 * (its path is one CPU thread); the single-pair host entry points above settle their launch themselves. */
int dsdtm_sparse_align_check(dsdtm_ctx* ctx, void* hip_stream);

/* Streams and hipGraphs. Launches of one context may be in flight on up to 16 different streams at once (a 17th
 * stream takes over the bookkeeping of the one idle longest, after the host has waited for the event behind that
 * entry's last launch — no device-wide synchronisation once the table is full). A launch captured into a
 * hipGraph owns a pair-counter word of the context for good and is preceded by a memset node, so graph replays
 * need nothing from the host; a context has 256 such words: capture once and replay — an application that
 * re-captures per frame gets DSDTM_ERR_INVALID from the 257th capture on and needs another context. Shapes that
 * spread one pair over several compute units (few pairs of more than 448 features) are ordered against each
 * other across streams by the library and are not used inside a capture (the single-CU kernels run instead).
 * Feature counts: up to 704 per pair (batches) or 16 384 per pair (few pairs: one pair over up to 64 compute
 * units) run register-resident kernels; beyond that (max 32 767) a pair runs on ONE compute unit through HBM
 * scratch — correct, but several times slower per feature (16 385 features: 1.95 ms against 0.38 ms for 16 384).
 * A context is used by one thread at a time (its stream bookkeeping is not locked). */

/* Bytes of scratch HBM the batch call needs for `batch` (0 when a register-resident kernel applies:
 * max_features <= 704, or few pairs of up to 16 384 features, which are spread over several compute units —
 * this function does not know the device and may over-estimate for those shapes). A live launch takes the
 * scratch from its stream's own workspace, which the context grows on demand (launches on different streams
 * never share scratch). A launch that is being captured into a hipGraph cannot allocate: call dsdtm_reserve
 * first; captured launches of one context share that reserved workspace, so their replays must not overlap. */
size_t dsdtm_sparse_align_workspace_bytes(const dsdtm_batch_desc* batch);
int dsdtm_reserve(dsdtm_ctx* ctx, size_t workspace_bytes);

/* ---- Feature_Alignment::Align2DGaussNewton ------------------------------------- */
/*
 * Replaces: static bool Feature_Alignment::Align2DGaussNewton(const cv::Mat& tCurImg,
 *           uchar* tPatch_WithBoarder, uchar* tPatch, int MaxIters, Eigen::Vector2d& tCurPx)
 *           (include/Feature_alignment.h:85, src/Feature_alignment.cpp:318-417)
 * for M independent features at once (the speculative form SearchLocalPoints needs,
 * SURVEY.md §7 "Sequential semantics of SearchLocalPoints").
 *
 * cur          : pyramid of the current frame (host)
 * patch_border : M x 100 u8  (10x10 bordered reference patch, mPatch_WithBoarder)
 * patch        : M x 64  u8  (8x8 reference patch, mPatch)
 * level        : M           search level of each feature (image = cur level[level[i]])
 * px_xy        : M x 2 double, in level coordinates; in: start, out: result. Written back
 *                even when not converged (NaN included), as the reference does (:414).
 * converged    : M x u8      the bool the reference returns
 */
int dsdtm_align2d_batch(dsdtm_ctx* ctx, const dsdtm_pyramid* cur,
                        const uint8_t* patch_border, const uint8_t* patch,
                        const int32_t* level, double* px_xy, uint8_t* converged,
                        int max_iters, int m);

/* Device-resident form. cur_pyr is one packed pyramid (same packing rules as the batch
 * descriptor above: level l at cur_pyr + level_offset[l]). */
typedef struct dsdtm_image_desc {
    int levels;
    int width[DSDTM_MAX_LEVELS];
    int height[DSDTM_MAX_LEVELS];
    int stride[DSDTM_MAX_LEVELS];
    size_t level_offset[DSDTM_MAX_LEVELS];
    size_t bytes;           /* size of the packed pyramid allocation */
    const uint8_t* data;    /* device */
} dsdtm_image_desc;

int dsdtm_align2d_batch_device(dsdtm_ctx* ctx, const dsdtm_image_desc* cur,
                               const uint8_t* patch_border, const uint8_t* patch,
                               const int32_t* level, double* px_xy, uint8_t* converged,
                               int max_iters, int m, void* hip_stream);

/* ---- Frame::ComputeImagePyramid ------------------------------ */
/*
 * Replaces the cv::pyrDown chain of Frame::ComputeImagePyramid (src/Frame.cpp:74-81)
 * for n_images packed pyramids on the device: level 0 must already be present at
 * pyr + i*pyr_pitch + level_offset[0]; levels 1..levels-1 are written. Bit-exact
 * OpenCV 8-bit semantics: separable [1 4 6 4 1], BORDER_REFLECT_101, (sum+128)>>8,
 * output size ((w+1)/2, (h+1)/2). Up to 32 images whose levels have widths that are multiples
 * of 8 (3..5 levels) are built by ONE launch (intermediate levels in LDS, a new frame of the
 * live tracker); larger batches and other shapes by one launch per level. Same bytes either way.
 */
int dsdtm_pyrdown_batch_device(dsdtm_ctx* ctx, uint8_t* pyr, size_t pyr_pitch, int n_images,
                               int levels, const int* width, const int* height,
                               const int* stride, const size_t* level_offset,
                               void* hip_stream);

/* Host convenience: builds levels 1.. of one pyramid whose level 0 is given. `out` levels
 * 1..levels-1 must point at caller-owned buffers of stride[l]*height[l] bytes. */
int dsdtm_pyrdown(dsdtm_ctx* ctx, const uint8_t* level0, int width, int height, int stride,
                  int levels, uint8_t* const* out_levels, const int* out_stride);

/* ---- Warp prelude of FindMatchDirect ------------------------- */
/*
 * Replaces, for M candidates at once, SolveAffineMatrix (src/Feature_alignment.cpp:160-190),
 * GetBestSearchLevel (:192-204), WarpAffine (:206-259) and GetPatchNoBoarder (:261-275).
 * One candidate = one (MapPoint, closest-observation reference feature) pair whose
 * reference keyframe is one of `n_kf` keyframes.
 *
 * kf_pyr[k]        : pyramid of reference keyframe k (host)
 * T_kf_w           : n_kf x 12, keyframe poses (world -> keyframe camera)
 * T_cur_w          : 12, current frame pose
 * cand_kf          : M, index of the candidate's reference keyframe
 * ref_px           : M x 2 float, Feature::mpx of the reference feature (level-0 coords)
 * ref_level        : M, Feature::mlevel of the reference feature
 * ref_bearing      : M x 3 double, Feature::mNormal
 * p_world          : M x 3 double, MapPoint position
 * max_search_level : Camera.MaxPyraLevels - 3 (:144)
 * outputs          : affine (M x 4 double, row-major 2x2 A_cur<-ref), search_level (M),
 *                    patch_border (M x 100), patch (M x 64)
 */
int dsdtm_warp_patches(dsdtm_ctx* ctx, const dsdtm_pyramid* kf_pyr, int n_kf,
                       const dsdtm_camera* cam, const double* T_kf_w, const double T_cur_w[12],
                       const int32_t* cand_kf, const float* ref_px, const int32_t* ref_level,
                       const double* ref_bearing, const double* p_world,
                       int max_search_level, int m,
                       double* affine, int32_t* search_level,
                       uint8_t* patch_border, uint8_t* patch);

#ifdef __cplusplus
}
#endif
#endif /* DSDTM_AMD_H */
