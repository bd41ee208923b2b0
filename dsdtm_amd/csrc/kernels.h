// kernels.h — launch-side declarations shared by the .hip kernels and api.cpp.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/dsdtm_amd.h"

namespace dsdtm {

struct LevelGeom {
    int w, h, stride;
    uint32_t off;   // byte offset of the level inside one packed pyramid
};

// Arguments of the sparse-alignment kernels (passed by value; all pointers are device pointers).
struct SAKernelArgs {
    const uint8_t* ref_pyr;
    const uint8_t* cur_pyr;
    const float* px_xy;
    const double* bearing;
    const double* p_world;
    const uint8_t* initial;
    const int32_t* n_features;
    const double* T_ref_w;
    double* T_cur_w;
    int32_t* n_tracked;
    dsdtm_align_stats* stats;
    double* workspace;
    unsigned* pair_counter;     // device word, zeroed before each launch: next pair index for the persistent slots
    unsigned* timeout_flag;     // host-mapped word of the launching context (multi-CU launches: a word of their own): set
                                // to 1 by a wave whose bounded wait ran out; read by the host once the stream has drained
    unsigned spin_limit;        // team kernel: polls before a wait gives up (0 = the default, 2^24); tests shorten it
    unsigned team_epoch;        // team kernel: number of this launch among the context's team launches (part of the exchange tags)
    int debug_drop;             // tests only: member 1 of every two-member pair exits at once (its partner's waits run out)
    int ws_sort;                // workspace kernels with LDS windows: the pair's features are walked in image-row order
    unsigned long long pyr_pitch;
    int n_pairs, max_features;
    int max_level, min_level, max_iters, min_fts;
    float fx, fy, cx, cy, f;
    LevelGeom lv[DSDTM_MAX_LEVELS];
};

// Diagnostic switches (A/B runs, tests). They exist as VARIABLES only in the diagnostic build (build.py --diag,
// -DDSDTM_DIAG: libdsdtm_amd_diag.so), where they are process-wide, read from the environment once — by the first
// dsdtm_create — and changed afterwards through dsdtm_debug_set_option. In the RELEASE build (libdsdtm_amd.so, the
// library bench.py, smoke() and the parity tests load) they are the compile-time constants below: nothing reads the
// environment, nothing can change them, and the kernels only they select are not compiled in.
struct Options {
    int no_team = 0;           // DSDTM_NO_TEAM: large single pairs take the one-CU kernels instead of a team
    int team_min = 449;        // DSDTM_TEAM_MIN: feature count from which few large pairs run as teams
    int team_spread_min = 33;  // DSDTM_TEAM_SPREAD_MIN: team size from which members are spread over all XCDs
    int ws_from = 704;         // DSDTM_WS_FROM: feature counts above this run the workspace kernel in batches
    int ws_no_windows = 0;     // DSDTM_WS_NO_WINDOWS
    int match_group = 0;       // DSDTM_MATCH_GROUP: candidates per workgroup of the fused FindMatchDirect kernel (0 = 16; 32 and 64 measured slower)
    int fmd_no_xcd = 0;        // DSDTM_FMD_NO_XCD: the fused FindMatchDirect kernel with plain block numbering (A/B)
    int fmd_split = 0;         // DSDTM_FMD_SPLIT: FindMatchDirect as two kernels with the patches through HBM (rounds 1-4; A/B)
    int ws_no_sort = 0;        // DSDTM_WS_NO_SORT: the workspace kernels walk a pair's features in list order (A/B, tests)
    int ws_no_duo = 0;         // DSDTM_WS_NO_DUO: 1025..2048 patches on one compute unit (HBM workspace) instead of two
    int pyr_fused = 1;         // DSDTM_PYR_FUSED: 0 never / 1 up to 32 images / 2 whenever the shape allows
    int pyr_band = 0;          // DSDTM_PYR_BAND: rows of the coarsest level per workgroup of the fused kernel (0: auto)
    int no_zero_copy = 0;      // DSDTM_NO_ZERO_COPY: single-call entry points copy instead of mapping the pinned block
    int po_no_cache = 0;       // DSDTM_PO_NO_CACHE: pose refinement without features in registers
    int a2d_tree = 0;          // DSDTM_A2D_TREE: Align2D with DPP tree sums (cost comparison only; not bit-identical)
    int a2d_group = 4;         // DSDTM_A2D_GROUP: Align2D features per wavefront (4; 8 = diagnostic: 114 VGPRs, measured 7 % slower)
    int warp_group = 0;        // DSDTM_WARP_GROUP: candidates per workgroup of the warp prelude (2, 8, 16, 32, 64; 0: by batch size)
    int team_no_wrap_clear = 0;  // DSDTM_TEAM_NO_WRAP_CLEAR: the team ring is NOT re-zeroed when the tag epoch wraps (A/B of that hazard only)
    int no_recover = 0;        // DSDTM_NO_RECOVER: a multi-CU launch that timed out is reported, not re-run (tests)
    int po_rows = 1;           // DSDTM_PO_ROWS: batched pose refinement with four frames per wavefront (0: one frame per wavefront, rounds 1-5; A/B)
};
#ifdef DSDTM_DIAG
Options& options();
#else
inline constexpr Options kReleaseOptions{};
inline constexpr const Options& options() { return kReleaseOptions; }
#endif

constexpr int SA_PPW = 2;      // pair slots per workgroup of the <= 320-feature register kernel (see sparse_align.hip)
enum SAVariant { SA_REG320 = 0, SA_REG448 = 1, SA_WS = 2, SA_REG128 = 3, SA_REG192 = 4, SA_REG256 = 5, SA_REG704 = 6 };
SAVariant sparse_align_pick_variant(int max_features);
size_t sparse_align_workspace_bytes(int n_pairs, int max_features);
// allow_multi_cu == false: shapes that would run one pair on two compute units take the one-CU kernel instead
hipError_t sparse_align_launch(const SAKernelArgs& args, SAVariant variant, int num_cus, hipStream_t stream, bool allow_multi_cu = true);
// true when sparse_align_launch (with allow_multi_cu) would run this feature count as pairs on two compute units
bool sparse_align_uses_duo(int max_features, bool have_workspace);
// team kernel (one pair over K workgroups, 704 < N <= 4096, few pairs): K or 0; bytes of the zeroed team buffers
int sparse_align_team_size(int n_pairs, int max_features, int num_cus);
size_t sparse_align_team_bytes(int n_pairs);
// drop_members > 0 (tests only): the last members of every team are not launched, so the others' waits run out
hipError_t sparse_align_launch_team(const SAKernelArgs& args, int k, hipStream_t stream, int drop_members = 0);
#ifdef DSDTM_DIAG
int sparse_align_occupancy(int variant);   // occupancy API answer (workgroups per CU)
hipError_t sparse_align_launch_stamps(const SAKernelArgs& args, int num_cus, hipStream_t stream);
#endif

// Feature detector (per-cell part): FAST-10 score map, then non-max + Shi-Tomasi + best corner per cell — for
// n_frames packed pyramids at once (frame f: pyr + f * pyr_pitch, score + f * pyr_pitch, cell_key / occupied / the
// outputs + f * cells).
struct DetectArgs {
    const uint8_t* pyr;            // n_frames packed pyramids (device)
    uint8_t* score;                // score maps, same layout and pitch as the pyramids
    unsigned long long* cell_key;  // n_frames * cells keys, zeroed before the launch
    const uint8_t* occupied;       // n_frames * cells, may be null
    uint8_t* keep;                 // diagnostic (dsdtm_debug_fast10, one frame): 1 where a corner survives the non-max step; else null
    // optional decoded outputs (batch entry; null: the host decodes the keys): n_frames * cells each
    float* cell_score; int32_t* cell_x; int32_t* cell_y; int32_t* cell_level;
    size_t pyr_pitch;
    int n_frames, levels;
    int cell_size, grid_cols, grid_rows, barrier;
    float detection_threshold;
    LevelGeom lv[DSDTM_MAX_LEVELS];
};
hipError_t detect_launch(const DetectArgs& args, hipStream_t stream);

// Align2D: one wavefront per feature.
struct A2DKernelArgs {
    const uint8_t* cur_pyr;       // packed pyramid (device)
    const uint8_t* patch_border;  // M x 100
    const uint8_t* patch;         // M x 64
    const int32_t* level;         // M
    double* px_xy;                // M x 2 (in/out)
    uint8_t* converged;           // M
    int m, max_iters, levels;
    int px_level0;                // 1: px_xy is in level-0 pixels (divided by 2^level on load, multiplied back on store)
    const int32_t* frame;         // optional, M: feature i sits on the pyramid cur_pyr + frame[i] * pyr_pitch (batches of frames)
    int n_frames;                 // with `frame`: indices outside [0, n_frames) are reported as not converged, never dereferenced
    size_t pyr_pitch;
    LevelGeom lv[DSDTM_MAX_LEVELS];
};
hipError_t align2d_launch(const A2DKernelArgs& args, hipStream_t stream);

// pyrDown: one launch per level over n_images packed pyramids.
hipError_t pyrdown_launch(uint8_t* pyr, size_t pyr_pitch, int n_images, int sw, int sh, int sstride,
                          size_t soff, int dstride, size_t doff, hipStream_t stream);
// pyrDown: all levels of n_images packed pyramids in ONE launch (band buffers of the intermediate levels in LDS).
// *launched == false: the shape is not eligible (odd or narrow levels, unaligned buffers) and nothing was enqueued —
// the caller falls back to one pyrdown_launch per level. band <= 0: chosen from the batch size.
hipError_t pyrdown_fused_launch(uint8_t* pyr, size_t pyr_pitch, int n_images, int levels, const int* w, const int* h,
                                const int* stride, const size_t* off, int band, hipStream_t stream, bool* launched);

// host-mapped pinned memory -> device by a kernel (a frame's level 0 + one small second range; see pyrdown.hip)
hipError_t ingest_launch(const void* src, void* dst, size_t bytes, const void* src2, void* dst2, size_t bytes2, hipStream_t stream);

// Warp prelude: groups of 2 / 64 candidates per 128-thread workgroup (lane = candidate for the FP64 chain, thread = sample for the patches).
struct WarpKernelArgs {
    const uint8_t* kf_pyr;        // n_kf packed pyramids, pitch kf_pitch ...
    const uint8_t* const* kf_ptrs; // ... or (non-null) n_kf base pointers of separately allocated pyramids (device array)
    size_t kf_pitch;
    const double* T_kf_w;         // n_kf x 12
    const int32_t* cand_kf;
    const float* ref_px;
    const int32_t* ref_level;
    const double* ref_bearing;
    const double* p_world;
    double* affine;               // M x 4
    int32_t* search_level;        // M
    uint8_t* patch_border;        // M x 100
    uint8_t* patch;               // M x 64
    double T_cur_w[12];
    const double* T_cur_w_arr;    // optional (batches of current frames): poses, 12 doubles each, indexed by cand_frame
    const int32_t* cand_frame;    // optional, M: the candidate's current frame
    int n_frames;                 // with cand_frame: indices outside [0, n_frames) are rejected like an invalid cand_kf
    int m, n_kf, max_search_level, levels;
    int no_xcd;                   // match_kernel: plain block numbering (A/B)
    float fx, fy, cx, cy;
    LevelGeom lv[DSDTM_MAX_LEVELS];
};
hipError_t warp_launch(const WarpKernelArgs& args, hipStream_t stream);
// FindMatchDirect in one launch (match.hip): the warp prelude and Align2D on patches that stay in LDS. `wa.patch_border`,
// `wa.patch` and `aa.patch_border`, `aa.patch`, `aa.level` are not used (the search level goes out through wa.search_level).
hipError_t match_launch(const WarpKernelArgs& wa, const A2DKernelArgs& aa, hipStream_t stream);

// Pose-only refinement (Optimizer::PoseOptimization): one wavefront per frame.
struct PoseOptArgs {
    int n_frames, max_features, max_iterations;
    const int32_t* n_features;     // per frame; null = max_features for every frame
    const double* bearing;         // n_frames x max_features x 3
    const double* p_world;         // n_frames x max_features x 3
    const int32_t* level;          // n_frames x max_features
    const uint8_t* use;            // n_frames x max_features
    double* T_cur_w;               // n_frames x 12 (in/out)
    double* residual_norm;         // n_frames x max_features
    dsdtm_pose_opt_summary* summary;   // n_frames
    // dsdtm_track_frame (one frame whose feature count is only known on the device): the instantiation is picked by the
    // RANGE the live count falls in, as dsdtm_pose_optimization picks it on the host — every candidate instantiation is
    // enqueued and the ones whose range (only_lo, only_hi] does not hold the count return at once. 0, 0: no filter.
    int only_lo = 0, only_hi = 0;
    int force_variant = 0;         // 0: by n_frames / max_features; 1: one wave per frame; 2: four waves, features in registers (<= 256);
                                   // 3: 1 or 2 by the frame's live count on the device (dsdtm_track_frame)
    double* T_mirror = nullptr;    // optional, n_frames x 12: the refined pose once more (dsdtm_track_frame: T_cur_w in device memory — the
                                   // kernel does not start with a read over the link — and the caller's pinned block here)
};
hipError_t pose_opt_launch(const PoseOptArgs& args, hipStream_t stream);

// One tracked frame in one submission (track.hip): what sits between Run, FindMatchDirect and the pose refinement.
struct TrackArgs {
    // from Run (device memory). The caller reads pose, count and statistics in its pinned block: block 0 of the fused
    // reprojection + FindMatchDirect kernel forwards them there (run_out_n16 x 16 bytes; posted writes — nothing on the device waits for the link)
    const double* T_run; const int32_t* n_tracked; int min_tracked;
    const void* run_out_dev; void* run_out_host; int run_out_n16;
    // the local map as the caller flattened it (device memory: copied up on a second stream while Run runs)
    const double* T_kf_w; const uint8_t* const* kf_ptrs; int n_kf;
    const double* mp_world; const int32_t* mp_found; const uint8_t* mp_bad; int n_points;
    const int32_t* obs_offset; const int32_t* obs_kf; const float* obs_px; const int32_t* obs_level; const double* obs_bearing;
    const uint8_t* mask; int mask_stride;
    float fx, fy, cx, cy; int width, height, levels;
    int cell_size, grid_cols, grid_rows, max_matches;
    int8_t disc_hw[128];                              // cv::circle's row half-widths for radius cell_size (track_disc_half_widths)
    // device scratch: the columns the FindMatchDirect kernel reads / writes (candidate = map point)
    double* pw; int32_t* cell; double* px0; double* px; int32_t* cand_kf; int32_t* cand_frame; float* ref_px; int32_t* ref_level;
    double* ref_bearing; uint8_t* init_blocked; int32_t* search_level; uint8_t* converged;
    // outputs of the replay: the match list and counts (host-mapped), the pose refinement's feature columns (device)
    dsdtm_track_match* matches; int32_t* counts;      // counts: [0] points in the grid, [1] matches, [2] 1 = the full scan ran
    double* T_opt;                                    // device: the pose refinement's in/out pose, seeded with T_run
    double* po_bearing; double* po_world; int32_t* po_level; uint8_t* po_use; int32_t* po_n;
};
// reprojection of every local map point + FindMatchDirect for it (wa / aa: the fused FindMatchDirect kernel's arguments over the
// candidate columns of `args`)
hipError_t track_match_launch(const TrackArgs& args, const WarpKernelArgs& wa, const A2DKernelArgs& aa, hipStream_t stream);
hipError_t track_replay_launch(const TrackArgs& args, hipStream_t stream);
size_t track_replay_lds_bytes(int n_points, int n_cells, int radius);
void track_disc_half_widths(int radius, int8_t* hw);   // radius <= 127

#ifdef DSDTM_DIAG
// device self-test of the FP64 building blocks (wave reduction, LDLT, SE3); see selftest.hip (diagnostic build only)
hipError_t selftest_launch(const double* in, double* out, int n_cases, hipStream_t stream);
#endif

}  // namespace dsdtm
