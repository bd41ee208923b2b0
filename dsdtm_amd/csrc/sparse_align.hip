// sparse_align.hip — Sprase_ImgAlign::Run on gfx950: persistent pair slots, register-resident patches.
//
// Replaces reference src/Sprase_ImageAlign.cpp:29-60 (Run), :62-166 (GetJocabianMat),
// :169-193 (GetJocabianBA), :240-299 (ComputeResiduals), :301-344 (GaussNewtonSolver).
//
// MI355X design (not a translation of the CPU loops):
//  * One launch runs whole alignments — all levels, all Gauss-Newton iterations, the 6x6 solves and
//    the accept/revert logic — with no host round trip and no global synchronisation. The grid is
//    one workgroup per CU; a workgroup holds PPW independent PAIR SLOTS (NPW patch waves + 1 solver
//    wave each, 12 waves = 3 per SIMD at 168 VGPRs = the CU's whole register file), and every slot
//    pulls frame pairs from a global counter until the batch is exhausted.
//  * lane = patch. A 4x4 patch needs the 6x6 grid of bilinearly interpolated reference
//    intensities around it (reference intensity + the neighbours its central-difference
//    gradients use, :147-158): 32 live doubles per patch, kept in the lane's VGPRs for the whole
//    pyramid level — the register file (512 KB per CU) is the largest on-chip memory, so the
//    "reference patch + Jacobian cache" of the CPU code (896 B/patch in mRefPatch/mJocabianPatch)
//    never exists in memory at all.
//  * inverse-compositional structure: J_px = dx*A + dy*B with A,B per patch (:160), so
//      sum_px J J^T = Sxx A A^T + Sxy (A B^T + B A^T) + Syy B B^T     (per level constants)
//      sum_px J r   = A * sum(dx r) + B * sum(dy r)
//    Per pixel the loop is 4 bilinear FMAs + residual + 3 accumulations instead of the CPU's 36+6
//    MACs. H changes between iterations only when the set of visible patches does: every 16-lane
//    row caches its H partial with the visibility ballot it was built for; the all-visible H of a
//    level is published right after the precompute and factorised by the solver WHILE the first
//    pass runs — into H^+ (one lane-parallel substitution of the six unit vectors), so that every
//    iteration's solve is the matrix-vector product x = H^+ b.
//  * the 7x7 (ref) u8 footprints are gathered with one dwordx3 load per row straight from the packed
//    pyramid, all seven in flight together (a scattered wave-load costs ~64 L1 tag lookups whatever
//    its width, and every row is its own cold cache line). The 5x5 (cur) footprint of a patch moves
//    by a fraction of a pixel per iteration: it is gathered once per level into a 5-row x 12-byte
//    window per patch in LDS and re-read from there; a lane refills its window only when the
//    floor position leaves it (the CU's 32 KB L1 cannot hold the 1500 lines a pair-level touches).
//  * reductions: DPP row rotations inside 16-lane rows (no LDS traffic), one LDS slot per row; the
//    slot's SOLVER wave (owns no patches, so its registers and the patch registers never share live
//    ranges) sums the partials lane-parallel, computes x = H^+ b, a series SE(3) exp and the
//    accept/revert logic, and publishes R|t + a control word through LDS. In an idle window of the
//    current pair it also stages the next pair's prologue (pose block in LDS, feature lines warmed
//    with LDS-DMA), so that a slot switches pairs without dependent HBM round trips.
//  * hand-over between the waves of a slot uses monotonic LDS counters (B0/BH/B1/B2/ACK below), not
//    s_barrier, so the slots of a workgroup run independently: one slot's solve overlaps the
//    other's pass.
//  * all decision-carrying arithmetic is FP64 (chi2 accept/revert compares values that differ in
//    the 5th-6th digit).
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

#include <mutex>

#include "../../include/dsdtm_amd.h"
#include "device_math.h"
#include "kernels.h"

namespace dsdtm {

// partial sums of one 16-lane DPP row (register kernel) or of one wave (workspace kernel) in LDS
struct WavePartial {
    double b[6];
    double chi2;
    int cnt;
    int h_changed;
    int n_ref;
    int pad;
    double H[21];
};
static_assert(offsetof(WavePartial, chi2) == 48 && offsetof(WavePartial, H) == 72, "solver_step indexes WavePartial as doubles");

// Pose block of a pair: what solver_init computes. BlockState holds the live one (`u`) and a staged
// one (`nx`) that the solver prepares for its slot's next pair while the current one is running.
struct UnitPose {
    double q[4], t[3];          // T_c2r
    double qr[4], tr[3];        // T_ref_w
    double Cref[3];             // reference camera centre in world (Frame::mOw)          | published to the
    double R[9], tt[3];         // rotation matrix + translation of T_c2r for the pass    | patch waves
};
constexpr int UNIT_POSE_DOUBLES = sizeof(UnitPose) / sizeof(double);
static_assert(UNIT_POSE_DOUBLES == 29, "commit_next copies UnitPose lane-parallel");

struct BlockState {
    UnitPose u;                 // live: q/t/qr/tr solver-private, Cref/R/tt read by the patch waves
    // solver-private state
    double qo[4], to[3];        // T_c2r before the last accepted step (tT_c2rOld)
    double Rs[9];               // rotation matrix of the quaternion state u.q (the published u.R is Rs * dR of the
                                // last step in matrix form: the same rotation to ~1e-16, available ~1 k cycles earlier)
    double chi2;
    double Hsum[21];            // sum of the per-row H partials of the last change of the visible set
    double Hinv[36];            // H^+ = P^T L^-T D^+ L^-1 P by columns (ldlt6_factor + ldlt6_apply on the unit vectors:
                                // Eigen's pivoted LDLT with pseudo-inverse of D), formed once per change of the
                                // visible set; every iteration then only needs x = H^+ b
    double bsum[7];             // b totals + chi2 total of the current iteration
    UnitPose nx;                // staged pose block of pair nx_pair (prepare_next)
    int nx_pair;                // -1: nothing staged
    // published to the patch waves
    int ctrl;                   // 0 continue, 1 level finished
    int n_vis;
    // pair-local synchronisation (used when several pairs share a workgroup): monotonic counters
    unsigned arrive;            // +1 per patch wave whose pass partials are in LDS (B1)
    unsigned seq;               // number of states published by the solver (1 after solver_init)
    unsigned arrive_h;          // +1 per patch wave whose all-visible H partials are in LDS (BH); a counter of
                                // its own: a wave signals BH and B1 back to back without waiting in between
    unsigned ack;               // +1 per patch wave that is completely done with the slot's current pair: the
                                // solver waits for all of them before it rewrites pair/run/ctrl/R for the next one
    int pair;                   // pair index this slot is working on (>= n_pairs: batch exhausted)
    int run;                    // 0: the pair is skipped (Min_fts rule), handled by the solver alone
};

// ---- pair-local synchronisation ------------------------------------------------------------
// With PPW > 1 pairs per workgroup each pair runs its own pass/solve pipeline; s_barrier would
// couple them, so patch waves and the pair's solver wave hand over through two monotonic LDS
// counters instead. All waves of a workgroup are co-resident, every wait has its producer in
// flight, and every spin is bounded (a broken protocol ends the kernel instead of hanging the GPU).
constexpr unsigned SPIN_LIMIT = 1u << 24;
// s_sleep units between two polls: the solver waiting for the patch waves (and for acknowledgements) / the patch
// waves waiting for the solver (polling less often leaves issue slots and LDS to the others: +1 %)
constexpr int SLEEP_ARRIVE = 1, SLEEP_SEQ = 4;
// A spin that runs out means the protocol is broken (or, in the multi-CU kernels, that a partner workgroup is not
// resident): the wave leaves the wait (so the kernel always terminates) and raises the launch's timeout word — a
// host-mapped word owned by the launching CONTEXT (SAKernelArgs::timeout_flag; multi-CU launches get a word of their
// own), which the host reads after the stream has drained: no device-global state, nothing shared between contexts.
__device__ __forceinline__ void spin_timeout(unsigned* flag) {
    __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__device__ __forceinline__ void pair_signal_arrive(unsigned* counter, int lane) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // this wave's LDS stores first
    if (lane == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void pair_wait_arrive(unsigned* counter, unsigned target, unsigned* tflag) {
    unsigned spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target && ++spins < SPIN_LIMIT)
        __builtin_amdgcn_s_sleep(SLEEP_ARRIVE);
    if (spins >= SPIN_LIMIT) spin_timeout(tflag);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ void pair_publish(BlockState& s, unsigned seq, int lane) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_store(&s.seq, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void pair_wait_seq(BlockState& s, unsigned seq, unsigned* tflag) {
    unsigned spins = 0;
    while (__hip_atomic_load(&s.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < seq && ++spins < SPIN_LIMIT)
        __builtin_amdgcn_s_sleep(SLEEP_SEQ);
    if (spins >= SPIN_LIMIT) spin_timeout(tflag);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__device__ __forceinline__ void store_se3(double* q, double* t, const SE3d& T) {
    q[0] = T.qw; q[1] = T.qx; q[2] = T.qy; q[3] = T.qz;
    t[0] = T.tx; t[1] = T.ty; t[2] = T.tz;
}
__device__ __forceinline__ SE3d load_se3(const double* q, const double* t) {
    SE3d T;
    T.qw = q[0]; T.qx = q[1]; T.qy = q[2]; T.qz = q[3];
    T.tx = t[0]; T.ty = t[1]; T.tz = t[2];
    return T;
}

// 12 bytes at a 4-byte aligned address as ONE global_load_dwordx3: a scattered wave-load costs ~64
// L1 tag lookups whatever its width, so footprint rows are fetched with as few instructions as possible.
struct __attribute__((packed, aligned(4))) U32x3 { uint32_t a, b, c; };
struct __attribute__((packed, aligned(4))) U32x2 { uint32_t a, b; };

// Footprint gathers: plain loads (measured: `nt` on them costs 8 %).
__device__ __forceinline__ U32x3 gather_x3(const uint32_t* p) { return *(const U32x3*)p; }
__device__ __forceinline__ U32x2 gather_x2(const uint32_t* p) { return *(const U32x2*)p; }

// byte k (0..3) of a dword as double (v_cvt_f32_ubyteK + v_cvt_f64_f32; exact)
__device__ __forceinline__ double ub(uint32_t w, int k) {
    return (double)(float)((w >> (8 * k)) & 0xffu);
}

// Per-patch state that lives across the Gauss-Newton iterations of one level.
// The interpolated reference intensities are doubles, what the reference caches in mRefPatch. (They are EXACT: the
// subpixel offsets of a float pixel position >= 3 * 2^level have at most 22 fractional bits each, so every weighted
// sum of four bytes fits 8 + 44 bits — no rounding whatever the order of operations.)
struct PatchRegs {
    double g[6][6];   // bilinear reference intensities on the 6x6 grid (corners unused)
    double X[3];      // 3-D point in the reference camera, bearing * |P_w - C_ref| (:117-119), kept in
                      // normalised form (x/z, y/z, 1/z)
    bool valid;
};

// LLVM's LICM would hoist everything derived from the per-level patch state (the 32 gradient
// differences, A/B, the 21 Hessian entries) out of the Gauss-Newton loop and keep it live in
// VGPRs — several hundred bytes per lane, which is exactly the cache this kernel avoids. An empty
// asm with a read-write VGPR operand makes the state opaque once per iteration (no instruction is
// emitted), so derived values are recomputed where they are used.
__device__ __forceinline__ void pin_patch(PatchRegs& P) {
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            if ((r == 0 || r == 5) && (c == 0 || c == 5)) continue;
            asm volatile("" : "+v"(P.g[r][c]));
        }
    asm volatile("" : "+v"(P.X[0]), "+v"(P.X[1]), "+v"(P.X[2]));
}

// GetJocabianBA (:169-193) scaled by f*scale (:160; Camera.f, quirk Q1): the non-zero entries
// of Jt.row(0)*fs = [a0,0,a1,a2,a3,a4] and Jt.row(1)*fs = [0,b0,b1,b2,b3,b4].
// Xn is the point in normalised form (x/z, y/z, 1/z), computed once per alignment, so that neither
// this nor the projection needs a division inside the Gauss-Newton loop:
//   x/z^2 = xn*zi, x*y/z^2 = xn*yn, 1 + x^2/z^2 = 1 + xn^2, ...
__device__ __forceinline__ void patch_AB(double fs, const double* Xn, double* A, double* B) {
    const double xn = Xn[0], yn = Xn[1], zi = Xn[2];
    const double zfs = zi * fs;
    A[0] = -zfs;                         // J(0,0) = -1/z
    A[1] = xn * zfs;                     // J(0,2) = x/z^2
    A[2] = (xn * yn) * fs;               // J(0,3) = x*y/z^2
    A[3] = -(1.0 + xn * xn) * fs;        // J(0,4) = -(1 + x^2/z^2)
    A[4] = yn * fs;                      // J(0,5) = y/z
    B[0] = -zfs;                         // J(1,1) = -1/z
    B[1] = yn * zfs;                     // J(1,2) = y/z^2
    B[2] = (1.0 + yn * yn) * fs;         // J(1,3) = 1 + y^2/z^2
    B[3] = -(xn * yn) * fs;              // J(1,4) = -x*y/z^2
    B[4] = -xn * fs;                     // J(1,5) = -x/z
}

// Level-independent part of GetJocabianMat (reference :84-119) for one feature: the feature
// columns are read from HBM once per alignment, not once per level.
struct FeatureRegs {
    float px, py;     // Feature::mpx
    double X[3];      // bearing * |P_w - C_ref|  (:117-119) as (x/z, y/z, 1/z)
    bool ok;          // mbInitial && P_w != 0 (:86, :95)
};

struct FeatureRaw {   // the raw loads, issued before the first barrier so their latency overlaps solver_init
    float px, py;
    double b0, b1, b2, w0, w1, w2;
    bool initial;
};

__device__ __forceinline__ FeatureRaw load_feature_raw(const SAKernelArgs& a, size_t fidx, bool live) {
    FeatureRaw r;
    r.px = r.py = 0.0f; r.b0 = r.b1 = r.b2 = r.w0 = r.w1 = r.w2 = 0.0; r.initial = false;
    if (live) {
        r.initial = a.initial[fidx] != 0;
        r.px = a.px_xy[2 * fidx]; r.py = a.px_xy[2 * fidx + 1];
        r.b0 = a.bearing[3 * fidx]; r.b1 = a.bearing[3 * fidx + 1]; r.b2 = a.bearing[3 * fidx + 2];
        r.w0 = a.p_world[3 * fidx]; r.w1 = a.p_world[3 * fidx + 1]; r.w2 = a.p_world[3 * fidx + 2];
    }
    return r;
}

__device__ __forceinline__ FeatureRegs make_feature(const FeatureRaw& r, const double* Cref) {
    FeatureRegs f;
    f.px = r.px; f.py = r.py;
    const bool is_zero = (r.w0 == 0.0 && r.w1 == 0.0 && r.w2 == 0.0);    // :95 isZero(0)
    f.ok = r.initial && !is_zero;
    const double d0 = r.w0 - Cref[0], d1 = r.w1 - Cref[1], d2 = r.w2 - Cref[2];
    const double depth = sqrt(d0 * d0 + d1 * d1 + d2 * d2);
    const double x = r.b0 * depth, y = r.b1 * depth, z = r.b2 * depth;
    const double zi = 1.0 / z;           // the only division per feature and alignment
    f.X[0] = x * zi; f.X[1] = y * zi; f.X[2] = zi;
    return f;
}

// Per-level part of GetJocabianMat for one feature (reference :89-100, :123-162), producing the
// register-resident state. Branch-free: lanes without a valid patch run the same instructions on a
// harmless address (the level's first bytes) and end up with P.valid == false and an unused grid, so
// the seven row gathers and the arithmetic sit in straight-line code. (Gathering the NEXT level's
// rows while a level waits for its solver, parked in LDS, was measured: the level start gets 5 k
// cycles shorter and the launch no faster — the cold lines cost the same L1 fill time wherever they
// are issued. Non-temporal gathers: 8 % slower.)
typedef __attribute__((address_space(3))) uint32_t LdsU32;
struct RefGeom {
    bool valid;       // mbInitial && P_w != 0 && inside the level's border (:86, :95-100)
    int fu, fv;       // floor of the level position (3,3 for lanes without a patch)
    double su, sv;    // subpixel offsets
};
__device__ __forceinline__ RefGeom ref_geom(const FeatureRegs& F, const LevelGeom& lg, int level) {
    RefGeom g;
    const float scale_f = 1.0f / (float)(1 << level);                  // :65 tScale (float)
    const double scale = (double)scale_f;
    const double px = (double)F.px * scale;                            // :89-91
    const double py = (double)F.py * scale;
    const double boarder = 3.0;                                        // :67 int(0.5*4+1)
    g.valid = F.ok && !(px - boarder < 0 || py - boarder < 0 || px + boarder >= (double)lg.w ||
                        py + boarder >= (double)lg.h || !(px == px) || !(py == py));
    const double fu_d = floor(px), fv_d = floor(py);                   // :123-132
    g.fu = g.valid ? (int)fu_d : 3; g.fv = g.valid ? (int)fv_d : 3;
    g.su = px - fu_d; g.sv = py - fv_d;
    return g;
}
__device__ __forceinline__ uint32_t ref_row_offset(const LevelGeom& lg, const RefGeom& g, int r) {
    return g.valid ? lg.off + (uint32_t)(g.fv - 3 + r) * (uint32_t)lg.stride + (uint32_t)(g.fu - 3) : lg.off;
}
// 7x7 u8 footprint rows fv-3..fv+3, cols fu-3..fu+3, fetched as aligned dwords. All seven row gathers
// are issued before the first one is consumed: the rows are cold (each its own cache line), and seven
// dependent round trips would be seven HBM latencies per level.
__device__ __forceinline__ void ref_rows_issue(const SAKernelArgs& a, const LevelGeom& lg, const uint8_t* __restrict__ ref_base,
                                               const RefGeom& g, U32x3* w) {
    const uint32_t* __restrict__ img32 = (const uint32_t*)ref_base;
    const uint32_t last_dw = (uint32_t)(a.pyr_pitch >> 2) - 1u;
#pragma unroll
    for (int r = 0; r < 7; ++r) {
        // (dw+2 may be the dword after the last pixel of the pyramid: shift the window back by one
        // dword there; wave-uniformly false except at the very end of the allocation)
        w[r] = gather_x3(img32 + min(ref_row_offset(lg, g, r) >> 2, last_dw - 2u));
    }
}
// row r -> its 7 bytes: rlo = bytes 0..3, rhi = bytes 4..6 (+ one unused)
__device__ __forceinline__ void ref_rows_unpack(const SAKernelArgs& a, const LevelGeom& lg, const RefGeom& g, const U32x3* w,
                                                uint32_t* rlo, uint32_t* rhi) {
    const uint32_t last_dw = (uint32_t)(a.pyr_pitch >> 2) - 1u;
#pragma unroll
    for (int r = 0; r < 7; ++r) {
        const uint32_t o = ref_row_offset(lg, g, r);
        const uint32_t dw = o >> 2;
        const uint32_t sh = (o & 3u) * 8u;
        const uint32_t dwc = min(dw, last_dw - 2u);
        const uint32_t w0 = (dwc == dw) ? w[r].a : w[r].b, w1 = (dwc == dw) ? w[r].b : w[r].c, w2 = (dwc == dw) ? w[r].c : 0u;
        rlo[r] = __builtin_amdgcn_alignbit(w1, w0, sh);
        rhi[r] = __builtin_amdgcn_alignbit(w2, w1, sh);
    }
}
// The grid g[r][c] = bilinear value at pixel (fv-3+r, fu-3+c) + subpixel offset, built row by row.
// Patch pixel (i,k), i,k in 0..3, is g[i+1][k+1] (offsets -2..+1 from floor, quirk Q4); its
// gradient neighbours are g[i+1][k], g[i+1][k+2], g[i][k+1], g[i+2][k+1] (:150-158).
__device__ __forceinline__ void grid_from_rows(const FeatureRegs& F, const RefGeom& g, const uint32_t* rlo, const uint32_t* rhi,
                                               PatchRegs& P) {
    // lanes without a valid patch get a harmless point: their A, B (patch_AB) stay finite, so that zeroing the three
    // gradient sums is enough to keep them out of the H block (patch_hess_factors)
    P.X[0] = g.valid ? F.X[0] : 0.0; P.X[1] = g.valid ? F.X[1] : 0.0; P.X[2] = g.valid ? F.X[2] : 1.0;
    P.valid = g.valid;
    const double omx = 1.0 - g.su, omy = 1.0 - g.sv;
    const double w00 = omx * omy, w01 = g.su * omy, w10 = omx * g.sv, w11 = g.su * g.sv;
    double top[7], bot[7];
#pragma unroll
    for (int r = 0; r < 7; ++r) {
        const uint32_t lo = rlo[r], hi = rhi[r];
        bot[0] = ub(lo, 0); bot[1] = ub(lo, 1); bot[2] = ub(lo, 2); bot[3] = ub(lo, 3);
        bot[4] = ub(hi, 0); bot[5] = ub(hi, 1); bot[6] = ub(hi, 2);
        if (r > 0) {
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                if ((r - 1 == 0 || r - 1 == 5) && (c == 0 || c == 5)) { P.g[r - 1][c] = 0.0; continue; }   // corners: unused
                P.g[r - 1][c] = (w00 * top[c] + w01 * top[c + 1] + w10 * bot[c] + w11 * bot[c + 1]);
            }
        }
#pragma unroll
        for (int c = 0; c < 7; ++c) top[c] = bot[c];
        __builtin_amdgcn_sched_barrier(0);   // one footprint row in flight (two: measured flat, profiles/r06_headline.txt)
    }
}

__device__ __forceinline__ void precompute_patch(const SAKernelArgs& a, const LevelGeom& lg, int level,
                                                 const uint8_t* __restrict__ ref_base,  // pair's ref pyramid
                                                 const FeatureRegs& F, PatchRegs& P) {
    const RefGeom g = ref_geom(F, lg, level);
    uint32_t rlo[7], rhi[7];
    {
        U32x3 w[7];
        ref_rows_issue(a, lg, ref_base, g, w);
        ref_rows_unpack(a, lg, g, w, rlo, rhi);
    }
    __builtin_amdgcn_sched_barrier(0);
    grid_from_rows(F, g, rlo, rhi, P);
}

// Per-patch Gauss-Newton matrix  Sxx A A^T + Sxy (A B^T + B A^T) + Syy B B^T  (upper triangle,
// row-major index q). PatchHess holds the per-patch factors; entry<I,J>() evaluates one element so
// that a caller can reduce the 21 entries one at a time without keeping all of them live.
struct PatchHess {
    double sxx, sxy, syy;
    double A[6], B[6];
    // A = [a0, 0, A2..A5], B = [0, b0, B2..B5] (GetJocabianBA's zeros): with P_j = sxx A_j + sxy B_j and
    // Q_j = sxy A_j + syy B_j the entry (i, j) is A_i P_j + B_i Q_j — two operations instead of six, and nothing is
    // multiplied by the structural zeros (the level start is bound by VALU issue like the pass)
    double Pj[6], Qj[6];
    template <int I, int J>
    __device__ __forceinline__ double entry() const {
        if constexpr (I == 0 && J == 0) return A[0] * Pj[0];
        else if constexpr (I == 0 && J == 1) return A[0] * Pj[1];
        else if constexpr (I == 0) return A[0] * Pj[J];
        else if constexpr (I == 1 && J == 1) return B[1] * Qj[1];
        else if constexpr (I == 1) return B[1] * Qj[J];
        else return A[I] * Pj[J] + B[I] * Qj[J];
    }
};

// The per-patch factors from the three gradient sums (sums of the doubled central differences, see the callers) and
// the patch's point. `use` = the patch contributes (valid / visible): otherwise its three sums, hence all 21 entries,
// are zero (Xn must still be finite).
__device__ __forceinline__ PatchHess patch_hess_from_sums(double sxx, double sxy, double syy, const double* Xn, double fs, bool use) {
    PatchHess h;
    // the 0.5 of the central differences as ONE exact scaling of the three sums (power of two)
    sxx = use ? sxx * 0.25 : 0.0; sxy = use ? sxy * 0.25 : 0.0; syy = use ? syy * 0.25 : 0.0;
    h.sxx = sxx; h.sxy = sxy; h.syy = syy;
    double A5[5], B5[5];
    patch_AB(fs, Xn, A5, B5);
    h.A[0] = A5[0]; h.A[1] = 0.0;   h.A[2] = A5[1]; h.A[3] = A5[2]; h.A[4] = A5[3]; h.A[5] = A5[4];
    h.B[0] = 0.0;   h.B[1] = B5[0]; h.B[2] = B5[1]; h.B[3] = B5[2]; h.B[4] = B5[3]; h.B[5] = B5[4];
    h.Pj[0] = sxx * h.A[0]; h.Qj[0] = sxy * h.A[0];               // B_0 = 0
    h.Pj[1] = sxy * h.B[1]; h.Qj[1] = syy * h.B[1];               // A_1 = 0
#pragma unroll
    for (int j = 2; j < 6; ++j) {
        h.Pj[j] = sxx * h.A[j] + sxy * h.B[j];
        h.Qj[j] = sxy * h.A[j] + syy * h.B[j];
    }
    return h;
}
__device__ __forceinline__ PatchHess patch_hess_factors(const PatchRegs& P, double fs, bool use = true) {
    double sxx = 0.0, sxy = 0.0, syy = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double dx = P.g[i + 1][k + 2] - P.g[i + 1][k];
            const double dy = P.g[i + 2][k + 1] - P.g[i][k + 1];
            sxx += dx * dx; sxy += dx * dy; syy += dy * dy;
        }
    return patch_hess_from_sums(sxx, sxy, syy, P.X, fs, use);
}

// entry q (row-major upper triangle, 0..20) by compile-time index; 0 beyond
template <int Q>
__device__ __forceinline__ double patch_hess_entry_q(const PatchHess& h) {
    if constexpr (Q >= 21) return 0.0;
    else {
        constexpr int I = Q < 6 ? 0 : Q < 11 ? 1 : Q < 15 ? 2 : Q < 18 ? 3 : Q < 20 ? 4 : 5;
        constexpr int J = I + (Q - (I * 6 - (I * (I - 1)) / 2));
        return h.entry<I, J>();
    }
}
// The 21 entries of the per-patch matrix summed over each 16-lane DPP row, three packed butterflies of eight
// values (row_reduce8): a lane ends up with the row totals of entries 8 g + row_reduce8_index(lane), g = 0..2, and
// the lanes with bit 2 clear store them.
template <int G, typename HP>
__device__ __forceinline__ void patch_hess_rows_group(const PatchHess& ph, bool use, int lane, HP Hout) {
    double v[8];
    (void)use;       // the factors were built with `use`: a patch that does not contribute has zero sums
    v[0] = patch_hess_entry_q<8 * G + 0>(ph); v[1] = patch_hess_entry_q<8 * G + 1>(ph);
    v[2] = patch_hess_entry_q<8 * G + 2>(ph); v[3] = patch_hess_entry_q<8 * G + 3>(ph);
    v[4] = patch_hess_entry_q<8 * G + 4>(ph); v[5] = patch_hess_entry_q<8 * G + 5>(ph);
    v[6] = patch_hess_entry_q<8 * G + 6>(ph); v[7] = patch_hess_entry_q<8 * G + 7>(ph);
    const double t = row_reduce8(v, lane);
    const int q = 8 * G + row_reduce8_index(lane);
    if (!(lane & 4) && q < 21) Hout[q] = t;
    __builtin_amdgcn_sched_barrier(0);       // one group of eight live at a time
}
template <typename HP>   // double* or an LDS-qualified double*
__device__ __forceinline__ void patch_hess_rows(const PatchHess& ph, bool use, int lane, HP Hout) {
    patch_hess_rows_group<0>(ph, use, lane, Hout);
    patch_hess_rows_group<1>(ph, use, lane, Hout);
    patch_hess_rows_group<2>(ph, use, lane, Hout);
}

// calls f(q, value) for the 21 upper-triangular entries in row-major order
template <int I, int J, typename F>
__device__ __forceinline__ void patch_hess_foreach(const PatchHess& h, F&& f) {
    f(I * 6 - (I * (I - 1)) / 2 + (J - I), h.entry<I, J>());
    if constexpr (J < 5) patch_hess_foreach<I, J + 1>(h, f);
    else if constexpr (I < 5) patch_hess_foreach<I + 1, I + 1>(h, f);
}

// WIN_NL > 0: the current-image footprint is served from a per-patch WINDOW in LDS. Between two
// Gauss-Newton iterations a patch moves by a fraction of a pixel, but every footprint row is its own
// cache line and a pair-level touches far more lines (300 x 5 x 128 B) than the CU's 32 KB L1 holds,
// so each pass would re-fetch all of them from L2 (measured: 5.5 k cycles per pass at levels 0/1
// against 2.9 k when the gathers hit). The window holds WIN_ROWS rows x 12 bytes (3 aligned
// dwords from column u0-3 of rows v0-(WIN_ROWS-1)/2 ..) around the position (u0, v0) of the pass that
// filled it; a pass whose floor position is within [u0-1, u0+3] x [v0-t, v0+t], t = (WIN_ROWS-5)/2,
// reads LDS only, otherwise the lane refills its window from the pyramid. Layout [row*3+dword][lane]
// (`win` points at this lane's column, planes are WIN_NL dwords apart): conflict-free.
constexpr int WIN_ROWS = 5;
constexpr uint32_t WIN_EMPTY = 0xffffffffu;


// tT_c2r * X -> pixel of the current level (reference :254-262). Returns the visibility test of :262.
__device__ __forceinline__ bool project_patch(const SAKernelArgs& a, const LevelGeom& lg, double scale, const double* X,
                                              const double* __restrict__ sR, const double* __restrict__ st,
                                              double& u, double& v) {
    // :254 tT_c2r * X, divided through by z > 0 (the projection below is scale invariant):
    // (R X + t)/z = R (x/z, y/z, 1) + t/z   (R, t are LDS broadcasts)
    const double pxc = sR[0] * X[0] + sR[1] * X[1] + sR[2] + st[0] * X[2];
    const double pyc = sR[3] * X[0] + sR[4] * X[1] + sR[5] + st[1] * X[2];
    const double pzc = sR[6] * X[0] + sR[7] * X[1] + sR[8] + st[2] * X[2];
    // Camera2Pixel (src/Camera.cpp:167-171), * tScale (:255)
    // one reciprocal for both coordinates (the reference divides twice; <= 1 ulp on u,v)
    const double izc = rcp_f64_newton(pzc);
    u = ((double)a.fx * pxc * izc + (double)a.cx) * scale;
    v = ((double)a.fy * pyc * izc + (double)a.cy) * scale;
    // :262 with mnboarder = 3 on floored ints: floor(u) >= 3 && floor(u)+3 < cols (NaN fails). Evaluated without
    // short-circuits: the pass is bound by VALU issue, and nested early-outs made the compiler re-zero the seven
    // result registers on every arm (21 moves on the visible path)
    // four v_cmp into SGPR pairs + three s_and (scalar unit) + the combined mask as the lane condition; written with `&`
    // on the bools, the vectoriser packs the four bits into an i4 and the backend spends ~15 VALU instructions on it
    const unsigned long long m = __builtin_amdgcn_ballot_w64(u >= 3.0) & __builtin_amdgcn_ballot_w64(u < (double)(lg.w - 3)) &
                                 __builtin_amdgcn_ballot_w64(v >= 3.0) & __builtin_amdgcn_ballot_w64(v < (double)(lg.h - 3));
    return __builtin_amdgcn_inverse_ballot_w64(m);
}

// A window fill in flight: the row gathers have been issued, window_commit parks them in LDS.
struct WinFill {
    U32x3 w[WIN_ROWS];
    int u_i, v_i;
};

__device__ __forceinline__ void window_issue(const SAKernelArgs& a, const LevelGeom& lg, const uint8_t* __restrict__ cur_base,
                                             int u_i, int v_i, WinFill& f) {
    constexpr int HR = (WIN_ROWS - 1) / 2;
    const uint32_t* __restrict__ img32 = (const uint32_t*)cur_base;
    const uint32_t last_dw = (uint32_t)(a.pyr_pitch >> 2) - 1u;
    f.u_i = u_i; f.v_i = v_i;
    // (u_i >= 3, HR <= v_i < h - HR: in range; the last dword of the allocation is handled as in
    // precompute_patch)
#pragma unroll
    for (int r = 0; r < WIN_ROWS; ++r) {
        const uint32_t o = lg.off + (uint32_t)(v_i - HR + r) * (uint32_t)lg.stride + (uint32_t)(u_i - 3);
        f.w[r] = gather_x3(img32 + min(o >> 2, last_dw - 2u));
    }
}

template <int WIN_NL>
__device__ __forceinline__ void window_commit(const SAKernelArgs& a, const LevelGeom& lg, const WinFill& f, LdsU32* win, uint32_t& worg) {
    constexpr int HR = (WIN_ROWS - 1) / 2;
    const uint32_t last_dw = (uint32_t)(a.pyr_pitch >> 2) - 1u;
#pragma unroll
    for (int r = 0; r < WIN_ROWS; ++r) {
        const uint32_t o = lg.off + (uint32_t)(f.v_i - HR + r) * (uint32_t)lg.stride + (uint32_t)(f.u_i - 3);
        const bool in = min(o >> 2, last_dw - 2u) == (o >> 2);
        const uint32_t d0 = in ? f.w[r].a : f.w[r].b, d1 = in ? f.w[r].b : f.w[r].c, d2 = in ? f.w[r].c : 0u;
        // parked with column u_i - 3 in byte 0 of the row's first dword (9..12 valid bytes): a pass then needs the same
        // dword offset and byte shift for all five rows and no per-row address arithmetic (the pass is bound by VALU
        // issue; this is done once per fill)
        const uint32_t sh = (o & 3u) * 8u;
        win[(r * 3 + 0) * WIN_NL] = __builtin_amdgcn_alignbit(d1, d0, sh);
        win[(r * 3 + 1) * WIN_NL] = __builtin_amdgcn_alignbit(d2, d1, sh);
        win[(r * 3 + 2) * WIN_NL] = d2 >> sh;
    }
    worg = (uint32_t)f.u_i | ((uint32_t)f.v_i << 16);
}

// ComputeResiduals for one patch (reference :252-296). Returns visibility; produces chi2 and
// b = sum_px J*res (as A*gx + B*gy) of this patch.
template <int WIN_NL = 0>
__device__ __forceinline__ bool residual_patch(const SAKernelArgs& a, const LevelGeom& lg, double scale, double fs,
                                               const uint8_t* __restrict__ cur_base, const PatchRegs& P,
                                               const double* __restrict__ sR, const double* __restrict__ st,
                                               double& chi2, double* b, LdsU32* win = nullptr, uint32_t* worg = nullptr) {
    chi2 = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) b[i] = 0.0;
    double u, v;
    if (!(project_patch(a, lg, scale, P.X, sR, st, u, v) & P.valid)) return false;    // ONE exit for lanes without a visible patch
    const double fu_d = floor(u), fv_d = floor(v);
    const int u_i = (int)fu_d, v_i = (int)fv_d;
    const double su = u - fu_d, sv = v - fv_d;
    const double tl = (1.0 - su) * (1.0 - sv);
    const double trw = su * (1.0 - sv);
    const double bl = (1.0 - su) * sv;
    const double br = su * sv;

    const uint32_t* __restrict__ img32 = (const uint32_t*)cur_base;
    uint32_t wlo[5], whi[5];
    if constexpr (WIN_NL > 0) {
        constexpr int T = (WIN_ROWS - 5) / 2, HR = (WIN_ROWS - 1) / 2;
        int u0 = (int)(*worg & 0xffffu), v0 = (int)(*worg >> 16);
        const bool covered = (unsigned)(u_i - u0 + 1) <= 4u && (unsigned)(v_i - v0 + T) <= (unsigned)(2 * T);
        if (!covered) {
            WinFill f;
            window_issue(a, lg, cur_base, u_i, v_i, f);
            window_commit<WIN_NL>(a, lg, f, win, *worg);
            u0 = u_i; v0 = v_i;
        }
        static_assert(WIN_ROWS == 5, "aligned windows: the window's rows are the pass's rows");
        {
            const uint32_t p = (uint32_t)(u_i - u0 + 1);          // byte of column u_i - 2 in the parked rows: 0..4
            const LdsU32* wc = win + (p >> 2) * (uint32_t)WIN_NL; // dword 0 or 1 of every row
            const uint32_t sh = (p & 3u) * 8u;
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                const uint32_t lo = wc[(r * 3) * WIN_NL], hi = wc[(r * 3 + 1) * WIN_NL];
                wlo[r] = __builtin_amdgcn_alignbit(hi, lo, sh);     // bytes 0..3 of the row
                whi[r] = hi >> sh;                                  // byte 4 in bits 0..7
            }
        }
    } else {
        U32x2 wr[5];
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            const uint32_t o = lg.off + (uint32_t)(v_i - 2 + r) * (uint32_t)lg.stride + (uint32_t)(u_i - 2);
            wr[r] = gather_x2(img32 + (o >> 2));                // one global_load_dwordx2
        }
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            const uint32_t o = lg.off + (uint32_t)(v_i - 2 + r) * (uint32_t)lg.stride + (uint32_t)(u_i - 2);
            const uint32_t sh = (o & 3u) * 8u;
            wlo[r] = __builtin_amdgcn_alignbit(wr[r].b, wr[r].a, sh);  // bytes 0..3 of the row
            whi[r] = wr[r].b >> sh;                                     // byte 4 in bits 0..7
        }
    }
    // two footprint rows in flight, ping-ponged by the (static) row parity so that no register
    // copies are needed between rows
    double rw[2][5];
    rw[0][0] = ub(wlo[0], 0); rw[0][1] = ub(wlo[0], 1); rw[0][2] = ub(wlo[0], 2); rw[0][3] = ub(wlo[0], 3); rw[0][4] = ub(whi[0], 0);
    double c2a = 0.0, c2b = 0.0, gxa = 0.0, gxb = 0.0, gya = 0.0, gyb = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        double* top = rw[i & 1];
        double* bot = rw[(i + 1) & 1];
        bot[0] = ub(wlo[i + 1], 0); bot[1] = ub(wlo[i + 1], 1); bot[2] = ub(wlo[i + 1], 2);
        bot[3] = ub(wlo[i + 1], 3); bot[4] = ub(whi[i + 1], 0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double cur = tl * top[k] + trw * top[k + 1] + bl * bot[k] + br * bot[k + 1];   // :281
            const double res = cur - P.g[i + 1][k + 1];                                   // :282
            const double ddx = P.g[i + 1][k + 2] - P.g[i + 1][k];      // 2*dx (:150)
            const double ddy = P.g[i + 2][k + 1] - P.g[i][k + 1];      // 2*dy (:155)
            if (k & 1) { c2b += res * res; gxb += ddx * res; gyb += ddy * res; }
            else       { c2a += res * res; gxa += ddx * res; gya += ddy * res; }
        }
        // keep the rows in program order: without this the scheduler converts the whole 5x5
        // footprint to doubles up front (50 live VGPRs instead of the two rows in flight)
        __builtin_amdgcn_sched_barrier(0);
    }
    chi2 = c2a + c2b;
    // JRes += J*res (:291) with J = dx*A + dy*B, written out in the normalised point (patch_AB's entries) so that the
    // ten products A_i, B_i are never formed: with gX = fs/2 * sum(2dx res), gY likewise and s = xn gX + yn gY
    //   b = [-zi gX, -zi gY, zi s, yn s + gY, -(xn s + gX), yn gX - xn gY]          (11 instead of 24 instructions)
    // (the 0.5 of the central differences is folded into fs: exact, a power of two)
    const double fsh = 0.5 * fs;
    const double gX = (gxa + gxb) * fsh, gY = (gya + gyb) * fsh;
    const double xn = P.X[0], yn = P.X[1], zi = P.X[2];
    const double sxy = xn * gX + yn * gY;
    b[0] = -(zi * gX);
    b[1] = -(zi * gY);
    b[2] = zi * sxy;
    b[3] = yn * sxy + gY;
    b[4] = -(xn * sxy + gX);
    b[5] = yn * gX - xn * gY;
    return true;
}

__device__ __forceinline__ void stats_clear(const SAKernelArgs& a, int pair) {
    if (!a.stats) return;
    dsdtm_align_stats* st = a.stats + pair;
    for (int l = 0; l < DSDTM_MAX_LEVELS; ++l) {
        st->iters[l] = 0; st->n_ref[l] = 0; st->n_vis[l] = 0; st->exit_code[l] = 0; st->chi2[l] = 0.0;
    }
}

typedef __attribute__((address_space(3))) BlockState LdsBlockState;   // ds_read/ds_write instead of flat accesses

// Solver wave, pair prologue: mT_c2r = cur.pose * ref.pose^-1 (:43); C_ref = (T_ref_w^-1).translation
// (Frame::Set_Pose, src/Frame.cpp:167-174).
// The pose block of a pair from its inputs, into `dst` (LDS). Its callers are out of line, data in
// and out through memory only (like factor_hinv_to_lds): they run once per pair and their SE(3)
// temporaries stay out of the register allocation of the per-iteration solver loop.
typedef __attribute__((address_space(3))) UnitPose LdsUnitPose;
__device__ __forceinline__ void unit_pose_compute(const double* T_ref_w_pair, const double* T_cur_w_pair,
                                                  LdsUnitPose* dst, int lane) {
    const SE3d Tr = se3_from_rt(T_ref_w_pair);
    const SE3d Tri = se3_inverse(Tr);
    const SE3d Tc = se3_from_rt(T_cur_w_pair);
    const SE3d T = se3_mul(Tc, Tri);
    double R[9];
    quat_to_matrix(T, R);
    if (lane == 0) {
        LdsUnitPose& u = *dst;
        u.q[0] = T.qw; u.q[1] = T.qx; u.q[2] = T.qy; u.q[3] = T.qz; u.t[0] = T.tx; u.t[1] = T.ty; u.t[2] = T.tz;
        u.qr[0] = Tr.qw; u.qr[1] = Tr.qx; u.qr[2] = Tr.qy; u.qr[3] = Tr.qz; u.tr[0] = Tr.tx; u.tr[1] = Tr.ty; u.tr[2] = Tr.tz;
        u.Cref[0] = Tri.tx; u.Cref[1] = Tri.ty; u.Cref[2] = Tri.tz;
#pragma unroll
        for (int i = 0; i < 9; ++i) u.R[i] = R[i];
        u.tt[0] = T.tx; u.tt[1] = T.ty; u.tt[2] = T.tz;
    }
}

// the solver-private rest of a pair start; `u` is in place
__device__ __forceinline__ void unit_state_reset(LdsBlockState& s, int lane) {
    if (lane < 7) {
        const double v = lane < 4 ? s.u.q[lane] : s.u.t[lane - 4];    // tT_c2rOld = T_c2r
        if (lane < 4) s.qo[lane] = v; else s.to[lane - 4] = v;
    }
    if (lane >= 7 && lane < 16) s.Rs[lane - 7] = s.u.R[lane - 7];
    if (lane == 0) { s.chi2 = 0.0; s.n_vis = 0; s.ctrl = 0; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// Pair prologue on the spot (nothing was staged): the slow path, ~10 k cycles of HBM latency + SE(3) math
__device__ __attribute__((noinline)) void solver_init(const double* T_ref_w_pair, const double* T_cur_w_pair,
                                                      LdsBlockState* sp, int lane) {
    unit_pose_compute(T_ref_w_pair, T_cur_w_pair, (LdsUnitPose*)&sp->u, lane);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    unit_state_reset(*sp, lane);
}

// Pair prologue ahead of time: called by the solver wave for its slot's NEXT pair in the idle window
// of an iteration of the current one, i.e. while the patch waves run a pass. Stages the pose block
// in s.nx and pulls the pair's feature columns towards this CU (LDS-DMA into a sink: no destination
// registers, nothing waits for them), so that the pair later starts on a 29-double LDS copy and cache
// hits instead of dependent HBM round trips (~15 k cycles per pair start otherwise).
typedef __attribute__((address_space(3))) void LdsVoid;
__device__ __forceinline__ void touch_range(const void* base, size_t bytes, int lane, LdsVoid* sink) {
    const char* p = (const char*)base;
    for (size_t off = (size_t)lane * 64; off < bytes; off += 64 * 64)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + off), sink, 4, 0, 0);
}
__device__ __attribute__((noinline)) void prepare_next(const double* T_ref_w_pair, const double* T_cur_w_pair,
                                                       const float* px, const double* be, const double* pw, const uint8_t* ini,
                                                       int nfm, int pair, LdsBlockState* sp, LdsVoid* sink, int lane) {
    touch_range(px, 8 * (size_t)nfm, lane, sink);
    touch_range(be, 24 * (size_t)nfm, lane, sink);
    touch_range(pw, 24 * (size_t)nfm, lane, sink);
    touch_range(ini, (size_t)nfm, lane, sink);
    unit_pose_compute(T_ref_w_pair, T_cur_w_pair, (LdsUnitPose*)&sp->nx, lane);
    if (lane == 0) sp->nx_pair = pair;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// staged pose block -> live one (lane-parallel LDS copy)
__device__ __forceinline__ void commit_next(LdsBlockState& s, int lane) {
    if (lane < UNIT_POSE_DOUBLES) {
        const __attribute__((address_space(3))) double* src = (const __attribute__((address_space(3))) double*)&s.nx;
        __attribute__((address_space(3))) double* dst = (__attribute__((address_space(3))) double*)&s.u;
        dst[lane] = src[lane];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    unit_state_reset(s, lane);
}

// Factorise BlockState::Hsum and form H^+ in BlockState (LDS). x = H^+ b = sum_j b_j * column_j is the
// substitution H.ldlt().solve(b) the reference runs every iteration (:318) up to rounding (the solve
// is linear in b), including the zeroed components of a rank-deficient system. The six columns are
// the solves of the six unit vectors — one per lane, so all of H^+ costs ONE substitution.
// Deliberately NOT inlined: it runs only when the visible set changed (about once per pyramid level),
// and keeping its ~70 live registers out of solver_step's allocation keeps the whole kernel inside
// the 168-VGPR budget of a 12-wave workgroup without spills on the per-iteration path.
// Deliberately NOT inlined (rare path; keeps its ~70 live registers out of the solver loop's allocation).
__device__ __attribute__((noinline)) void factor_hinv_general(LdsBlockState* sp, int lane) {
    LdsBlockState& s = *sp;
    double H[21], Fm[21], Fdinv[6];
    int tr0, tr1, tr2, tr3, tr4;
    unsigned dmask;
#pragma unroll
    for (int i = 0; i < 21; ++i) H[i] = s.Hsum[i];
    ldlt6_factor(H, Fm, Fdinv, tr0, tr1, tr2, tr3, tr4, dmask);
    const int j = lane < 6 ? lane : 5;                    // lane j < 6: column j = solve(e_j)
    double e[6], col[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) e[i] = (i == j) ? 1.0 : 0.0;
    ldlt6_apply(Fm, Fdinv, tr0, tr1, tr2, tr3, tr4, dmask, e, col);
    if (lane < 6) {
#pragma unroll
        for (int i = 0; i < 6; ++i) s.Hinv[j * 6 + i] = col[i];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// Sum the per-row H partials and form H^+. Lane 6 i + j (< 36) sums entry (i, j) over the NP slots in fixed
// order, the 36 lanes invert the matrix in place (gj6_invert_lanes) and park H^+ in LDS; a matrix that is not safely
// positive definite (no visible patch, rank-deficient) goes through the pivoted LDLT of the general path instead,
// which reproduces Eigen's rank cutoff and pseudo-inverse. Called once per level on the all-visible H —
// speculatively, while the patch waves run the level's first pass — and again only if a row later reports a
// different visible set. (Round 2: 7.7 k cycles per call with the pivoted factorisation + six substitutions.)
// Out of line on purpose: it runs about once per level, and inlined its temporaries pushed the per-iteration solver
// code into spills (18 VGPRs, solve 2.06 k -> 3.3 k cycles). Data in and out through LDS only.
typedef __attribute__((address_space(3))) WavePartial LdsWavePartial;
template <int NP>
__device__ __attribute__((noinline)) void refresh_hinv_to_lds(const LdsWavePartial* s_part, LdsBlockState* sp, int lane) {
    constexpr int WP = sizeof(WavePartial) / sizeof(double);
    LdsBlockState& s = *sp;
    const __attribute__((address_space(3))) double* base = (const __attribute__((address_space(3))) double*)s_part;
    const int l36 = lane < 36 ? lane : 35;
    const int i = l36 / 6, j = l36 - 6 * i;
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    const int q = lo * 6 - (lo * (lo - 1)) / 2 + (hi - lo);          // row-major upper triangle
    double acc = 0.0;
#pragma unroll
    for (int w = 0; w < NP; ++w) acc += base[w * WP + 9 + q];         // H[] starts at double 9 of a slot (all loads in flight)
    if (lane < 36 && i <= j) s.Hsum[q] = acc;                         // for the general path
    const bool ok = gj6_invert_lanes(acc, lane);
    if (ok && lane < 36) s.Hinv[j * 6 + i] = acc;                     // H^+ by columns
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (!ok) factor_hinv_general(sp, lane);                           // rare; parks H^+ in LDS
}
template <int NP>
__device__ __forceinline__ void solver_refresh_H(const WavePartial* s_part, BlockState& s, int lane, double* hrow) {
    refresh_hinv_to_lds<NP>((const LdsWavePartial*)s_part, (LdsBlockState*)&s, lane);
    const int li = lane < 6 ? lane : 5;
#pragma unroll
    for (int jj = 0; jj < 6; ++jj) hrow[jj] = s.Hinv[jj * 6 + li];    // lane i < 6 keeps row i of H^+ (stored by columns)
}

// what solver_step hands to solver_commit (wave-uniform values)
struct SolverCarry {
    double x[6], chi2_prev, chi2_new;
    int level, it, cnt, n_ref, ctrl, exit_code;
    bool fast;                  // the step was published in matrix form: the quaternion state is still to be updated
};

// Solver wave, one Gauss-Newton iteration (reference :310-343), up to the point where the patch waves can go
// on: new R/t and the control word are in LDS when it returns (the caller publishes them), the rest of the
// iteration follows in solver_commit. Executed uniformly by all 64 lanes of the solver wave (same cost as one
// lane); lane 0 stores. Returns ctrl.
template <int NP, class Args>   // NP = number of partial slots (rows or waves); Args: SAKernelArgs or SolverArgs
__device__ __forceinline__ int solver_step(const Args& a, int pair, int level, int it,
                                           const WavePartial* s_part, BlockState& s, int lane,
                                           double* hrow /* lane i < 6: row i of H^+ */, SolverCarry& c,
                                           unsigned long long* tacc = nullptr /* diagnostic build only */) {
    unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0;
    if (tacc) ts0 = __builtin_amdgcn_s_memtime();
    // Cross-wave totals, lane-parallel: lane q<21 sums H[q], lanes 21..26 sum b, lane 27 chi2
    // (fixed wave order -> deterministic). Totals go through LDS so that only the values the
    // solve needs are ever live in registers.
    static_assert(NP <= 64, "one lane per partial slot");
    // counters: lane w < NP reads slot w; cnt and n_ref (< 2^15 patches per pair, checked by the
    // launcher) share one int and are summed over the wave with one DPP reduction
    int packed = 0;
    bool chg = false;
    if (lane < NP) {
        packed = s_part[lane].cnt | (s_part[lane].n_ref << 16);
        chg = s_part[lane].h_changed != 0;
    }
    packed = wave_sum_i32(packed);
    const int cnt = packed & 0xffff;
    const int n_ref = (packed >> 16) & 0xffff;
    const int changed = __ballot(chg) != 0ull;
    if (changed) solver_refresh_H<NP>(s_part, s, lane, hrow);   // rare: a row's visible set differs from the cached one
    double bs[6], chi2s;
    {
        constexpr int WP = sizeof(WavePartial) / sizeof(double);
        const double* base = (const double*)s_part;
        // doubles inside WavePartial: b[0..5] at 0, chi2 at 6
        if constexpr (NP % 4 == 0) {
            // lane = 4*v + c sums value v over the slots of chunk c (NP/4 each); the four chunk sums of a
            // quad are folded with two DPP quad permutes; readlane broadcasts the totals. Fixed order
            // => deterministic; no LDS round trip on the per-iteration path.
            constexpr int CH = NP / 4;
            const int v = (lane >> 2) < 7 ? (lane >> 2) : 6, c = lane & 3;
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < CH; ++k) acc += base[(c * CH + k) * WP + v];
            acc += dpp_f64<0xB1, 0xf>(acc);    // quad_perm [1,0,3,2]
            acc += dpp_f64<0x4E, 0xf>(acc);    // quad_perm [2,3,0,1]
            const int lo = __double2loint(acc), hi = __double2hiint(acc);
#pragma unroll
            for (int i = 0; i < 6; ++i)
                bs[i] = __hiloint2double(__builtin_amdgcn_readlane(hi, 4 * i), __builtin_amdgcn_readlane(lo, 4 * i));
            chi2s = __hiloint2double(__builtin_amdgcn_readlane(hi, 24), __builtin_amdgcn_readlane(lo, 24));
        } else {
            if (lane < 7) {
                double acc = 0.0;
#pragma unroll 4
                for (int w = 0; w < NP; ++w) acc += base[w * WP + lane];   // fixed slot order: deterministic
                s.bsum[lane] = acc;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
            for (int i = 0; i < 6; ++i) bs[i] = s.bsum[i];
            chi2s = s.bsum[6];
        }
    }
    if (tacc) { asm volatile("" : "+v"(bs[0])); ts1 = __builtin_amdgcn_s_memtime(); }
    const double chi2New = chi2s / (double)(16 * cnt);     // :298 (0/0 -> NaN)
    // H.ldlt().solve(JRes) (:318) as x = H^+ b: lane i < 6 holds row i of H^+ and computes x_i;
    // readlane broadcasts the six results
    __builtin_amdgcn_sched_barrier(0);   // keep the solver's sub-steps in order: short live ranges, no spills
    double x[6];
    {
        double xr = 0.0;
#pragma unroll
        for (int j = 0; j < 6; ++j) xr += hrow[j] * bs[j];
        const int lo = __double2loint(xr), hi = __double2hiint(xr);
#pragma unroll
        for (int i = 0; i < 6; ++i)
            x[i] = __hiloint2double(__builtin_amdgcn_readlane(hi, i), __builtin_amdgcn_readlane(lo, i));
    }
    __builtin_amdgcn_sched_barrier(0);
    if (tacc) { asm volatile("" : "+v"(x[0])); ts2 = __builtin_amdgcn_s_memtime(); }
    const bool stop = (x[0] != x[0]);                      // :321 isnan(x(0))
    const double chi2_prev = s.chi2;
    int ctrl = 0, exit_code = 0;
    c.fast = false; c.level = level; c.it = it; c.cnt = cnt; c.n_ref = n_ref; c.chi2_prev = chi2_prev; c.chi2_new = chi2New;
#pragma unroll
    for (int i = 0; i < 6; ++i) c.x[i] = x[i];
    if ((it > 0 && chi2New > chi2_prev) || stop) {         // :328-332  tT_c2r = tT_c2rOld
        const SE3d To = load_se3(s.qo, s.to);
        double Rn[9];
        quat_to_matrix(To, Rn);
        if (lane == 0) {
            store_se3(s.u.q, s.u.t, To);
#pragma unroll
            for (int i = 0; i < 9; ++i) { s.u.R[i] = Rn[i]; s.Rs[i] = Rn[i]; }
            s.u.tt[0] = To.tx; s.u.tt[1] = To.ty; s.u.tt[2] = To.tz;
        }
        ctrl = 1;
        exit_code = stop ? 3 : 1;
    } else {
        const double theta_sq = x[3] * x[3] + x[4] * x[4] + x[5] * x[5];
        if (theta_sq < 0.01) {
            // The pass only needs R and t of T_c2r * exp(x) (:335, right-multiply): in matrix form that is
            // R_state * dR, t_state + R_state * dt — about half the dependent instructions of the quaternion
            // route. It is published first; solver_commit then brings the Sophus-style quaternion state (what
            // the next step composes with, what a revert restores and what Run finally returns) up to date
            // while the patch waves are already running the next pass.
            double dR[9], dt[3];
            se3_exp_matrix_small(x, theta_sq, dR, dt);
            __builtin_amdgcn_sched_barrier(0);
            double Rn[9], tn[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const double r0 = s.Rs[3 * i], r1 = s.Rs[3 * i + 1], r2 = s.Rs[3 * i + 2];
                Rn[3 * i] = r0 * dR[0] + r1 * dR[3] + r2 * dR[6];
                Rn[3 * i + 1] = r0 * dR[1] + r1 * dR[4] + r2 * dR[7];
                Rn[3 * i + 2] = r0 * dR[2] + r1 * dR[5] + r2 * dR[8];
                tn[i] = s.u.t[i] + (r0 * dt[0] + r1 * dt[1] + r2 * dt[2]);
            }
            if (lane == 0) {
#pragma unroll
                for (int i = 0; i < 9; ++i) s.u.R[i] = Rn[i];
                s.u.tt[0] = tn[0]; s.u.tt[1] = tn[1]; s.u.tt[2] = tn[2];
            }
            c.fast = true;
        } else {                                           // a large rotation step: the closed forms (out of line), state first
            const SE3d dT = se3_exp_call(x[0], x[1], x[2], x[3], x[4], x[5]);
            const SE3d Tcur = load_se3(s.u.q, s.u.t);
            const SE3d Tn = se3_mul(Tcur, dT);             // :335 right-multiply
            double Rn[9];
            quat_to_matrix(Tn, Rn);
            if (lane == 0) {
                store_se3(s.qo, s.to, Tcur);
                store_se3(s.u.q, s.u.t, Tn);
#pragma unroll
                for (int i = 0; i < 9; ++i) { s.u.R[i] = Rn[i]; s.Rs[i] = Rn[i]; }
                s.u.tt[0] = Tn.tx; s.u.tt[1] = Tn.ty; s.u.tt[2] = Tn.tz;
                s.chi2 = chi2New;
            }
        }
        double mx = 0.0;
#pragma unroll
        for (int i = 0; i < 6; ++i) mx = fmax(mx, fabs(x[i]));
        if (mx <= 1e-8) { ctrl = 1; exit_code = 2; }        // :341
    }
    ctrl = __builtin_amdgcn_readfirstlane(ctrl);
    c.ctrl = ctrl; c.exit_code = exit_code;
    if (tacc) { ts3 = __builtin_amdgcn_s_memtime(); tacc[0] += ts1 - ts0; tacc[1] += ts2 - ts1; tacc[2] += ts3 - ts2; }
    if (lane == 0) {
        s.ctrl = ctrl;
        s.n_vis = cnt;
    }
    return ctrl;
}

// Solver wave, after the new R/t/ctrl have been published: the part of the iteration nobody waits for.
// T_c2r = T_c2r * SE3::exp(x) on the quaternion state (:335), its rotation matrix for the next step,
// tT_c2rOld and chi2 (:313-314), and the statistics.
template <class Args>
__device__ __forceinline__ void solver_commit(const Args& a, int pair, BlockState& s, int lane, const SolverCarry& c, bool write_stats = true) {
    if (c.fast) {                                          // the step went out on the series path: |omega|^2 < 0.01
        const SE3d dT = se3_exp_small(c.x);
        __builtin_amdgcn_sched_barrier(0);
        const SE3d Tcur = load_se3(s.u.q, s.u.t);
        const SE3d Tn = se3_mul(Tcur, dT);
        __builtin_amdgcn_sched_barrier(0);
        double Rn[9];
        quat_to_matrix(Tn, Rn);
        if (lane == 0) {
            store_se3(s.qo, s.to, Tcur);
            store_se3(s.u.q, s.u.t, Tn);
#pragma unroll
            for (int i = 0; i < 9; ++i) s.Rs[i] = Rn[i];
            s.chi2 = c.chi2_new;
        }
    }
    if (lane == 0 && write_stats && a.stats && (c.ctrl || c.it == a.max_iters - 1)) {
        dsdtm_align_stats* st = a.stats + pair;
        st->iters[c.level] = c.it + 1;
        st->n_ref[c.level] = c.n_ref;
        st->n_vis[c.level] = c.cnt;
        st->exit_code[c.level] = c.exit_code;
        st->chi2[c.level] = (c.ctrl == 1 && c.exit_code != 2) ? c.chi2_prev : c.chi2_new;
    }
}

// Solver wave, epilogue: tCurFrame->Set_Pose(mT_c2r * tRefFrame->Get_Pose()) (:57), return mnPts (:59)
__device__ __attribute__((noinline)) void solver_finish(double* T_cur_w_pair, int32_t* n_tracked_pair, LdsBlockState* sp, int lane) {
    LdsBlockState& s = *sp;
    SE3d T, Tr;
    T.qw = s.u.q[0]; T.qx = s.u.q[1]; T.qy = s.u.q[2]; T.qz = s.u.q[3]; T.tx = s.u.t[0]; T.ty = s.u.t[1]; T.tz = s.u.t[2];
    Tr.qw = s.u.qr[0]; Tr.qx = s.u.qr[1]; Tr.qy = s.u.qr[2]; Tr.qz = s.u.qr[3]; Tr.tx = s.u.tr[0]; Tr.ty = s.u.tr[1]; Tr.tz = s.u.tr[2];
    const SE3d To = se3_mul(T, Tr);
    double R[9];
    quat_to_matrix(To, R);
    if (lane == 0) {
        double* out = T_cur_w_pair;
        out[0] = R[0]; out[1] = R[1]; out[2] = R[2];  out[3] = To.tx;
        out[4] = R[3]; out[5] = R[4]; out[6] = R[5];  out[7] = To.ty;
        out[8] = R[6]; out[9] = R[7]; out[10] = R[8]; out[11] = To.tz;
        *n_tracked_pair = s.n_vis;
    }
}

// The workspace kernel's solving wave is also a patch wave: with the solver inlined, its temporaries and the two dozen
// 64-bit constants of the series (which the compiler materialises in VGPRs ahead of the kernel's loops) share the
// register file with the pass. In the two-member kernel, with the exchange on top, they were spilled at 256 VGPRs and
// reloaded from scratch one by one inside the solve (188 B of scratch per lane): out of line the solver has its own
// allocation, +7 % (256 x 2000 patches 0.420 -> 0.448 M alignments/s). The one-member kernel keeps the solver inline (out
// of line: -1..2 %, the call and the carry through LDS sit on the iteration's critical path). Everything goes in and out
// through LDS (pointers qualified as such: a generic pointer would turn every access into a flat load).
struct SolverArgs {               // what solver_step / solver_commit read of the kernel arguments
    dsdtm_align_stats* stats;     // null: no statistics from this workgroup
    int max_iters;
};
template <int NP>
__device__ __attribute__((noinline)) void ws_solver_step(const LdsWavePartial* part, LdsBlockState* sp,
                                                         __attribute__((address_space(3))) SolverCarry* cp,
                                                         int pair, int level, int it, int lane) {
    BlockState& s = *(BlockState*)sp;
    // row of H^+ from LDS at every step (refresh_hinv_to_lds parks it there): nothing of the solver is live across the pass
    double hrow[6];
    const int li = lane < 6 ? lane : 5;
#pragma unroll
    for (int jj = 0; jj < 6; ++jj) hrow[jj] = s.Hinv[jj * 6 + li];
    const SolverArgs al{nullptr, 0};
    SolverCarry c;
    (void)solver_step<NP>(al, pair, level, it, (const WavePartial*)part, s, lane, hrow, c);
    if (lane == 0) *(SolverCarry*)cp = c;
}
__device__ __attribute__((noinline)) void ws_solver_commit(LdsBlockState* sp, const __attribute__((address_space(3))) SolverCarry* cp,
                                                           dsdtm_align_stats* stats, int max_iters, int pair, int lane) {
    BlockState& s = *(BlockState*)sp;
    const SolverCarry c = *(const SolverCarry*)cp;
    const SolverArgs al{stats, max_iters};
    solver_commit(al, pair, s, lane, c, true);
}

// Hand-over protocol of the register kernel (pair-local counters, see pair_signal_arrive & co.):
//   per pair:      ACK      all patch waves are done with the previous pair's state -> ack += NPW (the solver
//                                                                     may rewrite pair/run/ctrl/R for the next pair)
//                  B0       pair index published, pose block live  -> seq += 1 (patch waves may read pair/Cref/R/t)
//   per level:     BH       all-visible H partials are in LDS      -> arrive_h += NPW (solver sums + factorises
//                                                                     them while the first pass runs)
//   per iteration: B1       partials of all patch waves are in LDS -> arrive += NPW (solver may read them)
//                  B2       solver has published R/t/ctrl           -> seq += 1 (patch waves may read them)
// The solver rewrites R/t/ctrl only after the next B1, which every patch wave signals only after it
// has read them, so nothing else is needed at level boundaries. The generic (workspace) kernel
// below uses plain s_barrier for B0/B1/B2.

// ---------------------------------------------------------------------------------------------
// Register-resident kernel: NPW patch waves (one patch per lane, NPW*64 >= n_features) + 1 solver
// wave.
// ---------------------------------------------------------------------------------------------
// STAMPS = diagnostic instantiation only (dsdtm_debug_sparse_align_stamps): the solver wave
// accumulates shader-clock cycles spent waiting for the patch waves and solving, and writes them to
// a.workspace[pair*8 ..]; the timed/product instantiation contains no stamp.
//
// PPW = pairs per workgroup. A 6-wave group at ~166 VGPRs leaves the CU at one resident group
// (the dispatcher does not co-schedule two such groups), but ONE 12-wave workgroup at 3 waves/SIMD
// fits: PPW = 2 runs two independent pair pipelines in one workgroup, synchronised by the
// pair-local counters above instead of s_barrier, so one pair's solve overlaps the other's pass and
// the 10 patch waves balance over the 4 SIMDs.
// LDS of the register kernel. One struct
// so that the layout is ours: the small, hot structures sit at the lowest addresses (ds_read/ds_write
// immediate offsets reach 64 KB; behind a large array every access would need extra address
// arithmetic — measured: +75 % on the solver's partial sums), the per-patch arrays behind them.
template <int NPW, int PPW>
struct RegSmem {
    BlockState st[PPW];
    WavePartial part[PPW][NPW * 4];
    uint32_t sink[PPW][64];                              // LDS-DMA target of prepare_next's cache warm-up (never read)
    uint32_t win[PPW][WIN_ROWS * 3 * NPW * 64];          // current-image footprint windows (residual_patch), [plane][lane]
};

template <int NPW, int PPW, bool STAMPS = false>
__global__ __launch_bounds__(PPW * (NPW + 1) * 64) void sparse_align_reg_kernel(const SAKernelArgs a) {
    constexpr int NP = NPW * 4;            // one partial slot per 16-lane DPP row
    constexpr int WPP = NPW + 1;           // waves per pair
    __shared__ RegSmem<NPW, PPW> sm;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int gwave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int slot = gwave / WPP;                              // which pair slot of this workgroup
    const int wave = gwave - slot * WPP;                       // wave inside the slot
    const int ltid = tid - slot * WPP * 64;                    // thread inside the slot
    {
        // LDS is uninitialised at kernel start: zero the slot-local counters and run the one real
        // workgroup barrier of this kernel while every wave is still present
        if (tid < PPW) { sm.st[tid].arrive = 0u; sm.st[tid].arrive_h = 0u; sm.st[tid].seq = 0u; sm.st[tid].ack = 0u; sm.st[tid].nx_pair = -1; }
        __syncthreads();
    }
    WavePartial* s_part = sm.part[slot];
    BlockState& s = sm.st[slot];

    // PERSISTENT SLOTS: the grid is one workgroup per CU; each slot (NPW patch waves + 1 solver wave)
    // pulls pair indices from a global counter until the batch is exhausted, independently of the
    // workgroup's other slot. A workgroup that held its CU until BOTH of its pairs finished lost
    // ~11 % to the spread of pair durations (10..25 Gauss-Newton iterations). The slot-local
    // counters are monotonic, so they simply keep counting from one pair to the next.
    // (Finer work units — a pair cut at a pyramid level, the 7-double pose handed from CU to CU
    // through a completion-ordered queue — were built and measured: the tail of a 1024-pair launch
    // is bounded by the finest level, which cannot be cut, and the idle time at the end only fell
    // from ~12 % to ~9.5 %, which the hand-over cost ate. Whole pairs stay the unit.)
    if (wave == NPW) {
        // ------------------------------ solver wave ------------------------------
        unsigned expected = 0, expected_h = 0;                         // B1 / BH arrivals consumed so far
        unsigned published = 0;                                        // states published so far
        // The solve is the serial section of this pair's iteration, and this wave shares its SIMD
        // with patch waves of the workgroup's other slot: give it issue priority.
        if (PPW > 1) __builtin_amdgcn_s_setprio(2);
        // the first pair of every slot is assigned statically (no atomic round trip on the start-up
        // path); the counter hands out the pairs behind those. claim_issue only starts the atomic, its
        // value is waited for where claim_value reads it.
        const int first_dynamic = (int)gridDim.x * PPW;
        auto claim_issue = [&]() -> int {
            int v = 0;
            if (lane == 0) v = first_dynamic + (int)atomicAdd(a.pair_counter, 1u);
            return v;
        };
        auto claim_value = [&](int raw) -> int { return __builtin_amdgcn_readfirstlane(raw); };
        // Every pair that is started issues exactly one claim, so the claim that returns n_pairs - 1 is the last
        // one of this launch: its owner puts the counter back to zero for the launch that uses it next (the host
        // needs no memset node per launch).
        auto claim_taken = [&](int v) -> int {
            if (lane == 0 && v - first_dynamic == a.n_pairs - 1)
                __hip_atomic_store(a.pair_counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return v;
        };
        int pair = (int)blockIdx.x * PPW + slot;
        unsigned acks = 0;                                             // acknowledgements expected so far
        while (true) {
            // every patch wave has finished reading the previous pair's state (pair/run, final ctrl)
            pair_wait_arrive(&s.ack, acks, a.timeout_flag);
            const bool have = pair < a.n_pairs;
            const int nf = have ? (a.n_features ? a.n_features[pair] : a.max_features) : 0;
            // Run(): "Too few features to track" (:34-38) -> return 0, pose untouched
            const bool run = have && nf >= a.min_fts && a.max_level - 1 >= a.min_level;
            const bool staged = have && __builtin_amdgcn_readfirstlane(s.nx_pair) == pair;
            if (lane == 0) { s.pair = pair; s.run = run ? 1 : 0; s.nx_pair = -1; }
            if (!have) { pair_publish(s, ++published, lane); break; }  // B0 with "no more work"
            acks += NPW;
            if (!run) {
                if (lane == 0) { a.n_tracked[pair] = 0; stats_clear(a, pair); }
                pair_publish(s, ++published, lane);                    // B0 with "skip"
                pair = claim_taken(claim_value(claim_issue()));
                continue;
            }
            const int next_raw = claim_issue();                        // latency hidden under this pair
            unsigned long long t_wait = 0, t_solve = 0, t_first = 0, n_it = 0, t_begin = 0, t_refresh = 0;
            unsigned long long t_sub[3] = {0, 0, 0};
            unsigned long long rt_begin = 0;
            if (STAMPS) { t_begin = __builtin_amdgcn_s_memtime(); rt_begin = __builtin_amdgcn_s_memrealtime(); }
            if (staged) commit_next(*(LdsBlockState*)&s, lane);        // prepared while the previous pair was running
            else solver_init(a.T_ref_w + 12 * (size_t)pair, a.T_cur_w + 12 * (size_t)pair, (LdsBlockState*)&s, lane);
            if (lane == 0) stats_clear(a, pair);
            pair_publish(s, ++published, lane);                        // B0
            bool prepared = false;
            for (int level = a.max_level - 1; level >= a.min_level; --level) {
                if (lane == 0) {                                       // GaussNewtonSolver entry (:304-308)
                    s.chi2 = 0.0;
#pragma unroll
                    for (int i = 0; i < 4; ++i) s.qo[i] = s.u.q[i];
#pragma unroll
                    for (int i = 0; i < 3; ++i) s.to[i] = s.u.t[i];
                }
                // speculative: the patch waves publish the all-visible H partials right after the
                // level's precompute; sum + factorise them here while they run the first pass
                expected_h += NPW;
                pair_wait_arrive(&s.arrive_h, expected_h, a.timeout_flag);             // BH
                double hrow[6];
                unsigned long long tr0 = 0;
                if (STAMPS) tr0 = __builtin_amdgcn_s_memtime();
                solver_refresh_H<NP>(s_part, s, lane, hrow);
                if (STAMPS) { asm volatile("" : "+v"(hrow[0])); t_refresh += __builtin_amdgcn_s_memtime() - tr0; }
                for (int it = 0; it < a.max_iters; ++it) {
                    unsigned long long t0 = 0, t1 = 0, t2 = 0;
                    if (STAMPS) t0 = __builtin_amdgcn_s_memtime();
                    expected += NPW;
                    pair_wait_arrive(&s.arrive, expected, a.timeout_flag);             // B1
                    if (STAMPS) t1 = __builtin_amdgcn_s_memtime();
                    SolverCarry carry;
                    const int ctrl = solver_step<NP>(a, pair, level, it, s_part, s, lane, hrow, carry, STAMPS ? t_sub : nullptr);
                    if (STAMPS) {
                        t2 = __builtin_amdgcn_s_memtime();
                        if (it == 0) t_first += t1 - t0; else t_wait += t1 - t0;
                        t_solve += t2 - t1;
                        n_it += 1;
                    }
                    pair_publish(s, ++published, lane);                // B2
                    solver_commit(a, pair, s, lane, carry);            // quaternion state, chi2, statistics: behind the hand-over
                    if (ctrl) break;
                    // the rest of this iteration's window (the patch waves are running their next pass)
                    if (!prepared && level < a.max_level - 1) {
                        // once per pair, not on its first level (the claim has long returned by now): the
                        // next pair's prologue
                        prepared = true;
                        const int np = claim_value(next_raw);
                        if (np < a.n_pairs) {
                            const size_t nfm = (size_t)a.max_features;
                            prepare_next(a.T_ref_w + 12 * (size_t)np, a.T_cur_w + 12 * (size_t)np, a.px_xy + 2 * np * nfm,
                                         a.bearing + 3 * np * nfm, a.p_world + 3 * np * nfm, a.initial + np * nfm,
                                         a.max_features, np, (LdsBlockState*)&s, (LdsVoid*)sm.sink[slot], lane);
                        }
                    }
                }
            }
            solver_finish(a.T_cur_w + 12 * (size_t)pair, a.n_tracked + pair, (LdsBlockState*)&s, lane);
            if (STAMPS && lane == 0 && a.workspace) {
                unsigned long long* o = (unsigned long long*)a.workspace + (size_t)pair * 8;
                o[0] = t_first; o[1] = t_wait; o[2] = t_solve; o[3] = n_it;
                o[4] = __builtin_amdgcn_s_memtime() - t_begin;
                o[5] = t_sub[0];
                o[6] = t_sub[1];
                o[7] = t_sub[2];
                ((unsigned long long*)a.workspace)[(size_t)a.n_pairs * 48 + pair] = t_refresh;
                // the device-wide 100 MHz clock at the pair's begin and end + the slot that ran it: which slots sit idle, and
                // for how long, once the pair counter has run dry (tools/stamps.py: the tail of a solo launch)
                unsigned long long* rt = (unsigned long long*)a.workspace + (size_t)a.n_pairs * 49 + (size_t)pair * 3;
                rt[0] = rt_begin; rt[1] = __builtin_amdgcn_s_memrealtime(); rt[2] = (unsigned long long)blockIdx.x * PPW + slot;
            }
            pair = claim_taken(claim_value(next_raw));
        }
        return;
    }

    // ---------------------------------- patch waves ----------------------------------
    unsigned seen = 0;                                                 // states consumed so far
    unsigned pairs_done = 0;                                           // pairs this slot has finished (issue priority alternates, below)
    const int row = lane >> 4;
    const bool row_writer = (lane & 15) == 15;
    WavePartial& my_part = s_part[wave * 4 + row];
    while (true) {
        pair_wait_seq(s, ++seen, a.timeout_flag);                                      // B0 of the slot's next pair
        const int pair = __builtin_amdgcn_readfirstlane(s.pair);
        if (pair >= a.n_pairs) break;                                  // batch exhausted
        if (!s.run) {                                                  // Min_fts rule handled by the solver
            pair_signal_arrive(&s.ack, lane);
            continue;
        }
        const int nf = a.n_features ? a.n_features[pair] : a.max_features;
        const uint8_t* __restrict__ ref_base = a.ref_pyr + (size_t)pair * a.pyr_pitch;
        const uint8_t* __restrict__ cur_base = a.cur_pyr + (size_t)pair * a.pyr_pitch;
        const FeatureRaw fraw = load_feature_raw(a, (size_t)pair * a.max_features + ltid, ltid < nf);
        unsigned long long st_pre = 0, st_pass = 0, st_h = 0, st_bar = 0;
        unsigned long long st_lvl[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // pass cycles / pass count of levels 0..3
        unsigned long long st_plv[4] = {0, 0, 0, 0}, st_first[4] = {0, 0, 0, 0};   // precompute / first-pass cycles of levels 0..3
        unsigned long long st_bfirst[4] = {0, 0, 0, 0};                            // wait for the solver after the first pass
        FeatureRegs F;
        {
            const double Cref[3] = {s.u.Cref[0], s.u.Cref[1], s.u.Cref[2]};
            F = make_feature(fraw, Cref);
        }
        for (int level = a.max_level - 1; level >= a.min_level; --level) {
            // ISSUE PRIORITY BETWEEN THE TWO SLOTS. The second slot's waves are the younger half of the workgroup and lose the issue
            // arbitration to the first slot's: measured (tools/stamps.py) its pairs took 97.0 us against 85.3 us, so the first
            // slot sat idle for the last 23 us of a launch on the average CU. A static priority would only swap the roles
            // (measured); alternating it PER PAIR — second slot favoured on its first pair, first slot on its second, ... —
            // evens them out: 91.2 / 92.0 us per pair, both slots end within 2.3 us of each other, one launch 204.7 -> 196 us.
            // (Per level instead of per pair: half the gain; with three or four slots per workgroup — 190 / 120 patches — the same
            // rule measured flat or slightly negative, so only the two-slot shapes use it; profiles/r05_tail.txt §5.)
            if (PPW == 2) {
                if ((pairs_done + slot) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
            }
            const LevelGeom lg = a.lv[level];
            const double scale = (double)(1.0f / (float)(1 << level));
            const double fs = (double)a.f * scale;
            PatchRegs P;
            uint32_t worg = WIN_EMPTY;                                 // the level's window is filled by its first pass
            unsigned long long tp0 = 0;
            if (STAMPS) tp0 = __builtin_amdgcn_s_memtime();
            LdsU32* const win = (LdsU32*)&sm.win[slot][ltid];
            precompute_patch(a, lg, level, ref_base, F, P);
            if (STAMPS) {
                pin_patch(P);   // make the stamp wait for the precompute results
                const unsigned long long dtp = __builtin_amdgcn_s_memtime() - tp0;
                st_pre += dtp;
#pragma unroll
                for (int l = 0; l < 4; ++l) if (l == level) st_plv[l] += dtp;
            }
            const unsigned long long valid_mask = __ballot(P.valid);
            const int n_ref_row = __popc((unsigned)(valid_mask >> (16 * row)) & 0xffffu);
            // Speculative H of the level: every valid patch visible (what the first pass finds almost
            // always). Published before the pass so that the solver can factorise it meanwhile.
            unsigned long long cached_mask = valid_mask;
            {
                unsigned long long th0 = 0;
                if (STAMPS) th0 = __builtin_amdgcn_s_memtime();
                const PatchHess ph = patch_hess_factors(P, fs, P.valid);
                patch_hess_rows(ph, P.valid, lane, my_part.H);
                pair_signal_arrive(&s.arrive_h, lane);                 // BH
                if (STAMPS) st_h += __builtin_amdgcn_s_memtime() - th0;
            }

            for (int it = 0; it < a.max_iters; ++it) {
                double chi2, b[6];
                unsigned long long tq0 = 0, tq1 = 0;
                if (STAMPS) tq0 = __builtin_amdgcn_s_memtime();
                pin_patch(P);
                // (The slot's waves 0 and 4 sit on the same SIMD and run their passes in phase; the arbiter
                // favours the older wave 0: 4.0 k vs 5.2 k cycles per pass, and the solver waits for the
                // slower one. Giving wave 4 issue priority for part of its pass moves cycles between the
                // two but not the slower finish: measured, no gain.)
                const bool vis = residual_patch<NPW * 64>(a, lg, scale, fs, cur_base, P, s.u.R, s.u.tt, chi2, b,
                                                                              win, &worg);
                const unsigned long long vmask = __ballot(vis);
                // reduce to the 16-lane DPP rows only; the solver's lane-parallel summation folds the 4*NPW row
                // partials
                {
                    // b[0..5] and chi2 in ONE packed butterfly (row_reduce8): the lane holds the row total of value
                    // row_reduce8_index(lane); doubles 0..6 of a WavePartial are b[0..5], chi2
                    const double v8[8] = {b[0], b[1], b[2], b[3], b[4], b[5], chi2, 0.0};
                    const double t = row_reduce8(v8, lane);
                    const int q = row_reduce8_index(lane);
                    if (!(lane & 4) && q < 7) ((double*)&my_part)[q] = t;
                }
                if (row_writer) {
                    my_part.cnt = __popc((unsigned)(vmask >> (16 * row)) & 0xffffu);
                    my_part.n_ref = n_ref_row;
                }
                if (STAMPS) {
                    tq1 = __builtin_amdgcn_s_memtime(); st_pass += tq1 - tq0;
#pragma unroll
                    for (int l = 0; l < 4; ++l) if (l == level) { st_lvl[l] += tq1 - tq0; st_lvl[4 + l] += 1; if (it == 0) st_first[l] += tq1 - tq0; }
                }
                const bool h_changed = (vmask != cached_mask);        // wave-uniform, rare
                if (h_changed) {
                    const PatchHess ph = patch_hess_factors(P, fs, vis);
                    patch_hess_rows(ph, vis, lane, my_part.H);
                    cached_mask = vmask;
                }
                if (row_writer) my_part.h_changed = h_changed ? 1 : 0;
                unsigned long long tq2 = 0;
                if (STAMPS) { tq2 = __builtin_amdgcn_s_memtime(); st_h += tq2 - tq1; }
                ++seen;
                pair_signal_arrive(&s.arrive, lane);                   // B1
                pair_wait_seq(s, seen, a.timeout_flag);                                // B2
                if (STAMPS) {
                    const unsigned long long dtb = __builtin_amdgcn_s_memtime() - tq2;
                    st_bar += dtb;
                    if (it == 0) {
#pragma unroll
                        for (int l = 0; l < 4; ++l) if (l == level) st_bfirst[l] += dtb;
                    }
                }
                if (s.ctrl) break;
            }
        }
        ++pairs_done;
        pair_signal_arrive(&s.ack, lane);                              // done with this pair's shared state
        if (STAMPS && lane == 0 && a.workspace) {                      // every patch wave: pass and barrier cycles
            unsigned long long* o = (unsigned long long*)a.workspace + (size_t)a.n_pairs * 20 + (size_t)pair * 16;
            o[wave] = st_pass; o[8 + wave] = st_bar;
        }
        if (STAMPS && ltid == 0 && a.workspace) {                      // wave 0: per-level precompute and first-pass cycles
            unsigned long long* o = (unsigned long long*)a.workspace + (size_t)a.n_pairs * 36 + (size_t)pair * 8;
#pragma unroll
            for (int l = 0; l < 4; ++l) { o[l] = st_plv[l]; o[4 + l] = st_first[l]; }
            unsigned long long* o2 = (unsigned long long*)a.workspace + (size_t)a.n_pairs * 44 + (size_t)pair * 4;
#pragma unroll
            for (int l = 0; l < 4; ++l) o2[l] = st_bfirst[l];
        }
        if (STAMPS && ltid == 0 && a.workspace) {
            unsigned long long* o = (unsigned long long*)a.workspace + (size_t)a.n_pairs * 8 + (size_t)pair * 12;
            o[0] = st_pre; o[1] = st_pass; o[2] = st_h; o[3] = st_bar;
#pragma unroll
            for (int l = 0; l < 8; ++l) o[4 + l] = st_lvl[l];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Generic kernel for any feature count: each lane of the NPW patch waves loops over patches
// p = tid, tid + NPW*64, ...; the per-patch state is parked — in LDS up to 1024 patches per workgroup (one
// pair per compute unit, or one pair on two compute units up to 2048: MEMBERS below), else in an HBM
// workspace laid out [plane][patch] so that the lanes of a wave read consecutive dwords.
// ---------------------------------------------------------------------------------------------
// What is parked per patch is the INPUT of the grid, not the grid: the 7x7 u8 reference footprint as unpacked
// row words (14 dwords), the feature's pixel (2 floats) and its 3-D point (3 doubles) — 88 bytes instead of the
// 280 bytes of the 32 interpolated doubles + X. Every pass rebuilds the grid in registers with the same
// grid_from_rows the register kernels run once per level (same inputs, same arithmetic: bit-identical values).
// The first version parked the grid itself and was bound by exactly that traffic: 1024 pairs of 1000 patches moved
// 8.7 GB per launch through HBM (rocprofv3 FETCH_SIZE/WRITE_SIZE, 9.7x the algorithmic bytes) at 6.2 TB/s.
// The workgroup has no solver wave (a variant with 7 + 1 waves was measured: 1.5 % slower): 8 patch waves, the last of
// which also solves.
constexpr int WS_NPW = 8;
constexpr int WS_THREADS = WS_NPW * 64;
constexpr int WS_DWORDS = 22;   // rlo[7], rhi[7], px, py, X[3] as dword pairs
__host__ __device__ inline size_t ws_doubles_per_pair(int max_features) {
    const size_t npad = ((size_t)max_features + 63) / 64 * 64;
    return npad * WS_DWORDS / 2;
}

struct WsPatch {
    uint32_t rlo[7], rhi[7];
    FeatureRegs F;               // px, py, X; ok = the patch passed the reference-side checks of this level
};

// WP = uint32_t* (HBM workspace, planes `npad` dwords apart) or an LDS-qualified pointer (planes WCAP dwords apart)
template <typename WP>
__device__ __forceinline__ void ws_store(WP ws, size_t npad, int p, const WsPatch& w) {
#pragma unroll
    for (int r = 0; r < 7; ++r) { ws[(size_t)r * npad + p] = w.rlo[r]; ws[(size_t)(7 + r) * npad + p] = w.rhi[r]; }
    ws[(size_t)14 * npad + p] = __float_as_uint(w.F.px);
    ws[(size_t)15 * npad + p] = __float_as_uint(w.F.py);
    // invalid patches are parked with a NaN depth
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double v = w.F.ok ? w.F.X[i] : __longlong_as_double(0x7ff8000000000000ll);
        ws[(size_t)(16 + 2 * i) * npad + p] = (uint32_t)__double2loint(v);
        ws[(size_t)(17 + 2 * i) * npad + p] = (uint32_t)__double2hiint(v);
    }
}

// The same in two parts (kernels with LDS windows): the seven footprint rows are parked at every level start, the feature —
// pixel and normalised point, which do not depend on the level — once per pair; an unusable FEATURE (!mbInitial, zero map
// point: :86, :95) is parked with the sign bit of its pixel's x set (a negative x fails every level's border test anyway) and,
// as in ws_store, a NaN depth; whether a usable feature passes a level's border test is a bit in its thread.
template <typename WP>
__device__ __forceinline__ void ws_store_rows(WP ws, size_t npad, int p, const uint32_t* rlo, const uint32_t* rhi) {
#pragma unroll
    for (int r = 0; r < 7; ++r) { ws[(size_t)r * npad + p] = rlo[r]; ws[(size_t)(7 + r) * npad + p] = rhi[r]; }
}
template <typename WP>
__device__ __forceinline__ void ws_store_feature(WP ws, size_t npad, int p, const FeatureRegs& F) {
    ws[(size_t)14 * npad + p] = __float_as_uint(F.px) | (F.ok ? 0u : 0x80000000u);
    ws[(size_t)15 * npad + p] = __float_as_uint(F.py);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double v = F.ok ? F.X[i] : __longlong_as_double(0x7ff8000000000000ll);
        ws[(size_t)(16 + 2 * i) * npad + p] = (uint32_t)__double2loint(v);
        ws[(size_t)(17 + 2 * i) * npad + p] = (uint32_t)__double2hiint(v);
    }
}

// The HBM workspace is streamed: every parked dword is read exactly once per pass, by the thread that wrote it, and a
// pass reads far more than the L2 keeps until the next one — non-temporal loads leave the cache to the footprint gathers.
__device__ __forceinline__ uint32_t ws_word(const uint32_t* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ uint32_t ws_word(const LdsU32* p) { return *p; }
template <typename WP>
__device__ __forceinline__ void ws_load(WP ws, size_t npad, int p, WsPatch& w) {
    uint32_t d[WS_DWORDS];
#pragma unroll
    for (int i = 0; i < WS_DWORDS; ++i) d[i] = ws_word(ws + (size_t)i * npad + p);
#pragma unroll
    for (int r = 0; r < 7; ++r) { w.rlo[r] = d[r]; w.rhi[r] = d[7 + r]; }
    w.F.px = __uint_as_float(d[14]);
    w.F.py = __uint_as_float(d[15]);
#pragma unroll
    for (int i = 0; i < 3; ++i) w.F.X[i] = __hiloint2double((int)d[17 + 2 * i], (int)d[16 + 2 * i]);
    w.F.ok = (w.F.X[2] == w.F.X[2]);
}

template <typename WP>
__device__ __forceinline__ void ws_load_feature(WP ws, size_t npad, int p, FeatureRegs& F) {
    uint32_t d[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) d[i] = ws_word(ws + (size_t)(14 + i) * npad + p);
    F.px = __uint_as_float(d[0]);
    F.py = __uint_as_float(d[1]);
#pragma unroll
    for (int i = 0; i < 3; ++i) F.X[i] = __hiloint2double((int)d[3 + 2 * i], (int)d[2 + 2 * i]);
    F.ok = (d[0] >> 31) == 0u;
}

// the register-resident patch state of one parked patch (what precompute_patch leaves behind in the register kernels)
__device__ __forceinline__ void ws_patch_regs(const WsPatch& w, const LevelGeom& lg, int level, PatchRegs& P) {
    const RefGeom g = ref_geom(w.F, lg, level);          // valid == w.F.ok: the border test passed when the patch was parked
    grid_from_rows(w.F, g, w.rlo, w.rhi, P);
}

// WCAP > 0: current-image footprint windows in LDS as in the register kernels (residual_patch, WIN_NL = WCAP): one
// column of 15 dwords + the window's origin per PATCH (not per lane; a thread always handles the same patches), in
// dynamic LDS — 64 KB for up to 1024 patches, 128 KB for up to 2048. Without them every pass gathers its five
// footprint rows per patch through the CU's texture-address path again (the round-2 kernel: 1024 pairs of 1000
// patches 1.03 ms per launch).
// No solver wave — the workgroup is NPW patch waves and the last of them runs the solve between
// the two barriers of an iteration. This kernel's patch state does not live in registers across passes (the grid is
// rebuilt from the parked inputs every pass), so one wave can be both; at ~205 VGPRs a CU holds 8 waves, and 8 patch
// waves instead of 7 + 1 turn the 1000 patches of BASELINE config 3 from three rounds of 448 lanes (the third 23 %
// full) into two rounds of 512, and 2000 patches from five rounds into four.
// PARK_LDS: the parked grid inputs (88 B per patch) live in LDS behind the windows instead of the HBM workspace —
// up to 1024 patches (4 + 60 + 88 KB of dynamic LDS: one workgroup per compute unit), nothing of a pass touches HBM
// but the pose; without it a pass streams them from the workspace (written once per level, re-read every pass).
// MEMBERS = 2 (with PARK_LDS, 1025..2048 patches): ONE pair on TWO workgroups = two compute units, each with its half
// of the patches wholly in its LDS. Per Gauss-Newton iteration the two solving waves exchange their workgroup's partial
// (30 doubles) through tagged 64-bit words in HBM — the team kernel's exchange: the tag is the iteration number, a
// reader that finds it holds that iteration's payload, two banks by parity — and both run the same solver_step on the
// same two partials in the same order: the same pose on both, bit for bit, without a broadcast; member 0 writes the
// results. Unlike the team kernel's launches these are BATCHES (more workgroups than compute units), so the members
// of a pair must not wait for each other across the dispatch order: members are blocks b and b + 8 — the same XCD,
// and consecutive in that XCD's in-order dispatch queue — so at most one pair per XCD is ever split at the leading
// edge of what is resident, and the complete pairs behind it always finish and free its partner's compute unit.
// Every wait is bounded (spin limit -> timeout flag -> the pair stops and drains its barriers).
constexpr int DUO_WPD = sizeof(WavePartial) / sizeof(double);
constexpr size_t DUO_BYTES = 2 * 2 * DUO_WPD * 2 * sizeof(unsigned long long);      // banks x members x words
template <int NPW, int WCAP, bool PARK_LDS = false, int MEMBERS = 1>
__global__ __launch_bounds__(NPW * 64) __attribute__((amdgpu_waves_per_eu(1, NPW / 4)))
void sparse_align_ws_kernel(const SAKernelArgs a) {
    static_assert(MEMBERS == 1 || (MEMBERS == 2 && PARK_LDS), "two members: everything of a half in LDS");
    constexpr int PT = NPW * 64;   // patch threads
    constexpr int SW = NPW - 1;    // the wave that also solves
    __shared__ WavePartial s_part[NPW];
    __shared__ WavePartial s_mpart[MEMBERS == 2 ? 2 : 1];              // two members: one partial per member, member order
    __shared__ BlockState s;
    __shared__ SolverCarry s_carry;                                    // two members: solver_step -> solver_commit, both out of line
    extern __shared__ __attribute__((aligned(16))) uint32_t ws_win[];   // [16][WCAP]: 15 window planes + origins

    int pair = blockIdx.x, member = 0;
    if constexpr (MEMBERS == 2) {
        const int r = (int)blockIdx.x & 15;
        member = r >> 3;
        pair = ((int)blockIdx.x >> 4) * 8 + (r & 7);
        if (pair >= a.n_pairs || (member == 1 && a.debug_drop)) return;
    }
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nf = a.n_features ? a.n_features[pair] : a.max_features;

    if (nf < a.min_fts || a.max_level - 1 < a.min_level) {             // the same decision in both members
        if (member == 0 && tid == 0) { a.n_tracked[pair] = 0; stats_clear(a, pair); }
        return;
    }

    // the younger half of the workgroup's waves loses the issue arbitration to the older half, and every barrier waits for it:
    // a static priority for it (two-member pairs: 0.458 -> 0.470 M/s on 256 x 2000 patches; one member: flat, not used)
    if constexpr (MEMBERS == 2) { if (wave >= NPW / 2) __builtin_amdgcn_s_setprio(1); }
    const bool solves = wave == SW;
    if (solves) {
        solver_init(a.T_ref_w + 12 * (size_t)pair, a.T_cur_w + 12 * (size_t)pair, (LdsBlockState*)&s, lane);
        if (member == 0 && lane == 0) stats_clear(a, pair);
    }

    const uint8_t* __restrict__ ref_base = a.ref_pyr + (size_t)pair * a.pyr_pitch;
    const uint8_t* __restrict__ cur_base = a.cur_pyr + (size_t)pair * a.pyr_pitch;
    const size_t npad = ((size_t)a.max_features + 63) / 64 * 64;
    // this workgroup's patches [p0, p1); their LDS entries are indexed by p - p0
    const int half = MEMBERS == 2 ? (int)((npad / 64 + 1) / 2) * 64 : (int)npad;
    const int p0 = member * half, p1 = (int)npad < p0 + half ? (int)npad : p0 + half;
    const size_t pstride = PARK_LDS ? (size_t)WCAP : npad;                        // dwords between two parked planes
    auto ws = [&]() {
        if constexpr (PARK_LDS) return (LdsU32*)(ws_win + 16 * WCAP);
        else return (uint32_t*)(a.workspace + (size_t)pair * ws_doubles_per_pair(a.max_features));
    }();
    unsigned long long* const twords = MEMBERS == 2 ? (unsigned long long*)((char*)a.workspace + (size_t)pair * DUO_BYTES) : nullptr;
    unsigned g_it = 0;                                                 // iterations so far, over all levels (tag of the exchange)
    bool dead = false;                                                 // a wait for the partner ran out: drain, keep the barriers matched
    // FEATURE ORDER. The reference walks mvFeatures in list order (src/Sprase_ImageAlign.cpp:84-103) and sums in that order;
    // here a thread's patches are the entries tid, tid + PT, .. of the pair's list ORDERED BY IMAGE ROW (then column, then
    // list index: a total order, so the result depends neither on timing nor on the order the caller handed the list in):
    // the lanes of a wave then gather their footprint rows from neighbouring image rows and share cache lines (measured on
    // the caller's side in round 4: +7 % on 1000- and 2000-patch pairs, poses equal to 2e-15). One counting sort per pair in
    // the LDS the level starts fill later: rows are the bins (counts and bin starts are order-free), a feature's place inside
    // its bin is its rank among the bin's keys. n_tracked and the statistics are order-free; both members of a two-member
    // pair order the same list and take their halves of it.
    // Everything in LDS (PARK_LDS): the sort leaves place[i] (scratch), the features are then loaded in LIST order — coalesced
    // columns — and each is parked at its place, once per pair. Parked in HBM: the thread of a place gathers its feature (s_perm).
    __shared__ uint16_t s_perm[(WCAP > 0 && !PARK_LDS) ? WCAP : 1];
    uint32_t* const s_place = ws_win + 2 * 1024 + 2048 + 64;           // [npad], scratch, valid until the first level start
    const bool ordered = WCAP > 0 && a.ws_sort;
    if constexpr (WCAP > 0) {
        if (ordered) {
            constexpr int BINS = 1024, KPT = (2048 + PT - 1) / PT;     // row bins; keys per thread (WCAP > 0: at most 2048 patches)
            uint32_t* const start = ws_win;                            // [BINS] first place of a bin (then: its end)
            uint32_t* const cur = ws_win + BINS;                       // [BINS] next free place of a bin
            uint32_t* const tmp = ws_win + 2 * BINS;                   // [npad] keys in bin order, arbitrary inside a bin
            uint32_t* const wtot = ws_win + 2 * BINS + 2048;           // [NPW] bin counts per wave of the scan
            const float ysc = a.lv[0].h <= BINS ? 1.0f : (float)BINS / (float)a.lv[0].h;
            const float xsc = a.lv[0].w <= 2047 ? 1.0f : 2046.0f / (float)a.lv[0].w;   // (column keys 0..2046: see below)
            for (int i = tid; i < BINS; i += PT) start[i] = 0u;
            uint32_t key[KPT];
#pragma unroll
            for (int k = 0; k < KPT; ++k) {
                const int i = tid + k * PT;
                key[k] = 0xffffffffu;
                if (i < nf) {
                    const size_t fi = (size_t)pair * a.max_features + i;
                    const float x = a.px_xy[2 * fi], y = a.px_xy[2 * fi + 1];
                    const int yk = (int)fminf(fmaxf(y == y ? y * ysc : 0.0f, 0.0f), (float)(BINS - 1));
                    // column key at most 2046: feature 2047 of a 2048-feature pair whose pixel clamps to the last row bin must not
                    // produce the all-ones key, which means "no feature" below (round-5 advice)
                    const int xk = (int)fminf(fmaxf(x == x ? x * xsc : 0.0f, 0.0f), 2046.0f);
                    key[k] = ((uint32_t)yk << 22) | ((uint32_t)xk << 11) | (uint32_t)i;
                }
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < KPT; ++k)
                if (key[k] != 0xffffffffu) atomicAdd(&start[key[k] >> 22], 1u);
            __syncthreads();
            {   // exclusive scan of the BINS counts: BINS / PT consecutive bins per thread, wave scan, wave totals
                constexpr int BPT = BINS / PT;
                static_assert(BINS % PT == 0 && BPT >= 1, "bins per thread");
                uint32_t c[BPT], sum = 0;
#pragma unroll
                for (int q = 0; q < BPT; ++q) { c[q] = start[tid * BPT + q]; sum += c[q]; }
                uint32_t inc = sum;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const uint32_t up = (uint32_t)__shfl_up((int)inc, d);
                    if (lane >= d) inc += up;
                }
                if (lane == 63) wtot[wave] = inc;
                __syncthreads();
                uint32_t base = inc - sum;
                for (int w = 0; w < wave; ++w) base += wtot[w];
#pragma unroll
                for (int q = 0; q < BPT; ++q) { start[tid * BPT + q] = base; cur[tid * BPT + q] = base; base += c[q]; }
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < KPT; ++k)
                if (key[k] != 0xffffffffu) tmp[atomicAdd(&cur[key[k] >> 22], 1u)] = key[k];
            __syncthreads();
            for (int q = tid; q < (int)npad; q += PT) {
                if (q >= nf) {                                         // not a feature: stays where it is (fi >= nf: dead)
                    if constexpr (PARK_LDS) s_place[q] = (uint32_t)q;
                    else if (q >= p0 && q < p1) s_perm[q - p0] = (uint16_t)q;
                    continue;
                }
                const uint32_t kq = tmp[q];
                const uint32_t bin = kq >> 22, s0 = start[bin], e0 = cur[bin];
                uint32_t rank = 0;
                for (uint32_t r = s0; r < e0; ++r) rank += tmp[r] < kq ? 1u : 0u;
                const int place = (int)(s0 + rank);
                if constexpr (PARK_LDS) s_place[kq & 0x7ffu] = (uint32_t)place;
                else if (place >= p0 && place < p1) s_perm[place - p0] = (uint16_t)(kq & 0x7ffu);
            }
        }
    }
    __syncthreads();                                                   // B0
    if constexpr (PARK_LDS) {
        // the pair's features, read in list order (every column coalesced), each parked at its place of this workgroup's half
        const double Cref[3] = {s.u.Cref[0], s.u.Cref[1], s.u.Cref[2]};
        for (int i = tid; i < (int)npad; i += PT) {
            const FeatureRegs F = make_feature(load_feature_raw(a, (size_t)pair * a.max_features + (i < a.max_features ? i : 0), i < nf), Cref);
            const int place = ordered ? (int)s_place[i] : i;
            if (place >= p0 && place < p1) ws_store_feature(ws, pstride, place - p0, F);
        }
        __syncthreads();
    }

    for (int level = a.max_level - 1; level >= a.min_level; --level) {
        const LevelGeom lg = a.lv[level];
        const double scale = (double)(1.0f / (float)(1 << level));
        const double fs = (double)a.f * scale;
        int n_valid_lane = 0;
        if (solves && lane == 0) {                                     // GaussNewtonSolver entry (:304-308)
            s.chi2 = 0.0;
#pragma unroll
            for (int i = 0; i < 4; ++i) s.qo[i] = s.u.q[i];
#pragma unroll
            for (int i = 0; i < 3; ++i) s.to[i] = s.u.t[i];
        }
        // bit k: the thread's k-th patch passed this level's reference-side border test (kernels with LDS windows: <= 4 per thread)
        unsigned long long lvl_valid = 0ull;
        {
            const double Cref[3] = {s.u.Cref[0], s.u.Cref[1], s.u.Cref[2]};
            unsigned long long kbit = 1ull;
            for (int p = p0 + tid; p < p1; p += PT, kbit <<= 1) {
                const int lp = p - p0;
                const int slot = PARK_LDS ? lp : p;
                if constexpr (WCAP > 0) ws_win[15 * WCAP + lp] = WIN_EMPTY;     // the level's windows are filled by its first pass
                WsPatch w;
                if (!PARK_LDS && (WCAP == 0 || level == a.max_level - 1)) {
                    const int fi = ordered ? (int)s_perm[lp] : p;     // the pair's fi-th feature sits in this thread's slot p
                    w.F = make_feature(load_feature_raw(a, (size_t)pair * a.max_features + (fi < a.max_features ? fi : 0), fi < nf), Cref);
                    if constexpr (WCAP > 0) ws_store_feature(ws, pstride, slot, w.F);      // once per pair: read back at the finer levels
                } else {
                    ws_load_feature(ws, pstride, slot, w.F);          // parked once per pair (above / at the coarsest level)
                }
                const RefGeom g = ref_geom(w.F, lg, level);
                {
                    U32x3 rows[7];
                    ref_rows_issue(a, lg, ref_base, g, rows);
                    ref_rows_unpack(a, lg, g, rows, w.rlo, w.rhi);
                }
                if constexpr (WCAP > 0) {
                    ws_store_rows(ws, pstride, slot, w.rlo, w.rhi);
                    if (g.valid) lvl_valid |= kbit;
                } else {
                    w.F.ok = g.valid;
                    ws_store(ws, pstride, slot, w);                   // read back only by this same thread
                }
                n_valid_lane += g.valid ? 1 : 0;
            }
        }
        const int n_ref_wave = (int)wave_sum_to_lane63((double)n_valid_lane);   // valid in lane 63

        // H only depends on which patches are visible (inverse-compositional Jacobians): it is summed on the
        // level's first pass and again only when a lane of this wave sees a patch enter or leave the image —
        // the same numbers the reference recomputes every iteration (:289-291), bit for bit.
        unsigned long long vis_old = 0ull;
        const bool maskable = (size_t)(p1 - p0) <= (size_t)PT * 64;    // one bit per patch of this lane
        for (int it = 0; it < a.max_iters; ++it) {
            double b[6] = {0, 0, 0, 0, 0, 0};
            double chi2 = 0.0;
            int cnt = 0;
            unsigned long long vis_new = 0ull, bit = 1ull;
            for (int p = p0 + tid; p < p1; p += PT, bit <<= 1) {
                const int lp = p - p0;
                PatchRegs P;
                {
                    WsPatch w;
                    ws_load(ws, pstride, PARK_LDS ? lp : p, w);
                    if constexpr (WCAP > 0) w.F.ok = w.F.ok && (lvl_valid & bit) != 0ull;
                    ws_patch_regs(w, lg, level, P);
                }
                double c2, bp[6];
                bool vis;
                if constexpr (WCAP > 0) {
                    uint32_t worg = ws_win[15 * WCAP + lp];
                    const uint32_t worg0 = worg;
                    vis = residual_patch<WCAP>(a, lg, scale, fs, cur_base, P, s.u.R, s.u.tt, c2, bp, (LdsU32*)&ws_win[lp], &worg);
                    if (worg != worg0) ws_win[15 * WCAP + lp] = worg;
                } else {
                    vis = residual_patch<>(a, lg, scale, fs, cur_base, P, s.u.R, s.u.tt, c2, bp);
                }
                if (vis) {
#pragma unroll
                    for (int i = 0; i < 6; ++i) b[i] += bp[i];
                    chi2 += c2;
                    cnt += 1;
                    vis_new |= bit;
                }
            }
            // H is summed in a loop of its own (over the patches the pass found visible), so that its 21
            // accumulators are not live in the residual loop above: with them there the kernel spilled
            const bool h_new = (it == 0) || !maskable || __ballot(vis_new != vis_old) != 0ull;
            vis_old = vis_new;
            if (h_new) {
                double H[21];
#pragma unroll
                for (int i = 0; i < 21; ++i) H[i] = 0.0;
                bit = 1ull;
                for (int p = p0 + tid; p < p1; p += PT, bit <<= 1) {
                    const int lp = PARK_LDS ? p - p0 : p;
                    PatchRegs P;
                    if (maskable) {
                        if (!(vis_new & bit)) continue;
                        WsPatch w;
                        ws_load(ws, pstride, lp, w);                   // (visible => it passed the level's border test)
                        ws_patch_regs(w, lg, level, P);
                    } else {                                           // more patches per lane than mask bits: project again
                        WsPatch w;
                        ws_load(ws, pstride, lp, w);
                        ws_patch_regs(w, lg, level, P);
                        double u, v;
                        if (!P.valid || !project_patch(a, lg, scale, P.X, s.u.R, s.u.tt, u, v)) continue;
                    }
                    const PatchHess ph = patch_hess_factors(P, fs);
                    patch_hess_foreach<0, 0>(ph, [&](int q, double v) { H[q] += v; });
                }
#pragma unroll
                for (int i = 0; i < 21; ++i) {
                    const double hs = wave_sum_to_lane63(H[i]);
                    if (lane == 63) s_part[wave].H[i] = hs;
                }
            }
#pragma unroll
            for (int i = 0; i < 6; ++i) b[i] = wave_sum_to_lane63(b[i]);
            chi2 = wave_sum_to_lane63(chi2);
            const double cntd = wave_sum_to_lane63((double)cnt);
            if (lane == 63) {
#pragma unroll
                for (int i = 0; i < 6; ++i) s_part[wave].b[i] = b[i];
                s_part[wave].chi2 = chi2;
                s_part[wave].cnt = (int)cntd;
                s_part[wave].n_ref = n_ref_wave;
                s_part[wave].h_changed = h_new ? 1 : 0;
            }
            __syncthreads();                                           // B1
            SolverCarry carry1;                                        // one member: solver inline, the carry in registers
            if (solves) {
                auto* const carry = (__attribute__((address_space(3))) SolverCarry*)&s_carry;
                if constexpr (MEMBERS == 2) {
                    if (dead) {
                        if (lane == 0) s.ctrl = 1;
                    } else {
                        // this workgroup's partial: lane i < WPD folds double i of its wave partials in wave order (doubles 7
                        // and 8 are the packed counters: cnt | h_changed and n_ref | pad), publishes it as tagged words and
                        // polls the partner's
                        const unsigned long long tag = (unsigned long long)(g_it + 1u) << 32;
                        unsigned long long* const bank = twords + (size_t)(g_it & 1u) * 2 * DUO_WPD * 2;
                        if (lane < DUO_WPD) {
                            unsigned long long raw;
                            if (lane == 7 || lane == 8) {
                                int lo = 0, hi = 0;
#pragma unroll
                                for (int w = 0; w < NPW; ++w) {
                                    const int* q = (const int*)((const double*)&s_part[w] + lane);
                                    lo += q[0];
                                    hi |= (lane == 7) ? q[1] : 0;
                                }
                                raw = (unsigned long long)(unsigned)lo | ((unsigned long long)(unsigned)hi << 32);
                            } else {
                                double acc = 0.0;
#pragma unroll
                                for (int w = 0; w < NPW; ++w) acc += ((const double*)&s_part[w])[lane];
                                raw = (unsigned long long)__double_as_longlong(acc);
                            }
                            ((unsigned long long*)&s_mpart[member])[lane] = raw;
                            unsigned long long* dst = bank + ((size_t)member * DUO_WPD + lane) * 2;
                            __hip_atomic_store(dst, (raw & 0xffffffffull) | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store(dst + 1, (raw >> 32) | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        // (~1 s of polling: a partner is at most one pair's duration — milliseconds — behind in the dispatch order)
                        const unsigned spin_limit = a.spin_limit ? a.spin_limit : (1u << 20);
                        const int other = member ^ 1;
                        unsigned spins = 0;
                        for (;;) {
                            bool pending = false;
                            if (lane < DUO_WPD) {
                                const unsigned long long* src = bank + ((size_t)other * DUO_WPD + lane) * 2;
                                const unsigned long long w0 = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                const unsigned long long w1 = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                if ((w0 & 0xffffffff00000000ull) != tag || (w1 & 0xffffffff00000000ull) != tag) pending = true;
                                else ((unsigned long long*)&s_mpart[other])[lane] = (w0 & 0xffffffffull) | (w1 << 32);
                            }
                            if (__ballot(pending) == 0ull) break;
                            if (++spins >= spin_limit) break;
                            __builtin_amdgcn_s_sleep(1);
                        }
                        const bool ok = spins < spin_limit;
                        if (!ok) spin_timeout(a.timeout_flag);
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                        ws_solver_step<2>((const LdsWavePartial*)s_mpart, (LdsBlockState*)&s, carry, pair, level, it, lane);
                        if (!ok) { dead = true; if (lane == 0) s.ctrl = 1; }
                    }
                    ++g_it;
                } else {
                    // row of H^+ from LDS at every step (refresh_hinv_to_lds parks it there): nothing of the solver
                    // is live in registers across the pass
                    double hrow[6];
                    const int li = lane < 6 ? lane : 5;
#pragma unroll
                    for (int jj = 0; jj < 6; ++jj) hrow[jj] = s.Hinv[jj * 6 + li];
                    (void)solver_step<NPW>(a, pair, level, it, s_part, s, lane, hrow, carry1);
                }
            }
            __syncthreads();                                           // B2
            if constexpr (MEMBERS == 2) {
                if (solves && !dead)                                   // member 0 alone writes statistics
                    ws_solver_commit((LdsBlockState*)&s, (const __attribute__((address_space(3))) SolverCarry*)&s_carry,
                                     member == 0 ? a.stats : nullptr, a.max_iters, pair, lane);
            } else {
                if (solves) solver_commit(a, pair, s, lane, carry1);
            }
            if (s.ctrl) break;
        }
    }
    if (solves && member == 0) {
        solver_finish(a.T_cur_w + 12 * (size_t)pair, a.n_tracked + pair, (LdsBlockState*)&s, lane);
    }
}

// ---------------------------------------------------------------------------------------------
// Team kernel: ONE pair spread over K workgroups (K compute units) for feature counts beyond one
// workgroup's registers — the live tracker's case (BASELINE configs 3 and 5: 1000 / 2000 patches, one
// pair at a time), where the workspace kernel above makes a single CU loop over 3..5 chunks per pass.
// Member m keeps patches [m*448, (m+1)*448) in registers (one per lane, footprint windows in LDS, H summed
// on visibility changes only: the register kernel's pass). Per Gauss-Newton iteration every member publishes
// its seven wave partials in a team buffer in HBM (agent-scope atomic stores, a release fence, then its
// flag word; double-buffered by iteration parity) and reads the other members' (poll the flags, acquire,
// agent-scope loads). Every member's solver wave then runs the ordinary solver_step over the same 7*K
// partials in the same order, so all members hold the same pose bit for bit without a broadcast: ONE
// exchange per iteration. Member 0 alone writes pose, count and statistics. Members of a team are
// workgroups b, b + P, b + 2P.. with P a multiple of 8, i.e. on the same XCD (one L2). Launched only
// when every team is resident at once (teams * K <= half the CUs); a wait that never ends raises the
// same timeout flag as the register kernel's hand-over, stops the pair and drains its barriers.
// ---------------------------------------------------------------------------------------------
// The exchange buffer holds TAGGED WORDS: every 64-bit word carries 32 bits of payload and, in its upper
// half, the number of the iteration it belongs to. 64-bit atomic stores and loads are single-copy atomic, so
// (with the launch's epoch, so that the words of earlier launches in the same buffer never match: no memset per launch);
// a reader that finds the expected tag in a word holds that iteration's payload — no flag, no fence and no
// second round trip: the writer fires its stores and goes on, a reader polls the words themselves until
// every tag matches. Two banks (iteration parity): a member can only overwrite the bank of iteration g at
// iteration g + 2, which it reaches after every other member has published g + 1, i.e. has read all of g.
constexpr int TEAM_WPD = sizeof(WavePartial) / sizeof(double);          // doubles of one (member) partial
constexpr int TEAM_MAX_MEMBERS = 64;                                     // one solver lane per member partial
constexpr size_t TEAM_BYTES = 2 * TEAM_MAX_MEMBERS * TEAM_WPD * 2 * sizeof(unsigned long long);   // banks x members x words
static_assert(TEAM_BYTES <= 65536, "team buffer");

// K = member capacity of the instantiation: teams of 2..16 members run the instantiation of exactly their size,
// larger ones (up to 64 members = 16 384 features) the 64-member instantiation with `k` members at run time (the
// absent members' partials are zero, so the solver's fixed-order sums are unchanged by them).
template <int K, int NPW>
__global__ __launch_bounds__((NPW + 1) * 64) void sparse_align_team_kernel(const SAKernelArgs a, int pairs_pad, int k) {
    constexpr int PT = NPW * 64, WPD = TEAM_WPD;
    static_assert(K <= TEAM_MAX_MEMBERS && WPD <= 32, "one solver lane per member partial / per double of a partial");
    __shared__ WavePartial s_part[NPW];     // this member's wave partials
    __shared__ WavePartial s_mpart[K];      // one partial per member (its waves summed in wave order), member order
    __shared__ BlockState s;
    __shared__ uint32_t s_win[WIN_ROWS * 3 * PT];

    const int pair = (int)blockIdx.x % pairs_pad, member = (int)blockIdx.x / pairs_pad;
    if (pair >= a.n_pairs) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nf = a.n_features ? a.n_features[pair] : a.max_features;
    if (nf < a.min_fts || a.max_level - 1 < a.min_level) {              // the same decision in every member
        if (member == 0 && tid == 0) { a.n_tracked[pair] = 0; stats_clear(a, pair); }
        return;
    }
    unsigned long long* const twords = (unsigned long long*)((char*)a.workspace + (size_t)pair * TEAM_BYTES);
    WavePartial* const my_part = s_part;

    if (wave == NPW) {
        // Every member runs the SAME solver on the SAME K member partials in the same order, so every member
        // holds the same pose, bit for bit, without a broadcast: one exchange per iteration. Member 0 alone
        // writes results.
        SAKernelArgs am = a;
        if (member != 0) am.stats = nullptr;
        unsigned g = 0;                                                  // iterations so far, over all levels
        const unsigned spin_limit = a.spin_limit ? a.spin_limit : SPIN_LIMIT;
        if (k < K)                                                       // absent members contribute zeros
            for (int i = lane; i < (K - k) * WPD; i += 64) ((unsigned long long*)&s_mpart[k])[i] = 0ull;
        solver_init(a.T_ref_w + 12 * (size_t)pair, a.T_cur_w + 12 * (size_t)pair, (LdsBlockState*)&s, lane);
        if (member == 0 && lane == 0) stats_clear(a, pair);
        __syncthreads();                                                 // B0
        bool dead = false;                                               // a team wait timed out: drain, keep the barriers matched
        for (int level = a.max_level - 1; level >= a.min_level; --level) {
            if (lane == 0) {
                s.chi2 = 0.0;
#pragma unroll
                for (int i = 0; i < 4; ++i) s.qo[i] = s.u.q[i];
#pragma unroll
                for (int i = 0; i < 3; ++i) s.to[i] = s.u.t[i];
            }
            double hrow[6];
            for (int it = 0; it < a.max_iters; ++it) {
                __syncthreads();                                         // B1: this member's wave partials are in s_part
                int ctrl;
                if (dead) {
                    ctrl = 1;
                    if (lane == 0) s.ctrl = 1;
                } else {
                    // tag = launch epoch (20 bits) | iteration (12 bits): words a previous launch left in this ring slot never
                    // match, so the buffer needs no memset per launch (one stream operation less in front of every Run)
                    const unsigned long long tag = (unsigned long long)(((a.team_epoch & 0xfffffu) << 12) | ((g + 1u) & 0xfffu)) << 32;
                    unsigned long long* const bank = twords + (size_t)(g & 1u) * TEAM_MAX_MEMBERS * WPD * 2;
                    // this member's partial: lane i < WPD folds double i of its wave partials in wave order
                    // (doubles 7 and 8 are the packed counters: cnt | h_changed and n_ref | pad)
                    if (lane < WPD) {
                        unsigned long long raw;
                        if (lane == 7 || lane == 8) {
                            int lo = 0, hi = 0;
#pragma unroll
                            for (int w = 0; w < NPW; ++w) {
                                const int* q = (const int*)((const double*)&s_part[w] + lane);
                                lo += q[0];
                                hi |= (lane == 7) ? q[1] : 0;
                            }
                            raw = (unsigned long long)(unsigned)lo | ((unsigned long long)(unsigned)hi << 32);
                        } else {
                            double acc = 0.0;
#pragma unroll
                            for (int w = 0; w < NPW; ++w) acc += ((const double*)&s_part[w])[lane];
                            raw = (unsigned long long)__double_as_longlong(acc);
                        }
                        ((unsigned long long*)&s_mpart[member])[lane] = raw;
                        unsigned long long* dst = bank + ((size_t)member * WPD + lane) * 2;
                        __hip_atomic_store(dst, (raw & 0xffffffffull) | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(dst + 1, (raw >> 32) | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    // the other members' partials: poll the tagged words themselves
                    unsigned spins = 0;
                    for (;;) {
                        bool pending = false;
                        for (int i = lane; i < k * WPD; i += 64) {
                            if (i / WPD == member) continue;
                            const unsigned long long w0 = __hip_atomic_load(bank + 2 * (size_t)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            const unsigned long long w1 = __hip_atomic_load(bank + 2 * (size_t)i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if ((w0 & 0xffffffff00000000ull) != tag || (w1 & 0xffffffff00000000ull) != tag) pending = true;
                            else ((unsigned long long*)s_mpart)[i] = (w0 & 0xffffffffull) | (w1 << 32);
                        }
                        if (__ballot(pending) == 0ull) break;
                        if (++spins >= spin_limit) break;
                        __builtin_amdgcn_s_sleep(1);
                    }
                    const bool ok = spins < spin_limit;
                    if (!ok) spin_timeout(a.timeout_flag);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    SolverCarry carry;
                    ctrl = solver_step<K>(am, pair, level, it, s_mpart, s, lane, hrow, carry);
                    if (!ok) { ctrl = 1; dead = true; if (lane == 0) s.ctrl = 1; }
                    ++g;
                    __syncthreads();                                     // B2
                    solver_commit(am, pair, s, lane, carry);
                    if (ctrl) break;
                    continue;
                }
                ++g;
                __syncthreads();                                         // B2
                if (ctrl) break;
            }
        }
        if (member == 0) {
            solver_finish(a.T_cur_w + 12 * (size_t)pair, a.n_tracked + pair, (LdsBlockState*)&s, lane);
        }
        return;
    }

    // ---- patch waves: one patch per lane, in registers ------------------------------------------
    const uint8_t* __restrict__ ref_base = a.ref_pyr + (size_t)pair * a.pyr_pitch;
    const uint8_t* __restrict__ cur_base = a.cur_pyr + (size_t)pair * a.pyr_pitch;
    const int p = member * PT + tid;                                     // this lane's feature
    const FeatureRaw fraw = load_feature_raw(a, (size_t)pair * a.max_features + (p < a.max_features ? p : 0), p < nf && p < a.max_features);
    __syncthreads();                                                     // B0
    FeatureRegs F;
    {
        const double Cref[3] = {s.u.Cref[0], s.u.Cref[1], s.u.Cref[2]};
        F = make_feature(fraw, Cref);
    }
    LdsU32* const win = (LdsU32*)&s_win[tid];
    for (int level = a.max_level - 1; level >= a.min_level; --level) {
        const LevelGeom lg = a.lv[level];
        const double scale = (double)(1.0f / (float)(1 << level));
        const double fs = (double)a.f * scale;
        PatchRegs P;
        uint32_t worg = WIN_EMPTY;
        precompute_patch(a, lg, level, ref_base, F, P);
        const int n_ref_wave = __popcll(__ballot(P.valid));
        unsigned long long cached_mask = 0ull;
        for (int it = 0; it < a.max_iters; ++it) {
            double chi2, b[6];
            pin_patch(P);
            const bool vis = residual_patch<PT>(a, lg, scale, fs, cur_base, P, s.u.R, s.u.tt, chi2, b, win, &worg);
            const unsigned long long vmask = __ballot(vis);
            const bool h_new = (it == 0) || (vmask != cached_mask);
#pragma unroll
            for (int i = 0; i < 6; ++i) b[i] = wave_sum_to_lane63(b[i]);
            chi2 = wave_sum_to_lane63(chi2);
            if (lane == 63) {
#pragma unroll
                for (int i = 0; i < 6; ++i) my_part[wave].b[i] = b[i];
                my_part[wave].chi2 = chi2;
                my_part[wave].cnt = __popcll(vmask);
                my_part[wave].n_ref = n_ref_wave;
                my_part[wave].h_changed = h_new ? 1 : 0;
            }
            if (h_new) {
                const PatchHess ph = patch_hess_factors(P, fs);
                double* Hout = my_part[wave].H;
                patch_hess_foreach<0, 0>(ph, [&](int q, double v) {
                    const double hs = wave_sum_to_lane63(vis ? v : 0.0);
                    if (lane == 63) Hout[q] = hs;
                    __builtin_amdgcn_sched_barrier(0);   // one entry live at a time
                });
                cached_mask = vmask;
            }
            __syncthreads();                                             // B1
            __syncthreads();                                             // B2
            if (s.ctrl) break;
        }
    }
}

static thread_local int g_team_drop = 0;      // tests only (sparse_align_launch_team's drop_members)
template <int K, int NPW>
static hipError_t launch_team(const SAKernelArgs& args, int pairs_pad, int k, hipStream_t stream) {
    hipLaunchKernelGGL((sparse_align_team_kernel<K, NPW>), dim3((unsigned)(pairs_pad * (k - g_team_drop))), dim3((NPW + 1) * 64), 0, stream, args, pairs_pad, k);
    return hipGetLastError();
}

// Team shape (few pairs of more than 448 features): members of 4 patch waves (256 patches: every SIMD of a member's CU carries one patch wave, the
// pass runs at its uncontended ~3 k cycles and the level start's gathers spread over more CUs; members of 7
// waves were 15 % slower), K = ceil(N / 256) = 2..64 members (N <= 16 384: one solver lane per member partial).
// Returns K, or 0 when the team kernel does not apply (then: workspace kernel).
constexpr int TEAM_NPW = 4;
static int team_pairs_pad(int n_pairs, int k) {
    // members of a team are workgroups b, b + P, b + 2P ..: P a multiple of 8 keeps a team on one XCD (one L2, 32
    // CUs); teams of more than 32 members cannot fit one XCD and are spread over all of them instead (P odd)
    if (k < options().team_spread_min) return (n_pairs + 7) / 8 * 8;
    return n_pairs | 1;
}
int sparse_align_team_size(int n_pairs, int max_features, int num_cus) {
    // from 449 features a team of 2..3 CUs beats the 11 + 1 wave register kernel on one CU (N = 600: 0.113 vs
    // 0.121 ms per Run); below, one CU wins (N = 300: 0.108 vs 0.115 ms).
    if (max_features < options().team_min || n_pairs <= 0) return 0;
    const int k = (max_features + TEAM_NPW * 64 - 1) / (TEAM_NPW * 64);
    // every member must be resident at once (they spin on each other): the LIVE workgroups (n_pairs * k; the
    // padding workgroups exit at once) may take half the CUs
    if (k > TEAM_MAX_MEMBERS || n_pairs * k > num_cus / 2) return 0;
    return k;
}
size_t sparse_align_team_bytes(int n_pairs) { return (size_t)n_pairs * TEAM_BYTES; }

template <int K>
static hipError_t launch_team_k(const SAKernelArgs& args, int k, int pairs_pad, hipStream_t stream) {
    if constexpr (K > 16) return launch_team<TEAM_MAX_MEMBERS, TEAM_NPW>(args, pairs_pad, k, stream);
    else {
        if (k == K) return launch_team<K, TEAM_NPW>(args, pairs_pad, k, stream);
        return launch_team_k<K + 1>(args, k, pairs_pad, stream);
    }
}

hipError_t sparse_align_launch_team(const SAKernelArgs& args, int k, hipStream_t stream, int drop_members) {
    if (args.n_pairs <= 0) return hipSuccess;
    if (k < 2 || k > TEAM_MAX_MEMBERS || drop_members < 0 || drop_members >= k) return hipErrorInvalidValue;
    g_team_drop = drop_members;
    const hipError_t e = launch_team_k<2>(args, k, team_pairs_pad(args.n_pairs, k), stream);
    g_team_drop = 0;
    return e;
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
// Register-kernel shapes: NPW patch waves + 1 solver wave per pair slot, PPW slots per workgroup,
// always <= 12 waves (3 per SIMD at 168 VGPRs = the whole register file of a CU):
//   <= 128 features: 2+1 waves x 4 slots     <= 192: 3+1 x 3     <= 256: 4+1 x 2
//   <= 320 features: 5+1 x 2 (BASELINE shape) <= 448: 7+1 x 1     <= 704: 11+1 x 1
SAVariant sparse_align_pick_variant(int max_features) {
    if (max_features > options().ws_from) return SA_WS;   // 704 unless a diagnostic run moved it
    if (max_features <= 128) return SA_REG128;
    if (max_features <= 192) return SA_REG192;
    if (max_features <= 256) return SA_REG256;
    if (max_features <= 320) return SA_REG320;
    if (max_features <= 448) return SA_REG448;
    if (max_features <= 704) return SA_REG704;
    return SA_WS;
}

size_t sparse_align_workspace_bytes(int n_pairs, int max_features) {
    if (sparse_align_pick_variant(max_features) != SA_WS) return 0;
    if (!options().ws_no_windows && (max_features + 63) / 64 * 64 <= 1024) return 0;     // parked in LDS
    // (1025..2048 patches: the two-member kernel needs DUO_BYTES per pair, less than the parking space reserved here)
    return (size_t)n_pairs * ws_doubles_per_pair(max_features) * sizeof(double);
}

#ifdef DSDTM_DIAG
int sparse_align_occupancy(int variant) {
    int nb = -1;
    if (variant == SA_REG320) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, sparse_align_reg_kernel<5, SA_PPW, false>, SA_PPW * 6 * 64, 0);
    else if (variant == SA_REG448) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, sparse_align_reg_kernel<7, 1, false>, 8 * 64, 0);
    else if (variant == SA_WS) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, sparse_align_ws_kernel<WS_NPW, 0>, WS_THREADS, 0);
    return nb;
}
#endif

// persistent grid: one workgroup per CU (fewer when the batch is small); slots pull pairs dynamically
static unsigned persistent_grid(int n_pairs, int ppw, int num_cus) {
    const int need = (n_pairs + ppw - 1) / ppw;
    return (unsigned)(need < num_cus ? need : num_cus);
}

template <int NPW, int PPW, bool STAMPS>
static hipError_t launch_reg(const SAKernelArgs& args, int num_cus, hipStream_t stream) {
    static_assert(sizeof(RegSmem<NPW, PPW>) <= 64 * 1024, "static LDS");
    hipLaunchKernelGGL((sparse_align_reg_kernel<NPW, PPW, STAMPS>), dim3(persistent_grid(args.n_pairs, PPW, num_cus)),
                       dim3(PPW * (NPW + 1) * 64), 0, stream, args);
    return hipGetLastError();
}

#ifdef DSDTM_DIAG
hipError_t sparse_align_launch_stamps(const SAKernelArgs& args, int num_cus, hipStream_t stream) {
    if (args.n_pairs <= 0) return hipSuccess;
    return launch_reg<5, SA_PPW, true>(args, num_cus, stream);
}
#endif

// More than 64 KB of dynamic LDS is an opt-in per kernel AND per device (the sharded entry launches from one
// process on up to eight devices): set once for every (instantiation, device) pair, on the calling thread's device.
template <auto Kernel>
static hipError_t optin_dynamic_lds(size_t bytes) {
    static std::mutex m;
    static bool done[64] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipFuncSetAttribute((const void*)Kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    std::lock_guard<std::mutex> g(m);
    if (!done[dev]) {
        e = hipFuncSetAttribute((const void*)Kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return e;
        done[dev] = true;
    }
    return hipSuccess;
}

bool sparse_align_uses_duo(int max_features, bool have_workspace) {
    if (sparse_align_pick_variant(max_features) != SA_WS) return false;
    const int npad = (max_features + 63) / 64 * 64;
    // (only where two halves cost no more lane rounds than the whole: 2000 patches = 2 x 2 rounds of 512 lanes
    // against 4; 1500 patches would be 2 x 2 against 3 — measured 10 % slower on two compute units)
    return !options().ws_no_windows && npad > 1024 && npad <= 2048 && !options().ws_no_duo && have_workspace &&
           2 * (((npad / 64 + 1) / 2 * 64 + WS_THREADS - 1) / WS_THREADS) <= (npad + WS_THREADS - 1) / WS_THREADS;
}

hipError_t sparse_align_launch(const SAKernelArgs& args, SAVariant variant, int num_cus, hipStream_t stream, bool allow_multi_cu) {
    if (args.n_pairs <= 0) return hipSuccess;
    switch (variant) {
        case SA_REG128: return launch_reg<2, 4, false>(args, num_cus, stream);
        case SA_REG192: return launch_reg<3, 3, false>(args, num_cus, stream);
        case SA_REG256: return launch_reg<4, 2, false>(args, num_cus, stream);
        case SA_REG320: return launch_reg<5, SA_PPW, false>(args, num_cus, stream);
        case SA_REG448: return launch_reg<7, 1, false>(args, num_cus, stream);
        case SA_REG704: return launch_reg<11, 1, false>(args, num_cus, stream);
        case SA_WS: {
            const int npad = (args.max_features + 63) / 64 * 64;
            const bool ws_windows = !options().ws_no_windows;
            const dim3 grid((unsigned)args.n_pairs);
            if (ws_windows && npad <= 1024) {
                // everything a pass needs in LDS: window origins + windows + parked grid inputs (4 + 60 + 88 KB)
                constexpr size_t lds = (16 + WS_DWORDS) * 1024 * sizeof(uint32_t);
                const hipError_t attr = optin_dynamic_lds<sparse_align_ws_kernel<WS_NPW, 1024, true>>(lds);
                if (attr != hipSuccess) return attr;
                hipLaunchKernelGGL((sparse_align_ws_kernel<WS_NPW, 1024, true>), grid, dim3(WS_THREADS), lds, stream, args);
            } else if (allow_multi_cu && sparse_align_uses_duo(args.max_features, args.workspace != nullptr)) {
                // one pair on two compute units, each half wholly in LDS; the exchange words are zeroed per launch
                constexpr size_t lds = (16 + WS_DWORDS) * 1024 * sizeof(uint32_t);
                const hipError_t attr = optin_dynamic_lds<sparse_align_ws_kernel<WS_NPW, 1024, true, 2>>(lds);
                if (attr != hipSuccess) return attr;
                const hipError_t ez = hipMemsetAsync(args.workspace, 0, (size_t)args.n_pairs * DUO_BYTES, stream);
                if (ez != hipSuccess) return ez;
                const dim3 grid2((unsigned)((args.n_pairs + 7) / 8 * 16));
                hipLaunchKernelGGL((sparse_align_ws_kernel<WS_NPW, 1024, true, 2>), grid2, dim3(WS_THREADS), lds, stream, args);
            } else if (ws_windows && npad <= 2048) {
                const hipError_t attr = optin_dynamic_lds<sparse_align_ws_kernel<WS_NPW, 2048>>(16 * 2048 * sizeof(uint32_t));
                if (attr != hipSuccess) return attr;
                hipLaunchKernelGGL((sparse_align_ws_kernel<WS_NPW, 2048>), grid, dim3(WS_THREADS), 16 * 2048 * sizeof(uint32_t), stream, args);
            } else {
                hipLaunchKernelGGL((sparse_align_ws_kernel<WS_NPW, 0>), grid, dim3(WS_THREADS), 0, stream, args);
            }
            break;
        }
    }
    return hipGetLastError();
}

}  // namespace dsdtm
