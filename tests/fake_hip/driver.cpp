// tests/fake_hip/driver.cpp — scenarios that drive the HOST side of the C ABI (dsdtm_amd/csrc/api.cpp: stream rings, pair
// counters, recover slots, the team epoch, the sharded / streamed entries, frames, the single-call entries' packing) against the
// fake HIP runtime, under AddressSanitizer + UBSan (or ThreadSanitizer for the two-thread scenario). TEST INFRASTRUCTURE ONLY.
//   usage: driver [scenario ...]      (no argument: all of them); prints "ok <name>" per scenario, exits non-zero on failure
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "../../include/dsdtm_amd.h"
#include "fake_hip.h"

extern "C" long long dsdtm_debug_recovered_launches(dsdtm_ctx*);
extern "C" long long dsdtm_debug_team_seq(dsdtm_ctx*, long long);
extern "C" int dsdtm_debug_set_option(const char*, int);

#define CHECK(cond)                                                                                  \
    do {                                                                                             \
        if (!(cond)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); return false; } \
    } while (0)

static const int W = 64, H = 48, L = 3;
struct Geometry { int w[8], h[8], st[8]; size_t off[8]; size_t pitch; };
static Geometry geometry() {
    Geometry g{};
    size_t o = 0;
    for (int l = 0; l < L; ++l) { g.w[l] = l ? (g.w[l - 1] + 1) / 2 : W; g.h[l] = l ? (g.h[l - 1] + 1) / 2 : H; g.st[l] = g.w[l]; g.off[l] = o; o += ((size_t)g.w[l] * g.h[l] + 63) / 64 * 64; }
    g.pitch = (o + 255) / 256 * 256;
    return g;
}
static dsdtm_camera camera() { return dsdtm_camera{60.f, 60.f, 32.f, 24.f, 60.f, W, H}; }
static dsdtm_align_params params() { return dsdtm_align_params{L, 0, 10, 15}; }

// a batch whose arrays live in "device" memory (hipMalloc of the fake: the host heap) or in plain host memory
struct Batch {
    dsdtm_batch_desc b{};
    std::vector<void*> mem;
    bool device;
    Batch(int n_pairs, int n_features, bool on_device) : device(on_device) {
        const Geometry g = geometry();
        b.n_pairs = n_pairs; b.max_features = n_features; b.levels = L; b.pyr_pitch = g.pitch;
        for (int l = 0; l < L; ++l) { b.width[l] = g.w[l]; b.height[l] = g.h[l]; b.stride[l] = g.st[l]; b.level_offset[l] = g.off[l]; }
        const size_t P = (size_t)n_pairs, N = (size_t)n_features;
        b.ref_pyr = (const uint8_t*)get(P * g.pitch); b.cur_pyr = (const uint8_t*)get(P * g.pitch);
        b.px_xy = (const float*)get(P * N * 8); b.bearing = (const double*)get(P * N * 24); b.p_world = (const double*)get(P * N * 24);
        b.initial = (const uint8_t*)get(P * N); b.T_ref_w = (const double*)get(P * 96); b.T_cur_w = (double*)get(P * 96);
        b.n_tracked = (int32_t*)get(P * 4); b.stats = (dsdtm_align_stats*)get(P * sizeof(dsdtm_align_stats));
        for (size_t i = 0; i < P * 12; ++i) b.T_cur_w[i] = 0.5;
    }
    void* get(size_t bytes) {
        void* p = nullptr;
        if (device) { if (hipMalloc(&p, bytes) != hipSuccess) std::abort(); }
        else p = std::malloc(bytes ? bytes : 1);
        std::memset(p, 0, bytes);
        mem.push_back(p);
        return p;
    }
    void release() { for (void* p : mem) { if (device) (void)hipFree(p); else std::free(p); } mem.clear(); }
    ~Batch() { release(); }
};

static bool no_fake_errors() {
    for (const std::string& e : fake_hip_errors()) std::fprintf(stderr, "fake kernel: %s\n", e.c_str());
    return fake_hip_errors().empty();
}

// ---- 1. more streams than the ring table tracks: the 17th takes over the entry idle longest ----
static bool ring_table_past_16_streams() {
    dsdtm_ctx* ctx = nullptr;
    CHECK(dsdtm_create(0, &ctx) == DSDTM_OK);
    const dsdtm_camera cam = camera(); const dsdtm_align_params prm = params();
    std::vector<hipStream_t> st(20);
    std::vector<Batch*> bt;
    for (auto& s : st) CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking) == hipSuccess);
    for (size_t i = 0; i < st.size(); ++i) bt.push_back(new Batch(5, 100, true));
    for (int round = 0; round < 3; ++round) {
        for (size_t i = 0; i < st.size(); ++i) CHECK(dsdtm_sparse_align_batch_device(ctx, &bt[i]->b, &cam, &prm, st[i]) == DSDTM_OK);
        for (size_t i = 0; i < st.size(); ++i) {
            CHECK(dsdtm_sparse_align_check(ctx, st[i]) == DSDTM_OK);
            CHECK(bt[i]->b.T_cur_w[3] > 0.0 && bt[i]->b.n_tracked[4] == 100);         // the launch ran (marker = its sequence number)
        }
    }
    CHECK(no_fake_errors());                       // e.g. a pair-counter word shared by two launches that did not drain in between
    for (Batch* b : bt) delete b;
    dsdtm_destroy(ctx);
    for (auto& s : st) CHECK(hipStreamDestroy(s) == hipSuccess);
    return true;
}

// ---- 2. a timeout raised on a stream whose ring entry is handed to another stream before anybody checked (r04 advice) ----
static bool evicted_timeout_is_reported() {
    dsdtm_ctx* ctx = nullptr;
    CHECK(dsdtm_create(0, &ctx) == DSDTM_OK);
    const dsdtm_camera cam = camera(); const dsdtm_align_params prm = params();
    std::vector<hipStream_t> st(18);
    std::vector<Batch*> bt;
    for (auto& s : st) CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking) == hipSuccess);
    for (size_t i = 0; i < st.size(); ++i) bt.push_back(new Batch(3, 100, true));
    fake_hip_timeout_next(0, 1);                                                 // the first one-CU launch that RUNS raises its word
    CHECK(dsdtm_sparse_align_batch_device(ctx, &bt[0]->b, &cam, &prm, st[0]) == DSDTM_OK);
    CHECK(hipStreamSynchronize(st[0]) == hipSuccess);                            // it ran; nobody checked
    for (size_t i = 1; i < st.size(); ++i) CHECK(dsdtm_sparse_align_batch_device(ctx, &bt[i]->b, &cam, &prm, st[i]) == DSDTM_OK);   // evicts stream 0's entry
    CHECK(dsdtm_sparse_align_check(ctx, st[5]) == DSDTM_ERR_HIP);                // not lost with the entry
    CHECK(dsdtm_sparse_align_check(ctx, st[5]) == DSDTM_OK);                     // reported once
    for (size_t i = 1; i < st.size(); ++i) CHECK(dsdtm_sparse_align_check(ctx, st[i]) == DSDTM_OK);
    // the stopped launch left its pair-counter word behind: the entry's next user must not start on it
    CHECK(dsdtm_sparse_align_batch_device(ctx, &bt[0]->b, &cam, &prm, st[0]) == DSDTM_OK);
    CHECK(dsdtm_sparse_align_check(ctx, st[0]) == DSDTM_OK);
    for (Batch* b : bt) delete b;
    dsdtm_destroy(ctx);
    for (auto& s : st) CHECK(hipStreamDestroy(s) == hipSuccess);
    return true;
}

// ---- 3. more unsettled multi-CU launches than recover slots; the re-run goes to the launch's OWN stream (r04 advice) ----
static bool recover_slots_past_64() {
    dsdtm_ctx* ctx = nullptr;
    CHECK(dsdtm_create(0, &ctx) == DSDTM_OK);
    const dsdtm_camera cam = camera(); const dsdtm_align_params prm = params();
    hipStream_t st;
    CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess);
    std::vector<Batch*> bt;
    for (int i = 0; i < 70; ++i) bt.push_back(new Batch(2, 600, true));          // 2 pairs x 600 features: a team of 3 per pair
    fake_hip_timeout_next(1, 0);                                                 // the first team launch that runs times out
    for (int i = 0; i < 70; ++i) {
        CHECK(dsdtm_sparse_align_batch_device(ctx, &bt[i]->b, &cam, &prm, st) == DSDTM_OK);
        CHECK(dsdtm_debug_recovered_launches(ctx) == (i >= 64 ? 1 : 0));         // the 65th launch settled the oldest
    }
    CHECK(dsdtm_sparse_align_check(ctx, st) == DSDTM_OK);
    CHECK(dsdtm_debug_recovered_launches(ctx) == 1);
    bool saw_rerun = false;
    for (const fake_launch& fl : fake_hip_log())
        if (fl.T_cur_w == bt[0]->b.T_cur_w && !fl.timed_out) {
            CHECK(fl.kind == FAKE_SA_ONE_CU);                                    // re-run on the one-CU kernels ...
            CHECK(fl.stream == st);                                              // ... on the stream it was launched on (ordered with later work there)
            saw_rerun = true;
        }
    CHECK(saw_rerun && bt[0]->b.T_cur_w[3] > 0.0 && bt[0]->b.n_tracked[0] == 600);   // re-seeded, re-run: not the aborted launch's -1
    for (int i = 1; i < 70; ++i) CHECK(bt[i]->b.T_cur_w[3] > 0.0);
    CHECK(no_fake_errors());
    for (Batch* b : bt) delete b;
    dsdtm_destroy(ctx);
    CHECK(hipStreamDestroy(st) == hipSuccess);
    return true;
}

// ---- 4. the team ring across the wrap of its 20-bit tag epoch (r04 advice) ----
static bool team_epoch_wrap(bool clear_on_wrap) {
    dsdtm_ctx* ctx = nullptr;
    CHECK(dsdtm_create(0, &ctx) == DSDTM_OK);
    CHECK(dsdtm_debug_set_option("team_no_wrap_clear", clear_on_wrap ? 0 : 1) == DSDTM_OK);
    const dsdtm_camera cam = camera(); const dsdtm_align_params prm = params();
    Batch big(2, 1000, true), small(2, 500, true);
    auto run = [&](Batch& b) { return dsdtm_sparse_align_batch_device(ctx, &b.b, &cam, &prm, nullptr) == DSDTM_OK && dsdtm_sparse_align_check(ctx, nullptr) == DSDTM_OK; };
    CHECK(dsdtm_debug_team_seq(ctx, 0) == 0);
    CHECK(run(big));                                  // epoch 1, ring slot 1: its words stay behind
    dsdtm_debug_team_seq(ctx, 0xffffd);
    CHECK(run(small)); CHECK(run(big));               // 0xffffe, 0xfffff
    CHECK(run(big));                                  // wraps: epoch 0x100001 — tag 1, ring slot 1 again
    CHECK(dsdtm_debug_team_seq(ctx, -1) == 1);
    CHECK(run(small));
    const bool clean = fake_hip_errors().empty();
    CHECK(dsdtm_debug_set_option("team_no_wrap_clear", 0) == DSDTM_OK);
    dsdtm_destroy(ctx);
    if (clear_on_wrap) { CHECK(no_fake_errors()); }
    else CHECK(!clean);                               // without the re-zeroing the fake kernel DOES meet a stale word with its own tag: the scenario bites
    return true;
}
static bool team_epoch_wrap_is_cleared() { return team_epoch_wrap(true); }
static bool team_epoch_wrap_hazard_is_real() { return team_epoch_wrap(false); }

// ---- 5. the sharded entry: results fetched again after a transparent re-run (r04 advice) ----
static bool sharded_refetches_after_rerun() {
    dsdtm_ctx* ctx[2] = {nullptr, nullptr};
    for (auto& c : ctx) CHECK(dsdtm_create(0, &c) == DSDTM_OK);
    const dsdtm_camera cam = camera(); const dsdtm_align_params prm = params();
    Batch hb(6, 600, false);                          // host arrays; 3 pairs per shard, teams of 3
    fake_hip_timeout_next(1, 0);
    CHECK(dsdtm_sparse_align_batch_sharded(ctx, 2, &hb.b, &cam, &prm) == DSDTM_OK);
    for (int i = 0; i < 6; ++i) CHECK(hb.b.T_cur_w[12 * i + 3] > 0.0 && hb.b.n_tracked[i] == 600);   // never the aborted launch's -1
    CHECK(dsdtm_debug_recovered_launches(ctx[0]) + dsdtm_debug_recovered_launches(ctx[1]) == 1);
    CHECK(no_fake_errors());
    for (auto& c : ctx) dsdtm_destroy(c);
    return true;
}

// ---- 6. an error in the middle of a shard: no copy touches the caller's arrays after the return, no record survives that
//         points into a staging buffer the next call regrows (r04 advice) ----
static bool sharded_error_midway() {
    for (int fail_at = 1; fail_at <= 14; ++fail_at) {
        dsdtm_ctx* ctx[2] = {nullptr, nullptr};
        for (auto& c : ctx) CHECK(dsdtm_create(0, &c) == DSDTM_OK);
        const dsdtm_camera cam = camera(); const dsdtm_align_params prm = params();
        {
            Batch hb(4, 600, false);
            fake_hip_timeout_next(1, 0);              // ... and the launch that does get through times out
            fake_hip_fail("hipMemcpyAsync", fail_at); // uploads (1-9), the recover seed copy (10), downloads (11-13) of whichever shard gets there first
            const int rc = dsdtm_sparse_align_batch_sharded(ctx, 2, &hb.b, &cam, &prm);
            CHECK(rc == DSDTM_ERR_HIP || rc == DSDTM_OK);
            CHECK(fake_hip_pending() == 0);           // nothing is still queued that names the caller's arrays
        }                                             // the caller frees its arrays here
        fake_hip_timeout_next(0, 0);
        fake_hip_drain_all();
        Batch big(24, 600, false);                    // a larger batch: both staging buffers are regrown (the old ones freed)
        CHECK(dsdtm_sparse_align_batch_sharded(ctx, 2, &big.b, &cam, &prm) == DSDTM_OK);
        for (int i = 0; i < 24; ++i) CHECK(big.b.T_cur_w[12 * i + 3] > 0.0);
        for (auto& c : ctx) dsdtm_destroy(c);
        fake_hip_reset();
    }
    return true;
}

// ---- 7. the streamed entry: chained and separate frames over three contexts, an error in the middle ----
static bool streamed_entry() {
    const Geometry g = geometry();
    const dsdtm_camera cam = camera(); const dsdtm_align_params prm = params();
    for (int chained = 0; chained < 2; ++chained)
        for (int fail_at = 0; fail_at <= 12; fail_at += 4) {
            dsdtm_ctx* ctx[3] = {nullptr, nullptr, nullptr};
            for (auto& c : ctx) CHECK(dsdtm_create(0, &c) == DSDTM_OK);
            {
                const int P = 11, N = 120;
                Batch hb(P, N, false);
                std::vector<uint8_t> ref((size_t)(P + 1) * W * H, 9), cur((size_t)P * W * H, 7);
                dsdtm_stream_desc s{};
                s.n_pairs = P; s.max_features = N; s.levels = L; s.width = W; s.height = H; s.row_stride = W; s.image_pitch = (size_t)W * H;
                s.ref_image = ref.data(); s.cur_image = chained ? nullptr : cur.data();
                s.px_xy = hb.b.px_xy; s.bearing = hb.b.bearing; s.p_world = hb.b.p_world; s.initial = hb.b.initial;
                s.T_ref_w = hb.b.T_ref_w; s.T_cur_w = hb.b.T_cur_w; s.n_tracked = hb.b.n_tracked; s.stats = hb.b.stats;
                if (fail_at) fake_hip_fail("hipMemcpyAsync", fail_at);
                const int rc = dsdtm_sparse_align_batch_streamed(ctx, 3, &s, 2, &cam, &prm);
                CHECK(fail_at ? (rc == DSDTM_ERR_HIP || rc == DSDTM_OK) : rc == DSDTM_OK);
                CHECK(fake_hip_pending() == 0);
                if (!fail_at) for (int i = 0; i < P; ++i) CHECK(hb.b.T_cur_w[12 * i + 3] > 0.0 && hb.b.n_tracked[i] == N);
                (void)g;
            }
            fake_hip_drain_all();
            for (auto& c : ctx) dsdtm_destroy(c);
            fake_hip_reset();
        }
    return true;
}

// ---- 8. frames: the buffer pool, destruction after the context, a context pointer that is already gone ----
static bool frame_lifetime() {
    dsdtm_ctx* ctx = nullptr;
    CHECK(dsdtm_create(0, &ctx) == DSDTM_OK);
    std::vector<uint8_t> img((size_t)W * H, 3);
    const long mallocs0 = fake_hip_calls("hipMalloc");
    for (int i = 0; i < 20; ++i) {                    // a tracker's frames: one created, one destroyed per image
        dsdtm_frame* f = nullptr;
        CHECK(dsdtm_frame_create_from_image(ctx, img.data(), W, H, W, L, &f) == DSDTM_OK);
        dsdtm_frame_destroy(ctx, f);
    }
    CHECK(fake_hip_calls("hipMalloc") - mallocs0 <= 3);   // the pyramid buffer is recycled (+ staging): not 20 allocations
    dsdtm_frame *a = nullptr, *b = nullptr;
    CHECK(dsdtm_frame_create_from_image(ctx, img.data(), W, H, W, L, &a) == DSDTM_OK);
    CHECK(dsdtm_frame_create_from_image(ctx, img.data(), W, H, W, L, &b) == DSDTM_OK);
    dsdtm_destroy(ctx);
    dsdtm_frame_destroy(nullptr, a);                  // after its context, without one
    dsdtm_frame_destroy(ctx, b);                      // after its context, WITH the dangling pointer: must not be dereferenced
    CHECK(fake_hip_live_allocations() == 0);          // everything went back (frames, pool, staging, counters, flags)
    return true;
}

// ---- 9. the single-call entries: every byte they pack into / unpack from the staging blocks is inside them ----
static bool single_call_entries() {
    dsdtm_ctx* ctx = nullptr;
    CHECK(dsdtm_create(0, &ctx) == DSDTM_OK);
    const Geometry g = geometry();
    const dsdtm_camera cam = camera(); const dsdtm_align_params prm = params();
    std::vector<std::vector<uint8_t>> lv(L);
    dsdtm_pyramid pyr{};
    pyr.levels = L;
    for (int l = 0; l < L; ++l) { lv[l].assign((size_t)(g.w[l] + 5) * g.h[l], 1); pyr.data[l] = lv[l].data(); pyr.width[l] = g.w[l]; pyr.height[l] = g.h[l]; pyr.stride[l] = g.w[l] + 5; }
    for (int n : {0, 7, 15, 100, 600, 1000, 2000}) {
        std::vector<float> px(2 * (size_t)n + 1);
        std::vector<double> be(3 * (size_t)n + 1), pw(3 * (size_t)n + 1);
        std::vector<uint8_t> ini((size_t)n + 1, 1);
        double Tr[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0}, Tc[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
        int nt = -5;
        dsdtm_align_stats st;
        CHECK(dsdtm_sparse_align(ctx, &pyr, &pyr, &cam, px.data(), be.data(), pw.data(), ini.data(), n, Tr, Tc, &prm, &nt, &st) == DSDTM_OK);
        CHECK(nt == (n >= 15 ? n : 0));
        dsdtm_frame *fr = nullptr, *fc = nullptr;
        CHECK(dsdtm_frame_create(ctx, &pyr, &fr) == DSDTM_OK && dsdtm_frame_create_from_image(ctx, lv[0].data(), W, H, W + 5, L, &fc) == DSDTM_OK);
        if (n == 600) fake_hip_timeout_next(1, 0);    // a team of one pair that times out: re-run before the entry returns
        CHECK(dsdtm_sparse_align_frames(ctx, fr, fc, &cam, px.data(), be.data(), pw.data(), ini.data(), n, Tr, Tc, &prm, &nt, nullptr) == DSDTM_OK);
        CHECK(nt == (n >= 15 ? n : 0) && (n < 15 || Tc[3] > 0.0));
        // FindMatchDirect / Align2D / warp / pose refinement / detector / pyrDown on the same sizes
        const int m = n;
        std::vector<int32_t> ck((size_t)m + 1, 0), rl((size_t)m + 1, 0), sl((size_t)m + 1), lvl((size_t)m + 1, 0);
        std::vector<double> pxy(2 * (size_t)m + 1, 10.0), aff(4 * (size_t)m + 1), rn((size_t)m + 1);
        std::vector<uint8_t> conv((size_t)m + 1), pb(100 * (size_t)m + 1), pp(64 * (size_t)m + 1), use((size_t)m + 1, 1);
        const dsdtm_frame* kf[1] = {fr};
        CHECK(dsdtm_match_candidates_frames(ctx, fc, kf, 1, &cam, Tr, Tc, ck.data(), px.data(), rl.data(), be.data(), pw.data(), 1, 10, m, pxy.data(), sl.data(), conv.data()) == DSDTM_OK);
        CHECK(dsdtm_warp_patches(ctx, &pyr, 1, &cam, Tr, Tc, ck.data(), px.data(), rl.data(), be.data(), pw.data(), 1, m, aff.data(), sl.data(), pb.data(), pp.data()) == DSDTM_OK);
        CHECK(dsdtm_align2d_batch(ctx, &pyr, pb.data(), pp.data(), lvl.data(), pxy.data(), conv.data(), 10, m) == DSDTM_OK);
        dsdtm_pose_opt_params pop{100, 0};
        dsdtm_pose_opt_summary sm;
        CHECK(dsdtm_pose_optimization(ctx, be.data(), pw.data(), lvl.data(), use.data(), m, Tc, &pop, rn.data(), &sm) == DSDTM_OK);
        dsdtm_frame_destroy(ctx, fr); dsdtm_frame_destroy(ctx, fc);
    }
    {
        const dsdtm_detect_params dp{8, (W + 7) / 8, (H + 7) / 8, L, 20, 5.0f};
        const size_t G = (size_t)dp.grid_cols * dp.grid_rows;
        std::vector<float> sc(G);
        std::vector<int32_t> cx(G), cy(G), cl(G);
        CHECK(dsdtm_detect_cells(ctx, &pyr, nullptr, &dp, sc.data(), cx.data(), cy.data(), cl.data()) == DSDTM_OK);
        std::vector<std::vector<uint8_t>> out(L);
        uint8_t* outs[8] = {nullptr};
        int strides[8] = {0};
        for (int l = 1; l < L; ++l) { out[l].resize((size_t)(g.w[l] + 3) * g.h[l]); outs[l] = out[l].data(); strides[l] = g.w[l] + 3; }
        CHECK(dsdtm_pyrdown(ctx, lv[0].data(), W, H, W + 5, L, outs, strides) == DSDTM_OK);
    }
    CHECK(no_fake_errors());
    dsdtm_destroy(ctx);
    CHECK(fake_hip_live_allocations() == 0);
    return true;
}

// ---- 10. dsdtm_track_frame: the pinned block it packs (image, features, the flattened local map, results) ----
static bool track_frame_packing() {
    dsdtm_ctx* ctx = nullptr;
    CHECK(dsdtm_create(0, &ctx) == DSDTM_OK);
    const dsdtm_camera cam = camera();
    std::vector<uint8_t> img((size_t)(W + 3) * H, 5), mask((size_t)(W + 1) * H, 255);
    dsdtm_frame *ref = nullptr, *k0 = nullptr, *k1 = nullptr;
    CHECK(dsdtm_frame_create_from_image(ctx, img.data(), W, H, W + 3, L, &ref) == DSDTM_OK);
    CHECK(dsdtm_frame_create_from_image(ctx, img.data(), W, H, W + 3, L, &k0) == DSDTM_OK);
    CHECK(dsdtm_frame_create_from_image(ctx, img.data(), W, H, W + 3, L, &k1) == DSDTM_OK);
    const dsdtm_frame* kf[2] = {k0, k1};
    for (int n : {0, 40, 600})
        for (int M : {0, 1, 333, 4096})
            for (int with_mask = 0; with_mask < 2; ++with_mask) {
                std::vector<float> px(2 * (size_t)n + 1), opx;
                std::vector<double> be(3 * (size_t)n + 1), pw(3 * (size_t)n + 1), mpw(3 * (size_t)M + 1), ob, Tk(24, 0.0);
                std::vector<uint8_t> ini((size_t)n + 1, 1), bad((size_t)M + 1, 0);
                std::vector<int32_t> found((size_t)M + 1, 2), off((size_t)M + 1, 0), okf, olv;
                for (int i = 0; i < M; ++i) { for (int o = 0; o < i % 3; ++o) { okf.push_back(o % 2); olv.push_back(o % L); opx.push_back(1.f); opx.push_back(2.f); ob.insert(ob.end(), {0., 0., 1.}); } off[(size_t)i + 1] = (int32_t)okf.size(); }
                double T[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
                dsdtm_track_desc d{};
                d.image = img.data(); d.width = W; d.height = H; d.stride = W + 3; d.levels = L;
                d.ref = ref; d.n_ref_features = n;
                if (n) { d.ref_px_xy = px.data(); d.ref_bearing = be.data(); d.ref_p_world = pw.data(); d.ref_initial = ini.data(); }   // (none: NULL columns)
                d.T_ref_w = T; d.T_seed = T; d.align = params(); d.min_tracked = 0;
                d.kf = kf; d.n_kf = 2; d.T_kf_w = Tk.data(); d.n_points = M;
                if (M) { d.mp_world = mpw.data(); d.mp_found = found.data(); d.mp_bad = bad.data(); d.obs_offset = off.data(); }                // (no points: NULL columns)
                if (!okf.empty()) { d.obs_kf = okf.data(); d.obs_px = opx.data(); d.obs_level = olv.data(); d.obs_bearing = ob.data(); }
                if (with_mask) { d.mask = mask.data(); d.mask_stride = W + 1; }
                d.cell_size = 8; d.max_pyr_levels = L + 1; d.max_matches = 200; d.align2d_iters = 10; d.pose_opt.max_iterations = 100;
                dsdtm_track_result r;
                std::vector<dsdtm_track_match> ms(200);
                std::vector<double> rn(200);
                if (n == 600) fake_hip_timeout_next(1, 0);        // Run as a team that times out: the whole chain is re-issued on one CU
                const int rc = dsdtm_track_frame(ctx, &cam, &d, &r, ms.data(), rn.data());
                if (rc != DSDTM_OK) std::fprintf(stderr, "track n=%d M=%d mask=%d: %s\n", n, M, with_mask, dsdtm_last_error(ctx));
                CHECK(rc == DSDTM_OK);
                CHECK(r.frame != nullptr && r.n_tracked == (n >= 15 ? n : 0) && r.n_matches == 0 && fake_hip_pending() == 0);
                dsdtm_frame_destroy(ctx, r.frame);
            }
    dsdtm_frame_destroy(ctx, ref); dsdtm_frame_destroy(ctx, k0); dsdtm_frame_destroy(ctx, k1);
    CHECK(no_fake_errors());
    dsdtm_destroy(ctx);
    CHECK(fake_hip_live_allocations() == 0);
    return true;
}

// ---- 10b. dsdtm_track_frame when a launch or a copy inside it fails: an error, nothing pending, nothing leaked, and the next frame works ----
static bool track_frame_failures() {
    dsdtm_ctx* ctx = nullptr;
    CHECK(dsdtm_create(0, &ctx) == DSDTM_OK);
    const dsdtm_camera cam = camera();
    std::vector<uint8_t> img((size_t)W * H, 5);
    dsdtm_frame *ref = nullptr, *k0 = nullptr;
    CHECK(dsdtm_frame_create_from_image(ctx, img.data(), W, H, W, L, &ref) == DSDTM_OK);
    CHECK(dsdtm_frame_create_from_image(ctx, img.data(), W, H, W, L, &k0) == DSDTM_OK);
    const dsdtm_frame* kf[1] = {k0};
    const int n = 40, M = 50;
    std::vector<float> px(2 * n), opx(2 * M, 1.f);
    std::vector<double> be(3 * n), pw(3 * n), mpw(3 * M), ob(3 * M, 0.0), Tk(12, 0.0);
    std::vector<uint8_t> ini(n, 1), bad(M, 0);
    std::vector<int32_t> found(M, 2), off(M + 1), okf(M, 0), olv(M, 0);
    for (int i = 0; i <= M; ++i) off[i] = i;
    double T[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    dsdtm_track_desc d{};
    d.image = img.data(); d.width = W; d.height = H; d.stride = W; d.levels = L;
    d.ref = ref; d.n_ref_features = n; d.ref_px_xy = px.data(); d.ref_bearing = be.data(); d.ref_p_world = pw.data(); d.ref_initial = ini.data();
    d.T_ref_w = T; d.T_seed = T; d.align = params(); d.min_tracked = 0;
    d.kf = kf; d.n_kf = 1; d.T_kf_w = Tk.data(); d.n_points = M;
    d.mp_world = mpw.data(); d.mp_found = found.data(); d.mp_bad = bad.data(); d.obs_offset = off.data();
    d.obs_kf = okf.data(); d.obs_px = opx.data(); d.obs_level = olv.data(); d.obs_bearing = ob.data();
    d.cell_size = 8; d.max_pyr_levels = L + 1; d.max_matches = 200; d.align2d_iters = 10; d.pose_opt.max_iterations = 100;
    dsdtm_track_result r;
    std::vector<dsdtm_track_match> ms(200);
    std::vector<double> rn(200);
    CHECK(dsdtm_track_frame(ctx, &cam, &d, &r, ms.data(), rn.data()) == DSDTM_OK && r.frame && r.n_tracked == n);
    dsdtm_frame_destroy(ctx, r.frame);
    const char* points[] = {"ingest_launch", "pyrdown_launch", "sparse_align_launch", "hipMemcpyAsync", "hipStreamSynchronize",
                            "track_match_launch", "track_replay_launch", "pose_opt_launch"};
    for (const char* api : points) {
        fake_hip_fail(api, 1);
        const int rc = dsdtm_track_frame(ctx, &cam, &d, &r, ms.data(), rn.data());
        CHECK(rc == DSDTM_ERR_HIP && r.frame == nullptr && fake_hip_pending() == 0);
        CHECK(dsdtm_track_frame(ctx, &cam, &d, &r, ms.data(), rn.data()) == DSDTM_OK && r.frame && r.n_tracked == n);   // and the context is usable
        dsdtm_frame_destroy(ctx, r.frame);
    }
    {   // an image that lives in device memory: read from there (aligned, contiguous) or copied device to device (strided)
        uint8_t* dimg = nullptr;
        CHECK(hipMalloc((void**)&dimg, (size_t)(W + 16) * H) == hipSuccess);
        memset(dimg, 5, (size_t)(W + 16) * H);
        dsdtm_track_desc dd = d;
        dd.image = dimg;
        CHECK(dsdtm_track_frame(ctx, &cam, &dd, &r, ms.data(), rn.data()) == DSDTM_OK && r.frame && r.n_tracked == n);
        dsdtm_frame_destroy(ctx, r.frame);
        dd.stride = W + 16;
        CHECK(dsdtm_track_frame(ctx, &cam, &dd, &r, ms.data(), rn.data()) == DSDTM_OK && r.frame && r.n_tracked == n);
        dsdtm_frame_destroy(ctx, r.frame);
        CHECK(hipFree(dimg) == hipSuccess);
    }
    dsdtm_frame_destroy(ctx, ref); dsdtm_frame_destroy(ctx, k0);
    CHECK(no_fake_errors());
    dsdtm_destroy(ctx);
    CHECK(fake_hip_live_allocations() == 0);
    return true;
}

// ---- 11. two contexts driven from two threads at once (the ThreadSanitizer build) ----
static bool two_contexts_two_threads() {
    bool ok[2] = {false, false};
    auto work = [&](int t) {
        dsdtm_ctx* ctx = nullptr;
        if (dsdtm_create(0, &ctx) != DSDTM_OK) return;
        const dsdtm_camera cam = camera(); const dsdtm_align_params prm = params();
        hipStream_t st;
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return;
        bool good = true;
        for (int i = 0; i < 30 && good; ++i) {
            Batch b(3, i % 2 ? 600 : 100, true);
            good = dsdtm_sparse_align_batch_device(ctx, &b.b, &cam, &prm, st) == DSDTM_OK && dsdtm_sparse_align_check(ctx, st) == DSDTM_OK && b.b.T_cur_w[3] > 0.0;
            std::vector<uint8_t> img((size_t)W * H, 1);
            dsdtm_frame* f = nullptr;
            good = good && dsdtm_frame_create_from_image(ctx, img.data(), W, H, W, L, &f) == DSDTM_OK;
            dsdtm_frame_destroy(ctx, f);
        }
        {
            Batch hb(5, 100, false);
            dsdtm_ctx* one[1] = {ctx};
            good = good && dsdtm_sparse_align_batch_sharded(one, 1, &hb.b, &cam, &prm) == DSDTM_OK;
        }
        dsdtm_destroy(ctx);
        (void)hipStreamDestroy(st);
        ok[t] = good;
    };
    std::thread a(work, 0), b(work, 1);
    a.join(); b.join();
    CHECK(ok[0] && ok[1]);
    return true;
}

int main(int argc, char** argv) {
    const std::vector<std::pair<std::string, std::function<bool()>>> all = {
        {"ring_table_past_16_streams", ring_table_past_16_streams}, {"evicted_timeout_is_reported", evicted_timeout_is_reported},
        {"recover_slots_past_64", recover_slots_past_64}, {"team_epoch_wrap_is_cleared", team_epoch_wrap_is_cleared},
        {"team_epoch_wrap_hazard_is_real", team_epoch_wrap_hazard_is_real}, {"sharded_refetches_after_rerun", sharded_refetches_after_rerun},
        {"sharded_error_midway", sharded_error_midway}, {"streamed_entry", streamed_entry}, {"frame_lifetime", frame_lifetime},
        {"single_call_entries", single_call_entries}, {"track_frame_packing", track_frame_packing}, {"track_frame_failures", track_frame_failures},
        {"two_contexts_two_threads", two_contexts_two_threads}};
    int failed = 0;
    for (const auto& sc : all) {
        bool wanted = argc < 2;
        for (int i = 1; i < argc; ++i) wanted = wanted || sc.first == argv[i];
        if (!wanted) continue;
        fake_hip_reset();
        const bool ok = sc.second();
        std::printf("%s %s\n", ok ? "ok" : "FAILED", sc.first.c_str());
        std::fflush(stdout);
        failed += ok ? 0 : 1;
    }
    return failed ? 1 : 0;
}
