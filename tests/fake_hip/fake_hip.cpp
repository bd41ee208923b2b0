// tests/fake_hip/fake_hip.cpp — the fake HIP runtime + fake kernel launchers behind tests/test_fake_hip_cpu.py.
// TEST INFRASTRUCTURE ONLY (see hip/hip_runtime.h beside this file). What it models:
//   * "device" and pinned memory live in the host heap (so AddressSanitizer sees every access the host side makes or causes);
//   * every asynchronous operation — copies, memsets, event records, waits, kernel launches — is QUEUED on its stream and runs
//     only when something waits for it (stream / event / device synchronise, hipFree, hipMemset): a copy whose buffer died
//     too early, or results read before the wait, are caught instead of passing by luck;
//   * a kernel "launch" of the library (kernels.h) is a queued operation that reads the first and last byte of every input
//     it names, writes its outputs with a marker of the launch, checks the invariants the real kernels rely on (pair counter
//     zero at start, no stale tag of the same epoch in a team slot) and raises its timeout word when the test asked for it;
//   * any HIP entry can be told to fail at its n-th call.
#include "fake_hip.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "kernels.h"

struct fake_event { fake_stream* s = nullptr; unsigned long long ticket = 0; bool alive = true; };
struct FakeOp { std::function<void()> run; };
struct fake_stream {
    std::deque<FakeOp> q;
    unsigned long long issued = 0, done = 0;
    bool alive = true, draining = false;
    int id = 0;
};

namespace {
std::recursive_mutex g_mu;
typedef std::lock_guard<std::recursive_mutex> Lock;
// (registries that are never destroyed: the objects stay reachable for LeakSanitizer, which then reports the LIBRARY's leaks only)
std::vector<fake_stream*>& g_streams = *new std::vector<fake_stream*>();
std::vector<fake_event*>& g_events = *new std::vector<fake_event*>();          // kept for the life of the process: a use after destruction is REPORTED, and LeakSanitizer sees them as reachable
fake_stream g_null_stream;
std::map<void*, std::pair<size_t, int>> g_mem;      // pointer -> (bytes, hipMemoryType)
std::map<std::string, long> g_calls, g_fail_at;
std::vector<fake_launch> g_log;
std::vector<std::string> g_errors;
long g_timeout_multi = 0, g_timeout_one = 0, g_kernel_seq = 0;
int g_next_stream_id = 1;

fake_stream* S(hipStream_t s) { return s ? s : &g_null_stream; }

void fatal(const char* msg) { std::fprintf(stderr, "fake_hip: %s\n", msg); std::abort(); }

bool should_fail(const char* name) {
    const long n = ++g_calls[name];
    auto it = g_fail_at.find(name);
    if (it != g_fail_at.end() && it->second == n) { g_fail_at.erase(it); return true; }
    return false;
}
#define FAIL_POINT(name) do { Lock l_(g_mu); if (should_fail(name)) return hipErrorUnknown; } while (0)

void drain(fake_stream* st, unsigned long long upto) {
    if (st->draining) return;                       // (a wait on an event of the stream being drained: already satisfied in order)
    st->draining = true;
    while (st->done < upto && !st->q.empty()) {
        FakeOp op = std::move(st->q.front());
        st->q.pop_front();
        op.run();
        st->done++;
    }
    st->draining = false;
}
void drain_all_locked() {
    drain(&g_null_stream, ~0ull);
    for (fake_stream* s : g_streams) if (s->alive) drain(s, ~0ull);
}
void enqueue(hipStream_t s, std::function<void()> f) {
    fake_stream* st = S(s);
    if (!st->alive) fatal("operation enqueued on a destroyed stream");
    st->q.push_back(FakeOp{std::move(f)});
    st->issued++;
}
// reads one byte (ASan checks it): the first and the last byte of a range the kernel would read
void touch(const void* p, size_t bytes) {
    if (!p || !bytes) return;
    volatile const uint8_t* b = (const uint8_t*)p;
    (void)b[0]; (void)b[bytes - 1];
}
void touch_w(void* p, size_t bytes) {
    if (!p || !bytes) return;
    volatile uint8_t* b = (uint8_t*)p;
    b[0] = b[0]; b[bytes - 1] = b[bytes - 1];
}
long take(long& counter) { if (counter > 0) { counter--; return 1; } return 0; }
}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// control interface (fake_hip.h)
// ---------------------------------------------------------------------------------------------------------------------
void fake_hip_reset() {
    Lock l(g_mu);
    drain_all_locked();
    g_calls.clear(); g_fail_at.clear(); g_log.clear(); g_errors.clear();
    g_timeout_multi = g_timeout_one = 0;
}
void fake_hip_fail(const char* api, long nth_call_from_now) { Lock l(g_mu); g_fail_at[api] = g_calls[api] + nth_call_from_now; }
void fake_hip_timeout_next(long multi_cu, long one_cu) { Lock l(g_mu); g_timeout_multi = multi_cu; g_timeout_one = one_cu; }
void fake_hip_drain_all() { Lock l(g_mu); drain_all_locked(); }
long fake_hip_calls(const char* api) { Lock l(g_mu); return g_calls[api]; }
std::vector<fake_launch> fake_hip_log() { Lock l(g_mu); return g_log; }
std::vector<std::string> fake_hip_errors() { Lock l(g_mu); return g_errors; }
size_t fake_hip_live_allocations() { Lock l(g_mu); return g_mem.size(); }
size_t fake_hip_pending() {
    Lock l(g_mu);
    size_t n = g_null_stream.q.size();
    for (fake_stream* s : g_streams) n += s->q.size();
    return n;
}

// ---------------------------------------------------------------------------------------------------------------------
// the runtime
// ---------------------------------------------------------------------------------------------------------------------
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : (e == hipErrorUnknown ? "hipErrorUnknown (injected)" : "hipError"); }
hipError_t hipGetLastError() { return hipSuccess; }
hipError_t hipGetDeviceCount(int* n) { FAIL_POINT("hipGetDeviceCount"); *n = 2; return hipSuccess; }
hipError_t hipSetDevice(int) { FAIL_POINT("hipSetDevice"); return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { memset(p, 0, sizeof *p); strcpy(p->gcnArchName, "gfx950:sramecc+:xnack-"); p->multiProcessorCount = 256; return hipSuccess; }
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 8; return hipSuccess; }
hipError_t hipDeviceSynchronize() { FAIL_POINT("hipDeviceSynchronize"); Lock l(g_mu); drain_all_locked(); return hipSuccess; }

static hipError_t alloc(void** p, size_t bytes, int type, const char* name) {
    FAIL_POINT(name);
    void* m = malloc(bytes ? bytes : 1);
    if (!m) return hipErrorOutOfMemory;
    memset(m, 0xA5, bytes ? bytes : 1);             // uninitialised "device" memory is not zero
    Lock l(g_mu);
    g_mem[m] = {bytes, type};
    *p = m;
    return hipSuccess;
}
hipError_t hipMalloc(void** p, size_t bytes) { return alloc(p, bytes, hipMemoryTypeDevice, "hipMalloc"); }
hipError_t hipHostMalloc(void** p, size_t bytes, unsigned) { return alloc(p, bytes, hipMemoryTypeHost, "hipHostMalloc"); }
static hipError_t release(void* p) {
    if (!p) return hipSuccess;
    Lock l(g_mu);
    drain_all_locked();                             // hipFree / hipHostFree wait for the device
    if (!g_mem.erase(p)) fatal("free of a pointer this runtime did not allocate (or a double free)");
    free(p);
    return hipSuccess;
}
hipError_t hipFree(void* p) { return release(p); }
hipError_t hipHostFree(void* p) { return release(p); }
hipError_t hipHostGetDevicePointer(void** dev, void* host, unsigned) { FAIL_POINT("hipHostGetDevicePointer"); *dev = host; return hipSuccess; }
hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void* p) {
    Lock l(g_mu);
    for (auto& kv : g_mem)
        if ((const uint8_t*)p >= (const uint8_t*)kv.first && (const uint8_t*)p < (const uint8_t*)kv.first + kv.second.first) { a->type = (hipMemoryType)kv.second.second; a->device = 0; return hipSuccess; }
    return hipErrorInvalidValue;
}
hipError_t hipMemset(void* p, int v, size_t bytes) { FAIL_POINT("hipMemset"); Lock l(g_mu); drain_all_locked(); memset(p, v, bytes); return hipSuccess; }
hipError_t hipMemsetAsync(void* p, int v, size_t bytes, hipStream_t s) {
    FAIL_POINT("hipMemsetAsync");
    Lock l(g_mu);
    enqueue(s, [=]() { memset(p, v, bytes); });
    return hipSuccess;
}
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind, hipStream_t s) {
    FAIL_POINT("hipMemcpyAsync");
    Lock l(g_mu);
    enqueue(s, [=]() { memmove(dst, src, bytes); });
    return hipSuccess;
}
hipError_t hipMemcpy2DAsync(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, hipMemcpyKind, hipStream_t s) {
    FAIL_POINT("hipMemcpy2DAsync");
    Lock l(g_mu);
    enqueue(s, [=]() { for (size_t y = 0; y < height; ++y) memmove((uint8_t*)dst + y * dpitch, (const uint8_t*)src + y * spitch, width); });
    return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) {
    FAIL_POINT("hipStreamCreateWithFlags");
    Lock l(g_mu);
    fake_stream* st = new fake_stream();
    st->id = g_next_stream_id++;
    g_streams.push_back(st);
    *s = st;
    return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t s) {
    Lock l(g_mu);
    if (!s || !s->alive) fatal("hipStreamDestroy of a dead stream");
    drain(s, ~0ull);
    s->alive = false;                               // (kept allocated: a later use is reported, not a wild access)
    return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t s) {
    FAIL_POINT("hipStreamSynchronize");
    Lock l(g_mu);
    fake_stream* st = S(s);
    if (!st->alive) fatal("hipStreamSynchronize of a destroyed stream");
    drain(st, ~0ull);
    return hipSuccess;
}
hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus* st) { *st = hipStreamCaptureStatusNone; return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { FAIL_POINT("hipEventCreateWithFlags"); Lock l(g_mu); *e = new fake_event(); g_events.push_back(*e); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { Lock l(g_mu); if (!e || !e->alive) fatal("hipEventDestroy of a dead event"); e->alive = false; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
    FAIL_POINT("hipEventRecord");
    Lock l(g_mu);
    if (!e->alive) fatal("hipEventRecord on a destroyed event");
    enqueue(s, []() {});
    e->s = S(s); e->ticket = S(s)->issued;
    return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t e) {
    FAIL_POINT("hipEventSynchronize");
    Lock l(g_mu);
    if (!e->alive) fatal("hipEventSynchronize on a destroyed event");
    if (e->s) drain(e->s, e->ticket);               // (an event outlives its stream: the stream object is kept)
    return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned) {
    FAIL_POINT("hipStreamWaitEvent");
    Lock l(g_mu);
    fake_stream* es = e->s;
    const unsigned long long ticket = e->ticket;
    enqueue(s, [=]() { if (es) drain(es, ticket); });
    return hipSuccess;
}

// ---------------------------------------------------------------------------------------------------------------------
// the library's kernel launchers (kernels.h), faked
// ---------------------------------------------------------------------------------------------------------------------
namespace dsdtm {

static void sa_kernel(const SAKernelArgs a, int kind, int team_k, hipStream_t stream) {
    Lock l(g_mu);
    const bool multi = kind != FAKE_SA_ONE_CU;
    enqueue(stream, [=]() {
        const long seq = ++g_kernel_seq;
        const bool to = multi ? take(g_timeout_multi) : take(g_timeout_one);
        char msg[256];
        if (*a.pair_counter != 0u && kind != FAKE_SA_TEAM) {       // (the team kernel has no pair counter)
            snprintf(msg, sizeof msg, "launch %ld: the pair counter word is %u at the start of a launch", seq, *a.pair_counter);
            g_errors.push_back(msg);
        }
        size_t pyr_end = 0;
        for (int lv = 0; lv < DSDTM_MAX_LEVELS; ++lv)
            if (a.lv[lv].w > 0) pyr_end = std::max(pyr_end, (size_t)a.lv[lv].off + (size_t)a.lv[lv].stride * a.lv[lv].h);
        const size_t P = (size_t)a.n_pairs, N = (size_t)a.max_features;
        touch(a.ref_pyr, (P - 1) * a.pyr_pitch + pyr_end); touch(a.cur_pyr, (P - 1) * a.pyr_pitch + pyr_end);
        touch(a.px_xy, P * N * 8); touch(a.bearing, P * N * 24); touch(a.p_world, P * N * 24); touch(a.initial, P * N);
        touch(a.n_features, a.n_features ? P * 4 : 0); touch(a.T_ref_w, P * 96);
        touch_w(a.T_cur_w, P * 96); touch_w(a.n_tracked, P * 4); touch_w(a.stats, a.stats ? P * sizeof(dsdtm_align_stats) : 0);
        if (kind == FAKE_SA_TEAM) {
            // the exchange words of this launch's ring slot: a word that already carries this launch's 20-bit epoch is a partial
            // the real kernel would ACCEPT from a launch 2^20 launches ago
            uint32_t* w = (uint32_t*)a.workspace;
            const size_t words = sparse_align_team_bytes(a.n_pairs) / 4;
            const uint32_t tag = a.team_epoch & 0xfffffu;
            for (size_t i = 0; i < words; ++i)
                if (w[i] == tag && tag != 0u) { snprintf(msg, sizeof msg, "team launch %ld (epoch %u): a stale exchange word with this launch's tag", seq, a.team_epoch); g_errors.push_back(msg); break; }
            for (size_t i = 0; i < words; ++i) w[i] = tag;
        } else if (a.workspace) touch_w(a.workspace, 8);
        for (size_t i = 0; i < P; ++i) {
            const int nf = a.n_features ? a.n_features[i] : a.max_features;
            if (nf < a.min_fts) { a.n_tracked[i] = 0; continue; }          // Run: too few features, pose untouched
            a.T_cur_w[12 * i + 3] = to ? -1.0 : (double)seq;               // marker: which launch produced this pose (-1: aborted)
            a.n_tracked[i] = to ? -1 : nf;
            if (a.stats) { memset(&a.stats[i], 0, sizeof(dsdtm_align_stats)); a.stats[i].iters[0] = (int32_t)seq; }
        }
        if (to) { *a.timeout_flag = 1u; if (kind != FAKE_SA_TEAM) *a.pair_counter = 7u; }   // a stopped pair may leave its counter behind
        fake_launch fl;
        fl.kind = kind; fl.stream = stream; fl.T_cur_w = a.T_cur_w; fl.seq = seq; fl.team_k = team_k; fl.epoch = a.team_epoch; fl.timed_out = to;
        fl.n_pairs = a.n_pairs; fl.max_features = a.max_features;
        g_log.push_back(fl);
    });
}

SAVariant sparse_align_pick_variant(int n) {
    if (n > 704) return SA_WS;
    if (n <= 128) return SA_REG128;
    if (n <= 192) return SA_REG192;
    if (n <= 256) return SA_REG256;
    if (n <= 320) return SA_REG320;
    if (n <= 448) return SA_REG448;
    return SA_REG704;
}
size_t sparse_align_workspace_bytes(int n_pairs, int n) {
    if (sparse_align_pick_variant(n) != SA_WS || (n + 63) / 64 * 64 <= 1024) return 0;
    return (size_t)n_pairs * (size_t)((n + 63) / 64 * 64) * 88;
}
bool sparse_align_uses_duo(int n, bool have_ws) { const int npad = (n + 63) / 64 * 64; return sparse_align_pick_variant(n) == SA_WS && npad > 1024 && npad <= 2048 && have_ws && npad >= 1792; }
int sparse_align_team_size(int n_pairs, int n, int num_cus) {
    if (n < 449 || n_pairs <= 0) return 0;
    const int k = (n + 255) / 256;
    if (k > 64 || n_pairs * k > num_cus / 2) return 0;
    return k;
}
size_t sparse_align_team_bytes(int n_pairs) { return (size_t)n_pairs * 64 * 32 * 8; }
hipError_t sparse_align_launch(const SAKernelArgs& a, SAVariant, int, hipStream_t stream, bool allow_multi_cu) {
    FAIL_POINT("sparse_align_launch");
    if (a.n_pairs <= 0) return hipSuccess;
    const bool duo = allow_multi_cu && sparse_align_uses_duo(a.max_features, a.workspace != nullptr);
    sa_kernel(a, duo ? FAKE_SA_DUO : FAKE_SA_ONE_CU, 0, stream);
    return hipSuccess;
}
hipError_t sparse_align_launch_team(const SAKernelArgs& a, int k, hipStream_t stream, int) {
    FAIL_POINT("sparse_align_launch_team");
    if (a.n_pairs <= 0) return hipSuccess;
    sa_kernel(a, FAKE_SA_TEAM, k, stream);
    return hipSuccess;
}
#ifdef DSDTM_DIAG
int sparse_align_occupancy(int) { return 1; }
hipError_t sparse_align_launch_stamps(const SAKernelArgs& a, int, hipStream_t stream) { sa_kernel(a, FAKE_SA_ONE_CU, 0, stream); return hipSuccess; }
hipError_t selftest_launch(const double* in, double* out, int n, hipStream_t stream) {
    Lock l(g_mu);
    enqueue(stream, [=]() { touch(in, (size_t)n * 33 * 8); memset(out, 0, (size_t)n * 120 * 8); });
    return hipSuccess;
}
#endif

static void generic(const char* name, hipStream_t stream, std::function<void()> body) {
    Lock l(g_mu);
    std::string nm = name;
    enqueue(stream, [=]() {
        body();
        fake_launch fl;
        fl.kind = FAKE_OTHER; fl.stream = stream; fl.seq = ++g_kernel_seq; fl.name = nm;
        g_log.push_back(fl);
    });
}
static size_t pyr_bytes(const LevelGeom* lv) {
    size_t e = 0;
    for (int l = 0; l < DSDTM_MAX_LEVELS; ++l) if (lv[l].w > 0) e = std::max(e, (size_t)lv[l].off + (size_t)lv[l].stride * lv[l].h);
    return e;
}
hipError_t detect_launch(const DetectArgs& a, hipStream_t stream) {
    FAIL_POINT("detect_launch");
    generic("detect", stream, [=]() {
        const size_t F = (size_t)a.n_frames, G = (size_t)a.grid_cols * a.grid_rows, pe = pyr_bytes(a.lv);
        touch(a.pyr, (F - 1) * a.pyr_pitch + pe); touch_w(a.score, (F - 1) * a.pyr_pitch + pe); touch(a.occupied, a.occupied ? F * G : 0);
        for (size_t i = 0; i < F * G; ++i) a.cell_key[i] = 0ull;
        if (a.keep) touch_w(a.keep, pe);
        if (a.cell_score) for (size_t i = 0; i < F * G; ++i) { a.cell_score[i] = a.detection_threshold; a.cell_x[i] = a.cell_y[i] = a.cell_level[i] = 0; }
    });
    return hipSuccess;
}
hipError_t align2d_launch(const A2DKernelArgs& a, hipStream_t stream) {
    FAIL_POINT("align2d_launch");
    generic("align2d", stream, [=]() {
        const size_t M = (size_t)a.m;
        touch(a.cur_pyr, pyr_bytes(a.lv)); touch(a.patch_border, M * 100); touch(a.patch, M * 64); touch(a.level, M * 4);
        touch_w(a.px_xy, M * 16);
        for (size_t i = 0; i < M; ++i) a.converged[i] = 1;
    });
    return hipSuccess;
}
hipError_t pyrdown_launch(uint8_t* pyr, size_t pitch, int n, int sw, int sh, int sstride, size_t soff, int dstride, size_t doff, hipStream_t stream) {
    FAIL_POINT("pyrdown_launch");
    generic("pyrdown", stream, [=]() {
        for (int i = 0; i < n; ++i) {
            touch(pyr + (size_t)i * pitch + soff, (size_t)sstride * sh - (size_t)(sstride - sw));
            memset(pyr + (size_t)i * pitch + doff, 0x11, (size_t)dstride * ((sh + 1) / 2) - (size_t)(dstride - (sw + 1) / 2));
        }
    });
    return hipSuccess;
}
hipError_t ingest_launch(const void* src, void* dst, size_t bytes, const void* src2, void* dst2, size_t bytes2, hipStream_t stream) {
    FAIL_POINT("ingest_launch");
    if ((((size_t)src | (size_t)dst | (size_t)src2 | (size_t)dst2) & 15) || (bytes2 & 15)) return hipErrorInvalidValue;
    generic("ingest", stream, [=]() {                // (host-mapped pinned memory -> device: both ranges whole)
        if (bytes) memcpy(dst, src, bytes);
        if (bytes2) memcpy(dst2, src2, bytes2);
    });
    return hipSuccess;
}
hipError_t pyrdown_fused_launch(uint8_t*, size_t, int, int, const int*, const int*, const int*, const size_t*, int, hipStream_t, bool* launched) {
    *launched = false;                              // (the per-level launches above cover the same host path)
    return hipSuccess;
}
static void warp_body(const WarpKernelArgs& a) {
    const size_t M = (size_t)a.m;
    touch(a.T_kf_w, (size_t)a.n_kf * 96); touch(a.cand_kf, M * 4); touch(a.ref_px, M * 8); touch(a.ref_level, M * 4);
    touch(a.ref_bearing, M * 24); touch(a.p_world, M * 24);
    if (a.kf_ptrs) { touch(a.kf_ptrs, (size_t)a.n_kf * sizeof(void*)); for (int k = 0; k < a.n_kf; ++k) touch(a.kf_ptrs[k], pyr_bytes(a.lv)); }
    else touch(a.kf_pyr, (size_t)(a.n_kf - 1) * a.kf_pitch + pyr_bytes(a.lv));
    if (a.cand_frame) { touch(a.cand_frame, M * 4); touch(a.T_cur_w_arr, 96); }
    for (size_t i = 0; i < M; ++i) a.search_level[i] = 0;
    if (a.affine) touch_w(a.affine, M * 32);
}
hipError_t warp_launch(const WarpKernelArgs& a, hipStream_t stream) {
    FAIL_POINT("warp_launch");
    generic("warp", stream, [=]() { warp_body(a); memset(a.patch_border, 0, (size_t)a.m * 100); memset(a.patch, 0, (size_t)a.m * 64); });
    return hipSuccess;
}
hipError_t match_launch(const WarpKernelArgs& wa, const A2DKernelArgs& aa, hipStream_t stream) {
    FAIL_POINT("match_launch");
    generic("match", stream, [=]() {
        warp_body(wa);
        const size_t M = (size_t)aa.m;
        touch(aa.cur_pyr, pyr_bytes(aa.lv)); touch_w(aa.px_xy, M * 16);
        for (size_t i = 0; i < M; ++i) aa.converged[i] = 1;
    });
    return hipSuccess;
}
hipError_t pose_opt_launch(const PoseOptArgs& a, hipStream_t stream) {
    FAIL_POINT("pose_opt_launch");
    generic("pose_opt", stream, [=]() {
        const size_t F = (size_t)a.n_frames, N = (size_t)a.max_features;
        touch(a.n_features, a.n_features ? F * 4 : 0); touch(a.bearing, F * N * 24); touch(a.p_world, F * N * 24); touch(a.level, F * N * 4);
        touch(a.use, F * N); touch_w(a.T_cur_w, F * 96); touch_w(a.residual_norm, F * N * 8);
        for (size_t f = 0; f < F; ++f) { memset(&a.summary[f], 0, sizeof(dsdtm_pose_opt_summary)); a.summary[f].termination = DSDTM_PO_NO_RESIDUALS; }
    });
    return hipSuccess;
}
size_t track_replay_lds_bytes(int n_points, int n_cells, int radius) { return ((size_t)n_points + 63) / 64 * 64 * 22 + (size_t)n_cells * 8 + (size_t)radius * 2 + 256; }   // (the real layout's size)
void track_disc_half_widths(int radius, int8_t* hw) { for (int i = 0; i <= radius; ++i) hw[i] = (int8_t)radius; }
hipError_t track_match_launch(const TrackArgs& a, const WarpKernelArgs& wa, const A2DKernelArgs& aa, hipStream_t stream) {
    FAIL_POINT("track_match_launch");
    if (wa.m != a.n_points || aa.m != a.n_points) return hipErrorInvalidValue;
    generic("track_match", stream, [=]() {
        const size_t M = (size_t)a.n_points;
        touch(a.T_run, 96); touch(a.n_tracked, 4); touch(a.T_kf_w, (size_t)a.n_kf * 96); touch(a.kf_ptrs, (size_t)a.n_kf * sizeof(void*));
        touch(a.mp_world, M * 24); touch(a.mp_found, M * 4); touch(a.mp_bad, M); touch(a.obs_offset, (M + 1) * 4);
        const size_t nnz = M ? (size_t)a.obs_offset[M] : 0;
        touch(a.obs_kf, nnz * 4); touch(a.obs_px, nnz * 8); touch(a.obs_level, nnz * 4); touch(a.obs_bearing, nnz * 24);
        touch(a.mask, a.mask ? (size_t)a.mask_stride * a.height : 0);
        memmove(a.T_opt, a.T_run, 96);
        memmove(a.run_out_host, a.run_out_dev, (size_t)a.run_out_n16 * 16);      // Run's pose, count, statistics -> the pinned block
        touch_w(a.pw, M * 24); touch_w(a.px0, M * 16); touch_w(a.px, M * 16); touch_w(a.ref_px, M * 8); touch_w(a.ref_bearing, M * 24); touch_w(a.init_blocked, M);
        for (size_t i = 0; i < M; ++i) { a.cell[i] = -1; a.cand_kf[i] = -1; a.cand_frame[i] = 0; a.ref_level[i] = 0; a.init_blocked[i] = 0; }
        // the FindMatchDirect half over the columns just written
        if (wa.T_cur_w_arr != a.T_run || wa.cand_kf != a.cand_kf || wa.p_world != a.pw || aa.px_xy != a.px) abort();
        warp_body(wa);
        touch(aa.cur_pyr, pyr_bytes(aa.lv)); touch_w(aa.px_xy, M * 16);
        for (size_t i = 0; i < M; ++i) aa.converged[i] = 1;
    });
    return hipSuccess;
}
hipError_t track_replay_launch(const TrackArgs& a, hipStream_t stream) {
    FAIL_POINT("track_replay_launch");
    generic("track_replay", stream, [=]() {
        const size_t M = (size_t)a.n_points, MM = (size_t)a.max_matches;
        touch(a.cell, M * 4); touch(a.converged, M); touch(a.search_level, M * 4);
        touch_w(a.matches, MM * sizeof(dsdtm_track_match)); touch_w(a.po_bearing, MM * 24); touch_w(a.po_world, MM * 24);
        touch_w(a.po_level, MM * 4); touch_w(a.po_use, MM);
        a.po_n[0] = 0; a.counts[0] = 0; a.counts[1] = 0; a.counts[2] = 0;
    });
    return hipSuccess;
}

}  // namespace dsdtm
