// api.cpp — the C ABI of include/dsdtm_amd.h: context, host staging, launches.
//
// Host entry points pack their inputs into one pinned buffer, move it with a single
// hipMemcpyAsync, launch on the context's stream and copy the (small) results back. Device
// entry points only enqueue kernels. There is no CPU compute path in this library.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "../../include/dsdtm_amd.h"
#include "kernels.h"

using namespace dsdtm;

struct dsdtm_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    char err[512] = {0};
    // staging
    void* h_pinned = nullptr;
    size_t h_cap = 0;
    void* d_stage = nullptr;
    size_t d_cap = 0;
    // workspace for the generic sparse-align kernel
    void* d_ws = nullptr;
    size_t ws_cap = 0;
    // Pair counters of the persistent sparse-align kernel (a launch's slots pull pair indices from one word; the
    // word is zero when a launch starts and the launch's last claim puts it back to zero). A word must never be
    // shared by two launches that can run at the same time, so:
    //  * every stream that launches through this context owns a ring of COUNTERS_PER_STREAM words (a stream runs
    //    its launches in order, so the previous user of a word has finished when the next one starts);
    //  * a launch captured into a hipGraph gets a word of its own from the graph pool, for the life of the
    //    context, and a memset node in front of it (a hipGraphExec never overlaps itself).
    unsigned* d_counter = nullptr;
    // + the stream's workspace, and an event recorded behind the entry's last launch: what a 17th stream waits for
    // before it takes the entry over (the old stream itself may be gone by then)
    struct StreamRing { hipStream_t stream; unsigned seq; bool used; void* d_ws; size_t ws_cap; unsigned long long last_use; hipEvent_t last; bool last_recorded; };
    unsigned long long ring_tick = 0;
    static constexpr int MAX_STREAMS = 16, COUNTERS_PER_STREAM = 8, GRAPH_COUNTERS = 256;
    StreamRing rings[MAX_STREAMS] = {};
    int graph_counters_used = 0;
    // Exchange buffers of the team kernel: a ring of 8 launches x 64 pairs. Team launches of one context are
    // totally ordered (a launch on another stream first waits for the previous team launch's event), because the
    // members of a team spin on each other and must all be resident at once.
    uint8_t* d_team = nullptr;
    unsigned team_seq = 0;
    unsigned long long team_wraps = 0;    // times the 20-bit launch epoch of the exchange tags wrapped (ring re-zeroed each time)
    hipEvent_t team_event = nullptr;      // recorded behind the last team launch that ran on a caller's stream
    hipStream_t team_last_stream = nullptr;
    bool team_any = false;
    // Hand-over timeout words: host-mapped pinned memory of THIS context (a wave whose bounded wait ran out stores 1;
    // the host reads the word once the launch's stream has drained — no copy, no device-global state):
    //   [0, MAX_STREAMS)            one per stream ring: the one-CU launches of that stream
    //   FLAG_GRAPH                  launches captured into hipGraphs
    //   FLAG_SINGLE                 the synchronous single-pair entry points
    //   FLAG_RECOVER + slot         one per multi-CU launch (team / two-member kernels) that has not been settled yet
    static constexpr int RECOVER_SLOTS = 64;
    static constexpr int FLAG_GRAPH = MAX_STREAMS, FLAG_SINGLE = MAX_STREAMS + 1, FLAG_RECOVER = MAX_STREAMS + 2,
                         N_FLAGS = FLAG_RECOVER + RECOVER_SLOTS;
    volatile unsigned* h_flags = nullptr;
    unsigned* d_flags = nullptr;
    // Multi-CU launches wait on partner workgroups, i.e. on the GPU's dispatch: a wait that runs out (a foreign load on
    // the device, another partition mode) must not fail the caller. Every such launch leaves what a re-run needs —
    // descriptor, a copy of its seed poses, an event — and is settled by dsdtm_sparse_align_check (or when its slot is
    // needed again): timeout word clear -> forgotten; set -> poses re-seeded, the same batch relaunched on the one-CU
    // kernels, which cannot wait for anything outside their own workgroup.
    struct Recover {
        bool used = false;
        dsdtm_batch_desc b; dsdtm_camera cam; dsdtm_align_params prm;
        hipStream_t stream = nullptr;
        hipEvent_t done = nullptr;
        void* d_seed = nullptr; size_t seed_cap = 0;
        unsigned long long order = 0;
    };
    Recover rec[RECOVER_SLOTS];
    unsigned long long rec_tick = 0;
    unsigned long long recovered = 0;     // launches re-run on the one-CU kernels so far (dsdtm_debug_recovered_launches)
    bool evicted_timeout = false;         // a stream ring was handed to another stream while its timeout word was set
    bool multi_cu_ok = true;              // the dispatch assumptions of the multi-CU kernels hold on this device
    // dsdtm_sparse_align_batch_streamed: two copy streams and one event per chunk, created on first use
    hipStream_t copy_stream[2] = {nullptr, nullptr};
    std::vector<hipEvent_t> copy_events;
    hipEvent_t copy_fence = nullptr;
    int num_cus = 256;
    // Pyramid buffers of destroyed frames, kept for the next frame of the same size: a live tracker creates and destroys one
    // frame per image, and hipMalloc / hipFree (which waits for the whole device) cost more than the pyramid kernel.
    hipEvent_t track_event = nullptr;     // dsdtm_track_frame: the local map has arrived (copied up on copy_stream[0] beside Run)
    struct PooledFrame { size_t pitch; uint8_t* d; };
    static constexpr size_t FRAME_POOL = 8;
    std::vector<PooledFrame> frame_pool;
};

// Contexts that are alive (dsdtm_frame_destroy may be handed a context that is already gone: it must not be dereferenced)
static std::mutex g_live_mutex;
static std::vector<dsdtm_ctx*> g_live_ctx;
static bool ctx_is_live(dsdtm_ctx* ctx) {
    std::lock_guard<std::mutex> g(g_live_mutex);
    for (dsdtm_ctx* c : g_live_ctx) if (c == ctx) return true;
    return false;
}

static thread_local char g_create_err[512] = "";

static void set_err(dsdtm_ctx* ctx, const char* fmt, ...) {
    char* dst = ctx ? ctx->err : g_create_err;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(dst, 512, fmt, ap);
    va_end(ap);
}

#define HIP_TRY(ctx, call)                                                                   \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            set_err(ctx, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return DSDTM_ERR_HIP;                                                            \
        }                                                                                    \
    } while (0)

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- diagnostic switches (kernels.h: Options) — diagnostic build only ----------------------------
// The release library has no switches: options() is a compile-time constant there, nothing reads the environment,
// and no dsdtm_debug_* symbol exists (tests/test_capi_cpu.py checks `nm -D` and `strings`).
#ifdef DSDTM_DIAG
namespace {
struct OptionKey { const char* key; const char* env; int dsdtm::Options::*field; bool flag; };
const OptionKey kOptionKeys[] = {
    {"no_team", "DSDTM_NO_TEAM", &dsdtm::Options::no_team, true},
    {"team_min", "DSDTM_TEAM_MIN", &dsdtm::Options::team_min, false},
    {"team_spread_min", "DSDTM_TEAM_SPREAD_MIN", &dsdtm::Options::team_spread_min, false},
    {"ws_from", "DSDTM_WS_FROM", &dsdtm::Options::ws_from, false},
    {"ws_no_windows", "DSDTM_WS_NO_WINDOWS", &dsdtm::Options::ws_no_windows, true},
    {"ws_no_duo", "DSDTM_WS_NO_DUO", &dsdtm::Options::ws_no_duo, true},
    {"ws_no_sort", "DSDTM_WS_NO_SORT", &dsdtm::Options::ws_no_sort, true},
    {"fmd_split", "DSDTM_FMD_SPLIT", &dsdtm::Options::fmd_split, true},
    {"fmd_no_xcd", "DSDTM_FMD_NO_XCD", &dsdtm::Options::fmd_no_xcd, true},
    {"match_group", "DSDTM_MATCH_GROUP", &dsdtm::Options::match_group, false},
    {"pyr_fused", "DSDTM_PYR_FUSED", &dsdtm::Options::pyr_fused, false},
    {"pyr_band", "DSDTM_PYR_BAND", &dsdtm::Options::pyr_band, false},
    {"no_zero_copy", "DSDTM_NO_ZERO_COPY", &dsdtm::Options::no_zero_copy, true},
    {"po_no_cache", "DSDTM_PO_NO_CACHE", &dsdtm::Options::po_no_cache, true},
    {"a2d_tree", "DSDTM_A2D_TREE", &dsdtm::Options::a2d_tree, true},
    {"no_recover", "DSDTM_NO_RECOVER", &dsdtm::Options::no_recover, true},
    {"team_no_wrap_clear", "DSDTM_TEAM_NO_WRAP_CLEAR", &dsdtm::Options::team_no_wrap_clear, true},
    {"warp_group", "DSDTM_WARP_GROUP", &dsdtm::Options::warp_group, false},
    {"po_rows", "DSDTM_PO_ROWS", &dsdtm::Options::po_rows, false},
    {"a2d_group", "DSDTM_A2D_GROUP", &dsdtm::Options::a2d_group, false},
};
std::once_flag g_options_once;
void options_from_env() {
    dsdtm::Options& o = dsdtm::options();
    for (const OptionKey& k : kOptionKeys)
        if (const char* v = getenv(k.env)) o.*(k.field) = k.flag ? 1 : atoi(v);    // a flag is set by its presence
}
}  // namespace
dsdtm::Options& dsdtm::options() {
    static Options o;
    return o;
}
#endif  // DSDTM_DIAG

extern "C" {

#ifdef DSDTM_DIAG
const char* dsdtm_version(void) { return "dsdtm_amd 0.6 (gfx950, HIP; FP64 reference grid; DIAGNOSTIC build)"; }
#else
const char* dsdtm_version(void) { return "dsdtm_amd 0.6 (gfx950, HIP; FP64 reference grid)"; }
#endif

int dsdtm_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return DSDTM_ERR_NO_DEVICE;
    return n;
}

const char* dsdtm_last_error(const dsdtm_ctx* ctx) { return ctx ? ctx->err : g_create_err; }

int dsdtm_create(int device, dsdtm_ctx** out) {
    if (!out) return DSDTM_ERR_INVALID;
    *out = nullptr;
#ifdef DSDTM_DIAG
    std::call_once(g_options_once, options_from_env);     // diagnostic build: the only place the library reads the environment
#endif
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_err(nullptr, "no HIP device visible (%s); dsdtm_amd has no CPU fallback",
                e != hipSuccess ? hipGetErrorString(e) : "device count 0");
        return DSDTM_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= n) {
        set_err(nullptr, "device %d out of range (%d visible)", device, n);
        return DSDTM_ERR_NO_DEVICE;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) {
        set_err(nullptr, "hipGetDeviceProperties(%d) failed", device);
        return DSDTM_ERR_NO_DEVICE;
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_err(nullptr, "device %d is %s; this library contains gfx950 (MI355X) code objects only", device,
                prop.gcnArchName);
        return DSDTM_ERR_NO_DEVICE;
    }
    dsdtm_ctx* ctx = new (std::nothrow) dsdtm_ctx();
    if (!ctx) return DSDTM_ERR_NOMEM;
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        set_err(nullptr, "hipSetDevice/hipStreamCreate failed on device %d", device);
        delete ctx;
        return DSDTM_ERR_HIP;
    }
    ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    {
        // The team / two-member kernels place the members of a pair on one XCD by workgroup number (b, b + 8, ...) and
        // rely on all 256 CUs being one partition: anything else (CPX/DPX/QPX modes, fewer XCDs) runs the one-CU kernels.
        int xccs = 0;
        if (hipDeviceGetAttribute(&xccs, hipDeviceAttributeNumberOfXccs, device) != hipSuccess) xccs = 0;
        ctx->multi_cu_ok = (xccs == 8 && ctx->num_cus == 256);
    }
    const size_t counter_bytes = sizeof(unsigned) * (dsdtm_ctx::MAX_STREAMS * dsdtm_ctx::COUNTERS_PER_STREAM + dsdtm_ctx::GRAPH_COUNTERS);
    void* hf = nullptr;
    void* hfd = nullptr;
    if (hipHostMalloc(&hf, sizeof(unsigned) * dsdtm_ctx::N_FLAGS, hipHostMallocMapped) == hipSuccess &&
        hipHostGetDevicePointer(&hfd, hf, 0) == hipSuccess) {
        memset(hf, 0, sizeof(unsigned) * dsdtm_ctx::N_FLAGS);
        ctx->h_flags = (volatile unsigned*)hf; ctx->d_flags = (unsigned*)hfd;
    }
    if (!ctx->d_flags ||
        hipMalloc((void**)&ctx->d_counter, counter_bytes) != hipSuccess ||
        hipMalloc((void**)&ctx->d_team, 8 * sparse_align_team_bytes(64)) != hipSuccess ||
        hipMemset(ctx->d_counter, 0, counter_bytes) != hipSuccess ||
        hipMemset(ctx->d_team, 0, 8 * sparse_align_team_bytes(64)) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->team_event, hipEventDisableTiming) != hipSuccess) {
        if (ctx->d_counter) (void)hipFree(ctx->d_counter);
        if (ctx->d_team) (void)hipFree(ctx->d_team);
        if (hf) (void)hipHostFree(hf);
        set_err(nullptr, "hipMalloc failed on device %d", device);
        (void)hipStreamDestroy(ctx->stream);
        delete ctx;
        return DSDTM_ERR_NOMEM;
    }
    try {
        std::lock_guard<std::mutex> g(g_live_mutex);
        g_live_ctx.push_back(ctx);
    } catch (...) { dsdtm_destroy(ctx); return DSDTM_ERR_NOMEM; }
    *out = ctx;
    return DSDTM_OK;
}

#ifdef DSDTM_DIAG
// Diagnostic switches by name (tests, A/B tools): the keys of kOptionKeys, process-wide. Returns DSDTM_ERR_INVALID
// for an unknown key. Not for production use: no entry point synchronises against a concurrent change.
int dsdtm_debug_set_option(const char* key, int value) {
    std::call_once(g_options_once, options_from_env);
    for (const OptionKey& k : kOptionKeys)
        if (key && strcmp(key, k.key) == 0) { dsdtm::options().*(k.field) = value; return DSDTM_OK; }
    return DSDTM_ERR_INVALID;
}
int dsdtm_debug_get_option(const char* key, int* value) {
    std::call_once(g_options_once, options_from_env);
    for (const OptionKey& k : kOptionKeys)
        if (key && value && strcmp(key, k.key) == 0) { *value = dsdtm::options().*(k.field); return DSDTM_OK; }
    return DSDTM_ERR_INVALID;
}
#endif  // DSDTM_DIAG

void dsdtm_destroy(dsdtm_ctx* ctx) {
    if (!ctx) return;
    {
        std::lock_guard<std::mutex> g(g_live_mutex);
        for (size_t i = 0; i < g_live_ctx.size(); ++i)
            if (g_live_ctx[i] == ctx) { g_live_ctx.erase(g_live_ctx.begin() + (long)i); break; }
    }
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) { (void)hipStreamSynchronize(ctx->stream); (void)hipStreamDestroy(ctx->stream); }
    if (ctx->h_pinned) (void)hipHostFree(ctx->h_pinned);
    if (ctx->d_stage) (void)hipFree(ctx->d_stage);
    if (ctx->d_ws) (void)hipFree(ctx->d_ws);
    for (auto& r : ctx->rings) { if (r.d_ws) (void)hipFree(r.d_ws); if (r.last) (void)hipEventDestroy(r.last); }
    if (ctx->d_counter) (void)hipFree(ctx->d_counter);
    if (ctx->d_team) (void)hipFree(ctx->d_team);
    if (ctx->team_event) (void)hipEventDestroy(ctx->team_event);
    for (auto& r : ctx->rec) { if (r.d_seed) (void)hipFree(r.d_seed); if (r.done) (void)hipEventDestroy(r.done); }
    for (auto& cs : ctx->copy_stream) if (cs) { (void)hipStreamSynchronize(cs); (void)hipStreamDestroy(cs); }
    for (auto& e : ctx->copy_events) (void)hipEventDestroy(e);
    if (ctx->copy_fence) (void)hipEventDestroy(ctx->copy_fence);
    if (ctx->h_flags) (void)hipHostFree((void*)ctx->h_flags);
    for (auto& pf : ctx->frame_pool) (void)hipFree(pf.d);
    if (ctx->track_event) (void)hipEventDestroy(ctx->track_event);
    delete ctx;
}

}  // extern "C"

static int ensure_stage(dsdtm_ctx* ctx, size_t bytes) {
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (bytes > ctx->h_cap) {
        if (ctx->h_pinned) (void)hipHostFree(ctx->h_pinned);
        ctx->h_pinned = nullptr; ctx->h_cap = 0;
        const size_t cap = align_up(bytes + bytes / 4, 1 << 16);
        HIP_TRY(ctx, hipHostMalloc(&ctx->h_pinned, cap, hipHostMallocDefault));
        ctx->h_cap = cap;
    }
    if (bytes > ctx->d_cap) {
        if (ctx->d_stage) (void)hipFree(ctx->d_stage);
        ctx->d_stage = nullptr; ctx->d_cap = 0;
        const size_t cap = align_up(bytes + bytes / 4, 1 << 16);
        HIP_TRY(ctx, hipMalloc(&ctx->d_stage, cap));
        ctx->d_cap = cap;
    }
    return DSDTM_OK;
}

extern "C" int dsdtm_reserve(dsdtm_ctx* ctx, size_t workspace_bytes) {
    if (!ctx) return DSDTM_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (workspace_bytes > ctx->ws_cap) {
        if (ctx->d_ws) { HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); (void)hipFree(ctx->d_ws); }
        ctx->d_ws = nullptr; ctx->ws_cap = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->d_ws, workspace_bytes));
        ctx->ws_cap = workspace_bytes;
    }
    return DSDTM_OK;
}

extern "C" size_t dsdtm_sparse_align_workspace_bytes(const dsdtm_batch_desc* b) {
    if (!b) return 0;
    return sparse_align_workspace_bytes(b->n_pairs, b->max_features);
}

static int validate_params(dsdtm_ctx* ctx, const dsdtm_align_params* p, int levels) {
    if (!p) { set_err(ctx, "params is NULL"); return DSDTM_ERR_INVALID; }
    if (p->max_level > levels || p->max_level > DSDTM_MAX_LEVELS || p->min_level < 0) {
        set_err(ctx, "level range [%d,%d) does not fit a %d-level pyramid", p->min_level, p->max_level, levels);
        return DSDTM_ERR_INVALID;
    }
    return DSDTM_OK;
}

// behind a launch that used ring entry `ring`: once every entry is taken, leave an event for a later hand-over
static int ring_mark_launch(dsdtm_ctx* ctx, int ring, hipStream_t stream) {
    if (ring < 0) return DSDTM_OK;
    bool full = true;
    for (const auto& r : ctx->rings) full = full && r.used;
    if (!full) return DSDTM_OK;
    HIP_TRY(ctx, hipEventRecord(ctx->rings[ring].last, stream));
    ctx->rings[ring].last_recorded = true;
    return DSDTM_OK;
}

#ifdef DSDTM_DIAG
static thread_local void* g_stamp_out = nullptr;   // device buffer, set only by the stamps debug entry
static thread_local int g_team_drop_members = 0;         // set only by dsdtm_debug_sparse_align_short_team
#else
static constexpr void* g_stamp_out = nullptr;      // (release build: neither diagnostic exists)
static constexpr int g_team_drop_members = 0;
#endif

// How a launch is accounted for (see dsdtm_ctx::h_flags / Recover)
struct LaunchMode {
    bool one_cu = false;        // never a team / two-member kernel (re-runs after a timeout, captured launches)
    bool single = false;        // synchronous single-pair entry: its own timeout word, settled by the caller itself
};
static int launch_batch(dsdtm_ctx* ctx, const dsdtm_batch_desc* b, const dsdtm_camera* cam, const dsdtm_align_params* prm,
                        void* hip_stream, const LaunchMode& mode, bool* multi_cu_used);
static int recover_settle(dsdtm_ctx* ctx, int slot, hipStream_t rerun_stream);
// Drops the re-run records of `stream` without settling them: for entry points that return an error and whose device
// buffers (the records hold raw pointers into them) may be freed or regrown before the next check on that stream.
static void recover_forget_stream(dsdtm_ctx* ctx, hipStream_t stream) {
    for (int i = 0; i < dsdtm_ctx::RECOVER_SLOTS; ++i)
        if (ctx->rec[i].used && ctx->rec[i].stream == stream) {
            ctx->rec[i].used = false;
            ctx->h_flags[dsdtm_ctx::FLAG_RECOVER + i] = 0;
        }
}

extern "C" int dsdtm_sparse_align_batch_device(dsdtm_ctx* ctx, const dsdtm_batch_desc* b, const dsdtm_camera* cam,
                                               const dsdtm_align_params* prm, void* hip_stream) {
    return launch_batch(ctx, b, cam, prm, hip_stream, LaunchMode{}, nullptr);
}

// A free recovery slot. When all are taken, the oldest launch is settled first (its event is waited for; a launch
// that timed out is re-run on its own stream): at most RECOVER_SLOTS unchecked multi-CU launches are in flight.
static int recover_acquire(dsdtm_ctx* ctx, int* slot_out) {
    int oldest = -1;
    for (int i = 0; i < dsdtm_ctx::RECOVER_SLOTS; ++i) {
        if (!ctx->rec[i].used) { *slot_out = i; return DSDTM_OK; }
        if (oldest < 0 || ctx->rec[i].order < ctx->rec[oldest].order) oldest = i;
    }
    // (settled on the stream it was launched on: a re-run must stay ordered with the later work queued there on the same
    // pose / count buffers; those buffers and the stream stay the caller's to keep alive until its check — dsdtm_amd.h)
    HIP_TRY(ctx, hipEventSynchronize(ctx->rec[oldest].done));
    if (int rc = recover_settle(ctx, oldest, ctx->rec[oldest].stream)) return rc;
    *slot_out = oldest;
    return DSDTM_OK;
}

// Settles slot `slot` (its launch has finished): forgotten when its timeout word is clear; otherwise the batch is
// re-seeded and re-run on the one-CU kernels on `rerun_stream`, which is then waited for.
static int recover_settle(dsdtm_ctx* ctx, int slot, hipStream_t rerun_stream) {
    dsdtm_ctx::Recover& r = ctx->rec[slot];
    if (!r.used) return DSDTM_OK;
    r.used = false;
    volatile unsigned* flag = ctx->h_flags + dsdtm_ctx::FLAG_RECOVER + slot;
    if (!*flag) return DSDTM_OK;
    *flag = 0;
    if (options().no_recover) {
        set_err(ctx, "sparse-align kernel: a wait for a partner workgroup timed out (re-run disabled: no_recover)");
        return DSDTM_ERR_HIP;
    }
    // The re-run is queued BEHIND whatever the caller issued on that stream since. A later unsettled launch of this stream that
    // names the same pose buffer (a loop into one buffer with no check in between) would have its results overwritten by this
    // older launch's: that cannot be repaired here, so it is reported instead of papered over (dsdtm_amd.h: unchecked launches
    // of one stream name distinct output buffers).
    for (int i = 0; i < dsdtm_ctx::RECOVER_SLOTS; ++i) {
        const dsdtm_ctx::Recover& y = ctx->rec[i];
        if (i != slot && y.used && y.stream == r.stream && y.order > r.order && y.b.T_cur_w == r.b.T_cur_w) {
            set_err(ctx, "sparse-align kernel: a wait for a partner workgroup timed out in a launch whose pose buffer a later unchecked "
                         "launch of the same stream reuses: it cannot be re-run (check the stream between launches into one buffer)");
            return DSDTM_ERR_HIP;
        }
    }
    HIP_TRY(ctx, hipMemcpyAsync(r.b.T_cur_w, r.d_seed, (size_t)r.b.n_pairs * 96, hipMemcpyDeviceToDevice, rerun_stream));
    LaunchMode m;
    m.one_cu = true;
    if (int rc = launch_batch(ctx, &r.b, &r.cam, &r.prm, rerun_stream, m, nullptr)) return rc;
    ctx->recovered += 1;
    return dsdtm_sparse_align_check(ctx, rerun_stream);
}

static int launch_batch(dsdtm_ctx* ctx, const dsdtm_batch_desc* b, const dsdtm_camera* cam, const dsdtm_align_params* prm,
                        void* hip_stream, const LaunchMode& mode, bool* multi_cu_used) {
    if (multi_cu_used) *multi_cu_used = false;
    if (!ctx) return DSDTM_ERR_INVALID;
    if (!b || !cam) { set_err(ctx, "batch/cam is NULL"); return DSDTM_ERR_INVALID; }
    if (int rc = validate_params(ctx, prm, b->levels)) return rc;
    if (b->n_pairs < 0 || b->max_features < 0 || b->max_features > 32767 || b->levels <= 0 || b->levels > DSDTM_MAX_LEVELS) {
        set_err(ctx, "bad batch geometry"); return DSDTM_ERR_INVALID;
    }
    if (b->n_pairs == 0) return DSDTM_OK;
    if (!b->ref_pyr || !b->cur_pyr || !b->T_ref_w || !b->T_cur_w || !b->n_tracked ||
        (b->max_features > 0 && (!b->px_xy || !b->bearing || !b->p_world || !b->initial))) {
        set_err(ctx, "batch descriptor has NULL device pointers"); return DSDTM_ERR_INVALID;
    }
    if ((b->pyr_pitch & 3) || b->pyr_pitch == 0 || b->pyr_pitch > 0xffffffffull ||
        (((size_t)b->ref_pyr) & 3) || (((size_t)b->cur_pyr) & 3)) {
        set_err(ctx, "pyramid base/pitch must be 4-byte aligned and pitch < 4 GiB"); return DSDTM_ERR_INVALID;
    }
    SAKernelArgs a;
    memset(&a, 0, sizeof a);
    for (int l = 0; l < b->levels; ++l) {
        const size_t end = b->level_offset[l] + (size_t)b->stride[l] * b->height[l];
        if (b->width[l] <= 0 || b->height[l] <= 0 || b->stride[l] < b->width[l] || end > b->pyr_pitch) {
            set_err(ctx, "level %d does not fit inside pyr_pitch", l); return DSDTM_ERR_INVALID;
        }
        a.lv[l].w = b->width[l]; a.lv[l].h = b->height[l]; a.lv[l].stride = b->stride[l];
        a.lv[l].off = (uint32_t)b->level_offset[l];
    }
    a.ref_pyr = b->ref_pyr; a.cur_pyr = b->cur_pyr; a.px_xy = b->px_xy; a.bearing = b->bearing;
    a.p_world = b->p_world; a.initial = b->initial; a.n_features = b->n_features;
    a.T_ref_w = b->T_ref_w; a.T_cur_w = b->T_cur_w; a.n_tracked = b->n_tracked; a.stats = b->stats;
    a.pyr_pitch = b->pyr_pitch; a.n_pairs = b->n_pairs; a.max_features = b->max_features;
    a.max_level = prm->max_level; a.min_level = prm->min_level; a.max_iters = prm->max_iters; a.min_fts = prm->min_fts;
    a.fx = cam->fx; a.fy = cam->fy; a.cx = cam->cx; a.cy = cam->cy; a.f = cam->f;
    a.ws_sort = options().ws_no_sort ? 0 : 1;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t stream = (hipStream_t)hip_stream;
    int ring = -1;                                     // this stream's entry of ctx->rings (not while capturing)
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    const bool capturing = stream && hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
    // the word the launch's persistent slots pull pair indices from (see dsdtm_ctx::d_counter)
    if (capturing) {
        if (ctx->graph_counters_used >= dsdtm_ctx::GRAPH_COUNTERS) {
            set_err(ctx, "more than %d launches captured into hipGraphs through one context: create another context",
                    dsdtm_ctx::GRAPH_COUNTERS);
            return DSDTM_ERR_INVALID;
        }
        a.pair_counter = ctx->d_counter + dsdtm_ctx::MAX_STREAMS * dsdtm_ctx::COUNTERS_PER_STREAM + ctx->graph_counters_used++;
        HIP_TRY(ctx, hipMemsetAsync(a.pair_counter, 0, sizeof(unsigned), stream));   // memset node: replays heal themselves
        a.timeout_flag = ctx->d_flags + dsdtm_ctx::FLAG_GRAPH;
    } else {
        int ri = -1;
        for (int i = 0; i < dsdtm_ctx::MAX_STREAMS && ri < 0; ++i)
            if (ctx->rings[i].used && ctx->rings[i].stream == stream) ri = i;
        for (int i = 0; i < dsdtm_ctx::MAX_STREAMS && ri < 0; ++i)
            if (!ctx->rings[i].used) {
                hipEvent_t ev = nullptr;
                HIP_TRY(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
                ctx->rings[i] = dsdtm_ctx::StreamRing{stream, 0u, true, nullptr, 0, 0ull, ev, false};
                ri = i;
            }
        if (ri < 0) {
            // A 17th stream (applications that keep creating streams): the entry that has been idle longest is handed
            // over once the launches behind it have finished — the host waits for the event recorded behind the
            // entry's last launch (an event outlives its stream; no device-wide synchronisation, which would also be
            // illegal while another stream is capturing). After that nothing uses the entry's counters or workspace.
            // Events are only recorded once all entries are taken (rings_full), so that applications with few
            // streams pay nothing; an entry last used before that is waited for with the device, once.
            ri = 0;
            for (int i = 1; i < dsdtm_ctx::MAX_STREAMS; ++i)
                if (ctx->rings[i].last_use < ctx->rings[ri].last_use) ri = i;
            if (ctx->rings[ri].last_recorded) HIP_TRY(ctx, hipEventSynchronize(ctx->rings[ri].last));
            else HIP_TRY(ctx, hipDeviceSynchronize());
            ctx->rings[ri].last_recorded = false;
            ctx->rings[ri].stream = stream;
            ctx->rings[ri].seq = 0;
            // (the previous owner's launches have drained: its word starts clean — a timeout it raised and nobody checked for
            // is not lost with it: the next check on any stream of this context reports it)
            if (ctx->h_flags[ri]) ctx->evicted_timeout = true;
            ctx->h_flags[ri] = 0;
        }
        ctx->rings[ri].last_use = ++ctx->ring_tick;
        a.pair_counter = ctx->d_counter + ri * dsdtm_ctx::COUNTERS_PER_STREAM + (ctx->rings[ri].seq++ % dsdtm_ctx::COUNTERS_PER_STREAM);
        ring = ri;
        a.timeout_flag = ctx->d_flags + ri;
    }
    if (mode.single) a.timeout_flag = ctx->d_flags + dsdtm_ctx::FLAG_SINGLE;
#ifdef DSDTM_DIAG
    if (g_stamp_out) {   // diagnostic path of dsdtm_debug_sparse_align_stamps
        if (b->max_features > 320) { set_err(ctx, "stamps: <=320 features only"); return DSDTM_ERR_INVALID; }
        a.workspace = (double*)g_stamp_out;
        HIP_TRY(ctx, sparse_align_launch_stamps(a, ctx->num_cus, stream));
        return DSDTM_OK;
    }
#endif
    // Few pairs of more than 448 features: one pair over K compute units. The members of a team spin on each
    // other, so all of a launch's workgroups must be resident together: sparse_align_team_size admits a launch
    // only when it fills at most half the CUs, and the team launches of a context are totally ordered — one on
    // another stream first waits for the previous one's event — so two of them never compete for CUs (kernels
    // of OTHER kinds on other streams can still delay a member: such a wait ends when their workgroups drain; a
    // wait that does not end raises the timeout flag, see dsdtm_sparse_align_check). Not used while capturing
    // (a graph replay could not be ordered against live team launches): those shapes take the one-CU kernels.
    // Multi-CU kernels only where their dispatch assumptions hold, never in a re-run, never while capturing (a graph
    // replay could neither be ordered against live team launches nor be re-run after a timeout).
    const bool multi_cu = ctx->multi_cu_ok && !mode.one_cu && !capturing;
    const int team_k = (multi_cu && b->n_pairs <= 64 && !options().no_team) ? sparse_align_team_size(b->n_pairs, b->max_features, ctx->num_cus) : 0;
    const SAVariant v = sparse_align_pick_variant(b->max_features);
    const size_t ws = sparse_align_workspace_bytes(b->n_pairs, b->max_features);
    const bool duo = !team_k && multi_cu && sparse_align_uses_duo(b->max_features, ws != 0);
    int slot = -1;
    if ((team_k || duo) && !mode.single) {
        // what a re-run needs, should a wait for a partner workgroup run out (dsdtm_ctx::Recover)
        if (int rc = recover_acquire(ctx, &slot)) return rc;
        dsdtm_ctx::Recover& r = ctx->rec[slot];
        const size_t seed_bytes = (size_t)b->n_pairs * 96;
        if (seed_bytes > r.seed_cap) {
            if (r.d_seed) (void)hipFree(r.d_seed);     // (the slot is free: nothing in flight uses its buffer)
            r.d_seed = nullptr; r.seed_cap = 0;
            HIP_TRY(ctx, hipMalloc(&r.d_seed, align_up(seed_bytes, 4096)));
            r.seed_cap = align_up(seed_bytes, 4096);
        }
        if (!r.done) HIP_TRY(ctx, hipEventCreateWithFlags(&r.done, hipEventDisableTiming));
        HIP_TRY(ctx, hipMemcpyAsync(r.d_seed, b->T_cur_w, seed_bytes, hipMemcpyDeviceToDevice, stream));
        r.b = *b; r.cam = *cam; r.prm = *prm; r.stream = stream; r.order = ++ctx->rec_tick;
        ctx->h_flags[dsdtm_ctx::FLAG_RECOVER + slot] = 0;
        a.timeout_flag = ctx->d_flags + dsdtm_ctx::FLAG_RECOVER + slot;
    }
    if (multi_cu_used) *multi_cu_used = team_k || duo;
    auto launched_multi_cu = [&]() -> int {
        if (slot < 0) return DSDTM_OK;
        HIP_TRY(ctx, hipEventRecord(ctx->rec[slot].done, stream));
        ctx->rec[slot].used = true;
        return DSDTM_OK;
    };
    // Few pairs of more than 448 features: one pair over K compute units. The members of a team spin on each
    // other, so all of a launch's workgroups must be resident together: sparse_align_team_size admits a launch
    // only when it fills at most half the CUs, and the team launches of a context are totally ordered — one on
    // another stream first waits for the previous one's event — so two of them never compete for CUs (kernels
    // of OTHER kinds on other streams can still delay a member: such a wait ends when their workgroups drain; a
    // wait that does not end raises the launch's timeout word and the batch is re-run on the one-CU kernels, see
    // recover_settle).
    if (const int k = team_k) {
        if (ctx->team_any && ctx->team_last_stream != stream) {
            // the context's own stream is alive as long as the context: its event is recorded on demand;
            // launches on a caller's stream left theirs behind (that stream may be gone by now)
            if (ctx->team_last_stream == ctx->stream) HIP_TRY(ctx, hipEventRecord(ctx->team_event, ctx->stream));
            HIP_TRY(ctx, hipStreamWaitEvent(stream, ctx->team_event, 0));
        }
        // (the exchange words carry the launch's epoch in their tags — never 0, the buffers were zeroed when the context
        // was created — so a ring slot is reused without clearing it.) The tags hold 20 bits of epoch: ring slot and epoch
        // repeat together every 2^20 team launches, and a word that member m of pair p last wrote exactly then (team sizes
        // vary from frame to frame) would carry a tag this launch accepts. So when the epoch wraps, all eight ring slots
        // are zeroed on the launch stream — behind every earlier team launch (they are totally ordered, above) — once per
        // 2^20 launches; the reference's Run is stateless across calls (src/Sprase_ImageAlign.cpp:22-27).
        ctx->team_seq += 1;
        if ((ctx->team_seq & 0xfffffu) == 0u) {
            ctx->team_seq += 1;                           // epoch 0 = "never written"
            if (!options().team_no_wrap_clear)
                HIP_TRY(ctx, hipMemsetAsync(ctx->d_team, 0, 8 * sparse_align_team_bytes(64), stream));
            ctx->team_wraps += 1;
        }
        uint8_t* tslot = ctx->d_team + (size_t)(ctx->team_seq & 7u) * sparse_align_team_bytes(64);
        a.workspace = (double*)tslot;
        a.team_epoch = ctx->team_seq;
        if (g_team_drop_members) a.spin_limit = 1u << 12;      // the test's waits give up after ~4 k polls
        HIP_TRY(ctx, sparse_align_launch_team(a, k, stream, g_team_drop_members));
        if (int rc = launched_multi_cu()) return rc;
        if (int rc = ring_mark_launch(ctx, ring, stream)) return rc;
        if (stream != ctx->stream) HIP_TRY(ctx, hipEventRecord(ctx->team_event, stream));
        ctx->team_any = true; ctx->team_last_stream = stream;
        return DSDTM_OK;
    }
    if (ws) {
        // Scratch of the workspace kernel. A live launch uses its STREAM's workspace (the stream runs its launches
        // in order, so launches on different streams never share scratch; grown on demand, which synchronises that
        // stream once). A captured launch cannot allocate: it uses the context's reserved workspace (dsdtm_reserve),
        // which graph replays then share — replays of graphs captured through one context must not overlap.
        if (capturing) {
            if (ws > ctx->ws_cap) {
                set_err(ctx, "workspace of %zu bytes needed: call dsdtm_reserve before capturing", ws);
                return DSDTM_ERR_INVALID;
            }
            a.workspace = (double*)ctx->d_ws;
        } else {
            dsdtm_ctx::StreamRing& r = ctx->rings[ring];
            if (ws > r.ws_cap) {
                if (r.d_ws) { HIP_TRY(ctx, hipStreamSynchronize(stream)); (void)hipFree(r.d_ws); r.d_ws = nullptr; r.ws_cap = 0; }
                HIP_TRY(ctx, hipMalloc(&r.d_ws, ws));
                r.ws_cap = ws;
            }
            a.workspace = (double*)r.d_ws;
        }
    }
    if (duo && g_team_drop_members) { a.spin_limit = 1u << 12; a.debug_drop = 1; }    // tests: member 1 of every pair stays away
    HIP_TRY(ctx, sparse_align_launch(a, v, ctx->num_cus, stream, duo));
    if (int rc = launched_multi_cu()) return rc;
    return ring_mark_launch(ctx, ring, stream);
}

// Result check for callers of the asynchronous batch entry point: waits for `hip_stream`, then settles the launches
// issued on it since the last check. Multi-CU launches (team / two-member kernels) whose wait for a partner workgroup
// ran out are re-run on the one-CU kernels here — the caller sees DSDTM_OK and the results the reference's Run would
// have produced. DSDTM_ERR_HIP remains for a hand-over that failed INSIDE one workgroup (a broken protocol, never
// observed) or a re-run that failed too; the results of this stream's launches since the last check are then not to
// be trusted. Everything read here belongs to this context: no device-wide synchronisation, no device-global state.
extern "C" int dsdtm_sparse_align_check(dsdtm_ctx* ctx, void* hip_stream) {
    if (!ctx) return DSDTM_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t stream = (hipStream_t)hip_stream;
    HIP_TRY(ctx, hipStreamSynchronize(stream));
    // this stream's multi-CU launches, in launch order (a re-run may feed a later launch of the same stream: the caller
    // sees final results only after this function returns)
    for (;;) {
        int next = -1;
        for (int i = 0; i < dsdtm_ctx::RECOVER_SLOTS; ++i)
            if (ctx->rec[i].used && ctx->rec[i].stream == stream && (next < 0 || ctx->rec[i].order < ctx->rec[next].order)) next = i;
        if (next < 0) break;
        if (int rc = recover_settle(ctx, next, stream)) return rc;
    }
    int ring = -1;
    for (int i = 0; i < dsdtm_ctx::MAX_STREAMS; ++i)
        if (ctx->rings[i].used && ctx->rings[i].stream == stream) ring = i;
    bool bad = false;
    if (ring >= 0 && ctx->h_flags[ring]) {
        ctx->h_flags[ring] = 0;
        // a pair that was stopped may have left its counter word behind: this stream's words (it has drained)
        (void)hipMemsetAsync(ctx->d_counter + ring * dsdtm_ctx::COUNTERS_PER_STREAM, 0, sizeof(unsigned) * dsdtm_ctx::COUNTERS_PER_STREAM, stream);
        (void)hipStreamSynchronize(stream);
        bad = true;
    }
    if (ctx->h_flags[dsdtm_ctx::FLAG_GRAPH]) { ctx->h_flags[dsdtm_ctx::FLAG_GRAPH] = 0; bad = true; }   // (graph launches zero their own counters)
    if (ctx->evicted_timeout) { ctx->evicted_timeout = false; bad = true; }
    if (bad) {
        set_err(ctx, "sparse-align kernel: a hand-over wait inside a workgroup timed out (results of this stream since the last check are invalid)");
        return DSDTM_ERR_HIP;
    }
    return DSDTM_OK;
}

#ifdef DSDTM_DIAG
extern "C" long long dsdtm_debug_recovered_launches(dsdtm_ctx* ctx) { return ctx ? (long long)ctx->recovered : -1; }
// Tests: the context's team-launch counter (its low 20 bits are the epoch of the exchange tags, its low 3 bits the ring slot),
// so that the epoch wrap — 2^20 team launches away in real use — can be crossed by a handful of launches. Returns the number
// of wraps seen so far. The caller's team launches must have drained.
extern "C" long long dsdtm_debug_team_seq(dsdtm_ctx* ctx, long long set_to) {
    if (!ctx) return -1;
    if (set_to >= 0) ctx->team_seq = (unsigned)set_to;
    return (long long)ctx->team_wraps;
}
#endif  // DSDTM_DIAG

// ---- the batch from host memory over several contexts / devices ------------------------------
extern "C" void dsdtm_shard_range(int n_pairs, int n_shards, int shard, int* lo, int* hi) {
    const int per = n_shards > 0 ? (n_pairs + n_shards - 1) / n_shards : n_pairs;
    int l = shard * per, h = l + per;
    if (l > n_pairs) l = n_pairs;
    if (h > n_pairs) h = n_pairs;
    if (n_pairs <= 0 || shard < 0 || shard >= n_shards) l = h = 0;
    if (lo) *lo = l;
    if (hi) *hi = h;
}

// One shard: pairs [lo, hi) of the host batch on `ctx` (its device, its stream). Device scratch comes from the
// context's staging buffer (device side only: the caller's arrays may be pageable, the copies are then staged by
// the runtime; pinned arrays are copied directly).
static int sharded_issue(dsdtm_ctx* ctx, const dsdtm_batch_desc* hb, const dsdtm_camera* cam, const dsdtm_align_params* prm,
                         int lo, int hi) {
    const int n = hi - lo;
    if (n <= 0) return DSDTM_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t nf = (size_t)hb->max_features, pit = hb->pyr_pitch;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t o_ref = take(n * pit), o_cur = take(n * pit), o_px = take(n * nf * 8), o_be = take(n * nf * 24),
                 o_pw = take(n * nf * 24), o_in = take(n * nf), o_nf = take(hb->n_features ? n * 4 : 0),
                 o_tr = take(n * 96), o_tc = take(n * 96), o_nt = take(n * 4),
                 o_st = take(hb->stats ? n * sizeof(dsdtm_align_stats) : 0);
    if (off > ctx->d_cap) {
        if (ctx->d_stage) { HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); (void)hipFree(ctx->d_stage); }
        ctx->d_stage = nullptr; ctx->d_cap = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->d_stage, off));
        ctx->d_cap = off;
    }
    uint8_t* d = (uint8_t*)ctx->d_stage;
    hipStream_t st = ctx->stream;
    auto up = [&](size_t o, const void* src, size_t bytes) {
        return bytes ? hipMemcpyAsync(d + o, src, bytes, hipMemcpyHostToDevice, st) : hipSuccess;
    };
    HIP_TRY(ctx, up(o_ref, hb->ref_pyr + (size_t)lo * pit, n * pit));
    HIP_TRY(ctx, up(o_cur, hb->cur_pyr + (size_t)lo * pit, n * pit));
    HIP_TRY(ctx, up(o_px, hb->px_xy + (size_t)lo * nf * 2, n * nf * 8));
    HIP_TRY(ctx, up(o_be, hb->bearing + (size_t)lo * nf * 3, n * nf * 24));
    HIP_TRY(ctx, up(o_pw, hb->p_world + (size_t)lo * nf * 3, n * nf * 24));
    HIP_TRY(ctx, up(o_in, hb->initial + (size_t)lo * nf, n * nf));
    if (hb->n_features) HIP_TRY(ctx, up(o_nf, hb->n_features + lo, n * 4));
    HIP_TRY(ctx, up(o_tr, hb->T_ref_w + (size_t)lo * 12, n * 96));
    HIP_TRY(ctx, up(o_tc, hb->T_cur_w + (size_t)lo * 12, n * 96));
    dsdtm_batch_desc b = *hb;
    b.n_pairs = n;
    b.ref_pyr = d + o_ref; b.cur_pyr = d + o_cur;
    b.px_xy = (const float*)(d + o_px); b.bearing = (const double*)(d + o_be); b.p_world = (const double*)(d + o_pw);
    b.initial = d + o_in; b.n_features = hb->n_features ? (const int32_t*)(d + o_nf) : nullptr;
    b.T_ref_w = (const double*)(d + o_tr); b.T_cur_w = (double*)(d + o_tc);
    b.n_tracked = (int32_t*)(d + o_nt); b.stats = hb->stats ? (dsdtm_align_stats*)(d + o_st) : nullptr;
    if (int rc = dsdtm_sparse_align_batch_device(ctx, &b, cam, prm, st)) return rc;
    auto download = [&]() -> int {
        HIP_TRY(ctx, hipMemcpyAsync(hb->T_cur_w + (size_t)lo * 12, d + o_tc, n * 96, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipMemcpyAsync(hb->n_tracked + lo, d + o_nt, n * 4, hipMemcpyDeviceToHost, st));
        if (hb->stats) HIP_TRY(ctx, hipMemcpyAsync(hb->stats + lo, d + o_st, n * sizeof(dsdtm_align_stats), hipMemcpyDeviceToHost, st));
        return DSDTM_OK;
    };
    // the downloads overlap nothing when queued before the check, but the check may re-seed and re-run a multi-CU launch whose
    // wait for a partner workgroup ran out (recover_settle): the host arrays then hold the aborted launch's poses and are
    // fetched again
    const unsigned long long rerun0 = ctx->recovered;
    if (int rc = download()) return rc;
    if (int rc = dsdtm_sparse_align_check(ctx, st)) return rc;
    if (ctx->recovered != rerun0) {
        if (int rc = download()) return rc;
        HIP_TRY(ctx, hipStreamSynchronize(st));
    }
    return DSDTM_OK;
}

// A shard that fails half-way still has copies from / into the caller's arrays in flight: they are waited for before
// the error is returned, so the caller may free or reuse its memory as after any other return.
static int sharded_one(dsdtm_ctx* ctx, const dsdtm_batch_desc* hb, const dsdtm_camera* cam, const dsdtm_align_params* prm,
                       int lo, int hi) {
    const int rc = sharded_issue(ctx, hb, cam, prm, lo, hi);
    if (rc != DSDTM_OK) {
        (void)hipStreamSynchronize(ctx->stream);
        recover_forget_stream(ctx, ctx->stream);      // their records point into the staging buffer, which the next call may regrow
    }
    return rc;
}

extern "C" int dsdtm_sparse_align_batch_sharded(dsdtm_ctx* const* ctx, int n_ctx, const dsdtm_batch_desc* hb,
                                                const dsdtm_camera* cam, const dsdtm_align_params* prm) {
    if (!ctx || n_ctx <= 0 || !hb || !cam || !prm) return DSDTM_ERR_INVALID;
    for (int g = 0; g < n_ctx; ++g) if (!ctx[g]) return DSDTM_ERR_INVALID;
    if (hb->n_pairs < 0 || !hb->ref_pyr || !hb->cur_pyr || !hb->T_ref_w || !hb->T_cur_w || !hb->n_tracked ||
        (hb->max_features > 0 && (!hb->px_xy || !hb->bearing || !hb->p_world || !hb->initial))) {
        set_err(ctx[0], "sharded batch: NULL host pointers"); return DSDTM_ERR_INVALID;
    }
    for (int g = 0; g < n_ctx; ++g)
        for (int h = g + 1; h < n_ctx; ++h)
            if (ctx[g] == ctx[h]) { set_err(ctx[0], "sharded batch: a context appears twice (one context per shard)"); return DSDTM_ERR_INVALID; }
    // the descriptor is checked BEFORE any shard sizes its staging from it or a thread is started
    if (hb->max_features < 0 || hb->max_features > 32767 || hb->levels <= 0 || hb->levels > DSDTM_MAX_LEVELS ||
        hb->pyr_pitch == 0 || (hb->pyr_pitch & 3) || hb->pyr_pitch > 0xffffffffull) {
        set_err(ctx[0], "sharded batch: bad geometry (max_features 0..32767, levels 1..%d, pyr_pitch a non-zero multiple of 4 below 4 GiB)",
                DSDTM_MAX_LEVELS);
        return DSDTM_ERR_INVALID;
    }
    if (int rc0 = validate_params(ctx[0], prm, hb->levels)) return rc0;
    for (int l = 0; l < hb->levels; ++l) {
        const size_t end = hb->level_offset[l] + (size_t)hb->stride[l] * hb->height[l];
        if (hb->width[l] <= 0 || hb->height[l] <= 0 || hb->stride[l] < hb->width[l] || end > hb->pyr_pitch) {
            set_err(ctx[0], "sharded batch: level %d does not fit inside pyr_pitch", l); return DSDTM_ERR_INVALID;
        }
    }
    std::vector<int> rc(n_ctx, DSDTM_OK);
    std::vector<std::thread> th;
    bool spawn_failed = false;
    try { th.reserve((size_t)n_ctx); } catch (...) { set_err(ctx[0], "sharded batch: out of memory"); return DSDTM_ERR_NOMEM; }
    for (int g = 1; g < n_ctx && !spawn_failed; ++g) {
        int lo, hi;
        dsdtm_shard_range(hb->n_pairs, n_ctx, g, &lo, &hi);
        try {
            th.emplace_back([=, &rc]() { rc[g] = sharded_one(ctx[g], hb, cam, prm, lo, hi); });
        } catch (...) {           // no exception crosses the C boundary: the shards already started are joined below
            spawn_failed = true;
            rc[g] = DSDTM_ERR_NOMEM;
            set_err(ctx[g], "sharded batch: could not start the host thread of shard %d", g);
        }
    }
    if (!spawn_failed) {
        int lo, hi;
        dsdtm_shard_range(hb->n_pairs, n_ctx, 0, &lo, &hi);
        rc[0] = sharded_one(ctx[0], hb, cam, prm, lo, hi);      // the calling thread takes the first shard
    }
    for (auto& t : th) t.join();
    for (int g = 0; g < n_ctx; ++g) if (rc[g] != DSDTM_OK) return rc[g];
    return DSDTM_OK;
}

// ---- the batch STREAMED from host memory: level 0 only, chunked, copy overlapped with compute ---------------
// One shard = pairs [lo, hi) of the host sequence on `ctx`: chunks of `chunk` pairs; per chunk the level-0 images and
// feature columns go up on one of two copy streams, the pyramids are built by the device pyrDown and the chunk is
// aligned on the context's stream behind the copy's event, the results come back behind the alignment — the upload of
// chunk j + 1 overlaps pyramids + alignment + download of chunk j. Device memory holds the whole shard (0.41 MB per
// 640x480 frame: thousands of frames are a few GB of 288).
static int streamed_issue(dsdtm_ctx* ctx, const dsdtm_stream_desc* s, const dsdtm_camera* cam, const dsdtm_align_params* prm,
                          int chunk, int lo, int hi) {
    const int n = hi - lo;
    if (n <= 0) return DSDTM_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const bool chained = s->cur_image == nullptr;
    // device pyramid layout: the one dsdtm_frame_create_from_image uses (levels 64-byte aligned, pitch 256)
    int w[DSDTM_MAX_LEVELS], h[DSDTM_MAX_LEVELS], st[DSDTM_MAX_LEVELS];
    size_t offl[DSDTM_MAX_LEVELS], o = 0;
    for (int l = 0; l < s->levels; ++l) {
        w[l] = l ? (w[l - 1] + 1) / 2 : s->width; h[l] = l ? (h[l - 1] + 1) / 2 : s->height; st[l] = w[l];
        offl[l] = o; o += align_up((size_t)w[l] * h[l], 64);
    }
    const size_t pit = align_up(o, 256), img = (size_t)s->width * s->height, nf = (size_t)s->max_features;
    const int n_frames = chained ? n + 1 : n;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t v = off; off = align_up(off + bytes, 256); return v; };
    const size_t o_ref = take((size_t)n_frames * pit), o_cur = chained ? o_ref + pit : take((size_t)n * pit),
                 o_px = take(n * nf * 8), o_be = take(n * nf * 24), o_pw = take(n * nf * 24), o_in = take(n * nf),
                 o_nf = take(s->n_features ? (size_t)n * 4 : 0), o_tr = take((size_t)n * 96), o_tc = take((size_t)n * 96),
                 o_nt = take((size_t)n * 4), o_st = take(s->stats ? n * sizeof(dsdtm_align_stats) : 0);
    if (off > ctx->d_cap) {
        if (ctx->d_stage) { HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); (void)hipFree(ctx->d_stage); }
        ctx->d_stage = nullptr; ctx->d_cap = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->d_stage, off));
        ctx->d_cap = off;
    }
    uint8_t* d = (uint8_t*)ctx->d_stage;
    for (int i = 0; i < 2; ++i)
        if (!ctx->copy_stream[i]) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->copy_stream[i], hipStreamNonBlocking));
    const int n_chunks = (n + chunk - 1) / chunk;
    while ((int)ctx->copy_events.size() < n_chunks) {
        hipEvent_t e = nullptr;
        HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->copy_events.push_back(e);
    }
    // the copy streams start behind whatever the context's stream still does with the staging buffer
    if (!ctx->copy_fence) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->copy_fence, hipEventDisableTiming));
    HIP_TRY(ctx, hipEventRecord(ctx->copy_fence, ctx->stream));
    for (int i = 0; i < 2; ++i) HIP_TRY(ctx, hipStreamWaitEvent(ctx->copy_stream[i], ctx->copy_fence, 0));
    // level 0 of frames [f0, f1) of host array `src` (frame 0 = the shard's first) -> device pyramids at d + base
    auto up_images = [&](hipStream_t cs, const uint8_t* src, size_t base, int f0, int f1) -> hipError_t {
        if (f1 <= f0) return hipSuccess;
        if (s->row_stride == s->width)      // whole images: ONE strided copy (an image is a "row" of W*H bytes)
            return hipMemcpy2DAsync(d + base + (size_t)f0 * pit, pit, src + (size_t)f0 * s->image_pitch, s->image_pitch, img,
                                    (size_t)(f1 - f0), hipMemcpyHostToDevice, cs);
        for (int f = f0; f < f1; ++f) {     // padded rows: one 2-D copy per image
            const hipError_t e = hipMemcpy2DAsync(d + base + (size_t)f * pit, (size_t)s->width, src + (size_t)f * s->image_pitch,
                                                  (size_t)s->row_stride, (size_t)s->width, (size_t)s->height, hipMemcpyHostToDevice, cs);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    };
    auto up = [&](hipStream_t cs, size_t dst, const void* src, size_t bytes) {
        return bytes ? hipMemcpyAsync(d + dst, src, bytes, hipMemcpyHostToDevice, cs) : hipSuccess;
    };
    const uint8_t* ref_h = s->ref_image + (size_t)lo * s->image_pitch;
    const uint8_t* cur_h = chained ? nullptr : s->cur_image + (size_t)lo * s->image_pitch;
    dsdtm_batch_desc b;
    memset(&b, 0, sizeof b);
    b.max_features = s->max_features; b.levels = s->levels; b.pyr_pitch = pit;
    for (int l = 0; l < s->levels; ++l) { b.width[l] = w[l]; b.height[l] = h[l]; b.stride[l] = st[l]; b.level_offset[l] = offl[l]; }
    hipStream_t comp = ctx->stream;
    for (int j = 0; j < n_chunks; ++j) {
        const int p0 = j * chunk, p1 = p0 + chunk < n ? p0 + chunk : n;      // pairs of this chunk (shard-relative)
        const size_t np = (size_t)(p1 - p0);
        hipStream_t cs = ctx->copy_stream[j & 1];
        // frames that arrive with this chunk — chained: the `cur` frames of its pairs (+ frame 0 with the first chunk)
        const int f0 = chained ? (j == 0 ? 0 : p0 + 1) : p0, f1 = chained ? p1 + 1 : p1;
        HIP_TRY(ctx, up_images(cs, ref_h, o_ref, f0, f1));
        if (!chained) HIP_TRY(ctx, up_images(cs, cur_h, o_cur, p0, p1));
        const size_t g0 = (size_t)(lo + p0);                                   // first pair of the chunk in the host arrays
        HIP_TRY(ctx, up(cs, o_px + p0 * nf * 8, s->px_xy + g0 * nf * 2, np * nf * 8));
        HIP_TRY(ctx, up(cs, o_be + p0 * nf * 24, s->bearing + g0 * nf * 3, np * nf * 24));
        HIP_TRY(ctx, up(cs, o_pw + p0 * nf * 24, s->p_world + g0 * nf * 3, np * nf * 24));
        HIP_TRY(ctx, up(cs, o_in + p0 * nf, s->initial + g0 * nf, np * nf));
        if (s->n_features) HIP_TRY(ctx, up(cs, o_nf + (size_t)p0 * 4, s->n_features + g0, np * 4));
        HIP_TRY(ctx, up(cs, o_tr + (size_t)p0 * 96, s->T_ref_w + g0 * 12, np * 96));
        HIP_TRY(ctx, up(cs, o_tc + (size_t)p0 * 96, s->T_cur_w + g0 * 12, np * 96));
        HIP_TRY(ctx, hipEventRecord(ctx->copy_events[j], cs));
        HIP_TRY(ctx, hipStreamWaitEvent(comp, ctx->copy_events[j], 0));
        // Frame::ComputeImagePyramid (src/Frame.cpp:74-81) for the frames that just arrived, on the device
        if (int rc = dsdtm_pyrdown_batch_device(ctx, d + o_ref + (size_t)f0 * pit, pit, f1 - f0, s->levels, w, h, st, offl, comp)) return rc;
        if (!chained)
            if (int rc = dsdtm_pyrdown_batch_device(ctx, d + o_cur + (size_t)p0 * pit, pit, p1 - p0, s->levels, w, h, st, offl, comp)) return rc;
        b.n_pairs = p1 - p0;
        b.ref_pyr = d + o_ref + (size_t)p0 * pit; b.cur_pyr = d + o_cur + (size_t)p0 * pit;
        b.px_xy = (const float*)(d + o_px + p0 * nf * 8); b.bearing = (const double*)(d + o_be + p0 * nf * 24);
        b.p_world = (const double*)(d + o_pw + p0 * nf * 24); b.initial = d + o_in + p0 * nf;
        b.n_features = s->n_features ? (const int32_t*)(d + o_nf + (size_t)p0 * 4) : nullptr;
        b.T_ref_w = (const double*)(d + o_tr + (size_t)p0 * 96); b.T_cur_w = (double*)(d + o_tc + (size_t)p0 * 96);
        b.n_tracked = (int32_t*)(d + o_nt + (size_t)p0 * 4);
        b.stats = s->stats ? (dsdtm_align_stats*)(d + o_st + p0 * sizeof(dsdtm_align_stats)) : nullptr;
        if (int rc = dsdtm_sparse_align_batch_device(ctx, &b, cam, prm, comp)) return rc;
        // (multi-CU shapes: results are final after the check below; a chunk's download is repeated there if it was re-run)
        HIP_TRY(ctx, hipMemcpyAsync(s->T_cur_w + g0 * 12, d + o_tc + (size_t)p0 * 96, np * 96, hipMemcpyDeviceToHost, comp));
        HIP_TRY(ctx, hipMemcpyAsync(s->n_tracked + g0, d + o_nt + (size_t)p0 * 4, np * 4, hipMemcpyDeviceToHost, comp));
        if (s->stats) HIP_TRY(ctx, hipMemcpyAsync(s->stats + g0, d + o_st + p0 * sizeof(dsdtm_align_stats), np * sizeof(dsdtm_align_stats), hipMemcpyDeviceToHost, comp));
    }
    const unsigned long long rerun0 = ctx->recovered;
    if (int rc = dsdtm_sparse_align_check(ctx, comp)) return rc;
    if (ctx->recovered != rerun0) {          // a chunk was re-run on the one-CU kernels: fetch everything again
        HIP_TRY(ctx, hipMemcpyAsync(s->T_cur_w + (size_t)lo * 12, d + o_tc, (size_t)n * 96, hipMemcpyDeviceToHost, comp));
        HIP_TRY(ctx, hipMemcpyAsync(s->n_tracked + lo, d + o_nt, (size_t)n * 4, hipMemcpyDeviceToHost, comp));
        if (s->stats) HIP_TRY(ctx, hipMemcpyAsync(s->stats + lo, d + o_st, n * sizeof(dsdtm_align_stats), hipMemcpyDeviceToHost, comp));
        HIP_TRY(ctx, hipStreamSynchronize(comp));
    }
    return DSDTM_OK;
}

// As sharded_one: no copy touches the caller's arrays after an error return.
static int streamed_one(dsdtm_ctx* ctx, const dsdtm_stream_desc* s, const dsdtm_camera* cam, const dsdtm_align_params* prm,
                        int chunk, int lo, int hi) {
    const int rc = streamed_issue(ctx, s, cam, prm, chunk, lo, hi);
    if (rc != DSDTM_OK) {
        for (int i = 0; i < 2; ++i)
            if (ctx->copy_stream[i]) (void)hipStreamSynchronize(ctx->copy_stream[i]);
        (void)hipStreamSynchronize(ctx->stream);
        recover_forget_stream(ctx, ctx->stream);      // as sharded_one
    }
    return rc;
}

extern "C" int dsdtm_sparse_align_batch_streamed(dsdtm_ctx* const* ctx, int n_ctx, const dsdtm_stream_desc* s, int chunk_pairs,
                                                 const dsdtm_camera* cam, const dsdtm_align_params* prm) {
    if (!ctx || n_ctx <= 0 || !s || !cam || !prm) return DSDTM_ERR_INVALID;
    for (int g = 0; g < n_ctx; ++g) if (!ctx[g]) return DSDTM_ERR_INVALID;
    for (int g = 0; g < n_ctx; ++g)
        for (int k = g + 1; k < n_ctx; ++k)
            if (ctx[g] == ctx[k]) { set_err(ctx[0], "streamed batch: a context appears twice (one context per shard)"); return DSDTM_ERR_INVALID; }
    if (s->n_pairs < 0 || !s->ref_image || !s->T_ref_w || !s->T_cur_w || !s->n_tracked ||
        (s->max_features > 0 && (!s->px_xy || !s->bearing || !s->p_world || !s->initial))) {
        set_err(ctx[0], "streamed batch: NULL host pointers"); return DSDTM_ERR_INVALID;
    }
    if (s->max_features < 0 || s->max_features > 32767 || s->levels <= 0 || s->levels > DSDTM_MAX_LEVELS || s->width <= 0 ||
        s->height <= 0 || s->row_stride < s->width || s->image_pitch < (size_t)s->row_stride * (s->height - 1) + s->width) {
        set_err(ctx[0], "streamed batch: bad geometry"); return DSDTM_ERR_INVALID;
    }
    if (int rc0 = validate_params(ctx[0], prm, s->levels)) return rc0;
    const int chunk = chunk_pairs > 0 ? chunk_pairs : 128;
    std::vector<int> rc(n_ctx, DSDTM_OK);
    std::vector<std::thread> th;
    bool spawn_failed = false;
    try { th.reserve((size_t)n_ctx); } catch (...) { set_err(ctx[0], "streamed batch: out of memory"); return DSDTM_ERR_NOMEM; }
    for (int g = 1; g < n_ctx && !spawn_failed; ++g) {
        int lo, hi;
        dsdtm_shard_range(s->n_pairs, n_ctx, g, &lo, &hi);
        try {
            th.emplace_back([=, &rc]() { rc[g] = streamed_one(ctx[g], s, cam, prm, chunk, lo, hi); });
        } catch (...) {
            spawn_failed = true;
            rc[g] = DSDTM_ERR_NOMEM;
            set_err(ctx[g], "streamed batch: could not start the host thread of shard %d", g);
        }
    }
    if (!spawn_failed) {
        int lo, hi;
        dsdtm_shard_range(s->n_pairs, n_ctx, 0, &lo, &hi);
        rc[0] = streamed_one(ctx[0], s, cam, prm, chunk, lo, hi);
    }
    for (auto& t : th) t.join();
    for (int g = 0; g < n_ctx; ++g) if (rc[g] != DSDTM_OK) return rc[g];
    return DSDTM_OK;
}

// ---- host staging helpers ---------------------------------------------------------------
struct PackedPyr {
    int levels;
    int w[DSDTM_MAX_LEVELS], h[DSDTM_MAX_LEVELS];
    size_t off[DSDTM_MAX_LEVELS];
    size_t bytes;
};

static int plan_pyramid(dsdtm_ctx* ctx, const dsdtm_pyramid* p, PackedPyr* out) {
    if (!p || p->levels <= 0 || p->levels > DSDTM_MAX_LEVELS) { set_err(ctx, "bad pyramid"); return DSDTM_ERR_INVALID; }
    size_t off = 0;
    out->levels = p->levels;
    for (int l = 0; l < p->levels; ++l) {
        if (!p->data[l] || p->width[l] <= 0 || p->height[l] <= 0 || p->stride[l] < p->width[l]) {
            set_err(ctx, "bad pyramid level %d", l); return DSDTM_ERR_INVALID;
        }
        out->w[l] = p->width[l]; out->h[l] = p->height[l]; out->off[l] = off;
        off += align_up((size_t)p->width[l] * p->height[l], 64);
    }
    out->bytes = off;
    return DSDTM_OK;
}

static void pack_pyramid(const dsdtm_pyramid* p, const PackedPyr& pl, uint8_t* dst) {
    for (int l = 0; l < pl.levels; ++l) {
        uint8_t* d = dst + pl.off[l];
        const uint8_t* s = p->data[l];
        if (p->stride[l] == pl.w[l]) memcpy(d, s, (size_t)pl.w[l] * pl.h[l]);
        else for (int y = 0; y < pl.h[l]; ++y) memcpy(d + (size_t)y * pl.w[l], s + (size_t)y * p->stride[l], pl.w[l]);
    }
}

// One alignment on the context's stream. The pyramids are either packed into the staging buffer
// together with the features (ref/cur given, dev_ref/dev_cur null) or already on the device.
static int sparse_align_one(dsdtm_ctx* ctx, const PackedPyr& pl, const dsdtm_pyramid* ref, const dsdtm_pyramid* cur,
                            const uint8_t* dev_ref, const uint8_t* dev_cur, size_t dev_pitch,
                            const dsdtm_camera* cam, const float* px_xy, const double* bearing,
                            const double* p_world, const uint8_t* initial, int n_features,
                            const double T_ref_w[12], double T_cur_w[12], const dsdtm_align_params* prm,
                            int* n_tracked, dsdtm_align_stats* stats) {
    if (int rc = validate_params(ctx, prm, pl.levels)) return rc;
    // Run() (:34-38): too few features -> 0, pose untouched. Decided on the host: no launch needed.
    if (stats) memset(stats, 0, sizeof *stats);
    *n_tracked = 0;
    if (n_features < prm->min_fts || prm->max_level - 1 < prm->min_level) return DSDTM_OK;

    const bool staged_pyr = dev_ref == nullptr;
    const size_t nf = (size_t)n_features;
    const size_t pitch = staged_pyr ? align_up(pl.bytes, 256) : dev_pitch;
    size_t o = 0;
    const size_t o_ref = o; if (staged_pyr) o += pitch;
    const size_t o_cur = o; if (staged_pyr) o += pitch;
    const size_t o_bear = o; o += align_up(nf * 24, 256);
    const size_t o_pw = o; o += align_up(nf * 24, 256);
    const size_t o_tr = o; o += 256;
    const size_t o_px = o; o += align_up(nf * 8, 256);
    const size_t o_ini = o; o += align_up(nf, 256);
    const size_t in_bytes = o;
    const size_t o_tc = o; o += 256;          // in (seed) and out
    const size_t o_nt = o; o += 256;
    const size_t o_st = o; o += align_up(sizeof(dsdtm_align_stats), 256);
    const size_t total = o;
    if (int rc = ensure_stage(ctx, total)) return rc;
    uint8_t* h = (uint8_t*)ctx->h_pinned;
    uint8_t* d = (uint8_t*)ctx->d_stage;
    if (staged_pyr) {
        pack_pyramid(ref, pl, h + o_ref);
        pack_pyramid(cur, pl, h + o_cur);
    }
    memcpy(h + o_bear, bearing, nf * 24);
    memcpy(h + o_pw, p_world, nf * 24);
    memcpy(h + o_tr, T_ref_w, 96);
    memcpy(h + o_px, px_xy, nf * 8);
    memcpy(h + o_ini, initial, nf);
    memcpy(h + o_tc, T_cur_w, 96);
    // Resident frames: the kernel reads features and poses straight from the pinned staging block (once per
    // pair, ~17 KB over PCIe) and writes its few results there: no H2D / D2H copy operations around the launch.
    const bool zero_copy_enabled = !options().no_zero_copy;
    const bool zero_copy = zero_copy_enabled && !staged_pyr;
    if (zero_copy) {
        void* hd = nullptr;
        HIP_TRY(ctx, hipHostGetDevicePointer(&hd, ctx->h_pinned, 0));
        d = (uint8_t*)hd;
        memset(h + o_nt, 0, total - o_nt);
    } else {
        HIP_TRY(ctx, hipMemcpyAsync(d, h, in_bytes + 256, hipMemcpyHostToDevice, ctx->stream));
    }

    dsdtm_batch_desc b;
    memset(&b, 0, sizeof b);
    b.n_pairs = 1; b.max_features = n_features; b.levels = pl.levels;
    for (int l = 0; l < pl.levels; ++l) { b.width[l] = pl.w[l]; b.height[l] = pl.h[l]; b.stride[l] = pl.w[l]; b.level_offset[l] = pl.off[l]; }
    b.pyr_pitch = pitch;
    b.ref_pyr = staged_pyr ? d + o_ref : dev_ref; b.cur_pyr = staged_pyr ? d + o_cur : dev_cur;
    b.px_xy = (const float*)(d + o_px);
    b.bearing = (const double*)(d + o_bear); b.p_world = (const double*)(d + o_pw); b.initial = d + o_ini;
    b.n_features = nullptr; b.T_ref_w = (const double*)(d + o_tr); b.T_cur_w = (double*)(d + o_tc);
    b.n_tracked = (int32_t*)(d + o_nt); b.stats = (dsdtm_align_stats*)(d + o_st);
    // Synchronous single-pair entry: its own timeout word, settled right here. A team launch whose wait for a partner
    // workgroup ran out (a foreign load on the device) is re-run on the one-CU kernels from the seed pose, which is
    // still in the pinned block: Run never fails for scheduling reasons (src/Sprase_ImageAlign.cpp:29-60).
    volatile unsigned* h_flag = ctx->h_flags + dsdtm_ctx::FLAG_SINGLE;
    *h_flag = 0;
    LaunchMode mode;
    mode.single = true;
    for (int attempt = 0; attempt < 2; ++attempt) {
        bool multi_cu = false;
        if (int rc_launch = launch_batch(ctx, &b, cam, prm, ctx->stream, mode, &multi_cu)) return rc_launch;
        if (!zero_copy) HIP_TRY(ctx, hipMemcpyAsync(h + o_tc, d + o_tc, total - o_tc, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (!*h_flag) break;
        *h_flag = 0;
        if (!multi_cu || attempt == 1 || options().no_recover) {
            // (a pair that was stopped may have left its counter behind: the context stream's words)
            for (int i = 0; i < dsdtm_ctx::MAX_STREAMS; ++i)
                if (ctx->rings[i].used && ctx->rings[i].stream == ctx->stream)
                    (void)hipMemset(ctx->d_counter + i * dsdtm_ctx::COUNTERS_PER_STREAM, 0, sizeof(unsigned) * dsdtm_ctx::COUNTERS_PER_STREAM);
            set_err(ctx, multi_cu ? "sparse-align kernel: a wait for a partner workgroup timed out (re-run disabled: no_recover)"
                                  : "sparse-align kernel: intra-workgroup hand-over timed out");
            return DSDTM_ERR_HIP;
        }
        memcpy(h + o_tc, T_cur_w, 96);                      // the seed again
        if (zero_copy) memset(h + o_nt, 0, total - o_nt);
        else HIP_TRY(ctx, hipMemcpyAsync(d + o_tc, h + o_tc, 96, hipMemcpyHostToDevice, ctx->stream));
        mode.one_cu = true;
        ctx->recovered += 1;
    }
    memcpy(T_cur_w, h + o_tc, 96);
    *n_tracked = *(const int32_t*)(h + o_nt);
    if (stats) memcpy(stats, h + o_st, sizeof *stats);
    return DSDTM_OK;
}

extern "C" int dsdtm_sparse_align(dsdtm_ctx* ctx, const dsdtm_pyramid* ref, const dsdtm_pyramid* cur,
                                  const dsdtm_camera* cam, const float* px_xy, const double* bearing,
                                  const double* p_world, const uint8_t* initial, int n_features,
                                  const double T_ref_w[12], double T_cur_w[12], const dsdtm_align_params* prm,
                                  int* n_tracked, dsdtm_align_stats* stats) {
    if (!ctx) return DSDTM_ERR_INVALID;
    if (!ref || !cur || !cam || !T_ref_w || !T_cur_w || !n_tracked || n_features < 0 ||
        (n_features > 0 && (!px_xy || !bearing || !p_world || !initial))) {
        set_err(ctx, "NULL argument"); return DSDTM_ERR_INVALID;
    }
    PackedPyr pr, pc;
    if (int rc = plan_pyramid(ctx, ref, &pr)) return rc;
    if (int rc = plan_pyramid(ctx, cur, &pc)) return rc;
    if (pr.levels != pc.levels) { set_err(ctx, "ref/cur pyramids differ in level count"); return DSDTM_ERR_INVALID; }
    for (int l = 0; l < pr.levels; ++l)
        if (pr.w[l] != pc.w[l] || pr.h[l] != pc.h[l]) { set_err(ctx, "ref/cur level %d differ in size", l); return DSDTM_ERR_INVALID; }
    return sparse_align_one(ctx, pr, ref, cur, nullptr, nullptr, 0, cam, px_xy, bearing, p_world, initial, n_features,
                            T_ref_w, T_cur_w, prm, n_tracked, stats);
}

// ---- frames that stay on the device --------------------------------------------------------------
struct dsdtm_frame {
    dsdtm_ctx* owner;    // compared only, never dereferenced through the frame (a frame may outlive its context)
    int device;
    uint8_t* d;          // packed pyramid (the layout of plan_pyramid), its own allocation
    size_t pitch;
    PackedPyr pl;
};

static int frame_alloc(dsdtm_ctx* ctx, const PackedPyr& pl, dsdtm_frame** out) {
    dsdtm_frame* f = new (std::nothrow) dsdtm_frame();
    if (!f) return DSDTM_ERR_NOMEM;
    f->owner = ctx; f->device = ctx->device; f->pl = pl; f->pitch = align_up(pl.bytes, 256); f->d = nullptr;
    for (size_t i = 0; i < ctx->frame_pool.size(); ++i)
        if (ctx->frame_pool[i].pitch == f->pitch) {             // a destroyed frame's buffer of this size (its users have drained:
            f->d = ctx->frame_pool[i].d;                        // every entry that takes a dsdtm_frame waits for its stream)
            ctx->frame_pool.erase(ctx->frame_pool.begin() + (long)i);
            *out = f;
            return DSDTM_OK;
        }
    if (hipSetDevice(ctx->device) != hipSuccess || hipMalloc((void**)&f->d, f->pitch) != hipSuccess) {
        set_err(ctx, "hipMalloc of a %zu-byte frame failed", f->pitch);
        delete f;
        return DSDTM_ERR_NOMEM;
    }
    *out = f;
    return DSDTM_OK;
}

extern "C" int dsdtm_frame_create(dsdtm_ctx* ctx, const dsdtm_pyramid* pyr, dsdtm_frame** out) {
    if (!ctx || !out) return DSDTM_ERR_INVALID;
    *out = nullptr;
    PackedPyr pl;
    if (int rc = plan_pyramid(ctx, pyr, &pl)) return rc;
    if (int rc = ensure_stage(ctx, pl.bytes)) return rc;
    dsdtm_frame* f = nullptr;
    if (int rc = frame_alloc(ctx, pl, &f)) return rc;
    pack_pyramid(pyr, pl, (uint8_t*)ctx->h_pinned);
    hipError_t e = hipMemcpyAsync(f->d, ctx->h_pinned, pl.bytes, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);      // the pinned buffer is reused by the next call
    if (e != hipSuccess) { set_err(ctx, "frame upload failed: %s", hipGetErrorString(e)); dsdtm_frame_destroy(ctx, f); return DSDTM_ERR_HIP; }
    *out = f;
    return DSDTM_OK;
}

extern "C" int dsdtm_frame_create_from_image(dsdtm_ctx* ctx, const uint8_t* level0, int width, int height, int stride,
                                             int levels, dsdtm_frame** out) {
    if (!ctx || !out) return DSDTM_ERR_INVALID;
    *out = nullptr;
    if (!level0 || width <= 0 || height <= 0 || stride < width || levels <= 0 || levels > DSDTM_MAX_LEVELS) {
        set_err(ctx, "bad argument"); return DSDTM_ERR_INVALID;
    }
    PackedPyr pl;
    int st[DSDTM_MAX_LEVELS];
    pl.levels = levels;
    size_t o = 0;
    for (int l = 0; l < levels; ++l) {
        pl.w[l] = l ? (pl.w[l - 1] + 1) / 2 : width; pl.h[l] = l ? (pl.h[l - 1] + 1) / 2 : height; st[l] = pl.w[l];
        pl.off[l] = o; o += align_up((size_t)pl.w[l] * pl.h[l], 64);
    }
    pl.bytes = o;
    if (int rc = ensure_stage(ctx, (size_t)width * height)) return rc;
    dsdtm_frame* f = nullptr;
    if (int rc = frame_alloc(ctx, pl, &f)) return rc;
    uint8_t* hp = (uint8_t*)ctx->h_pinned;
    if (stride == width) memcpy(hp, level0, (size_t)width * height);
    else for (int y = 0; y < height; ++y) memcpy(hp + (size_t)y * width, level0 + (size_t)y * stride, width);
    // level 0 enters through a kernel on the compute stream (ingest_kernel, pyrdown.hip): no copy operation in front of the
    // pyramid kernel — the hand-over from the copy engine to the compute queue costs more than the 8 us of the copy itself
    void* hpd = nullptr;
    hipError_t e = hipHostGetDevicePointer(&hpd, hp, 0);
    if (e == hipSuccess) e = ingest_launch(hpd, f->d, (size_t)width * height, nullptr, nullptr, 0, ctx->stream);
    int rc = DSDTM_OK;
    if (e == hipSuccess) rc = dsdtm_pyrdown_batch_device(ctx, f->d, f->pitch, 1, levels, pl.w, pl.h, st, pl.off, ctx->stream);
    if (e == hipSuccess && rc == DSDTM_OK) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess || rc != DSDTM_OK) {
        if (e != hipSuccess) { set_err(ctx, "frame upload failed: %s", hipGetErrorString(e)); rc = DSDTM_ERR_HIP; }
        dsdtm_frame_destroy(ctx, f);
        return rc;
    }
    *out = f;
    return DSDTM_OK;
}

extern "C" void dsdtm_frame_destroy(dsdtm_ctx* ctx, dsdtm_frame* f) {
    if (!f) return;
    // its own, LIVE context: the buffer goes to that context's pool (no hipFree, which would wait for the whole device)
    if (ctx && f->d && ctx_is_live(ctx) && f->owner == ctx && ctx->frame_pool.size() < dsdtm_ctx::FRAME_POOL) {
        try { ctx->frame_pool.push_back(dsdtm_ctx::PooledFrame{f->pitch, f->d}); f->d = nullptr; } catch (...) {}
        if (!f->d) { delete f; return; }
    }
    (void)hipSetDevice(f->device);
    if (f->d) (void)hipFree(f->d);       // hipFree waits for the device's pending work
    delete f;
}

extern "C" int dsdtm_sparse_align_frames(dsdtm_ctx* ctx, const dsdtm_frame* ref, const dsdtm_frame* cur,
                                         const dsdtm_camera* cam, const float* px_xy, const double* bearing,
                                         const double* p_world, const uint8_t* initial, int n_features,
                                         const double T_ref_w[12], double T_cur_w[12], const dsdtm_align_params* prm,
                                         int* n_tracked, dsdtm_align_stats* stats) {
    if (!ctx) return DSDTM_ERR_INVALID;
    if (!ref || !cur || !cam || !T_ref_w || !T_cur_w || !n_tracked || n_features < 0 ||
        (n_features > 0 && (!px_xy || !bearing || !p_world || !initial))) {
        set_err(ctx, "NULL argument"); return DSDTM_ERR_INVALID;
    }
    if (ref->owner != ctx || cur->owner != ctx) { set_err(ctx, "frame belongs to another context"); return DSDTM_ERR_INVALID; }
    if (ref->pl.levels != cur->pl.levels) { set_err(ctx, "ref/cur pyramids differ in level count"); return DSDTM_ERR_INVALID; }
    for (int l = 0; l < ref->pl.levels; ++l)
        if (ref->pl.w[l] != cur->pl.w[l] || ref->pl.h[l] != cur->pl.h[l]) { set_err(ctx, "ref/cur level %d differ in size", l); return DSDTM_ERR_INVALID; }
    return sparse_align_one(ctx, ref->pl, nullptr, nullptr, ref->d, cur->d, ref->pitch, cam, px_xy, bearing, p_world, initial,
                            n_features, T_ref_w, T_cur_w, prm, n_tracked, stats);
}

// ---- feature detector (per-cell part) ----------------------------------------------------------
static int detect_cells_one(dsdtm_ctx* ctx, const PackedPyr& pl, const dsdtm_pyramid* host_pyr, const uint8_t* dev_pyr,
                            const uint8_t* grid_occupied, const dsdtm_detect_params* prm, float* cell_score,
                            int32_t* cell_x, int32_t* cell_y, int32_t* cell_level) {
    if (!prm || !cell_score || !cell_x || !cell_y || !cell_level) { set_err(ctx, "NULL argument"); return DSDTM_ERR_INVALID; }
    if (prm->cell_size <= 0 || prm->grid_cols <= 0 || prm->grid_rows <= 0 || prm->levels <= 0 || prm->levels > pl.levels ||
        prm->barrier < 0 || prm->barrier > 254 || !(prm->detection_threshold >= 0.0f) ||
        (long long)prm->grid_cols * prm->grid_rows > (1 << 24)) {
        set_err(ctx, "bad detector parameters"); return DSDTM_ERR_INVALID;
    }
    for (int l = 0; l < prm->levels; ++l)
        if (pl.w[l] >= (1 << 14) || pl.h[l] >= (1 << 14)) { set_err(ctx, "level %d larger than 16383 pixels", l); return DSDTM_ERR_INVALID; }
    const size_t G = (size_t)prm->grid_cols * prm->grid_rows;
    const size_t pitch = align_up(pl.bytes, 256);
    size_t o = 0;
    const size_t o_pyr = o; if (!dev_pyr) o += pitch;
    const size_t o_occ = o; o += align_up(G, 256);
    const size_t in_bytes = o;
    const size_t o_key = o; o += align_up(G * 8, 256);
    const size_t o_score = o; o += pitch;
    const size_t total = o;
    if (int rc = ensure_stage(ctx, total)) return rc;
    uint8_t* h = (uint8_t*)ctx->h_pinned;
    uint8_t* d = (uint8_t*)ctx->d_stage;
    if (!dev_pyr) pack_pyramid(host_pyr, pl, h + o_pyr);
    if (grid_occupied) memcpy(h + o_occ, grid_occupied, G); else memset(h + o_occ, 0, G);
    HIP_TRY(ctx, hipMemcpyAsync(d, h, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(d + o_key, 0, G * 8, ctx->stream));
    DetectArgs a;
    memset(&a, 0, sizeof a);
    a.pyr = dev_pyr ? dev_pyr : d + o_pyr; a.score = d + o_score; a.cell_key = (unsigned long long*)(d + o_key);
    a.occupied = d + o_occ;
    a.pyr_pitch = pitch; a.n_frames = 1; a.levels = prm->levels;
    a.cell_size = prm->cell_size; a.grid_cols = prm->grid_cols; a.grid_rows = prm->grid_rows; a.barrier = prm->barrier;
    a.detection_threshold = prm->detection_threshold;
    for (int l = 0; l < prm->levels; ++l) { a.lv[l].w = pl.w[l]; a.lv[l].h = pl.h[l]; a.lv[l].stride = pl.w[l]; a.lv[l].off = (uint32_t)pl.off[l]; }
    HIP_TRY(ctx, detect_launch(a, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(h + o_key, d + o_key, G * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const unsigned long long* keys = (const unsigned long long*)(h + o_key);
    for (size_t k = 0; k < G; ++k) {
        if (!keys[k]) { cell_score[k] = prm->detection_threshold; cell_x[k] = 0; cell_y[k] = 0; cell_level[k] = 0; continue; }   // :74
        const uint32_t bits = (uint32_t)(keys[k] >> 32), order = 0xffffffffu - (uint32_t)keys[k];
        const int L = (int)(order >> 28), y = (int)((order >> 14) & 0x3fffu), x = (int)(order & 0x3fffu);
        memcpy(&cell_score[k], &bits, 4);
        cell_x[k] = x << L; cell_y[k] = y << L; cell_level[k] = L;                                  // :106
    }
    return DSDTM_OK;
}

extern "C" int dsdtm_detect_cells(dsdtm_ctx* ctx, const dsdtm_pyramid* pyr, const uint8_t* grid_occupied,
                                  const dsdtm_detect_params* prm, float* cell_score, int32_t* cell_x, int32_t* cell_y,
                                  int32_t* cell_level) {
    if (!ctx) return DSDTM_ERR_INVALID;
    PackedPyr pl;
    if (int rc = plan_pyramid(ctx, pyr, &pl)) return rc;
    return detect_cells_one(ctx, pl, pyr, nullptr, grid_occupied, prm, cell_score, cell_x, cell_y, cell_level);
}

extern "C" int dsdtm_detect_cells_frame(dsdtm_ctx* ctx, const dsdtm_frame* frame, const uint8_t* grid_occupied,
                                        const dsdtm_detect_params* prm, float* cell_score, int32_t* cell_x, int32_t* cell_y,
                                        int32_t* cell_level) {
    if (!ctx) return DSDTM_ERR_INVALID;
    if (!frame) { set_err(ctx, "NULL frame"); return DSDTM_ERR_INVALID; }
    if (frame->owner != ctx) { set_err(ctx, "frame belongs to another context"); return DSDTM_ERR_INVALID; }
    return detect_cells_one(ctx, frame->pl, nullptr, frame->d, grid_occupied, prm, cell_score, cell_x, cell_y, cell_level);
}

// The image part of Feature_detector::detect for n_frames packed device pyramids in one go (independent sequences,
// keyframes of an offline run): every pointer is a device pointer, nothing is copied, asynchronous on hip_stream.
extern "C" int dsdtm_detect_cells_batch_device(dsdtm_ctx* ctx, const uint8_t* pyr, size_t pyr_pitch, int n_frames, int levels,
                                               const int* width, const int* height, const int* stride, const size_t* level_offset,
                                               const uint8_t* grid_occupied, const dsdtm_detect_params* prm,
                                               uint8_t* score_scratch, unsigned long long* key_scratch,
                                               float* cell_score, int32_t* cell_x, int32_t* cell_y, int32_t* cell_level,
                                               void* hip_stream) {
    if (!ctx) return DSDTM_ERR_INVALID;
    if (!pyr || !width || !height || !stride || !level_offset || !prm || !score_scratch || !key_scratch || !cell_score || !cell_x ||
        !cell_y || !cell_level || n_frames < 0 || levels <= 0 || levels > DSDTM_MAX_LEVELS) { set_err(ctx, "bad argument"); return DSDTM_ERR_INVALID; }
    if (prm->cell_size <= 0 || prm->grid_cols <= 0 || prm->grid_rows <= 0 || prm->levels <= 0 || prm->levels > levels ||
        prm->barrier < 0 || prm->barrier > 254 || !(prm->detection_threshold >= 0.0f) ||
        (long long)prm->grid_cols * prm->grid_rows > (1 << 24)) { set_err(ctx, "bad detector parameters"); return DSDTM_ERR_INVALID; }
    // the kernels fetch pyramid and score rows as dwords and index frames through the grid's z dimension
    if ((pyr_pitch & 3) || (((size_t)pyr) & 3) || (((size_t)score_scratch) & 3)) {
        set_err(ctx, "detector batch: pyramid base, score scratch and pyr_pitch must be 4-byte aligned"); return DSDTM_ERR_INVALID;
    }
    if ((long long)n_frames * prm->levels > 65535) {
        set_err(ctx, "detector batch: n_frames * levels = %lld exceeds 65535 (one launch indexes frames through grid.z): split the batch",
                (long long)n_frames * prm->levels);
        return DSDTM_ERR_INVALID;
    }
    if (n_frames == 0) return DSDTM_OK;
    DetectArgs a;
    memset(&a, 0, sizeof a);
    for (int l = 0; l < prm->levels; ++l) {
        if (width[l] <= 0 || height[l] <= 0 || stride[l] < width[l] || width[l] >= (1 << 14) || height[l] >= (1 << 14) ||
            level_offset[l] + (size_t)stride[l] * height[l] > pyr_pitch) { set_err(ctx, "level %d does not fit", l); return DSDTM_ERR_INVALID; }
        a.lv[l].w = width[l]; a.lv[l].h = height[l]; a.lv[l].stride = stride[l]; a.lv[l].off = (uint32_t)level_offset[l];
    }
    const size_t G = (size_t)prm->grid_cols * prm->grid_rows;
    a.pyr = pyr; a.score = score_scratch; a.cell_key = key_scratch; a.occupied = grid_occupied;
    a.cell_score = cell_score; a.cell_x = cell_x; a.cell_y = cell_y; a.cell_level = cell_level;
    a.pyr_pitch = pyr_pitch; a.n_frames = n_frames; a.levels = prm->levels;
    a.cell_size = prm->cell_size; a.grid_cols = prm->grid_cols; a.grid_rows = prm->grid_rows; a.barrier = prm->barrier;
    a.detection_threshold = prm->detection_threshold;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemsetAsync(key_scratch, 0, (size_t)n_frames * G * 8, (hipStream_t)hip_stream));
    HIP_TRY(ctx, detect_launch(a, (hipStream_t)hip_stream));
    return DSDTM_OK;
}

#ifdef DSDTM_DIAG
// Diagnostic (tests): the FAST-10 score map and the non-max survivors of ONE 8-bit image, as the detector's
// two passes produce them on the device. score/keep: width*height bytes each.
extern "C" int dsdtm_debug_fast10(dsdtm_ctx* ctx, const uint8_t* img, int width, int height, int stride, int barrier,
                                  uint8_t* score, uint8_t* keep) {
    if (!ctx || !img || !score || !keep || width <= 0 || height <= 0 || stride < width || width >= (1 << 14) || height >= (1 << 14))
        return DSDTM_ERR_INVALID;
    const size_t n = (size_t)width * height, pitch = align_up(n, 256);
    if (int rc = ensure_stage(ctx, 3 * pitch + 256)) return rc;
    uint8_t* h = (uint8_t*)ctx->h_pinned;
    uint8_t* d = (uint8_t*)ctx->d_stage;
    for (int y = 0; y < height; ++y) memcpy(h + (size_t)y * width, img + (size_t)y * stride, width);
    HIP_TRY(ctx, hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(d + pitch, 0, 2 * pitch + 256, ctx->stream));
    DetectArgs a;
    memset(&a, 0, sizeof a);
    a.pyr = d; a.score = d + pitch; a.keep = d + 2 * pitch; a.cell_key = (unsigned long long*)(d + 3 * pitch);
    a.cell_size = 1 << 20; a.grid_cols = 1; a.grid_rows = 1; a.barrier = barrier; a.detection_threshold = 3.0e38f;
    a.lv[0].w = width; a.lv[0].h = height; a.lv[0].stride = width; a.lv[0].off = 0;
    a.pyr_pitch = pitch; a.n_frames = 1; a.levels = 1;
    HIP_TRY(ctx, detect_launch(a, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(h, d + pitch, 2 * pitch, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(score, h, n);
    memcpy(keep, h + pitch, n);
    return DSDTM_OK;
}
#endif  // DSDTM_DIAG

// ---- Align2D ------------------------------------------------------------------------------
extern "C" int dsdtm_align2d_batch_device(dsdtm_ctx* ctx, const dsdtm_image_desc* cur, const uint8_t* patch_border,
                                          const uint8_t* patch, const int32_t* level, double* px_xy,
                                          uint8_t* converged, int max_iters, int m, void* hip_stream) {
    if (!ctx) return DSDTM_ERR_INVALID;
    if (!cur || m < 0 || cur->levels <= 0 || cur->levels > DSDTM_MAX_LEVELS) { set_err(ctx, "bad image descriptor"); return DSDTM_ERR_INVALID; }
    if (m == 0) return DSDTM_OK;
    if (!cur->data || !patch_border || !patch || !level || !px_xy || !converged) { set_err(ctx, "NULL device pointer"); return DSDTM_ERR_INVALID; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    A2DKernelArgs a;
    memset(&a, 0, sizeof a);
    for (int l = 0; l < cur->levels; ++l) {
        const size_t end = cur->level_offset[l] + (size_t)cur->stride[l] * cur->height[l];
        if (cur->width[l] <= 0 || cur->height[l] <= 0 || cur->stride[l] < cur->width[l] || end > cur->bytes) {
            set_err(ctx, "level %d does not fit inside the packed pyramid", l); return DSDTM_ERR_INVALID;
        }
        a.lv[l].w = cur->width[l]; a.lv[l].h = cur->height[l]; a.lv[l].stride = cur->stride[l];
        a.lv[l].off = (uint32_t)cur->level_offset[l];
    }
    a.cur_pyr = cur->data; a.patch_border = patch_border; a.patch = patch; a.level = level;
    a.px_xy = px_xy; a.converged = converged; a.m = m; a.max_iters = max_iters; a.levels = cur->levels;
    HIP_TRY(ctx, align2d_launch(a, (hipStream_t)hip_stream));
    return DSDTM_OK;
}

extern "C" int dsdtm_align2d_batch(dsdtm_ctx* ctx, const dsdtm_pyramid* cur, const uint8_t* patch_border,
                                   const uint8_t* patch, const int32_t* level, double* px_xy, uint8_t* converged,
                                   int max_iters, int m) {
    if (!ctx) return DSDTM_ERR_INVALID;
    if (m < 0 || (m > 0 && (!patch_border || !patch || !level || !px_xy || !converged))) { set_err(ctx, "NULL argument"); return DSDTM_ERR_INVALID; }
    PackedPyr pc;
    if (int rc = plan_pyramid(ctx, cur, &pc)) return rc;
    if (m == 0) return DSDTM_OK;
    for (int i = 0; i < m; ++i)
        if (level[i] < 0 || level[i] >= pc.levels) { set_err(ctx, "level[%d]=%d outside the pyramid", i, level[i]); return DSDTM_ERR_INVALID; }
    const size_t M = (size_t)m;
    size_t o = 0;
    const size_t o_pyr = o; o += align_up(pc.bytes, 256);
    const size_t o_pb = o; o += align_up(M * 100, 256);
    const size_t o_p = o; o += align_up(M * 64, 256);
    const size_t o_lv = o; o += align_up(M * 4, 256);
    const size_t o_px = o; o += align_up(M * 16, 256);
    const size_t in_bytes = o;
    const size_t o_cv = o; o += align_up(M, 256);
    const size_t total = o;
    if (int rc = ensure_stage(ctx, total)) return rc;
    uint8_t* h = (uint8_t*)ctx->h_pinned;
    uint8_t* d = (uint8_t*)ctx->d_stage;
    pack_pyramid(cur, pc, h + o_pyr);
    memcpy(h + o_pb, patch_border, M * 100);
    memcpy(h + o_p, patch, M * 64);
    memcpy(h + o_lv, level, M * 4);
    memcpy(h + o_px, px_xy, M * 16);
    HIP_TRY(ctx, hipMemcpyAsync(d, h, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    dsdtm_image_desc img;
    memset(&img, 0, sizeof img);
    img.levels = pc.levels;
    for (int l = 0; l < pc.levels; ++l) { img.width[l] = pc.w[l]; img.height[l] = pc.h[l]; img.stride[l] = pc.w[l]; img.level_offset[l] = pc.off[l]; }
    img.bytes = pc.bytes; img.data = d + o_pyr;
    if (int rc = dsdtm_align2d_batch_device(ctx, &img, d + o_pb, d + o_p, (const int32_t*)(d + o_lv),
                                            (double*)(d + o_px), d + o_cv, max_iters, m, ctx->stream)) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(h + o_px, d + o_px, total - o_px, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(px_xy, h + o_px, M * 16);
    memcpy(converged, h + o_cv, M);
    return DSDTM_OK;
}

// ---- pyrDown ------------------------------------------------------------------------------
extern "C" int dsdtm_pyrdown_batch_device(dsdtm_ctx* ctx, uint8_t* pyr, size_t pyr_pitch, int n_images, int levels,
                                          const int* width, const int* height, const int* stride,
                                          const size_t* level_offset, void* hip_stream) {
    if (!ctx) return DSDTM_ERR_INVALID;
    if (!pyr || !width || !height || !stride || !level_offset || levels <= 0 || levels > DSDTM_MAX_LEVELS || n_images < 0) {
        set_err(ctx, "bad argument"); return DSDTM_ERR_INVALID;
    }
    for (int l = 0; l < levels; ++l) {
        if (width[l] <= 0 || height[l] <= 0 || stride[l] < width[l] ||
            level_offset[l] + (size_t)stride[l] * height[l] > pyr_pitch) { set_err(ctx, "level %d does not fit", l); return DSDTM_ERR_INVALID; }
        if (l > 0 && (width[l] != (width[l - 1] + 1) / 2 || height[l] != (height[l - 1] + 1) / 2)) {
            set_err(ctx, "level %d is not ((w+1)/2,(h+1)/2) of level %d", l, l - 1); return DSDTM_ERR_INVALID;
        }
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // Few images (the live tracker's new frame): ONE launch builds every level, intermediate levels in LDS — the call is
    // bound by launch latency and dependent round trips, 0.0145 -> 0.0079 ms for a 640x480x4 pyramid. Large batches are
    // bound by HBM and run one launch per level (the fused kernel's LDS stages issue no loads: 2..30 % slower from 64
    // images on, `tools/pyr_ab.py`). option pyr_fused = 0 / 2: never / whenever the shape allows (A/B, tests);
    // pyr_band: rows of the coarsest level per workgroup.
    const int fused_mode = options().pyr_fused;
    if (fused_mode == 2 || (fused_mode == 1 && n_images <= 32)) {
        bool launched = false;
        HIP_TRY(ctx, pyrdown_fused_launch(pyr, pyr_pitch, n_images, levels, width, height, stride, level_offset,
                                          options().pyr_band, (hipStream_t)hip_stream, &launched));
        if (launched) return DSDTM_OK;
    }
    for (int l = 1; l < levels; ++l)
        HIP_TRY(ctx, pyrdown_launch(pyr, pyr_pitch, n_images, width[l - 1], height[l - 1], stride[l - 1],
                                    level_offset[l - 1], stride[l], level_offset[l], (hipStream_t)hip_stream));
    return DSDTM_OK;
}

extern "C" int dsdtm_pyrdown(dsdtm_ctx* ctx, const uint8_t* level0, int width, int height, int stride, int levels,
                             uint8_t* const* out_levels, const int* out_stride) {
    if (!ctx) return DSDTM_ERR_INVALID;
    if (!level0 || width <= 0 || height <= 0 || stride < width || levels <= 0 || levels > DSDTM_MAX_LEVELS ||
        (levels > 1 && (!out_levels || !out_stride))) { set_err(ctx, "bad argument"); return DSDTM_ERR_INVALID; }
    int w[DSDTM_MAX_LEVELS], h[DSDTM_MAX_LEVELS], st[DSDTM_MAX_LEVELS];
    size_t off[DSDTM_MAX_LEVELS];
    size_t o = 0;
    for (int l = 0; l < levels; ++l) {
        w[l] = l ? (w[l - 1] + 1) / 2 : width; h[l] = l ? (h[l - 1] + 1) / 2 : height; st[l] = w[l];
        off[l] = o; o += align_up((size_t)w[l] * h[l], 64);
    }
    const size_t pitch = align_up(o, 256);
    if (int rc = ensure_stage(ctx, pitch)) return rc;
    uint8_t* hp = (uint8_t*)ctx->h_pinned;
    uint8_t* d = (uint8_t*)ctx->d_stage;
    for (int y = 0; y < height; ++y) memcpy(hp + (size_t)y * width, level0 + (size_t)y * stride, width);
    HIP_TRY(ctx, hipMemcpyAsync(d, hp, (size_t)width * height, hipMemcpyHostToDevice, ctx->stream));
    if (int rc = dsdtm_pyrdown_batch_device(ctx, d, pitch, 1, levels, w, h, st, off, ctx->stream)) return rc;
    if (levels > 1) HIP_TRY(ctx, hipMemcpyAsync(hp + off[1], d + off[1], o - off[1], hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int l = 1; l < levels; ++l) {
        if (!out_levels[l] || out_stride[l] < w[l]) { set_err(ctx, "bad output level %d", l); return DSDTM_ERR_INVALID; }
        for (int y = 0; y < h[l]; ++y) memcpy(out_levels[l] + (size_t)y * out_stride[l], hp + off[l] + (size_t)y * w[l], w[l]);
    }
    return DSDTM_OK;
}

// ---- warp prelude ---------------------------------------------------------------------------
extern "C" int dsdtm_warp_patches(dsdtm_ctx* ctx, const dsdtm_pyramid* kf_pyr, int n_kf, const dsdtm_camera* cam,
                                  const double* T_kf_w, const double T_cur_w[12], const int32_t* cand_kf,
                                  const float* ref_px, const int32_t* ref_level, const double* ref_bearing,
                                  const double* p_world, int max_search_level, int m, double* affine,
                                  int32_t* search_level, uint8_t* patch_border, uint8_t* patch) {
    if (!ctx) return DSDTM_ERR_INVALID;
    if (!kf_pyr || n_kf <= 0 || !cam || !T_kf_w || !T_cur_w || m < 0 ||
        (m > 0 && (!cand_kf || !ref_px || !ref_level || !ref_bearing || !p_world || !search_level || !patch_border || !patch))) {
        set_err(ctx, "NULL argument"); return DSDTM_ERR_INVALID;
    }
    if (m == 0) return DSDTM_OK;
    PackedPyr p0;
    if (int rc = plan_pyramid(ctx, &kf_pyr[0], &p0)) return rc;
    for (int k = 1; k < n_kf; ++k) {
        PackedPyr pk;
        if (int rc = plan_pyramid(ctx, &kf_pyr[k], &pk)) return rc;
        if (pk.levels != p0.levels || memcmp(pk.w, p0.w, sizeof(int) * p0.levels) || memcmp(pk.h, p0.h, sizeof(int) * p0.levels)) {
            set_err(ctx, "keyframe %d pyramid geometry differs from keyframe 0", k); return DSDTM_ERR_INVALID;
        }
    }
    for (int i = 0; i < m; ++i) {
        if (cand_kf[i] < 0 || cand_kf[i] >= n_kf || ref_level[i] < 0 || ref_level[i] >= p0.levels) {
            set_err(ctx, "candidate %d: keyframe %d / level %d out of range", i, cand_kf[i], ref_level[i]); return DSDTM_ERR_INVALID;
        }
    }
    const size_t M = (size_t)m;
    const size_t pitch = align_up(p0.bytes, 256);
    size_t o = 0;
    const size_t o_pyr = o; o += pitch * (size_t)n_kf;
    const size_t o_tk = o; o += align_up((size_t)n_kf * 96, 256);
    const size_t o_rb = o; o += align_up(M * 24, 256);
    const size_t o_pw = o; o += align_up(M * 24, 256);
    const size_t o_ck = o; o += align_up(M * 4, 256);
    const size_t o_rl = o; o += align_up(M * 4, 256);
    const size_t o_rp = o; o += align_up(M * 8, 256);
    const size_t in_bytes = o;
    const size_t o_af = o; o += align_up(M * 32, 256);
    const size_t o_sl = o; o += align_up(M * 4, 256);
    const size_t o_pb = o; o += align_up(M * 100, 256);
    const size_t o_pp = o; o += align_up(M * 64, 256);
    const size_t total = o;
    if (int rc = ensure_stage(ctx, total)) return rc;
    uint8_t* h = (uint8_t*)ctx->h_pinned;
    uint8_t* d = (uint8_t*)ctx->d_stage;
    for (int k = 0; k < n_kf; ++k) pack_pyramid(&kf_pyr[k], p0, h + o_pyr + pitch * (size_t)k);
    memcpy(h + o_tk, T_kf_w, (size_t)n_kf * 96);
    memcpy(h + o_rb, ref_bearing, M * 24);
    memcpy(h + o_pw, p_world, M * 24);
    memcpy(h + o_ck, cand_kf, M * 4);
    memcpy(h + o_rl, ref_level, M * 4);
    memcpy(h + o_rp, ref_px, M * 8);
    HIP_TRY(ctx, hipMemcpyAsync(d, h, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    WarpKernelArgs a;
    memset(&a, 0, sizeof a);
    a.kf_pyr = d + o_pyr; a.kf_pitch = pitch; a.T_kf_w = (const double*)(d + o_tk);
    a.cand_kf = (const int32_t*)(d + o_ck); a.ref_px = (const float*)(d + o_rp);
    a.ref_level = (const int32_t*)(d + o_rl); a.ref_bearing = (const double*)(d + o_rb);
    a.p_world = (const double*)(d + o_pw); a.affine = (double*)(d + o_af);
    a.search_level = (int32_t*)(d + o_sl); a.patch_border = d + o_pb; a.patch = d + o_pp;
    memcpy(a.T_cur_w, T_cur_w, 96);
    a.m = m; a.n_kf = n_kf; a.max_search_level = max_search_level; a.levels = p0.levels;
    a.fx = cam->fx; a.fy = cam->fy; a.cx = cam->cx; a.cy = cam->cy;
    for (int l = 0; l < p0.levels; ++l) { a.lv[l].w = p0.w[l]; a.lv[l].h = p0.h[l]; a.lv[l].stride = p0.w[l]; a.lv[l].off = (uint32_t)p0.off[l]; }
    HIP_TRY(ctx, warp_launch(a, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(h + o_af, d + o_af, total - o_af, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (affine) memcpy(affine, h + o_af, M * 32);
    memcpy(search_level, h + o_sl, M * 4);
    memcpy(patch_border, h + o_pb, M * 100);
    memcpy(patch, h + o_pp, M * 64);
    return DSDTM_OK;
}

// ---- FindMatchDirect for M candidates on device-resident frames: warp prelude + Align2D, one call ----
extern "C" int dsdtm_match_candidates_frames(dsdtm_ctx* ctx, const dsdtm_frame* cur, const dsdtm_frame* const* kf, int n_kf,
                                             const dsdtm_camera* cam, const double* T_kf_w, const double T_cur_w[12],
                                             const int32_t* cand_kf, const float* ref_px, const int32_t* ref_level,
                                             const double* ref_bearing, const double* p_world, int max_search_level,
                                             int max_iters, int m, double* px_xy, int32_t* search_level, uint8_t* converged) {
    if (!ctx) return DSDTM_ERR_INVALID;
    if (!cur || !kf || n_kf <= 0 || n_kf > 4096 || !cam || !T_kf_w || !T_cur_w || m < 0 ||
        (m > 0 && (!cand_kf || !ref_px || !ref_level || !ref_bearing || !p_world || !px_xy || !search_level || !converged))) {
        set_err(ctx, "NULL argument"); return DSDTM_ERR_INVALID;
    }
    if (m == 0) return DSDTM_OK;
    if (cur->owner != ctx) { set_err(ctx, "frame belongs to another context"); return DSDTM_ERR_INVALID; }
    const PackedPyr& p0 = cur->pl;
    for (int k = 0; k < n_kf; ++k) {
        if (!kf[k] || kf[k]->owner != ctx) { set_err(ctx, "keyframe %d: NULL or foreign frame", k); return DSDTM_ERR_INVALID; }
        const PackedPyr& pk = kf[k]->pl;
        if (pk.levels != p0.levels || memcmp(pk.w, p0.w, sizeof(int) * p0.levels) || memcmp(pk.h, p0.h, sizeof(int) * p0.levels)) {
            set_err(ctx, "keyframe %d pyramid geometry differs from the current frame", k); return DSDTM_ERR_INVALID;
        }
    }
    if (max_search_level < 0 || max_search_level >= p0.levels) { set_err(ctx, "max_search_level outside the pyramid"); return DSDTM_ERR_INVALID; }
    for (int i = 0; i < m; ++i) {
        if (cand_kf[i] < 0 || cand_kf[i] >= n_kf || ref_level[i] < 0 || ref_level[i] >= p0.levels) {
            set_err(ctx, "candidate %d: keyframe %d / level %d out of range", i, cand_kf[i], ref_level[i]); return DSDTM_ERR_INVALID;
        }
    }
    const size_t M = (size_t)m;
    size_t o = 0;
    const size_t o_ptr = o; o += align_up((size_t)n_kf * sizeof(void*), 256);
    const size_t o_tk = o; o += align_up((size_t)n_kf * 96, 256);
    const size_t o_rb = o; o += align_up(M * 24, 256);
    const size_t o_pw = o; o += align_up(M * 24, 256);
    const size_t o_ck = o; o += align_up(M * 4, 256);
    const size_t o_rl = o; o += align_up(M * 4, 256);
    const size_t o_rp = o; o += align_up(M * 8, 256);
    const size_t o_px = o; o += align_up(M * 16, 256);          // in and out
    const size_t in_bytes = o;
    const size_t o_sl = o; o += align_up(M * 4, 256);
    const size_t o_cv = o; o += align_up(M, 256);
    const size_t out_end = o;
    const size_t o_af = o; o += align_up(M * 32, 256);          // device only
    const size_t o_pb = o; o += align_up(M * 100, 256);
    const size_t o_pp = o; o += align_up(M * 64, 256);
    const size_t total = o;
    if (int rc = ensure_stage(ctx, total)) return rc;
    uint8_t* h = (uint8_t*)ctx->h_pinned;
    uint8_t* d = (uint8_t*)ctx->d_stage;
    for (int k = 0; k < n_kf; ++k) ((const uint8_t**)(h + o_ptr))[k] = kf[k]->d;
    memcpy(h + o_tk, T_kf_w, (size_t)n_kf * 96);
    memcpy(h + o_rb, ref_bearing, M * 24);
    memcpy(h + o_pw, p_world, M * 24);
    memcpy(h + o_ck, cand_kf, M * 4);
    memcpy(h + o_rl, ref_level, M * 4);
    memcpy(h + o_rp, ref_px, M * 8);
    memcpy(h + o_px, px_xy, M * 16);
    // Every input is read once per candidate and every result written once: the two kernels take them straight
    // from / to the pinned block (no copy operations around the launches); only the warped patches, which the
    // second kernel re-reads at every iteration, live in device memory.
    const bool zero_copy = !options().no_zero_copy;
    uint8_t* io = d;
    if (zero_copy) {
        void* hd = nullptr;
        HIP_TRY(ctx, hipHostGetDevicePointer(&hd, ctx->h_pinned, 0));
        io = (uint8_t*)hd;
    } else {
        HIP_TRY(ctx, hipMemcpyAsync(d, h, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    }
    WarpKernelArgs a;
    memset(&a, 0, sizeof a);
    a.kf_ptrs = (const uint8_t* const*)(io + o_ptr); a.T_kf_w = (const double*)(io + o_tk);
    a.cand_kf = (const int32_t*)(io + o_ck); a.ref_px = (const float*)(io + o_rp);
    a.ref_level = (const int32_t*)(io + o_rl); a.ref_bearing = (const double*)(io + o_rb);
    a.p_world = (const double*)(io + o_pw); a.affine = (double*)(d + o_af);
    a.search_level = (int32_t*)(io + o_sl); a.patch_border = d + o_pb; a.patch = d + o_pp;
    memcpy(a.T_cur_w, T_cur_w, 96);
    a.m = m; a.n_kf = n_kf; a.max_search_level = max_search_level; a.levels = p0.levels;
    a.fx = cam->fx; a.fy = cam->fy; a.cx = cam->cx; a.cy = cam->cy;
    for (int l = 0; l < p0.levels; ++l) { a.lv[l].w = p0.w[l]; a.lv[l].h = p0.h[l]; a.lv[l].stride = p0.w[l]; a.lv[l].off = (uint32_t)p0.off[l]; }
    A2DKernelArgs b;
    memset(&b, 0, sizeof b);
    b.cur_pyr = cur->d; b.patch_border = d + o_pb; b.patch = d + o_pp; b.level = (const int32_t*)(io + o_sl);
    b.px_xy = (double*)(io + o_px); b.converged = io + o_cv; b.m = m; b.max_iters = max_iters; b.levels = p0.levels;
    b.px_level0 = 1;
    for (int l = 0; l < p0.levels; ++l) b.lv[l] = a.lv[l];
#ifdef DSDTM_DIAG
    if (options().fmd_split) {                 // rounds 1-4 (diagnostic build): two kernels, the warped patches through device memory
        HIP_TRY(ctx, warp_launch(a, ctx->stream));
        HIP_TRY(ctx, align2d_launch(b, ctx->stream));
    } else
#endif
    {
        a.affine = nullptr; a.no_xcd = options().fmd_no_xcd;                    // (nobody reads it here)
        HIP_TRY(ctx, match_launch(a, b, ctx->stream));
    }
    if (!zero_copy) HIP_TRY(ctx, hipMemcpyAsync(h + o_px, d + o_px, out_end - o_px, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(px_xy, h + o_px, M * 16);
    memcpy(search_level, h + o_sl, M * 4);
    memcpy(converged, h + o_cv, M);
    return DSDTM_OK;
}

// FindMatchDirect for the candidates of MANY current frames in one go (independent sequences): every pointer is a device
// pointer, nothing is copied, asynchronous on hip_stream.
extern "C" int dsdtm_match_candidates_batch_device(dsdtm_ctx* ctx, const uint8_t* cur_pyr, int n_frames, const uint8_t* kf_pyr, int n_kf,
                                                   size_t pyr_pitch, int levels, const int* width, const int* height, const int* stride,
                                                   const size_t* level_offset, const dsdtm_camera* cam, const double* T_kf_w,
                                                   const double* T_cur_w, const int32_t* cand_frame, const int32_t* cand_kf,
                                                   const float* ref_px, const int32_t* ref_level, const double* ref_bearing,
                                                   const double* p_world, int max_search_level, int max_iters, int m,
                                                   uint8_t* scratch, double* px_xy, int32_t* search_level, uint8_t* converged,
                                                   void* hip_stream) {
    if (!ctx) return DSDTM_ERR_INVALID;
    if (!cur_pyr || !kf_pyr || n_frames <= 0 || n_kf <= 0 || !width || !height || !stride || !level_offset || !cam || !T_kf_w || !T_cur_w ||
        levels <= 0 || levels > DSDTM_MAX_LEVELS || m < 0 || (pyr_pitch & 3) ||
        (m > 0 && (!cand_frame || !cand_kf || !ref_px || !ref_level || !ref_bearing || !p_world || !scratch || !px_xy || !search_level || !converged))) {
        set_err(ctx, "bad argument"); return DSDTM_ERR_INVALID;
    }
    if (max_search_level < 0 || max_search_level >= levels) { set_err(ctx, "max_search_level outside the pyramid"); return DSDTM_ERR_INVALID; }
    if (((size_t)scratch) & 15) { set_err(ctx, "scratch must be 16-byte aligned (the warped patches are written as whole dwords)"); return DSDTM_ERR_INVALID; }
    if (m == 0) return DSDTM_OK;
    const size_t M = (size_t)m;
    WarpKernelArgs a;
    memset(&a, 0, sizeof a);
    for (int l = 0; l < levels; ++l) {
        if (width[l] <= 0 || height[l] <= 0 || stride[l] < width[l] || level_offset[l] + (size_t)stride[l] * height[l] > pyr_pitch) {
            set_err(ctx, "level %d does not fit", l); return DSDTM_ERR_INVALID;
        }
        a.lv[l].w = width[l]; a.lv[l].h = height[l]; a.lv[l].stride = stride[l]; a.lv[l].off = (uint32_t)level_offset[l];
    }
    uint8_t* affine = scratch;                                  // M x 32, then M x 100, then M x 64 (256-byte aligned sections)
    uint8_t* pb = scratch + align_up(M * 32, 256);
    uint8_t* pp = pb + align_up(M * 100, 256);
    a.kf_pyr = kf_pyr; a.kf_pitch = pyr_pitch; a.T_kf_w = T_kf_w; a.T_cur_w_arr = T_cur_w; a.cand_frame = cand_frame;
    a.cand_kf = cand_kf; a.ref_px = ref_px; a.ref_level = ref_level; a.ref_bearing = ref_bearing; a.p_world = p_world;
    a.affine = (double*)affine; a.search_level = search_level; a.patch_border = pb; a.patch = pp;
    a.m = m; a.n_kf = n_kf; a.max_search_level = max_search_level; a.levels = levels; a.n_frames = n_frames;
    a.fx = cam->fx; a.fy = cam->fy; a.cx = cam->cx; a.cy = cam->cy;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    A2DKernelArgs b;
    memset(&b, 0, sizeof b);
    b.cur_pyr = cur_pyr; b.patch_border = pb; b.patch = pp; b.level = search_level; b.px_xy = px_xy; b.converged = converged;
    b.m = m; b.max_iters = max_iters; b.levels = levels; b.px_level0 = 1; b.frame = cand_frame; b.n_frames = n_frames; b.pyr_pitch = pyr_pitch;
    for (int l = 0; l < levels; ++l) b.lv[l] = a.lv[l];
#ifdef DSDTM_DIAG
    if (options().fmd_split) {                 // rounds 1-4 (diagnostic build): two kernels, the warped patches through `scratch`
        HIP_TRY(ctx, warp_launch(a, (hipStream_t)hip_stream));
        HIP_TRY(ctx, align2d_launch(b, (hipStream_t)hip_stream));
    } else
#endif
    {                                          // one kernel, the patches stay in LDS (`scratch` is not touched)
        a.affine = nullptr; a.no_xcd = options().fmd_no_xcd;
        HIP_TRY(ctx, match_launch(a, b, (hipStream_t)hip_stream));
    }
    return DSDTM_OK;
}
extern "C" size_t dsdtm_match_candidates_scratch_bytes(int m) {
    const size_t M = m > 0 ? (size_t)m : 0;
    return align_up(M * 32, 256) + align_up(M * 100, 256) + align_up(M * 64, 256);
}

// ---- Optimizer::PoseOptimization ---------------------------------------------------------------
extern "C" int dsdtm_pose_optimization_batch_device(dsdtm_ctx* ctx, int n_frames, int max_features,
                                                    const int32_t* n_features, const double* bearing,
                                                    const double* p_world, const int32_t* level, const uint8_t* use,
                                                    double* T_cur_w, const dsdtm_pose_opt_params* params,
                                                    double* residual_norm, dsdtm_pose_opt_summary* summary,
                                                    void* hip_stream) {
    if (!ctx) return DSDTM_ERR_INVALID;
    if (n_frames < 0 || max_features < 0 || !params || params->max_iterations < 0) {
        set_err(ctx, "bad argument"); return DSDTM_ERR_INVALID;
    }
    if (n_frames == 0) return DSDTM_OK;
    if (!T_cur_w || !summary || (max_features > 0 && (!bearing || !p_world || !level || !use || !residual_norm))) {
        set_err(ctx, "NULL argument"); return DSDTM_ERR_INVALID;
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    PoseOptArgs a;
    a.n_frames = n_frames; a.max_features = max_features; a.max_iterations = params->max_iterations;
    a.n_features = n_features; a.bearing = bearing; a.p_world = p_world; a.level = level; a.use = use;
    a.T_cur_w = T_cur_w; a.residual_norm = residual_norm; a.summary = summary;
    HIP_TRY(ctx, pose_opt_launch(a, (hipStream_t)hip_stream));
    return DSDTM_OK;
}

extern "C" int dsdtm_pose_optimization(dsdtm_ctx* ctx, const double* bearing, const double* p_world,
                                       const int32_t* level, const uint8_t* use, int n_features,
                                       double T_cur_w[12], const dsdtm_pose_opt_params* params,
                                       double* residual_norm, dsdtm_pose_opt_summary* summary) {
    if (!ctx) return DSDTM_ERR_INVALID;
    if (!T_cur_w || !params || !summary || n_features < 0 ||
        (n_features > 0 && (!bearing || !p_world || !level || !use || !residual_norm))) {
        set_err(ctx, "NULL argument"); return DSDTM_ERR_INVALID;
    }
    for (int i = 0; i < n_features; ++i)
        if (use[i] && (level[i] < 0 || level[i] >= DSDTM_MAX_LEVELS)) {
            set_err(ctx, "feature %d: level %d out of range", i, level[i]); return DSDTM_ERR_INVALID;
        }
    const size_t N = (size_t)n_features;
    size_t o = 0;
    const size_t o_T = o;   o += 12 * 8;                       // in/out
    const size_t o_sm = o;  o += align_up(sizeof(dsdtm_pose_opt_summary), 8);   // out
    const size_t o_rn = o;  o += N * 8;                        // out
    const size_t out_end = o;
    const size_t o_b = o;   o += N * 24;
    const size_t o_p = o;   o += N * 24;
    const size_t o_l = o;   o += N * 4;
    const size_t o_u = o;   o += align_up(N, 8);
    const size_t total = o;
    if (int rc = ensure_stage(ctx, total)) return rc;
    uint8_t* h = (uint8_t*)ctx->h_pinned;
    uint8_t* d = (uint8_t*)ctx->d_stage;
    memcpy(h + o_T, T_cur_w, 96);
    if (N) {
        memcpy(h + o_b, bearing, N * 24);
        memcpy(h + o_p, p_world, N * 24);
        memcpy(h + o_l, level, N * 4);
        memcpy(h + o_u, use, N);
    }
    // Up to 512 features the kernel keeps a lane's features in registers (pose_opt.hip, CACHED): every input is read once,
    // so it reads them straight from the pinned block and writes pose, summary and norms there — no copy operations
    // around the launch (DSDTM_NO_ZERO_COPY=1 restores them). Larger frames re-read their columns at every evaluation
    // and are copied to the device first.
    const bool zero_copy_enabled = !options().no_zero_copy;
    const bool zero_copy = zero_copy_enabled && n_features > 64 && n_features <= 512 && !options().po_no_cache;
    if (zero_copy) {
        void* hd = nullptr;
        HIP_TRY(ctx, hipHostGetDevicePointer(&hd, ctx->h_pinned, 0));
        d = (uint8_t*)hd;
    } else {
        HIP_TRY(ctx, hipMemcpyAsync(d, h, total, hipMemcpyHostToDevice, ctx->stream));
    }
    if (int rc = dsdtm_pose_optimization_batch_device(ctx, 1, n_features, nullptr, (const double*)(d + o_b),
                                                      (const double*)(d + o_p), (const int32_t*)(d + o_l), d + o_u,
                                                      (double*)(d + o_T), params, (double*)(d + o_rn),
                                                      (dsdtm_pose_opt_summary*)(d + o_sm), ctx->stream)) return rc;
    if (!zero_copy) HIP_TRY(ctx, hipMemcpyAsync(h, d, out_end, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(T_cur_w, h + o_T, 96);
    memcpy(summary, h + o_sm, sizeof(*summary));
    if (summary->n_residual_blocks > 0) memcpy(residual_norm, h + o_rn, (size_t)summary->n_residual_blocks * 8);
    return DSDTM_OK;
}

// ---- one tracked frame in ONE submission (src/Tracking.cpp:199-256) ------------------------------
// new frame -> Run -> ReprojectPoint + Get_ClosetObs for every local map point -> FindMatchDirect for all of them -> the cell
// walk of SearchLocalPoints replayed on the device -> PoseOptimization, enqueued back to back on the context's stream; the host
// waits once. What crosses the link crosses it once and off the kernels' critical paths: level 0 and Run's inputs are read from
// host-mapped pinned memory by ONE kernel into HBM, the local map goes up on a second stream while Run runs, every later kernel
// works on device memory and the few results are written (posted) to the pinned block by whichever kernel has them first.
extern "C" int dsdtm_track_frame(dsdtm_ctx* ctx, const dsdtm_camera* cam, const dsdtm_track_desc* d, dsdtm_track_result* res,
                                 dsdtm_track_match* matches, double* residual_norm) {
    if (!ctx) return DSDTM_ERR_INVALID;
    if (!cam || !d || !res || !matches || !residual_norm) { set_err(ctx, "NULL argument"); return DSDTM_ERR_INVALID; }
    memset(res, 0, sizeof *res);
    if (!d->image || d->width <= 0 || d->height <= 0 || d->stride < d->width || d->levels <= 0 || d->levels > DSDTM_MAX_LEVELS ||
        d->width > 16384 || d->height > 16384) { set_err(ctx, "track: bad image geometry"); return DSDTM_ERR_INVALID; }
    if (!d->ref || !d->T_ref_w || !d->T_seed || d->n_ref_features < 0 ||
        (d->n_ref_features > 0 && (!d->ref_px_xy || !d->ref_bearing || !d->ref_p_world || !d->ref_initial))) {
        set_err(ctx, "track: NULL reference-frame argument"); return DSDTM_ERR_INVALID;
    }
    if (d->n_kf < 0 || d->n_kf > 4096 || d->n_points < 0 || d->n_points > 4096 || (d->n_kf > 0 && (!d->kf || !d->T_kf_w)) ||
        (d->n_points > 0 && (!d->mp_world || !d->mp_found || !d->mp_bad || !d->obs_offset))) {
        set_err(ctx, "track: bad local map (at most 4096 keyframes and 4096 map points per call)"); return DSDTM_ERR_INVALID;
    }
    if (d->cell_size <= 0 || d->cell_size > 127 || d->max_matches <= 0 || d->max_matches > 256 || d->align2d_iters < 0 ||
        d->pose_opt.max_iterations < 0 || (d->mask && d->mask_stride < d->width)) {
        set_err(ctx, "track: bad search parameters (cell_size 1..127, max_matches 1..256)"); return DSDTM_ERR_INVALID;
    }
    const int max_search_level = d->max_pyr_levels - 3;                                      // src/Feature_alignment.cpp:144
    if (max_search_level < 0 || max_search_level >= d->levels) { set_err(ctx, "track: max_pyr_levels - 3 outside the pyramid"); return DSDTM_ERR_INVALID; }
    // the new frame's geometry (Frame::ComputeImagePyramid) must be the reference frame's and the keyframes'
    PackedPyr pl;
    int st[DSDTM_MAX_LEVELS];
    pl.levels = d->levels;
    {
        size_t o = 0;
        for (int l = 0; l < d->levels; ++l) {
            pl.w[l] = l ? (pl.w[l - 1] + 1) / 2 : d->width; pl.h[l] = l ? (pl.h[l - 1] + 1) / 2 : d->height; st[l] = pl.w[l];
            pl.off[l] = o; o += align_up((size_t)pl.w[l] * pl.h[l], 64);
        }
        pl.bytes = o;
    }
    auto same_geometry = [&](const dsdtm_frame* f) {
        return f && f->owner == ctx && f->pl.levels == pl.levels && !memcmp(f->pl.w, pl.w, sizeof(int) * pl.levels) &&
               !memcmp(f->pl.h, pl.h, sizeof(int) * pl.levels);
    };
    if (!same_geometry(d->ref)) { set_err(ctx, "track: the reference frame is NULL, foreign or of another geometry"); return DSDTM_ERR_INVALID; }
    for (int k = 0; k < d->n_kf; ++k)
        if (!same_geometry(d->kf[k])) { set_err(ctx, "track: keyframe %d is NULL, foreign or of another geometry", k); return DSDTM_ERR_INVALID; }
    if (int rc = validate_params(ctx, &d->align, pl.levels)) return rc;
    const int M = d->n_points;
    const int nnz = M > 0 ? d->obs_offset[M] : 0;
    if (M > 0) {
        if (d->obs_offset[0] != 0 || nnz < 0 || (nnz > 0 && (!d->obs_kf || !d->obs_px || !d->obs_level || !d->obs_bearing))) {
            set_err(ctx, "track: bad observation arrays"); return DSDTM_ERR_INVALID;
        }
        for (int i = 0; i < M; ++i)
            if (d->obs_offset[i + 1] < d->obs_offset[i]) { set_err(ctx, "track: obs_offset is not monotone at point %d", i); return DSDTM_ERR_INVALID; }
        for (int j = 0; j < nnz; ++j)
            if (d->obs_kf[j] < 0 || d->obs_kf[j] >= d->n_kf || d->obs_level[j] < 0 || d->obs_level[j] >= pl.levels) {
                set_err(ctx, "track: observation %d names keyframe %d / level %d out of range", j, d->obs_kf[j], d->obs_level[j]);
                return DSDTM_ERR_INVALID;
            }
    }
    const int grid_rows = (d->height + d->cell_size - 1) / d->cell_size, grid_cols = (d->width + d->cell_size - 1) / d->cell_size;   // :29-30
    if ((long long)grid_rows * grid_cols > 4096) { set_err(ctx, "track: more than 4096 grid cells"); return DSDTM_ERR_INVALID; }
    const size_t lds = track_replay_lds_bytes(M, grid_rows * grid_cols, d->cell_size);
    if (lds > 160 * 1024 - 256) { set_err(ctx, "track: %d map points over %d cells do not fit one workgroup's LDS", M, grid_rows * grid_cols); return DSDTM_ERR_INVALID; }

    const size_t img = (size_t)d->width * d->height, nf = (size_t)d->n_ref_features, Mz = (size_t)M, NZ = (size_t)nnz,
                 MM = (size_t)d->max_matches, NK = (size_t)d->n_kf;
    // ---- the pinned block (host-mapped): inputs, then results ----
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o = align_up(o + bytes, 256); return at; };
    const size_t h_img = take(img);
    // (Run's range: what it reads, then its in/out pose, count and statistics — moved to the device in one piece, see below)
    const size_t h_run = o;
    const size_t h_bear = take(nf * 24), h_pw = take(nf * 24), h_tr = take(96), h_px = take(nf * 8), h_ini = take(nf);
    const size_t h_T = take(96), h_nt = take(4), h_st = take(sizeof(dsdtm_align_stats));
    const size_t run_bytes = o - h_run, run_out_bytes = o - h_T;
    // (the local map and the mask: one contiguous range, copied to the device in one piece)
    const size_t h_map = o;
    const size_t h_mask = take(d->mask ? img : 0);
    const size_t h_mpw = take(Mz * 24), h_found = take(Mz * 4), h_bad = take(Mz), h_off = take((Mz + 1) * 4), h_okf = take(NZ * 4),
                 h_opx = take(NZ * 8), h_olv = take(NZ * 4), h_ob = take(NZ * 24), h_Tkf = take(NK * 96), h_kfp = take(NK * sizeof(void*));
    const size_t map_bytes = o - h_map;
    const size_t h_cnt = take(64), h_match = take(MM * sizeof(dsdtm_track_match)), h_Topt = take(96),
                 h_sm = take(sizeof(dsdtm_pose_opt_summary)), h_rn = take(MM * 8);
    const size_t h_total = o;
    // ---- device scratch ----
    o = 0;
    const size_t g_run = take(run_bytes), g_map = take(map_bytes), g_pw = take(Mz * 24), g_cell = take(Mz * 4),
                 g_px0 = take(Mz * 16), g_px = take(Mz * 16), g_ck = take(Mz * 4), g_cf = take(Mz * 4), g_rp = take(Mz * 8),
                 g_rl = take(Mz * 4), g_rb = take(Mz * 24), g_ib = take(Mz), g_sl = take(Mz * 4), g_cv = take(Mz),
                 g_pob = take(MM * 24), g_pow = take(MM * 24), g_pol = take(MM * 4), g_pou = take(MM), g_pon = take(4), g_Topt = take(96);
    const size_t g_total = o;
    if (int rc = ensure_stage(ctx, h_total > g_total ? h_total : g_total)) return rc;
    uint8_t* h = (uint8_t*)ctx->h_pinned;
    uint8_t* g = (uint8_t*)ctx->d_stage;
    void* hd_ = nullptr;
    HIP_TRY(ctx, hipHostGetDevicePointer(&hd_, ctx->h_pinned, 0));
    uint8_t* hd = (uint8_t*)hd_;                       // the pinned block as the device sees it
    hipStream_t stream = ctx->stream;

    dsdtm_frame* f = nullptr;
    if (int rc = frame_alloc(ctx, pl, &f)) return rc;
    auto fail = [&](int rc) {
        if (ctx->copy_stream[0]) (void)hipStreamSynchronize(ctx->copy_stream[0]);
        (void)hipStreamSynchronize(stream);
        dsdtm_frame_destroy(ctx, f);
        return rc;
    };
#define TRACK_TRY(call)                                                                                        \
    do {                                                                                                       \
        hipError_t e_ = (call);                                                                                \
        if (e_ != hipSuccess) { set_err(ctx, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); return fail(DSDTM_ERR_HIP); } \
    } while (0)

    // 1. the new frame: level 0 crosses the link, the pyramid is built on the device (src/Frame.cpp:35-41, :74-81); with it
    //    what Run(cur, ref) reads and its in/out pose: ONE kernel (ingest_kernel, pyrdown.hip) on the compute stream reads both
    //    from host-mapped pinned memory — no copy operation in front of the first kernel (the hand-over from the copy engine
    //    costs 7-8 us), and Run reads its 57 bytes per feature from HBM instead of over the link.
    // (an image the caller keeps in pinned memory — hipHostMalloc / hipHostRegister, e.g. the capture buffer — is read straight
    // from there; anything else is staged through the context's pinned block first: 5-15 us of host memcpy for 640x480)
    // (an image that already lives in this device's memory — a capture or decode pipeline on the GPU — is read from there by the same
    // kernel, or by a device-to-device copy when it is strided or not 16-byte aligned)
    bool pinned_image = false, device_image = false;
    {
        hipPointerAttribute_t pa_;
        if (hipPointerGetAttributes(&pa_, d->image) == hipSuccess) {
            device_image = pa_.type == hipMemoryTypeDevice;
            pinned_image = pa_.type == hipMemoryTypeHost && d->stride == d->width;
            if (device_image && pa_.device != ctx->device) { set_err(ctx, "track: the image lives on device %d, the context on %d", pa_.device, ctx->device); return fail(DSDTM_ERR_INVALID); }
        } else (void)hipGetLastError();                // (an ordinary host pointer: not an error)
    }
    const void* img_dev = nullptr;                     // the image as the device sees it
    if (device_image) {
        if (d->stride == d->width && !(((size_t)d->image) & 15)) img_dev = d->image;
    } else if (pinned_image && !(((size_t)d->image) & 15)) {
        void* p_ = nullptr;
        if (hipHostGetDevicePointer(&p_, (void*)d->image, 0) == hipSuccess) img_dev = p_;
        else (void)hipGetLastError();
    }
    if (!pinned_image && !device_image) {
        if (d->stride == d->width) memcpy(h + h_img, d->image, img);
        else for (int y = 0; y < d->height; ++y) memcpy(h + h_img + (size_t)y * d->width, d->image + (size_t)y * d->stride, (size_t)d->width);
        img_dev = hd + h_img;
    }
    if (nf) {
        memcpy(h + h_bear, d->ref_bearing, nf * 24);
        memcpy(h + h_pw, d->ref_p_world, nf * 24);
        memcpy(h + h_px, d->ref_px_xy, nf * 8);
        memcpy(h + h_ini, d->ref_initial, nf);
    }
    memcpy(h + h_tr, d->T_ref_w, 96);
    auto seed_run = [&]() {
        memcpy(h + h_T, d->T_seed, 96);                              // cur->Set_Pose(last->Get_Pose()) (src/Tracking.cpp:201)
        memset(h + h_nt, 0, 4); memset(h + h_st, 0, sizeof(dsdtm_align_stats));
    };
    seed_run();
    if (img_dev) TRACK_TRY(ingest_launch(img_dev, f->d, img, hd + h_run, g + g_run, run_bytes, stream));
    else {      // (a pinned image the device cannot address, or one that is not 16-byte aligned; a strided or odd device image: the copy engine)
        if (device_image) TRACK_TRY(hipMemcpy2DAsync(f->d, (size_t)d->width, d->image, (size_t)d->stride, (size_t)d->width, (size_t)d->height, hipMemcpyDeviceToDevice, stream));
        else TRACK_TRY(hipMemcpyAsync(f->d, pinned_image ? (const void*)d->image : (const void*)(h + h_img), img, hipMemcpyHostToDevice, stream));
        TRACK_TRY(ingest_launch(nullptr, nullptr, 0, hd + h_run, g + g_run, run_bytes, stream));
    }
    if (int rc = dsdtm_pyrdown_batch_device(ctx, f->d, f->pitch, 1, pl.levels, pl.w, pl.h, st, pl.off, stream)) return fail(rc);

    // 2. Run(cur, ref) on the device copy of its range
    // Run() (:34-38): too few features -> 0, the pose untouched. Decided on the host: no launch.
    const bool run = d->n_ref_features >= d->align.min_fts && d->align.max_level - 1 >= d->align.min_level && d->n_ref_features > 0;
    uint8_t* const gr = g + g_run - h_run;              // device address of the pinned block's offset X inside Run's range: gr + X
    dsdtm_batch_desc b;
    memset(&b, 0, sizeof b);
    b.n_pairs = 1; b.max_features = d->n_ref_features; b.levels = pl.levels;
    for (int l = 0; l < pl.levels; ++l) { b.width[l] = pl.w[l]; b.height[l] = pl.h[l]; b.stride[l] = pl.w[l]; b.level_offset[l] = pl.off[l]; }
    b.pyr_pitch = d->ref->pitch;
    b.ref_pyr = d->ref->d; b.cur_pyr = f->d;
    b.px_xy = (const float*)(gr + h_px); b.bearing = (const double*)(gr + h_bear); b.p_world = (const double*)(gr + h_pw);
    b.initial = gr + h_ini; b.T_ref_w = (const double*)(gr + h_tr); b.T_cur_w = (double*)(gr + h_T);
    b.n_tracked = (int32_t*)(gr + h_nt); b.stats = (dsdtm_align_stats*)(gr + h_st);
    if (d->ref->pitch != f->pitch) { set_err(ctx, "track: pyramid pitches differ"); return fail(DSDTM_ERR_INVALID); }

    // 3. the local map, packed once (the retry below re-uses it)
    bool packed_map = false;
    auto pack_map = [&]() {
        if (packed_map) return;
        packed_map = true;
        if (d->mask) for (int y = 0; y < d->height; ++y) memcpy(h + h_mask + (size_t)y * d->width, d->mask + (size_t)y * d->mask_stride, (size_t)d->width);
        if (M > 0) {
            memcpy(h + h_mpw, d->mp_world, Mz * 24); memcpy(h + h_found, d->mp_found, Mz * 4); memcpy(h + h_bad, d->mp_bad, Mz);
            memcpy(h + h_off, d->obs_offset, (Mz + 1) * 4);
        } else memset(h + h_off, 0, 4);
        if (nnz > 0) {
            memcpy(h + h_okf, d->obs_kf, NZ * 4); memcpy(h + h_opx, d->obs_px, NZ * 8); memcpy(h + h_olv, d->obs_level, NZ * 4);
            memcpy(h + h_ob, d->obs_bearing, NZ * 24);
        }
        if (d->n_kf > 0) memcpy(h + h_Tkf, d->T_kf_w, NK * 96);
        for (int k = 0; k < d->n_kf; ++k) ((const uint8_t**)(h + h_kfp))[k] = d->kf[k]->d;
    };

    TrackArgs t;
    memset(&t, 0, sizeof t);
    t.T_run = (const double*)(gr + h_T); t.n_tracked = (const int32_t*)(gr + h_nt); t.min_tracked = d->min_tracked;
    t.run_out_dev = gr + h_T; t.run_out_host = hd + h_T; t.run_out_n16 = (int)(run_out_bytes / 16);
    const uint8_t* gm = g + g_map - h_map;                // the device copy of the map range: same offsets as in the pinned block
    t.T_kf_w = (const double*)(gm + h_Tkf); t.kf_ptrs = (const uint8_t* const*)(gm + h_kfp); t.n_kf = d->n_kf;
    t.mp_world = (const double*)(gm + h_mpw); t.mp_found = (const int32_t*)(gm + h_found); t.mp_bad = gm + h_bad; t.n_points = M;
    t.obs_offset = (const int32_t*)(gm + h_off); t.obs_kf = (const int32_t*)(gm + h_okf); t.obs_px = (const float*)(gm + h_opx);
    t.obs_level = (const int32_t*)(gm + h_olv); t.obs_bearing = (const double*)(gm + h_ob);
    t.mask = d->mask ? gm + h_mask : nullptr; t.mask_stride = d->width;
    t.fx = cam->fx; t.fy = cam->fy; t.cx = cam->cx; t.cy = cam->cy; t.width = cam->width; t.height = cam->height; t.levels = pl.levels;
    t.cell_size = d->cell_size; t.grid_cols = grid_cols; t.grid_rows = grid_rows; t.max_matches = d->max_matches;
    track_disc_half_widths(d->cell_size, t.disc_hw);
    t.pw = (double*)(g + g_pw); t.cell = (int32_t*)(g + g_cell); t.px0 = (double*)(g + g_px0); t.px = (double*)(g + g_px);
    t.cand_kf = (int32_t*)(g + g_ck); t.cand_frame = (int32_t*)(g + g_cf); t.ref_px = (float*)(g + g_rp); t.ref_level = (int32_t*)(g + g_rl);
    t.ref_bearing = (double*)(g + g_rb); t.init_blocked = g + g_ib; t.search_level = (int32_t*)(g + g_sl); t.converged = g + g_cv;
    t.matches = (dsdtm_track_match*)(hd + h_match); t.counts = (int32_t*)(hd + h_cnt); t.T_opt = (double*)(g + g_Topt);
    t.po_bearing = (double*)(g + g_pob); t.po_world = (double*)(g + g_pow); t.po_level = (int32_t*)(g + g_pol); t.po_use = g + g_pou;
    t.po_n = (int32_t*)(g + g_pon);

    WarpKernelArgs wa;
    memset(&wa, 0, sizeof wa);
    for (int l = 0; l < pl.levels; ++l) { wa.lv[l].w = pl.w[l]; wa.lv[l].h = pl.h[l]; wa.lv[l].stride = pl.w[l]; wa.lv[l].off = (uint32_t)pl.off[l]; }
    wa.kf_ptrs = t.kf_ptrs; wa.T_kf_w = t.T_kf_w; wa.T_cur_w_arr = t.T_run; wa.cand_frame = t.cand_frame;
    wa.cand_kf = t.cand_kf; wa.ref_px = t.ref_px; wa.ref_level = t.ref_level; wa.ref_bearing = t.ref_bearing; wa.p_world = t.pw;
    wa.search_level = t.search_level; wa.m = M; wa.n_kf = d->n_kf; wa.max_search_level = max_search_level; wa.levels = pl.levels;
    wa.n_frames = 1; wa.fx = cam->fx; wa.fy = cam->fy; wa.cx = cam->cx; wa.cy = cam->cy; wa.no_xcd = options().fmd_no_xcd;
    A2DKernelArgs aa;
    memset(&aa, 0, sizeof aa);
    aa.cur_pyr = f->d; aa.level = t.search_level; aa.px_xy = t.px; aa.converged = t.converged; aa.m = M; aa.max_iters = d->align2d_iters;
    aa.levels = pl.levels; aa.px_level0 = 1; aa.frame = t.cand_frame; aa.n_frames = 1; aa.pyr_pitch = f->pitch;
    for (int l = 0; l < pl.levels; ++l) aa.lv[l] = wa.lv[l];

    PoseOptArgs pa;
    pa.n_frames = 1; pa.max_features = d->max_matches; pa.max_iterations = d->pose_opt.max_iterations;
    pa.n_features = t.po_n; pa.bearing = t.po_bearing; pa.p_world = t.po_world; pa.level = t.po_level; pa.use = t.po_use;
    pa.T_cur_w = t.T_opt; pa.T_mirror = (double*)(hd + h_Topt); pa.residual_norm = (double*)(hd + h_rn); pa.summary = (dsdtm_pose_opt_summary*)(hd + h_sm);

    volatile unsigned* h_flag = ctx->h_flags + dsdtm_ctx::FLAG_SINGLE;
    *h_flag = 0;
    LaunchMode mode;
    mode.single = true;
    for (int attempt = 0; attempt < 2; ++attempt) {
        bool multi_cu = false;
        if (attempt) {                                               // once more: Run's range again, from the seed
            seed_run();
            TRACK_TRY(ingest_launch(nullptr, nullptr, 0, hd + h_run, g + g_run, run_bytes, stream));
        }
        memset(h + h_cnt, 0, 64); memset(h + h_sm, 0, sizeof(dsdtm_pose_opt_summary));
        if (run) { if (int rc = launch_batch(ctx, &b, cam, &d->align, stream, mode, &multi_cu)) return fail(rc); }
        if (!packed_map) {
            // the local map does not depend on Run: it is packed and copied up on a second stream WHILE Run runs (its reads —
            // observation -> keyframe pose -> reference feature — are dependent chains: from host-mapped memory they cost the
            // reprojection 16 us, from HBM 4)
            pack_map();
            if (!ctx->copy_stream[0]) TRACK_TRY(hipStreamCreateWithFlags(&ctx->copy_stream[0], hipStreamNonBlocking));
            TRACK_TRY(hipMemcpyAsync(g + g_map, h + h_map, map_bytes, hipMemcpyHostToDevice, ctx->copy_stream[0]));
            // (waited for by the HOST, while Run runs: a device-side event wait costs the stream a 6-us bubble in front of the
            // next kernel; the kernels behind Run are still enqueued long before Run ends)
            TRACK_TRY(hipStreamSynchronize(ctx->copy_stream[0]));
        }
        // 4. ReprojectPoint + Get_ClosetObs for every point; FindMatchDirect for every point that passed; the cell walk; the
        //    features of the matches; PoseOptimization on them — the instantiation picked on the device by the match count,
        //    as dsdtm_pose_optimization picks it on the host (same arithmetic, same bits as the four-call chain)
        TRACK_TRY(track_match_launch(t, wa, aa, stream));
        TRACK_TRY(track_replay_launch(t, stream));
        pa.force_variant = 3;                                      // one wave / four waves by the match count, chosen by the kernel
        TRACK_TRY(pose_opt_launch(pa, stream));
        TRACK_TRY(hipStreamSynchronize(stream));
        if (!*h_flag) break;
        *h_flag = 0;
        if (!multi_cu || attempt == 1 || options().no_recover) {
            for (int i = 0; i < dsdtm_ctx::MAX_STREAMS; ++i)
                if (ctx->rings[i].used && ctx->rings[i].stream == ctx->stream)
                    (void)hipMemset(ctx->d_counter + i * dsdtm_ctx::COUNTERS_PER_STREAM, 0, sizeof(unsigned) * dsdtm_ctx::COUNTERS_PER_STREAM);
            set_err(ctx, multi_cu ? "sparse-align kernel: a wait for a partner workgroup timed out (re-run disabled: no_recover)"
                                  : "sparse-align kernel: intra-workgroup hand-over timed out");
            return fail(DSDTM_ERR_HIP);
        }
        mode.one_cu = true;                                          // Run cannot fail for scheduling reasons: once more, on one compute unit
        ctx->recovered += 1;
    }
#undef TRACK_TRY
    res->frame = f;
    memcpy(res->T_run, h + h_T, 96);
    res->n_tracked = *(const int32_t*)(h + h_nt);
    memcpy(&res->stats, h + h_st, sizeof res->stats);
    res->lost = res->n_tracked < d->min_tracked ? 1 : 0;
    const int32_t* cnt = (const int32_t*)(h + h_cnt);
    res->n_in_grid = cnt[0];
    res->n_matches = res->lost ? 0 : cnt[1];
    res->replay_full_scan = cnt[2] == 1 ? 1 : 0;
    if (cnt[2] == 2 && !res->lost) { set_err(ctx, "track: the replay of the cell walk did not settle"); dsdtm_frame_destroy(ctx, f); res->frame = nullptr; return DSDTM_ERR_HIP; }
    if (res->lost) {
        memcpy(res->T_opt, res->T_run, 96);
    } else {
        memcpy(res->T_opt, h + h_Topt, 96);
        memcpy(&res->summary, h + h_sm, sizeof res->summary);
        if (res->n_matches > 0) memcpy(matches, h + h_match, (size_t)res->n_matches * sizeof(dsdtm_track_match));
        if (res->summary.n_residual_blocks > 0) memcpy(residual_norm, h + h_rn, (size_t)res->summary.n_residual_blocks * 8);
    }
    return DSDTM_OK;
}

#ifdef DSDTM_DIAG
// ---- debug: device self-test of the FP64 building blocks (not in the public header) ------------
extern "C" int dsdtm_debug_selftest(dsdtm_ctx* ctx, const double* in, double* out, int n_cases) {
    if (!ctx || !in || !out || n_cases < 0) return DSDTM_ERR_INVALID;
    if (n_cases == 0) return DSDTM_OK;
    const size_t ib = align_up((size_t)n_cases * 33 * 8, 256), ob = (size_t)n_cases * 120 * 8;   // selftest.hip: SELFTEST_OUT
    if (int rc = ensure_stage(ctx, ib + ob)) return rc;
    uint8_t* h = (uint8_t*)ctx->h_pinned;
    uint8_t* d = (uint8_t*)ctx->d_stage;
    memcpy(h, in, (size_t)n_cases * 33 * 8);
    HIP_TRY(ctx, hipMemcpyAsync(d, h, ib, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, selftest_launch((const double*)d, (double*)(d + ib), n_cases, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(h + ib, d + ib, ob, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(out, h + ib, ob);
    return DSDTM_OK;
}

// ---- debug: in-kernel stamps of the register kernel (diagnostic instantiation, never timed) ----
// stamps_dev: device buffer of n_pairs*8 uint64: [0] cycles the solver waited for the first pass of
// each level (precompute + pass), [1] waits of the later passes, [2] solve cycles, [3] iterations,
// [4] whole-block cycles, [5] s_memrealtime at the end, [6] s_memtime at the start.
extern "C" int dsdtm_debug_sparse_align_stamps(dsdtm_ctx* ctx, const dsdtm_batch_desc* b, const dsdtm_camera* cam,
                                               const dsdtm_align_params* prm, void* stamps_dev, void* hip_stream) {
    if (!ctx || !stamps_dev) return DSDTM_ERR_INVALID;
    g_stamp_out = stamps_dev;
    const int rc = dsdtm_sparse_align_batch_device(ctx, b, cam, prm, hip_stream);
    g_stamp_out = nullptr;
    return rc;
}

// ---- debug: a team launch whose last member is missing (not in the public header) ----------------
// The members that do run wait for a partner that never arrives: their bounded waits must run out, the pair must
// stop without hanging the device, and dsdtm_sparse_align_check must report it (tests/test_sparse_align_gpu.py).
extern "C" int dsdtm_debug_sparse_align_short_team(dsdtm_ctx* ctx, const dsdtm_batch_desc* b, const dsdtm_camera* cam,
                                                   const dsdtm_align_params* prm, void* hip_stream) {
    if (!ctx) return DSDTM_ERR_INVALID;
    g_team_drop_members = 1;
    const int rc = dsdtm_sparse_align_batch_device(ctx, b, cam, prm, hip_stream);
    g_team_drop_members = 0;
    return rc;
}

// tests: every team / two-member launch of the calling thread loses its last member until this is called with 0
extern "C" void dsdtm_debug_drop_team_members(int n) { g_team_drop_members = n > 0 ? 1 : 0; }

extern "C" int dsdtm_debug_occupancy(dsdtm_ctx* ctx, int variant) {
    if (!ctx) return DSDTM_ERR_INVALID;
    (void)hipSetDevice(ctx->device);
    return sparse_align_occupancy(variant);
}
#endif  // DSDTM_DIAG
