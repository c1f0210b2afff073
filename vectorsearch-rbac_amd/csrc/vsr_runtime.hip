// vsr_runtime.hip — host side of libvsrbac: the C ABI of include/vsrbac.h over the gfx950 kernels.
// No CPU compute path: every search / distance entry point launches HIP kernels or fails.
#include "../../include/vsrbac.h"
#include "vsr_device.h"
#include "vsr_hnsw.h"
#include "vsr_hnsw_build.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <numeric>
#include <string>
#include <unordered_map>
#include <vector>

using namespace vsr;

// ---------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;

// threshold seeding (see search_impl): the sample pass visits every SEED_STRIDE-th tile with 1/SEED_BLOCK_DIV of
// the workgroups (measured on MI355X: 256 / 1 is the cheapest sample that still seeds tightly)
static uint32_t SEED_STRIDE = 256;
static uint32_t SEED_BLOCK_DIV = 1;

static int fail(int status, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return status;
}

#define HIPCHK(expr)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(e_ == hipErrorOutOfMemory ? VSR_ERR_OOM : VSR_ERR_HIP, "%s: %s", #expr, \
                        hipGetErrorString(e_));                                               \
    } while (0)

// helpers of the other translation units of the library (vsr_kmeans.hip)
int vsr_kmeans_fail(const char* what, const char* why, bool oom)
{
    return fail(oom ? VSR_ERR_OOM : VSR_ERR_HIP, "%s: %s", what, why);
}

// ---------------------------------------------------------------------------------------------
// small RAII buffers (grow-only workspaces)
// ---------------------------------------------------------------------------------------------
struct DevBuf {
    void*  p = nullptr;
    size_t cap = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    int reserve(size_t bytes)
    {
        if (bytes <= cap) return VSR_OK;
        const bool regrow = p != nullptr;
        if (p) (void) hipFree(p);
        p = nullptr;
        cap = 0;
        // a quarter of slack the first time; a buffer that had to grow once doubles: hipFree / hipMalloc synchronise the device,
        // and a serving process whose batches differ by a few per cent should stop paying that after its first few calls
        size_t want = std::max(bytes, (size_t) 4096);
        want += regrow ? want : want / 4;
        HIPCHK(hipMalloc(&p, want));
        cap = want;
        return VSR_OK;
    }
    void release()
    {
        if (p) (void) hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct PinBuf {
    void*  p = nullptr;
    size_t cap = 0;
    PinBuf() = default;
    PinBuf(const PinBuf&) = delete;
    PinBuf& operator=(const PinBuf&) = delete;
    ~PinBuf() { release(); }
    int reserve(size_t bytes)
    {
        if (bytes <= cap) return VSR_OK;
        const bool regrow = p != nullptr;
        if (p) (void) hipHostFree(p);
        p = nullptr;
        cap = 0;
        size_t want = std::max(bytes, (size_t) 4096);
        want += regrow ? want : want / 4;                    // (as DevBuf)
        HIPCHK(hipHostMalloc(&p, want, hipHostMallocMapped));
        HIPCHK(hipHostGetDevicePointer(&dp, p, 0));
        cap = want;
        return VSR_OK;
    }
    void release()
    {
        if (p) (void) hipHostFree(p);
        p = nullptr;
        dp = nullptr;
        cap = 0;
    }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
    void* dp = nullptr;            // the same memory as the device sees it (kernels read the staging block directly)
};

struct EventPair {
    hipEvent_t a, b;
    int kind;   // 0 = scan (1 query/pass), 1 = scan (shared pass), 2 = select, 3 = sample scan, 4 = seed select, 5 = whole search
};

struct vsr_ctx {
    int             device = 0;
    hipStream_t     stream = nullptr;
    hipStream_t     own_stream = nullptr;
    hipDeviceProp_t prop;
    // workspaces
    DevBuf d_desc;       // queries (padded) + q norms + scan groups + select queries, one upload
    DevBuf d_partial;
    DevBuf d_cand;       // K1m / K2 candidate buffers
    DevBuf d_flags;      // per-query screening flags of the last call
    DevBuf d_tau;        // seeded thresholds (sample pass)
    DevBuf d_samp;       // K2w: per-query sample buffers
    DevBuf d_qcnt;       // K2w: [candidate counts | sample counts]
    bool   hint_u8 = false;       // vsr_set_query_hint: the caller's DEVICE-resident queries are integers 0..255
    bool   int8_this_call = false;   // search_impl -> make_plan: the queries of this call qualify for the int8 planes
    bool   q8_ok = true;          // ... and every hinted query so far really was (else the hint is dropped)
    PinBuf h_q8;                  // one word the staging kernel sets when a query is not (read without synchronising)
    bool   no_fused = false;      // VSR_NO_FUSED: nq == 1 takes the general path (staging, K1, K5)
    bool   seeding = true;        // seed thresholds of big shared passes from a 1/32 sample pass
    int64_t seed_min_rows = 2000000;
    uint32_t sample_stride = 16;  // K2w sample launch: every 16th tile of a workgroup (VSR_SAMPLE_STRIDE)
    int64_t seed_min_pass_rows = 2048;   // average rows per pass below which the warm-up it removes is too small to pay
    int32_t* d_flag_total = nullptr;   // running count of flagged queries (device)
    bool   screening = true;      // allow K2 (MFMA screening + exact re-rank) for shared passes
    int64_t flagged_seen = 0;
    DevBuf d_out;        // host-API outputs
    DevBuf d_misc;
    DevBuf d_dbg;                 // VSR_FUSED_DBG: timestamps of the one-query launch
    PinBuf h_dbg;
    DevBuf d_done;                // nq == 1 fused path: arrival counters of the in-kernel merge tree (zero between calls)
    DevBuf d_redo;                // vsr_search_device_exact: queries and results of the flagged queries
    PinBuf h_desc;
    PinBuf h_out;
    hipEvent_t desc_done = nullptr;   // staging buffer reuse guard
    bool desc_pending = false;
    // measurement
    int profiling = 0;             // 0 off, 1 HIP events around every launch class, 2 around the main scan launch only
    std::vector<EventPair> pending;
    std::vector<hipEvent_t> event_pool;
    vsr_stats stats{};
    // knobs
    int block_budget = 0;          // 0 = 4 * CUs
    bool fused_dbg = false;        // VSR_FUSED_DBG=1
    bool scan_lane = false;        // VSR_SCAN_LANE=1
    hipEvent_t lane_in = nullptr, lane_out = nullptr;
    int fused_fan = 0;             // VSR_FUSED_FAN: lists per first-level merge of the one-query launch (0: the planner's rule)
    int min_rows_per_block = 256;
    int min_shared_rows = 2048;    // rows per workgroup of a shared pass (VSR_MIN_SHARED_ROWS)
    int max_qb = 16;               // queries per shared pass.  32 (two MFMA query groups) does not pay at d = 128; the planner
                                   // picks it by itself for long rows when the query groups fill it (make_plan)
    uint32_t debug = 0;            // VSR_DEBUG != 0: vsr_stats_get prints host-side timings (measurement only)
    double extra_ms[2] = {0, 0};   // sample scan, seed select (profiling only)
    double host_us[3] = {0, 0, 0}; // VSR_DEBUG: host time in make_plan / waiting for the staging buffer / whole search_impl
    long   host_calls = 0;
    std::string last_kernel;       // main scan kernel of the last search (vsr_last_scan_kernel)
    bool no_classes = false;       // VSR_NO_CLASSES=1: scan role partitions whole (A/B measurements)
    bool max_qb_set = false;       // VSR_MAX_QB / vsr_tune chose the queries per pass: the planner does not override it
    bool no_xcd_map = false;       // VSR_NO_XCD_MAP=1: workgroups in pass order instead of XCD-aware bundles (A/B)
    bool no_mq = false;            // VSR_NO_MQ=1: keep shared passes on K1 (A/B measurements)
    bool no_wide = false;          // VSR_NO_WIDE=1: shared passes on K2 (wave-private tiles) instead of K2w (A/B)
    bool no_gemm = false;          // VSR_NO_GEMM=1: wide passes over long rows on K2w instead of K2g (A/B)
    bool no_k2i = true;            // VSR_K2I=1: the int8 main launch as K2i's per-wave streams instead of K2w's workgroup tiles (A/B;
                                   // measured on the headline step: K2w 0.342 ms, K2i 0.366 ms -- K2w stays the default)
    bool last_k2i = false;         // the last main launch was eligible for K2i
    bool k2i_sample = true;        // VSR_NO_K2I_SAMPLE=1: the int8 sample pass on K2w's kernel instead of K2i's streams (A/B)
    bool no_scan8 = false;         // VSR_NO_SCAN8=1: one-query calls on the fp32 rows even when the int8 planes apply (A/B)
    bool k2i_wide = false;         // VSR_K2I_WIDE=1 (with VSR_K2I=1): 128-column passes on K2i
    int  force_epi = -1;           // VSR_FORCE_EPI=0|1: the main launch's survivor handling regardless of the estimate (tests)
    int  screen_level = 2;         // search_impl -> make_plan: 2 = every screening tier, 1 = no coarse tier (K2g), 0 = exact only
    bool last_coarse = false;      // the last search screened on the coarse planes: its flagged queries go to the fine tier first

    ~vsr_ctx()                     // also runs on vsr_open's error returns: nothing allocated so far is leaked
    {
        for (auto& ep : pending) {
            (void) hipEventDestroy(ep.a);
            (void) hipEventDestroy(ep.b);
        }
        for (auto ev : event_pool) (void) hipEventDestroy(ev);
        // (every DevBuf / PinBuf member releases itself: ~DevBuf, ~PinBuf)
        if (desc_done) (void) hipEventDestroy(desc_done);
        if (d_flag_total) (void) hipFree(d_flag_total);
        if (lane_in) (void) hipEventDestroy(lane_in);
        if (lane_out) (void) hipEventDestroy(lane_out);
        if (own_stream) (void) hipStreamDestroy(own_stream);
    }
};

static std::atomic<uint64_t> g_filter_id{0};

struct vsr_filter {
    // never reused, unlike the address: what the index-side caches (vsr_ivf::parts / view_bitmaps, vsr_hnsw::bitmaps) are
    // keyed by, so that a filter allocated where a freed one used to live can never inherit that filter's permissions
    const uint64_t id = g_filter_id.fetch_add(1, std::memory_order_relaxed) + 1;
    vsr_corpus* corpus = nullptr;
    int         mode = VSR_FILTER_RANGES;
    bool        cached = false;
    uint2*      d_tiles = nullptr;     // RANGES
    uint32_t    n_tiles = 0;
    uint64_t*   d_bitmap = nullptr;    // BITMAP (or impure partition: tiles + bitmap)
    bool        owns_bitmap = false;
    int64_t     allowed_rows = 0;
    int64_t     scanned_rows = 0;
    // pre-filter of a role set = union of disjoint permission classes (documents with the same role signature);
    // the planner scans class by class so that queries of different roles share the classes they have in common
    std::vector<vsr_filter*> parts;
    bool parts_only = false;               // the filter has no tile list of its own: always scanned part by part (IVF probes)
    // planner scratch (one planner per context at a time): group id of this filter in the plan being built
    mutable uint64_t plan_epoch = 0;
    mutable uint32_t plan_group = 0;
};

struct vsr_corpus {
    vsr_ctx*    ctx = nullptr;
    // the corpus's scan lane (VSR_SCAN_LANE=1): the main scan launches of ALL sessions over this corpus queue up on this one
    // stream, so two of them never share the GPU (their short kernels still run beside the other sessions' scans)
    mutable hipStream_t scan_stream = nullptr;
    int64_t     n = 0;
    int         dim = 0;
    uint32_t    stride4 = 0;
    int64_t     row_offset = 0;
    KernelShape shape{};
    float4*     d_rows = nullptr;
    float*      d_norm2 = nullptr;
    uint4*      d_scr = nullptr;         // K2w screening planes (bf16 hi / mid split of the rows), nullptr: not built
    // list-ordered VIEW of another corpus (IVFFlat, vsr_ivf_load): rows / norms / planes / tile lists are the view's own,
    // in list order; identity arrays, RBAC tables and the fp32 rows the exact re-rank gathers stay in `base`, and keys carry
    // base rows through d_rank (physical row -> base row)
    vsr_corpus* base = nullptr;
    uint32_t*   d_rank = nullptr;
    uint2*      d_all_tiles = nullptr;   // identity tile list (K2w always walks an explicit list: unfiltered passes use this)
    uint32_t    pstride4 = 0;            // 16-byte chunks per plane row
    bool        scr_has_mid = true;      // false: every element is exactly a bf16 value (e.g. SIFT's 0..255 integers)
    uint4*      d_scr_c = nullptr;       // K2g coarse planes (hi = bf16(x) only, rows padded to whole 64-element K-steps): long rows
    uint32_t    cstride4 = 0;            // 16-byte chunks per coarse plane row
    uint4*      d_scr8 = nullptr;        // int8 planes (x - 128, 128 bytes per row): corpus of integers 0..255, d <= 128; L2 only
    float*      d_norm2_8 = nullptr;     // sum (x - 128)^2 per row
    float*      d_norm2_max = nullptr;   // max |row|^2 (error bound of K2 screening); +Inf if any |row|^2 is not finite
    bool        k2_safe = true;          // false: some |row|^2 is Inf / NaN (non-finite or huge elements) -> exact kernels only
    int64_t*    d_block = nullptr;
    int32_t*    d_doc = nullptr;
    int64_t*    d_orig = nullptr;
    uint32_t*   d_row_docidx = nullptr;
    // host-side identity (internal order)
    std::vector<int64_t> h_orig;
    std::vector<int32_t> docs;            // sorted unique document ids
    std::vector<uint32_t> doc_row_start;  // docs.size() + 1
    // RBAC
    bool rbac = false;
    std::vector<int32_t> roles;           // sorted unique role ids
    uint32_t words = 0;
    std::vector<uint64_t> doc_mask;       // docs.size() * words
    uint64_t* d_doc_mask = nullptr;
    std::unordered_map<int32_t, std::vector<int32_t>> user_roles;
    std::map<std::pair<int, std::vector<int32_t>>, vsr_filter*> cache;
    // permission classes: documents grouped by identical role signature (doc_mask row)
    std::vector<uint32_t> doc_class;                 // per document
    std::vector<std::vector<uint64_t>> class_sig;    // per class
    std::vector<vsr_filter*> class_filters;          // per class, built on first use (RANGES, owned by the corpus)
    std::vector<vsr_filter*> class_bitmap_filters;   // per class, BITMAP mode: aligned windows + the class's own bitmap
    uint32_t* d_doc_class = nullptr;                 // class of every document (device copy of doc_class)
    // indexes loaded over this corpus: a filter that dies (vsr_filter_free, vsr_rbac_load) is purged from their caches
    std::vector<struct vsr_ivf*>  ivf_indexes;
    std::vector<struct vsr_hnsw*> hnsw_indexes;

    ~vsr_corpus();                 // frees the device arrays and cached filters (also on vsr_corpus_load's error returns)
};

// ---------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------
extern "C" int vsr_abi_version(void) { return VSR_ABI_VERSION; }

extern "C" const char* vsr_last_error(void) { return g_last_error.c_str(); }

extern "C" const char* vsr_status_string(int s)
{
    switch (s) {
    case VSR_OK: return "ok";
    case VSR_ERR_INVALID: return "invalid argument";
    case VSR_ERR_DIM_MISMATCH: return "different vector dimensions";
    case VSR_ERR_NO_DEVICE: return "no usable gfx950 device";
    case VSR_ERR_HIP: return "HIP error";
    case VSR_ERR_OOM: return "out of device memory";
    case VSR_ERR_UNSUPPORTED: return "unsupported";
    case VSR_ERR_NO_RBAC: return "RBAC tables not loaded";
    default: return "unknown";
    }
}

extern "C" int vsr_open(int device, vsr_ctx** out)
{
    if (!out) return fail(VSR_ERR_INVALID, "vsr_open: out is NULL");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(VSR_ERR_NO_DEVICE, "vsr_open: no HIP device (%s); libvsrbac has no CPU path",
                    e == hipSuccess ? "count = 0" : hipGetErrorString(e));
    if (device < 0 || device >= count) return fail(VSR_ERR_INVALID, "vsr_open: device %d of %d", device, count);
    std::unique_ptr<vsr_ctx> ctx(new vsr_ctx());
    ctx->device = device;
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipGetDeviceProperties(&ctx->prop, device));
    if (strncmp(ctx->prop.gcnArchName, "gfx950", 6) != 0)
        return fail(VSR_ERR_NO_DEVICE, "vsr_open: device %d is %s; this library is built for gfx950 only", device,
                    ctx->prop.gcnArchName);
    HIPCHK(hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking));
    ctx->stream = ctx->own_stream;
    HIPCHK(hipEventCreateWithFlags(&ctx->desc_done, hipEventDisableTiming));
    HIPCHK(hipMalloc(&ctx->d_flag_total, 64));
    HIPCHK(hipMemset(ctx->d_flag_total, 0, 64));
    HIPCHK(hipMemset(reinterpret_cast<char*>(ctx->d_flag_total) + 32, 0xFF, 8));   // ScanParams::ones
    const char* env;
    if ((env = getenv("VSR_BLOCK_BUDGET"))) ctx->block_budget = atoi(env);
    if ((env = getenv("VSR_FUSED_FAN"))) ctx->fused_fan = atoi(env);
    if ((env = getenv("VSR_SCAN_LANE"))) ctx->scan_lane = atoi(env) != 0;
    if ((env = getenv("VSR_FUSED_DBG"))) ctx->fused_dbg = atoi(env) != 0;
    if ((env = getenv("VSR_MIN_ROWS_PER_BLOCK"))) ctx->min_rows_per_block = std::max(1, atoi(env));
    if ((env = getenv("VSR_MAX_QB"))) { ctx->max_qb = std::max(1, atoi(env)); ctx->max_qb_set = true; }
    if ((env = getenv("VSR_NO_MQ"))) ctx->no_mq = atoi(env) != 0;
    if ((env = getenv("VSR_NO_WIDE"))) ctx->no_wide = atoi(env) != 0;
    if ((env = getenv("VSR_NO_GEMM"))) ctx->no_gemm = atoi(env) != 0;
    if ((env = getenv("VSR_K2I"))) ctx->no_k2i = atoi(env) == 0;
    if ((env = getenv("VSR_FORCE_EPI"))) ctx->force_epi = atoi(env) != 0;
    if ((env = getenv("VSR_K2I_WIDE"))) ctx->k2i_wide = atoi(env) != 0;
    if ((env = getenv("VSR_NO_SCAN8"))) ctx->no_scan8 = atoi(env) != 0;
    if ((env = getenv("VSR_NO_K2I_SAMPLE"))) ctx->k2i_sample = atoi(env) == 0;
    if ((env = getenv("VSR_NO_CLASSES"))) ctx->no_classes = atoi(env) != 0;
    if ((env = getenv("VSR_DEBUG"))) ctx->debug = (uint32_t) atoi(env);
    if ((env = getenv("VSR_NO_SEED"))) ctx->seeding = atoi(env) == 0;
    if ((env = getenv("VSR_NO_FUSED"))) ctx->no_fused = atoi(env) != 0;
    if ((env = getenv("VSR_NO_XCD_MAP"))) ctx->no_xcd_map = atoi(env) != 0;
    if ((env = getenv("VSR_MIN_SHARED_ROWS"))) ctx->min_shared_rows = std::max(64, atoi(env));
    if ((env = getenv("VSR_SEED_MIN_PASS"))) ctx->seed_min_pass_rows = atoll(env);
    if ((env = getenv("VSR_SAMPLE_STRIDE"))) ctx->sample_stride = (uint32_t) std::min(512, std::max(2, atoi(env)));
    if ((env = getenv("VSR_SEED_STRIDE"))) SEED_STRIDE = (uint32_t) std::max(2, atoi(env));
    if ((env = getenv("VSR_SEED_DIV"))) SEED_BLOCK_DIV = (uint32_t) std::max(1, atoi(env));
    if ((env = getenv("VSR_NO_SCREENING"))) ctx->screening = atoi(env) == 0;
    *out = ctx.release();
    return VSR_OK;
}

int vsr_ctx_device(const vsr_ctx* ctx, hipStream_t* stream)
{
    if (stream) *stream = ctx->stream;
    return ctx->device;
}

extern "C" int vsr_close(vsr_ctx* ctx)
{
    if (!ctx) return VSR_OK;
    (void) hipSetDevice(ctx->device);
    (void) hipStreamSynchronize(ctx->stream);
    delete ctx;
    return VSR_OK;
}

extern "C" int vsr_set_stream(vsr_ctx* ctx, void* s)
{
    if (!ctx) return fail(VSR_ERR_INVALID, "vsr_set_stream: ctx is NULL");
    ctx->stream = s ? reinterpret_cast<hipStream_t>(s) : ctx->own_stream;
    return VSR_OK;
}

extern "C" int vsr_synchronize(vsr_ctx* ctx)
{
    if (!ctx) return fail(VSR_ERR_INVALID, "vsr_synchronize: ctx is NULL");
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return VSR_OK;
}

extern "C" int vsr_device_info(vsr_ctx* ctx, char* name, int name_len, int* cus, int64_t* hbm)
{
    if (!ctx) return fail(VSR_ERR_INVALID, "vsr_device_info: ctx is NULL");
    if (name && name_len > 0)      // some boxes report an empty marketing name: say what the architecture implies
        snprintf(name, (size_t) name_len, "%s (%s)", ctx->prop.name[0] ? ctx->prop.name : "AMD Instinct (CDNA4)",
                 ctx->prop.gcnArchName);
    if (cus) *cus = ctx->prop.multiProcessorCount;
    if (hbm) *hbm = (int64_t) ctx->prop.totalGlobalMem;
    return VSR_OK;
}

extern "C" int vsr_tune(vsr_ctx* ctx, int block_budget, int min_rows_per_block, int max_qb)
{
    if (!ctx) return fail(VSR_ERR_INVALID, "vsr_tune: ctx is NULL");
    if (block_budget >= 0) ctx->block_budget = block_budget;
    if (min_rows_per_block > 0) ctx->min_rows_per_block = min_rows_per_block;
    if (max_qb > 0) { ctx->max_qb = max_qb; ctx->max_qb_set = true; }
    return VSR_OK;
}

// ---------------------------------------------------------------------------------------------
// measurement
// ---------------------------------------------------------------------------------------------
static hipEvent_t take_event(vsr_ctx* ctx)
{
    if (!ctx->event_pool.empty()) {
        hipEvent_t e = ctx->event_pool.back();
        ctx->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void) hipEventCreate(&e);
    return e;
}

static void drain_events(vsr_ctx* ctx)
{
    for (auto& ep : ctx->pending) {
        float ms = 0.f;
        if (hipEventSynchronize(ep.b) == hipSuccess && hipEventElapsedTime(&ms, ep.a, ep.b) == hipSuccess) {
            if (ep.kind == 5) {
                ctx->stats.search_ms += ms;
            } else if (ep.kind >= 3) {
                ctx->extra_ms[ep.kind - 3] += ms;
            } else if (ep.kind < 2) {
                ctx->stats.scan_ms[ep.kind] += ms;
                ctx->stats.scan_launches[ep.kind]++;
            } else {
                ctx->stats.select_ms += ms;
                ctx->stats.select_launches++;
            }
        }
        ctx->event_pool.push_back(ep.a);
        ctx->event_pool.push_back(ep.b);
    }
    ctx->pending.clear();
}

extern "C" int vsr_profiling(vsr_ctx* ctx, int enable)
{
    if (!ctx) return fail(VSR_ERR_INVALID, "vsr_profiling: ctx is NULL");
    ctx->profiling = enable < 0 ? 0 : enable;
    return VSR_OK;
}

extern "C" int vsr_stats_get(vsr_ctx* ctx, vsr_stats* out)
{
    if (!ctx || !out) return fail(VSR_ERR_INVALID, "vsr_stats_get: NULL argument");
    HIPCHK(hipStreamSynchronize(ctx->stream));
    drain_events(ctx);
    *out = ctx->stats;
    if (ctx->debug) {              // VSR_DEBUG=1: host-side timing of the searches since the last call (stderr)
        fprintf(stderr, "[vsr debug] sample_scan_ms=%.3f seed_select_ms=%.3f\n", ctx->extra_ms[0], ctx->extra_ms[1]);
        ctx->extra_ms[0] = ctx->extra_ms[1] = 0;
        if (ctx->host_calls)
            fprintf(stderr, "[vsr debug] host per search: plan %.1f us, staging wait %.1f us, total %.1f us (%ld calls)\n",
                    ctx->host_us[0] / ctx->host_calls, ctx->host_us[1] / ctx->host_calls, ctx->host_us[2] / ctx->host_calls,
                    ctx->host_calls);
        ctx->host_us[0] = ctx->host_us[1] = ctx->host_us[2] = 0;
        ctx->host_calls = 0;
    }
    return VSR_OK;
}

extern "C" int vsr_stats_reset(vsr_ctx* ctx)
{
    if (!ctx) return fail(VSR_ERR_INVALID, "vsr_stats_reset: ctx is NULL");
    HIPCHK(hipStreamSynchronize(ctx->stream));
    drain_events(ctx);
    ctx->stats = vsr_stats{};
    return VSR_OK;
}

// ---------------------------------------------------------------------------------------------
// corpus
// ---------------------------------------------------------------------------------------------
static void drop_cached_filters(vsr_corpus* c);
// index-side caches derived from filters (defined with the indexes): drop what belongs to `f` (nullptr: everything)
static void purge_index_caches(vsr_corpus* c, const vsr_filter* f);
static void ranges_to_tiles(const std::vector<std::pair<uint32_t, uint32_t>>& ranges, int rw, std::vector<uint2>& tiles);

extern "C" int vsr_corpus_free(vsr_corpus* c)
{
    if (!c) return VSR_OK;
    (void) hipSetDevice(c->ctx->device);
    (void) hipStreamSynchronize(c->ctx->stream);
    delete c;
    return VSR_OK;
}

vsr_corpus::~vsr_corpus()
{
    if (scan_stream) { (void) hipStreamSynchronize(scan_stream); (void) hipStreamDestroy(scan_stream); }
    drop_cached_filters(this);
    void* ptrs[] = {d_rows, d_scr, d_scr_c, d_scr8, d_norm2_8, d_all_tiles, d_doc_class, d_rank, d_norm2, d_norm2_max, d_block, d_doc, d_orig, d_row_docidx, d_doc_mask};
    for (void* p : ptrs)
        if (p) (void) hipFree(p);
}

extern "C" int64_t vsr_corpus_rows(const vsr_corpus* c) { return c ? c->n : 0; }
extern "C" int vsr_corpus_dim(const vsr_corpus* c) { return c ? c->dim : 0; }

extern "C" int vsr_corpus_load(vsr_ctx* ctx, const float* rows, int64_t n, int dim, const int64_t* block_ids,
                               const int32_t* doc_ids, int64_t row_offset, vsr_corpus** out)
{
    if (!ctx || !out) return fail(VSR_ERR_INVALID, "vsr_corpus_load: NULL argument");
    *out = nullptr;
    if (n < 0 || (n > 0 && !rows)) return fail(VSR_ERR_INVALID, "vsr_corpus_load: rows is NULL");
    if (dim < 1 || dim > 16000)      // VECTOR_MAX_DIM, pgvector/src/vector.h:4
        return fail(VSR_ERR_INVALID, "vsr_corpus_load: vector must have between 1 and 16000 dimensions (got %d)", dim);
    if (n + row_offset >= 0xFFFFFFFFll) return fail(VSR_ERR_UNSUPPORTED, "vsr_corpus_load: more than 2^32-2 rows per shard");
    HIPCHK(hipSetDevice(ctx->device));

    std::unique_ptr<vsr_corpus> c(new vsr_corpus());
    c->ctx = ctx;
    c->n = n;
    c->dim = dim;
    c->stride4 = (uint32_t) ((dim + 3) / 4);
    c->row_offset = row_offset;
    c->shape = scan_shape_for_dim(dim);

    // internal order: (document_id, block_id); identity when the input is already sorted that way
    std::vector<int64_t> perm((size_t) n);
    std::iota(perm.begin(), perm.end(), (int64_t) 0);
    auto doc_of = [&](int64_t r) { return doc_ids ? doc_ids[r] : 0; };
    auto blk_of = [&](int64_t r) { return block_ids ? block_ids[r] : r; };
    bool sorted = true;
    for (int64_t i = 1; i < n && sorted; ++i) {
        const int32_t da = doc_of(i - 1), db = doc_of(i);
        if (da > db || (da == db && blk_of(i - 1) > blk_of(i))) sorted = false;
    }
    if (!sorted)
        std::stable_sort(perm.begin(), perm.end(), [&](int64_t a, int64_t b) {
            const int32_t da = doc_of(a), db = doc_of(b);
            if (da != db) return da < db;
            return blk_of(a) < blk_of(b);
        });
    c->h_orig = perm;

    std::vector<int32_t> h_doc((size_t) n);
    std::vector<int64_t> h_blk((size_t) n);
    std::vector<uint32_t> h_docidx((size_t) n);
    for (int64_t i = 0; i < n; ++i) {
        h_doc[(size_t) i] = doc_of(perm[(size_t) i]);
        h_blk[(size_t) i] = blk_of(perm[(size_t) i]);
        if (i == 0 || h_doc[(size_t) i] != h_doc[(size_t) i - 1]) {
            c->docs.push_back(h_doc[(size_t) i]);
            c->doc_row_start.push_back((uint32_t) i);
        }
        h_docidx[(size_t) i] = (uint32_t) (c->docs.size() - 1);
    }
    c->doc_row_start.push_back((uint32_t) n);

    const size_t row_bytes = (size_t) c->stride4 * 16;
    const size_t alloc_rows = (size_t) std::max<int64_t>(n, 1);
    HIPCHK(hipMalloc(&c->d_rows, alloc_rows * row_bytes + 1024));
    HIPCHK(hipMalloc(&c->d_norm2, alloc_rows * sizeof(float)));
    HIPCHK(hipMalloc(&c->d_norm2_max, 64));
    HIPCHK(hipMemset(c->d_norm2_max, 0, 64));
    HIPCHK(hipMalloc(&c->d_block, alloc_rows * sizeof(int64_t)));
    HIPCHK(hipMalloc(&c->d_doc, alloc_rows * sizeof(int32_t)));
    HIPCHK(hipMalloc(&c->d_orig, alloc_rows * sizeof(int64_t)));
    HIPCHK(hipMalloc(&c->d_row_docidx, alloc_rows * sizeof(uint32_t)));

    if (n > 0) {
        if (sorted && dim % 4 == 0) {
            HIPCHK(hipMemcpy(c->d_rows, rows, (size_t) n * row_bytes, hipMemcpyHostToDevice));
        } else {
            // permute + zero-pad through a bounded host staging buffer
            const size_t chunk_rows = std::max<size_t>(1, (64u << 20) / row_bytes);
            std::vector<float> stage(chunk_rows * c->stride4 * 4);
            for (int64_t base = 0; base < n; base += (int64_t) chunk_rows) {
                const int64_t m = std::min<int64_t>((int64_t) chunk_rows, n - base);
                std::fill(stage.begin(), stage.begin() + (size_t) m * c->stride4 * 4, 0.0f);
                for (int64_t i = 0; i < m; ++i)
                    memcpy(&stage[(size_t) i * c->stride4 * 4], rows + (size_t) perm[(size_t) (base + i)] * dim,
                           (size_t) dim * sizeof(float));
                HIPCHK(hipMemcpy(reinterpret_cast<char*>(c->d_rows) + (size_t) base * row_bytes, stage.data(),
                                 (size_t) m * row_bytes, hipMemcpyHostToDevice));
            }
        }
        HIPCHK(hipMemcpy(c->d_block, h_blk.data(), (size_t) n * sizeof(int64_t), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(c->d_doc, h_doc.data(), (size_t) n * sizeof(int32_t), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(c->d_orig, perm.data(), (size_t) n * sizeof(int64_t), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(c->d_row_docidx, h_docidx.data(), (size_t) n * sizeof(uint32_t), hipMemcpyHostToDevice));
        HIPCHK(launch_row_norms(c->d_rows, (uint32_t) n, c->stride4, c->d_norm2, ctx->stream));
        HIPCHK(launch_norm_max(c->d_norm2, (uint32_t) n, c->d_norm2_max, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        // K2 multiplies the zero padding of its query fragments with row data and needs a finite error bound: a corpus
        // holding NaN / Inf (pgvector rejects those on input, vector.c:101-113) or overflowing norms stays on K1 / K1m
        float nmax = 0.0f;
        HIPCHK(hipMemcpy(&nmax, c->d_norm2_max, sizeof(float), hipMemcpyDeviceToHost));
        c->k2_safe = std::isfinite(nmax);
        // K2w multiplies bf16 hi / mid planes of the rows on the matrix cores (16x the fp32 MFMA rate); the planes are a
        // second, equally large image of the corpus, built once here (288 GB of HBM: the SIFT10M planes are 5 GB)
        if (c->k2_safe && mfmaw_supported(c->stride4) && !getenv("VSR_NO_PLANES")) {
            uint32_t* d_any = reinterpret_cast<uint32_t*>(c->d_norm2_max) + 8;      // spare word of the 64-byte block
            HIPCHK(launch_check_bf16_exact(c->d_rows, (uint32_t) n, c->stride4, d_any, ctx->stream));
            uint32_t any = 1;
            HIPCHK(hipMemcpyAsync(&any, d_any, sizeof any, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(hipStreamSynchronize(ctx->stream));
            c->scr_has_mid = any != 0 || getenv("VSR_NO_HIONLY") != nullptr;
            c->pstride4 = plane_stride4(dim, !c->scr_has_mid);
            HIPCHK(hipMalloc(&c->d_scr, alloc_rows * (size_t) c->pstride4 * 16 + 1024));
            HIPCHK(launch_split_planes(c->d_rows, (uint32_t) n, c->stride4, c->d_scr, c->pstride4, !c->scr_has_mid, ctx->stream));
            std::vector<uint2> all;
            ranges_to_tiles({{0u, (uint32_t) n}}, c->shape.rw, all);
            HIPCHK(hipMalloc(&c->d_all_tiles, all.size() * sizeof(uint2)));
            HIPCHK(hipMemcpy(c->d_all_tiles, all.data(), all.size() * sizeof(uint2), hipMemcpyHostToDevice));
            HIPCHK(hipStreamSynchronize(ctx->stream));
            // long rows (the 768-d configurations) also get the COARSE planes of K2g: hi = bf16(x) alone, half the bytes of
            // the hi + mid planes and one product per element; wide passes (> 128 queries per filter part) screen on them
            if (mfmaw_qmax(c->pstride4, !c->scr_has_mid) > 64 && !getenv("VSR_NO_COARSE")) {
                c->cstride4 = coarse_stride4(dim);
                HIPCHK(hipMalloc(&c->d_scr_c, coarse_plane_u4((uint64_t) alloc_rows, c->cstride4) * 16 + 1024));
                HIPCHK(launch_split_coarse(c->d_rows, (uint32_t) n, c->stride4, c->d_scr_c, c->cstride4, ctx->stream));
                HIPCHK(hipStreamSynchronize(ctx->stream));
            }
            // SIFT-like corpora (every element an integer 0..255, d <= 128) also get int8 planes: a quarter of the fp32
            // bytes per row and v_mfma_i32_16x16x64_i8; used for L2 searches whose queries are such integers too
            if (!c->scr_has_mid && dim <= 128 && !getenv("VSR_NO_INT8")) {
                HIPCHK(hipMemsetAsync(d_any, 0, sizeof(uint32_t), ctx->stream));
                HIPCHK(launch_check_u8_exact(c->d_rows, (uint32_t) n, c->stride4, d_any, ctx->stream));
                HIPCHK(hipMemcpyAsync(&any, d_any, sizeof any, hipMemcpyDeviceToHost, ctx->stream));
                HIPCHK(hipStreamSynchronize(ctx->stream));
                if (any == 0) {
                    // (K2i streams whole 16-row list tiles: a tile that starts at the last row reads 15 rows / norms past it)
                    HIPCHK(hipMalloc(&c->d_scr8, alloc_rows * (size_t) 128 + 4096));
                    HIPCHK(hipMalloc(&c->d_norm2_8, (alloc_rows + 64) * sizeof(float)));
                    HIPCHK(launch_split_planes8(c->d_rows, (uint32_t) n, c->stride4, (uint32_t) dim, c->d_scr8, c->d_norm2_8, ctx->stream));
                    HIPCHK(hipStreamSynchronize(ctx->stream));
                }
            }
        }
    }
    *out = c.release();
    return VSR_OK;
}

// ---------------------------------------------------------------------------------------------
// RBAC
// ---------------------------------------------------------------------------------------------
static void drop_cached_filters(vsr_corpus* c)
{
    purge_index_caches(c, nullptr);                  // the indexes' view-order bitmaps and probe parts of every filter
    for (vsr_filter* f : c->class_filters)
        if (f) {
            if (f->d_tiles) (void) hipFree(f->d_tiles);
            delete f;
        }
    c->class_filters.clear();
    for (vsr_filter* f : c->class_bitmap_filters)
        if (f) {
            if (f->d_tiles) (void) hipFree(f->d_tiles);
            if (f->d_bitmap && f->owns_bitmap) (void) hipFree(f->d_bitmap);
            delete f;
        }
    c->class_bitmap_filters.clear();
    c->class_sig.clear();
    c->doc_class.clear();
    for (auto& kv : c->cache) {
        vsr_filter* f = kv.second;
        if (f->d_tiles) (void) hipFree(f->d_tiles);
        if (f->d_bitmap && f->owns_bitmap) (void) hipFree(f->d_bitmap);
        delete f;
    }
    c->cache.clear();
}

extern "C" int vsr_rbac_load(vsr_corpus* c, const int32_t* ur_user, const int32_t* ur_role, int64_t n_ur,
                             const int32_t* pa_role, const int32_t* pa_doc, int64_t n_pa)
{
    if (!c) return fail(VSR_ERR_INVALID, "vsr_rbac_load: corpus is NULL");
    if ((n_ur > 0 && (!ur_user || !ur_role)) || (n_pa > 0 && (!pa_role || !pa_doc)) || n_ur < 0 || n_pa < 0)
        return fail(VSR_ERR_INVALID, "vsr_rbac_load: NULL table");
    HIPCHK(hipSetDevice(c->ctx->device));
    HIPCHK(hipStreamSynchronize(c->ctx->stream));
    drop_cached_filters(c);

    c->roles.clear();
    for (int64_t i = 0; i < n_ur; ++i) c->roles.push_back(ur_role[i]);
    for (int64_t i = 0; i < n_pa; ++i) c->roles.push_back(pa_role[i]);
    std::sort(c->roles.begin(), c->roles.end());
    c->roles.erase(std::unique(c->roles.begin(), c->roles.end()), c->roles.end());
    c->words = (uint32_t) std::max<size_t>(1, (c->roles.size() + 63) / 64);

    c->user_roles.clear();
    for (int64_t i = 0; i < n_ur; ++i) c->user_roles[ur_user[i]].push_back(ur_role[i]);
    for (auto& kv : c->user_roles) {
        std::sort(kv.second.begin(), kv.second.end());
        kv.second.erase(std::unique(kv.second.begin(), kv.second.end()), kv.second.end());
    }

    c->doc_mask.assign(c->docs.size() * c->words, 0);
    for (int64_t i = 0; i < n_pa; ++i) {
        auto d = std::lower_bound(c->docs.begin(), c->docs.end(), pa_doc[i]);
        if (d == c->docs.end() || *d != pa_doc[i]) continue;     // permission on a document with no rows here
        const size_t di = (size_t) (d - c->docs.begin());
        const size_t ri = (size_t) (std::lower_bound(c->roles.begin(), c->roles.end(), pa_role[i]) - c->roles.begin());
        c->doc_mask[di * c->words + ri / 64] |= 1ull << (ri % 64);
    }
    if (c->d_doc_mask) (void) hipFree(c->d_doc_mask);
    c->d_doc_mask = nullptr;
    const size_t bytes = std::max<size_t>(8, c->doc_mask.size() * sizeof(uint64_t));
    HIPCHK(hipMalloc(&c->d_doc_mask, bytes));
    if (!c->doc_mask.empty())
        HIPCHK(hipMemcpy(c->d_doc_mask, c->doc_mask.data(), c->doc_mask.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
    // permission classes = distinct role signatures of the documents
    {
        std::map<std::vector<uint64_t>, uint32_t> ids;
        c->doc_class.assign(c->docs.size(), 0);
        std::vector<uint64_t> sig(c->words);
        for (size_t di = 0; di < c->docs.size(); ++di) {
            std::copy(c->doc_mask.begin() + (long) (di * c->words), c->doc_mask.begin() + (long) ((di + 1) * c->words), sig.begin());
            auto it = ids.find(sig);
            if (it == ids.end()) {
                it = ids.emplace(sig, (uint32_t) c->class_sig.size()).first;
                c->class_sig.push_back(sig);
            }
            c->doc_class[di] = it->second;
        }
        c->class_filters.assign(c->class_sig.size(), nullptr);
        c->class_bitmap_filters.assign(c->class_sig.size(), nullptr);
        if (c->d_doc_class) (void) hipFree(c->d_doc_class);
        c->d_doc_class = nullptr;
        HIPCHK(hipMalloc(&c->d_doc_class, std::max<size_t>(4, c->doc_class.size() * sizeof(uint32_t))));
        if (!c->doc_class.empty())
            HIPCHK(hipMemcpy(c->d_doc_class, c->doc_class.data(), c->doc_class.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    c->rbac = true;
    return VSR_OK;
}

// ---------------------------------------------------------------------------------------------
// filters
// ---------------------------------------------------------------------------------------------
static int upload_tiles(vsr_filter* f, const std::vector<uint2>& tiles)
{
    f->n_tiles = (uint32_t) tiles.size();
    if (tiles.empty()) return VSR_OK;
    HIPCHK(hipMalloc(&f->d_tiles, tiles.size() * sizeof(uint2)));
    HIPCHK(hipMemcpy(f->d_tiles, tiles.data(), tiles.size() * sizeof(uint2), hipMemcpyHostToDevice));
    return VSR_OK;
}

// contiguous permitted row ranges -> the RW-aligned row windows that hold at least one permitted row
// (bitmap mode: the per-row bits decide inside each window; windows without a set bit are never visited)
static int64_t ranges_to_aligned_tiles(const std::vector<std::pair<uint32_t, uint32_t>>& ranges, int rw, int64_t n,
                                       std::vector<uint2>& tiles)
{
    int64_t rows = 0;
    int64_t last = -1;
    for (auto& r : ranges)
        for (int64_t t = r.first / rw; t <= (int64_t) (r.second - 1) / rw; ++t) {
            if (t == last) continue;
            last = t;
            const uint32_t s = (uint32_t) (t * rw);
            const uint32_t cnt = (uint32_t) std::min<int64_t>(rw, n - (int64_t) s);
            tiles.push_back(make_uint2(s, cnt));
            rows += cnt;
        }
    return rows;
}

// contiguous permitted row ranges -> tiles of <= RW rows
static void ranges_to_tiles(const std::vector<std::pair<uint32_t, uint32_t>>& ranges, int rw, std::vector<uint2>& tiles)
{
    for (auto& r : ranges)
        for (uint32_t s = r.first; s < r.second; s += (uint32_t) rw)
            tiles.push_back(make_uint2(s, std::min<uint32_t>((uint32_t) rw, r.second - s)));
}

static size_t bitmap_words(int64_t n) { return (size_t) ((n + 63) / 64) + 2; }   // + pad for the 2-word window

static int alloc_bitmap(vsr_filter* f)
{
    const size_t bytes = bitmap_words(f->corpus->n) * sizeof(uint64_t);
    HIPCHK(hipMalloc(&f->d_bitmap, bytes));
    HIPCHK(hipMemsetAsync(f->d_bitmap, 0, bytes, f->corpus->ctx->stream));
    f->owns_bitmap = true;
    return VSR_OK;
}

static void free_filter(vsr_filter* f)
{
    if (!f) return;
    if (f->d_tiles) (void) hipFree(f->d_tiles);
    if (f->d_bitmap && f->owns_bitmap) (void) hipFree(f->d_bitmap);
    delete f;
}

static std::vector<uint64_t> role_mask(const vsr_corpus* c, const std::vector<int32_t>& roles)
{
    std::vector<uint64_t> m(c->words, 0);
    for (int32_t r : roles) {
        auto it = std::lower_bound(c->roles.begin(), c->roles.end(), r);
        if (it == c->roles.end() || *it != r) continue;
        const size_t ri = (size_t) (it - c->roles.begin());
        m[ri / 64] |= 1ull << (ri % 64);
    }
    return m;
}

static bool doc_allowed(const vsr_corpus* c, size_t di, const std::vector<uint64_t>& m)
{
    for (uint32_t w = 0; w < c->words; ++w)
        if (c->doc_mask[di * c->words + w] & m[w]) return true;
    return false;
}

constexpr size_t MAX_CLASSES = 4096;        // beyond this (e.g. random RBAC: a signature per document) filters stay whole
constexpr size_t MAX_PARTS = 64;

// the rows of one permission class as a RANGES filter (built once, owned by the corpus)
static int class_filter(vsr_corpus* c, uint32_t cls, vsr_filter** out)
{
    if (c->class_filters[cls]) {
        *out = c->class_filters[cls];
        return VSR_OK;
    }
    std::unique_ptr<vsr_filter, void (*)(vsr_filter*)> f(new vsr_filter(), free_filter);
    f->corpus = c;
    f->mode = VSR_FILTER_RANGES;
    f->cached = true;
    std::vector<std::pair<uint32_t, uint32_t>> ranges;
    int64_t rows = 0;
    for (size_t di = 0; di < c->docs.size(); ++di) {
        if (c->doc_class[di] != cls) continue;
        const uint32_t s = c->doc_row_start[di], e = c->doc_row_start[di + 1];
        rows += e - s;
        if (!ranges.empty() && ranges.back().second == s) ranges.back().second = e;
        else ranges.emplace_back(s, e);
    }
    std::vector<uint2> tiles;
    ranges_to_tiles(ranges, c->shape.rw, tiles);
    int rc = upload_tiles(f.get(), tiles);
    if (rc) return rc;
    f->allowed_rows = f->scanned_rows = rows;
    c->class_filters[cls] = f.release();
    *out = c->class_filters[cls];
    return VSR_OK;
}

static int alloc_bitmap(vsr_filter* f);
static int64_t ranges_to_aligned_tiles(const std::vector<std::pair<uint32_t, uint32_t>>& ranges, int rw, int64_t n,
                                       std::vector<uint2>& tiles);

// the rows of one permission class in post-filter form: the RW-aligned windows that hold at least one of its rows and the
// class's own permission bitmap, tested per row in the distance loop (built once, owned by the corpus)
static int class_bitmap_filter(vsr_corpus* c, uint32_t cls, vsr_filter** out)
{
    if (c->class_bitmap_filters[cls]) {
        *out = c->class_bitmap_filters[cls];
        return VSR_OK;
    }
    std::unique_ptr<vsr_filter, void (*)(vsr_filter*)> f(new vsr_filter(), free_filter);
    f->corpus = c;
    f->mode = VSR_FILTER_BITMAP;
    f->cached = true;
    std::vector<std::pair<uint32_t, uint32_t>> ranges;
    int64_t rows = 0;
    for (size_t di = 0; di < c->docs.size(); ++di) {
        if (c->doc_class[di] != cls) continue;
        const uint32_t s = c->doc_row_start[di], e = c->doc_row_start[di + 1];
        rows += e - s;
        if (!ranges.empty() && ranges.back().second == s) ranges.back().second = e;
        else ranges.emplace_back(s, e);
    }
    int rc = alloc_bitmap(f.get());
    if (rc) return rc;
    HIPCHK(launch_build_class_bitmap(c->d_row_docidx, (uint32_t) c->n, c->d_doc_class, cls, f->d_bitmap, c->ctx->stream));
    std::vector<uint2> tiles;
    f->scanned_rows = ranges_to_aligned_tiles(ranges, c->shape.rw, c->n, tiles);
    rc = upload_tiles(f.get(), tiles);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(c->ctx->stream));
    f->allowed_rows = rows;
    c->class_bitmap_filters[cls] = f.release();
    *out = c->class_bitmap_filters[cls];
    return VSR_OK;
}

static int build_role_filter(vsr_corpus* c, const std::vector<int32_t>& roles, int mode, vsr_filter** out)
{
    std::unique_ptr<vsr_filter, void (*)(vsr_filter*)> f(new vsr_filter(), free_filter);
    f->corpus = c;
    f->mode = mode;
    const std::vector<uint64_t> m = role_mask(c, roles);
    std::vector<std::pair<uint32_t, uint32_t>> ranges;
    int64_t allowed = 0;
    for (size_t di = 0; di < c->docs.size(); ++di) {
        if (!doc_allowed(c, di, m)) continue;
        const uint32_t s = c->doc_row_start[di], e = c->doc_row_start[di + 1];
        allowed += e - s;
        if (!ranges.empty() && ranges.back().second == s) ranges.back().second = e;
        else ranges.emplace_back(s, e);
    }
    f->allowed_rows = allowed;
    if (mode == VSR_FILTER_RANGES) {
        std::vector<uint2> tiles;
        ranges_to_tiles(ranges, c->shape.rw, tiles);
        int rc = upload_tiles(f.get(), tiles);
        if (rc) return rc;
        f->scanned_rows = allowed;
        // the same row set as a union of permission classes (used when many queries are searched together)
        if (c->class_sig.size() <= MAX_CLASSES) {
            for (uint32_t cls = 0; cls < (uint32_t) c->class_sig.size(); ++cls) {
                bool hit = false;
                for (uint32_t w = 0; w < c->words; ++w) hit |= (c->class_sig[cls][w] & m[w]) != 0;
                if (!hit) continue;
                vsr_filter* part = nullptr;
                if ((rc = class_filter(c, cls, &part))) return rc;
                if (part->n_tiles) f->parts.push_back(part);
            }
            if (f->parts.size() > MAX_PARTS || f->parts.size() < 2) f->parts.clear();
        }
    } else {
        int rc = alloc_bitmap(f.get());
        if (rc) return rc;
        vsr_ctx* ctx = c->ctx;
        rc = ctx->d_misc.reserve(c->words * sizeof(uint64_t));
        if (rc) return rc;
        HIPCHK(hipMemcpyAsync(ctx->d_misc.p, m.data(), c->words * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(launch_build_bitmap(c->d_row_docidx, (uint32_t) c->n, c->d_doc_mask, c->words,
                                   ctx->d_misc.as<uint64_t>(), f->d_bitmap, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));      // m is a stack-owned host buffer
        std::vector<uint2> tiles;
        f->scanned_rows = ranges_to_aligned_tiles(ranges, c->shape.rw, c->n, tiles);
        rc = upload_tiles(f.get(), tiles);
        if (rc) return rc;
        // the same row set class by class, each class with its own bitmap: queries of different roles then share the
        // classes they have in common exactly as in pre-filter mode (the bit test per row stays in the distance loop)
        if (c->class_sig.size() <= MAX_CLASSES) {
            for (uint32_t cls = 0; cls < (uint32_t) c->class_sig.size(); ++cls) {
                bool hit = false;
                for (uint32_t w = 0; w < c->words; ++w) hit |= (c->class_sig[cls][w] & m[w]) != 0;
                if (!hit) continue;
                vsr_filter* part = nullptr;
                if ((rc = class_bitmap_filter(c, cls, &part))) return rc;
                if (part->n_tiles) f->parts.push_back(part);
            }
            if (f->parts.size() > MAX_PARTS || f->parts.size() < 2) f->parts.clear();
        }
    }
    *out = f.release();
    return VSR_OK;
}

extern "C" int vsr_filter_for_roles(vsr_corpus* c, const int32_t* role_ids, int n_roles, int mode, vsr_filter** out)
{
    if (!c || !out || n_roles < 0 || (n_roles > 0 && !role_ids)) return fail(VSR_ERR_INVALID, "vsr_filter_for_roles: bad argument");
    *out = nullptr;
    if (mode != VSR_FILTER_RANGES && mode != VSR_FILTER_BITMAP) return fail(VSR_ERR_INVALID, "vsr_filter_for_roles: mode %d", mode);
    if (!c->rbac) return fail(VSR_ERR_NO_RBAC, "vsr_filter_for_roles: call vsr_rbac_load first");
    HIPCHK(hipSetDevice(c->ctx->device));
    std::vector<int32_t> roles(role_ids, role_ids + n_roles);
    std::sort(roles.begin(), roles.end());
    roles.erase(std::unique(roles.begin(), roles.end()), roles.end());
    auto key = std::make_pair(mode, roles);
    auto it = c->cache.find(key);
    if (it != c->cache.end()) {
        *out = it->second;
        return VSR_OK;
    }
    vsr_filter* f = nullptr;
    int rc = build_role_filter(c, roles, mode, &f);
    if (rc) return rc;
    f->cached = true;
    c->cache[key] = f;
    *out = f;
    return VSR_OK;
}

extern "C" int vsr_filter_for_user(vsr_corpus* c, int32_t user_id, int mode, vsr_filter** out)
{
    if (!c || !out) return fail(VSR_ERR_INVALID, "vsr_filter_for_user: NULL argument");
    if (!c->rbac) return fail(VSR_ERR_NO_RBAC, "vsr_filter_for_user: call vsr_rbac_load first");
    auto it = c->user_roles.find(user_id);
    static const std::vector<int32_t> none;
    const std::vector<int32_t>& roles = it == c->user_roles.end() ? none : it->second;   // unknown user: sees nothing
    return vsr_filter_for_roles(c, roles.data(), (int) roles.size(), mode, out);
}

extern "C" int vsr_filter_from_bytemask(vsr_corpus* c, const uint8_t* allowed, int mode, vsr_filter** out)
{
    if (!c || !out || (!allowed && c->n > 0)) return fail(VSR_ERR_INVALID, "vsr_filter_from_bytemask: NULL argument");
    *out = nullptr;
    if (mode != VSR_FILTER_RANGES && mode != VSR_FILTER_BITMAP) return fail(VSR_ERR_INVALID, "vsr_filter_from_bytemask: mode %d", mode);
    HIPCHK(hipSetDevice(c->ctx->device));
    std::unique_ptr<vsr_filter, void (*)(vsr_filter*)> f(new vsr_filter(), free_filter);
    f->corpus = c;
    f->mode = mode;
    int64_t cnt = 0;
    for (int64_t i = 0; i < c->n; ++i) cnt += allowed[i] != 0;
    f->allowed_rows = cnt;
    std::vector<std::pair<uint32_t, uint32_t>> ranges;
    for (int64_t i = 0; i < c->n; ++i) {
        if (!allowed[c->h_orig[(size_t) i]]) continue;
        if (!ranges.empty() && ranges.back().second == (uint32_t) i) ranges.back().second++;
        else ranges.emplace_back((uint32_t) i, (uint32_t) i + 1);
    }
    if (mode == VSR_FILTER_RANGES) {
        std::vector<uint2> tiles;
        ranges_to_tiles(ranges, c->shape.rw, tiles);
        int rc = upload_tiles(f.get(), tiles);
        if (rc) return rc;
        f->scanned_rows = cnt;
    } else {
        int rc = alloc_bitmap(f.get());
        if (rc) return rc;
        vsr_ctx* ctx = c->ctx;
        rc = ctx->d_misc.reserve((size_t) std::max<int64_t>(c->n, 1));
        if (rc) return rc;
        if (c->n > 0) {
            HIPCHK(hipMemcpyAsync(ctx->d_misc.p, allowed, (size_t) c->n, hipMemcpyHostToDevice, ctx->stream));
            HIPCHK(launch_pack_bytemask(ctx->d_misc.as<uint8_t>(), c->d_orig, (uint32_t) c->n, f->d_bitmap, ctx->stream));
        }
        HIPCHK(hipStreamSynchronize(ctx->stream));
        std::vector<uint2> tiles;
        f->scanned_rows = ranges_to_aligned_tiles(ranges, c->shape.rw, c->n, tiles);
        rc = upload_tiles(f.get(), tiles);
        if (rc) return rc;
    }
    *out = f.release();
    return VSR_OK;
}

extern "C" int vsr_filter_from_documents(vsr_corpus* c, const int32_t* doc_ids, int64_t n_docs, int32_t user_id,
                                         vsr_filter** out)
{
    if (!c || !out || n_docs < 0 || (n_docs > 0 && !doc_ids)) return fail(VSR_ERR_INVALID, "vsr_filter_from_documents: bad argument");
    *out = nullptr;
    HIPCHK(hipSetDevice(c->ctx->device));
    std::unique_ptr<vsr_filter, void (*)(vsr_filter*)> f(new vsr_filter(), free_filter);
    f->corpus = c;
    f->mode = VSR_FILTER_RANGES;
    std::vector<int32_t> want(doc_ids, doc_ids + n_docs);
    std::sort(want.begin(), want.end());
    want.erase(std::unique(want.begin(), want.end()), want.end());
    std::vector<uint64_t> um;
    if (user_id >= 0) {
        if (!c->rbac) return fail(VSR_ERR_NO_RBAC, "vsr_filter_from_documents: call vsr_rbac_load first");
        auto it = c->user_roles.find(user_id);
        static const std::vector<int32_t> none;
        um = role_mask(c, it == c->user_roles.end() ? none : it->second);
    }
    std::vector<std::pair<uint32_t, uint32_t>> ranges;
    int64_t scanned = 0, allowed = 0;
    for (int32_t d : want) {
        auto it = std::lower_bound(c->docs.begin(), c->docs.end(), d);
        if (it == c->docs.end() || *it != d) continue;
        const size_t di = (size_t) (it - c->docs.begin());
        const uint32_t s = c->doc_row_start[di], e = c->doc_row_start[di + 1];
        scanned += e - s;
        if (user_id < 0 || doc_allowed(c, di, um)) allowed += e - s;
        if (!ranges.empty() && ranges.back().second == s) ranges.back().second = e;
        else ranges.emplace_back(s, e);
    }
    std::vector<uint2> tiles;
    ranges_to_tiles(ranges, c->shape.rw, tiles);
    int rc = upload_tiles(f.get(), tiles);
    if (rc) return rc;
    f->allowed_rows = allowed;
    f->scanned_rows = scanned;
    if (user_id >= 0) {
        // impure partition: the user's permission bitmap rides along with the partition's tiles
        vsr_filter* ub = nullptr;
        rc = vsr_filter_for_user(c, user_id, VSR_FILTER_BITMAP, &ub);
        if (rc) return rc;
        f->d_bitmap = ub->d_bitmap;
        f->owns_bitmap = false;
    }
    *out = f.release();
    return VSR_OK;
}

extern "C" int vsr_filter_free(vsr_filter* f)
{
    if (!f || f->cached) return VSR_OK;    // cached filters belong to the corpus
    (void) hipSetDevice(f->corpus->ctx->device);
    (void) hipStreamSynchronize(f->corpus->ctx->stream);
    purge_index_caches(f->corpus, f);
    free_filter(f);
    return VSR_OK;
}

extern "C" int64_t vsr_filter_allowed_rows(const vsr_filter* f) { return f ? f->allowed_rows : 0; }
extern "C" int64_t vsr_filter_scanned_rows(const vsr_filter* f) { return f ? f->scanned_rows : 0; }

// ---------------------------------------------------------------------------------------------
// search
// ---------------------------------------------------------------------------------------------
namespace {

constexpr uint32_t SEL_FANIN = 64;           // partial lists one K5 workgroup merges; more -> two levels

struct Plan {
    // slot i = caller query i; passes address their queries through q_slots
    std::vector<uint32_t>    q_slots;        // per pass: the slots of its queries, concatenated
    std::vector<ScanGroup>   groups;         // one K1 / K1m / K2 launch
    uint32_t                 n_blocks = 0;
    int                      qi = 1;         // K1 sub-batch width (1 or 4)
    bool                     mq = false;     // shared passes run on K1m (vsr_mq.h)
    bool                     k2 = false;     // shared passes run on K2 / K2w (MFMA screening) + K5r
    bool                     k2w = false;    // ... on K2w: workgroup-shared row tiles, up to 128 queries per pass (vsr_mfmaw.h)
    bool                     int8 = false;   // ... on the corpus's int8 planes (L2, integer 0..255 rows and queries)
    bool                     k2g = false;    // ... on K2g: long rows, 256-query passes, coarse planes (vsr_gemm.h); implies k2w
    uint32_t                 keep = 0;       // partial list length kp (K2: 2k screening survivors; else k)
    uint32_t                 rerank_base = 0;  // K2: first partial list holding the per-query screening survivors
    uint32_t                 n_scan_lists = 0;
    uint32_t                 qmax = 1;       // query slots per workgroup
    std::vector<uint32_t>    list_ids;       // K5 indirection: per query the indices of its partial lists
    std::vector<SelectQuery> sel1;           // level-1 K5 items (only for queries with many partial lists)
    std::vector<SelectQuery> selq;           // final K5 item per query (slot order)
    std::vector<ScanGroup>   groups_s;       // sample pass (threshold seeding): same passes, fewer workgroups
    bool                     k2i_sample = false;   // int8 planes: the sample pass runs as K2i's per-wave streams (vsr_i8s.h, SAMPLE)
    std::vector<SelectQuery> seedq;          // per query: merge the sample pass's lists into a seed threshold
    uint32_t                 n_blocks_s = 0;
    uint32_t                 n_partial_s = 0;
    uint32_t                 n_partial = 0;  // scan partial lists + level-1 K5 outputs (+ K2 survivor lists)
    bool                     sel_wave = false;  // K5 items are small enough for the one-wave-per-query radix select
    std::vector<uint2>       block_map;      // shared-pass launches: workgroup -> (group, block), XCD-aware (see make_plan)
    uint32_t                 n_launch = 0;   // workgroups of the main launch (= block_map.size() when mapped)
    int64_t                  scan_rows = 0;
    int64_t                  scan_bytes = 0;
    float                    kp_frac = 0;      // K2w: kp * sampling fraction of the densest pass (expected top-kp rows in a sample)
    uint32_t                 sample_stride = 1;  // K2w: the sample launch visits every sample_stride-th tile of a workgroup
    int64_t                  scan_pairs = 0;   // sum over passes of rows * queries
    int64_t                  unique_rows = 0;  // distinct filter parts' rows (capped at the corpus size)

    void reset()                             // keeps the vectors' capacity: one plan per batch, no allocation once warm
    {
        q_slots.clear(); groups.clear(); list_ids.clear(); block_map.clear(); n_launch = 0; sel1.clear(); selq.clear(); groups_s.clear(); seedq.clear(); k2i_sample = false;
        n_blocks = 0; qi = 1; mq = false; k2 = false; k2w = false; int8 = false; k2g = false; keep = 0; rerank_base = 0; n_scan_lists = 0; qmax = 1;
        n_blocks_s = 0; n_partial_s = 0; n_partial = 0; scan_rows = 0; scan_bytes = 0; sel_wave = false;
        scan_pairs = 0; unique_rows = 0; kp_frac = 0; sample_stride = 1;
    }
};

struct PassItem {
    const vsr_filter* part;                  // atomic filter scanned (nullptr = whole corpus)
    uint32_t          slot;
};

}  // namespace

// Queries -> passes.  A filter that is a union of permission classes (vsr_filter::parts) is scanned class by class, so
// that every query whose role sees a class shares that class's pass: the corpus is then read at most
// ceil(queries of the class / qmax) times per class instead of once per role partition.
// Returns false when the K2w plan it built cannot be seeded safely (the caller then plans again with allow_wide = false).
static bool make_plan(const vsr_ctx* ctx, const vsr_corpus* c, int nq, int k, int metric, bool allow_screening,
                      bool allow_wide, bool allow_gemm, const vsr_filter* const* filters, Plan& plan)
{
    auto fof = [&](uint32_t q) { return filters ? filters[q] : nullptr; };

    // (filter part, query slot) items grouped by part: group ids in first-seen order, then a counting sort (stable, so
    // the slots of a part stay ascending).  No comparison sort, no per-query allocation: the planner runs once per batch.
    // plan marks live in the (shared) filters, so the epoch must be unique across host threads: a corpus handed from one
    // thread to another must never meet a stale mark that equals the new thread's counter
    static std::atomic<uint64_t> g_epoch{0};
    const uint64_t epoch = g_epoch.fetch_add(1, std::memory_order_relaxed) + 1;
    static thread_local std::vector<PassItem> raw, items;
    static thread_local std::vector<uint32_t> gid, gcount;
    static thread_local std::vector<const vsr_filter*> gpart;
    raw.clear(); gid.clear(); gcount.clear(); gpart.clear();
    const bool decompose = nq >= 32 && !ctx->no_classes;
    uint32_t null_group = 0xFFFFFFFFu;
    auto group_of = [&](const vsr_filter* f) -> uint32_t {
        if (!f) {
            if (null_group == 0xFFFFFFFFu) {
                null_group = (uint32_t) gpart.size();
                gpart.push_back(nullptr);
                gcount.push_back(0);
            }
            return null_group;
        }
        if (f->plan_epoch != epoch) {
            f->plan_epoch = epoch;
            f->plan_group = (uint32_t) gpart.size();
            gpart.push_back(f);
            gcount.push_back(0);
        }
        return f->plan_group;
    };
    for (uint32_t q = 0; q < (uint32_t) nq; ++q) {
        const vsr_filter* f = fof(q);
        if (f && !f->parts.empty() && (decompose || f->parts_only)) {
            for (const vsr_filter* part : f->parts) {
                const uint32_t g = group_of(part);
                raw.push_back({part, q});
                gid.push_back(g);
                gcount[g]++;
            }
        } else {
            const uint32_t g = group_of(f);
            raw.push_back({f, q});
            gid.push_back(g);
            gcount[g]++;
        }
    }
    {
        uint32_t run = 0;
        for (auto& cnt : gcount) { const uint32_t n = cnt; cnt = run; run += n; }     // counts -> start offsets
        items.resize(raw.size());
        for (size_t i = 0; i < raw.size(); ++i) items[gcount[gid[i]]++] = raw[i];
    }

    for (const vsr_filter* part : gpart) plan.unique_rows += part ? part->scanned_rows : c->n;
    plan.unique_rows = std::min<int64_t>(plan.unique_rows, c->n);

    const bool mq_ok = mq_supported(c->dim) && mq_qmax(c->dim) >= 4 && !ctx->no_mq;
    // K2 / K2w: matrix-core screening keeps 2k (>= 32) candidates per query, K5r re-ranks them exactly
    const uint32_t keep = (uint32_t) std::max(2 * k, 32);
    const bool k2_any = allow_screening && ctx->screening && c->k2_safe && metric != VSR_METRIC_L1 && mq_supported(c->dim) &&
                        ctx->max_qb >= 16;
    const bool k2w_ok = k2_any && allow_wide && c->d_scr && keep <= GQ_MAX_KP && !ctx->no_wide && ctx->seeding;
    const bool k2_ok = k2w_ok || (k2_any && mfma_cap_for_k(keep) <= 8192 && mfma_lds_bytes(c->stride4) <= 150 * 1024);
    int qmax;
    int wq = mfmaw_qmax(c->pstride4, !c->scr_has_mid);
    // K2g (coarse planes, 256-query passes): when nearly all (part, query) items sit in parts seen by more than 128
    // queries -- unfiltered batches, a few big partitions -- and the coarser screen's larger survivor list fits
    const uint32_t keep_c = (uint32_t) std::max(4 * k, 128);
    bool k2g = k2w_ok && allow_gemm && c->d_scr_c && !ctx->no_gemm && ctx->screen_level >= 2 && keep_c <= GQ_MAX_KP && !ctx->max_qb_set;
    if (k2g) {
        uint64_t big = 0, all = 0;
        for (size_t g = 0; g < gcount.size(); ++g) {
            const uint32_t cnt = gcount[g] - (g ? gcount[g - 1] : 0u);         // gcount holds end offsets after the scatter
            all += cnt;
            if (cnt > 128) big += cnt;
        }
        k2g = all > 0 && big * 10 >= all * 9;
    }
    if (k2g) wq = (int) GM_QMAX;
    // int8 planes on K2i: every wave holds the B fragments of the whole pass, and 128 columns fit its registers -- a class seen
    // by 330 queries is streamed 3 times instead of 6.  The sample launch (K2w's kernel: 64 columns) gets such a pass as two.
    // (measured on the headline step: 12 % fewer pass rows, but the 8-group instantiation -- 223 VGPRs, its candidate masks
    // spilled to lanes -- is slower per row: 0.43 ms against 0.357 ms for 64-column passes.  Opt-in: VSR_K2I_WIDE=1.)
    const bool i8wide = k2w_ok && !k2g && c->d_scr8 && metric == VSR_METRIC_L2 && ctx->int8_this_call && !ctx->no_k2i &&
                        ctx->k2i_wide && c->shape.rw == 16 && !ctx->max_qb_set;
    if (i8wide) wq = 128;
    if (wq > 64 && !k2g && !i8wide) {
        // long rows: 128-query passes (two groups per wave: a heavier kernel that also fetches the second group's fragments
        // where a pass has none) pay when nearly all (part, query) items sit in parts seen by more than 64 queries --
        // unfiltered batches: 1M x 768 x 1000 queries 9.9 -> 7.9 ms; a role mix (1000 users over 100 roles) would lose:
        // 1.10 -> 1.39 ms
        uint64_t big = 0, all = 0;
        for (size_t g = 0; g < gcount.size(); ++g) {
            const uint32_t cnt = gcount[g] - (g ? gcount[g - 1] : 0u);         // gcount holds end offsets after the scatter
            all += cnt;
            if (cnt > 64) big += cnt;
        }
        if (big * 10 < all * 9) wq = 64;
    }
    if (k2w_ok) qmax = ctx->max_qb_set ? std::min(ctx->max_qb, wq) : wq;
    else if (k2_ok) qmax = std::min(ctx->max_qb_set ? ctx->max_qb : 16, mfma_qmax(c->stride4));
    else qmax = std::min(ctx->max_qb_set ? ctx->max_qb : 16, mq_ok ? mq_qmax(c->dim) : scan_qmax(c->dim, k));
    if (k2_ok && !k2w_ok && !ctx->max_qb_set && qmax >= 16 && c->stride4 > 64) {
        // long rows (d > 256): a pass costs mostly its row bytes, so two 16-query MFMA groups per pass (half the passes)
        // pay off -- but only when the query groups fill them (an unfiltered 1000-query batch: 17 % less time at
        // d = 768; role partitions with ~25 queries per class: 2.7x more, the second group would be mostly padding)
        uint64_t used = 0, slots = 0;
        for (size_t g = 0; g < gcount.size(); ++g) {
            const uint32_t cnt = gcount[g] - (g ? gcount[g - 1] : 0u);    // gcount holds end offsets after the scatter
            used += cnt;
            slots += (uint64_t) (cnt + 31) / 32 * 32;
        }
        if (slots && used * 10 >= slots * 9) qmax = std::min(32, mfma_qmax(c->stride4));
    }
    qmax = k2w_ok ? std::max(16, qmax / 16 * 16) : qmax >= 4 ? qmax / 4 * 4 : 1;

    struct Pass { const vsr_filter* f; uint32_t q_off, q_count; int64_t rows; uint32_t n_tiles; int64_t cost; };
    static thread_local std::vector<Pass> passes;
    passes.clear();
    uint32_t widest = 1;
    for (size_t s = 0; s < items.size();) {
        size_t e = s;
        while (e < items.size() && items[e].part == items[s].part) ++e;
        const vsr_filter* f = items[s].part;
        // K2w: a part seen by more queries than one pass holds is cut into equal passes (330 queries -> 3 x 110, not
        // 128 + 128 + 74), each a whole number of 16-query MFMA groups
        size_t per = (size_t) qmax;
        if (k2w_ok && e - s > (size_t) qmax) {
            const size_t n_pass = (e - s + (size_t) qmax - 1) / (size_t) qmax;
            per = std::min<size_t>((size_t) qmax, ((e - s + n_pass - 1) / n_pass + 15) / 16 * 16);
        }
        for (size_t b = s; b < e;) {
            const uint32_t cnt = (uint32_t) std::min<size_t>(e - b, per);
            Pass pd;
            pd.f = f;
            pd.q_off = (uint32_t) plan.q_slots.size();
            pd.q_count = cnt;
            pd.rows = f ? f->scanned_rows : c->n;
            pd.n_tiles = f ? f->n_tiles : (uint32_t) ((c->n + c->shape.rw - 1) / c->shape.rw);
            // relative cost of a row of this pass: K2w passes are bound by the row stream up to ~3 query groups and by
            // the matrix pipe beyond (a 64-row tile costs 4 * groups * d/4 MFMAs), so fat passes get more workgroups
            const int64_t groups = (cnt + 15) / 16;
            pd.cost = std::max<int64_t>(pd.rows, 1) * (k2w_ok ? std::max<int64_t>(32, 10 * groups) : 32);
            for (uint32_t i = 0; i < cnt; ++i) plan.q_slots.push_back(items[b + i].slot);
            passes.push_back(pd);
            widest = std::max(widest, cnt);
            b += cnt;
        }
        s = e;
    }
    // one launch: the shared-pass kernels as soon as any pass carries more than one query
    plan.qi = widest > 1 ? 4 : 1;
    plan.qmax = plan.qi == 1 ? 1 : (widest + 3) / 4 * 4;
    plan.k2 = plan.qi == 4 && k2_ok;
    plan.k2w = plan.k2 && k2w_ok;
    plan.mq = plan.qi == 4 && mq_ok && !plan.k2;
    plan.keep = plan.k2 ? keep : (uint32_t) k;
    if (plan.k2w) plan.qmax = (uint32_t) wq;                // query slots per workgroup: one (long rows: two) 16-query groups per wave
    else if (plan.k2) plan.qmax = plan.qmax > 16 ? 32 : 16;
    plan.int8 = plan.k2w && c->d_scr8 && metric == VSR_METRIC_L2 && ctx->int8_this_call;
    if (plan.int8) plan.keep = (uint32_t) std::max(k, 32);  // exact screening: no second half of survivors to re-rank
    plan.k2g = plan.k2w && k2g && !plan.int8;
    if (plan.k2g) plan.keep = keep_c;                       // coarse screening: a wider survivor list for the exact re-rank

    int64_t total_rows = 0, total_cost = 0;
    for (auto& p : passes) {
        total_rows += std::max<int64_t>(p.rows, 1);
        total_cost += p.cost;
    }
    // Workgroups per launch: 4 per CU (two resident at a time), and for big shared-pass launches one per ~13k scanned
    // rows up to 16 per CU -- finer blocks even out the passes' very different lengths over the chip (10M rows, 1000
    // queries: main launch alone 2.38 -> 2.12 ms with 8 per CU).  The sample launch then keeps ~2 workgroups per CU.
    const int64_t cus = ctx->prop.multiProcessorCount;
    // (one query per call: 2 per CU -- one resident round -- halves the lists the in-kernel merge tree has to combine:
    // 0.094 -> 0.079 ms per call on SIFT10M role partitions)
    int64_t budget = ctx->block_budget > 0 ? ctx->block_budget : nq == 1 ? 2 * cus : 4 * cus;
    uint32_t seed_div = SEED_BLOCK_DIV;
    if (ctx->block_budget <= 0 && plan.qi == 4) {
        // K2w keeps 3 workgroups per CU resident and its passes differ a lot in cost per row: ~4 rounds of workgroups
        // even them out (10M rows, 1000 queries: main launch alone 0.97 -> 0.75 ms from 4 to 12 per CU)
        // (K2g: one 8-wave workgroup per CU; ~3 rounds, at least ~8 of its 256-row tiles per workgroup)
        const int64_t want = plan.k2g ? std::min<int64_t>(total_rows / 2048, 3 * cus)
                           : plan.k2w ? std::min<int64_t>(total_rows / 4096, 12 * cus) : std::min<int64_t>(total_rows / 13000, 16 * cus);
        if (want > budget) {
            budget = want;
            seed_div = std::max<uint32_t>(seed_div, (uint32_t) (budget / (2 * cus)));
        }
    }
    if (plan.k2w) {
        // the sample launch visits every ss-th tile: a workgroup of it is all prologue and memory latency, so it gets ONE
        // resident round of workgroups (each then walks ~12 tiles instead of three rounds walking 4: 72 -> ~45 us on the
        // 10M-row corpus); launches that fit one round anyway (a shard) keep the main launch's workgroups
        const int64_t slots = (plan.k2g ? 1 : plan.int8 ? 4 : 3) * cus;
        seed_div = (uint32_t) std::max<int64_t>(1, (budget + slots - 1) / slots);
    }

    // blocks per pass, then the partial lists of every query as CSR (count, prefix, fill): no per-query vectors
    static thread_local std::vector<uint32_t> loff, lcur, lids, lids_s;
    static thread_local std::vector<double> gdens, gdens_s; // per group (sample group): permitted fraction of the rows its tiles cover
    gdens.clear();
    gdens_s.clear();
    loff.assign((size_t) nq + 1, 0);
    for (auto& p : passes) {
        if (p.n_tiles == 0 || p.rows == 0) continue;       // empty filter part: nothing to scan
        int64_t nb = (int64_t) (((__int128) p.cost * budget + total_cost - 1) / total_cost);
        const int64_t min_rows = p.q_count > 1 ? std::max<int64_t>(ctx->min_rows_per_block, ctx->min_shared_rows) : ctx->min_rows_per_block;
        nb = std::min<int64_t>(nb, std::max<int64_t>(1, p.rows / min_rows));   // shared passes need rows to prune on
        nb = std::min<int64_t>(nb, std::max<uint32_t>(1, p.n_tiles));
        nb = std::max<int64_t>(nb, 1);
        ScanGroup g;
        g.tiles = p.f ? p.f->d_tiles : plan.k2w ? c->d_all_tiles : nullptr;
        g.bitmap = p.f ? p.f->d_bitmap : nullptr;
        g.n_tiles = p.n_tiles;
        g.q_begin = p.q_off;
        g.q_count = p.q_count;
        g.block_begin = plan.n_blocks;
        g.n_blocks = (uint32_t) nb;
        g.partial_begin = plan.n_partial;
        plan.groups.push_back(g);
        gdens.push_back(p.f && p.f->scanned_rows > 0 ? (double) p.f->allowed_rows / (double) p.f->scanned_rows : 1.0);
        ScanGroup gs = g;                                   // the same pass in the sample launch (buffers alias:
        gs.n_blocks = (uint32_t) std::max<int64_t>(1, nb / seed_div);         // it finishes before the main launch)
        gs.block_begin = plan.n_blocks_s;
        gs.partial_begin = plan.n_partial_s;
        if (i8wide && p.q_count > 64) {                     // a 128-column pass: two sample groups of at most 64 columns
            ScanGroup g1 = gs;
            g1.q_count = (p.q_count / 2 + 15) / 16 * 16;
            plan.groups_s.push_back(g1);
            gdens_s.push_back(gdens.back());
            plan.n_blocks_s += g1.n_blocks;
            plan.n_partial_s += g1.n_blocks * g1.q_count;
            gs.q_begin += g1.q_count;
            gs.q_count = p.q_count - g1.q_count;
            gs.block_begin = plan.n_blocks_s;
            gs.partial_begin = plan.n_partial_s;
            plan.n_partial_s -= gs.n_blocks * p.q_count - gs.n_blocks * gs.q_count;   // (the common accounting below adds the whole pass)
        }
        plan.groups_s.push_back(gs);
        gdens_s.push_back(gdens.back());
        for (uint32_t qi = 0; qi < p.q_count; ++qi) loff[plan.q_slots[p.q_off + qi] + 1] += g.n_blocks;
        plan.n_blocks += g.n_blocks;
        plan.n_partial += g.n_blocks * p.q_count;
        plan.n_blocks_s += gs.n_blocks;
        plan.n_partial_s += gs.n_blocks * p.q_count;
        plan.scan_rows += p.rows;
        plan.scan_pairs += p.rows * (int64_t) p.q_count;
        plan.scan_bytes += p.rows * (int64_t) c->dim * 4 + (g.bitmap ? (p.rows + 7) / 8 : 0) + (int64_t) p.q_count * k * 12 +
                           (plan.k2 ? p.rows * 4 : 0);     // K2 also reads |row|^2
    }
    plan.n_scan_lists = plan.n_partial;
    plan.n_launch = plan.n_blocks;
    if (plan.k2 || plan.mq) {
        // XCD-aware workgroup order.  Consecutive passes over the same rows (one permission class scanned for several
        // query groups) are split into the same block ranges; block j of all of them forms a bundle that should run on
        // ONE XCD at the same time, so that the rows are fetched over the fabric once and re-read from that XCD's L2.
        // Workgroups are dealt round-robin over the 8 XCDs (id % 8 = one XCD, MI355X_MICROARCH.md): lane l owns the
        // ids l, l+8, l+16, ...; every bundle is appended whole to the currently shortest lane.
        constexpr uint32_t XCDS = 8;
        static thread_local std::vector<uint2> lane[XCDS];
        for (auto& l : lane) l.clear();
        size_t gi = 0;
        while (gi < plan.groups.size()) {
            size_t ge = gi + 1;
            while (ge < plan.groups.size() && plan.groups[ge].tiles == plan.groups[gi].tiles &&
                   plan.groups[ge].bitmap == plan.groups[gi].bitmap && plan.groups[ge].n_tiles == plan.groups[gi].n_tiles &&
                   plan.groups[ge].n_blocks == plan.groups[gi].n_blocks)
                ++ge;
            for (uint32_t j = 0; j < plan.groups[gi].n_blocks; ++j) {
                uint32_t best = 0;
                for (uint32_t l = 1; l < XCDS; ++l)
                    if (lane[l].size() < lane[best].size()) best = l;
                for (size_t g2 = gi; g2 < ge; ++g2) lane[best].push_back(make_uint2((uint32_t) g2, j));
            }
            gi = ge;
        }
        size_t longest = 0;
        for (auto& l : lane) longest = std::max(longest, l.size());
        plan.block_map.assign(longest * XCDS, make_uint2(0xFFFFFFFFu, 0u));
        for (uint32_t l = 0; l < XCDS; ++l)
            for (size_t t = 0; t < lane[l].size(); ++t) plan.block_map[t * XCDS + l] = lane[l][t];
        plan.n_launch = (uint32_t) plan.block_map.size();
    }
    if (plan.k2w) {
        // K2w keeps one candidate buffer per query: no partial lists, no K5 items.  What the plan still owes is the
        // threshold seeding.  The sample launch runs the same workgroups over every ss-th tile of theirs (at least one
        // each), so every pass is sampled at a fraction f >= 1/ss of its rows, and keeps per query only minima: one
        // entry per query column and wave-tile, or one per lane where a query's sample would otherwise be too thin
        // (`fine` passes).  The seed is the m-th smallest entry of a query, m = lambda + 6 sqrt(lambda) + 4 with
        // lambda = kp * f for the most densely sampled pass: more than m of the true top kp rows in the sample has
        // probability ~1e-8, so the seed ranks behind the kp-th row and admits about m / f rows of the query:
        // kp + 6 sqrt(kp / f) + 4 / f (~600 at f = 1/16, kp = 200).  Dropping sample entries (minima, buffer
        // overflow) can only raise the m-th smallest, i.e. loosen the seed.  A query whose sample cannot reach rank m
        // gets an open threshold; that is only safe when all of its rows fit its candidate buffer (GQ_CAP).
        const double tile_rows = plan.k2g ? 256.0 : 64.0;   // rows per workgroup tile of the kernel
        static thread_local std::vector<double> est;
        // K2g on a small corpus: its 256-row tiles make a thin sample; it is sampled more densely (stride 8, 4, 2) before
        // the plan is given up.  (At the sizes it is built for -- millions of rows -- the first stride holds.)
        bool ok = false;
        if (plan.int8 && ctx->k2i_sample && c->shape.rw == 16) {
            // The sample pass as per-wave streams (vsr_i8s.h, SAMPLE): stages of 32 rows, every ss-th stage of a workgroup's
            // range; each of the <= 4 waves that get a stage keeps 4 lanes' minima per query column over its whole stream.
            // Fewer entries than K2w's per-tile minima, so the plan takes it only when every query's sample stays thick enough.
            const double ss = ctx->sample_stride;
            double frac = 1.0 / ss;
            for (const ScanGroup& gs : plan.groups_s) {
                const double t32 = std::ceil((double) gs.n_tiles * c->shape.rw / 32.0);
                const double per_block = std::ceil(t32 / gs.n_blocks);
                const double sampled = std::min(t32, gs.n_blocks * std::ceil(per_block / ss));
                if (t32 > 0) frac = std::max(frac, sampled / t32);
            }
            const double lambda = (double) plan.keep * frac;
            const uint32_t seed_m = (uint32_t) std::ceil(lambda + 6.0 * std::sqrt(lambda)) + 4;
            est.assign((size_t) nq, 0.0);
            for (size_t gi = 0; gi < plan.groups_s.size(); ++gi) {
                const ScanGroup& gs = plan.groups_s[gi];
                const double t32 = std::ceil((double) gs.n_tiles * c->shape.rw / 32.0);
                const double per_block = std::floor(t32 / gs.n_blocks);                  // (the shortest block of the group)
                const double st = std::max(1.0, std::ceil(per_block / ss));              // stages a workgroup samples
                const double waves = std::min(4.0, st);
                const double rows_per_entry = st / waves * 8.0;                          // 2 row blocks x 4 rows per lane and stage
                const double p_entry = std::min(1.0, gdens_s[gi] * rows_per_entry);
                for (uint32_t qi = 0; qi < gs.q_count; ++qi) est[plan.q_slots[gs.q_begin + qi]] += gs.n_blocks * waves * 4.0 * p_entry;
            }
            bool thick = seed_m <= GQ_SAMPLE_CAP / 4;
            for (uint32_t q = 0; q < (uint32_t) nq && thick; ++q) {
                const vsr_filter* f = fof(q);
                const int64_t allowed = f ? f->allowed_rows : c->n;
                const bool exact_count = !f || f->allowed_rows == f->scanned_rows;
                if (allowed > (int64_t) GQ_CAP && est[q] < (exact_count ? 1.5 * seed_m + 16.0 : 2.5 * seed_m)) thick = false;
            }
            if (thick) {
                ok = true;
                plan.k2i_sample = true;
                plan.sample_stride = ctx->sample_stride;
                plan.kp_frac = (float) lambda;
                for (ScanGroup& gs : plan.groups_s) gs.partial_begin = 0u;
            }
        }
        // A sample too thin for some query at the configured stride is taken more densely (8, 4, 2) before the plan is given up:
        // K2g's 256-row tiles on a small corpus, and K2w over many small parts (IVFFlat lists: probes x ~1000 rows per query
        // used to fall back to the legacy kernels as soon as one query's lists added up to more than its candidate buffer).
        // Among the strides that make a valid plan the first one is preferred that also seeds every query with more than a
        // few thousand rows: a query that fits its buffer may run with an open threshold, but then EVERY one of its rows is a
        // candidate (IVFFlat, 4 probes of ~1000 rows: 4000 appended keys per query, slower than 10 probes with seeds).
        auto evaluate = [&](uint32_t stride, bool& soft_thin) -> bool {
            plan.sample_stride = stride;
            const double ss = stride;
            double frac = 1.0 / ss;
            for (size_t gi = 0; gi < plan.groups_s.size(); ++gi) {
                const ScanGroup& gs = plan.groups_s[gi];
                const double t64 = std::ceil((double) gs.n_tiles * c->shape.rw / tile_rows);
                const double per_block = std::ceil(t64 / gs.n_blocks);
                const double sampled = std::min(t64, gs.n_blocks * std::ceil(per_block / ss));
                if (t64 > 0) frac = std::max(frac, sampled / t64);
            }
            const double lambda = (double) plan.keep * frac;
            const uint32_t seed_m = (uint32_t) std::ceil(lambda + 6.0 * std::sqrt(lambda)) + 4;
            plan.kp_frac = (float) lambda;
            bool good = seed_m <= GQ_SAMPLE_CAP / 4;
            // fine passes: any query whose per-column minima (one per 64 rows of a sampled tile; K2g: one per 128) would be
            // fewer than 4 m: one minimum per lane instead (4 x as many)
            est.assign((size_t) nq, 0.0);
            for (size_t gi = 0; gi < plan.groups_s.size(); ++gi) {
                ScanGroup& gs = plan.groups_s[gi];
                bool fine = false;
                for (uint32_t qi = 0; qi < gs.q_count; ++qi) {
                    const vsr_filter* f = fof(plan.q_slots[gs.q_begin + qi]);
                    const double allowed = f ? (double) f->allowed_rows : (double) c->n;
                    fine |= allowed / ((plan.k2g ? 128.0 : 64.0) * ss) < 4.0 * seed_m;
                }
                gs.partial_begin = fine ? 1u : 0u;                     // (K2w / K2g have no partial lists: the field carries the flag)
                const double t64 = std::ceil((double) gs.n_tiles * c->shape.rw / tile_rows);
                const double per_block = std::ceil(t64 / gs.n_blocks);
                const double sampled = std::min(t64, gs.n_blocks * std::ceil(per_block / ss));
                const uint32_t ngt = (gs.q_count + 15) / 16;
                const double waves_per_col = plan.k2g ? 2.0 : ngt == 1 ? 4.0 : ngt == 2 ? 2.0 : 1.0;     // row split (vsr_mfmaw.h)
                const double entries_per_tile = waves_per_col * (fine ? 4.0 : 1.0);
                const double rows_per_entry = tile_rows / entries_per_tile;
                const double p_entry = std::min(1.0, gdens_s[gi] * rows_per_entry);    // a bitmap may leave an entry without rows
                for (uint32_t qi = 0; qi < gs.q_count; ++qi) est[plan.q_slots[gs.q_begin + qi]] += sampled * entries_per_tile * p_entry;
            }
            soft_thin = false;
            for (uint32_t q = 0; q < (uint32_t) nq; ++q) {
                const vsr_filter* f = fof(q);
                const int64_t allowed = f ? f->allowed_rows : c->n;
                // the sample must be thick enough to reach rank m (with a margin where a bitmap makes the count random),
                // unless all of the query's rows fit its buffer anyway
                const bool exact_count = !f || f->allowed_rows == f->scanned_rows;
                const bool thin = est[q] < (exact_count ? 1.25 * seed_m + 8.0 : 2.0 * seed_m);
                if (allowed > (int64_t) GQ_CAP && thin) good = false;
                if (allowed > (int64_t) (8 * plan.keep) && allowed > 2048 && thin) soft_thin = true;
            }
            return good;
        };
        uint32_t first_ok = 0, chosen = 0;
        for (uint32_t stride = ctx->sample_stride; !ok && !chosen && stride >= 2; stride /= 2) {
            bool soft = false;
            if (evaluate(stride, soft)) {
                if (!first_ok) first_ok = stride;
                if (!soft) chosen = stride;
            }
        }
        if (!ok && (chosen || first_ok)) {
            bool soft = false;
            ok = evaluate(chosen ? chosen : first_ok, soft);      // (sets the plan's stride, seed fraction and the groups' flags)
        }
        plan.selq.resize((size_t) nq);
        for (uint32_t q = 0; q < (uint32_t) nq; ++q) {
            const vsr_filter* f = fof(q);
            const int64_t allowed = f ? f->allowed_rows : c->n;
            SelectQuery sq;
            sq.ids_begin = 0;
            sq.n_lists = 0;
            sq.out_slot = q;
            sq.dst_list = SEL_FINAL;
            sq.allowed = (uint32_t) std::min<int64_t>(allowed, 0xFFFFFFFFll);
            sq.pad = 0;
            plan.selq[q] = sq;
        }
        return ok;
    }
    for (int q = 0; q < nq; ++q) loff[(size_t) q + 1] += loff[(size_t) q];
    lcur.assign(loff.begin(), loff.end() - 1);
    lids.resize(loff[(size_t) nq]);
    const bool same_blocks = seed_div == 1;                // sample lists mirror the main lists one to one
    if (!same_blocks) lids_s.resize(loff[(size_t) nq]);
    static thread_local std::vector<uint32_t> lcnt_s;
    lcnt_s.assign((size_t) nq, 0);
    for (size_t gi = 0; gi < plan.groups.size(); ++gi) {
        const ScanGroup& g = plan.groups[gi];
        const ScanGroup& gs = plan.groups_s[gi];
        for (uint32_t qi = 0; qi < g.q_count; ++qi) {
            const uint32_t slot = plan.q_slots[g.q_begin + qi];
            uint32_t at = lcur[slot];
            for (uint32_t b = 0; b < g.n_blocks; ++b) lids[at + b] = g.partial_begin + qi * g.n_blocks + b;
            if (!same_blocks) {
                // the sample pass has at most as many lists: kept left-packed in the same CSR range
                uint32_t as = loff[slot] + lcnt_s[slot];
                for (uint32_t b = 0; b < gs.n_blocks; ++b) lids_s[as + b] = gs.partial_begin + qi * gs.n_blocks + b;
                lcnt_s[slot] += gs.n_blocks;
            }
            lcur[slot] = at + g.n_blocks;
        }
    }

    // K5 items.  Queries with many partial lists get a first level of fan-in-list merges.  When every query fits two
    // levels of the wave-per-query selection (<= 4096 keys per item) that kernel and its smaller fan-in are used.
    uint32_t most_lists = 0;
    for (int q = 0; q < nq; ++q) most_lists = std::max(most_lists, loff[(size_t) q + 1] - loff[(size_t) q]);
    const uint32_t wave_fanin = select_wave_fanin(plan.keep);
    plan.sel_wave = wave_fanin > 0 && (uint64_t) most_lists <= (uint64_t) wave_fanin * wave_fanin;
    const uint32_t fanin = plan.sel_wave ? wave_fanin : SEL_FANIN;
    plan.selq.resize((size_t) nq);
    plan.seedq.resize((size_t) nq);
    plan.list_ids.reserve(lids.size() * 2 + 64);
    static thread_local std::vector<uint32_t> level2;
    for (uint32_t q = 0; q < (uint32_t) nq; ++q) {
        const vsr_filter* f = fof(q);
        const uint32_t allowed = (uint32_t) std::min<int64_t>(f ? f->allowed_rows : c->n, 0xFFFFFFFFll);
        const uint32_t* ls = lids.data() + loff[q];
        uint32_t n_ls = loff[q + 1] - loff[q];
        if (n_ls > fanin) {
            level2.clear();
            for (uint32_t j = 0; j < n_ls; j += fanin) {
                SelectQuery s1;
                s1.ids_begin = (uint32_t) plan.list_ids.size();
                s1.n_lists = std::min<uint32_t>(fanin, n_ls - j);
                s1.out_slot = 0;
                s1.dst_list = plan.n_partial;
                s1.allowed = 0;
                s1.pad = 0;
                plan.list_ids.insert(plan.list_ids.end(), ls + j, ls + j + s1.n_lists);
                plan.sel1.push_back(s1);
                level2.push_back(plan.n_partial++);
            }
            ls = level2.data();
            n_ls = (uint32_t) level2.size();
        }
        SelectQuery sq;
        sq.ids_begin = (uint32_t) plan.list_ids.size();
        sq.n_lists = n_ls;
        sq.out_slot = q;
        sq.dst_list = SEL_FINAL;
        sq.allowed = allowed;
        sq.pad = 0;
        plan.list_ids.insert(plan.list_ids.end(), ls, ls + n_ls);
        plan.selq[q] = sq;
        SelectQuery sd = sq;                                // seed item: the sample pass's lists of the same query
        sd.ids_begin = (uint32_t) plan.list_ids.size();
        sd.dst_list = SEL_SEED;
        const uint32_t* sl = same_blocks ? lids.data() + loff[q] : lids_s.data() + loff[q];
        sd.n_lists = same_blocks ? loff[q + 1] - loff[q] : lcnt_s[q];     // same_blocks: identical list numbering
        if (plan.sel_wave && sd.n_lists > 64) sd.n_lists = 64;            // a subset of the sample only loosens the seed
        plan.list_ids.insert(plan.list_ids.end(), sl, sl + sd.n_lists);
        plan.seedq[q] = sd;
    }
    if (plan.k2) {          // the final K5 of every query writes its kp screening survivors as list rerank_base + slot
        plan.rerank_base = plan.n_partial;
        for (size_t s = 0; s < plan.selq.size(); ++s) plan.selq[s].dst_list = plan.rerank_base + (uint32_t) s;
        plan.n_partial += (uint32_t) plan.selq.size();
    }
    return true;
}

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// the instantiation the main scan launch of a plan resolves to (bench.py reports it beside the roofline)
static std::string scan_kernel_name(const Plan& plan, const vsr_corpus* c, int metric, bool k2i = false)
{
    static const char* mname[] = {"L2", "IP", "COSINE", "L1"};
    char buf[160];
    const uint32_t nstage = (c->stride4 + 15) / 16;
    if (plan.k2g)
        snprintf(buf, sizeof buf, "vsr::gemm_screen_kernel<%s, SAMPLE=false> (K2g, bf16 coarse planes)", mname[metric]);
    else if (plan.k2w && plan.int8 && k2i)
        snprintf(buf, sizeof buf, "vsr::i8_stream_kernel<NQG=4> (K2i, int8 planes, %s)", mname[metric]);
    else if (plan.k2w && plan.int8)
        snprintf(buf, sizeof buf, "vsr::mfma_wide_kernel<%s, NCH=1, SAMPLE=false, PL=int8> (K2w, int8 planes)", mname[metric]);
    else if (plan.k2w)
        snprintf(buf, sizeof buf, "vsr::mfma_wide_kernel<%s, NCH=%u, SAMPLE=false, HO=%s> (K2w, bf16 %s planes)", mname[metric],
                 c->pstride4 / 16, c->scr_has_mid ? "false" : "true", c->scr_has_mid ? "hi+mid" : "hi-only");
    else if (plan.k2)
        snprintf(buf, sizeof buf, "vsr::mfma_scan_kernel<%s, NSTR=%d, SAMPLE=false, NG=%d> (K2)", mname[metric],
                 nstage > 4 ? 0 : 4, plan.qmax > 16 ? 2 : 1);
    else if (plan.mq)
        snprintf(buf, sizeof buf, "vsr::mq_scan_kernel<%s, SAMPLE=false> (K1m)", mname[metric]);
    else
        snprintf(buf, sizeof buf, "vsr::scan_kernel<%s, LPR=%d, C=%d, R=%d, QI=%d> (K1)", mname[metric], c->shape.lpr,
                 c->shape.c, c->shape.r, plan.qi);
    return buf;
}

// K2w launch sequence (plan.k2w): staging -> sample pass -> threshold seeds -> main pass -> select + exact re-rank.
// Five launches, one candidate buffer per query, no partial lists (see vsr_mfmaw.h).
static int search_wide(vsr_ctx* ctx, vsr_corpus* c, const Plan& plan, const float* h_queries, const float* d_queries, int nq,
                       int dim, int k, int metric, int64_t* d_blk, int32_t* d_doc, int64_t* d_row, float* d_dist, int32_t* d_cnt,
                       uint64_t* d_keys)
{
    const uint32_t kp = plan.keep;
    const vsr_corpus* idc = c->base ? c->base : c;          // identity arrays and re-rank rows (c may be a list-ordered view)
    const size_t qfloats = (size_t) c->stride4 * 4;
    const size_t q_pstride = c->scr_has_mid ? c->pstride4 : 2 * (size_t) c->pstride4;    // query planes keep hi and mid
    // staging block: [queries | q_norm2 | query planes || scan groups | sample groups | pass query slots | per-query items | block map]
    const size_t off_q = 0;
    const size_t off_qn = align_up(off_q + (size_t) nq * qfloats * sizeof(float), 256);
    const size_t off_qp = align_up(off_qn + (size_t) nq * sizeof(float), 256);
    const size_t off_qc = align_up(off_qp + (size_t) nq * q_pstride * 16, 256);            // K2g: coarse query planes
    const size_t off_q8 = align_up(off_qc + (plan.k2g ? coarse_plane_u4((uint64_t) nq, c->cstride4) * 16 : 0), 256);   // int8 query planes, |q-128|^2, validity
    const size_t off_qn8 = align_up(off_q8 + (plan.int8 ? (size_t) nq * 128 : 0), 256);
    const size_t off_qb = align_up(off_qn8 + (plan.int8 ? (size_t) nq * sizeof(float) : 0), 256);
    const size_t off_g = align_up(off_qb + (plan.int8 ? (size_t) nq * sizeof(uint32_t) : 0), 256);    // copied from here on
    const size_t off_gs = align_up(off_g + plan.groups.size() * sizeof(ScanGroup), 256);
    const size_t off_qs = align_up(off_gs + plan.groups_s.size() * sizeof(ScanGroup), 256);
    const size_t off_sq = align_up(off_qs + plan.q_slots.size() * sizeof(uint32_t), 256);
    const size_t off_bm = align_up(off_sq + plan.selq.size() * sizeof(SelectQuery), 256);
    const size_t total = align_up(off_bm + plan.block_map.size() * sizeof(uint2), 256);
    // the pinned block holds only what the host writes: the queries (host API) and the descriptors
    const size_t h_q_bytes = h_queries ? align_up((size_t) nq * qfloats * sizeof(float), 256) : 0;
    const size_t h_total = h_q_bytes + (total - off_g);

    int rc;
    if (ctx->desc_pending) {       // the previous batch's staging kernel still owns the pinned block
        const auto w0 = std::chrono::steady_clock::now();
        HIPCHK(hipEventSynchronize(ctx->desc_done));
        ctx->desc_pending = false;
        {
            const double waited = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - w0).count();
            ctx->host_us[1] += waited;
            ctx->stats.host_wait_ms += waited * 1e-3;
        }
    }
    if ((rc = ctx->h_desc.reserve(h_total))) return rc;
    if ((rc = ctx->d_desc.reserve(total))) return rc;
    if ((rc = ctx->d_flags.reserve((size_t) nq * sizeof(int32_t)))) return rc;
    if ((rc = ctx->d_tau.reserve((size_t) nq * sizeof(uint64_t)))) return rc;
    if ((rc = ctx->d_cand.reserve((size_t) nq * GQ_CAP * sizeof(uint64_t)))) return rc;
    if ((rc = ctx->d_samp.reserve((size_t) nq * GQ_SAMPLE_CAP * sizeof(uint64_t)))) return rc;
    if ((rc = ctx->d_qcnt.reserve((size_t) 2 * nq * sizeof(uint32_t)))) return rc;
    char* hs = ctx->h_desc.as<char>();
    char* ds = ctx->d_desc.as<char>();
    if (h_queries) {
        float* hq = reinterpret_cast<float*>(hs);
        for (int s = 0; s < nq; ++s) {
            float* dst = hq + (size_t) s * qfloats;
            memcpy(dst, h_queries + (size_t) s * dim, (size_t) dim * sizeof(float));
            for (size_t j = (size_t) dim; j < qfloats; ++j) dst[j] = 0.0f;
        }
    }
    char* hd_desc = hs + h_q_bytes;                         // host image of [off_g, total)
    memcpy(hd_desc + (off_g - off_g), plan.groups.data(), plan.groups.size() * sizeof(ScanGroup));
    memcpy(hd_desc + (off_gs - off_g), plan.groups_s.data(), plan.groups_s.size() * sizeof(ScanGroup));
    memcpy(hd_desc + (off_qs - off_g), plan.q_slots.data(), plan.q_slots.size() * sizeof(uint32_t));
    memcpy(hd_desc + (off_sq - off_g), plan.selq.data(), plan.selq.size() * sizeof(SelectQuery));
    memcpy(hd_desc + (off_bm - off_g), plan.block_map.data(), plan.block_map.size() * sizeof(uint2));

    hipEvent_t w0 = nullptr, w1 = nullptr;                  // profiling level 1: the whole search on the device
    if (ctx->profiling == 1) {
        w0 = take_event(ctx);
        w1 = take_event(ctx);
        HIPCHK(hipEventRecord(w0, ctx->stream));
    }
    uint32_t* qcnt = ctx->d_qcnt.as<uint32_t>();
    uint32_t* scnt = qcnt + nq;
    {
        StageParams st{};
        const char* hd = reinterpret_cast<const char*>(ctx->h_desc.dp);
        st.src16 = reinterpret_cast<const uint4*>(hd + h_q_bytes);
        st.dst16 = reinterpret_cast<uint4*>(ds + off_g);
        st.n16 = (uint32_t) ((total - off_g) / 16);
        st.q_src = d_queries ? d_queries : reinterpret_cast<const float*>(hd);
        st.q_stride = d_queries ? (uint32_t) dim : (uint32_t) qfloats;
        st.q_dst = reinterpret_cast<float*>(ds + off_q);
        st.dim = (uint32_t) dim;
        st.qfloats = (uint32_t) qfloats;
        st.nq = (uint32_t) nq;
        st.q_norm2 = reinterpret_cast<float*>(ds + off_qn);
        st.q_scr = reinterpret_cast<uint4*>(ds + off_qp);
        st.pstride4 = c->pstride4;
        st.plane_ho = c->scr_has_mid ? 0u : 1u;
        st.flags = ctx->d_flags.as<int32_t>();
        st.tau = ctx->d_tau.as<uint64_t>();
        st.qcnt = qcnt;
        st.scnt = scnt;
        if (plan.k2g) {
            st.q_scr = nullptr;                             // only the coarse planes are read
            st.q_scr_c = reinterpret_cast<uint4*>(ds + off_qc);
            st.cstride4 = c->cstride4;
        }
        if (plan.int8) {
            if (!ctx->h_q8.p) {
                if ((rc = ctx->h_q8.reserve(64))) return rc;
                memset(ctx->h_q8.p, 0, 64);
            }
            st.q_scr8 = reinterpret_cast<uint4*>(ds + off_q8);
            st.q_norm2_8 = reinterpret_cast<float*>(ds + off_qn8);
            st.q8_bad = reinterpret_cast<uint32_t*>(ds + off_qb);
            st.q8_bad_host = reinterpret_cast<uint32_t*>(ctx->h_q8.dp);
        }
        HIPCHK(launch_stage(st, ctx->stream));
    }
    HIPCHK(hipEventRecord(ctx->desc_done, ctx->stream));
    ctx->desc_pending = true;

    ScanParams sp{};
    sp.rows = c->d_rows;
    sp.norm2 = c->d_norm2;
    sp.n_rows = (uint32_t) c->n;
    sp.stride4 = c->stride4;
    sp.queries = reinterpret_cast<const float*>(ds + off_q);
    sp.q_norm2 = reinterpret_cast<const float*>(ds + off_qn);
    sp.scr = c->d_scr;
    sp.q_scr = reinterpret_cast<const uint4*>(ds + off_qp);
    sp.pstride4 = c->pstride4;
    sp.plane_ho = c->scr_has_mid ? 0u : 1u;
    if (plan.int8) {                                        // same kernel, int8 planes: 8 chunks per row, their own norms
        sp.norm2 = c->d_norm2_8;
        sp.q_norm2 = reinterpret_cast<const float*>(ds + off_qn8);
        sp.scr = c->d_scr8;
        sp.q_scr = reinterpret_cast<const uint4*>(ds + off_q8);
        sp.pstride4 = 8;
        sp.plane_ho = 2u;
    }
    if (plan.k2g) {
        sp.scr_c = c->d_scr_c;
        sp.q_scr_c = reinterpret_cast<const uint4*>(ds + off_qc);
        sp.cstride4 = c->cstride4;
    }
    auto launch_pass = [&](uint32_t blocks, hipStream_t st) { return plan.k2g ? launch_gemm(sp, metric, blocks, st) : launch_mfmaw(sp, metric, blocks, st); };
    sp.q_slots = reinterpret_cast<const uint32_t*>(ds + off_qs);
    sp.kp = sp.k = kp;
    sp.qmax = plan.qmax;
    sp.rw = (uint32_t) c->shape.rw;
    sp.err = reinterpret_cast<uint32_t*>(ctx->d_flag_total) + 4;
    sp.ones = reinterpret_cast<const uint64_t*>(reinterpret_cast<const char*>(ctx->d_flag_total) + 32);
    sp.rank = c->d_rank;

    if (plan.n_blocks) {
        // ---- sample pass: every sample_stride-th tile, open threshold, into the queries' sample buffers ----
        hipEvent_t a0 = nullptr, a1 = nullptr;              // profiling level 1: sample pass + seed select together
        if (ctx->profiling == 1) {
            a0 = take_event(ctx); a1 = take_event(ctx);
            HIPCHK(hipEventRecord(a0, ctx->stream));
        }
        sp.groups = reinterpret_cast<const ScanGroup*>(ds + off_gs);
        sp.n_groups = (uint32_t) plan.groups_s.size();
        sp.sample_stride = plan.sample_stride;
        sp.tau_init = nullptr;
        sp.block_map = nullptr;
        sp.qcand = ctx->d_samp.as<uint64_t>();
        sp.qcnt = scnt;
        sp.capq = GQ_SAMPLE_CAP;
        sp.k2i = plan.k2i_sample ? 2u : 0u;                 // (bit 1: the sample launch on K2i; bit 0: the main launch)
        HIPCHK(launch_pass(plan.n_blocks_s, ctx->stream));
        HIPCHK(launch_seed_select(ctx->d_samp.as<uint64_t>(), scnt, GQ_SAMPLE_CAP, plan.kp_frac, ctx->d_tau.as<uint64_t>(),
                                  (uint32_t) nq, ctx->stream));
        if (a0) {
            HIPCHK(hipEventRecord(a1, ctx->stream));
            ctx->pending.push_back({a0, a1, 3});
        }
        // ---- main pass ----
        hipStream_t main_stream = ctx->stream;
        if (ctx->scan_lane) {                               // the main launch goes to the corpus's lane, behind this batch's seeds
            if (!c->scan_stream) HIPCHK(hipStreamCreateWithFlags(&c->scan_stream, hipStreamNonBlocking));
            if (!ctx->lane_in) {
                HIPCHK(hipEventCreateWithFlags(&ctx->lane_in, hipEventDisableTiming));
                HIPCHK(hipEventCreateWithFlags(&ctx->lane_out, hipEventDisableTiming));
            }
            HIPCHK(hipEventRecord(ctx->lane_in, ctx->stream));
            HIPCHK(hipStreamWaitEvent(c->scan_stream, ctx->lane_in, 0));
            main_stream = c->scan_stream;
        }
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (ctx->profiling) {
            e0 = take_event(ctx);
            e1 = take_event(ctx);
            HIPCHK(hipEventRecord(e0, main_stream));
        }
        sp.groups = reinterpret_cast<const ScanGroup*>(ds + off_g);
        sp.n_groups = (uint32_t) plan.groups.size();
        sp.sample_stride = 1;
        {
            // survivors a wave-tile (64 rows x 16 queries) can expect: what a query admits over the rows it scans
            const double f = plan.kp_frac > 0 ? plan.kp_frac / kp : 1.0 / 16;
            const double admitted = kp + 6.0 * std::sqrt(kp / f) + 4.0 / f;
            const double rows_per_query = (double) plan.scan_pairs / std::max(1, nq);
            sp.epi = rows_per_query > 0 && 1024.0 * admitted / rows_per_query <= 4.0 ? 1u : 0u;
        }
        if (ctx->force_epi >= 0) sp.epi = (uint32_t) ctx->force_epi;
        if (plan.int8 && plan.qmax > 64) sp.epi = 1u;       // 128-column passes exist on K2i only (its parking area takes bursts)
        sp.k2i = plan.int8 && !plan.k2g && sp.epi == 1 && !ctx->no_k2i && sp.rw == 16 && sp.qmax <= 128 ? 1u : 0u;
        ctx->last_k2i = sp.k2i != 0;
        sp.tau_init = ctx->d_tau.as<uint64_t>();
        sp.qcand = ctx->d_cand.as<uint64_t>();
        sp.qcnt = qcnt;
        sp.capq = GQ_CAP;
        if (!plan.block_map.empty() && !ctx->no_xcd_map) sp.block_map = reinterpret_cast<const uint2*>(ds + off_bm);
        HIPCHK(launch_pass(sp.block_map ? plan.n_launch : plan.n_blocks, main_stream));
        if (e0) {
            HIPCHK(hipEventRecord(e1, main_stream));
            ctx->pending.push_back({e0, e1, 1});
        }
        if (ctx->scan_lane) {                               // the selection waits for the lane
            HIPCHK(hipEventRecord(ctx->lane_out, main_stream));
            HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->lane_out, 0));
        }
        ctx->last_kernel = scan_kernel_name(plan, c, metric, ctx->last_k2i);
        ctx->stats.scan_bytes[1] += plan.scan_bytes;
        ctx->stats.scan_rows[1] += plan.scan_rows;
        ctx->stats.scan_pairs[1] += plan.scan_pairs;
        ctx->stats.unique_rows[1] += plan.unique_rows;
    }

    hipEvent_t s0 = nullptr, s1 = nullptr;
    if (ctx->profiling == 1) {
        s0 = take_event(ctx);
        s1 = take_event(ctx);
        HIPCHK(hipEventRecord(s0, ctx->stream));
    }
    RerankParams rr{};
    rr.queries = reinterpret_cast<const SelectQuery*>(ds + off_sq);
    rr.rows = idc->d_rows;
    rr.stride4 = c->stride4;
    rr.queries_f = reinterpret_cast<const float*>(ds + off_q);
    rr.kp = kp;
    rr.k = (uint32_t) k;
    rr.metric = metric;
    rr.dim = c->dim;
    rr.norm2_max = idc->d_norm2_max;
    rr.row_offset = (uint32_t) idc->row_offset;
    rr.block_ids = idc->d_block;
    rr.doc_ids = idc->d_doc;
    rr.orig_rows = idc->d_orig;
    rr.out_block = d_blk;
    rr.out_doc = d_doc;
    rr.out_row = d_row;
    rr.out_dist = d_dist;
    rr.out_keys = d_keys;
    rr.out_count = d_cnt;
    rr.qcand = ctx->d_cand.as<uint64_t>();
    rr.qcnt = qcnt;
    rr.capq = GQ_CAP;
    rr.err_g = plan.k2g ? coarse_err_g(c->dim) : plane_err_g(c->dim);
    rr.err_tight = plan.k2g ? 1u : 0u;
    rr.qbad = plan.int8 ? reinterpret_cast<const uint32_t*>(ds + off_qb) : nullptr;
    rr.exact_screen = plan.int8 ? 1u : 0u;
    rr.seeded = 1;
    rr.tau_init = ctx->d_tau.as<uint64_t>();
    rr.out_flags = ctx->d_flags.as<int32_t>();
    rr.flagged_total = ctx->d_flag_total;
    HIPCHK(launch_select_rerank(rr, (uint32_t) nq, ctx->stream));
    if (s0) {
        HIPCHK(hipEventRecord(s1, ctx->stream));
        ctx->pending.push_back({s0, s1, 2});
    }
    if (w0) {
        HIPCHK(hipEventRecord(w1, ctx->stream));
        ctx->pending.push_back({w0, w1, 5});
    }
    ctx->stats.queries += nq;
    return VSR_OK;
}

// Shared by the host and device entry points.  d_queries == nullptr: queries come from `h_queries`.
// `ctx` is the session the search runs in (stream, workspaces, counters): the corpus's own context, or another context
// of the same device (vsr_search_device_on) so that two batches over one corpus can be in flight at once.
static int search_impl(vsr_ctx* ctx, vsr_corpus* c, const float* h_queries, const float* d_queries, int nq, int dim, int k,
                       int metric, const vsr_filter* const* filters, int64_t* d_blk, int32_t* d_doc, int64_t* d_row,
                       float* d_dist, int32_t* d_cnt, uint64_t* d_keys, int level)
{
    // level: 2 = every screening tier (coarse planes for wide passes over long rows, K2g), 1 = fine planes only (K2w / K2),
    // 0 = exact kernels only.  A query flagged at one level is re-run at the next lower one (host_search,
    // vsr_search_device_exact).
    const bool allow_screening = level >= 1;
    ctx->screen_level = level;
    const auto h0 = std::chrono::steady_clock::now();
    struct HostTimer {
        vsr_ctx* ctx;
        std::chrono::steady_clock::time_point t0;
        ~HostTimer()
        {
            const double spent = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            ctx->host_us[2] += spent;
            ctx->stats.host_ms += spent * 1e-3;
            ctx->host_calls++;
        }
    } host_timer{ctx, h0};
    // int8 planes (SIFT-like corpora): host queries are checked here; device-resident queries only under the caller's
    // hint (vsr_set_query_hint), validated by the staging kernel -- a violating query is flagged, and the hint is dropped
    // once the kernel's pinned word shows one (read without synchronising: at worst a batch late)
    if (ctx->q8_ok && ctx->h_q8.p && *reinterpret_cast<volatile uint32_t*>(ctx->h_q8.p)) ctx->q8_ok = false;
    ctx->int8_this_call = false;
    if (c->d_scr8 && metric == VSR_METRIC_L2 && allow_screening) {
        if (h_queries) {
            bool ok = true;
            const size_t total = (size_t) nq * dim;
            for (size_t i = 0; i < total && ok; ++i) {
                const float v = h_queries[i];
                ok = v >= 0.0f && v <= 255.0f && v == floorf(v);
            }
            ctx->int8_this_call = ok;
        } else {
            ctx->int8_this_call = ctx->hint_u8 && ctx->q8_ok;
        }
    }
    static thread_local Plan plan;
    plan.reset();
    if (!make_plan(ctx, c, nq, k, metric, allow_screening, true, true, filters, plan)) {
        const bool was_gemm = plan.k2g;
        plan.reset();                                       // K2g could not be seeded safely: K2w; K2w neither: legacy shared passes
        if (!was_gemm || !make_plan(ctx, c, nq, k, metric, allow_screening, true, false, filters, plan)) {
            plan.reset();
            (void) make_plan(ctx, c, nq, k, metric, allow_screening, false, false, filters, plan);
        }
    }
    ctx->last_coarse = plan.k2g;
    if (plan.k2w) {
        ctx->host_us[0] += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - h0).count();
        return search_wide(ctx, c, plan, h_queries, d_queries, nq, dim, k, metric, d_blk, d_doc, d_row, d_dist, d_cnt, d_keys);
    }
    ctx->host_us[0] += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - h0).count();

    const uint32_t kp = plan.keep;
    const vsr_corpus* idc = c->base ? c->base : c;          // identity arrays and re-rank rows (c may be a list-ordered view)
    const size_t qfloats = (size_t) c->stride4 * 4;

    // ---- one query, one pass: the whole search in ONE launch (FusedTail, vsr_device.h) ----
    // conditions: the query already lies in device memory in the row layout (d a multiple of 4: no padding to add),
    // no |q|^2 needed (not cosine), one plain pass on K1, k small enough for the workgroup's LDS top-k buffer
    if (nq == 1 && !ctx->no_fused && plan.groups.size() == 1 && plan.qi == 1 && !plan.k2 && !plan.mq &&
        dim % 4 == 0 && metric != VSR_METRIC_COSINE && plan.groups[0].n_blocks <= 64u * 64u && ctx->profiling != 1) {
        const uint32_t cap = std::max<uint32_t>(4096, scan_cap_for_k((int) kp, c->dim));   // the merge's LDS layout (vsr_scan.h, FUSED_*)
        const ScanGroup& g = plan.groups[0];
        const uint32_t per_merge = kp ? 8192u / kp : 0u;    // lists one merge takes (16 keys per thread, vsr_scan.h)
        uint32_t fan = per_merge ? std::max<uint32_t>(16, (g.n_blocks + per_merge - 1) / per_merge) : 0u;
        if (ctx->fused_fan >= 16 && per_merge >= 16) fan = std::min<uint32_t>((uint32_t) ctx->fused_fan, per_merge);   // development knob (>= 16: d_done holds 256 group counters)
        if (kp <= 512 && fan && fan <= per_merge) {
            int rc;
            const uint32_t n_g = (g.n_blocks + fan - 1) / fan;
            if ((rc = ctx->d_partial.reserve((size_t) (g.n_blocks + n_g) * kp * sizeof(uint64_t)))) return rc;
            if ((rc = ctx->d_flags.reserve(sizeof(int32_t)))) return rc;
            if (!ctx->d_done.p) {
                if ((rc = ctx->d_done.reserve((size_t) (1 + 4096 / 16 + 64) * sizeof(uint32_t)))) return rc;
                HIPCHK(hipMemsetAsync(ctx->d_done.p, 0, ctx->d_done.cap, ctx->stream));
            }
            const float* q_dev = d_queries;
            if (!q_dev) {
                // host query: the kernel reads it straight out of the pinned staging block (512 bytes, cached after the
                // first workgroup), which the previous call's kernel must have left
                if (ctx->desc_pending) {
                    HIPCHK(hipEventSynchronize(ctx->desc_done));
                    ctx->desc_pending = false;
                }
                if ((rc = ctx->h_desc.reserve(qfloats * sizeof(float)))) return rc;
                memcpy(ctx->h_desc.p, h_queries, (size_t) dim * sizeof(float));
                q_dev = reinterpret_cast<const float*>(ctx->h_desc.dp);
            }
            ScanParams sp{};
            sp.rows = c->d_rows;
            sp.norm2 = c->d_norm2;
            sp.n_rows = (uint32_t) c->n;
            sp.stride4 = c->stride4;
            sp.queries = q_dev;
            sp.partial = ctx->d_partial.as<uint64_t>();
            sp.kp = sp.k = kp;
            sp.cap = cap;
            sp.qmax = 1;
            sp.rw = (uint32_t) c->shape.rw;
            sp.err = reinterpret_cast<uint32_t*>(ctx->d_flag_total) + 4;
            sp.ones = reinterpret_cast<const uint64_t*>(reinterpret_cast<const char*>(ctx->d_flag_total) + 32);
            sp.rank = c->d_rank;
            sp.sample_stride = 1;
            sp.n_groups = 1;
            sp.fused.enable = 1;
            sp.fused.fan = fan;
            sp.fused.group = g;
            sp.fused.group.block_begin = 0;
            sp.fused.group.partial_begin = 0;
            sp.fused.done = ctx->d_done.as<uint32_t>();
            sp.fused.block_ids = idc->d_block;
            sp.fused.doc_ids = idc->d_doc;
            sp.fused.orig_rows = idc->d_orig;
            sp.fused.out_block = d_blk;
            sp.fused.out_doc = d_doc;
            sp.fused.out_row = d_row;
            sp.fused.out_dist = d_dist;
            sp.fused.out_keys = d_keys;
            sp.fused.out_count = d_cnt;
            sp.fused.out_flag = ctx->d_flags.as<int32_t>();
            sp.fused.row_offset = (uint32_t) idc->row_offset;
            sp.fused.metric = metric;
            if (ctx->fused_dbg) {                           // development: timestamps of the finishing workgroup, printed by the next call
                if (!ctx->d_dbg.p) {
                    if ((rc = ctx->d_dbg.reserve(64))) return rc;
                    if ((rc = ctx->h_dbg.reserve(64))) return rc;
                } else {
                    HIPCHK(hipStreamSynchronize(ctx->stream));
                    const uint64_t* t = ctx->h_dbg.as<uint64_t>();
                    HIPCHK(hipMemcpy(ctx->h_dbg.p, ctx->d_dbg.p, 64, hipMemcpyDeviceToHost));
                    fprintf(stderr, "fused_dbg us: scan %.1f compact+publish %.1f wait %.1f merge1 %.1f publish+wait %.1f merge2 %.1f tail %.1f total %.1f\n",
                            (t[1] - t[0]) / 100.0, (t[2] - t[1]) / 100.0, t[3] ? (t[3] - t[2]) / 100.0 : 0.0, t[3] ? (t[4] - t[3]) / 100.0 : 0.0,
                            (t[5] - (t[4] ? t[4] : t[2])) / 100.0, (t[6] - t[5]) / 100.0, (t[7] - t[6]) / 100.0, (t[7] - t[0]) / 100.0);
                }
                sp.fused.dbg = ctx->d_dbg.as<uint64_t>();
            }
            hipEvent_t e0 = nullptr, e1 = nullptr;
            if (ctx->profiling) {
                e0 = take_event(ctx);
                e1 = take_event(ctx);
                HIPCHK(hipEventRecord(e0, ctx->stream));
            }
            // SIFT-like corpus and query (u8-exact, L2): the same launch over the int8 planes -- a quarter of the bytes per
            // row, identical distances (vsr_scan.h, scan8_fused_kernel)
            const bool scan8 = c->d_scr8 && !c->base && metric == VSR_METRIC_L2 && ctx->int8_this_call && c->shape.rw == 16 &&
                               !ctx->no_scan8;
            if (scan8) {
                if (!ctx->h_q8.p) {
                    if ((rc = ctx->h_q8.reserve(64))) return rc;
                    memset(ctx->h_q8.p, 0, 64);
                }
                sp.scr = c->d_scr8;
                sp.norm2 = c->d_norm2_8;
                sp.pstride4 = 8;
                sp.plane_ho = 2u;
                sp.fused.flag_total = ctx->d_flag_total;
                HIPCHK(launch_scan8_fused(sp, (uint32_t) dim, reinterpret_cast<uint32_t*>(ctx->h_q8.dp), g.n_blocks, ctx->stream));
            } else
                HIPCHK(launch_scan(sp, metric, c->dim, 1, g.n_blocks, ctx->stream));
            if (!d_queries) {
                HIPCHK(hipEventRecord(ctx->desc_done, ctx->stream));
                ctx->desc_pending = true;
            }
            if (e0) {
                HIPCHK(hipEventRecord(e1, ctx->stream));
                ctx->pending.push_back({e0, e1, 0});
            }
            ctx->last_kernel = scan8 ? std::string("vsr::scan8_fused_kernel (K1 on the int8 planes) + in-kernel merge")
                                     : scan_kernel_name(plan, c, metric) + " + in-kernel merge";
            ctx->stats.scan_bytes[0] += plan.scan_bytes;
            ctx->stats.scan_rows[0] += plan.scan_rows;
            ctx->stats.scan_pairs[0] += plan.scan_pairs;
            ctx->stats.unique_rows[0] += plan.unique_rows;
            ctx->stats.queries += 1;
            return VSR_OK;
        }
    }
    // one staging block: [queries | q_norm2 | scan groups | sample groups | pass query slots | K5 items | list ids]
    const size_t off_q = 0;
    const size_t off_qn = align_up(off_q + (size_t) nq * qfloats * sizeof(float), 256);
    const size_t off_g = align_up(off_qn + (size_t) nq * sizeof(float), 256);
    const size_t off_gs = align_up(off_g + plan.groups.size() * sizeof(ScanGroup), 256);
    const size_t off_qs = align_up(off_gs + plan.groups_s.size() * sizeof(ScanGroup), 256);
    const size_t off_s1 = align_up(off_qs + plan.q_slots.size() * sizeof(uint32_t), 256);
    const size_t off_sq = align_up(off_s1 + plan.sel1.size() * sizeof(SelectQuery), 256);
    const size_t off_sd = align_up(off_sq + plan.selq.size() * sizeof(SelectQuery), 256);
    const size_t off_li = align_up(off_sd + plan.seedq.size() * sizeof(SelectQuery), 256);
    const size_t off_bm = align_up(off_li + plan.list_ids.size() * sizeof(uint32_t), 256);
    const size_t total = align_up(off_bm + plan.block_map.size() * sizeof(uint2), 256);

    int rc;
    if (ctx->desc_pending) {       // the previous batch's staging kernel still owns the pinned block
        const auto w0 = std::chrono::steady_clock::now();
        HIPCHK(hipEventSynchronize(ctx->desc_done));
        ctx->desc_pending = false;
        {
            const double waited = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - w0).count();
            ctx->host_us[1] += waited;
            ctx->stats.host_wait_ms += waited * 1e-3;
        }
    }
    if ((rc = ctx->h_desc.reserve(total))) return rc;
    if ((rc = ctx->d_desc.reserve(total))) return rc;
    if ((rc = ctx->d_partial.reserve(std::max<size_t>(8, (size_t) plan.n_partial * kp * sizeof(uint64_t))))) return rc;
    if ((rc = ctx->d_flags.reserve((size_t) nq * sizeof(int32_t)))) return rc;
    if ((rc = ctx->d_tau.reserve((size_t) nq * sizeof(uint64_t)))) return rc;
    char* hs = ctx->h_desc.as<char>();
    char* ds = ctx->d_desc.as<char>();
    if (h_queries) {
        float* hq = reinterpret_cast<float*>(hs + off_q);
        for (int s = 0; s < nq; ++s) {
            float* dst = hq + (size_t) s * qfloats;
            memcpy(dst, h_queries + (size_t) s * dim, (size_t) dim * sizeof(float));
            for (size_t j = (size_t) dim; j < qfloats; ++j) dst[j] = 0.0f;
        }
    }
    memcpy(hs + off_g, plan.groups.data(), plan.groups.size() * sizeof(ScanGroup));
    memcpy(hs + off_gs, plan.groups_s.data(), plan.groups_s.size() * sizeof(ScanGroup));
    memcpy(hs + off_qs, plan.q_slots.data(), plan.q_slots.size() * sizeof(uint32_t));
    memcpy(hs + off_s1, plan.sel1.data(), plan.sel1.size() * sizeof(SelectQuery));
    memcpy(hs + off_sq, plan.selq.data(), plan.selq.size() * sizeof(SelectQuery));
    memcpy(hs + off_sd, plan.seedq.data(), plan.seedq.size() * sizeof(SelectQuery));
    memcpy(hs + off_li, plan.list_ids.data(), plan.list_ids.size() * sizeof(uint32_t));
    memcpy(hs + off_bm, plan.block_map.data(), plan.block_map.size() * sizeof(uint2));
    hipEvent_t w0 = nullptr, w1 = nullptr;                  // profiling level 1: the whole search on the device
    if (ctx->profiling == 1) {
        w0 = take_event(ctx);
        w1 = take_event(ctx);
        HIPCHK(hipEventRecord(w0, ctx->stream));
    }
    {
        // ONE staging kernel instead of an SDMA copy + gather + norm + two fills: it pulls the descriptor block out of
        // the pinned host buffer, pads the queries to the row stride (from the caller's device buffer, or from the
        // staged host copy), computes |q|^2 with the arithmetic of the row norms, clears the per-query flags and seeds.
        StageParams st{};                                   // (no query planes, no K2w counters on this path)
        const char* hd = reinterpret_cast<const char*>(ctx->h_desc.dp);
        st.src16 = reinterpret_cast<const uint4*>(hd + off_g);
        st.dst16 = reinterpret_cast<uint4*>(ds + off_g);
        st.n16 = (uint32_t) ((total - off_g) / 16);
        st.q_src = d_queries ? d_queries : reinterpret_cast<const float*>(hd + off_q);
        st.q_stride = d_queries ? (uint32_t) dim : (uint32_t) qfloats;
        st.q_dst = reinterpret_cast<float*>(ds + off_q);
        st.dim = (uint32_t) dim;
        st.qfloats = (uint32_t) qfloats;
        st.nq = (uint32_t) nq;
        st.q_norm2 = reinterpret_cast<float*>(ds + off_qn);
        st.flags = ctx->d_flags.as<int32_t>();
        st.tau = ctx->d_tau.as<uint64_t>();
        HIPCHK(launch_stage(st, ctx->stream));
    }
    HIPCHK(hipEventRecord(ctx->desc_done, ctx->stream));
    ctx->desc_pending = true;

    ScanParams sp{};
    sp.rows = c->d_rows;
    sp.norm2 = c->d_norm2;
    sp.n_rows = (uint32_t) c->n;
    sp.stride4 = c->stride4;
    sp.queries = reinterpret_cast<const float*>(ds + off_q);
    sp.q_norm2 = reinterpret_cast<const float*>(ds + off_qn);
    sp.partial = ctx->d_partial.as<uint64_t>();
    sp.kp = kp;
    sp.k = kp;
    sp.cap = plan.k2 ? mfma_cap_for_k(kp) : scan_cap_for_k((int) kp, c->dim);
    sp.qmax = plan.qmax;
    sp.rw = (uint32_t) c->shape.rw;
    sp.cand = nullptr;
    sp.err = reinterpret_cast<uint32_t*>(ctx->d_flag_total) + 4;    // bounds-guard word (checked by vsr_screening_check)
    sp.ones = reinterpret_cast<const uint64_t*>(reinterpret_cast<const char*>(ctx->d_flag_total) + 32);
    sp.rank = c->d_rank;
    sp.block_map = nullptr;
    if (plan.mq || plan.k2) {
        if ((rc = ctx->d_cand.reserve(std::max<size_t>(8, (size_t) plan.n_scan_lists * cand_pitch(sp.cap) * sizeof(uint64_t))))) return rc;
        sp.cand = ctx->d_cand.as<uint64_t>();
    }
    sp.groups = reinterpret_cast<const ScanGroup*>(ds + off_g);
    sp.n_groups = (uint32_t) plan.groups.size();
    sp.q_slots = reinterpret_cast<const uint32_t*>(ds + off_qs);

    SelectParams sel{};
    sel.partial = ctx->d_partial.as<uint64_t>();
    sel.list_ids = reinterpret_cast<const uint32_t*>(ds + off_li);
    sel.kp = kp;
    // short candidate streams (<= 16k keys per query) merge faster with small workgroups
    uint32_t max_lists = 1;
    for (auto& q : plan.selq) max_lists = std::max(max_lists, q.n_lists);
    for (auto& q : plan.sel1) max_lists = std::max(max_lists, q.n_lists);
    uint32_t max_seed_lists = 1;
    for (auto& q : plan.seedq) max_seed_lists = std::max(max_seed_lists, q.n_lists);
    if (!plan.sel_wave) max_lists = std::max(max_lists, max_seed_lists);
    const int sel_threads = plan.sel_wave ? 64 : (uint64_t) max_lists * kp <= 16384 ? 256 : 1024;
    sel.cap = plan.sel_wave ? std::max<uint32_t>(1024, max_lists * kp) : select_cap(kp, sel_threads);   // wave: key capacity
    sel.metric = metric;
    sel.row_offset = (uint32_t) idc->row_offset;
    sel.block_ids = idc->d_block;
    sel.doc_ids = idc->d_doc;
    sel.orig_rows = idc->d_orig;
    sel.out_block = d_blk;
    sel.out_doc = d_doc;
    sel.out_row = d_row;
    sel.out_dist = d_dist;
    sel.out_keys = d_keys;
    sel.out_count = d_cnt;
    sel.out_flags = ctx->d_flags.as<int32_t>();
    sel.flagged_total = ctx->d_flag_total;
    sel.tau_out = nullptr;
    sel.seeded = 0;

    // ---- threshold seeding: a 1/SEED_STRIDE sample pass of the same launch, then the m-th best sampled candidate of
    // each query becomes the initial threshold of the main pass (all rows at or before it stay eligible; a query whose
    // seed turns out too tight is flagged by K5 / K5r and re-run unseeded) ----
    // m-th best of a 1/SEED_STRIDE sample: mean lambda = kp / SEED_STRIDE rows of the true top-kp fall into the
    // sample; lambda + 6 sigma + 4 makes a too-tight seed a ~1e-8 event (and a detected one: K5 / K5r flag it).
    // The m-th best of the whole sample only involves the m best of every sample list, so those lists are short.
    // Every sample workgroup visits at least one 64-row tile, so a pass cut into many workgroups is sampled more densely
    // than 1 / SEED_STRIDE: lambda uses the densest pass's fraction (a larger m only loosens the seed).
    double frac = 1.0 / SEED_STRIDE;
    for (const ScanGroup& g : plan.groups_s) {
        const double rows = (double) g.n_tiles * c->shape.rw;
        if (rows > 0) frac = std::max(frac, std::min(1.0, 64.0 * g.n_blocks / rows));
    }
    const double lambda = (double) kp * frac;
    const uint32_t seed_m = (uint32_t) std::ceil(lambda + 6.0 * std::sqrt(lambda)) + 4;
    constexpr uint32_t SEED_LIST = 64;                      // keys a sample-pass workgroup publishes per query (one tile: no selection)
    const bool seed = allow_screening && ctx->screening && ctx->seeding && (plan.k2 || plan.mq) && plan.n_blocks > 0 &&
                      seed_m <= SEED_LIST && kp >= SEED_LIST &&
                      plan.scan_rows >= ctx->seed_min_rows &&
                      plan.scan_rows / (int64_t) std::max<size_t>(1, plan.groups.size()) >= ctx->seed_min_pass_rows;
    sp.sample_stride = 1;
    sp.tau_init = nullptr;
    if (seed) {
        sp.sample_stride = SEED_STRIDE;
        sp.groups = reinterpret_cast<const ScanGroup*>(ds + off_gs);
        sp.n_groups = (uint32_t) plan.groups_s.size();
        hipEvent_t a0 = nullptr, a1 = nullptr, b0 = nullptr, b1 = nullptr;
        if (ctx->profiling == 1) {
            a0 = take_event(ctx); a1 = take_event(ctx); b0 = take_event(ctx); b1 = take_event(ctx);
            HIPCHK(hipEventRecord(a0, ctx->stream));
        }
        sp.kp = sp.k = SEED_LIST;
        if (plan.k2) HIPCHK(launch_mfma(sp, metric, plan.n_blocks_s, ctx->stream));
        else HIPCHK(launch_mq(sp, metric, plan.n_blocks_s, ctx->stream));
        sp.kp = sp.k = kp;
        if (a0) {
            HIPCHK(hipEventRecord(a1, ctx->stream));
            HIPCHK(hipEventRecord(b0, ctx->stream));
        }
        sp.groups = reinterpret_cast<const ScanGroup*>(ds + off_g);
        sp.n_groups = (uint32_t) plan.groups.size();
        SelectParams seeds = sel;
        seeds.kp = SEED_LIST;
        seeds.k = seed_m;
        if (plan.sel_wave) seeds.cap = std::max<uint32_t>(1024, max_seed_lists * SEED_LIST);   // <= 64 lists (planner)
        seeds.tau_out = ctx->d_tau.as<uint64_t>();
        seeds.queries = reinterpret_cast<const SelectQuery*>(ds + off_sd);
        HIPCHK(launch_select(seeds, (uint32_t) plan.seedq.size(), sel_threads, ctx->stream));
        if (a0) {
            HIPCHK(hipEventRecord(b1, ctx->stream));
            ctx->pending.push_back({a0, a1, 3});
            ctx->pending.push_back({b0, b1, 4});
        }
        sp.sample_stride = 1;
        sp.tau_init = ctx->d_tau.as<uint64_t>();
        sel.seeded = 1;
    }

    const int cls = plan.qi == 4 ? 1 : 0;
    if (plan.n_blocks) {
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (ctx->profiling) {
            e0 = take_event(ctx);
            e1 = take_event(ctx);
            HIPCHK(hipEventRecord(e0, ctx->stream));
        }
        if (!plan.block_map.empty() && !ctx->no_xcd_map) sp.block_map = reinterpret_cast<const uint2*>(ds + off_bm);
        const uint32_t launch_blocks = sp.block_map ? plan.n_launch : plan.n_blocks;
        if (plan.k2) HIPCHK(launch_mfma(sp, metric, launch_blocks, ctx->stream));
        else if (plan.mq) HIPCHK(launch_mq(sp, metric, launch_blocks, ctx->stream));
        else HIPCHK(launch_scan(sp, metric, c->dim, plan.qi, plan.n_blocks, ctx->stream));
        if (e0) {
            HIPCHK(hipEventRecord(e1, ctx->stream));
            ctx->pending.push_back({e0, e1, cls});
        }
        ctx->last_kernel = scan_kernel_name(plan, c, metric);
        ctx->stats.scan_bytes[cls] += plan.scan_bytes;
        ctx->stats.scan_rows[cls] += plan.scan_rows;
        ctx->stats.scan_pairs[cls] += plan.scan_pairs;
        ctx->stats.unique_rows[cls] += plan.unique_rows;
    }

    sel.k = kp;
    hipEvent_t s0 = nullptr, s1 = nullptr;
    if (ctx->profiling == 1) {
        s0 = take_event(ctx);
        s1 = take_event(ctx);
        HIPCHK(hipEventRecord(s0, ctx->stream));
    }
    if (!plan.sel1.empty()) {
        sel.queries = reinterpret_cast<const SelectQuery*>(ds + off_s1);
        HIPCHK(launch_select(sel, (uint32_t) plan.sel1.size(), sel_threads, ctx->stream));
    }
    sel.queries = reinterpret_cast<const SelectQuery*>(ds + off_sq);
    HIPCHK(launch_select(sel, (uint32_t) nq, sel_threads, ctx->stream));
    if (plan.k2) {
        RerankParams rr{};
        rr.lists = ctx->d_partial.as<uint64_t>() + (size_t) plan.rerank_base * kp;
        rr.queries = reinterpret_cast<const SelectQuery*>(ds + off_sq);
        rr.rows = idc->d_rows;
        rr.stride4 = c->stride4;
        rr.queries_f = reinterpret_cast<const float*>(ds + off_q);
        rr.kp = kp;
        rr.k = (uint32_t) k;
        rr.metric = metric;
        rr.dim = c->dim;
        rr.norm2_max = idc->d_norm2_max;
        rr.row_offset = (uint32_t) idc->row_offset;
        rr.block_ids = idc->d_block;
        rr.doc_ids = idc->d_doc;
        rr.orig_rows = idc->d_orig;
        rr.out_block = d_blk;
        rr.out_doc = d_doc;
        rr.out_row = d_row;
        rr.out_dist = d_dist;
        rr.out_keys = d_keys;
        rr.out_count = d_cnt;
        rr.err_g = (float) (c->dim + 8) * 5.9604645e-8f;    // K2's fp32 MFMA chain: (d + 8) * 2^-24
        rr.seeded = seed ? 1 : 0;
        rr.tau_init = seed ? ctx->d_tau.as<uint64_t>() : nullptr;
        rr.out_flags = ctx->d_flags.as<int32_t>();
        rr.flagged_total = ctx->d_flag_total;
        HIPCHK(launch_rerank(rr, (uint32_t) nq, ctx->stream));
    }
    if (s0) {
        HIPCHK(hipEventRecord(s1, ctx->stream));
        ctx->pending.push_back({s0, s1, 2});
    }
    if (w0) {
        HIPCHK(hipEventRecord(w1, ctx->stream));
        ctx->pending.push_back({w0, w1, 5});
    }
    ctx->stats.queries += nq;
    return VSR_OK;
}

static int check_search_args(const vsr_corpus* c, const void* queries, int nq, int dim, int k, int metric,
                             const vsr_filter* const* filters, const char* who)
{
    if (!c) return fail(VSR_ERR_INVALID, "%s: corpus is NULL", who);
    if (nq < 0 || (nq > 0 && !queries)) return fail(VSR_ERR_INVALID, "%s: queries is NULL", who);
    if (dim != c->dim) return fail(VSR_ERR_DIM_MISMATCH, "different vector dimensions %d and %d", c->dim, dim);
    if (k < 1) return fail(VSR_ERR_INVALID, "%s: k must be >= 1 (got %d)", who, k);
    if (k > VSR_MAX_K) return fail(VSR_ERR_UNSUPPORTED, "%s: k = %d exceeds VSR_MAX_K = %d", who, k, VSR_MAX_K);
    if (metric < VSR_METRIC_L2 || metric > VSR_METRIC_L1) return fail(VSR_ERR_INVALID, "%s: metric %d", who, metric);
    if (filters)
        for (int i = 0; i < nq; ++i)
            if (filters[i] && filters[i]->corpus != c) return fail(VSR_ERR_INVALID, "%s: filter %d belongs to another corpus", who, i);
    return VSR_OK;
}

extern "C" int vsr_search_device_on(vsr_ctx* session, vsr_corpus* c, const float* d_queries, int nq, int dim, int k,
                                    int metric, const vsr_filter* const* filters, int64_t* d_blk, int32_t* d_doc,
                                    int64_t* d_row, float* d_dist, int32_t* d_cnt, uint64_t* d_keys)
{
    int rc = check_search_args(c, d_queries, nq, dim, k, metric, filters, "vsr_search_device");
    if (rc) return rc;
    vsr_ctx* ctx = session ? session : c->ctx;
    if (ctx->device != c->ctx->device) return fail(VSR_ERR_INVALID, "vsr_search_device_on: session and corpus are on different devices");
    if (nq == 0) return VSR_OK;
    if (!d_blk || !d_dist || !d_cnt) return fail(VSR_ERR_INVALID, "vsr_search_device: output is NULL");
    HIPCHK(hipSetDevice(ctx->device));
    if (!d_doc) {
        if ((rc = ctx->d_misc.reserve((size_t) nq * k * sizeof(int32_t)))) return rc;
        d_doc = ctx->d_misc.as<int32_t>();
    }
    return search_impl(ctx, c, nullptr, d_queries, nq, dim, k, metric, filters, d_blk, d_doc, d_row, d_dist, d_cnt, d_keys, 2);
}

extern "C" int vsr_search_device(vsr_corpus* c, const float* d_queries, int nq, int dim, int k, int metric,
                                 const vsr_filter* const* filters, int64_t* d_blk, int32_t* d_doc, int64_t* d_row,
                                 float* d_dist, int32_t* d_cnt, uint64_t* d_keys)
{
    return vsr_search_device_on(nullptr, c, d_queries, nq, dim, k, metric, filters, d_blk, d_doc, d_row, d_dist, d_cnt,
                                d_keys);
}

// The device API for callers that cannot tolerate an unproven row: search, wait, re-run what was flagged, patch.
extern "C" int vsr_search_device_exact(vsr_ctx* session, vsr_corpus* c, const float* d_queries, int nq, int dim, int k,
                                       int metric, const vsr_filter* const* filters, int64_t* d_blk, int32_t* d_doc,
                                       int64_t* d_row, float* d_dist, int32_t* d_cnt, uint64_t* d_keys, int32_t* n_rerun)
{
    if (n_rerun) *n_rerun = 0;
    int rc = vsr_search_device_on(session, c, d_queries, nq, dim, k, metric, filters, d_blk, d_doc, d_row, d_dist, d_cnt, d_keys);
    if (rc || nq == 0) return rc;
    vsr_ctx* ctx = session ? session : c->ctx;
    std::vector<int32_t> flags((size_t) nq, 0);
    HIPCHK(hipMemcpyAsync(flags.data(), ctx->d_flags.p, (size_t) nq * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    std::vector<int> redo;
    for (int i = 0; i < nq; ++i)
        if (flags[(size_t) i]) redo.push_back(i);
    if (redo.empty()) return VSR_OK;
    ctx->flagged_seen += (int64_t) redo.size();
    if (n_rerun) *n_rerun = (int32_t) redo.size();
    // flagged on the coarse planes: the fine planes next; flagged there (or no coarse tier involved): the exact kernels
    for (int level = ctx->last_coarse ? 1 : 0; level >= 0 && !redo.empty(); --level) {
        const size_t nr = redo.size(), nk = nr * (size_t) k;
        // workspace: [queries | block | row | doc | dist | keys | counts] of the flagged queries
        const size_t o_q = 0, o_blk = align_up(o_q + nr * (size_t) dim * 4, 256), o_row = align_up(o_blk + nk * 8, 256),
                     o_doc = align_up(o_row + nk * 8, 256), o_dist = align_up(o_doc + nk * 4, 256),
                     o_key = align_up(o_dist + nk * 4, 256), o_cnt = align_up(o_key + nk * 8, 256),
                     total = align_up(o_cnt + nr * 4, 256);
        if ((rc = ctx->d_redo.reserve(total))) return rc;
        char* w = ctx->d_redo.as<char>();
        std::vector<const vsr_filter*> f2(nr, nullptr);
        for (size_t j = 0; j < nr; ++j) {
            HIPCHK(hipMemcpyAsync(w + o_q + j * (size_t) dim * 4, d_queries + (size_t) redo[j] * dim, (size_t) dim * 4,
                                  hipMemcpyDeviceToDevice, ctx->stream));
            if (filters) f2[j] = filters[redo[j]];
        }
        rc = search_impl(ctx, c, nullptr, reinterpret_cast<const float*>(w + o_q), (int) nr, dim, k, metric, f2.data(),
                         reinterpret_cast<int64_t*>(w + o_blk), reinterpret_cast<int32_t*>(w + o_doc),
                         reinterpret_cast<int64_t*>(w + o_row), reinterpret_cast<float*>(w + o_dist),
                         reinterpret_cast<int32_t*>(w + o_cnt), reinterpret_cast<uint64_t*>(w + o_key), level);
        if (rc) return rc;
        for (size_t j = 0; j < nr; ++j) {
            const size_t src = j * (size_t) k, dst = (size_t) redo[j] * k;
            auto patch = [&](void* to, const void* from, size_t bytes) {
                return hipMemcpyAsync(to, from, bytes, hipMemcpyDeviceToDevice, ctx->stream);
            };
            HIPCHK(patch(d_blk + dst, reinterpret_cast<int64_t*>(w + o_blk) + src, (size_t) k * 8));
            if (d_row) HIPCHK(patch(d_row + dst, reinterpret_cast<int64_t*>(w + o_row) + src, (size_t) k * 8));
            if (d_doc) HIPCHK(patch(d_doc + dst, reinterpret_cast<int32_t*>(w + o_doc) + src, (size_t) k * 4));
            HIPCHK(patch(d_dist + dst, reinterpret_cast<float*>(w + o_dist) + src, (size_t) k * 4));
            if (d_keys) HIPCHK(patch(d_keys + dst, reinterpret_cast<uint64_t*>(w + o_key) + src, (size_t) k * 8));
            HIPCHK(patch(d_cnt + redo[j], reinterpret_cast<int32_t*>(w + o_cnt) + j, 4));
        }
        HIPCHK(hipStreamSynchronize(ctx->stream));
        HIPCHK(hipMemcpy(flags.data(), ctx->d_flags.p, nr * sizeof(int32_t), hipMemcpyDeviceToHost));
        std::vector<int> still;
        for (size_t j = 0; j < nr; ++j)
            if (flags[j]) still.push_back(redo[j]);
        // the exact path never flags; anything else is a library fault and must not be published
        if (level == 0 && !still.empty())
            return fail(VSR_ERR_HIP, "vsr_search_device_exact: query %d still flagged after the exact re-run", still[0]);
        redo.swap(still);
    }
    return VSR_OK;
}

// Host-buffer search on a corpus or on a list-ordered view of one: the search, then the exact re-run of flagged queries.
static int host_search(vsr_corpus* c, const float* queries, int nq, int dim, int k, int metric,
                       const vsr_filter* const* filters, int64_t* out_blk, int32_t* out_doc, int64_t* out_row,
                       float* out_dist, int32_t* out_cnt)
{
    int rc;
    vsr_ctx* ctx = c->ctx;
    HIPCHK(hipSetDevice(ctx->device));
    const size_t nk = (size_t) nq * k;
    const size_t o_blk = 0, o_row = align_up(o_blk + nk * 8, 256), o_doc = align_up(o_row + nk * 8, 256),
                 o_dist = align_up(o_doc + nk * 4, 256), o_cnt = align_up(o_dist + nk * 4, 256),
                 total = align_up(o_cnt + (size_t) nq * 4, 256);
    // [results | flags] in pinned memory.  Small results (the harness's one query per call) are written there by the kernels
    // themselves -- the block is mapped into the device's address space -- so a call is its launches, a 4-byte-per-query copy of
    // the flags and ONE wait; larger ones go through device memory and one packed copy.
    const size_t o_flags = total, total_h = align_up(o_flags + (size_t) nq * 4, 256);
    const bool direct = total <= 64 * 1024;
    if (!direct && (rc = ctx->d_out.reserve(total))) return rc;
    if ((rc = ctx->h_out.reserve(total_h))) return rc;
    char* d = direct ? static_cast<char*>(ctx->h_out.dp) : ctx->d_out.as<char>();
    char* h = ctx->h_out.as<char>();
    auto run = [&](const float* qs, int n, const vsr_filter* const* fs, int level) -> int {
        int r = search_impl(ctx, c, qs, nullptr, n, dim, k, metric, fs, reinterpret_cast<int64_t*>(d + o_blk),
                            reinterpret_cast<int32_t*>(d + o_doc), reinterpret_cast<int64_t*>(d + o_row),
                            reinterpret_cast<float*>(d + o_dist), reinterpret_cast<int32_t*>(d + o_cnt), nullptr, level);
        if (r) return r;
        if (!direct) HIPCHK(hipMemcpyAsync(h, d, total, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipMemcpyAsync(h + o_flags, ctx->d_flags.p, (size_t) n * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        return VSR_OK;
    };
    if ((rc = run(queries, nq, filters, 2))) return rc;
    memcpy(out_blk, h + o_blk, nk * 8);
    if (out_row) memcpy(out_row, h + o_row, nk * 8);
    if (out_doc) memcpy(out_doc, h + o_doc, nk * 4);
    memcpy(out_dist, h + o_dist, nk * 4);
    memcpy(out_cnt, h + o_cnt, (size_t) nq * 4);

    // screening flags: re-run the (rare) flagged queries one tier down -- coarse planes -> fine planes -> exact kernels
    std::vector<int32_t> flags(reinterpret_cast<const int32_t*>(h + o_flags), reinterpret_cast<const int32_t*>(h + o_flags) + nq);
    std::vector<int> redo;
    for (int i = 0; i < nq; ++i)
        if (flags[(size_t) i]) redo.push_back(i);
    if (!redo.empty()) ctx->flagged_seen += (int64_t) redo.size();
    for (int level = ctx->last_coarse ? 1 : 0; level >= 0 && !redo.empty(); --level) {
        std::vector<float> q2(redo.size() * (size_t) dim);
        std::vector<const vsr_filter*> f2(redo.size(), nullptr);
        for (size_t j = 0; j < redo.size(); ++j) {
            memcpy(&q2[j * (size_t) dim], queries + (size_t) redo[j] * dim, (size_t) dim * sizeof(float));
            if (filters) f2[j] = filters[redo[j]];
        }
        if ((rc = run(q2.data(), (int) redo.size(), f2.data(), level))) return rc;
        memcpy(flags.data(), h + o_flags, redo.size() * sizeof(int32_t));
        std::vector<int> still;
        for (size_t j = 0; j < redo.size(); ++j) {
            if (flags[j] && level > 0) {                      // unproven again: one more tier down
                still.push_back(redo[j]);
                continue;
            }
            const size_t src = j * (size_t) k, dst = (size_t) redo[j] * k;
            memcpy(out_blk + dst, reinterpret_cast<int64_t*>(h + o_blk) + src, (size_t) k * 8);
            if (out_row) memcpy(out_row + dst, reinterpret_cast<int64_t*>(h + o_row) + src, (size_t) k * 8);
            if (out_doc) memcpy(out_doc + dst, reinterpret_cast<int32_t*>(h + o_doc) + src, (size_t) k * 4);
            memcpy(out_dist + dst, reinterpret_cast<float*>(h + o_dist) + src, (size_t) k * 4);
            out_cnt[redo[j]] = reinterpret_cast<int32_t*>(h + o_cnt)[j];
        }
        redo.swap(still);
    }
    return VSR_OK;
}

extern "C" int vsr_search(vsr_corpus* c, const float* queries, int nq, int dim, int k, int metric,
                          const vsr_filter* const* filters, int64_t* out_blk, int32_t* out_doc, int64_t* out_row,
                          float* out_dist, int32_t* out_cnt)
{
    int rc = check_search_args(c, queries, nq, dim, k, metric, filters, "vsr_search");
    if (rc) return rc;
    if (c->base) return fail(VSR_ERR_INVALID, "vsr_search: this corpus is an index view; use the index's search function");
    if (nq == 0) return VSR_OK;
    if (!out_blk || !out_dist || !out_cnt) return fail(VSR_ERR_INVALID, "vsr_search: output is NULL");
    return host_search(c, queries, nq, dim, k, metric, filters, out_blk, out_doc, out_row, out_dist, out_cnt);
}

extern "C" int vsr_last_scan_kernel(vsr_ctx* ctx, char* name, int name_len)
{
    if (!ctx || !name || name_len < 1) return fail(VSR_ERR_INVALID, "vsr_last_scan_kernel: bad argument");
    snprintf(name, (size_t) name_len, "%s", ctx->last_kernel.c_str());
    return VSR_OK;
}

extern "C" int vsr_set_query_hint(vsr_ctx* ctx, int u8_queries)
{
    if (!ctx) return fail(VSR_ERR_INVALID, "vsr_set_query_hint: ctx is NULL");
    ctx->hint_u8 = u8_queries != 0;
    if (ctx->hint_u8) {                                     // a fresh promise: forget earlier violations
        ctx->q8_ok = true;
        if (ctx->h_q8.p) *reinterpret_cast<volatile uint32_t*>(ctx->h_q8.p) = 0;
    }
    return VSR_OK;
}

extern "C" int vsr_set_screening(vsr_ctx* ctx, int enable)
{
    if (!ctx) return fail(VSR_ERR_INVALID, "vsr_set_screening: ctx is NULL");
    ctx->screening = enable != 0;
    return VSR_OK;
}

extern "C" int vsr_screening_check(vsr_ctx* ctx, int64_t* flagged_total, int32_t* flags_last_call, int nq)
{
    if (!ctx) return fail(VSR_ERR_INVALID, "vsr_screening_check: ctx is NULL");
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    int32_t words[8] = {0};
    HIPCHK(hipMemcpy(words, ctx->d_flag_total, sizeof words, hipMemcpyDeviceToHost));
    const int32_t total = words[0];
    if (words[4] != 0)
        return fail(VSR_ERR_HIP, "internal bounds guard tripped in a scan kernel (bits %d): results are not valid", words[4]);
    if (flagged_total) *flagged_total = total;
    if (flags_last_call && nq > 0) {
        if (!ctx->d_flags.p || ctx->d_flags.cap < (size_t) nq * sizeof(int32_t))
            memset(flags_last_call, 0, (size_t) nq * sizeof(int32_t));
        else
            HIPCHK(hipMemcpy(flags_last_call, ctx->d_flags.p, (size_t) nq * sizeof(int32_t), hipMemcpyDeviceToHost));
    }
    return VSR_OK;
}

extern "C" int vsr_merge_topk_device(vsr_ctx* ctx, const uint64_t* d_keys, const int64_t* d_blk, const int32_t* d_doc,
                                     const float* d_dist, int n_parts, int nq, int k, int64_t* o_blk, int32_t* o_doc,
                                     float* o_dist, uint64_t* o_keys, int32_t* o_cnt)
{
    if (!ctx || !d_keys || !d_blk || !d_doc || !d_dist || !o_blk || !o_doc || !o_dist || !o_cnt)
        return fail(VSR_ERR_INVALID, "vsr_merge_topk_device: NULL argument");
    if (n_parts < 1 || nq < 0 || k < 1) return fail(VSR_ERR_INVALID, "vsr_merge_topk_device: bad sizes");
    if (nq == 0) return VSR_OK;
    if ((int64_t) n_parts * k > 8192) return fail(VSR_ERR_UNSUPPORTED, "vsr_merge_topk_device: n_parts * k > 8192");
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(launch_merge_lists(d_keys, d_blk, d_doc, d_dist, (uint32_t) n_parts, (uint32_t) nq, (uint32_t) k, 0, o_blk,
                              o_doc, o_dist, o_keys, o_cnt, ctx->stream));
    return VSR_OK;
}

extern "C" int64_t vsr_packed_result_bytes(int nq, int k)
{
    return nq < 0 || k < 1 ? 0 : (int64_t) nq * k * 24;
}

extern "C" int vsr_merge_topk_packed_device(vsr_ctx* ctx, const void* d_packed, int n_parts, int nq, int k,
                                            int64_t* o_blk, int32_t* o_doc, float* o_dist, uint64_t* o_keys, int32_t* o_cnt)
{
    if (!ctx || !d_packed || !o_blk || !o_doc || !o_dist || !o_cnt) return fail(VSR_ERR_INVALID, "vsr_merge_topk_packed_device: NULL argument");
    if (n_parts < 1 || nq < 0 || k < 1) return fail(VSR_ERR_INVALID, "vsr_merge_topk_packed_device: bad sizes");
    if (nq == 0) return VSR_OK;
    if ((int64_t) n_parts * k > 8192) return fail(VSR_ERR_UNSUPPORTED, "vsr_merge_topk_packed_device: n_parts * k > 8192");
    HIPCHK(hipSetDevice(ctx->device));
    const size_t nk = (size_t) nq * k;
    const char* base = reinterpret_cast<const char*>(d_packed);
    HIPCHK(launch_merge_lists(reinterpret_cast<const uint64_t*>(base), reinterpret_cast<const int64_t*>(base + nk * 8),
                              reinterpret_cast<const int32_t*>(base + nk * 16), reinterpret_cast<const float*>(base + nk * 20),
                              (uint32_t) n_parts, (uint32_t) nq, (uint32_t) k, nk * 24, o_blk, o_doc, o_dist, o_keys, o_cnt,
                              ctx->stream));
    return VSR_OK;
}

extern "C" int vsr_pair_distances(vsr_ctx* ctx, int metric, const float* a, const float* b, int64_t n_pairs, int dim_a,
                                  int dim_b, int b_broadcast, double* out)
{
    if (!ctx || n_pairs < 0 || (n_pairs > 0 && (!a || !b || !out))) return fail(VSR_ERR_INVALID, "vsr_pair_distances: NULL argument");
    if (dim_a != dim_b) return fail(VSR_ERR_DIM_MISMATCH, "different vector dimensions %d and %d", dim_a, dim_b);
    if (dim_a < 1) return fail(VSR_ERR_INVALID, "vsr_pair_distances: dim %d", dim_a);
    if (metric < VSR_METRIC_L2 || metric > VSR_METRIC_L1) return fail(VSR_ERR_INVALID, "vsr_pair_distances: metric %d", metric);
    if (n_pairs == 0) return VSR_OK;
    HIPCHK(hipSetDevice(ctx->device));
    const size_t a_bytes = (size_t) n_pairs * dim_a * sizeof(float);
    const size_t b_bytes = (size_t) (b_broadcast ? 1 : n_pairs) * dim_a * sizeof(float);
    const size_t o_a = 0, o_b = align_up(a_bytes, 256), o_out = align_up(o_b + b_bytes, 256);
    int rc = ctx->d_misc.reserve(o_out + (size_t) n_pairs * sizeof(double));
    if (rc) return rc;
    char* d = ctx->d_misc.as<char>();
    HIPCHK(hipMemcpyAsync(d + o_a, a, a_bytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(d + o_b, b, b_bytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(launch_pair_distances(reinterpret_cast<float*>(d + o_a), reinterpret_cast<float*>(d + o_b), n_pairs, dim_a,
                                 b_broadcast, metric, reinterpret_cast<double*>(d + o_out), ctx->stream));
    HIPCHK(hipMemcpyAsync(out, d + o_out, (size_t) n_pairs * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return VSR_OK;
}

// ---------------------------------------------------------------------------------------------
// opclass support functions, batched (vector.c:692-711, 756-808)
// ---------------------------------------------------------------------------------------------
static int vector_fn(vsr_ctx* ctx, int mode, const float* a, const float* b, int64_t n, int dim_a, int dim_b, int b_broadcast,
                     double* out_d, float* out_f, const char* who)
{
    if (!ctx || n < 0 || (n > 0 && (!a || (mode == 2 && !b) || (mode == 1 ? !out_f : !out_d))))
        return fail(VSR_ERR_INVALID, "%s: NULL argument", who);
    if (mode == 2 && dim_a != dim_b) return fail(VSR_ERR_DIM_MISMATCH, "different vector dimensions %d and %d", dim_a, dim_b);
    if (dim_a < 1) return fail(VSR_ERR_INVALID, "%s: dim %d", who, dim_a);
    if (n == 0) return VSR_OK;
    HIPCHK(hipSetDevice(ctx->device));
    const size_t a_bytes = (size_t) n * dim_a * sizeof(float);
    const size_t b_bytes = mode == 2 ? (size_t) (b_broadcast ? 1 : n) * dim_a * sizeof(float) : 0;
    const size_t o_bytes = mode == 1 ? a_bytes : (size_t) n * sizeof(double);
    const size_t o_b = align_up(a_bytes, 256), o_out = align_up(o_b + b_bytes, 256), o_flag = align_up(o_out + o_bytes, 256);
    int rc = ctx->d_misc.reserve(o_flag + 64);
    if (rc) return rc;
    char* d = ctx->d_misc.as<char>();
    HIPCHK(hipMemcpyAsync(d, a, a_bytes, hipMemcpyHostToDevice, ctx->stream));
    if (b_bytes) HIPCHK(hipMemcpyAsync(d + o_b, b, b_bytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemsetAsync(d + o_flag, 0, 4, ctx->stream));
    HIPCHK(launch_vector_fn(mode, reinterpret_cast<float*>(d), reinterpret_cast<float*>(d + o_b), n, dim_a, b_broadcast,
                            reinterpret_cast<double*>(d + o_out), reinterpret_cast<float*>(d + o_out),
                            reinterpret_cast<int*>(d + o_flag), ctx->stream));
    int overflow = 0;
    HIPCHK(hipMemcpyAsync(mode == 1 ? (void*) out_f : (void*) out_d, d + o_out, o_bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(&overflow, d + o_flag, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (overflow) return fail(VSR_ERR_INVALID, "value out of range: overflow");        // float_overflow_error(), vector.c:803
    return VSR_OK;
}

extern "C" int vsr_vector_norms(vsr_ctx* ctx, const float* a, int64_t n, int dim, double* out)
{
    return vector_fn(ctx, 0, a, nullptr, n, dim, dim, 0, out, nullptr, "vsr_vector_norms");
}

extern "C" int vsr_l2_normalize(vsr_ctx* ctx, const float* a, int64_t n, int dim, float* out)
{
    return vector_fn(ctx, 1, a, nullptr, n, dim, dim, 0, nullptr, out, "vsr_l2_normalize");
}

extern "C" int vsr_spherical_distances(vsr_ctx* ctx, const float* a, const float* b, int64_t n, int dim_a, int dim_b,
                                       int b_broadcast, double* out)
{
    return vector_fn(ctx, 2, a, b, n, dim_a, dim_b, b_broadcast, out, nullptr, "vsr_spherical_distances");
}

// ---------------------------------------------------------------------------------------------
// K3: IVFFlat list probe (ivfscan.c:36-176, 339-389) over a list-ordered view of the corpus
// ---------------------------------------------------------------------------------------------
// [lists][dim] -> [dim][lists]: the layout ivf_probe_kernel reads (vsr_kernels.hip)
static std::vector<float> transpose_centers(const float* centers, int lists, int dim)
{
    std::vector<float> t((size_t) lists * dim);
    for (int c = 0; c < lists; ++c)
        for (int j = 0; j < dim; ++j) t[(size_t) j * lists + c] = centers[(size_t) c * dim + j];
    return t;
}

struct vsr_ivf {
    vsr_corpus* main = nullptr;
    vsr_corpus* view = nullptr;                      // list-ordered rows; view->base = main
    int         lists = 0;
    float*      d_centers = nullptr;
    std::vector<uint32_t> list_start;                // lists + 1 offsets into the view
    std::vector<vsr_filter*> list_filters;           // one RANGES filter per list (tiles over the view)
    struct ViewBitmap { uint64_t* d = nullptr; std::vector<uint64_t> h; };
    std::map<uint64_t, ViewBitmap> view_bitmaps;                          // a base filter (by vsr_filter::id) as a bitmap in view order
    std::map<std::pair<uint64_t, int>, vsr_filter*> parts;                // (base filter id, list) -> part of a probe
    DevBuf d_q, d_probe;
};

extern "C" int vsr_ivf_free(vsr_ivf* ivf)
{
    if (!ivf) return VSR_OK;
    if (ivf->main) {
        (void) hipSetDevice(ivf->main->ctx->device);
        (void) hipStreamSynchronize(ivf->main->ctx->stream);
    }
    if (ivf->main) {
        auto& reg = ivf->main->ivf_indexes;
        reg.erase(std::remove(reg.begin(), reg.end(), ivf), reg.end());
    }
    for (auto& kv : ivf->parts) delete kv.second;    // tiles / bitmaps are borrowed
    for (auto& kv : ivf->view_bitmaps)
        if (kv.second.d) (void) hipFree(kv.second.d);
    for (vsr_filter* f : ivf->list_filters) free_filter(f);
    if (ivf->d_centers) (void) hipFree(ivf->d_centers);
    ivf->d_q.release();
    ivf->d_probe.release();
    delete ivf->view;                                // frees the view's own arrays only
    delete ivf;
    return VSR_OK;
}

extern "C" int vsr_ivf_load(vsr_corpus* c, const float* centers, int lists, const int32_t* row_list, vsr_ivf** out)
{
    if (!c || !out || !centers || (c->n > 0 && !row_list)) return fail(VSR_ERR_INVALID, "vsr_ivf_load: NULL argument");
    *out = nullptr;
    if (c->base) return fail(VSR_ERR_INVALID, "vsr_ivf_load: the corpus is itself a view");
    if (lists < 1 || lists > 32768)      /* reloption lists: 1 .. IVFFLAT_MAX_LISTS (ivfflat.h:42-44) */
        return fail(VSR_ERR_UNSUPPORTED, "vsr_ivf_load: lists must be between 1 and 32768 (got %d)", lists);
    vsr_ctx* ctx = c->ctx;
    HIPCHK(hipSetDevice(ctx->device));
    const int64_t n = c->n;
    for (int64_t i = 0; i < n; ++i)
        if (row_list[i] < 0 || row_list[i] >= lists) return fail(VSR_ERR_INVALID, "vsr_ivf_load: row %lld has list %d", (long long) i, row_list[i]);
    std::unique_ptr<vsr_ivf> ivf(new vsr_ivf());
    ivf->main = c;
    ivf->lists = lists;
    // view order: by list, then by base row (= (document_id, block_id) order inside a list)
    std::vector<uint32_t> count((size_t) lists + 1, 0);
    for (int64_t r = 0; r < n; ++r) count[(size_t) row_list[c->h_orig[(size_t) r]] + 1]++;
    for (int l = 0; l < lists; ++l) count[(size_t) l + 1] += count[(size_t) l];
    ivf->list_start = count;
    std::vector<uint32_t> rank((size_t) std::max<int64_t>(n, 1));
    {
        std::vector<uint32_t> cur(count.begin(), count.end() - 1);
        for (int64_t r = 0; r < n; ++r) rank[cur[(size_t) row_list[c->h_orig[(size_t) r]]]++] = (uint32_t) r;
    }
    std::unique_ptr<vsr_corpus> v(new vsr_corpus());
    v->ctx = ctx;
    v->n = n;
    v->dim = c->dim;
    v->stride4 = c->stride4;
    v->row_offset = c->row_offset;
    v->shape = c->shape;
    v->base = c;
    v->k2_safe = c->k2_safe;
    const size_t alloc_rows = (size_t) std::max<int64_t>(n, 1), row_bytes = (size_t) c->stride4 * 16;
    HIPCHK(hipMalloc(&v->d_rank, alloc_rows * sizeof(uint32_t)));
    HIPCHK(hipMemcpy(v->d_rank, rank.data(), alloc_rows * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIPCHK(hipMalloc(&v->d_rows, alloc_rows * row_bytes + 1024));
    HIPCHK(hipMalloc(&v->d_norm2, alloc_rows * sizeof(float)));
    HIPCHK(launch_gather_rows(c->d_rows, c->d_norm2, v->d_rank, (uint32_t) n, c->stride4, v->d_rows, v->d_norm2, ctx->stream));
    if (c->d_scr) {                                  // the view's own screening planes, in its order
        v->scr_has_mid = c->scr_has_mid;
        v->pstride4 = c->pstride4;
        HIPCHK(hipMalloc(&v->d_scr, alloc_rows * (size_t) v->pstride4 * 16 + 1024));
        HIPCHK(launch_split_planes(v->d_rows, (uint32_t) n, v->stride4, v->d_scr, v->pstride4, !v->scr_has_mid, ctx->stream));
        std::vector<uint2> all;
        ranges_to_tiles({{0u, (uint32_t) n}}, v->shape.rw, all);
        HIPCHK(hipMalloc(&v->d_all_tiles, std::max<size_t>(8, all.size() * sizeof(uint2))));
        if (!all.empty()) HIPCHK(hipMemcpy(v->d_all_tiles, all.data(), all.size() * sizeof(uint2), hipMemcpyHostToDevice));
    }
    HIPCHK(hipMalloc(&ivf->d_centers, (size_t) lists * c->dim * sizeof(float)));
    {                                                       // transposed for the probe kernel: element j of every list contiguous
        std::vector<float> ct = transpose_centers(centers, lists, c->dim);
        HIPCHK(hipMemcpy(ivf->d_centers, ct.data(), ct.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ivf->view = v.release();
    ivf->list_filters.assign((size_t) lists, nullptr);
    for (int l = 0; l < lists; ++l) {
        std::unique_ptr<vsr_filter, void (*)(vsr_filter*)> f(new vsr_filter(), free_filter);
        f->corpus = ivf->view;
        f->mode = VSR_FILTER_RANGES;
        f->cached = true;
        std::vector<uint2> tiles;
        const uint32_t s0 = ivf->list_start[(size_t) l], s1 = ivf->list_start[(size_t) l + 1];
        if (s1 > s0) ranges_to_tiles({{s0, s1}}, ivf->view->shape.rw, tiles);
        int rc = upload_tiles(f.get(), tiles);
        if (rc) { vsr_ivf_free(ivf.release()); return rc; }
        f->allowed_rows = f->scanned_rows = s1 - s0;
        ivf->list_filters[(size_t) l] = f.release();
    }
    c->ivf_indexes.push_back(ivf.get());
    *out = ivf.release();
    return VSR_OK;
}

// (base filter, list) as a filter of the view: the list's tiles and the base filter's bitmap in view order
static int ivf_part(vsr_ivf* ivf, const vsr_filter* bf, int list, vsr_filter** out)
{
    if (!bf) {
        *out = ivf->list_filters[(size_t) list];
        return VSR_OK;
    }
    auto key = std::make_pair(bf->id, list);
    auto it = ivf->parts.find(key);
    if (it != ivf->parts.end()) {
        *out = it->second;
        return VSR_OK;
    }
    vsr_corpus* v = ivf->view;
    vsr_ctx* ctx = v->ctx;
    auto& vb = ivf->view_bitmaps[bf->id];
    if (!vb.d) {
        const size_t words = bitmap_words(v->n);
        HIPCHK(hipMalloc(&vb.d, words * sizeof(uint64_t)));
        HIPCHK(hipMemsetAsync(vb.d, 0, words * sizeof(uint64_t), ctx->stream));
        HIPCHK(launch_view_bitmap(v->d_rank, (uint32_t) v->n, bf->d_tiles, bf->n_tiles, bf->d_bitmap, vb.d, ctx->stream));
        vb.h.resize(words);
        HIPCHK(hipMemcpyAsync(vb.h.data(), vb.d, words * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    const vsr_filter* lf = ivf->list_filters[(size_t) list];
    std::unique_ptr<vsr_filter> f(new vsr_filter());
    f->corpus = v;
    f->mode = VSR_FILTER_BITMAP;
    f->cached = true;
    f->d_tiles = lf->d_tiles;
    f->n_tiles = lf->n_tiles;
    f->d_bitmap = vb.d;
    f->owns_bitmap = false;
    f->scanned_rows = lf->scanned_rows;
    int64_t allowed = 0;
    for (uint32_t p = ivf->list_start[(size_t) list]; p < ivf->list_start[(size_t) list + 1]; ++p)
        allowed += (vb.h[p >> 6] >> (p & 63)) & 1ull;
    f->allowed_rows = allowed;
    *out = ivf->parts[key] = f.release();
    return VSR_OK;
}

// GetScanLists + the per-query filters of GetScanItems: the probe launch on device-resident queries, the probed list
// ids back to the host (nq x probes x 4 bytes: the planner that groups queries by list is host code), one parts-only
// filter of the view per query.
static int ivf_plan(vsr_ivf* ivf, const float* d_queries, int nq, int dim, int probes, int metric,
                    const vsr_filter* const* filters, std::vector<std::unique_ptr<vsr_filter>>& owned,
                    std::vector<const vsr_filter*>& fl)
{
    vsr_ctx* ctx = ivf->main->ctx;
    int rc;
    if ((rc = ivf->d_probe.reserve((size_t) nq * probes * sizeof(int32_t)))) return rc;
    // cosine opclass: the caller passes normalised queries and the index distance is the negative inner product
    // (vector.sql:323-327)
    HIPCHK(launch_ivf_probe(d_queries, (uint32_t) dim, (uint32_t) nq, ivf->d_centers, dim, ivf->lists, probes,
                            metric == VSR_METRIC_L2 ? M_L2 : M_IP, ivf->d_probe.as<int32_t>(), ctx->stream));
    std::vector<int32_t> probe((size_t) nq * probes);
    HIPCHK(hipMemcpyAsync(probe.data(), ivf->d_probe.p, probe.size() * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    owned.resize((size_t) nq);
    fl.assign((size_t) nq, nullptr);
    for (int q = 0; q < nq; ++q) {
        std::unique_ptr<vsr_filter> f(new vsr_filter());
        f->corpus = ivf->view;
        f->mode = VSR_FILTER_RANGES;
        f->parts_only = true;
        for (int j = 0; j < probes; ++j) {
            const int32_t l = probe[(size_t) q * probes + j];
            if (l < 0) continue;
            vsr_filter* part = nullptr;
            if ((rc = ivf_part(ivf, filters ? filters[q] : nullptr, l, &part))) return rc;
            if (part->n_tiles == 0) continue;
            f->parts.push_back(part);
            f->allowed_rows += part->allowed_rows;
            f->scanned_rows += part->scanned_rows;
        }
        fl[(size_t) q] = f.get();
        owned[(size_t) q] = std::move(f);
    }
    return VSR_OK;
}

static int ivf_check(vsr_ivf* ivf, const float* queries, int nq, int dim, int k, int& probes, int metric,
                     const vsr_filter* const* filters, const void* o1, const void* o2, const void* o3, const char* who)
{
    if (!ivf) return fail(VSR_ERR_INVALID, "%s: index is NULL", who);
    int rc = check_search_args(ivf->main, queries, nq, dim, k, metric, filters, who);
    if (rc) return rc;
    if (metric == VSR_METRIC_L1) return fail(VSR_ERR_UNSUPPORTED, "%s: ivfflat has no L1 operator class", who);
    if (probes < 1) return fail(VSR_ERR_INVALID, "%s: probes must be >= 1 (got %d)", who, probes);   /* ivfflat.c:41-45 */
    if (nq > 0 && (!o1 || !o2 || !o3)) return fail(VSR_ERR_INVALID, "%s: output is NULL", who);
    probes = std::min(probes, ivf->lists);
    return VSR_OK;
}

extern "C" int vsr_ivf_search(vsr_ivf* ivf, const float* queries, int nq, int dim, int k, int probes, int metric,
                              const vsr_filter* const* filters, int64_t* out_blk, int32_t* out_doc, int64_t* out_row,
                              float* out_dist, int32_t* out_cnt)
{
    int rc = ivf_check(ivf, queries, nq, dim, k, probes, metric, filters, out_blk, out_dist, out_cnt, "vsr_ivf_search");
    if (rc || nq == 0) return rc;
    vsr_ctx* ctx = ivf->main->ctx;
    HIPCHK(hipSetDevice(ctx->device));
    if ((rc = ivf->d_q.reserve((size_t) nq * dim * sizeof(float)))) return rc;
    HIPCHK(hipMemcpyAsync(ivf->d_q.p, queries, (size_t) nq * dim * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    std::vector<std::unique_ptr<vsr_filter>> owned;
    std::vector<const vsr_filter*> fl;
    if ((rc = ivf_plan(ivf, ivf->d_q.as<float>(), nq, dim, probes, metric, filters, owned, fl))) return rc;
    return host_search(ivf->view, queries, nq, dim, k, metric, fl.data(), out_blk, out_doc, out_row, out_dist, out_cnt);
}

// The same with queries and results resident on the device.  Returns when every query is proven exact over its lists
// (vsr_search_device_exact's contract); only the probed list ids cross PCIe.
extern "C" int vsr_ivf_search_device(vsr_ivf* ivf, const float* d_queries, int nq, int dim, int k, int probes, int metric,
                                     const vsr_filter* const* filters, int64_t* d_blk, int32_t* d_doc, int64_t* d_row,
                                     float* d_dist, int32_t* d_cnt)
{
    int rc = ivf_check(ivf, d_queries, nq, dim, k, probes, metric, filters, d_blk, d_dist, d_cnt, "vsr_ivf_search_device");
    if (rc || nq == 0) return rc;
    vsr_ctx* ctx = ivf->main->ctx;
    HIPCHK(hipSetDevice(ctx->device));
    std::vector<std::unique_ptr<vsr_filter>> owned;
    std::vector<const vsr_filter*> fl;
    if ((rc = ivf_plan(ivf, d_queries, nq, dim, probes, metric, filters, owned, fl))) return rc;
    return vsr_search_device_exact(nullptr, ivf->view, d_queries, nq, dim, k, metric, fl.data(), d_blk, d_doc, d_row, d_dist,
                                   d_cnt, nullptr, nullptr);
}

extern "C" int vsr_ivf_probe(vsr_ivf* ivf, const float* queries, int nq, int dim, int probes, int metric, int32_t* out_lists)
{
    if (!ivf || !queries || !out_lists || nq < 0) return fail(VSR_ERR_INVALID, "vsr_ivf_probe: NULL argument");
    if (dim != ivf->main->dim) return fail(VSR_ERR_DIM_MISMATCH, "different vector dimensions %d and %d", ivf->main->dim, dim);
    if (probes < 1) return fail(VSR_ERR_INVALID, "vsr_ivf_probe: probes must be >= 1 (got %d)", probes);
    if (nq == 0) return VSR_OK;
    probes = std::min(probes, ivf->lists);
    vsr_ctx* ctx = ivf->main->ctx;
    HIPCHK(hipSetDevice(ctx->device));
    int rc;
    if ((rc = ivf->d_q.reserve((size_t) nq * dim * sizeof(float)))) return rc;
    if ((rc = ivf->d_probe.reserve((size_t) nq * probes * sizeof(int32_t)))) return rc;
    HIPCHK(hipMemcpyAsync(ivf->d_q.p, queries, (size_t) nq * dim * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(launch_ivf_probe(ivf->d_q.as<float>(), (uint32_t) dim, (uint32_t) nq, ivf->d_centers, dim, ivf->lists, probes,
                            metric == VSR_METRIC_L2 ? M_L2 : M_IP, ivf->d_probe.as<int32_t>(), ctx->stream));
    HIPCHK(hipMemcpyAsync(out_lists, ivf->d_probe.p, (size_t) nq * probes * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return VSR_OK;
}

// Index build, the part that touches every row (ivfbuild.c:141-227: InsertTuple finds the nearest list of each heap
// row): all corpus rows against the centres, in the index's arithmetic, on the GPU.  The k-means that produces the
// centres from the sampled rows (ivfbuild.c:404-445 ComputeCenters, ivfkmeans.c) is vsr_ivf_kmeans (vsr_kmeans.hip).
extern "C" int vsr_ivf_assign(vsr_corpus* c, const float* centers, int lists, int metric, int32_t* out_row_list)
{
    if (!c || !centers || (c->n > 0 && !out_row_list)) return fail(VSR_ERR_INVALID, "vsr_ivf_assign: NULL argument");
    if (c->base) return fail(VSR_ERR_INVALID, "vsr_ivf_assign: the corpus is a view");
    if (lists < 1 || lists > 32768) return fail(VSR_ERR_UNSUPPORTED, "vsr_ivf_assign: lists must be between 1 and 32768 (got %d)", lists);
    if (metric != VSR_METRIC_L2 && metric != VSR_METRIC_IP && metric != VSR_METRIC_COSINE)
        return fail(VSR_ERR_UNSUPPORTED, "vsr_ivf_assign: metric %d has no ivfflat opclass", metric);
    vsr_ctx* ctx = c->ctx;
    HIPCHK(hipSetDevice(ctx->device));
    const int64_t n = c->n;
    if (n == 0) return VSR_OK;
    DevBuf d_centers, d_out;
    struct Guard { DevBuf &a, &b; ~Guard() { a.release(); b.release(); } } guard{d_centers, d_out};
    int rc;
    if ((rc = d_centers.reserve((size_t) lists * c->dim * sizeof(float)))) return rc;
    if ((rc = d_out.reserve((size_t) n * sizeof(int32_t)))) return rc;
    const std::vector<float> ct = transpose_centers(centers, lists, c->dim);      // (outlives the copy: synchronised below)
    HIPCHK(hipMemcpyAsync(d_centers.p, ct.data(), ct.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    // cosine opclass: rows and centres are compared by the negative inner product (spherical k-means, vector.sql:323-327)
    HIPCHK(launch_ivf_probe(reinterpret_cast<const float*>(c->d_rows), (uint32_t) c->stride4 * 4, (uint32_t) n,
                            d_centers.as<float>(), c->dim, lists, 1, metric == VSR_METRIC_L2 ? M_L2 : M_IP, d_out.as<int32_t>(),
                            ctx->stream));
    std::vector<int32_t> by_internal((size_t) n);
    HIPCHK(hipMemcpyAsync(by_internal.data(), d_out.p, (size_t) n * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (int64_t r = 0; r < n; ++r) out_row_list[c->h_orig[(size_t) r]] = by_internal[(size_t) r];     // caller's row order
    return VSR_OK;
}

// ---------------------------------------------------------------------------------------------
// K4: HNSW graph search (hnswscan.c:15-45, hnswutils.c:813-976) over a graph built elsewhere
// ---------------------------------------------------------------------------------------------
struct vsr_hnsw {
    vsr_corpus* corpus = nullptr;
    int32_t n_elem = 0, entry = -1, entry_level = -1, m = 0, max_level = 1;
    int32_t *d_elem_row = nullptr, *d_level = nullptr, *d_nbr0 = nullptr, *d_up_slot = nullptr, *d_up_nbr = nullptr,
            *d_tid_count = nullptr, *d_tids = nullptr;
    std::map<uint64_t, uint64_t*> bitmaps;           // filters (by vsr_filter::id) without a full bitmap of their own, as one
    DevBuf d_q, d_vis, d_out, d_bm;
    PinBuf h_out, h_bm;
    int last_mode = -1;                              // visited form of the last launch (HnswVisited)
    int predicate_aware = 0;                         // vsr_hnsw_set_predicate_aware
};

extern "C" int vsr_hnsw_free(vsr_hnsw* h)
{
    if (!h) return VSR_OK;
    if (h->corpus) {
        (void) hipSetDevice(h->corpus->ctx->device);
        (void) hipStreamSynchronize(h->corpus->ctx->stream);
    }
    if (h->corpus) {
        auto& reg = h->corpus->hnsw_indexes;
        reg.erase(std::remove(reg.begin(), reg.end(), h), reg.end());
    }
    void* ptrs[] = {h->d_elem_row, h->d_level, h->d_nbr0, h->d_up_slot, h->d_up_nbr, h->d_tid_count, h->d_tids};
    for (void* p : ptrs)
        if (p) (void) hipFree(p);
    for (auto& kv : h->bitmaps)
        if (kv.second) (void) hipFree(kv.second);
    delete h;
    return VSR_OK;
}

extern "C" int vsr_hnsw_load(vsr_corpus* c, int m, int32_t n_elem, int32_t entry, const int32_t* level, const int32_t* nbr0,
                             const int32_t* tid_count, const int64_t* tids, const int32_t* up_slot, const int32_t* up_nbr,
                             int32_t n_upper, int32_t max_level, vsr_hnsw** out)
{
    if (!c || !out) return fail(VSR_ERR_INVALID, "vsr_hnsw_load: NULL argument");
    *out = nullptr;
    if (c->base) return fail(VSR_ERR_INVALID, "vsr_hnsw_load: the corpus is a view");
    if (m < 2 || m > 100)        /* reloption m: 2 .. HNSW_MAX_M (hnsw.h:36-40) */
        return fail(VSR_ERR_UNSUPPORTED, "vsr_hnsw_load: m must be between 2 and 100 (got %d)", m);
    if (n_elem < 0 || (n_elem > 0 && (!level || !nbr0 || !tid_count || !tids || !up_slot)) || max_level < 1 || n_upper < 0 ||
        (n_upper > 0 && !up_nbr) || entry >= n_elem)
        return fail(VSR_ERR_INVALID, "vsr_hnsw_load: bad graph arrays");
    vsr_ctx* ctx = c->ctx;
    HIPCHK(hipSetDevice(ctx->device));
    std::unique_ptr<vsr_hnsw> h(new vsr_hnsw());
    h->corpus = c;
    h->n_elem = n_elem;
    h->entry = n_elem > 0 ? entry : -1;
    h->m = m;
    h->max_level = max_level;
    // heap TIDs arrive as caller row indices; the kernels work on internal rows
    std::vector<int32_t> inv((size_t) std::max<int64_t>(c->n, 1), -1);
    for (int64_t r = 0; r < c->n; ++r) inv[(size_t) c->h_orig[(size_t) r]] = (int32_t) r;
    std::vector<int32_t> itids((size_t) std::max(n_elem, 1) * 10, -1), erow((size_t) std::max(n_elem, 1), 0);
    for (int32_t e = 0; e < n_elem; ++e) {
        if (tid_count[e] < 1 || tid_count[e] > 10 || level[e] < 0 || level[e] > max_level)
            return fail(VSR_ERR_INVALID, "vsr_hnsw_load: element %d has %d heap TIDs / level %d", e, tid_count[e], level[e]);
        for (int t = 0; t < tid_count[e]; ++t) {
            const int64_t row = tids[(size_t) e * 10 + t];
            if (row < 0 || row >= c->n) return fail(VSR_ERR_INVALID, "vsr_hnsw_load: element %d points at row %lld", e, (long long) row);
            itids[(size_t) e * 10 + t] = inv[(size_t) row];
        }
        erow[(size_t) e] = itids[(size_t) e * 10];
        for (int j = 0; j < 2 * m; ++j)
            if (nbr0[(size_t) e * 2 * m + j] >= n_elem) return fail(VSR_ERR_INVALID, "vsr_hnsw_load: neighbour out of range");
    }
    h->entry_level = h->entry >= 0 ? level[h->entry] : -1;
    auto up = [&](int32_t** d, const int32_t* src, size_t count) -> int {
        HIPCHK(hipMalloc(d, std::max<size_t>(4, count * sizeof(int32_t))));
        if (count) HIPCHK(hipMemcpy(*d, src, count * sizeof(int32_t), hipMemcpyHostToDevice));
        return VSR_OK;
    };
    int rc;
    if ((rc = up(&h->d_elem_row, erow.data(), (size_t) n_elem)) || (rc = up(&h->d_level, level, (size_t) n_elem)) ||
        (rc = up(&h->d_nbr0, nbr0, (size_t) n_elem * 2 * m)) || (rc = up(&h->d_up_slot, up_slot, (size_t) n_elem)) ||
        (rc = up(&h->d_up_nbr, up_nbr, (size_t) n_upper * max_level * m)) || (rc = up(&h->d_tid_count, tid_count, (size_t) n_elem)) ||
        (rc = up(&h->d_tids, itids.data(), (size_t) n_elem * 10))) {
        vsr_hnsw_free(h.release());
        return rc;
    }
    c->hnsw_indexes.push_back(h.get());
    *out = h.release();
    return VSR_OK;
}

// CREATE INDEX ... USING hnsw on the GPU (vsr_hnsw_build.hip): batched insertion over the corpus's rows (element e = internal
// row e), levels from a seeded xorshift64* stream.  Returns a loaded index, as vsr_hnsw_load would from the same graph.
extern "C" int vsr_hnsw_build(vsr_corpus* c, int m, int ef_construction, int metric, uint64_t seed, vsr_hnsw** out)
{
    if (!c || !out) return fail(VSR_ERR_INVALID, "vsr_hnsw_build: NULL argument");
    *out = nullptr;
    if (c->base) return fail(VSR_ERR_INVALID, "vsr_hnsw_build: the corpus is a view");
    if (m < 2 || m > 100) return fail(VSR_ERR_UNSUPPORTED, "vsr_hnsw_build: m must be between 2 and 100 (got %d)", m);
    if (ef_construction < 4 || ef_construction > 1000 || ef_construction < 2 * m)      /* hnsw.c:62-63, hnswbuild.c:677-679 */
        return fail(VSR_ERR_UNSUPPORTED, "vsr_hnsw_build: ef_construction must be between 4 and 1000 and at least 2 * m (got %d)",
                    ef_construction);
    if (metric != VSR_METRIC_L2 && metric != VSR_METRIC_IP && metric != VSR_METRIC_COSINE)
        return fail(VSR_ERR_UNSUPPORTED, "vsr_hnsw_build: L2, inner product and cosine operator classes only");
    vsr_ctx* ctx = c->ctx;
    HIPCHK(hipSetDevice(ctx->device));
    const int64_t n = c->n;
    if (n > 0x7FFFFFF0ll) return fail(VSR_ERR_UNSUPPORTED, "vsr_hnsw_build: too many rows");
    std::unique_ptr<vsr_hnsw> h(new vsr_hnsw());
    h->corpus = c;
    h->n_elem = (int32_t) n;
    h->m = m;
    // levels: level = floor(-ln(u) * ml), ml = 1 / ln(m) (hnswutils.c:243), capped like HnswGetMaxLevel (hnsw.h:89)
    int cap = (8192 - 24 - 8 - 4 - 4) / 6 / m - 2;
    cap = std::min(cap, 255);
    uint64_t rs = seed * 0x9E3779B97F4A7C15ULL + 0x1234567ULL;               // xorshift64* (the serial CPU restatement draws the same stream)
    if (!rs) rs = 1;
    auto next = [&]() {
        uint64_t x = rs;
        x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
        rs = x;
        return x * 0x2545F4914F6CDD1DULL;
    };
    (void) next();
    const double ml = 1.0 / std::log((double) m);
    std::vector<int32_t> level((size_t) std::max<int64_t>(n, 1), 0), up_slot((size_t) std::max<int64_t>(n, 1), -1);
    int32_t max_level = 1, n_upper = 0;
    for (int64_t e = 0; e < n; ++e) {
        const double u = (double) (next() >> 11) * (1.0 / 9007199254740992.0);
        int lv = (int) (-std::log(u) * ml);
        lv = std::min(lv, cap);
        level[(size_t) e] = lv;
        if (lv >= 1) {
            up_slot[(size_t) e] = n_upper++;
            max_level = std::max(max_level, lv);
        }
    }
    h->max_level = max_level;
    const size_t alloc = (size_t) std::max<int64_t>(n, 1);
    float *d_dist0 = nullptr, *d_up_dist = nullptr;
    uint64_t *d_key[2] = {nullptr, nullptr}, *d_val[2] = {nullptr, nullptr};
    uint32_t* d_cnt = nullptr;
    void* d_tmp = nullptr;
    auto cleanup = [&]() {
        void* ptrs[] = {d_dist0, d_up_dist, d_key[0], d_key[1], d_val[0], d_val[1], d_cnt, d_tmp};
        for (void* q : ptrs)
            if (q) (void) hipFree(q);
    };
    auto bail = [&](int rc) {
        cleanup();
        vsr_hnsw_free(h.release());
        return rc;
    };
#define HB_CHK(call)                                                                                         \
    do {                                                                                                     \
        hipError_t e_ = (call);                                                                              \
        if (e_ != hipSuccess) return bail(fail(VSR_ERR_HIP, "vsr_hnsw_build: %s", hipGetErrorString(e_)));   \
    } while (0)
    const size_t up_words = (size_t) std::max(n_upper, 1) * max_level * m;
    HB_CHK(hipMalloc(&h->d_level, alloc * 4));
    HB_CHK(hipMalloc(&h->d_up_slot, alloc * 4));
    HB_CHK(hipMalloc(&h->d_nbr0, alloc * 2 * m * 4));
    HB_CHK(hipMalloc(&h->d_up_nbr, up_words * 4));
    HB_CHK(hipMalloc(&d_dist0, alloc * 2 * m * 4));
    HB_CHK(hipMalloc(&d_up_dist, up_words * 4));
    HB_CHK(hipMemcpy(h->d_level, level.data(), alloc * 4, hipMemcpyHostToDevice));
    HB_CHK(hipMemcpy(h->d_up_slot, up_slot.data(), alloc * 4, hipMemcpyHostToDevice));
    HB_CHK(hipMemsetAsync(h->d_nbr0, 0xFF, alloc * 2 * m * 4, ctx->stream));
    HB_CHK(hipMemsetAsync(h->d_up_nbr, 0xFF, up_words * 4, ctx->stream));

    HnswBuildParams bp{};
    bp.rows = c->d_rows;
    bp.stride4 = c->stride4;
    bp.metric = metric == VSR_METRIC_L2 ? M_L2 : M_IP;
    bp.m = (uint32_t) m;
    bp.efc = (uint32_t) ef_construction;
    bp.max_level = (uint32_t) max_level;
    bp.nbr0 = h->d_nbr0;
    bp.dist0 = d_dist0;
    bp.up_slot = h->d_up_slot;
    bp.up_nbr = h->d_up_nbr;
    bp.up_dist = d_up_dist;
    bp.level = h->d_level;
    bp.caps = (uint32_t) (ef_construction + 2 * m);
    uint32_t slots = 4096;
    while (slots < (uint32_t) ef_construction * 2u * (uint32_t) m * 2u && slots < 32768u) slots <<= 1;
    bp.hash_slots = slots;
    const size_t per = ((size_t) bp.caps * 8 + ((bp.caps + 15) & ~15u) + (size_t) HB_NBR * 12 + (size_t) bp.caps * 4 + (size_t) slots * 4 + 15) &
                       ~(size_t) 15;
    if (per > HN_LDS_BUDGET) return bail(fail(VSR_ERR_UNSUPPORTED, "vsr_hnsw_build: ef_construction = %d with m = %d does not fit the LDS", ef_construction, m));
    bp.lds_per_wave = (uint32_t) per;
    bp.wpb = (uint32_t) std::min<size_t>(4, HN_LDS_BUDGET / per);
    bp.err = reinterpret_cast<uint32_t*>(ctx->d_flag_total) + 4;
    const uint32_t batch_max = 4096;
    bp.rec_cap = batch_max * (uint32_t) (2 * m + std::min(max_level, 4) * m);
    HB_CHK(hipMalloc(&d_key[0], (size_t) bp.rec_cap * 8));
    HB_CHK(hipMalloc(&d_key[1], (size_t) bp.rec_cap * 8));
    HB_CHK(hipMalloc(&d_val[0], (size_t) bp.rec_cap * 8));
    HB_CHK(hipMalloc(&d_val[1], (size_t) bp.rec_cap * 8));
    HB_CHK(hipMalloc(&d_cnt, 64));
    const size_t tmp_bytes = vsr_hnsw_build_sort_bytes(bp.rec_cap);
    HB_CHK(hipMalloc(&d_tmp, std::max<size_t>(tmp_bytes, 256)));
    bp.rec_count = d_cnt;

    int32_t entry = -1, entry_level = -1;
    for (int64_t done = 0; done < n;) {
        // a batch never exceeds 1/8 of the graph it is inserted into: its elements do not see each other
        const int64_t b = std::max<int64_t>(1, std::min<int64_t>({done / 8, (int64_t) batch_max, n - done}));
        bp.entry = entry;
        bp.entry_level = entry_level;
        bp.first = (uint32_t) done;
        bp.count = (uint32_t) b;
        bp.rec_key = d_key[0];
        bp.rec_val = d_val[0];
        HB_CHK(vsr_hnsw_build_batch(bp, d_tmp, tmp_bytes, d_key[1], d_val[1], ctx->stream));
        for (int64_t e = done; e < done + b; ++e)            // HnswUpdateGraphInMemory: a higher element becomes the entry point
            if (entry < 0 || level[(size_t) e] > entry_level) {
                entry = (int32_t) e;
                entry_level = level[(size_t) e];
            }
        done += b;
    }
#undef HB_CHK
    HIPCHK(hipStreamSynchronize(ctx->stream));
    cleanup();
    h->entry = n > 0 ? entry : -1;
    h->entry_level = n > 0 ? entry_level : -1;
    // element e holds internal row e alone
    std::vector<int32_t> erow(alloc, 0), tcount(alloc, 1), itids(alloc * 10, -1);
    for (int64_t e = 0; e < n; ++e) {
        erow[(size_t) e] = (int32_t) e;
        itids[(size_t) e * 10] = (int32_t) e;
    }
    auto up = [&](int32_t** d, const int32_t* src, size_t count) -> int {
        HIPCHK(hipMalloc(d, std::max<size_t>(4, count * sizeof(int32_t))));
        if (count) HIPCHK(hipMemcpy(*d, src, count * sizeof(int32_t), hipMemcpyHostToDevice));
        return VSR_OK;
    };
    int rc;
    if ((rc = up(&h->d_elem_row, erow.data(), alloc)) || (rc = up(&h->d_tid_count, tcount.data(), alloc)) ||
        (rc = up(&h->d_tids, itids.data(), alloc * 10))) {
        vsr_hnsw_free(h.release());
        return rc;
    }
    c->hnsw_indexes.push_back(h.get());
    *out = h.release();
    return VSR_OK;
}

// what a build left: elements, entry point, its level, the highest level (for reports and tests)
extern "C" int vsr_hnsw_set_predicate_aware(vsr_hnsw* h, int on)
{
    if (!h) return fail(VSR_ERR_INVALID, "vsr_hnsw_set_predicate_aware: index is NULL");
    h->predicate_aware = on ? 1 : 0;
    return VSR_OK;
}

extern "C" int vsr_hnsw_info(const vsr_hnsw* h, int32_t* n_elem, int32_t* entry, int32_t* entry_level, int32_t* max_level)
{
    if (!h) return fail(VSR_ERR_INVALID, "vsr_hnsw_info: index is NULL");
    if (n_elem) *n_elem = h->n_elem;
    if (entry) *entry = h->entry;
    if (entry_level) *entry_level = h->entry_level;
    if (max_level) *max_level = h->max_level;
    return VSR_OK;
}

// A filter of the corpus is about to die (vsr_filter_free) or all of them are (vsr_rbac_load, corpus teardown): the
// indexes forget what they derived from it.  The caller has synchronised the corpus's stream.
static void purge_index_caches(vsr_corpus* c, const vsr_filter* f)
{
    for (vsr_ivf* ivf : c->ivf_indexes) {
        for (auto it = ivf->parts.begin(); it != ivf->parts.end();) {
            if (!f || it->first.first == f->id) {
                delete it->second;                           // tiles / bitmap are borrowed
                it = ivf->parts.erase(it);
            } else
                ++it;
        }
        for (auto it = ivf->view_bitmaps.begin(); it != ivf->view_bitmaps.end();) {
            if (!f || it->first == f->id) {
                if (it->second.d) (void) hipFree(it->second.d);
                it = ivf->view_bitmaps.erase(it);
            } else
                ++it;
        }
    }
    for (vsr_hnsw* h : c->hnsw_indexes) {
        for (auto it = h->bitmaps.begin(); it != h->bitmaps.end();) {
            if (!f || it->first == f->id) {
                if (it->second) (void) hipFree(it->second);
                it = h->bitmaps.erase(it);
            } else
                ++it;
        }
    }
}

// the rows a filter admits as a bitmap over internal rows
static int hnsw_filter_bitmap(vsr_hnsw* h, const vsr_filter* f, const uint64_t** out)
{
    vsr_corpus* c = h->corpus;
    if (f->mode == VSR_FILTER_BITMAP && f->d_bitmap) {       // role / byte-mask filters in post-filter mode: already one
        *out = f->d_bitmap;
        return VSR_OK;
    }
    auto it = h->bitmaps.find(f->id);
    if (it == h->bitmaps.end()) {
        uint64_t* d = nullptr;
        const size_t words = bitmap_words(c->n);
        HIPCHK(hipMalloc(&d, words * sizeof(uint64_t)));
        HIPCHK(hipMemsetAsync(d, 0, words * sizeof(uint64_t), c->ctx->stream));
        HIPCHK(launch_view_bitmap(nullptr, (uint32_t) c->n, f->d_tiles, f->n_tiles, f->d_bitmap, d, c->ctx->stream));
        it = h->bitmaps.emplace(f->id, d).first;
    }
    *out = it->second;
    return VSR_OK;
}

// One launch over queries resident in device memory (rows of q_stride floats), results into device arrays; bitmaps: one
// device pointer per query (d_bm, may be nullptr).  status_out / visited_out are optional device arrays.
static int hnsw_launch(vsr_hnsw* h, vsr_ctx* ctx, const float* d_q, uint32_t q_stride, int nq, int k, int ef, int metric,
                       const uint64_t* const* d_bm, bool force_global, int64_t* d_blk, int32_t* d_doc, int64_t* d_row, float* d_dist,
                       int32_t* d_cnt, int64_t* d_vis, int32_t* d_status)
{
    vsr_corpus* c = h->corpus;
    HnswParams p{};
    p.rows = c->d_rows;
    p.stride4 = c->stride4;
    p.metric = metric;
    p.queries = d_q;
    p.q_stride = q_stride;
    p.dim = (uint32_t) c->dim;
    p.nq = (uint32_t) nq;
    p.n_elem = (uint32_t) h->n_elem;
    p.entry = h->entry;
    p.entry_level = h->entry_level;
    p.m = (uint32_t) h->m;
    p.max_level = (uint32_t) h->max_level;
    p.elem_row = h->d_elem_row;
    p.nbr0 = h->d_nbr0;
    p.up_slot = h->d_up_slot;
    p.up_nbr = h->d_up_nbr;
    p.level = h->d_level;
    p.tid_count = h->d_tid_count;
    p.tids = h->d_tids;
    p.bitmaps = d_bm;
    p.predicate_aware = h->predicate_aware;
    p.ef = (uint32_t) ef;
    p.k = (uint32_t) k;
    p.caps = (uint32_t) (2 * ef + 2 * h->m + 64);
    if (!hnsw_plan(p, force_global)) {                      // S does not fit beside anything: a shorter tail behind W
        p.caps = (uint32_t) (ef + 2 * h->m + 64);
        if (!hnsw_plan(p, force_global)) return fail(VSR_ERR_UNSUPPORTED, "vsr_hnsw_search: ef_search = %d does not fit the LDS", ef);
    }
    // development / tests: VSR_HNSW_VISITED=hash[:slots] forces the LDS hash table (with `slots` entries, a power of two) on a
    // graph small enough for the LDS bitmap, so that the table and its overflow re-run can be exercised on small graphs
    if (!force_global) {
        const char* env = getenv("VSR_HNSW_VISITED");
        if (env && !strncmp(env, "hash", 4)) {
            uint32_t slots = env[4] == ':' ? (uint32_t) atoi(env + 5) : 4096u;
            while (slots & (slots - 1)) slots &= slots - 1;
            slots = std::max(64u, slots);
            const size_t fixed = hnsw_lds_fixed(p.caps);
            if (fixed + (size_t) slots * 4 <= HN_LDS_BUDGET) {
                p.vis_mode = VIS_LDS_HASH;
                p.vis_words = slots;
                p.lds_per_query = (uint32_t) ((fixed + (size_t) slots * 4 + 15) & ~(size_t) 15);
                p.qpb = 1;
            }
        } else if (env && !strcmp(env, "global")) {
            (void) hnsw_plan(p, true);
        }
    }
    if (p.vis_mode == VIS_GLOBAL) {
        int rc = h->d_vis.reserve((size_t) nq * p.vis_words * 4);
        if (rc) return rc;
        HIPCHK(hipMemsetAsync(h->d_vis.p, 0, (size_t) nq * p.vis_words * 4, ctx->stream));
        p.visited = h->d_vis.as<uint32_t>();
    }
    p.block_ids = c->d_block;
    p.doc_ids = c->d_doc;
    p.orig_rows = c->d_orig;
    p.out_block = d_blk;
    p.out_doc = d_doc;
    p.out_row = d_row;
    p.out_dist = d_dist;
    p.out_count = d_cnt;
    p.out_visited = d_vis;
    p.out_status = d_status;
    p.err = reinterpret_cast<uint32_t*>(ctx->d_flag_total) + 4;
    HIPCHK(launch_hnsw_search(p, ctx->stream));
    h->last_mode = p.vis_mode;
    return VSR_OK;
}

static int hnsw_check(vsr_hnsw* h, const void* queries, int nq, int dim, int k, int ef, int metric, const vsr_filter* const* filters,
                      const char* who)
{
    if (!h) return fail(VSR_ERR_INVALID, "%s: index is NULL", who);
    int rc = check_search_args(h->corpus, queries, nq, dim, k, metric, filters, who);
    if (rc) return rc;
    if (metric == VSR_METRIC_L1) return fail(VSR_ERR_UNSUPPORTED, "%s: L1 graphs are not supported", who);
    if (ef < 1 || ef > 5000)      /* hnsw.ef_search: 1 .. HNSW_MAX_EF_SEARCH (hnsw.c:86-89, hnsw.h:44) */
        return fail(VSR_ERR_INVALID, "%s: ef_search must be between 1 and 5000 (got %d)", who, ef);
    return VSR_OK;
}

// per-query permission bitmaps as a device array of pointers (nullptr entries: no filter); any_filter = false: none at all
static int hnsw_bitmaps(vsr_hnsw* h, vsr_ctx* ctx, const vsr_filter* const* filters, int q0, int n, bool& any_filter)
{
    any_filter = false;
    std::vector<const uint64_t*> bms((size_t) n, nullptr);
    int rc;
    for (int i = 0; i < n; ++i)
        if (filters && filters[q0 + i]) {
            if ((rc = hnsw_filter_bitmap(h, filters[q0 + i], &bms[(size_t) i]))) return rc;
            any_filter = true;
        }
    if (!any_filter) return VSR_OK;
    if ((rc = h->d_bm.reserve((size_t) n * sizeof(uint64_t*)))) return rc;
    if ((rc = h->h_bm.reserve((size_t) n * sizeof(uint64_t*)))) return rc;
    memcpy(h->h_bm.p, bms.data(), (size_t) n * sizeof(uint64_t*));
    HIPCHK(hipMemcpyAsync(h->d_bm.p, h->h_bm.p, (size_t) n * sizeof(uint64_t*), hipMemcpyHostToDevice, ctx->stream));
    return VSR_OK;
}

extern "C" int vsr_hnsw_search_device(vsr_hnsw* h, const float* d_queries, int nq, int dim, int k, int ef, int metric,
                                      const vsr_filter* const* filters, int64_t* d_blk, int32_t* d_doc, int64_t* d_row,
                                      float* d_dist, int32_t* d_cnt, int64_t* d_visited)
{
    int rc = hnsw_check(h, d_queries, nq, dim, k, ef, metric, filters, "vsr_hnsw_search_device");
    if (rc) return rc;
    if (nq == 0) return VSR_OK;
    if (!d_blk || !d_dist || !d_cnt) return fail(VSR_ERR_INVALID, "vsr_hnsw_search_device: output is NULL");
    vsr_ctx* ctx = h->corpus->ctx;
    HIPCHK(hipSetDevice(ctx->device));
    if (!d_doc) {
        if ((rc = h->d_out.reserve((size_t) nq * k * sizeof(int32_t)))) return rc;
        d_doc = h->d_out.as<int32_t>();
    }
    bool any_filter = false;
    if (h->h_bm.p) HIPCHK(hipStreamSynchronize(ctx->stream));            // the pinned pointer block of the previous call
    if ((rc = hnsw_bitmaps(h, ctx, filters, 0, nq, any_filter))) return rc;
    return hnsw_launch(h, ctx, d_queries, (uint32_t) dim, nq, k, ef, metric, any_filter ? h->d_bm.as<const uint64_t*>() : nullptr, false,
                       d_blk, d_doc, d_row, d_dist, d_cnt, d_visited, nullptr);
}

extern "C" int vsr_hnsw_search(vsr_hnsw* h, const float* queries, int nq, int dim, int k, int ef, int metric,
                               const vsr_filter* const* filters, int64_t* out_blk, int32_t* out_doc, int64_t* out_row,
                               float* out_dist, int32_t* out_cnt, int64_t* out_visited)
{
    int rc = hnsw_check(h, queries, nq, dim, k, ef, metric, filters, "vsr_hnsw_search");
    if (rc) return rc;
    if (nq == 0) return VSR_OK;
    if (!out_blk || !out_dist || !out_cnt) return fail(VSR_ERR_INVALID, "vsr_hnsw_search: output is NULL");
    vsr_corpus* c = h->corpus;
    vsr_ctx* ctx = c->ctx;
    HIPCHK(hipSetDevice(ctx->device));
    // one launch for the whole call; results come back in one copy.  Queries whose LDS visited table overflowed (big graphs
    // only) are re-run with the global bitmap
    const size_t nk = (size_t) nq * k;
    const size_t o_blk = 0, o_row = align_up(o_blk + nk * 8, 256), o_doc = align_up(o_row + nk * 8, 256),
                 o_dist = align_up(o_doc + nk * 4, 256), o_cnt = align_up(o_dist + nk * 4, 256),
                 o_vis = align_up(o_cnt + (size_t) nq * 4, 256), o_st = align_up(o_vis + (size_t) nq * 8, 256),
                 total = align_up(o_st + (size_t) nq * 4, 256);
    if ((rc = h->d_q.reserve((size_t) nq * dim * sizeof(float)))) return rc;
    if ((rc = h->d_out.reserve(total))) return rc;
    if ((rc = h->h_out.reserve(total))) return rc;
    char* d = h->d_out.as<char>();
    char* hh = h->h_out.as<char>();
    auto run = [&](const float* qs, int n, const vsr_filter* const* fs, bool force_global) -> int {
        bool any_filter = false;
        int r;
        HIPCHK(hipMemcpyAsync(h->d_q.p, qs, (size_t) n * dim * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
        if ((r = hnsw_bitmaps(h, ctx, fs, 0, n, any_filter))) return r;
        if ((r = hnsw_launch(h, ctx, h->d_q.as<float>(), (uint32_t) dim, n, k, ef, metric,
                             any_filter ? h->d_bm.as<const uint64_t*>() : nullptr, force_global, reinterpret_cast<int64_t*>(d + o_blk),
                             reinterpret_cast<int32_t*>(d + o_doc), reinterpret_cast<int64_t*>(d + o_row),
                             reinterpret_cast<float*>(d + o_dist), reinterpret_cast<int32_t*>(d + o_cnt),
                             reinterpret_cast<int64_t*>(d + o_vis), reinterpret_cast<int32_t*>(d + o_st))))
            return r;
        HIPCHK(hipMemcpyAsync(hh, d, total, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        return VSR_OK;
    };
    if ((rc = run(queries, nq, filters, false))) return rc;
    memcpy(out_blk, hh + o_blk, nk * 8);
    if (out_row) memcpy(out_row, hh + o_row, nk * 8);
    if (out_doc) memcpy(out_doc, hh + o_doc, nk * 4);
    memcpy(out_dist, hh + o_dist, nk * 4);
    memcpy(out_cnt, hh + o_cnt, (size_t) nq * 4);
    if (out_visited) memcpy(out_visited, hh + o_vis, (size_t) nq * 8);
    std::vector<int> redo;
    for (int i = 0; i < nq; ++i)
        if (reinterpret_cast<const int32_t*>(hh + o_st)[i]) redo.push_back(i);
    if (!redo.empty()) {
        std::vector<float> q2(redo.size() * (size_t) dim);
        std::vector<const vsr_filter*> f2(redo.size(), nullptr);
        for (size_t j = 0; j < redo.size(); ++j) {
            memcpy(&q2[j * (size_t) dim], queries + (size_t) redo[j] * dim, (size_t) dim * sizeof(float));
            if (filters) f2[j] = filters[redo[j]];
        }
        if ((rc = run(q2.data(), (int) redo.size(), f2.data(), true))) return rc;
        for (size_t j = 0; j < redo.size(); ++j) {
            const size_t src = j * (size_t) k, dst = (size_t) redo[j] * k;
            memcpy(out_blk + dst, reinterpret_cast<int64_t*>(hh + o_blk) + src, (size_t) k * 8);
            if (out_row) memcpy(out_row + dst, reinterpret_cast<int64_t*>(hh + o_row) + src, (size_t) k * 8);
            if (out_doc) memcpy(out_doc + dst, reinterpret_cast<int32_t*>(hh + o_doc) + src, (size_t) k * 4);
            memcpy(out_dist + dst, reinterpret_cast<float*>(hh + o_dist) + src, (size_t) k * 4);
            out_cnt[redo[j]] = reinterpret_cast<int32_t*>(hh + o_cnt)[j];
            if (out_visited) out_visited[redo[j]] = reinterpret_cast<int64_t*>(hh + o_vis)[j];
        }
    }
    return VSR_OK;
}
