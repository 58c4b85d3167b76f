// extern "C" surface of libgnark_amd.so (include/gnark_amd.h): context, buffers, MSM, NTT, group helpers,
// profiling.  Groth16 lives in groth16.hip.  Every entry point selects the context's device itself and takes the
// context mutex (one proof at a time per device, as icicle.go:821-823).
#include <stdarg.h>
#include <stdlib.h>

#include <memory>
#include <new>
#include <vector>

#include "hostops.hip.h"

namespace ga {

static thread_local char g_err[1024] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char* get_error() { return g_err; }

int abi_exception_code(const char* entry) noexcept {
    try {
        throw;   // (Lippincott function: re-raise the exception being handled to classify it)
    } catch (const std::bad_alloc&) {
        set_error("out of host memory (std::bad_alloc) under %s", entry);
        return GA_ERR_NOMEM;
    } catch (const std::exception& e) {
        set_error("internal error under %s: %s", entry, e.what());
        return GA_ERR_STATE;
    } catch (...) {
        set_error("internal error under %s: unknown exception", entry);
        return GA_ERR_STATE;
    }
}
static thread_local const char* g_entry = "";
EntryScope::EntryScope(const char* name) : prev(g_entry) { g_entry = name; }
EntryScope::~EntryScope() { g_entry = prev; }
const char* current_entry() { return g_entry; }
static uint64_t fnv1a(const char* s) {
    uint64_t h = 1469598103934665603ull;
    for (; *s; s++) h = (h ^ (uint8_t)*s) * 1099511628211ull;
    return h ? h : 1;
}

static std::atomic<int> g_table_c{0};
int table_c_override() { return g_table_c.load(); }
static thread_local int g_lane = 0;
int current_lane() { return g_lane; }
LaneScope::LaneScope(int lane) : prev(g_lane) { g_lane = lane; }
LaneScope::~LaneScope() { g_lane = prev; }

static thread_local char g_alloc_why[160] = {0};
const char* device_malloc_error(hipError_t e) { return g_alloc_why[0] ? g_alloc_why : hipGetErrorString(e); }

hipError_t device_malloc_bytes(void** p, size_t bytes) {
    // The last GA_HBM_RESERVE_MB (default 1024) MiB of the device stay free for the runtime's own dispatch-time allocations.
    // The reserve is a rule about what THIS allocation leaves behind, so it is checked before the hipMalloc, under the device's
    // mutex (two lanes cannot both pass on the same free bytes).  Allocations below 16 MiB may use the upper half of the reserve: a
    // few-KB buffer is not refused because another process or rank sharing the device has eaten a little into it.
    static const size_t reserve = []() {
        const char* e = getenv("GA_HBM_RESERVE_MB");
        return (size_t)(e ? strtoull(e, nullptr, 10) : 1024ull) << 20;
    }();
    static const size_t small = 16ull << 20;
    // one mutex per DEVICE (the reserve is a per-device rule; ga_g16_prove_multi allocates on several devices from one process and
    // must not serialise them on each other)
    constexpr int MAX_DEV = 64;
    static std::mutex alloc_mu[MAX_DEV];
    *p = nullptr;
    g_alloc_why[0] = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    std::lock_guard<std::mutex> g(alloc_mu[dev % MAX_DEV]);
    if (reserve) {
        size_t free_b = 0, total_b = 0;
        const size_t floor_b = bytes >= small ? reserve : reserve / 2;
        const hipError_t qe = hipMemGetInfo(&free_b, &total_b);
        if (qe != hipSuccess) {   // without the free-memory figure the reserve cannot be honoured: refuse rather than skip it silently
            (void)hipGetLastError();
            snprintf(g_alloc_why, sizeof(g_alloc_why), "refused: hipMemGetInfo failed (%s), the GA_HBM_RESERVE_MB rule cannot be checked", hipGetErrorString(qe));
            return hipErrorOutOfMemory;
        }
        if (free_b < bytes || free_b - bytes < floor_b) {
            snprintf(g_alloc_why, sizeof(g_alloc_why), "refused: %zu MiB free, the allocation would leave less than the %zu MiB reserve (GA_HBM_RESERVE_MB)",
                     free_b >> 20, reserve >> 20);
            return hipErrorOutOfMemory;
        }
    }
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();   // clear the sticky error: the caller reports this failure itself
        *p = nullptr;
        return e;
    }
    return hipSuccess;
}

int Ctx::scratch_get(const char* base_key, size_t bytes, void** out) {
    // every lane has its own namespace: a lane-1 proof never shares a buffer with the lane-0 work running beside it
    // GA_FAULT_THROW=<entry point> (tests of the ABI's exception barrier): a host allocation failing deep inside that entry point
    if (const uint64_t f = tun.fault_throw.load(std::memory_order_relaxed))
        if (f == fnv1a(current_entry())) throw std::bad_alloc();
    const int lane = current_lane();
    if (lane >= 2 && tun.fault_lane2_nomem.load(std::memory_order_relaxed)) {
        set_error("device scratch '%s@%d': GA_FAULT_LANE2_NOMEM is set (test knob: lanes 2/3 cannot allocate)", base_key, lane);
        return GA_ERR_NOMEM;
    }
    const std::string lane_key = lane ? std::string(base_key) + "@" + std::to_string(lane) : std::string(base_key);
    const char* key = lane_key.c_str();
    std::lock_guard<std::mutex> g(scratch_mu);
    auto it = scratch.find(key);
    if (it != scratch.end() && it->second.second >= bytes) {
        *out = it->second.first;
        return GA_OK;
    }
    if (it != scratch.end()) {
        hipFree(it->second.first);
        scratch.erase(it);
    }
    void* p = nullptr;
    // growth slack so that slowly growing requests do not reallocate every time -- capped: 1/8 of a multi-GiB sort buffer is a
    // gigabyte nobody uses (10 GiB of a 2^26 proof's 94 GiB of scratch)
    const size_t slack = bytes / 8 < (64u << 20) ? bytes / 8 : (size_t)(64u << 20);
    size_t want = bytes + slack + 256;
    hipError_t e = device_malloc(&p, want);
    if (e != hipSuccess) {
        set_error("device scratch '%s': device_malloc(%zu) failed: %s", key, want, device_malloc_error(e));
        return GA_ERR_NOMEM;
    }
    scratch[key] = std::make_pair(p, want);
    *out = p;
    return GA_OK;
}

void Ctx::scratch_free_all() {
    std::lock_guard<std::mutex> g(scratch_mu);
    for (auto& kv : scratch) hipFree(kv.second.first);
    scratch.clear();
}

void Ctx::scratch_free_lanes(int first_lane) {
    std::lock_guard<std::mutex> g(scratch_mu);
    for (auto it = scratch.begin(); it != scratch.end();) {
        const size_t at = it->first.rfind('@');
        const int lane = at == std::string::npos ? 0 : atoi(it->first.c_str() + at + 1);
        if (lane >= first_lane) {
            hipFree(it->second.first);
            it = scratch.erase(it);
        } else
            ++it;
    }
}

void Tunables::read_env() {
    // a prover on another lane may be reading the knobs while lane 0 refreshes them: every field is an atomic, parsed into a
    // local first and stored only when the environment changed it, so a reader never sees the defaults flicker
    auto num = [](const char* name, uint64_t dflt) -> uint64_t {
        const char* e = getenv(name);
        return e ? strtoull(e, nullptr, 10) : dflt;
    };
    auto put64 = [](std::atomic<uint64_t>& f, uint64_t v) {
        if (f.load(std::memory_order_relaxed) != v) f.store(v, std::memory_order_relaxed);
    };
    auto put = [](std::atomic<int>& f, int v) {
        if (f.load(std::memory_order_relaxed) != v) f.store(v, std::memory_order_relaxed);
    };
    put64(msm_max_chunk, num("GA_MSM_MAX_CHUNK", 0));
    put64(reduce_lazy_min, num("GA_REDUCE_LAZY_MIN", 1u << 14));
    put(g16_share_min_pct, (int)num("GA_G16_SHARE_MIN_PCT", 90));
    put(g16_lanes, (int)num("GA_G16_LANES", 2));
    put64(g16_table_budget_pct, num("GA_G16_TABLE_BUDGET_PCT", 0));
    put(g16_split, (int)num("GA_G16_SPLIT", 1));
    put(g16_batch_tables, (int)num("GA_G16_BATCH_TABLES", 1));
    put(ntt_coset_fold, (int)num("GA_NTT_COSET_FOLD", 1));
    put(ntt_wave_local, (int)num("GA_NTT_WAVE_LOCAL", 1));
    put(ntt_direct, (int)num("GA_NTT_DIRECT", 1));
    put(table_c, (int)num("GA_TABLE_C", 0));
    put(msm_exact_redo, (int)num("GA_MSM_EXACT_REDO", 0));
    put64(msm_fuse_min, num("GA_MSM_FUSE_MIN", 1ull << 21));
    put64(msm_task_exact_min, num("GA_MSM_TASK_EXACT_MIN", 1ull << 25));
    put(msm_xcd, (int)num("GA_MSM_XCD", 3));
    {
        const uint64_t g = num("GA_MSM_P1_GRID", 512);
        put64(msm_p1_grid, g ? g : 1);
    }
    {
        const char* e = getenv("GA_FAULT_THROW");
        put64(fault_throw, (e && *e) ? fnv1a(e) : 0);
    }
    put(fault_lane2_nomem, (int)num("GA_FAULT_LANE2_NOMEM", 0));
    const int grp = (int)num("GA_MSM_GROUP", 0);
    put(msm_group, (grp >= 2 && grp <= 256 && (grp & (grp - 1)) == 0) ? grp : 0);
    const uint64_t seg = num("GA_MSM_MIN_SEG", 256);
    if (seg >= 32) put64(msm_min_seg, seg);
    g_table_c = table_c.load();
}

typedef CtxLock Lock;

// Bring inputs to the device when they are host pointers.
struct Staged {
    Ctx* ctx;
    const void* dev = nullptr;
    void* owned = nullptr;
    int stage(const void* p, size_t bytes, bool on_device) {
        if (on_device || bytes == 0) {
            dev = p;
            return GA_OK;
        }
        hipError_t e = device_malloc(&owned, bytes);
        if (e != hipSuccess) {
            set_error("device_malloc(%zu) failed: %s", bytes, device_malloc_error(e));
            return GA_ERR_NOMEM;
        }
        GA_HIP_CHECK(hipMemcpyAsync(owned, p, bytes, hipMemcpyHostToDevice, ctx->work_stream()));
        GA_HIP_CHECK(hipStreamSynchronize(ctx->work_stream()));   // host memory must not be referenced after return
        dev = owned;
        return GA_OK;
    }
    ~Staged() {
        if (owned) hipFree(owned);
    }
};

template <class C, int G>
static int msm_impl(Ctx* ctx, const void* bases, const void* scalars, size_t n, unsigned flags, int win_lo, int win_hi,
                    void* out, int* c_out, int* nwin_out, bool want_windows) {
    typedef typename GroupField<C, G>::F F;
    int c, nwin;
    GA_CHECK(msm_plan<C>(G, n, &c, &nwin));
    if (c_out) *c_out = c;
    if (nwin_out) *nwin_out = nwin;
    if (win_hi < 0) win_hi = nwin;
    if (win_lo < 0 || win_hi > nwin || win_lo >= win_hi) {
        set_error("msm: window range [%d,%d) outside [0,%d)", win_lo, win_hi, nwin);
        return GA_ERR_INVALID;
    }
    // Split along the point axis when one call would overflow the 2^31 (point, window) pair space -- the analogue of
    // msmChunkedG1/G2 (icicle.go:362-467); partial results are added on the host.  Never needed up to 2^26 points.
    size_t max_chunk = ((size_t)1 << 31) / (size_t)(win_hi - win_lo) - 1;
    if (ctx->tun.msm_max_chunk > 0 && ctx->tun.msm_max_chunk < max_chunk)   // GA_MSM_MAX_CHUNK, like ICICLE's chunk-cap override (icicle.go:577-584)
        max_chunk = (size_t)ctx->tun.msm_max_chunk;
    if (!want_windows && n <= max_chunk) {   // the usual case: one launch sequence, the Horner step shares the window reduction's host chain
        Staged sb{ctx}, ss{ctx};
        GA_CHECK(sb.stage(bases, n * sizeof(Affine<F>), flags & GA_BASES_ON_DEVICE));
        GA_CHECK(ss.stage(scalars, n * 32, flags & GA_SCALARS_ON_DEVICE));
        XYZZ<F> sum;
        GA_CHECK((msm_windows_device<C, G>(ctx, sb.dev, ss.dev, n, (flags & GA_SCALARS_MONTGOMERY) != 0, c, win_lo, win_hi, &sum, true)));
        for (int k = 0; k < c * win_lo; k++) sum = dbl(sum);
        host_store_jac<F>(out, sum);
        return GA_OK;
    }
    std::vector<XYZZ<F>> W(win_hi - win_lo, xyzz_inf<F>());
    for (size_t done = 0; done < n || n == 0; ) {
        const size_t cn = n - done < max_chunk ? n - done : max_chunk;
        Staged sb{ctx}, ss{ctx};
        const char* bp = reinterpret_cast<const char*>(bases) + done * sizeof(Affine<F>);
        const char* sp = reinterpret_cast<const char*>(scalars) + done * 32;
        GA_CHECK(sb.stage(bp, cn * sizeof(Affine<F>), flags & GA_BASES_ON_DEVICE));
        GA_CHECK(ss.stage(sp, cn * 32, flags & GA_SCALARS_ON_DEVICE));
        std::vector<XYZZ<F>> part(win_hi - win_lo);
        GA_CHECK((msm_windows_device<C, G>(ctx, sb.dev, ss.dev, cn, (flags & GA_SCALARS_MONTGOMERY) != 0, c, win_lo, win_hi, part.data())));
        for (size_t w = 0; w < W.size(); w++) W[w] = add(W[w], part[w]);
        done += cn;
        if (n == 0) break;
    }
    if (want_windows) {
        char* o = reinterpret_cast<char*>(out);
        for (size_t w = 0; w < W.size(); w++) host_store_jac<F>(o + w * sizeof(Jac<F>), W[w]);
    } else {
        host_store_jac<F>(out, host_horner(W.data(), (int)W.size(), c));
    }
    return GA_OK;
}

}  // namespace ga

using namespace ga;

#define GA_DISPATCH_GROUP(group, ...)                           \
    switch (group) {                                            \
        case GA_G1: {                                           \
            constexpr int G = GA_G1;                            \
            __VA_ARGS__;                                        \
        } break;                                                \
        case GA_G2: {                                           \
            constexpr int G = GA_G2;                            \
            __VA_ARGS__;                                        \
        } break;                                                \
        default:                                                \
            set_error("unknown group id %d", (int)(group));     \
            return GA_ERR_INVALID;                              \
    }

extern "C" {

const char* ga_last_error(void) { return get_error(); }
const char* ga_version(void) { return "gnark_amd 0.1 (gfx950; Groth16/PLONK prover kernels: MSM G1/G2, NTT; BN254, BLS12-381)"; }

int ga_device_count(int* count) try {
    GA_ABI_ENTRY();
    GA_HIP_CHECK(hipGetDeviceCount(count));
    return GA_OK;
} GA_ABI_CATCH

int ga_ctx_create(int device, ga_ctx** out) try {
    GA_ABI_ENTRY();
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        set_error("no HIP device available (%s); libgnark_amd has no CPU fallback", e == hipSuccess ? "count=0" : hipGetErrorString(e));
        return GA_ERR_HIP;
    }
    if (device < 0 || device >= count) {
        set_error("device %d out of range (have %d)", device, count);
        return GA_ERR_INVALID;
    }
    GA_HIP_CHECK(hipSetDevice(device));
    {
        hipDeviceProp_t prop;
        GA_HIP_CHECK(hipGetDeviceProperties(&prop, device));
        if (prop.warpSize != 64) {   // (common.hip.h: every kernel is written for 64-lane wavefronts)
            set_error("device %d runs %d-lane wavefronts; libgnark_amd's kernels need 64 (gfx950)", device, prop.warpSize);
            return GA_ERR_INVALID;
        }
    }
    Ctx* c = new Ctx();
    c->device = device;
    hipStream_t* slots[3 + GA_NUM_LANES] = {&c->lane_stream[0], &c->lane_stream[1], &c->lane_stream[2], &c->lane_stream[3],
                                            &c->copy_stream, &c->slot_stream[0], &c->slot_stream[1]};
    // (measured and dropped, profiles/README.md round 3 batch H: creating the partner lanes -- or the witness lanes -- with the highest
    // stream priority changes nothing: 140.3 / 140.9 ms per proof without, 141.1 / 141.9 / 141.0 with)
    for (int k = 0; k < 3 + GA_NUM_LANES; k++) {
        hipError_t se = hipStreamCreateWithFlags(slots[k], hipStreamNonBlocking);
        if (se != hipSuccess) {
            set_error("hipStreamCreate failed: %s", hipGetErrorString(se));
            for (int q = 0; q < k; q++) hipStreamDestroy(*slots[q]);
            delete c;
            return GA_ERR_HIP;
        }
    }
    c->stream = c->lane_stream[0];
    c->tun.read_env();
    *out = reinterpret_cast<ga_ctx*>(c);
    return GA_OK;
} GA_ABI_CATCH

void ga_ctx_destroy(ga_ctx* h) try {
    GA_ABI_ENTRY();
    if (!h) return;
    Ctx* c = reinterpret_cast<Ctx*>(h);
    hipSetDevice(c->device);
    for (int l = 0; l < GA_NUM_LANES; l++) hipStreamSynchronize(c->lane_stream[l]);
    c->scratch_free_all();
    if (c->spare_domain) ntt_domain_delete(c->spare_domain);
    c->spare_domain = nullptr;
    for (void*& q : c->spare_vectors.p) {
        hipFree(q);
        q = nullptr;
    }
    c->spare_vectors.have = false;
    for (auto& s : c->stages) {
        hipEventDestroy(s.a);
        hipEventDestroy(s.b);
    }
    for (int l = 0; l < GA_NUM_LANES; l++) hipStreamDestroy(c->lane_stream[l]);
    for (int l = 0; l < GA_NUM_LANES; l++)
        if (c->host_pin[l]) hipHostFree(c->host_pin[l]);
    hipStreamDestroy(c->copy_stream);
    hipStreamDestroy(c->slot_stream[0]);
    hipStreamDestroy(c->slot_stream[1]);
    delete c;
} GA_ABI_CATCH_VOID

int ga_device_info(ga_ctx* h, char* name, size_t name_len, uint64_t* total_bytes, uint64_t* free_bytes) try {
    GA_ABI_ENTRY();
    Ctx* c = reinterpret_cast<Ctx*>(h);
    Lock l(c);
    hipDeviceProp_t p;
    GA_HIP_CHECK(hipGetDeviceProperties(&p, c->device));
    if (name && name_len) snprintf(name, name_len, "%s (%s, %d CUs)", p.name, p.gcnArchName, p.multiProcessorCount);
    size_t f = 0, t = 0;
    GA_HIP_CHECK(hipMemGetInfo(&f, &t));
    if (total_bytes) *total_bytes = t;
    if (free_bytes) *free_bytes = f;
    return GA_OK;
} GA_ABI_CATCH

int ga_malloc(ga_ctx* h, size_t bytes, void** dptr) try {
    GA_ABI_ENTRY();
    Ctx* c = reinterpret_cast<Ctx*>(h);
    Lock l(c);
    hipError_t e = device_malloc(dptr, bytes ? bytes : 16);
    if (e != hipSuccess) {
        set_error("device_malloc(%zu) failed: %s", bytes, device_malloc_error(e));
        return GA_ERR_NOMEM;
    }
    return GA_OK;
} GA_ABI_CATCH

int ga_free(ga_ctx* h, void* dptr) try {
    GA_ABI_ENTRY();
    Ctx* c = reinterpret_cast<Ctx*>(h);
    Lock l(c);
    GA_HIP_CHECK(hipStreamSynchronize(c->stream));
    GA_HIP_CHECK(hipFree(dptr));
    return GA_OK;
} GA_ABI_CATCH

int ga_copy_to_device(ga_ctx* h, void* dst, const void* src, size_t bytes) try {
    GA_ABI_ENTRY();
    Ctx* c = reinterpret_cast<Ctx*>(h);
    Lock l(c);
    GA_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    GA_HIP_CHECK(hipStreamSynchronize(c->stream));
    return GA_OK;
} GA_ABI_CATCH

int ga_copy_to_host(ga_ctx* h, void* dst, const void* src, size_t bytes) try {
    GA_ABI_ENTRY();
    Ctx* c = reinterpret_cast<Ctx*>(h);
    Lock l(c);
    GA_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    GA_HIP_CHECK(hipStreamSynchronize(c->stream));
    return GA_OK;
} GA_ABI_CATCH

int ga_sync(ga_ctx* h) try {   // (every entry point returns with its own work finished; this waits for both lanes' streams)
    GA_ABI_ENTRY();
    Ctx* c = reinterpret_cast<Ctx*>(h);
    Lock l(c);
    for (int l = 0; l < GA_NUM_LANES; l++) GA_HIP_CHECK(hipStreamSynchronize(c->lane_stream[l]));
    return GA_OK;
} GA_ABI_CATCH

// ---- MSM ------------------------------------------------------------------------------------------------
int ga_msm(ga_ctx* h, int curve, int group, const void* bases, const void* scalars, size_t n, unsigned flags, void* out_jac) try {
    GA_ABI_ENTRY();
    Ctx* c = reinterpret_cast<Ctx*>(h);
    if (!c || !out_jac || (n && (!bases || !scalars))) {
        set_error("ga_msm: null argument");
        return GA_ERR_INVALID;
    }
    Lock l(c);
    GA_DISPATCH_CURVE(curve, GA_DISPATCH_GROUP(group, return (msm_impl<C, G>(c, bases, scalars, n, flags, 0, -1, out_jac, nullptr, nullptr, false))));
    return GA_OK;
} GA_ABI_CATCH

int ga_msm_windows(ga_ctx* h, int curve, int group, const void* bases, const void* scalars, size_t n, unsigned flags, int win_lo,
                   int win_hi, void* out_windows, int* window_bits, int* num_windows) try {
    GA_ABI_ENTRY();
    Ctx* c = reinterpret_cast<Ctx*>(h);
    if (!c || !out_windows || (n && (!bases || !scalars))) {
        set_error("ga_msm_windows: null argument");
        return GA_ERR_INVALID;
    }
    Lock l(c);
    GA_DISPATCH_CURVE(curve, GA_DISPATCH_GROUP(group, return (msm_impl<C, G>(c, bases, scalars, n, flags, win_lo, win_hi, out_windows,
                                                                            window_bits, num_windows, true))));
    return GA_OK;
} GA_ABI_CATCH

int ga_msm_plan(int curve, int group, size_t n, int* window_bits, int* num_windows) try {
    GA_ABI_ENTRY();
    GA_DISPATCH_CURVE(curve, return msm_plan<C>(group, n, window_bits, num_windows));
    return GA_OK;
} GA_ABI_CATCH

int ga_msm_combine_windows(int curve, int group, const void* windows, int num_windows, int window_bits, void* out_jac) try {
    GA_ABI_ENTRY();
    if (!windows || !out_jac || num_windows <= 0 || window_bits <= 0) {
        set_error("ga_msm_combine_windows: bad argument");
        return GA_ERR_INVALID;
    }
    GA_DISPATCH_CURVE(curve, GA_DISPATCH_GROUP(group, {
                          typedef typename GroupField<C, G>::F F;
                          std::vector<XYZZ<F>> W(num_windows);
                          for (int w = 0; w < num_windows; w++)
                              W[w] = host_load_jac<F>(reinterpret_cast<const char*>(windows) + (size_t)w * sizeof(Jac<F>));
                          host_store_jac<F>(out_jac, host_horner(W.data(), num_windows, window_bits));
                      }));
    return GA_OK;
} GA_ABI_CATCH

// ---- precomputed tables ------------------------------------------------------------------------------------
struct MsmTable {
    Ctx* ctx;
    int curve, group, c, nwin;
    size_t n;
    void* d_table;
    uint64_t bytes;
};

int ga_msm_table_create(ga_ctx* h, int curve, int group, const void* bases, size_t n, unsigned flags, ga_msm_table** out) try {
    GA_ABI_ENTRY();
    Ctx* c = reinterpret_cast<Ctx*>(h);
    if (!c || !out || !bases || n == 0) {
        set_error("ga_msm_table_create: bad argument");
        return GA_ERR_INVALID;
    }
    Lock l(c);
    struct Owner {   // (the table object and its device memory are released on every way out but the successful one)
        MsmTable* t;
        ~Owner() {
            if (t) {
                hipFree(t->d_table);
                delete t;
            }
        }
    } own{new MsmTable{c, curve, group, 0, 0, n, nullptr, 0}};
    MsmTable* t = own.t;
    int rc = GA_OK;
    GA_DISPATCH_CURVE(curve, GA_DISPATCH_GROUP(group, {
                          typedef typename GroupField<C, G>::F F;
                          rc = msm_plan_table<C>(n, &t->c, &t->nwin, (flags & GA_TABLE_BATCHED) != 0);
                          t->bytes = (uint64_t)t->nwin * n * msm_table_point_bytes<C, G>();
                          Staged sb{c};
                          if (rc == GA_OK) rc = sb.stage(bases, n * sizeof(Affine<F>), flags & GA_BASES_ON_DEVICE);
                          if (rc == GA_OK && device_malloc(&t->d_table, t->bytes) != hipSuccess) {
                              set_error("ga_msm_table_create: device_malloc(%llu) failed", (unsigned long long)t->bytes);
                              rc = GA_ERR_NOMEM;
                          }
                          if (rc == GA_OK) rc = msm_table_build<C, G>(c, sb.dev, n, t->c, t->d_table);
                          if (rc == GA_OK && hipStreamSynchronize(c->stream) != hipSuccess) {
                              set_error("ga_msm_table_create: table kernel failed");
                              rc = GA_ERR_HIP;
                          }
                      }));
    if (rc != GA_OK) return rc;
    own.t = nullptr;
    *out = reinterpret_cast<ga_msm_table*>(t);
    return GA_OK;
} GA_ABI_CATCH

void ga_msm_table_destroy(ga_msm_table* th) try {
    GA_ABI_ENTRY();
    MsmTable* t = reinterpret_cast<MsmTable*>(th);
    if (!t) return;
    Lock l(t->ctx);
    hipStreamSynchronize(t->ctx->stream);
    t->ctx->forget_table(t->d_table);
    hipFree(t->d_table);
    delete t;
} GA_ABI_CATCH_VOID

int ga_msm_table_info(ga_msm_table* th, int* window_bits, int* num_windows, uint64_t* table_bytes) try {
    GA_ABI_ENTRY();
    MsmTable* t = reinterpret_cast<MsmTable*>(th);
    if (!t) return GA_ERR_INVALID;
    if (window_bits) *window_bits = t->c;
    if (num_windows) *num_windows = t->nwin;
    if (table_bytes) *table_bytes = t->bytes;
    return GA_OK;
} GA_ABI_CATCH

int ga_msm_table_run(ga_msm_table* th, const void* scalars, unsigned flags, void* out_jac) try {
    GA_ABI_ENTRY();
    MsmTable* t = reinterpret_cast<MsmTable*>(th);
    if (!t || !scalars || !out_jac) {
        set_error("ga_msm_table_run: null argument");
        return GA_ERR_INVALID;
    }
    Ctx* c = t->ctx;
    LaneLock l(c);   // a second caller (the PLONK prover commits from several goroutines) runs beside the first on lane 1
    GA_DISPATCH_CURVE(t->curve, GA_DISPATCH_GROUP(t->group, {
                          typedef typename GroupField<C, G>::F F;
                          Staged ss{c};
                          GA_CHECK(ss.stage(scalars, t->n * 32, flags & GA_SCALARS_ON_DEVICE));
                          XYZZ<F> sum;
                          GA_CHECK((msm_table_device<C, G>(c, t->d_table, ss.dev, t->n, (flags & GA_SCALARS_MONTGOMERY) != 0, t->c, &sum)));
                          host_store_jac<F>(out_jac, sum);
                      }));
    return GA_OK;
} GA_ABI_CATCH

int ga_msm_table_run_batch(ga_msm_table* th, const void* const* scalars, uint32_t k, unsigned flags, void* out_jacs) try {
    GA_ABI_ENTRY();
    MsmTable* t = reinterpret_cast<MsmTable*>(th);
    if (!t || !scalars || !out_jacs || k == 0 || k > 16) {
        set_error("ga_msm_table_run_batch: null argument or batch size %u outside [1, 16]", k);
        return GA_ERR_INVALID;
    }
    for (uint32_t j = 0; j < k; j++)
        if (!scalars[j]) {
            set_error("ga_msm_table_run_batch: scalars[%u] is null", j);
            return GA_ERR_INVALID;
        }
    if ((uint64_t)k * t->nwin * t->n >= (1ull << 31)) {
        set_error("ga_msm_table_run_batch: %u vectors x %d windows x %zu points exceed the 2^31 pair index space; use smaller batches", k,
                  t->nwin, (size_t)t->n);
        return GA_ERR_INVALID;
    }
    Ctx* c = t->ctx;
    LaneLock l(c);   // a second caller (the PLONK prover commits from several goroutines) runs beside the first on lane 1
    GA_DISPATCH_CURVE(t->curve, GA_DISPATCH_GROUP(t->group, {
                          typedef typename GroupField<C, G>::F F;
                          std::vector<std::unique_ptr<Staged>> st;
                          std::vector<const void*> dev(k);
                          for (uint32_t j = 0; j < k; j++) {
                              st.emplace_back(new Staged{c});
                              GA_CHECK(st.back()->stage(scalars[j], t->n * 32, flags & GA_SCALARS_ON_DEVICE));
                              dev[j] = st.back()->dev;
                          }
                          std::vector<XYZZ<F>> sums(k);
                          GA_CHECK((msm_table_device_batch<C, G>(c, t->d_table, dev.data(), (int)k, t->n, (flags & GA_SCALARS_MONTGOMERY) != 0,
                                                                 t->c, sums.data())));
                          for (uint32_t j = 0; j < k; j++) host_store_jac<F>((char*)out_jacs + j * sizeof(Jac<F>), sums[j]);
                      }));
    return GA_OK;
} GA_ABI_CATCH

int ga_msm_table_run_windows(ga_msm_table* th, const void* scalars, unsigned flags, int win_lo, int win_hi, void* out_jac) try {
    GA_ABI_ENTRY();
    MsmTable* t = reinterpret_cast<MsmTable*>(th);
    if (!t || !scalars || !out_jac) {
        set_error("ga_msm_table_run_windows: null argument");
        return GA_ERR_INVALID;
    }
    if (win_lo < 0 || win_hi > t->nwin || win_hi < win_lo) {
        set_error("ga_msm_table_run_windows: window range [%d,%d) outside [0,%d)", win_lo, win_hi, t->nwin);
        return GA_ERR_INVALID;
    }
    Ctx* c = t->ctx;
    LaneLock l(c);   // a second caller (the PLONK prover commits from several goroutines) runs beside the first on lane 1
    GA_DISPATCH_CURVE(t->curve, GA_DISPATCH_GROUP(t->group, {
                          typedef typename GroupField<C, G>::F F;
                          Staged ss{c};
                          GA_CHECK(ss.stage(scalars, t->n * 32, flags & GA_SCALARS_ON_DEVICE));
                          XYZZ<F> sum;
                          GA_CHECK((msm_table_device<C, G>(c, t->d_table, ss.dev, t->n, (flags & GA_SCALARS_MONTGOMERY) != 0, t->c, &sum, win_lo,
                                                           win_hi)));
                          host_store_jac<F>(out_jac, sum);
                      }));
    return GA_OK;
} GA_ABI_CATCH

int ga_fr_linear_combination(ga_ctx* h, int curve, uint64_t n, int k, const void* const* vecs, const void* scalars, void* out,
                             int on_device) try {
    GA_ABI_ENTRY();
    Ctx* c = reinterpret_cast<Ctx*>(h);
    if (!c || !vecs || !scalars || (!out && n)) {
        set_error("ga_fr_linear_combination: null argument");
        return GA_ERR_INVALID;
    }
    for (int j = 0; j < k; j++)
        if (!vecs[j]) {
            set_error("ga_fr_linear_combination: vector %d is null", j);
            return GA_ERR_INVALID;
        }
    Lock l(c);
    GA_DISPATCH_CURVE(curve, GA_CHECK(fr_vec_lincomb<C>(c, n, k, vecs, scalars, out, on_device != 0)));
    return GA_OK;
} GA_ABI_CATCH

// p(point) for a polynomial in canonical form (iop.Polynomial.Evaluate / evaluateBlinded, prove.go:1186-1215)
int ga_fr_poly_evaluate(ga_ctx* h, int curve, const void* poly, uint64_t n, const void* point, void* value_out, int on_device) try {
    GA_ABI_ENTRY();
    Ctx* c = reinterpret_cast<Ctx*>(h);
    if (!c || (!poly && n) || !point || !value_out) {
        set_error("ga_fr_poly_evaluate: null argument");
        return GA_ERR_INVALID;
    }
    Lock l(c);
    if (n == 0) {
        memset(value_out, 0, 32);
        return GA_OK;
    }
    Staged sp{c};
    GA_CHECK(sp.stage(poly, n * 32, on_device));
    void* q;
    GA_CHECK(c->scratch_get("kzg_quotient", n * 32, &q));
    GA_DISPATCH_CURVE(curve, GA_CHECK(kzg_domain_divide<C>(c, sp.dev, n, point, q, value_out)));
    return GA_OK;
} GA_ABI_CATCH

// kzg.Open(p, point, pk): claimed value p(point) and the commitment to (p(X) - p(point)) / (X - point) over the pinned SRS
int ga_kzg_open(ga_msm_table* th, const void* poly, size_t n, unsigned flags, const void* point, void* claimed_value_out, void* h_out) try {
    GA_ABI_ENTRY();
    MsmTable* t = reinterpret_cast<MsmTable*>(th);
    if (!t || !poly || !point || !claimed_value_out || !h_out || n == 0) {
        set_error("ga_kzg_open: null argument or empty polynomial");
        return GA_ERR_INVALID;
    }
    if (t->group != GA_G1 || n - 1 > t->n) {
        set_error("ga_kzg_open: the SRS table must be G1 and hold at least len(p) - 1 = %zu points (it has %zu)", n - 1, t->n);
        return GA_ERR_INVALID;
    }
    Ctx* c = t->ctx;
    Lock l(c);
    GA_DISPATCH_CURVE(t->curve, {
        typedef typename GroupField<C, GA_G1>::F F;
        Staged sp{c};
        GA_CHECK(sp.stage(poly, n * 32, flags & GA_SCALARS_ON_DEVICE));
        const size_t qn = n > t->n ? n : t->n;
        void* q;
        GA_CHECK(c->scratch_get("kzg_quotient", qn * 32, &q));
        GA_HIP_CHECK(hipMemsetAsync(q, 0, qn * 32, c->stream));
        GA_CHECK(kzg_domain_divide<C>(c, sp.dev, n, point, q, claimed_value_out));
        XYZZ<F> sum;
        GA_CHECK((msm_table_device<C, GA_G1>(c, t->d_table, q, t->n, true, t->c, &sum)));
        host_store_jac<F>(h_out, sum);
    });
    return GA_OK;
} GA_ABI_CATCH

// ---- host group helpers ---------------------------------------------------------------------------------
int ga_jac_add(int curve, int group, const void* a, const void* b, void* out) try {
    GA_ABI_ENTRY();
    GA_DISPATCH_CURVE(curve, GA_DISPATCH_GROUP(group, {
                          typedef typename GroupField<C, G>::F F;
                          host_store_jac<F>(out, add(host_load_jac<F>(a), host_load_jac<F>(b)));
                      }));
    return GA_OK;
} GA_ABI_CATCH

int ga_jac_to_affine(int curve, int group, const void* a, void* out) try {
    GA_ABI_ENTRY();
    GA_DISPATCH_CURVE(curve, GA_DISPATCH_GROUP(group, {
                          typedef typename GroupField<C, G>::F F;
                          host_store_affine<F>(out, host_load_jac<F>(a));
                      }));
    return GA_OK;
} GA_ABI_CATCH

int ga_jac_scalar_mul(int curve, int group, const void* a, const void* k, void* out) try {
    GA_ABI_ENTRY();
    GA_DISPATCH_CURVE(curve, GA_DISPATCH_GROUP(group, {
                          typedef typename GroupField<C, G>::F F;
                          uint32_t kw[8];
                          memcpy(kw, k, 32);
                          host_store_jac<F>(out, scalar_mul(host_load_jac<F>(a), kw, 8));
                      }));
    return GA_OK;
} GA_ABI_CATCH

// ---- NTT ------------------------------------------------------------------------------------------------
int ga_domain_create(ga_ctx* h, int curve, uint64_t cardinality, ga_domain** out) try {
    GA_ABI_ENTRY();
    Ctx* c = reinterpret_cast<Ctx*>(h);
    if (!c || !out || cardinality == 0) {
        set_error("ga_domain_create: bad argument");
        return GA_ERR_INVALID;
    }
    Lock l(c);
    Domain* d = nullptr;
    GA_DISPATCH_CURVE(curve, GA_CHECK(ntt_domain_new<C>(c, cardinality, &d)));
    *out = reinterpret_cast<ga_domain*>(d);
    return GA_OK;
} GA_ABI_CATCH

void ga_domain_destroy(ga_domain* dh) try {
    GA_ABI_ENTRY();
    if (!dh) return;
    Domain* d = reinterpret_cast<Domain*>(dh);
    Lock l(ntt_domain_ctx(d));
    hipStreamSynchronize(ntt_domain_ctx(d)->stream);
    ntt_domain_delete(d);
} GA_ABI_CATCH_VOID

int ga_fft(ga_domain* dh, void* data, int direction, int decimation, int on_coset, int on_device) try {
    GA_ABI_ENTRY();
    Domain* d = reinterpret_cast<Domain*>(dh);
    if (!d || !data || (direction != GA_FFT_FORWARD && direction != GA_FFT_INVERSE) || (decimation != GA_DIF && decimation != GA_DIT)) {
        set_error("ga_fft: bad argument");
        return GA_ERR_INVALID;
    }
    Ctx* c = ntt_domain_ctx(d);
    Lock l(c);
    size_t bytes = ntt_domain_size(d) * 32;
    void* dev = data;
    if (!on_device) {
        GA_CHECK(c->scratch_get("fft_io", bytes, &dev));
        GA_HIP_CHECK(hipMemcpyAsync(dev, data, bytes, hipMemcpyHostToDevice, c->stream));
    }
    GA_DISPATCH_CURVE(ntt_domain_curve(d), GA_CHECK(ntt_domain_fft<C>(d, dev, direction, decimation, on_coset)));
    if (!on_device) GA_HIP_CHECK(hipMemcpyAsync(data, dev, bytes, hipMemcpyDeviceToHost, c->stream));
    GA_HIP_CHECK(hipStreamSynchronize(c->stream));
    return GA_OK;
} GA_ABI_CATCH

int ga_plonk_quotient(ga_domain* dh0, ga_domain* dh1, const ga_plonk_quotient_in* in, void* h_out) try {
    GA_ABI_ENTRY();
    Domain *d0 = reinterpret_cast<Domain*>(dh0), *d1 = reinterpret_cast<Domain*>(dh1);
    if (!d0 || !d1 || !in || !h_out || !in->l || !in->r || !in->o || !in->z || !in->ql || !in->qr || !in->qm || !in->qo || !in->qk ||
        !in->s1 || !in->s2 || !in->s3 || !in->bl || !in->br || !in->bo || !in->bz || !in->alpha || !in->beta || !in->gamma ||
        (in->nb_bsb && (!in->qcp || !in->pi2))) {
        set_error("ga_plonk_quotient: null argument");
        return GA_ERR_INVALID;
    }
    if (in->nb_bsb > (uint32_t)PLONK_MAX_BSB || ntt_domain_curve(d0) != ntt_domain_curve(d1) || ntt_domain_ctx(d0) != ntt_domain_ctx(d1)) {
        set_error("ga_plonk_quotient: more than %d BSB22 gates, or the two domains differ in curve/context", PLONK_MAX_BSB);
        return GA_ERR_INVALID;
    }
    Ctx* c = ntt_domain_ctx(d0);
    Lock l(c);
    PlonkQuotientArgs a;
    memset(&a, 0, sizeof(a));
    a.nb_bsb = in->nb_bsb;
    const void* fixed[PLONK_NB_FIXED] = {in->l, in->r, in->o, in->z, in->ql, in->qr, in->qm, in->qo, in->qk, in->s1, in->s2, in->s3};
    for (int k = 0; k < PLONK_NB_FIXED; k++) a.polys[k] = fixed[k];
    for (uint32_t k = 0; k < in->nb_bsb; k++) {
        if (!in->qcp[k] || !in->pi2[k]) {
            set_error("ga_plonk_quotient: null Qcp / Pi2 polynomial %u", k);
            return GA_ERR_INVALID;
        }
        a.polys[PLONK_NB_FIXED + 2 * k] = in->qcp[k];
        a.polys[PLONK_NB_FIXED + 2 * k + 1] = in->pi2[k];
    }
    a.lagrange_mask = in->lagrange_mask;
    a.on_device = (in->flags & GA_PLONK_ON_DEVICE) != 0;
    a.bl = in->bl; a.br = in->br; a.bo = in->bo; a.bz = in->bz;
    a.alpha = in->alpha; a.beta = in->beta; a.gamma = in->gamma;
    GA_DISPATCH_CURVE(ntt_domain_curve(d0), GA_CHECK(plonk_domain_quotient<C>(d0, d1, a, h_out)));
    return GA_OK;
} GA_ABI_CATCH

static int plonk_args_from(const ga_plonk_quotient_in* in, bool need_fixed, bool need_var, PlonkQuotientArgs* a) {
    memset(a, 0, sizeof(*a));
    if (in->nb_bsb > (uint32_t)PLONK_MAX_BSB) {
        set_error("plonk: more than %d BSB22 gates", PLONK_MAX_BSB);
        return GA_ERR_INVALID;
    }
    a->nb_bsb = in->nb_bsb;
    const void* fixed[PLONK_NB_FIXED] = {in->l, in->r, in->o, in->z, in->ql, in->qr, in->qm, in->qo, in->qk, in->s1, in->s2, in->s3};
    const bool is_fixed[PLONK_NB_FIXED] = {false, false, false, false, true, true, true, true, false, true, true, true};
    for (int k = 0; k < PLONK_NB_FIXED; k++) {
        if ((is_fixed[k] ? need_fixed : need_var) && !fixed[k]) {
            set_error("plonk: polynomial %d of the input is null", k);
            return GA_ERR_INVALID;
        }
        a->polys[k] = fixed[k];
    }
    for (uint32_t k = 0; k < in->nb_bsb; k++) {
        if ((need_fixed && (!in->qcp || !in->qcp[k])) || (need_var && (!in->pi2 || !in->pi2[k]))) {
            set_error("plonk: null Qcp / Pi2 polynomial %u", k);
            return GA_ERR_INVALID;
        }
        a->polys[PLONK_NB_FIXED + 2 * k] = in->qcp ? in->qcp[k] : nullptr;
        a->polys[PLONK_NB_FIXED + 2 * k + 1] = in->pi2 ? in->pi2[k] : nullptr;
    }
    if (need_var && (!in->bl || !in->br || !in->bo || !in->bz || !in->alpha || !in->beta || !in->gamma)) {
        set_error("plonk: blinding polynomials and challenges are required");
        return GA_ERR_INVALID;
    }
    a->lagrange_mask = in->lagrange_mask;
    a->on_device = (in->flags & GA_PLONK_ON_DEVICE) != 0;
    a->bl = in->bl; a->br = in->br; a->bo = in->bo; a->bz = in->bz;
    a->alpha = in->alpha; a->beta = in->beta; a->gamma = in->gamma;
    return GA_OK;
}

int ga_plonk_pk_create(ga_domain* dh0, ga_domain* dh1, const ga_plonk_quotient_in* in, ga_plonk_pk** out) try {
    GA_ABI_ENTRY();
    Domain *d0 = reinterpret_cast<Domain*>(dh0), *d1 = reinterpret_cast<Domain*>(dh1);
    if (!d0 || !d1 || !in || !out || ntt_domain_curve(d0) != ntt_domain_curve(d1) || ntt_domain_ctx(d0) != ntt_domain_ctx(d1)) {
        set_error("ga_plonk_pk_create: null argument, or the two domains differ in curve/context");
        return GA_ERR_INVALID;
    }
    Ctx* c = ntt_domain_ctx(d0);
    Lock l(c);
    PlonkQuotientArgs a;
    GA_CHECK(plonk_args_from(in, true, false, &a));
    PlonkFixed* fx = nullptr;
    GA_DISPATCH_CURVE(ntt_domain_curve(d0), GA_CHECK(plonk_domain_fixed_create<C>(d0, d1, a, &fx)));
    *out = reinterpret_cast<ga_plonk_pk*>(fx);
    return GA_OK;
} GA_ABI_CATCH

void ga_plonk_pk_destroy(ga_plonk_pk* p) try {
    GA_ABI_ENTRY();
    PlonkFixed* fx = reinterpret_cast<PlonkFixed*>(p);
    if (!fx) return;
    Ctx* c = ntt_domain_ctx(plonk_fixed_domain0(fx));
    Lock l(c);
    hipStreamSynchronize(c->stream);
    plonk_fixed_delete(fx);
} GA_ABI_CATCH_VOID

int ga_plonk_quotient_pinned(ga_plonk_pk* p, const ga_plonk_quotient_in* in, void* h_out) try {
    GA_ABI_ENTRY();
    PlonkFixed* fx = reinterpret_cast<PlonkFixed*>(p);
    if (!fx || !in || !h_out) {
        set_error("ga_plonk_quotient_pinned: null argument");
        return GA_ERR_INVALID;
    }
    Domain* d0 = plonk_fixed_domain0(fx);
    Ctx* c = ntt_domain_ctx(d0);
    Lock l(c);
    PlonkQuotientArgs a;
    GA_CHECK(plonk_args_from(in, false, true, &a));
    GA_DISPATCH_CURVE(ntt_domain_curve(d0), GA_CHECK(plonk_domain_quotient_pinned<C>(fx, a, h_out)));
    return GA_OK;
} GA_ABI_CATCH

int ga_plonk_build_z(ga_domain* dh0, const void* lv, const void* rv, const void* ov, const int64_t* permutation, const void* beta,
                     const void* gamma, int on_device, void* z_out) try {
    GA_ABI_ENTRY();
    Domain* d0 = reinterpret_cast<Domain*>(dh0);
    if (!d0 || !lv || !rv || !ov || !permutation || !beta || !gamma || !z_out) {
        set_error("ga_plonk_build_z: null argument");
        return GA_ERR_INVALID;
    }
    Ctx* c = ntt_domain_ctx(d0);
    Lock l(c);
    if (!on_device) {
        const uint64_t n3 = 3 * ntt_domain_size(d0);
        for (uint64_t i = 0; i < n3; i++)
            if (permutation[i] < 0 || (uint64_t)permutation[i] >= n3) {
                set_error("ga_plonk_build_z: permutation[%llu] = %lld is outside [0, 3n)", (unsigned long long)i, (long long)permutation[i]);
                return GA_ERR_INVALID;
            }
    }
    GA_DISPATCH_CURVE(ntt_domain_curve(d0), GA_CHECK(plonk_domain_build_z<C>(d0, lv, rv, ov, permutation, beta, gamma, on_device != 0, z_out)));
    return GA_OK;
} GA_ABI_CATCH

int ga_fr_batch_invert(ga_ctx* h, int curve, void* v, uint64_t n, int on_device) try {
    GA_ABI_ENTRY();
    Ctx* c = reinterpret_cast<Ctx*>(h);
    if (!c || (!v && n)) {
        set_error("ga_fr_batch_invert: null argument");
        return GA_ERR_INVALID;
    }
    Lock l(c);
    GA_DISPATCH_CURVE(curve, GA_CHECK(fr_vec_batch_inverse<C>(c, v, n, on_device != 0)));
    return GA_OK;
} GA_ABI_CATCH

int ga_compute_h(ga_domain* dh, const void* a, const void* b, const void* cc, uint64_t n_constraints, void* h_out, int on_device) try {
    GA_ABI_ENTRY();
    Domain* d = reinterpret_cast<Domain*>(dh);
    if (!d || !a || !b || !cc || !h_out || n_constraints > ntt_domain_size(d)) {
        set_error("ga_compute_h: bad argument (n_constraints must be <= domain cardinality)");
        return GA_ERR_INVALID;
    }
    Ctx* c = ntt_domain_ctx(d);
    Lock l(c);
    const uint64_t n = ntt_domain_size(d);
    const size_t full = n * 32, part = n_constraints * 32;
    void *da, *db, *dc;
    GA_CHECK(c->scratch_get("compute_h_a", full, &da));
    GA_CHECK(c->scratch_get("compute_h_b", full, &db));
    GA_CHECK(c->scratch_get("compute_h_c", full, &dc));
    const void* src[3] = {a, b, cc};
    void* dst[3] = {da, db, dc};
    for (int k = 0; k < 3; k++) {
        GA_HIP_CHECK(hipMemcpyAsync(dst[k], src[k], part, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->stream));
        if (full > part) GA_HIP_CHECK(hipMemsetAsync(reinterpret_cast<char*>(dst[k]) + part, 0, full - part, c->stream));
    }
    GA_DISPATCH_CURVE(ntt_domain_curve(d), GA_CHECK(ntt_domain_compute_h<C>(d, da, db, dc)));
    GA_HIP_CHECK(hipMemcpyAsync(h_out, da, full, on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->stream));
    GA_HIP_CHECK(hipStreamSynchronize(c->stream));
    return GA_OK;
} GA_ABI_CATCH

// ---- profiling ------------------------------------------------------------------------------------------
int ga_profile_enable(ga_ctx* h, int on) try {
    GA_ABI_ENTRY();
    Ctx* c = reinterpret_cast<Ctx*>(h);
    Lock l(c);
    c->profiling = on != 0;
    return GA_OK;
} GA_ABI_CATCH

int ga_profile_reset(ga_ctx* h) try {
    GA_ABI_ENTRY();
    Ctx* c = reinterpret_cast<Ctx*>(h);
    Lock l(c);
    hipStreamSynchronize(c->stream);
    for (auto& s : c->stages) {
        hipEventDestroy(s.a);
        hipEventDestroy(s.b);
    }
    c->stages.clear();
    return GA_OK;
} GA_ABI_CATCH

int ga_profile_read(ga_ctx* h, char* buf, size_t cap) try {
    GA_ABI_ENTRY();
    Ctx* c = reinterpret_cast<Ctx*>(h);
    Lock l(c);
    GA_HIP_CHECK(hipStreamSynchronize(c->stream));
    std::string out;
    for (auto& s : c->stages) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, s.a, s.b) != hipSuccess) ms = -1;
        char tmp[160];
        snprintf(tmp, sizeof(tmp), "%s=%.6f;", s.name.c_str(), ms);
        out += tmp;
    }
    if (!buf || cap == 0) return GA_ERR_INVALID;
    snprintf(buf, cap, "%s", out.c_str());
    return GA_OK;
} GA_ABI_CATCH

// ---- test / bench support -------------------------------------------------------------------------------
int ga_gen_bases(ga_ctx* h, int curve, int group, uint64_t seed, size_t n, void* bases_dev, void* dlogs_dev) try {
    GA_ABI_ENTRY();
    Ctx* c = reinterpret_cast<Ctx*>(h);
    Lock l(c);
    GA_DISPATCH_CURVE(curve, GA_DISPATCH_GROUP(group, GA_CHECK((util_gen_bases<C, G>(c, seed, n, bases_dev, dlogs_dev)))));
    GA_HIP_CHECK(hipStreamSynchronize(c->stream));
    return GA_OK;
} GA_ABI_CATCH

int ga_gen_bases_at(ga_ctx* h, int curve, int group, uint64_t seed, uint64_t first, size_t n, void* bases_dev, void* dlogs_dev) try {
    GA_ABI_ENTRY();
    Ctx* c = reinterpret_cast<Ctx*>(h);
    Lock l(c);
    GA_DISPATCH_CURVE(curve, GA_DISPATCH_GROUP(group, GA_CHECK((util_gen_bases<C, G>(c, seed, n, bases_dev, dlogs_dev, first)))));
    GA_HIP_CHECK(hipStreamSynchronize(c->stream));
    return GA_OK;
} GA_ABI_CATCH

int ga_gen_scalars(ga_ctx* h, int curve, uint64_t seed, size_t n, void* scalars_dev) try {
    GA_ABI_ENTRY();
    Ctx* c = reinterpret_cast<Ctx*>(h);
    Lock l(c);
    GA_DISPATCH_CURVE(curve, GA_CHECK(util_gen_scalars<C>(c, seed, n, scalars_dev)));
    GA_HIP_CHECK(hipStreamSynchronize(c->stream));
    return GA_OK;
} GA_ABI_CATCH

int ga_fr_dot(ga_ctx* h, int curve, const void* a_dev, const void* b_dev, size_t n, void* out_host) try {
    GA_ABI_ENTRY();
    Ctx* c = reinterpret_cast<Ctx*>(h);
    Lock l(c);
    GA_DISPATCH_CURVE(curve, GA_CHECK(util_fr_dot<C>(c, a_dev, b_dev, n, out_host)));
    return GA_OK;
} GA_ABI_CATCH

int ga_fr_vec_mul(ga_ctx* h, int curve, const void* a, const void* b, size_t n, void* out, int on_device) try {
    GA_ABI_ENTRY();
    Ctx* c = reinterpret_cast<Ctx*>(h);
    if (!c || (n && (!a || !b || !out))) {
        set_error("ga_fr_vec_mul: null argument");
        return GA_ERR_INVALID;
    }
    Lock l(c);
    if (n == 0) return GA_OK;
    Staged sa{c}, sb{c};
    GA_CHECK(sa.stage(a, n * 32, on_device));
    GA_CHECK(sb.stage(b, n * 32, on_device));
    void* d_out = out;
    if (!on_device) GA_CHECK(c->scratch_get("fr_vec_out", n * 32, &d_out));
    GA_DISPATCH_CURVE(curve, GA_CHECK(util_fr_vec_mul<C>(c, sa.dev, sb.dev, n, d_out)));
    if (!on_device) GA_HIP_CHECK(hipMemcpyAsync(out, d_out, n * 32, hipMemcpyDeviceToHost, c->stream));
    GA_HIP_CHECK(hipStreamSynchronize(c->stream));
    return GA_OK;
} GA_ABI_CATCH

int ga_generator_mul(int curve, int group, const void* k, void* out_jac) try {
    GA_ABI_ENTRY();
    GA_DISPATCH_CURVE(curve, GA_DISPATCH_GROUP(group, {
                          typedef typename GroupField<C, G>::F F;
                          uint32_t kw[8];
                          memcpy(kw, k, 32);
                          host_store_jac<F>(out_jac, scalar_mul(to_xyzz(Generator<C, G>::get()), kw, 8));
                      }));
    return GA_OK;
} GA_ABI_CATCH

int ga_clock_probe(ga_ctx* h, uint32_t micros, double* mhz_out) try {
    GA_ABI_ENTRY();
    Ctx* c = reinterpret_cast<Ctx*>(h);
    if (!c || !mhz_out || micros == 0) {
        set_error("ga_clock_probe: bad argument");
        return GA_ERR_INVALID;
    }
    return util_clock_probe(c, micros, mhz_out);   // (no context lock: it is meant to run BESIDE whatever holds it)
} GA_ABI_CATCH

int ga_microbench(ga_ctx* h, char* buf, size_t cap) try {
    GA_ABI_ENTRY();
    Ctx* c = reinterpret_cast<Ctx*>(h);
    Lock l(c);
    return util_microbench(c, buf, cap);
} GA_ABI_CATCH

}  // extern "C"
