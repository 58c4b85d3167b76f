// Groth16 prover core: device-resident proving key + one-call proof from the solver's output.
//
// Mirrors backend/groth16/bn254/prove.go:130-315 (CPU) and backend/accelerated/icicle/groth16/bn254/icicle.go:784-1360
// (GPU):  computeH -> filter wire values -> 4 G1 MSMs + 1 G2 MSM -> host epilogue with the caller-supplied randomness
// (r, s); BSB22 commitments are the two extra MSMs per commitment of ga_g16_commit (prove.go:84,114) plus the K filter.  The host
// side is C++ because the reference's host side is compiled Go and no Go toolchain exists in the build image (INTEGRATION.md shows
// the cgo binding that calls this file's entry points).
#include <algorithm>
#include <condition_variable>
#include <memory>
#include <new>
#include <stdexcept>
#include <string>
#include <chrono>
#include <thread>

#include "hostops.hip.h"
#include "keyio.hip.h"

namespace ga {

// dst[idx[i]] = src[i] for elements of `chunks` 16-byte pieces (building the wire-indexed base arrays at pin time)
static __global__ void g16_scatter_points_kernel(u32x4* __restrict__ dst, const u32x4* __restrict__ src, const uint32_t* __restrict__ idx,
                                                 uint64_t n, uint32_t chunks) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t i = t / chunks, k = t % chunks;
    if (i >= n) return;
    dst[(uint64_t)idx[i] * chunks + k] = src[i * chunks + k];
}

struct G16Pk {
    Ctx* ctx = nullptr;
    int curve = 0;
    uint64_t n = 0;            // domain cardinality
    uint64_t nb_wires = 0;
    Domain* dom = nullptr;
    void *d_a = nullptr, *d_b = nullptr, *d_z = nullptr, *d_k = nullptr, *d_b2 = nullptr;
    uint64_t len_a = 0, len_b = 0, len_z = 0, len_k = 0, len_b2 = 0;
    uint32_t *d_idx_a = nullptr, *d_idx_b = nullptr;   // wire indices kept for the A / B MSMs (prove.go:147-168)
    uint32_t* d_idx_k = nullptr;   // wire indices feeding the K MSM when committed wires are left out (prove.go:231-235); null = W[nbPublic:]
    uint64_t len_k_remove = 0;
    std::vector<void*> d_ck_basis, d_ck_sigma;   // pinned pedersen keys (setup.go:260-287, icicle.go:231-261)
    std::vector<uint64_t> ck_len;
    // precomputed window-multiple tables (msm.hip.h): a vector with its tab_* flag set points to windows x len points and c_* is the
    // window width.  Per VECTOR since round 5: when the five tables do not fit the free HBM together (2^26 constraints: 288 GiB), the
    // ones that pay most per byte are built -- A, B (G1), K (they share one witness sort), then Z, then the twice as large G2.B --
    // and the rest stay plain affine arrays that run as un-pinned MSMs.
    bool tables = false;   // any of the five
    bool tab_a = false, tab_b = false, tab_z = false, tab_k = false, tab_b2 = false;
    int c_a = 0, c_b = 0, c_z = 0, c_k = 0;
    // Wire-indexed tables: when a base vector covers (almost) every wire, its table is laid out by WIRE id with (0,0) at the
    // wires it lacks (infinity entries are skipped by the bucket kernel), so that the digit extraction + radix sort of the whole
    // witness is done ONCE and shared by the A, B (G1 and G2) and K MSMs instead of once per filtered copy of the witness.
    bool share_a = false, share_b = false, share_k = false;   // (each implies its tab_* flag)
    bool share_b2 = false;                                    // G2.B wire-indexed too: it reuses the shared sort (share_b and tab_b2)
    int c_w = 0;
    // multi-GPU partition B: this key holds slice [off, off+len) of every base vector (ga_g16_key.shard_index/count)
    uint32_t shard_index = 0, shard_count = 1;
    uint64_t off_k = 0, off_z = 0, full_len_k = 0;
    // multi-GPU partition A (scalar windows, BASELINE config 4's wording): the WHOLE key is pinned on every device and this one
    // accumulates only share win_index of win_count of the Pippenger windows of every MSM; partial results add up
    uint32_t win_index = 0, win_count = 1;
    uint64_t w_lo = 0, w_hi = 0;   // wire range [w_lo, w_hi) the A and B gather lists (and a filtered K list) of this shard touch
    std::vector<uint8_t> alpha1, beta1, delta1, beta2, delta2;   // affine images (host)
    // fixed-base tables of delta1 / delta2 for the host epilogue: entry [w*15 + d-1] = d * 2^(4w) * delta (XYZZ images), built on first use
    std::once_flag delta_tab_once;
    std::vector<uint8_t> delta1_tab, delta2_tab;
    // In-flight users: every entry point that takes the key holds one PkUse for its whole duration (the host epilogue included,
    // which runs outside the device lock); ga_g16_pk_destroy waits for them, so a caller that frees the key from one thread while
    // another is still proving (Go: `defer pk.FreeGPUResources()` beside a second goroutine's Prove) gets a late free, not a
    // use-after-free.
    std::mutex use_mu;
    std::condition_variable use_cv;
    int users = 0;
    bool dying = false;
    // A key whose base vectors are still ON THEIR WAY (ga_g16_prove_oneshot: the key goes up as plain vectors, is used for ONE proof
    // and dropped -- the Go package's default, PinToGPU = false): an uploader thread copies them in the order the proof consumes
    // them (A, B, K, G2.B, Z) while the proof already runs; an MSM waits for ITS vector (await_vector), not for the key.
    struct Pending {
        std::mutex mu;
        std::condition_variable cv;
        bool done[GA_KEY_NB_VECTORS] = {false, false, false, false, false};
        size_t bytes[GA_KEY_NB_VECTORS] = {0, 0, 0, 0, 0};   // allocation sizes (the buffers go back to the context's spare set)
        // recorded on the uploader's stream behind a vector's copies: the consumer's STREAM waits for it (hipStreamWaitEvent), no host
        // thread does -- a hipStreamSynchronize of the upload stream was seen to return only when another thread's wait for a 54 ms
        // bucket kernel did (profiles/r06_h_oneshot_timeline.txt)
        hipEvent_t ev[GA_KEY_NB_VECTORS] = {nullptr, nullptr, nullptr, nullptr, nullptr};
        ~Pending() {
            for (hipEvent_t e : ev)
                if (e) hipEventDestroy(e);
        }
        int rc = GA_OK;
        std::string err;
        std::thread uploader;
    };
    std::unique_ptr<Pending> pending;
};

static void trace_event(const char* what, int arg, double extra_ms);
// the vector `which` of a key is on the device (always true for a key made by ga_g16_pk_create / the builder / a key file)
static int await_vector(G16Pk* pk, int which) {
    G16Pk::Pending* pd = pk->pending.get();
    if (!pd) return GA_OK;
    const auto t0 = std::chrono::steady_clock::now();
    std::unique_lock<std::mutex> g(pd->mu);
    pd->cv.wait(g, [&] { return pd->done[which] || pd->rc != GA_OK; });
    trace_event("MSM waited for vector", which, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    if (pd->rc != GA_OK) {
        set_error("%s", pd->err.c_str());
        return pd->rc;
    }
    g.unlock();
    GA_HIP_CHECK(hipStreamWaitEvent(pk->ctx->work_stream(), pd->ev[which], 0));   // the MSM about to be launched on this lane starts behind the copies
    return GA_OK;
}

struct PkUse {
    G16Pk* pk;
    bool ok = false;
    explicit PkUse(G16Pk* p) : pk(p) {
        if (!pk) return;
        std::lock_guard<std::mutex> g(pk->use_mu);
        if (pk->dying) return;
        pk->users++;
        ok = true;
    }
    ~PkUse() {
        if (!ok) return;
        std::lock_guard<std::mutex> g(pk->use_mu);
        if (--pk->users == 0) pk->use_cv.notify_all();
    }
    PkUse(const PkUse&) = delete;
    PkUse& operator=(const PkUse&) = delete;
};
#define GA_PK_USE(pk, what)                                                      \
    PkUse _pk_use(pk);                                                           \
    if (!_pk_use.ok) {                                                           \
        set_error(what ": the proving key is being destroyed");                  \
        return GA_ERR_STATE;                                                     \
    }

// GA_TRACE_PIN=1, process-wide clock: one line per event of a one-shot proof (uploader, waits) on stderr
static void trace_event(const char* what, int arg, double extra_ms = -1.0) {
    static const bool on = getenv("GA_TRACE_PIN") != nullptr;
    static const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    if (!on) return;
    const double t = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (extra_ms >= 0) fprintf(stderr, "[one-shot] %10.2f ms  %s %d (%.2f ms)\n", t, what, arg, extra_ms);
    else fprintf(stderr, "[one-shot] %10.2f ms  %s %d\n", t, what, arg);
}

// GA_TRACE_PIN=1: milestones of a key's way to the device on stderr (ms since the first mark of the calling thread) -- tools/exp
struct PinTrace {
    bool on;
    std::chrono::steady_clock::time_point t0;
    PinTrace() : on(getenv("GA_TRACE_PIN") != nullptr), t0(std::chrono::steady_clock::now()) {}
    void mark(const char* what) const {
        if (on) fprintf(stderr, "[pin] %8.2f ms  %s\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), what);
    }
};

static int upload(Ctx* ctx, const void* src, size_t bytes, void** dst) {
    *dst = nullptr;
    hipError_t e = device_malloc(dst, bytes ? bytes : 16);
    if (e != hipSuccess) {
        set_error("proving key upload: device_malloc(%zu) failed: %s", bytes, device_malloc_error(e));
        return GA_ERR_NOMEM;
    }
    if (bytes) GA_HIP_CHECK(hipMemcpyAsync(*dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return GA_OK;
}

static void pk_free(G16Pk* pk) {
    if (!pk) return;
    if (pk->pending && pk->pending->uploader.joinable()) pk->pending->uploader.join();   // (it writes into the buffers freed below)
    if (pk->pending && pk->ctx && !pk->tables) {   // a one-shot key: its plain vector buffers stay with the context for the next one
        const char* e = getenv("GA_DOMAIN_SPARE");
        Ctx* c = pk->ctx;
        std::lock_guard<std::mutex> g(c->spare_mu);
        if (!(e && atoi(e) == 0) && !c->spare_vectors.have) {
            void** slot[GA_KEY_NB_VECTORS] = {&pk->d_a, &pk->d_b, &pk->d_z, &pk->d_k, &pk->d_b2};
            for (int w = 0; w < GA_KEY_NB_VECTORS; w++) {
                c->spare_vectors.p[w] = *slot[w];
                c->spare_vectors.bytes[w] = pk->pending->bytes[w];
                *slot[w] = nullptr;
            }
            c->spare_vectors.have = true;
        }
    }
    if (pk->ctx)
        for (const void* t : {pk->d_a, pk->d_b, pk->d_z, pk->d_k, pk->d_b2}) pk->ctx->forget_table(t);
    hipFree(pk->d_a);
    hipFree(pk->d_b);
    hipFree(pk->d_z);
    hipFree(pk->d_k);
    hipFree(pk->d_b2);
    hipFree(pk->d_idx_a);
    hipFree(pk->d_idx_b);
    hipFree(pk->d_idx_k);
    for (void* p : pk->d_ck_basis) hipFree(p);
    for (void* p : pk->d_ck_sigma) hipFree(p);
    if (pk->dom) ntt_domain_give_spare(pk->ctx, pk->dom);   // (kept for the next key of this size: common.hip.h)
    delete pk;
}

// ---- staged key construction (ga_g16_builder_*) -------------------------------------------------------------------------------
// The proving key reaches the device vector by vector, chunk by chunk: every call takes ONE flat pointer to pointer-free memory
// and has copied what it needs when it returns.  This is the shape cgo wants (no Go pointer stored inside a C struct, nothing
// retained after the call) and the shape a streaming reader of the 6-9 GiB key files wants (keyio.hip: ReadDump / ReadFrom feed
// chunks from a pinned staging buffer).  ga_g16_pk_create(struct) is a thin wrapper over it.
struct G16Stage {
    Ctx* ctx = nullptr;
    int curve = 0;
    uint64_t n = 0, nb_wires = 0;
    uint32_t shard_index = 0, shard_count = 1;
    uint32_t win_index = 0, win_count = 1;
    struct Vec {
        void* d = nullptr;
        uint64_t total = 0, lo = 0, cnt = 0, seen = 0;
        bool reserved = false;
    } v[GA_KEY_NB_VECTORS];
    std::vector<uint8_t> inf[2];                 // InfinityA, InfinityB (Go []bool images)
    bool have_inf[2] = {false, false};
    // the wire ids each mask keeps, ascending (the gather lists of prove.go:147-168): built by a helper thread as soon as a mask is
    // set -- beside the uploads of the vectors, which keep the calling thread busy for 19 ms per GiB -- and joined by stage_finish
    struct WireList {
        std::unique_ptr<uint32_t[]> ids;
        uint64_t size = 0;
        std::thread job;
    } lists[2];
    void start_list(int which) {
        WireList& L = lists[which];
        if (L.job.joinable()) L.job.join();
        L.ids.reset(new uint32_t[inf[which].size() + 8]);   // (uninitialised on purpose: 64 MiB at 2^24 wires)
        L.size = 0;
        const uint8_t* m = inf[which].data();
        const uint64_t nw = inf[which].size();
        uint32_t* out = L.ids.get();
        uint64_t* size = &L.size;
        L.job = std::thread([m, nw, out, size]() {
            uint32_t* p = out;
            uint64_t i = 0;
            for (; i + 8 <= nw; i += 8) {   // masks are zero almost everywhere: eight wires per test
                uint64_t w;
                memcpy(&w, m + i, 8);
                if (w == 0) {
                    for (int k = 0; k < 8; k++) p[k] = (uint32_t)(i + k);
                    p += 8;
                } else {
                    for (int k = 0; k < 8; k++)
                        if (!m[i + k]) *p++ = (uint32_t)(i + k);
                }
            }
            for (; i < nw; i++)
                if (!m[i]) *p++ = (uint32_t)i;
            *size = (uint64_t)(p - out);
        });
    }
    std::vector<uint8_t> pts[GA_KEY_NB_POINTS];  // alpha1, beta1, delta1, beta2, delta2
    std::vector<void*> d_ck_basis, d_ck_sigma;
    std::vector<uint64_t> ck_len;
    std::vector<uint64_t> k_remove;
    std::thread* early_uploader = nullptr;       // one-shot keys: the thread already filling the vectors' buffers (pk_create_from_struct)
    ~G16Stage() {
        for (auto& L : lists)
            if (L.job.joinable()) L.job.join();
        for (auto& x : v) hipFree(x.d);
        for (void* p : d_ck_basis) hipFree(p);
        for (void* p : d_ck_sigma) hipFree(p);
    }
};

static size_t stage_point_bytes(int curve, int which) {
    const size_t fp = curve == GA_BN254 ? 32 : 48;
    return which == GA_KEY_G2_B ? 4 * fp : 2 * fp;
}

static int stage_reserve(G16Stage* st, int which, uint64_t total) {
    if (which < 0 || which >= GA_KEY_NB_VECTORS) {
        set_error("proving key: unknown vector id %d", which);
        return GA_ERR_INVALID;
    }
    G16Stage::Vec& x = st->v[which];
    if (x.reserved) {
        set_error("proving key: vector %d reserved twice", which);
        return GA_ERR_STATE;
    }
    const uint64_t base = total / st->shard_count, rem = total % st->shard_count, k = st->shard_index;   // same split as multigpu.shard_range
    x.total = total;
    x.lo = k * base + (k < rem ? k : rem);
    x.cnt = base + (k < rem ? 1 : 0);
    const size_t bytes = x.cnt * stage_point_bytes(st->curve, which);
    hipError_t e = device_malloc(&x.d, bytes ? bytes : 16);
    if (e != hipSuccess) {
        set_error("proving key upload: device_malloc(%zu) failed: %s", bytes, device_malloc_error(e));
        return GA_ERR_NOMEM;
    }
    x.reserved = true;
    return GA_OK;
}

// points [seen, seen + count) of the full vector; only the part inside this shard's range is copied.  `pinned`: the source is
// page-locked memory owned by the caller for the duration of the call (keyio's staging buffers) -- the copy is then truly
// asynchronous and the caller synchronises; otherwise the stream is drained before returning.
static int stage_append(G16Stage* st, int which, const void* points, uint64_t count, bool pinned = false) {
    if (which < 0 || which >= GA_KEY_NB_VECTORS || !st->v[which].reserved) {
        set_error("proving key: append to vector %d before ga_g16_builder_reserve", which);
        return GA_ERR_STATE;
    }
    G16Stage::Vec& x = st->v[which];
    if (x.seen + count > x.total) {
        set_error("proving key: vector %d overflows its reserved length %llu", which, (unsigned long long)x.total);
        return GA_ERR_INVALID;
    }
    const size_t psz = stage_point_bytes(st->curve, which);
    const uint64_t b0 = x.seen > x.lo ? x.seen : x.lo;
    const uint64_t e0 = x.seen + count < x.lo + x.cnt ? x.seen + count : x.lo + x.cnt;
    if (e0 > b0) {
        GA_HIP_CHECK(hipMemcpyAsync((char*)x.d + (b0 - x.lo) * psz, (const char*)points + (b0 - x.seen) * psz, (e0 - b0) * psz,
                                    hipMemcpyHostToDevice, st->ctx->stream));
        if (!pinned) GA_HIP_CHECK(hipStreamSynchronize(st->ctx->stream));   // no host pointer survives this call
    }
    x.seen += count;
    return GA_OK;
}

template <class C>
static int stage_finish(G16Stage* st, int precompute, G16Pk** out) {
    typedef Fe<typename C::FpP> F1;
    typedef Fe2<typename C::FpP> F2;
    Ctx* ctx = st->ctx;
    const size_t s1 = sizeof(Affine<F1>), s2 = sizeof(Affine<F2>);
    for (int w = 0; w < GA_KEY_NB_VECTORS; w++)
        if (!st->v[w].reserved || st->v[w].seen != st->v[w].total) {
            set_error("proving key: vector %d incomplete (%llu of %llu points)", w, (unsigned long long)st->v[w].seen,
                      (unsigned long long)st->v[w].total);
            return GA_ERR_STATE;
        }
    for (int q = 0; q < GA_KEY_NB_POINTS; q++)
        if (st->pts[q].empty()) {
            set_error("proving key: point %d (alpha1, beta1, delta1, beta2, delta2) not set", q);
            return GA_ERR_STATE;
        }
    if (!st->have_inf[0] || !st->have_inf[1]) {
        set_error("proving key: InfinityA / InfinityB not set");
        return GA_ERR_STATE;
    }
    const uint64_t len_a = st->v[GA_KEY_G1_A].total, len_b = st->v[GA_KEY_G1_B].total, len_z = st->v[GA_KEY_G1_Z].total,
                   len_k = st->v[GA_KEY_G1_K].total, len_b2 = st->v[GA_KEY_G2_B].total;
    if (len_z + 1 != st->n) {
        set_error("proving key: len(G1.Z)=%llu but domain cardinality is %llu (expected n-1, setup.go:248-249)",
                  (unsigned long long)len_z, (unsigned long long)st->n);
        return GA_ERR_INVALID;
    }
    if (st->nb_wires >= (1ull << 32)) {
        set_error("proving key: %llu wires exceed the 32-bit wire index space", (unsigned long long)st->nb_wires);
        return GA_ERR_INVALID;
    }
    if (len_a > st->nb_wires || len_b > st->nb_wires || len_k > st->nb_wires || st->k_remove.size() > st->nb_wires ||
        len_k + st->k_remove.size() > st->nb_wires) {   // (nbWires - len(K) - len(k_remove) = nbPublic >= 0; the wire-indexed layouts rely on it)
        set_error("proving key: len(A) %llu, len(B) %llu, len(K) %llu + %zu removed wires do not fit %llu wires", (unsigned long long)len_a,
                  (unsigned long long)len_b, (unsigned long long)len_k, st->k_remove.size(), (unsigned long long)st->nb_wires);
        return GA_ERR_INVALID;
    }
    if (st->inf[0].size() != st->nb_wires || st->inf[1].size() != st->nb_wires) {
        set_error("proving key: InfinityA / InfinityB must have one entry per wire");
        return GA_ERR_INVALID;
    }
    if (st->win_count > 1 && (st->shard_count > 1 || st->win_index >= st->win_count)) {
        set_error("proving key: window sharding (%u of %u) cannot be combined with base-range sharding, and the index must be below the count",
                  st->win_index, st->win_count);
        return GA_ERR_INVALID;
    }
    PinTrace tr;
    for (int k = 0; k < 2; k++) {
        if (!st->lists[k].job.joinable()) st->start_list(k);   // (a caller that set the mask through a path without the early start)
        st->lists[k].job.join();
    }
    const uint32_t *ia = st->lists[0].ids.get(), *ib = st->lists[1].ids.get();
    if (st->lists[0].size != len_a || st->lists[1].size != len_b || len_b2 != len_b) {
        set_error("proving key: InfinityA/B masks disagree with len(A)/len(B), or len(G2.B) != len(G1.B)");
        return GA_ERR_INVALID;
    }
    G16Pk* pk = new G16Pk();
    pk->ctx = ctx;
    pk->curve = C::ID;
    pk->n = st->n;
    pk->nb_wires = st->nb_wires;
    pk->shard_count = st->shard_count;
    pk->shard_index = st->shard_index;
    pk->win_count = st->win_count ? st->win_count : 1;
    pk->win_index = st->win_index;
    auto take = [&](int which, void** slot, uint64_t* len) {   // the device buffer changes owner
        *slot = st->v[which].d;
        *len = st->v[which].cnt;
        st->v[which].d = nullptr;
    };
    take(GA_KEY_G1_A, &pk->d_a, &pk->len_a);
    take(GA_KEY_G1_B, &pk->d_b, &pk->len_b);
    take(GA_KEY_G1_Z, &pk->d_z, &pk->len_z);
    take(GA_KEY_G1_K, &pk->d_k, &pk->len_k);
    take(GA_KEY_G2_B, &pk->d_b2, &pk->len_b2);
    const uint64_t lo_a = st->v[GA_KEY_G1_A].lo, lo_b = st->v[GA_KEY_G1_B].lo, lo_k = st->v[GA_KEY_G1_K].lo;
    pk->off_k = lo_k;
    pk->off_z = st->v[GA_KEY_G1_Z].lo;
    pk->full_len_k = len_k;
    {   // wire range of this shard: the sorted gather lists are sliced contiguously, so min/max are the slice ends
        uint64_t lo = st->nb_wires, hi = 0;
        auto span = [&](const uint32_t* v, uint64_t off, uint64_t cnt) {
            if (cnt == 0) return;
            lo = lo < v[off] ? lo : v[off];
            hi = hi > (uint64_t)v[off + cnt - 1] + 1 ? hi : (uint64_t)v[off + cnt - 1] + 1;
        };
        span(ia, lo_a, pk->len_a);
        span(ib, lo_b, pk->len_b);
        pk->w_lo = st->shard_count == 1 ? 0 : lo;
        pk->w_hi = st->shard_count == 1 ? st->nb_wires : hi;
    }
    tr.mark("finish: gather lists built on the host");
    int rc = GA_OK;
    pk->dom = ntt_domain_take_spare(ctx, C::ID, pk->n);   // the domain of the key this context freed last, when it has this size
    if (!pk->dom) rc = ntt_domain_new<C>(ctx, pk->n, &pk->dom);
    tr.mark("finish: ntt_domain_new returned");
    if (rc == GA_OK) rc = upload(ctx, ia + lo_a, pk->len_a * 4, (void**)&pk->d_idx_a);
    if (rc == GA_OK) rc = upload(ctx, ib + lo_b, pk->len_b * 4, (void**)&pk->d_idx_b);
    // K filter with commitments: wireValues[nbPublic:] minus the private committed and commitment wires (prove.go:231-235)
    std::vector<uint32_t> ik;
    pk->len_k_remove = st->k_remove.size();
    if (rc == GA_OK && pk->len_k_remove) {
        const uint64_t nrem = st->k_remove.size();
        if (len_k + nrem > st->nb_wires) {
            set_error("proving key: len(K)+len(k_remove) > nbWires");
            rc = GA_ERR_INVALID;
        } else {
            const uint64_t nb_public = st->nb_wires - len_k - nrem;
            ik.reserve(len_k);
            uint64_t j = 0;
            bool ok = true;
            for (uint64_t i = 0; i < nrem; i++)
                ok = ok && st->k_remove[i] >= nb_public && st->k_remove[i] < st->nb_wires && (i == 0 || st->k_remove[i] > st->k_remove[i - 1]);
            for (uint64_t i = nb_public; ok && i < st->nb_wires; i++) {
                if (j < nrem && st->k_remove[j] == i) j++;
                else ik.push_back((uint32_t)i);
            }
            if (!ok || ik.size() != len_k) {
                set_error("proving key: k_remove must be strictly increasing wire ids in [nbPublic, nbWires)");
                rc = GA_ERR_INVALID;
            }
        }
        if (rc == GA_OK) rc = upload(ctx, ik.data() + lo_k, pk->len_k * 4, (void**)&pk->d_idx_k);
        if (rc == GA_OK && pk->len_k && st->shard_count > 1) {
            pk->w_lo = pk->w_lo < ik[lo_k] ? pk->w_lo : ik[lo_k];
            pk->w_hi = pk->w_hi > (uint64_t)ik[lo_k + pk->len_k - 1] + 1 ? pk->w_hi : (uint64_t)ik[lo_k + pk->len_k - 1] + 1;
        }
    }
    pk->d_ck_basis.swap(st->d_ck_basis);
    pk->d_ck_sigma.swap(st->d_ck_sigma);
    pk->ck_len = st->ck_len;
    if (rc == GA_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) {   // no host pointer survives this call
        set_error("proving key upload: stream synchronize failed");
        rc = GA_ERR_HIP;
    }
    tr.mark("finish: gather lists uploaded, stream drained");
    // ---- optional precomputation: [2^(c*w)]P for every window (one shared bucket set per MSM afterwards) ----------
    if (rc == GA_OK && precompute >= 0) {
        {   // a key that is here to stay: the buffers kept for one-shot keys (6-9 GiB) go back to the device before the tables are sized
            std::lock_guard<std::mutex> g(ctx->spare_mu);
            for (void*& q : ctx->spare_vectors.p) {
                hipFree(q);
                q = nullptr;
            }
            ctx->spare_vectors.have = false;
        }
        int nw = 0;
        const size_t t1 = msm_table_point_bytes<C, GA_G1>(), t2 = msm_table_point_bytes<C, GA_G2>();
        // share the witness sort between the vectors that cover at least GA_G16_SHARE_MIN_PCT % of the wires (default 90: a
        // sparse vector would make the lanes of the bucket kernel idle on its missing wires, and waste table memory)
        const int share_pct = ctx->tun.g16_share_min_pct;
        auto dense = [&](uint64_t len) {
            return pk->shard_count == 1 && pk->nb_wires < (1ull << 27) && len > 0 && (double)len * 100.0 >= (double)pk->nb_wires * share_pct;
        };
        pk->share_a = dense(pk->len_a);
        pk->share_b = dense(pk->len_b);
        pk->share_k = dense(pk->len_k);
        bool plan_ok = msm_plan_table<C>(pk->nb_wires, &pk->c_w, &nw) == GA_OK;
        const uint64_t wide = (uint64_t)nw * pk->nb_wires;
        plan_ok = msm_plan_table<C>(pk->len_a, &pk->c_a, &nw) == GA_OK && plan_ok;
        plan_ok = msm_plan_table<C>(pk->len_b, &pk->c_b, &nw) == GA_OK && plan_ok;
        plan_ok = msm_plan_table<C>(pk->len_z, &pk->c_z, &nw) == GA_OK && plan_ok;
        plan_ok = msm_plan_table<C>(pk->len_k, &pk->c_k, &nw) == GA_OK && plan_ok;
        if (!plan_ok && precompute > 0) rc = GA_ERR_INVALID;   // vectors beyond the table index space: the caller asked for tables explicitly
        size_t free_b = 0, total_b = 0;
        hipMemGetInfo(&free_b, &total_b);
        // Which vectors get a table.  precompute > 0: all five (the caller insists; a table that does not fit fails the call).
        // precompute == 0: as many as fit the free HBM next to the per-proof scratch, in the order of what a table buys per byte:
        // A, B1, K (48 GiB each at 2^26 BN254; with all three the witness is sorted once instead of three times), Z, and last G2.B
        // (twice the bytes for the smallest relative gain).
        // (a vector the planner refused -- beyond the 2^31 pair space, or a forced GA_TABLE_C too narrow -- has c_* = 0: no table, no size)
        auto nwin_of = [](int cbits) -> uint64_t { return cbits > 0 ? (uint64_t)(C::FrP::BITS / cbits + 1) : 0; };
        const uint64_t bytes_a = !plan_ok ? 0 : pk->share_a ? wide * t1 : nwin_of(pk->c_a) * pk->len_a * t1;
        const uint64_t bytes_b = !plan_ok ? 0 : pk->share_b ? wide * t1 : nwin_of(pk->c_b) * pk->len_b * t1;
        const uint64_t bytes_b2 = !plan_ok ? 0 : pk->share_b ? wide * t2 : nwin_of(pk->c_b) * pk->len_b * t2;
        const uint64_t bytes_z = !plan_ok ? 0 : nwin_of(pk->c_z) * pk->len_z * t1;
        const uint64_t bytes_k = !plan_ok ? 0 : pk->share_k ? wide * t1 : nwin_of(pk->c_k) * pk->len_k * t1;
        if (rc == GA_OK && plan_ok) {
            if (precompute > 0) {
                pk->tab_a = pk->tab_b = pk->tab_k = pk->tab_z = pk->tab_b2 = true;
            } else {
                // What the tables may take = free HBM - what a single caller's proof will allocate on this context (measured: 1.15 KB per
                // constraint at 2^26 with three tables -- sort pairs, task lists, hat-domain copies of the plain vectors, input slots, NTT
                // tables --, 1.25 KB allowed, + 10 %) - 4 GiB.  At 2^26 BN254 on an empty device: 244 - 88 - 4 = 152 GiB -> A, B, K (144).
                // Scratch this context already holds (an earlier proof of this size) is credited against the allowance: it is not free
                // any more, but it is exactly what the allowance was for.
                uint64_t held = 0;
                {
                    std::lock_guard<std::mutex> g(ctx->scratch_mu);
                    for (const auto& kv : ctx->scratch) held += kv.second.second;
                }
                const double per_proof = (double)pk->n * 1280.0;
                const double allowance = per_proof > (double)held ? per_proof - (double)held : 0.0;
                double budget = (double)free_b - 1.1 * allowance - 4.0 * 1073741824.0;
                if (const uint64_t pct = ctx->tun.g16_table_budget_pct)   // GA_G16_TABLE_BUDGET_PCT (tests: partial tables on small keys)
                    budget = (double)pct / 100.0 * (double)(bytes_a + bytes_b + bytes_k + bytes_z + bytes_b2);
                struct Cand { bool* flag; uint64_t bytes; } order[5] = {{&pk->tab_a, bytes_a}, {&pk->tab_b, bytes_b}, {&pk->tab_k, bytes_k},
                                                                       {&pk->tab_z, bytes_z}, {&pk->tab_b2, bytes_b2}};
                for (auto& cnd : order) {
                    // (the plain array it replaces is freed once the table stands; while it is built both are resident)
                    if ((double)cnd.bytes <= budget) {
                        *cnd.flag = true;
                        budget -= (double)cnd.bytes;
                    }
                }
            }
        }
        pk->share_a = pk->share_a && pk->tab_a;
        pk->share_b = pk->share_b && pk->tab_b;
        pk->share_k = pk->share_k && pk->tab_k;
        pk->share_b2 = pk->share_b && pk->tab_b2;
        pk->tables = pk->tab_a || pk->tab_b || pk->tab_k || pk->tab_z || pk->tab_b2;
        if (rc == GA_OK && pk->tables) {
            auto make = [&](void** slot, uint64_t len, int c, size_t psz, auto build) -> int {
                if (len == 0) return GA_OK;
                const int nwin = C::FrP::BITS / c + 1;
                void* t = nullptr;
                if (device_malloc(&t, (uint64_t)nwin * len * psz) != hipSuccess) {
                    set_error("proving key: hipMalloc of a %llu-byte window table failed", (unsigned long long)((uint64_t)nwin * len * psz));
                    return GA_ERR_NOMEM;
                }
                int r = build(*slot, len, c, t);
                if (r == GA_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) r = GA_ERR_HIP;
                hipFree(*slot);
                *slot = t;
                return r;
            };
            auto b1 = [&](const void* src, uint64_t len, int c, void* t) { return msm_table_build<C, GA_G1>(ctx, src, len, c, t); };
            auto b2 = [&](const void* src, uint64_t len, int c, void* t) { return msm_table_build<C, GA_G2>(ctx, src, len, c, t); };
            // compact base array -> wire-indexed array with (0,0) at the missing wires
            auto widen = [&](void** slot, uint64_t len, const uint32_t* d_idx, size_t psz) -> int {
                void* wide_arr = nullptr;
                if (device_malloc(&wide_arr, pk->nb_wires * psz) != hipSuccess) {
                    set_error("proving key: hipMalloc of a wire-indexed base array failed");
                    return GA_ERR_NOMEM;
                }
                hipError_t we = hipMemsetAsync(wide_arr, 0, pk->nb_wires * psz, ctx->stream);
                const uint32_t chunks = (uint32_t)(psz / 16);
                const uint64_t threads = len * chunks;
                if (we == hipSuccess) {
                    hipLaunchKernelGGL(g16_scatter_points_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, ctx->stream,
                                       (u32x4*)wide_arr, (const u32x4*)*slot, d_idx, len, chunks);
                    we = hipGetLastError();
                }
                if (we == hipSuccess) we = hipStreamSynchronize(ctx->stream);
                if (we != hipSuccess) {
                    set_error("proving key: building a wire-indexed base array failed: %s", hipGetErrorString(we));
                    hipFree(wide_arr);
                    return GA_ERR_HIP;
                }
                hipFree(*slot);
                *slot = wide_arr;
                return GA_OK;
            };
            uint32_t* d_ik = pk->d_idx_k;   // wire ids of K's entries: the remove-list gather, or nbPublic + i
            if (pk->share_k && !d_ik) {
                const uint64_t nbp = pk->nb_wires - pk->len_k;
                std::vector<uint32_t> ikk(pk->len_k);
                for (uint64_t i = 0; i < pk->len_k; i++) ikk[i] = (uint32_t)(nbp + i);
                rc = upload(ctx, ikk.data(), pk->len_k * 4, (void**)&d_ik);
                if (rc == GA_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = GA_ERR_HIP;
            }
            if (rc == GA_OK && pk->share_a) rc = widen(&pk->d_a, pk->len_a, pk->d_idx_a, s1);
            if (rc == GA_OK && pk->share_b) rc = widen(&pk->d_b, pk->len_b, pk->d_idx_b, s1);
            if (rc == GA_OK && pk->share_b2) rc = widen(&pk->d_b2, pk->len_b2, pk->d_idx_b, s2);
            if (rc == GA_OK && pk->share_k) rc = widen(&pk->d_k, pk->len_k, d_ik, s1);
            if (d_ik && d_ik != pk->d_idx_k) hipFree(d_ik);
            const uint64_t nwr = pk->nb_wires;
            if (rc == GA_OK && pk->tab_a) rc = pk->share_a ? make(&pk->d_a, nwr, pk->c_w, t1, b1) : make(&pk->d_a, pk->len_a, pk->c_a, t1, b1);
            if (rc == GA_OK && pk->tab_b) rc = pk->share_b ? make(&pk->d_b, nwr, pk->c_w, t1, b1) : make(&pk->d_b, pk->len_b, pk->c_b, t1, b1);
            if (rc == GA_OK && pk->tab_z) rc = make(&pk->d_z, pk->len_z, pk->c_z, t1, b1);
            if (rc == GA_OK && pk->tab_k) rc = pk->share_k ? make(&pk->d_k, nwr, pk->c_w, t1, b1) : make(&pk->d_k, pk->len_k, pk->c_k, t1, b1);
            if (rc == GA_OK && pk->tab_b2) rc = pk->share_b2 ? make(&pk->d_b2, nwr, pk->c_w, t2, b2) : make(&pk->d_b2, pk->len_b2, pk->c_b, t2, b2);
            if (rc != GA_OK) pk->tables = false;
        }
    }
    if (!pk->tables) {
        pk->share_a = pk->share_b = pk->share_k = pk->share_b2 = false;
        pk->tab_a = pk->tab_b = pk->tab_k = pk->tab_z = pk->tab_b2 = false;
    }
    if (rc != GA_OK) {
        if (st->early_uploader && st->early_uploader->joinable()) st->early_uploader->join();   // (it writes into the buffers pk_free frees)
        pk_free(pk);
        return rc;
    }
    pk->alpha1 = st->pts[GA_KEY_G1_ALPHA];
    pk->beta1 = st->pts[GA_KEY_G1_BETA];
    pk->delta1 = st->pts[GA_KEY_G1_DELTA];
    pk->beta2 = st->pts[GA_KEY_G2_BETA];
    pk->delta2 = st->pts[GA_KEY_G2_DELTA];
    (void)s2;
    *out = pk;
    return GA_OK;
}

static int stage_set_point(G16Stage* st, int which, const void* affine) {
    if (which < 0 || which >= GA_KEY_NB_POINTS || !affine) {
        set_error("proving key: bad point id %d or null pointer", which);
        return GA_ERR_INVALID;
    }
    const size_t fp = st->curve == GA_BN254 ? 32 : 48;
    const size_t bytes = (which == GA_KEY_G2_BETA || which == GA_KEY_G2_DELTA) ? 4 * fp : 2 * fp;
    st->pts[which].assign((const uint8_t*)affine, (const uint8_t*)affine + bytes);
    return GA_OK;
}

static int stage_add_commitment_key(G16Stage* st, const void* basis, const void* sigma, uint64_t len) {
    if (len && (!basis || !sigma)) {
        set_error("proving key: null commitment key basis");
        return GA_ERR_INVALID;
    }
    const size_t s1 = stage_point_bytes(st->curve, GA_KEY_G1_A);
    void *db = nullptr, *ds = nullptr;
    int rc = upload(st->ctx, basis, len * s1, &db);
    if (rc == GA_OK) rc = upload(st->ctx, sigma, len * s1, &ds);
    if (rc == GA_OK && hipStreamSynchronize(st->ctx->stream) != hipSuccess) rc = GA_ERR_HIP;
    if (rc != GA_OK) {
        hipFree(db);
        hipFree(ds);
        return rc;
    }
    st->d_ck_basis.push_back(db);
    st->d_ck_sigma.push_back(ds);
    st->ck_len.push_back(len);
    return GA_OK;
}

// ga_g16_pk_create: the struct-of-pointers form of the same thing (C and ctypes callers; from Go only with runtime.Pinner)
// defer_uploads (ga_g16_prove_oneshot): the five base vectors get their device buffers here but are copied by an uploader thread
// that this function starts before it returns -- plain vectors only (no tables), the whole key on one device; the caller must keep
// the host vectors alive until the thread has been joined (pk_free does).
static int pk_create_from_struct(Ctx* ctx, const ga_g16_key* key, G16Pk** out, bool defer_uploads = false) {
    if (!key->g1_alpha || !key->g1_beta || !key->g1_delta || !key->g2_beta || !key->g2_delta || !key->infinity_a || !key->infinity_b ||
        (key->len_a && !key->g1_a) || (key->len_b && !key->g1_b) || (key->len_z && !key->g1_z) || (key->len_k && !key->g1_k) ||
        (key->len_b2 && !key->g2_b)) {
        set_error("ga_g16_pk_create: null pointer inside ga_g16_key");
        return GA_ERR_INVALID;
    }
    if (key->len_a + key->nb_infinity_a != key->nb_wires || key->len_b + key->nb_infinity_b != key->nb_wires) {
        set_error("proving key: len(A)+NbInfinityA, len(B)+NbInfinityB must equal nbWires and len(G2.B)==len(G1.B)");
        return GA_ERR_INVALID;
    }
    if (key->nb_commitments && (!key->ck_basis || !key->ck_basis_exp_sigma || !key->ck_len)) {
        set_error("proving key: nb_commitments > 0 but the commitment key arrays are null");
        return GA_ERR_INVALID;
    }
    if (key->len_k_remove && !key->k_remove) {
        set_error("proving key: k_remove missing");
        return GA_ERR_INVALID;
    }
    PinTrace tr;
    G16Stage st;
    st.ctx = ctx;
    st.curve = key->curve;
    st.n = key->domain_cardinality;
    st.nb_wires = key->nb_wires;
    st.shard_count = key->shard_count ? key->shard_count : 1;
    st.shard_index = key->shard_index;
    if (st.shard_index >= st.shard_count) {
        set_error("proving key: shard_index %u >= shard_count %u", st.shard_index, st.shard_count);
        return GA_ERR_INVALID;
    }
    st.win_count = key->window_shard_count ? key->window_shard_count : 1;
    st.win_index = key->window_shard_index;
    // the masks first: their gather lists are built by helper threads while this thread is busy with the 6-9 GiB of uploads below
    st.inf[0].assign(key->infinity_a, key->infinity_a + key->nb_wires);
    st.inf[1].assign(key->infinity_b, key->infinity_b + key->nb_wires);
    st.have_inf[0] = st.have_inf[1] = true;
    st.start_list(0);
    st.start_list(1);
    const void* vec[GA_KEY_NB_VECTORS] = {key->g1_a, key->g1_b, key->g1_z, key->g1_k, key->g2_b};
    const uint64_t len[GA_KEY_NB_VECTORS] = {key->len_a, key->len_b, key->len_z, key->len_k, key->len_b2};
    if (defer_uploads && (st.shard_count != 1 || st.win_count != 1)) {
        set_error("ga_g16_prove_oneshot: the key must be whole (no base-range or window sharding)");
        return GA_ERR_INVALID;
    }
    size_t alloc_bytes[GA_KEY_NB_VECTORS];
    for (int w = 0; w < GA_KEY_NB_VECTORS; w++) alloc_bytes[w] = len[w] ? (size_t)len[w] * stage_point_bytes(key->curve, w) : 16;
    if (defer_uploads) {   // the buffers of the previous one-shot key, when they have the sizes this one needs
        std::lock_guard<std::mutex> g(ctx->spare_mu);
        Ctx::SpareVectors& sp = ctx->spare_vectors;
        if (sp.have) {
            bool fits = true;
            for (int w = 0; w < GA_KEY_NB_VECTORS; w++) fits = fits && sp.bytes[w] == alloc_bytes[w];
            for (int w = 0; w < GA_KEY_NB_VECTORS; w++) {
                if (fits) {
                    G16Stage::Vec& x = st.v[w];
                    x.d = sp.p[w];
                    x.total = x.cnt = len[w];
                    x.lo = 0;
                    x.reserved = true;
                } else {
                    hipFree(sp.p[w]);
                }
                sp.p[w] = nullptr;
            }
            sp.have = false;
        }
    }
    for (int w = 0; w < GA_KEY_NB_VECTORS; w++) {
        if (!st.v[w].reserved) GA_CHECK(stage_reserve(&st, w, len[w]));
        if (defer_uploads) st.v[w].seen = st.v[w].total;   // (the uploader below fills the buffer)
        else GA_CHECK(stage_append(&st, w, vec[w], len[w], /*pinned=*/true));   // one drain below instead of five
        tr.mark("vector reserved + appended");
    }
    GA_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    tr.mark("uploads drained");
    // one-shot: the uploader starts NOW, on the buffers just reserved -- the masks, gather lists and domain below (30-40 ms at 2^24)
    // are built while the first vector is already on its way
    std::unique_ptr<G16Pk::Pending> pending;
    struct JoinOnError {   // an error return below must not free the buffers (G16Stage's destructor) under a running uploader
        G16Pk::Pending* pd = nullptr;
        ~JoinOnError() {
            if (pd && pd->uploader.joinable()) pd->uploader.join();
        }
    } join_on_error;
    if (defer_uploads) {
        pending.reset(new G16Pk::Pending());
        G16Pk::Pending* pd = pending.get();
        join_on_error.pd = pd;
        st.early_uploader = &pd->uploader;
        for (int w = 0; w < GA_KEY_NB_VECTORS; w++) {
            pd->bytes[w] = alloc_bytes[w];
            GA_HIP_CHECK(hipEventCreateWithFlags(&pd->ev[w], hipEventDisableTiming));
        }
        // in the order the proof consumes them: A, B, K (G1), B (G2) on the witness lane, Z last (it waits for h anyway)
        struct Job { int which; void* dst; const void* src; size_t bytes; int prio; };
        std::vector<Job> jobs;
        void* const dst[GA_KEY_NB_VECTORS] = {st.v[GA_KEY_G1_A].d, st.v[GA_KEY_G1_B].d, st.v[GA_KEY_G1_Z].d, st.v[GA_KEY_G1_K].d, st.v[GA_KEY_G2_B].d};
        int order = 0;
        for (int w : {GA_KEY_G1_A, GA_KEY_G1_B, GA_KEY_G1_K, GA_KEY_G2_B, GA_KEY_G1_Z}) {
            // turn priorities (common.hip.h): W 0 | A 1, B 2 | the solver's A, B, C 3 | K 4, G2.B 5, Z 6.  (K before G2.B, and its MSM
            // before G2.B's in witness_msms: copies make little progress while the G2 bucket kernel runs -- 2 GiB in 72 ms where
            // they take 38 -- so as much as possible is on the device before that kernel starts)
            static const int prio[5] = {1, 2, 4, 5, 6};
            jobs.push_back(Job{w, dst[w], vec[w], (size_t)len[w] * stage_point_bytes(key->curve, w), prio[order++]});
        }
        const int device = ctx->device;
        pd->uploader = std::thread([pd, jobs, device, ctx]() {
            int rc = GA_OK;
            std::string err;
            hipStream_t up = nullptr;
            try {
                if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&up, hipStreamNonBlocking) != hipSuccess) {
                    rc = GA_ERR_HIP;
                    err = "one-shot key upload: no stream";
                }
                for (const Job& j : jobs) {
                    if (rc != GA_OK) break;
                    hipError_t e = j.bytes ? ctx->h2d_pageable(j.dst, j.src, j.bytes, up, j.prio) : hipSuccess;
                    if (e == hipSuccess) e = hipEventRecord(pd->ev[j.which], up);
                    std::lock_guard<std::mutex> g(pd->mu);
                    if (e != hipSuccess) {
                        rc = GA_ERR_HIP;
                        err = std::string("one-shot key upload: ") + hipGetErrorString(e);
                    } else {
                        pd->done[j.which] = true;
                        pd->cv.notify_all();
                        trace_event("uploaded vector", j.which);
                    }
                }
            } catch (...) {   // (an exception leaving a thread function terminates the process)
                rc = GA_ERR_STATE;
                err = "one-shot key upload: exception in the uploader thread";
            }
            if (up) {   // the host vectors may be released once this thread has been joined: every copy must have left them
                if (hipStreamSynchronize(up) != hipSuccess && rc == GA_OK) {
                    rc = GA_ERR_HIP;
                    err = "one-shot key upload: stream synchronize failed";
                }
                hipStreamDestroy(up);
            }
            if (rc != GA_OK) {
                std::lock_guard<std::mutex> g(pd->mu);
                pd->rc = rc;
                pd->err = err;
                pd->cv.notify_all();
            }
        });
    }

    const void* pt[GA_KEY_NB_POINTS] = {key->g1_alpha, key->g1_beta, key->g1_delta, key->g2_beta, key->g2_delta};
    for (int q = 0; q < GA_KEY_NB_POINTS; q++) GA_CHECK(stage_set_point(&st, q, pt[q]));
    for (uint32_t i = 0; i < key->nb_commitments; i++) GA_CHECK(stage_add_commitment_key(&st, key->ck_basis[i], key->ck_basis_exp_sigma[i], key->ck_len[i]));
    if (key->len_k_remove) st.k_remove.assign(key->k_remove, key->k_remove + key->len_k_remove);
    tr.mark("points, infinity masks, commitment keys staged");
    G16Pk* pk = nullptr;
    GA_DISPATCH_CURVE(key->curve, GA_CHECK(stage_finish<C>(&st, defer_uploads ? -1 : key->precompute, &pk)));
    tr.mark("stage_finish");
    if (defer_uploads) {
        pk->pending = std::move(pending);   // (the key now owns the uploader: pk_free joins it)
        join_on_error.pd = nullptr;
    }
    *out = pk;
    return GA_OK;
}


// ---- key files -> staged key (keyio.hip.h has the formats) --------------------------------------------------------------------------
// `count` encoded points of group G from `src`: decoded on the device, the part inside [keep_lo, keep_lo + keep_cnt) lands at d_dst
template <class C, int G>
static int decode_stream(Ctx* ctx, Staging& sg, ByteSource& src, uint64_t count, bool compressed, void* d_dst, uint64_t keep_lo,
                         uint64_t keep_cnt) {
    typedef typename GroupField<C, G>::F F;
    const size_t enc = compressed ? sizeof(F) : 2 * sizeof(F), psz = sizeof(Affine<F>);
    const uint64_t per_chunk = Staging::BYTES / enc;
    GA_HIP_CHECK(hipMemsetAsync(sg.d_bad, 0, 4, ctx->stream));
    int k = 0;
    for (uint64_t done = 0; done < count; k ^= 1) {
        const uint64_t cn = count - done < per_chunk ? count - done : per_chunk;
        GA_HIP_CHECK(hipEventSynchronize(sg.ev[k]));   // the previous copy out of this staging buffer has finished
        GA_CHECK(src.read(sg.h[k], cn * enc));
        GA_HIP_CHECK(hipMemcpyAsync(sg.d_bytes, sg.h[k], cn * enc, hipMemcpyHostToDevice, ctx->stream));
        GA_HIP_CHECK(hipEventRecord(sg.ev[k], ctx->stream));
        hipLaunchKernelGGL((key_decode_kernel<C, G>), dim3((unsigned)((cn + 63) / 64)), dim3(64), 0, ctx->stream, (const uint8_t*)sg.d_bytes, cn,
                           compressed ? 1 : 0, sg.d_points, sg.d_bad);
        GA_KERNEL_CHECK();
        const uint64_t b0 = done > keep_lo ? done : keep_lo;
        const uint64_t e0 = done + cn < keep_lo + keep_cnt ? done + cn : keep_lo + keep_cnt;
        if (e0 > b0)
            GA_HIP_CHECK(hipMemcpyAsync((char*)d_dst + (b0 - keep_lo) * psz, (const char*)sg.d_points + (b0 - done) * psz, (e0 - b0) * psz,
                                        hipMemcpyDeviceToDevice, ctx->stream));
        done += cn;
    }
    uint32_t bad = 0;
    GA_HIP_CHECK(hipMemcpyAsync(&bad, sg.d_bad, 4, hipMemcpyDeviceToHost, ctx->stream));
    GA_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (bad) {
        set_error("key file: %u of %llu points do not decode (bad flag bits, coordinate >= p, or not on the curve)", bad,
                  (unsigned long long)count);
        return GA_ERR_INVALID;
    }
    return GA_OK;
}

// a []G1Affine / []G2Affine of the Encoder: u32 BE length, then the points (all compressed or all uncompressed)
// mode: the encoding of the STREAM (1 compressed, 0 raw), fixed once from [alpha]1 -- which is never infinity -- by pk_read; -1 =
// unknown, guess from the first byte of the vector (a vector that starts with a point at infinity is then ambiguous on BN254,
// whose infinity flag is the same in both encodings)
template <class C, int G>
static int read_encoded_vector(G16Stage* st, Staging& sg, ByteSource& src, int which, void** d_plain, uint64_t* len_out, int mode = -1) {
    typedef typename GroupField<C, G>::F F;
    uint32_t len = 0;
    GA_CHECK(src.u32be(&len));
    bool compressed = mode != 0;
    if (len && mode < 0) {
        uint8_t b0;
        GA_CHECK(src.peek(&b0, 1));
        PointFlags f;
        if (!point_flags<C>(b0, &f)) {
            set_error("key file: malformed flag bits 0x%02x at the head of a point vector", b0);
            return GA_ERR_INVALID;
        }
        compressed = f.compressed || (f.infinity && C::ID == GA_BLS12_381 && (b0 & 0x80));
    }
    if (len_out) *len_out = len;
    GA_CHECK(src.expect(len, compressed ? sizeof(Affine<F>) / 2 : sizeof(Affine<F>), "a point vector"));   // before any allocation
    if (which >= 0) {
        GA_CHECK(stage_reserve(st, which, len));
        G16Stage::Vec& x = st->v[which];
        GA_CHECK((decode_stream<C, G>(st->ctx, sg, src, len, compressed, x.d, x.lo, x.cnt)));
        x.seen = len;
        return GA_OK;
    }
    *d_plain = nullptr;   // a commitment basis: kept whole
    hipError_t e = device_malloc(d_plain, len ? (size_t)len * sizeof(Affine<F>) : 16);
    if (e != hipSuccess) {
        set_error("key file: hipMalloc of a commitment basis failed: %s", hipGetErrorString(e));
        return GA_ERR_NOMEM;
    }
    return decode_stream<C, G>(st->ctx, sg, src, len, compressed, *d_plain, 0, len);
}

// a slice of unsafe.WriteSlice: u64 LE length + gnark's own memory image; no arithmetic, file -> pinned buffer -> HBM
template <class C, int G>
static int read_dumped_vector(G16Stage* st, Staging& sg, ByteSource& src, int which, void** d_plain, uint64_t* len_out) {
    typedef typename GroupField<C, G>::F F;
    const size_t psz = sizeof(Affine<F>);
    uint64_t len = 0;
    GA_CHECK(src.u64le(&len));
    if (len >= (1ull << 40)) {
        set_error("key dump: implausible slice length %llu", (unsigned long long)len);
        return GA_ERR_INVALID;
    }
    if (len_out) *len_out = len;
    GA_CHECK(src.expect(len, psz, "a dumped slice"));   // before any allocation
    char* plain = nullptr;
    if (which >= 0) GA_CHECK(stage_reserve(st, which, len));
    else {
        hipError_t e = device_malloc((void**)&plain, len ? len * psz : 16);
        if (e != hipSuccess) {
            set_error("key dump: hipMalloc of a commitment basis failed: %s", hipGetErrorString(e));
            return GA_ERR_NOMEM;
        }
        *d_plain = plain;
    }
    const uint64_t per_chunk = Staging::BYTES / psz;
    int k = 0;
    for (uint64_t done = 0; done < len; k ^= 1) {
        const uint64_t cn = len - done < per_chunk ? len - done : per_chunk;
        GA_HIP_CHECK(hipEventSynchronize(sg.ev[k]));
        GA_CHECK(src.read(sg.h[k], cn * psz));
        if (which >= 0) GA_CHECK(stage_append(st, which, sg.h[k], cn, /*pinned=*/true));
        else GA_HIP_CHECK(hipMemcpyAsync(plain + done * psz, sg.h[k], cn * psz, hipMemcpyHostToDevice, st->ctx->stream));
        GA_HIP_CHECK(hipEventRecord(sg.ev[k], st->ctx->stream));
        done += cn;
    }
    GA_HIP_CHECK(hipStreamSynchronize(st->ctx->stream));
    return GA_OK;
}

// one point of the header (alpha, beta, delta): host arithmetic; advances the source by its encoded length
// *mode: -1 = not known yet, 0 = the stream holds uncompressed points, 1 = compressed; set by the first finite point.  BN254 has
// one flag value (0b01) for infinity in both modes, so an infinity point takes the size of the stream's mode (compressed when the
// mode is still unknown -- the size gnark-crypto's SetBytes consumes for it).
template <class C, int G>
static int read_header_point(ByteSource& src, std::vector<uint8_t>* out_image, int* mode = nullptr) {
    typedef typename GroupField<C, G>::F F;
    uint8_t buf[2 * sizeof(F)];
    GA_CHECK(src.peek(buf, 1));
    PointFlags f;
    if (!point_flags<C>(buf[0], &f)) {
        set_error("key file: malformed flag bits 0x%02x", buf[0]);
        return GA_ERR_INVALID;
    }
    bool compressed = f.compressed;
    if (f.infinity && C::ID == GA_BN254) compressed = !(mode && *mode == 0);
    if (!f.infinity && mode && *mode < 0) *mode = f.compressed ? 1 : 0;
    const size_t len = compressed ? sizeof(F) : 2 * sizeof(F);
    GA_CHECK(src.read(buf, len));
    Affine<F> p;
    if (!point_decode<C, G>(buf, f.compressed, &p)) {
        set_error("key file: header point does not decode");
        return GA_ERR_INVALID;
    }
    out_image->assign(reinterpret_cast<const uint8_t*>(&p), reinterpret_cast<const uint8_t*>(&p) + sizeof(p));
    return GA_OK;
}

// fft.Domain.WriteTo: cardinality + five fr elements (+ the withPrecompute byte of newer gnark-crypto versions, detected by
// trying to decode [alpha]1 right after it)
template <class C>
static int read_domain(ByteSource& src, uint64_t* cardinality) {
    typedef Fe<typename C::FpP> F1;
    GA_CHECK(src.u64be(cardinality));
    uint8_t skip[5 * 32];
    GA_CHECK(src.read(skip, sizeof skip));
    if (*cardinality == 0 || (*cardinality & (*cardinality - 1)) || *cardinality > (1ull << C::FrP::ADICITY)) {
        set_error("key file: domain cardinality %llu is not a power of two within the field's 2-adicity", (unsigned long long)*cardinality);
        return GA_ERR_INVALID;
    }
    uint8_t win[1 + 2 * sizeof(F1)];
    GA_CHECK(src.peek(win, sizeof win));
    auto decodes = [&](const uint8_t* b) {
        PointFlags f;
        Affine<F1> p;
        return point_flags<C>(b[0], &f) && !f.infinity && point_decode<C, GA_G1>(b, f.compressed, &p);
    };
    if (win[0] <= 1 && decodes(win + 1)) {
        uint8_t flag;
        return src.read(&flag, 1);   // withPrecompute
    }
    if (decodes(win)) return GA_OK;
    set_error("key file: [alpha]1 does not decode after the domain block (neither with nor without the withPrecompute byte)");
    return GA_ERR_INVALID;
}

static bool source_is_dump(ByteSource& src) {
    uint8_t m[8];
    static const uint8_t marker[8] = {0xef, 0xbe, 0xad, 0xde, 0, 0, 0, 0};   // uint64(0xdeadbeef) as this (little-endian) platform stores it
    return src.peek(m, 8) == GA_OK && memcmp(m, marker, 8) == 0;
}

template <class C>
static int pk_read(Ctx* ctx, ByteSource& src, int32_t precompute, uint32_t shard_index, uint32_t shard_count, const uint64_t* k_remove,
                   uint64_t len_k_remove, G16Pk** out) {
    G16Stage st;
    st.ctx = ctx;
    st.curve = C::ID;
    st.shard_index = shard_index;
    st.shard_count = shard_count ? shard_count : 1;
    if (st.shard_index >= st.shard_count) {
        set_error("proving key: shard_index %u >= shard_count %u", st.shard_index, st.shard_count);
        return GA_ERR_INVALID;
    }
    Staging sg;
    GA_CHECK(sg.init());
    const bool dump = source_is_dump(src);
    if (dump) {
        uint8_t m[8];
        GA_CHECK(src.read(m, 8));
    }
    GA_CHECK(read_domain<C>(src, &st.n));
    auto header_tail = [&]() -> int {   // nbWires, NbInfinityA, NbInfinityB, InfinityA, InfinityB, nbCommitments (marshal.go:263-270,335-349)
        uint64_t nb_wires, nia, nib;
        GA_CHECK(src.u64be(&nb_wires));
        GA_CHECK(src.u64be(&nia));
        GA_CHECK(src.u64be(&nib));
        if (nb_wires >= (1ull << 32)) {
            set_error("key file: %llu wires", (unsigned long long)nb_wires);
            return GA_ERR_INVALID;
        }
        if (nia > nb_wires || nib > nb_wires) {
            set_error("key file: %llu / %llu infinity entries for %llu wires", (unsigned long long)nia, (unsigned long long)nib, (unsigned long long)nb_wires);
            return GA_ERR_INVALID;
        }
        if (src.fd < 0 && 2 * nb_wires > src.mem_len - src.mem_pos) {   // (never size a buffer from an untrusted count alone)
            set_error("key image: unexpected end of input (%llu wires announced, %zu bytes left)", (unsigned long long)nb_wires, src.mem_len - src.mem_pos);
            return GA_ERR_INVALID;
        }
        st.nb_wires = nb_wires;
        for (int k = 0; k < 2; k++) {
            st.inf[k].clear();
            for (uint64_t done = 0; done < nb_wires;) {   // grown as the bytes really arrive
                const uint64_t cn = nb_wires - done < (1u << 20) ? nb_wires - done : (1u << 20);
                st.inf[k].resize(done + cn);
                GA_CHECK(src.read(st.inf[k].data() + done, cn));
                done += cn;
            }
            st.have_inf[k] = true;
            uint64_t ones = 0;
            for (uint8_t b : st.inf[k]) ones += b != 0;
            if (ones != (k == 0 ? nia : nib)) {
                set_error("key file: Infinity%c holds %llu set entries, the header says %llu", k == 0 ? 'A' : 'B', (unsigned long long)ones,
                          (unsigned long long)(k == 0 ? nia : nib));
                return GA_ERR_INVALID;
            }
        }
        return GA_OK;
    };
    uint32_t nb_commitments = 0;
    int mode = -1;   // compressed (1) or raw (0) stream: taken from [alpha]1, the first point, which is never infinity
    if (!dump) {   // ReadFrom order, marshal.go:316-330
        GA_CHECK((read_header_point<C, GA_G1>(src, &st.pts[GA_KEY_G1_ALPHA], &mode)));
        GA_CHECK((read_header_point<C, GA_G1>(src, &st.pts[GA_KEY_G1_BETA], &mode)));
        GA_CHECK((read_header_point<C, GA_G1>(src, &st.pts[GA_KEY_G1_DELTA], &mode)));
        for (int w : {GA_KEY_G1_A, GA_KEY_G1_B, GA_KEY_G1_Z, GA_KEY_G1_K}) GA_CHECK((read_encoded_vector<C, GA_G1>(&st, sg, src, w, nullptr, nullptr, mode)));
        GA_CHECK((read_header_point<C, GA_G2>(src, &st.pts[GA_KEY_G2_BETA], &mode)));
        GA_CHECK((read_header_point<C, GA_G2>(src, &st.pts[GA_KEY_G2_DELTA], &mode)));
        GA_CHECK((read_encoded_vector<C, GA_G2>(&st, sg, src, GA_KEY_G2_B, nullptr, nullptr, mode)));
        GA_CHECK(header_tail());
        GA_CHECK(src.u32be(&nb_commitments));
    } else {       // ReadDump order, marshal.go:459-478
        GA_CHECK((read_header_point<C, GA_G1>(src, &st.pts[GA_KEY_G1_ALPHA])));
        GA_CHECK((read_header_point<C, GA_G1>(src, &st.pts[GA_KEY_G1_BETA])));
        GA_CHECK((read_header_point<C, GA_G1>(src, &st.pts[GA_KEY_G1_DELTA])));
        GA_CHECK((read_header_point<C, GA_G2>(src, &st.pts[GA_KEY_G2_BETA])));
        GA_CHECK((read_header_point<C, GA_G2>(src, &st.pts[GA_KEY_G2_DELTA])));
        GA_CHECK(header_tail());
        GA_CHECK(src.u32be(&nb_commitments));
        for (int w : {GA_KEY_G1_A, GA_KEY_G1_B, GA_KEY_G1_Z, GA_KEY_G1_K}) GA_CHECK((read_dumped_vector<C, GA_G1>(&st, sg, src, w, nullptr, nullptr)));
        GA_CHECK((read_dumped_vector<C, GA_G2>(&st, sg, src, GA_KEY_G2_B, nullptr, nullptr)));
    }
    if (nb_commitments > 4096) {
        set_error("key file: implausible number of commitment keys %u", nb_commitments);
        return GA_ERR_INVALID;
    }
    for (uint32_t i = 0; i < nb_commitments; i++) {   // pedersen.ProvingKey: Basis, BasisExpSigma
        void *db = nullptr, *ds = nullptr;
        uint64_t lb = 0, ls = 0;
        int rc = dump ? read_dumped_vector<C, GA_G1>(&st, sg, src, -1, &db, &lb) : read_encoded_vector<C, GA_G1>(&st, sg, src, -1, &db, &lb, mode);
        if (rc == GA_OK) rc = dump ? read_dumped_vector<C, GA_G1>(&st, sg, src, -1, &ds, &ls) : read_encoded_vector<C, GA_G1>(&st, sg, src, -1, &ds, &ls, mode);
        if (rc == GA_OK && lb != ls) {
            set_error("key file: commitment key %u has %llu basis points and %llu sigma points", i, (unsigned long long)lb, (unsigned long long)ls);
            rc = GA_ERR_INVALID;
        }
        if (rc != GA_OK) {
            hipFree(db);
            hipFree(ds);
            return rc;
        }
        st.d_ck_basis.push_back(db);
        st.d_ck_sigma.push_back(ds);
        st.ck_len.push_back(lb);
    }
    if (len_k_remove) st.k_remove.assign(k_remove, k_remove + len_k_remove);
    return stage_finish<C>(&st, precompute, out);
}

// ---- key writers: the host description (ga_g16_key) -> WriteTo / WriteRawTo / WriteDump bytes ------------------------------------------
template <class C, int G>
static int write_encoded_vector(Ctx* ctx, Staging& sg, ByteSink& dst, const void* pts, uint64_t len, bool compressed) {
    typedef typename GroupField<C, G>::F F;
    if (len >= (1ull << 32)) {
        set_error("key writer: a vector of %llu points does not fit the u32 length prefix", (unsigned long long)len);
        return GA_ERR_INVALID;
    }
    GA_CHECK(dst.u32be((uint32_t)len));
    const size_t enc = compressed ? sizeof(F) : 2 * sizeof(F), psz = sizeof(Affine<F>);
    const uint64_t per_chunk = Staging::BYTES / psz;
    for (uint64_t done = 0; done < len;) {
        const uint64_t cn = len - done < per_chunk ? len - done : per_chunk;
        GA_HIP_CHECK(hipMemcpyAsync(sg.d_points, (const char*)pts + done * psz, cn * psz, hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL((key_encode_kernel<C, G>), dim3((unsigned)((cn + 63) / 64)), dim3(64), 0, ctx->stream, (const void*)sg.d_points, cn,
                           compressed ? 1 : 0, sg.d_bytes);
        GA_KERNEL_CHECK();
        GA_HIP_CHECK(hipMemcpyAsync(sg.h[0], sg.d_bytes, cn * enc, hipMemcpyDeviceToHost, ctx->stream));
        GA_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        GA_CHECK(dst.write(sg.h[0], cn * enc));
        done += cn;
    }
    return GA_OK;
}
template <class C, int G>
static int write_header_point(ByteSink& dst, const void* affine, bool compressed) {
    typedef typename GroupField<C, G>::F F;
    Affine<F> p;
    memcpy(&p, affine, sizeof p);
    uint8_t buf[2 * sizeof(F)];
    point_encode<C, G>(p, compressed, buf);
    return dst.write(buf, compressed ? sizeof(F) : 2 * sizeof(F));
}
template <class C>
static int write_domain(ByteSink& dst, uint64_t n) {
    typedef typename C::FrP FrP;
    typedef Fe<FrP> F;
    const int logn = ilog2_u64(n);
    if ((1ull << logn) != n || logn > FrP::ADICITY) {
        set_error("key writer: domain cardinality %llu", (unsigned long long)n);
        return GA_ERR_INVALID;
    }
    F w = fe_const<FrP>(FrP::ROOT), wi = fe_const<FrP>(FrP::ROOT_INV);
    for (int k = 0; k < FrP::ADICITY - logn; k++) {
        w = sqr(w);
        wi = sqr(wi);
    }
    F card = fe_zero<FrP>();
    card.l[0] = (uint32_t)n;
    card.l[1] = (uint32_t)(n >> 32);
    const F elems[5] = {inv(to_mont(card)), w, wi, fe_const<FrP>(FrP::GEN), fe_const<FrP>(FrP::GEN_INV)};
    GA_CHECK(dst.u64be(n));
    for (const F& e : elems) {
        uint8_t b[32];
        fe_to_be_bytes(e, b);
        GA_CHECK(dst.write(b, 32));
    }
    const uint8_t with_precompute = 1;
    return dst.write(&with_precompute, 1);
}
template <class C>
static int key_write(Ctx* ctx, const ga_g16_key* key, int format, ByteSink& dst) {
    typedef Fe<typename C::FpP> F1;
    typedef Fe2<typename C::FpP> F2;
    const bool dump = format == GA_KEY_FORMAT_DUMP, compressed = format == GA_KEY_FORMAT_COMPRESSED;
    Staging sg;
    GA_CHECK(sg.init());
    if (dump) GA_CHECK(dst.u64le(0xdeadbeefull));
    GA_CHECK(write_domain<C>(dst, key->domain_cardinality));
    const bool hc = compressed;   // header points follow the encoder's mode (raw for the dump)
    auto tail = [&]() -> int {
        GA_CHECK(dst.u64be(key->nb_wires));
        GA_CHECK(dst.u64be(key->nb_infinity_a));
        GA_CHECK(dst.u64be(key->nb_infinity_b));
        GA_CHECK(dst.write(key->infinity_a, key->nb_wires));
        GA_CHECK(dst.write(key->infinity_b, key->nb_wires));
        return dst.u32be(key->nb_commitments);
    };
    auto slice = [&](const void* p, uint64_t len, size_t psz) -> int {
        GA_CHECK(dst.u64le(len));
        return dst.write(p, len * psz);
    };
    GA_CHECK((write_header_point<C, GA_G1>(dst, key->g1_alpha, hc)));
    GA_CHECK((write_header_point<C, GA_G1>(dst, key->g1_beta, hc)));
    GA_CHECK((write_header_point<C, GA_G1>(dst, key->g1_delta, hc)));
    if (!dump) {
        GA_CHECK((write_encoded_vector<C, GA_G1>(ctx, sg, dst, key->g1_a, key->len_a, compressed)));
        GA_CHECK((write_encoded_vector<C, GA_G1>(ctx, sg, dst, key->g1_b, key->len_b, compressed)));
        GA_CHECK((write_encoded_vector<C, GA_G1>(ctx, sg, dst, key->g1_z, key->len_z, compressed)));
        GA_CHECK((write_encoded_vector<C, GA_G1>(ctx, sg, dst, key->g1_k, key->len_k, compressed)));
    }
    GA_CHECK((write_header_point<C, GA_G2>(dst, key->g2_beta, hc)));
    GA_CHECK((write_header_point<C, GA_G2>(dst, key->g2_delta, hc)));
    if (!dump) GA_CHECK((write_encoded_vector<C, GA_G2>(ctx, sg, dst, key->g2_b, key->len_b2, compressed)));
    GA_CHECK(tail());
    if (dump) {
        GA_CHECK(slice(key->g1_a, key->len_a, sizeof(Affine<F1>)));
        GA_CHECK(slice(key->g1_b, key->len_b, sizeof(Affine<F1>)));
        GA_CHECK(slice(key->g1_z, key->len_z, sizeof(Affine<F1>)));
        GA_CHECK(slice(key->g1_k, key->len_k, sizeof(Affine<F1>)));
        GA_CHECK(slice(key->g2_b, key->len_b2, sizeof(Affine<F2>)));
    }
    for (uint32_t i = 0; i < key->nb_commitments; i++) {
        if (dump) {
            GA_CHECK(slice(key->ck_basis[i], key->ck_len[i], sizeof(Affine<F1>)));
            GA_CHECK(slice(key->ck_basis_exp_sigma[i], key->ck_len[i], sizeof(Affine<F1>)));
        } else {
            GA_CHECK((write_encoded_vector<C, GA_G1>(ctx, sg, dst, key->ck_basis[i], key->ck_len[i], compressed)));
            GA_CHECK((write_encoded_vector<C, GA_G1>(ctx, sg, dst, key->ck_basis_exp_sigma[i], key->ck_len[i], compressed)));
        }
    }
    return GA_OK;
}

// Proof.ReadFrom (marshal.go:62-86): Ar | Bs | Krs | u32 n | n commitments | CommitmentPok, compressed or uncompressed points
template <class C>
static int proof_unmarshal(const uint8_t* data, size_t len, void* proof_out, void* commitments_out, uint32_t max_commitments,
                           uint32_t* n_commitments, void* pok_out, size_t* consumed) {
    typedef Fe<typename C::FpP> F1;
    typedef Fe2<typename C::FpP> F2;
    ByteSource src;
    src.mem = data;
    src.mem_len = len;
    std::vector<uint8_t> img;
    int mode = -1;
    char* o = reinterpret_cast<char*>(proof_out);
    GA_CHECK((read_header_point<C, GA_G1>(src, &img, &mode)));
    memcpy(o, img.data(), sizeof(Affine<F1>));
    GA_CHECK((read_header_point<C, GA_G2>(src, &img, &mode)));
    memcpy(o + sizeof(Affine<F1>), img.data(), sizeof(Affine<F2>));
    GA_CHECK((read_header_point<C, GA_G1>(src, &img, &mode)));
    memcpy(o + sizeof(Affine<F1>) + sizeof(Affine<F2>), img.data(), sizeof(Affine<F1>));
    uint32_t n = 0;
    GA_CHECK(src.u32be(&n));
    if (n > max_commitments || (n && !commitments_out)) {
        set_error("proof: %u commitments, room for %u", n, max_commitments);
        return GA_ERR_INVALID;
    }
    for (uint32_t i = 0; i < n; i++) {
        GA_CHECK((read_header_point<C, GA_G1>(src, &img, &mode)));
        memcpy(reinterpret_cast<char*>(commitments_out) + (size_t)i * sizeof(Affine<F1>), img.data(), sizeof(Affine<F1>));
    }
    GA_CHECK((read_header_point<C, GA_G1>(src, &img, &mode)));
    if (pok_out) memcpy(pok_out, img.data(), sizeof(Affine<F1>));
    if (n_commitments) *n_commitments = n;
    if (consumed) *consumed = src.mem_pos;
    return GA_OK;
}

// ---- the device part of a proof, in pieces (a multi-GPU proof runs them on different devices) ----------------------------------
// witness_msms : upload W (only the wire range this shard's bases cover), filter, MSM A, B (G1 + G2), K      prove.go:147-237,283
// h_chain      : v <- FFT_coset(iFFT(v)) for one of the solver's A, B, C                                       prove.go:362-368
// h_combine    : h <- iFFT_coset((a*b - c) * den), bit-reversed like pk.G1.Z                                  prove.go:377-386
// z_msm        : MSM over this shard's slice of pk.G1.Z and h                                                  prove.go:225-227
// W -> device, only the wire range this shard reads.  Returns when the copy has been handed to the DMA engine from pageable
// memory, i.e. when the host buffer has been consumed -- callers start the (PCIe-competing) upload of A, B, C only after it.
static int witness_upload(G16Pk* pk, const SlotLease& slot, const void* w, uint64_t nb_public, hipStream_t up_stream = nullptr) {
    Ctx* ctx = pk->ctx;
    if (nb_public > pk->nb_wires || pk->nb_wires - nb_public != pk->full_len_k + pk->len_k_remove) {
        set_error("prove: inconsistent sizes (nbWires %llu - nbPublic %llu != len(K) %llu + len(k_remove) %llu)", (unsigned long long)pk->nb_wires,
                  (unsigned long long)nb_public, (unsigned long long)pk->full_len_k, (unsigned long long)pk->len_k_remove);
        return GA_ERR_INVALID;
    }
    void* d_w;
    GA_CHECK(ctx->scratch_get(slot.name("g16_w").c_str(), pk->nb_wires * 32, &d_w));
    if (!up_stream) up_stream = ctx->work_stream();
    // the wire range this shard reads: everything for an unsharded key, ~1/N of W for shard k of N (the gather lists of a
    // shard are contiguous pieces of the sorted wire lists); K's range depends on nbPublic
    uint64_t lo = pk->w_lo, hi = pk->w_hi;
    if (pk->len_k && !pk->d_idx_k) {
        const uint64_t klo = nb_public + pk->off_k, khi = klo + pk->len_k;
        lo = lo < klo ? lo : klo;
        hi = hi > khi ? hi : khi;
    }
    if (hi > pk->nb_wires) hi = pk->nb_wires;
    if (lo > hi) lo = hi;
    // (timed only on the main stream: a staging thread runs outside the device lock that guards the profiler's stage list)
    StageTimer tm(up_stream == ctx->work_stream() ? ctx : nullptr, "g16_h2d_w", up_stream);
    if (hi > lo) GA_HIP_CHECK(ctx->h2d_pageable((char*)d_w + lo * 32, (const char*)w + lo * 32, (hi - lo) * 32, up_stream, /*prio=*/0));
    return GA_OK;
}

// What the two halves of a split proof share (prove_partial): the sort of the whole witness, made once on the lane of the
// witness MSMs, and the K MSM, which goes to whichever lane gets to it first.
struct WitnessShared {
    MsmPrepared prep_w;            // digits + sort of W (wire-indexed tables); arrays live in the witness lane's scratch
    bool w_live = false;
    void* d_w = nullptr;           // W on the device (the slot's buffer, in the witness lane's scratch namespace)
    hipEvent_t w_ev = nullptr;     // recorded on the witness lane's stream once W is on the device and prep_w has been launched
    std::mutex mu;
    std::condition_variable cv;
    bool posted = false, failed = false;   // host-side: prep_w / w_ev are valid (or never will be)
    std::atomic<int> k_owner{-1};          // lane that claimed the K MSM
    void post(bool ok) {
        {
            std::lock_guard<std::mutex> g(mu);
            posted = true;
            failed = !ok;
        }
        cv.notify_all();
    }
    bool wait_posted() {
        std::unique_lock<std::mutex> g(mu);
        cv.wait(g, [&] { return posted; });
        return !failed;
    }
    bool claim_k(int lane) {
        int expected = -1;
        return k_owner.compare_exchange_strong(expected, lane);
    }
    ~WitnessShared() {
        if (w_ev) hipEventDestroy(w_ev);
    }
};

// this device's share of the windows of a table with window width c (everything unless the key is window-sharded)
template <class C>
static void g16_window_share(const G16Pk* pk, int c, int* lo, int* hi) {
    window_share(C::FrP::BITS / c + 1, pk->win_index, pk->win_count, lo, hi);
}

// one G1 MSM over a compact (not wire-indexed) table with its own digits + sort; `prep` is left holding them
template <class C>
static int g16_table_msm_g1(G16Pk* pk, const void* table, const void* scal, uint64_t len, int c, MsmPrepared* prep, bool* prep_live,
                            XYZZ<Fe<typename C::FpP>>* out) {
    typedef Fe<typename C::FpP> F1;
    int lo, hi;
    g16_window_share<C>(pk, c, &lo, &hi);
    *prep_live = false;
    if (len == 0 || hi <= lo) {
        *out = xyzz_inf<F1>();
        return GA_OK;
    }
    GA_CHECK(msm_prepare_table_scalars<C>(pk->ctx, scal, len, true, c, prep, 0, lo, hi));
    *prep_live = true;
    return msm_table_device_reuse<C, GA_G1>(pk->ctx, table, *prep, out);
}

// the K MSM (prove.go:231-237) on the CALLING thread's lane: over the shared witness sort when K's table is wire-indexed,
// otherwise over its own gather + sort (scratch of the calling lane).  `sh` must have been posted.
template <class C>
static int k_msm(G16Pk* pk, uint64_t nb_public, WitnessShared& sh, XYZZ<Fe<typename C::FpP>>* out) {
    typedef Fe<typename C::FpP> F1;
    Ctx* ctx = pk->ctx;
    if (pk->share_k) {
        if (!sh.w_live) {
            *out = xyzz_inf<F1>();
            return GA_OK;
        }
        return msm_table_device_reuse<C, GA_G1>(ctx, pk->d_k, sh.prep_w, out);
    }
    void* const d_w = sh.d_w;   // (not scratch_get: the calling thread may be on the partner lane, whose namespace differs)
    const void* d_wk = (const char*)d_w + (nb_public + pk->off_k) * 32;
    if (pk->d_idx_k) {
        void* g;
        GA_CHECK(ctx->scratch_get("g16_wk", pk->len_k * 32 + 32, &g));
        GA_CHECK(util_gather_fr<C>(ctx, g, d_w, pk->d_idx_k, pk->len_k));
        d_wk = g;
    }
    if (pk->tab_k) {
        MsmPrepared prep;
        bool live;
        return g16_table_msm_g1<C>(pk, pk->d_k, d_wk, pk->len_k, pk->c_k, &prep, &live, out);
    }
    GA_CHECK(await_vector(pk, GA_KEY_G1_K));
    return host_msm<C, GA_G1>(ctx, pk->d_k, d_wk, pk->len_k, true, out, pk->win_index, pk->win_count);
}

// The witness MSMs A, B (G1 and G2) and -- unless the partner lane claims it first -- K, on the calling thread's lane.
// `sh` is posted as soon as the shared witness sort has been launched; o_k is written only when *did_k comes back true.
template <class C>
static int witness_msms(G16Pk* pk, const SlotLease& slot, uint64_t nb_public, WitnessShared& sh, XYZZ<Fe<typename C::FpP>>* o_ar,
                        XYZZ<Fe<typename C::FpP>>* o_bs1, XYZZ<Fe<typename C::FpP>>* o_k, XYZZ<Fe2<typename C::FpP>>* o_bs2, bool* did_k) {
    typedef Fe<typename C::FpP> F1;
    typedef Fe2<typename C::FpP> F2;
    Ctx* ctx = pk->ctx;
    struct PostGuard {   // a failure before the post must still release the partner lane
        WitnessShared& sh;
        bool done = false;
        ~PostGuard() {
            if (!done) sh.post(false);
        }
    } pg{sh};
    *did_k = false;
    if (nb_public > pk->nb_wires || pk->nb_wires - nb_public != pk->full_len_k + pk->len_k_remove) {
        set_error("prove: inconsistent sizes (nbWires %llu - nbPublic %llu != len(K) %llu + len(k_remove) %llu)", (unsigned long long)pk->nb_wires,
                  (unsigned long long)nb_public, (unsigned long long)pk->full_len_k, (unsigned long long)pk->len_k_remove);
        return GA_ERR_INVALID;
    }
    hipStream_t st = ctx->work_stream();
    void *d_w, *d_wa, *d_wb;
    GA_CHECK(ctx->scratch_get(slot.name("g16_w").c_str(), pk->nb_wires * 32, &d_w));
    GA_CHECK(ctx->scratch_get("g16_wa", pk->len_a * 32 + 32, &d_wa));
    GA_CHECK(ctx->scratch_get("g16_wb", pk->len_b * 32 + 32, &d_wb));
    // digits + sort of the WHOLE witness once (scratch slot 1), reused by every wire-indexed table
    if (pk->share_a || pk->share_b || pk->share_k) {
        int lo, hi;
        g16_window_share<C>(pk, pk->c_w, &lo, &hi);
        if (hi > lo) {
            GA_CHECK(msm_prepare_table_scalars<C>(ctx, d_w, pk->nb_wires, true, pk->c_w, &sh.prep_w, 1, lo, hi));
            sh.w_live = true;
        }
    }
    sh.d_w = d_w;
    GA_HIP_CHECK(hipEventRecord(sh.w_ev, st));
    pg.done = true;
    sh.post(true);
    // ---- wire filtering (prove.go:147-168) ------------------------------------------------------------
    if (!pk->share_a) GA_CHECK(util_gather_fr<C>(ctx, d_wa, d_w, pk->d_idx_a, pk->len_a));
    if (!pk->share_b || !pk->share_b2) GA_CHECK(util_gather_fr<C>(ctx, d_wb, d_w, pk->d_idx_b, pk->len_b));   // (G2.B may be plain beside a wire-indexed G1.B)
    // ---- the witness MSMs (prove.go:194,207,237,283) ---------------------------------------------------
    XYZZ<F1> ar, bs1;
    XYZZ<F2> bs2;
    {   // every vector on its own kind of path: wire-indexed table over the shared sort, compact table with its own digits + sort, or
        // plain bases (no table: un-pinned MSM, one bucket set per window + Horner)
        MsmPrepared prep;
        bool prep_live = false;   // `prep` holds the digits of wB for G2.B
        auto shared_g1 = [&](const void* table, XYZZ<F1>* out) -> int {
            if (!sh.w_live) {
                *out = xyzz_inf<F1>();
                return GA_OK;
            }
            return msm_table_device_reuse<C, GA_G1>(ctx, table, sh.prep_w, out);
        };
        // The wire-indexed G1 tables (A, B1 and -- when this lane gets it -- K) in ONE pass of the bucket kernel, the merge and the window
        // reduction over the shared witness sort (msm_table_device_reuse_multi): one kernel tail, one reduction with k x the waves and
        // one host synchronisation instead of k of each (prove.go:194,207,237: three MultiExp over the same wireValues).  A table the
        // bucket kernel has found degenerate (a DummySetup key) keeps its own pass with the complete loop; GA_G16_BATCH_TABLES=0: round 5.
        bool multi_a = false, multi_b = false;
        if (sh.w_live && ctx->tun.g16_batch_tables.load(std::memory_order_relaxed)) {
            const void* tabs[3];
            XYZZ<F1>* dst[3];
            int nt = 0;
            auto want = [&](bool shared, const void* table, XYZZ<F1>* out) {
                if (!shared || ctx->is_degenerate(table)) return false;
                tabs[nt] = table;
                dst[nt++] = out;
                return true;
            };
            multi_a = want(pk->share_a, pk->d_a, &ar);
            multi_b = want(pk->share_b, pk->d_b, &bs1);
            const bool k_fits = pk->share_k && !ctx->is_degenerate(pk->d_k);
            if (nt + (k_fits ? 1 : 0) >= 2) {
                if (k_fits && sh.claim_k(current_lane())) {
                    want(true, pk->d_k, o_k);
                    *did_k = true;
                }
                if (nt >= 2) {
                    XYZZ<F1> sums[3];
                    GA_CHECK((msm_table_device_reuse_multi<C, GA_G1>(ctx, tabs, nt, sh.prep_w, sums)));
                    for (int i = 0; i < nt; i++) *dst[i] = sums[i];
                } else {   // (K went to the partner lane after all: one table left)
                    GA_CHECK(shared_g1(tabs[0], dst[0]));
                }
            } else {
                multi_a = multi_b = false;
            }
        }
        if (multi_a) {
        } else if (pk->share_a) GA_CHECK(shared_g1(pk->d_a, &ar));
        else if (pk->tab_a) GA_CHECK(g16_table_msm_g1<C>(pk, pk->d_a, d_wa, pk->len_a, pk->c_a, &prep, &prep_live, &ar));
        else {
            GA_CHECK(await_vector(pk, GA_KEY_G1_A));
            GA_CHECK((host_msm<C, GA_G1>(ctx, pk->d_a, d_wa, pk->len_a, true, &ar, pk->win_index, pk->win_count)));
        }
        prep_live = false;   // (whatever A left in `prep` is not wB's)
        if (multi_b) {
        } else if (pk->share_b) GA_CHECK(shared_g1(pk->d_b, &bs1));
        else if (pk->tab_b) GA_CHECK(g16_table_msm_g1<C>(pk, pk->d_b, d_wb, pk->len_b, pk->c_b, &prep, &prep_live, &bs1));
        else {
            GA_CHECK(await_vector(pk, GA_KEY_G1_B));
            GA_CHECK((host_msm<C, GA_G1>(ctx, pk->d_b, d_wb, pk->len_b, true, &bs1, pk->win_index, pk->win_count)));
        }
        if (pk->pending && !*did_k && sh.claim_k(current_lane())) {   // a key still on its way: K's MSM before G2.B's (upload order)
            GA_CHECK(k_msm<C>(pk, nb_public, sh, o_k));
            *did_k = true;
        }
        if (pk->share_b2) {   // G2.B wire-indexed: the shared witness sort again
            if (sh.w_live) GA_CHECK((msm_table_device_reuse<C, GA_G2>(ctx, pk->d_b2, sh.prep_w, &bs2)));
            else bs2 = xyzz_inf<F2>();
        } else if (pk->tab_b2) {   // compact G2.B table (window width c_b): the digits / sort of wB that G1.B's compact table has just made, or its own
            int lo, hi;
            g16_window_share<C>(pk, pk->c_b, &lo, &hi);
            if (pk->len_b2 == 0 || hi <= lo) {
                bs2 = xyzz_inf<F2>();
            } else {
                if (!prep_live) GA_CHECK(msm_prepare_table_scalars<C>(ctx, d_wb, pk->len_b2, true, pk->c_b, &prep, 0, lo, hi));
                GA_CHECK((msm_table_device_reuse<C, GA_G2>(ctx, pk->d_b2, prep, &bs2)));
            }
        } else {
            GA_CHECK(await_vector(pk, GA_KEY_G2_B));
            GA_CHECK((host_msm<C, GA_G2>(ctx, pk->d_b2, d_wb, pk->len_b2, true, &bs2, pk->win_index, pk->win_count)));
        }
    }
    *o_ar = ar;
    *o_bs1 = bs1;
    *o_bs2 = bs2;
    if (!*did_k && sh.claim_k(current_lane())) {
        GA_CHECK(k_msm<C>(pk, nb_public, sh, o_k));
        *did_k = true;
    }
    return GA_OK;
}

// v (host, n_constraints elements) -> device buffer d_v (n elements, zero-padded) on `up_stream`; the main stream waits for it
static int h_upload(G16Pk* pk, const void* v, uint64_t n_constraints, void* d_v, hipStream_t up_stream) {
    const uint64_t n = pk->n;
    if (n_constraints > n) {
        set_error("prove: %llu constraints exceed the domain cardinality %llu", (unsigned long long)n_constraints, (unsigned long long)n);
        return GA_ERR_INVALID;
    }
    GA_HIP_CHECK(pk->ctx->h2d_pageable(d_v, v, n_constraints * 32, up_stream));
    if (n > n_constraints)   // computeH pads to the domain size (prove.go:356-359)
        GA_HIP_CHECK(hipMemsetAsync((char*)d_v + n_constraints * 32, 0, (n - n_constraints) * 32, up_stream));
    return GA_OK;
}

template <class C>
static int z_msm(G16Pk* pk, const void* d_h_slice, XYZZ<Fe<typename C::FpP>>* out) {
    typedef Fe<typename C::FpP> F1;
    Ctx* ctx = pk->ctx;
    if (pk->len_z == 0) {
        *out = xyzz_inf<F1>();
        return GA_OK;
    }
    if (pk->tab_z) {
        int lo, hi;
        window_share(C::FrP::BITS / pk->c_z + 1, pk->win_index, pk->win_count, &lo, &hi);
        if (hi <= lo) {
            *out = xyzz_inf<F1>();
            return GA_OK;
        }
        MsmPrepared prep;
        GA_CHECK(msm_prepare_table_scalars<C>(ctx, d_h_slice, pk->len_z, true, pk->c_z, &prep, 0, lo, hi));
        return msm_table_device_reuse<C, GA_G1>(ctx, pk->d_z, prep, out);
    }
    GA_CHECK(await_vector(pk, GA_KEY_G1_Z));
    return host_msm<C, GA_G1>(ctx, pk->d_z, d_h_slice, pk->len_z, true, out, pk->win_index, pk->win_count);
}

// RAII for the pieces that must not outlive an early return
struct EventGuard {
    hipEvent_t ev = nullptr;
    ~EventGuard() {
        if (ev) hipEventDestroy(ev);
    }
};
struct ThreadJoiner {
    std::thread& t;
    ~ThreadJoiner() {
        if (t.joinable()) t.join();
    }
};

// all four vectors of a solution into the slot's staging buffers on the slot's own stream (a caller that found the device busy
// with another proof does this while it waits); returns when the copies have completed, so the host buffers are free again
static int preload_solution(G16Pk* pk, const SlotLease& slot, const void* w, const void* a, const void* b, const void* c,
                            uint64_t n_constraints, uint64_t nb_public) {
    Ctx* ctx = pk->ctx;
    const uint64_t n = pk->n;
    hipStream_t st = ctx->slot_stream[slot.slot];
    GA_CHECK(witness_upload(pk, slot, w, nb_public, st));
    const void* src[3] = {a, b, c};
    static const char* const names[3] = {"h_a", "h_b", "h_c"};
    for (int k = 0; k < 3; k++) {
        void* d;
        GA_CHECK(ctx->scratch_get(slot.name(names[k]).c_str(), n * 32, &d));
        GA_CHECK(h_upload(pk, src[k], n_constraints, d, st));
    }
    GA_HIP_CHECK(hipStreamSynchronize(st));
    return GA_OK;
}

// The device part of a proof on this key's shard: computeH + the five MSMs over the pinned slices.
// Outputs (before randomisation): A-sum, B1-sum, K-sum + Z-sum (G1), B2-sum (G2) -- to be added across shards.
// preloaded: W, A, B, C already sit in the slot's buffers (preload_solution).
//
// Schedule.  The calling thread uploads W and runs the witness MSMs (A, B1, B2) on its own lane.  A helper thread uploads A, B, C
// on the slot's copy stream (pageable H2D copies block the thread that issues them) and -- when the partner lane is free -- also
// runs the H side there: each chain FFT_coset(iFFT(.)) as soon as its vector has landed, the point-wise step, the last
// transform, then the Z MSM over h.  The K MSM goes to whichever lane reaches it first.  The two halves share the device: the
// sorts, reduction tails, host round trips and launch gaps of one run under the bucket kernels of the other.  Without a partner
// lane (profiling on, GA_G16_SPLIT=0, or the lane is taken) the helper only uploads and the H side follows on the caller's lane,
// as in round 2.  Everything is joined before this function returns, so no host pointer outlives the call.
template <class C>
static int prove_partial(G16Pk* pk, const SlotLease& slot, bool preloaded, const void* w, const void* a, const void* b, const void* c,
                         uint64_t n_constraints, uint64_t nb_public, XYZZ<Fe<typename C::FpP>>* o_ar, XYZZ<Fe<typename C::FpP>>* o_bs1,
                         XYZZ<Fe<typename C::FpP>>* o_krs, XYZZ<Fe2<typename C::FpP>>* o_bs2) {
    typedef Fe<typename C::FpP> F1;
    Ctx* ctx = pk->ctx;
    const uint64_t n = pk->n;
    if (n_constraints > n) {
        set_error("prove: %llu constraints exceed the domain cardinality %llu", (unsigned long long)n_constraints, (unsigned long long)n);
        return GA_ERR_INVALID;
    }
    void* d_h[3];
    static const char* const names[3] = {"h_a", "h_b", "h_c"};
    for (int k = 0; k < 3; k++) GA_CHECK(ctx->scratch_get(slot.name(names[k]).c_str(), n * 32, &d_h[k]));
    const int lane = current_lane();
    const int partner = lane + 1;   // lanes pair up: (0, 1) and (2, 3)
    const bool may_split = !ctx->profiling && ctx->tun.g16_split && (lane == 0 || lane == 2);
    WitnessShared sh;
    GA_HIP_CHECK(hipEventCreateWithFlags(&sh.w_ev, hipEventDisableTiming));
    EventGuard ev[3];
    for (int k = 0; k < 3; k++) GA_HIP_CHECK(hipEventCreateWithFlags(&ev[k].ev, hipEventDisableTiming));
    // W first: the witness MSMs only need W, and the (PCIe-competing) upload of A, B, C starts when this one has been handed over
    if (!preloaded) GA_CHECK(witness_upload(pk, slot, w, nb_public));
    XYZZ<F1> k_part = xyzz_inf<F1>(), k_part_h = xyzz_inf<F1>(), z_part = xyzz_inf<F1>();
    bool split = false, h_did_k = false;
    int h_rc = GA_OK;
    std::string h_err;
    hipStream_t up = ctx->slot_stream[slot.slot];   // the slot's own copy stream: two proofs in flight do not queue their uploads
    std::thread helper;
    if (!preloaded || may_split)
        helper = std::thread([&]() {
          try {   // (an exception leaving a thread function terminates the process: host allocations below can throw)
            auto fail = [&](int rc, const char* what) {
                h_rc = rc;
                h_err = std::string(what) + ": " + get_error();
            };
            if (hipSetDevice(ctx->device) != hipSuccess) {
                set_error("hipSetDevice(%d) failed", ctx->device);
                return fail(GA_ERR_HIP, "selecting the device");
            }
            std::unique_lock<std::mutex> pl(ctx->lane_mu[partner < GA_NUM_LANES ? partner : 1], std::defer_lock);
            if (may_split && pl.try_lock()) split = true;
            LaneScope on_lane(split ? partner : lane);
            if (!preloaded) {
                const void* src[3] = {a, b, c};
                for (int k = 0; k < 3; k++) {
                    if (h_upload(pk, src[k], n_constraints, d_h[k], up) != GA_OK) return fail(GA_ERR_HIP, "uploading A, B, C");
                    if (hipEventRecord(ev[k].ev, up) != hipSuccess) {
                        set_error("hipEventRecord failed");
                        return fail(GA_ERR_HIP, "uploading A, B, C");
                    }
                }
            }
            auto drain_uploads = [&]() -> bool {
                if (preloaded || hipStreamSynchronize(up) == hipSuccess) return true;
                set_error("hipStreamSynchronize failed on the copy stream");
                fail(GA_ERR_HIP, "uploading A, B, C");
                return false;
            };
            if (!split) {
                drain_uploads();
                return;
            }
            ctx->stat_split++;
            hipStream_t st = ctx->work_stream();
            int rc = GA_OK;
            for (int k = 0; k < 3 && rc == GA_OK; k++) {
                if (!preloaded && hipStreamWaitEvent(st, ev[k].ev, 0) != hipSuccess) {
                    set_error("hipStreamWaitEvent failed");
                    rc = GA_ERR_HIP;
                    break;
                }
                rc = ntt_domain_h_chain<C>(pk->dom, d_h[k]);
            }
            if (rc == GA_OK) rc = ntt_domain_h_combine<C>(pk->dom, d_h[0], d_h[1], d_h[2]);   // h in d_h[0], bit-reversed like pk.G1.Z
            if (rc == GA_OK) rc = z_msm<C>(pk, (const char*)d_h[0] + pk->off_z * 32, &z_part);
            if (rc != GA_OK) {
                hipStreamSynchronize(st);
                drain_uploads();
                return fail(rc, "computeH / Z MSM");
            }
            if (hipStreamSynchronize(st) != hipSuccess) {   // (z_msm returns synchronised unless this shard's Z slice is empty)
                set_error("hipStreamSynchronize failed on the H lane");
                return fail(GA_ERR_HIP, "computeH / Z MSM");
            }
            if (!drain_uploads()) return;
            // the K MSM, if the witness lane has not got to it yet (its W and its witness sort are on the device: w_ev)
            if (sh.wait_posted() && sh.claim_k(partner)) {
                if (hipStreamWaitEvent(st, sh.w_ev, 0) != hipSuccess) {
                    set_error("hipStreamWaitEvent failed");
                    return fail(GA_ERR_HIP, "K MSM");
                }
                rc = k_msm<C>(pk, nb_public, sh, &k_part_h);
                if (rc != GA_OK) return fail(rc, "K MSM");
                h_did_k = true;
            }
          } catch (...) {
              h_rc = GA_ERR_STATE;
              h_err = "exception on the helper thread (out of host memory?)";
          }
        });
    ThreadJoiner joiner{helper};
    bool did_k = false;
    const int w_rc = witness_msms<C>(pk, slot, nb_public, sh, o_ar, o_bs1, &k_part, o_bs2, &did_k);
    if (helper.joinable()) helper.join();
    if (w_rc != GA_OK) {
        hipStreamSynchronize(ctx->work_stream());
        return w_rc;
    }
    if (h_rc != GA_OK) {
        set_error("prove (H side): %s", h_err.c_str());
        return h_rc;
    }
    if (!split) {
        // ---- H (prove.go:134,346-389), then the MSM over pk.G1.Z (prove.go:225-227), on this lane ------------
        if (!preloaded) GA_HIP_CHECK(hipStreamWaitEvent(ctx->work_stream(), ev[2].ev, 0));
        GA_CHECK(ntt_domain_compute_h<C>(pk->dom, d_h[0], d_h[1], d_h[2]));   // h in d_h[0], bit-reversed like pk.G1.Z
        GA_CHECK(z_msm<C>(pk, (const char*)d_h[0] + pk->off_z * 32, &z_part));
    }
    if (!did_k && !h_did_k) {
        set_error("prove: the K MSM was claimed by no lane");
        return GA_ERR_STATE;
    }
    *o_krs = add(did_k ? k_part : k_part_h, z_part);
    return GA_OK;
}

// fixed-base scalar multiplication on the host: table[w*15 + d-1] = d * 16^w * P for the 64 4-bit windows of a 256-bit scalar
template <class F>
static void fixed_base_table(const XYZZ<F>& p, std::vector<uint8_t>& blob) {
    blob.resize((size_t)64 * 15 * sizeof(XYZZ<F>));
    XYZZ<F>* t = reinterpret_cast<XYZZ<F>*>(blob.data());
    XYZZ<F> base = p;
    for (int w = 0; w < 64; w++) {
        XYZZ<F> acc = base;
        for (int d = 1; d <= 15; d++) {
            t[w * 15 + d - 1] = acc;
            acc = add(acc, base);
        }
        base = acc;   // 16 * base
    }
}
template <class F>
static XYZZ<F> fixed_base_mul(const std::vector<uint8_t>& blob, const uint32_t* k8) {
    const XYZZ<F>* t = reinterpret_cast<const XYZZ<F>*>(blob.data());
    XYZZ<F> r = xyzz_inf<F>();
    for (int w = 0; w < 64; w++) {
        const uint32_t d = (k8[w / 8] >> (4 * (w % 8))) & 15u;
        if (d) r = add(r, t[w * 15 + d - 1]);
    }
    return r;
}

// Host epilogue with the prover's randomness (prove.go:171-185,199-200,212-214,241-269,287-292) on the SUMMED partials.
template <class C>
static int finish(G16Pk* pk, XYZZ<Fe<typename C::FpP>> ar, XYZZ<Fe<typename C::FpP>> bs1, XYZZ<Fe<typename C::FpP>> krs,
                  XYZZ<Fe2<typename C::FpP>> bs2, const void* r_mont, const void* s_mont, void* proof_out) {
    typedef typename C::FrP FrP;
    typedef Fe<typename C::FpP> F1;
    typedef Fe2<typename C::FpP> F2;
    StageTimer tm(pk->ctx, "g16_epilogue_host");
    Fe<FrP> r, s;
    memcpy(r.l, r_mont, 32);
    memcpy(s.l, s_mont, 32);
    Fe<FrP> kr = neg(mul(r, s));
    Fe<FrP> rc = from_mont(r), sc = from_mont(s), krc = from_mont(kr);
    // [delta]*r, *s, *kr and [delta2]*s are fixed-base: 4-bit windows over a per-key table (63 additions instead of 254 doublings
    // + ~127 additions each; the epilogue sits on the critical path of a single proof: 1.5 -> 0.6 ms)
    std::call_once(pk->delta_tab_once, [&]() {
        fixed_base_table<F1>(host_load_affine<F1>(pk->delta1.data()), pk->delta1_tab);
        fixed_base_table<F2>(host_load_affine<F2>(pk->delta2.data()), pk->delta2_tab);
    });
    XYZZ<F1> d_r = fixed_base_mul<F1>(pk->delta1_tab, rc.l), d_s = fixed_base_mul<F1>(pk->delta1_tab, sc.l), d_kr = fixed_base_mul<F1>(pk->delta1_tab, krc.l);
    bs1 = add(add(bs1, host_load_affine<F1>(pk->beta1.data())), d_s);
    ar = add(add(ar, host_load_affine<F1>(pk->alpha1.data())), d_r);
    krs = add(krs, d_kr);
    krs = add(krs, scalar_mul2(ar, sc.l, bs1, rc.l, 8));   // s*Ar + r*Bs1 on one doubling chain
    bs2 = add(add(bs2, fixed_base_mul<F2>(pk->delta2_tab, sc.l)), host_load_affine<F2>(pk->beta2.data()));
    char* o = reinterpret_cast<char*>(proof_out);
    host_store_affine<F1>(o, ar);
    host_store_affine<F2>(o + sizeof(Affine<F1>), bs2);
    host_store_affine<F1>(o + sizeof(Affine<F1>) + sizeof(Affine<F2>), krs);
    return GA_OK;
}

// BSB22: commitment_i = <values, Basis_i> (pedersen Commit, prove.go:84) and its proof of knowledge <values, BasisExpSigma_i>
// (ProveKnowledge, prove.go:114) -- the ICICLE prover's two commitment MSM blocks (icicle.go:834-873,904-944) in one call.
template <class C>
static int commit(G16Pk* pk, uint32_t index, const void* values, uint64_t n_values, void* commitment_out, void* pok_out) {
    typedef Fe<typename C::FpP> F1;
    Ctx* ctx = pk->ctx;
    if (index >= pk->ck_len.size() || n_values != pk->ck_len[index]) {
        set_error("commit: commitment %u of %zu, %llu values for a basis of %llu points", index, pk->ck_len.size(),
                  (unsigned long long)n_values, (unsigned long long)(index < pk->ck_len.size() ? pk->ck_len[index] : 0));
        return GA_ERR_INVALID;
    }
    XYZZ<F1> com = xyzz_inf<F1>(), pok = xyzz_inf<F1>();
    if (n_values) {
        void* d_v;
        GA_CHECK(ctx->scratch_get("g16_commit_values", n_values * 32, &d_v));
        GA_HIP_CHECK(hipMemcpyAsync(d_v, values, n_values * 32, hipMemcpyHostToDevice, ctx->stream));
        GA_CHECK((host_msm<C, GA_G1>(ctx, pk->d_ck_basis[index], d_v, n_values, true, &com)));
        GA_CHECK((host_msm<C, GA_G1>(ctx, pk->d_ck_sigma[index], d_v, n_values, true, &pok)));
    }
    host_store_affine<F1>(commitment_out, com);
    host_store_affine<F1>(pok_out, pok);
    return GA_OK;
}

// sum_i challenge^i * poks[i]  (Horner from the last point)
template <class C>
static int fold_pok(const void* poks, uint64_t n, const void* challenge_mont, void* out) {
    typedef Fe<typename C::FpP> F1;
    uint32_t ch[8];
    host_fr_canonical<typename C::FrP>(challenge_mont, ch);
    XYZZ<F1> acc = xyzz_inf<F1>();
    for (uint64_t i = n; i-- > 0;) {
        acc = scalar_mul(acc, ch, 8);
        acc = add(acc, host_load_affine<F1>((const char*)poks + i * sizeof(Affine<F1>)));
    }
    host_store_affine<F1>(out, acc);
    return GA_OK;
}

// ---- compressed point encoding (marshal.go:33-58 via gnark-crypto's encoder [EXT]) ---------
template <class P>
static bool lex_largest(const Fe<P>& y_mont) {
    // y > (p-1)/2 on the canonical value
    Fe<P> y = from_mont(y_mont);
    uint32_t half[P::N];
    for (int i = 0; i < P::N; i++) half[i] = (P::MOD[i] >> 1) | (i + 1 < P::N ? (P::MOD[i + 1] << 31) : 0);
    for (int i = P::N - 1; i >= 0; i--) {
        if (y.l[i] > half[i]) return true;
        if (y.l[i] < half[i]) return false;
    }
    return false;
}
template <class P>
static void be_bytes(const Fe<P>& x_mont, uint8_t* out) {
    Fe<P> x = from_mont(x_mont);
    for (int i = 0; i < P::N; i++) {
        uint32_t v = x.l[P::N - 1 - i];
        out[4 * i] = v >> 24;
        out[4 * i + 1] = v >> 16;
        out[4 * i + 2] = v >> 8;
        out[4 * i + 3] = v;
    }
}
template <class C>
static void flag_bytes(uint8_t* out, bool inf, bool largest) {
    if (C::ID == GA_BN254) out[0] |= inf ? 0x40 : (largest ? 0xC0 : 0x80);
    else out[0] |= inf ? 0xC0 : (0x80 | (largest ? 0x20 : 0));
}
template <class C>
static size_t compress_g1(const void* aff, uint8_t* out) {
    typedef typename C::FpP P;
    Affine<Fe<P>> a;
    memcpy(&a, aff, sizeof(a));
    const size_t nb = P::N * 4;
    memset(out, 0, nb);
    if (is_inf(a)) {
        flag_bytes<C>(out, true, false);
        return nb;
    }
    be_bytes<P>(a.x, out);
    flag_bytes<C>(out, false, lex_largest<P>(a.y));
    return nb;
}
template <class C>
static size_t compress_g2(const void* aff, uint8_t* out) {
    typedef typename C::FpP P;
    Affine<Fe2<P>> a;
    memcpy(&a, aff, sizeof(a));
    const size_t nb = P::N * 4;
    memset(out, 0, 2 * nb);
    if (is_inf(a)) {
        flag_bytes<C>(out, true, false);
        return 2 * nb;
    }
    be_bytes<P>(a.x.c1, out);        // A1 || A0
    be_bytes<P>(a.x.c0, out + nb);
    bool largest = is_zero(a.y.c1) ? lex_largest<P>(a.y.c0) : lex_largest<P>(a.y.c1);
    flag_bytes<C>(out, false, largest);
    return 2 * nb;
}

// uncompressed encodings (curve.RawEncoding(), Proof.WriteRawTo marshal.go:25-30): x | y big-endian, G2 coordinates A1 | A0;
// infinity = flag byte 0x40 followed by zeros
template <class C>
static size_t raw_g1(const void* aff, uint8_t* out) {
    typedef typename C::FpP P;
    Affine<Fe<P>> a;
    memcpy(&a, aff, sizeof(a));
    const size_t nb = P::N * 4;
    memset(out, 0, 2 * nb);
    if (is_inf(a)) {
        out[0] = 0x40;
        return 2 * nb;
    }
    be_bytes<P>(a.x, out);
    be_bytes<P>(a.y, out + nb);
    return 2 * nb;
}
template <class C>
static size_t raw_g2(const void* aff, uint8_t* out) {
    typedef typename C::FpP P;
    Affine<Fe2<P>> a;
    memcpy(&a, aff, sizeof(a));
    const size_t nb = P::N * 4;
    memset(out, 0, 4 * nb);
    if (is_inf(a)) {
        out[0] = 0x40;
        return 4 * nb;
    }
    be_bytes<P>(a.x.c1, out);
    be_bytes<P>(a.x.c0, out + nb);
    be_bytes<P>(a.y.c1, out + 2 * nb);
    be_bytes<P>(a.y.c0, out + 3 * nb);
    return 4 * nb;
}

template <class C>
static int marshal(const void* proof, const void* commitments, uint32_t ncom, const void* pok, uint8_t* out, size_t cap, size_t* len,
                   bool raw = false) {
    typedef Fe<typename C::FpP> F1;
    typedef Fe2<typename C::FpP> F2;
    const size_t nb = C::FpP::N * 4;
    const size_t need = (raw ? 2 : 1) * (nb + 2 * nb + nb + (size_t)ncom * nb + nb) + 4;
    if (cap < need) {
        set_error("proof marshal: buffer too small (%zu < %zu)", cap, need);
        return GA_ERR_INVALID;
    }
    const char* p = reinterpret_cast<const char*>(proof);
    size_t o = 0;
    auto g1 = [&](const void* a, uint8_t* dst) { return raw ? raw_g1<C>(a, dst) : compress_g1<C>(a, dst); };
    o += g1(p, out + o);
    o += raw ? raw_g2<C>(p + sizeof(Affine<F1>), out + o) : compress_g2<C>(p + sizeof(Affine<F1>), out + o);
    o += g1(p + sizeof(Affine<F1>) + sizeof(Affine<F2>), out + o);
    out[o] = ncom >> 24;   // uint32 big-endian number of commitments (the slice encoder's length prefix)
    out[o + 1] = ncom >> 16;
    out[o + 2] = ncom >> 8;
    out[o + 3] = ncom;
    o += 4;
    for (uint32_t i = 0; i < ncom; i++) o += g1((const char*)commitments + i * sizeof(Affine<F1>), out + o);
    Affine<F1> inf;
    memset(&inf, 0, sizeof(inf));
    o += g1(pok ? pok : &inf, out + o);   // CommitmentPok (infinity without commitments)
    *len = o;
    return GA_OK;
}

// One proof over several devices from ONE process: keys[i] = shard i of n of the same proving key, each in a context on its own
// device.  One host thread per device for the MSMs plus one for the H side (second lane of the context); the three chains of computeH
// run on the first three devices beside their witness MSMs (with n >= 3 every device uploads 1/n of A, B, C over its own PCIe link and
// forwards the rows to the chain owners over xGMI), b and c travel to device 0 (hipMemcpyPeerAsync), device 0 finishes h and sends
// every device its slice; partial sums are added on the host.  With n = 1 this is ga_g16_prove.
struct MultiShared {
    std::mutex mu;
    std::condition_variable cv;
    int arrived[4] = {0, 0, 0, 0};
    bool failed = false;
    std::string err;
    uint32_t n = 0;
    void fail(const char* msg) {
        std::lock_guard<std::mutex> g(mu);
        if (!failed) {
            failed = true;
            err = msg;
        }
        cv.notify_all();
    }
    // all n threads meet here; returns false when some thread failed (everyone then unwinds)
    bool barrier(int k) {
        std::unique_lock<std::mutex> g(mu);
        arrived[k]++;
        cv.notify_all();
        cv.wait(g, [&] { return failed || arrived[k] == (int)n; });
        return !failed;
    }
};

template <class C>
static int prove_multi(G16Pk* const* pks, uint32_t n, const void* w, const void* a, const void* b, const void* c, uint64_t n_constraints,
                       uint64_t nb_public, const void* r, const void* s, void* proof_out) {
    typedef Fe<typename C::FpP> F1;
    typedef Fe2<typename C::FpP> F2;
    struct Part {
        XYZZ<F1> ar, bs1, k, z;
        XYZZ<F2> bs2;
    };
    std::vector<Part> parts(n);
    const uint64_t N = pks[0]->n;
    // which device runs which chain: a on 0, b on 1 (or 0), c on 2 (or 0)
    const uint32_t owner[3] = {0, n >= 2 ? 1u : 0u, n >= 3 ? 2u : 0u};
    const void* src[3] = {a, b, c};
    static const char* const names[3] = {"h_a", "h_b", "h_c"};
    void* chain_buf[3] = {nullptr, nullptr, nullptr};   // on the owner's device
    void* dev0_buf[3] = {nullptr, nullptr, nullptr};     // on device 0
    std::vector<void*> h_slice(n, nullptr);
    MultiShared sh, hs;   // sh: the device workers; hs: their H-side helper threads
    sh.n = n;
    hs.n = n;
    const bool sliced = n >= 3;
    const uint64_t cshare = (n_constraints + n - 1) / n;
    auto worker = [&](uint32_t t) {
        G16Pk* pk = pks[t];
        Ctx* ctx = pk->ctx;
        SlotLease slot(ctx);
        std::lock_guard<std::mutex> g(ctx->mu);   // one proof at a time per device (icicle.go:821-823)
        ctx->tun.read_env();
        auto bail = [&](const char* what) {   // releases the workers AND the H-side helpers waiting on their barriers
            const std::string m = std::string(what) + ": " + get_error();
            sh.fail(m.c_str());
            hs.fail(m.c_str());
        };
        bool ok = hipSetDevice(ctx->device) == hipSuccess;
        if (!ok) set_error("hipSetDevice(%d) failed", ctx->device);
        // buffers first, so that peers can address them after barrier 0
        for (int k = 0; ok && k < 3; k++) {
            if (owner[k] == t) ok = ctx->scratch_get(slot.name(names[k]).c_str(), N * 32, &chain_buf[k]) == GA_OK;
            if (t == 0 && ok) ok = ctx->scratch_get(slot.name(names[k]).c_str(), N * 32, &dev0_buf[k]) == GA_OK;
        }
        // a base-range shard receives its slice of h, a window shard all of it
        if (ok && pk->len_z) ok = t == 0 || ctx->scratch_get("h_slice", pk->len_z * 32, &h_slice[t]) == GA_OK;
        if (!ok) bail("multi-device prove: buffers");
        if (!sh.barrier(0)) return;
        if (witness_upload(pk, slot, w, nb_public) != GA_OK) {
            bail("multi-device prove: uploading W");
            ok = false;
        }
        // The H side of this device on a helper thread, on the context's second lane (own stream and scratch), BESIDE the witness
        // MSMs: uploads, the chain(s) this device owns, the hop of b / c to device 0.  With three or more devices no PCIe link
        // carries a whole vector: every device uploads rows [t*cshare, ...) of A, B and C over its own link into a staging buffer
        // and forwards them to the chain owners over xGMI (hipMemcpyPeerAsync); the owners start when all pieces have landed.
        bool owns = false;
        for (int k = 0; k < 3; k++) owns = owns || owner[k] == t;
        int h_rc = GA_OK;
        std::string h_err;
        std::thread helper;
        if (ok && (owns || sliced)) {
            helper = std::thread([&, t]() {
                auto hfail = [&](const char* what) {
                    h_rc = GA_ERR_HIP;
                    h_err = std::string(what) + ": " + get_error();
                    hs.fail(h_err.c_str());
                };
                if (hipSetDevice(ctx->device) != hipSuccess) {
                    set_error("hipSetDevice(%d) failed", ctx->device);
                    return hfail("multi-device prove (H side)");
                }
                std::lock_guard<std::mutex> l1(ctx->lane_mu[1]);
                LaneScope lane(1);
                hipStream_t st = ctx->work_stream();
                bool hok = true;
                if (sliced) {
                    const uint64_t lo = std::min<uint64_t>((uint64_t)t * cshare, n_constraints), hi = std::min<uint64_t>(lo + cshare, n_constraints);
                    void* stage = nullptr;
                    hok = ctx->scratch_get("h_stage", 3 * cshare * 32 + 32, &stage) == GA_OK;
                    for (int k = 0; hok && k < 3; k++) {
                        char* mine = (char*)stage + (uint64_t)k * cshare * 32;
                        if (hi > lo) {
                            hok = hipMemcpyAsync(mine, (const char*)src[k] + lo * 32, (hi - lo) * 32, hipMemcpyHostToDevice, st) == hipSuccess;
                            if (hok)
                                hok = hipMemcpyPeerAsync((char*)chain_buf[k] + lo * 32, pks[owner[k]]->ctx->device, mine, ctx->device, (hi - lo) * 32,
                                                         st) == hipSuccess;
                        }
                        if (hok && owner[k] == t && N > n_constraints)   // computeH pads to the domain size (prove.go:356-359)
                            hok = hipMemsetAsync((char*)chain_buf[k] + n_constraints * 32, 0, (N - n_constraints) * 32, st) == hipSuccess;
                    }
                    if (hok) hok = hipStreamSynchronize(st) == hipSuccess;
                    if (!hok) {
                        if (!get_error()[0]) set_error("HIP error while uploading / forwarding the rows of A, B, C");
                        return hfail("multi-device prove: uploading A, B, C");
                    }
                    if (!hs.barrier(0)) return;   // every device's rows have landed on the chain owners
                } else {
                    for (int k = 0; hok && k < 3; k++)
                        if (owner[k] == t) hok = h_upload(pk, src[k], n_constraints, chain_buf[k], st) == GA_OK;
                    if (!hok) return hfail("multi-device prove: uploading A, B, C");
                }
                for (int k = 0; hok && k < 3; k++)
                    if (owner[k] == t) {
                        hok = ntt_domain_h_chain<C>(pk->dom, chain_buf[k]) == GA_OK;
                        if (hok && t != 0)
                            hok = hipMemcpyPeerAsync(dev0_buf[k], pks[0]->ctx->device, chain_buf[k], ctx->device, N * 32, st) == hipSuccess;
                    }
                if (hok) hok = hipStreamSynchronize(st) == hipSuccess;
                if (!hok) {
                    if (!get_error()[0]) set_error("HIP error in the computeH chain");
                    return hfail("multi-device prove: computeH chain");
                }
            });
        }
        ThreadJoiner joiner{helper};
        if (ok) {
            WitnessShared wsh;
            bool did_k = false;
            if (hipEventCreateWithFlags(&wsh.w_ev, hipEventDisableTiming) != hipSuccess ||
                witness_msms<C>(pk, slot, nb_public, wsh, &parts[t].ar, &parts[t].bs1, &parts[t].k, &parts[t].bs2, &did_k) != GA_OK) {
                bail("multi-device prove: witness MSMs");
                ok = false;
            }
        }
        if (helper.joinable()) helper.join();
        if (h_rc != GA_OK) {
            sh.fail(h_err.c_str());
            ok = false;
        }
        if (!sh.barrier(1)) return;
        if (t == 0) {
            ok = ntt_domain_h_combine<C>(pk->dom, dev0_buf[0], dev0_buf[1], dev0_buf[2]) == GA_OK;
            for (uint32_t q = 1; ok && q < n; q++)   // every device gets its slice of h[:n-1]
                if (pks[q]->len_z)
                    ok = hipMemcpyPeerAsync(h_slice[q], pks[q]->ctx->device, (const char*)dev0_buf[0] + pks[q]->off_z * 32, ctx->device,
                                            pks[q]->len_z * 32, ctx->stream) == hipSuccess;
            if (ok) ok = hipStreamSynchronize(ctx->stream) == hipSuccess;
            if (!ok) {
                if (!get_error()[0]) set_error("HIP error while finishing / scattering h");
                bail("multi-device prove: h");
            }
            h_slice[0] = (char*)dev0_buf[0] + pk->off_z * 32;
        }
        if (!sh.barrier(2)) return;
        if (z_msm<C>(pk, h_slice[t], &parts[t].z) != GA_OK) bail("multi-device prove: Z MSM");
        sh.barrier(3);
    };
    std::vector<std::thread> threads;
    for (uint32_t t = 1; t < n; t++) threads.emplace_back(worker, t);
    worker(0);
    for (auto& th : threads) th.join();
    if (sh.failed) {
        set_error("%s", sh.err.c_str());
        return GA_ERR_HIP;
    }
    XYZZ<F1> ar = xyzz_inf<F1>(), bs1 = xyzz_inf<F1>(), krs = xyzz_inf<F1>();
    XYZZ<F2> bs2 = xyzz_inf<F2>();
    for (uint32_t t = 0; t < n; t++) {
        ar = add(ar, parts[t].ar);
        bs1 = add(bs1, parts[t].bs1);
        krs = add(krs, add(parts[t].k, parts[t].z));
        bs2 = add(bs2, parts[t].bs2);
    }
    return finish<C>(pks[0], ar, bs1, krs, bs2, r, s, proof_out);
}

}  // namespace ga

using namespace ga;

extern "C" {

int ga_g16_pk_create(ga_ctx* h, const ga_g16_key* key, ga_g16_pk** out) try {
    GA_ABI_ENTRY();
    Ctx* ctx = reinterpret_cast<Ctx*>(h);
    if (!ctx || !key || !out) {
        set_error("ga_g16_pk_create: null argument");
        return GA_ERR_INVALID;
    }
    CtxLock g(ctx);
    G16Pk* pk = nullptr;
    GA_CHECK(pk_create_from_struct(ctx, key, &pk));
    *out = reinterpret_cast<ga_g16_pk*>(pk);
    return GA_OK;
} GA_ABI_CATCH

int ga_g16_builder_create(ga_ctx* h, int curve, uint64_t domain_cardinality, uint64_t nb_wires, uint32_t shard_index,
                          uint32_t shard_count, ga_g16_builder** out) try {
    GA_ABI_ENTRY();
    Ctx* ctx = reinterpret_cast<Ctx*>(h);
    if (!ctx || !out || (curve != GA_BN254 && curve != GA_BLS12_381) || domain_cardinality == 0) {
        set_error("ga_g16_builder_create: bad argument");
        return GA_ERR_INVALID;
    }
    if (shard_count == 0) shard_count = 1;
    if (shard_index >= shard_count) {
        set_error("proving key: shard_index %u >= shard_count %u", shard_index, shard_count);
        return GA_ERR_INVALID;
    }
    G16Stage* st = new G16Stage();
    st->ctx = ctx;
    st->curve = curve;
    st->n = domain_cardinality;
    st->nb_wires = nb_wires;
    st->shard_index = shard_index;
    st->shard_count = shard_count;
    *out = reinterpret_cast<ga_g16_builder*>(st);
    return GA_OK;
} GA_ABI_CATCH

#define GA_STAGE(b)                                     \
    G16Stage* st = reinterpret_cast<G16Stage*>(b);      \
    if (!st) {                                          \
        set_error("ga_g16_builder: null builder");      \
        return GA_ERR_INVALID;                          \
    }                                                   \
    CtxLock g(st->ctx)

int ga_g16_builder_reserve(ga_g16_builder* b, int which, uint64_t total_len) try {
    GA_ABI_ENTRY();
    GA_STAGE(b);
    return stage_reserve(st, which, total_len);
} GA_ABI_CATCH

int ga_g16_builder_append(ga_g16_builder* b, int which, const void* points, uint64_t count) try {
    GA_ABI_ENTRY();
    GA_STAGE(b);
    if (count && !points) {
        // skipping is allowed for points that are none of this shard's business
        if (which < 0 || which >= GA_KEY_NB_VECTORS || !st->v[which].reserved) {
            set_error("ga_g16_builder_append: skip on vector %d before ga_g16_builder_reserve", which);
            return GA_ERR_STATE;
        }
        G16Stage::Vec& x = st->v[which];
        const bool outside = x.seen + count <= x.lo || x.seen >= x.lo + x.cnt;
        if (!outside || x.seen + count > x.total) {
            set_error("ga_g16_builder_append: null pointer for points [%llu, %llu) of vector %d, of which this shard keeps [%llu, %llu)",
                      (unsigned long long)x.seen, (unsigned long long)(x.seen + count), which, (unsigned long long)x.lo, (unsigned long long)(x.lo + x.cnt));
            return GA_ERR_INVALID;
        }
        x.seen += count;
        return GA_OK;
    }
    return stage_append(st, which, points, count);
} GA_ABI_CATCH

int ga_g16_builder_set_point(ga_g16_builder* b, int which, const void* affine) try {
    GA_ABI_ENTRY();
    GA_STAGE(b);
    return stage_set_point(st, which, affine);
} GA_ABI_CATCH

int ga_g16_builder_set_infinity(ga_g16_builder* b, int which, const uint8_t* mask, uint64_t nb_wires) try {
    GA_ABI_ENTRY();
    GA_STAGE(b);
    if ((which != 0 && which != 1) || !mask || nb_wires != st->nb_wires) {
        set_error("ga_g16_builder_set_infinity: which must be 0/1 and the mask must have nbWires = %llu entries", (unsigned long long)st->nb_wires);
        return GA_ERR_INVALID;
    }
    if (st->lists[which].job.joinable()) st->lists[which].job.join();   // (a mask set twice: the list of the old one must not outlive it)
    st->inf[which].assign(mask, mask + nb_wires);
    st->have_inf[which] = true;
    st->start_list(which);   // (beside whatever the caller appends next)
    return GA_OK;
} GA_ABI_CATCH

int ga_g16_builder_add_commitment_key(ga_g16_builder* b, const void* basis, const void* sigma, uint64_t len) try {
    GA_ABI_ENTRY();
    GA_STAGE(b);
    return stage_add_commitment_key(st, basis, sigma, len);
} GA_ABI_CATCH

int ga_g16_builder_set_k_remove(ga_g16_builder* b, const uint64_t* ids, uint64_t len) try {
    GA_ABI_ENTRY();
    GA_STAGE(b);
    if (len && !ids) {
        set_error("ga_g16_builder_set_k_remove: null pointer");
        return GA_ERR_INVALID;
    }
    if (len > st->nb_wires) {
        set_error("ga_g16_builder_set_k_remove: %llu removed wires for %llu wires", (unsigned long long)len, (unsigned long long)st->nb_wires);
        return GA_ERR_INVALID;
    }
    st->k_remove.assign(ids, ids + len);
    return GA_OK;
} GA_ABI_CATCH

int ga_g16_builder_set_window_shard(ga_g16_builder* b, uint32_t index, uint32_t count) try {
    GA_ABI_ENTRY();
    GA_STAGE(b);
    if (count == 0 || index >= count) {
        set_error("ga_g16_builder_set_window_shard: index %u must be below count %u", index, count);
        return GA_ERR_INVALID;
    }
    st->win_index = index;
    st->win_count = count;
    return GA_OK;
} GA_ABI_CATCH

int ga_g16_builder_finish(ga_g16_builder* b, int32_t precompute, ga_g16_pk** out) try {
    GA_ABI_ENTRY();
    G16Stage* st = reinterpret_cast<G16Stage*>(b);
    if (!st || !out) {
        set_error("ga_g16_builder_finish: null argument");
        return GA_ERR_INVALID;
    }
    int rc;
    {
        CtxLock g(st->ctx);
        G16Pk* pk = nullptr;
        rc = GA_ERR_INVALID;
        try {
            if (st->curve == GA_BN254) rc = stage_finish<Bn254>(st, precompute, &pk);
            else if (st->curve == GA_BLS12_381) rc = stage_finish<Bls12381>(st, precompute, &pk);
        } catch (const std::exception& e) {   // (host allocations sized by the key: no exception crosses the C ABI)
            set_error("ga_g16_builder_finish: %s", e.what());
            rc = GA_ERR_NOMEM;
        }
        if (rc == GA_OK) *out = reinterpret_cast<ga_g16_pk*>(pk);
    }
    delete st;   // consumed either way: a failed finish leaves nothing half-built behind
    return rc;
} GA_ABI_CATCH

void ga_g16_builder_destroy(ga_g16_builder* b) try {
    GA_ABI_ENTRY();
    G16Stage* st = reinterpret_cast<G16Stage*>(b);
    if (!st) return;
    Ctx* ctx = st->ctx;
    CtxLock g(ctx);
    hipStreamSynchronize(ctx->stream);
    delete st;
} GA_ABI_CATCH_VOID

static void pk_destroy_impl(G16Pk* pk) {
    if (!pk) return;
    {   // wait for every entry point still working on this key (provers on other lanes, epilogues outside the device lock)
        std::unique_lock<std::mutex> u(pk->use_mu);
        pk->dying = true;
        pk->use_cv.wait(u, [&] { return pk->users == 0; });
    }
    CtxLock g(pk->ctx);
    for (int l = 0; l < GA_NUM_LANES; l++) hipStreamSynchronize(pk->ctx->lane_stream[l]);
    pk_free(pk);
}

void ga_g16_pk_destroy(ga_g16_pk* p) try {
    GA_ABI_ENTRY();
    pk_destroy_impl(reinterpret_cast<G16Pk*>(p));
} GA_ABI_CATCH_VOID

static int g16_prove_impl(ga_g16_pk* p, const void* w, const void* a, const void* b, const void* c, uint64_t n_constraints,
                 uint64_t nb_public, const void* r, const void* s, void* proof_out) {
    G16Pk* pk = reinterpret_cast<G16Pk*>(p);
    if (!pk || !w || !a || !b || !c || !r || !s || !proof_out) {
        set_error("ga_g16_prove: null argument");
        return GA_ERR_INVALID;
    }
    GA_PK_USE(pk, "ga_g16_prove");
    if (pk->shard_count != 1 || pk->win_count != 1) {
        set_error("ga_g16_prove: this key holds one share of a sharded key (base range %u/%u, windows %u/%u); use ga_g16_prove_multi or "
                  "ga_g16_prove_partial + ga_g16_finish", pk->shard_index, pk->shard_count, pk->win_index, pk->win_count);
        return GA_ERR_STATE;
    }
    // Two callers may be inside at once (two goroutines proving on one device).  The first holds the device lock and works on
    // lanes 0/1 (witness MSMs / H side, prove_partial).  The second finds the device busy and computes its proof on lanes 2/3 --
    // own streams, own scratch namespaces -- so the two proofs run CONCURRENTLY: uploads hide behind the other proof's kernels
    // and the kernels interleave.  When lane 2 is taken as well, or while the profiler records stages, or with GA_G16_LANES=1,
    // the caller stages its solution in its input slot and queues for the device.  The host epilogue always runs outside the
    // device lock.  ga_g16_lane_stats reports how the calls of a context were scheduled.
    // A lane-2 proof that cannot get its scratch (precompute = 0 fills HBM with tables beside ONE caller's scratch: at 2^26 a second
    // caller's 77 GiB are not there) gives back what lanes 2/3 hold and queues for the device like a third caller would -- a proof
    // is slower then, never failed.
    Ctx* ctx = pk->ctx;
    SlotLease slot(ctx);
    std::unique_lock<std::mutex> dev(ctx->mu, std::try_to_lock);
    std::unique_lock<std::mutex> lane2(ctx->lane_mu[2], std::defer_lock);
    hipSetDevice(ctx->device);
    for (int attempt = 0;; attempt++) {
        bool preloaded = false;
        int lane = 0;
        if (!dev.owns_lock()) {
            if (attempt == 0 && !ctx->profiling && ctx->tun.g16_lanes > 1 && lane2.try_lock()) {
                lane = 2;
                ctx->stat_lane2++;
            } else {
                GA_CHECK(preload_solution(pk, slot, w, a, b, c, n_constraints, nb_public));
                preloaded = true;
                dev.lock();
                ctx->stat_queued++;
            }
        }
        if (lane == 0 && !preloaded) ctx->stat_lane0++;
        int partial_rc = GA_OK;
        {
            LaneScope on_lane(lane);
            if (lane == 0) ctx->tun.read_env();
            GA_DISPATCH_CURVE(pk->curve, {
                XYZZ<Fe<typename C::FpP>> ar, bs1, krs;
                XYZZ<Fe2<typename C::FpP>> bs2;
                partial_rc = prove_partial<C>(pk, slot, preloaded, w, a, b, c, n_constraints, nb_public, &ar, &bs1, &krs, &bs2);
                if (partial_rc == GA_OK) {
                    const bool profiling = ctx->profiling;
                    if (lane == 0 && !profiling) dev.unlock();   // the stage list of the profiler is guarded by the device lock
                    if (lane == 2) lane2.unlock();
                    return finish<C>(pk, ar, bs1, krs, bs2, r, s, proof_out);
                }
            });
        }
        if (partial_rc != GA_ERR_NOMEM || lane != 2) return partial_rc;
        // (everything prove_partial started on lanes 2/3 has joined; hipFree synchronises with what their streams still hold)
        ctx->scratch_free_lanes(2);
        ctx->stat_lane2--;
        lane2.unlock();
    }
}

// out[0..3] = ga_g16_prove calls of this context that ran on lanes 0/1, on lanes 2/3 beside another proof, that staged their
// inputs and queued for the device, and proofs whose H side ran on a partner lane (any entry point); out[4..5] = device bytes
// of scratch held by lanes 0/1 and by lanes 2/3
int ga_g16_lane_stats(ga_ctx* h, uint64_t* out6) try {
    GA_ABI_ENTRY();
    Ctx* ctx = reinterpret_cast<Ctx*>(h);
    if (!ctx || !out6) {
        set_error("ga_g16_lane_stats: null argument");
        return GA_ERR_INVALID;
    }
    out6[0] = ctx->stat_lane0;
    out6[1] = ctx->stat_lane2;
    out6[2] = ctx->stat_queued;
    out6[3] = ctx->stat_split;
    out6[4] = out6[5] = 0;
    std::lock_guard<std::mutex> g(ctx->scratch_mu);
    for (const auto& kv : ctx->scratch) {
        const size_t at = kv.first.rfind('@');
        const int lane = at == std::string::npos ? 0 : atoi(kv.first.c_str() + at + 1);
        out6[lane < 2 ? 4 : 5] += kv.second.second;
    }
    return GA_OK;
} GA_ABI_CATCH

static int g16_prove_partial_impl(ga_g16_pk* p, const void* w, const void* a, const void* b, const void* c, uint64_t n_constraints,
                         uint64_t nb_public, void* partials_out) {
    G16Pk* pk = reinterpret_cast<G16Pk*>(p);
    if (!pk || !w || !a || !b || !c || !partials_out) {
        set_error("ga_g16_prove_partial: null argument");
        return GA_ERR_INVALID;
    }
    GA_PK_USE(pk, "ga_g16_prove_partial");
    SlotLease slot(pk->ctx);   // always slot first, device lock second (ga_g16_prove's order)
    CtxLock g(pk->ctx);
    GA_DISPATCH_CURVE(pk->curve, {
        typedef Fe<typename C::FpP> F1;
        typedef Fe2<typename C::FpP> F2;
        XYZZ<F1> ar, bs1, krs;
        XYZZ<F2> bs2;
        GA_CHECK(prove_partial<C>(pk, slot, false, w, a, b, c, n_constraints, nb_public, &ar, &bs1, &krs, &bs2));
        char* o = reinterpret_cast<char*>(partials_out);
        host_store_jac<F1>(o, ar);
        host_store_jac<F1>(o + sizeof(Jac<F1>), bs1);
        host_store_jac<F1>(o + 2 * sizeof(Jac<F1>), krs);
        host_store_jac<F2>(o + 3 * sizeof(Jac<F1>), bs2);
    });
    return GA_OK;
}

int ga_g16_finish(ga_g16_pk* p, const void* partials_sum, const void* r, const void* s, void* proof_out) try {
    GA_ABI_ENTRY();
    G16Pk* pk = reinterpret_cast<G16Pk*>(p);
    if (!pk || !partials_sum || !r || !s || !proof_out) {
        set_error("ga_g16_finish: null argument");
        return GA_ERR_INVALID;
    }
    GA_PK_USE(pk, "ga_g16_finish");
    GA_DISPATCH_CURVE(pk->curve, {
        typedef Fe<typename C::FpP> F1;
        typedef Fe2<typename C::FpP> F2;
        const char* i = reinterpret_cast<const char*>(partials_sum);
        return finish<C>(pk, host_load_jac<F1>(i), host_load_jac<F1>(i + sizeof(Jac<F1>)), host_load_jac<F1>(i + 2 * sizeof(Jac<F1>)),
                         host_load_jac<F2>(i + 3 * sizeof(Jac<F1>)), r, s, proof_out);
    });
    return GA_OK;
} GA_ABI_CATCH

// ---- pieces of a sharded proof (multi-GPU orchestration by the caller: gnark_amd/multigpu.py over RCCL, or ga_g16_prove_multi) ----
int ga_g16_shard_layout(ga_g16_pk* p, uint64_t* out6) try {
    GA_ABI_ENTRY();
    G16Pk* pk = reinterpret_cast<G16Pk*>(p);
    uint64_t* out8 = out6;
    if (!pk || !out6) {
        set_error("ga_g16_shard_layout: null argument");
        return GA_ERR_INVALID;
    }
    out6[0] = pk->off_z;
    out6[1] = pk->len_z;
    out6[2] = pk->w_lo;
    out6[3] = pk->w_hi;
    out6[4] = pk->n;
    out6[5] = pk->nb_wires;
    out8[6] = pk->win_index;
    out8[7] = pk->win_count;   // (the plain count: which vectors carry tables is ga_g16_table_layout's answer)
    return GA_OK;
} GA_ABI_CATCH

// out2[0]: which vectors carry a window table (bit 0 G1.A, 1 G1.B, 2 G1.Z, 3 G1.K, 4 G2.B); out2[1]: which of those are laid out by
// wire id over the shared witness sort (bit 0 A, 1 B, 3 K, 4 G2.B)
int ga_g16_table_layout(ga_g16_pk* p, uint64_t* out2) try {
    GA_ABI_ENTRY();
    G16Pk* pk = reinterpret_cast<G16Pk*>(p);
    if (!pk || !out2) {
        set_error("ga_g16_table_layout: null argument");
        return GA_ERR_INVALID;
    }
    out2[0] = (uint64_t)pk->tab_a | (uint64_t)pk->tab_b << 1 | (uint64_t)pk->tab_z << 2 | (uint64_t)pk->tab_k << 3 | (uint64_t)pk->tab_b2 << 4;
    out2[1] = (uint64_t)pk->share_a | (uint64_t)pk->share_b << 1 | (uint64_t)pk->share_k << 3 | (uint64_t)pk->share_b2 << 4;
    return GA_OK;
} GA_ABI_CATCH

static int g16_witness_partial_impl(ga_g16_pk* p, const void* w, uint64_t nb_public, void* partials_out) {
    G16Pk* pk = reinterpret_cast<G16Pk*>(p);
    if (!pk || !w || !partials_out) {
        set_error("ga_g16_witness_partial: null argument");
        return GA_ERR_INVALID;
    }
    GA_PK_USE(pk, "ga_g16_witness_partial");
    SlotLease slot(pk->ctx);
    CtxLock g(pk->ctx);
    GA_DISPATCH_CURVE(pk->curve, {
        typedef Fe<typename C::FpP> F1;
        typedef Fe2<typename C::FpP> F2;
        XYZZ<F1> ar, bs1, krs;
        XYZZ<F2> bs2;
        GA_CHECK(witness_upload(pk, slot, w, nb_public));
        WitnessShared sh;
        GA_HIP_CHECK(hipEventCreateWithFlags(&sh.w_ev, hipEventDisableTiming));
        bool did_k = false;
        GA_CHECK(witness_msms<C>(pk, slot, nb_public, sh, &ar, &bs1, &krs, &bs2, &did_k));
        char* o = reinterpret_cast<char*>(partials_out);
        host_store_jac<F1>(o, ar);
        host_store_jac<F1>(o + sizeof(Jac<F1>), bs1);
        host_store_jac<F1>(o + 2 * sizeof(Jac<F1>), krs);
        host_store_jac<F2>(o + 3 * sizeof(Jac<F1>), bs2);
    });
    return GA_OK;
}

int ga_g16_h_chain(ga_g16_pk* p, const void* v, uint64_t n_constraints, void* out_dev) try {
    GA_ABI_ENTRY();
    G16Pk* pk = reinterpret_cast<G16Pk*>(p);
    if (!pk || !v || !out_dev) {
        set_error("ga_g16_h_chain: null argument");
        return GA_ERR_INVALID;
    }
    GA_PK_USE(pk, "ga_g16_h_chain");
    LaneLock g(pk->ctx);   // beside the witness MSMs of the same shard when the caller runs them from another thread
    GA_CHECK(h_upload(pk, v, n_constraints, out_dev, pk->ctx->work_stream()));
    GA_DISPATCH_CURVE(pk->curve, GA_CHECK(ntt_domain_h_chain<C>(pk->dom, out_dev)));
    GA_HIP_CHECK(hipStreamSynchronize(pk->ctx->work_stream()));   // the buffer is handed to another stream / device next
    return GA_OK;
} GA_ABI_CATCH

int ga_g16_h_chain_dev(ga_g16_pk* p, void* buf_dev, uint64_t n_constraints) try {
    GA_ABI_ENTRY();
    G16Pk* pk = reinterpret_cast<G16Pk*>(p);
    if (!pk || !buf_dev) {
        set_error("ga_g16_h_chain_dev: null argument");
        return GA_ERR_INVALID;
    }
    GA_PK_USE(pk, "ga_g16_h_chain_dev");
    if (n_constraints > pk->n) {
        set_error("ga_g16_h_chain_dev: %llu constraints exceed the domain cardinality %llu", (unsigned long long)n_constraints,
                  (unsigned long long)pk->n);
        return GA_ERR_INVALID;
    }
    LaneLock g(pk->ctx);
    hipStream_t st = pk->ctx->work_stream();
    if (pk->n > n_constraints)   // computeH pads to the domain size (prove.go:356-359)
        GA_HIP_CHECK(hipMemsetAsync((char*)buf_dev + n_constraints * 32, 0, (pk->n - n_constraints) * 32, st));
    GA_DISPATCH_CURVE(pk->curve, GA_CHECK(ntt_domain_h_chain<C>(pk->dom, buf_dev)));
    GA_HIP_CHECK(hipStreamSynchronize(st));
    return GA_OK;
} GA_ABI_CATCH

int ga_g16_h_combine(ga_g16_pk* p, void* a_dev, const void* b_dev, const void* c_dev) try {
    GA_ABI_ENTRY();
    G16Pk* pk = reinterpret_cast<G16Pk*>(p);
    if (!pk || !a_dev || !b_dev || !c_dev) {
        set_error("ga_g16_h_combine: null argument");
        return GA_ERR_INVALID;
    }
    GA_PK_USE(pk, "ga_g16_h_combine");
    LaneLock g(pk->ctx);
    GA_DISPATCH_CURVE(pk->curve, GA_CHECK(ntt_domain_h_combine<C>(pk->dom, a_dev, b_dev, c_dev)));
    GA_HIP_CHECK(hipStreamSynchronize(pk->ctx->work_stream()));
    return GA_OK;
} GA_ABI_CATCH

int ga_g16_z_partial(ga_g16_pk* p, const void* h_slice_dev, void* partial_out) try {
    GA_ABI_ENTRY();
    G16Pk* pk = reinterpret_cast<G16Pk*>(p);
    if (!pk || (!h_slice_dev && pk->len_z) || !partial_out) {
        set_error("ga_g16_z_partial: null argument");
        return GA_ERR_INVALID;
    }
    GA_PK_USE(pk, "ga_g16_z_partial");
    CtxLock g(pk->ctx);
    GA_DISPATCH_CURVE(pk->curve, {
        typedef Fe<typename C::FpP> F1;
        XYZZ<F1> z;
        GA_CHECK(z_msm<C>(pk, h_slice_dev, &z));
        host_store_jac<F1>(partial_out, z);
    });
    return GA_OK;
} GA_ABI_CATCH

static int g16_prove_multi_impl(ga_g16_pk* const* keys, uint32_t n, const void* w, const void* a, const void* b, const void* c,
                       uint64_t n_constraints, uint64_t nb_public, const void* r, const void* s, void* proof_out) {
    if (!keys || n == 0 || n > 64 || !w || !a || !b || !c || !r || !s || !proof_out) {
        set_error("ga_g16_prove_multi: null argument or unsupported device count");
        return GA_ERR_INVALID;
    }
    G16Pk* const* pks = reinterpret_cast<G16Pk* const*>(keys);
    for (uint32_t t = 0; t < n; t++) {
        const bool by_range = pks[t] && pks[t]->shard_count == n && pks[t]->shard_index == t && pks[t]->win_count == 1;
        const bool by_window = pks[t] && pks[t]->win_count == n && pks[t]->win_index == t && pks[t]->shard_count == 1;
        if (!pks[t] || pks[t]->curve != pks[0]->curve || pks[t]->n != pks[0]->n || pks[t]->nb_wires != pks[0]->nb_wires ||
            !(n == 1 || by_range || by_window) || (pks[t]->win_count > 1) != (pks[0]->win_count > 1)) {
            set_error("ga_g16_prove_multi: keys[%u] must be shard %u of %u of the same proving key (all by base range or all by windows)", t, t, n);
            return GA_ERR_INVALID;
        }
        for (uint32_t q = 0; q < t; q++)
            if (pks[q]->ctx == pks[t]->ctx) {
                set_error("ga_g16_prove_multi: keys[%u] and keys[%u] share a context; one context per shard", q, t);
                return GA_ERR_INVALID;
            }
    }
    if (n == 1) return ga_g16_prove(keys[0], w, a, b, c, n_constraints, nb_public, r, s, proof_out);
    std::vector<std::unique_ptr<PkUse>> uses;
    for (uint32_t t = 0; t < n; t++) {
        uses.emplace_back(new PkUse(pks[t]));
        if (!uses.back()->ok) {
            set_error("ga_g16_prove_multi: keys[%u] is being destroyed", t);
            return GA_ERR_STATE;
        }
    }
    // One multi-device proof at a time per process: every worker thread holds its device's lock while it waits for the others at
    // the barriers, so two calls over the same devices could each hold one lock the other needs (A holds dev0 and waits for its
    // worker on dev1, B holds dev1 and waits for its worker on dev0).  A sharded proof occupies all its devices anyway.
    static std::mutex multi_mu;
    std::lock_guard<std::mutex> multi_guard(multi_mu);
    for (uint32_t t = 0; t < n; t++)   // peer access both ways between device 0 and the others (errors = already enabled / same device)
        for (uint32_t q = 0; q < n; q++)
            if (q != t && (t == 0 || q == 0) && pks[t]->ctx->device != pks[q]->ctx->device) {
                hipSetDevice(pks[t]->ctx->device);
                (void)hipDeviceEnablePeerAccess(pks[q]->ctx->device, 0);
                (void)hipGetLastError();
            }
    GA_DISPATCH_CURVE(pks[0]->curve, return (prove_multi<C>(pks, n, w, a, b, c, n_constraints, nb_public, r, s, proof_out)));
    return GA_OK;
}

// ---- key files and proof bytes ------------------------------------------------------------------------------------------------
static int pk_read_any(ga_ctx* h, int curve, ByteSource& src, int32_t precompute, uint32_t shard_index, uint32_t shard_count,
                       const uint64_t* k_remove, uint64_t len_k_remove, ga_g16_pk** out, uint64_t* bytes_read) {
    Ctx* ctx = reinterpret_cast<Ctx*>(h);
    if (!ctx || !out || (len_k_remove && !k_remove)) {
        set_error("ga_g16_pk_read: null argument");
        return GA_ERR_INVALID;
    }
    CtxLock g(ctx);
    G16Pk* pk = nullptr;
    GA_DISPATCH_CURVE(curve, GA_CHECK(pk_read<C>(ctx, src, precompute, shard_index, shard_count, k_remove, len_k_remove, &pk)));
    *out = reinterpret_cast<ga_g16_pk*>(pk);
    if (bytes_read) *bytes_read = src.consumed;
    return GA_OK;
}

int ga_g16_pk_read_mem(ga_ctx* h, int curve, const uint8_t* data, size_t len, int32_t precompute, uint32_t shard_index, uint32_t shard_count,
                       const uint64_t* k_remove, uint64_t len_k_remove, ga_g16_pk** out, uint64_t* bytes_read) try {
    GA_ABI_ENTRY();
    if (!data) {
        set_error("ga_g16_pk_read_mem: null data");
        return GA_ERR_INVALID;
    }
    ByteSource src;
    src.mem = data;
    src.mem_len = len;
    return pk_read_any(h, curve, src, precompute, shard_index, shard_count, k_remove, len_k_remove, out, bytes_read);
} GA_ABI_CATCH

int ga_g16_pk_read_fd(ga_ctx* h, int curve, int fd, int32_t precompute, uint32_t shard_index, uint32_t shard_count, const uint64_t* k_remove,
                      uint64_t len_k_remove, ga_g16_pk** out, uint64_t* bytes_read) try {
    GA_ABI_ENTRY();
    if (fd < 0) {
        set_error("ga_g16_pk_read_fd: bad file descriptor");
        return GA_ERR_INVALID;
    }
    ByteSource src;
    src.fd = fd;
    return pk_read_any(h, curve, src, precompute, shard_index, shard_count, k_remove, len_k_remove, out, bytes_read);
} GA_ABI_CATCH

int ga_g16_key_write_fd(ga_ctx* h, const ga_g16_key* key, int format, int fd, uint64_t* bytes_written) try {
    GA_ABI_ENTRY();
    Ctx* ctx = reinterpret_cast<Ctx*>(h);
    if (!ctx || !key || fd < 0 || format < GA_KEY_FORMAT_COMPRESSED || format > GA_KEY_FORMAT_DUMP) {
        set_error("ga_g16_key_write_fd: bad argument");
        return GA_ERR_INVALID;
    }
    if (!key->g1_alpha || !key->g1_beta || !key->g1_delta || !key->g2_beta || !key->g2_delta || !key->infinity_a || !key->infinity_b ||
        (key->len_a && !key->g1_a) || (key->len_b && !key->g1_b) || (key->len_z && !key->g1_z) || (key->len_k && !key->g1_k) ||
        (key->len_b2 && !key->g2_b) || (key->nb_commitments && (!key->ck_basis || !key->ck_basis_exp_sigma || !key->ck_len))) {
        set_error("ga_g16_key_write_fd: null pointer inside ga_g16_key");
        return GA_ERR_INVALID;
    }
    CtxLock g(ctx);
    ByteSink dst;
    dst.fd = fd;
    GA_DISPATCH_CURVE(key->curve, GA_CHECK(key_write<C>(ctx, key, format, dst)));
    if (bytes_written) *bytes_written = dst.written;
    return GA_OK;
} GA_ABI_CATCH

int ga_g16_proof_unmarshal(int curve, const uint8_t* data, size_t len, void* proof_out, void* commitments_out, uint32_t max_commitments,
                           uint32_t* n_commitments, void* pok_out, size_t* consumed) try {
    GA_ABI_ENTRY();
    if (!data || !proof_out) {
        set_error("ga_g16_proof_unmarshal: null argument");
        return GA_ERR_INVALID;
    }
    GA_DISPATCH_CURVE(curve, return proof_unmarshal<C>(data, len, proof_out, commitments_out, max_commitments, n_commitments, pok_out, consumed));
    return GA_OK;
} GA_ABI_CATCH

int ga_point_unmarshal(int curve, int group, const uint8_t* data, size_t len, void* affine_out, size_t* consumed) try {
    GA_ABI_ENTRY();
    if (!data || !affine_out) {
        set_error("ga_point_unmarshal: null argument");
        return GA_ERR_INVALID;
    }
    ByteSource src;
    src.mem = data;
    src.mem_len = len;
    std::vector<uint8_t> img;
    GA_DISPATCH_CURVE(curve, {
        if (group == GA_G1) GA_CHECK((read_header_point<C, GA_G1>(src, &img)));
        else if (group == GA_G2) GA_CHECK((read_header_point<C, GA_G2>(src, &img)));
        else {
            set_error("unknown group id %d", group);
            return GA_ERR_INVALID;
        }
    });
    memcpy(affine_out, img.data(), img.size());
    if (consumed) *consumed = src.mem_pos;
    return GA_OK;
} GA_ABI_CATCH

int ga_g16_proof_marshal(int curve, const void* proof, uint8_t* out, size_t cap, size_t* len) try {
    GA_ABI_ENTRY();
    if (!proof || !out || !len) {
        set_error("ga_g16_proof_marshal: null argument");
        return GA_ERR_INVALID;
    }
    GA_DISPATCH_CURVE(curve, return marshal<C>(proof, nullptr, 0, nullptr, out, cap, len));
    return GA_OK;
} GA_ABI_CATCH

int ga_g16_proof_marshal_bsb22(int curve, const void* proof, const void* commitments, uint32_t n, const void* pok, uint8_t* out,
                               size_t cap, size_t* len) try {
    GA_ABI_ENTRY();
    if (!proof || !out || !len || (n && !commitments)) {
        set_error("ga_g16_proof_marshal_bsb22: null argument");
        return GA_ERR_INVALID;
    }
    GA_DISPATCH_CURVE(curve, return marshal<C>(proof, commitments, n, pok, out, cap, len));
    return GA_OK;
} GA_ABI_CATCH

int ga_g16_proof_marshal_raw(int curve, const void* proof, const void* commitments, uint32_t n, const void* pok, uint8_t* out,
                             size_t cap, size_t* len) try {
    GA_ABI_ENTRY();
    if (!proof || !out || !len || (n && !commitments)) {
        set_error("ga_g16_proof_marshal_raw: null argument");
        return GA_ERR_INVALID;
    }
    GA_DISPATCH_CURVE(curve, return marshal<C>(proof, commitments, n, pok, out, cap, len, true));
    return GA_OK;
} GA_ABI_CATCH

int ga_g1_marshal_uncompressed(int curve, const void* affine, uint8_t* out, size_t cap, size_t* len) try {
    GA_ABI_ENTRY();
    if (!affine || !out || !len) {
        set_error("ga_g1_marshal_uncompressed: null argument");
        return GA_ERR_INVALID;
    }
    GA_DISPATCH_CURVE(curve, {
        typedef typename C::FpP P;
        const size_t nb = P::N * 4;
        if (cap < 2 * nb) {
            set_error("ga_g1_marshal_uncompressed: buffer too small");
            return GA_ERR_INVALID;
        }
        Affine<Fe<P>> a;
        memcpy(&a, affine, sizeof(a));
        memset(out, 0, 2 * nb);
        if (is_inf(a)) {
            out[0] = C::ID == GA_BN254 ? 0x40 : 0x40;   // mUncompressedInfinity: 0b01<<6 (BN254), 0b010<<5 (BLS12-381)
        } else {
            be_bytes<P>(a.x, out);
            be_bytes<P>(a.y, out + nb);
        }
        *len = 2 * nb;
    });
    return GA_OK;
} GA_ABI_CATCH

int ga_g16_commit(ga_g16_pk* p, uint32_t index, const void* values, uint64_t n_values, void* commitment_out, void* pok_out) try {
    GA_ABI_ENTRY();
    G16Pk* pk = reinterpret_cast<G16Pk*>(p);
    if (!pk || (!values && n_values) || !commitment_out || !pok_out) {
        set_error("ga_g16_commit: null argument");
        return GA_ERR_INVALID;
    }
    GA_PK_USE(pk, "ga_g16_commit");
    CtxLock g(pk->ctx);
    GA_DISPATCH_CURVE(pk->curve, return commit<C>(pk, index, values, n_values, commitment_out, pok_out));
    return GA_OK;
} GA_ABI_CATCH

int ga_g16_fold_pok(int curve, const void* poks, uint64_t n, const void* challenge, void* out) try {
    GA_ABI_ENTRY();
    if ((!poks && n) || !challenge || !out) {
        set_error("ga_g16_fold_pok: null argument");
        return GA_ERR_INVALID;
    }
    GA_DISPATCH_CURVE(curve, return fold_pok<C>(poks, n, challenge, out));
    return GA_OK;
} GA_ABI_CATCH


int ga_g16_prove(ga_g16_pk* p, const void* w, const void* a, const void* b, const void* c, uint64_t n_constraints,
                 uint64_t nb_public, const void* r, const void* s, void* proof_out) try {
    GA_ABI_ENTRY();
    return g16_prove_impl(p, w, a, b, c, n_constraints, nb_public, r, s, proof_out);
} GA_ABI_CATCH
// One proof on a key that is NOT kept on the device (the Go package's default, PinToGPU = false, as icicle.go:797-805): the key
// goes up as plain vectors WHILE the proof runs -- the uploader thread of pk_create_from_struct copies A, B, K, G2.B, Z in the order
// the MSMs consume them, every MSM waits for its own vector only -- and is dropped afterwards.  Same proof bytes as
// ga_g16_pk_create(precompute = -1) + ga_g16_prove + ga_g16_pk_destroy, in about the time of the longer of the two (PCIe, device)
// instead of their sum.  No host pointer is used after the call returns (the uploader is joined before the key is freed).
int ga_g16_prove_oneshot(ga_ctx* h, const ga_g16_key* key, const void* w, const void* a, const void* b, const void* c, uint64_t n_constraints,
                         uint64_t nb_public, const void* r, const void* s, void* proof_out) try {
    GA_ABI_ENTRY();
    Ctx* ctx = reinterpret_cast<Ctx*>(h);
    if (!ctx || !key || !w || !a || !b || !c || !r || !s || !proof_out) {
        set_error("ga_g16_prove_oneshot: null argument");
        return GA_ERR_INVALID;
    }
    trace_event("ga_g16_prove_oneshot", 0);
    struct InFlight {   // pageable uploads of this context take turns while the key is on its way (common.hip.h)
        Ctx* c;
        explicit InFlight(Ctx* x) : c(x) { c->oneshot_inflight++; }
        ~InFlight() { c->oneshot_inflight--; }
    } inflight(ctx);
    G16Pk* pk = nullptr;
    {
        CtxLock g(ctx);
        GA_CHECK(pk_create_from_struct(ctx, key, &pk, /*defer_uploads=*/true));
    }
    struct Drop {   // whatever happens below, the uploader is joined and the key freed before the host vectors go out of scope
        G16Pk* pk;
        ~Drop() {
            pk_destroy_impl(pk);
            trace_event("key dropped", 0);
        }
    } drop{pk};
    trace_event("key reserved, uploader started", 0);
    const int rc = g16_prove_impl(reinterpret_cast<ga_g16_pk*>(pk), w, a, b, c, n_constraints, nb_public, r, s, proof_out);
    trace_event("proof done", rc);
    return rc;
} GA_ABI_CATCH
int ga_g16_prove_partial(ga_g16_pk* p, const void* w, const void* a, const void* b, const void* c, uint64_t n_constraints,
                         uint64_t nb_public, void* partials_out) try {
    GA_ABI_ENTRY();
    return g16_prove_partial_impl(p, w, a, b, c, n_constraints, nb_public, partials_out);
} GA_ABI_CATCH
int ga_g16_witness_partial(ga_g16_pk* p, const void* w, uint64_t nb_public, void* partials_out) try {
    GA_ABI_ENTRY();
    return g16_witness_partial_impl(p, w, nb_public, partials_out);
} GA_ABI_CATCH
int ga_g16_prove_multi(ga_g16_pk* const* keys, uint32_t n, const void* w, const void* a, const void* b, const void* c,
                       uint64_t n_constraints, uint64_t nb_public, const void* r, const void* s, void* proof_out) try {
    GA_ABI_ENTRY();
    return g16_prove_multi_impl(keys, n, w, a, b, c, n_constraints, nb_public, r, s, proof_out);
} GA_ABI_CATCH
}  // extern "C"
