// Context, error handling, stage profiler and device scratch management shared by every translation unit
// of libgnark_amd.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <chrono>
#include <condition_variable>
#include <atomic>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "../../include/gnark_amd.h"
#include "ec.hip.h"

namespace ga {

// Every kernel of the library assumes 64-lane wavefronts: lane = tid & 63 and wave = tid >> 6, wave-local LDS exchanges without an
// s_barrier (wave_lds_sync, the NTT rounds below slot bit 8), a 16-wave scan of 1024 threads (msm_block_excl_scan_1024).  ROCm 7 defines
// no compile-time wavefront-size macro any more; GFX9 / CDNA (gfx950 included) has no wave32 mode, so the device pass insists on that
// family here, and ga_ctx_create refuses a device whose hipDeviceProp_t::warpSize is not 64.  GA_REQUIRE_WAVE64() marks the code
// that would silently compute wrong results on 32-lane waves.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__GFX9__)
#error "libgnark_amd: the kernels assume the 64-lane wavefronts of GFX9 / CDNA (gfx950); a wave32 target would compile and compute wrong transforms and sorts"
#endif
// (GA_REQUIRE_WAVE64 is DOCUMENTATION: it expands to a tautology and checks nothing by itself; the checks are the #error above
// and the warpSize test in ga_ctx_create)
#define GA_REQUIRE_WAVE64() static_assert(true, "64-lane wavefronts: enforced for the whole device pass at the top of common.hip.h")

// Rendezvous of the lanes of ONE wave around LDS traffic among themselves: a wave's LDS instructions are served in issue order, so
// a ds_write by one lane is visible to a later ds_read of another lane of the same wave without an s_barrier; what is needed is
// that the compiler keeps the two sides in program order (the fences emit no instruction).  The functional emulation runs lanes as
// fibers and has them meet for real.
__device__ __forceinline__ void wave_lds_sync() {
    GA_REQUIRE_WAVE64();
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#elif defined(GA_HIP_EMULATION)
    hipemu::wave_barrier();
#endif
}

struct Domain;   // ntt.hip.h

// ---- errors --------------------------------------------------------------------------------------
void set_error(const char* fmt, ...);
const char* get_error();

// ---- the exception barrier of the C ABI ------------------------------------------------------------------------------------------
// No C++ exception may cross an extern "C" entry point (a std::bad_alloc / std::system_error unwinding into cgo aborts the Go
// process; the reference turns device errors into Go errors, icicle.go:122-208).  EVERY entry point is a function-try-block
//     int ga_x(...) try { GA_ABI_ENTRY(); ... } GA_ABI_CATCH
// whose handler maps the exception to an error code + ga_last_error text (abi_exception_code); the RAII guards inside (locks, lanes,
// slot leases, staged buffers, thread joiners) release on the way out, so the context stays usable.
// GA_ABI_ENTRY also names the entry point for the fault knob GA_FAULT_THROW=<entry point> (tests only): Ctx::scratch_get then throws
// std::bad_alloc when called under that entry point on the calling thread.
int abi_exception_code(const char* entry) noexcept;   // call inside a catch (...) handler
struct EntryScope {
    const char* prev;
    explicit EntryScope(const char* name);
    ~EntryScope();
};
const char* current_entry();         // the innermost entry point of the calling thread ("" outside the library)
#define GA_ABI_ENTRY() ::ga::EntryScope _ga_entry_scope(__func__)
#define GA_ABI_CATCH \
    catch (...) { return ::ga::abi_exception_code(__func__); }
#define GA_ABI_CATCH_VOID \
    catch (...) { (void)::ga::abi_exception_code(__func__); }

#define GA_HIP_CHECK(expr)                                                                              \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess) {                                                                         \
            ::ga::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return GA_ERR_HIP;                                                                          \
        }                                                                                               \
    } while (0)

// Every device allocation of the library goes through here.  Two things hipMalloc does not do for a long-lived server process:
//  * the ROCm runtime allocates the private-segment ("scratch") memory of a kernel at dispatch time and ABORTS THE PROCESS when it
//    cannot (rocdevice.cpp "Aborting with error : HSA_STATUS_ERROR_OUT_OF_RESOURCES", reproduced in round 3 with
//    tools/exp/oom_repro.py: a key pinned into the last 96 MiB of HBM, then the table kernel's 896 B of scratch per lane) -- so an
//    allocation that would leave less than GA_HBM_RESERVE_MB (default 1024) free is refused as out-of-memory instead;
//  * a failed hipMalloc leaves its error in the thread's sticky last-error slot, where the next kernel-launch check would find it.
hipError_t device_malloc_bytes(void** p, size_t bytes);
const char* device_malloc_error(hipError_t e);   // why the last device_malloc of this thread failed (names the HBM reserve when that was the cause)
template <class T>
inline hipError_t device_malloc(T** p, size_t bytes) {
    return device_malloc_bytes(reinterpret_cast<void**>(p), bytes);
}

#define GA_CHECK(expr)               \
    do {                             \
        int _r = (expr);             \
        if (_r != GA_OK) return _r;  \
    } while (0)

#define GA_KERNEL_CHECK() GA_HIP_CHECK(hipGetLastError())

// ---- curve tags ----------------------------------------------------------------------------------
struct Bn254 {
    static constexpr int ID = GA_BN254;
    using FpP = BN254_Fp;
    using FrP = BN254_Fr;
};
struct Bls12381 {
    static constexpr int ID = GA_BLS12_381;
    using FpP = BLS12_381_Fp;
    using FrP = BLS12_381_Fr;
};
template <class C, int G> struct GroupField;
template <class C> struct GroupField<C, GA_G1> { using F = Fe<typename C::FpP>; };
template <class C> struct GroupField<C, GA_G2> { using F = Fe2<typename C::FpP>; };

GA_HD uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

template <class C, int G> struct Generator;
template <class C> struct Generator<C, GA_G1> {
    typedef typename C::FpP P;
    GA_HD static Affine<Fe<P>> get() { return {fe_const<P>(P::G1X), fe_const<P>(P::G1Y)}; }
};
template <class C> struct Generator<C, GA_G2> {
    typedef typename C::FpP P;
    GA_HD static Affine<Fe2<P>> get() {
        return {{fe_const<P>(P::G2X0), fe_const<P>(P::G2X1)}, {fe_const<P>(P::G2Y0), fe_const<P>(P::G2Y1)}};
    }
};

// ---- profiler: hipEvent pairs around named stages on the context's stream ----------------------------
struct StageRec {
    std::string name;
    hipEvent_t a, b;
};

// Run-time knobs, read from the environment ONCE per ABI entry point (Lock's constructor), never inside the launch paths:
//   GA_MSM_MAX_CHUNK      split an MSM along the point axis into chunks of at most this many points (msmChunkedG1/G2 analogue)
//   GA_REDUCE_LAZY_MIN    bucket count from which the window reduction runs in the lazy representation
//   GA_G16_SHARE_MIN_PCT  a Groth16 base vector shares the single witness sort when it covers at least this % of the wires
//   GA_G16_TABLE_BUDGET_PCT  % of the bytes of a key's five window tables that precompute = 0 may spend, instead of what the free HBM allows (tests: partial tables)
//   GA_MSM_MIN_SEG        shortest task length the bucket lists are cut into (points per task)
//   GA_MSM_FUSE_MIN       (point, window) pairs from which an MSM sorts with the fused two-level sort instead of the library's (2^21)
//   GA_MSM_XCD            fused sort: bit 0 per-XCD slices in the first level (from 2^24 pairs; bit 2: at any size), bit 1 XCD swizzle in the second (3; A/B knob)
//   GA_FAULT_THROW        tests: name of an entry point under which the next device-scratch request throws std::bad_alloc
//   GA_MSM_P1_GRID        workgroups of the fused sort's first level (each walks tiles b, b + grid, ...; 512; tests: 1 .. 9)
//   GA_MSM_EXACT_REDO     1: tasks flagged by the fast bucket loop go straight to the exact-arithmetic kernel (tests)
//   GA_TABLE_C            force the window width of precomputed tables built from now on (experiments; 0 = planned)
//   GA_G16_LANES          1: a second concurrent ga_g16_prove caller queues for the device instead of proving on its own lanes
//   GA_G16_SPLIT          0: a proof keeps its H side (computeH, Z MSM) on the lane of its witness MSMs instead of a partner lane
//   GA_NTT_COSET_FOLD     0: coset FFTs scale their input by the coset powers (round 2) instead of running over a coset twiddle table
//   GA_NTT_WAVE_LOCAL     0: every round of an NTT pass ends in a workgroup barrier (round 3) instead of wave-local exchanges
//   GA_NTT_DIRECT         0: the first / last round of an NTT pass goes through LDS instead of moving its quads to / from HBM itself
// (GA_HBM_RESERVE_MB is read once per process by device_malloc: see there.)
// The fields are relaxed atomics: the entry point that holds lane 0 refreshes them while provers on the other lanes read them.
struct Tunables {
    std::atomic<uint64_t> msm_max_chunk{0};          // 0 = only the 2^31 pair-space limit
    std::atomic<uint64_t> reduce_lazy_min{1u << 14};
    std::atomic<int> g16_share_min_pct{90};
    std::atomic<uint64_t> g16_table_budget_pct{0};    // 0 = from the free HBM; else the % of the five tables' bytes precompute = 0 may spend (tests)
    std::atomic<int> g16_lanes{2};
    std::atomic<int> g16_split{1};
    std::atomic<int> g16_batch_tables{1};            // 1: the wire-indexed G1 tables of a proof (A, B1, K) in one pass of every MSM kernel
    std::atomic<int> ntt_coset_fold{1};
    std::atomic<int> ntt_wave_local{1};
    std::atomic<int> ntt_direct{1};
    std::atomic<int> table_c{0};
    std::atomic<uint64_t> msm_min_seg{256};
    std::atomic<int> msm_exact_redo{0};
    std::atomic<uint64_t> msm_fuse_min{1ull << 21};   // pairs from which the digits are fused with the first sort pass (msm.hip.h 1b)
    std::atomic<int> msm_group{0};                   // buckets per running-sum group of the window reduction (0 = MSM_GROUP)
    std::atomic<uint64_t> fault_throw{0};            // FNV-1a of GA_FAULT_THROW (0 = unset): the entry point under which scratch_get throws (tests)
    std::atomic<int> fault_lane2_nomem{0};           // GA_FAULT_LANE2_NOMEM (tests): scratch requests of lanes 2/3 fail as if HBM were exhausted
    std::atomic<uint64_t> msm_p1_grid{512};         // blocks of the first sort level (a block walks several tiles)
    std::atomic<uint64_t> msm_task_exact_min{1ull << 25};   // pairs from which the task list is sorted on exact lengths (msm.hip.h 3)
    std::atomic<int> msm_xcd{3};                     // fused sort placement: bit 0 per-XCD slices of the first level's groups, bit 1 XCD swizzle of the second level's segments
    void read_env();
};

// Lanes: a host thread inside an entry point works on ONE lane of its context = one stream + one namespace of the scratch map.
// Lane 0 is the context's main stream, guarded by Ctx::mu (every entry point); lanes 1..3 are guarded by Ctx::lane_mu[lane].
//   * a proof runs on a PAIR of lanes: the witness MSMs on the caller's lane (0, or 2 for a second concurrent caller) and its H
//     side -- uploads of A, B, C, computeH, the Z MSM -- on the partner lane (1, or 3) from a helper thread, so that the sorts,
//     reduction tails and host round trips of one half run under the bucket kernels of the other (groth16.hip prove_partial);
//   * a second ga_g16_prove caller computes its proof CONCURRENTLY on lanes 2/3 instead of queueing behind the first: its
//     uploads hide behind the other proof's kernels;
//   * the table MSM entry points and the pieces of a sharded proof take lane 1 when lane 0 is busy (LaneLock).
// The lane is a thread-local of the calling thread (abi.hip), so the launch paths pick the right stream / scratch without extra
// parameters.  Results never depend on the interleaving.
constexpr int GA_NUM_LANES = 4;
int current_lane();
int table_c_override();   // Tunables::table_c of the last read_env (process-wide: msm_plan_table has no context)
struct LaneScope {
    int prev;
    explicit LaneScope(int lane);
    ~LaneScope();
};

struct Ctx {
    int device = 0;
    // A few words of PINNED host memory per lane: counters an entry point copies back while more launches follow (the count of
    // flagged tasks of an MSM).  Into pageable memory hipMemcpyAsync blocks the host until the stream gets there -- every launch
    // behind it then starts from an empty queue (round 4: 0.1-0.2 ms of bubbles per MSM, 7 % of a 2^20 one).
    uint32_t* host_pin[GA_NUM_LANES] = {nullptr, nullptr, nullptr, nullptr};
    uint32_t* pinned_words() {   // 64 words for the calling thread's lane, or null (callers fall back to a blocking copy)
        const int l = current_lane();
        if (!host_pin[l] && hipHostMalloc((void**)&host_pin[l], 256, 0) != hipSuccess) {
            host_pin[l] = nullptr;
            (void)hipGetLastError();
        }
        return host_pin[l];
    }
    Tunables tun;
    hipStream_t stream = nullptr;                        // lane 0
    hipStream_t lane_stream[GA_NUM_LANES] = {nullptr, nullptr, nullptr, nullptr};   // [0] == stream
    std::mutex lane_mu[GA_NUM_LANES];                    // at most one thread per lane ([0] unused: lane 0 is guarded by `mu`)
    hipStream_t work_stream() const { return lane_stream[current_lane()]; }
    // how ga_g16_prove calls were scheduled since the context was created (ga_g16_lane_stats)
    std::atomic<uint64_t> stat_lane0{0}, stat_lane2{0}, stat_queued{0}, stat_split{0};
    hipStream_t copy_stream = nullptr;   // uploads that overlap kernels (groth16.hip)
    std::mutex mu;
    // Two input slots per context (W, A, B, C staging buffers each): while one proof computes under `mu`, a second caller of
    // ga_g16_prove stages its solution in the other slot over PCIe, so that back-to-back proofs from two host threads (two
    // goroutines) hide the upload of the next proof behind the kernels of the current one.
    std::mutex slot_mu;
    std::condition_variable slot_cv;
    bool slot_busy[2] = {false, false};
    hipStream_t slot_stream[2] = {nullptr, nullptr};
    std::mutex scratch_mu;   // the scratch map is touched by the staging thread outside `mu`
    // While a one-shot proof is in flight (ga_g16_prove_oneshot: 6-9 GiB of key on their way beside W and the solver's A, B, C) the
    // pageable host-to-device copies of this context take turns, 128 MiB at a time, in the order they asked (a ticket lock): two
    // such copies at once share the link at 24 + 24 GB/s where one alone gets 56 (profiles/r06_j_h2d_concurrency.json).  Otherwise
    // they run as they come: two pinned-key proofs in flight LOSE with the turns (122 -> 131 ms per proof, r06_k).
    std::atomic<int> oneshot_inflight{0};
    // (turns are taken by PRIORITY = the order in which the proof needs the data: W, key A, key B, the solver's A, B, C, key G2.B,
    // K, Z; equal priorities in arrival order; a copy gives way at 128 MiB boundaries)
    struct TurnLock {
        std::mutex mu;
        std::condition_variable cv;
        bool busy = false;
        uint64_t seq = 0;
        std::set<std::pair<int, uint64_t>> waiting;
        void lock(int prio) {
            std::unique_lock<std::mutex> g(mu);
            const std::pair<int, uint64_t> me(prio, seq++);
            waiting.insert(me);
            cv.wait(g, [&] { return !busy && *waiting.begin() == me; });
            waiting.erase(waiting.begin());
            busy = true;
        }
        void unlock() {
            {
                std::lock_guard<std::mutex> g(mu);
                busy = false;
            }
            cv.notify_all();
        }
    } h2d_turn;
    hipError_t h2d_pageable(void* dst, const void* src, size_t bytes, hipStream_t st, int prio = 3) {
        if (oneshot_inflight.load(std::memory_order_relaxed) == 0) return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st);
        const size_t chunk = (size_t)128 << 20;
        for (size_t off = 0; off < bytes; off += chunk) {
            const size_t nb = bytes - off < chunk ? bytes - off : chunk;
            static const bool trace = getenv("GA_TRACE_H2D") != nullptr;
            const auto t0 = std::chrono::steady_clock::now();
            h2d_turn.lock(prio);
            const auto t1 = std::chrono::steady_clock::now();
            const hipError_t e = hipMemcpyAsync((char*)dst + off, (const char*)src + off, nb, hipMemcpyHostToDevice, st);
            h2d_turn.unlock();
            if (trace)
                fprintf(stderr, "[h2d] prio %d chunk %zu: waited %.2f ms, copy call %.2f ms\n", prio, off / chunk,
                        std::chrono::duration<double, std::milli>(t1 - t0).count(),
                        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    }
    std::mutex spare_mu;
    Domain* spare_domain = nullptr;   // ntt_domain_give_spare / ntt_domain_take_spare
    // ... and so do the five vector buffers of the last ONE-SHOT key (groth16.hip): a caller that uploads its key for every proof gets
    // them back instead of 6-9 GiB of hipMalloc / hipFree per proof (5-200 ms, r06_k)
    struct SpareVectors {
        void* p[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
        size_t bytes[5] = {0, 0, 0, 0, 0};
        bool have = false;
    } spare_vectors;
    std::atomic<bool> profiling{false};
    std::vector<StageRec> stages;
    // reusable device scratch, grown on demand (keyed by purpose)
    std::map<std::string, std::pair<void*, size_t>> scratch;

    // base tables on which the fast bucket loop flagged most of its tasks (all bases equal: a DummySetup key); msm.hip.h
    std::mutex degenerate_mu;
    std::set<const void*> degenerate;
    bool is_degenerate(const void* table) {
        std::lock_guard<std::mutex> g(degenerate_mu);
        return degenerate.count(table) != 0;
    }
    void mark_degenerate(const void* table) {
        std::lock_guard<std::mutex> g(degenerate_mu);
        degenerate.insert(table);
    }
    void forget_table(const void* table) {
        std::lock_guard<std::mutex> g(degenerate_mu);
        degenerate.erase(table);
        sparse_sets.erase(table);
    }
    // tables whose SMALL shared bucket set turned out mostly empty on the previous call (a witness of zeros and ones puts nearly
    // every (scalar, window) pair into the skip bucket): the lazy window reduction flags most groups there, so the next call takes
    // the exact kernel again (msm.hip.h, dense_set)
    // The verdict is re-examined: every 16th call on such a table takes the lazy pass again and note_sparse_set renews or drops it
    // (one 0/1-heavy witness must not send a table to the slow exact kernel for the rest of its life).
    std::map<const void*, uint32_t> sparse_sets;   // table -> calls since the verdict
    bool is_sparse_set(const void* table) {
        std::lock_guard<std::mutex> g(degenerate_mu);
        auto it = sparse_sets.find(table);
        if (it == sparse_sets.end()) return false;
        return (++it->second % 16) != 0;
    }
    void note_sparse_set(const void* table, bool sparse) {
        std::lock_guard<std::mutex> g(degenerate_mu);
        if (sparse)
            sparse_sets.emplace(table, 0u);   // (keeps the running count of an existing verdict)
        else
            sparse_sets.erase(table);
    }

    int scratch_get(const char* key, size_t bytes, void** out);
    void scratch_free_all();
    void scratch_free_lanes(int first_lane);   // give back the scratch of lanes >= first_lane (their streams must be idle)
};

// every ABI entry point: take the context's mutex (one proof at a time per device, icicle.go:821-823), select its device (HIP's
// current device is per OS thread and goroutines migrate) and refresh the run-time knobs
struct CtxLock {
    std::lock_guard<std::mutex> g;
    explicit CtxLock(Ctx* c) : g(c->mu) {
        hipSetDevice(c->device);
        c->tun.read_env();
    }
};

// Device access for an entry point that may run BESIDE another one on the same context (ga_g16_h_chain / ga_g16_h_combine next to
// the witness MSMs of a sharded proof): lane 0 under Ctx::mu when the device is free, lane 1 under Ctx::lane_mu when lane 0 is busy,
// otherwise wait for lane 0.  (ga_g16_prove has its own variant that pre-stages the solution before it waits.)
struct LaneLock {
    std::unique_lock<std::mutex> dev, l1;
    int lane = 0;
    LaneScope* scope = nullptr;
    explicit LaneLock(Ctx* c) : dev(c->mu, std::try_to_lock), l1(c->lane_mu[1], std::defer_lock) {
        hipSetDevice(c->device);
        if (!dev.owns_lock()) {
            if (!c->profiling && c->tun.g16_lanes > 1 && l1.try_lock()) lane = 1;
            else dev.lock();
        }
        if (lane == 0) c->tun.read_env();
        scope = new LaneScope(lane);
    }
    ~LaneLock() { delete scope; }
    LaneLock(const LaneLock&) = delete;
    LaneLock& operator=(const LaneLock&) = delete;
};

// ownership of one input slot of a context for the duration of a proof
struct SlotLease {
    Ctx* ctx;
    int slot;
    explicit SlotLease(Ctx* c) : ctx(c), slot(0) {
        std::unique_lock<std::mutex> g(c->slot_mu);
        c->slot_cv.wait(g, [&] { return !c->slot_busy[0] || !c->slot_busy[1]; });
        slot = c->slot_busy[0] ? 1 : 0;
        c->slot_busy[slot] = true;
    }
    ~SlotLease() {
        {
            std::lock_guard<std::mutex> g(ctx->slot_mu);
            ctx->slot_busy[slot] = false;
        }
        ctx->slot_cv.notify_one();
    }
    // per-slot scratch name: "g16_w" -> "g16_w#1"
    std::string name(const char* base) const { return std::string(base) + (slot ? "#1" : "#0"); }
};

struct StageTimer {
    Ctx* ctx;
    int idx = -1;
    hipStream_t st;
    StageTimer(Ctx* c, const char* name, hipStream_t stream = nullptr) : ctx(c), st(stream ? stream : (c ? c->work_stream() : nullptr)) {
        if (!c || !c->profiling || current_lane() != 0) return;   // the stage list belongs to lane 0 (guarded by Ctx::mu)
        StageRec r;
        r.name = name;
        if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
        hipEventRecord(r.a, st);
        c->stages.push_back(r);
        idx = (int)c->stages.size() - 1;
    }
    ~StageTimer() {
        if (idx >= 0) hipEventRecord(ctx->stages[idx].b, st);
    }
};

struct DeviceGuard {
    explicit DeviceGuard(int dev) { hipSetDevice(dev); }
};

static inline int ilog2_u64(uint64_t x) {
    int l = 0;
    while ((1ull << l) < x) l++;
    return l;
}

// ---- per-curve entry points; explicitly instantiated in msm_*.hip / ntt_*.hip / util_*.hip -------------
// MSM: accumulates windows [win_lo, win_hi) of sum scalars[i]*bases[i]; writes (win_hi-win_lo) XYZZ window sums
// (host memory, XYZZ<F> images).  d_bases / d_scalars are device pointers.
// combine: h_window_sums receives ONE point, sum_w 2^(c (w - win_lo)) W_w (the Horner step folded into the window reduction's host tail).
template <class C, int G>
int msm_windows_device(Ctx* ctx, const void* d_bases, const void* d_scalars, size_t n, bool scalars_mont, int c,
                       int win_lo, int win_hi, void* h_window_sums, bool combine = false);
template <class C>
int msm_plan(int group, size_t n, int* c, int* nwin);
// window width for the precomputed-table mode (one shared bucket set; table = nwin x n affine points)
template <class C>
int msm_plan_table(size_t n, int* c, int* nwin, bool batched = false);

// Output of the group-independent half of an MSM (digits -> sort -> bucket offsets -> length-ordered task list).
// The arrays live in the context's scratch; they stay valid until the next msm_prepare on the same context, so one
// prepared scalar vector can feed several base vectors (G1.B and G2.B share wireValuesB, prove.go:194,283).
struct MsmPrepared {
    size_t n = 0;
    int c = 0, nwin = 0, win_lo = 0, win_hi = 0;
    int nsets = 0;          // bucket sets = window sums produced: (win_hi - win_lo), or 1 in table mode
    bool table = false;
    uint32_t half = 0, nb = 0, seg = 0;
    uint64_t m = 0, max_tasks = 0;
    uint32_t *vals = nullptr, *task_off = nullptr, *task_start = nullptr, *task_key = nullptr, *task_perm = nullptr;   // task_key: the SORTED quantised keys (0xFFFFFFFF = padding)
    uint32_t* task_key_by_id = nullptr;   // unsorted, exact: seg - len of task id
    uint32_t* task_dest = nullptr;        // slot of the task's sum in [bucket sums | partial sums]
};

template <class C, int G>
int msm_table_build(Ctx* ctx, const void* d_bases, size_t n, int c, void* d_table);
// bytes per table entry: tables are stored unpacked (field29.hip.h: 29/28-bit limbs, one per word, padded to 16 B)
template <class C, int G>
size_t msm_table_point_bytes();
template <class C, int G>
int msm_table_device(Ctx* ctx, const void* d_table, const void* d_scalars, size_t n, bool scalars_mont, int c, void* h_sum, int win_lo = 0,
                     int win_hi = -1);
template <class C, int G>
int msm_table_device_batch(Ctx* ctx, const void* d_table, const void* const* d_scalars, int batch, size_t n, bool scalars_mont, int c,
                           void* h_sums);
// slot 0/1 selects one of two scratch sets (the witness sort shared by several tables stays live in set 1 while set 0 is reused)
template <class C>
int msm_prepare_table_scalars(Ctx* ctx, const void* d_scalars, size_t n, bool scalars_mont, int c, MsmPrepared* P, int slot = 0,
                              int win_lo = 0, int win_hi = -1);
template <class C, int G>
int msm_table_device_reuse(Ctx* ctx, const void* d_table, const MsmPrepared& P, void* h_sum);
// the same over 2..4 tables of one shape in ONE pass of every kernel (msm.hip.h): h_sums[i] = the XYZZ sum over tables[i]
template <class C, int G>
int msm_table_device_reuse_multi(Ctx* ctx, const void* const* tables, int ntab, const MsmPrepared& P, void* h_sums);

struct Domain;   // ntt.hip.h
template <class C> int ntt_domain_new(Ctx* ctx, uint64_t n, Domain** out);
template <class C> int ntt_domain_fft(Domain* d, void* d_data, int direction, int decimation, int on_coset);
template <class C> int ntt_domain_compute_h(Domain* d, void* d_a, void* d_b, void* d_c);
template <class C> int ntt_domain_h_chain(Domain* d, void* d_v);                                        // v <- FFT_coset(iFFT(v))
template <class C> int ntt_domain_h_combine(Domain* d, void* d_a, const void* d_b, const void* d_c);    // a <- h (bit-reversed)
void ntt_domain_delete(Domain* d);
// The NTT domain of the last proving key freed on a context stays with the context (like its scratch): a caller that re-pins its key
// for every proof -- the Go package's default, PinToGPU = false -- gets it back instead of rebuilding 3 GiB of twiddle tables (11 ms
// of a 160 ms pin at 2^24).  One domain at most; GA_DOMAIN_SPARE=0 frees it at once.
Domain* ntt_domain_take_spare(Ctx* ctx, int curve, uint64_t n);   // null when the spare does not match
void ntt_domain_give_spare(Ctx* ctx, Domain* d);                  // the previous spare is deleted
int ntt_domain_curve(const Domain* d);
uint64_t ntt_domain_size(const Domain* d);
Ctx* ntt_domain_ctx(const Domain* d);

// PLONK quotient / grand product on device (plonk.hip.h)
constexpr int PLONK_MAX_BSB = 16;
constexpr int PLONK_NB_FIXED = 12;   // L R O Z Ql Qr Qm Qo Qk S1 S2 S3 (plonk prove.go:44-59; ZS is Z shifted by one)
struct PlonkQuotientArgs {
    uint32_t nb_bsb;
    const void* polys[PLONK_NB_FIXED + 2 * PLONK_MAX_BSB];   // fixed ids, then (Qcp_i, committed polynomial i) pairs
    uint64_t lagrange_mask;   // bit k: polys[k] holds evaluations on the small domain (Lagrange, regular) instead of canonical coefficients
    bool on_device;           // polys and h_out are device pointers
    const void *bl, *br, *bo, *bz;   // blinding polynomials: 2, 2, 2, 3 coefficients
    const void *alpha, *beta, *gamma;
};
struct PlonkFixed;   // plonk.hip.h: the circuit-constant coset evaluations pinned in HBM
template <class C> int plonk_domain_quotient(Domain* d0, Domain* d1, const PlonkQuotientArgs& args, void* h_out);
template <class C> int plonk_domain_fixed_create(Domain* d0, Domain* d1, const PlonkQuotientArgs& args, PlonkFixed** out);
template <class C> int plonk_domain_quotient_pinned(PlonkFixed* fx, const PlonkQuotientArgs& args, void* h_out);
void plonk_fixed_delete(PlonkFixed* fx);
Domain* plonk_fixed_domain0(PlonkFixed* fx);
template <class C> int plonk_domain_build_z(Domain* d0, const void* L, const void* R, const void* O, const int64_t* perm, const void* beta,
                                            const void* gamma, bool on_device, void* z_out);
template <class C> int fr_vec_batch_inverse(Ctx* ctx, void* v, uint64_t n, bool on_device);
template <class C> int fr_vec_lincomb(Ctx* ctx, uint64_t n, int k, const void* const* vecs, const void* scalars, void* out, bool on_device);
template <class C> int kzg_domain_divide(Ctx* ctx, const void* d_poly, uint64_t n, const void* z_mont, void* d_quot, void* value_out);

// utility kernels (util_*.hip)
template <class C, int G> int util_gen_bases(Ctx* ctx, uint64_t seed, size_t n, void* d_bases, void* d_dlogs, uint64_t first = 0);
template <class C> int util_gen_scalars(Ctx* ctx, uint64_t seed, size_t n, void* d_scalars);
template <class C> int util_fr_dot(Ctx* ctx, const void* d_a, const void* d_b, size_t n, void* h_out);
template <class C> int util_fr_vec_mul(Ctx* ctx, const void* d_a, const void* d_b, size_t n, void* d_out);
template <class C> int util_gather_fr(Ctx* ctx, void* d_dst, const void* d_src, const uint32_t* d_idx, size_t n);
int util_microbench(Ctx* ctx, char* buf, size_t cap);
int util_clock_probe(Ctx* ctx, uint32_t micros, double* mhz_out);

#define GA_DISPATCH_CURVE(curve, ...)                                \
    switch (curve) {                                                 \
        case GA_BN254: {                                             \
            using C = ::ga::Bn254;                                   \
            __VA_ARGS__;                                             \
        } break;                                                     \
        case GA_BLS12_381: {                                         \
            using C = ::ga::Bls12381;                                \
            __VA_ARGS__;                                             \
        } break;                                                     \
        default:                                                     \
            ::ga::set_error("unknown curve id %d", (int)(curve));    \
            return GA_ERR_INVALID;                                   \
    }

}  // namespace ga
