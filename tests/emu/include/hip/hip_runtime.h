// TEST INFRASTRUCTURE ONLY -- a tiny functional emulation of the subset of the HIP runtime and device
// language that gnark_amd/csrc uses, so that kernel *logic* (indexing, barriers, wave shuffles) can be
// exercised by `pytest -m "not gpu"` in a container without a GPU.
//
// It is never part of the product: gnark_amd/_lib.py only ever loads the hipcc-built libgnark_amd.so and
// raises if it is missing; tests/emu builds a separate libgnark_amd_emu.so (tests/emu/build_emu.sh) by
// compiling the unmodified product sources with this directory first on the include path.
//
// Model: a small pool of OS worker threads takes the blocks of a launch; the threads of a block are ucontext fibers inside
// one worker, scheduled round-robin and parked at __syncthreads() (block barrier) or at the two meeting points of a
// __shfl*/__ballot (wave barrier: 64 consecutive threads); a thread that has returned no longer counts for either barrier,
// as on the GPU.  `__shared__` maps to `static thread_local` (one copy per worker = per resident block).
#pragma once
#include <pthread.h>
#include <stdint.h>
#include <sys/mman.h>
#include <ucontext.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <tuple>
#include <utility>
#include <vector>

#define GA_HIP_EMULATION 1

#define __host__
#define __device__
#define __global__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static thread_local
#define __launch_bounds__(...)
#define __restrict__ __restrict
#define HIP_DYNAMIC_SHARED(type, var) type* var = reinterpret_cast<type*>(::hipemu::dyn_smem());

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

struct uint4 {
    unsigned x, y, z, w;
} __attribute__((aligned(16)));
struct uint2 {
    unsigned x, y;
} __attribute__((aligned(8)));
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }

namespace hipemu {
struct Wave {
    uint64_t scratch[64];
    unsigned live = 0, arrived = 0;
};
// Switching fibers: glibc's swapcontext saves and restores the signal mask -- two system calls per switch, a third of the CPU suite's
// time in the kernel -- so, outside AddressSanitizer builds (whose stack bookkeeping follows ucontext), a fiber is a saved stack
// pointer and a switch is hipemu_switch (emu_impl.cpp): push the callee-saved registers, swap rsp, pop, ret.
#if defined(__x86_64__) && !defined(__SANITIZE_ADDRESS__) && !defined(HIPEMU_UCONTEXT)
#define HIPEMU_ASM_SWITCH 1
#endif
struct Fiber {
    ucontext_t ctx;
    void* sp = nullptr;   // HIPEMU_ASM_SWITCH: the fiber's saved stack pointer
    int state = 3;   // 0 runnable, 1 parked at the block barrier, 2 parked at its wave barrier, 3 finished
    char* stack = nullptr;
};
struct Idx {
    unsigned x, y, z;
};
// one per worker thread: the block it is currently running
struct Block {
    std::vector<Fiber> fibers;
    ucontext_t sched;
    void* sched_sp = nullptr;   // HIPEMU_ASM_SWITCH: the scheduler's saved stack pointer
    unsigned cur = 0, nthreads = 0, live = 0, arrived = 0;
    std::vector<Wave> waves;
    std::vector<char> smem;
    std::function<void()> body;
};
inline Block*& cur_block() {
    static thread_local Block* b = nullptr;
    return b;
}
inline void* dyn_smem() { return cur_block()->smem.data(); }
void block_barrier();
void wave_barrier();
}  // namespace hipemu

extern thread_local hipemu::Idx threadIdx;
extern thread_local hipemu::Idx blockIdx;
extern thread_local dim3 blockDim;
extern thread_local dim3 gridDim;
#ifdef GA_HIP_EMULATION_IMPL
thread_local hipemu::Idx threadIdx;
thread_local hipemu::Idx blockIdx;
thread_local dim3 blockDim;
thread_local dim3 gridDim;
#endif

static const int warpSize = 64;

inline void __syncthreads() { hipemu::block_barrier(); }

namespace hipemu {
inline Wave& my_wave() { return cur_block()->waves[cur_block()->cur / 64]; }
inline unsigned lane() { return cur_block()->cur % 64; }
template <class T>
inline T wave_exchange(T v, unsigned src) {
    static_assert(sizeof(T) <= 8, "shuffle of <= 8 bytes only");
    Wave& w = my_wave();
    uint64_t raw = 0;
    memcpy(&raw, &v, sizeof(T));
    w.scratch[lane()] = raw;
    wave_barrier();
    uint64_t got = w.scratch[src % 64];
    wave_barrier();
    T out;
    memcpy(&out, &got, sizeof(T));
    return out;
}
}  // namespace hipemu

template <class T>
inline T __shfl(T v, int src, int width = 64) {
    unsigned l = hipemu::lane();
    unsigned base = l - (l % width);
    return hipemu::wave_exchange(v, base + ((unsigned)src % width));
}
template <class T>
inline T __shfl_xor(T v, int mask, int width = 64) {
    unsigned l = hipemu::lane();
    unsigned t = l ^ (unsigned)mask;
    if ((t / width) != (l / width)) t = l;
    return hipemu::wave_exchange(v, t);
}
template <class T>
inline T __shfl_down(T v, unsigned d, int width = 64) {
    unsigned l = hipemu::lane();
    unsigned t = l + d;
    if ((t / width) != (l / width)) t = l;
    return hipemu::wave_exchange(v, t);
}
template <class T>
inline T __shfl_up(T v, unsigned d, int width = 64) {
    unsigned l = hipemu::lane();
    unsigned t = (l % width) >= d ? l - d : l;
    return hipemu::wave_exchange(v, t);
}
inline unsigned long long __ballot(int pred) {
    hipemu::Wave& w = hipemu::my_wave();
    w.scratch[hipemu::lane()] = pred ? 1 : 0;
    hipemu::wave_barrier();
    unsigned long long m = 0;
    const unsigned nl = std::min(64u, hipemu::cur_block()->nthreads - (hipemu::cur_block()->cur / 64) * 64);
    for (unsigned i = 0; i < nl; i++) m |= (unsigned long long)(w.scratch[i] & 1) << i;
    hipemu::wave_barrier();
    return m;
}
inline int __any(int p) { return __ballot(p) != 0; }
inline int __all(int p) { return __ballot(p) == ~0ull; }
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline int __popc(unsigned x) { return __builtin_popcount(x); }
inline int __ffsll(unsigned long long x) { return __builtin_ffsll((long long)x); }
inline int __clz(unsigned x) { return x ? __builtin_clz(x) : 32; }
inline int __clzll(unsigned long long x) { return x ? __builtin_clzll(x) : 64; }
inline unsigned __brev(unsigned x) {
    unsigned r = 0;
    for (int i = 0; i < 32; i++) r |= ((x >> i) & 1u) << (31 - i);
    return r;
}
inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((uint64_t)a * b) >> 32); }

template <class T>
inline T atomicAdd(T* p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
template <class T>
inline T atomicMax(T* p, T v) {
    T old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
    }
    return old;
}
template <class T>
inline T atomicOr(T* p, T v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
template <class T>
inline T atomicExch(T* p, T v) { return __atomic_exchange_n(p, v, __ATOMIC_RELAXED); }
template <class T>
inline T atomicCAS(T* p, T cmp, T v) {
    __atomic_compare_exchange_n(p, &cmp, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED);
    return cmp;
}
inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }

// ---- runtime API ---------------------------------------------------------------------------------
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotReady = 600 };
typedef struct hipemuStream* hipStream_t;
struct hipemuEvent {
    std::chrono::steady_clock::time_point t;
};
typedef hipemuEvent* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipHostMallocDefault = 0, hipStreamNonBlocking = 1, hipEventDefault = 0, hipEventDisableTiming = 2 };

struct hipDeviceProp_t {
    char name[256];
    char gcnArchName[256];
    size_t totalGlobalMem;
    int multiProcessorCount;
    int warpSize;
};

inline const char* hipGetErrorString(hipError_t e) { return e == 0 ? "hipSuccess" : "hipemu error"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDevice(int* d) {
    *d = 0;
    return hipSuccess;
}
inline hipError_t hipGetDeviceCount(int* n) {
    *n = 1;
    return hipSuccess;
}
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 0, hipDeviceAttributeWallClockRate = 1 };
inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t a, int) {
    *v = a == hipDeviceAttributeMultiprocessorCount ? 1 : 100000;
    return hipSuccess;
}
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
    memset(p, 0, sizeof(*p));
    strcpy(p->name, "hipemu (CPU functional emulation)");
    strcpy(p->gcnArchName, "emu");
    p->totalGlobalMem = 8ull << 30;
    p->multiProcessorCount = 1;
    p->warpSize = 64;
    return hipSuccess;
}
inline hipError_t hipMemGetInfo(size_t* f, size_t* t) {
    *f = 16ull << 30;
    *t = 32ull << 30;
    return hipSuccess;
}
inline hipError_t hipMalloc(void** p, size_t n) {
    *p = aligned_alloc(256, (n + 255) / 256 * 256 + 256);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
template <class T>
inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
inline hipError_t hipFree(void* p) {
    free(p);
    return hipSuccess;
}
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
template <class T>
inline hipError_t hipHostMalloc(T** p, size_t n, unsigned f = 0) { return hipMalloc((void**)p, n); }
inline hipError_t hipHostFree(void* p) { return hipFree(p); }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) {
    memmove(d, s, n);
    return hipSuccess;
}
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t = nullptr) { return hipMemcpy(d, s, n, k); }
inline hipError_t hipMemcpyPeerAsync(void* d, int, const void* s, int, size_t n, hipStream_t = nullptr) { return hipMemcpy(d, s, n, hipMemcpyDeviceToDevice); }
inline hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }
inline hipError_t hipMemset(void* d, int v, size_t n) {
    memset(d, v, n);
    return hipSuccess;
}
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = nullptr) { return hipMemset(d, v, n); }
inline hipError_t hipStreamCreate(hipStream_t* s) {
    *s = nullptr;
    return hipSuccess;
}
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) {
    *s = nullptr;
    return hipSuccess;
}
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned = 0) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t* e) {
    *e = new hipemuEvent();
    return hipSuccess;
}
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipEventDestroy(hipEvent_t e) {
    delete e;
    return hipSuccess;
}
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) {
    e->t = std::chrono::steady_clock::now();
    return hipSuccess;
}
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}

// ---- kernel launch --------------------------------------------------------------------------------
namespace hipemu {
// runs `body` once per thread of one block, as fibers of the calling worker (implemented in emu_impl.cpp)
void run_block(unsigned nthreads, size_t shmem, dim3 block_dim, dim3 grid_dim, Idx block_idx, const std::function<void()>& body);
// distributes the blocks of a grid over the worker pool and waits for all of them
void run_grid(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body);
}  // namespace hipemu

template <class F, class... Args>
inline void hipLaunchKernelGGL(F kernel, dim3 grid, dim3 block, size_t shmem, hipStream_t, Args... args) {
    auto tup = std::make_tuple(args...);
    std::function<void()> body = [&]() { std::apply(kernel, tup); };
    hipemu::run_grid(grid, block, shmem, body);
}
