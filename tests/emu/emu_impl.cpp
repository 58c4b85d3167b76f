// TEST INFRASTRUCTURE: the fiber scheduler and worker pool of the HIP emulation (see include/hip/hip_runtime.h), storage for
// the emulated built-in variables, and a stub for the (inline-asm) microbenchmarks.
#define GA_HIP_EMULATION_IMPL
#include <hip/hip_runtime.h>
#include <stdio.h>

namespace hipemu {

static constexpr size_t STACK_BYTES = 512 * 1024;

#ifdef HIPEMU_ASM_SWITCH
// void hipemu_switch(void** save_sp, void* load_sp): the System V callee-saved registers go on the current stack, its pointer is
// stored, the other stack's registers are popped and `ret` continues where that stack last called hipemu_switch (or, for a fresh
// fiber, at the trampoline whose address fiber_init put there).
extern "C" void hipemu_switch(void** save_sp, void* load_sp);
asm(R"(
    .text
    .globl hipemu_switch
    .type hipemu_switch,@function
hipemu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size hipemu_switch,.-hipemu_switch
)");
static void trampoline();
static void* fiber_init(char* stack) {
    uintptr_t top = ((uintptr_t)stack + STACK_BYTES) & ~(uintptr_t)15;
    uint64_t* p = (uint64_t*)top;
    *--p = 0;                        // where a return address would be: the trampoline never returns (rsp = 8 mod 16 at its entry)
    *--p = (uint64_t)&trampoline;    // popped by hipemu_switch's ret
    for (int i = 0; i < 6; i++) *--p = 0;   // rbp, rbx, r12..r15
    return p;
}
#define HIPEMU_TO_SCHED(f, b) hipemu_switch(&(f).sp, (b)->sched_sp)
#define HIPEMU_TO_FIBER(b, f) hipemu_switch(&(b)->sched_sp, (f).sp)
#else
#define HIPEMU_TO_SCHED(f, b) swapcontext(&(f).ctx, &(b)->sched)
#define HIPEMU_TO_FIBER(b, f) swapcontext(&(b)->sched, &(f).ctx)
#endif

static void release_block(Block* b) {
    for (auto& f : b->fibers)
        if (f.state == 1) f.state = 0;
    b->arrived = 0;
}
static void release_wave(Block* b, unsigned w) {
    for (unsigned t = w * 64; t < std::min(b->nthreads, (w + 1) * 64); t++)
        if (b->fibers[t].state == 2) b->fibers[t].state = 0;
    b->waves[w].arrived = 0;
}

void block_barrier() {
    Block* b = cur_block();
    Fiber& f = b->fibers[b->cur];
    if (++b->arrived == b->live) {   // last one in: everybody (including this fiber) goes on
        release_block(b);
        return;
    }
    f.state = 1;
    HIPEMU_TO_SCHED(f, b);
}

void wave_barrier() {
    Block* b = cur_block();
    const unsigned w = b->cur / 64;
    Fiber& f = b->fibers[b->cur];
    if (++b->waves[w].arrived == b->waves[w].live) {
        release_wave(b, w);
        return;
    }
    f.state = 2;
    HIPEMU_TO_SCHED(f, b);
}

static void trampoline() {
    Block* b = cur_block();
    b->body();
    // the thread has returned: it no longer takes part in barriers
    const unsigned t = b->cur, w = t / 64;
    b->fibers[t].state = 3;
    b->live--;
    b->waves[w].live--;
    if (b->live > 0 && b->arrived == b->live) release_block(b);
    if (b->waves[w].live > 0 && b->waves[w].arrived == b->waves[w].live) release_wave(b, w);
    HIPEMU_TO_SCHED(b->fibers[t], b);   // never resumed
}

void run_block(unsigned nthreads, size_t shmem, dim3 block_dim, dim3 grid_dim, Idx block_idx, const std::function<void()>& body) {
    static thread_local Block blk;
    Block* b = &blk;
    cur_block() = b;
    if (b->fibers.size() < nthreads) b->fibers.resize(nthreads);
    b->nthreads = nthreads;
    b->live = nthreads;
    b->arrived = 0;
    b->waves.assign((nthreads + 63) / 64, Wave());
    for (unsigned t = 0; t < nthreads; t++) b->waves[t / 64].live++;
    if (b->smem.size() < shmem + 64) b->smem.resize(shmem + 64);
    b->body = body;
    blockDim = block_dim;
    gridDim = grid_dim;
    blockIdx = block_idx;
    auto set_tid = [&](unsigned t) {
        threadIdx.x = t % block_dim.x;
        threadIdx.y = (t / block_dim.x) % block_dim.y;
        threadIdx.z = t / (block_dim.x * block_dim.y);
    };
    for (unsigned t = 0; t < nthreads; t++) {
        Fiber& f = b->fibers[t];
        if (!f.stack) {
            f.stack = (char*)mmap(nullptr, STACK_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE | MAP_STACK, -1, 0);
            if (f.stack == MAP_FAILED) {
                fprintf(stderr, "hipemu: cannot map a fiber stack\n");
                abort();
            }
        }
#ifdef HIPEMU_ASM_SWITCH
        f.sp = fiber_init(f.stack);
#else
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack;
        f.ctx.uc_stack.ss_size = STACK_BYTES;
        f.ctx.uc_link = nullptr;
        makecontext(&f.ctx, trampoline, 0);
#endif
        f.state = 0;
    }
    unsigned done = 0;
    while (done < nthreads) {
        bool progress = false;
        for (unsigned t = 0; t < nthreads; t++) {
            Fiber& f = b->fibers[t];
            if (f.state != 0) continue;
            b->cur = t;
            set_tid(t);
            HIPEMU_TO_FIBER(b, f);
            progress = true;
            if (f.state == 3) done++;
        }
        if (!progress) {
            fprintf(stderr, "hipemu: barrier deadlock (a __syncthreads/__shfl not reached by every live thread)\n");
            abort();
        }
    }
}

// ---- worker pool: blocks of one launch are independent, so they run on up to 8 OS threads -------------------------------
struct Pool {
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::vector<std::thread> threads;
    uint64_t generation = 0;
    unsigned pending = 0;
    bool stop = false;
    // current job
    dim3 grid, block;
    size_t shmem = 0;
    const std::function<void()>* body = nullptr;
    std::atomic<uint64_t> next{0};
    uint64_t total = 0;

    void work() {
        for (;;) {
            uint64_t i = next.fetch_add(1);
            if (i >= total) return;
            Idx bi;
            bi.x = (unsigned)(i % grid.x);
            bi.y = (unsigned)((i / grid.x) % grid.y);
            bi.z = (unsigned)(i / ((uint64_t)grid.x * grid.y));
            run_block(block.x * block.y * block.z, shmem, block, grid, bi, *body);
        }
    }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> l(mu);
                cv_work.wait(l, [&] { return stop || generation != seen; });
                if (stop) return;
                seen = generation;
            }
            work();
            std::unique_lock<std::mutex> l(mu);
            if (--pending == 0) cv_done.notify_all();
        }
    }
    Pool() {
        unsigned n = std::thread::hardware_concurrency();
        n = n == 0 ? 4 : (n > 8 ? 8 : n);
        for (unsigned i = 0; i < n; i++) threads.emplace_back([this] { loop(); });
    }
    ~Pool() {
        {
            std::unique_lock<std::mutex> l(mu);
            stop = true;
        }
        cv_work.notify_all();
        for (auto& t : threads) t.join();
    }
};

void run_grid(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body) {
    static Pool pool;
    static std::mutex launch_mu;   // one launch at a time (the product serialises launches per context anyway)
    std::lock_guard<std::mutex> g(launch_mu);
    const uint64_t total = (uint64_t)grid.x * grid.y * grid.z;
    if (total == 0) return;
    std::unique_lock<std::mutex> l(pool.mu);
    pool.grid = grid;
    pool.block = block;
    pool.shmem = shmem;
    pool.body = &body;
    pool.total = total;
    pool.next = 0;
    pool.pending = (unsigned)pool.threads.size();
    pool.generation++;
    pool.cv_work.notify_all();
    pool.cv_done.wait(l, [&] { return pool.pending == 0; });
}

}  // namespace hipemu

namespace ga {
struct Ctx;
int util_microbench(Ctx*, char* buf, size_t cap) {
    snprintf(buf, cap, "emulation=1;");
    return 0;
}
int util_clock_probe(Ctx*, uint32_t, double* mhz_out) {
    *mhz_out = 0.0;   // no shader clock under the emulation
    return 0;
}
}  // namespace ga
