// TEST INFRASTRUCTURE ONLY -- runtime of the CPU kernel emulator (see include/hip/hip_runtime.h).
// Work-items of a workgroup are fibers on one OS thread; workgroups are spread over OS threads.
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <atomic>
#include <mutex>
#include <thread>
#include <vector>

// AddressSanitizer build (tests/emu/build_emu.py sanitize=True): the fibers' stack switches are announced to the sanitizer, and a launch's
// dynamic LDS is an allocation of EXACTLY the requested size (red zones right behind it) instead of a reused, larger one.
#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define EMU_ASAN 1
#endif
#endif
#ifdef EMU_ASAN
#include <sanitizer/common_interface_defs.h>
#endif

extern "C" void emu_ctx_switch(void** from_sp, void* to_sp);
asm(R"(
.text
.globl emu_ctx_switch
.type emu_ctx_switch,@function
emu_ctx_switch:
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
.size emu_ctx_switch,.-emu_ctx_switch
)");

namespace emu {

static constexpr size_t kStack = 256 * 1024;
static constexpr int kMaxThreads = 1024;

struct Wave {
    int first, nlanes, alive, arrived, gen;
    alignas(16) char slots[64][16];
};
struct Fiber {
    void* sp;
    dim3 tid;
    int flat;
    Wave* wave;
    bool done;
    void* fake;          // EMU_ASAN: the sanitizer's fake-stack handle of this fiber while it is switched out
    const void* bottom;  // EMU_ASAN: lowest address of the fiber's stack
};
struct Block {
    Fiber fib[kMaxThreads];
    Wave waves[kMaxThreads / 64];
    int n, alive, arrived, gen;
    void* main_sp;
    const std::function<void()>* body;
    char* stacks;
    char* dyn;
    size_t dyn_cap;
};

thread_local Fiber* cur = nullptr;
thread_local dim3 tl_block_idx;
thread_local Block* tl_blk = nullptr;
dim3 g_grid, g_block;

dim3& cur_tid() { return cur->tid; }
void* dyn_smem() { return tl_blk->dyn; }
int lane_id() { return cur->flat & 63; }
void* wave_slot(int lane) { return cur->wave->slots[lane]; }

static void switch_to(Fiber* f) {
    Fiber* prev = cur;
    cur = f;
#ifdef EMU_ASAN
    __sanitizer_start_switch_fiber(prev->done ? nullptr : &prev->fake, f->bottom, kStack);
#endif
    emu_ctx_switch(&prev->sp, f->sp);
#ifdef EMU_ASAN
    __sanitizer_finish_switch_fiber(prev->fake, nullptr, nullptr);     // (back on `prev`)
#endif
}

static void die(const char* msg) {
    fprintf(stderr, "[emu] fatal: %s (block %u,%u,%u)\n", msg, tl_block_idx.x, tl_block_idx.y, tl_block_idx.z);
    abort();
}

static void yield_block() {
    Block* b = tl_blk;
    int i = cur->flat;
    for (int s = 1; s <= b->n; ++s) {
        Fiber* f = &b->fib[(i + s) % b->n];
        if (!f->done && f != cur) { switch_to(f); return; }
    }
    die("deadlock: work-item waits at a barrier but no other work-item is alive");
}

static void yield_wave() {
    Wave* w = cur->wave;
    Block* b = tl_blk;
    int i = cur->flat - w->first;
    for (int s = 1; s <= w->nlanes; ++s) {
        Fiber* f = &b->fib[w->first + (i + s) % w->nlanes];
        if (!f->done && f != cur) { switch_to(f); return; }
    }
    die("deadlock: lane waits at a wave collective but no other lane of the wave is alive");
}

void yield_any() {
    Block* b = tl_blk;
    int i = cur->flat;
    for (int s = 1; s < b->n; ++s) {
        Fiber* f = &b->fib[(i + s) % b->n];
        if (!f->done && f != cur) { switch_to(f); return; }
    }
    std::this_thread::yield();  // alone in the workgroup: let the other workgroups' OS threads run
}

void block_barrier() {
    Block* b = tl_blk;
    b->arrived++;
    const int g = b->gen;
    if (b->arrived >= b->alive) { b->arrived = 0; b->gen++; return; }
    while (b->gen == g) yield_block();
}

void wave_barrier() {
    Wave* w = cur->wave;
    w->arrived++;
    const int g = w->gen;
    if (w->arrived >= w->alive) { w->arrived = 0; w->gen++; return; }
    while (w->gen == g) yield_wave();
}

#ifdef EMU_ASAN
thread_local const void* tl_main_bottom = nullptr;
thread_local size_t tl_main_size = 0;
#endif

static void fiber_main() {
    Block* b = tl_blk;
#ifdef EMU_ASAN
    {   // first time on this stack; whoever switched here came either from the OS thread's stack (its bounds are remembered) or a fiber's
        const void* ob; size_t os;
        __sanitizer_finish_switch_fiber(nullptr, &ob, &os);
        if (cur->flat == 0) { tl_main_bottom = ob; tl_main_size = os; }
    }
#endif
    (*b->body)();
    Fiber* me = cur;
    me->done = true;
    b->alive--;
    me->wave->alive--;
    // a finished work-item no longer participates in barriers (s_barrier counts live waves only)
    if (b->arrived > 0 && b->arrived >= b->alive) { b->arrived = 0; b->gen++; }
    if (me->wave->arrived > 0 && me->wave->arrived >= me->wave->alive) { me->wave->arrived = 0; me->wave->gen++; }
    for (int s = 1; s <= b->n; ++s) {
        Fiber* f = &b->fib[(me->flat + s) % b->n];
        if (!f->done) { switch_to(f); die("resumed a finished fiber"); }
    }
    // last one out: back to the OS thread
    void* dummy;
    cur = nullptr;
#ifdef EMU_ASAN
    __sanitizer_start_switch_fiber(nullptr, tl_main_bottom, tl_main_size);
#endif
    emu_ctx_switch(&dummy, b->main_sp);
    die("unreachable");
}

static std::mutex g_pool_mu;
static std::vector<Block*> g_pool;

static void release_block_ctx(Block* b) {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    g_pool.push_back(b);
}

static Block* get_block_ctx(size_t shmem) {
    Block* blk = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        if (!g_pool.empty()) { blk = g_pool.back(); g_pool.pop_back(); }
    }
    if (!blk) {
        blk = new Block();
        blk->stacks = (char*)mmap(nullptr, kStack * kMaxThreads, PROT_READ | PROT_WRITE,
                                  MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (blk->stacks == (char*)MAP_FAILED) { perror("mmap"); abort(); }
        blk->dyn = nullptr;
        blk->dyn_cap = 0;
    }
#ifdef EMU_ASAN
    // exactly `shmem` bytes ending at the allocation's end: an LDS access past the size the launch asked for lands in a red zone
    free(blk->dyn);
    blk->dyn = shmem ? (char*)aligned_alloc(16, (shmem + 15) / 16 * 16) : nullptr;
    blk->dyn_cap = shmem;
#else
    if (shmem > blk->dyn_cap) {
        free(blk->dyn);
        blk->dyn = (char*)aligned_alloc(64, (shmem + 63) / 64 * 64);
        blk->dyn_cap = shmem;
    }
#endif
    return blk;
}

static void run_block(Block* b, dim3 bidx, dim3 block, size_t shmem, const std::function<void()>& body) {
    tl_blk = b;
    tl_block_idx = bidx;
    const int n = (int)(block.x * block.y * block.z);
    if (n > kMaxThreads) die("block too large");
    b->n = n; b->alive = n; b->arrived = 0; b->gen = 0; b->body = &body;
    if (shmem) memset(b->dyn, 0xFF, shmem);  // LDS is uninitialised on hardware: poison with NaNs
    const int nw = (n + 63) / 64;
    for (int w = 0; w < nw; ++w) {
        Wave& W = b->waves[w];
        W.first = w * 64;
        W.nlanes = (n - w * 64 < 64) ? (n - w * 64) : 64;
        W.alive = W.nlanes; W.arrived = 0; W.gen = 0;
    }
    for (int i = 0; i < n; ++i) {
        Fiber& f = b->fib[i];
        f.flat = i;
        f.tid = dim3(i % block.x, (i / block.x) % block.y, i / (block.x * block.y));
        f.wave = &b->waves[i / 64];
        f.done = false;
        f.fake = nullptr;
        f.bottom = b->stacks + (size_t)i * kStack;
        uintptr_t top = (uintptr_t)(b->stacks + (size_t)(i + 1) * kStack);
        top &= ~(uintptr_t)15;
        void** sp = (void**)top;
        *--sp = nullptr;               // fake return address of fiber_main
        *--sp = (void*)&fiber_main;    // 'ret' target of the first switch
        for (int r = 0; r < 6; ++r) *--sp = nullptr;
        f.sp = sp;
    }
    cur = &b->fib[0];
#ifdef EMU_ASAN
    void* main_fake = nullptr;
    __sanitizer_start_switch_fiber(&main_fake, b->fib[0].bottom, kStack);
#endif
    emu_ctx_switch(&b->main_sp, b->fib[0].sp);
#ifdef EMU_ASAN
    __sanitizer_finish_switch_fiber(main_fake, nullptr, nullptr);
#endif
    cur = nullptr;
}

void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body) {
    g_grid = grid;
    g_block = block;
    const long total = (long)grid.x * grid.y * grid.z;
    if (total <= 0) return;
    unsigned hw = std::thread::hardware_concurrency();
    const char* env = getenv("AICG_EMU_THREADS");
    if (env) hw = (unsigned)atoi(env);
    if (hw < 4) hw = 4;  // kernels whose workgroups wait for each other (the 4-workgroup GRU: a quad is contiguous in dispatch order)
                         // need their partners running concurrently, even if that means time-sharing one core -- also when
                         // AICG_EMU_THREADS asks for fewer
    const long nthreads = total < (long)hw ? total : (long)hw;
    std::atomic<long> next{0};
    auto worker = [&]() {
        Block* ctx = get_block_ctx(shmem);
        for (;;) {
            long i = next.fetch_add(1);
            if (i >= total) break;
            dim3 b((unsigned)(i % grid.x), (unsigned)((i / grid.x) % grid.y), (unsigned)(i / ((long)grid.x * grid.y)));
            run_block(ctx, b, block, shmem, body);
        }
        release_block_ctx(ctx);
    };
    if (nthreads <= 1) { worker(); return; }
    std::vector<std::thread> th;
    for (long t = 0; t < nthreads; ++t) th.emplace_back(worker);
    for (auto& t : th) t.join();
}

}  // namespace emu
