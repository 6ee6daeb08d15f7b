// hipemu_runtime.cpp -- fiber scheduler behind tests/hipemu/include/hip/hip_runtime.h (test infrastructure only).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdlib>

// Fiber switch: callee-saved registers + the two floating-point control words on the fiber's own stack, then the stack pointers are
// exchanged.  (swapcontext() makes a signal-mask system call on every switch: an emulated solve is tens of millions of switches, and the
// CPU suite spent most of its eleven minutes in the kernel.)  x86-64 System V only -- like the rest of this emulator's build.
extern "C" void hipemu_switch(void **save_sp, void *const *load_sp);
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
    subq $8, %rsp
    stmxcsr (%rsp)
    fnstcw 4(%rsp)
    movq %rsp, (%rdi)
    movq (%rsi), %rsp
    ldmxcsr (%rsp)
    fldcw 4(%rsp)
    addq $8, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size hipemu_switch, .-hipemu_switch
)");

namespace hipemu {

ThreadCtx *g_cur = nullptr;
const char *g_kernel_name = "?";
alignas(64) static char dyn_smem_pool[160 * 1024];
char *g_dyn_smem = dyn_smem_pool;

namespace {
constexpr size_t kStack = 256 * 1024;
struct Fiber {
    void *sp = nullptr; // saved stack pointer while the fiber is not running
    // what the fiber waits for (the scheduler does not resume it before that has happened: a barrier of 256 fibers was 256 pointless
    // switches per scheduler pass otherwise): the block barrier's / its wave exchange's generation it arrived in, -1 = runnable
    int wait_bar_gen = -1, wait_wave_gen = -1;
    ThreadCtx tc;
    bool done = false;
    char *stack = nullptr;
};
struct WaveState {
    unsigned char buf[64 * 32];
    int arrived = 0, generation = 0, alive = 0, readers = 0;
};
std::vector<Fiber> fibers;
std::vector<WaveState> waves;
void *sched_sp = nullptr;
int n_threads = 0, bar_arrived = 0, bar_generation = 0, n_done = 0;
long g_sync_events = 0; // completed barriers + wave exchanges (progress for the deadlock detector)
long launch_counter = 0;
int g_order = [] { const char *e = std::getenv("HIPEMU_ORDER"); return !e ? 0 : (e[0] == 'r' ? 1 : (e[0] == 'i' ? 2 : 0)); }();
std::function<void()> *cur_body = nullptr;
Fiber *cur_fiber = nullptr;

void yield() {
    Fiber *f = cur_fiber;
    hipemu_switch(&f->sp, &sched_sp);
}
void trampoline() {
    (*cur_body)();
    cur_fiber->done = true;
    ++n_done;
    waves[cur_fiber->tc.wave].alive--;
    hipemu_switch(&cur_fiber->sp, &sched_sp);
    std::abort(); // a finished fiber is never resumed
}
// A fresh fiber: the frame hipemu_switch pops (control words, six registers, return address = trampoline) at the top of its stack, laid out
// so that trampoline() starts with the stack alignment of a called function.
void prepare(Fiber &f) {
    uintptr_t top = (reinterpret_cast<uintptr_t>(f.stack) + kStack) & ~(uintptr_t)15;
    uint64_t *p = reinterpret_cast<uint64_t *>(top);
    *--p = 0;                                        // where a caller's return address would be
    *--p = reinterpret_cast<uint64_t>(&trampoline);  // popped by `ret`
    for (int k = 0; k < 6; ++k) *--p = 0;            // rbp rbx r12 r13 r14 r15
    *--p = 0x1F80u | ((uint64_t)0x037Fu << 32);      // MXCSR, x87 control word: the defaults
    f.sp = p;
}
} // namespace

void spin_yield() { yield(); }

void syncthreads() {
    if (n_done != 0) {
        std::fprintf(stderr, "hipemu: __syncthreads() reached after %d thread(s) of the block already returned\n", n_done);
        std::abort();
    }
    int gen = bar_generation;
    if (++bar_arrived == n_threads) {
        bar_arrived = 0;
        ++bar_generation;
        ++g_sync_events;
    } else {
        cur_fiber->wait_bar_gen = gen;
        while (bar_generation == gen) yield();
        cur_fiber->wait_bar_gen = -1;
    }
}

void wave_exchange(const void *in, void *out_all, size_t elem) {
    WaveState &w = waves[g_cur->wave];
    if (elem > 32) std::abort();
    // phase 0: wait until the previous exchange has been fully read
    while (w.readers != 0) yield();
    std::memcpy(w.buf + (size_t)g_cur->lane * elem, in, elem);
    int gen = w.generation;
    int n_in_wave = w.alive;
    if (++w.arrived == n_in_wave) {
        w.arrived = 0;
        w.readers = n_in_wave;
        ++w.generation;
        ++g_sync_events;
    } else {
        cur_fiber->wait_wave_gen = gen;
        while (w.generation == gen) yield();
        cur_fiber->wait_wave_gen = -1;
    }
    std::memcpy(out_all, w.buf, 64 * elem);
    --w.readers;
}

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

std::vector<Guarded> &guarded() {
    static std::vector<Guarded> g;
    return g;
}
void check_guards(const char *where) {
    for (const Guarded &a : guarded())
        for (int k = -256; k < 256; ++k)
            if ((unsigned char)(k < 0 ? a.p[k] : a.p[a.n + k]) != 0x5C) {
                std::fprintf(stderr, "hipemu: %s (kernel %s): write %d bytes past the end of a device allocation of %zu bytes\n", where, g_kernel_name ? g_kernel_name : "?", k, a.n);
                std::abort();
            }
}
void launch(dim3 grid, dim3 block, size_t shmem, Stream *s, std::function<void()> body) {
    auto run = [grid, block, shmem, body]() mutable {
        ++launch_counter;
        const int nt = (int)(block.x * block.y * block.z);
        if (shmem > sizeof(dyn_smem_pool)) {
            std::fprintf(stderr, "hipemu: dynamic LDS request %zu exceeds 160 KiB\n", shmem);
            std::abort();
        }
        if ((int)fibers.size() < nt) {
            size_t old = fibers.size();
            fibers.resize(nt);
            for (size_t i = old; i < fibers.size(); ++i) fibers[i].stack = (char *)std::malloc(kStack);
        }
        waves.resize((nt + 63) / 64);
        for (unsigned bz = 0; bz < grid.z; ++bz)
            for (unsigned by = 0; by < grid.y; ++by)
                for (unsigned bx = 0; bx < grid.x; ++bx) {
                    n_threads = nt;
                    bar_arrived = 0;
                    n_done = 0;
                    for (auto &w : waves) w.arrived = 0, w.alive = 0, w.readers = 0;
                    cur_body = &body;
                    for (int t = 0; t < nt; ++t) {
                        Fiber &f = fibers[t];
                        f.done = false;
                        f.wait_bar_gen = -1, f.wait_wave_gen = -1;
                        f.tc.tIdx = {(unsigned)(t % block.x), (unsigned)((t / block.x) % block.y), (unsigned)(t / (block.x * block.y))};
                        f.tc.bIdx = {bx, by, bz};
                        f.tc.bDim = block;
                        f.tc.gDim = grid;
                        f.tc.linear_tid = t;
                        f.tc.lane = t & 63;
                        f.tc.wave = t >> 6;
                        waves[t >> 6].alive++;
                        prepare(f);
                    }
                    int remaining = nt;
                    long spins = 0, seen_events = g_sync_events;
                    while (remaining > 0) {
                        int progressed = 0;
                        for (int tt = 0; tt < nt; ++tt) {
                            // HIPEMU_ORDER=reverse / interleave: run the fibers of a barrier interval in another order, to
                            // flush out code that only works because thread 0 happens to run first
                            const int t = g_order == 1 ? nt - 1 - tt : (g_order == 2 ? ((tt & 1) ? nt - 1 - (tt >> 1) : (tt >> 1)) : tt);
                            Fiber &f = fibers[t];
                            if (f.done) continue;
                            if (f.wait_bar_gen >= 0 && f.wait_bar_gen == bar_generation) continue;               // still at the barrier
                            if (f.wait_wave_gen >= 0 && f.wait_wave_gen == waves[f.tc.wave].generation) continue; // still in the exchange
                            cur_fiber = &f;
                            g_cur = &f.tc;
                            hipemu_switch(&sched_sp, &f.sp);
                            if (f.done) {
                                --remaining;
                                ++progressed;
                            }
                        }
                        // a completed barrier / wave exchange is progress too: a long kernel may pass tens of thousands of them
                        // before its first thread retires
                        if (g_sync_events != seen_events) seen_events = g_sync_events, spins = 0;
                        if (!progressed && ++spins > 20000) {
                            std::fprintf(stderr, "hipemu: kernel %s (%d threads) block (%u,%u,%u) deadlocked: bar_arrived=%d n_done=%d (divergent barrier / shuffle?)\n", g_kernel_name, nt, bx, by, bz, bar_arrived, n_done);
                            for (size_t w = 0; w < waves.size(); ++w) std::fprintf(stderr, "  wave %zu: arrived=%d alive=%d readers=%d\n", w, waves[w].arrived, waves[w].alive, waves[w].readers);
                            std::fprintf(stderr, "  threads still running:");
                            for (int t = 0, shown = 0; t < nt && shown < 16; ++t)
                                if (!fibers[t].done) std::fprintf(stderr, " %d", t), ++shown;
                            std::fprintf(stderr, "\n");
                            std::abort();
                        }
                        if (progressed) spins = 0;
                    }
                }
        g_cur = nullptr;
        check_guards("after the launch");
    };
    if (s && s->capturing)
        s->graph->push_back(run);
    else
        run();
}

} // namespace hipemu
