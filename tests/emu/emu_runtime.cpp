// TEST INFRASTRUCTURE ONLY: fiber scheduler behind the CPU emulation of the HIP device language
// (see hip/hip_runtime.h).
#include "hip/hip_runtime.h"
#include <ucontext.h>
#include <cstdlib>
#include <cstdio>

emu_dim3 threadIdx, blockIdx, blockDim, gridDim;

namespace emu {
uint64_t g_xbuf[1024];
uint64_t g_ybuf[1024];

static ucontext_t g_sched;
static std::vector<ucontext_t> g_ctx;
static std::vector<char *> g_stack;
static std::vector<int> g_alive;
static const std::function<void()> *g_body = nullptr;
static int g_cur = 0;
static const size_t kStack = 1 << 20;

static void trampoline() {
    (*g_body)();
    g_alive[g_cur] = 0;
    swapcontext(&g_ctx[g_cur], &g_sched);
}

// A fiber parks at a workgroup barrier (2) or at a wavefront-scope exchange (1: shuffles, ballots, matrix instructions).  The
// scheduler releases a wave when every live lane of it is parked at scope 1, and the workgroup when every live fiber is parked
// at scope 2 -- so wave-uniform branches around wave operations (one wave skips a tile product) keep their meaning.
static std::vector<int> g_wait;
void barrier() { g_wait[g_cur] = 2; swapcontext(&g_ctx[g_cur], &g_sched); }
void wave_barrier() { g_wait[g_cur] = 1; swapcontext(&g_ctx[g_cur], &g_sched); }

void launch_impl(unsigned grid, unsigned block, const std::function<void()> &body) {
    g_body = &body;
    if (g_stack.size() < block) {
        for (size_t i = g_stack.size(); i < block; ++i) g_stack.push_back((char *)std::malloc(kStack));
    }
    g_ctx.resize(block);
    g_alive.assign(block, 1);
    gridDim.x = grid; blockDim.x = block;
    for (unsigned b = 0; b < grid; ++b) {
        blockIdx.x = b;
        for (unsigned t = 0; t < block; ++t) {
            getcontext(&g_ctx[t]);
            g_ctx[t].uc_stack.ss_sp = g_stack[t];
            g_ctx[t].uc_stack.ss_size = kStack;
            g_ctx[t].uc_link = &g_sched;
            makecontext(&g_ctx[t], trampoline, 0);
            g_alive[t] = 1;
        }
        g_wait.assign(block, 0);
        bool any = true;
        while (any) {
            any = false;
            bool ran = false;
            for (unsigned t = 0; t < block; ++t)
                if (g_alive[t] && g_wait[t] == 0) {
                    g_cur = (int)t; threadIdx.x = t;
                    swapcontext(&g_sched, &g_ctx[t]);
                    ran = true;
                }
            bool released = false, all_block = true;
            for (unsigned w0 = 0; w0 < block; w0 += 64) {
                bool some = false, all_wave = true;
                for (unsigned t = w0; t < w0 + 64 && t < block; ++t)
                    if (g_alive[t]) { any = true; some = true; all_wave = all_wave && g_wait[t] == 1; all_block = all_block && g_wait[t] == 2; }
                if (some && all_wave) {
                    for (unsigned t = w0; t < w0 + 64 && t < block; ++t) g_wait[t] = 0;
                    released = true;
                }
            }
            if (any && all_block) {
                for (unsigned t = 0; t < block; ++t) g_wait[t] = 0;
                released = true;
            }
            if (any && !ran && !released) {
                std::fprintf(stderr, "emu: deadlock -- lanes of one wave wait at different scopes (divergent barrier)\n");
                std::abort();
            }
        }
    }
}
}  // namespace emu
