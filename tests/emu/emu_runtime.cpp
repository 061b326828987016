// TEST INFRASTRUCTURE — CPU SIMT emulator for the one-wave-per-game kernels.
//
// Runs the *product kernel source* (sprl_amd/csrc/step_kernel.h, compiled with -DSPRL_EMU) on the CPU: a
// "wave" is 64 cooperative fibers (ucontext) in one OS thread; every cross-lane primitive of wave.h is a
// rendezvous at which each lane deposits a value and resumes once all 64 have arrived.  The scheduler
// asserts that all lanes sit at the same collective (same call site, same sequence number), i.e. it
// detects exactly the divergence bugs that would be silent data corruption on the GPU.  Blocks are spread
// over a small thread pool.  Nothing here is linked into, or reachable from, the product library.
#include <ucontext.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

namespace emu {

typedef void (*block_fn)(void* arg, int block);

struct Wave {
    ucontext_t sched;
    ucontext_t ctx[64];
    char* stacks = nullptr;
    bool done[64];
    uint64_t slots[2][64];
    int sites[2][64];
    long phase[64];
    block_fn fn;
    void* arg;
    int block;
};

static constexpr size_t STACK_BYTES = 256 * 1024;
static thread_local Wave* tl_wave = nullptr;
static thread_local int tl_lane = 0;

int lane() { return tl_lane; }
int block() { return tl_wave->block; }

const uint64_t* exchange(uint64_t v, int site) {
    Wave* w = tl_wave;
    const int l = tl_lane;
    const long ph = w->phase[l]++;
    w->slots[ph & 1][l] = v;
    w->sites[ph & 1][l] = site;
    swapcontext(&w->ctx[l], &w->sched);
    // resumed by the scheduler once every lane has deposited its value for collective `ph`
    return w->slots[ph & 1];
}

void fatal(const char* what, const uint64_t* slots) {
    fprintf(stderr, "emu: %s (block %d, lane %d)\n  slots:", what, tl_wave->block, tl_lane);
    for (int i = 0; i < 64; ++i) fprintf(stderr, " %llx", (unsigned long long)slots[i]);
    fprintf(stderr, "\n");
    abort();
}

static void fiber_main() {
    Wave* w = tl_wave;
    const int l = tl_lane;
    w->fn(w->arg, w->block);
    w->done[l] = true;
    swapcontext(&w->ctx[l], &w->sched);
}

static void run_block(Wave* w, block_fn fn, void* arg, int block) {
    w->fn = fn;
    w->arg = arg;
    w->block = block;
    tl_wave = w;
    for (int l = 0; l < 64; ++l) {
        w->done[l] = false;
        w->phase[l] = 0;
        getcontext(&w->ctx[l]);
        w->ctx[l].uc_stack.ss_sp = w->stacks + (size_t)l * STACK_BYTES;
        w->ctx[l].uc_stack.ss_size = STACK_BYTES;
        w->ctx[l].uc_link = &w->sched;
        makecontext(&w->ctx[l], (void (*)())fiber_main, 0);
    }
    for (;;) {
        int ndone = 0;
        for (int l = 0; l < 64; ++l) {
            if (w->done[l]) { ++ndone; continue; }
            tl_lane = l;
            swapcontext(&w->sched, &w->ctx[l]);
            if (w->done[l]) ++ndone;
        }
        if (ndone == 64) break;
        if (ndone != 0) {
            fprintf(stderr, "emu: divergence in block %d: %d lanes finished while others wait at a collective\n",
                    block, ndone);
            abort();
        }
        const long ph = w->phase[0] - 1;
        for (int l = 1; l < 64; ++l) {
            if (w->phase[l] != w->phase[0] || w->sites[ph & 1][l] != w->sites[ph & 1][0]) {
                fprintf(stderr, "emu: divergence in block %d: lane %d at collective #%ld (site %d), lane 0 at #%ld (site %d)\n",
                        block, l, w->phase[l] - 1, w->sites[(w->phase[l] - 1) & 1][l], ph, w->sites[ph & 1][0]);
                abort();
            }
        }
    }
}

void launch(block_fn fn, void* arg, int nblocks) {
    unsigned hw = std::thread::hardware_concurrency();
    int nthreads = (int)(hw ? hw : 4);
    if (nthreads > nblocks) nthreads = nblocks;
    if (nthreads < 1) nthreads = 1;
    std::atomic<int> next{0};
    auto worker = [&]() {
        Wave* w = new Wave();
        w->stacks = (char*)malloc(64 * STACK_BYTES);
        for (;;) {
            int b = next.fetch_add(1);
            if (b >= nblocks) break;
            run_block(w, fn, arg, b);
        }
        free(w->stacks);
        delete w;
    };
    if (nthreads == 1) {
        worker();
        return;
    }
    std::vector<std::thread> ts;
    for (int i = 0; i < nthreads; ++i) ts.emplace_back(worker);
    for (auto& t : ts) t.join();
}

}  // namespace emu
