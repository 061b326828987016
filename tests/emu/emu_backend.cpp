// TEST INFRASTRUCTURE — backend.h on the CPU SIMT emulator (see emu_runtime.cpp).  Builds
// tests/emu/libsprl_emu.so = engine.cpp + the product kernel source compiled with -DSPRL_EMU, so that
// `pytest -m "not gpu"` can check kernel logic against the oracle without a GPU.  Never shipped, never
// loaded by the sprl_amd package.
#include <stdlib.h>
#include <string.h>

#include <stdio.h>

#include <chrono>
#include <map>
#include <mutex>
#include <string>

#include "../../sprl_amd/csrc/backend.h"
#include "../../sprl_amd/csrc/records_kernel.h"
#include "../../sprl_amd/csrc/step_kernel.h"
#include "../../sprl_amd/csrc/step_kernel_wide.h"

namespace emu {
typedef void (*block_fn)(void* arg, int block);
void launch(block_fn fn, void* arg, int nblocks);
}

namespace {
alignas(16) static thread_local unsigned char tl_lds[160 * 1024];

template <class G>
void block_entry(void* arg, int block) {
    const EngineParams* P = (const EngineParams*)arg;
    static_assert(sizeof(sprl::WaveLds<G>) <= sizeof(tl_lds), "emulated LDS too small");
    sprl::step_game<G>(*P, block, reinterpret_cast<sprl::WaveLds<G>*>(tl_lds));
}
template <class G>
void block_entry_match(void* arg, int block) {
    const EngineParams* P = (const EngineParams*)arg;
    sprl::step_match<G>(*P, block, reinterpret_cast<sprl::WaveLds<G>*>(tl_lds));
}
template <class G>
void block_entry_match_wide(void* arg, int block) {
    const EngineParams* P = static_cast<const EngineParams*>(arg);
    sprlw::step_match<G>(*P, block, reinterpret_cast<sprlw::WaveLdsW<G>*>(tl_lds));
}

template <class G>
void block_entry_wide(void* arg, int block) {
    const EngineParams* P = (const EngineParams*)arg;
    static_assert(sizeof(sprlw::WaveLdsW<G>) <= sizeof(tl_lds), "emulated LDS too small");
    sprlw::step_game<G>(*P, block, reinterpret_cast<sprlw::WaveLdsW<G>*>(tl_lds));
}
}  // namespace

namespace be {
const char* name() { return "cpu-simt-emulator (tests only)"; }
static const char* g_err = "emulator error";
const char* last_error() { return g_err; }
bool available(std::string*) { return true; }
int init(int, std::string*) { return 0; }
// Test hook of the EMULATOR only (never in the product library): SPRL_EMU_HBM_BYTES bounds the live "device" memory so that the
// host code's out-of-memory paths (SPRL_E_NOMEM, the worker's back-off) can be exercised on the CPU; SPRL_EMU_HBM_REPORT prints the
// high-water mark at exit.
static std::mutex g_mem_mu;
static std::map<void*, size_t> g_mem_live;
static size_t g_mem_now = 0, g_mem_peak = 0;
static void mem_report() { fprintf(stderr, "emu hbm peak bytes: %zu\n", g_mem_peak); }
void* dmalloc(size_t bytes) {
    std::lock_guard<std::mutex> lock(g_mem_mu);
    static const char* lim = getenv("SPRL_EMU_HBM_BYTES");
    static const bool report = getenv("SPRL_EMU_HBM_REPORT") != nullptr && atexit(mem_report) == 0;
    (void)report;
    if (lim && g_mem_now + bytes > (size_t)strtoull(lim, nullptr, 10)) {
        g_err = "emulated HBM exhausted";
        return nullptr;
    }
    void* p = calloc(1, bytes);
    if (p) {
        g_mem_live[p] = bytes;
        g_mem_now += bytes;
        if (g_mem_now > g_mem_peak) g_mem_peak = g_mem_now;
    }
    return p;
}
void dfree(void* p) {
    std::lock_guard<std::mutex> lock(g_mem_mu);
    auto it = g_mem_live.find(p);
    if (it != g_mem_live.end()) {
        g_mem_now -= it->second;
        g_mem_live.erase(it);
    }
    free(p);
}
int h2d(void* dst, const void* src, size_t n) { memcpy(dst, src, n); return 0; }
int d2h(void* dst, const void* src, size_t n) { memcpy(dst, src, n); return 0; }
int dmemset(void* dst, int v, size_t n) { memset(dst, v, n); return 0; }
int sync() { return 0; }
static int g_stream_token;                       // own_stream on the emulator: kernels run synchronously on the calling thread, a
void* stream_create() { return &g_stream_token; }
void* stream_create_priority(int) { return &g_stream_token; }      // (launches are synchronous here: events and waits are no-ops)
void* event_new() { return &g_stream_token; }
void event_free(void*) {}
void event_record(void*, void*) {}
void stream_wait(void*, void*) {}   // "stream" is just a non-null token (engines on different host threads are independent)
void stream_destroy(void*) {}
void set_stream(void*) {}
void bind(int, void*) {}
void* current_stream() { return nullptr; }
int launch_step(int game, const EngineParams& P) {
    EngineParams copy = P;
    if (game == SPRL_GAME_OTHELLO) emu::launch(block_entry<Othello>, &copy, P.num_slots);
    else if (game == SPRL_GAME_GO7) emu::launch(block_entry<Go7>, &copy, P.num_slots);
    else if (game == SPRL_GAME_GO9) emu::launch(block_entry_wide<GoN<9>>, &copy, P.num_slots);
    else if (game == SPRL_GAME_GO19) emu::launch(block_entry_wide<GoN<19>>, &copy, P.num_slots);
    else if (game == SPRL_GAME_GO7W) emu::launch(block_entry_wide<GoN<7>>, &copy, P.num_slots);
    else emu::launch(block_entry<ConnectFour>, &copy, P.num_slots);
    return 0;
}
int launch_match(int game, const EngineParams& P) {
    EngineParams copy = P;
    if (game == SPRL_GAME_OTHELLO) emu::launch(block_entry_match<Othello>, &copy, P.num_slots);
    else if (game == SPRL_GAME_GO7) emu::launch(block_entry_match<Go7>, &copy, P.num_slots);
    else if (game == SPRL_GAME_CONNECT_FOUR) emu::launch(block_entry_match<ConnectFour>, &copy, P.num_slots);
    else if (game == SPRL_GAME_GO9) emu::launch(block_entry_match_wide<GoN<9>>, &copy, P.num_slots);
    else if (game == SPRL_GAME_GO19) emu::launch(block_entry_match_wide<GoN<19>>, &copy, P.num_slots);
    else if (game == SPRL_GAME_GO7W) emu::launch(block_entry_match_wide<GoN<7>>, &copy, P.num_slots);
    else return -1;
    return 0;
}
int launch_compact(const EngineParams& P, int floats_per_leaf) {
    uint32_t run = 0;
    for (int s = 0; s < P.num_slots; ++s) {
        P.leaf_offset[s] = run;
        for (uint32_t q = 0; q < P.leaf_count[s]; ++q)
            memcpy(P.nn_dense + (size_t)(run + q) * floats_per_leaf,
                   P.nn_in + ((size_t)s * P.max_queue + q) * floats_per_leaf, (size_t)floats_per_leaf * sizeof(float));
        run += P.leaf_count[s];
    }
    P.counters->leaf_total = run;
    P.counters->leaf_rows += run;
    P.counters->active_last = P.counters->active_slots;
    P.counters->active_slots = 0;
    return 0;
}
int launch_records_scan(const EngineParams& P) {
    int32_t run = 0;
    for (int g = 0; g < P.num_games; ++g) {
        P.rec_offsets[g] = run;
        run += P.rec_nplies[g];
    }
    P.rec_offsets[P.num_games] = run;
    return 0;
}
template <class G>
static void pack_all(const EngineParams& P, const RecPacked& o, int use_sym) {
    for (int g = 0; g < P.num_games; ++g)
        for (int lane = 0; lane < 64; ++lane) rec_pack_game<G>(P, o, g, lane, use_sym);
}
template <class G>
static void expand_all(const EngineParams& P, const RecExpanded& o) {
    for (int g = 0; g < P.num_games; ++g)
        for (int lane = 0; lane < 64; ++lane) rec_expand_game<G>(P, o, g, lane);
}
int launch_records_pack(int game, const EngineParams& P, const RecPacked& o, int use_sym) {
    if (game == SPRL_GAME_OTHELLO) pack_all<Othello>(P, o, use_sym);
    else if (game == SPRL_GAME_CONNECT_FOUR) pack_all<ConnectFour>(P, o, use_sym);
    else if (game == SPRL_GAME_GO7) pack_all<Go7>(P, o, use_sym);
    else if (game == SPRL_GAME_GO9) pack_all<GoN<9>>(P, o, use_sym);
    else if (game == SPRL_GAME_GO19) pack_all<GoN<19>>(P, o, use_sym);
    else pack_all<GoN<7>>(P, o, use_sym);
    return 0;
}
int launch_records_expand(int game, const EngineParams& P, const RecExpanded& o) {
    if (game == SPRL_GAME_OTHELLO) expand_all<Othello>(P, o);
    else if (game == SPRL_GAME_CONNECT_FOUR) expand_all<ConnectFour>(P, o);
    else if (game == SPRL_GAME_GO7) expand_all<Go7>(P, o);
    else if (game == SPRL_GAME_GO9) expand_all<GoN<9>>(P, o);
    else if (game == SPRL_GAME_GO19) expand_all<GoN<19>>(P, o);
    else expand_all<GoN<7>>(P, o);
    return 0;
}
void* mark() {
    auto* t = new std::chrono::steady_clock::time_point(std::chrono::steady_clock::now());
    return t;
}
double elapsed_ms(void* a, void* b) {
    auto* ta = (std::chrono::steady_clock::time_point*)a;
    auto* tb = (std::chrono::steady_clock::time_point*)b;
    return std::chrono::duration<double, std::milli>(*tb - *ta).count();
}
void mark_free(void* m) { delete (std::chrono::steady_clock::time_point*)m; }
static double g_busy_sum = 0.0;                 // the emulator runs one kernel at a time: busy time = sum of the durations
void* chain_new() { return nullptr; }
void chain_free(void*) {}
double resolve_logged(void*, void* a, void* b) {
    const double ms = elapsed_ms(a, b);
    mark_free(a);
    mark_free(b);
    g_busy_sum += ms;
    return ms;
}
double busy_ms(double* sum_ms) {
    if (sum_ms) *sum_ms = g_busy_sum;
    return g_busy_sum;
}
void busy_reset() { g_busy_sum = 0.0; }
}  // namespace be
