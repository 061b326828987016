// backend_hip.hip — gfx950 implementation of backend.h: device memory, the tree-kernel launch, HIP events.
#include <hip/hip_runtime.h>

#include <string>

#include "busy_log.h"
#include "backend.h"
#include "records_kernel.h"
#include "step_kernel.h"
#include "step_kernel_wide.h"

namespace {

thread_local std::string g_err;
thread_local hipStream_t g_stream = nullptr;   // the calling thread's engine stream (engine.cpp sets it at every API entry)

bool ok(hipError_t e, const char* what) {
    if (e == hipSuccess) return true;
    g_err = std::string(what) + ": " + hipGetErrorString(e);
    return false;
}

// One workgroup = one wavefront = one game tree.  blockIdx -> slot is the identity so that a game is
// revisited by the same XCD every launch (workgroups are dealt round-robin over the 8 XCDs), which keeps
// the top of its tree in that XCD's L2.
template <class G>
__global__ void __launch_bounds__(64) step_kernel(EngineParams P) {
    __shared__ sprl::WaveLds<G> lds;
    const int slot = (int)blockIdx.x;
    if (slot < P.num_slots) sprl::step_game<G>(P, slot, &lds);
}

template <class G>
__global__ void __launch_bounds__(64) match_kernel(EngineParams P) {
    __shared__ sprl::WaveLds<G> lds;
    const int slot = (int)blockIdx.x;
    if (slot < P.num_slots) sprl::step_match<G>(P, slot, &lds);
}

template <class G>
__global__ void __launch_bounds__(64) match_kernel_wide(EngineParams P) {
    __shared__ sprlw::WaveLdsW<G> lds;
    const int slot = (int)blockIdx.x;
    if (slot < P.num_slots) sprlw::step_match<G>(P, slot, &lds);
}

template <class G>
__global__ void __launch_bounds__(64) step_kernel_wide(EngineParams P) {
    __shared__ sprlw::WaveLdsW<G> lds;
    const int slot = (int)blockIdx.x;
    if (slot < P.num_slots) sprlw::step_game<G>(P, slot, &lds);
}

// Exclusive scan of the per-slot leaf counts (slot order => the dense batch order is deterministic).
// One 1024-thread workgroup: each thread owns a contiguous chunk of slots, chunk totals are scanned in LDS.
__global__ void __launch_bounds__(1024) leaf_scan_kernel(const uint32_t* count, uint32_t* offset, Counters* counters, int n) {
    __shared__ uint32_t part[1024];
    const int t = (int)threadIdx.x;
    const int chunk = (n + 1023) / 1024;
    const int lo = t * chunk, hi = min(n, lo + chunk);
    uint32_t sum = 0;
    for (int i = lo; i < hi; ++i) sum += count[i];
    part[t] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        uint32_t v = t >= off ? part[t - off] : 0u;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint32_t run = part[t] - sum;
    for (int i = lo; i < hi; ++i) {
        offset[i] = run;
        run += count[i];
    }
    if (t == 1023) {
        counters->leaf_total = part[1023];
        counters->leaf_rows += part[1023];
        counters->active_last = counters->active_slots;     // (the step kernel has finished: this launch follows it on its stream)
        counters->active_slots = 0;
    }
}

// One wavefront per slot copies its <= max_queue leaves (16 B per lane per step) to their dense rows.
__global__ void __launch_bounds__(64) leaf_gather_kernel(EngineParams P, int floats_per_leaf) {
    const int slot = (int)blockIdx.x;
    const uint32_t cnt = P.leaf_count[slot];
    if (cnt == 0) return;
    const uint32_t off = P.leaf_offset[slot];
    const int vec = (floats_per_leaf % 4 == 0) ? floats_per_leaf / 4 : 0;   // 16-byte copies only when rows stay aligned
    for (uint32_t q = 0; q < cnt; ++q) {
        const float4* src = (const float4*)(P.nn_in + ((size_t)slot * P.max_queue + q) * floats_per_leaf);
        float4* dst = (float4*)(P.nn_dense + (size_t)(off + q) * floats_per_leaf);
        for (int i = (int)threadIdx.x; i < vec; i += 64) dst[i] = src[i];
        for (int i = vec * 4 + (int)threadIdx.x; i < floats_per_leaf; i += 64)
            P.nn_dense[(size_t)(off + q) * floats_per_leaf + i] = P.nn_in[((size_t)slot * P.max_queue + q) * floats_per_leaf + i];
    }
}

// Exclusive prefix sum of the per-game ply counts (one 1024-thread workgroup, as leaf_scan_kernel): rec_offsets[g],
// rec_offsets[num_games] = total plies of the run.
__global__ void __launch_bounds__(1024) records_scan_kernel(const int32_t* nplies, int32_t* offsets, int n) {
    __shared__ int32_t part[1024];
    const int t = (int)threadIdx.x;
    const int chunk = (n + 1023) / 1024;
    const int lo = t * chunk, hi = min(n, lo + chunk);
    int32_t sum = 0;
    for (int i = lo; i < hi; ++i) sum += nplies[i];
    part[t] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        int32_t v = t >= off ? part[t - off] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int32_t run = part[t] - sum;
    for (int i = lo; i < hi; ++i) {
        offsets[i] = run;
        run += nplies[i];
    }
    if (t == 1023) offsets[n] = part[1023];
}

template <class G>
__global__ void __launch_bounds__(64) records_pack_kernel(EngineParams P, RecPacked out, int use_sym) {
    rec_pack_game<G>(P, out, (int)blockIdx.x, (int)threadIdx.x, use_sym);
}

template <class G>
__global__ void __launch_bounds__(64) records_expand_kernel(EngineParams P, RecExpanded out) {
    rec_expand_game<G>(P, out, (int)blockIdx.x, (int)threadIdx.x);
}

}  // namespace

namespace be {

const char* name() { return "hip-gfx950"; }

bool available(std::string* why) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        if (why) *why = "no HIP device visible";
        return false;
    }
    return true;
}

int init(int device, std::string* err) {
    int n = 0;
    if (!ok(hipGetDeviceCount(&n), "hipGetDeviceCount") || n <= 0) {
        if (err) *err = g_err.empty() ? "no HIP device visible" : g_err;
        return -1;
    }
    if (device < 0 || device >= n) {
        if (err) *err = "device ordinal out of range";
        return -1;
    }
    if (!ok(hipSetDevice(device), "hipSetDevice")) {
        if (err) *err = g_err;
        return -1;
    }
    hipDeviceProp_t prop;
    if (!ok(hipGetDeviceProperties(&prop, device), "hipGetDeviceProperties")) {
        if (err) *err = g_err;
        return -1;
    }
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) {
        if (err) *err = std::string("device is ") + prop.gcnArchName + ", this library contains gfx950 code only";
        return -1;
    }
    return 0;
}

void* dmalloc(size_t bytes) {
    void* p = nullptr;
    if (!ok(hipMalloc(&p, bytes), "hipMalloc")) return nullptr;
    return p;
}
void dfree(void* p) {
    if (p) (void)hipFree(p);
}
int h2d(void* dst, const void* src, size_t n) {
    return ok(hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, g_stream), "hipMemcpy h2d") && ok(hipStreamSynchronize(g_stream), "h2d sync") ? 0 : -1;
}
int d2h(void* dst, const void* src, size_t n) {
    return ok(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, g_stream), "hipMemcpy d2h") && ok(hipStreamSynchronize(g_stream), "d2h sync") ? 0 : -1;
}
int dmemset(void* dst, int v, size_t n) { return ok(hipMemsetAsync(dst, v, n, g_stream), "hipMemsetAsync") ? 0 : -1; }
int sync() { return ok(hipStreamSynchronize(g_stream), "hipStreamSynchronize") ? 0 : -1; }
void* stream_create() {
    hipStream_t s = nullptr;
    if (!ok(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), "hipStreamCreate")) return nullptr;
    return (void*)s;
}
void* stream_create_priority(int high) {
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);      // numerically lower = higher priority
    hipStream_t s = nullptr;
    if (!ok(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, high ? greatest : least), "hipStreamCreateWithPriority")) return nullptr;
    return (void*)s;
}
void* event_new() {
    hipEvent_t e = nullptr;
    return ok(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate") ? (void*)e : nullptr;
}
void event_free(void* ev) {
    if (ev) (void)hipEventDestroy((hipEvent_t)ev);
}
void event_record(void* ev, void* stream) { (void)hipEventRecord((hipEvent_t)ev, (hipStream_t)stream); }
void stream_wait(void* stream, void* ev) { (void)hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)ev, 0); }
void stream_destroy(void* s) {
    if (s) (void)hipStreamDestroy((hipStream_t)s);
}
void set_stream(void* s) { g_stream = (hipStream_t)s; }
void bind(int device, void* s) {
    static thread_local int bound = -1;
    if (bound != device && hipSetDevice(device) == hipSuccess) bound = device;
    g_stream = (hipStream_t)s;
}
void* current_stream() { return (void*)g_stream; }

int launch_step(int game, const EngineParams& P) {
    dim3 grid((unsigned)P.num_slots), block(64);
    if (game == SPRL_GAME_OTHELLO) hipLaunchKernelGGL(step_kernel<Othello>, grid, block, 0, g_stream, P);
    else if (game == SPRL_GAME_GO7) hipLaunchKernelGGL(step_kernel<Go7>, grid, block, 0, g_stream, P);
    else if (game == SPRL_GAME_GO9) hipLaunchKernelGGL(step_kernel_wide<GoN<9>>, grid, block, 0, g_stream, P);
    else if (game == SPRL_GAME_GO19) hipLaunchKernelGGL(step_kernel_wide<GoN<19>>, grid, block, 0, g_stream, P);
    else if (game == SPRL_GAME_GO7W) hipLaunchKernelGGL(step_kernel_wide<GoN<7>>, grid, block, 0, g_stream, P);
    else hipLaunchKernelGGL(step_kernel<ConnectFour>, grid, block, 0, g_stream, P);
    return ok(hipGetLastError(), "step_kernel launch") ? 0 : -1;
}

int launch_match(int game, const EngineParams& P) {
    dim3 grid((unsigned)P.num_slots), block(64);
    if (game == SPRL_GAME_OTHELLO) hipLaunchKernelGGL(match_kernel<Othello>, grid, block, 0, g_stream, P);
    else if (game == SPRL_GAME_GO7) hipLaunchKernelGGL(match_kernel<Go7>, grid, block, 0, g_stream, P);
    else if (game == SPRL_GAME_CONNECT_FOUR) hipLaunchKernelGGL(match_kernel<ConnectFour>, grid, block, 0, g_stream, P);
    else if (game == SPRL_GAME_GO9) hipLaunchKernelGGL(match_kernel_wide<GoN<9>>, grid, block, 0, g_stream, P);
    else if (game == SPRL_GAME_GO19) hipLaunchKernelGGL(match_kernel_wide<GoN<19>>, grid, block, 0, g_stream, P);
    else if (game == SPRL_GAME_GO7W) hipLaunchKernelGGL(match_kernel_wide<GoN<7>>, grid, block, 0, g_stream, P);
    else {
        g_err = "unknown game";
        return -1;
    }
    return ok(hipGetLastError(), "match_kernel launch") ? 0 : -1;
}

int launch_compact(const EngineParams& P, int floats_per_leaf) {
    hipLaunchKernelGGL(leaf_scan_kernel, dim3(1), dim3(1024), 0, g_stream, P.leaf_count, P.leaf_offset, P.counters, P.num_slots);
    hipLaunchKernelGGL(leaf_gather_kernel, dim3((unsigned)P.num_slots), dim3(64), 0, g_stream, P, floats_per_leaf);
    return ok(hipGetLastError(), "leaf compaction launch") ? 0 : -1;
}

int launch_records_scan(const EngineParams& P) {
    hipLaunchKernelGGL(records_scan_kernel, dim3(1), dim3(1024), 0, g_stream, P.rec_nplies, P.rec_offsets, P.num_games);
    return ok(hipGetLastError(), "records scan launch") ? 0 : -1;
}

#define SPRL_FOR_GAME(game, CALL)                                           \
    switch (game) {                                                         \
    case SPRL_GAME_OTHELLO: CALL(Othello); break;                           \
    case SPRL_GAME_CONNECT_FOUR: CALL(ConnectFour); break;                  \
    case SPRL_GAME_GO7: CALL(Go7); break;                                   \
    case SPRL_GAME_GO9: CALL(GoN<9>); break;                                \
    case SPRL_GAME_GO19: CALL(GoN<19>); break;                              \
    case SPRL_GAME_GO7W: CALL(GoN<7>); break;                               \
    default: g_err = "unknown game"; return -1;                             \
    }

int launch_records_pack(int game, const EngineParams& P, const RecPacked& out, int use_sym) {
    const dim3 grid((unsigned)P.num_games), block(64);
#define SPRL_CALL(G) hipLaunchKernelGGL(records_pack_kernel<G>, grid, block, 0, g_stream, P, out, use_sym)
    SPRL_FOR_GAME(game, SPRL_CALL)
#undef SPRL_CALL
    return ok(hipGetLastError(), "records pack launch") ? 0 : -1;
}

int launch_records_expand(int game, const EngineParams& P, const RecExpanded& out) {
    const dim3 grid((unsigned)P.num_games), block(64);
#define SPRL_CALL(G) hipLaunchKernelGGL(records_expand_kernel<G>, grid, block, 0, g_stream, P, out)
    SPRL_FOR_GAME(game, SPRL_CALL)
#undef SPRL_CALL
    return ok(hipGetLastError(), "records expand launch") ? 0 : -1;
}

void* mark() {
    hipEvent_t e = busy::get_event();                  // pooled per process and device
    if (!e) return nullptr;
    (void)hipEventRecord(e, g_stream);
    return (void*)e;
}
double elapsed_ms(void* a, void* b) {
    float ms = 0.0f;
    (void)hipEventSynchronize((hipEvent_t)b);
    (void)hipEventElapsedTime(&ms, (hipEvent_t)a, (hipEvent_t)b);
    return (double)ms;
}
void mark_free(void* m) { busy::put_event((hipEvent_t)m); }

static busy::Log g_tree_busy;
void* chain_new() { return new busy::Chain(); }
void chain_free(void* chain) {
    if (!chain) return;
    static_cast<busy::Chain*>(chain)->release();
    delete static_cast<busy::Chain*>(chain);
}
double resolve_logged(void* chain, void* a, void* b) {
    double ms = 0.0;
    (void)hipEventSynchronize((hipEvent_t)b);
    const auto iv = static_cast<busy::Chain*>(chain)->resolve((hipEvent_t)a, (hipEvent_t)b, &ms);
    busy::put_event((hipEvent_t)b);
    g_tree_busy.add(iv);
    return ms;
}
double busy_ms(double* sum_ms) { return g_tree_busy.union_ms(sum_ms); }
void busy_reset() { g_tree_busy.reset(); }

const char* last_error() { return g_err.c_str(); }
}  // namespace be
