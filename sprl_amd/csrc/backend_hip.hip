// backend_hip.hip — gfx950 implementation of backend.h: device memory, the tree-kernel launch, HIP events.
#include <hip/hip_runtime.h>

#include <string>

#include "backend.h"
#include "step_kernel.h"

namespace {

thread_local std::string g_err;

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
    __shared__ uint32_t lds_path[G::MAX_DEPTH];
    const int slot = (int)blockIdx.x;
    if (slot < P.num_slots) sprl::step_game<G>(P, slot, lds_path);
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
int h2d(void* dst, const void* src, size_t n) { return ok(hipMemcpy(dst, src, n, hipMemcpyHostToDevice), "hipMemcpy h2d") ? 0 : -1; }
int d2h(void* dst, const void* src, size_t n) { return ok(hipMemcpy(dst, src, n, hipMemcpyDeviceToHost), "hipMemcpy d2h") ? 0 : -1; }
int dmemset(void* dst, int v, size_t n) { return ok(hipMemsetAsync(dst, v, n, 0), "hipMemsetAsync") ? 0 : -1; }
int sync() { return ok(hipStreamSynchronize(0), "hipStreamSynchronize") ? 0 : -1; }

int launch_step(int game, const EngineParams& P) {
    dim3 grid((unsigned)P.num_slots), block(64);
    if (game == SPRL_GAME_OTHELLO) hipLaunchKernelGGL(step_kernel<Othello>, grid, block, 0, 0, P);
    else hipLaunchKernelGGL(step_kernel<ConnectFour>, grid, block, 0, 0, P);
    return ok(hipGetLastError(), "step_kernel launch") ? 0 : -1;
}

void* mark() {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    (void)hipEventRecord(e, 0);
    return (void*)e;
}
double elapsed_ms(void* a, void* b) {
    float ms = 0.0f;
    (void)hipEventSynchronize((hipEvent_t)b);
    (void)hipEventElapsedTime(&ms, (hipEvent_t)a, (hipEvent_t)b);
    return (double)ms;
}
void mark_free(void* m) {
    if (m) (void)hipEventDestroy((hipEvent_t)m);
}

const char* last_error() { return g_err.c_str(); }
}  // namespace be
