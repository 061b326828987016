// wave.h — the few cross-lane primitives the tree kernels are written against.
//
// One game tree is walked by ONE 64-lane wavefront (a 64-thread workgroup), lanes <-> board cells /
// actions.  Device build (hipcc, gfx950): thin wrappers over the CDNA4 cross-lane instructions
// (v_readlane / ds_bpermute / s_ballot).  With -DSPRL_EMU (g++, tests/emu only) the same kernel source
// runs on a CPU SIMT emulator — 64 cooperative fibers per "wave" that rendezvous at every collective and
// assert convergence — so kernel logic can be sanitised and parity-tested without a GPU.  The emulator
// is test infrastructure: the product library never contains or falls back to it.
#ifndef SPRL_WAVE_H
#define SPRL_WAVE_H

#include <stdint.h>

#ifdef SPRL_EMU
// ------------------------------------------------------------------------------------------------
// CPU SIMT emulator interface (implemented in tests/emu/emu_runtime.cpp)
// ------------------------------------------------------------------------------------------------
#define SPRL_DEV static inline
#define SPRL_DEV_NOINLINE static
namespace emu {
int lane();
int block();
// every lane contributes `v`; returns a pointer to the 64 contributed values (valid until the lane's
// next collective).  `site` identifies the call site for the convergence assert.
const uint64_t* exchange(uint64_t v, int site);
void fatal(const char* what, const uint64_t* slots);
}  // namespace emu

namespace wv {
static inline int lane() { return emu::lane(); }
static inline uint64_t ballot(bool p) {
    const uint64_t* s = emu::exchange(p ? 1 : 0, __LINE__);
    uint64_t m = 0;
    for (int i = 0; i < 64; ++i) m |= (s[i] & 1ull) << i;
    return m;
}
static inline uint32_t shfl_u32(uint32_t v, int src) { return (uint32_t)emu::exchange(v, __LINE__)[src & 63]; }
static inline uint32_t bcast_u32(uint32_t v, int src) {
    const uint64_t* s = emu::exchange(((uint64_t)(uint32_t)src << 32) | v, __LINE__);
    for (int i = 1; i < 64; ++i)
        if ((s[i] >> 32) != (s[0] >> 32)) emu::fatal("bcast source lane is not wave-uniform", s);
    return (uint32_t)s[src & 63];
}
static inline float fmax_all(float v) {
    uint32_t u;
    __builtin_memcpy(&u, &v, 4);
    const uint64_t* s = emu::exchange(u, __LINE__);
    float m = -__builtin_inff();
    for (int i = 0; i < 64; ++i) {
        uint32_t w = (uint32_t)s[i];
        float f;
        __builtin_memcpy(&f, &w, 4);
        if (f > m) m = f;
    }
    return m;
}
template <typename T> static inline T uni(T v) {
    uint64_t u = 0;
    __builtin_memcpy(&u, &v, sizeof(T));
    const uint64_t* s = emu::exchange(u, __LINE__);
    for (int i = 1; i < 64; ++i)
        if (s[i] != s[0]) emu::fatal("value claimed uniform is not", s);
    return v;
}
static inline uint32_t atomic_add_u32(uint32_t* p, uint32_t v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline uint32_t atomic_xor_u32(uint32_t* p, uint32_t v) { return __atomic_fetch_xor(p, v, __ATOMIC_RELAXED); }
static inline uint32_t atomic_or_u32(uint32_t* p, uint32_t v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
static inline unsigned long long atomic_add_u64(unsigned long long* p, unsigned long long v) {
    return __atomic_fetch_add(p, v, __ATOMIC_RELAXED);
}
static inline uint32_t atomic_cas_u32(uint32_t* p, uint32_t expect, uint32_t desired) {
    __atomic_compare_exchange_n(p, &expect, desired, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED);
    return expect;
}
static inline void atomic_store_u32(uint32_t* p, uint32_t v) { __atomic_store_n(p, v, __ATOMIC_RELAXED); }
// lockstep point: lanes run sequentially between collectives in the emulator, so code that loads a
// wave-uniform location and later stores to it needs one of these between the load and the store
static inline void sync() { (void)emu::exchange(0, __LINE__); }
static inline void wave_fence() { sync(); }
static inline void agent_release() {}
static inline void agent_acquire() {}
}  // namespace wv

#else
// ------------------------------------------------------------------------------------------------
// gfx950 device build
// ------------------------------------------------------------------------------------------------
#include <hip/hip_runtime.h>
#define SPRL_DEV __device__ __forceinline__
// Out-of-line device calls were measured and rejected: passing the wave state by reference puts it in scratch
// (360 B/lane) and the launch got 1.8x slower, so the rare heavy paths stay inlined too.
#define SPRL_DEV_NOINLINE __device__ __forceinline__

namespace wv {
SPRL_DEV int lane() { return (int)(threadIdx.x & 63u); }
SPRL_DEV uint64_t ballot(bool p) { return __ballot(p); }
SPRL_DEV uint32_t shfl_u32(uint32_t v, int src) {
    return (uint32_t)__builtin_amdgcn_ds_bpermute((src & 63) << 2, (int)v);
}
SPRL_DEV uint32_t bcast_u32(uint32_t v, int src) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, __builtin_amdgcn_readfirstlane(src));
}
// Wave-wide max on the VALU's data-parallel-primitive paths (quad_perm, row_mirror, row_bcast): six VALU
// instructions of a few cycles each, instead of a butterfly of six dependent ds_bpermute round trips through LDS
// (~100+ cycles each), which sat on the critical path of every level of a descent.
SPRL_DEV float fmax_all(float v) {
#define SPRL_DPP_MAX(ctrl, row_mask)                                                                          \
    {                                                                                                         \
        const int o = __builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), ctrl, row_mask, 0xf, false); \
        const float f = __int_as_float(o);                                                                    \
        v = f > v ? f : v;                                                                                    \
    }
    SPRL_DPP_MAX(0xB1, 0xf)    // quad_perm [1,0,3,2]
    SPRL_DPP_MAX(0x4E, 0xf)    // quad_perm [2,3,0,1]
    SPRL_DPP_MAX(0x141, 0xf)   // row_half_mirror
    SPRL_DPP_MAX(0x140, 0xf)   // row_mirror: every lane of a 16-lane row now holds the row max
    SPRL_DPP_MAX(0x142, 0xa)   // row_bcast:15 into rows 1 and 3
    SPRL_DPP_MAX(0x143, 0xc)   // row_bcast:31 into rows 2 and 3: lane 63 holds the wave max
#undef SPRL_DPP_MAX
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
SPRL_DEV uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
SPRL_DEV int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
SPRL_DEV float uni(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
SPRL_DEV uint64_t uni(uint64_t v) {
    uint32_t lo = uni((uint32_t)v), hi = uni((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}
SPRL_DEV uint32_t atomic_add_u32(uint32_t* p, uint32_t v) { return atomicAdd(p, v); }
SPRL_DEV uint32_t atomic_xor_u32(uint32_t* p, uint32_t v) { return atomicXor(p, v); }
SPRL_DEV uint32_t atomic_or_u32(uint32_t* p, uint32_t v) { return atomicOr(p, v); }
SPRL_DEV unsigned long long atomic_add_u64(unsigned long long* p, unsigned long long v) { return atomicAdd(p, v); }
SPRL_DEV uint32_t atomic_cas_u32(uint32_t* p, uint32_t expect, uint32_t desired) { return atomicCAS(p, expect, desired); }
SPRL_DEV void atomic_store_u32(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// Order this wave's earlier global stores before its later loads from OTHER lanes of the same wave.  A wavefront's
// vector memory instructions are issued and performed in order through one L1, so wavefront scope needs no
// instruction on gfx9 (LLVM AMDGPU memory model: no code is emitted for wavefront-scope fences); this only stops
// the compiler from reordering across it.
SPRL_DEV void wave_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }
// Ownership of a memory region moving between wavefronts that may sit on different XCDs: the per-XCD L2s are
// write-back and not coherent with each other, so the old owner writes its dirty lines back (buffer_wbl2) before
// publishing the hand-off and the new owner invalidates its L1 after winning it (MI355X_MICROARCH.md,
// "Workgroup dispatch, XCD placement & inter-workgroup visibility").
SPRL_DEV void agent_release() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
SPRL_DEV void agent_acquire() {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
// lanes of a wavefront execute in lockstep: nothing to do on hardware (see the emulator's sync())
SPRL_DEV void sync() { __builtin_amdgcn_wave_barrier(); }
}  // namespace wv
#endif

#if defined(SPRL_PHASE_TIMERS) && !defined(SPRL_EMU)
#define SPRL_TIC(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define SPRL_TOC(acc, var) (acc) += __builtin_amdgcn_s_memtime() - (var)
#else
#define SPRL_TIC(var) do { } while (0)
#define SPRL_TOC(acc, var) do { } while (0)
#endif

namespace wv {
SPRL_DEV float shfl_f32(float v, int src) {
    uint32_t u;
    __builtin_memcpy(&u, &v, 4);
    u = shfl_u32(u, src);
    __builtin_memcpy(&v, &u, 4);
    return v;
}
SPRL_DEV float bcast_f32(float v, int src) {
    uint32_t u;
    __builtin_memcpy(&u, &v, 4);
    u = bcast_u32(u, src);
    __builtin_memcpy(&v, &u, 4);
    return v;
}
SPRL_DEV int popc64(uint64_t m) { return __builtin_popcountll(m); }
SPRL_DEV int ctz64(uint64_t m) { return __builtin_ctzll(m); }
SPRL_DEV uint64_t lt_mask(int l) { return (1ull << l) - 1ull; }
// index of the n-th (0-based) set bit of a wave-uniform mask; n < popc(m)
SPRL_DEV int nth_set_bit(uint64_t m, int n) {
    for (int i = 0; i < n; ++i) m &= m - 1;
    return __builtin_ctzll(m);
}
}  // namespace wv

#endif  // SPRL_WAVE_H
