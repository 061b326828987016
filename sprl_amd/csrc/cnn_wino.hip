// cnn_wino.hip — the trunk convolution of the policy/value CNN as a hand-written gfx950 kernel.
//
// conv3x3 (64 -> 64 channels, padding 1, boards up to 8x8: grid_networks.py:8-27) + BatchNorm/bias (folded scale and
// shift) + optional residual + ReLU in ONE launch, fp32 end to end:
//   Winograd F(4x4, 3x3): a board is 2x2 output tiles of 4x4; per tile and channel the 6x6 input patch d is
//   transformed (V = B^T d B), the 36 transform positions are 36 independent [64 x 64] x [64 x tiles] products on the
//   fp32 matrix cores (v_mfma_f32_16x16x4_f32), and Y = A^T M A is transformed back.  4x fewer multiplies than the
//   direct convolution (2.25x fewer than the F(2x2,3x3) library kernel it replaces).
//
// Activation layout "W", chosen so that both ends of this kernel move whole 256-byte rows:
//   x[n][g][i][cs][tile][j],  channel k = 16 (g >> 2) + 4 cs + (g & 3),  cell (row, col) = (4 ty + i, 4 tx + j),
//   tile = 2 ty + tx; 4096 floats per board, cells outside an H x W board hold zeros.
// A channel quad of the K loop (one MFMA K step) is one group g: 1 KB contiguous per board.
//
// Workgroup = 4 waves = 4 boards = 16 tiles, two workgroups per CU; wave kb owns output channels 16kb..16kb+15 for ALL 36
// transform positions of the 16 tiles (36 accumulator tiles, 144 registers), so a lane ends the K loop holding every
// position of its four (channel, tile) pairs and the inverse transform needs no exchange.  K loop over the 16 groups, two
// groups (8 channels) per phase:
//   * activations: a chunk of 8 channels per board group, zero-bordered 10x10 images in LDS (double buffered; strides
//     chosen so that the 32 lanes of a bank group read 32 different banks);
//   * B operand: for every chunk the threads build V[2 groups][36][4 ch][16 tiles] ONCE into LDS (thread = one channel, one
//     tile, three of the six transform rows: two factored 1-D transforms of its 6x6 patch), double buffered, in the lane
//     order the MFMA wants, so a B fetch is one conflict-free ds_read;
//   * A operand: weights pre-transformed on the host (U = G g G^T, torch_eval.cpp: wino_transform) and stored in lane
//     order, four transform positions per 16-byte load (590 KB per layer, L2-resident), through a 36-register ring: a quad
//     is reloaded for the next K step right behind the four MFMAs that used it;
//   * output: inverse transform in registers, scale/shift, residual, ReLU, 16-byte stores that are contiguous over 16 lanes.
// All global memory goes through buffer descriptors (wave-uniform byte offset + one 32-bit per-lane offset: no 64-bit address
// arithmetic in vector registers; boards past the batch read as zero and are never stored - the hardware range check replaces
// every `n < batch` branch).  Memory waits are ordered for the in-order return of a wave's vector loads: the activation chunk
// c+3 is requested at the start of the phase's second K step and goes to LDS between the two K steps of the NEXT phase; the
// filter quads re-issued behind it are first needed a whole K step later.
// Measured on MI355X at 14 400 boards (tools/wino_lab.hip): 232 us per launch = 73 TFLOP/s of Winograd-
// domain fp32 MFMA work (0.46 of the 157.3 TFLOP/s matrix peak; the round-1 kernel: 287 us, 0.38).  Compile this file with
// -fno-slp-vectorize (Makefile): the SLP vectoriser's v_pk_* instructions and the v_mov shuffles feeding them cost 3-4 %;
// beside fp32 MFMAs a SIMD retires only about three other vector instructions per MFMA (tools/mfma_valu_probe.hip).
// The experiments behind these choices (filters streamed as 3x3 taps or as half-transformed T = G g with the transform in
// registers, 16-byte V reads, persistent workgroups, a second activation register set) live in tools/wino_variants.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

constexpr int RS = 10;                   // row stride of a zero-bordered board image
constexpr int IA = 112, IB = 226;        // board b sits at (b & 1) * IA + (b >> 1) * IB   (== 16 and 2 mod 32)

__device__ __forceinline__ int board_off(int b) { return (b & 1) * IA + (b >> 1) * IB; }

// Y = A^T m A for one (channel, tile): rows of the 4x4 output
__device__ __forceinline__ void inverse_transform(const float (&m)[6][6], float (&o)[4][4]) {
    float tm[4][6];
#pragma unroll
    for (int b = 0; b < 6; ++b) {
        const float s12 = m[1][b] + m[2][b], d12 = m[1][b] - m[2][b], s34 = m[3][b] + m[4][b], d34 = m[3][b] - m[4][b];
        tm[0][b] = m[0][b] + s12 + s34;
        tm[1][b] = d12 + 2.0f * d34;
        tm[2][b] = s12 + 4.0f * s34;
        tm[3][b] = d12 + 8.0f * d34 + m[5][b];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float s12 = tm[i][1] + tm[i][2], d12 = tm[i][1] - tm[i][2], s34 = tm[i][3] + tm[i][4], d34 = tm[i][3] - tm[i][4];
        o[i][0] = tm[i][0] + s12 + s34;
        o[i][1] = d12 + 2.0f * d34;
        o[i][2] = s12 + 4.0f * s34;
        o[i][3] = d12 + 8.0f * d34 + tm[i][5];
    }
}

// The same for TWO output components at once.  Stage 1 (columns) runs on register pairs - acc[p][0:1] and acc[p][2:3] are aligned
// pairs of the 16x16x4 accumulator tile, so v_pk_add_f32 / v_pk_fma_f32 do two components per instruction; stage 2 (rows) reads
// the halves of those pairs (sub-registers are free) and writes each component's 4x4 output into its own registers, ready for the
// 16-byte stores - no shuffles.  60 + 2 x 40 = 140 instructions for two components instead of 200.
__device__ __forceinline__ void inverse_transform_pair(const f2 (&m)[6][6], float (&o0)[4][4], float (&o1)[4][4]) {
    f2 tm[4][6];
#pragma unroll
    for (int b = 0; b < 6; ++b) {
        const f2 s12 = m[1][b] + m[2][b], d12 = m[1][b] - m[2][b], s34 = m[3][b] + m[4][b], d34 = m[3][b] - m[4][b];
        tm[0][b] = m[0][b] + s12 + s34;
        tm[1][b] = d12 + 2.0f * d34;
        tm[2][b] = s12 + 4.0f * s34;
        tm[3][b] = d12 + 8.0f * d34 + m[5][b];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        {
            const float s12 = tm[i][1][0] + tm[i][2][0], d12 = tm[i][1][0] - tm[i][2][0], s34 = tm[i][3][0] + tm[i][4][0], d34 = tm[i][3][0] - tm[i][4][0];
            o0[i][0] = tm[i][0][0] + s12 + s34;
            o0[i][1] = d12 + 2.0f * d34;
            o0[i][2] = s12 + 4.0f * s34;
            o0[i][3] = d12 + 8.0f * d34 + tm[i][5][0];
        }
        {
            const float s12 = tm[i][1][1] + tm[i][2][1], d12 = tm[i][1][1] - tm[i][2][1], s34 = tm[i][3][1] + tm[i][4][1], d34 = tm[i][3][1] - tm[i][4][1];
            o1[i][0] = tm[i][0][1] + s12 + s34;
            o1[i][1] = d12 + 2.0f * d34;
            o1[i][2] = s12 + 4.0f * s34;
            o1[i][3] = d12 + 8.0f * d34 + tm[i][5][1];
        }
    }
}

// head convolutions fused behind the last trunk convolution (HEADS >= 1: 2 policy + 1 value head channels)
struct HeadArgs {
    const float *hw, *hb;   // [3][64] weights (policy rows first), [3] biases
    float* maps_out;        // HEADS = 1: ReLU'd head maps [batch][3 * H * W]; the FC layers run in sprl_tail_fc (cnn_epilogue.hip)
    // HEADS = 2: the FC layers too (grid_networks.py:45,50-51) - the forward ends in this kernel, the maps never leave LDS
    const float *pfc_w, *pfc_b;      // [2 * H * W][A] (transposed Linear weight), [A]
    const float *vfc1_w, *vfc1_b;    // [H * W][HID], [HID]
    const float *vfc2_w, *vfc2_b;    // [HID], [1]
    float *logits, *value;           // [batch][A], [batch]
    int A, HID;
    unsigned magic_a, magic_h;       // ceil(2^20 / A), ceil(2^20 / HID): t / A == (t * magic_a) >> 20 for t < 512
};
// The stem (3 input planes -> 64 channels, 3x3, folded BN, ReLU; grid_networks.py:36-38,56) fused into the FIRST trunk convolution
// (STEM = 1, round 4): the workgroup computes the stem output of its four boards in its prologue - one wave per board, out[64 ch]
// [64 cells] = W[64][27] x patches[27][64] on v_mfma_f32_16x16x4_f32 exactly as cnn_epilogue.hip: stem_mfma_kernel does - stores it
// to x0 (layout W; the next convolution's residual input) and then runs the convolution on it, reading it back through L2 (the
// lines were written by this CU a moment ago: `s_waitcnt vmcnt(0)` + barrier, same XCD).  The accumulators are not live yet, so
// the 28 weight registers and 32 scale / shift registers cost nothing; the plane images sit in the V buffer, which is free until
// the first V is built.  One launch and one pass of 16 KB per board through HBM less per forward (the stem kernel wrote x0,
// this convolution read it back from HBM).
struct StemArgs {
    const float* planes;    // [batch][3][H][W] input planes
    const float* w;         // [64][27] stem convolution weights
    const float *scale, *shift;   // [64] folded BatchNorm of the stem
    float* x0;              // = the convolution's input x: written here, [batch][4096] in layout W
};

// HEADS = 2 task split: the policy contraction (<= 128 long) in four parts, the hidden layer's (<= 64) in two; a task is one output
// column x one part x the four boards of the workgroup, at most 32 weights long
constexpr int FC_MAXA = 96, FC_LP = 32;

// ---- compile-time switches of the kernels below (product values; tools/conv_ab.py and tools/build_plugin_variant.sh build the others) ----
#ifndef SPRL_WINO_LD_AUX
#define SPRL_WINO_LD_AUX 2                            // cache policy of the 8x8 kernel's activation / residual loads (2 = nt: streamed)
#endif
#ifndef SPRL_WINO_ST_AUX
#define SPRL_WINO_ST_AUX 2                            // ... and of its output stores (default policy for either: -1.3 / -2.9 % in the bench, profiles/r04zo_*)
#endif
#ifndef SPRL_WINO_F4_INTERLEAVE
#define SPRL_WINO_F4_INTERLEAVE 0                     // lab: F(4x4) any-board kernel - the transform in pieces behind the filter quads of K step 2c+1 (first quad n - 1)
#endif
#ifndef SPRL_WINO_F4_REQ_HEAD
#define SPRL_WINO_F4_REQ_HEAD 0                       // F(4x4) any-board kernel: the activation request at the head of the phase (0: behind the transform)
#endif
// 8x8 kernel: n > 0 = the input transform of chunk c+1 is issued in six pieces behind filter quads n-1 .. n+4 of K step 2c+1 (under the
// wave's own MFMAs) instead of as one block between the K steps.  Together with the rolling B reads (SPRL_WINO_BROLL = 4; without
// them every piece cuts the K step's LDS reads off from their MFMAs: -1 ... -2 %): lab +0.2 ... +2.8 %, bench 455.3 / 456.0 against
// 451.8 / 451.4 games/s in one call (profiles/r04zzm_*, r04zzn_*).  0 = the round-3 form.
#ifndef SPRL_WINO_INTERLEAVE
#define SPRL_WINO_INTERLEAVE 1
#endif
#ifndef SPRL_WINO_INTERLEAVE_PLAIN
#define SPRL_WINO_INTERLEAVE_PLAIN SPRL_WINO_INTERLEAVE      // lab: another first quad for the variants without the residual rows / with the heads
#endif
#ifndef SPRL_WINO_WGROUP
#define SPRL_WINO_WGROUP 0                            // lab: 8x8 kernel on a group-major activation layout (see xvoff)
#endif
#ifndef SPRL_WINO_REQ_POS
#define SPRL_WINO_REQ_POS 0                           // 8x8 kernel: the activation request at the head (0) / middle (1) / end (2) of K step 2c+1
#endif
#ifndef SPRL_WINO_GLOAD_BRANCH
#define SPRL_WINO_GLOAD_BRANCH 0                      // lab: 1 = the activation request of a phase sits behind a branch (the round-3 form)
#endif
#ifndef SPRL_WINO_PRIO
#define SPRL_WINO_PRIO 0                              // lab: 8x8 kernel - static priority for one of the two waves of a SIMD (1: odd wave slot, 2: even, 3: odd workgroup)
#endif
#ifndef SPRL_WINO_PRIO_LEVEL
#define SPRL_WINO_PRIO_LEVEL 1
#endif
// B operands of a K step (V from LDS) read a fixed number of MFMA pairs AHEAD of their use (rolling, asm reads with counted waits:
// see the K step of the 8x8 kernel).  Measured per kernel (tools/conv_ab.py, profiles/r04zl_conv_ab_broll.log, r04zzb_*): the F(4x4)
// any-board kernel gains 1.3 - 3.5 % (19x19), the 8x8 kernel nothing by itself (+-1 %: its SIMD's other wave already covers the LDS
// round trips) and the F(3x3) kernel has no registers for it (61 spilled).  0 = the compiler's own batches of eight.
#ifndef SPRL_WINO_BROLL
#define SPRL_WINO_BROLL 4                             // 8x8 kernel (alone +-1 %; it is what makes the interleaved transform pay, see SPRL_WINO_INTERLEAVE)
#endif
#ifndef SPRL_WINO_BROLL_F4
#define SPRL_WINO_BROLL_F4 4                          // any-board kernel, F(4x4,3x3)
#endif
#ifndef SPRL_WINO_BROLL_F3
#define SPRL_WINO_BROLL_F3 0                          // any-board kernel, F(3x3,3x3)
#endif
#ifndef SPRL_WINO_DEEP4
#define SPRL_WINO_DEEP4 0                             // 1: the F(4x4) layout-T kernel also keeps two activation chunks in flight
#endif
#ifdef SPRL_WINO_LAB
__constant__ int wino_lab_dbg;
#define LAB_OFF(bit) (wino_lab_dbg & (1 << (bit)))
#else
#define LAB_OFF(bit) 0
#endif
#if defined(SPRL_WINO_LAB) || defined(SPRL_WINO_TRACE)
// timeline (tools/wino8_trace.py, a build with -DSPRL_WINO_TRACE: the product kernel plus the stamps): lane 0 of every wave of the
// workgroups [first, first + count) writes the shader clock at the marked points of the 8x8 kernel to
// trace[(block - first) * 4 + wave][0..127]; stamp 63 is the wave's HW_ID (CU / SIMD / wave slot), 62 its XCC_ID; 64 + 2c, 65 + 2c: inside
// K step 2c+1 (behind the activation request, behind its fifth filter quad).
// (`tracing` / `trace_row` are set once at the top of the kernel; the stamps of a phase wait in scalar registers and are written
// together at its end, so that a traced wave pays one wait for the clock reads per phase)
#define SPRL_WINO_STAMPS 1
__device__ unsigned long long* wino_lab_trace;
__device__ int wino_lab_trace_first, wino_lab_trace_count;
#define LAB_STAMP(id)                                                                                                              \
    do {                                                                                                                           \
        if (tracing) {                                                                                                             \
            const unsigned long long now_ = clock64();                                                                             \
            if (lane == 0) trace_row[id] = now_;                                                                                   \
        }                                                                                                                          \
    } while (0)
#define LAB_STAMP_LATER(var) do { if (tracing) var = clock64(); } while (0)
#define LAB_STAMP_WRITE(id, var) do { if (tracing && lane == 0) trace_row[id] = var; } while (0)
#else
#define LAB_STAMP(id) ((void)0)
#define LAB_STAMP_LATER(var) ((void)0)
#define LAB_STAMP_WRITE(id, var) ((void)0)
#endif

constexpr int NIMG2 = 4, NTHR2 = 256;
constexpr int CS2 = 449;                 // channel-slot stride for 4 boards (== 1 mod 32)
constexpr int IN_BUF2 = 8 * CS2;
constexpr int V_G2 = 36 * 64;            // V of one group: [p][c_sub][16 tiles]
constexpr int LDS_FLOATS2 = 2 * IN_BUF2 + 4 * V_G2;      // 65.6 KB

template <int H, int W, int HEADS = 0, int RES = 1, int STEM = 0>
__global__ void __launch_bounds__(NTHR2, 2) wino_conv64_kernel(const float* x, const float* __restrict__ u,
                                                               const float* __restrict__ scale, const float* __restrict__ shift,
                                                               const float* __restrict__ res, float* __restrict__ y, int batch,
                                                               int relu, const unsigned* __restrict__ batch_dev, HeadArgs ha,
                                                               StemArgs sa) {
    // batch_dev != null: the number of boards is on the device (the engine's leaf count of this round), `batch` is the
    // capacity the grid was sized for; workgroups past the real count leave at once
    if (batch_dev) {
        const int real = (int)*batch_dev;
        batch = real < batch ? real : batch;
    }
    if ((int)blockIdx.x * NIMG2 >= batch) return;
#if SPRL_WINO_PRIO
    {   // lab (profiles/r04zr_conv_ab_prio.log): a static priority for one of the two waves that share a SIMD (told apart by their
        // wave slot) - the arbiter already prefers the older wave, so this changes little: +0.6 ... +5.5 % in the lab, nothing in the bench
        const unsigned hw_id = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (3 << 11));      // HW_ID[3:0]: wave slot on the SIMD
        const bool hi = SPRL_WINO_PRIO == 1 ? (hw_id & 1) : SPRL_WINO_PRIO == 2 ? !(hw_id & 1) : (blockIdx.x & 1);
        if (hi) __builtin_amdgcn_s_setprio(SPRL_WINO_PRIO_LEVEL);
    }
#endif
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS2];
    float* const in_buf = lds;                        // [2][IN_BUF2]
    float* const v_buf = lds + 2 * IN_BUF2;           // [2 phases][2 groups][V_G2]
    const int tid = (int)threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c_sub = lane >> 4, tl = lane & 15;
    const int gl = wave & 1, wa = wave >> 1;          // producer role: group of the chunk, transform rows 3wa..3wa+2
    const int kb = wave;                              // consumer role: output channels 16kb..16kb+15
    const int n0 = (int)blockIdx.x * NIMG2;

#ifdef SPRL_WINO_STAMPS
    const bool tracing = wino_lab_trace && (int)blockIdx.x >= wino_lab_trace_first && (int)blockIdx.x < wino_lab_trace_first + wino_lab_trace_count;
    unsigned long long* const trace_row = wino_lab_trace + (((int)blockIdx.x - wino_lab_trace_first) * 4 + wave) * 128;
    unsigned long long st0_ = 0, st1_ = 0, st2_ = 0, st3_ = 0, st4_ = 0, st5_ = 0, st6_ = 0, st7_ = 0;
    if (tracing && lane == 0) {
        trace_row[63] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));       // HW_ID
        trace_row[62] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));      // XCC_ID
    }
#endif
    LAB_STAMP(0);                                      // wave started
    // borders stay zero for the whole kernel.  (The fill BEHIND the first requests of the prologue - on their way to HBM while the 28
    // LDS stores per thread are issued - was measured: 1.1 ... 1.6 % slower, profiles/r04zw_conv_ab_fill_behind_requests.log.)
    for (int i = tid; i < 2 * IN_BUF2; i += NTHR2) lds[i] = 0.0f;

    int ldst[2], xvoff[2];                            // LDS float index / global byte offset of this thread's two 16-byte pieces of a chunk
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int f = tid + NTHR2 * it;
        const int b = f >> 7, rem = f & 127;
        const int g2 = rem >> 6, i = (rem >> 4) & 3, cs = (rem >> 2) & 3, tile = rem & 3;
        ldst[it] = (g2 * 4 + cs) * CS2 + board_off(b) + (4 * (tile >> 1) + i + 1) * RS + 4 * (tile & 1) + 1;
#if SPRL_WINO_WGROUP
        // lab: layout W', group-major over the four boards of a workgroup - x[n / 4][g][n % 4][256] - so that a chunk of the workgroup
        // (2 groups x 4 boards) is ONE contiguous 8 KB piece instead of four 2 KB pieces 16 KB apart (batch a multiple of 4)
        xvoff[it] = (n0 * 4096 + g2 * 1024 + b * 256 + (rem & 63) * 4) * 4;
#else
        xvoff[it] = ((n0 + b) * 4096 + rem * 4) * 4;  // the per-lane offset carries the board: the range check drops boards >= batch
#endif
    }
    const int patch0 = (gl * 4 + c_sub) * CS2 + board_off(tl >> 2) + ((tl >> 1) & 1) * 4 * RS + (tl & 1) * 4 + wa * RS;
    const int vdst0 = gl * V_G2 + (3 * wa) * 6 * 64 + lane;
    const unsigned act_bytes = (unsigned)batch * 16384u;            // the host keeps batch * 16 KB below 4 GB
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, act_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)res, 0, res ? act_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)y, 0, (y && !HEADS) ? act_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc((void*)u, 0, 36u * 4096u * 4u, 0x00020000);
    const int ulane = lane * 16;
    const int tile = tl & 3, ty = tile >> 1, tx = tile & 1;          // this lane's tile
    // Output rows: per-lane offset + an IMMEDIATE row offset, scalar offset 0.  With a scalar-REGISTER offset the compiler assumes
    // a 16-byte buffer store needs no wait state before its data registers are overwritten (LLVM createsVALUHazard: "hazard only
    // exists if the instruction is not using a register in the soffset field") and schedules a VALU write into them right behind
    // the store; on gfx950 that corrupted dword 1 of lanes 12-15 of every 16 (found with tools/wino_lab.hip).
#if SPRL_WINO_WGROUP
    const int ovoff = (n0 * 4096 + kb * 4096 + (tl >> 2) * 256 + c_sub * 16 + tile * 4) * 4;      // + r * 4096 + i * 256 bytes
#else
    const int ovoff = ((n0 + (tl >> 2)) * 4096 + kb * 1024 + c_sub * 16 + tile * 4) * 4;
#endif

    f4 acc[36];                                       // first written by the first K step (C operand = 0): no zero fill
    f4 pre[2];                                        // the activation chunk in flight
    auto gload_to = [&](int chunk, f4 (&dst)[2]) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            if (LAB_OFF(0) && chunk > 2) continue;     // lab: no activation loads behind the prologue's
            // (lab bit 8: every workgroup reads the first four boards - the same stream of requests, answered by L2 instead of HBM)
            int voff = LAB_OFF(8) ? xvoff[it] - n0 * 16384 : xvoff[it];
            // (lab bit 9: the boards of a 64 MB window - too large for the L2s, small enough for the memory-side cache)
            if (LAB_OFF(9)) voff = xvoff[it] - (n0 - n0 % 4096) * 16384;
            // chunk < 0 = "nothing to request": the load is issued all the same, with an offset the range check rejects (no memory
            // access, zeros come back at once).  A BRANCH around the loads costs more than the loads: the compiler's wait counts
            // must hold on both paths, so behind a skipped-or-not pair of loads every `vmcnt(n)` of the K step is two too small on
            // the path that did issue them - and the last filter quads of the step then wait for the HBM trip of the activation
            // chunk itself (tools/conv_ab.py, profiles/r04zt_conv_ab_request_without_branch.log: +0.2 ... +2.7 %).
            if (chunk < 0) voff = (int)0x80000000;
            dst[it] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rx, voff, (chunk < 0 ? 0 : chunk) * (SPRL_WINO_WGROUP ? 8192 : 2048), SPRL_WINO_LD_AUX));
        }
    };
    auto lstore_from = [&](float* buf, const f4 (&src)[2]) {
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int j = 0; j < 4; ++j) buf[ldst[it] + j] = src[it][j];
    };
    // B^T rows written as fused multiply-adds (two instructions for 4 a - 5 b + c, one for c - 4 b): the kernel is bound by the
    // number of VALU instructions a SIMD retires beside its MFMAs (round 3), so every instruction counts
    auto produce = [&](int c) {
        const float* pp = in_buf + (c & 1) * IN_BUF2 + patch0;
        float* vd = v_buf + (c & 1) * 2 * V_G2 + vdst0;
        // stage 1 (over the patch rows) on PAIRS of columns - packed v_pk_* instructions, two columns each; stage 2 (over the
        // columns) reads the halves of those pairs
        f2 wr[3][3];
        if (LAB_OFF(2)) return;                       // lab: no input transform
        if (wa == 0) {                                // (the wave-uniform branch outside the loop: one scheduling region per role)
#pragma unroll
            for (int jp = 0; jp < 3; ++jp) {
                const int j = 2 * jp;
                const f2 e0 = { pp[j], pp[j + 1] }, e1 = { pp[RS + j], pp[RS + j + 1] }, e2 = { pp[2 * RS + j], pp[2 * RS + j + 1] },
                         e3 = { pp[3 * RS + j], pp[3 * RS + j + 1] }, e4 = { pp[4 * RS + j], pp[4 * RS + j + 1] };
                const f2 p = e4 - 4.0f * e2, q = e3 - 4.0f * e1;
                wr[0][jp] = (4.0f * e0 + e4) - 5.0f * e2;
                wr[1][jp] = p + q;
                wr[2][jp] = p - q;
            }
        } else {
#pragma unroll
            for (int jp = 0; jp < 3; ++jp) {
                const int j = 2 * jp;
                const f2 e0 = { pp[j], pp[j + 1] }, e1 = { pp[RS + j], pp[RS + j + 1] }, e2 = { pp[2 * RS + j], pp[2 * RS + j + 1] },
                         e3 = { pp[3 * RS + j], pp[3 * RS + j + 1] }, e4 = { pp[4 * RS + j], pp[4 * RS + j + 1] };
                const f2 p = e3 - e1, d = e2 - e0;
                wr[0][jp] = p + 2.0f * d;
                wr[1][jp] = p - 2.0f * d;
                wr[2][jp] = (4.0f * e0 + e4) - 5.0f * e2;
            }
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float w0 = wr[r][0][0], w1 = wr[r][0][1], w2 = wr[r][1][0], w3 = wr[r][1][1], w4 = wr[r][2][0], w5 = wr[r][2][1];
            const float p = __builtin_fmaf(-4.0f, w2, w4), q = __builtin_fmaf(-4.0f, w1, w3), p2 = w4 - w2, d2 = w3 - w1;
            vd[(r * 6 + 0) * 64] = __builtin_fmaf(-5.0f, w2, __builtin_fmaf(4.0f, w0, w4));
            vd[(r * 6 + 1) * 64] = p + q;
            vd[(r * 6 + 2) * 64] = p - q;
            vd[(r * 6 + 3) * 64] = __builtin_fmaf(2.0f, d2, p2);
            vd[(r * 6 + 4) * 64] = __builtin_fmaf(-2.0f, d2, p2);
            vd[(r * 6 + 5) * 64] = __builtin_fmaf(-5.0f, w3, __builtin_fmaf(4.0f, w1, w5));
        }
    };
#if SPRL_WINO_INTERLEAVE
    constexpr int ILV0 = (RES && !HEADS) ? SPRL_WINO_INTERLEAVE : SPRL_WINO_INTERLEAVE_PLAIN;      // first piece behind filter quad ILV0 - 1
    // the same transform in six PIECES (stage 1 per pair of columns, stage 2 per row) that K step 2c+1 issues between its filter
    // quads - the wave's own MFMAs cover its transform, instead of the other wave of the SIMD having to
    f2 wri[3][3];
    auto produce_piece = [&](int c, auto piece_c) {
        constexpr int piece = decltype(piece_c)::value;
        const float* pp = in_buf + (c & 1) * IN_BUF2 + patch0;
        float* vd = v_buf + (c & 1) * 2 * V_G2 + vdst0;
        if constexpr (piece < 3) {
            constexpr int jp = piece, j = 2 * jp;
            const f2 e0 = { pp[j], pp[j + 1] }, e1 = { pp[RS + j], pp[RS + j + 1] }, e2 = { pp[2 * RS + j], pp[2 * RS + j + 1] },
                     e3 = { pp[3 * RS + j], pp[3 * RS + j + 1] }, e4 = { pp[4 * RS + j], pp[4 * RS + j + 1] };
            if (wa == 0) {
                const f2 p = e4 - 4.0f * e2, q = e3 - 4.0f * e1;
                wri[0][jp] = (4.0f * e0 + e4) - 5.0f * e2;
                wri[1][jp] = p + q;
                wri[2][jp] = p - q;
            } else {
                const f2 p = e3 - e1, d = e2 - e0;
                wri[0][jp] = p + 2.0f * d;
                wri[1][jp] = p - 2.0f * d;
                wri[2][jp] = (4.0f * e0 + e4) - 5.0f * e2;
            }
        } else {
            constexpr int r = piece - 3;
            const float w0 = wri[r][0][0], w1 = wri[r][0][1], w2 = wri[r][1][0], w3 = wri[r][1][1], w4 = wri[r][2][0], w5 = wri[r][2][1];
            const float p = __builtin_fmaf(-4.0f, w2, w4), q = __builtin_fmaf(-4.0f, w1, w3), p2 = w4 - w2, d2 = w3 - w1;
            vd[(r * 6 + 0) * 64] = __builtin_fmaf(-5.0f, w2, __builtin_fmaf(4.0f, w0, w4));
            vd[(r * 6 + 1) * 64] = p + q;
            vd[(r * 6 + 2) * 64] = p - q;
            vd[(r * 6 + 3) * 64] = __builtin_fmaf(2.0f, d2, p2);
            vd[(r * 6 + 4) * 64] = __builtin_fmaf(-2.0f, d2, p2);
            vd[(r * 6 + 5) * 64] = __builtin_fmaf(-5.0f, w3, __builtin_fmaf(4.0f, w1, w5));
        }
    };
#endif
    // A operand: U4[p / 4][s][kb][lane][p % 4] (16-byte loads, four transform positions each); 36-register ring
    f4 a[9];
    auto aload = [&](int s, int q4) {
        a[q4] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(ru, ulane, (q4 * 16 + s) * 4096 + kb * 1024, 0));
    };
    // one K step = one group of 4 input channels: 36 MFMAs; `chunk` >= 0: request that activation chunk behind the filters.
    // FIRST: the accumulators start from the MFMA's constant-zero C operand (no 144-instruction zero fill per wave)
    auto kstep = [&](const float* vg, int s, int chunk, auto first, auto with_gload, int cprod = -1) {
        constexpr bool FIRST = decltype(first)::value;
        constexpr bool WITH_GLOAD = decltype(with_gload)::value;      // this K step carries the phase's activation request
        __builtin_amdgcn_sched_barrier(0);
#if SPRL_WINO_GLOAD_BRANCH
        if (chunk >= 0) gload_to(chunk, pre);         // lab: the round-3 form
#else
        if (WITH_GLOAD && SPRL_WINO_REQ_POS == 0) {    // issued whether or not there is a chunk left to request (see gload_to)
            gload_to(chunk, pre);
            __builtin_amdgcn_sched_barrier(0);         // at the HEAD of the K step: as far ahead of the filter quads requested behind it as it gets
            LAB_STAMP_LATER(st6_);                     // request issued
        }
#endif
#if SPRL_WINO_BROLL
        // B operands (V from LDS) requested a fixed number of MFMA pairs AHEAD of their use, one ds_read2st64 behind every pair:
        // in the plain form the compiler reads eight values, waits, issues their eight MFMAs and only then reads the next eight -
        // the matrix pipe runs dry for an LDS round trip after every batch unless the SIMD's other wave has MFMAs ready.  The
        // compiler's scheduler regroups such reads whatever the source order (and sched_group_barrier only fixes the classes),
        // so the reads and their counted waits are written as asm; the MFMAs hang on the waits through the register operands.
        constexpr int BD = SPRL_WINO_BROLL;
        static_assert(BD >= 1 && BD <= 8, "pairs in flight");
        const unsigned vaddr = (unsigned)(size_t)(__attribute__((address_space(3))) const float*)(vg + lane);
        f2 bq[BD];
#define SPRL_BREAD(dst, pr) asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(dst) : "v"(vaddr), "n"(2 * (pr)), "n"(2 * (pr) + 1))
#define SPRL_BPRE(pr) if constexpr ((pr) < BD) SPRL_BREAD(bq[(pr) % BD], (pr));
#if SPRL_WINO_INTERLEAVE
        // the transform's piece (pr >> 1) - (INTERLEAVE - 1) behind filter quad pr >> 1 (the rolling B reads stay BD pairs ahead across it)
#define SPRL_BPIECE(pr)                                                                                                                \
        if constexpr (((pr) & 1) && ((pr) >> 1) >= ILV0 - 1 && ((pr) >> 1) < ILV0 + 5) {                                               \
            if (WITH_GLOAD && cprod >= 0) {                                                                                            \
                __builtin_amdgcn_sched_barrier(0);                                                                                     \
                produce_piece(cprod, std::integral_constant<int, ((pr) >> 1) - (ILV0 - 1)>{});                                          \
                __builtin_amdgcn_sched_barrier(0);                                                                                     \
            }                                                                                                                          \
        }
#else
#define SPRL_BPIECE(pr)
#endif
#define SPRL_BMFMA(p, bv)                                                                                                              \
        if (FIRST) acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(p) >> 2][(p) & 3], bv, (f4){ 0.0f, 0.0f, 0.0f, 0.0f }, 0, 0, 0);         \
        else acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(p) >> 2][(p) & 3], bv, acc[p], 0, 0, 0);
        // pair pr: wait until only the reads younger than this pair's are outstanding, two MFMAs, the read of pair pr + BD
#define SPRL_BSTEP(pr)                                                                                                                 \
        asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(bq[(pr) % BD]) : "n"((17 - (pr)) < (BD - 1) ? (17 - (pr)) : (BD - 1)));          \
        SPRL_BMFMA(2 * (pr), bq[(pr) % BD][0])                                                                                        \
        SPRL_BMFMA(2 * (pr) + 1, bq[(pr) % BD][1])                                                                                    \
        if constexpr ((pr) + BD < 18) SPRL_BREAD(bq[(pr) % BD], (pr) + BD);                                                           \
        if (((pr) & 1) && s + 1 < 16 && !LAB_OFF(6)) aload(s + 1, (pr) >> 1);                                                         \
        SPRL_BPIECE(pr)                                                                                                                \
        if constexpr (WITH_GLOAD && ((SPRL_WINO_REQ_POS == 1 && (pr) == 9) || (SPRL_WINO_REQ_POS == 2 && (pr) == 17))) {               \
            __builtin_amdgcn_sched_barrier(0);         /* lab: the request in the middle / at the end of the K step */                 \
            gload_to(chunk, pre);                                                                                                      \
            __builtin_amdgcn_sched_barrier(0);                                                                                         \
        }
        SPRL_BPRE(0) SPRL_BPRE(1) SPRL_BPRE(2) SPRL_BPRE(3) SPRL_BPRE(4) SPRL_BPRE(5) SPRL_BPRE(6) SPRL_BPRE(7)
        SPRL_BSTEP(0) SPRL_BSTEP(1) SPRL_BSTEP(2) SPRL_BSTEP(3) SPRL_BSTEP(4) SPRL_BSTEP(5) SPRL_BSTEP(6) SPRL_BSTEP(7) SPRL_BSTEP(8)
        SPRL_BSTEP(9) SPRL_BSTEP(10) SPRL_BSTEP(11) SPRL_BSTEP(12) SPRL_BSTEP(13) SPRL_BSTEP(14) SPRL_BSTEP(15) SPRL_BSTEP(16) SPRL_BSTEP(17)
#undef SPRL_BSTEP
#undef SPRL_BPIECE
#undef SPRL_BMFMA
#undef SPRL_BPRE
#undef SPRL_BREAD
        __builtin_amdgcn_sched_barrier(0);
        return;
#endif
#pragma unroll
        for (int q4 = 0; q4 < 9; ++q4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int p = q4 * 4 + e;
#if defined(SPRL_WINO_LAB_BREUSE)                     // lab build: every B operand (LDS read) feeds TWO MFMAs - half the K loop's LDS reads
                const float bv = vg[(p & ~1) * 64 + lane];
#else
                const float bv = vg[p * 64 + lane];
#endif
                if (FIRST) acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q4][e], bv, (f4){ 0.0f, 0.0f, 0.0f, 0.0f }, 0, 0, 0);
                else acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q4][e], bv, acc[p], 0, 0, 0);
            }
            if (s + 1 < 16 && !LAB_OFF(6)) aload(s + 1, q4);     // (lab bit 6: the filter quads are loaded once and reused)
#if SPRL_WINO_INTERLEAVE
            if (WITH_GLOAD && q4 >= SPRL_WINO_INTERLEAVE - 1 && q4 < SPRL_WINO_INTERLEAVE + 5 && cprod >= 0) {      // (wave-uniform; the last phase has no transform)
                __builtin_amdgcn_sched_barrier(0);
                if (q4 == SPRL_WINO_INTERLEAVE - 1) produce_piece(cprod, std::integral_constant<int, 0>{});
                if (q4 == SPRL_WINO_INTERLEAVE) produce_piece(cprod, std::integral_constant<int, 1>{});
                if (q4 == SPRL_WINO_INTERLEAVE + 1) produce_piece(cprod, std::integral_constant<int, 2>{});
                if (q4 == SPRL_WINO_INTERLEAVE + 2) produce_piece(cprod, std::integral_constant<int, 3>{});
                if (q4 == SPRL_WINO_INTERLEAVE + 3) produce_piece(cprod, std::integral_constant<int, 4>{});
                if (q4 == SPRL_WINO_INTERLEAVE + 4) produce_piece(cprod, std::integral_constant<int, 5>{});
                __builtin_amdgcn_sched_barrier(0);
            }
#endif
            if (WITH_GLOAD && ((SPRL_WINO_REQ_POS == 1 && q4 == 4) || (SPRL_WINO_REQ_POS == 2 && q4 == 8))) {
                __builtin_amdgcn_sched_barrier(0);     // lab: the request in the middle / at the end of the K step
                gload_to(chunk, pre);
                __builtin_amdgcn_sched_barrier(0);
            }
#ifdef SPRL_WINO_STAMPS
            if (WITH_GLOAD && q4 == 4) {
                __builtin_amdgcn_sched_barrier(0);
                LAB_STAMP_LATER(st7_);                 // five of the nine filter quads of the K step consumed
            }
#endif
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // The phase loop is rotated by one K step: K step 0 runs before the loop (it is the one that defines the accumulators), an
    // iteration is then {chunk c+2 to LDS, K step 2c+1, V(c+1), barrier, K step 2c+2}.  Same order of operations as
    // {K step 2c, ..., K step 2c+1, V(c+1), barrier} per phase, only the loop boundary sits elsewhere.
    // (Two activation chunks in flight - a second register set, the phases unrolled in pairs, as the F(3x3) any-board kernel does -
    // were measured here in round 3 and dropped: at 144 accumulators the pair of phases spills 42 registers, 210.5 -> 248.1 us,
    // profiles/r03m_wino_lab_two_chunks_in_flight.log.  THREE workgroups per CU - the 36 positions in two passes of 18, 168 registers,
    // 47 KB of LDS, the partial inverse transform of pass 0 parked in y - also measured and dropped: 278.7 us against 210.8 us;
    // the second read of the activations and the round trip of the partial sums move 2.6x the bytes and every pass pays its own
    // prologue.  The kernel is in the history at commit e8a3335, the run in profiles/r03w_lab_3wg.log.
    // EIGHT boards per workgroup - 8 waves, wave (kb, ph) owns 16 channels x 32 tiles x 18 of the 36 positions, so every filter
    // register feeds two MFMAs and the filter stream is halved; the partial inverse transforms of the two position halves are
    // exchanged through LDS - was built too (commit 1e0bdac): correct, and the filter stream stops mattering (filters loaded once
    // would save 3 % instead of 25 %), but with 131 KB of LDS there is ONE workgroup per CU, its eight waves reach the transform,
    // the barrier and the output stage together, and nothing covers those stretches: 228.8 us against 210.8 us in the lab, 389.7
    // against 419.1 games/s in the bench (profiles/r03x_*, r03y_bench_8b.json).  EIGHT WAVES of 128 registers on the same four
    // boards (wave (kb, ph) with 72 accumulators, four waves per SIMD, two workgroups of eight per CU; commit e7706f2, see
    // DESIGN.md section 5): 227.8 us against 212.7 us - more streams do not help either.  What the PMC says instead: the filter
    // stream is 11.7 TB/s of L2-to-CU traffic, two thirds of the practical L2 rate, at an L2 latency of 271 cycles - a bandwidth,
    // not a latency (tools/conv_pmc.sh, profiles/r03z_conv_pmc_*).)
    f4 rres[4][4];
    auto rload = [&](int r) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            rres[r][i] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rr, ovoff + (r * (SPRL_WINO_WGROUP ? 4096 : 1024) + i * 256), 0, SPRL_WINO_LD_AUX));
    };
    auto phase = [&](int c) {
        const float* vs = v_buf + (c & 1) * 2 * V_G2;
        LAB_STAMP_LATER(st0_);                         // phase start
        if (c + 2 < 8) lstore_from(in_buf + (c & 1) * IN_BUF2, pre);      // chunk c+2 -> in_buf[c & 1] (V(c) was built in phase c-1)
        LAB_STAMP_LATER(st1_);                         // activation chunk stored (it had to arrive)
        // the first residual rows are requested before the LAST K step: the activation registers are dead, no filter quad is
        // requested behind them any more (nothing the K loop waits for queues up behind these HBM loads), and the output stage
        // finds them there one K step later
        // (rows 0-2 there and row 3 behind the first component were measured too: the third row costs registers in the last
        // phases - 203.0 -> 208.6 us, 426.7 -> 410 games/s.)
        if (RES && c == 7) {
            rload(0);
            rload(1);
        }
        kstep(vs + V_G2, 2 * c + 1, c + 3 < 8 ? c + 3 : -1, std::false_type{}, std::true_type{}, (SPRL_WINO_INTERLEAVE && c + 1 < 8) ? c + 1 : -1);
        LAB_STAMP_LATER(st2_);                         // K step 2c+1 issued
        if (!SPRL_WINO_INTERLEAVE && c + 1 < 8) produce(c + 1);
        LAB_STAMP_LATER(st3_);                         // V(c+1) written
        __syncthreads();
        LAB_STAMP_LATER(st4_);                         // barrier passed
        if (c + 1 < 8) kstep(v_buf + ((c + 1) & 1) * 2 * V_G2, 2 * c + 2, -1, std::false_type{}, std::false_type{});
        LAB_STAMP_LATER(st5_);                         // K step 2c+2 issued
        LAB_STAMP_WRITE(4 + 6 * c, st0_);
        LAB_STAMP_WRITE(5 + 6 * c, st1_);
        LAB_STAMP_WRITE(6 + 6 * c, st2_);
        LAB_STAMP_WRITE(7 + 6 * c, st3_);
        LAB_STAMP_WRITE(8 + 6 * c, st4_);
        LAB_STAMP_WRITE(9 + 6 * c, st5_);
        LAB_STAMP_WRITE(64 + 2 * c, st6_);
        LAB_STAMP_WRITE(65 + 2 * c, st7_);
    };

    if constexpr (STEM) {
        // ---- the stem of this workgroup's four boards: wave w = board n0 + w (see StemArgs) ----
        static_assert(!SPRL_WINO_WGROUP, "the group-major lab layout is wired into the plain / residual variants only");
        // The filter quads of K step 0 are requested first and arrive under the stem; channels 0..15 of the stem output ARE the
        // activation chunks 0 and 1 (groups 0..3), so they go straight into the two LDS images as well - the K loop's first V is
        // built from them without a trip to memory; the chunks from 2 on are read back from x0 (L2) behind the drain + barrier.
#pragma unroll
        for (int q4 = 0; q4 < 9; ++q4) aload(0, q4);
        __syncthreads();                               // zero fill done (the stem writes the interiors of both images)
        constexpr int HW = H * W;
        float* const im = v_buf + wave * 300;          // zero-bordered 10x10 images of the board's three planes
        for (int i = lane; i < 300; i += 64) im[i] = 0.0f;
        const int n = n0 + wave;
        const int qs = lane >> 4, l16 = lane & 15;     // K slot of this lane inside a step, column of the 16x16 tile
        const int stile = l16 >> 2, sj = l16 & 3;
        if (n < batch) {
            for (int e = lane; e < 3 * HW; e += 64) {
                const int pl = e / HW, cell = e % HW;
                im[pl * 100 + (cell / W + 1) * 10 + cell % W + 1] = sa.planes[(size_t)n * 3 * HW + e];
            }
        }
        float wf[4][7];                                // A fragments: lane -> (channel 16 kb + l16, K index 4 s + qs)
        int qoff[7];                                   // B fragments: offset of tap q = plane * 100 + dy * 10 + dx
#pragma unroll
        for (int s7 = 0; s7 < 7; ++s7) {
            const int q = 4 * s7 + qs;
            qoff[s7] = q < 27 ? (q / 9) * 100 + ((q % 9) / 3) * 10 + (q % 3) : -1;
#pragma unroll
            for (int kb4 = 0; kb4 < 4; ++kb4) wf[kb4][s7] = q < 27 ? sa.w[(16 * kb4 + l16) * 27 + q] : 0.0f;
        }
        float ssc[4][4], ssh[4][4];                    // channel 16 kb + 4 qs + r  (row of the accumulator tile)
#pragma unroll
        for (int kb4 = 0; kb4 < 4; ++kb4)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                ssc[kb4][r] = sa.scale[16 * kb4 + 4 * qs + r];
                ssh[kb4][r] = sa.shift[16 * kb4 + 4 * qs + r];
            }
        const __amdgpu_buffer_rsrc_t rx0 = __builtin_amdgcn_make_buffer_rsrc((void*)sa.x0, 0, act_bytes, 0x00020000);
        const int cell0 = (4 * (stile >> 1)) * 10 + 4 * (stile & 1) + sj;      // + cb * 10 for row-in-tile cb
        const int scol = 4 * (stile & 1) + sj;
        const int svoff = (n * 4096 + qs * 16 + l16) * 4;       // the per-lane offset carries the board: boards >= batch are not stored
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            f4 sacc[4];
#pragma unroll
            for (int s7 = 0; s7 < 7; ++s7) {
                const float bv = qoff[s7] >= 0 ? im[qoff[s7] + cell0 + cb * 10] : 0.0f;
#pragma unroll
                for (int kb4 = 0; kb4 < 4; ++kb4)
                    sacc[kb4] = s7 == 0 ? __builtin_amdgcn_mfma_f32_16x16x4f32(wf[kb4][s7], bv, (f4){ 0.0f, 0.0f, 0.0f, 0.0f }, 0, 0, 0)
                                        : __builtin_amdgcn_mfma_f32_16x16x4f32(wf[kb4][s7], bv, sacc[kb4], 0, 0, 0);
            }
            const bool on_board = 4 * (stile >> 1) + cb < H && scol < W;
#pragma unroll
            for (int kb4 = 0; kb4 < 4; ++kb4)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = sacc[kb4][r] * ssc[kb4][r] + ssh[kb4][r];      // (two roundings, as the stand-alone stem kernel)
                    const float o = on_board ? (v > 0.0f ? v : 0.0f) : 0.0f;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o), rx0, svoff, ((4 * kb4 + r) * 256 + cb * 64) * 4, 0);
                    if (kb4 == 0)                      // channel 4 qs + r = group r, slot qs: chunk r >> 1, group r & 1 of the chunk
                        in_buf[(r >> 1) * IN_BUF2 + ((r & 1) * 4 + qs) * CS2 + board_off(wave) + (4 * (stile >> 1) + cb + 1) * RS + 4 * (stile & 1) + sj + 1] = o;
                }
        }
        // every wave's stores have reached L2 before any wave of the workgroup reads x0 back (the loads below miss this CU's L1:
        // nothing of these boards has been loaded in this launch)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    } else {   // the first two chunks are requested together: one HBM round trip before the first V can be built, not two
        // (chunk 2 requested here as well, behind the filter quads of K step 0 and into registers of its own, was measured at the
        // end of round 3: 203.0 -> 209.0 us; earlier in the round in front of them: 1-2 % slower too)
        f4 first[2];
        gload_to(0, first);
        gload_to(1, pre);
#pragma unroll
        for (int q4 = 0; q4 < 9; ++q4) aload(0, q4);
        __syncthreads();                               // zero fill done
        lstore_from(in_buf, first);
        lstore_from(in_buf + IN_BUF2, pre);
    }
    gload_to(2, pre);
    LAB_STAMP(1);                                      // first chunks arrived and stored
    __syncthreads();
    produce(0);
    __syncthreads();
    LAB_STAMP(2);                                      // first V built
    kstep(v_buf, 0, -1, std::true_type{}, std::false_type{});
    LAB_STAMP(3);
    for (int c = 0; c < 8; ++c) phase(c);             // stays a rolled loop: peeled or fully unrolled forms measured 5 % slower

    // ---- inverse transform in registers + epilogue ----
#ifdef SPRL_WINO_LAB
    if (LAB_OFF(7)) {                                 // lab: no output stage (one store keeps the accumulators alive)
        f4 sum = acc[0];
#pragma unroll
        for (int q = 1; q < 36; ++q) sum += acc[q];
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, sum), ry, ovoff, 0, 2);
        return;
    }
#endif
    const float relu_floor = relu ? 0.0f : -__builtin_inff();      // ReLU as one v_max against a scalar (no select per element)
    constexpr int OC = 3;                              // HEADS: 2 policy + 1 value head channels
    float hp[HEADS ? OC : 1][4][4];                    // HEADS: this lane's share of the 1x1 head convolutions (its 4 channels)
    if (HEADS) {
#pragma unroll
        for (int o = 0; o < OC; ++o)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) hp[HEADS ? o : 0][i][j] = 0.0f;
    }
    float opair[2][4][4];                              // the 4x4 outputs of the two components of a pair
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        __builtin_amdgcn_sched_barrier(0);
        if ((r & 1) == 0) {                            // components r, r + 1 together: stage 1 of the inverse transform on register pairs
            f2 m[6][6];
#pragma unroll
            for (int p = 0; p < 36; ++p) m[p / 6][p % 6] = (f2){ acc[p][r], acc[p][r + 1] };
            inverse_transform_pair(m, opair[0], opair[1]);
        }
        const float (&o)[4][4] = opair[r & 1];
        const int k = 16 * kb + 4 * c_sub + r;
        const float sc = scale[k], sh = shift[k];
        float hwk[OC];
        if (HEADS) {
#pragma unroll
            for (int oc = 0; oc < OC; ++oc) hwk[oc] = ha.hw[oc * 64 + k];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j] = __builtin_fmaf(o[i][j], sc, sh);
                if (RES) v[j] += rres[r][i][j];
                v[j] = __builtin_fmaxf(v[j], relu_floor);
                if (4 * ty + i >= H || 4 * tx + j >= W) v[j] = 0.0f;       // cells off the board stay zero
            }
            if (HEADS) {
#pragma unroll
                for (int oc = 0; oc < OC; ++oc)
#pragma unroll
                    for (int j = 0; j < 4; ++j) hp[HEADS ? oc : 0][i][j] += hwk[oc] * v[j];
            } else {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), ry, ovoff + (r * (SPRL_WINO_WGROUP ? 4096 : 1024) + i * 256), 0, SPRL_WINO_ST_AUX);
            }
        }
        if (RES && r + 2 < 4) rload(r + 2);
    }
    LAB_STAMP(52);                                     // output stage issued
    if (!HEADS) return;

    // ---- head convolutions fused behind the LAST trunk convolution: only the ReLU'd head maps are written ----
    // Every lane holds the head-convolution partial sums of its 4 channels for its 16 cells; the 16 partials of a cell
    // (4 waves x 4 lane groups) are summed through LDS in a fixed order.  The LDS images are dead by now.
    constexpr int PROW = OC * 256 + 16;                // partial row stride: 32 lanes of a bank group -> 32 banks
    float* const part = lds;                           // [kb * 4 + c_sub][o][(i * 4 + j) * 16 + tl]
    float* const maps = lds + 16 * PROW;               // HEADS = 1: [board][o * HW + row * W + col]; HEADS = 2: [o * HW + cell][board]
    constexpr int HW = H * W;
    constexpr int PIN = 2 * HW, VIN = HW;
#pragma unroll
    for (int oc = 0; oc < OC; ++oc)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) part[(kb * 4 + c_sub) * PROW + oc * 256 + (i * 4 + j) * 16 + tl] = hp[HEADS ? oc : 0][i][j];
    // HEADS = 2: the FC weights of this thread's (at most two) tasks are requested HERE, behind the stores of the partial sums (the
    // accumulators are dead, and not earlier: the compiler sinks the whole output stage to these stores) - they come from L2 while
    // the partial sums go through LDS.  Task A of thread t: policy task t (part t / A, column t % A); task B: hidden task t (half
    // t / HID, unit t % HID) for t < 2 HID, else policy task 256 + t - 2 HID.  Every load is unconditional on a clamped index (no
    // branch per load), what lies outside the task is replaced by zero.
    constexpr int LP = HEADS == 2 ? (PIN + 3) / 4 : 1, LV = HEADS == 2 ? (VIN + 1) / 2 : 1;      // task lengths (<= FC_LP)
    float fwa[LP], fwh[LV], fwc[LP];                  // weights of: policy task tid, hidden task tid, policy task 256 + tid - 2 HID
    int qa0 = -1, qh0 = -1, qc0 = -1, fca = 0, fch = 0, fcc = 0, fkqa = 0, fkqh = 0, fkqc = 0;     // first index (-1: no task), column, part
    float fc_bias[HEADS == 2 ? 5 : 1];
    if constexpr (HEADS == 2) {
        // Buffer descriptors over the two weight matrices: a thread without the task gets an offset past the end, and so does a part
        // that runs past the contraction - the range check answers with zeros.  All loads are issued back to back (scalar offset =
        // row k of the part); nothing waits for them before the barrier below.
        const int A = ha.A, HID = ha.HID;
        const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)ha.pfc_w, 0, (unsigned)(PIN * A) * 4u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)ha.vfc1_w, 0, (unsigned)(VIN * HID) * 4u, 0x00020000);
        constexpr int NONE = 0x7fff0000;
        // t / A and t / HID for t < 512 without an integer division: multiply by ceil(2^20 / d) (exact for t * d < 2^20)
        const unsigned ma = ha.magic_a, mh = ha.magic_h;
        const bool has_a = tid < 4 * A, has_h = tid < 2 * HID;
        const int t2 = 256 + tid - 2 * HID;
        const bool has_c = !has_h && t2 < 4 * A;
        fkqa = has_a ? (int)(((unsigned)tid * ma) >> 20) : 0;
        fca = has_a ? tid - fkqa * A : 0;
        qa0 = has_a ? fkqa * LP : -1;
        fkqh = has_h ? (int)(((unsigned)tid * mh) >> 20) : 0;
        fch = has_h ? tid - fkqh * HID : 0;
        qh0 = has_h ? fkqh * LV : -1;
        fkqc = has_c ? (int)(((unsigned)t2 * ma) >> 20) : 0;
        fcc = has_c ? t2 - fkqc * A : 0;
        qc0 = has_c ? fkqc * LP : -1;
        const int voa = has_a ? (qa0 * A + fca) * 4 : NONE, voh = has_h ? (qh0 * HID + fch) * 4 : NONE, voc = has_c ? (qc0 * A + fcc) * 4 : NONE;
        // a wave none of whose threads has a task kind skips that kind's loads (wave-uniform: Othello - waves 0, 1 take the hidden
        // layer, wave 2 the four left-over policy tasks)
        const int w0 = wave * 64;
        if (w0 < 4 * A) {
#pragma unroll
            for (int k = 0; k < LP; ++k) fwa[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rp, voa, k * A * 4, 0));
        }
        if (w0 < 2 * HID) {
#pragma unroll
            for (int k = 0; k < LV; ++k) fwh[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rv, voh, k * HID * 4, 0));
        }
        if (w0 + 63 >= 2 * HID && 256 + w0 - 2 * HID < 4 * A) {
#pragma unroll
            for (int k = 0; k < LP; ++k) fwc[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rp, voc, k * A * 4, 0));
        }
        const int c64 = tid & 63;
        fc_bias[0] = ha.pfc_b[c64 < A ? c64 : 0];
        fc_bias[1] = ha.pfc_b[c64 + 64 < A ? c64 + 64 : 0];
        fc_bias[2] = ha.vfc1_b[c64 < HID ? c64 : 0];
        fc_bias[3] = ha.vfc2_w[c64 < HID ? c64 : 0];
        fc_bias[4] = ha.vfc2_b[0];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < OC; ++q) {
        const int idx = tid + NTHR2 * q;               // (o, cell) pairs: 768 per workgroup
        const int oc = idx >> 8, cp = idx & 255;
        float sum = 0.0f;
#pragma unroll
        for (int rw = 0; rw < 16; ++rw) sum += part[rw * PROW + oc * 256 + cp];
        const int ij = cp >> 4, t16 = cp & 15;
        const int b = t16 >> 2, tl4 = t16 & 3;
        const int row = 4 * (tl4 >> 1) + (ij >> 2), col = 4 * (tl4 & 1) + (ij & 3);
        sum += ha.hb[oc];
        if (row < H && col < W) {
            if constexpr (HEADS == 2) maps[(oc * HW + row * W + col) * 4 + b] = sum > 0.0f ? sum : 0.0f;
            else maps[b * (OC * HW) + oc * HW + row * W + col] = sum > 0.0f ? sum : 0.0f;
        }
    }
    __syncthreads();
    if constexpr (HEADS == 1) {
        for (int i = tid; i < NIMG2 * OC * HW; i += NTHR2) {
            const int b = i / (OC * HW);
            if (n0 + b < batch) ha.maps_out[(size_t)n0 * (OC * HW) + i] = maps[i];
        }
        return;
    }
    if constexpr (HEADS == 2) {
        // ---- the FC layers: logits = maps[0 : PIN] x pfc_w + b; value = tanh(relu(maps[PIN :] x vfc1_w + b1) . vfc2_w + b2) ----
        // `part` is dead (the barrier above): the partial sums of the tasks go there.  A map value of the four boards is ONE
        // 16-byte LDS read (a broadcast: the threads of a task part read the same address).
        const int A = ha.A, HID = ha.HID;
        const f4* const mq = (const f4*)maps;
        float* const pp = lds;                          // [part 0..3][board][FC_MAXA]
        float* const hpart = lds + 4 * 4 * FC_MAXA;     // [half 0..1][board][64]
        auto task = [&](int q0, const auto& w, int qmax) {
            f4 acc4 = { 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
            for (int k = 0; k < (int)(sizeof(w) / sizeof(float)); ++k) {
                const int q = q0 + k < qmax ? q0 + k : qmax;      // (past the contraction: weight 0 times a finite map value)
                const f4 m = mq[q];
                acc4[0] = __builtin_fmaf(m[0], w[k], acc4[0]);
                acc4[1] = __builtin_fmaf(m[1], w[k], acc4[1]);
                acc4[2] = __builtin_fmaf(m[2], w[k], acc4[2]);
                acc4[3] = __builtin_fmaf(m[3], w[k], acc4[3]);
            }
            return acc4;
        };
        if (qa0 >= 0) {
            const f4 r = task(qa0, fwa, PIN - 1);
#pragma unroll
            for (int b = 0; b < 4; ++b) pp[(fkqa * 4 + b) * FC_MAXA + fca] = r[b];
        }
        if (qh0 >= 0) {
            const f4 r = task(PIN + qh0, fwh, PIN + VIN - 1);
#pragma unroll
            for (int b = 0; b < 4; ++b) hpart[(fkqh * 4 + b) * 64 + fch] = r[b];
        }
        if (qc0 >= 0) {
            const f4 r = task(qc0, fwc, PIN - 1);
#pragma unroll
            for (int b = 0; b < 4; ++b) pp[(fkqc * 4 + b) * FC_MAXA + fcc] = r[b];
        }
        __syncthreads();
        const int b = wave, c64 = lane;                 // thread = (board, column): the wave of a board reduces its hidden layer
        const bool live = n0 + b < batch;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int a = c64 + 64 * e;
            if (a < A) {
                const float v = ((pp[(0 * 4 + b) * FC_MAXA + a] + pp[(1 * 4 + b) * FC_MAXA + a]) + pp[(2 * 4 + b) * FC_MAXA + a]) +
                                pp[(3 * 4 + b) * FC_MAXA + a] + fc_bias[e];
                if (live) ha.logits[(size_t)(n0 + b) * A + a] = v;
            }
        }
        float hsum = 0.0f;
        if (c64 < HID) {
            const float h = (hpart[(0 * 4 + b) * 64 + c64] + hpart[(1 * 4 + b) * 64 + c64]) + fc_bias[2];
            hsum = (h > 0.0f ? h : 0.0f) * fc_bias[3];
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) hsum += __shfl_xor(hsum, m, 64);
        if (lane == 0 && live) ha.value[n0 + b] = tanhf(hsum + fc_bias[4]);
    }
}

// ---------------------------------------------------------------------------------------------------
// General boards (any H x W, e.g. Go 9x9 and 19x19), plain NCHW activations in and out: same Winograd / MFMA core as
// version 2, but a workgroup takes 16 consecutive TILES of the batch (boards are ceil(H/M) x ceil(W/M) tiles, a tile's
// (M+2) x (M+2) input patch reaches into its neighbours), so the loader gathers the patch values of every (channel, tile)
// straight from global memory into LDS patches [slot][tile][PS] (PS odd, slot stride = 16 mod 32: a 32-lane bank group reads
// 32 banks).
//
// Two tilings, chosen by the host per board size (sprl_wino_nchw_tile):
//   M = 4: F(4x4, 3x3), 36 transform positions per tile - 19x19 (5x5 tiles, 361 of 400 cells useful);
//   M = 3: F(3x3, 3x3), 25 transform positions per tile - 9x9 (3x3 tiles, no padding: 225 position-products per board where
//          F(4x4) needs 324).  Interpolation points 0, 1, -1, 2, inf:
//            B^T = [2 -1 -2 1 0; 0 2 1 -1 0; 0 -2 3 -1 0; 0 -1 0 1 0; 0 2 -1 -2 1]
//            G   = [1/2 0 0; 1/2 1/2 1/2; 1/6 -1/6 1/6; 1/6 1/3 2/3; 0 0 1]       (torch_eval.cpp: wino_transform)
//            A^T = [1 1 1 1 0; 0 1 -1 2 0; 0 1 1 4 1]
// ---------------------------------------------------------------------------------------------------
constexpr unsigned SLACK_G = 32;                      // bytes that must be readable behind x and res (see the kernel)

// tools/nchw_lab.py builds this file with -DSPRL_WINO_LAB: a run-time mask switches stages of the any-board kernel off so that
// their cost can be read from the launch time (results are then wrong).  Never defined in the product build.

template <int M>
struct NchwGeom {
    static constexpr int PT = M + 2;                  // patch width
    static constexpr int NP = PT * PT;                // transform positions
    static constexpr int NQ = (NP + 3) / 4;           // filter quads (16-byte A loads of four positions)
    static constexpr int PS = M == 4 ? 37 : 25;       // patch stride (odd)
    static constexpr int SS = 16 * PS;                // channel-slot stride (== 16 mod 32)
    static constexpr int IN_BUF = 8 * SS;
    static constexpr int VG = NP * 64;                // V of one group: [p][c_sub][16 tiles]
    static constexpr int LDS_FLOATS = 2 * IN_BUF + 4 * VG;      // M = 4: 74.8 KB, M = 3: 50.6 KB
    static constexpr int R0 = 3;                      // transform rows of producer half 0 (half 1: PT - 3)
    static constexpr int NLD = R0 * PT;               // patch values a loader thread holds (half 1 of M = 3 uses 2 rows of them)
};

// Y = A^T m A for one (channel, tile), F(3x3, 3x3): rows of the 3x3 output
__device__ __forceinline__ void inverse_transform3(const float (&m)[5][5], float (&o)[3][3]) {
    float tm[3][5];
#pragma unroll
    for (int b = 0; b < 5; ++b) {
        const float s12 = m[1][b] + m[2][b], d12 = m[1][b] - m[2][b];
        tm[0][b] = m[0][b] + s12 + m[3][b];
        tm[1][b] = d12 + 2.0f * m[3][b];
        tm[2][b] = s12 + 4.0f * m[3][b] + m[4][b];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float s12 = tm[i][1] + tm[i][2], d12 = tm[i][1] - tm[i][2];
        o[i][0] = tm[i][0] + s12 + tm[i][3];
        o[i][1] = d12 + 2.0f * tm[i][3];
        o[i][2] = s12 + 4.0f * tm[i][3] + tm[i][4];
    }
}

// Rows [I0, I0 + NR) of Y = A^T m A (the output stage of layout T works on row bands so that a band's four channels can be
// stored as 16-byte vectors without holding the whole tile of all four channels in registers)
template <int M, int I0, int NR>
__device__ __forceinline__ void inverse_rows(const float (&m)[M + 2][M + 2], float (&o)[NR][M]) {
    constexpr int PT = M + 2;
    float tm[NR][PT];
#pragma unroll
    for (int b = 0; b < PT; ++b) {
        const float s12 = m[1][b] + m[2][b], d12 = m[1][b] - m[2][b];
        if constexpr (M == 4) {
            const float s34 = m[3][b] + m[4][b], d34 = m[3][b] - m[4][b];
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const int row = I0 + i;
                tm[i][b] = row == 0 ? m[0][b] + s12 + s34 : row == 1 ? __builtin_fmaf(2.0f, d34, d12)
                         : row == 2 ? __builtin_fmaf(4.0f, s34, s12) : __builtin_fmaf(8.0f, d34, d12) + m[5][b];
            }
        } else {
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const int row = I0 + i;
                tm[i][b] = row == 0 ? m[0][b] + s12 + m[3][b] : row == 1 ? __builtin_fmaf(2.0f, m[3][b], d12)
                                                                          : __builtin_fmaf(4.0f, m[3][b], s12) + m[4][b];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const float s12 = tm[i][1] + tm[i][2], d12 = tm[i][1] - tm[i][2];
        if constexpr (M == 4) {
            const float s34 = tm[i][3] + tm[i][4], d34 = tm[i][3] - tm[i][4];
            o[i][0] = tm[i][0] + s12 + s34;
            o[i][1] = __builtin_fmaf(2.0f, d34, d12);
            o[i][2] = __builtin_fmaf(4.0f, s34, s12);
            o[i][3] = __builtin_fmaf(8.0f, d34, d12) + tm[i][5];
        } else {
            o[i][0] = tm[i][0] + s12 + tm[i][3];
            o[i][1] = __builtin_fmaf(2.0f, tm[i][3], d12);
            o[i][2] = __builtin_fmaf(4.0f, tm[i][3], s12) + tm[i][4];
        }
    }
}

// LAY = 0: NCHW activations (the round-2 form: patch rows gathered 16 + 8 bytes at a time at 4-byte alignment, outputs stored one
//          float at a time).  tools/nchw_lab.py (profiles/r03h_nchw_lab_*.log) showed that form bound by its memory INSTRUCTIONS,
//          not by bandwidth or MFMA: at 8192 9x9 boards 361 us, of which the output stores 96 us, the activation loads 89 us, the
//          residual loads 49 us - while the MFMA loop + transforms alone run in 185 us.
// LAY = 1: layout T, made for this kernel:  x[q][cell][n * tiles + tile][e]  with channel 4 q + e, cell = i * M + j inside the
//          M x M tile, tile = ty * TX + tx, n = board; the tile index runs over the whole BATCH (round 4; before: x[n][q][cell]
//          [tile][e]), so the 16 consecutive tiles of a workgroup are ONE aligned 256-byte row per (q, cell) whatever boards
//          they belong to - with the board-major form a workgroup's rows were cut at board boundaries into unaligned pieces and
//          the kernel moved 1.46 x (9x9) / 1.67 x (19x19) its algorithmic bytes through HBM (profiles/conv_kernel_traffic_go*.json).
//          64 * M^2 * tiles floats per board (9x9 with M = 3: exactly 64 * 81, no padding; 19x19 with M = 4: 64 * 400); the
//          row pitch is cap * tiles, cap = the `batch` argument (the capacity when the count is on the device).  Every global access is then an aligned 16-byte vector of four channels, and lanes that hold neighbouring
//          tiles touch neighbouring vectors: a patch is (M+2)^2 vector loads shared over 8 loader threads per (tile, channel quad),
//          an output tile M^2 vector stores, the residual M^2 vector loads.  Cells of a tile that lie off the board are never read
//          (the loader answers them with zero through the range check) and may hold anything.
template <int M, int OCC, int RES, int LAY>
__global__ void __launch_bounds__(NTHR2, OCC) wino_conv64_nchw_kernel(const float* __restrict__ x, const float* __restrict__ u,
                                                                    const float* __restrict__ scale, const float* __restrict__ shift,
                                                                    const float* __restrict__ res, float* __restrict__ y, int batch,
                                                                    int H, int W, int relu, const unsigned* __restrict__ batch_dev) {
    using Gm = NchwGeom<M>;
    constexpr int PT = Gm::PT, NP = Gm::NP, NQ = Gm::NQ, PS = Gm::PS, SS = Gm::SS, IN_BUF = Gm::IN_BUF, VG = Gm::VG, NLD = Gm::NLD;
    // batch_dev != null: the number of boards is on the device (the engine's leaf count of this round), `batch` is the capacity
    // the grid was sized for; workgroups whose 16 tiles lie past the real count leave at once
    const int cap = batch;                            // boards the activation buffers are laid out for (layout T: see TT)
    if (batch_dev) {
        const int real = (int)*batch_dev;
        batch = real < batch ? real : batch;
    }
    const int TX = (W + M - 1) / M, TY = (H + M - 1) / M, TPB = TX * TY;
    const int TT = cap * TPB;                         // layout T: tiles of the whole buffer - the tile index runs over the BATCH
    if ((long long)blockIdx.x * 16 >= (long long)batch * TPB) return;
    __shared__ __attribute__((aligned(16))) float lds[Gm::LDS_FLOATS];
    float* const in_buf = lds;
    float* const v_buf = lds + 2 * IN_BUF;
    const int tid = (int)threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c_sub = lane >> 4, tl = lane & 15;
    const int gl = wave & 1, wa = wave >> 1;
    const int kb = wave;
    const int HW = H * W;
    // this thread's tile (the same one as loader, tid & 15, and as MFMA column / output lane, lane & 15)
    const int total_tiles = batch * TPB;              // the host keeps batch * tiles-per-board below 2^31
    const int tg = (int)blockIdx.x * 16 + tl;
    const int t_n = tg < total_tiles ? tg / TPB : -1;
    const int t_tt = tg < total_tiles ? tg % TPB : 0;
    const int t_row0 = M * (t_tt / TX), t_col0 = M * (t_tt % TX);

    f4 acc[NP];                                       // first written by the first K step (C operand = 0): no zero fill

    // chunk = groups 2c, 2c+1: 8 channel slots x 16 tiles x NP patch values.  Loader thread = (tile, slot, half): the patch
    // rows 3 half .. 3 half + 2 of one (channel slot, tile) (M = 3, half 1: rows 3 and 4)
    constexpr int MC = M * M, NPC = (NP + 7) / 8;     // layout T: cells per tile, patch cells per loader thread
    const int ld_tile = tl, ld_slot = (tid >> 4) & 7, ld_half = tid >> 7;
    const int ld_quad = (tid >> 4) & 1, ld_part = tid >> 5;      // layout T: loader thread = (tile, channel quad of the chunk, part)
    const int ld_n = t_n;
    const int ld_row = t_row0 - 1 + 3 * ld_half, ld_col = t_col0 - 1;
    const int ld_lds = ld_slot * SS + ld_tile * PS + 3 * PT * ld_half;
    const int ld_rows = (M == 3 && ld_half == 1) ? 2 : 3;
    // Every global access goes through a buffer descriptor with a per-lane byte offset; an element off the board (or a tile
    // past the batch) gets bit 31 set in its offset, which the hardware range check answers with 0 for loads and drops for
    // stores (the host keeps the tensors below 2 GiB) - no predicated loads, no branches, no 64-bit address arithmetic.
    constexpr int OOB = (int)0x80000000;
    // (layout T interleaves the boards: the range check cannot cut at the real count, tiles past it carry t_n = -1 -> OOB)
    const unsigned act_bytes = LAY ? (unsigned)cap * (unsigned)(64 * MC * TPB * 4) : (unsigned)batch * 64u * (unsigned)HW * 4u;
    // the input descriptor starts 16 bytes before x: a patch's first column is col0 - 1, so with the bias no offset is ever
    // negative (a negative per-lane offset plus a positive instruction offset must not depend on how the range check wraps)
    // A patch row (6 or 5 floats) is one 16-byte + one 8- or 4-byte load and a residual row one 16-byte load, at 4-byte alignment;
    // at a board's edge they run into the next row / plane (those values are replaced by zeros below / never stored) and at the
    // very end of the tensor up to 20 bytes past it, at its very start 4 bytes before it (the left neighbour of column 0): x must
    // be readable from 16 bytes before its start, x and res for 32 bytes beyond their end (SLACK_G).
    const __amdgpu_buffer_rsrc_t rx = LAY ? __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, act_bytes, 0x00020000)
                                          : __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)x - 16), 0, act_bytes + 16u + SLACK_G, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)res, 0, res ? act_bytes + (LAY ? 0u : SLACK_G) : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)y, 0, act_bytes, 0x00020000);
    int ld_rowoff[3];
    bool ld_colok[PT];
#pragma unroll
    for (int j = 0; j < PT; ++j) ld_colok[j] = ld_col + j >= 0 && ld_col + j < W;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int row = ld_row + i;
        // (board, channel slot 0 of the chunk, row, first patch column) in bytes; the chunk's channel is added per load
        ld_rowoff[i] = (ld_n >= 0 && row >= 0 && row < H && i < ld_rows) ? ((ld_n * 64) * HW + row * W + ld_col) * 4 + 16 : OOB;
    }
    // layout T: byte offset of (board, quad ld_quad, patch cell) for this thread's patch cells pc = ld_part, ld_part + 8, ...;
    // the chunk's first quad (2 * chunk) is added as the instructions' scalar offset
    int ld_toff[NPC];
#pragma unroll
    for (int e = 0; e < NPC; ++e) {
        const int pc = ld_part + 8 * e;
        const int R = t_row0 + pc / PT - 1, Cc = t_col0 + pc % PT - 1;
        const bool ok = LAY && pc < NP && t_n >= 0 && R >= 0 && R < H && Cc >= 0 && Cc < W;
        ld_toff[e] = ok ? ((ld_quad * MC + (R % M) * M + Cc % M) * TT + t_n * TPB + (R / M) * TX + Cc / M) * 16 : OOB;
    }
    // DEEP (layout T, F(3x3,3x3)): a phase of this tiling is only 50 MFMAs per wave (0.7 us), shorter than a trip to HBM, and
    // tools/nchw_lab.py showed the kernel waiting on its own prefetches (with the MFMA loop removed it lost only 28 % of its
    // time).  There are registers to spare at 100 accumulators, so the activation chunks are requested TWO phases ahead (two
    // register sets, phases unrolled in pairs) and the filter quads two K steps ahead (two rings).
#ifndef SPRL_WINO_F3_RINGS
#define SPRL_WINO_F3_RINGS 2                          // lab: 1 = one filter ring (quads one K step ahead) in the F(3x3) layout-T kernel
#endif
    constexpr bool DEEP = LAY == 1 && M == 3 && SPRL_WINO_F3_RINGS == 2;     // filter rings two K steps deep
    constexpr bool DEEP_ACT = LAY == 1 && (M == 3 || SPRL_WINO_DEEP4);      // activation chunks two phases deep
    float pre0[LAY ? 4 * NPC : NLD], pre1[DEEP_ACT ? 4 * NPC : 1];
    auto gload = [&](int chunk, float* pre) {
        if (LAB_OFF(0)) return;                        // lab: no activation loads
        if constexpr (LAY == 1) {
#pragma unroll
            for (int e = 0; e < NPC; ++e) {
                // chunk < 0: nothing left to request - the loads are issued all the same, with offsets the range check rejects (no
                // memory access).  A branch around them would cost more: the compiler's wait counts must hold on both of its paths,
                // so the `vmcnt(n)` of the next K step come out NPC too small on the path that did issue the loads, and its last
                // filter quads wait for the HBM trip of the activation chunk (see the 8x8 kernel's gload_to).
                const f4 v = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rx, chunk < 0 ? OOB : ld_toff[e], (chunk < 0 ? 0 : chunk) * (2 * MC * 16) * TT, 0));
                pre[4 * e] = v[0]; pre[4 * e + 1] = v[1]; pre[4 * e + 2] = v[2]; pre[4 * e + 3] = v[3];
            }
            return;
        }
        const int g = 2 * chunk + (ld_slot >> 2);
        const int k = 16 * (g >> 2) + 4 * (ld_slot & 3) + (g & 3);
        const int koff = k * HW * 4;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const f4 a4 = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rx, ld_rowoff[i] + koff, 0, 0));
            pre[i * PT + 0] = a4[0]; pre[i * PT + 1] = a4[1]; pre[i * PT + 2] = a4[2]; pre[i * PT + 3] = a4[3];
            if (M == 4) {
                const f2 a2 = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(rx, ld_rowoff[i] + koff + 16, 0, 0));
                pre[i * PT + 4] = a2[0]; pre[i * PT + (M == 4 ? 5 : 4)] = a2[1];
            } else {
                pre[i * PT + 4] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, ld_rowoff[i] + koff + 16, 0, 0));
            }
        }
    };
    auto lstore = [&](float* buf, const float* pre) {
        if (LAB_OFF(1)) return;                        // lab: no patch stores to LDS
        if constexpr (LAY == 1) {
#pragma unroll
            for (int e = 0; e < NPC; ++e) {
                const int pc = ld_part + 8 * e;
                if (pc < NP) {
#pragma unroll
                    for (int ch = 0; ch < 4; ++ch) buf[(ld_quad * 4 + ch) * SS + ld_tile * PS + pc] = pre[4 * e + ch];
                }
            }
            return;
        }
#pragma unroll
        for (int q = 0; q < NLD; ++q)
            if (M == 4 || q < 2 * PT || ld_half == 0) buf[ld_lds + q] = ld_colok[q % PT] ? pre[q] : 0.0f;      // columns off the board
    };
    // producer half 1 reads from patch row 1 on (M = 4: its rows 3..5 use patch rows 1..5) / sits at patch row 3 (M = 3)
    const int patch0 = (gl * 4 + c_sub) * SS + tl * PS + wa * (M == 4 ? PT : 3 * PT);
    const int vdst0 = gl * VG + (3 * wa) * PT * 64 + lane;
    auto produce = [&](int c) {
        const float* pp = in_buf + (c & 1) * IN_BUF + patch0;
        float* vd = v_buf + (c & 1) * 2 * VG + vdst0;
        if (LAB_OFF(2)) return;                        // lab: no input transform
        if (M == 4) {
            f2 wr[3][3];                               // stage 1 on pairs of columns (packed instructions), as in the 8x8 kernel
            if (wa == 0) {
#pragma unroll
                for (int jp = 0; jp < 3; ++jp) {
                    const int j = 2 * jp;
                    const f2 e0 = { pp[j], pp[j + 1] }, e1 = { pp[6 + j], pp[7 + j] }, e2 = { pp[12 + j], pp[13 + j] },
                             e3 = { pp[18 + j], pp[19 + j] }, e4 = { pp[24 + j], pp[25 + j] };
                    const f2 p = e4 - 4.0f * e2, q = e3 - 4.0f * e1;
                    wr[0][jp] = (4.0f * e0 + e4) - 5.0f * e2;
                    wr[1][jp] = p + q;
                    wr[2][jp] = p - q;
                }
            } else {
#pragma unroll
                for (int jp = 0; jp < 3; ++jp) {
                    const int j = 2 * jp;
                    const f2 e0 = { pp[j], pp[j + 1] }, e1 = { pp[6 + j], pp[7 + j] }, e2 = { pp[12 + j], pp[13 + j] },
                             e3 = { pp[18 + j], pp[19 + j] }, e4 = { pp[24 + j], pp[25 + j] };
                    const f2 p = e3 - e1, d = e2 - e0;
                    wr[0][jp] = p + 2.0f * d;
                    wr[1][jp] = p - 2.0f * d;
                    wr[2][jp] = (4.0f * e0 + e4) - 5.0f * e2;
                }
            }
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const float w0 = wr[r][0][0], w1 = wr[r][0][1], w2 = wr[r][1][0], w3 = wr[r][1][1], w4 = wr[r][2][0], w5 = wr[r][2][1];
                const float p = __builtin_fmaf(-4.0f, w2, w4), q = __builtin_fmaf(-4.0f, w1, w3), p2 = w4 - w2, d2 = w3 - w1;
                vd[(r * 6 + 0) * 64] = __builtin_fmaf(-5.0f, w2, __builtin_fmaf(4.0f, w0, w4));
                vd[(r * 6 + 1) * 64] = p + q;
                vd[(r * 6 + 2) * 64] = p - q;
                vd[(r * 6 + 3) * 64] = __builtin_fmaf(2.0f, d2, p2);
                vd[(r * 6 + 4) * 64] = __builtin_fmaf(-2.0f, d2, p2);
                vd[(r * 6 + 5) * 64] = __builtin_fmaf(-5.0f, w3, __builtin_fmaf(4.0f, w1, w5));
            }
        } else {
            // F(3x3,3x3): half 0 builds transform rows 0..2 (from patch rows 0..3), half 1 rows 3..4 (from patch rows 1..4, i.e.
            // pp already points at patch row 3: rows 1..4 are pp[-2 PT] .. pp[PT])
            float wr[3][5];
            if (wa == 0) {
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    const float e0 = pp[j], e1 = pp[5 + j], e2 = pp[10 + j], e3 = pp[15 + j];
                    const float t = e3 - e1;                               // B^T rows: [2 -1 -2 1 0], [0 2 1 -1 0], [0 -2 3 -1 0]
                    wr[0][j] = __builtin_fmaf(2.0f, e0 - e2, t);
                    wr[1][j] = (e2 + e1) - t;
                    wr[2][j] = __builtin_fmaf(3.0f, e2, __builtin_fmaf(-2.0f, e1, -e3));
                }
            } else {
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    const float e1 = pp[j - 10], e2 = pp[j - 5], e3 = pp[j], e4 = pp[5 + j];
                    wr[0][j] = e3 - e1;                                    // [0 -1 0 1 0]
                    wr[1][j] = __builtin_fmaf(2.0f, e1 - e3, e4 - e2);     // [0 2 -1 -2 1]
                    wr[2][j] = 0.0f;
                }
            }
            const int nr = wa == 0 ? 3 : 2;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                if (r < nr) {
                    const float w0 = wr[r][0], w1 = wr[r][1], w2 = wr[r][2], w3 = wr[r][3], w4 = wr[r][4];
                    const float t = w3 - w1;
                    vd[(r * 5 + 0) * 64] = __builtin_fmaf(2.0f, w0 - w2, t);
                    vd[(r * 5 + 1) * 64] = (w2 + w1) - t;
                    vd[(r * 5 + 2) * 64] = __builtin_fmaf(3.0f, w2, __builtin_fmaf(-2.0f, w1, -w3));
                    vd[(r * 5 + 3) * 64] = t;
                    vd[(r * 5 + 4) * 64] = __builtin_fmaf(2.0f, w1 - w3, w4 - w2);
                }
            }
        }
    };
    // F(4x4), lab (SPRL_WINO_F4_INTERLEAVE): the transform in six pieces issued behind filter quads of K step 2c+1 (see the 8x8 kernel)
    f2 wri4[3][3];
    auto produce_piece = [&](int c, auto piece_c) {
        constexpr int piece = decltype(piece_c)::value;
        if constexpr (M == 4) {
            const float* pp = in_buf + (c & 1) * IN_BUF + patch0;
            float* vd = v_buf + (c & 1) * 2 * VG + vdst0;
            if constexpr (piece < 3) {
                constexpr int jp = piece, j = 2 * jp;
                const f2 e0 = { pp[j], pp[j + 1] }, e1 = { pp[6 + j], pp[7 + j] }, e2 = { pp[12 + j], pp[13 + j] },
                         e3 = { pp[18 + j], pp[19 + j] }, e4 = { pp[24 + j], pp[25 + j] };
                if (wa == 0) {
                    const f2 p = e4 - 4.0f * e2, q = e3 - 4.0f * e1;
                    wri4[0][jp] = (4.0f * e0 + e4) - 5.0f * e2;
                    wri4[1][jp] = p + q;
                    wri4[2][jp] = p - q;
                } else {
                    const f2 p = e3 - e1, d = e2 - e0;
                    wri4[0][jp] = p + 2.0f * d;
                    wri4[1][jp] = p - 2.0f * d;
                    wri4[2][jp] = (4.0f * e0 + e4) - 5.0f * e2;
                }
            } else {
                constexpr int r = piece - 3;
                const float w0 = wri4[r][0][0], w1 = wri4[r][0][1], w2 = wri4[r][1][0], w3 = wri4[r][1][1], w4 = wri4[r][2][0], w5 = wri4[r][2][1];
                const float p = __builtin_fmaf(-4.0f, w2, w4), q = __builtin_fmaf(-4.0f, w1, w3), p2 = w4 - w2, d2 = w3 - w1;
                vd[(r * 6 + 0) * 64] = __builtin_fmaf(-5.0f, w2, __builtin_fmaf(4.0f, w0, w4));
                vd[(r * 6 + 1) * 64] = p + q;
                vd[(r * 6 + 2) * 64] = p - q;
                vd[(r * 6 + 3) * 64] = __builtin_fmaf(2.0f, d2, p2);
                vd[(r * 6 + 4) * 64] = __builtin_fmaf(-2.0f, d2, p2);
                vd[(r * 6 + 5) * 64] = __builtin_fmaf(-5.0f, w3, __builtin_fmaf(4.0f, w1, w5));
            }
        }
    };
    constexpr bool F4_ILV = M == 4 && LAY == 1 && SPRL_WINO_F4_INTERLEAVE > 0 && SPRL_WINO_BROLL_F4 > 0;
    const f4* ua = (const f4*)u + kb * 64 + lane;
    f4 a0[NQ], a1[DEEP ? NQ : 1];                     // filter quads of the even / odd K steps (one ring unless DEEP)
    auto aload = [&](int s, int k, f4* a) { a[k] = ua[(size_t)k * (16 * 4 * 64) + s * 256]; };
    // one K step (group of 4 input channels): NP MFMAs from ring `a`; FIRST: the accumulators start from the constant-zero C
    // operand; behind each quad's MFMAs the quad of K step s + AHEAD is requested into the same ring
    auto kstep = [&](const float* vg, int s, auto first, f4* a, int cprod = -1) {
        constexpr bool FIRST = decltype(first)::value;
        constexpr int AHEAD = DEEP ? 2 : 1;
        if (!FIRST && LAB_OFF(3)) return;              // lab: only the first K step (no MFMA loop)
        constexpr int BD = M == 4 ? SPRL_WINO_BROLL_F4 : SPRL_WINO_BROLL_F3;
        if constexpr (BD > 0) {   // rolling B-operand prefetch, as in the 8x8 kernel's K step (see there); NP = 25: the last "pair" is one position
            constexpr int NPR = (NP + 1) / 2;
            const unsigned vaddr = (unsigned)(size_t)(__attribute__((address_space(3))) const float*)(vg + lane);
            f2 bq[BD];
#define SPRL_BREAD(dst, pr)                                                                                                            \
            if constexpr (2 * (pr) + 1 < NP) asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(dst) : "v"(vaddr), "n"(2 * (pr)), "n"(2 * (pr) + 1)); \
            else asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst[0]) : "v"(vaddr), "n"(2 * (pr) * 256));
#define SPRL_BPRE(pr) if constexpr ((pr) < BD && (pr) < NPR) { SPRL_BREAD(bq[(pr) % BD], (pr)) }
#define SPRL_BMFMA(p, bv)                                                                                                              \
            if constexpr ((p) < NP) {                                                                                                  \
                if (FIRST) acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(p) >> 2][(p) & 3], bv, (f4){ 0.0f, 0.0f, 0.0f, 0.0f }, 0, 0, 0); \
                else acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(p) >> 2][(p) & 3], bv, acc[p], 0, 0, 0);                         \
            }
#define SPRL_BSTEP(pr)                                                                                                                 \
            if constexpr ((pr) < NPR) {                                                                                                \
                asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(bq[(pr) % BD]) : "n"((NPR - 1 - (pr)) < (BD - 1) ? (NPR - 1 - (pr)) : (BD - 1))); \
                SPRL_BMFMA(2 * (pr), bq[(pr) % BD][0])                                                                                \
                SPRL_BMFMA(2 * (pr) + 1, bq[(pr) % BD][1])                                                                            \
                if constexpr ((pr) + BD < NPR) { SPRL_BREAD(bq[(pr) % BD], (pr) + BD) }                                               \
                if ((((pr) & 1) || 2 * (pr) + 2 >= NP) && s + AHEAD < 16) aload(s + AHEAD, (pr) >> 1, a);                             \
                if constexpr (F4_ILV && ((pr) & 1) && ((pr) >> 1) >= SPRL_WINO_F4_INTERLEAVE - 1 && ((pr) >> 1) < SPRL_WINO_F4_INTERLEAVE + 5) { \
                    if (cprod >= 0) {                                                                                                  \
                        __builtin_amdgcn_sched_barrier(0);                                                                             \
                        produce_piece(cprod, std::integral_constant<int, (F4_ILV ? ((pr) >> 1) - (SPRL_WINO_F4_INTERLEAVE - 1) : 0)>{}); \
                        __builtin_amdgcn_sched_barrier(0);                                                                             \
                    }                                                                                                                  \
                }                                                                                                                      \
            }
            SPRL_BPRE(0) SPRL_BPRE(1) SPRL_BPRE(2) SPRL_BPRE(3) SPRL_BPRE(4) SPRL_BPRE(5) SPRL_BPRE(6) SPRL_BPRE(7)
            SPRL_BSTEP(0) SPRL_BSTEP(1) SPRL_BSTEP(2) SPRL_BSTEP(3) SPRL_BSTEP(4) SPRL_BSTEP(5) SPRL_BSTEP(6) SPRL_BSTEP(7) SPRL_BSTEP(8)
            SPRL_BSTEP(9) SPRL_BSTEP(10) SPRL_BSTEP(11) SPRL_BSTEP(12) SPRL_BSTEP(13) SPRL_BSTEP(14) SPRL_BSTEP(15) SPRL_BSTEP(16) SPRL_BSTEP(17)
#undef SPRL_BSTEP
#undef SPRL_BMFMA
#undef SPRL_BPRE
#undef SPRL_BREAD
            return;
        }
#pragma unroll
        for (int k = 0; k < NQ; ++k) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (k * 4 + q < NP) {
                    const float b = vg[(k * 4 + q) * 64 + lane];
                    if (FIRST) acc[k * 4 + q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k][q], b, (f4){ 0.0f, 0.0f, 0.0f, 0.0f }, 0, 0, 0);
                    else acc[k * 4 + q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k][q], b, acc[k * 4 + q], 0, 0, 0);
                }
            }
            if (s + AHEAD < 16) aload(s + AHEAD, k, a);
        }
    };
    // rotated by one K step like the 8x8 kernel's loop: K step 0 runs before the loop, an iteration is {K step 2c+1, V(c+1),
    // chunk c+2 to LDS, request chunk c+3 (DEEP: c+4), barrier, K step 2c+2}
    // layout T: the residual rows of the first output band are requested while the last K steps still run (the activation
    // registers are dead by then) - at the point behind which no filter quad is requested any more, so that nothing the K loop
    // waits for queues up behind these HBM loads: DEEP after K step 13, otherwise before K step 15
    constexpr int NR_T = M == 4 ? 2 : M;               // rows per output band (see the output stage)
    constexpr bool EARLY_RES = RES && LAY == 1;
    const int q_out = 4 * kb + c_sub;                  // layout T: this lane's output channels 4 q_out .. 4 q_out + 3
    const int obase = (LAY == 1 && t_n >= 0) ? (q_out * MC * TT + tg) * 16 : OOB;
    f4 rv0[EARLY_RES ? NR_T * M : 1];
    auto early_res = [&]() {
        if constexpr (EARLY_RES) {
#pragma unroll
            for (int cc = 0; cc < NR_T * M; ++cc)
                rv0[cc] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rr, obase + cc * TT * 16, 0, 0));
        }
    };
    // REQ_HEAD (layout T, one chunk in flight = F(4x4)): chunk c+2 goes to LDS and chunk c+3 is requested at the HEAD of the phase,
    // in front of K step 2c+1, as the 8x8 kernel does - loads return in issue order, so the filter quads requested behind an
    // activation request wait for its trip to HBM; from the head of the phase that trip has K step 2c+1 and the transform to
    // complete before K step 2c+2 asks for its quads, from its old place behind the transform only the barrier.
    constexpr bool REQ_HEAD = LAY == 1 && !DEEP_ACT && SPRL_WINO_F4_REQ_HEAD;
    auto phase = [&](int c, float* pre) {
        const float* vs = v_buf + (c & 1) * 2 * VG;
        if (EARLY_RES && !DEEP && c == 7) early_res();
        if constexpr (REQ_HEAD) {
            if (c + 2 < 8) lstore(in_buf + (c & 1) * IN_BUF, pre);
            gload(c + 3 < 8 ? c + 3 : -1, pre);
            __builtin_amdgcn_sched_barrier(0);
        }
        kstep(vs + VG, 2 * c + 1, std::false_type{}, DEEP ? a1 : a0, (F4_ILV && c + 1 < 8) ? c + 1 : -1);
        __builtin_amdgcn_sched_barrier(0);
        if (EARLY_RES && DEEP && c == 6) early_res();
        if (!F4_ILV && c + 1 < 8) produce(c + 1);
        if constexpr (!REQ_HEAD) {
            if (c + 2 < 8) lstore(in_buf + (c & 1) * IN_BUF, pre);
            if (LAY == 1 && !SPRL_WINO_GLOAD_BRANCH) gload(c + (DEEP_ACT ? 4 : 3) < 8 ? c + (DEEP_ACT ? 4 : 3) : -1, pre);      // unconditional (see gload)
            else if (c + (DEEP_ACT ? 4 : 3) < 8) gload(c + (DEEP_ACT ? 4 : 3), pre);
        }
        __syncthreads();
        if (c + 1 < 8) kstep(v_buf + ((c + 1) & 1) * 2 * VG, 2 * c + 2, std::false_type{}, a0);
    };

    {   // chunks 0 and 1 are requested together into registers of their own (no accumulator is live yet): one trip to HBM before
        // the first V can be built, not two; the chunks of the later phases go out behind the filter quads of the first K steps
        float c0[LAY ? 4 * NPC : NLD], c1[LAY ? 4 * NPC : NLD];
        gload(0, c0);
        gload(1, c1);
#pragma unroll
        for (int k = 0; k < NQ; ++k) aload(0, k, a0);
        if (DEEP) {
#pragma unroll
            for (int k = 0; k < NQ; ++k) aload(1, k, a1);
        }
        gload(2, pre0);
        if (DEEP_ACT) gload(3, pre1);
        lstore(in_buf, c0);
        lstore(in_buf + IN_BUF, c1);
    }
    __syncthreads();
    produce(0);
    __syncthreads();
    kstep(v_buf, 0, std::true_type{}, a0);
    if constexpr (DEEP_ACT) {
        for (int c = 0; c < 8; c += 2) {              // chunk c + 2 waits in pre0, chunk c + 3 in pre1
            phase(c, pre0);
            phase(c + 1, pre1);
        }
    } else {
        for (int c = 0; c < 8; ++c) phase(c, pre0);
    }

    if constexpr (LAY == 1) {
        // ---- layout T: inverse transform per row band, the four channels of a cell leave as one 16-byte vector ----
        constexpr int NR = NR_T;                       // rows per band: M = 3 one band of the whole tile, M = 4 two bands of two rows
        const f4 sc4 = *(const f4*)(scale + 4 * q_out), sh4 = *(const f4*)(shift + 4 * q_out);
        const float relu_floor = relu ? 0.0f : -__builtin_inff();
#pragma unroll
        for (int band = 0; band < M / NR; ++band) {
            f4 outv[NR * M], rv[NR * M];
            if (RES) {
                // (requesting band 1's rows here as well, before band 0 is worked on, was measured: 303 -> 319 us at 2048 19x19 boards)
#pragma unroll
                for (int c = 0; c < NR * M; ++c) {
                    if (band == 0) rv[c] = rv0[EARLY_RES ? c : 0];      // (requested during the last K steps)
                    else rv[c] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rr, obase + (band * NR * M + c) * TT * 16, 0, 0));
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float m[PT][PT];
#pragma unroll
                for (int p = 0; p < NP; ++p) m[p / PT][p % PT] = acc[p][r];
                float o[NR][M];
                if (band == 0) inverse_rows<M, 0, NR>(m, o);
                else inverse_rows<M, (M == 4 ? 2 : 0), NR>(m, o);
#pragma unroll
                for (int i = 0; i < NR; ++i)
#pragma unroll
                    for (int j = 0; j < M; ++j) outv[i * M + j][r] = __builtin_fmaf(o[i][j], sc4[r], sh4[r]);
            }
#pragma unroll
            for (int c = 0; c < NR * M; ++c) {
                f4 v = outv[c];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (RES) v[r] += rv[c][r];
                    v[r] = __builtin_fmaxf(v[r], relu_floor);
                }
                if (LAB_OFF(4) && v[0] != 12345.0f) continue;      // lab: no output stores
                // (per-lane offset + nothing else: no scalar-register offset on a 16-byte store - the gfx950 store hazard)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), ry, obase + (band * NR * M + c) * TT * 16, 0, 0);
            }
        }
        return;
    }
    // ---- inverse transform in registers + epilogue, NCHW ----
    // byte offset of (board, channel 0, row0 + i, col0) per output row, bit 31 set when the row is off the board / past the batch
    int o_rowoff[M], o_colsel[M];
#pragma unroll
    for (int i = 0; i < M; ++i) o_rowoff[i] = (t_n >= 0 && t_row0 + i < H) ? ((t_n * 64) * HW + (t_row0 + i) * W + t_col0) * 4 : OOB;
#pragma unroll
    for (int j = 0; j < M; ++j) o_colsel[j] = (t_col0 + j < W) ? 0 : OOB;
    // The per-lane offsets carry the lane's first channel (16 kb + 4 c_sub); output component r adds r * HW * 4 bytes, which is
    // wave-uniform and goes into the instructions' SCALAR offset - no vector arithmetic per store (bit 31 keeps marking
    // "off the board": the hardware adds the scalar offset after the range check's comparison value can no longer wrap).
    const int koff0 = (16 * kb + 4 * c_sub) * HW * 4;
#pragma unroll
    for (int i = 0; i < M; ++i) o_rowoff[i] = o_rowoff[i] < 0 ? OOB : o_rowoff[i] + koff0;
    // residual values of output component r (channel 16 kb + 4 c_sub + r), requested one component ahead of their use
    f4 rres[2][M];
    auto rload = [&](int r, f4 (&dst)[M]) {
#pragma unroll
        for (int i = 0; i < M; ++i) dst[i] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rr, o_rowoff[i], r * HW * 4, 0));
    };
    if (RES) rload(0, rres[0]);
    const float relu_floor = relu ? 0.0f : -__builtin_inff();      // ReLU as one v_max against a scalar (no select per element)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        __builtin_amdgcn_sched_barrier(0);
        if (RES && r + 1 < 4) rload(r + 1, rres[(r + 1) & 1]);
        float m[PT][PT];
#pragma unroll
        for (int p = 0; p < NP; ++p) m[p / PT][p % PT] = acc[p][r];
        float o[M][M];
        if constexpr (M == 4) inverse_transform(m, o);
        else inverse_transform3(m, o);
        const int k = 16 * kb + 4 * c_sub + r;
        const float sc = scale[k], sh = shift[k];
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
            for (int j = 0; j < M; ++j) {
                float v = __builtin_fmaf(o[i][j], sc, sh);
                if (RES) v += rres[r & 1][i][j];
                v = __builtin_fmaxf(v, relu_floor);
                if (LAB_OFF(4) && v != 12345.0f) continue;     // lab: no output stores
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ry, (o_rowoff[i] | o_colsel[j]) + j * 4, r * HW * 4, 0);
            }
    }
}

}  // namespace

// sprl_wino_weight_layout(): 2 = U4[p / 4][s][kb][lane][p % 4] (what wino_transform must produce; 1 was U2[p][s][kb][lane])
extern "C" int sprl_wino_weight_layout(void) { return 2; }

namespace {
template <int HEADS>
int launch_conv64(const float* x, const float* u, const float* scale, const float* shift, const float* res, float* y, int batch,
                  int H, int W, int relu, const unsigned* batch_dev, const HeadArgs& ha_in, void* stream, const StemArgs* stem = nullptr) {
    if (batch <= 0) return 0;
    if ((long long)batch * 16384LL >= 0xFFFFFFFFLL) return -1;      // byte offsets of the buffer descriptors are 32 bits
    const HeadArgs& ha = ha_in;
    const StemArgs sa = stem ? *stem : StemArgs{};
    const dim3 grid((unsigned)((batch + NIMG2 - 1) / NIMG2)), block(NTHR2);
    hipStream_t st = (hipStream_t)stream;
    // RES = 0: the first convolution of a residual block has no residual input - no loads, no adds for it
#define SPRL_LAUNCH_CONV(HH, WW)                                                                                                        \
    do {                                                                                                                                \
        if constexpr (HEADS != 0) {                                                                                                     \
            hipLaunchKernelGGL((wino_conv64_kernel<HH, WW, HEADS, 1, 0>), grid, block, 0, st, x, u, scale, shift, res, y, batch, relu, batch_dev, ha, sa); \
        } else if (stem) {                                                                                                              \
            hipLaunchKernelGGL((wino_conv64_kernel<HH, WW, 0, 0, 1>), grid, block, 0, st, x, u, scale, shift, res, y, batch, relu, batch_dev, ha, sa);     \
        } else if (res) {                                                                                                               \
            hipLaunchKernelGGL((wino_conv64_kernel<HH, WW, 0, 1, 0>), grid, block, 0, st, x, u, scale, shift, res, y, batch, relu, batch_dev, ha, sa);     \
        } else {                                                                                                                        \
            hipLaunchKernelGGL((wino_conv64_kernel<HH, WW, 0, 0, 0>), grid, block, 0, st, x, u, scale, shift, res, y, batch, relu, batch_dev, ha, sa);     \
        }                                                                                                                               \
    } while (0)
    if (H == 8 && W == 8) SPRL_LAUNCH_CONV(8, 8);
    else if (H == 6 && W == 7) SPRL_LAUNCH_CONV(6, 7);
    else if (H == 7 && W == 7) SPRL_LAUNCH_CONV(7, 7);
    else return -1;
#undef SPRL_LAUNCH_CONV
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
}  // namespace

// x, y, res: activations in layout W (4096 floats per board; res may be null; y must not alias x); u: 36*64*64 pre-transformed
// weights in A-operand order (torch_eval.cpp: wino_transform); scale/shift: [64].  batch_dev: optional device pointer to the
// real board count (<= batch, the capacity).  Returns 0, or -1 when the board shape has no kernel here.
extern "C" int sprl_wino_conv64_dev(const float* x, const float* u, const float* scale, const float* shift, const float* res,
                                    float* y, int batch, int H, int W, int relu, const unsigned* batch_dev, void* stream) {
    return launch_conv64<0>(x, u, scale, shift, res, y, batch, H, W, relu, batch_dev, HeadArgs{}, stream);
}
// The FIRST trunk convolution with the stem in its prologue (3 input planes; no residual input): planes [batch][3][H][W] ->
// x0 = ReLU(BN(conv3x3(planes))) (written, layout W: the next convolution's residual) -> y = ReLU(BN(conv3x3(x0))).
// stem_w: [64][27]; y must not alias x0.  Replaces sprl_stem_conv3x3_w + sprl_wino_conv64_dev (one launch less per forward).
extern "C" int sprl_wino_conv64_stem(const float* planes, const float* stem_w, const float* stem_scale, const float* stem_shift, float* x0,
                                     const float* u, const float* scale, const float* shift, float* y, int batch, int H, int W,
                                     const unsigned* batch_dev, void* stream) {
    StemArgs sa{ planes, stem_w, stem_scale, stem_shift, x0 };
    return launch_conv64<0>(x0, u, scale, shift, nullptr, y, batch, H, W, 1, batch_dev, HeadArgs{}, stream, &sa);
}
extern "C" int sprl_wino_conv64(const float* x, const float* u, const float* scale, const float* shift, const float* res,
                                float* y, int batch, int H, int W, int relu, void* stream) {
    return sprl_wino_conv64_dev(x, u, scale, shift, res, y, batch, H, W, relu, nullptr, stream);
}

// The last trunk convolution with both 1x1 head convolutions fused behind it (2 policy + 1 value head channels): the trunk
// output is never written, only the ReLU'd head maps [batch][3 * H * W] (policy maps first); the FC layers follow in
// sprl_tail_fc (cnn_epilogue.hip), which keeps their weights in LDS for 16 boards at a time.  hw/hb: [3][64] / [3].
extern "C" int sprl_wino_conv64_heads(const float* x, const float* u, const float* scale, const float* shift, const float* res,
                                      int batch, int H, int W, const unsigned* batch_dev, const float* hw, const float* hb,
                                      float* maps_out, void* stream) {
    HeadArgs ha{};
    ha.hw = hw; ha.hb = hb; ha.maps_out = maps_out;
    return launch_conv64<1>(x, u, scale, shift, res, nullptr, batch, H, W, 1, batch_dev, ha, stream);
}

// The last trunk convolution with the WHOLE tail behind it: both 1x1 head convolutions, the policy FC, and the value head's
// FC -> ReLU -> FC -> tanh (grid_networks.py:44-51), written straight into logits [batch][A] / value [batch] - the forward ends
// in this launch, neither the trunk output nor the head maps reach memory (round 4: replaces sprl_wino_conv64_heads +
// sprl_tail_fc, one launch and 106-workgroup grid less per forward).  pfc_w: [2 H W][A] (transposed Linear weight), vfc1_w:
// [H W][HID], vfc2_w: [HID].  Returns -1 when the shape is not covered (A > 96, HID > 64: the caller uses the two-kernel form).
extern "C" int sprl_wino_conv64_heads_fc(const float* x, const float* u, const float* scale, const float* shift, const float* res,
                                         int batch, int H, int W, const unsigned* batch_dev, const float* hw, const float* hb,
                                         const float* pfc_w, const float* pfc_b, const float* vfc1_w, const float* vfc1_b,
                                         const float* vfc2_w, const float* vfc2_b, float* logits, float* value, int A, int HID,
                                         void* stream) {
    if (A < 1 || A > FC_MAXA || HID < 1 || HID > 64 || (2 * H * W + 3) / 4 > FC_LP || (H * W + 1) / 2 > FC_LP) return -1;
    if (4 * A > 256 + (256 - 2 * HID)) return -1;     // every task needs a thread (two tasks per thread at most)
    HeadArgs ha{};
    ha.hw = hw; ha.hb = hb;
    ha.pfc_w = pfc_w; ha.pfc_b = pfc_b; ha.vfc1_w = vfc1_w; ha.vfc1_b = vfc1_b; ha.vfc2_w = vfc2_w; ha.vfc2_b = vfc2_b;
    ha.logits = logits; ha.value = value; ha.A = A; ha.HID = HID;
    ha.magic_a = ((1u << 20) + (unsigned)A - 1u) / (unsigned)A;
    ha.magic_h = ((1u << 20) + (unsigned)HID - 1u) / (unsigned)HID;
    return launch_conv64<2>(x, u, scale, shift, res, nullptr, batch, H, W, 1, batch_dev, ha, stream);
}

// Any board size, NCHW activations [batch][64][H][W] in and out (res may be null; y must not alias x).  `u`: the weights in the
// Winograd domain of the tiling sprl_wino_nchw_tile(H, W) selects (torch_eval.cpp: wino_transform with that tile size).
// x must be readable from 16 bytes before its start, x and res for sprl_wino_nchw_slack() bytes behind their end (patch rows
// are fetched 16 + 8 bytes at a time, starting one column left of the tile).  batch_dev: optional device pointer to the real
// board count (<= batch, the capacity the buffers and the grid are sized for).
extern "C" int sprl_wino_nchw_slack(void) { return (int)SLACK_G; }
// output tile size of the any-board kernel for an H x W board: 3 (F(3x3,3x3), 25 positions per tile) when that needs fewer
// position-products than 4 (F(4x4,3x3), 36 per tile) - 9x9: 9 x 25 = 225 against 9 x 36 = 324; 19x19: 4 (900 against 1225)
extern "C" int sprl_wino_nchw_tile(int H, int W) {
    const long long w4 = (long long)((H + 3) / 4) * ((W + 3) / 4) * 36, w3 = (long long)((H + 2) / 3) * ((W + 2) / 3) * 25;
    return w3 < w4 ? 3 : 4;
}
// occ: workgroups per CU the F(3x3) build is held to (0 = default: 2 on layout T, 3 on NCHW; a lab parameter, see below)
static int launch_any_board(const float* x, const float* u, const float* scale, const float* shift, const float* res, float* y, int batch,
                            int H, int W, int relu, int tile, int layout_t, int occ, const unsigned* batch_dev, void* stream) {
    if (batch <= 0) return 0;
    if (H < 1 || W < 1 || H > 64 || W > 64 || (tile != 3 && tile != 4)) return -1;
    const long long tiles = (long long)batch * ((H + tile - 1) / tile) * ((W + tile - 1) / tile);
    if (tiles > 0x7fffffffLL - 16) return -1;
    const long long board_floats = layout_t ? 64LL * tile * tile * ((H + tile - 1) / tile) * ((W + tile - 1) / tile) : 64LL * H * W;
    if ((long long)batch * board_floats * 4 >= 0x7fffff00LL) return -1;   // per-lane byte offsets: bit 31 marks "off the board"
    const dim3 grid((unsigned)((tiles + 15) / 16)), block(NTHR2);
    // F(3x3): 50 KB of LDS per workgroup, so three fit a CU if the kernel is held to 168 registers (11 of them then spill);
    // occ = 2 selects the two-per-CU build without spills (measured: DESIGN.md section 5)
    // (layout T with its two-phase-deep prefetch uses the registers of the two-per-CU build)
    const int f3_occ = occ ? occ : (layout_t ? 2 : 3);
    // RES = 0: the first convolution of a residual block has no residual input - no loads, no adds for it
#define SPRL_LAUNCH_NCHW(MM, OO, LL)                                                                                                      \
    do {                                                                                                                                  \
        if (res)                                                                                                                          \
            hipLaunchKernelGGL((wino_conv64_nchw_kernel<MM, OO, 1, LL>), grid, block, 0, (hipStream_t)stream, x, u, scale, shift, res, y, batch, H, W, relu, batch_dev); \
        else                                                                                                                              \
            hipLaunchKernelGGL((wino_conv64_nchw_kernel<MM, OO, 0, LL>), grid, block, 0, (hipStream_t)stream, x, u, scale, shift, res, y, batch, H, W, relu, batch_dev); \
    } while (0)
    if (layout_t) {
#if SPRL_WINO_DEEP4
        if (tile == 4 && occ == 1) SPRL_LAUNCH_NCHW(4, 1, 1);   // lab: one workgroup per CU, 512 registers
        else
#endif
        if (tile == 4) SPRL_LAUNCH_NCHW(4, 2, 1);
        else if (f3_occ == 2) SPRL_LAUNCH_NCHW(3, 2, 1);
        else SPRL_LAUNCH_NCHW(3, 3, 1);
    } else {
        if (tile == 4) SPRL_LAUNCH_NCHW(4, 2, 0);
        else if (f3_occ == 2) SPRL_LAUNCH_NCHW(3, 2, 0);
        else SPRL_LAUNCH_NCHW(3, 3, 0);
    }
#undef SPRL_LAUNCH_NCHW
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
extern "C" int sprl_wino_conv64_nchw_tiled(const float* x, const float* u, const float* scale, const float* shift, const float* res,
                                           float* y, int batch, int H, int W, int relu, int tile, const unsigned* batch_dev, void* stream) {
    return launch_any_board(x, u, scale, shift, res, y, batch, H, W, relu, tile, 0, 0, batch_dev, stream);
}
// Activations in layout T (see the kernel): x, res, y hold sprl_wino_t_board_floats(H, W, tile) floats per board; `u` from
// sprl_wino_transform_weights_t (input channels of K step s are 4 s .. 4 s + 3).
extern "C" int sprl_wino_t_board_floats(int H, int W, int tile) { return 64 * tile * tile * ((H + tile - 1) / tile) * ((W + tile - 1) / tile); }
extern "C" int sprl_wino_conv64_t(const float* x, const float* u, const float* scale, const float* shift, const float* res, float* y,
                                  int batch, int H, int W, int relu, int tile, const unsigned* batch_dev, void* stream) {
    return launch_any_board(x, u, scale, shift, res, y, batch, H, W, relu, tile, 1, 0, batch_dev, stream);
}
// lab: the same with the occupancy variant chosen by the caller (tools/nchw_lab.py); no environment switch picks it
extern "C" int sprl_wino_conv64_t_occ(const float* x, const float* u, const float* scale, const float* shift, const float* res, float* y,
                                      int batch, int H, int W, int relu, int tile, int occ, const unsigned* batch_dev, void* stream) {
    return launch_any_board(x, u, scale, shift, res, y, batch, H, W, relu, tile, 1, occ, batch_dev, stream);
}
extern "C" int sprl_wino_conv64_nchw_tiled_occ(const float* x, const float* u, const float* scale, const float* shift, const float* res,
                                               float* y, int batch, int H, int W, int relu, int tile, int occ, const unsigned* batch_dev,
                                               void* stream) {
    return launch_any_board(x, u, scale, shift, res, y, batch, H, W, relu, tile, 0, occ, batch_dev, stream);
}
// F(4x4,3x3) tiling, as before
extern "C" int sprl_wino_conv64_nchw_dev(const float* x, const float* u, const float* scale, const float* shift, const float* res,
                                         float* y, int batch, int H, int W, int relu, const unsigned* batch_dev, void* stream) {
    return sprl_wino_conv64_nchw_tiled(x, u, scale, shift, res, y, batch, H, W, relu, 4, batch_dev, stream);
}
extern "C" int sprl_wino_conv64_nchw(const float* x, const float* u, const float* scale, const float* shift, const float* res,
                                     float* y, int batch, int H, int W, int relu, void* stream) {
    return sprl_wino_conv64_nchw_tiled(x, u, scale, shift, res, y, batch, H, W, relu, 4, nullptr, stream);
}

#ifdef SPRL_WINO_STAMPS
extern "C" int sprl_wino_lab_set_trace(unsigned long long* trace, int first, int count) {
    return hipMemcpyToSymbol(HIP_SYMBOL(wino_lab_trace), &trace, sizeof(trace)) == hipSuccess &&
                   hipMemcpyToSymbol(HIP_SYMBOL(wino_lab_trace_first), &first, sizeof(int)) == hipSuccess &&
                   hipMemcpyToSymbol(HIP_SYMBOL(wino_lab_trace_count), &count, sizeof(int)) == hipSuccess
               ? 0 : -1;
}
#endif
#ifdef SPRL_WINO_LAB
extern "C" int sprl_wino_lab_set_dbg(int mask) { return hipMemcpyToSymbol(HIP_SYMBOL(wino_lab_dbg), &mask, sizeof(int)) == hipSuccess ? 0 : -1; }
#endif
