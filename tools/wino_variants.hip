// EXPERIMENTS (diagnostic, never shipped): every variant of the trunk convolution tried in round 2, switched by template
// flags and timed by tools/wino_lab.hip.  The product kernel is sprl_amd/csrc/cnn_wino.hip.
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
// Common to both kernel versions below, K loop over the 16 groups, two groups (8 channels) per phase:
//   * activations: a chunk of 8 channels per board group, zero-bordered 10x10 images in LDS (double buffered; strides
//     chosen so that the 32 lanes of a bank group read 32 different banks);
//   * B operand: for every chunk the threads build V[2 groups][36][4 ch][tiles] ONCE into LDS (thread = one channel, one
//     tile, three of the six transform rows: two factored 1-D transforms of its 6x6 patch, ~80 VALU operations), double
//     buffered, in the lane order the MFMA wants, so a B fetch is one conflict-free ds_read;
//   * A operand: weights pre-transformed on the host (U = G g G^T) and stored in lane order, four transform positions per
//     16-byte load (590 KB per layer, L2-resident), through a 36-register ring one group ahead of the MFMAs;
//   * output: every lane ends the K loop with all 36 positions of its four (channel, tile) pairs: inverse transform in
//     registers, scale/shift, residual, ReLU, 16-byte stores that are contiguous over 16 lanes (256-byte rows of layout W).
// Version 2 (the product): 4 waves = 4 boards = 16 tiles per workgroup, two workgroups per CU.
// Version 3 (SPRL_WINO_V3=1): 8 waves = 8 boards = 32 tiles, one workgroup per CU, weight fragments shared by two waves
// through the vector cache.  Both measure the same in the whole network.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

constexpr int NIMG = 8;                  // boards per workgroup
constexpr int NTHR = 512;
constexpr int RS = 10;                   // row stride of a zero-bordered board image
constexpr int IA = 112, IB = 226;        // board b sits at (b & 1) * IA + (b >> 1) * IB   (== 16 and 2 mod 32)
constexpr int CS = 929;                  // channel-slot stride (== 1 mod 32)
constexpr int IN_BUF = 8 * CS;           // 8 channel slots = 2 groups
constexpr int V_BUF = 36 * 128;          // V[p][tb][c_sub][16 tiles]
constexpr int LDS_FLOATS = 2 * IN_BUF + 4 * V_BUF;       // 133 KB; the output exchange (72 KB) reuses it

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

// ABL: ablation switches for tools/wino_ablate.hip only (0 in the product): 1 no output stage, 2 no V production,
// 4 no weight loads, 8 no MFMAs, 16 no activation loads, 32 weights from an 8 KB footprint, 64 no stores
//
// Version 3 (8 waves = 8 boards = 32 tiles, one workgroup per CU): wave (kb, tb) owns output channels 16kb..16kb+15 for
// all 36 transform positions of tile block tb.  The two waves of a kb request the same weight fragments within a few
// hundred cycles of each other, so the second request is served by the CU's vector cache and the L2 -> CU weight traffic
// is half of version 2's (590 KB per 8 boards instead of per 4) - the quantity that bounds version 2 (about 70 GB/s per
// CU from L2).  In-register inverse transform as in version 2; no second resident workgroup to overlap with.
template <int H, int W, int ABL = 0>
__global__ void __launch_bounds__(NTHR) wino_conv64_kernel(const float* __restrict__ x, const float* __restrict__ u,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           const float* __restrict__ res, float* __restrict__ y, int batch,
                                                           int relu, const unsigned* __restrict__ batch_dev) {
    if (batch_dev) {
        const int real = (int)*batch_dev;
        batch = real < batch ? real : batch;
        if ((int)blockIdx.x * NIMG >= batch) return;
    }
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    float* const in_buf = lds;                        // [2][IN_BUF]
    float* const v_buf = lds + 2 * IN_BUF;            // [2 phases][2 groups][V_BUF]
    const int tid = (int)threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c_sub = lane >> 4, tl = lane & 15;
    const int n0 = (int)blockIdx.x * NIMG;
    // producer role: group gl of the chunk, tile block tbp, transform rows xi in 3wa..3wa+2 (all six nu)
    const int gl = wave & 1, tbp = (wave >> 1) & 1, wa = wave >> 2;
    // consumer role: output channels 16kb.., tile block tb (waves kb and kb + 4 share a SIMD and the weight fragments)
    const int kb = wave & 3, tb = wave >> 2;

    for (int i = tid; i < 2 * IN_BUF; i += NTHR) lds[i] = 0.0f;   // borders stay zero for the whole kernel

    f4 acc[36];
#pragma unroll
    for (int q = 0; q < 36; ++q) acc[q] = (f4){ 0.0f, 0.0f, 0.0f, 0.0f };

    f4 pre[2];                                        // one chunk in flight from HBM (a full phase to land)
    int ldst[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int f = tid + NTHR * it;
        const int b = f >> 7, rem = f & 127;
        const int g2 = rem >> 6, i = (rem >> 4) & 3, cs = (rem >> 2) & 3, tile = rem & 3;
        ldst[it] = (g2 * 4 + cs) * CS + board_off(b) + (4 * (tile >> 1) + i + 1) * RS + 4 * (tile & 1) + 1;
    }
    auto gload = [&](int chunk) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int f = tid + NTHR * it;
            const int n = n0 + (f >> 7);
            pre[it] = (ABL & 16) ? (f4){ 1.0f, 1.0f, 1.0f, 1.0f }
                      : n < batch ? __builtin_nontemporal_load((const f4*)(x + (size_t)n * 4096 + (size_t)chunk * 512 + (size_t)(f & 127) * 4))
                                  : (f4){ 0.0f, 0.0f, 0.0f, 0.0f };
        }
    };
    auto lstore = [&](float* buf) {
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int j = 0; j < 4; ++j) buf[ldst[it] + j] = pre[it][j];
    };
    const int patch0 = (gl * 4 + c_sub) * CS + board_off(tbp * 4 + (tl >> 2)) + ((tl >> 1) & 1) * 4 * RS + (tl & 1) * 4 + wa * RS;
    const int vdst0 = gl * V_BUF + (3 * wa) * 6 * 128 + tbp * 64 + lane;
    auto produce = [&](int c) {
        const float* pp = in_buf + (c & 1) * IN_BUF + patch0;
        float* vd = v_buf + (c & 1) * 2 * V_BUF + vdst0;
        float wr[3][6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const float e0 = pp[j], e1 = pp[RS + j], e2 = pp[2 * RS + j], e3 = pp[3 * RS + j], e4 = pp[4 * RS + j];
            const float st = 4.0f * e0 - 5.0f * e2 + e4;
            if (wa == 0) {
                const float p = e4 - 4.0f * e2, q = e3 - 4.0f * e1;
                wr[0][j] = st;
                wr[1][j] = p + q;
                wr[2][j] = p - q;
            } else {
                const float p = e3 - e1, q = 2.0f * (e2 - e0);
                wr[0][j] = p + q;
                wr[1][j] = p - q;
                wr[2][j] = st;
            }
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float w0 = wr[r][0], w1 = wr[r][1], w2 = wr[r][2], w3 = wr[r][3], w4 = wr[r][4], w5 = wr[r][5];
            const float p = w4 - 4.0f * w2, q = w3 - 4.0f * w1, p2 = w4 - w2, q2 = 2.0f * (w3 - w1);
            vd[(r * 6 + 0) * 128] = 4.0f * w0 - 5.0f * w2 + w4;
            vd[(r * 6 + 1) * 128] = p + q;
            vd[(r * 6 + 2) * 128] = p - q;
            vd[(r * 6 + 3) * 128] = p2 + q2;
            vd[(r * 6 + 4) * 128] = p2 - q2;
            vd[(r * 6 + 5) * 128] = 4.0f * w1 - 5.0f * w3 + w5;
        }
    };
    // A operand: U4[p / 4][s][kb][lane][p % 4]; a 36-register ring one group ahead
    const f4* ua = (const f4*)u + kb * 64 + lane;
    f4 a[9];
    auto aload = [&](int s, int k) {
        a[k] = (ABL & 4) ? (f4){ (float)s, 1.0f, 2.0f, (float)k } : (ABL & 32) ? ua[(k & 1) * 256] : ua[(size_t)k * (16 * 4 * 64) + s * 256];
    };
    const float* vsrc = v_buf + tb * 64 + lane;
    auto mma = [&](const float* vg, int k) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float b = vg[(k * 4 + q) * 128];
            if (ABL & 8) acc[k * 4 + q][0] += a[k][q] * b;
            else acc[k * 4 + q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k][q], b, acc[k * 4 + q], 0, 0, 0);
        }
    };
    // the two waves of a SIMD (tb 0 / tb 1) run produce and MFMA in opposite order
    auto phase = [&](int c) {
        const float* vs = vsrc + (c & 1) * 2 * V_BUF;
        if (tb == 0 && c + 1 < 8 && !(ABL & 2)) produce(c + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2) {
            const int s = 2 * c + g2;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                mma(vs + g2 * V_BUF, k);
                if (s + 1 < 16) aload(s + 1, k);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (tb != 0 && c + 1 < 8 && !(ABL & 2)) produce(c + 1);
        if (c + 2 < 8) {
            lstore(in_buf + (c & 1) * IN_BUF);         // in_buf[c & 1]: V(c) was built in phase c - 1
            if (c + 3 < 8) gload(c + 3);
        }
        __syncthreads();
    };

    const int t_out = 16 * tb + tl;                   // this lane's tile
    const int n = n0 + (t_out >> 2), tile = t_out & 3;
    const size_t plane0 = (size_t)n * 4096 + (size_t)(4 * kb) * 256 + (size_t)(c_sub * 16 + tile * 4);
    f4 rres[4][4];
    auto rload = [&](int r) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            rres[r][i] = (res && n < batch) ? __builtin_nontemporal_load((const f4*)(res + plane0 + (size_t)r * 256 + (size_t)i * 64))
                                            : (f4){ 0.0f, 0.0f, 0.0f, 0.0f };
    };

    gload(0);
    __syncthreads();                                   // zero fill done
    lstore(in_buf);
    gload(1);
#pragma unroll
    for (int k = 0; k < 9; ++k) aload(0, k);
    lstore(in_buf + IN_BUF);
    gload(2);
    __syncthreads();
    produce(0);
    __syncthreads();
    for (int c = 0; c < 8; ++c) phase(c);

    if (ABL & 1) {
        float sum = 0.0f;
#pragma unroll
        for (int q = 0; q < 36; ++q) sum += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
        if (sum == 123.456f) y[tid] = sum;
        return;
    }

    // ---- inverse transform in registers + epilogue ----
    const int ty = tile >> 1, tx = tile & 1;
    rload(0);
    rload(1);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        __builtin_amdgcn_sched_barrier(0);
        float m[6][6];
#pragma unroll
        for (int p = 0; p < 36; ++p) m[p / 6][p % 6] = acc[p][r];
        float o[4][4];
        inverse_transform(m, o);
        const int k = 16 * kb + 4 * c_sub + r;
        const float sc = scale[k], sh = shift[k];
        if (n < batch) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f4 v;
                const f4 rv = rres[r][i];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[j] = o[i][j] * sc + sh + rv[j];
                    if (relu) v[j] = v[j] > 0.0f ? v[j] : 0.0f;
                    if (4 * ty + i >= H || 4 * tx + j >= W) v[j] = 0.0f;       // cells off the board stay zero
                }
                if (!(ABL & 64) || v[0] == 123.456f) __builtin_nontemporal_store(v, (f4*)(y + plane0 + (size_t)r * 256 + (size_t)i * 64));
            }
        }
        if (r + 2 < 4) rload(r + 2);
    }
}

// ---------------------------------------------------------------------------------------------------
// Version 2: 4 waves = 4 boards = 16 tiles per workgroup, TWO workgroups per CU.
// Wave kb owns output channels 16kb..16kb+15 for ALL 36 transform positions of the 16 tiles (36 accumulator tiles,
// 144 registers), so a lane ends the K loop holding every position of its four (channel, tile) pairs and the inverse
// transform needs no exchange.  The two resident workgroups of a CU drift out of phase, so the prologue / output stage of one
// overlaps the MFMA phases of the other (`stagger` can force an offset in the first round; off by default), and so does their HBM
// traffic.  Same LDS images, V production, weight layout and activation layout as above.
// ---------------------------------------------------------------------------------------------------
// network tail fused into the last trunk convolution (TAIL = 1: 2 policy + 1 value head channels)
struct TailArgs {
    const float *hw, *hb, *pfc_w, *pfc_b, *vfc1_w, *vfc1_b, *vfc2_w, *vfc2_b;
    float *logits, *value;
    int A, HID;
    float* maps_out;        // TAIL = 2: only the head maps [batch][3 * H * W] are written (the FC layers run in their own kernel)
};

constexpr int NIMG2 = 4, NTHR2 = 256;
constexpr int CS2 = 449;                 // channel-slot stride for 4 boards (== 1 mod 32)
constexpr int IN_BUF2 = 8 * CS2;
constexpr int V_G2 = 36 * 64;            // V of one group: [p][c_sub][16 tiles]
constexpr int LDS_FLOATS2 = 2 * IN_BUF2 + 4 * V_G2;      // 65.6 KB

template <int H, int W, int ABL = 0, int TAIL = 0>
__global__ void __launch_bounds__(NTHR2, 2) wino_conv64_v2_kernel(const float* __restrict__ x, const float* __restrict__ u,
                                                                  const float* __restrict__ scale, const float* __restrict__ shift,
                                                                  const float* __restrict__ res, float* __restrict__ y, int batch,
                                                                  int relu, int stagger, const unsigned* __restrict__ batch_dev,
                                                                  TailArgs ta) {
    // batch_dev != null: the number of boards is on the device (the engine's leaf count of this round), `batch` is the
    // capacity the grid was sized for; workgroups past the real count leave at once
    if (batch_dev) {
        const int real = (int)*batch_dev;
        batch = real < batch ? real : batch;
        if ((int)blockIdx.x * NIMG2 >= batch) return;
    }
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS2];
    float* const in_buf = lds;                        // [2][IN_BUF2]
    float* const v_buf = lds + 2 * IN_BUF2;           // [2 phases][2 groups][V_G2]
    const int tid = (int)threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c_sub = lane >> 4, tl = lane & 15;
    const int n0 = (int)blockIdx.x * NIMG2;
    const int gl = wave & 1, wa = wave >> 1;          // producer role: group of the chunk, transform rows 3wa..3wa+2
    const int kb = wave;                              // consumer role: output channels 16kb..16kb+15

    for (int i = tid; i < 2 * IN_BUF2; i += NTHR2) lds[i] = 0.0f;    // borders stay zero for the whole kernel

    // optional, first round only: the workgroup in the second wave slot of its SIMD starts `stagger` sleeps late
    if (stagger > 0 && (int)blockIdx.x < 2 * 256) {
        const unsigned slot = __builtin_amdgcn_s_getreg((3 << 11) | 4);   // HW_ID.WAVE_ID
        if (slot & 1u)
            for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(127);
    }

    f4 acc[36];
#pragma unroll
    for (int q = 0; q < 36; ++q) acc[q] = (f4){ 0.0f, 0.0f, 0.0f, 0.0f };

    f4 pre[2];
    int ldst[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int f = tid + NTHR2 * it;
        const int b = f >> 7, rem = f & 127;
        const int g2 = rem >> 6, i = (rem >> 4) & 3, cs = (rem >> 2) & 3, tile = rem & 3;
        ldst[it] = (g2 * 4 + cs) * CS2 + board_off(b) + (4 * (tile >> 1) + i + 1) * RS + 4 * (tile & 1) + 1;
    }
    auto gload_to = [&](int chunk, f4 (&dst)[2]) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int f = tid + NTHR2 * it;
            const int n = n0 + (f >> 7);
            dst[it] = (ABL & 16) ? (f4){ 1.0f, 1.0f, 1.0f, 1.0f }
                      : n < batch ? __builtin_nontemporal_load((const f4*)(x + (size_t)n * 4096 + (size_t)chunk * 512 + (size_t)(f & 127) * 4))
                                  : (f4){ 0.0f, 0.0f, 0.0f, 0.0f };
        }
    };
    auto lstore_from = [&](float* buf, const f4 (&src)[2]) {
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int j = 0; j < 4; ++j) buf[ldst[it] + j] = src[it][j];
    };
    auto gload = [&](int chunk) { gload_to(chunk, pre); };
    auto lstore = [&](float* buf) { lstore_from(buf, pre); };
    const int patch0 = (gl * 4 + c_sub) * CS2 + board_off(tl >> 2) + ((tl >> 1) & 1) * 4 * RS + (tl & 1) * 4 + wa * RS;
    const int vdst0 = gl * V_G2 + (3 * wa) * 6 * 64 + lane;
    auto produce = [&](int c) {
        const float* pp = in_buf + (c & 1) * IN_BUF2 + patch0;
        float* vd = v_buf + (c & 1) * 2 * V_G2 + vdst0;
        float wr[3][6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const float e0 = pp[j], e1 = pp[RS + j], e2 = pp[2 * RS + j], e3 = pp[3 * RS + j], e4 = pp[4 * RS + j];
            const float st = 4.0f * e0 - 5.0f * e2 + e4;
            if (wa == 0) {
                const float p = e4 - 4.0f * e2, q = e3 - 4.0f * e1;
                wr[0][j] = st;
                wr[1][j] = p + q;
                wr[2][j] = p - q;
            } else {
                const float p = e3 - e1, q = 2.0f * (e2 - e0);
                wr[0][j] = p + q;
                wr[1][j] = p - q;
                wr[2][j] = st;
            }
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float w0 = wr[r][0], w1 = wr[r][1], w2 = wr[r][2], w3 = wr[r][3], w4 = wr[r][4], w5 = wr[r][5];
            const float p = w4 - 4.0f * w2, q = w3 - 4.0f * w1, p2 = w4 - w2, q2 = 2.0f * (w3 - w1);
            vd[(r * 6 + 0) * 64] = 4.0f * w0 - 5.0f * w2 + w4;
            vd[(r * 6 + 1) * 64] = p + q;
            vd[(r * 6 + 2) * 64] = p - q;
            vd[(r * 6 + 3) * 64] = p2 + q2;
            vd[(r * 6 + 4) * 64] = p2 - q2;
            vd[(r * 6 + 5) * 64] = 4.0f * w1 - 5.0f * w3 + w5;
        }
    };
    // A operand: U4[p / 4][s][kb][lane][p % 4] (16-byte loads, four transform positions each); a 36-register ring that
    // runs one group (36 MFMAs) ahead
    const f4* ua = (const f4*)u + kb * 64 + lane;
    f4 a[9];
    auto aload = [&](int s, int k) {
        a[k] = (ABL & 4) ? (f4){ (float)s, 1.0f, 2.0f, (float)k } : (ABL & 32) ? ua[(k & 1) * 256] : ua[(size_t)k * (16 * 4 * 64) + s * 256];
    };
    auto mma = [&](const float* vg, int k) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float b = vg[(k * 4 + q) * 64 + lane];
            if (ABL & 8) acc[k * 4 + q][0] += a[k][q] * b;
            else acc[k * 4 + q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k][q], b, acc[k * 4 + q], 0, 0, 0);
        }
    };
    auto phase = [&](int c) {
        const float* vs = v_buf + (c & 1) * 2 * V_G2;
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2) {
            const int s = 2 * c + g2;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                mma(vs + g2 * V_G2, k);
                if (s + 1 < 16) aload(s + 1, k);
            }
        }
#ifndef WINO_NO_SCHED_BARRIER
        __builtin_amdgcn_sched_barrier(0);
#endif
        if (c + 1 < 8 && !(ABL & 2)) produce(c + 1);
        if (c + 2 < 8) {
            lstore(in_buf + (c & 1) * IN_BUF2);        // in_buf[c & 1]: V(c) was built in phase c - 1
            if (c + 3 < 8) gload(c + 3);
        }
        __syncthreads();
    };

    const int n = n0 + (tl >> 2), tile = tl & 3;      // this lane's tile
    const size_t plane0 = (size_t)n * 4096 + (size_t)(4 * kb) * 256 + (size_t)(c_sub * 16 + tile * 4);
    f4 rres[4][4];
    auto rload = [&](int r) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            rres[r][i] = (res && n < batch) ? __builtin_nontemporal_load((const f4*)(res + plane0 + (size_t)r * 256 + (size_t)i * 64))
                                            : (f4){ 0.0f, 0.0f, 0.0f, 0.0f };
    };

    {   // the first two chunks are requested together: one HBM round trip before the first V can be built, not two
        f4 first[2];
        gload_to(0, first);
        gload(1);
#pragma unroll
        for (int k = 0; k < 9; ++k) aload(0, k);
        __syncthreads();                               // zero fill done
        lstore_from(in_buf, first);
        lstore(in_buf + IN_BUF2);
    }
    gload(2);
    __syncthreads();
    produce(0);
    __syncthreads();
    for (int c = 0; c < 8; ++c) phase(c);

    if (ABL & 1) {
        float sum = 0.0f;
#pragma unroll
        for (int q = 0; q < 36; ++q) sum += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
        if (sum == 123.456f) y[tid] = sum;
        return;
    }

    // ---- inverse transform in registers + epilogue ----
    const int ty = tile >> 1, tx = tile & 1;
    rload(0);
    rload(1);
    constexpr int OC = 3;                              // TAIL: 2 policy + 1 value head channels
    float hp[TAIL ? OC : 1][4][4];                     // TAIL: this lane's share of the 1x1 head convolutions (its 4 channels)
    if (TAIL) {
#pragma unroll
        for (int o = 0; o < OC; ++o)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) hp[TAIL ? o : 0][i][j] = 0.0f;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        __builtin_amdgcn_sched_barrier(0);
        float m[6][6];
#pragma unroll
        for (int p = 0; p < 36; ++p) m[p / 6][p % 6] = acc[p][r];
        float o[4][4];
        inverse_transform(m, o);
        const int k = 16 * kb + 4 * c_sub + r;
        const float sc = scale[k], sh = shift[k];
        float hwk[OC];
        if (TAIL) {
#pragma unroll
            for (int oc = 0; oc < OC; ++oc) hwk[oc] = ta.hw[oc * 64 + k];
        }
        if (n < batch) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f4 v;
                const f4 rv = rres[r][i];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[j] = o[i][j] * sc + sh + rv[j];
                    if (relu) v[j] = v[j] > 0.0f ? v[j] : 0.0f;
                    if (4 * ty + i >= H || 4 * tx + j >= W) v[j] = 0.0f;       // cells off the board stay zero
                }
                if (TAIL) {
#pragma unroll
                    for (int oc = 0; oc < OC; ++oc)
#pragma unroll
                        for (int j = 0; j < 4; ++j) hp[TAIL ? oc : 0][i][j] += hwk[oc] * v[j];
                } else if (!(ABL & 64) || v[0] == 123.456f) {
                    __builtin_nontemporal_store(v, (f4*)(y + plane0 + (size_t)r * 256 + (size_t)i * 64));
                }
            }
        }
        if (r + 2 < 4) rload(r + 2);
    }
    if (!TAIL) return;

    // ---- fused tail (last trunk convolution only): head maps -> policy FC, value FC -> ReLU -> FC -> tanh ----
    // The trunk output never goes to memory.  Every lane holds the head-convolution partial sums of its 4 channels for its
    // 16 cells; the 16 partials of a cell (4 waves x 4 lane groups) are summed through LDS in a fixed order.
    constexpr int PROW = OC * 256 + 16;                // partial row stride: 32 lanes of a bank group -> 32 banks
    float* const part = lds;                           // [kb * 4 + c_sub][o][(i * 4 + j) * 16 + tl]
    float* const maps = lds + 16 * PROW;               // [board][o * HW + row * W + col]
    constexpr int HW = H * W;
#pragma unroll
    for (int oc = 0; oc < OC; ++oc)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) part[(kb * 4 + c_sub) * PROW + oc * 256 + (i * 4 + j) * 16 + tl] = hp[TAIL ? oc : 0][i][j];
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
        sum += ta.hb[oc];
        if (row < H && col < W) maps[b * (OC * HW) + oc * HW + row * W + col] = sum > 0.0f ? sum : 0.0f;
    }
    __syncthreads();
    if (TAIL == 2) {
        for (int i = tid; i < NIMG2 * OC * HW; i += NTHR2) {
            const int b = i / (OC * HW);
            if (n0 + b < batch) ta.maps_out[(size_t)n0 * (OC * HW) + i] = maps[i];
        }
        return;
    }
    {
        const int b = wave, nb = n0 + b;               // one wave per board
        const float* pm = maps + b * (OC * HW);        // policy maps [2 * HW], then the value map [HW]
        const float* vm = pm + 2 * HW;
        if (nb < batch) {
            for (int a = lane; a < ta.A; a += 64) {
                float sum = ta.pfc_b[a];
                for (int q = 0; q < 2 * HW; ++q) sum += pm[q] * ta.pfc_w[q * ta.A + a];
                ta.logits[(size_t)nb * ta.A + a] = sum;
            }
        }
        float hsum = 0.0f;
        if (lane < ta.HID) {
            float h = ta.vfc1_b[lane];
            for (int q = 0; q < HW; ++q) h += vm[q] * ta.vfc1_w[q * ta.HID + lane];
            hsum = (h > 0.0f ? h : 0.0f) * ta.vfc2_w[lane];
        }
#pragma unroll
        for (int msk = 32; msk >= 1; msk >>= 1) hsum += __shfl_xor(hsum, msk, 64);
        if (lane == 0 && nb < batch) ta.value[nb] = tanhf(hsum + ta.vfc2_b[0]);
    }
}

// ---------------------------------------------------------------------------------------------------
// Version 4: version 2's structure (4 waves = 4 boards = 16 tiles per workgroup, two workgroups per CU, wave kb owns output
// channels 16kb..16kb+15 for all 36 transform positions), but the A operand is no longer streamed from L2 in the Winograd
// domain.  Version 2 is bound by that stream: 36 x 64 x 64 floats = 590 KB per workgroup of 4 boards, about 28 B/clk per CU
// with two resident workgroups - the rate an L2-served stream into one CU reaches.  Version 4 streams the 3x3 filters
// themselves (9 floats per (out, in) pair: a quarter of the bytes) and builds U = G g G^T in registers in front of the MFMAs
// that use it: per K step and lane 33 + 66 vector operations (rows of T = G g just in time, then each row of U = T G^T),
// which the scheduler places into the gaps between the step's 36 MFMAs.
//   * filters: two register sets; a step's loads go out at its START and are first used a whole step later;
//   * activations: chunk c+3 is requested after the second K step's filter loads (so no filter wait stands behind a young
//     HBM request: vmcnt retires in order) and goes to LDS between the two K steps of the next phase;
//   * all global memory through buffer descriptors: wave-uniform byte offset in a scalar register + one 32-bit per-lane
//     offset, rows past the batch read as zero and are never stored (hardware range check): no `n < batch` branches.
// FLAGS (experiments, tools/wino_lab.hip): 1 = no scheduling barrier between the MFMA block and the V production,
//   2 = persistent workgroups (grid = resident count, loop over board groups), 4 = explicit MFMA / VALU / LDS-read groups.
// ---------------------------------------------------------------------------------------------------
// yv[0..5] = G x for one column x = (x0, x1, x2): the 6x3 filter transform of F(4x4, 3x3)
__device__ __forceinline__ void filt6(float x0, float x1, float x2, float (&yv)[6]) {
    const float t = x0 + x2;
    const float p = x0 * (1.0f / 24.0f) + x2 * (1.0f / 6.0f), q = x1 * (1.0f / 12.0f);
    yv[0] = x0 * 0.25f;
    yv[1] = (t + x1) * (-1.0f / 6.0f);
    yv[2] = (t - x1) * (-1.0f / 6.0f);
    yv[3] = p + q;
    yv[4] = p - q;
    yv[5] = x2;
}

constexpr int G9_FLOATS = 16 * 4 * 2 * 64 * 4 + 16 * 4 * 64;      // G8[s][kb][2][lane][4] (taps 0..7) + G1[s][kb][lane] (tap 8)

template <int H, int W, int FLAGS, int TAIL = 0>
__global__ void __launch_bounds__(NTHR2, 2) wino_conv64_v4_kernel(const float* __restrict__ x, const float* __restrict__ wts,
                                                                  const float* __restrict__ scale, const float* __restrict__ shift,
                                                                  const float* __restrict__ res, float* __restrict__ y, int batch,
                                                                  int relu, const unsigned* __restrict__ batch_dev, TailArgs ta) {
    constexpr bool PERSIST = (FLAGS & 2) != 0;
    // batch_dev != null: the number of boards is on the device (the engine's leaf count of this round), `batch` is the
    // capacity the grid was sized for; workgroups past the real count leave at once
    if (batch_dev) {
        const int real = (int)*batch_dev;
        batch = real < batch ? real : batch;
    }
    const int ngroups = (batch + NIMG2 - 1) / NIMG2;
    if ((int)blockIdx.x >= ngroups) return;
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS2];
    float* const in_buf = lds;                        // [2][IN_BUF2]
    float* const v_buf = lds + 2 * IN_BUF2;           // [2 phases][2 groups][V_G2]
    const int tid = (int)threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c_sub = lane >> 4, tl = lane & 15;
    const int gl = wave & 1, wa = wave >> 1;          // producer role: group of the chunk, transform rows 3wa..3wa+2
    const int kb = wave;                              // consumer role: output channels 16kb..16kb+15

    for (int i = tid; i < 2 * IN_BUF2; i += NTHR2) lds[i] = 0.0f;    // borders stay zero for the whole kernel

    int ldst[2];
    unsigned xoff[2];                                 // bytes
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int f = tid + NTHR2 * it;
        const int b = f >> 7, rem = f & 127;
        const int g2 = rem >> 6, i = (rem >> 4) & 3, cs = (rem >> 2) & 3, tile = rem & 3;
        ldst[it] = (g2 * 4 + cs) * CS2 + board_off(b) + (4 * (tile >> 1) + i + 1) * RS + 4 * (tile & 1) + 1;
        xoff[it] = (unsigned)(b * 4096 + rem * 4) * 4u;
    }
    const int patch0 = (gl * 4 + c_sub) * CS2 + board_off(tl >> 2) + ((tl >> 1) & 1) * 4 * RS + (tl & 1) * 4 + wa * RS;
    const int vdst0 = gl * V_G2 + (3 * wa) * 6 * 64 + lane;
    const unsigned act_bytes = (unsigned)batch * 16384u;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, act_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)res, 0, res ? act_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)y, 0, (y && !TAIL) ? act_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)wts, 0, (FLAGS & 8192) ? 36u * 4096u * 4u : (FLAGS & 4096) ? 24u * 4096u * 4u : (unsigned)G9_FLOATS * 4u, 0x00020000);
    const int wlane16 = lane * 16, wlane4 = lane * 4;
    const int ooff = ((tl >> 2) * 4096 + c_sub * 16 + (tl & 3) * 4) * 4;   // bytes: this lane's board, channel slot and tile
    const int tile = tl & 3, ty = tile >> 1, tx = tile & 1;

    auto group = [&](const int grp) {
        const int n0 = grp * NIMG2;
        // the filters are the same for every board group: keep the optimiser from hoisting the first K step's loads and
        // transforms out of the persistent loop (36 registers held across everything else)
        int wbase = kb * 2048;
        if (PERSIST) asm volatile("" : "+s"(wbase));
        f4 acc[36];
#pragma unroll
        for (int q = 0; q < 36; ++q) acc[q] = (f4){ 0.0f, 0.0f, 0.0f, 0.0f };
        f4 pre[2];
        // per-lane byte offsets carry the board (so the range check drops boards past the batch whatever the scalar offset adds)
        const int xvoff[2] = { (int)xoff[0] + n0 * 16384, (int)xoff[1] + n0 * 16384 };
        auto gload_to = [&](int chunk, f4 (&dst)[2]) {
#pragma unroll
            for (int it = 0; it < 2; ++it)
                dst[it] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rx, xvoff[it], chunk * 2048, 2));
        };
        auto lstore_from = [&](float* buf, const f4 (&src)[2]) {
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int j = 0; j < 4; ++j) buf[ldst[it] + j] = src[it][j];
        };
        auto produce = [&](int c) {
            const float* pp = in_buf + (c & 1) * IN_BUF2 + patch0;
            float* vd = v_buf + (c & 1) * 2 * V_G2 + vdst0;
            float wr[3][6];
            constexpr float C6 = -1.0f / 6.0f, C24 = 1.0f / 24.0f;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const float e0 = pp[j], e1 = pp[RS + j], e2 = pp[2 * RS + j], e3 = pp[3 * RS + j], e4 = pp[4 * RS + j];
                if (FLAGS & 16) {                         // rows scaled by (1/4, -1/6, -1/6, 1/24, 1/24, 1)
                    if (wa == 0) {
                        const float p = C6 * e4 - 4.0f * C6 * e2, q = C6 * e3 - 4.0f * C6 * e1;
                        wr[0][j] = e0 - 1.25f * e2 + 0.25f * e4;
                        wr[1][j] = p + q;
                        wr[2][j] = p - q;
                    } else {
                        const float p = C24 * (e3 - e1), q = 2.0f * C24 * (e2 - e0);
                        wr[0][j] = p + q;
                        wr[1][j] = p - q;
                        wr[2][j] = 4.0f * e0 - 5.0f * e2 + e4;
                    }
                    continue;
                }
                const float st = 4.0f * e0 - 5.0f * e2 + e4;
                if (wa == 0) {
                    const float p = e4 - 4.0f * e2, q = e3 - 4.0f * e1;
                    wr[0][j] = st;
                    wr[1][j] = p + q;
                    wr[2][j] = p - q;
                } else {
                    const float p = e3 - e1, q = 2.0f * (e2 - e0);
                    wr[0][j] = p + q;
                    wr[1][j] = p - q;
                    wr[2][j] = st;
                }
            }
            float vv[18];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const float w0 = wr[r][0], w1 = wr[r][1], w2 = wr[r][2], w3 = wr[r][3], w4 = wr[r][4], w5 = wr[r][5];
                if (FLAGS & 16) {                         // columns scaled the same way
                    const float p = C6 * w4 - 4.0f * C6 * w2, q = C6 * w3 - 4.0f * C6 * w1, p2 = C24 * (w4 - w2), q2 = 2.0f * C24 * (w3 - w1);
                    vv[r * 6 + 0] = w0 - 1.25f * w2 + 0.25f * w4;
                    vv[r * 6 + 1] = p + q;
                    vv[r * 6 + 2] = p - q;
                    vv[r * 6 + 3] = p2 + q2;
                    vv[r * 6 + 4] = p2 - q2;
                    vv[r * 6 + 5] = 4.0f * w1 - 5.0f * w3 + w5;
                    continue;
                }
                const float p = w4 - 4.0f * w2, q = w3 - 4.0f * w1, p2 = w4 - w2, q2 = 2.0f * (w3 - w1);
                vv[r * 6 + 0] = 4.0f * w0 - 5.0f * w2 + w4;
                vv[r * 6 + 1] = p + q;
                vv[r * 6 + 2] = p - q;
                vv[r * 6 + 3] = p2 + q2;
                vv[r * 6 + 4] = p2 - q2;
                vv[r * 6 + 5] = 4.0f * w1 - 5.0f * w3 + w5;
            }
            if (FLAGS & 8) {
                // V[group][p / 4][lane][p % 4]: this thread owns positions 18 wa .. 18 wa + 17 = four whole quads and half of quad 4
                float* vq = v_buf + (c & 1) * 2 * V_G2 + gl * V_G2 + lane * 4;
                if (wa == 0) {
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) *(f4*)(vq + q4 * 256) = (f4){ vv[4 * q4], vv[4 * q4 + 1], vv[4 * q4 + 2], vv[4 * q4 + 3] };
                    *(f2*)(vq + 4 * 256) = (f2){ vv[16], vv[17] };
                } else {
                    *(f2*)(vq + 4 * 256 + 2) = (f2){ vv[0], vv[1] };
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) *(f4*)(vq + (5 + q4) * 256) = (f4){ vv[2 + 4 * q4], vv[3 + 4 * q4], vv[4 + 4 * q4], vv[5 + 4 * q4] };
                }
            } else {
#pragma unroll
                for (int q = 0; q < 18; ++q) vd[q * 64] = vv[q];
            }
        };
        // ---- A operand: the 3x3 filters of K step s for this lane's (out channel, in channel), two register sets ----
        f4 gq0[2], gq1[2];
        float gq2[2];
        f4 ring[(FLAGS & 8192) ? 9 : (FLAGS & 4096) ? 6 : 1];   // FLAGS 4096: T'[6 rows][3] (+pad) / 8192: U'[36] of the current K step
        auto ring_load = [&](int s, int i) {
            // T18: T4[s][kb][row][lane][4] (6 KB per wave and step); U36: U4[quad][s][kb][lane][4] (9 KB per wave and step)
            const int off = (FLAGS & 8192) ? (i * 16 + s) * 4096 + kb * 1024 : (s * 4 + kb) * 6144 + i * 1024;
            ring[((FLAGS & 8192) || (FLAGS & 4096)) ? i : 0] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rw, wlane16, off, 0));
        };
        auto wload = [&](int s) {
            if (FLAGS & (4096 | 8192)) {
                if (s == 0) {
#pragma unroll
                    for (int i = 0; i < ((FLAGS & 8192) ? 9 : 6); ++i) ring_load(0, i);
                }
                return;
            }
            gq0[s & 1] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rw, wlane16, wbase + s * 8192, 0));
            gq1[s & 1] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rw, wlane16, wbase + s * 8192 + 1024, 0));
            gq2[s & 1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, wlane4, (wbase >> 3) + 16 * 8192 + s * 1024, 0));
        };
        f4 bq[9];                                         // FLAGS & 8: the step's B operands, one 16-byte LDS read per four positions
        auto mma1 = [&](const float* vg, int p, float av) {
            const float b = (FLAGS & 8) ? bq[p >> 2][p & 3] : vg[p * 64 + lane];
            acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b, acc[p], 0, 0, 0);
        };
        // one K step = one group of 4 input channels: 36 MFMAs; `chunk` >= 0: request that activation chunk behind the filters
        auto kstep = [&](const float* vg, int s, int chunk) {
            __builtin_amdgcn_sched_barrier(0);
            if (FLAGS & 8192) {                           // U' streamed from L2 (no filter arithmetic): a quad is reloaded behind its MFMAs
                if (chunk >= 0) gload_to(chunk, pre);
                if (FLAGS & 8) {
#pragma unroll
                    for (int q4 = 0; q4 < 9; ++q4) bq[q4] = *(const f4*)(vg + q4 * 256 + lane * 4);
                }
#pragma unroll
                for (int q4 = 0; q4 < 9; ++q4) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) mma1(vg, q4 * 4 + e, ring[(FLAGS & 8192) ? q4 : 0][e]);
                    if (s + 1 < 16) ring_load(s + 1, q4);
                }
                __builtin_amdgcn_sched_barrier(0);
                return;
            }
            if (FLAGS & 4096) {                           // T' = G' g streamed (18 floats): 6 operations per row of U', one row ahead
                if (chunk >= 0) gload_to(chunk, pre);
                if (FLAGS & 8) {
#pragma unroll
                    for (int q4 = 0; q4 < 9; ++q4) bq[q4] = *(const f4*)(vg + q4 * 256 + lane * 4);
                }
                auto f6 = [](float x0, float x1, float x2, float (&yv)[6]) {
                    const float t = x0 + x2, p = x0 + 4.0f * x2;
                    yv[0] = x0;
                    yv[1] = t + x1;
                    yv[2] = t - x1;
                    yv[3] = p + 2.0f * x1;
                    yv[4] = p - 2.0f * x1;
                    yv[5] = x2;
                };
                float ua[6], ub[6];
                f6(ring[(FLAGS & 4096) ? 0 : 0][0], ring[0][1], ring[0][2], ua);
                if (s + 1 < 16) ring_load(s + 1, 0);
#pragma unroll
                for (int a = 0; a < 6; ++a) {
                    float (&cur)[6] = (a & 1) ? ub : ua;
                    float (&nxt)[6] = (a & 1) ? ua : ub;
                    if (a + 1 < 6) {
                        const int r1 = (FLAGS & 4096) ? a + 1 : 0;
                        f6(ring[r1][0], ring[r1][1], ring[r1][2], nxt);
                        if (s + 1 < 16) ring_load(s + 1, a + 1);
                    }
#pragma unroll
                    for (int b = 0; b < 6; ++b) mma1(vg, a * 6 + b, cur[b]);
                }
                __builtin_amdgcn_sched_barrier(0);
                return;
            }
            if (s + 1 < 16 && !(FLAGS & 128)) wload(s + 1);
            if (chunk >= 0 && !(FLAGS & 1024)) gload_to(chunk, pre);
            __builtin_amdgcn_sched_barrier(0);            // the loads stay here, a whole K step ahead of their first use
            if (FLAGS & 8) {
#pragma unroll
                for (int q4 = 0; q4 < 9; ++q4) bq[q4] = *(const f4*)(vg + q4 * 256 + lane * 4);
            }
            const f4 gn0 = gq0[s & 1], gn1 = gq1[s & 1];
            const float gn2 = gq2[s & 1];
            const float g[3][3] = { { gn0[0], gn0[1], gn0[2] }, { gn0[3], gn1[0], gn1[1] }, { gn1[2], gn1[3], gn2 } };
            float uu[6];
            if (FLAGS & 16) {
                // U' = G' g G'^T, G' = [[1,0,0],[1,1,1],[1,-1,1],[1,2,4],[1,-2,4],[0,0,1]] (the row scales live in V): 6 operations
                // per 6-vector instead of 11 -> 18 + 36 = 54 per K step
                auto f6 = [](float x0, float x1, float x2, float (&yv)[6]) {
                    const float t = x0 + x2, p = x0 + 4.0f * x2;
                    yv[0] = x0;
                    yv[1] = t + x1;
                    yv[2] = t - x1;
                    yv[3] = p + 2.0f * x1;
                    yv[4] = p - 2.0f * x1;
                    yv[5] = x2;
                };
                if (FLAGS & 2048) {
                    // software pipeline: the six U values of row r+1 are computed BEFORE the six MFMAs of row r are issued, so no
                    // MFMA reads a register a VALU instruction has just written (tools/mfma_valu_probe: +10..15 cycles per MFMA)
                    float ua[6], ub[6], t1[3], t2[3], t3[3], t4[3];
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const float t = g[0][j] + g[2][j], p = g[0][j] + 4.0f * g[2][j];
                        t1[j] = t + g[1][j];
                        t2[j] = t - g[1][j];
                        t3[j] = p + 2.0f * g[1][j];
                        t4[j] = p - 2.0f * g[1][j];
                    }
                    f6(g[0][0], g[0][1], g[0][2], ua);
                    __builtin_amdgcn_sched_barrier(0);
                    f6(t1[0], t1[1], t1[2], ub);
#pragma unroll
                    for (int b = 0; b < 6; ++b) mma1(vg, b, ua[b]);
                    __builtin_amdgcn_sched_barrier(0);
                    f6(t2[0], t2[1], t2[2], ua);
#pragma unroll
                    for (int b = 0; b < 6; ++b) mma1(vg, 6 + b, ub[b]);
                    __builtin_amdgcn_sched_barrier(0);
                    f6(t3[0], t3[1], t3[2], ub);
#pragma unroll
                    for (int b = 0; b < 6; ++b) mma1(vg, 12 + b, ua[b]);
                    __builtin_amdgcn_sched_barrier(0);
                    f6(t4[0], t4[1], t4[2], ua);
#pragma unroll
                    for (int b = 0; b < 6; ++b) mma1(vg, 18 + b, ub[b]);
                    __builtin_amdgcn_sched_barrier(0);
                    f6(g[2][0], g[2][1], g[2][2], ub);
#pragma unroll
                    for (int b = 0; b < 6; ++b) mma1(vg, 24 + b, ua[b]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int b = 0; b < 6; ++b) mma1(vg, 30 + b, ub[b]);
                    __builtin_amdgcn_sched_barrier(0);
                    return;
                }
                f6(g[0][0], g[0][1], g[0][2], uu);                               // row 0 of T' = g[0][.]
#pragma unroll
                for (int b = 0; b < 6; ++b) mma1(vg, b, uu[b]);
                float ta[3], tb[3];
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const float t = g[0][j] + g[2][j];
                    ta[j] = t + g[1][j];
                    tb[j] = t - g[1][j];
                }
                f6(ta[0], ta[1], ta[2], uu);
#pragma unroll
                for (int b = 0; b < 6; ++b) mma1(vg, 6 + b, uu[b]);
                f6(tb[0], tb[1], tb[2], uu);
#pragma unroll
                for (int b = 0; b < 6; ++b) mma1(vg, 12 + b, uu[b]);
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const float p = g[0][j] + 4.0f * g[2][j];
                    ta[j] = p + 2.0f * g[1][j];
                    tb[j] = p - 2.0f * g[1][j];
                }
                f6(ta[0], ta[1], ta[2], uu);
#pragma unroll
                for (int b = 0; b < 6; ++b) mma1(vg, 18 + b, uu[b]);
                f6(tb[0], tb[1], tb[2], uu);
#pragma unroll
                for (int b = 0; b < 6; ++b) mma1(vg, 24 + b, uu[b]);
                f6(g[2][0], g[2][1], g[2][2], uu);                               // row 5 of T' = g[2][.]
#pragma unroll
                for (int b = 0; b < 6; ++b) mma1(vg, 30 + b, uu[b]);
                __builtin_amdgcn_sched_barrier(0);
                return;
            }
            if (FLAGS & 64) {                             // ablation: no filter transform, 36 MFMAs on raw taps
#pragma unroll
                for (int p = 0; p < 36; ++p) mma1(vg, p, g[(p / 3) % 3][p % 3]);
                __builtin_amdgcn_sched_barrier(0);
                return;
            }
            filt6(g[0][0] * 0.25f, g[0][1] * 0.25f, g[0][2] * 0.25f, uu);      // row 0 of T = G g is g[0][.] / 4
#pragma unroll
            for (int b = 0; b < 6; ++b) mma1(vg, b, uu[b]);
            {   // rows 1, 2
                float t1[3], t2[3];
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const float t = g[0][j] + g[2][j];
                    t1[j] = (t + g[1][j]) * (-1.0f / 6.0f);
                    t2[j] = (t - g[1][j]) * (-1.0f / 6.0f);
                }
                filt6(t1[0], t1[1], t1[2], uu);
#pragma unroll
                for (int b = 0; b < 6; ++b) mma1(vg, 6 + b, uu[b]);
                filt6(t2[0], t2[1], t2[2], uu);
#pragma unroll
                for (int b = 0; b < 6; ++b) mma1(vg, 12 + b, uu[b]);
            }
            {   // rows 3, 4
                float t3[3], t4[3];
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const float p = g[0][j] * (1.0f / 24.0f) + g[2][j] * (1.0f / 6.0f), q = g[1][j] * (1.0f / 12.0f);
                    t3[j] = p + q;
                    t4[j] = p - q;
                }
                filt6(t3[0], t3[1], t3[2], uu);
#pragma unroll
                for (int b = 0; b < 6; ++b) mma1(vg, 18 + b, uu[b]);
                filt6(t4[0], t4[1], t4[2], uu);
#pragma unroll
                for (int b = 0; b < 6; ++b) mma1(vg, 24 + b, uu[b]);
            }
            filt6(g[2][0], g[2][1], g[2][2], uu);                               // row 5 of T is g[2][.]
#pragma unroll
            for (int b = 0; b < 6; ++b) mma1(vg, 30 + b, uu[b]);
            if (FLAGS & 4) {
#pragma unroll
                for (int i = 0; i < 36; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // 1 MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);    // 3 VALU
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);    // 1 LDS read
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        auto phase = [&](int c) {
            const float* vs = v_buf + (c & 1) * 2 * V_G2;
            kstep(vs, 2 * c, -1);
            if (c + 2 < 8) lstore_from(in_buf + (c & 1) * IN_BUF2, pre);      // chunk c+2 -> in_buf[c & 1] (V(c) was built in phase c-1)
            kstep(vs + V_G2, 2 * c + 1, c + 3 < 8 ? c + 3 : -1);
            if (FLAGS & 1) __builtin_amdgcn_sched_barrier(0x2);                  // (experiment) VALU may cross
            if (c + 1 < 8 && !(FLAGS & 256)) produce(c + 1);
            __syncthreads();
        };

        // Output rows: per-lane offset + an IMMEDIATE row offset, scalar offset 0.  With a scalar-register offset the compiler
        // assumes a 16-byte buffer store needs no wait state before its data registers are overwritten (LLVM createsVALUHazard:
        // "hazard only exists if the instruction is not using a register in the soffset field") and schedules a v_pk_mov into
        // them right behind the store; on gfx950 that corrupted dword 1 of lanes 12-15 of every 16 (found with tools/wino_lab).
        const int ovoff = ooff + n0 * 16384 + kb * 4096;
        f4 rres[4][4];
        auto rload = [&](int r) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                rres[r][i] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rr, ovoff + (r * 1024 + i * 256), 0, 2));
        };

        unsigned long long* const stamps = (FLAGS & 32) ? (unsigned long long*)ta.maps_out + (size_t)grp * 16 : nullptr;
        auto stamp = [&](int slot) {
            if ((FLAGS & 32) && tid == 0) stamps[slot] = __builtin_amdgcn_s_memtime();
        };
        stamp(0);
        {   // the first two chunks are requested together: one HBM round trip before the first V can be built, not two
            f4 first[2];
            gload_to(0, first);
            gload_to(1, pre);
            wload(0);
            if (FLAGS & 128) { gq0[1] = gq0[0]; gq1[1] = gq1[0]; gq2[1] = gq2[0]; }
            __syncthreads();                               // zero fill done / the previous group's LDS reads are over
            lstore_from(in_buf, first);
            lstore_from(in_buf + IN_BUF2, pre);
        }
        gload_to(2, pre);
        __syncthreads();
        produce(0);
        __syncthreads();
        stamp(1);
        for (int c = 0; c < 8; ++c) {
            phase(c);
            stamp(2 + c);
        }

        if (FLAGS & 512) {                                // ablation: no output stage
            float sum = 0.0f;
#pragma unroll
            for (int q = 0; q < 36; ++q) sum += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
            if (sum == 123.456f) y[tid] = sum;
            return;
        }
        // ---- inverse transform in registers + epilogue ----
        rload(0);
        rload(1);
        constexpr int OC = 3;                              // TAIL: 2 policy + 1 value head channels
        float hp[TAIL ? OC : 1][4][4];                     // TAIL: this lane's share of the 1x1 head convolutions (its 4 channels)
        if (TAIL) {
#pragma unroll
            for (int o = 0; o < OC; ++o)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) hp[TAIL ? o : 0][i][j] = 0.0f;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            __builtin_amdgcn_sched_barrier(0);
            float m[6][6];
#pragma unroll
            for (int p = 0; p < 36; ++p) m[p / 6][p % 6] = acc[p][r];
            float o[4][4];
            inverse_transform(m, o);
            const int k = 16 * kb + 4 * c_sub + r;
            const float sc = scale[k], sh = shift[k];
            float hwk[OC];
            if (TAIL) {
#pragma unroll
                for (int oc = 0; oc < OC; ++oc) hwk[oc] = ta.hw[oc * 64 + k];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f4 v;
                const f4 rv = rres[r][i];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[j] = o[i][j] * sc + sh + rv[j];
                    if (relu) v[j] = v[j] > 0.0f ? v[j] : 0.0f;
                    if (4 * ty + i >= H || 4 * tx + j >= W) v[j] = 0.0f;       // cells off the board stay zero
                }
                if (TAIL) {
#pragma unroll
                    for (int oc = 0; oc < OC; ++oc)
#pragma unroll
                        for (int j = 0; j < 4; ++j) hp[TAIL ? oc : 0][i][j] += hwk[oc] * v[j];
                } else {
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), ry, ovoff + (r * 1024 + i * 256), 0, 2);
                }
            }
            if (r + 2 < 4) rload(r + 2);
        }
        stamp(10);
        if (!TAIL) return;

        // ---- head convolutions fused behind the LAST trunk convolution: only the ReLU'd head maps are written ----
        // Every lane holds the head-convolution partial sums of its 4 channels for its 16 cells; the 16 partials of a cell
        // (4 waves x 4 lane groups) are summed through LDS in a fixed order.  The LDS images are dead by now.
        constexpr int PROW = OC * 256 + 16;                // partial row stride: 32 lanes of a bank group -> 32 banks
        float* const part = lds;                           // [kb * 4 + c_sub][o][(i * 4 + j) * 16 + tl]
        float* const maps = lds + 16 * PROW;               // [board][o * HW + row * W + col]
        constexpr int HW = H * W;
#pragma unroll
        for (int oc = 0; oc < OC; ++oc)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) part[(kb * 4 + c_sub) * PROW + oc * 256 + (i * 4 + j) * 16 + tl] = hp[TAIL ? oc : 0][i][j];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < OC; ++q) {
            const int idx = tid + NTHR2 * q;               // (o, cell) pairs: 768 per workgroup
            const int oc = idx >> 8, cp = idx & 255;
            float sum = 0.0f;
#pragma unroll
            for (int rw16 = 0; rw16 < 16; ++rw16) sum += part[rw16 * PROW + oc * 256 + cp];
            const int ij = cp >> 4, t16 = cp & 15;
            const int b = t16 >> 2, tl4 = t16 & 3;
            const int row = 4 * (tl4 >> 1) + (ij >> 2), col = 4 * (tl4 & 1) + (ij & 3);
            sum += ta.hb[oc];
            if (row < H && col < W) maps[b * (OC * HW) + oc * HW + row * W + col] = sum > 0.0f ? sum : 0.0f;
        }
        __syncthreads();
        for (int i = tid; i < NIMG2 * OC * HW; i += NTHR2) {
            const int b = i / (OC * HW);
            if (n0 + b < batch) ta.maps_out[(size_t)n0 * (OC * HW) + i] = maps[i];
        }
        if (PERSIST) {                                     // the next group needs its zero borders back
            __syncthreads();
            for (int i = tid; i < 2 * IN_BUF2; i += NTHR2) lds[i] = 0.0f;
        }
    };
    if (PERSIST) {
        for (int grp = (int)blockIdx.x; grp < ngroups; grp += (int)gridDim.x) group(grp);
    } else {
        group((int)blockIdx.x);
    }
}

template <int FLAGS>
static int sprl_wino_conv64_v4_launch(const float* x, const float* wts, const float* scale, const float* shift, const float* res,
                                      float* y, int batch, int relu, const unsigned* batch_dev, void* stream, float* stamps = nullptr) {
    const int ngroups = (batch + NIMG2 - 1) / NIMG2;
    const int grid = (FLAGS & 2) ? (ngroups < 512 ? ngroups : 512) : ngroups;
    TailArgs ta{};
    ta.maps_out = stamps;
    hipLaunchKernelGGL((wino_conv64_v4_kernel<8, 8, FLAGS>), dim3((unsigned)grid), dim3(NTHR2), 0, (hipStream_t)stream, x, wts,
                       scale, shift, res, y, batch, relu, batch_dev, ta);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ---------------------------------------------------------------------------------------------------
// General boards (any H x W, e.g. Go 9x9 and 19x19), plain NCHW activations in and out: same Winograd / MFMA core as
// version 2, but a workgroup takes 16 consecutive TILES of the batch (boards are ceil(H/4) x ceil(W/4) tiles, a tile's
// 6x6 input patch reaches into its neighbours), so the loader gathers the 36 patch values of every (channel, tile) straight
// from global memory into LDS patches [slot][tile][37] (slot stride 592 = 16 mod 32: a 32-lane bank group reads 32 banks).
// Used by the library-convolution path for its trunk convolutions (torch_eval.cpp) - the stem and the heads stay as they are.
// ---------------------------------------------------------------------------------------------------
constexpr int PS_G = 37, SS_G = 16 * PS_G;            // patch stride, channel-slot stride
constexpr int IN_BUF_G = 8 * SS_G;
constexpr int LDS_FLOATS_G = 2 * IN_BUF_G + 4 * V_G2;  // 74.8 KB: two workgroups per CU
constexpr int NLD_G = (8 * 16 * 36) / NTHR2;          // 18 patch values per thread and chunk

__global__ void __launch_bounds__(NTHR2, 2) wino_conv64_nchw_kernel(const float* __restrict__ x, const float* __restrict__ u,
                                                                    const float* __restrict__ scale, const float* __restrict__ shift,
                                                                    const float* __restrict__ res, float* __restrict__ y, int batch,
                                                                    int H, int W, int relu) {
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS_G];
    float* const in_buf = lds;
    float* const v_buf = lds + 2 * IN_BUF_G;
    const int tid = (int)threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c_sub = lane >> 4, tl = lane & 15;
    const int gl = wave & 1, wa = wave >> 1;
    const int kb = wave;
    const int TX = (W + 3) >> 2, TY = (H + 3) >> 2, TPB = TX * TY;
    const int HW = H * W;
    // this thread's tile (the same one as loader, tid & 15, and as MFMA column / output lane, lane & 15)
    const int total_tiles = batch * TPB;              // the host keeps batch * tiles-per-board below 2^31
    const int tg = (int)blockIdx.x * 16 + tl;
    const int t_n = tg < total_tiles ? tg / TPB : -1;
    const int t_tt = tg < total_tiles ? tg % TPB : 0;
    const int t_row0 = 4 * (t_tt / TX), t_col0 = 4 * (t_tt % TX);

    f4 acc[36];
#pragma unroll
    for (int q = 0; q < 36; ++q) acc[q] = (f4){ 0.0f, 0.0f, 0.0f, 0.0f };

    // chunk = groups 2c, 2c+1: 8 channel slots x 16 tiles x 36 patch values.  Loader thread = (tile, slot, half): the
    // three patch rows 3 half .. 3 half + 2 of one (channel slot, tile), 18 values
    const int ld_tile = tl, ld_slot = (tid >> 4) & 7, ld_half = tid >> 7;
    const int ld_n = t_n;
    const int ld_row = t_row0 - 1 + 3 * ld_half, ld_col = t_col0 - 1;
    const int ld_lds = ld_slot * SS_G + ld_tile * PS_G + 18 * ld_half;
    bool colok[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) colok[j] = ld_n >= 0 && ld_col + j >= 0 && ld_col + j < W;
    float pre[NLD_G];
    auto gload = [&](int chunk) {
        const int g = 2 * chunk + (ld_slot >> 2);
        const int k = 16 * (g >> 2) + 4 * (ld_slot & 3) + (g & 3);
        const float* src = x + ((size_t)(ld_n < 0 ? 0 : ld_n) * 64 + (size_t)k) * HW + ld_col;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int row = ld_row + i;
            const bool rowok = row >= 0 && row < H;
#pragma unroll
            for (int j = 0; j < 6; ++j) pre[i * 6 + j] = (rowok && colok[j]) ? src[row * W + j] : 0.0f;
        }
    };
    auto lstore = [&](float* buf) {
#pragma unroll
        for (int q = 0; q < NLD_G; ++q) buf[ld_lds + q] = pre[q];
    };
    const int patch0 = (gl * 4 + c_sub) * SS_G + tl * PS_G + wa * 6;
    const int vdst0 = gl * V_G2 + (3 * wa) * 6 * 64 + lane;
    auto produce = [&](int c) {
        const float* pp = in_buf + (c & 1) * IN_BUF_G + patch0;
        float* vd = v_buf + (c & 1) * 2 * V_G2 + vdst0;
        float wr[3][6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const float e0 = pp[j], e1 = pp[6 + j], e2 = pp[12 + j], e3 = pp[18 + j], e4 = pp[24 + j];
            const float st = 4.0f * e0 - 5.0f * e2 + e4;
            if (wa == 0) {
                const float p = e4 - 4.0f * e2, q = e3 - 4.0f * e1;
                wr[0][j] = st;
                wr[1][j] = p + q;
                wr[2][j] = p - q;
            } else {
                const float p = e3 - e1, q = 2.0f * (e2 - e0);
                wr[0][j] = p + q;
                wr[1][j] = p - q;
                wr[2][j] = st;
            }
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float w0 = wr[r][0], w1 = wr[r][1], w2 = wr[r][2], w3 = wr[r][3], w4 = wr[r][4], w5 = wr[r][5];
            const float p = w4 - 4.0f * w2, q = w3 - 4.0f * w1, p2 = w4 - w2, q2 = 2.0f * (w3 - w1);
            vd[(r * 6 + 0) * 64] = 4.0f * w0 - 5.0f * w2 + w4;
            vd[(r * 6 + 1) * 64] = p + q;
            vd[(r * 6 + 2) * 64] = p - q;
            vd[(r * 6 + 3) * 64] = p2 + q2;
            vd[(r * 6 + 4) * 64] = p2 - q2;
            vd[(r * 6 + 5) * 64] = 4.0f * w1 - 5.0f * w3 + w5;
        }
    };
    const f4* ua = (const f4*)u + kb * 64 + lane;
    f4 a[9];
    auto aload = [&](int s, int k) { a[k] = ua[(size_t)k * (16 * 4 * 64) + s * 256]; };
    auto mma = [&](const float* vg, int k) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float b = vg[(k * 4 + q) * 64 + lane];
            acc[k * 4 + q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k][q], b, acc[k * 4 + q], 0, 0, 0);
        }
    };
    auto phase = [&](int c) {
        const float* vs = v_buf + (c & 1) * 2 * V_G2;
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2) {
            const int s = 2 * c + g2;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                mma(vs + g2 * V_G2, k);
                if (s + 1 < 16) aload(s + 1, k);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (c + 1 < 8) produce(c + 1);
        if (c + 2 < 8) {
            lstore(in_buf + (c & 1) * IN_BUF_G);
            if (c + 3 < 8) gload(c + 3);
        }
        __syncthreads();
    };

    gload(0);
    lstore(in_buf);
    gload(1);
#pragma unroll
    for (int k = 0; k < 9; ++k) aload(0, k);
    lstore(in_buf + IN_BUF_G);
    gload(2);
    __syncthreads();
    produce(0);
    __syncthreads();
    for (int c = 0; c < 8; ++c) phase(c);

    // ---- inverse transform in registers + epilogue, NCHW ----
    const int n = t_n, row0 = t_row0, col0 = t_col0;
    // residual values of output component r (channel 16 kb + 4 c_sub + r), requested one component ahead of their use
    float rres[2][16];
    auto rload = [&](int r, float (&dst)[16]) {
        const size_t plane = ((size_t)(n < 0 ? 0 : n) * 64 + (size_t)(16 * kb + 4 * c_sub + r)) * HW;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = row0 + i, col = col0 + j;
                dst[i * 4 + j] = (res && n >= 0 && row < H && col < W) ? res[plane + (size_t)row * W + col] : 0.0f;
            }
    };
    rload(0, rres[0]);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        __builtin_amdgcn_sched_barrier(0);
        if (r + 1 < 4) rload(r + 1, rres[(r + 1) & 1]);
        float m[6][6];
#pragma unroll
        for (int p = 0; p < 36; ++p) m[p / 6][p % 6] = acc[p][r];
        float o[4][4];
        inverse_transform(m, o);
        const int k = 16 * kb + 4 * c_sub + r;
        const float sc = scale[k], sh = shift[k];
        if (n >= 0) {
            const size_t plane = ((size_t)n * 64 + (size_t)k) * HW;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = row0 + i, col = col0 + j;
                    if (row < H && col < W) {
                        float v = o[i][j] * sc + sh + rres[r & 1][i * 4 + j];
                        if (relu) v = v > 0.0f ? v : 0.0f;
                        y[plane + (size_t)row * W + col] = v;
                    }
                }
        }
    }
}

}  // namespace

// sprl_wino_weight_layout(): 2 = U4[p / 4][s][kb][lane][p % 4] (what wino_transform must produce; 1 was U2[p][s][kb][lane])
extern "C" int sprl_wino_weight_layout(void) { return 2; }

// x, y, res: activations in layout W (4096 floats per board; res may be null; y must not alias x); u: 36*64*64 pre-transformed
// weights in A-operand order (torch_eval.cpp: wino_transform); scale/shift: [64].  Returns 0, or -1 when the board shape has
// no kernel here.
extern "C" int sprl_wino_conv64_dev(const float* x, const float* u, const float* scale, const float* shift, const float* res,
                                    float* y, int batch, int H, int W, int relu, const unsigned* batch_dev, void* stream);
extern "C" int sprl_wino_conv64(const float* x, const float* u, const float* scale, const float* shift, const float* res,
                                float* y, int batch, int H, int W, int relu, void* stream) {
    return sprl_wino_conv64_dev(x, u, scale, shift, res, y, batch, H, W, relu, nullptr, stream);
}

// batch_dev: optional device pointer to the real board count (<= batch, the capacity); version 2 kernel only
extern "C" int sprl_wino_conv64_dev(const float* x, const float* u, const float* scale, const float* shift, const float* res,
                                    float* y, int batch, int H, int W, int relu, const unsigned* batch_dev, void* stream) {
    if (batch <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    static const int version = getenv("SPRL_WINO_V3") ? 3 : 2;
    static const int stagger = getenv("SPRL_WINO_STAGGER") ? atoi(getenv("SPRL_WINO_STAGGER")) : 0;
    if (version == 2) {
        const dim3 grid((unsigned)((batch + NIMG2 - 1) / NIMG2)), block(NTHR2);
        if (H == 8 && W == 8) hipLaunchKernelGGL((wino_conv64_v2_kernel<8, 8>), grid, block, 0, st, x, u, scale, shift, res, y, batch, relu, stagger, batch_dev, TailArgs{});
        else if (H == 6 && W == 7) hipLaunchKernelGGL((wino_conv64_v2_kernel<6, 7>), grid, block, 0, st, x, u, scale, shift, res, y, batch, relu, stagger, batch_dev, TailArgs{});
        else if (H == 7 && W == 7) hipLaunchKernelGGL((wino_conv64_v2_kernel<7, 7>), grid, block, 0, st, x, u, scale, shift, res, y, batch, relu, stagger, batch_dev, TailArgs{});
        else return -1;
        return hipGetLastError() == hipSuccess ? 0 : -2;
    }
    const dim3 grid((unsigned)((batch + NIMG - 1) / NIMG)), block(NTHR);
    if (H == 8 && W == 8) hipLaunchKernelGGL((wino_conv64_kernel<8, 8>), grid, block, 0, st, x, u, scale, shift, res, y, batch, relu, batch_dev);
    else if (H == 6 && W == 7) hipLaunchKernelGGL((wino_conv64_kernel<6, 7>), grid, block, 0, st, x, u, scale, shift, res, y, batch, relu, batch_dev);
    else if (H == 7 && W == 7) hipLaunchKernelGGL((wino_conv64_kernel<7, 7>), grid, block, 0, st, x, u, scale, shift, res, y, batch, relu, batch_dev);
    else return -1;
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// The last trunk convolution with the whole network tail fused behind it (2 policy + 1 value head channels): the trunk
// output is never written; logits [batch][A] and value [batch] go straight to the caller's buffers.  hw/hb: [3][64] / [3]
// head convolutions (policy rows first), pfc_w: [2*H*W][A], vfc1_w: [H*W][HID], vfc2_w: [HID].  -1: shape not covered.
extern "C" int sprl_wino_conv64_tail(const float* x, const float* u, const float* scale, const float* shift, const float* res,
                                     int batch, int H, int W, const unsigned* batch_dev, const float* hw, const float* hb,
                                     const float* pfc_w, const float* pfc_b, const float* vfc1_w, const float* vfc1_b,
                                     const float* vfc2_w, const float* vfc2_b, float* logits, float* value, int A, int HID,
                                     void* stream) {
    if (batch <= 0) return 0;
    if (HID > 64 || A < 1) return -1;
    static const int stagger = getenv("SPRL_WINO_STAGGER") ? atoi(getenv("SPRL_WINO_STAGGER")) : 0;
    const TailArgs ta{ hw, hb, pfc_w, pfc_b, vfc1_w, vfc1_b, vfc2_w, vfc2_b, logits, value, A, HID, nullptr };
    const dim3 grid((unsigned)((batch + NIMG2 - 1) / NIMG2)), block(NTHR2);
    hipStream_t st = (hipStream_t)stream;
    if (H == 8 && W == 8) hipLaunchKernelGGL((wino_conv64_v2_kernel<8, 8, 0, 1>), grid, block, 0, st, x, u, scale, shift, res, nullptr, batch, 1, stagger, batch_dev, ta);
    else if (H == 6 && W == 7) hipLaunchKernelGGL((wino_conv64_v2_kernel<6, 7, 0, 1>), grid, block, 0, st, x, u, scale, shift, res, nullptr, batch, 1, stagger, batch_dev, ta);
    else if (H == 7 && W == 7) hipLaunchKernelGGL((wino_conv64_v2_kernel<7, 7, 0, 1>), grid, block, 0, st, x, u, scale, shift, res, nullptr, batch, 1, stagger, batch_dev, ta);
    else return -1;
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// Same, but only the ReLU'd head maps [batch][3 * H * W] (policy maps first) are produced; the FC layers follow in
// sprl_tail_fc (cnn_epilogue.hip), which keeps their weights in LDS for 16 boards at a time.
extern "C" int sprl_wino_conv64_heads(const float* x, const float* u, const float* scale, const float* shift, const float* res,
                                      int batch, int H, int W, const unsigned* batch_dev, const float* hw, const float* hb,
                                      float* maps_out, void* stream) {
    if (batch <= 0) return 0;
    static const int stagger = getenv("SPRL_WINO_STAGGER") ? atoi(getenv("SPRL_WINO_STAGGER")) : 0;
    TailArgs ta{};
    ta.hw = hw;
    ta.hb = hb;
    ta.maps_out = maps_out;
    const dim3 grid((unsigned)((batch + NIMG2 - 1) / NIMG2)), block(NTHR2);
    hipStream_t st = (hipStream_t)stream;
    if (H == 8 && W == 8) hipLaunchKernelGGL((wino_conv64_v2_kernel<8, 8, 0, 2>), grid, block, 0, st, x, u, scale, shift, res, nullptr, batch, 1, stagger, batch_dev, ta);
    else if (H == 6 && W == 7) hipLaunchKernelGGL((wino_conv64_v2_kernel<6, 7, 0, 2>), grid, block, 0, st, x, u, scale, shift, res, nullptr, batch, 1, stagger, batch_dev, ta);
    else if (H == 7 && W == 7) hipLaunchKernelGGL((wino_conv64_v2_kernel<7, 7, 0, 2>), grid, block, 0, st, x, u, scale, shift, res, nullptr, batch, 1, stagger, batch_dev, ta);
    else return -1;
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// Any board size, NCHW activations [batch][64][H][W] in and out (res may be null; y must not alias x); same weights `u`.
extern "C" int sprl_wino_conv64_nchw(const float* x, const float* u, const float* scale, const float* shift, const float* res,
                                     float* y, int batch, int H, int W, int relu, void* stream) {
    if (batch <= 0) return 0;
    if (H < 1 || W < 1 || H > 64 || W > 64) return -1;
    const long long tiles = (long long)batch * ((H + 3) / 4) * ((W + 3) / 4);
    if (tiles > 0x7fffffffLL - 16) return -1;
    const dim3 grid((unsigned)((tiles + 15) / 16)), block(NTHR2);
    hipLaunchKernelGGL(wino_conv64_nchw_kernel, grid, block, 0, (hipStream_t)stream, x, u, scale, shift, res, y, batch, H, W, relu);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
