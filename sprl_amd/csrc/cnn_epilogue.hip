// cnn_epilogue.hip — fused convolution epilogue for the policy/value CNN (gfx950).
//
// The reference network (src/networks/grid_networks.py:8-27,56-60) is conv3x3 -> BatchNorm -> ReLU (+ residual).
// The convolutions run in MIOpen (MFMA); everything between two convolutions is one pass of this kernel over the
// NCHW activation, in place:   y = max(0, x * scale[c] + shift[c] (+ residual)),
// with scale = gamma / sqrt(var + eps) and shift = (conv_bias - mean) * scale + beta folded on the host.
// HBM-bound: 16 B per lane per access, channel parameters broadcast from registers (H*W = 64 is a multiple of 4, so a
// float4 never straddles channels).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <mutex>

namespace {
template <bool RESIDUAL>
__global__ void __launch_bounds__(256) bn_relu_kernel(float4* __restrict__ x, const float4* __restrict__ res,
                                                      const float* __restrict__ scale, const float* __restrict__ shift,
                                                      int64_t n_vec, int vec_per_channel, int channels) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)((i / vec_per_channel) % channels);
        const float s = scale[c], t = shift[c];
        float4 v = x[i];
        v.x = v.x * s + t;
        v.y = v.y * s + t;
        v.z = v.z * s + t;
        v.w = v.w * s + t;
        if (RESIDUAL) {
            const float4 r = res[i];
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
        v.x = v.x > 0.0f ? v.x : 0.0f;
        v.y = v.y > 0.0f ? v.y : 0.0f;
        v.z = v.z > 0.0f ? v.z : 0.0f;
        v.w = v.w > 0.0f ? v.w : 0.0f;
        x[i] = v;
    }
}
// Both 1x1 head convolutions (policy: C -> PC channels, value: C -> VC channels; grid_networks.py:44,49) + bias + ReLU in
// ONE pass over the trunk output: thread = (sample, position), consecutive lanes read consecutive positions of one
// channel plane (coalesced), the OC = PC + VC weight rows sit in LDS.  Replaces two library convolutions, two bias
// adds and two ReLUs that would each stream the 64-channel activation again.
template <int OC>
__global__ void __launch_bounds__(256) heads_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                    const float* __restrict__ bias, float* __restrict__ out_p,
                                                    float* __restrict__ out_v, int64_t n_pos, int C, int HW, int PC, int Wd) {
    // Wd > 0: x is in layout W of cnn_wino.hip (board width Wd), else NCHW
    extern __shared__ float wsh[];                       // [OC][C] then [OC] biases
    for (int i = threadIdx.x; i < OC * C + OC; i += blockDim.x) wsh[i] = i < OC * C ? w[i] : bias[i - OC * C];
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_pos; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / HW;
        const int pos = (int)(i - b * HW);
        const int row = Wd > 0 ? pos / Wd : 0, col = Wd > 0 ? pos % Wd : 0;
        const float* xp = Wd > 0 ? x + b * 4096 + (row & 3) * 64 + ((row >> 2) * 2 + (col >> 2)) * 4 + (col & 3)
                                 : x + b * (int64_t)C * HW + pos;
        float acc[OC];
#pragma unroll
        for (int o = 0; o < OC; ++o) acc[o] = 0.0f;
        for (int c = 0; c < C; ++c) {
            const float v = Wd > 0 ? xp[(4 * (c >> 4) + (c & 3)) * 256 + ((c >> 2) & 3) * 16] : xp[(int64_t)c * HW];
#pragma unroll
            for (int o = 0; o < OC; ++o) acc[o] += wsh[o * C + c] * v;
        }
#pragma unroll
        for (int o = 0; o < OC; ++o) {
            float r = acc[o] + wsh[OC * C + o];
            r = r > 0.0f ? r : 0.0f;
            if (o < PC) out_p[(b * PC + o) * HW + pos] = r;
            else out_v[(b * (OC - PC) + (o - PC)) * HW + pos] = r;
        }
    }
}
// Stem convolution (P input planes -> 64 channels, 3x3, padding 1) + folded BatchNorm/bias + ReLU, written in layout W
// (cnn_wino.hip) for the trunk kernels.  K = 9 P is tiny (27 for Othello / Connect Four), so this is plain VALU work:
// thread = one board cell, the 3x3xP patch in registers, the weights wave-uniform (scalar loads), 64 channels per thread.
// Cells of the 8x8 frame that are off an H x W board are written as zeros (the trunk relies on it).
template <int P>
__global__ void __launch_bounds__(256) stem_kernel(const float* __restrict__ planes, const float* __restrict__ w,
                                                   const float* __restrict__ scale, const float* __restrict__ shift,
                                                   float* __restrict__ y, int batch, int H, int W,
                                                   const unsigned* __restrict__ batch_dev) {
    if (batch_dev) {
        const int real = (int)*batch_dev;
        batch = real < batch ? real : batch;
        if ((int)blockIdx.x * 4 >= batch) return;
    }
    __shared__ float img[4][P][100];                   // zero-bordered 10x10 images
    const int tid = (int)threadIdx.x;
    const int n0 = (int)blockIdx.x * 4;
    for (int i = tid; i < 4 * P * 100; i += 256) (&img[0][0][0])[i] = 0.0f;
    __syncthreads();
    const int HW = H * W;
    for (int e = tid; e < 4 * P * HW; e += 256) {
        const int b = e / (P * HW), r = e % (P * HW), p = r / HW, cell = r % HW;
        if (n0 + b < batch) img[b][p][(cell / W + 1) * 10 + cell % W + 1] = planes[(size_t)(n0 + b) * P * HW + r];
    }
    __syncthreads();
    const int b = tid >> 6, cell = tid & 63;
    const int i = cell >> 4, tile = (cell >> 2) & 3, j = cell & 3;
    const int row = 4 * (tile >> 1) + i, col = 4 * (tile & 1) + j;
    const int n = n0 + b;
    float d[P][9];
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
        for (int t = 0; t < 9; ++t) d[p][t] = img[b][p][(row + t / 3) * 10 + col + t % 3];
    const bool on_board = row < H && col < W;
    if (n >= batch) return;
    float* yo = y + (size_t)n * 4096 + i * 64 + tile * 4 + j;
    for (int g = 0; g < 16; ++g) {
#pragma unroll
        for (int cs = 0; cs < 4; ++cs) {
            const int k = 16 * (g >> 2) + 4 * cs + (g & 3);
            const float* wk = w + (size_t)k * P * 9;
            float acc = 0.0f;
#pragma unroll
            for (int p = 0; p < P; ++p)
#pragma unroll
                for (int t = 0; t < 9; ++t) acc += wk[p * 9 + t] * d[p][t];
            acc = acc * scale[k] + shift[k];
            yo[g * 256 + cs * 16] = on_board ? (acc > 0.0f ? acc : 0.0f) : 0.0f;
        }
    }
}
// Stem for 3 input planes on the fp32 matrix core: per board out[64 ch][64 cells] = W[64][27] x patches[27][64] as
// v_mfma_f32_16x16x4_f32 (K padded to 28 = 7 steps; exact fp32 multiply-adds like the VALU form).  One wave per board:
// the weight fragments (4 channel blocks x 7 steps) stay in 28 registers for the whole kernel, the patch fragments are
// single LDS reads from the zero-bordered planes, and with the cell order (row-in-tile, tile, column) a store
// instruction writes one full 256-byte row of layout W.
typedef float stem_f4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) stem_mfma_kernel(const float* __restrict__ planes, const float* __restrict__ w,
                                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                                        float* __restrict__ y, int batch, int H, int W,
                                                        const unsigned* __restrict__ batch_dev) {
    if (batch_dev) {
        const int real = (int)*batch_dev;
        batch = real < batch ? real : batch;
    }
    __shared__ float img[4][3][100];                   // per wave: zero-bordered 10x10 planes of its board
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qs = lane >> 4, l16 = lane & 15;         // K slot of this lane inside a step, column of the 16x16 tile
    const int tile = l16 >> 2, j = l16 & 3;
    // A fragments: lane -> (channel 16 kb + l16, K index 4 s + qs)
    float a[4][7];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int s7 = 0; s7 < 7; ++s7) {
            const int q = 4 * s7 + qs;
            a[kb][s7] = q < 27 ? w[(16 * kb + l16) * 27 + q] : 0.0f;
        }
    // B fragments: lane -> (K index 4 s + qs, cell (i = cb, tile, j)); offset of tap q = plane * 100 + dy * 10 + dx
    int qoff[7];
#pragma unroll
    for (int s7 = 0; s7 < 7; ++s7) {
        const int q = 4 * s7 + qs;
        qoff[s7] = q < 27 ? (q / 9) * 100 + ((q % 9) / 3) * 10 + (q % 3) : -1;
    }
    const int cell0 = (4 * (tile >> 1)) * 10 + 4 * (tile & 1) + j;     // + cb * 10 for row-in-tile cb
    const int col = 4 * (tile & 1) + j;
    float sc[4][4], sh[4][4];                          // channel 16 kb + 4 qs + r  (row of the accumulator tile)
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            sc[kb][r] = scale[16 * kb + 4 * qs + r];
            sh[kb][r] = shift[16 * kb + 4 * qs + r];
        }
    const int HW = H * W;
    float* im = &img[wave][0][0];
    for (int i = lane; i < 300; i += 64) im[i] = 0.0f;
    const int nwaves = (int)gridDim.x * 4;
    for (int n = (int)blockIdx.x * 4 + wave; n < batch; n += nwaves) {
        for (int e = lane; e < 3 * HW; e += 64) {
            const int p = e / HW, cell = e % HW;
            im[p * 100 + (cell / W + 1) * 10 + cell % W + 1] = planes[(size_t)n * 3 * HW + e];
        }
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            stem_f4 acc[4];
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) acc[kb] = (stem_f4){ 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
            for (int s7 = 0; s7 < 7; ++s7) {
                const float b = qoff[s7] >= 0 ? im[qoff[s7] + cell0 + cb * 10] : 0.0f;
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) acc[kb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kb][s7], b, acc[kb], 0, 0, 0);
            }
            const bool on_board = 4 * (tile >> 1) + cb < H && col < W;
            float* yo = y + (size_t)n * 4096 + cb * 64 + qs * 16 + l16;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = acc[kb][r] * sc[kb][r] + sh[kb][r];
                    yo[(4 * kb + r) * 256] = on_board ? (v > 0.0f ? v : 0.0f) : 0.0f;
                }
        }
    }
}

// The whole tail of the network in one pass over the trunk output (layout W): both 1x1 head convolutions + bias + ReLU
// (grid_networks.py:44,49), the policy FC, and the value FC -> ReLU -> FC -> tanh (:45,50-51), written straight into the
// engine's logits / value buffers.  16 boards per workgroup; the head maps of a board stay in LDS between the two
// stages, the FC weights (<= 50 KB) sit in LDS.  HBM-bound on the 16 KB per board it reads.
constexpr int TAIL_NB = 16, TAIL_MAXIN = 128, TAIL_MAXA = 80, TAIL_HID = 64;

template <int PC, int VC>
__global__ void __launch_bounds__(256) tail_kernel(const float* __restrict__ x, const float* __restrict__ hw,
                                                   const float* __restrict__ hb, const float* __restrict__ pfc_w,
                                                   const float* __restrict__ pfc_b, const float* __restrict__ vfc1_w,
                                                   const float* __restrict__ vfc1_b, const float* __restrict__ vfc2_w,
                                                   const float* __restrict__ vfc2_b, float* __restrict__ logits,
                                                   float* __restrict__ value, int batch, int H, int W, int A, int HID,
                                                   const unsigned* __restrict__ batch_dev, const float* __restrict__ maps_in) {
    // maps_in != null: the head maps [batch][OC * H * W] were already produced (by the last trunk convolution): FC layers only
    if (batch_dev) {
        const int real = (int)*batch_dev;
        batch = real < batch ? real : batch;
        if ((int)blockIdx.x * TAIL_NB >= batch) return;
    }
    constexpr int OC = PC + VC;
    __shared__ float s_hw[OC * 64 + OC];
    __shared__ float s_maps[TAIL_NB][OC * 64];         // [board][o][cell], cell = row * W + col
    __shared__ float s_pw[TAIL_MAXIN * TAIL_MAXA];
    __shared__ float s_vw[64 * TAIL_HID];
    const int tid = (int)threadIdx.x;
    const int HW = H * W, PIN = PC * HW, VIN = VC * HW;
    for (int i = tid; i < OC * 64 + OC; i += 256) s_hw[i] = i < OC * 64 ? hw[i] : hb[i - OC * 64];
    for (int i = tid; i < PIN * A; i += 256) s_pw[i] = pfc_w[i];
    for (int i = tid; i < VIN * HID; i += 256) s_vw[i] = vfc1_w[i];
    if (maps_in) {
        const int n_first = (int)blockIdx.x * TAIL_NB;
        for (int i = tid; i < TAIL_NB * OC * HW; i += 256) {
            const int bb = i / (OC * HW);
            if (n_first + bb < batch) s_maps[bb][i - bb * (OC * HW)] = maps_in[(size_t)n_first * (OC * HW) + i];
        }
    }
    __syncthreads();

    // stage 1: thread = (board, row-in-tile i, tile): 4 cells x OC maps, all 64 channels
    const int b = tid >> 4, i = (tid >> 2) & 3, tile = tid & 3;
    const int n = (int)blockIdx.x * TAIL_NB + b;
    float acc[OC][4];
#pragma unroll
    for (int o = 0; o < OC; ++o)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[o][j] = 0.0f;
    if (n < batch && !maps_in) {
        typedef float f4v __attribute__((ext_vector_type(4)));
        const f4v* xp = (const f4v*)(x + (size_t)n * 4096 + i * 64 + tile * 4);
#pragma unroll 4
        for (int g = 0; g < 16; ++g)
#pragma unroll
            for (int cs = 0; cs < 4; ++cs) {
                const f4v v = __builtin_nontemporal_load(xp + g * 64 + cs * 4);
                const int k = 16 * (g >> 2) + 4 * cs + (g & 3);
#pragma unroll
                for (int o = 0; o < OC; ++o) {
                    const float w = s_hw[o * 64 + k];
                    acc[o][0] += w * v.x;
                    acc[o][1] += w * v.y;
                    acc[o][2] += w * v.z;
                    acc[o][3] += w * v.w;
                }
            }
    }
    const int row = 4 * (tile >> 1) + i, col0 = 4 * (tile & 1);
#pragma unroll
    for (int o = 0; o < OC; ++o)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (!maps_in && row < H && col0 + j < W) {
                const float r = acc[o][j] + s_hw[OC * 64 + o];
                s_maps[b][o * HW + row * W + col0 + j] = r > 0.0f ? r : 0.0f;
            }
    __syncthreads();

    // stage 2: thread = (board, t16): policy outputs a = t16, t16 + 16, ...; hidden units j = t16, t16 + 16, ...
    const int t16 = tid & 15;
    if (n < batch) {
        const float* pm = s_maps[b];                   // policy maps first: [PC * HW], then value maps [VC * HW]
        for (int a = t16; a < A; a += 16) {
            float sum = pfc_b[a];
            for (int q = 0; q < PIN; ++q) sum += pm[q] * s_pw[q * A + a];
            logits[(size_t)n * A + a] = sum;
        }
        const float* vm = pm + PIN;
        float part = 0.0f;
        for (int j = t16; j < HID; j += 16) {
            float h = vfc1_b[j];
            for (int q = 0; q < VIN; ++q) h += vm[q] * s_vw[q * HID + j];
            h = h > 0.0f ? h : 0.0f;
            part += h * vfc2_w[j];
        }
#pragma unroll
        for (int m = 8; m >= 1; m >>= 1) part += __shfl_xor(part, m, 16);
        if (t16 == 0) value[n] = tanhf(part + vfc2_b[0]);
    }
}

// The FC layers of the tail on the fp32 matrix cores, for head maps that the last trunk convolution already produced
// (maps_in of tail_kernel above; sprl_wino_conv64_heads).  tail_kernel's stage 2 walks the FC weights in LDS once per board and
// output - two LDS reads per multiply-add, 40 us per round of 13.5 k Othello boards, LDS-bound.  Here a workgroup takes 64 boards,
// a wave 16 of them: logits[16 boards][A] = maps[16][PIN] x W[PIN][A] as v_mfma_f32_16x16x4_f32 tiles with the boards as rows
// (A operand: one map value per lane, B operand: one weight per lane, both conflict-free LDS reads: the map rows are 194 floats
// apart, the weight rows 80), the hidden layer of the value head the same way, then bias / ReLU / the 64-long dot product with
// vfc2 (a 16-lane reduction) / tanh in registers.  Weights past A / HID and map columns past the board are zero in LDS, so the
// contraction is padded to a multiple of 4 without a branch.
constexpr int TFC_NB = 64, TFC_LDW = 80, TFC_LDM = 194, TFC_MAXNBLK = TAIL_MAXA / 16;

__global__ void __launch_bounds__(256) tail_fc_mfma_kernel(const float* __restrict__ maps_in, const float* __restrict__ pfc_w,
                                                           const float* __restrict__ pfc_b, const float* __restrict__ vfc1_w,
                                                           const float* __restrict__ vfc1_b, const float* __restrict__ vfc2_w,
                                                           const float* __restrict__ vfc2_b, float* __restrict__ logits,
                                                           float* __restrict__ value, int batch, int PIN, int VIN, int A, int HID,
                                                           const unsigned* __restrict__ batch_dev) {
    typedef float f4v __attribute__((ext_vector_type(4)));
    if (batch_dev) {
        const int real = (int)*batch_dev;
        batch = real < batch ? real : batch;
    }
    const int n0 = (int)blockIdx.x * TFC_NB;
    if (n0 >= batch) return;
    const int KP = (PIN + 3) >> 2, KV = (VIN + 3) >> 2, OCHW = PIN + VIN;
    extern __shared__ float fsh[];
    float* const s_pw = fsh;                           // [4 KP][80]
    float* const s_vw = s_pw + 4 * KP * TFC_LDW;       // [4 KV][80]
    float* const s_maps = s_vw + 4 * KV * TFC_LDW;     // [64][194]
    float* const s_pb = s_maps + TFC_NB * TFC_LDM;     // [80] policy bias, [64] hidden bias, [64] vfc2
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // LDS fill, eight independent loads in flight per thread (a rolled one-load loop is a chain of ~100 L2 round trips)
    auto fill = [&](float* dst, int count, auto&& src) {
        for (int i0 = tid; i0 < count; i0 += 256 * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = i0 + 256 * u < count ? src(i0 + 256 * u) : 0.0f;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (i0 + 256 * u < count) dst[i0 + 256 * u] = v[u];
        }
    };
    fill(s_pw, 4 * KP * TFC_LDW, [&](int i) {
        const int q = i / TFC_LDW, a = i - q * TFC_LDW;
        return (q < PIN && a < A) ? pfc_w[q * A + a] : 0.0f;
    });
    fill(s_vw, 4 * KV * TFC_LDW, [&](int i) {
        const int q = i / TFC_LDW, j = i - q * TFC_LDW;
        return (q < VIN && j < HID) ? vfc1_w[q * HID + j] : 0.0f;
    });
    fill(s_maps, TFC_NB * TFC_LDM, [&](int i) {
        const int b = i / TFC_LDM, c = i - b * TFC_LDM;
        return (c < OCHW && n0 + b < batch) ? maps_in[(size_t)(n0 + b) * OCHW + c] : 0.0f;
    });
    if (tid < TFC_LDW) s_pb[tid] = tid < A ? pfc_b[tid] : 0.0f;
    if (tid < 64) {
        s_pb[TFC_LDW + tid] = tid < HID ? vfc1_b[tid] : 0.0f;
        s_pb[TFC_LDW + 64 + tid] = tid < HID ? vfc2_w[tid] : 0.0f;
    }
    __syncthreads();

    const int row = lane & 15, kq = lane >> 4;         // A operand: board `row` of this wave, contraction index 4 s + kq
    const float* am = s_maps + (wave * 16 + row) * TFC_LDM + kq;
    const float* bw = s_pw + kq * TFC_LDW + row;        // B operand: weight row 4 s + kq, output column 16 nb + (lane & 15)
    const int nblk = (A + 15) >> 4;
    f4v accp[TFC_MAXNBLK], accv[4];
#pragma unroll
    for (int nb = 0; nb < TFC_MAXNBLK; ++nb) accp[nb] = (f4v){ 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) accv[nb] = (f4v){ 0.0f, 0.0f, 0.0f, 0.0f };
    for (int s = 0; s < KP; ++s) {
        const float av = am[4 * s];
#pragma unroll
        for (int nb = 0; nb < TFC_MAXNBLK; ++nb)
            if (nb < nblk) accp[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bw[4 * s * TFC_LDW + 16 * nb], accp[nb], 0, 0, 0);
    }
    const float* bv = s_vw + kq * TFC_LDW + row;
    for (int s = 0; s < KV; ++s) {
        const float av = am[PIN + 4 * s];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) accv[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[4 * s * TFC_LDW + 16 * nb], accv[nb], 0, 0, 0);
    }
    // D: lane holds boards 4 kq + r (r = 0..3) of this wave, column (lane & 15) of each 16-wide block
    const int nrow0 = n0 + wave * 16 + 4 * kq;
#pragma unroll
    for (int nb = 0; nb < TFC_MAXNBLK; ++nb) {
        const int a = 16 * nb + row;
        if (nb < nblk && a < A) {
            const float bias = s_pb[a];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (nrow0 + r < batch) logits[(size_t)(nrow0 + r) * A + a] = accp[nb][r] + bias;
        }
    }
    float part[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        const float hb1 = s_pb[TFC_LDW + 16 * nb + row], w2 = s_pb[TFC_LDW + 64 + 16 * nb + row];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float h = accv[nb][r] + hb1;
            part[r] += (h > 0.0f ? h : 0.0f) * w2;
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int m = 8; m >= 1; m >>= 1) part[r] += __shfl_xor(part[r], m, 16);
        if (row == 0 && nrow0 + r < batch) value[nrow0 + r] = tanhf(part[r] + vfc2_b[0]);
    }
}
}  // namespace

// The same tail for ANY board (Go 9x9 / 19x19), trunk output in NCHW, in two kernels that read the board count from device
// memory (batch_dev), so nothing here needs the host: they replace the heads kernel + three library GEMMs (+ their at::empty
// allocations and output copies) of the earlier wide-board path.
//   tail_heads_value_kernel: both 1x1 head convolutions + bias + ReLU for NB boards (maps in LDS), the policy maps written out
//       [batch][PC * HW], and the value head FC -> ReLU -> FC -> tanh;
//   policy_fc_kernel: logits[B][A] = maps[B][PC * HW] x W[PC * HW][A] + b as a (boards / NB) x (A / 128) grid - at 19x19 the
//       722 x 362 product of one round's ~1 k boards is 370 workgroups of a few microseconds (a single-kernel form with one
//       workgroup per 8 boards walked the 722-long contraction twice per workgroup and cost 10 % of the forward).  Thread =
//       one output column x one half of the contraction; consecutive threads read consecutive weights of a row (coalesced, each
//       weight once per workgroup, reused for NB boards in registers), the map values are LDS broadcasts.
template <int OC, int NB>
__global__ void __launch_bounds__(256) tail_heads_value_kernel(const float* __restrict__ x, const float* __restrict__ hw,
                                                               const float* __restrict__ hb, const float* __restrict__ vfc1_w,
                                                               const float* __restrict__ vfc1_b, const float* __restrict__ vfc2_w,
                                                               const float* __restrict__ vfc2_b, float* __restrict__ pmaps,
                                                               float* __restrict__ value, int batch, int HW, int PC, int HID,
                                                               const unsigned* __restrict__ batch_dev, int tile_m, int Wb) {
    // tile_m = 0: x is NCHW; 3 / 4: x is layout T of the any-board trunk kernel (board width Wb; row pitch = capacity x tiles)
    const int cap = batch;
    if (batch_dev) {
        const int real = (int)*batch_dev;
        batch = real < batch ? real : batch;
    }
    const int n0 = (int)blockIdx.x * NB;
    if (n0 >= batch) return;
    extern __shared__ float tsh[];
    float* const s_hw = tsh;                           // [OC][64] + [OC]
    float* const s_maps = tsh + OC * 64 + OC;          // [NB][OC * HW]: policy maps first, then value maps
    float* const s_hid = s_maps + NB * OC * HW;        // [NB][HID]
    const int tid = (int)threadIdx.x;
    const int nb = batch - n0 < NB ? batch - n0 : NB;
    for (int i = tid; i < OC * 64 + OC; i += 256) s_hw[i] = i < OC * 64 ? hw[i] : hb[i - OC * 64];
    for (int i = tid; i < NB * OC * HW; i += 256) s_maps[i] = 0.0f;     // boards past the batch contribute zeros below
    __syncthreads();
    // heads: thread = (board, cell); consecutive lanes read consecutive cells of one channel plane
    for (int e = tid; e < nb * HW; e += 256) {
        const int b = e / HW, cell = e - b * HW;
        float acc[OC];
#pragma unroll
        for (int o = 0; o < OC; ++o) acc[o] = 0.0f;
        if (tile_m) {
            typedef float tf4 __attribute__((ext_vector_type(4)));
            const int Hb = HW / Wb, row = cell / Wb, col = cell - row * Wb;
            const int TXt = (Wb + tile_m - 1) / tile_m, TPB = TXt * ((Hb + tile_m - 1) / tile_m), MC = tile_m * tile_m;
            const int cell_t = (row % tile_m) * tile_m + col % tile_m, tile_t = (row / tile_m) * TXt + col / tile_m;
            const size_t TT = (size_t)cap * TPB;
            const tf4* xp = (const tf4*)x + (size_t)cell_t * TT + (size_t)(n0 + b) * TPB + tile_t;
#pragma unroll 4
            for (int q = 0; q < 16; ++q) {
                const tf4 v = xp[(size_t)q * MC * TT];
#pragma unroll
                for (int o = 0; o < OC; ++o)
                    acc[o] += s_hw[o * 64 + 4 * q] * v[0] + s_hw[o * 64 + 4 * q + 1] * v[1] + s_hw[o * 64 + 4 * q + 2] * v[2] + s_hw[o * 64 + 4 * q + 3] * v[3];
            }
        } else {
            const float* xp = x + ((size_t)(n0 + b) * 64) * HW + cell;
#pragma unroll 8
            for (int c = 0; c < 64; ++c) {
                const float v = xp[(size_t)c * HW];
#pragma unroll
                for (int o = 0; o < OC; ++o) acc[o] += s_hw[o * 64 + c] * v;
            }
        }
#pragma unroll
        for (int o = 0; o < OC; ++o) {
            const float r = acc[o] + s_hw[OC * 64 + o];
            s_maps[b * (OC * HW) + o * HW + cell] = r > 0.0f ? r : 0.0f;
        }
    }
    __syncthreads();
    const int PIN = PC * HW, VIN = (OC - PC) * HW;
    for (int e = tid; e < nb * PIN; e += 256) {
        const int b = e / PIN, q = e - b * PIN;
        pmaps[(size_t)(n0 + b) * PIN + q] = s_maps[b * (OC * HW) + q];
    }
    // value head: FC1 + ReLU, thread = (hidden unit j, quarter of the contraction); then FC2 + tanh per board
    const int j = tid & 63, part = tid >> 6;           // HID <= 64 (checked by the host)
    {
        float acc[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[b] = 0.0f;
        const int q0 = (VIN * part) / 4, q1 = (VIN * (part + 1)) / 4;
        if (j < HID)
            for (int q = q0; q < q1; ++q) {
                const float w = vfc1_w[(size_t)q * HID + j];
#pragma unroll
                for (int b = 0; b < NB; ++b) acc[b] += s_maps[b * (OC * HW) + PIN + q] * w;
            }
        // the four partial sums of a hidden unit are added in a fixed order through LDS
        float* const s_part = s_hid + NB * 64;         // [4][NB][64]
#pragma unroll
        for (int b = 0; b < NB; ++b) s_part[(part * NB + b) * 64 + j] = acc[b];
        __syncthreads();
        if (part == 0 && j < HID) {
            const float bias = vfc1_b[j], w2 = vfc2_w[j];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const float h = ((s_part[(0 * NB + b) * 64 + j] + s_part[(1 * NB + b) * 64 + j]) + s_part[(2 * NB + b) * 64 + j]) +
                                s_part[(3 * NB + b) * 64 + j] + bias;
                s_hid[b * 64 + j] = (h > 0.0f ? h : 0.0f) * w2;
            }
        }
        __syncthreads();
    }
    if (tid < nb) {
        float sum = vfc2_b[0];
        for (int jj = 0; jj < HID; ++jj) sum += s_hid[tid * 64 + jj];
        value[n0 + tid] = tanhf(sum);
    }
}

template <int NB>
__global__ void __launch_bounds__(256) policy_fc_kernel(const float* __restrict__ pmaps, const float* __restrict__ pfc_w,
                                                        const float* __restrict__ pfc_b, float* __restrict__ logits, int batch,
                                                        int PIN, int A, const unsigned* __restrict__ batch_dev) {
    if (batch_dev) {
        const int real = (int)*batch_dev;
        batch = real < batch ? real : batch;
    }
    const int n0 = (int)blockIdx.x * NB;
    if (n0 >= batch) return;
    extern __shared__ float fsh[];
    float* const s_in = fsh;                           // [NB][PIN]
    float* const s_red = fsh + NB * PIN;               // [NB][128]: the second half's partial sums
    const int tid = (int)threadIdx.x;
    const int nb = batch - n0 < NB ? batch - n0 : NB;
    for (int i = tid; i < NB * PIN; i += 256) s_in[i] = i < nb * PIN ? pmaps[(size_t)n0 * PIN + i] : 0.0f;
    __syncthreads();
    const int al = tid & 127, kh = tid >> 7;
    const int a = (int)blockIdx.y * 128 + al;
    const int q0 = kh ? PIN / 2 : 0, q1 = kh ? PIN : PIN / 2;
    float acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b] = 0.0f;
    if (a < A) {
        const float* wp = pfc_w + a;
        int q = q0;
        for (; q + 4 <= q1; q += 4) {                  // four weights in flight per thread
            const float w0 = wp[(size_t)q * A], w1 = wp[(size_t)(q + 1) * A], w2 = wp[(size_t)(q + 2) * A], w3 = wp[(size_t)(q + 3) * A];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const float* m = s_in + b * PIN + q;
                acc[b] = acc[b] + m[0] * w0 + m[1] * w1 + m[2] * w2 + m[3] * w3;
            }
        }
        for (; q < q1; ++q) {
            const float w0 = wp[(size_t)q * A];
#pragma unroll
            for (int b = 0; b < NB; ++b) acc[b] += s_in[b * PIN + q] * w0;
        }
    }
    if (kh) {
#pragma unroll
        for (int b = 0; b < NB; ++b) s_red[b * 128 + al] = acc[b];
    }
    __syncthreads();
    if (!kh && a < A) {
        const float bias = pfc_b[a];
#pragma unroll
        for (int b = 0; b < NB; ++b)
            if (b < nb) logits[(size_t)(n0 + b) * A + a] = (acc[b] + s_red[b * 128 + al]) + bias;
    }
}

// planes: [batch][P][H][W] (the engine's dense network batch), w: [64][P][3][3], y: layout W.  -1: no kernel for this P.
// Stem for boards wider than 8 (Go 9x9 / 19x19; any H x W), NCHW in and out: conv3x3 (P planes -> 64 channels, padding 1) +
// folded BatchNorm/bias + ReLU.  thread = one board cell, its 3x3xP patch in registers (the planes are read straight from
// global memory: neighbouring threads read neighbouring cells), weights wave-uniform (scalar loads), 64 channels per thread,
// stores coalesced over the cells of a channel.  Replaces the library's convolution for this layer (its generic kernel took
// 3 ms per call on 17-plane Go batches, profiles/r01k_go_kernel_stats.csv).
template <int P>
__global__ void __launch_bounds__(256) stem_nchw_kernel(const float* __restrict__ planes, const float* __restrict__ w,
                                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                                        float* __restrict__ y, long long cells, int H, int W) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= cells) return;
    const int HW = H * W;
    const long long n = e / HW;
    const int cell = (int)(e % HW), row = cell / W, col = cell % W;
    const float* in = planes + (size_t)n * P * HW;
    float d[P][9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int r = row + t / 3 - 1, c = col + t % 3 - 1;
        const bool ok = r >= 0 && r < H && c >= 0 && c < W;
        const int off = ok ? r * W + c : 0;
#pragma unroll
        for (int p = 0; p < P; ++p) d[p][t] = ok ? in[(size_t)p * HW + off] : 0.0f;
    }
    float* yo = y + (size_t)n * 64 * HW + cell;
    for (int k = 0; k < 64; ++k) {
        const float* wk = w + (size_t)k * P * 9;
        float acc = 0.0f;
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
            for (int t = 0; t < 9; ++t) acc += wk[p * 9 + t] * d[p][t];
        acc = acc * scale[k] + shift[k];
        yo[(size_t)k * HW] = acc > 0.0f ? acc : 0.0f;
    }
}

// The same stem on the fp32 matrix cores (v_mfma_f32_16x16x4_f32), any board and plane count: out[64 ch][cells] =
// W[64][9 P] x patches[9 P][cells].  A wave takes 16 consecutive cells of the batch (board-major, so a store instruction
// writes 64 contiguous bytes per channel) and all 64 channels.  K is ordered (plane group of 4, tap): in step s = 9 pg + t the
// four K slots are planes 4 pg .. 4 pg + 3 at tap t (planes >= P carry zero weights), so a lane's patch value sits at
// tapoff[t] + planeoff[pg] - two small per-lane tables instead of one offset per step.  The patch values come straight from the
// NCHW planes through a buffer descriptor (off-board taps and missing planes get offsets past the tensor, which read as 0); the
// A fragments of all steps sit in LDS in lane order, one 16-byte read per step for the four channel blocks.
// Go 9x9, 8192 boards x 17 planes: 0.84 ms (VALU form above) -> see DESIGN.md section 5.
template <int P>
__global__ void __launch_bounds__(256, 3) stem_mfma_nchw_kernel(const float* __restrict__ planes, const float* __restrict__ w,
                                                             const float* __restrict__ scale, const float* __restrict__ shift,
                                                             float* __restrict__ y, long long cells, int H, int W,
                                                             const unsigned* __restrict__ batch_dev, int tile_m) {
    // tile_m = 0: y is NCHW; 3 or 4: y is layout T of the any-board trunk kernel (cnn_wino.hip): [q][cell][n * tiles + tile][4 channels]
    constexpr int PG = (P + 3) / 4, KS = PG * 9;
    const long long cap_boards = cells / ((long long)H * W);      // layout T: the row pitch is capacity x tiles per board
    if (batch_dev) {                                   // the real board count is on the device; `cells` is the capacity
        const long long real = (long long)*batch_dev * H * W;
        cells = real < cells ? real : cells;
        if ((long long)blockIdx.x * 64 >= cells) return;
    }
    __shared__ __attribute__((aligned(16))) float wsh[KS * 64 * 4];          // [step][lane][channel block]
    __shared__ float scsh[2][64];
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qs = lane >> 4, l16 = lane & 15;
    for (int e = tid; e < KS * 64 * 4; e += 256) {
        const int kb = e & 3, ln = (e >> 2) & 63, st = e >> 8;
        const int plane = 4 * (st / 9) + (ln >> 4), tap = st % 9;
        wsh[e] = plane < P ? w[((size_t)(16 * kb + (ln & 15)) * P + plane) * 9 + tap] : 0.0f;
    }
    if (tid < 64) {
        scsh[0][tid] = scale[tid];
        scsh[1][tid] = shift[tid];
    }
    __syncthreads();
    const int HW = H * W;
    const unsigned in_bytes = (unsigned)(cells * P * 4);                      // the host keeps this below 1 GiB
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)planes, 0, in_bytes, 0x00020000);
    const long long tiles = (cells + 15) / 16;
    for (long long tile = (long long)blockIdx.x * 4 + wave; tile < tiles; tile += (long long)gridDim.x * 4) {
        const long long e = tile * 16 + l16;
        const bool live = e < cells;
        const int n = live ? (int)(e / HW) : 0, cell = live ? (int)(e % HW) : 0;
        const int row = cell / W, col = cell % W;
        int tapoff[9], pgoff[PG];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int r = row + t / 3 - 1, c = col + t % 3 - 1;
            tapoff[t] = (live && r >= 0 && r < H && c >= 0 && c < W) ? (n * P * HW + r * W + c) * 4 : (int)0x80000000;
        }
#pragma unroll
        for (int pg = 0; pg < PG; ++pg) pgoff[pg] = 4 * pg + qs < P ? (4 * pg + qs) * HW * 4 : 0x40000000;
        float b[KS];
#pragma unroll
        for (int st = 0; st < KS; ++st)
            b[st] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rp, tapoff[st % 9] + pgoff[st / 9], 0, 0));
        stem_f4 acc[4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) acc[kb] = (stem_f4){ 0.0f, 0.0f, 0.0f, 0.0f };
        stem_f4 a = *(const stem_f4*)&wsh[lane * 4];
#pragma unroll
        for (int st = 0; st < KS; ++st) {
            // the next step's fragments are requested before this step's MFMAs and nothing moves across the barrier: without it
            // the scheduler hoists all KS reads (4 KS registers) to the top
            const stem_f4 an = *(const stem_f4*)&wsh[((st + 1 < KS ? st + 1 : st) * 64 + lane) * 4];
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) acc[kb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kb], b[st], acc[kb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            a = an;
        }
        // accumulator row r of block kb = channel 16 kb + 4 qs + r, column = this lane's cell
        if (live && tile_m) {
            // layout T: the four channels of a block (quad 4 kb + qs) of this cell are one 16-byte vector
            const int TXt = (W + tile_m - 1) / tile_m, TPB = TXt * ((H + tile_m - 1) / tile_m), MC = tile_m * tile_m;
            const int cell_t = (row % tile_m) * tile_m + col % tile_m, tile_t = (row / tile_m) * TXt + col / tile_m;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                stem_f4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int k = 16 * kb + 4 * qs + r;
                    const float t = acc[kb][r] * scsh[0][k] + scsh[1][k];
                    v[r] = t > 0.0f ? t : 0.0f;
                }
                *(stem_f4*)(y + (((size_t)(4 * kb + qs) * MC + cell_t) * (size_t)(cap_boards * TPB) + (size_t)n * TPB + tile_t) * 4) = v;
            }
        } else if (live) {
            float* yo = y + ((size_t)n * 64) * HW + cell;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int k = 16 * kb + 4 * qs + r;
                    const float v = acc[kb][r] * scsh[0][k] + scsh[1][k];
                    yo[(size_t)k * HW] = v > 0.0f ? v : 0.0f;
                }
        }
    }
}

// valu != 0: the VALU form (a lab / test variant; it cannot read the count from the device or write layout T)
static int stem_any_board(const float* planes, const float* w, const float* scale, const float* shift, float* y, long long batch, int P,
                          int H, int W, const unsigned* batch_dev, int tile_m, int valu, void* stream) {
    if (batch <= 0) return 0;
    const long long cells = batch * H * W;
    if ((batch_dev || tile_m) && (cells * P * 4 >= 0x40000000LL || (P != 3 && P != 17))) return -1;   // only the MFMA form does these
    if (cells * P * 4 < 0x40000000LL && (batch_dev || tile_m || !valu)) {
        long long blocks = (cells / 16 + 3) / 4;
        if (blocks > 256 * 6) blocks = 256 * 6;        // grid-stride over 16-cell tiles: the weight fragments are staged once per workgroup
        if (blocks < 1) blocks = 1;
        const dim3 g((unsigned)blocks), bl(256);
        if (P == 3) hipLaunchKernelGGL(stem_mfma_nchw_kernel<3>, g, bl, 0, (hipStream_t)stream, planes, w, scale, shift, y, cells, H, W, batch_dev, tile_m);
        else if (P == 17) hipLaunchKernelGGL(stem_mfma_nchw_kernel<17>, g, bl, 0, (hipStream_t)stream, planes, w, scale, shift, y, cells, H, W, batch_dev, tile_m);
        else return -1;
        return hipGetLastError() == hipSuccess ? 0 : -2;
    }
    const dim3 grid((unsigned)((cells + 255) / 256)), block(256);
    if (P == 3) hipLaunchKernelGGL(stem_nchw_kernel<3>, grid, block, 0, (hipStream_t)stream, planes, w, scale, shift, y, cells, H, W);
    else if (P == 17) hipLaunchKernelGGL(stem_nchw_kernel<17>, grid, block, 0, (hipStream_t)stream, planes, w, scale, shift, y, cells, H, W);
    else return -1;
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
extern "C" int sprl_stem_conv3x3_nchw_dev(const float* planes, const float* w, const float* scale, const float* shift, float* y,
                                          long long batch, int P, int H, int W, const unsigned* batch_dev, void* stream) {
    return stem_any_board(planes, w, scale, shift, y, batch, P, H, W, batch_dev, 0, 0, stream);
}
extern "C" int sprl_stem_conv3x3_nchw(const float* planes, const float* w, const float* scale, const float* shift, float* y,
                                      long long batch, int P, int H, int W, void* stream) {
    return stem_any_board(planes, w, scale, shift, y, batch, P, H, W, nullptr, 0, 0, stream);
}
// the VALU form of the same stem (tests, lab): no environment switch picks it, only this entry point
extern "C" int sprl_stem_conv3x3_nchw_valu(const float* planes, const float* w, const float* scale, const float* shift, float* y,
                                           long long batch, int P, int H, int W, void* stream) {
    return stem_any_board(planes, w, scale, shift, y, batch, P, H, W, nullptr, 0, 1, stream);
}
// the same stem writing layout T (tile = 3 or 4) for the any-board trunk kernel
extern "C" int sprl_stem_conv3x3_t(const float* planes, const float* w, const float* scale, const float* shift, float* y,
                                   long long batch, int P, int H, int W, int tile, const unsigned* batch_dev, void* stream) {
    if (tile != 3 && tile != 4) return -1;
    return stem_any_board(planes, w, scale, shift, y, batch, P, H, W, batch_dev, tile, 0, stream);
}

// valu != 0: the VALU form also for 3 planes (lab / test variant)
extern "C" int sprl_stem_conv3x3_w_form(const float* planes, const float* w, const float* scale, const float* shift, float* y,
                                        int batch, int P, int H, int W, int valu, const unsigned* batch_dev, void* stream) {
    if (batch <= 0) return 0;
    if (H > 8 || W > 8) return -1;
    const dim3 grid((unsigned)((batch + 3) / 4)), block(256);
    if (P == 3 && !valu) {
        int blocks = (batch + 3) / 4;
        if (blocks > 256 * 8) blocks = 256 * 8;        // grid-stride over boards: the weight fragments are loaded once per wave
        hipLaunchKernelGGL(stem_mfma_kernel, dim3((unsigned)blocks), block, 0, (hipStream_t)stream, planes, w, scale, shift, y, batch, H, W,
                           batch_dev);
    } else if (P == 3) hipLaunchKernelGGL(stem_kernel<3>, grid, block, 0, (hipStream_t)stream, planes, w, scale, shift, y, batch, H, W, batch_dev);
    else if (P == 17) hipLaunchKernelGGL(stem_kernel<17>, grid, block, 0, (hipStream_t)stream, planes, w, scale, shift, y, batch, H, W, batch_dev);
    else return -1;
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
extern "C" int sprl_stem_conv3x3_w(const float* planes, const float* w, const float* scale, const float* shift, float* y,
                                   int batch, int P, int H, int W, const unsigned* batch_dev, void* stream) {
    return sprl_stem_conv3x3_w_form(planes, w, scale, shift, y, batch, P, H, W, 0, batch_dev, stream);
}

// board_w > 0: x is in layout W (cnn_wino.hip; needs C == 64), board_w == 0: NCHW
extern "C" int sprl_heads_conv1x1_relu(const float* x, const float* w, const float* bias, float* out_p, float* out_v,
                                       int64_t batch, int C, int HW, int PC, int VC, int board_w, void* stream) {
    if (board_w > 0 && C != 64) return -1;
    const int OC = PC + VC;
    const int64_t n_pos = batch * HW;
    int64_t blocks = (n_pos + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    const size_t lds = (size_t)(OC * C + OC) * sizeof(float);
#define LAUNCH(N) hipLaunchKernelGGL(heads_kernel<N>, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, x, w, bias, out_p, out_v, n_pos, C, HW, PC, board_w)
    switch (OC) {
    case 2: LAUNCH(2); break;
    case 3: LAUNCH(3); break;
    case 4: LAUNCH(4); break;
    default: return -1;
    }
#undef LAUNCH
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// any H*W: one element per lane (boards such as 9x9 or 19x19, where a float4 would straddle channels)
template <bool RESIDUAL>
__global__ void __launch_bounds__(256) bn_relu_scalar_kernel(float* __restrict__ x, const float* __restrict__ res,
                                                             const float* __restrict__ scale, const float* __restrict__ shift,
                                                             int64_t n, int hw, int channels) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)((i / hw) % channels);
        float v = x[i] * scale[c] + shift[c];
        if (RESIDUAL) v += res[i];
        x[i] = v > 0.0f ? v : 0.0f;
    }
}

extern "C" int sprl_bn_relu_inplace(float* x, const float* residual, const float* scale, const float* shift,
                                    int64_t numel, int channels, int hw, void* stream) {
    if (hw % 4 != 0 || numel % 4 != 0) {
        int64_t blocks = (numel + 255) / 256;
        if (blocks > 256 * 32) blocks = 256 * 32;
        if (residual)
            hipLaunchKernelGGL(bn_relu_scalar_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, residual, scale,
                               shift, numel, hw, channels);
        else
            hipLaunchKernelGGL(bn_relu_scalar_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x,
                               (const float*)nullptr, scale, shift, numel, hw, channels);
        return hipGetLastError() == hipSuccess ? 0 : -2;
    }
    const int64_t n_vec = numel / 4;
    int64_t blocks = (n_vec + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;          // grid-stride over 16 workgroups per CU
    if (residual)
        hipLaunchKernelGGL(bn_relu_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (float4*)x,
                           (const float4*)residual, scale, shift, n_vec, hw / 4, channels);
    else
        hipLaunchKernelGGL(bn_relu_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (float4*)x,
                           (const float4*)nullptr, scale, shift, n_vec, hw / 4, channels);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int sprl_tail_fc(const float* x, const float* maps_in, const float* hw, const float* hb, const float* pfc_w,
                            const float* pfc_b, const float* vfc1_w, const float* vfc1_b, const float* vfc2_w, const float* vfc2_b,
                            float* logits, float* value, int batch, int H, int W, int PC, int VC, int A, int HID,
                            const unsigned* batch_dev, void* stream);
extern "C" int sprl_tail_fc_form(const float* x, const float* maps_in, const float* hw, const float* hb, const float* pfc_w,
                                 const float* pfc_b, const float* vfc1_w, const float* vfc1_b, const float* vfc2_w, const float* vfc2_b,
                                 float* logits, float* value, int batch, int H, int W, int PC, int VC, int A, int HID, int no_mfma,
                                 const unsigned* batch_dev, void* stream);

// x: trunk output in layout W; hw/hb: [PC + VC][64] / [PC + VC] head convolutions (policy rows first); pfc_w: [PC*H*W][A]
// (transposed Linear weight), vfc1_w: [VC*H*W][HID], vfc2_w: [HID]; logits: [batch][A], value: [batch].  -1: shape not covered.
extern "C" int sprl_tail_heads_fc(const float* x, const float* hw, const float* hb, const float* pfc_w, const float* pfc_b,
                                  const float* vfc1_w, const float* vfc1_b, const float* vfc2_w, const float* vfc2_b,
                                  float* logits, float* value, int batch, int H, int W, int PC, int VC, int A, int HID,
                                  const unsigned* batch_dev, void* stream) {
    return sprl_tail_fc(x, nullptr, hw, hb, pfc_w, pfc_b, vfc1_w, vfc1_b, vfc2_w, vfc2_b, logits, value, batch, H, W, PC, VC, A, HID,
                        batch_dev, stream);
}

// maps_in != null: FC layers only, on head maps [batch][(PC + VC) * H * W] produced by sprl_wino_conv64_heads
extern "C" int sprl_tail_fc(const float* x, const float* maps_in, const float* hw, const float* hb, const float* pfc_w,
                            const float* pfc_b, const float* vfc1_w, const float* vfc1_b, const float* vfc2_w, const float* vfc2_b,
                            float* logits, float* value, int batch, int H, int W, int PC, int VC, int A, int HID,
                            const unsigned* batch_dev, void* stream) {
    return sprl_tail_fc_form(x, maps_in, hw, hb, pfc_w, pfc_b, vfc1_w, vfc1_b, vfc2_w, vfc2_b, logits, value, batch, H, W, PC, VC, A, HID, 0,
                             batch_dev, stream);
}

// LDS the MFMA FC tail needs for a shape (bytes); the launcher raises the kernel's dynamic-LDS limit to TFC_LDS_LIMIT on every
// device it launches on and refuses shapes above it
constexpr size_t TFC_LDS_LIMIT = 128 * 1024;
extern "C" long long sprl_tail_fc_lds_bytes(int PIN, int VIN) {
    return (long long)(4 * ((PIN + 3) / 4) * TFC_LDW + 4 * ((VIN + 3) / 4) * TFC_LDW + TFC_NB * TFC_LDM + TFC_LDW + 128) * (long long)sizeof(float);
}
extern "C" long long sprl_tail_fc_lds_limit(void) { return (long long)TFC_LDS_LIMIT; }

// no_mfma != 0: the scalar FC stage of round 2 (lab variant)
extern "C" int sprl_tail_fc_form(const float* x, const float* maps_in, const float* hw, const float* hb, const float* pfc_w,
                                 const float* pfc_b, const float* vfc1_w, const float* vfc1_b, const float* vfc2_w, const float* vfc2_b,
                                 float* logits, float* value, int batch, int H, int W, int PC, int VC, int A, int HID, int no_mfma,
                                 const unsigned* batch_dev, void* stream) {
    if (batch <= 0) return 0;
    if (H > 8 || W > 8 || PC * H * W > TAIL_MAXIN || A > TAIL_MAXA || HID > TAIL_HID || VC * H * W > 64) return -1;
    const dim3 grid((unsigned)((batch + TAIL_NB - 1) / TAIL_NB)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (maps_in && !no_mfma) {                         // FC layers only: on the matrix cores, 64 boards per workgroup
        const int PIN = PC * H * W, VIN = VC * H * W;
        const size_t lds = (size_t)sprl_tail_fc_lds_bytes(PIN, VIN);
        if (PIN + 4 * ((VIN + 3) / 4) > TFC_LDM || 4 * ((PIN + 3) / 4) > TFC_LDM) return -1;      // (the padded contraction stays inside a map row)
        if (lds > TFC_LDS_LIMIT) return -1;
        // more than the 64 KB a kernel gets by default.  The attribute is PER DEVICE: one flag per device, set under a
        // mutex - engines on several devices, or several host threads (--populations), all find it set for their device.
        {
            static std::mutex mu;
            static bool attr_set[64] = { false };
            int dev = 0;
            if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -2;
            std::lock_guard<std::mutex> lock(mu);
            if (!attr_set[dev]) {
                if (hipFuncSetAttribute((const void*)tail_fc_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TFC_LDS_LIMIT) != hipSuccess)
                    return -2;
                attr_set[dev] = true;
            }
        }
        hipLaunchKernelGGL(tail_fc_mfma_kernel, dim3((unsigned)((batch + TFC_NB - 1) / TFC_NB)), block, lds, st, maps_in, pfc_w, pfc_b,
                           vfc1_w, vfc1_b, vfc2_w, vfc2_b, logits, value, batch, PIN, VIN, A, HID, batch_dev);
        return hipGetLastError() == hipSuccess ? 0 : -2;
    }
    if (PC == 2 && VC == 1)
        hipLaunchKernelGGL((tail_kernel<2, 1>), grid, block, 0, st, x, hw, hb, pfc_w, pfc_b, vfc1_w, vfc1_b, vfc2_w, vfc2_b, logits,
                           value, batch, H, W, A, HID, batch_dev, maps_in);
    else return -1;
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// x: trunk output NCHW [batch][64][H*W]; hw/hb: [PC + VC][64] / [PC + VC] head convolutions (policy rows first); pfc_w: [PC*H*W][A]
// (transposed Linear weight), vfc1_w: [VC*H*W][HID], vfc2_w: [HID]; pmaps: scratch [batch][PC*H*W]; logits: [batch][A], value:
// [batch]; batch_dev: optional device pointer to the real board count (<= batch).  -1: shape not covered (the caller falls back
// to the library GEMMs).
static int tail_any_board(const float* x, const float* hw, const float* hb, const float* pfc_w, const float* pfc_b, const float* vfc1_w,
                          const float* vfc1_b, const float* vfc2_w, const float* vfc2_b, float* pmaps, float* logits, float* value,
                          int batch, int H, int W, int PC, int VC, int A, int HID, const unsigned* batch_dev, int tile_m, void* stream) {
    if (batch <= 0) return 0;
    const int OC = PC + VC, HW = H * W, PIN = PC * HW;
    if (OC != 3 || PC != 2 || HID > 64) return -1;
    const bool big = OC * HW > 512;                    // 19x19: 1083 floats of maps per board -> 8 boards per workgroup
    const int NB = big ? 8 : 16;
    const size_t lds_a = (size_t)(OC * 64 + OC + NB * OC * HW + NB * 64 + 4 * NB * 64) * sizeof(float);
    const size_t lds_b = (size_t)(NB * PIN + NB * 128) * sizeof(float);
    if (lds_a > 64 * 1024 || lds_b > 64 * 1024) return -1;
    const dim3 grid_a((unsigned)((batch + NB - 1) / NB)), grid_b((unsigned)((batch + NB - 1) / NB), (unsigned)((A + 127) / 128)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (big) {
        hipLaunchKernelGGL((tail_heads_value_kernel<3, 8>), grid_a, block, lds_a, st, x, hw, hb, vfc1_w, vfc1_b, vfc2_w, vfc2_b, pmaps, value,
                           batch, HW, PC, HID, batch_dev, tile_m, W);
        hipLaunchKernelGGL((policy_fc_kernel<8>), grid_b, block, lds_b, st, pmaps, pfc_w, pfc_b, logits, batch, PIN, A, batch_dev);
    } else {
        hipLaunchKernelGGL((tail_heads_value_kernel<3, 16>), grid_a, block, lds_a, st, x, hw, hb, vfc1_w, vfc1_b, vfc2_w, vfc2_b, pmaps, value,
                           batch, HW, PC, HID, batch_dev, tile_m, W);
        hipLaunchKernelGGL((policy_fc_kernel<16>), grid_b, block, lds_b, st, pmaps, pfc_w, pfc_b, logits, batch, PIN, A, batch_dev);
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int sprl_tail_nchw(const float* x, const float* hw, const float* hb, const float* pfc_w, const float* pfc_b,
                              const float* vfc1_w, const float* vfc1_b, const float* vfc2_w, const float* vfc2_b, float* pmaps,
                              float* logits, float* value, int batch, int H, int W, int PC, int VC, int A, int HID,
                              const unsigned* batch_dev, void* stream) {
    return tail_any_board(x, hw, hb, pfc_w, pfc_b, vfc1_w, vfc1_b, vfc2_w, vfc2_b, pmaps, logits, value, batch, H, W, PC, VC, A, HID,
                          batch_dev, 0, stream);
}
// the same tail on a trunk output in layout T (tile = 3 or 4)
extern "C" int sprl_tail_t(const float* x, const float* hw, const float* hb, const float* pfc_w, const float* pfc_b,
                           const float* vfc1_w, const float* vfc1_b, const float* vfc2_w, const float* vfc2_b, float* pmaps,
                           float* logits, float* value, int batch, int H, int W, int PC, int VC, int A, int HID, int tile,
                           const unsigned* batch_dev, void* stream) {
    if (tile != 3 && tile != 4) return -1;
    return tail_any_board(x, hw, hb, pfc_w, pfc_b, vfc1_w, vfc1_b, vfc2_w, vfc2_b, pmaps, logits, value, batch, H, W, PC, VC, A, HID,
                          batch_dev, tile, stream);
}
