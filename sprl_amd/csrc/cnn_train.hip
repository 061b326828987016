// cnn_train.hip — what the TRAINER (sprl_amd/trainer_ops.py) needs to run the trunk convolutions of a training step
// on the hand-written Winograd / fp32-MFMA kernel of cnn_wino.hip instead of the library's: the forward convolution and the
// backward-data convolution of conv3x3(64 -> 64) are the same kernel (backward-data = the convolution of the output gradient with the
// filters transposed over the channels and rotated by 180 degrees).  The training graph around them (BatchNorm in training mode,
// ReLU, residual adds, the weight gradient) stays NCHW in the framework, so the tensors cross two cheap layout kernels:
//   NCHW [B][64][H][W]  <->  layout W [B][4096]  (cnn_wino.hip: x[n][g][i][cs][tile][j], cells off the board zero)
// and the filters - which change with every optimiser step - are brought to the Winograd domain ON THE DEVICE
// (U = G g G^T in double, the arithmetic and the A-operand layout of torch_eval.cpp: wino_transform).
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

// thread = one 16-byte piece of layout W: (board n, group g, row-in-tile i, channel slot cs, tile) -> columns 4 tx .. 4 tx + 3
__global__ void __launch_bounds__(256) nchw_to_w_kernel(const float* __restrict__ x, float* __restrict__ xw, int B, int H, int W) {
    const long long f = (long long)blockIdx.x * 256 + threadIdx.x;
    if (f >= (long long)B * 1024) return;
    const int n = (int)(f >> 10), rem = (int)(f & 1023);
    const int g = rem >> 6, i = (rem >> 4) & 3, cs = (rem >> 2) & 3, tile = rem & 3;
    const int k = 16 * (g >> 2) + 4 * cs + (g & 3), row = 4 * (tile >> 1) + i, col0 = 4 * (tile & 1);
    f4 v = { 0.0f, 0.0f, 0.0f, 0.0f };
    if (row < H) {
        const float* src = x + (((size_t)n * 64 + k) * H + row) * W + col0;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (col0 + j < W) v[j] = src[j];
    }
    *(f4*)(xw + f * 4) = v;
}

__global__ void __launch_bounds__(256) w_to_nchw_kernel(const float* __restrict__ xw, float* __restrict__ x, int B, int H, int W) {
    const long long f = (long long)blockIdx.x * 256 + threadIdx.x;
    if (f >= (long long)B * 1024) return;
    const int n = (int)(f >> 10), rem = (int)(f & 1023);
    const int g = rem >> 6, i = (rem >> 4) & 3, cs = (rem >> 2) & 3, tile = rem & 3;
    const int k = 16 * (g >> 2) + 4 * cs + (g & 3), row = 4 * (tile >> 1) + i, col0 = 4 * (tile & 1);
    if (row >= H) return;
    const f4 v = *(const f4*)(xw + f * 4);
    float* dst = x + (((size_t)n * 64 + k) * H + row) * W + col0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (col0 + j < W) dst[j] = v[j];
}

// thread = one (output channel ko, input channel ci) pair of the convolution the kernel will run.  flip_transpose = 0: the forward
// filters g[dy][dx] = w[ko][ci][dy][dx]; 1: the backward-data filters g[dy][dx] = w[ci][ko][2 - dy][2 - dx].
__global__ void __launch_bounds__(256) wino_weights_kernel(const float* __restrict__ w, float* __restrict__ u, int flip_transpose) {
    const int t = (int)blockIdx.x * 256 + (int)threadIdx.x;
    if (t >= 4096) return;
    const int ko = t >> 6, ci = t & 63;
    double g[9];
#pragma unroll
    for (int e = 0; e < 9; ++e) g[e] = flip_transpose ? (double)w[((size_t)ci * 64 + ko) * 9 + (8 - e)] : (double)w[((size_t)ko * 64 + ci) * 9 + e];
    const double G[6][3] = { { 1.0 / 4, 0, 0 },         { -1.0 / 6, -1.0 / 6, -1.0 / 6 }, { -1.0 / 6, 1.0 / 6, -1.0 / 6 },
                             { 1.0 / 24, 1.0 / 12, 1.0 / 6 }, { 1.0 / 24, -1.0 / 12, 1.0 / 6 }, { 0, 0, 1 } };
    double tm[6][3];
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int j = 0; j < 3; ++j) tm[a][j] = G[a][0] * g[j] + G[a][1] * g[3 + j] + G[a][2] * g[6 + j];
    // K-loop step s reads group s of layout W: input channel c = 16 (s >> 2) + 4 slot + (s & 3)
    const int s = 4 * (ci >> 4) + (ci & 3), slot = (ci >> 2) & 3, kb = ko >> 4, lane = slot * 16 + (ko & 15);
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            const double v = tm[a][0] * G[b][0] + tm[a][1] * G[b][1] + tm[a][2] * G[b][2];
            const int p = a * 6 + b;
            u[((((size_t)(p >> 2) * 16 + s) * 4 + kb) * 64 + lane) * 4 + (p & 3)] = (float)v;
        }
}

}  // namespace

extern "C" int sprl_train_nchw_to_w(const float* x, float* xw, int B, int H, int W, void* stream) {
    if (B <= 0) return 0;
    if (H < 1 || W < 1 || H > 8 || W > 8) return -1;
    hipLaunchKernelGGL(nchw_to_w_kernel, dim3((unsigned)(((long long)B * 1024 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, xw, B, H, W);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
extern "C" int sprl_train_w_to_nchw(const float* xw, float* x, int B, int H, int W, void* stream) {
    if (B <= 0) return 0;
    if (H < 1 || W < 1 || H > 8 || W > 8) return -1;
    hipLaunchKernelGGL(w_to_nchw_kernel, dim3((unsigned)(((long long)B * 1024 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, xw, x, B, H, W);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
// w: [64][64][3][3] on the device; u: 36 * 64 * 64 floats on the device in the layout sprl_wino_conv64 takes
extern "C" int sprl_train_wino_weights(const float* w, float* u, int flip_transpose, void* stream) {
    hipLaunchKernelGGL(wino_weights_kernel, dim3(16), dim3(256), 0, (hipStream_t)stream, w, u, flip_transpose);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
