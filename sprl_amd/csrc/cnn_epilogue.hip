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
}  // namespace

extern "C" int sprl_bn_relu_inplace(float* x, const float* residual, const float* scale, const float* shift,
                                    int64_t numel, int channels, int hw, void* stream) {
    if (hw % 4 != 0 || numel % 4 != 0) return -1;
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
