// Probe (diagnostic): do fp32 MFMAs (v_mfma_f32_16x16x4_f32) and fp32 VALU instructions overlap on gfx950?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/mfma_valu_probe tools/mfma_valu_probe.hip && tools/mfma_valu_probe
// (a) two waves on one SIMD: MFMA-only wave beside a VALU-only wave, against each of them alone;
// (b) one wave: 1 MFMA followed by n independent VALU instructions per iteration, n = 0..12;
// (c) one wave: 1 MFMA whose A operand is produced by a VALU instruction d instructions earlier.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int ITERS = 4096;

// role 0: MFMA loop (8 independent accumulators), role 1: VALU loop (8 independent chains), role 2: idle
template <int ROLE_LO, int ROLE_HI>
__global__ void __launch_bounds__(512) pair_kernel(float* out, unsigned long long* cyc) {
    const int wave = threadIdx.x >> 6;
    const int role = wave < 4 ? ROLE_LO : ROLE_HI;
    f4 acc[8];
    float v[8];
    for (int i = 0; i < 8; ++i) {
        acc[i] = (f4){ 0.f, 0.f, 0.f, 0.f };
        v[i] = (float)threadIdx.x * 1e-3f + i;
    }
    const float a = (float)threadIdx.x * 1e-4f, b = 1.0001f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (role == 0) {
        for (int it = 0; it < ITERS; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        }
    } else if (role == 1) {
        for (int it = 0; it < ITERS; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], b, a);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + v[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

// one wave per SIMD: per iteration 8 MFMAs, each followed by NV independent VALU instructions
template <int NV>
__global__ void __launch_bounds__(256) mix_kernel(float* out, unsigned long long* cyc) {
    f4 acc[8];
    float v[12];
    for (int i = 0; i < 8; ++i) acc[i] = (f4){ 0.f, 0.f, 0.f, 0.f };
    for (int i = 0; i < 12; ++i) v[i] = (float)threadIdx.x * 1e-3f + i;
    const float a = (float)threadIdx.x * 1e-4f, b = 1.0001f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV; ++j) v[j] = __builtin_fmaf(v[j], b, a);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0];
    for (int i = 0; i < 12; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

// one wave per SIMD: the A operand of each MFMA is produced by a VALU instruction D independent VALU instructions earlier
template <int D>
__global__ void __launch_bounds__(256) dep_kernel(float* out, unsigned long long* cyc) {
    f4 acc[8];
    float v[12];
    for (int i = 0; i < 8; ++i) acc[i] = (f4){ 0.f, 0.f, 0.f, 0.f };
    for (int i = 0; i < 12; ++i) v[i] = (float)threadIdx.x * 1e-3f + i;
    float a = (float)threadIdx.x * 1e-4f;
    const float b = 1.0001f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            a = __builtin_fmaf(a, b, 1e-7f);                       // producer of the MFMA's A operand
#pragma unroll
            for (int j = 0; j < D; ++j) v[j] = __builtin_fmaf(v[j], b, 0.5f);
            __builtin_amdgcn_sched_barrier(0);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = a;
    for (int i = 0; i < 8; ++i) s += acc[i][0];
    for (int i = 0; i < 12; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

static double run(void (*launch)(float*, unsigned long long*), float* out, unsigned long long* cyc, int waves, int lo, int hi) {
    hipMemset(cyc, 0, 256 * 8 * 8);
    launch(out, cyc);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 8);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0;
    int n = 0;
    for (int b = 0; b < 256; ++b)
        for (int w = lo; w < hi; ++w) { s += (double)h[b * 8 + w]; ++n; }
    (void)waves;
    return s / n;
}

#define PAIR(LO, HI) [](float* o, unsigned long long* c) { hipLaunchKernelGGL((pair_kernel<LO, HI>), dim3(256), dim3(512), 0, 0, o, c); }
#define MIX(N) [](float* o, unsigned long long* c) { hipLaunchKernelGGL((mix_kernel<N>), dim3(256), dim3(256), 0, 0, o, c); }
#define DEP(N) [](float* o, unsigned long long* c) { hipLaunchKernelGGL((dep_kernel<N>), dim3(256), dim3(256), 0, 0, o, c); }

int main() {
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&cyc, 256 * 8 * 8);
    const double per_mfma = 1.0 / (ITERS * 8.0), per_valu = 1.0 / (ITERS * 32.0);
    printf("(a) two waves per SIMD, cycles per instruction of the measured wave\n");
    printf("    MFMA wave alone (partner idle)        %.2f cyc/MFMA\n", run(PAIR(0, 2), out, cyc, 8, 0, 4) * per_mfma);
    printf("    MFMA wave beside MFMA wave            %.2f cyc/MFMA\n", run(PAIR(0, 0), out, cyc, 8, 0, 4) * per_mfma);
    printf("    VALU wave alone (partner idle)        %.2f cyc/v_fma\n", run(PAIR(1, 2), out, cyc, 8, 0, 4) * per_valu);
    printf("    VALU wave beside VALU wave            %.2f cyc/v_fma\n", run(PAIR(1, 1), out, cyc, 8, 0, 4) * per_valu);
    printf("    MFMA wave beside VALU wave            %.2f cyc/MFMA\n", run(PAIR(0, 1), out, cyc, 8, 0, 4) * per_mfma);
    printf("    VALU wave beside MFMA wave            %.2f cyc/v_fma\n", run(PAIR(0, 1), out, cyc, 8, 4, 8) * per_valu);
    printf("(b) one wave per SIMD: MFMA followed by n independent v_fma, cycles per MFMA\n");
    printf("    n=0 %.1f  n=2 %.1f  n=4 %.1f  n=6 %.1f  n=8 %.1f  n=10 %.1f  n=12 %.1f\n", run(MIX(0), out, cyc, 4, 0, 4) * per_mfma,
           run(MIX(2), out, cyc, 4, 0, 4) * per_mfma, run(MIX(4), out, cyc, 4, 0, 4) * per_mfma, run(MIX(6), out, cyc, 4, 0, 4) * per_mfma,
           run(MIX(8), out, cyc, 4, 0, 4) * per_mfma, run(MIX(10), out, cyc, 4, 0, 4) * per_mfma, run(MIX(12), out, cyc, 4, 0, 4) * per_mfma);
    printf("(c) one wave per SIMD: A operand produced d VALU instructions before the MFMA, cycles per MFMA\n");
    printf("    d=0 %.1f  d=1 %.1f  d=2 %.1f  d=4 %.1f  d=6 %.1f\n", run(DEP(0), out, cyc, 4, 0, 4) * per_mfma, run(DEP(1), out, cyc, 4, 0, 4) * per_mfma,
           run(DEP(2), out, cyc, 4, 0, 4) * per_mfma, run(DEP(4), out, cyc, 4, 0, 4) * per_mfma, run(DEP(6), out, cyc, 4, 0, 4) * per_mfma);
    return 0;
}
