// Ablation timing of the Winograd trunk convolution (diagnostic): which part of the kernel costs what.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/wino_ablate tools/wino_ablate.hip
#include "../sprl_amd/csrc/cnn_wino.hip"

#include <cstdio>
#include <vector>

template <int ABL>
static float run(const float* x, const float* u, const float* sc, const float* sh, const float* res, float* y, int B, int iters) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    dim3 grid((B + NIMG - 1) / NIMG), block(NTHR);
    for (int i = 0; i < 12; ++i) hipLaunchKernelGGL((wino_conv64_kernel<8, 8, ABL>), grid, block, 0, 0, x, u, sc, sh, res, y, B, 1, nullptr);
    hipEventRecord(a, 0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((wino_conv64_kernel<8, 8, ABL>), grid, block, 0, 0, x, u, sc, sh, res, y, B, 1, nullptr);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms * 1000.0f / iters;
}

template <int ABL>
static float run2(const float* x, const float* u, const float* sc, const float* sh, const float* res, float* y, int B, int iters, int stagger) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    dim3 grid((B + NIMG2 - 1) / NIMG2), block(NTHR2);
    for (int i = 0; i < 12; ++i) hipLaunchKernelGGL((wino_conv64_v2_kernel<8, 8, ABL>), grid, block, 0, 0, x, u, sc, sh, res, y, B, 1, stagger, nullptr, TailArgs{});
    hipEventRecord(a, 0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((wino_conv64_v2_kernel<8, 8, ABL>), grid, block, 0, 0, x, u, sc, sh, res, y, B, 1, stagger, nullptr, TailArgs{});
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms * 1000.0f / iters;
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 16384;
    float *x, *y, *r, *u, *sc, *sh;
    const size_t n = (size_t)B * 64 * 64;
    hipMalloc(&x, n * 4); hipMalloc(&y, n * 4); hipMalloc(&r, n * 4);
    hipMalloc(&u, 36 * 64 * 64 * 4); hipMalloc(&sc, 256); hipMalloc(&sh, 256);
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.0f - 0.5f;
    hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(r, h.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(u, h.data(), 36 * 64 * 64 * 4, hipMemcpyHostToDevice);
    hipMemcpy(sc, h.data(), 256, hipMemcpyHostToDevice);
    hipMemcpy(sh, h.data(), 256, hipMemcpyHostToDevice);
    printf("batch %d, us per launch\n", B);
    for (int sg = 0; sg <= 6; sg += 3) printf("v2 full, stagger %d            %8.1f\n", sg, run2<0>(x, u, sc, sh, r, y, B, 20, sg));
    printf("v2 no residual               %8.1f\n", run2<0>(x, u, sc, sh, nullptr, y, B, 20, 0));
    printf("v2 no stores                 %8.1f\n", run2<64>(x, u, sc, sh, r, y, B, 20, 0));
    printf("v2 no stores, no residual    %8.1f\n", run2<64>(x, u, sc, sh, nullptr, y, B, 20, 0));
    printf("v2 no output stage           %8.1f\n", run2<1>(x, u, sc, sh, r, y, B, 20, 0));
    printf("v2 no V production           %8.1f\n", run2<2>(x, u, sc, sh, r, y, B, 20, 0));
    printf("v2 no weight loads           %8.1f\n", run2<4>(x, u, sc, sh, r, y, B, 20, 0));
    printf("v2 weights from 8 KB footprint %8.1f\n", run2<32>(x, u, sc, sh, r, y, B, 20, 0));
    printf("v2 no output, tiny-footprint wts %6.1f\n", run2<33>(x, u, sc, sh, r, y, B, 20, 0));
    printf("v2 no output, no V, no wts   %8.1f\n", run2<7>(x, u, sc, sh, r, y, B, 20, 0));
    printf("v2 skeleton + MFMA           %8.1f\n", run2<23>(x, u, sc, sh, r, y, B, 20, 0));
    printf("full                         %8.1f\n", run<0>(x, u, sc, sh, r, y, B, 20));
    printf("no output stage              %8.1f\n", run<1>(x, u, sc, sh, r, y, B, 20));
    printf("no V production              %8.1f\n", run<2>(x, u, sc, sh, r, y, B, 20));
    printf("no weight loads              %8.1f\n", run<4>(x, u, sc, sh, r, y, B, 20));
    printf("no MFMA                      %8.1f\n", run<8>(x, u, sc, sh, r, y, B, 20));
    printf("no activation loads          %8.1f\n", run<16>(x, u, sc, sh, r, y, B, 20));
    printf("no output, no V              %8.1f\n", run<3>(x, u, sc, sh, r, y, B, 20));
    printf("no output, no V, no weights  %8.1f\n", run<7>(x, u, sc, sh, r, y, B, 20));
    printf("no output, no weights        %8.1f\n", run<5>(x, u, sc, sh, r, y, B, 20));
    printf("only loop skeleton + MFMA    %8.1f\n", run<23>(x, u, sc, sh, r, y, B, 20));
    return 0;
}
