// Experiment bench for the trunk convolution (diagnostic, never shipped): variants of the Winograd/MFMA kernel timed in
// interleaved rounds in ONE process on random data, each checked against a float64 direct convolution on the host.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/wino_lab tools/wino_lab.hip && tools/wino_lab [boards] [rounds]
// the variants file carries its own copies of the product's entry points: rename them, the real ones are linked in
#define sprl_wino_weight_layout var_wino_weight_layout
#define sprl_wino_conv64_dev var_wino_conv64_dev
#define sprl_wino_conv64_heads var_wino_conv64_heads
#define sprl_wino_conv64_nchw var_wino_conv64_nchw
#define sprl_wino_conv64 var_wino_conv64
#include "wino_variants.hip"   // the experimental variants (v2 = round-1 kernel, v4 = flag-switched experiments); the product kernel
                               // (sprl_amd/csrc/cnn_wino.hip) is variant v4<8192>: filters as U36 through a register ring

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

namespace {

#undef sprl_wino_weight_layout
#undef sprl_wino_conv64_dev
#undef sprl_wino_conv64_heads
#undef sprl_wino_conv64_nchw
#undef sprl_wino_conv64
extern "C" int sprl_wino_conv64(const float* x, const float* u, const float* scale, const float* shift, const float* res, float* y,
                                int batch, int H, int W, int relu, void* stream);

struct Variant {
    const char* name;
    void (*launch)(const float* x, const float* w, const float* sc, const float* sh, const float* res, float* y, int B);
    int wkind;   // which weight buffer the variant reads: 0 = U36 (product layout), 1 = G9
};

const float *g_u36 = nullptr, *g_g9 = nullptr;

// U = G g G^T in double, product layout U4[p / 4][s][kb][lane][p % 4] (as torch_eval.cpp: wino_transform)
void sprl_wino_transform_weights_host(const float* g, float* up) {
    static const double G[6][3] = { { 1.0 / 4, 0, 0 },         { -1.0 / 6, -1.0 / 6, -1.0 / 6 }, { -1.0 / 6, 1.0 / 6, -1.0 / 6 },
                                    { 1.0 / 24, 1.0 / 12, 1.0 / 6 }, { 1.0 / 24, -1.0 / 12, 1.0 / 6 }, { 0, 0, 1 } };
    for (int k = 0; k < 64; ++k)
        for (int c = 0; c < 64; ++c) {
            const float* gk = g + ((size_t)k * 64 + c) * 9;
            double t[6][3];
            for (int a = 0; a < 6; ++a)
                for (int j = 0; j < 3; ++j) t[a][j] = G[a][0] * gk[j] + G[a][1] * gk[3 + j] + G[a][2] * gk[6 + j];
            for (int a = 0; a < 6; ++a)
                for (int b = 0; b < 6; ++b) {
                    const double v = t[a][0] * G[b][0] + t[a][1] * G[b][1] + t[a][2] * G[b][2];
                    const int p = a * 6 + b, s = 4 * (c >> 4) + (c & 3), kb = k >> 4, lane = ((c >> 2) & 3) * 16 + (k & 15);
                    up[((((size_t)(p >> 2) * 16 + s) * 4 + kb) * 64 + lane) * 4 + (p & 3)] = (float)v;
                }
        }
}
// U' = G' g G'^T (integer G', the row scales live in the V transform), product layout U4[p / 4][s][kb][lane][p % 4]
void pack_u36_scaled_host(const float* g, float* up) {
    static const double G[6][3] = { { 1, 0, 0 }, { 1, 1, 1 }, { 1, -1, 1 }, { 1, 2, 4 }, { 1, -2, 4 }, { 0, 0, 1 } };
    for (int k = 0; k < 64; ++k)
        for (int c = 0; c < 64; ++c) {
            const float* gk = g + ((size_t)k * 64 + c) * 9;
            double t[6][3];
            for (int a = 0; a < 6; ++a)
                for (int j = 0; j < 3; ++j) t[a][j] = G[a][0] * gk[j] + G[a][1] * gk[3 + j] + G[a][2] * gk[6 + j];
            for (int a = 0; a < 6; ++a)
                for (int b = 0; b < 6; ++b) {
                    const double v = t[a][0] * G[b][0] + t[a][1] * G[b][1] + t[a][2] * G[b][2];
                    const int p = a * 6 + b, s = 4 * (c >> 4) + (c & 3), kb = k >> 4, lane = ((c >> 2) & 3) * 16 + (k & 15);
                    up[((((size_t)(p >> 2) * 16 + s) * 4 + kb) * 64 + lane) * 4 + (p & 3)] = (float)v;
                }
        }
}
// T' = G' g: T4[s][kb][row 6][lane][4] (three values + one pad float per row)
void pack_t18_host(const float* g, float* out) {
    static const double G[6][3] = { { 1, 0, 0 }, { 1, 1, 1 }, { 1, -1, 1 }, { 1, 2, 4 }, { 1, -2, 4 }, { 0, 0, 1 } };
    for (int k = 0; k < 64; ++k)
        for (int c = 0; c < 64; ++c) {
            const float* gk = g + ((size_t)k * 64 + c) * 9;
            const int s = 4 * (c >> 4) + (c & 3), kb = k >> 4, lane = ((c >> 2) & 3) * 16 + (k & 15);
            for (int a = 0; a < 6; ++a)
                for (int j = 0; j < 4; ++j)
                    out[((((size_t)s * 4 + kb) * 6 + a) * 64 + lane) * 4 + j] =
                        j < 3 ? (float)(G[a][0] * gk[j] + G[a][1] * gk[3 + j] + G[a][2] * gk[6 + j]) : 0.0f;
        }
}
// the 3x3 filters in A-operand lane order: G8[s][kb][2][lane][4] (filter taps 0..7), then G1[s][kb][lane] (tap 8)
void sprl_wino_pack_g9_host(const float* g, float* out) {
    for (int k = 0; k < 64; ++k)
        for (int c = 0; c < 64; ++c) {
            const float* gk = g + ((size_t)k * 64 + c) * 9;
            const int s = 4 * (c >> 4) + (c & 3), kb = k >> 4, lane = ((c >> 2) & 3) * 16 + (k & 15);
            for (int e = 0; e < 8; ++e) out[((((size_t)s * 4 + kb) * 2 + (e >> 2)) * 64 + lane) * 4 + (e & 3)] = gk[e];
            out[(size_t)16 * 4 * 2 * 64 * 4 + ((size_t)s * 4 + kb) * 64 + lane] = gk[8];
        }
}

template <int FLAGS>
void launch_v4(const float* x, const float* w, const float* sc, const float* sh, const float* res, float* y, int B) {
    sprl_wino_conv64_v4_launch<FLAGS>(x, w, sc, sh, res, y, B, 1, nullptr, nullptr);
}
void launch_v2(const float* x, const float* w, const float* sc, const float* sh, const float* res, float* y, int B) {
    const dim3 grid((unsigned)((B + NIMG2 - 1) / NIMG2)), block(NTHR2);
    hipLaunchKernelGGL((wino_conv64_v2_kernel<8, 8>), grid, block, 0, 0, x, w, sc, sh, res, y, B, 1, 0, nullptr, TailArgs{});
}

// layout W index of (board n, channel k, row, col)
size_t widx(int n, int k, int row, int col) {
    const int g = 4 * (k >> 4) + (k & 3), cs = (k >> 2) & 3, tile = 2 * (row >> 2) + (col >> 2), i = row & 3, j = col & 3;
    return (size_t)n * 4096 + (size_t)g * 256 + (size_t)i * 64 + (size_t)cs * 16 + (size_t)tile * 4 + (size_t)j;
}

// probe: buffer-descriptor copy with the kernel's addressing (scalar offset per workgroup + per-lane offset)
__global__ void __launch_bounds__(256) probe_copy(const float* x, float* y, int batch) {
    const unsigned bytes = (unsigned)batch * 16384u;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)y, 0, bytes, 0x00020000);
    const int n0 = (int)blockIdx.x * 4;
    for (int q = 0; q < 16; ++q) {
        const int voff = (int)threadIdx.x * 16;
        const u4 v = __builtin_amdgcn_raw_buffer_load_b128(rx, voff, n0 * 16384 + q * 4096, 2);
        __builtin_amdgcn_raw_buffer_store_b128(v, ry, voff, n0 * 16384 + q * 4096, 2);
    }
}

}  // namespace

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 14400;
    const int rounds = argc > 2 ? atoi(argv[2]) : 7;
    const size_t n = (size_t)B * 4096;
    std::mt19937 rng(12345);
    std::normal_distribution<float> nd(0.0f, 1.0f);
    std::vector<float> hx(n), hr(n), hw(64 * 64 * 9), hsc(64), hsh(64);
    for (auto& v : hx) v = nd(rng);
    for (auto& v : hr) v = nd(rng);
    for (auto& v : hw) v = nd(rng) * 0.06f;
    for (auto& v : hsc) v = 0.5f + (float)(rng() % 1000) / 1000.0f;
    for (auto& v : hsh) v = nd(rng) * 0.3f;
    std::vector<float> hu(36 * 64 * 64), hg(G9_FLOATS), hus(36 * 64 * 64), ht(24 * 64 * 64);
    sprl_wino_transform_weights_host(hw.data(), hu.data());
    sprl_wino_pack_g9_host(hw.data(), hg.data());
    pack_u36_scaled_host(hw.data(), hus.data());
    pack_t18_host(hw.data(), ht.data());

    float *x, *y, *r, *u, *g9, *sc, *sh, *us, *t18;
    hipMalloc(&us, hus.size() * 4); hipMalloc(&t18, ht.size() * 4);
    hipMemcpy(us, hus.data(), hus.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(t18, ht.data(), ht.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&x, n * 4); hipMalloc(&y, n * 4); hipMalloc(&r, n * 4);
    hipMalloc(&u, hu.size() * 4); hipMalloc(&g9, hg.size() * 4); hipMalloc(&sc, 256); hipMalloc(&sh, 256);
    hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(r, hr.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(u, hu.data(), hu.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(g9, hg.data(), hg.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(sc, hsc.data(), 256, hipMemcpyHostToDevice);
    hipMemcpy(sh, hsh.data(), 256, hipMemcpyHostToDevice);
    g_u36 = u; g_g9 = g9;

    // the product kernel itself (sprl_amd/csrc/cnn_wino.hip compiled into this binary).  A persistent form of it (grid = resident
    // workgroups, next group's first chunks and filter quads requested in the second half of the output stage) was measured here
    // in round 2 and dropped: 328 us against 219 us at 13 492 boards (profiles/r02d_wino_lab_persistent.log) - the two
    // workgroups of a CU start together and stay in phase, so the output stage of one no longer overlaps the MFMA phases of the
    // other, and keeping the per-lane indices alive over the group loop spills ~130 registers.
    std::vector<Variant> vs = {
        { "PRODUCT cnn_wino.hip", [](const float* x, const float* w, const float* sc, const float* sh, const float* r, float* y, int B) {
             sprl_wino_conv64(x, w, sc, sh, r, y, B, 8, 8, 1, nullptr); }, 0 },
        { "v2 product (U36 from L2)", launch_v2, 0 },
        { "v4 G9 scaled + row-ahead U + b128", launch_v4<8 + 16 + 2048>, 1 },
        { "v4 U36' ring (scaled), b32 V", launch_v4<16 + 8192>, 2 },
        { "v4 U36 ring (unscaled), b32 V", launch_v4<8192>, 0 },
        { "v4 U36 ring (unscaled), b128 V", launch_v4<8 + 8192>, 0 },
    };

    // float64 reference for boards 0..3 and the last board (direct 3x3 convolution, padding 1, + scale/shift + residual + ReLU)
    const int check_boards[5] = { 0, 1, 2, 3, B - 1 };
    std::vector<double> want(5 * 64 * 64);
    for (int cb = 0; cb < 5; ++cb) {
        const int nb = check_boards[cb];
        for (int k = 0; k < 64; ++k)
            for (int row = 0; row < 8; ++row)
                for (int col = 0; col < 8; ++col) {
                    double s = 0.0;
                    for (int c = 0; c < 64; ++c)
                        for (int dy = 0; dy < 3; ++dy)
                            for (int dx = 0; dx < 3; ++dx) {
                                const int rr = row + dy - 1, cc = col + dx - 1;
                                if (rr < 0 || rr > 7 || cc < 0 || cc > 7) continue;
                                s += (double)hw[((size_t)k * 64 + c) * 9 + dy * 3 + dx] * (double)hx[widx(nb, c, rr, cc)];
                            }
                    s = s * hsc[k] + hsh[k] + hr[widx(nb, k, row, col)];
                    want[((size_t)cb * 64 + k) * 64 + row * 8 + col] = s > 0.0 ? s : 0.0;
                }
    }

    std::vector<float> hy(n), hy0;
    printf("boards %d, rounds %d\n", B, rounds);
    {
        hipMemset(y, 0xff, n * 4);
        hipLaunchKernelGGL(probe_copy, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, 0, x, y, B);
        hipDeviceSynchronize();
        hipMemcpy(hy.data(), y, n * 4, hipMemcpyDeviceToHost);
        long bad = 0;
        for (size_t i = 0; i < n; ++i) bad += !(hy[i] == hx[i]);
        printf("  probe: buffer copy with scalar offsets: %ld of %zu elements differ\n", bad, n);
    }
    for (size_t vi = 0; vi < vs.size(); ++vi) {
        hipMemset(y, 0xff, n * 4);
        vs[vi].launch(x, vs[vi].wkind == 3 ? t18 : vs[vi].wkind == 2 ? us : vs[vi].wkind ? g9 : u, sc, sh, r, y, B);
        if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", vs[vi].name); return 1; }
        hipMemcpy(hy.data(), y, n * 4, hipMemcpyDeviceToHost);
        double err = 0.0;
        long nan_ref = 0, nan_all = 0, bad_all = 0;
        int shown = 0;
        long hist[8][4] = {};
        for (int cb = 0; cb < 5; ++cb)
            for (int k = 0; k < 64; ++k)
                for (int cell = 0; cell < 64; ++cell) {
                    const double d = std::fabs((double)hy[widx(check_boards[cb], k, cell >> 3, cell & 7)] - want[((size_t)cb * 64 + k) * 64 + cell]);
                    if (std::isnan(d)) ++nan_ref;
                    else if (d > err) err = d;
                    if (!(d < 1e-3) && shown < 6 && vi > 0) {
                        ++shown;
                        printf("    mismatch board %d k %d row %d col %d: got %g want %g\n", check_boards[cb], k, cell >> 3, cell & 7,
                               (double)hy[widx(check_boards[cb], k, cell >> 3, cell & 7)], want[((size_t)cb * 64 + k) * 64 + cell]);
                    }
                }
        double dv2 = 0.0;
        if (vi == 0) hy0 = hy;
        else
            for (size_t i = 0; i < n; ++i) {
                const double d = std::fabs((double)hy[i] - (double)hy0[i]);
                if (std::isnan(d)) ++nan_all;
                else {
                    if (d > dv2) dv2 = d;
                    if (d > 1e-3) {
                        ++bad_all;
                        // layout W: [n][g][i][cs][tile][j]
                        const size_t w = i & 4095;
                        ++hist[0][(i >> 12) & 3]; ++hist[1][w & 3]; ++hist[2][(w >> 2) & 3]; ++hist[3][(w >> 4) & 3];
                        ++hist[4][(w >> 6) & 3]; ++hist[5][(w >> 8) & 3]; ++hist[6][(w >> 10) & 3];
                        ++hist[7][((i >> 12) / 4) % 4];
                    }
                }
            }
        if (bad_all) {
            const char* names[8] = { "board%4", "j", "tile", "cs", "i", "g&3 (=r)", "g>>2 (=kb)", "group%4" };
            for (int h = 0; h < 8; ++h) printf("      mismatches by %-10s %ld %ld %ld %ld\n", names[h], hist[h][0], hist[h][1], hist[h][2], hist[h][3]);
        }
        printf("  check %-34s max|err| vs f64 direct (5 boards) %.3e (NaN %ld)   vs v2 (all boards): max|diff| %.3e, >1e-3: %ld, NaN: %ld\n",
               vs[vi].name, err, nan_ref, dv2, bad_all, nan_all);
    }

    {   // where a workgroup's time goes: shader-clock stamps of wave 0 (diagnostic build of the same kernel)
        const int ngroups = (B + 3) / 4;
        unsigned long long* st;
        hipMalloc(&st, (size_t)ngroups * 16 * 8);
        hipMemset(st, 0, (size_t)ngroups * 16 * 8);
        for (int rep = 0; rep < 3; ++rep) sprl_wino_conv64_v4_launch<32 + 8192>(x, u, sc, sh, r, y, B, 1, nullptr, nullptr, (float*)st);
        hipDeviceSynchronize();
        std::vector<unsigned long long> hs((size_t)ngroups * 16);
        hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost);
        const char* names[11] = { "prologue", "phase0", "phase1", "phase2", "phase3", "phase4", "phase5", "phase6", "phase7", "output", "total" };
        printf("  stamps (shader cycles per workgroup, wave 0; %d workgroups): median / p10 / p90\n", ngroups);
        for (int k = 0; k < 11; ++k) {
            std::vector<double> v;
            for (int gi = 0; gi < ngroups; ++gi) {
                const unsigned long long* q = &hs[(size_t)gi * 16];
                v.push_back(k < 10 ? (double)(q[k + 1] - q[k]) : (double)(q[10] - q[0]));
            }
            std::sort(v.begin(), v.end());
            printf("    %-9s %9.0f %9.0f %9.0f\n", names[k], v[v.size() / 2], v[v.size() / 10], v[v.size() * 9 / 10]);
        }
        unsigned long long tmin = ~0ull, tmax = 0;
        for (int gi = 0; gi < ngroups; ++gi) { tmin = std::min(tmin, hs[(size_t)gi * 16]); tmax = std::max(tmax, hs[(size_t)gi * 16 + 10]); }
        printf("    first start -> last end: %.0f cycles\n", (double)(tmax - tmin));
        hipFree(st);
    }

    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<std::vector<float>> t(vs.size());
    const int iters = 10;
    for (int rd = 0; rd < rounds + 1; ++rd)
        for (size_t vi = 0; vi < vs.size(); ++vi) {
            hipEventRecord(e0, 0);
            for (int i = 0; i < iters; ++i) vs[vi].launch(x, vs[vi].wkind == 3 ? t18 : vs[vi].wkind == 2 ? us : vs[vi].wkind ? g9 : u, sc, sh, r, y, B);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            if (rd > 0) t[vi].push_back(ms * 1000.0f / iters);
        }
    const double flop = (double)B * 1179648.0;
    for (size_t vi = 0; vi < vs.size(); ++vi) {
        std::sort(t[vi].begin(), t[vi].end());
        const double med = t[vi][t[vi].size() / 2], mn = t[vi][0];
        printf("%-36s median %7.1f us  min %7.1f us   %5.1f TFLOP/s (Winograd domain) = %.3f of 157.3\n", vs[vi].name, med, mn,
               flop / med / 1e6, flop / med / 1e6 / 157.3);
    }
    return 0;
}
