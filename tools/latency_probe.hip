// Dependent-load latency probe: every wavefront chases a pointer chain of 1 KiB "nodes" spread over a pool of the
// given size (like a tree descent: the next node's index comes out of the node just loaded), with as many waves
// in flight as the self-play kernel has games.  Prints ns per hop for several pool sizes / hot-set shapes.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/latency_probe tools/latency_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void __launch_bounds__(64) chase(const unsigned* __restrict__ pool, size_t stride_words, unsigned start_mod,
                                            int hops, unsigned* out) {
    const int l = threadIdx.x;
    unsigned idx = (blockIdx.x * 2654435761u) % start_mod;
    unsigned acc = 0;
    for (int i = 0; i < hops; ++i) {
        const unsigned* np = pool + (size_t)idx * stride_words;
        unsigned next = np[224];                 // "header" word, same address in every lane
        unsigned row = np[l] + np[64 + l] + np[128 + l];   // three coalesced rows
        acc += row;
        idx = next;
    }
    if (acc == 0xdeadbeefu) out[0] = acc;
    if (l == 0) out[1 + blockIdx.x] = idx;
}

int main(int argc, char** argv) {
    const int waves = argc > 1 ? atoi(argv[1]) : 4096;
    const int hops = 256;
    unsigned* out;
    hipMalloc(&out, (waves + 1) * 4);
    const size_t sizes_gib[] = { 1, 8, 32, 64, 128, 160 };
    for (size_t gib : sizes_gib) {
        const size_t nodes = gib << 20;
        unsigned* pool;
        if (hipMalloc(&pool, nodes * 1024) != hipSuccess) { printf("%zu GiB: alloc failed\n", gib); continue; }
        // two chain shapes: (a) uniformly random over the whole pool, (b) per-wave hot set of 2048 nodes (2 MiB)
        for (int shape = 0; shape < 2; ++shape) {
            std::vector<unsigned> next(nodes);
            unsigned long long x = 88172645463325252ull;
            const size_t region = shape == 0 ? nodes : 2048;
            for (size_t i = 0; i < nodes; ++i) {
                x ^= x << 13; x ^= x >> 7; x ^= x << 17;
                size_t base = shape == 0 ? 0 : (i / region) * region;
                next[i] = (unsigned)(base + (x % region));
            }
            // scatter the next indices into word 224 of each node
            std::vector<unsigned> node(256, 0);
            hipMemset(pool, 0, nodes * 1024);
            for (size_t i = 0; i < nodes; i += 1 << 16) {
                size_t n = nodes - i < (1u << 16) ? nodes - i : (1u << 16);
                std::vector<unsigned> buf(n * 256, 0);
                for (size_t k = 0; k < n; ++k) buf[k * 256 + 224] = next[i + k];
                hipMemcpy(pool + i * 256, buf.data(), n * 1024, hipMemcpyHostToDevice);
            }
            hipEvent_t a, b;
            hipEventCreate(&a); hipEventCreate(&b);
            hipLaunchKernelGGL(chase, dim3(waves), dim3(64), 0, 0, pool, (size_t)256, (unsigned)nodes, hops, out);
            hipDeviceSynchronize();
            hipEventRecord(a, 0);
            for (int r = 0; r < 5; ++r)
                hipLaunchKernelGGL(chase, dim3(waves), dim3(64), 0, 0, pool, (size_t)256, (unsigned)nodes, hops, out);
            hipEventRecord(b, 0);
            hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("pool %3zu GiB  %-28s waves %d : %.1f ns per dependent hop (kernel %.3f ms)\n", gib,
                   shape == 0 ? "uniform over pool" : "2 MiB hot set per start", waves, ms * 1e6 / 5 / hops, ms / 5);
            fflush(stdout);
        }
        hipFree(pool);
    }
    return 0;
}
