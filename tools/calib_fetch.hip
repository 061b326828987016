// Calibration for the FETCH_SIZE / WRITE_SIZE PMC counters on the tree kernel's own access pattern
// (MI355X_MICROARCH.md §HBM: "calibrate on a known byte count in your own access pattern").
// One wavefront per block walks `nodes_per_wave` pseudo-random 1 KiB nodes of a large pool exactly like
// step_kernel's load_node(): lane l reads N[l], W[l], P[l] (3 x 256 B rows, one dword per lane), child[l]
// (128 B, one ushort per lane) and the 64 B header (same address in every lane), then lane l writes N[l] and W[l]
// of the node back (2 x 4 B per lane).  Known bytes per node: 960 read, 512 written.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/calib_fetch tools/calib_fetch.hip
// Run:   rocprofv3 --pmc FETCH_SIZE -- tools/calib_fetch   (and WRITE_SIZE in a separate pass)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void __launch_bounds__(64) walk(unsigned char* pool, size_t num_nodes, int nodes_per_wave, float* sink) {
    const int l = threadIdx.x;
    unsigned long long x = 0x9E3779B97F4A7C15ull * (blockIdx.x + 1);
    float acc = 0.0f;
    for (int i = 0; i < nodes_per_wave; ++i) {
        x = x * 6364136223846793005ull + 1442695040888963407ull;
        size_t idx = (size_t)((x >> 20) % num_nodes);
        unsigned char* np = pool + idx * 1024;
        float n = ((float*)np)[l], w = ((float*)(np + 256))[l], p = ((float*)(np + 512))[l];
        unsigned short ch = ((unsigned short*)(np + 768))[l];
        const uint4* h = (const uint4*)(np + 896);
        uint4 h0 = h[0], h1 = h[1], h2 = h[2], h3 = h[3];
        acc += n + w + p + (float)ch + (float)(h0.x ^ h1.y ^ h2.z ^ h3.w);
        ((float*)np)[l] = n + 1.0f;
        ((float*)(np + 256))[l] = w - 1.0f;
    }
    if (acc == 123.456f) sink[0] = acc;
}

int main(int argc, char** argv) {
    const size_t gib = argc > 1 ? (size_t)atol(argv[1]) : 64;      // pool far larger than L2 + Infinity Cache
    const int waves = 4096, nodes_per_wave = 64, reps = 20;
    const size_t num_nodes = gib << 20;                             // gib GiB / 1 KiB
    unsigned char* pool;
    float* sink;
    if (hipMalloc(&pool, num_nodes * 1024) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return 1;
    hipMemset(pool, 0, num_nodes * 1024);
    hipDeviceSynchronize();
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(walk, dim3(waves), dim3(64), 0, 0, pool, num_nodes, nodes_per_wave, sink);
    hipDeviceSynchronize();
    printf("launches %d nodes_per_launch %d read_bytes_per_launch %zu write_bytes_per_launch %zu\n", reps,
           waves * nodes_per_wave, (size_t)waves * nodes_per_wave * 960, (size_t)waves * nodes_per_wave * 512);
    return 0;
}
