// microbench_stream.hip -- what a read-modify-write stream of 1 GiB (256 x 2^20 floats: K3's one-step-per-launch case) reaches on this
// GPU with different launch shapes: floats per thread, nontemporal accesses, block size.
//   hipcc --offload-arch=gfx950 -O3 -o tools/microbench_stream tools/microbench_stream.hip && tools/microbench_stream
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int V, bool NT, bool GS>
__global__ void rmw(f4* __restrict__ x, long long nquads, float a) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (GS) {  // grid-stride: V quads in flight per thread per round
        for (; i < nquads; i += stride * V) {
            f4 v[V];
#pragma unroll
            for (int u = 0; u < V; ++u)
                if (i + u * stride < nquads) v[u] = NT ? __builtin_nontemporal_load(x + i + u * stride) : x[i + u * stride];
#pragma unroll
            for (int u = 0; u < V; ++u)
                if (i + u * stride < nquads) {
                    v[u] = v[u] * a + 1.0f;
                    if (NT) __builtin_nontemporal_store(v[u], x + i + u * stride);
                    else x[i + u * stride] = v[u];
                }
        }
    } else {  // one pass: thread t takes quads t, t + T, ... (V of them)
        f4 v[V];
#pragma unroll
        for (int u = 0; u < V; ++u)
            if (i + u * stride < nquads) v[u] = NT ? __builtin_nontemporal_load(x + i + u * stride) : x[i + u * stride];
#pragma unroll
        for (int u = 0; u < V; ++u)
            if (i + u * stride < nquads) {
                v[u] = v[u] * a + 1.0f;
                if (NT) __builtin_nontemporal_store(v[u], x + i + u * stride);
                else x[i + u * stride] = v[u];
            }
    }
}
template <int V, bool NT, bool GS>
static void run(const char* name, f4* d, long long nquads, int block, int grid_cap) {
    long long threads = (nquads + V - 1) / V;
    long long grid = (threads + block - 1) / block;
    if (GS && grid > grid_cap) grid = grid_cap;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0);
        rmw<V, NT, GS><<<(unsigned)grid, block>>>(d, nquads, 0.999f);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    printf("%-44s block %4d grid %8lld: %.1f us = %.2f TB/s\n", name, block, grid, best * 1e3, 2.0 * nquads * 16 / (best * 1e-3) / 1e12);
}
int main() {
    const long long nquads = (256ll << 20) / 4;
    f4* d;
    hipMalloc(&d, nquads * 16);
    hipMemset(d, 0, nquads * 16);
    run<1, false, false>("1 quad per thread", d, nquads, 256, 0);
    run<1, true, false>("1 quad per thread, nontemporal", d, nquads, 256, 0);
    run<2, false, false>("2 quads per thread", d, nquads, 256, 0);
    run<4, false, false>("4 quads per thread", d, nquads, 256, 0);
    run<4, true, false>("4 quads per thread, nontemporal", d, nquads, 256, 0);
    run<8, false, false>("8 quads per thread", d, nquads, 256, 0);
    run<4, false, false>("4 quads per thread", d, nquads, 1024, 0);
    run<4, false, true>("grid-stride 4 in flight, 2048 blocks", d, nquads, 256, 2048);
    run<4, false, true>("grid-stride 4 in flight, 8192 blocks", d, nquads, 256, 8192);
    run<4, true, true>("grid-stride 4 in flight nt, 4096 blocks", d, nquads, 256, 4096);
    run<8, false, true>("grid-stride 8 in flight, 2048 blocks", d, nquads, 256, 2048);
    run<2, false, true>("grid-stride 2 in flight, 4096 x 1024", d, nquads, 1024, 4096);
    // a plain device-to-device copy of the same bytes (read 1 GiB + write 1 GiB)
    f4* d2;
    hipMalloc(&d2, nquads * 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipMemcpyAsync(d2, d, nquads * 16, hipMemcpyDeviceToDevice, 0);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    printf("%-44s: %.1f us = %.2f TB/s\n", "hipMemcpy device to device (1 GiB)", best * 1e3, 2.0 * nquads * 16 / (best * 1e-3) / 1e12);
    return 0;
}
