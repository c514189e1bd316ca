// microbench3.hip -- Philox throughput vs occupancy and ILP (development aid)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <functional>
struct u32x4 { uint32_t x, y, z, w; };
__device__ __forceinline__ u32x4 philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return u32x4{c0, c1, c2, c3};
}
template <int ILP>
__global__ __launch_bounds__(256) void probe(uint32_t* out, int iters, uint32_t seed) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (int it = 0; it < iters; it += ILP) {
#pragma unroll
        for (int j = 0; j < ILP; ++j) {
            u32x4 r = philox(t, it + j + (acc & 1), seed, j, 0x1234567u, 0x89abcdefu);
            acc ^= r.x ^ r.y ^ r.z ^ r.w;
        }
    }
    out[t] = acc;
}
static float time_ms(std::function<void()> f) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    f(); (void)hipDeviceSynchronize(); float best = 1e30f;
    for (int r = 0; r < 5; ++r) { (void)hipEventRecord(a); f(); (void)hipEventRecord(b); (void)hipEventSynchronize(b); float ms; (void)hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; }
    return best;
}
int main() {
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    uint32_t* out; (void)hipMalloc(&out, (size_t)prop.multiProcessorCount * 8 * 256 * 4);
    const int iters = 2048;
    for (int wps : {1, 2, 3, 4, 6, 8}) {
        int blocks = prop.multiProcessorCount * wps;
        float m1 = time_ms([&] { probe<1><<<blocks, 256>>>(out, iters, 1); });
        float m2 = time_ms([&] { probe<2><<<blocks, 256>>>(out, iters, 1); });
        float m4 = time_ms([&] { probe<4><<<blocks, 256>>>(out, iters, 1); });
        double calls = (double)blocks * 256 * iters;
        auto cyc = [&](float ms) { return ms * 1e-3 * 2.4e9 / (iters * (double)wps); };  // cycles per wave-call per SIMD
        printf("waves/SIMD=%d  ILP1 %7.1f G/s (%5.0f cyc/call)  ILP2 %7.1f G/s (%5.0f)  ILP4 %7.1f G/s (%5.0f)\n", wps,
               calls / m1 / 1e6, cyc(m1), calls / m2 / 1e6, cyc(m2), calls / m4 / 1e6, cyc(m4));
    }
    return 0;
}
