// Issue cost of v_mad_u64_u32 against v_mul_hi_u32 + v_mul_lo_u32 on gfx950 (not product code).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define ITERS 2048
__global__ __launch_bounds__(256) void p_mad64(uint32_t* out, uint32_t seed) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t x[8];
    uint64_t r[8];
    for (int i = 0; i < 8; ++i) x[i] = t * (2 * i + 1) + seed + i;
    uint32_t b = t ^ 0x5bd1e995u;
    uint64_t z = 0;
    asm volatile("" : "+v"(z));
    uint32_t acc = 0;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(r[i]) : "v"(x[i]), "v"(b), "v"(z) : "vcc");
            x[i] = (uint32_t)r[i];
        }
    }
    for (int i = 0; i < 8; ++i) acc ^= (uint32_t)r[i] ^ (uint32_t)(r[i] >> 32);
    out[t] = acc;
}
__global__ __launch_bounds__(256) void p_mulhilo(uint32_t* out, uint32_t seed) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t a0 = t + seed, a1 = t * 3 + seed, a2 = t * 5 + 1, a3 = t * 7 + 2;
    uint32_t h0 = 0, h1 = 0, h2 = 0, h3 = 0;
    uint32_t b = t ^ 0x5bd1e995u;
    for (int it = 0; it < ITERS; ++it)
        asm volatile(
            "v_mul_hi_u32 %4, %0, %8\n v_mul_lo_u32 %0, %0, %8\n v_mul_hi_u32 %5, %1, %8\n v_mul_lo_u32 %1, %1, %8\n"
            "v_mul_hi_u32 %6, %2, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_hi_u32 %7, %3, %8\n v_mul_lo_u32 %3, %3, %8\n"
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3)
            : "v"(b));
    out[t] = a0 ^ a1 ^ a2 ^ a3 ^ h0 ^ h1 ^ h2 ^ h3;
}
template <class F>
static float time_ms(F f) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    f(); (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) { (void)hipEventRecord(a); f(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; }
    return best;
}
int main() {
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    uint32_t* out;
    for (int wpb : {8, 4, 2}) {
        int blocks = prop.multiProcessorCount * wpb;
        (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
        float m1 = time_ms([&] { p_mad64<<<blocks, 256>>>(out, 1); });
        float m2 = time_ms([&] { p_mulhilo<<<blocks, 256>>>(out, 1); });
        printf("%d waves/SIMD: mad_u64_u32 %.2f cyc per 64-bit product; mul_hi+mul_lo %.2f cyc per 64-bit product (@2.4GHz)\n", wpb,
               m1 * 1e-3 * 2.4e9 / (ITERS * 8.0 * wpb), m2 * 1e-3 * 2.4e9 / (ITERS * 4.0 * wpb));
        (void)hipFree(out);
    }
    return 0;
}
