// Hierarchical grid barrier (per-XCD counters and flags on separate 4 KiB pages) vs a flat counter (not product code).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define PAGE 1024  // unsigned per 4 KiB
// layout: cnt[g] at g*PAGE, gen[g] at (8+g)*PAGE, root at 16*PAGE
template <int FENCE, int NG>
__global__ __launch_bounds__(1024) void bar_loop(unsigned* bar, float* data, int rounds) {
    unsigned epoch = 0;
    const unsigned g = blockIdx.x % NG;
    const unsigned gsize = (gridDim.x - g + NG - 1) / NG;
    for (int r = 0; r < rounds; ++r) {
        data[(size_t)blockIdx.x * 1024 + threadIdx.x] += 1.0f;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        epoch += 1;
        if (threadIdx.x == 0) {
            if (FENCE) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            unsigned a = __hip_atomic_fetch_add(&bar[g * PAGE], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (a + 1 == epoch * gsize) {
                unsigned b = __hip_atomic_fetch_add(&bar[16 * PAGE], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (b + 1 == epoch * NG)
                    for (int q = 0; q < NG; ++q) __hip_atomic_store(&bar[(8 + q) * PAGE], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            while (__hip_atomic_load(&bar[(8 + g) * PAGE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) __builtin_amdgcn_s_sleep(1);
            if (FENCE) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        }
        __syncthreads();
    }
}
template <int FENCE, int NG>
static void run(const char* name, int grid, unsigned* bar, float* data) {
    int rounds = 2000;
    void* args[] = {&bar, &data, &rounds};
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipMemset(bar, 0, 17 * PAGE * 4);
        (void)hipEventRecord(a);
        hipError_t e = hipLaunchCooperativeKernel((const void*)bar_loop<FENCE, NG>, dim3(grid), dim3(1024), args, 0, 0);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        if (rep == 1) printf("%-34s grid %4d, %d groups: %.2f us per barrier (%s)\n", name, grid, NG, ms * 1e3 / rounds, hipGetErrorString(e));
    }
}
int main() {
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    unsigned* bar; float* data;
    (void)hipMalloc(&bar, 17 * PAGE * 4); (void)hipMalloc(&data, 512 * 1024 * 4); (void)hipMemset(data, 0, 512 * 1024 * 4);
    int cus = prop.multiProcessorCount;
    run<0, 8>("hierarchical relaxed", cus, bar, data);
    run<1, 8>("hierarchical release/acquire", cus, bar, data);
    run<0, 1>("flat relaxed", cus, bar, data);
    run<1, 1>("flat release/acquire", cus, bar, data);
    run<1, 4>("hierarchical release/acquire", cus, bar, data);
    run<1, 8>("hierarchical release/acquire", cus / 2, bar, data);
    return 0;
}
