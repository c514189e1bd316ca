// Cost of one grid barrier on MI355X for a cooperative grid of one workgroup per CU (not product code).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
template <int MODE>  // 0: relaxed only, 1: + release/acquire fences (agent), 2: fences + dirty data each round
__global__ __launch_bounds__(1024) void bar_loop(unsigned* bar, float* data, int rounds, int sleep) {
    unsigned epoch = 0;
    for (int r = 0; r < rounds; ++r) {
        if (MODE == 2) data[(size_t)blockIdx.x * 1024 + threadIdx.x] += 1.0f;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        epoch += 1;
        if (threadIdx.x == 0) {
            if (MODE >= 1) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            __hip_atomic_fetch_add(&bar[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = epoch * gridDim.x;
            while (__hip_atomic_load(&bar[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if (sleep == 1) __builtin_amdgcn_s_sleep(1);
                else if (sleep == 2) __builtin_amdgcn_s_sleep(2);
                else if (sleep == 8) __builtin_amdgcn_s_sleep(8);
            }
            if (MODE >= 1) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        }
        __syncthreads();
    }
}
template <int MODE>
static void run(const char* name, int grid, int threads, int sleep, unsigned* bar, float* data) {
    int rounds = 2000;
    void* args[] = {&bar, &data, &rounds, &sleep};
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipMemset(bar, 0, 16);
        (void)hipEventRecord(a);
        hipError_t e = hipLaunchCooperativeKernel((const void*)bar_loop<MODE>, dim3(grid), dim3(threads), args, 0, 0);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        if (rep == 1) printf("%-28s grid %4d x %4d sleep %d: %.2f us per barrier (%s)\n", name, grid, threads, sleep, ms * 1e3 / rounds, hipGetErrorString(e));
    }
}
int main() {
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    unsigned* bar; float* data;
    (void)hipMalloc(&bar, 16); (void)hipMalloc(&data, 512 * 1024 * 4); (void)hipMemset(data, 0, 512 * 1024 * 4);
    int cus = prop.multiProcessorCount;
    for (int sleep : {0, 1, 2, 8}) run<0>("relaxed", cus, 1024, sleep, bar, data);
    for (int sleep : {0, 2}) run<1>("release/acquire", cus, 1024, sleep, bar, data);
    for (int sleep : {0, 2}) run<2>("release/acquire + dirty", cus, 1024, sleep, bar, data);
    run<1>("release/acquire", cus, 256, 2, bar, data);
    run<1>("release/acquire", cus / 2, 1024, 2, bar, data);
    run<1>("release/acquire", cus / 8, 1024, 2, bar, data);
    run<1>("release/acquire", 2 * cus, 1024, 2, bar, data);
    return 0;
}
