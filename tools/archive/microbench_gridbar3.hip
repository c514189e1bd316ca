// Which part of the agent-scope fences costs?  Hierarchical barrier keyed by XCC_ID (not product code).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define PAGE 1024
// cnt[g] at g*PAGE, gen[g] at (16+g)*PAGE, root at 32*PAGE, gsize[g] at 33*PAGE+g
// MODE 0: no fences; 1: all wbl2 only; 2: all inv sc1 only; 3: leader wbl2+inv sc1, others inv sc0; 4: all wbl2 + inv sc1
template <int MODE>
__global__ __launch_bounds__(1024) void bar_loop(unsigned* bar, float* data, int rounds, unsigned* info) {
    __shared__ unsigned s_g, s_gs, s_ng;
    if (threadIdx.x == 0) {
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        x &= 0xF;
        s_g = x;
        __hip_atomic_fetch_add(&bar[33 * PAGE + x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // flat barrier so that the group sizes are final
        __hip_atomic_fetch_add(&bar[34 * PAGE], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(&bar[34 * PAGE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) __builtin_amdgcn_s_sleep(1);
        unsigned ng = 0;
        for (int q = 0; q < 16; ++q) ng += __hip_atomic_load(&bar[33 * PAGE + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
        s_ng = ng;
        s_gs = __hip_atomic_load(&bar[33 * PAGE + x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (blockIdx.x < 16) info[blockIdx.x] = x | (s_gs << 8) | (ng << 16);
    }
    __syncthreads();
    const unsigned g = s_g, gsize = s_gs, ng = s_ng;
    unsigned epoch = 0;
    for (int r = 0; r < rounds; ++r) {
        data[(size_t)blockIdx.x * 1024 + threadIdx.x] += 1.0f;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        epoch += 1;
        if (threadIdx.x == 0) {
            if (MODE == 1 || MODE == 4) asm volatile("buffer_wbl2 sc1\n s_waitcnt vmcnt(0)" ::: "memory");
            unsigned a = __hip_atomic_fetch_add(&bar[g * PAGE], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bool leader = (a + 1 == epoch * gsize);
            if (leader) {
                if (MODE == 3) asm volatile("buffer_wbl2 sc1\n s_waitcnt vmcnt(0)\n buffer_inv sc1\n s_waitcnt vmcnt(0)" ::: "memory");
                unsigned b = __hip_atomic_fetch_add(&bar[32 * PAGE], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (b + 1 == epoch * ng)
                    for (int q = 0; q < 16; ++q) __hip_atomic_store(&bar[(16 + q) * PAGE], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            while (__hip_atomic_load(&bar[(16 + g) * PAGE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) __builtin_amdgcn_s_sleep(1);
            if (MODE == 2 || MODE == 4) asm volatile("buffer_inv sc1\n s_waitcnt vmcnt(0)" ::: "memory");
            if (MODE == 3 && !leader) asm volatile("buffer_inv sc0\n s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
    }
}
template <int MODE>
static void run(const char* name, int grid, unsigned* bar, float* data, unsigned* info) {
    int rounds = 2000;
    void* args[] = {&bar, &data, &rounds, &info};
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipMemset(bar, 0, 35 * PAGE * 4);
        (void)hipEventRecord(a);
        hipError_t e = hipLaunchCooperativeKernel((const void*)bar_loop<MODE>, dim3(grid), dim3(1024), args, 0, 0);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        if (rep == 1) printf("%-44s grid %4d: %.2f us per barrier (%s)\n", name, grid, ms * 1e3 / rounds, hipGetErrorString(e));
    }
}
int main() {
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    unsigned *bar, *info; float* data;
    (void)hipMalloc(&bar, 35 * PAGE * 4); (void)hipMalloc(&info, 64); (void)hipMalloc(&data, 512 * 1024 * 4); (void)hipMemset(data, 0, 512 * 1024 * 4);
    int cus = prop.multiProcessorCount;
    run<0>("xcc groups, no fences", cus, bar, data, info);
    run<1>("xcc groups, all wbl2", cus, bar, data, info);
    run<2>("xcc groups, all inv sc1", cus, bar, data, info);
    run<3>("xcc groups, leader wbl2+inv sc1, rest inv sc0", cus, bar, data, info);
    run<4>("xcc groups, all wbl2 + inv sc1", cus, bar, data, info);
    unsigned h[16]; (void)hipMemcpy(h, info, 64, hipMemcpyDeviceToHost);
    printf("block -> xcc (group size, groups):");
    for (int i = 0; i < 16; ++i) printf(" %u(%u,%u)", h[i] & 0xFF, (h[i] >> 8) & 0xFF, h[i] >> 16);
    printf("\n");
    return 0;
}
