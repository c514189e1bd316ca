// Ping-pong between two workgroups on the SAME XCD (blocks 0 and 8) and on DIFFERENT XCDs (blocks 0 and 1) with
// different access flavours: does a cheaper-than-agent-scope access see the other CU's store, and how fast?
// (research for a later round: XCD-local iterations of the dense sweep.  Not product code.)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
template <int MODE>
static __device__ __forceinline__ unsigned ld_flag(unsigned* p) {
    if (MODE == 0) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (MODE == 1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    unsigned v;
    if (MODE == 2) asm volatile("global_load_dword %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (MODE == 3) asm volatile("global_load_dword %0, %1, off nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int MODE>
static __device__ __forceinline__ void st_flag(unsigned* p, unsigned v) {
    if (MODE == 0) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
    if (MODE == 1) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); return; }
    asm volatile("global_store_dword %0, %1, off\n s_waitcnt vmcnt(0)" ::"v"(p), "v"(v) : "memory");
}
template <int MODE>
__global__ void pingpong(unsigned* flags, int partner_block, int rounds, unsigned* result) {
    if (threadIdx.x != 0) return;
    unsigned* a = flags;        // written by block 0
    unsigned* b = flags + 1024;  // written by the partner (separate 4 KiB page)
    long long t0 = wall_clock64();
    unsigned done = 0;
    if (blockIdx.x == 0) {
        for (int r = 1; r <= rounds; ++r) {
            st_flag<MODE>(a, (unsigned)r);
            int spins = 0;
            while (ld_flag<MODE>(b) < (unsigned)r && ++spins < 2000000) {}
            if (spins >= 2000000) break;
            done = r;
        }
        result[0] = done;
        result[1] = (unsigned)(wall_clock64() - t0);
    } else if (blockIdx.x == partner_block) {
        for (int r = 1; r <= rounds; ++r) {
            int spins = 0;
            while (ld_flag<MODE>(a) < (unsigned)r && ++spins < 2000000) {}
            if (spins >= 2000000) break;
            st_flag<MODE>(b, (unsigned)r);
        }
    }
}
template <int MODE>
static void run(const char* name, int partner, unsigned* flags, unsigned* res) {
    (void)hipMemset(flags, 0, 2048 * 4);
    (void)hipMemset(res, 0, 8);
    int rounds = 2000;
    pingpong<MODE><<<16, 64>>>(flags, partner, rounds, res);
    (void)hipDeviceSynchronize();
    unsigned h[2];
    (void)hipMemcpy(h, res, 8, hipMemcpyDeviceToHost);
    printf("%-34s partner block %2d (%s XCD): %4u / %d round trips, %.2f us each\n", name, partner, partner == 8 ? "same" : "other", h[0],
           rounds, h[0] ? h[1] / 100.0 / h[0] : 0.0);
}
int main() {
    unsigned *flags, *res;
    (void)hipMalloc(&flags, 2048 * 4);
    (void)hipMalloc(&res, 8);
    for (int partner : {8, 1}) {
        run<0>("agent-scope atomics", partner, flags, res);
        run<1>("workgroup-scope atomics", partner, flags, res);
        run<2>("plain store + sc0 load", partner, flags, res);
        run<3>("plain store + nt load", partner, flags, res);
    }
    return 0;
}
