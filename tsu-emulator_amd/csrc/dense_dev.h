// dense_dev.h -- device helpers shared by the grid-synchronising dense kernels (dense_coop.hip: k2_coop, k2_pipe; dense_own.hip: k2_own):
// the counter / error-flag page layout, bounded waits, agent-scope accesses and the decision helpers.
#pragma once
#include "dense.h"

#define CO_REFRESH 64          // sweeps between two full recomputations of the fields (bounds floating-point drift)
#define CO_TIMEOUT 400000000ll // wall_clock64 ticks (100 MHz): 4 s

#define BAR_PAGE 1024                 // unsigned per 4 KiB page
#define BAR_GROUPS 16
#define BAR_CNT(g) ((g) * BAR_PAGE)
#define BAR_GEN(g) ((BAR_GROUPS + (g)) * BAR_PAGE)
#define BAR_ROOT (2 * BAR_GROUPS * BAR_PAGE)
#define BAR_GSIZE(g) ((2 * BAR_GROUPS + 1) * BAR_PAGE + (g))
#define BAR_SETUP ((2 * BAR_GROUPS + 2) * BAR_PAGE)
#define BAR_ERR ((2 * BAR_GROUPS + 3) * BAR_PAGE)      // [0] error flag, [1] slowest fixed point, [2] not-converged flag
#define BAR_WORDS ((2 * BAR_GROUPS + 4) * BAR_PAGE)

static __device__ __forceinline__ bool bar_wait(unsigned* bar, const unsigned* word, unsigned target) {
    // one agent-scope load per poll (the load's own latency is the back-off); the error flag and the clock are looked at
    // every 64th poll only -- checking them every time doubled the time a barrier takes to release
    const long long t0 = wall_clock64();
    for (unsigned spins = 0;; ++spins) {
        if (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return true;
        if ((spins & 63u) == 63u) {
            if (__hip_atomic_load(&bar[BAR_ERR], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
            if (wall_clock64() - t0 > CO_TIMEOUT) {
                __hip_atomic_store(&bar[BAR_ERR], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
    }
}

// dense_decide (dense.h) with the rare close call out of line: the float64 exp and the Philox block it needs would
// otherwise be inlined at every decision site and push the kernel past its 128 VGPRs
static __device__ __noinline__ int co_decide_exact(double F, double T, uint32_t site, const double* __restrict__ uniforms,
                                                   uint32_t sweep, uint32_t tag, uint32_t k0, uint32_t k1) {
    const double u = uniforms ? uniforms[site] : dense_uniform(site, sweep, tag, k0, k1);
    return (u < sigmoid_clamped(F / T)) ? 1 : 0;
}
static __device__ __forceinline__ int co_decide(double F, double lg, double T, double invT, uint32_t site,
                                                const double* __restrict__ uniforms, uint32_t sweep, uint32_t tag, uint32_t k0,
                                                uint32_t k1) {
    const double xa = F * invT;
    if (fabs(fabs(xa) - 20.0) < 1e-9 || fabs(xa - lg) <= 1e-9 * (1.0 + fabs(lg)))
        return co_decide_exact(F, T, site, uniforms, sweep, tag, k0, k1);
    if (xa > 20.0) return 1;
    if (xa < -20.0) return 0;
    return xa > lg ? 1 : 0;
}
static __device__ __noinline__ double co_logit(uint32_t site, const double* __restrict__ uniforms, uint32_t sweep, uint32_t tag,
                                               uint32_t k0, uint32_t k1) {
    const double u = uniforms ? uniforms[site] : dense_uniform(site, sweep, tag, k0, k1);
    return log(u) - log1p(-u);
}

// agent-scope accesses for data shared between workgroups inside the kernel (see the barrier's note)
template <typename V>
static __device__ __forceinline__ V ld(const V* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename V>
static __device__ __forceinline__ void st(V* p, V v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

