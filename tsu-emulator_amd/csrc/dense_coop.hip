// dense_coop.hip -- K2 in ONE launch of a co-resident grid per call: natural-order sequential Gibbs sweeps on a dense J.
//
// Same mathematics as the superblock path in dense.hip (the sequential pass over a superblock of SB_SIZE positions is
// the unique fixed point of delta = decide(f + L delta), reached exactly by Jacobi iteration from delta = 0), but
//   * all phases of all sweeps run inside one grid of co-resident workgroups separated by grid barriers (release /
//     acquire at agent scope on one counter), so an iteration costs a barrier (~2 us) instead of a launch, and a
//     converged superblock costs nothing more (no budget of early-exit launches);
//   * after the first full triangular pass an iteration is INCREMENTAL: only the sites whose decision changed in the
//     previous iteration (a list of tens) are applied to the rows below them;
//   * the streaming passes (field, triangular pass, strip update of the later rows) read J with 16-byte loads,
//     eight in flight per lane.
// Per sweep: strips N^2 (every row times every superblock's columns: J exactly once) + triangular passes
// N*SB_SIZE/2 elements of J from HBM; the first sweep of a call and every 64th add a field pass of N^2.
//
// Phases (B = grid barrier):
//   A   f_i = b_i + J[i,:].s, logit_i = logit(u_i(sweep)); d0_i = decide(f_i) - s_i for the first superblock      B
//   per superblock [p0, p0 + cnt):
//   T   c_i = sum_{p0 <= j < i} J_ij d0_j; d1_i = decide(f_i + c_i) - s_i; changed sites -> list                  B
//   I*  while the list is not empty: c_i += sum_{(j, D) in list, j < i} J_ij D; re-decide; changed -> next list   B
//   C   s'_i = s_i + d1_i; later rows r: f_r += J[r, p0:p0+cnt].d1, and d0_r = decide(f_r) - s_r for the next
//       superblock's rows                                                                                         B
#include "dense.h"

#define CO_THREADS 1024
#define CO_MAX_N 65536         // bytes of LDS for the staged state vector (phase A)
#define CO_REFRESH 64          // sweeps between two full recomputations of the fields (bounds floating-point drift)
#define CO_SLOTS 64            // iterations recorded per superblock; more than that = not converged (never seen)
#define CO_TIMEOUT 400000000ll // wall_clock64 ticks (100 MHz): 4 s

struct CoopParams {
    const void* J;
    const double* bias;
    int8_t* s0;
    int8_t* s1;
    double* f;
    double* lg;
    double* corr;
    int8_t* d0;
    int8_t* d1;
    const double* uniforms;
    int* lists;
    int* counts;
    unsigned* bar;
    int n, n_sweeps;
    double T;
    uint32_t sweep0, tag, k0, k1;
    unsigned long long* timeline;  // TSU_K2_VERBOSE=2: wall_clock64 ticks per phase kind [A, T, I, C] and phase counts
};

// ---------------------------------------------------------------------------------------------------- grid barrier
// A flat counter with agent-scope release/acquire fences costs 7.3 us per barrier for 256 workgroups: every arrival
// is a serialised memory-side atomic and every workgroup pays an L2 write-back + invalidate.  Here
//   * the barrier is hierarchical over the XCDs (HW_REG_XCC_ID): workgroups arrive on their own XCD's counter, the
//     last arrival of an XCD arrives on the root counter, the last XCD raises one flag per XCD (counters and flags on
//     separate 4 KiB pages): 1.9 us (tools/microbench_gridbar3.hip);
//   * there are NO cache fences: everything one workgroup writes and another reads inside the kernel (fields, logits,
//     corrections, flips, change lists, counters, state) goes through agent-scope relaxed atomic loads and stores,
//     which are performed at the device's coherence point, past the per-CU L1 and the per-XCD L2 (the same property
//     the barrier's own counters rely on).  J, the bias and replayed uniforms are read-only and cached normally.
//     Vectors that every wave needs (the state in phase A, the flips in T and C) are staged once per workgroup into
//     LDS.  A wave's stores are complete (s_waitcnt vmcnt(0)) before its workgroup arrives.
#define BAR_PAGE 1024                 // unsigned per 4 KiB page
#define BAR_GROUPS 16
#define BAR_CNT(g) ((g) * BAR_PAGE)
#define BAR_GEN(g) ((BAR_GROUPS + (g)) * BAR_PAGE)
#define BAR_ROOT (2 * BAR_GROUPS * BAR_PAGE)
#define BAR_GSIZE(g) ((2 * BAR_GROUPS + 1) * BAR_PAGE + (g))
#define BAR_SETUP ((2 * BAR_GROUPS + 2) * BAR_PAGE)
#define BAR_ERR ((2 * BAR_GROUPS + 3) * BAR_PAGE)      // [0] error flag, [1] slowest fixed point, [2] not-converged flag
#define BAR_WORDS ((2 * BAR_GROUPS + 4) * BAR_PAGE)

struct GridBar {
    unsigned* bar;
    unsigned g, gsize, ng, epoch;
};

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

// once per kernel: which XCD am I on, how many workgroups share it, how many XCDs take part (one flat barrier)
static __device__ __forceinline__ bool grid_bar_init(GridBar& B, unsigned* bar) {
    __shared__ unsigned s_init[4];
    if (threadIdx.x == 0) {
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        x &= BAR_GROUPS - 1;
        __hip_atomic_fetch_add(&bar[BAR_GSIZE(x)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&bar[BAR_SETUP], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool ok = bar_wait(bar, &bar[BAR_SETUP], gridDim.x);
        unsigned ng = 0;
        for (int q = 0; q < BAR_GROUPS; ++q) ng += __hip_atomic_load(&bar[BAR_GSIZE(q)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
        s_init[0] = x;
        s_init[1] = __hip_atomic_load(&bar[BAR_GSIZE(x)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_init[2] = ng;
        s_init[3] = ok;
    }
    __syncthreads();
    B.bar = bar;
    B.g = s_init[0];
    B.gsize = s_init[1];
    B.ng = s_init[2];
    B.epoch = 0;
    return s_init[3] != 0;
}

// false once a wait has expired anywhere in the grid (all workgroups then leave the kernel)
static __device__ __noinline__ bool grid_barrier(GridBar& B) {
    __shared__ int s_ok;  // one verdict per workgroup, so that all its threads leave (or stay) together
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores are complete
    __syncthreads();
    B.epoch += 1;
    if (threadIdx.x == 0) {
        unsigned* bar = B.bar;
        const unsigned a = __hip_atomic_fetch_add(&bar[BAR_CNT(B.g)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool leader = a + 1 == B.epoch * B.gsize;
        if (leader) {
            const unsigned b = __hip_atomic_fetch_add(&bar[BAR_ROOT], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (b + 1 == B.epoch * B.ng)
                for (int q = 0; q < BAR_GROUPS; ++q) __hip_atomic_store(&bar[BAR_GEN(q)], B.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        s_ok = bar_wait(bar, &bar[BAR_GEN(B.g)], B.epoch);
    }
    __syncthreads();
    return s_ok != 0;
}

template <typename TJ>
struct JVec;
template <>
struct JVec<float> {
    static constexpr int W = 4;
    typedef float4 raw;
    static __device__ __forceinline__ double get(const raw& q, int e) { return e == 0 ? q.x : e == 1 ? q.y : e == 2 ? q.z : q.w; }
};
template <>
struct JVec<double> {
    static constexpr int W = 2;
    typedef double2 raw;
    static __device__ __forceinline__ double get(const raw& q, int e) { return e == 0 ? q.x : q.y; }
};

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

// the workgroup copies w[c0, c1) (c0 a multiple of 4) into LDS; the caller synchronises
static __device__ __forceinline__ void stage_weights(const int8_t* w, int c0, int c1, int8_t* wl) {
    const int nw4 = (c1 - c0) >> 2;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(w + c0);
    uint32_t* dst = reinterpret_cast<uint32_t*>(wl);
    for (int q = threadIdx.x; q < nw4; q += CO_THREADS) dst[q] = ld(src + q);
    for (int q = c0 + 4 * nw4 + threadIdx.x; q < c1; q += CO_THREADS) wl[q - c0] = ld(w + q);
}

// wave-wide sum over columns [c0, c1) of row[j] * wl[j - wbase], wl an int8 vector in LDS with entries in {-1, 0, 1}.
// c0, wbase and the row start are multiples of the vector width when VEC (checked by the host).  Eight 16-byte loads
// per lane are issued before the first is consumed, also on short rows (out-of-range slots re-read the lane's first
// vector with weight 0).
template <typename TJ, bool VEC>
static __device__ __noinline__ double wave_dot(const TJ* __restrict__ row, const int8_t* wl, int wbase, int c0, int c1,
                                                  int lane) {
    double acc = 0.0;
    if (VEC) {
        constexpr int W = JVec<TJ>::W, U = 8;
        const int c1v = c0 + (c1 - c0) / W * W;
        for (int j0 = c0 + lane * W; j0 < c1v; j0 += U * 64 * W) {
            typename JVec<TJ>::raw v[U];
            uint32_t wp[U];
#pragma unroll
            for (int q = 0; q < U; ++q) {
                const int jq = j0 + q * 64 * W;
                const bool ok = jq < c1v;
                v[q] = *reinterpret_cast<const typename JVec<TJ>::raw*>(row + (ok ? jq : j0));
                wp[q] = ok ? (W == 4 ? *reinterpret_cast<const uint32_t*>(wl + (jq - wbase))
                                     : (uint32_t)*reinterpret_cast<const uint16_t*>(wl + (jq - wbase)))
                           : 0u;
            }
#pragma unroll
            for (int q = 0; q < U; ++q)
#pragma unroll
                for (int e = 0; e < W; ++e) acc += JVec<TJ>::get(v[q], e) * (double)(int8_t)(wp[q] >> (8 * e));
        }
        for (int t = c1v + lane; t < c1; t += 64) acc += (double)row[t] * (double)wl[t - wbase];
    } else {
        for (int j = c0 + lane; j < c1; j += 64) acc += (double)row[j] * (double)wl[j - wbase];
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    return acc;  // valid in lane 0
}

template <typename TJ, bool VEC>
__global__ __launch_bounds__(CO_THREADS) void k2_coop(CoopParams P) {
    const int lane = threadIdx.x & 63;
    const int WPB = CO_THREADS / 64;
    const int gw = blockIdx.x * WPB + (threadIdx.x >> 6), NW = gridDim.x * WPB;
    const int gtid = blockIdx.x * CO_THREADS + threadIdx.x, NT = gridDim.x * CO_THREADS;
    const TJ* __restrict__ J = (const TJ*)P.J;
    const int n = P.n;
    const double T = P.T, invT = 1.0 / P.T;
    const int nsb = (n + SB_SIZE - 1) / SB_SIZE;
    GridBar B;
    if (!grid_bar_init(B, P.bar)) return;
    long long tl_last = wall_clock64();
    unsigned long long tl[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define TL_MARK(kind)                                  \
    if (P.timeline && gtid == 0) {                     \
        const long long now_ = wall_clock64();         \
        tl[kind] += (unsigned long long)(now_ - tl_last); \
        tl[4 + kind] += 1;                             \
        tl_last = now_;                                \
    }
    extern __shared__ int8_t wl[];  // staged weight vector: max(n, SB_SIZE) bytes
    int8_t* s = P.s0;
    int8_t* s_new = P.s1;
    int worst = 0;
    for (int sw = 0; sw < P.n_sweeps; ++sw) {
        const uint32_t t = P.sweep0 + (uint32_t)sw;
        const double* uni = P.uniforms ? P.uniforms + (size_t)sw * n : nullptr;
        // ---- A: field, logits, first guesses of superblock 0 -- on the first sweep of a call and every CO_REFRESH
        // sweeps; in between, phase C keeps every field current (f += J[:, sb] . flips for ALL rows), so that J is
        // streamed once per sweep (the strips) instead of once for the fields plus half again for the strips
        const bool fresh = (sw % CO_REFRESH) == 0;
        const bool next_incremental = sw + 1 < P.n_sweeps && ((sw + 1) % CO_REFRESH) != 0;
        const double* uni_next = P.uniforms ? P.uniforms + (size_t)(sw + 1) * n : nullptr;
        if (fresh) stage_weights(s, 0, n, wl);
        __syncthreads();
        for (int i = gw; fresh && i < n; i += NW) {
            const double acc = wave_dot<TJ, VEC>(J + (size_t)i * n, wl, 0, 0, n, lane);
            if (lane == 0) {
                const double F = acc + (P.bias ? P.bias[i] : 0.0);
                const double lg = co_logit((uint32_t)i, uni, t, P.tag, P.k0, P.k1);
                st(P.f + i, F);
                st(P.lg + i, lg);
                if (i < SB_SIZE)
                    st(P.d0 + i, (int8_t)(co_decide(F, lg, T, invT, (uint32_t)i, uni, t, P.tag, P.k0, P.k1) - (int)wl[i]));
            }
        }
        if (fresh) {
            if (!grid_barrier(B)) return;
            TL_MARK(0);
        }
        int* counts = P.counts + (size_t)sw * nsb * CO_SLOTS;
        for (int p0 = 0; p0 < n; p0 += SB_SIZE, counts += CO_SLOTS) {
            const int cnt = n - p0 < SB_SIZE ? n - p0 : SB_SIZE;
            const int pe = p0 + cnt;
            // ---- T: full triangular pass from the first guesses
            stage_weights(P.d0, p0, pe, wl);
            __syncthreads();
            for (int i = p0 + gw; i < pe; i += NW) {
                // the row's scalars travel together with its J loads, not after them
                const double fi = ld(P.f + i), lgi = ld(P.lg + i);
                const int si = ld(s + i);
                const double acc = wave_dot<TJ, VEC>(J + (size_t)i * n, wl, p0, p0, i, lane);
                if (lane == 0) {
                    const int dold = wl[i - p0];
                    const int dn = co_decide(fi + acc, lgi, T, invT, (uint32_t)i, uni, t, P.tag, P.k0, P.k1) - si;
                    st(P.corr + i, acc);
                    st(P.d1 + i, (int8_t)dn);
                    if (dn != dold) {
                        const int idx = atomicAdd(&counts[1], 1);
                        st(P.lists + SB_SIZE + idx, (i << 1) | (dn - dold > 0 ? 1 : 0));  // iteration 1 writes list 1
                    }
                }
            }
            if (!grid_barrier(B)) return;
            TL_MARK(1);
            // ---- I: incremental iterations until nothing changes
            int k = 1;
            while (true) {
                const int n_in = ld(&counts[k]);
                if (n_in == 0) break;
                if (k + 1 >= CO_SLOTS) {  // out of slots: report, leave the superblock as it is (the host redoes the call)
                    if (gtid == 0) st(&P.bar[BAR_ERR + 2], 1u);
                    break;
                }
                const int* lin = P.lists + (k & 1) * SB_SIZE;
                int* lout = P.lists + ((k + 1) & 1) * SB_SIZE;
                for (int i = p0 + gw; i < pe; i += NW) {
                    const double fi = ld(P.f + i), lgi = ld(P.lg + i), ci = ld(P.corr + i);
                    const int si = ld(s + i), dold = ld(P.d1 + i);
                    double acc = 0.0;
                    bool any = false;
                    for (int e = lane; e < n_in; e += 64) {
                        const int v = ld(lin + e), j = v >> 1;
                        if (j < i) {
                            const double x = (double)J[(size_t)i * n + j];
                            acc += (v & 1) ? x : -x;
                            any = true;
                        }
                    }
                    if (__ballot(any) == 0ull) continue;  // no earlier site changed: this row's decision stands
                    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
                    if (lane == 0) {
                        const double c = ci + acc;
                        st(P.corr + i, c);
                        const int dn = co_decide(fi + c, lgi, T, invT, (uint32_t)i, uni, t, P.tag, P.k0, P.k1) - si;
                        if (dn != dold) {
                            st(P.d1 + i, (int8_t)dn);
                            const int idx = atomicAdd(&counts[k + 1], 1);
                            st(lout + idx, (i << 1) | (dn - dold > 0 ? 1 : 0));
                        }
                    }
                }
                ++k;
                if (!grid_barrier(B)) return;
                TL_MARK(2);
            }
            if (k > worst) worst = k;
            // ---- C: commit the superblock; add its flips to the fields -- of the later rows, or of all rows when the
            // next sweep does not recompute them; first guesses of the next superblock; after the last superblock of a
            // sweep that hands its fields on: next sweep's logits and the first guesses of its superblock 0
            stage_weights(P.d1, p0, pe, wl);
            __syncthreads();
            for (int i = p0 + gtid; i < pe; i += NT) st(s_new + i, (int8_t)(ld(s + i) + wl[i - p0]));
            const bool hand_on = next_incremental && pe == n;
            for (int r = (next_incremental ? 0 : pe) + gw; r < n; r += NW) {
                const bool next_sb = r >= pe && r < pe + SB_SIZE;
                const double fr = ld(P.f + r), lgr = next_sb ? ld(P.lg + r) : 0.0;
                const int sr = next_sb ? (int)ld(s + r) : 0;
                int snr = 0;
                if (hand_on && r < SB_SIZE) snr = r >= p0 ? (int)ld(s + r) + (int)wl[r - p0] : (int)ld(s_new + r);
                const double acc = wave_dot<TJ, VEC>(J + (size_t)r * n, wl, p0, p0, pe, lane);
                if (lane == 0) {
                    const double F = fr + acc;
                    st(P.f + r, F);
                    if (next_sb) st(P.d0 + r, (int8_t)(co_decide(F, lgr, T, invT, (uint32_t)r, uni, t, P.tag, P.k0, P.k1) - sr));
                    if (hand_on) {
                        const double lgn = co_logit((uint32_t)r, uni_next, t + 1, P.tag, P.k0, P.k1);
                        st(P.lg + r, lgn);
                        if (r < SB_SIZE)
                            st(P.d0 + r, (int8_t)(co_decide(F, lgn, T, invT, (uint32_t)r, uni_next, t + 1, P.tag, P.k0, P.k1) - snr));
                    }
                }
            }
            if (!grid_barrier(B)) return;
            TL_MARK(3);
        }
        int8_t* tmp = s;
        s = s_new;
        s_new = tmp;
    }
    if (gtid == 0) P.bar[BAR_ERR + 1] = (unsigned)worst;
    if (P.timeline && gtid == 0)
        for (int q = 0; q < 8; ++q) P.timeline[q] = tl[q];
}

template <typename TJ>
static int coop_sweep(tsu_dense* d, double T, int n_sweeps, uint64_t seed, uint32_t sweep0, uint32_t replica, bool have_uni,
                      int* done) {
    tsu_ctx* ctx = d->ctx;
    const int n = d->n;
    *done = 0;
    if (n > CO_MAX_N) return TSU_OK;  // the staged state vector must fit in LDS: larger systems take the multi-launch path
    const bool vec = (n % JVec<TJ>::W) == 0;
    void (*kern)(CoopParams) = vec ? k2_coop<TJ, true> : k2_coop<TJ, false>;
    const size_t lds_bytes = (size_t)((n > SB_SIZE ? n : SB_SIZE) + 15) / 16 * 16;
    if (lds_bytes > 48 * 1024 && tsu_func_allow_lds(ctx, (const void*)kern, (int)lds_bytes) != hipSuccess) {
        (void)hipGetLastError();
        return TSU_OK;
    }
    int per_cu = 0;
    if (tsu_func_blocks_per_cu(ctx, (const void*)kern, CO_THREADS, lds_bytes, &per_cu) != hipSuccess || per_cu < 1) {
        (void)hipGetLastError();
        d->co_disabled = 1;
        return TSU_OK;
    }
    const int nsb = (n + SB_SIZE - 1) / SB_SIZE;
    const size_t count_ints = (size_t)n_sweeps * nsb * CO_SLOTS;
    if (!d->co_logit) {
        TSU_HIP_TRY(ctx, hipMalloc(&d->co_logit, (size_t)n * 8));
        TSU_HIP_TRY(ctx, hipMalloc(&d->co_corr, (size_t)n * 8));
        TSU_HIP_TRY(ctx, hipMalloc(&d->co_d0, (size_t)n));
        TSU_HIP_TRY(ctx, hipMalloc(&d->co_d1, (size_t)n));
        TSU_HIP_TRY(ctx, hipMalloc(&d->co_lists, 2 * SB_SIZE * sizeof(int)));
        TSU_HIP_TRY(ctx, hipMalloc(&d->co_bar, BAR_WORDS * sizeof(unsigned)));
    }
    if (d->co_counts_cap < count_ints) {
        if (d->co_counts) (void)hipFree(d->co_counts);
        d->co_counts = nullptr;
        d->co_counts_cap = 0;
        TSU_HIP_TRY(ctx, hipMalloc(&d->co_counts, count_ints * sizeof(int)));
        d->co_counts_cap = count_ints;
    }
    TSU_HIP_TRY(ctx, hipMemsetAsync(d->co_counts, 0, count_ints * sizeof(int), ctx->stream));
    TSU_HIP_TRY(ctx, hipMemsetAsync(d->co_bar, 0, BAR_WORDS * sizeof(unsigned), ctx->stream));
    CoopParams P;
    P.J = d->J;
    P.bias = d->bias;
    P.s0 = d->state;
    P.s1 = d->state2;
    P.f = d->field;
    P.lg = d->co_logit;
    P.corr = d->co_corr;
    P.d0 = d->co_d0;
    P.d1 = d->co_d1;
    P.uniforms = have_uni ? d->uniforms : nullptr;
    P.lists = d->co_lists;
    P.counts = d->co_counts;
    P.bar = d->co_bar;
    P.n = n;
    P.n_sweeps = n_sweeps;
    P.T = T;
    P.sweep0 = sweep0;
    P.tag = TSU_TAG_DENSE | (replica << 8);
    P.k0 = (uint32_t)seed;
    P.k1 = (uint32_t)(seed >> 32);
    const char* verbose = getenv("TSU_K2_VERBOSE");
    unsigned long long* d_tl = nullptr;
    if (verbose && atoi(verbose) >= 2) TSU_HIP_TRY(ctx, hipMalloc(&d_tl, 8 * sizeof(unsigned long long)));
    P.timeline = d_tl;
    // one workgroup per CU: every phase is either a stream (16 waves x 8 loads in flight per CU) or tiny, and fewer
    // arrivals make a cheaper barrier; small systems use fewer workgroups still
    int grid = ctx->cus;
    const int useful = (n + CO_THREADS / 64 - 1) / (CO_THREADS / 64);
    if (grid > useful) grid = useful;
    // The grid is one workgroup per CU at most and fits the device (occupancy query above); the launch goes through
    // tsu_launch_grid_sync (cooperative API by default).  The bounded wait in the barrier stays: it turns a GPU shared
    // with another process's long kernel into the fallback below instead of a hang.
    {
        const int rcx = tsu_grid_exclusive_begin(ctx);
        if (rcx != TSU_OK) return rcx;
    }
    hipError_t e = tsu_launch_grid_sync(ctx, (const void*)kern, dim3((unsigned)grid), dim3(CO_THREADS), &P, lds_bytes, ctx->stream);
    if (e != hipSuccess) {  // launch not possible in this configuration: not an error, use the other path
        (void)hipGetLastError();
        d->co_disabled = 1;
        return TSU_OK;
    }
    {
        const int rcx = tsu_grid_exclusive_end(ctx);
        if (rcx != TSU_OK) return rcx;
    }
    unsigned h[4];  // [1] = error flag, [2] = slowest fixed point, [3] = not-converged flag
    TSU_HIP_TRY(ctx, hipMemcpyAsync(h + 1, d->co_bar + BAR_ERR, 3 * sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
    TSU_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (h[1]) {
        // a workgroup waited 4 s at a barrier: the grid was not co-resident (GPU shared with another long-running
        // kernel).  The caller restores the state from its backup and continues on the one-launch-per-iteration path.
        fprintf(stderr, "[tsu] dense sweep: grid barrier timed out (GPU shared?); continuing with one launch per iteration\n");
        d->co_disabled = 1;
        if (d_tl) (void)hipFree(d_tl);
        return TSU_OK;
    }
    if (d_tl) {
        unsigned long long tl[8];
        (void)hipMemcpy(tl, d_tl, sizeof(tl), hipMemcpyDeviceToHost);
        (void)hipFree(d_tl);
        const char* nm[4] = {"A field", "T triangular", "I incremental", "C commit+strips"};
        for (int q = 0; q < 4; ++q)
            fprintf(stderr, "[tsu]   phase %-16s %6llu x  %8.1f us total  %6.2f us each\n", nm[q], tl[4 + q], tl[q] / 100.0, tl[4 + q] ? tl[q] / 100.0 / tl[4 + q] : 0.0);
    }
    if (verbose)
        fprintf(stderr, "[tsu] dense cooperative: n=%d, %d workgroups, %d sweeps, slowest fixed point after %u iterations%s\n", n, grid,
                n_sweeps, h[2], h[3] ? " (NOT converged)" : "");
    if (h[3]) return TSU_OK;  // *done stays 0: the caller restores the state and takes the other path
    if (n_sweeps & 1) {
        int8_t* tmp = d->state;
        d->state = d->state2;
        d->state2 = tmp;
    }
    *done = 1;
    return TSU_OK;
}

int tsu_dense_coop_sweep(tsu_dense* d, double T, int n_sweeps, uint64_t seed, uint32_t sweep0, uint32_t replica, bool have_uni,
                         int* done) {
    if (d->dtype == TSU_DTYPE_F64) return coop_sweep<double>(d, T, n_sweeps, seed, sweep0, replica, have_uni, done);
    return coop_sweep<float>(d, T, n_sweeps, seed, sweep0, replica, have_uni, done);
}
