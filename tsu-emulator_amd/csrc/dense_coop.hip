// dense_coop.hip -- K2 in ONE cooperative launch per call: natural-order sequential Gibbs sweeps on a dense J.
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
// Per sweep: field pass N^2 + strips N^2/2 + triangular passes N*SB_SIZE/2 elements of J from HBM.
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
// A flat counter costs 7.3 us per barrier for 256 workgroups (every arrival is a serialised memory-side atomic and
// every workgroup pays an L2 write-back + invalidate).  This one is hierarchical over the XCDs (tools/microbench_gridbar3:
// 2.4 us): workgroups arrive on their own XCD's counter (HW_REG_XCC_ID; counters and flags on separate 4 KiB pages);
// the last arrival of an XCD -- by then every store of that XCD has reached its L2 (s_waitcnt vmcnt(0) precedes each
// arrival) -- writes the L2 back, invalidates it, and arrives on the root counter; the last XCD raises one flag per
// XCD.  Every other workgroup invalidates on wake-up (agent scope: its CU's L1 -- a workgroup-scope invalidate is
// not enough, it leaves stale L1 lines and wrong results -- and the L2 again, which is cheap once it is clean).
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
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (__hip_atomic_load(&bar[BAR_ERR], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
        if (wall_clock64() - t0 > CO_TIMEOUT) {
            __hip_atomic_store(&bar[BAR_ERR], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
    }
    return true;
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
static __device__ __forceinline__ bool grid_barrier(GridBar& B) {
    __shared__ int s_ok;  // one verdict per workgroup, so that all its threads leave (or stay) together
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores have reached the L2
    __syncthreads();
    B.epoch += 1;
    if (threadIdx.x == 0) {
        unsigned* bar = B.bar;
        const unsigned a = __hip_atomic_fetch_add(&bar[BAR_CNT(B.g)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool leader = a + 1 == B.epoch * B.gsize;
        if (leader) {
            asm volatile("buffer_wbl2 sc1\n s_waitcnt vmcnt(0)\n buffer_inv sc1\n s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned b = __hip_atomic_fetch_add(&bar[BAR_ROOT], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (b + 1 == B.epoch * B.ng)
                for (int q = 0; q < BAR_GROUPS; ++q) __hip_atomic_store(&bar[BAR_GEN(q)], B.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const bool ok = bar_wait(bar, &bar[BAR_GEN(B.g)], B.epoch);
        if (!leader) asm volatile("buffer_inv sc1\n s_waitcnt vmcnt(0)" ::: "memory");
        s_ok = ok;
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

// wave-wide sum over columns [c0, c1) of row[j] * w[j], w an int8 vector with entries in {-1, 0, 1}.  c0 and the row
// start are multiples of the vector width when VEC (checked by the host).  Eight 16-byte loads per lane are issued
// before the first is consumed, also on short rows (out-of-range slots re-read the lane's first vector with weight 0).
template <typename TJ, bool VEC>
static __device__ __forceinline__ double wave_dot(const TJ* __restrict__ row, const int8_t* __restrict__ w, int c0, int c1,
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
                wp[q] = ok ? (W == 4 ? *reinterpret_cast<const uint32_t*>(w + jq) : (uint32_t)*reinterpret_cast<const uint16_t*>(w + jq)) : 0u;
            }
#pragma unroll
            for (int q = 0; q < U; ++q)
#pragma unroll
                for (int e = 0; e < W; ++e) acc += JVec<TJ>::get(v[q], e) * (double)(int8_t)(wp[q] >> (8 * e));
        }
        for (int t = c1v + lane; t < c1; t += 64) acc += (double)row[t] * (double)w[t];
    } else {
        for (int j = c0 + lane; j < c1; j += 64) acc += (double)row[j] * (double)w[j];
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
    int8_t* s = P.s0;
    int8_t* s_new = P.s1;
    int worst = 0;
    for (int sw = 0; sw < P.n_sweeps; ++sw) {
        const uint32_t t = P.sweep0 + (uint32_t)sw;
        const double* uni = P.uniforms ? P.uniforms + (size_t)sw * n : nullptr;
        // ---- A: field, logits, first guesses of superblock 0
        for (int i = gw; i < n; i += NW) {
            const double acc = wave_dot<TJ, VEC>(J + (size_t)i * n, s, 0, n, lane);
            if (lane == 0) {
                const double F = acc + (P.bias ? P.bias[i] : 0.0);
                const double u = uni ? uni[i] : dense_uniform((uint32_t)i, t, P.tag, P.k0, P.k1);
                const double lg = log(u) - log1p(-u);
                P.f[i] = F;
                P.lg[i] = lg;
                if (i < SB_SIZE) P.d0[i] = (int8_t)(dense_decide(F, lg, T, invT, (uint32_t)i, uni, t, P.tag, P.k0, P.k1) - (int)s[i]);
            }
        }
        if (!grid_barrier(B)) return;
        TL_MARK(0);
        int* counts = P.counts + (size_t)sw * nsb * CO_SLOTS;
        for (int p0 = 0; p0 < n; p0 += SB_SIZE, counts += CO_SLOTS) {
            const int cnt = n - p0 < SB_SIZE ? n - p0 : SB_SIZE;
            const int pe = p0 + cnt;
            // ---- T: full triangular pass from the first guesses
            for (int i = p0 + gw; i < pe; i += NW) {
                const double acc = wave_dot<TJ, VEC>(J + (size_t)i * n, P.d0, p0, i, lane);
                if (lane == 0) {
                    const int dold = P.d0[i];
                    const int dn = dense_decide(P.f[i] + acc, P.lg[i], T, invT, (uint32_t)i, uni, t, P.tag, P.k0, P.k1) - (int)s[i];
                    P.corr[i] = acc;
                    P.d1[i] = (int8_t)dn;
                    if (dn != dold) {
                        const int idx = atomicAdd(&counts[1], 1);
                        P.lists[SB_SIZE + idx] = (i << 1) | (dn - dold > 0 ? 1 : 0);  // iteration 1 writes list 1
                    }
                }
            }
            if (!grid_barrier(B)) return;
            TL_MARK(1);
            // ---- I: incremental iterations until nothing changes
            int k = 1;
            while (true) {
                const int n_in = __hip_atomic_load(&counts[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (n_in == 0) break;
                if (k + 1 >= CO_SLOTS) {  // out of slots: report, leave the superblock as it is (the host redoes the call)
                    if (gtid == 0) __hip_atomic_store(&P.bar[BAR_ERR + 2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                const int* __restrict__ lin = P.lists + (k & 1) * SB_SIZE;
                int* __restrict__ lout = P.lists + ((k + 1) & 1) * SB_SIZE;
                for (int i = p0 + gw; i < pe; i += NW) {
                    double acc = 0.0;
                    bool any = false;
                    for (int e = lane; e < n_in; e += 64) {
                        const int v = lin[e], j = v >> 1;
                        if (j < i) {
                            const double x = (double)J[(size_t)i * n + j];
                            acc += (v & 1) ? x : -x;
                            any = true;
                        }
                    }
                    if (__ballot(any) == 0ull) continue;  // no earlier site changed: this row's decision stands
                    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
                    if (lane == 0) {
                        const double c = P.corr[i] + acc;
                        P.corr[i] = c;
                        const int dold = P.d1[i];
                        const int dn = dense_decide(P.f[i] + c, P.lg[i], T, invT, (uint32_t)i, uni, t, P.tag, P.k0, P.k1) - (int)s[i];
                        if (dn != dold) {
                            P.d1[i] = (int8_t)dn;
                            const int idx = atomicAdd(&counts[k + 1], 1);
                            lout[idx] = (i << 1) | (dn - dold > 0 ? 1 : 0);
                        }
                    }
                }
                ++k;
                if (!grid_barrier(B)) return;
                TL_MARK(2);
            }
            if (k > worst) worst = k;
            // ---- C: commit the superblock, add its flips to every later field, first guesses of the next superblock
            for (int i = p0 + gtid; i < pe; i += NT) s_new[i] = (int8_t)(s[i] + P.d1[i]);
            for (int r = pe + gw; r < n; r += NW) {
                const double acc = wave_dot<TJ, VEC>(J + (size_t)r * n, P.d1, p0, pe, lane);
                if (lane == 0) {
                    const double F = P.f[r] + acc;
                    P.f[r] = F;
                    if (r < pe + SB_SIZE)
                        P.d0[r] = (int8_t)(dense_decide(F, P.lg[r], T, invT, (uint32_t)r, uni, t, P.tag, P.k0, P.k1) - (int)s[r]);
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
    const bool vec = (n % JVec<TJ>::W) == 0;
    void (*kern)(CoopParams) = vec ? k2_coop<TJ, true> : k2_coop<TJ, false>;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, CO_THREADS, 0) != hipSuccess || per_cu < 1) {
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
    void* args[] = {&P};
    hipError_t e = hipLaunchCooperativeKernel((const void*)kern, dim3((unsigned)grid), dim3(CO_THREADS), args, 0, ctx->stream);
    if (e != hipSuccess) {  // no cooperative launch on this device / configuration: not an error, use the other path
        (void)hipGetLastError();
        d->co_disabled = 1;
        return TSU_OK;
    }
    unsigned h[4];  // [1] = error flag, [2] = slowest fixed point, [3] = not-converged flag
    TSU_HIP_TRY(ctx, hipMemcpyAsync(h + 1, d->co_bar + BAR_ERR, 3 * sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
    TSU_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (h[1]) return tsu_fail(ctx, TSU_E_HIP, "dense_sweep: grid barrier timed out (cooperative kernel not co-resident?); state invalid");
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
