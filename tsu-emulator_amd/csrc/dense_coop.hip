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
#include "dense_dev.h"

#define CO_THREADS 1024
#define CO_MAX_N 65536         // bytes of LDS for the staged state vector (phase A)
#define CO_SLOTS 64            // iterations recorded per superblock; more than that = not converged (never seen)

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
struct GridBar {
    unsigned* bar;
    unsigned g, gsize, ng, epoch;
};

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
template <typename TJ, bool VEC, int U = 8>
static __device__ __noinline__ double wave_dot(const TJ* __restrict__ row, const int8_t* wl, int wbase, int c0, int c1,
                                                  int lane) {
    double acc = 0.0;
    if (VEC) {
        constexpr int W = JVec<TJ>::W;
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


// ==================================================================================================== k2_pipe
// The same fixed point, organised as a two-role pipeline so that the latency-bound iterations of superblock s+1 hide
// behind the bandwidth-bound strip of superblock s (profiles/r01_k2_notes.txt: ~40 grid-wide iterations per sweep at ~6 us
// sat serially next to 0.2 ms of J strips):
//   * STREAMERS (all workgroups but PP_NS): every wave owns a fixed set of rows (r = wave + m * waves) and keeps their
//     fields for the whole call; per superblock it adds the strip J[r, sb] . flips to them -- the rows of the superblock
//     that is solved NEXT first, published to P.f and signalled (C1), then all its other rows (C2).  This is the J stream:
//     every element of J once per sweep.
//   * SOLVERS (PP_NS workgroups): teams of PP_TW waves own 64 consecutive rows of the current superblock, lane = row.
//     Jacobi iteration on FLIP MASKS: every team publishes the 64-bit mask of its rows' current decisions as two 8-byte
//     {mask32, iteration tag} granules (one sc1 store each); every solver workgroup polls all 128 granules of the
//     iteration (validated by their tags: no barrier, no atomics, two buffers by iteration parity), turns the TOGGLES
//     against the previous iteration into a sorted list, and every row adds +-J^T[j][i] of the toggled earlier sites j
//     (coalesced: lanes are consecutive i) to its correction and decides again.  Unchanged masks = the fixed point.
//     One iteration is a publish -> poll hand-off (~1 us) plus one gather, instead of append atomics + a grid barrier.
//   Hand-offs between the roles are two monotone counters (C1: fields of the next superblock complete; SOLVED: flips of
//   the superblock final), everything shared goes through agent-scope (sc1) accesses drained before the signal.
// Superblock size of the pipeline: the fixed costs (hand-offs, commit, the iterations' latency) are per superblock, the
// exposed strip + triangle bytes and the iteration count grow with it.  Measured on the finished kernel (fp32, T = 1, ms per
// sweep, superblocks of 2048 / 4096 / 8192): n = 1024 0.053 / 0.059 / 0.068, 2048 0.061 / 0.067 / 0.083, 4096 0.106 / 0.103 / 0.122,
// 8192 0.185 / 0.170 / 0.190, 12288 0.283 / 0.262 / 0.273, 16384 0.373 / 0.333 / 0.363 -- 2048 for n <= 2048, 4096 above.
#define PP_SB_MAX 8192
#define PP_TEAMS 4                // 64-row groups per solver workgroup
#define PP_GRAN_MAX (PP_SB_MAX / 32)
#define PP_MAXR 24                // rows one streamer wave may own
#define PB_SOLVED BAR_CNT(1)       // counter words on pages of their own (the grid-barrier layout is not used here)
// C1 and T count ~240 streamer workgroups each: sharded eight ways by workgroup index (one atomic per ~12 ns on ONE word would put
// 3 us of arrivals in front of every wait; the guide's "fanin" row), the waiter's first eight lanes poll one shard each
#ifndef PP_SHARDS
#define PP_SHARDS 8
#endif
#define PB_C1(x) ((2 + (x)) * BAR_PAGE)              // pages 2 .. 2 + PP_SHARDS - 1
#define PB_T(x) ((2 + PP_SHARDS + (x)) * BAR_PAGE)   // the next PP_SHARDS pages (PP_SHARDS <= 16: the error words sit on page 35)
#define PP_BATCH 16               // J^T elements in flight per lane in the gather
#ifndef PP_U
#define PP_U 8                    // 16-byte loads in flight per lane in the streamers' strip passes
#endif


struct PipeParams {
    const void* J;
    const void* JT;
    const double* bias;
    int8_t* state;       // updated in place, superblock by superblock
    double* f;           // fields of the superblock that is solved next (written by streamers, read by solvers)
    int8_t* d1;          // flips of the superblock just solved, indexed from its first row
    const double* uniforms;
    unsigned long long* masks;  // [2][PP_GRAN] granules of the iteration + [PP_GRAN] first-guess granules (tag = superblock number)
    double* corr;               // corrections of the first guesses (triangular pass, written by the streamers)
    int8_t* d0;                 // first guesses of the superblock that is solved next, from its first row (written by the rows' owners)
    unsigned* bar;
    int n, n_sweeps;
    double T;
    uint32_t sweep0, tag, k0, k1;
    const double* temps;  // one temperature per sweep of the call (an annealing schedule), or nullptr: T
    int8_t* samples;      // the state after sweep rec_from + m rec_every (m = 1, 2, ...) goes to samples + (m - 1) n, or nullptr
    int rec_from, rec_every;
    double* fields_all;  // [n] fields of every row, kept from call to call (see dense.h)
    int resume;          // fields_all holds the fields of the state the call starts from: no pass over J to rebuild them
    int persist;         // leave the fields of the final state in fields_all (the call's last sweep then updates every row)
    int refresh_off;     // sweeps since the fields were last computed from scratch, at the start of the call
    unsigned long long* timeline;  // TSU_K2_VERBOSE=2: solver workgroup 0, ticks in [wait C1, poll, gather+decide, commit], [4] iterations, [5] toggles;
                                   // first streamer workgroup, wave 0: [6] wait SOLVED, [7] stage, [8] priority rows + signal, [9] other rows
};

// one lane polls, the workgroup follows its verdict
static __device__ __forceinline__ bool wg_wait(unsigned* bar, int word, unsigned target, int* s_ok) {
    if (threadIdx.x == 0) *s_ok = bar_wait(bar, &bar[word], target) ? 1 : 0;
    __syncthreads();
    const bool ok = *s_ok != 0;
    __syncthreads();
    return ok;
}

// wait until every streamer workgroup (indices first_wg .. n_wg - 1 of the grid) has arrived `events` times at the sharded counter
// whose shard x lives at word0 + x * stride: lanes 0 .. 7 of the workgroup's first wave poll one shard each
static __device__ __forceinline__ bool wg_wait_sharded(unsigned* bar, int word0, int stride, unsigned events, int first_wg, int n_wg, int* s_ok) {
    if (threadIdx.x < 64) {
        const int x = threadIdx.x & (PP_SHARDS - 1);
        const int hi = n_wg - 1 - x >= 0 ? (n_wg - 1 - x) / PP_SHARDS + 1 : 0, lo = first_wg - 1 - x >= 0 ? (first_wg - 1 - x) / PP_SHARDS + 1 : 0;
        const unsigned target = events * (unsigned)(hi - lo);
        const unsigned* word = bar + word0 + x * stride;
        bool ok = true;
        const long long t0 = wall_clock64();
        for (unsigned spins = 0;; ++spins) {
            const bool there = threadIdx.x >= PP_SHARDS || __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target;
            if (__ballot(!there) == 0ull) break;
            if ((spins & 63u) == 63u) {
                if (__hip_atomic_load(&bar[BAR_ERR], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) || wall_clock64() - t0 > CO_TIMEOUT) {
                    __hip_atomic_store(&bar[BAR_ERR], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = false;
                    break;
                }
            }
        }
        const bool all_ok = __ballot(!ok) == 0ull;  // (the lanes leave the loop one by one on an error: the wave is whole again here)
        if (threadIdx.x == 0) *s_ok = all_ok ? 1 : 0;
    }
    __syncthreads();
    const bool ok = *s_ok != 0;
    __syncthreads();
    return ok;
}

template <typename TJ, bool VEC, int SB, int U2>
__global__ __launch_bounds__(CO_THREADS) void k2_pipe(PipeParams P) {
    // superblock of SB positions: SB / 256 solver workgroups with four 64-row groups each, SB / 32 mask granules
    constexpr int NS = SB / 256, GRAN = SB / 32, GPL = GRAN / 64;
    static_assert(NS * PP_TEAMS * 64 == SB && (GPL == 1 || GPL == 2 || GPL == 4), "solver workgroups x 4 groups of 64 rows = one superblock");
    extern __shared__ int8_t wl[];  // streamers: staged state (n bytes) / staged flips (SB bytes)
    __shared__ double s_f[CO_THREADS / 64][PP_MAXR];
    __shared__ double s_red[CO_THREADS / 64][PP_TEAMS][64];  // partial sums: [wave][group of the workgroup][row]
    __shared__ int8_t s_sb[SB];          // solvers: state of the superblock at the start of the sweep
    __shared__ unsigned s_m[2][GRAN];      // solvers: masks of the last two iterations
    __shared__ unsigned short s_list[SB];
    __shared__ int s_nlist, s_ok, s_fail;
    __shared__ unsigned s_c1, s_t;
    __shared__ int8_t wl2[SB];           // streamers: first guesses of the superblock that is solved next
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const TJ* __restrict__ J = (const TJ*)P.J;
    const TJ* __restrict__ JT = (const TJ*)P.JT;
    const int n = P.n, nsb = (n + SB - 1) / SB;
    const double T = P.T;
    const unsigned n_stream = gridDim.x - NS;
    if (threadIdx.x == 0) {
        s_c1 = 0;
        s_t = 0;
        s_fail = 0;
    }
    __syncthreads();
    unsigned ev = 0;    // C1 events so far (one per field pass and one per strip, the same count in both roles)
    unsigned seq = 0;   // superblocks solved so far

    if (blockIdx.x >= NS) {
        // ------------------------------------------------------------------------------------------ streamer
        // row r belongs to wave ws = r mod W.  Wave-major numbering: the W does not divide a superblock (4096 rows over 3840 waves),
        // so 256 waves own TWO rows of every superblock -- numbered this way they are one wave in each workgroup rather than
        // every wave of sixteen workgroups, and no CU has twice the others' priority work
        const int W = (int)n_stream * (CO_THREADS / 64), ws = wv * (int)n_stream + (int)(blockIdx.x - NS);
        unsigned tev = 0;  // triangular passes so far
        // first guess of row r (of the superblock that starts at q0, solved in sweep number sw_of): the decision on the field alone
        auto first_guess = [&](int r, int q0, double F, int sw_of) {
            const uint32_t tt = P.sweep0 + (uint32_t)sw_of;
            const double* un = P.uniforms ? P.uniforms + (size_t)sw_of * n : nullptr;
            const double lgr = co_logit((uint32_t)r, un, tt, P.tag, P.k0, P.k1);
            const int sr = ld(P.state + r);
            const double Tw = P.temps ? P.temps[sw_of] : T;
            st(P.d0 + (r - q0), (int8_t)(co_decide(F, lgr, Tw, 1.0 / Tw, (uint32_t)r, un, tt, P.tag, P.k0, P.k1) - sr));
        };
        auto signal_c1 = [&]() {  // this wave's stores are complete; the wave that completes the workgroup tells the solvers
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) {
                const unsigned a = atomicAdd(&s_c1, 1u);
                if (a + 1 == ev * (CO_THREADS / 64)) __hip_atomic_fetch_add(&P.bar[PB_C1(blockIdx.x & (PP_SHARDS - 1))], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        };
        // ---- T: once EVERY row of superblock [q0, q1) has its field, its first guesses d0_i = decide(f_i) - s_i are known --
        // each row's owner computes its own with the field (the solvers compute the very same decisions for their masks) --
        // and the triangular pass over them -- the one heavy step of the fixed point, ~40 % of the sites flip -- is a
        // stream and is done here, by the rows' owners: corr_i = sum_{q0 <= j < i} J_ij d0_j.
        auto t_pass = [&](int q0, int q1) -> bool {
            ++tev;
            stage_weights(P.d0, 0, q1 - q0, wl2);
            __syncthreads();
            for (int r = ws; r < n; r += W) {
                if (r < q0 || r >= q1) continue;
                const double acc = wave_dot<TJ, VEC, PP_U>(J + (size_t)r * n, wl2, q0, q0, r, lane);
                if (lane == 0) st(P.corr + r, acc);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) {
                const unsigned a = atomicAdd(&s_t, 1u);
                if (a + 1 == tev * (CO_THREADS / 64)) __hip_atomic_fetch_add(&P.bar[PB_T(blockIdx.x & (PP_SHARDS - 1))], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            return true;
        };
        const bool stiming = P.timeline && blockIdx.x == NS && threadIdx.x == 0;
        unsigned long long stl[4] = {0, 0, 0, 0};
        long long stl_last = wall_clock64();
#define ST_MARK(kind)                                       \
    if (stiming) {                                          \
        const long long now_ = wall_clock64();              \
        stl[kind] += (unsigned long long)(now_ - stl_last); \
        stl_last = now_;                                    \
    }
        for (int sw = 0; sw < P.n_sweeps; ++sw) {
            // a sweep starts from complete fields: computed from scratch every CO_REFRESH sweeps (counted across calls) and at the
            // start of a call -- unless the previous call left them (resume) -- or handed on by the previous sweep
            const bool start_phase = sw == 0 || ((sw + P.refresh_off) % CO_REFRESH) == 0;
            const bool resume_now = sw == 0 && P.resume;
            const bool next_incremental = sw + 1 < P.n_sweeps && ((sw + 1 + P.refresh_off) % CO_REFRESH) != 0;  // priority rows for the next sweep
            const bool keep_all = next_incremental || (sw + 1 == P.n_sweeps && P.persist);                      // every row's field stays current
            if (start_phase) {
                // (from scratch: needs the complete state)
                if (!wg_wait(P.bar, PB_SOLVED, seq * NS, &s_ok)) return;
                if (!resume_now) {
                    stage_weights(P.state, 0, n, wl);
                    __syncthreads();
                }
                ++ev;
                int m = 0;
                for (int r = ws; r < n; r += W, ++m) {
                    const double acc = resume_now ? 0.0 : wave_dot<TJ, VEC>(J + (size_t)r * n, wl, 0, 0, n, lane);
                    if (lane == 0) {
                        const double F = resume_now ? P.fields_all[r] : acc + (P.bias ? P.bias[r] : 0.0);
                        s_f[wv][m] = F;
                        if (r < SB) {
                            st(P.f + r, F);
                            first_guess(r, 0, F, sw);
                        }
                    }
                }
                signal_c1();
                if (!wg_wait_sharded(P.bar, PB_C1(0), BAR_PAGE, ev, NS, (int)gridDim.x, &s_ok)) return;
                if (!t_pass(0, SB < n ? SB : n)) return;
                __syncthreads();  // wl is restaged below
            }
            for (int sb = 0; sb < nsb; ++sb) {
                const int p0 = sb * SB, pe = p0 + SB < n ? p0 + SB : n;
                ++seq;
                ST_MARK(3);
                if (!wg_wait(P.bar, PB_SOLVED, seq * NS, &s_ok)) return;
                ST_MARK(0);
                stage_weights(P.d1, 0, pe - p0, wl);
                __syncthreads();
                ST_MARK(1);
                ++ev;
                // the superblock that is solved next: the following one, or the first of the next sweep when the fields
                // are handed on; rows below row_lo are not needed again in this sweep and start afresh in the next
                const bool last = sb == nsb - 1;
                const int q0 = last ? 0 : pe, q1 = last ? (next_incremental ? (SB < n ? SB : n) : 0) : (pe + SB < n ? pe + SB : n);
                const int row_lo = keep_all ? 0 : pe;
                int m = 0;
                for (int r = ws; r < n; r += W, ++m) {
                    if (r < q0 || r >= q1) continue;
                    const double acc = wave_dot<TJ, VEC, PP_U>(J + (size_t)r * n, wl, p0, p0, pe, lane);
                    if (lane == 0) {
                        const double F = s_f[wv][m] + acc;
                        s_f[wv][m] = F;
                        st(P.f + r, F);
                        first_guess(r, q0, F, last ? sw + 1 : sw);
                    }
                }
                signal_c1();
                // nothing else is started until EVERY workgroup's priority rows are through: other loads would queue in
                // front of the stragglers' (a few waves own two priority rows), and the solvers' start is the critical path
                if (q1 > q0) {
                    if (!wg_wait_sharded(P.bar, PB_C1(0), BAR_PAGE, ev, NS, (int)gridDim.x, &s_ok)) return;
                    if (!t_pass(q0, q1)) return;
                }
                ST_MARK(2);
                m = 0;
                for (int r = ws; r < n; r += W, ++m) {
                    if (r < row_lo || (r >= q0 && r < q1)) continue;
                    const double acc = wave_dot<TJ, VEC, U2>(J + (size_t)r * n, wl, p0, p0, pe, lane);
                    if (lane == 0) s_f[wv][m] += acc;
                }
                __syncthreads();  // every wave is done with wl before it is restaged
            }
        }
        if (P.persist) {  // the fields of the final state, for the next call
            int m = 0;
            for (int r = ws; r < n; r += W, ++m)
                if (lane == 0) P.fields_all[r] = s_f[wv][m];
        }
        if (stiming)
            for (int x = 0; x < 4; ++x) P.timeline[6 + x] = stl[x];
#undef ST_MARK
        return;
    }

    // ---------------------------------------------------------------------------------------------- solver
    // Workgroup b owns the four 64-row groups {b, 31-b, 32+b, 63-b} of the superblock: the work of a group grows with its
    // index (a row reads the toggles of all EARLIER sites), and these quadruples carry the same total.  Waves 0..3 are the
    // groups' row owners (lane = row: field, logit, correction, decision); all 16 waves share the gather of the
    // workgroup's four groups (list entries c = wave mod 16) and hand their partial sums over through LDS.
    const int b = (int)blockIdx.x;
    const int grp[PP_TEAMS] = {b, 2 * NS - 1 - b, 2 * NS + b, 4 * NS - 1 - b};
    const bool owner = wv < PP_TEAMS;
    const int g = grp[wv & (PP_TEAMS - 1)];            // (owners) my group
    unsigned itag = 1;                                 // tag of the next iteration's granules (0 = never written)
    unsigned tev = 0;                                  // triangular passes so far
    const bool timing = P.timeline && blockIdx.x == 0 && threadIdx.x == 0;
    unsigned long long tl[6] = {0, 0, 0, 0, 0, 0};
    long long tl_last = wall_clock64();
#define PP_MARK(kind)                                      \
    if (timing) {                                          \
        const long long now_ = wall_clock64();             \
        tl[kind] += (unsigned long long)(now_ - tl_last);  \
        tl_last = now_;                                    \
    }
    for (int sw = 0; sw < P.n_sweeps; ++sw) {
        const uint32_t t = P.sweep0 + (uint32_t)sw;
        const double* uni = P.uniforms ? P.uniforms + (size_t)sw * n : nullptr;
        const double Tw = P.temps ? P.temps[sw] : T, invTw = 1.0 / Tw;  // (a schedule: one temperature per sweep)
        // the state after this sweep is one of the recorded ones: the commits below write it there too
        int8_t* rec = nullptr;
        if (P.samples && sw + 1 > P.rec_from && (sw + 1 - P.rec_from) % P.rec_every == 0)
            rec = P.samples + (size_t)((sw + 1 - P.rec_from) / P.rec_every - 1) * n;
        if (sw == 0 || ((sw + P.refresh_off) % CO_REFRESH) == 0) ++ev;
        for (int sb = 0; sb < nsb; ++sb) {
            const int p0 = sb * SB, pe = p0 + SB < n ? p0 + SB : n, cnt = pe - p0;
            const int il = 64 * g + lane, i = p0 + il;
            const bool active = owner && il < cnt;
            // my logit: before the fields arrive
            const double lg = active ? co_logit((uint32_t)i, uni, t, P.tag, P.k0, P.k1) : 0.0;
            PP_MARK(3);
            if (!wg_wait_sharded(P.bar, PB_C1(0), BAR_PAGE, ev, NS, (int)gridDim.x, &s_ok)) return;
            PP_MARK(0);
            // state of the superblock at the start of the sweep.  Read AFTER the fields have arrived: with a single superblock
            // its rows were committed by the other solver workgroups just before (the previous sweep's solve), and only the
            // strip that followed -- which waited for all of them -- orders those stores before this read.
            for (int x = threadIdx.x; x < SB; x += CO_THREADS) s_sb[x] = x < cnt ? ld(P.state + p0 + x) : (int8_t)0;
            __syncthreads();
            const double fi = active ? ld(P.f + i) : 0.0;
            const int si = s_sb[il];
            double corr = 0.0;
            bool flipped = active && co_decide(fi, lg, Tw, invTw, (uint32_t)i, uni, t, P.tag, P.k0, P.k1) != si;
            int k = 0;
            while (true) {
                // publish iteration k, then collect every group's iteration-k masks
                unsigned long long* slot = P.masks + (size_t)(k & 1) * GRAN;
                if (owner) {
                    const unsigned long long mk = __ballot(flipped);
                    if (lane < 2) st(slot + 2 * g + lane, (unsigned long long)(uint32_t)(mk >> (32 * lane)) | ((unsigned long long)(itag + (unsigned)k) << 32));
                }
                if (wv == PP_TEAMS) {
                    // (a wave that owns no rows) lane l collects GPL consecutive granules = the 32 GPL sites from
                    // 32 GPL l on; every granule validates itself by its tag
                    unsigned long long gv[GPL];
                    const long long t0 = wall_clock64();
                    for (unsigned spins = 0;; ++spins) {
                        bool okk = true;
#pragma unroll
                        for (int u = 0; u < GPL; ++u) {
                            gv[u] = ld(slot + GPL * lane + u);
                            okk = okk && (unsigned)(gv[u] >> 32) == itag + (unsigned)k;
                        }
                        if (__ballot(!okk) == 0ull) break;
                        if ((spins & 63u) == 63u) {
                            if (ld(&P.bar[BAR_ERR]) || wall_clock64() - t0 > CO_TIMEOUT) {
                                st(&P.bar[BAR_ERR], 1u);
                                s_fail = 1;
                                break;
                            }
                        }
                    }
                    // sorted list of the toggled sites: j | (contribution negative << 15); lane l writes its sites' entries
                    unsigned cur[GPL], tog[GPL];
                    int pc = 0;
#pragma unroll
                    for (int u = 0; u < GPL; ++u) {
                        cur[u] = (unsigned)gv[u];
                        const unsigned old = k ? s_m[(k - 1) & 1][GPL * lane + u] : 0u;
                        s_m[k & 1][GPL * lane + u] = cur[u];
                        tog[u] = cur[u] ^ old;
                        pc += __popc(tog[u]);
                    }
                    int off = pc;
                    for (int d = 1; d < 64; d <<= 1) {
                        const int v = __shfl_up(off, d, 64);
                        if (lane >= d) off += v;
                    }
                    if (lane == 63) s_nlist = off;
                    off -= pc;
#pragma unroll
                    for (int u = 0; u < GPL; ++u) {
                        unsigned tt = tog[u];
                        while (tt) {
                            const int bpos = __ffs((int)tt) - 1;
                            tt &= tt - 1;
                            const int j = 32 * (GPL * lane + u) + bpos;
                            const bool on = (cur[u] >> bpos) & 1u;           // the site is flipped now: + d_j, else - d_j
                            const bool neg = (s_sb[j] != 0) == on;           // d_j = -1 for a site that was 1
                            s_list[off++] = (unsigned short)(j | (neg ? 0x8000 : 0));
                        }
                    }
                }
                __syncthreads();
                if (s_fail) return;
                const int nl = s_nlist;
                PP_MARK(1);
                if (timing) {
                    tl[4] += 1;
                    tl[5] += (unsigned long long)nl;
                }
                if (k > 0 && nl == 0) break;  // nothing toggled anywhere: the fixed point
                if (k == 0) {
                    // iteration 0 -> 1: all first guesses at once = the streamers' triangular pass
                    ++tev;
                    if (!wg_wait_sharded(P.bar, PB_T(0), BAR_PAGE, tev, NS, (int)gridDim.x, &s_ok)) return;
                    if (owner) {
                        corr = active ? ld(P.corr + i) : 0.0;
                        flipped = active && co_decide(fi + corr, lg, Tw, invTw, (uint32_t)i, uni, t, P.tag, P.k0, P.k1) != si;
                    }
                    ++k;
                    PP_MARK(2);
                    continue;
                }
                // the toggles of earlier sites for the workgroup's four groups: PP_BATCH / 4 list entries x 4 groups per round,
                // all loads of a round in flight together (the lists are short after the first guesses: one round trip per
                // iteration).  The list ascends: a wave stops where its entries pass the end of the last group.
                {
                    constexpr int EB = PP_BATCH / PP_TEAMS, NWV = CO_THREADS / 64;
                    double acc[PP_TEAMS] = {0.0, 0.0, 0.0, 0.0};
                    const int j_last = 64 * (grp[PP_TEAMS - 1] + 1);
                    for (int c0 = wv; c0 < nl; c0 += NWV * EB) {
                        if ((int)(s_list[c0] & 0x7FFF) >= j_last) break;
                        TJ x[PP_TEAMS][EB];
                        int e[EB];
#pragma unroll
                        for (int u = 0; u < EB; ++u) {
                            const int c = c0 + u * NWV;
                            e[u] = c < nl ? (int)s_list[c] : 0x7FFF;
                            const int j = e[u] & 0x7FFF;
#pragma unroll
                            for (int gi = 0; gi < PP_TEAMS; ++gi) {
                                // unconditional loads (a predicated one becomes a branch and the loads serial round trips):
                                // entries beyond the group's end read row 0 and are weighted 0 below (j >= j_end > gl)
                                const int gl = 64 * grp[gi] + lane;
                                x[gi][u] = JT[(size_t)(p0 + (j < 64 * (grp[gi] + 1) ? j : 0)) * n + (size_t)(p0 + (gl < cnt ? gl : 0))];
                            }
                        }
#pragma unroll
                        for (int u = 0; u < EB; ++u) {
                            const int j = e[u] & 0x7FFF;
#pragma unroll
                            for (int gi = 0; gi < PP_TEAMS; ++gi) {
                                const double v = (j < 64 * grp[gi] + lane) ? (double)x[gi][u] : 0.0;  // only earlier sites
                                acc[gi] += (e[u] & 0x8000) ? -v : v;
                            }
                        }
                    }
#pragma unroll
                    for (int gi = 0; gi < PP_TEAMS; ++gi) s_red[wv][gi][lane] = acc[gi];
                }
                __syncthreads();
                if (owner) {
                    double tot = 0.0;
#pragma unroll
                    for (int u = 0; u < CO_THREADS / 64; ++u) tot += s_red[u][wv][lane];
                    corr += tot;
                    flipped = active && co_decide(fi + corr, lg, Tw, invTw, (uint32_t)i, uni, t, P.tag, P.k0, P.k1) != si;
                }
                ++k;
                if (k > SB + 2) {  // cannot happen (the iteration is exact after SB rounds): report, leave
                    if (threadIdx.x == 0) st(&P.bar[BAR_ERR + 2], 1u);
                    break;
                }
                __syncthreads();  // s_red and s_list are rewritten in the next round
                PP_MARK(2);
            }
            itag += (unsigned)k + 2u;
            // commit: flips for the streamers' strip, the new state in place (four rows per dword store)
            const int dn = flipped ? (si ? -1 : 1) : 0;
            const int b0 = dn & 0xFF, c0s = (si + dn) & 0xFF;
            unsigned pd = (unsigned)b0, ps = (unsigned)c0s;
            pd |= (unsigned)__shfl_down(b0, 1, 64) << 8 | (unsigned)__shfl_down(b0, 2, 64) << 16 | (unsigned)__shfl_down(b0, 3, 64) << 24;
            ps |= (unsigned)__shfl_down(c0s, 1, 64) << 8 | (unsigned)__shfl_down(c0s, 2, 64) << 16 | (unsigned)__shfl_down(c0s, 3, 64) << 24;
            if (owner && (lane & 3) == 0 && il < cnt) {  // cnt is a multiple of 4 (n is)
                st(reinterpret_cast<unsigned*>(P.d1 + il), pd);
                st(reinterpret_cast<unsigned*>(P.state + i), ps);
                if (rec) *reinterpret_cast<unsigned*>(rec + i) = ps;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            ++seq;
            ++ev;
            if (threadIdx.x == 0) __hip_atomic_fetch_add(&P.bar[PB_SOLVED], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (timing)
        for (int x = 0; x < 6; ++x) P.timeline[x] = tl[x];
#undef PP_MARK
}

template <typename TJ>
static int pipe_sweep(tsu_dense* d, double T, int n_sweeps, uint64_t seed, uint32_t sweep0, uint32_t replica, bool have_uni,
                      int* done, bool fields_were_valid, const double* temps_dev = nullptr, int8_t* samples_dev = nullptr, int rec_from = 0,
                      int rec_every = 1) {
    tsu_ctx* ctx = d->ctx;
    const int n = d->n;
    *done = 0;
    static int use_pipe = -1;
    if (use_pipe < 0) {
        const char* e = getenv("TSU_K2_PIPE");
        use_pipe = e ? atoi(e) : 1;
    }
    int grid = ctx->cus;
    // small systems: fewer streamer workgroups (every workgroup takes part in the hand-off counters, and a system of a few hundred
    // rows has no work for 4000 waves); from 2048 rows up every CU streams.  TSU_K2_PIPE_WG_PER_1024 = workgroups per 1024 rows (0: every CU)
    static int grid_env = -1;
    if (grid_env < 0) {
        const char* e = getenv("TSU_K2_PIPE_WG_PER_1024");
        grid_env = e ? atoi(e) : 128;  // measured (n = 580 / 1024: 36 / 42 us per sweep; with every CU 46 / 49, with 64 per 1024 rows 40 / 44)
    }
    static int sb_env = -1;
    if (sb_env < 0) {
        const char* e = getenv("TSU_K2_PIPE_SB");
        sb_env = e ? atoi(e) : 0;
    }
    // superblock: 4096 (2048 for systems that fit one): re-measured on the finished pipeline (profiles/r02_k2_notes.txt) -- 8192,
    // which had won up to n = 12288 half-way through its development, loses everywhere now
    const int sb = (sb_env == 2048 || sb_env == 4096 || sb_env == 8192) ? sb_env : (n <= 2048 ? 2048 : 4096);
    const int ns = sb / 256;
    if (grid_env > 0) {
        const int want = ns + (int)(((long long)n * grid_env + 1023) / 1024);
        if (want < grid) grid = want;
    }
    static int pipe_min = -1;
    if (pipe_min < 0) {
        const char* e = getenv("TSU_K2_PIPE_MIN");
        pipe_min = e ? atoi(e) : 452;  // just above the one-workgroup kernel (k2_wg: 576 fp32 / 448 fp64 sites): 45-50 us per sweep at
                                       // n = 580 .. 1020 against 50-62 us on the barrier kernel
    }
    if (!use_pipe || n > CO_MAX_N || n < pipe_min || grid < 2 * ns) return TSU_OK;
    if ((long long)(grid - ns) * (CO_THREADS / 64) * PP_MAXR < n) return TSU_OK;
    if (n % 4) return TSU_OK;  // state / flips travel as dwords
    const bool vec = (n % JVec<TJ>::W) == 0;
    // loads in flight per lane on the rows that are not urgent (they run beside the solvers' latency-bound iterations, whose
    // hand-offs and gathers queue behind them): ONE while those rows take less time than the iterations anyway (N = 16384:
    // 0.386 -> 0.353 ms per sweep; profiles/r02_k2_notes.txt), eight once the strips dominate
    const bool gentle = n <= 20480;
    void (*kern)(PipeParams) = sb == 8192 ? (vec ? k2_pipe<TJ, true, 8192, 1> : k2_pipe<TJ, false, 8192, 1>)
                               : sb == 2048 ? (vec ? k2_pipe<TJ, true, 2048, 1> : k2_pipe<TJ, false, 2048, 1>)
                               : gentle   ? (vec ? k2_pipe<TJ, true, 4096, 1> : k2_pipe<TJ, false, 4096, 1>)
                                          : (vec ? k2_pipe<TJ, true, 4096, 8> : k2_pipe<TJ, false, 4096, 8>);
    const size_t lds_bytes = (size_t)((n > sb ? n : sb) + 15) / 16 * 16;
    if (tsu_func_allow_lds(ctx, (const void*)kern, (int)lds_bytes) != hipSuccess) {
        (void)hipGetLastError();
        return TSU_OK;
    }
    int per_cu = 0;
    if (tsu_func_blocks_per_cu(ctx, (const void*)kern, CO_THREADS, lds_bytes, &per_cu) != hipSuccess || per_cu < 1) {
        (void)hipGetLastError();
        return TSU_OK;
    }
    if (!d->co_bar) TSU_HIP_TRY(ctx, hipMalloc(&d->co_bar, BAR_WORDS * sizeof(unsigned)));
    if (!d->co_d1) TSU_HIP_TRY(ctx, hipMalloc(&d->co_d1, (size_t)n));
    if (!d->pp_masks) TSU_HIP_TRY(ctx, hipMalloc(&d->pp_masks, 3 * PP_GRAN_MAX * sizeof(unsigned long long)));
    if (!d->co_corr) TSU_HIP_TRY(ctx, hipMalloc(&d->co_corr, (size_t)n * 8));
    if (!d->co_d0) TSU_HIP_TRY(ctx, hipMalloc(&d->co_d0, (size_t)n));
    if (!d->co_fields) TSU_HIP_TRY(ctx, hipMalloc(&d->co_fields, (size_t)n * 8));
    TSU_HIP_TRY(ctx, hipMemsetAsync(d->co_bar, 0, BAR_WORDS * sizeof(unsigned), ctx->stream));
    TSU_HIP_TRY(ctx, hipMemsetAsync(d->pp_masks, 0, 3 * PP_GRAN_MAX * sizeof(unsigned long long), ctx->stream));
    PipeParams P;
    P.J = d->J;
    P.JT = d->JT;
    P.bias = d->bias;
    P.state = d->state;
    P.f = d->field;
    P.d1 = d->co_d1;
    P.uniforms = have_uni ? d->uniforms : nullptr;
    P.masks = d->pp_masks;
    P.corr = d->co_corr;
    P.d0 = d->co_d0;
    P.bar = d->co_bar;
    P.n = n;
    P.n_sweeps = n_sweeps;
    P.T = T;
    P.sweep0 = sweep0;
    P.tag = TSU_TAG_DENSE | (replica << 8);
    P.k0 = (uint32_t)seed;
    P.k1 = (uint32_t)(seed >> 32);
    // fields from call to call: resume when the previous pipeline call left them for exactly this state and no refresh is due;
    // keep them from the second consecutive call on (a lone call does not pay for rows it would not need again)
    static int keep_fields = -1;
    if (keep_fields < 0) {
        const char* e = getenv("TSU_K2_KEEP_FIELDS");
        keep_fields = e ? atoi(e) : 1;
    }
    P.fields_all = d->co_fields;
    P.temps = temps_dev;
    P.samples = samples_dev;
    P.rec_from = rec_from;
    P.rec_every = rec_every > 0 ? rec_every : 1;
    P.resume = keep_fields && fields_were_valid && (d->since_refresh % CO_REFRESH) != 0 ? 1 : 0;
    P.refresh_off = P.resume ? d->since_refresh : 0;
    P.persist = keep_fields && d->pipe_streak >= 1 ? 1 : 0;
    const char* verbose = getenv("TSU_K2_VERBOSE");
    unsigned long long* d_tl = nullptr;
    if (verbose && atoi(verbose) >= 2) {
        TSU_HIP_TRY(ctx, hipMalloc(&d_tl, 10 * sizeof(unsigned long long)));
        TSU_HIP_TRY(ctx, hipMemsetAsync(d_tl, 0, 10 * sizeof(unsigned long long), ctx->stream));
    }
    P.timeline = d_tl;
    {
        const int rcx = tsu_grid_exclusive_begin(ctx);
        if (rcx != TSU_OK) return rcx;
    }
    hipError_t e = tsu_launch_grid_sync(ctx, (const void*)kern, dim3((unsigned)grid), dim3(CO_THREADS), &P, lds_bytes, ctx->stream);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        if (d_tl) (void)hipFree(d_tl);
        return TSU_OK;
    }
    {
        const int rcx = tsu_grid_exclusive_end(ctx);
        if (rcx != TSU_OK) return rcx;
    }
    unsigned h[4];
    TSU_HIP_TRY(ctx, hipMemcpyAsync(h + 1, d->co_bar + BAR_ERR, 3 * sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
    TSU_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (d_tl) {
        unsigned long long tl[10];
        (void)hipMemcpy(tl, d_tl, sizeof(tl), hipMemcpyDeviceToHost);
        (void)hipFree(d_tl);
        const double nsb_tot = (double)n_sweeps * ((n + sb - 1) / sb);
        fprintf(stderr, "[tsu] k2_pipe n=%d (superblocks of %d), %d sweeps; per superblock: wait C1 %.1f us, poll %.1f us, gather+decide %.1f us, commit+prologue %.1f us, "
                        "%.1f iterations, %.0f toggles\n", n, sb, n_sweeps, tl[0] / 100.0 / nsb_tot, tl[1] / 100.0 / nsb_tot, tl[2] / 100.0 / nsb_tot,
                tl[3] / 100.0 / nsb_tot, tl[4] / nsb_tot, tl[5] / nsb_tot);
        fprintf(stderr, "[tsu]   streamer wave: wait SOLVED %.1f us, stage flips %.1f us, priority rows + signal %.1f us, other rows %.1f us\n",
                tl[6] / 100.0 / nsb_tot, tl[7] / 100.0 / nsb_tot, tl[8] / 100.0 / nsb_tot, tl[9] / 100.0 / nsb_tot);
    }
    if (h[1] || h[3]) {
        // a wait expired (GPU shared with another long-running kernel) or -- never seen -- no fixed point: the caller
        // restores the state from its backup and the other paths take the call
        fprintf(stderr, "[tsu] dense sweep (pipeline): %s; continuing on the barrier path\n", h[1] ? "a wait timed out (GPU shared?)" : "no fixed point");
        d->pp_failed = 1;
        return TSU_OK;
    }
    *done = 1;
    d->n_pipe += 1;
    d->since_refresh = (P.refresh_off + n_sweeps) % CO_REFRESH;
    d->fields_valid = P.persist;
    d->pipe_streak += 1;
    return TSU_OK;
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
    // (the pipeline kernel may have created some of these already)
    if (!d->co_logit) TSU_HIP_TRY(ctx, hipMalloc(&d->co_logit, (size_t)n * 8));
    if (!d->co_corr) TSU_HIP_TRY(ctx, hipMalloc(&d->co_corr, (size_t)n * 8));
    if (!d->co_d0) TSU_HIP_TRY(ctx, hipMalloc(&d->co_d0, (size_t)n));
    if (!d->co_d1) TSU_HIP_TRY(ctx, hipMalloc(&d->co_d1, (size_t)n));
    if (!d->co_lists) TSU_HIP_TRY(ctx, hipMalloc(&d->co_lists, 2 * SB_SIZE * sizeof(int)));
    if (!d->co_bar) TSU_HIP_TRY(ctx, hipMalloc(&d->co_bar, BAR_WORDS * sizeof(unsigned)));
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

// the owner-computes kernel (dense_own.hip) for one state: same contract as pipe_sweep
static int own_try(tsu_dense* d, double T, int n_sweeps, uint64_t seed, uint32_t sweep0, uint32_t replica, bool have_uni, const int64_t* order_dev,
                   const double* temps_dev, int8_t* samples_dev, int rec_from, int rec_every, bool fields_were_valid, int* done) {
    OwnRep rep;
    rep.T = T;
    rep.sweep0 = sweep0;
    rep.tag = TSU_TAG_DENSE | (replica << 8);
    rep.k0 = (uint32_t)seed;
    rep.k1 = (uint32_t)(seed >> 32);
    return tsu_dense_own_run(d, 1, &rep, d->state, n_sweeps, have_uni ? d->uniforms : nullptr, order_dev, temps_dev, samples_dev, rec_from, rec_every,
                             fields_were_valid, true, done);
}

// a whole run (burn-in, then n_samples x n_sweeps sweeps with the state recorded after each group; or an annealing schedule: one
// temperature per sweep) in ONE launch -- the owner-computes kernel first, then the pipeline; *done = 0: neither takes it (the
// caller's loop of calls does).  order_dev: the caller's visiting orders ([n_total][n], owner-computes kernel only) or nullptr.
int tsu_dense_pipe_run(tsu_dense* d, double T, const double* temps_dev, int n_total, int rec_from, int rec_every, int8_t* samples_dev,
                       uint64_t seed, uint32_t sweep0, uint32_t replica, bool have_uni, int* done, const int64_t* order_dev) {
    *done = 0;
    bool fields_were_valid = d->fields_valid != 0;
    const int streak = d->pipe_streak;
    d->fields_valid = 0;
    d->rep_match = 0;
    d->pipe_streak = 0;
    if (n_total <= 0) return TSU_OK;
    if (!d->own_failed) {
        d->pipe_streak = streak;
        const int rc = own_try(d, T, n_total, seed, sweep0, replica, have_uni, order_dev, temps_dev, samples_dev, rec_from, rec_every, fields_were_valid, done);
        if (!*done) d->pipe_streak = 0;
        if (rc != TSU_OK || *done) return rc;
        if (d->own_failed) {  // it ran and gave up half way: back to the state at the start of the call (the caller's backup)
            TSU_HIP_TRY(d->ctx, hipMemcpyAsync(d->state, d->backup, (size_t)d->n, hipMemcpyDeviceToDevice, d->ctx->stream));
            fields_were_valid = false;
        }
    }
    if (d->pp_failed || order_dev) return TSU_OK;
    d->pipe_streak = streak;
    const int rc = d->dtype == TSU_DTYPE_F64
                       ? pipe_sweep<double>(d, T, n_total, seed, sweep0, replica, have_uni, done, fields_were_valid, temps_dev, samples_dev, rec_from, rec_every)
                       : pipe_sweep<float>(d, T, n_total, seed, sweep0, replica, have_uni, done, fields_were_valid, temps_dev, samples_dev, rec_from, rec_every);
    if (!*done) d->pipe_streak = 0;
    return rc;
}

int tsu_dense_coop_sweep(tsu_dense* d, double T, int n_sweeps, uint64_t seed, uint32_t sweep0, uint32_t replica, bool have_uni,
                         int* done, const int64_t* order_dev) {
    // first choice: the owner-computes kernel, then the two-role pipeline (both in place on d->state); they decline small / odd systems
    // and report a failed run with *done = 0, in which case the state is restored here and the barrier kernel below takes the call
    // (whatever happens below, the fields kept from the last call stop being those of d->state; a successful call sets the flag again)
    bool fields_were_valid = d->fields_valid != 0;
    const int streak = d->pipe_streak;
    d->fields_valid = 0;
    d->rep_match = 0;
    d->pipe_streak = 0;
    *done = 0;
    if (!d->own_failed) {
        d->pipe_streak = streak;
        const int rc = own_try(d, T, n_sweeps, seed, sweep0, replica, have_uni, order_dev, nullptr, nullptr, 0, 1, fields_were_valid, done);
        if (!*done) d->pipe_streak = 0;
        if (rc != TSU_OK || *done) return rc;
        if (d->own_failed) {
            TSU_HIP_TRY(d->ctx, hipMemcpyAsync(d->state, d->backup, (size_t)d->n, hipMemcpyDeviceToDevice, d->ctx->stream));
            fields_were_valid = false;
        }
    }
    if (order_dev) return TSU_OK;  // the other one-launch kernels run in natural order only
    if (!d->pp_failed) {
        d->pipe_streak = streak;
        const int rc = d->dtype == TSU_DTYPE_F64 ? pipe_sweep<double>(d, T, n_sweeps, seed, sweep0, replica, have_uni, done, fields_were_valid)
                                                 : pipe_sweep<float>(d, T, n_sweeps, seed, sweep0, replica, have_uni, done, fields_were_valid);
        if (!*done) d->pipe_streak = 0;
        if (rc != TSU_OK || *done) return rc;
        if (d->pp_failed)  // it ran and gave up half way: back to the state at the start of the call (the caller's backup)
            TSU_HIP_TRY(d->ctx, hipMemcpyAsync(d->state, d->backup, (size_t)d->n, hipMemcpyDeviceToDevice, d->ctx->stream));
    }
    if (d->dtype == TSU_DTYPE_F64) return coop_sweep<double>(d, T, n_sweeps, seed, sweep0, replica, have_uni, done);
    return coop_sweep<float>(d, T, n_sweeps, seed, sweep0, replica, have_uni, done);
}
