// dense.hip -- K2 sequential single-site heat-bath Gibbs on a dense coupling matrix (gfx950).
//
// Replaces GibbsSampler.gibbs_sweep on a dense J (tsu/gibbs.py:128-162): for every site in visiting
// order h_i = J[i,:].s + b_i (INCLUDING J_ii s_i, gibbs.py:97), p = sigmoid(h_i/T) with the +-20 clamp
// (gibbs.py:73-77), s_i <- (u < p) (gibbs.py:126).  The visiting order is the reference's (range(n) or
// the caller's permutation); what is parallel is the arithmetic, not the Markov chain.
//
// Paths, all with the same results (dispatch in tsu_dense_sweep), natural order by system size:
//   * k2_small (n <= 192 fp32 J / 128 fp64 J): a whole call (or sample_boltzmann run, annealing schedule, tempering
//     ladder) in one launch of a single wave, J in LDS;
//   * k2_wg (n <= 528 / 448): one workgroup, a thread per site, J^T columns from L2;
//   * larger: dense_coop.hip -- the whole call in one cooperative launch (superblock fixed point);
//   * the same fixed point with one launch per iteration (k2_sb_iter / k2_sb_finish) when a cooperative launch is
//     not available;
//   * k2_block: per sweep the field f = J s + b (k2_matvec), then the visiting order in blocks of 64 positions, one
//     launch per block in which every workgroup gathers the 64 x 64 sub-block into LDS, a wave resolves the 64
//     sites in order by jumping from flip to flip (ballot + ffs: a lane's pending decision only changes when an
//     earlier site of the block actually flips), and the block's flips are added to the other fields through rows
//     of J^T.  Serves custom visiting orders (np.random.permutation), tiny systems and the never-seen case of a
//     superblock that did not converge.
//
// Uniforms: replayed doubles from the host (bit-exact replay of np.random.rand) or Philox doubles keyed by
// (site, sweep): a = W[2(i&1)] >> 5, b = W[2(i&1)+1] >> 6, u = (a 2^26 + b) / 2^53 with
// W = Philox4x32-10(ctr = (i >> 1, 0, sweep, TAG_DENSE | replica << 8), key = seed).
#include <algorithm>
#include <vector>

#include "dense.h"

template <typename TJ>
__global__ __launch_bounds__(256) void k2_matvec(const TJ* __restrict__ J, const int8_t* __restrict__ s,
                                                const double* __restrict__ bias, double* __restrict__ f, int n) {
    // one wave per row; lanes stride the row
    int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= n) return;
    const TJ* row = J + (size_t)wave * n;
    double acc = 0.0;
    for (int j = lane; j < n; j += 64) acc += (double)row[j] * (double)s[j];
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (lane == 0) f[wave] = acc + (bias ? bias[wave] : 0.0);
}

// One block of 64 visiting-order positions per launch, ALL workgroups: every workgroup gathers the 64 x 64
// sub-block and resolves the block redundantly (deterministic, so every copy agrees; the chain is latency-bound and
// the other CUs would only wait), then applies the block's flips to its own slice of the remaining fields.  This
// halves the launches of the resolve + propagate pair and removes the single-workgroup kernel from the chain.
template <typename TJ>
__global__ __launch_bounds__(256) void k2_block(const TJ* __restrict__ J, const TJ* __restrict__ JT,
                                               const int8_t* __restrict__ s, int8_t* __restrict__ s_new,
                                               double* __restrict__ f, const int64_t* __restrict__ order,
                                               const double* __restrict__ uniforms, int n, int pos0, int cnt, double T,
                                               uint32_t sweep, uint32_t tag, uint32_t k0, uint32_t k1, int last_block) {
    __shared__ double sub[DB][DB + 1];  // sub[i][k] = J[site_k][site_i]
    __shared__ int sites[DB];
    __shared__ int sh_flips[DB];
    __shared__ int sh_nflip;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < DB) sites[tid] = tid < cnt ? (order ? (int)order[pos0 + tid] : pos0 + tid) : -1;
    __syncthreads();
    // gather: thread (wave, lane) loads rows i = wave, wave + 4, ... for column site_lane; all 16 loads are issued
    // before the first one is consumed (one L2/HBM latency per block instead of sixteen)
    {
        const int site = sites[lane];
        TJ v[DB / 4];
#pragma unroll
        for (int r = 0; r < DB / 4; ++r) {
            const int i = wave + 4 * r;
            v[r] = (lane < cnt && i < cnt) ? J[(size_t)site * n + sites[i]] : (TJ)0;
        }
#pragma unroll
        for (int r = 0; r < DB / 4; ++r) {
            const int i = wave + 4 * r;
            if (lane < cnt && i < cnt) sub[i][lane] = (double)v[r];
        }
    }
    __syncthreads();
    if (wave == 0) {
        const int site = sites[lane];
        double fk = 0.0, u = 2.0;
        int bit = 0;
        if (lane < cnt) {
            fk = f[site];
            bit = s[site];
            u = uniforms ? uniforms[pos0 + lane] : dense_uniform((uint32_t)site, sweep, tag, k0, k1);
        }
        // u < sigmoid(x) <=> x > logit(u): the logit is computed once per site, so the flip-to-flip chain below
        // carries one compare instead of a float64 exp.  The reference's own expression decides whenever x is
        // within a safety margin of the logit (and beyond the +-20 clamp, gibbs.py:73-76), so outcomes are unchanged.
        const double logit = (lane < cnt) ? (log(u) - log1p(-u)) : 0.0;
        const double margin = 1e-9 * (1.0 + fabs(logit));
        int nflip = 0, cursor = 0;
        const double invT = 1.0 / T;
        while (true) {
            const double xa = fk * invT;  // approximate x = fk / T: only used where it is decisive by a wide margin
            int cand;
            if (fabs(fabs(xa) - 20.0) < 1e-9 || fabs(xa - logit) <= margin) cand = (u < sigmoid_clamped(fk / T)) ? 1 : 0;
            else if (xa > 20.0) cand = 1;  // p = 1.0 > u
            else if (xa < -20.0) cand = 0;
            else cand = xa > logit ? 1 : 0;
            if (lane >= cnt) cand = 0;
            unsigned long long want = __ballot((lane >= cursor) && (lane < cnt) && (cand != bit));
            if (want == 0ull) break;
            int i = __ffsll((long long)want) - 1;
            int delta = __shfl(cand - bit, i, 64);
            if (lane == i) bit = cand;
            if (lane < cnt) fk += (double)delta * sub[i][lane];
            if (lane == 0) sh_flips[nflip] = (sites[i] << 1) | (delta > 0 ? 1 : 0);
            ++nflip;
            cursor = i + 1;
        }
        if (lane == 0) sh_nflip = nflip;
        // one copy publishes the block's new bits -- into the NEXT-state array: the other workgroups may still be
        // reading this block's old bits.  The block's own fields are not needed again before the next sweep's
        // field pass, and nobody writes them during this launch (the slice update below skips the block).
        if (blockIdx.x == 0 && lane < cnt) s_new[site] = (int8_t)bit;
    }
    __syncthreads();
    const int nflip = sh_nflip;
    if (nflip == 0 || last_block) return;
    for (int j = blockIdx.x * blockDim.x + tid; j < n; j += gridDim.x * blockDim.x) {
        bool inside = false;
        if (!order) inside = (j >= pos0 && j < pos0 + cnt);
        else
            for (int k = 0; k < cnt; ++k) inside |= (sites[k] == j);
        if (inside) continue;
        // eight independent loads in flight per step: the flips are few (tens), so this loop is latency-bound
        double acc = 0.0;
        for (int k0 = 0; k0 < nflip; k0 += 8) {
            TJ v[8];
            double w[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int k = k0 + q < nflip ? k0 + q : nflip - 1;
                const int fl = sh_flips[k];
                v[q] = JT[(size_t)(fl >> 1) * n + j];
                w[q] = k0 + q < nflip ? ((fl & 1) ? 1.0 : -1.0) : 0.0;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) acc += w[q] * (double)v[q];
        }
        f[j] += acc;
    }
}

// ================================================================== small systems: one wave, one launch per run
// n <= 64 M sites (M = 1..3 register slots per lane), natural order: the whole sample_boltzmann run (burn-in, then
// n_samples x n_sweeps sweeps with a recorded state after each group) inside ONE wave.  Lane k owns sites k, 64 + k,
// 128 + k; J sits in LDS as columns (fp32 J: up to 192 sites = 148 KB, fp64: up to 128); every sweep recomputes the
// fields from scratch (like the reference's np.dot per site) and resolves the sequential pass slot after slot by a
// fixed-point iteration over the slot's 64 sites.  No barrier, no launch and no global memory access inside a sweep.
#define K2W_MAX_SLOTS 3
// u < sigmoid(x) <=> x > logit(u): a float32 logit is computed once per site and sweep, outside the flip-to-flip
// chain, so a decision inside the chain is one multiply, a conversion and a few float compares.  The float quantities
// carry errors below 1e-5 (1 + |logit|); whenever x is within 1e-4 (1 + |logit|) of the logit, or within 1e-3 of the
// +-20 clamp, the reference's float64 expression (gibbs.py:73-77,126) decides, so outcomes are its own.
static __device__ __forceinline__ float k2w_logit(double u) {
    return __logf((float)u) - __logf((float)(1.0 - u));  // 1 - u in float64: no cancellation for u near 1
}
static __device__ __forceinline__ int k2w_decide(double fk, double u, float lg, double T, double invT) {
    const float xf = (float)(fk * invT);
    if (fabsf(fabsf(xf) - 20.0f) < 1e-3f || !(fabsf(xf - lg) > 1e-4f * (1.0f + fabsf(lg)))) return (u < sigmoid_clamped(fk / T)) ? 1 : 0;
    if (xf > 20.0f) return 1;
    if (xf < -20.0f) return 0;
    return xf > lg ? 1 : 0;
}

template <typename TJ, int M>
static __device__ __forceinline__ void k2_wave_body(const TJ* __restrict__ J, const double* __restrict__ bias, int8_t* __restrict__ state,
                                                    const double* __restrict__ uniforms, int8_t* __restrict__ samples, int n, double T,
                                                    const double* __restrict__ temps, int n_burnin, int n_sweeps, int n_samples,
                                                    uint32_t sweep0, uint32_t tag, uint32_t k0, uint32_t k1) {
    extern __shared__ unsigned char k2w_lds[];
    constexpr int STRIDE = 64 * M + 1;  // col[i * STRIDE + k] = J[k][i]; odd stride: the transposing stores do not conflict
    TJ* col = reinterpret_cast<TJ*>(k2w_lds);
    const int lane = threadIdx.x;
    for (int k = 0; k < n; ++k) {  // row k of J, read coalesced, stored as column entries
#pragma unroll
        for (int t = 0; t < M; ++t) {
            const int i = t * 64 + lane;
            if (i < n) col[i * STRIDE + k] = J[(size_t)k * n + i];
        }
    }
    for (int k = n; k < 64 * ((n + 63) / 64); ++k) {  // entries the idle lanes of the last slot read
#pragma unroll
        for (int t = 0; t < M; ++t) {
            const int i = t * 64 + lane;
            if (i < n) col[i * STRIDE + k] = (TJ)0;
        }
    }
    __syncthreads();
    bool on[M];
    double bk[M], fk[M], u[M], u_next[M];
    int bit[M];
    const int total = n_burnin + n_samples * n_sweeps;
#pragma unroll
    for (int t = 0; t < M; ++t) {
        const int site = t * 64 + lane;
        on[t] = site < n;
        bk[t] = (on[t] && bias) ? bias[site] : 0.0;
        bit[t] = on[t] ? state[site] : 0;
        u_next[t] = (on[t] && uniforms && total > 0) ? uniforms[site] : 2.0;
    }
    double invT = 1.0 / T;  // or one temperature per sweep (an annealing schedule)
    int next_record = n_burnin + n_sweeps - 1, rec = 0;
    for (int sw = 0; sw < total; ++sw) {
        if (temps) {
            T = temps[sw];
            invT = 1.0 / T;
        }
        unsigned long long up[M];
#pragma unroll
        for (int t = 0; t < M; ++t) {
            const int site = t * 64 + lane;
            u[t] = 0.5;  // idle lanes: any value with a finite logit (they never take part in a ballot)
            if (on[t]) u[t] = uniforms ? u_next[t] : dense_uniform((uint32_t)site, sweep0 + (uint32_t)sw, tag, k0, k1);
            if (on[t] && uniforms && sw + 1 < total) u_next[t] = uniforms[(size_t)(sw + 1) * n + site];  // in flight during this sweep
            up[t] = __ballot(bit[t] != 0);
            fk[t] = bk[t];
        }
#pragma unroll
        for (int tt = 0; tt < M; ++tt) {
            const int cnt = n - tt * 64 < 64 ? n - tt * 64 : 64;
            // ascending site index; branch-free (a site that is down adds +0.0) so that the LDS reads of several sites are in
            // flight together instead of one read latency per site
#pragma unroll 8
            for (int ii = 0; ii < cnt; ++ii) {
                const double b = ((up[tt] >> ii) & 1ull) ? 1.0 : 0.0;
                const TJ* c = col + (tt * 64 + ii) * STRIDE + lane;
#pragma unroll
                for (int t = 0; t < M; ++t) fk[t] += b * (double)c[t * 64];
            }
        }
#pragma unroll
        for (int s = 0; s < M; ++s) {
            // The sequential pass over the 64 sites of slot s is the unique solution of a triangular system: delta_k =
            // decide(f_k + sum_{j<k} J_kj delta_j) - bit_k.  It is solved by fixed-point iteration with all lanes
            // deciding at once: lane k is exact from iteration k on, in practice a handful of iterations suffice, and the
            // column reads of an iteration's changes are in flight together instead of one LDS latency per flip.
            const float lg = k2w_logit(u[s]);
            int d_prev = 0;     // this lane's current candidate flip (new bit - old bit)
            double corr = 0.0;  // sum over the earlier lanes' candidate flips of delta_j J[k][j]
            for (int iter = 0; iter < 66 && s * 64 < n; ++iter) {
                const int d_new = on[s] ? k2w_decide(fk[s] + corr, u[s], lg, T, invT) - bit[s] : 0;
                unsigned long long chg = __ballot(d_new != d_prev);
                if (chg == 0ull) break;
                const int diff = d_new - d_prev;
                d_prev = d_new;
                while (chg) {  // four changed sites per round: their column entries are read together
                    int j[4], dd[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        j[q] = chg ? __ffsll((long long)chg) - 1 : 0;
                        dd[q] = chg ? __builtin_amdgcn_readlane(diff, j[q]) : 0;
                        chg &= chg - 1ull;
                    }
                    TJ c[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) c[q] = col[(s * 64 + j[q]) * STRIDE + s * 64 + lane];
#pragma unroll
                    for (int q = 0; q < 4; ++q) corr += (lane > j[q]) ? (double)dd[q] * (double)c[q] : 0.0;
                }
            }
            bit[s] += d_prev;
            if (s + 1 < M && (s + 1) * 64 < n) {  // the later slots see every flip of this one
                unsigned long long flips = __ballot(d_prev != 0);
                while (flips) {
                    int j[4], dd[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        j[q] = flips ? __ffsll((long long)flips) - 1 : 0;
                        dd[q] = flips ? __builtin_amdgcn_readlane(d_prev, j[q]) : 0;
                        flips &= flips - 1ull;
                    }
#pragma unroll
                    for (int t = s + 1; t < M; ++t) {
                        TJ c[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) c[q] = col[(s * 64 + j[q]) * STRIDE + t * 64 + lane];
#pragma unroll
                        for (int q = 0; q < 4; ++q) fk[t] += (double)dd[q] * (double)c[q];
                    }
                }
            }
        }
        if (sw == next_record) {
#pragma unroll
            for (int t = 0; t < M; ++t)
                if (on[t]) samples[(size_t)rec * n + t * 64 + lane] = (int8_t)bit[t];
            ++rec;
            next_record += n_sweeps;
        }
    }
#pragma unroll
    for (int t = 0; t < M; ++t)
        if (on[t]) state[t * 64 + lane] = (int8_t)bit[t];
}

template <typename TJ, int M>
__global__ __launch_bounds__(64) void k2_small(const TJ* __restrict__ J, const double* __restrict__ bias, int8_t* __restrict__ state,
                                              const double* __restrict__ uniforms, int8_t* __restrict__ samples, int n, double T,
                                              const double* __restrict__ temps, int n_burnin, int n_sweeps, int n_samples,
                                              uint32_t sweep0, uint32_t tag, uint32_t k0, uint32_t k1) {
    k2_wave_body<TJ, M>(J, bias, state, uniforms, samples, n, T, temps, n_burnin, n_sweeps, n_samples, sweep0, tag, k0, k1);
}

// the replicas of a tempering ladder: block r sweeps state r at its own temperature with its own stream
struct K2Replica {
    double T;
    uint32_t sweep0, tag, k0, k1;
};

template <typename TJ, int M>
__global__ __launch_bounds__(64) void k2_small_replicas(const TJ* __restrict__ J, const double* __restrict__ bias, int8_t* __restrict__ states,
                                                       const double* __restrict__ uniforms, const K2Replica* __restrict__ reps, int n,
                                                       int n_sweeps) {
    const K2Replica rp = reps[blockIdx.x];
    k2_wave_body<TJ, M>(J, bias, states + (size_t)blockIdx.x * n, uniforms ? uniforms + (size_t)blockIdx.x * n_sweeps * n : nullptr, nullptr, n,
                        rp.T, nullptr, n_sweeps, 1, 0, rp.sweep0, rp.tag, rp.k0, rp.k1);
}

// largest n on the one-workgroup kernel: the measured crossover against the cooperative kernel
// (tools/dense_mid_times.py): fp32 J ~600 sites, fp64 J ~470 against the barrier kernel; against the pipeline (which now starts
// at n = 452 with a streamer grid sized to the system) fp32 ~530: n = 500 30 us here / 34 there, n = 576 41 / 36; TSU_K2_WG=n overrides (0: never)
static bool k2wg_takes(const tsu_dense* d) {
    static int use_wg = -2;
    if (use_wg == -2) {
        const char* e = getenv("TSU_K2_WG");
        use_wg = e ? atoi(e) : -1;
        if (use_wg > 1024) use_wg = 1024;
    }
    return d->n <= (use_wg >= 0 ? use_wg : (d->dtype == TSU_DTYPE_F64 ? 448 : 528));
}

// slots per lane the one-wave kernels need for this system, 0 if it does not fit one CU's LDS (TSU_K2_WAVE=0: at most 1)
static int k2w_slots(const tsu_dense* d) {
    static int enabled = -1;
    if (enabled < 0) {
        const char* e = getenv("TSU_K2_WAVE");
        enabled = e ? atoi(e) : 1;
    }
    const int m = (d->n + 63) / 64;
    const int cap = !enabled ? 1 : (d->dtype == TSU_DTYPE_F64 ? 2 : K2W_MAX_SLOTS);
    return m <= cap ? m : 0;
}

static size_t k2w_lds_bytes(const tsu_dense* d, int m) {
    return (size_t)d->n * (size_t)(64 * m + 1) * (d->dtype == TSU_DTYPE_F64 ? 8 : 4);
}

typedef void (*k2w_run_fn)(const void*, const double*, int8_t*, const double*, int8_t*, int, double, const double*, int, int, int, uint32_t,
                           uint32_t, uint32_t, uint32_t);
typedef void (*k2w_rep_fn)(const void*, const double*, int8_t*, const double*, const K2Replica*, int, int);

// one launch of the one-wave kernel (grid 1) for a system k2w_slots accepts
static hipError_t k2w_launch(const tsu_dense* d, int m, hipStream_t stream, const double* uniforms, int8_t* samples, double T,
                             const double* temps, int n_burnin, int n_sweeps, int n_samples, uint32_t sweep0, uint32_t tag, uint32_t k0,
                             uint32_t k1) {
    static const k2w_run_fn table[2][K2W_MAX_SLOTS] = {
        {(k2w_run_fn)k2_small<float, 1>, (k2w_run_fn)k2_small<float, 2>, (k2w_run_fn)k2_small<float, 3>},
        {(k2w_run_fn)k2_small<double, 1>, (k2w_run_fn)k2_small<double, 2>, nullptr}};
    const int ti = d->dtype == TSU_DTYPE_F64 ? 1 : 0;
    const k2w_run_fn fn = table[ti][m - 1];
    {
        hipError_t e = tsu_func_allow_lds(d->ctx, (const void*)fn, 160 * 1024);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(fn, dim3(1), dim3(64), k2w_lds_bytes(d, m), stream, (const void*)d->J, (const double*)d->bias, d->state, uniforms, samples,
                       d->n, T, temps, n_burnin, n_sweeps, n_samples, sweep0, tag, k0, k1);
    return hipGetLastError();
}

static hipError_t k2w_launch_replicas(const tsu_dense* d, int m, hipStream_t stream, int n_replicas, int8_t* states, const double* uniforms,
                                      const K2Replica* reps, int n_sweeps) {
    static const k2w_rep_fn table[2][K2W_MAX_SLOTS] = {
        {(k2w_rep_fn)k2_small_replicas<float, 1>, (k2w_rep_fn)k2_small_replicas<float, 2>, (k2w_rep_fn)k2_small_replicas<float, 3>},
        {(k2w_rep_fn)k2_small_replicas<double, 1>, (k2w_rep_fn)k2_small_replicas<double, 2>, nullptr}};
    const int ti = d->dtype == TSU_DTYPE_F64 ? 1 : 0;
    const k2w_rep_fn fn = table[ti][m - 1];
    {
        hipError_t e = tsu_func_allow_lds(d->ctx, (const void*)fn, 160 * 1024);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(fn, dim3((unsigned)n_replicas), dim3(64), k2w_lds_bytes(d, m), stream, (const void*)d->J, (const double*)d->bias, states,
                       uniforms, reps, d->n, n_sweeps);
    return hipGetLastError();
}

// ================================================================== mid-size systems: one workgroup, one launch per call
// 192 (128) < n <= 1024, natural order: thread i owns site i (field, logit, correction, bit in registers), J^T is read
// from global memory (L2-resident: 4 MB at n = 1024) one COLUMN per changed site -- coalesced across the threads, no
// reductions.  A sweep is the fixed point delta = decide(f + L delta) - s over the whole system as one block (site k is
// exact from iteration k on; about a dozen iterations in practice): every iteration lists the sites whose decision
// changed (ascending, so floating-point sums do not depend on timing) and every thread adds their column entries with
// j < i to its correction.  The flips are then added to every field (fields are handed from sweep to sweep and
// recomputed from scratch every K2WG_REFRESH sweeps).  An iteration costs two workgroup barriers and an L2 round trip instead
// of a grid barrier: n = 256 14 us per sweep against 43 us on the grid-wide path, n = 448 27 us against 46 us.
#define K2WG_MAX_N 1024
#define K2WG_REFRESH 64

struct K2wgList {
    int j[K2WG_MAX_N + 64];  // padded with (site 0, value 0) up to the next multiple of 64 entries
    int d[K2WG_MAX_N + 64];
    alignas(16) int wcount[16];  // list entries per wave (0 for waves the workgroup does not have)
};

// every thread calls it; sites with pred set are listed in ascending order with their value; returns the list length
static __device__ __forceinline__ int k2wg_build(K2wgList& L, bool pred, int value, int site, int wave, int lane) {
    const unsigned long long b = __ballot(pred);
    if (lane == 0) L.wcount[wave] = __popcll(b);
    __syncthreads();
    // all sixteen counts with four 16-byte LDS reads (waves that do not exist keep their 0): a loop over the waves paid
    // one LDS latency per wave
    int wc[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int4 v = reinterpret_cast<const int4*>(L.wcount)[q];
        wc[4 * q] = v.x; wc[4 * q + 1] = v.y; wc[4 * q + 2] = v.z; wc[4 * q + 3] = v.w;
    }
    int before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        before += w < wave ? wc[w] : 0;
        total += wc[w];
    }
    if (pred) {
        const int at = before + __popcll(b & ((1ull << lane) - 1ull));
        L.j[at] = site;
        L.d[at] = value;
    }
    if (threadIdx.x < 64) {  // padding: the readers take whole groups of 64 entries
        L.j[total + threadIdx.x] = 0;
        L.d[total + threadIdx.x] = 0;
    }
    __syncthreads();
    return total;
}

// one entry of J^T through a buffer descriptor: scalar row offset (the listed site) + one per-lane column offset for
// every load of a batch, so a load in flight holds nothing but its result register
typedef unsigned int k2wg_v2u __attribute__((ext_vector_type(2)));
template <typename TJ>
struct K2wgLoad;
template <>
struct K2wgLoad<float> {
    static __device__ __forceinline__ float ld(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
    }
};
template <>
struct K2wgLoad<double> {
    static __device__ __forceinline__ double ld(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
        return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
    }
};

// acc += sum over the listed sites j (only j < i when TRI) of d_j * JT[j][i], ascending j.  Lane l of every wave picks up
// entry e0 + l of the (padded) list with one LDS read per 64 entries; the entries then reach the scalar unit by
// v_readlane, and a batch of 32 column loads is in flight per thread (the loop is bound by the L2 round trip).
template <typename TJ, bool TRI>
static __device__ __forceinline__ double k2wg_apply(__amdgpu_buffer_rsrc_t JT, const K2wgList& L, int cnt, int n, int i, int lane, bool on,
                                                    double acc) {
    constexpr int B = 32;
    const int voff = (on ? i : 0) * (int)sizeof(TJ);  // idle threads read a valid entry and never use it
    const int row_bytes = n * (int)sizeof(TJ);
    for (int e0 = 0; e0 < cnt; e0 += 64) {
        const int my_j = L.j[e0 + lane], my_d = L.d[e0 + lane];
#pragma unroll
        for (int h = 0; h < 64; h += B) {
            if (e0 + h >= cnt) break;
            TJ x[B];
#pragma unroll
            for (int g = 0; g < B; g += 8) {  // groups of eight: a short list (most iterations) issues few loads
                if (e0 + h + g < cnt) {
#pragma unroll
                    for (int q = g; q < g + 8; ++q) x[q] = K2wgLoad<TJ>::ld(JT, voff, __builtin_amdgcn_readlane(my_j, h + q) * row_bytes);
                }
            }
#pragma unroll
            for (int g = 0; g < B; g += 8) {
                if (e0 + h + g < cnt) {
#pragma unroll
                    for (int q = g; q < g + 8; ++q) {
                        const int j = __builtin_amdgcn_readlane(my_j, h + q), dd = __builtin_amdgcn_readlane(my_d, h + q);
                        acc += (!TRI || j < i) ? (double)dd * (double)x[q] : 0.0;
                    }
                }
            }
        }
    }
    return acc;
}

template <typename TJ>
__global__ __launch_bounds__(1024) void k2_wg(const TJ* __restrict__ JTp, const double* __restrict__ bias, int8_t* __restrict__ state,
                                             const double* __restrict__ uniforms, int8_t* __restrict__ samples, int n, double T,
                                             const double* __restrict__ temps, int n_burnin, int n_per_sample, int n_samples,
                                             uint32_t sweep0, uint32_t tag, uint32_t k0, uint32_t k1, unsigned* __restrict__ info,
                                             const K2Replica* __restrict__ reps) {
    // n_burnin sweeps, then n_samples x (n_per_sample sweeps, record the state) -- a plain sweep call is (n, 1, 0);
    // temps: one temperature per sweep (an annealing schedule) or NULL; reps: workgroup r sweeps state r of a tempering
    // ladder at its own temperature with its own stream (its uniforms follow those of replica r - 1)
    __shared__ K2wgList L;
    if (threadIdx.x < 16) L.wcount[threadIdx.x] = 0;
    __syncthreads();
    if (reps) {
        const K2Replica rp = reps[blockIdx.x];
        T = rp.T;
        sweep0 = rp.sweep0;
        tag = rp.tag;
        k0 = rp.k0;
        k1 = rp.k1;
        state += (size_t)blockIdx.x * n;
        if (uniforms) uniforms += (size_t)blockIdx.x * n_burnin * n;
    }
    const __amdgpu_buffer_rsrc_t JT = __builtin_amdgcn_make_buffer_rsrc(const_cast<TJ*>(JTp), 0, n * n * (int)sizeof(TJ), 0x00020000);
    const int i = threadIdx.x, lane = i & 63, wave = i >> 6;
    const bool on = i < n;
    const double b = (on && bias) ? bias[i] : 0.0;
    double invT = 1.0 / T;
    int bit = on ? state[i] : 0;
    double f = 0.0;
    int worst = 0;
    const int n_sweeps = n_burnin + n_samples * n_per_sample;
    int next_record = n_burnin + n_per_sample - 1, rec = 0;
    for (int sw = 0; sw < n_sweeps; ++sw) {
        if (temps) {
            T = temps[sw];
            invT = 1.0 / T;
        }
        if (sw % K2WG_REFRESH == 0) {  // fields from scratch: b + sum over the sites that are up of their column
            const int cnt = k2wg_build(L, on && bit != 0, 1, i, wave, lane);
            f = k2wg_apply<TJ, false>(JT, L, cnt, n, i, lane, on, b);
        }
        double u = 0.5;
        if (on) u = uniforms ? uniforms[(size_t)sw * n + i] : dense_uniform((uint32_t)i, sweep0 + (uint32_t)sw, tag, k0, k1);
        const float lg = k2w_logit(u);
        int d_prev = 0;
        double corr = 0.0;
        int iter = 0;
        for (; iter < n + 2; ++iter) {
            const int d_new = on ? k2w_decide(f + corr, u, lg, T, invT) - bit : 0;
            const int cnt = k2wg_build(L, d_new != d_prev, d_new - d_prev, i, wave, lane);
            if (cnt == 0) break;
            d_prev = d_new;
            corr = k2wg_apply<TJ, true>(JT, L, cnt, n, i, lane, on, corr);
        }
        if (iter > worst) worst = iter;
        bit += d_prev;
        if (samples && sw == next_record) {
            if (on) samples[(size_t)rec * n + i] = (int8_t)bit;
            ++rec;
            next_record += n_per_sample;
        }
        if (sw + 1 < n_sweeps && (sw + 1) % K2WG_REFRESH != 0) {  // hand the fields on: every flip, every row
            const int cnt = k2wg_build(L, d_prev != 0, d_prev, i, wave, lane);
            f = k2wg_apply<TJ, false>(JT, L, cnt, n, i, lane, on, f);
        }
    }
    if (on) state[i] = (int8_t)bit;
    if (i == 0 && info) info[0] = (unsigned)worst;
}

static hipError_t k2wg_launch(const tsu_dense* d, hipStream_t stream, const double* uniforms, int8_t* samples, double T, const double* temps,
                               int n_burnin, int n_per_sample, int n_samples, uint32_t sweep0, uint32_t tag, uint32_t k0, uint32_t k1,
                               unsigned* info, int n_replicas = 1, int8_t* states = nullptr, const K2Replica* reps = nullptr) {
    const unsigned threads = (unsigned)((d->n + 63) / 64 * 64);
    int8_t* st = reps ? states : d->state;
    if (d->dtype == TSU_DTYPE_F64)
        k2_wg<double><<<(unsigned)n_replicas, threads, 0, stream>>>((const double*)d->JT, d->bias, st, uniforms, samples, d->n, T, temps, n_burnin,
                                                                  n_per_sample, n_samples, sweep0, tag, k0, k1, info, reps);
    else
        k2_wg<float><<<(unsigned)n_replicas, threads, 0, stream>>>((const float*)d->JT, d->bias, st, uniforms, samples, d->n, T, temps, n_burnin,
                                                                 n_per_sample, n_samples, sweep0, tag, k0, k1, info, reps);
    return hipGetLastError();
}

__global__ __launch_bounds__(1024) void k2_energy(const double* __restrict__ f, const int8_t* __restrict__ s,
                                                 const double* __restrict__ bias, double* __restrict__ out, int n) {
    // f = J s + b  =>  -1/2 s.(f - b) - b.s.  ONE workgroup, sums in a fixed order (strided per thread, a shuffle tree per wave, the
    // sixteen waves in order): two evaluations of the same state give the same bits, so that comparisons between states of
    // equal energy (simulated_annealing's running minimum, gibbs.py:384-391) do not depend on the order in which atomics land
    __shared__ double part[16];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 1024) {
        double b = bias ? bias[i] : 0.0, si = (double)s[i];
        acc += -0.5 * si * (f[i] - b) - b * si;
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int w = 0; w < 16; ++w) tot += part[w];
        *out = tot;
    }
}

// ================================================================== superblock fixed-point resolve
// The sequential sweep is the unique solution of a triangular system: delta_i = dec_i(f_i + sum_{j<i} J_ij delta_j).
// For a superblock of S consecutive positions that system is solved by fixed-point iteration with ALL sites of the
// block updated in parallel: iteration m is one row-parallel triangular matvec over the S x S sub-block (which
// stays in L2) plus one decision per site.  Site i is exact from iteration i on, so the iteration reaches the
// sequential result; in practice the intra-block corrections are small against the decision gaps and it converges
// in about ten iterations.  A launch finds "no site changed in the previous iteration" and returns at once, so a
// fixed launch budget costs little; if the budget is exhausted without convergence the whole call is re-run on the
// block-by-block path (k2_block) from a backup of the state.  Natural visiting order only.
#define SB_MAX_IT 32  // slots per superblock; the launch budget adapts below this

template <typename TJ>
__global__ __launch_bounds__(256) void k2_sb_iter(const TJ* __restrict__ J, const int8_t* __restrict__ s,
                                                 const double* __restrict__ f, const double* __restrict__ uniforms,
                                                 const int8_t* __restrict__ din, int8_t* __restrict__ dout,
                                                 double* __restrict__ logit, int* __restrict__ sync, int m, int n, int p0,
                                                 int cnt, double T, uint32_t sweep, uint32_t tag, uint32_t k0, uint32_t k1) {
    if (m > 0 && sync[m - 1] == 0) return;  // the previous iteration changed nothing: delta is the fixed point
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= cnt) return;
    const int site = p0 + wave;
    double c = 0.0;
    if (m > 0) {  // iteration 0 starts from delta = 0
        const TJ* row = J + (size_t)site * n + p0;
        for (int j = lane; j < wave; j += 64) {
            const int dj = din[j];
            if (dj) c += (double)dj * (double)row[j];
        }
        for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    }
    if (lane == 0) {
        double lg;
        double u = 0.0;
        if (m == 0) {
            u = uniforms ? uniforms[site] : dense_uniform((uint32_t)site, sweep, tag, k0, k1);
            lg = log(u) - log1p(-u);
            logit[wave] = lg;
        } else {
            lg = logit[wave];
        }
        const double F = f[site] + c;
        const double xa = F * (1.0 / T);
        int cand;
        if (fabs(fabs(xa) - 20.0) < 1e-9 || fabs(xa - lg) <= 1e-9 * (1.0 + fabs(lg))) {
            if (m > 0) u = uniforms ? uniforms[site] : dense_uniform((uint32_t)site, sweep, tag, k0, k1);
            cand = (u < sigmoid_clamped(F / T)) ? 1 : 0;  // the reference's own expression decides close calls
        } else if (xa > 20.0) cand = 1;
        else if (xa < -20.0) cand = 0;
        else cand = xa > lg ? 1 : 0;
        const int8_t dn = (int8_t)(cand - (int)s[site]);
        const int8_t dp = m > 0 ? din[wave] : (int8_t)0;
        dout[wave] = dn;
        if (dn != dp) atomicAdd(&sync[m], 1);
    }
}

// commit the superblock (new bits into the next-state array, convergence flag) and add its flips to the fields of
// every later site: one wave per later row, reading the S contiguous entries J[row][p0 .. p0+cnt)
template <typename TJ>
__global__ __launch_bounds__(256) void k2_sb_finish(const TJ* __restrict__ J, const int8_t* __restrict__ s,
                                                   int8_t* __restrict__ s_new, double* __restrict__ f,
                                                   const int8_t* __restrict__ d0, const int8_t* __restrict__ d1,
                                                   int* __restrict__ sync, int n, int p0, int cnt, int budget) {
    const int gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    // the fixed point sits in the buffer iteration mc wrote, mc = first iteration that changed nothing (later launches
    // returned without writing); without convergence take the last buffer written (the host re-runs the call anyway)
    int mc = budget - 1;
    for (int m = budget - 1; m >= 0; --m)
        if (sync[m] == 0) mc = m;
    const int8_t* __restrict__ dfinal = (mc & 1) ? d1 : d0;
    if (blockIdx.x == 0) {
        for (int i = threadIdx.x; i < cnt; i += blockDim.x) s_new[p0 + i] = (int8_t)(s[p0 + i] + dfinal[i]);
        if (threadIdx.x == 0) {
            int ok = 0;
            for (int m = 0; m < budget; ++m) ok |= (sync[m] == 0);
            sync[SB_MAX_IT] = ok;
        }
    }
    const int row = p0 + cnt + gw;
    if (row >= n) return;
    const TJ* jr = J + (size_t)row * n + p0;
    double c = 0.0;
    for (int j = lane; j < cnt; j += 64) {
        const int dj = dfinal[j];
        if (dj) c += (double)dj * (double)jr[j];
    }
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    if (lane == 0) f[row] += c;
}

template <typename TJ>
static int dense_sweep_superblocks(tsu_dense* d, double T, int n_sweeps, uint64_t seed, uint32_t sweep0, uint32_t replica,
                                   bool have_uni, int* converged) {
    if (d->sb_budget < 16) d->sb_budget = 20;
    const int budget = d->sb_budget;
    tsu_ctx* ctx = d->ctx;
    const TJ* J = (const TJ*)d->J;
    const int n = d->n;
    const uint32_t tag = TSU_TAG_DENSE | (replica << 8);
    const int nsb = (n + SB_SIZE - 1) / SB_SIZE;
    const size_t sync_ints = (size_t)nsb * n_sweeps * (SB_MAX_IT + 1);
    if (!d->delta[0]) {
        TSU_HIP_TRY(ctx, hipMalloc(&d->delta[0], SB_SIZE));
        TSU_HIP_TRY(ctx, hipMalloc(&d->delta[1], SB_SIZE));
        TSU_HIP_TRY(ctx, hipMalloc(&d->logit, SB_SIZE * sizeof(double)));
    }
    if ((size_t)d->sb_cap < sync_ints) {
        if (d->sb_sync) (void)hipFree(d->sb_sync);
        d->sb_sync = nullptr;
        d->sb_cap = 0;
        TSU_HIP_TRY(ctx, hipMalloc(&d->sb_sync, sync_ints * sizeof(int)));
        d->sb_cap = (int)sync_ints;
    }
    TSU_HIP_TRY(ctx, hipMemsetAsync(d->sb_sync, 0, sync_ints * sizeof(int), ctx->stream));
    TSU_HIP_TRY(ctx, hipMemcpyAsync(d->backup, d->state, (size_t)n, hipMemcpyDeviceToDevice, ctx->stream));
    const unsigned mv_grid = (unsigned)(((size_t)n * 64 + 255) / 256);
    int* sync = d->sb_sync;
    for (int s = 0; s < n_sweeps; ++s) {
        TSU_HIP_TRY(ctx, hipMemcpyAsync(d->state2, d->state, (size_t)n, hipMemcpyDeviceToDevice, ctx->stream));
        k2_matvec<TJ><<<mv_grid, 256, 0, ctx->stream>>>(J, d->state, d->bias, d->field, n);
        const double* uni = have_uni ? d->uniforms + (size_t)s * n : nullptr;
        for (int p0 = 0; p0 < n; p0 += SB_SIZE, sync += SB_MAX_IT + 1) {
            const int cnt = n - p0 < SB_SIZE ? n - p0 : SB_SIZE;
            const unsigned it_grid = (unsigned)((cnt * 64 + 255) / 256);
            for (int m = 0; m < budget; ++m)
                k2_sb_iter<TJ><<<it_grid, 256, 0, ctx->stream>>>(J, d->state, d->field, uni, d->delta[(m + 1) & 1], d->delta[m & 1],
                                                                 d->logit, sync, m, n, p0, cnt, T, sweep0 + (uint32_t)s, tag,
                                                                 (uint32_t)seed, (uint32_t)(seed >> 32));
            const int later = n - (p0 + cnt);
            const unsigned fin_grid = (unsigned)(((size_t)(later > 0 ? later : 1) * 64 + 255) / 256);
            k2_sb_finish<TJ><<<fin_grid, 256, 0, ctx->stream>>>(J, d->state, d->state2, d->field, d->delta[0], d->delta[1], sync, n, p0,
                                                               cnt, budget);
        }
        int8_t* t = d->state;
        d->state = d->state2;
        d->state2 = t;
    }
    TSU_HIP_TRY(ctx, hipGetLastError());
    // one synchronisation per call: did every superblock reach its fixed point within the launch budget?
    std::vector<int> h(sync_ints);
    TSU_HIP_TRY(ctx, hipMemcpyAsync(h.data(), d->sb_sync, sync_ints * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    TSU_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *converged = 1;
    int worst = 0;
    for (size_t b = 0; b < (size_t)nsb * n_sweeps; ++b) {
        if (!h[b * (SB_MAX_IT + 1) + SB_MAX_IT]) *converged = 0;
        int mc = budget;
        for (int m = budget - 1; m >= 0; --m)
            if (h[b * (SB_MAX_IT + 1) + m] == 0) mc = m;
        if (mc > worst) worst = mc;
    }
    // adapt the launch budget to this system (temperature, coupling strength): slowest fixed point + margin
    // (an unused launch exits at once and costs ~3 us; a failed attempt costs the whole call again)
    d->sb_budget = *converged ? (worst + 8 < 16 ? 16 : (worst + 8 > SB_MAX_IT ? SB_MAX_IT : worst + 8)) : SB_MAX_IT;
    if (getenv("TSU_K2_VERBOSE"))
        fprintf(stderr, "[tsu] dense superblocks: n=%d, %d superblocks x %d sweeps, slowest fixed point after %d iterations (budget %d, next %d)\n",
                n, nsb, n_sweeps, worst, budget, d->sb_budget);
    return TSU_OK;
}

template <typename TJ>
static int dense_sweep_impl(tsu_dense* d, double T, int n_sweeps, bool have_order, uint64_t seed, uint32_t sweep0,
                            uint32_t replica, bool have_uni) {
    tsu_ctx* ctx = d->ctx;
    const TJ* J = (const TJ*)d->J;
    const TJ* JT = (const TJ*)d->JT;
    int n = d->n;
    uint32_t tag = TSU_TAG_DENSE | (replica << 8);
    unsigned mv_grid = (unsigned)(((size_t)n * 64 + 255) / 256), pg_grid = (unsigned)((n + 255) / 256);
    for (int s = 0; s < n_sweeps; ++s) {
        TSU_HIP_TRY(ctx, hipMemcpyAsync(d->state2, d->state, (size_t)n, hipMemcpyDeviceToDevice, ctx->stream));
        k2_matvec<TJ><<<mv_grid, 256, 0, ctx->stream>>>(J, d->state, d->bias, d->field, n);
        const int64_t* ord = have_order ? d->order + (size_t)s * n : nullptr;
        const double* uni = have_uni ? d->uniforms + (size_t)s * n : nullptr;
        unsigned blk_grid = pg_grid < (unsigned)(2 * ctx->cus) ? pg_grid : (unsigned)(2 * ctx->cus);
        for (int pos0 = 0; pos0 < n; pos0 += DB) {
            int cnt = n - pos0 < DB ? n - pos0 : DB;
            int last = pos0 + cnt >= n;
            k2_block<TJ><<<last ? 1u : blk_grid, 256, 0, ctx->stream>>>(J, JT, d->state, d->state2, d->field, ord, uni, n, pos0,
                                                                       cnt, T, sweep0 + (uint32_t)s, tag, (uint32_t)seed,
                                                                       (uint32_t)(seed >> 32), last);
        }
        int8_t* t = d->state;
        d->state = d->state2;
        d->state2 = t;
    }
    TSU_HIP_TRY(ctx, hipGetLastError());
    return TSU_OK;
}

extern "C" {

int tsu_dense_create(tsu_ctx* ctx, int n, const void* J_host, int dtype, const double* bias_host, tsu_dense** out) {
    TSU_ENTER(ctx);
    if (!ctx || !out) return TSU_E_INVALID;
    *out = nullptr;
    TSU_REQUIRE(ctx, n >= 1 && J_host, "Coupling matrix must be square");
    TSU_REQUIRE(ctx, dtype == TSU_DTYPE_F64 || dtype == TSU_DTYPE_F32, "dense_create: bad dtype %d", dtype);
    tsu_dense* d = new (std::nothrow) tsu_dense();
    if (!d) return tsu_fail(ctx, TSU_E_NOMEM, "dense_create: host allocation failed");
    memset(d, 0, sizeof(*d));
    d->ctx = ctx;
    d->n = n;
    d->dtype = dtype;
    size_t esz = dtype == TSU_DTYPE_F64 ? 8 : 4, jb = (size_t)n * n * esz;
    // symmetric J: J^T aliases J; otherwise keep a transposed copy for the coalesced propagate pass
    bool sym = true;
    if (dtype == TSU_DTYPE_F64) {
        const double* A = (const double*)J_host;
        for (int i = 0; i < n && sym; ++i)
            for (int j = i + 1; j < n; ++j)
                if (A[(size_t)i * n + j] != A[(size_t)j * n + i]) { sym = false; break; }
    } else {
        const float* A = (const float*)J_host;
        for (int i = 0; i < n && sym; ++i)
            for (int j = i + 1; j < n; ++j)
                if (A[(size_t)i * n + j] != A[(size_t)j * n + i]) { sym = false; break; }
    }
    hipError_t e = hipMalloc(&d->J, jb);
    if (e == hipSuccess) e = hipMemcpyAsync(d->J, J_host, jb, hipMemcpyHostToDevice, ctx->stream);
    void* hostT = nullptr;
    if (e == hipSuccess && !sym) {
        hostT = malloc(jb);
        if (!hostT) e = hipErrorOutOfMemory;
        else {
            if (dtype == TSU_DTYPE_F64) {
                const double* A = (const double*)J_host; double* B = (double*)hostT;
                for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) B[(size_t)j * n + i] = A[(size_t)i * n + j];
            } else {
                const float* A = (const float*)J_host; float* B = (float*)hostT;
                for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) B[(size_t)j * n + i] = A[(size_t)i * n + j];
            }
            e = hipMalloc(&d->JT, jb);
            if (e == hipSuccess) e = hipMemcpyAsync(d->JT, hostT, jb, hipMemcpyHostToDevice, ctx->stream);
        }
    } else if (e == hipSuccess) {
        d->JT = d->J;
    }
    if (e == hipSuccess && bias_host) {
        e = hipMalloc(&d->bias, (size_t)n * 8);
        if (e == hipSuccess) e = hipMemcpyAsync(d->bias, bias_host, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream);
    }
    if (e == hipSuccess) e = hipMalloc(&d->state, (size_t)n);
    if (e == hipSuccess) e = hipMemsetAsync(d->state, 0, (size_t)n, ctx->stream);
    if (e == hipSuccess) e = hipMalloc(&d->state2, (size_t)n);
    if (e == hipSuccess) e = hipMalloc(&d->field, (size_t)n * 8);
    if (e == hipSuccess) e = hipMalloc(&d->d_energy, 8);
    if (e == hipSuccess) e = hipMalloc(&d->backup, (size_t)n);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (hostT) free(hostT);
    if (e != hipSuccess) {
        int rc = tsu_fail(ctx, e == hipErrorOutOfMemory ? TSU_E_NOMEM : TSU_E_HIP, "dense_create: %s", hipGetErrorString(e));
        tsu_dense_destroy(d);
        return rc;
    }
    *out = d;
    return TSU_OK;
}

int tsu_dense_destroy(tsu_dense* d) {
    TSU_ENTER(d ? d->ctx : nullptr);
    if (!d) return TSU_OK;
    (void)hipStreamSynchronize(d->ctx->stream);
    if (d->JT && d->JT != d->J) (void)hipFree(d->JT);
    if (d->J) (void)hipFree(d->J);
    if (d->bias) (void)hipFree(d->bias);
    if (d->state) (void)hipFree(d->state);
    if (d->state2) (void)hipFree(d->state2);
    if (d->field) (void)hipFree(d->field);
    if (d->order) (void)hipFree(d->order);
    if (d->uniforms) (void)hipFree(d->uniforms);
    if (d->d_energy) (void)hipFree(d->d_energy);
    if (d->delta[0]) (void)hipFree(d->delta[0]);
    if (d->delta[1]) (void)hipFree(d->delta[1]);
    if (d->logit) (void)hipFree(d->logit);
    if (d->sb_sync) (void)hipFree(d->sb_sync);
    if (d->backup) (void)hipFree(d->backup);
    if (d->samples) (void)hipFree(d->samples);
    if (d->temps) (void)hipFree(d->temps);
    if (d->rep_buf) (void)hipFree(d->rep_buf);
    if (d->rep_fields[0]) (void)hipFree(d->rep_fields[0]);
    if (d->rep_fields[1]) (void)hipFree(d->rep_fields[1]);
    free(d->rep_prev);
    if (d->h_flags) (void)hipHostFree(d->h_flags);
    if (d->h_stage) (void)hipHostFree(d->h_stage);
    if (d->h_rep) (void)hipHostFree(d->h_rep);
    if (d->co_logit) (void)hipFree(d->co_logit);
    if (d->co_corr) (void)hipFree(d->co_corr);
    if (d->co_d0) (void)hipFree(d->co_d0);
    if (d->co_d1) (void)hipFree(d->co_d1);
    if (d->co_lists) (void)hipFree(d->co_lists);
    if (d->co_counts) (void)hipFree(d->co_counts);
    if (d->co_bar) (void)hipFree(d->co_bar);
    if (d->pp_masks) (void)hipFree(d->pp_masks);
    if (d->co_fields) (void)hipFree(d->co_fields);
    if (d->own_gran) (void)hipFree(d->own_gran);
    delete d;
    return TSU_OK;
}

// every byte 0 or 1?  (eight at a time: the byte-by-byte check of eight replicas' states took 0.1 ms per call at n = 16384)
static bool dense_bits01(const int8_t* p, size_t n) {
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        uint64_t w;
        memcpy(&w, p + i, 8);
        if (w & 0xFEFEFEFEFEFEFEFEull) return false;
    }
    for (; i < n; ++i)
        if (p[i] != 0 && p[i] != 1) return false;
    return true;
}

// the handle's pinned staging buffer (n bytes for a state, 8 more for an energy), made on first use; nullptr: copy directly
static int8_t* dense_stage(tsu_dense* d) {
    if (!d->h_stage && hipHostMalloc((void**)&d->h_stage, (((size_t)d->n + 7) / 8) * 8 + 8, hipHostMallocDefault) != hipSuccess) {
        d->h_stage = nullptr;
        (void)hipGetLastError();
    }
    return d->h_stage;
}

int tsu_dense_set_state(tsu_dense* d, const int8_t* bits_host) {
    TSU_ENTER(d ? d->ctx : nullptr);
    if (!d) return TSU_E_INVALID;
    TSU_REQUIRE(d->ctx, bits_host != nullptr, "dense_set_state: NULL");
    TSU_REQUIRE(d->ctx, dense_bits01(bits_host, (size_t)d->n), "dense_set_state: state must be 0/1");
    TSU_HIP_TRY(d->ctx, hipMemcpyAsync(d->state, bits_host, (size_t)d->n, hipMemcpyHostToDevice, d->ctx->stream));
    TSU_HIP_TRY(d->ctx, hipStreamSynchronize(d->ctx->stream));
    d->fields_valid = 0;  // the pipeline's kept fields belong to the old state
    d->pipe_streak = 0;
    // ... but the state may be one the last replica call returned (a tempering loop asks for every replica's energy between its
    // sweeps, gibbs.py:303-323): its fields are still there
    d->rep_match = 0;
    for (int p = 0; p < d->rep_prev_n && !d->rep_match; ++p)
        if (memcmp(bits_host, d->rep_prev + (size_t)p * d->n, (size_t)d->n) == 0) d->rep_match = p + 1;
    return TSU_OK;
}

int tsu_dense_get_state(tsu_dense* d, int8_t* bits_host) {
    TSU_ENTER(d ? d->ctx : nullptr);
    if (!d) return TSU_E_INVALID;
    TSU_REQUIRE(d->ctx, bits_host != nullptr, "dense_get_state: NULL");
    int8_t* stage = dense_stage(d);
    TSU_HIP_TRY(d->ctx, hipMemcpyAsync(stage ? stage : bits_host, d->state, (size_t)d->n, hipMemcpyDeviceToHost, d->ctx->stream));
    TSU_HIP_TRY(d->ctx, hipStreamSynchronize(d->ctx->stream));
    if (stage) memcpy(bits_host, stage, (size_t)d->n);
    return TSU_OK;
}

int tsu_dense_sweep(tsu_dense* d, double T, int n_sweeps, const int64_t* order, uint64_t seed, uint32_t sweep0,
                    uint32_t replica, const double* replay_uniforms) {
    TSU_ENTER(d ? d->ctx : nullptr);
    if (!d) return TSU_E_INVALID;
    tsu_ctx* ctx = d->ctx;
    TSU_REQUIRE(ctx, T > 0.0, "Temperature must be positive");
    TSU_REQUIRE(ctx, n_sweeps >= 0, "dense_sweep: n_sweeps must be >= 0");
    if (n_sweeps == 0) return TSU_OK;
    size_t cnt = (size_t)n_sweeps * d->n;
    if (order) {
        if (d->order_cap < cnt) {
            if (d->order) (void)hipFree(d->order);
            d->order = nullptr;
            d->order_cap = 0;
            TSU_HIP_TRY(ctx, hipMalloc(&d->order, cnt * 8));
            d->order_cap = cnt;
        }
        // (the upload runs while the host checks the rows; nothing reads the device copy before the synchronisation below)
        TSU_HIP_TRY(ctx, hipMemcpyAsync(d->order, order, cnt * 8, hipMemcpyHostToDevice, ctx->stream));
        // every sweep must visit every site exactly once (np.random.permutation, gibbs.py:157): the sweep reads the
        // frozen current state for "old" bits, which is only right for a site's first visit
        std::vector<int> seen((size_t)d->n, -1);  // the row that saw the site last: one pass, no clearing per row
        bool ok = true;
        int bad_row = -1;
        for (int sw = 0; sw < n_sweeps && ok; ++sw) {
            const int64_t* row = order + (size_t)sw * d->n;
            for (int i = 0; i < d->n; ++i) {
                const long long v = row[i];
                if (v < 0 || v >= d->n || seen[(size_t)v] == sw) {
                    ok = false;
                    bad_row = sw;
                    break;
                }
                seen[(size_t)v] = sw;
            }
        }
        if (!ok) {
            (void)hipStreamSynchronize(ctx->stream);
            return tsu_fail(ctx, TSU_E_INVALID, "dense_sweep: order row %d is not a permutation of 0..%d", bad_row, d->n - 1);
        }
    }
    if (replay_uniforms) {
        if (d->uni_cap < cnt) {
            if (d->uniforms) (void)hipFree(d->uniforms);
            d->uniforms = nullptr;
            d->uni_cap = 0;
            TSU_HIP_TRY(ctx, hipMalloc(&d->uniforms, cnt * 8));
            d->uni_cap = cnt;
        }
        TSU_HIP_TRY(ctx, hipMemcpyAsync(d->uniforms, replay_uniforms, cnt * 8, hipMemcpyHostToDevice, ctx->stream));
    }
    if (order || replay_uniforms) TSU_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // host buffers are the caller's
    if (!order) {
        // systems of at most 192 (fp32) / 128 (fp64) sites: all sweeps of the call in one launch of a single wave
        const int wave_m = k2w_slots(d);
        if (wave_m) {
            TSU_HIP_TRY(ctx, k2w_launch(d, wave_m, ctx->stream, replay_uniforms ? d->uniforms : nullptr, nullptr, T, nullptr, n_sweeps, 1, 0, sweep0,
                                        TSU_TAG_DENSE | (replica << 8), (uint32_t)seed, (uint32_t)(seed >> 32)));
            return TSU_OK;
        }
    }
    if (!order && k2wg_takes(d)) {
        // mid-size systems: one workgroup, thread per site, all sweeps of the call in one launch
        static const bool verbose_wg = getenv("TSU_K2_VERBOSE") != nullptr;
        TSU_HIP_TRY(ctx, k2wg_launch(d, ctx->stream, replay_uniforms ? d->uniforms : nullptr, nullptr, T, nullptr, n_sweeps, 1, 0, sweep0,
                                     TSU_TAG_DENSE | (replica << 8), (uint32_t)seed, (uint32_t)(seed >> 32),
                                     verbose_wg ? (unsigned*)d->d_energy : nullptr));
        if (verbose_wg) {
            unsigned w = 0;
            TSU_HIP_TRY(ctx, hipMemcpyAsync(&w, d->d_energy, 4, hipMemcpyDeviceToHost, ctx->stream));
            TSU_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            fprintf(stderr, "[tsu] k2_wg n=%d: %d sweeps, slowest fixed point %u iterations\n", d->n, n_sweeps, w);
        }
        return TSU_OK;
    }
    if (order && d->n >= 2 * DB) {
        // a caller's visiting order (update_order="random"): the owner-computes kernel follows it in one launch (dense_own.hip);
        // it declines small systems and reports a failed run with done = 0 -- then the block-by-block path below takes the call
        TSU_HIP_TRY(ctx, hipMemcpyAsync(d->backup, d->state, (size_t)d->n, hipMemcpyDeviceToDevice, ctx->stream));
        int done = 0;
        const int rc = tsu_dense_coop_sweep(d, T, n_sweeps, seed, sweep0, replica, replay_uniforms != nullptr, &done, d->order);
        if (rc != TSU_OK) return rc;
        if (done) return TSU_OK;
    }
    static int use_sb = -1, use_coop = -1;
    if (use_sb < 0) {
        const char* e = getenv("TSU_K2_SUPERBLOCK");
        use_sb = e ? atoi(e) : 1;
        e = getenv("TSU_K2_COOP");
        use_coop = e ? atoi(e) : 1;
    }
    // natural order: the whole call in one cooperative launch; if that is unavailable, the same fixed point with one
    // launch per iteration; the block-by-block path serves custom orders, tiny systems and a superblock that did not
    // converge within its iteration slots
    if (!(use_coop && !d->co_disabled && !order && d->n >= 2 * DB)) {  // another path writes the state: the pipeline's kept fields go stale
        d->fields_valid = 0;
        d->rep_match = 0;
        d->pipe_streak = 0;
    }
    if (use_coop && !d->co_disabled && !order && d->n >= 2 * DB) {
        TSU_HIP_TRY(ctx, hipMemcpyAsync(d->backup, d->state, (size_t)d->n, hipMemcpyDeviceToDevice, ctx->stream));
        int done = 0;
        int rc = tsu_dense_coop_sweep(d, T, n_sweeps, seed, sweep0, replica, replay_uniforms != nullptr, &done);
        if (rc != TSU_OK) return rc;
        if (done) return TSU_OK;
        TSU_HIP_TRY(ctx, hipMemcpyAsync(d->state, d->backup, (size_t)d->n, hipMemcpyDeviceToDevice, ctx->stream));
    }
    if (use_sb && !order && d->n >= 2 * DB) {
        int ok = 0;
        int rc = d->dtype == TSU_DTYPE_F64
                     ? dense_sweep_superblocks<double>(d, T, n_sweeps, seed, sweep0, replica, replay_uniforms != nullptr, &ok)
                     : dense_sweep_superblocks<float>(d, T, n_sweeps, seed, sweep0, replica, replay_uniforms != nullptr, &ok);
        if (rc != TSU_OK) return rc;
        if (!ok) {
            // a superblock ran out of iteration launches: restore the state and retry once with the full budget
            TSU_HIP_TRY(ctx, hipMemcpyAsync(d->state, d->backup, (size_t)d->n, hipMemcpyDeviceToDevice, ctx->stream));
            rc = d->dtype == TSU_DTYPE_F64
                     ? dense_sweep_superblocks<double>(d, T, n_sweeps, seed, sweep0, replica, replay_uniforms != nullptr, &ok)
                     : dense_sweep_superblocks<float>(d, T, n_sweeps, seed, sweep0, replica, replay_uniforms != nullptr, &ok);
            if (rc != TSU_OK) return rc;
        }
        if (ok) return TSU_OK;
        // still not converged within SB_MAX_IT iterations: redo the call on the exact block-by-block path
        TSU_HIP_TRY(ctx, hipMemcpyAsync(d->state, d->backup, (size_t)d->n, hipMemcpyDeviceToDevice, ctx->stream));
    }
    if (d->dtype == TSU_DTYPE_F64)
        return dense_sweep_impl<double>(d, T, n_sweeps, order != nullptr, seed, sweep0, replica, replay_uniforms != nullptr);
    return dense_sweep_impl<float>(d, T, n_sweeps, order != nullptr, seed, sweep0, replica, replay_uniforms != nullptr);
}

// shared by tsu_dense_sample (one temperature) and tsu_dense_anneal (temps: one temperature per sweep, host array)
static int dense_run(tsu_dense* d, double T, const double* temps, int n_burnin, int n_sweeps, int n_samples, const int64_t* order,
                     uint64_t seed, uint32_t sweep0, uint32_t replica, const double* replay_uniforms, int8_t* samples_host) {
    tsu_ctx* ctx = d->ctx;
    TSU_REQUIRE(ctx, T > 0.0, "Temperature must be positive");
    TSU_REQUIRE(ctx, n_burnin >= 0 && n_sweeps > 0 && n_samples >= 0, "dense_sample: need n_burnin >= 0, n_sweeps > 0, n_samples >= 0");
    TSU_REQUIRE(ctx, n_samples == 0 || samples_host != nullptr, "dense_sample: NULL output");
    const long long total = (long long)n_burnin + (long long)n_samples * n_sweeps;
    TSU_REQUIRE(ctx, total <= (1ll << 30) && (uint64_t)sweep0 + (uint64_t)total <= (1ull << 32), "dense_sample: sweep counter overflow");
    const int n = d->n;
    const size_t out_bytes = (size_t)n_samples * n;
    if (d->samples_cap < out_bytes) {
        if (d->samples) (void)hipFree(d->samples);
        d->samples = nullptr;
        d->samples_cap = 0;
        TSU_HIP_TRY(ctx, hipMalloc(&d->samples, out_bytes));
        d->samples_cap = out_bytes;
    }
    const int wave_m = order ? 0 : k2w_slots(d);
    const bool wg = !order && !wave_m && k2wg_takes(d);
    if (wave_m || wg) {
        const double* temps_dev = nullptr;
        if (temps && total > 0) {
            if (d->temps_cap < (size_t)total) {
                if (d->temps) (void)hipFree(d->temps);
                d->temps = nullptr;
                d->temps_cap = 0;
                TSU_HIP_TRY(ctx, hipMalloc(&d->temps, (size_t)total * 8));
                d->temps_cap = (size_t)total;
            }
            TSU_HIP_TRY(ctx, hipMemcpyAsync(d->temps, temps, (size_t)total * 8, hipMemcpyHostToDevice, ctx->stream));
            temps_dev = d->temps;
        }
        const size_t cnt = (size_t)total * n;
        if (replay_uniforms && cnt) {
            if (d->uni_cap < cnt) {
                if (d->uniforms) (void)hipFree(d->uniforms);
                d->uniforms = nullptr;
                d->uni_cap = 0;
                TSU_HIP_TRY(ctx, hipMalloc(&d->uniforms, cnt * 8));
                d->uni_cap = cnt;
            }
            TSU_HIP_TRY(ctx, hipMemcpyAsync(d->uniforms, replay_uniforms, cnt * 8, hipMemcpyHostToDevice, ctx->stream));
        }
        const uint32_t tag = TSU_TAG_DENSE | (replica << 8);
        if (total > 0 && wave_m)
            TSU_HIP_TRY(ctx, k2w_launch(d, wave_m, ctx->stream, replay_uniforms ? d->uniforms : nullptr, d->samples, T, temps_dev, n_burnin, n_sweeps,
                                        n_samples, sweep0, tag, (uint32_t)seed, (uint32_t)(seed >> 32)));
        else if (total > 0)
            TSU_HIP_TRY(ctx, k2wg_launch(d, ctx->stream, replay_uniforms ? d->uniforms : nullptr, d->samples, T, temps_dev, n_burnin, n_sweeps,
                                         n_samples, sweep0, tag, (uint32_t)seed, (uint32_t)(seed >> 32), nullptr));
    } else {
        // larger systems in natural order: the whole run in ONE launch of the pipeline kernel (it records the states itself and takes
        // a temperature per sweep) -- a loop of calls costs ~60 us per call beside sweeps of 40-100 us at n = 1000-4000
        static int one_launch = -1;
        if (one_launch < 0) {
            const char* e = getenv("TSU_K2_RUN_ONE_LAUNCH");
            one_launch = e ? atoi(e) : 1;
        }
        const size_t ucnt = (size_t)total * n;
        if (one_launch && total > 0 && (order || !d->co_disabled) && n >= 2 * DB && (!replay_uniforms || ucnt * 8 <= ((size_t)1 << 29)) &&
            (!order || ucnt * 8 <= ((size_t)1 << 29))) {
            const double* temps_dev = nullptr;
            if (temps) {  // (a schedule: n_burnin == 0, n_sweeps == 1, one temperature per recorded state)
                if (d->temps_cap < (size_t)total) {
                    if (d->temps) (void)hipFree(d->temps);
                    d->temps = nullptr;
                    d->temps_cap = 0;
                    TSU_HIP_TRY(ctx, hipMalloc(&d->temps, (size_t)total * 8));
                    d->temps_cap = (size_t)total;
                }
                TSU_HIP_TRY(ctx, hipMemcpyAsync(d->temps, temps, (size_t)total * 8, hipMemcpyHostToDevice, ctx->stream));
                temps_dev = d->temps;
            }
            if (replay_uniforms) {
                if (d->uni_cap < ucnt) {
                    if (d->uniforms) (void)hipFree(d->uniforms);
                    d->uniforms = nullptr;
                    d->uni_cap = 0;
                    TSU_HIP_TRY(ctx, hipMalloc(&d->uniforms, ucnt * 8));
                    d->uni_cap = ucnt;
                }
                TSU_HIP_TRY(ctx, hipMemcpyAsync(d->uniforms, replay_uniforms, ucnt * 8, hipMemcpyHostToDevice, ctx->stream));
            }
            if (order) {  // (validated as permutations by tsu_dense_sweep's check, repeated here for the one-launch path)
                std::vector<char> seen((size_t)n);
                for (long long sw = 0; sw < total; ++sw) {
                    std::fill(seen.begin(), seen.end(), 0);
                    for (int i = 0; i < n; ++i) {
                        const long long v = order[(size_t)sw * n + i];
                        TSU_REQUIRE(ctx, v >= 0 && v < n && !seen[(size_t)v], "dense_sweep: order row %d is not a permutation of 0..%d", (int)sw, n - 1);
                        seen[(size_t)v] = 1;
                    }
                }
                if (d->order_cap < ucnt) {
                    if (d->order) (void)hipFree(d->order);
                    d->order = nullptr;
                    d->order_cap = 0;
                    TSU_HIP_TRY(ctx, hipMalloc(&d->order, ucnt * 8));
                    d->order_cap = ucnt;
                }
                TSU_HIP_TRY(ctx, hipMemcpyAsync(d->order, order, ucnt * 8, hipMemcpyHostToDevice, ctx->stream));
            }
            TSU_HIP_TRY(ctx, hipMemcpyAsync(d->backup, d->state, (size_t)n, hipMemcpyDeviceToDevice, ctx->stream));
            int done = 0;
            const int rc1 = tsu_dense_pipe_run(d, T, temps_dev, (int)total, n_burnin, n_sweeps, d->samples, seed, sweep0, replica, replay_uniforms != nullptr,
                                               &done, order ? d->order : nullptr);
            if (rc1 != TSU_OK) return rc1;
            if (done) {
                if (out_bytes) TSU_HIP_TRY(ctx, hipMemcpyAsync(samples_host, d->samples, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
                TSU_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
                return TSU_OK;
            }
            TSU_HIP_TRY(ctx, hipMemcpyAsync(d->state, d->backup, (size_t)n, hipMemcpyDeviceToDevice, ctx->stream));  // declined or gave up: the loop below
        }
        // otherwise the sweep paths above, one call per recorded state; samples gathered on the device
        // (with a schedule every recorded state is one sweep: n_burnin == 0 and n_sweeps == 1)
        int rc = tsu_dense_sweep(d, T, n_burnin, order, seed, sweep0, replica, replay_uniforms);
        if (rc != TSU_OK) return rc;
        size_t off = (size_t)n_burnin * n;
        uint32_t sw = sweep0 + (uint32_t)n_burnin;
        for (int k = 0; k < n_samples; ++k) {
            rc = tsu_dense_sweep(d, temps ? temps[k] : T, n_sweeps, order ? order + off : nullptr, seed, sw, replica,
                                 replay_uniforms ? replay_uniforms + off : nullptr);
            if (rc != TSU_OK) return rc;
            TSU_HIP_TRY(ctx, hipMemcpyAsync(d->samples + (size_t)k * n, d->state, (size_t)n, hipMemcpyDeviceToDevice, ctx->stream));
            off += (size_t)n_sweeps * n;
            sw += (uint32_t)n_sweeps;
        }
    }
    if (out_bytes) TSU_HIP_TRY(ctx, hipMemcpyAsync(samples_host, d->samples, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    TSU_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return TSU_OK;
}

int tsu_dense_sample(tsu_dense* d, double T, int n_burnin, int n_sweeps, int n_samples, const int64_t* order, uint64_t seed,
                     uint32_t sweep0, uint32_t replica, const double* replay_uniforms, int8_t* samples_host) {
    TSU_ENTER(d ? d->ctx : nullptr);
    if (!d) return TSU_E_INVALID;
    return dense_run(d, T, nullptr, n_burnin, n_sweeps, n_samples, order, seed, sweep0, replica, replay_uniforms, samples_host);
}

int tsu_dense_anneal(tsu_dense* d, const double* temperatures, int n_steps, const int64_t* order, uint64_t seed, uint32_t sweep0,
                     uint32_t replica, const double* replay_uniforms, int8_t* states_host) {
    TSU_ENTER(d ? d->ctx : nullptr);
    if (!d) return TSU_E_INVALID;
    TSU_REQUIRE(d->ctx, n_steps >= 0 && (n_steps == 0 || temperatures), "dense_anneal: need n_steps >= 0 and a temperature per step");
    for (int s = 0; s < n_steps; ++s) TSU_REQUIRE(d->ctx, temperatures[s] > 0.0, "Temperature must be positive");
    return dense_run(d, 1.0, temperatures, 0, 1, n_steps, order, seed, sweep0, replica, replay_uniforms, states_host);
}

int tsu_dense_sweep_replicas(tsu_dense* d, int n_replicas, const double* temperatures, int n_sweeps, int8_t* states_host,
                             const uint64_t* seeds, const uint32_t* sweep0s, const uint32_t* replicas, const double* replay_uniforms) {
    TSU_ENTER(d ? d->ctx : nullptr);
    if (!d) return TSU_E_INVALID;
    tsu_ctx* ctx = d->ctx;
    TSU_REQUIRE(ctx, n_replicas >= 1 && temperatures && states_host && seeds && sweep0s && replicas,
                "dense_sweep_replicas: temperatures, states, seeds, sweep0s and replicas are per-replica arrays");
    TSU_REQUIRE(ctx, n_sweeps >= 0, "dense_sweep: n_sweeps must be >= 0");
    const int n = d->n;
    for (int r = 0; r < n_replicas; ++r) {
        TSU_REQUIRE(ctx, temperatures[r] > 0.0, "Temperature must be positive");
        TSU_REQUIRE(ctx, dense_bits01(states_host + (size_t)r * n, (size_t)n), "dense_set_state: state must be 0/1");
    }
    if (n_sweeps == 0) return TSU_OK;
    d->rep_match = 0;  // (the kept rows are about to change)
    const int wave_m = k2w_slots(d);
    const bool wg = !wave_m && k2wg_takes(d);
    if (!wave_m && !wg) {
        // larger systems: up to eight replicas advance together in one launch of the owner-computes kernel (dense_own.hip: every row
        // of J^T is loaded once for all of them), groups of eight one after the other; if the kernel declines, one replica after the
        // other through the sweep paths
        bool all_done = true;
        int r_done = 0;  // replicas whose final states are already in states_host
        for (int r0 = 0; r0 < n_replicas && all_done; r0 += 8) {
            const int m = n_replicas - r0 < 8 ? n_replicas - r0 : 8;
            if (m == 1) {
                all_done = false;
                break;
            }
            const int mp = m <= 2 ? 2 : m <= 4 ? 4 : 8;  // (padding replicas repeat the group's first one; their results are dropped)
            const size_t sb_ = (size_t)mp * n, ub_ = replay_uniforms ? (size_t)mp * n_sweeps * n * 8 : 0;
            const size_t need_ = ((sb_ + 7) / 8) * 8 + ub_;
            if (d->rep_cap < need_) {
                if (d->rep_buf) (void)hipFree(d->rep_buf);
                d->rep_buf = nullptr;
                d->rep_cap = 0;
                TSU_HIP_TRY(ctx, hipMalloc(&d->rep_buf, need_));
                d->rep_cap = need_;
            }
            int8_t* ds = (int8_t*)d->rep_buf;
            double* du = ub_ ? (double*)((char*)d->rep_buf + ((sb_ + 7) / 8) * 8) : nullptr;
            if (!d->h_rep && hipHostMalloc((void**)&d->h_rep, (size_t)8 * n, hipHostMallocDefault) != hipSuccess) {
                d->h_rep = nullptr;
                (void)hipGetLastError();
            }
            // one group of replicas: their fields stay on the device from call to call; a state that comes back byte for byte as one
            // of those the last call returned (in any position: tempering swaps them) resumes from that state's fields
            const bool keep = n_replicas <= 8;
            if (keep && !d->rep_prev) {
                d->rep_prev = (int8_t*)malloc((size_t)8 * n);
                if (d->rep_prev && hipMalloc(&d->rep_fields[0], (size_t)8 * n * sizeof(double)) == hipSuccess &&
                    hipMalloc(&d->rep_fields[1], (size_t)8 * n * sizeof(double)) == hipSuccess) {
                    d->rep_prev_n = 0;
                } else {
                    if (d->rep_fields[0]) (void)hipFree(d->rep_fields[0]);
                    if (d->rep_fields[1]) (void)hipFree(d->rep_fields[1]);
                    d->rep_fields[0] = d->rep_fields[1] = nullptr;
                    free(d->rep_prev);
                    d->rep_prev = nullptr;
                    (void)hipGetLastError();
                }
            }
            const bool can_keep = keep && d->rep_prev && d->rep_fields[1];
            bool all_known = can_keep && d->rep_prev_n > 0;
            OwnRep reps[8];
            for (int q = 0; q < mp; ++q) {
                const int r = r0 + (q < m ? q : 0);
                reps[q].src = -1;
                if (can_keep)
                    for (int t = 0; t < d->rep_prev_n && reps[q].src < 0; ++t) {
                        const int p = (q + t) % d->rep_prev_n;  // (its own position first)
                        if (memcmp(states_host + (size_t)r * n, d->rep_prev + (size_t)p * n, (size_t)n) == 0) reps[q].src = p;
                    }
                all_known = all_known && reps[q].src >= 0;
                reps[q].T = temperatures[r];
                reps[q].sweep0 = sweep0s[r];
                reps[q].tag = TSU_TAG_DENSE | (replicas[r] << 8);
                reps[q].k0 = (uint32_t)seeds[r];
                reps[q].k1 = (uint32_t)(seeds[r] >> 32);
                if (d->h_rep) memcpy(d->h_rep + (size_t)q * n, states_host + (size_t)r * n, (size_t)n);
                else TSU_HIP_TRY(ctx, hipMemcpyAsync(ds + (size_t)q * n, states_host + (size_t)r * n, (size_t)n, hipMemcpyHostToDevice, ctx->stream));
                if (du)
                    TSU_HIP_TRY(ctx, hipMemcpyAsync(du + (size_t)q * n_sweeps * n, replay_uniforms + (size_t)r * n_sweeps * n, (size_t)n_sweeps * n * 8,
                                                    hipMemcpyHostToDevice, ctx->stream));
            }
            if (d->h_rep) TSU_HIP_TRY(ctx, hipMemcpyAsync(ds, d->h_rep, (size_t)mp * n, hipMemcpyHostToDevice, ctx->stream));
            int done = 0;
            const int rc = tsu_dense_own_run(d, mp, reps, ds, n_sweeps, du, nullptr, nullptr, nullptr, 0, 1, all_known, can_keep, &done);
            if (rc != TSU_OK) {
                d->rep_prev_n = 0;
                return rc;
            }
            if (!done) {
                d->own_failed = 0;  // (the replicas' states live in a scratch buffer: nothing of the system was touched)
                d->rep_prev_n = 0;
                all_done = false;
                break;
            }
            TSU_HIP_TRY(ctx, hipMemcpyAsync(d->h_rep ? d->h_rep : states_host + (size_t)r0 * n, ds, (size_t)m * n, hipMemcpyDeviceToHost, ctx->stream));
            TSU_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            if (d->h_rep) memcpy(states_host + (size_t)r0 * n, d->h_rep, (size_t)m * n);
            if (can_keep) {
                memcpy(d->rep_prev, states_host + (size_t)r0 * n, (size_t)m * n);
                d->rep_prev_n = m;
            }
            r_done = r0 + m;
        }
        if (all_done) return TSU_OK;
        for (int r = r_done; r < n_replicas; ++r) {  // (the groups that ran are finished: only the others)
            int rc = tsu_dense_set_state(d, states_host + (size_t)r * n);
            if (rc == TSU_OK)
                rc = tsu_dense_sweep(d, temperatures[r], n_sweeps, nullptr, seeds[r], sweep0s[r], replicas[r],
                                     replay_uniforms ? replay_uniforms + (size_t)r * n_sweeps * n : nullptr);
            if (rc == TSU_OK) rc = tsu_dense_get_state(d, states_host + (size_t)r * n);
            if (rc != TSU_OK) return rc;
        }
        return TSU_OK;
    }
    const size_t sbytes = (size_t)n_replicas * n, ubytes = replay_uniforms ? (size_t)n_replicas * n_sweeps * n * 8 : 0;
    const size_t rbytes = (size_t)n_replicas * sizeof(K2Replica);
    const size_t need = ((sbytes + 7) / 8) * 8 + ubytes + rbytes;
    if (d->rep_cap < need) {
        if (d->rep_buf) (void)hipFree(d->rep_buf);
        d->rep_buf = nullptr;
        d->rep_cap = 0;
        TSU_HIP_TRY(ctx, hipMalloc(&d->rep_buf, need));
        d->rep_cap = need;
    }
    int8_t* d_states = (int8_t*)d->rep_buf;
    double* d_uni = ubytes ? (double*)((char*)d->rep_buf + ((sbytes + 7) / 8) * 8) : nullptr;
    K2Replica* d_reps = (K2Replica*)((char*)d->rep_buf + ((sbytes + 7) / 8) * 8 + ubytes);
    std::vector<K2Replica> reps((size_t)n_replicas);
    for (int r = 0; r < n_replicas; ++r) {
        reps[(size_t)r].T = temperatures[r];
        reps[(size_t)r].sweep0 = sweep0s[r];
        reps[(size_t)r].tag = TSU_TAG_DENSE | (replicas[r] << 8);
        reps[(size_t)r].k0 = (uint32_t)seeds[r];
        reps[(size_t)r].k1 = (uint32_t)(seeds[r] >> 32);
    }
    TSU_HIP_TRY(ctx, hipMemcpyAsync(d_states, states_host, sbytes, hipMemcpyHostToDevice, ctx->stream));
    if (ubytes) TSU_HIP_TRY(ctx, hipMemcpyAsync(d_uni, replay_uniforms, ubytes, hipMemcpyHostToDevice, ctx->stream));
    TSU_HIP_TRY(ctx, hipMemcpyAsync(d_reps, reps.data(), rbytes, hipMemcpyHostToDevice, ctx->stream));
    if (wave_m)
        TSU_HIP_TRY(ctx, k2w_launch_replicas(d, wave_m, ctx->stream, n_replicas, d_states, d_uni, d_reps, n_sweeps));
    else  // one workgroup per replica
        TSU_HIP_TRY(ctx, k2wg_launch(d, ctx->stream, d_uni, nullptr, 1.0, nullptr, n_sweeps, 1, 0, 0, 0, 0, 0, nullptr, n_replicas, d_states, d_reps));
    TSU_HIP_TRY(ctx, hipMemcpyAsync(states_host, d_states, sbytes, hipMemcpyDeviceToHost, ctx->stream));
    TSU_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return TSU_OK;
}

int tsu_dense_energy(tsu_dense* d, double* energy) {
    TSU_ENTER(d ? d->ctx : nullptr);
    if (!d) return TSU_E_INVALID;
    tsu_ctx* ctx = d->ctx;
    TSU_REQUIRE(ctx, energy != nullptr, "dense_energy: NULL");
    // the fields f = J s + b: those the pipeline kept for exactly this state (a loop of sweep + energy calls then streams J once per
    // step instead of three times), or one pass over J
    const double* fields = d->field;
    if (d->fields_valid && d->co_fields) {
        fields = d->co_fields;
    } else if (d->rep_match > 0 && d->rep_match <= d->rep_prev_n && d->rep_fields[d->rep_cur]) {
        fields = d->rep_fields[d->rep_cur] + (size_t)(d->rep_match - 1) * d->n;
    } else {
        unsigned mv_grid = (unsigned)(((size_t)d->n * 64 + 255) / 256);
        if (d->dtype == TSU_DTYPE_F64)
            k2_matvec<double><<<mv_grid, 256, 0, ctx->stream>>>((const double*)d->J, d->state, d->bias, d->field, d->n);
        else
            k2_matvec<float><<<mv_grid, 256, 0, ctx->stream>>>((const float*)d->J, d->state, d->bias, d->field, d->n);
    }
    k2_energy<<<1, 1024, 0, ctx->stream>>>(fields, d->state, d->bias, d->d_energy, d->n);
    TSU_HIP_TRY(ctx, hipGetLastError());
    int8_t* stage = dense_stage(d);
    void* dst = stage ? (void*)(stage + (((size_t)d->n + 7) / 8) * 8) : (void*)energy;
    TSU_HIP_TRY(ctx, hipMemcpyAsync(dst, d->d_energy, 8, hipMemcpyDeviceToHost, ctx->stream));
    TSU_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (stage) memcpy(energy, dst, 8);
    return TSU_OK;
}

int tsu_dense_launch_counts(tsu_dense* d, uint64_t counts[2]) {
    if (!d || !counts) return TSU_E_INVALID;
    counts[0] = d->n_own;
    counts[1] = d->n_pipe;
    return TSU_OK;
}

int tsu_dense_energies(tsu_dense* d, const int8_t* states_host, int n_states, double* energies_host) {
    TSU_ENTER(d ? d->ctx : nullptr);
    if (!d) return TSU_E_INVALID;
    tsu_ctx* ctx = d->ctx;
    TSU_REQUIRE(ctx, n_states >= 0 && (n_states == 0 || (states_host && energies_host)), "dense_energies: NULL / negative count");
    if (n_states == 0) return TSU_OK;
    const size_t n = (size_t)d->n;
    TSU_REQUIRE(ctx, dense_bits01(states_host, n * (size_t)n_states), "dense_energies: states must be 0/1");
    // the states share the sample buffer (it is only ever a staging area of one call), the energies get a scratch array
    if (d->samples_cap < n * (size_t)n_states) {
        if (d->samples) (void)hipFree(d->samples);
        d->samples = nullptr;
        d->samples_cap = 0;
        TSU_HIP_TRY(ctx, hipMalloc(&d->samples, n * (size_t)n_states));
        d->samples_cap = n * (size_t)n_states;
    }
    double* d_e = nullptr;
    TSU_HIP_TRY(ctx, hipMalloc(&d_e, (size_t)n_states * 8));
    hipError_t e = hipMemcpyAsync(d->samples, states_host, n * (size_t)n_states, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_e, 0, (size_t)n_states * 8, ctx->stream);
    const unsigned mv_grid = (unsigned)((n * 64 + 255) / 256);
    for (int k = 0; k < n_states && e == hipSuccess; ++k) {  // one pass over J per state, all on the stream, one wait at the end
        const int8_t* s = d->samples + (size_t)k * n;
        // ... unless the state is one the last replica call returned: its fields are still on the device (a tempering ladder asks
        // for its chains' energies between their sweeps)
        const double* kept = nullptr;
        for (int p = 0; p < d->rep_prev_n && !kept && d->rep_fields[d->rep_cur]; ++p)
            if (memcmp(states_host + (size_t)k * n, d->rep_prev + (size_t)p * n, n) == 0) kept = d->rep_fields[d->rep_cur] + (size_t)p * n;
        if (kept) {
            k2_energy<<<1, 1024, 0, ctx->stream>>>(kept, s, d->bias, d_e + k, d->n);
            continue;
        }
        if (d->dtype == TSU_DTYPE_F64)
            k2_matvec<double><<<mv_grid, 256, 0, ctx->stream>>>((const double*)d->J, s, d->bias, d->field, d->n);
        else
            k2_matvec<float><<<mv_grid, 256, 0, ctx->stream>>>((const float*)d->J, s, d->bias, d->field, d->n);
        k2_energy<<<1, 1024, 0, ctx->stream>>>(d->field, s, d->bias, d_e + k, d->n);
    }
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(energies_host, d_e, (size_t)n_states * 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_e);
    if (e != hipSuccess) return tsu_fail(ctx, TSU_E_HIP, "dense_energies: %s", hipGetErrorString(e));
    return TSU_OK;
}

}  // extern "C"
