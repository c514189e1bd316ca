// dense_own.hip -- K2 "owner computes" (k2_own): sequential Gibbs sweeps on a dense J in ONE launch of a co-resident grid, every
// workgroup the owner of 64 M consecutive sites for the whole call.
//
// Same mathematics as the other dense paths (dense_coop.hip): the sequential pass over a superblock of positions is the unique
// fixed point of v = decide(F0 + L (v - v0)) (L = strictly lower triangle of the superblock's couplings in visiting order, v0 the
// values at its start), reached exactly by Jacobi iteration from v = v0.  What differs is who holds what:
//   * The FIELD of a site lives in its owner's registers (lane = site) from the first sweep to the last.  Nothing about fields ever
//     crosses workgroups, so the only inter-workgroup traffic is the VALUE MASKS of the sites that are being decided: one 8-byte
//     {32 values, tag} granule pair per 64 sites and generation, written with one sc1 store each and polled by whoever needs them
//     (each granule validates itself by its tag: no counters, no barriers, no atomics).
//   * J is read in AXPY form, one row of J^T per site whose value CHANGED: F[r] += +-J^T[j, r] for the owner's rows r (lanes are
//     consecutive r: coalesced).  A superblock in which 41 % of the sites flip therefore reads 41 % of its strip of J, and a
//     generation in which thirty sites toggle reads thirty row segments -- not the dot products over every column of the
//     superblock the row-owning pipeline (k2_pipe) computes.  Summation orders are fixed (list order, then a fixed tree over the
//     waves), so results are reproducible bit for bit.
//   * Natural order: the owners of a superblock's sites ("active" workgroups) run the generations among themselves; everybody else
//     waits for the superblock's FINAL masks (a ring of tagged granules) and applies the final flips to its rows in one pass.
//   * A caller's visiting order (update_order="random": one permutation per sweep): a superblock is a range of POSITIONS, its sites
//     are spread over all workgroups, so every workgroup is active in every superblock, follows every generation and applies every
//     toggle to all its rows (unmasked: the running field) and to the rows that come later in the order (masked: the decision).
//   * Replicas (tempering ladders, independent chains on ONE J): R states advance together, one deciding wave per replica: wave rho
//     polls replica rho's granules, scans its toggles and gathers short lists by itself (rep_step); long lists go to all sixteen
//     waves as a union list -- a row of J^T loaded once for the union of the replicas' toggles and applied to each replica's
//     accumulator with that replica's sign -- built by all sixteen waves (build_par).  The replicas' fields stay on the device
//     from call to call (dense.h: rep_fields).
//   * Solo generations (one chain): after the first correction pass a generation moves a handful of sites; the deciding wave then
//     runs it alone -- decide, publish, poll, scan, gather, reduce -- with no workgroup barrier; the other waves are called in for
//     long lists only.
#include <algorithm>
#include <vector>

#include "dense_dev.h"

#define OWN_THREADS 1024
#define OWN_WAVES (OWN_THREADS / 64)
#define OWN_RING 8              // superblocks whose final masks are kept (a workgroup is active at least once per sweep, and cannot be
                                // active before it has applied every earlier superblock: nobody lags more than a sweep = at most
                                // OWN_RING superblocks, checked by the host)
#define OWN_TAG_SPAN 16384u     // generation g of superblock number q carries tag q * SPAN + g + 1 (0 = never written)
#define OWN_MAX_R 8
#define OWN_SOLO_MAX 32           // toggle-list entries (before the workgroup's last row) a deciding wave gathers by itself (natural order; a caller's: 48)
#ifndef OWN_REP_QUAD
#define OWN_REP_QUAD 0          // replicas: 0 = one row per lane, scalar multipliers (own_axpy_rep); 1 = the quad layout in groups of four replicas
#endif
#define OWN_NGEN 16            // generation buffers (a ring by generation number): the deciders of a superblock move in lockstep and need two;
                                // the FOLLOWERS (below) read the same buffers without being waited for and may lag up to OWN_NGEN - 2 generations
                                // (they notice an overwritten buffer by its tag and fail the call -- the caller's other paths redo it)

struct OwnParams {
    const void* JT;
    const double* bias;
    int8_t* state;               // [R][n] values, committed in place superblock by superblock
    const double* uniforms;      // [R][n_sweeps][n] replayed uniforms (indexed by position) or nullptr
    unsigned long long* gen;     // [OWN_NGEN][R][G][2] granules of the running generations (ring by generation number)
    unsigned long long* fin;     // [OWN_RING][R][G][2] granules of the final values of a superblock (tag = superblock number + 1)
    unsigned* bar;               // error words (BAR_ERR page layout of dense_dev.h)
    const int64_t* order;        // [n_sweeps][n] visiting orders, or nullptr (natural order)
    const double* temps;         // one temperature per sweep (R == 1), or nullptr
    int8_t* samples;             // (R == 1) the state after sweep rec_from + m rec_every goes to samples + (m - 1) n, or nullptr
    double* fields_all;          // [R][n] fields kept from call to call (dense.h): read at the start of a resumed call (row rep[rho].src) ...
    double* fields_out;          // ... written at the end of a call that keeps them (several replicas: another buffer -- the rows change places)
    unsigned long long* timeline;
    int n, n_sweeps;
    int sbw;                     // workgroups per superblock (natural order); positions per superblock = sbw * 64 M
    int lmax;                    // toggle-list capacity (entries) = positions per superblock
    int rec_from, rec_every, resume, persist, refresh_off;
    int fail_at;                 // (tests) give up at this superblock number as if a wait had expired; < 0: never
    int solo;                    // one replica, 64 rows per workgroup: short generations by the deciding wave alone
    int solo_max;                // ... lists of up to this many entries
    OwnRep rep[OWN_MAX_R];
};

// the decision helpers of dense_dev.h with the uniform's index apart from the site (a replayed uniform belongs to the POSITION in
// the visiting order, gibbs.py:126; the Philox uniform to the site)
static __device__ __noinline__ int own_decide_exact(double F, double T, uint32_t site, uint32_t uidx, const double* __restrict__ uniforms,
                                                    uint32_t sweep, uint32_t tag, uint32_t k0, uint32_t k1) {
    const double u = uniforms ? uniforms[uidx] : dense_uniform(site, sweep, tag, k0, k1);
    return (u < sigmoid_clamped(F / T)) ? 1 : 0;
}
static __device__ __forceinline__ int own_decide(double F, double lg, double T, double invT, uint32_t site, uint32_t uidx,
                                                 const double* __restrict__ uniforms, uint32_t sweep, uint32_t tag, uint32_t k0, uint32_t k1) {
    const double xa = F * invT;
    if (fabs(fabs(xa) - 20.0) < 1e-9 || fabs(xa - lg) <= 1e-9 * (1.0 + fabs(lg)))
        return own_decide_exact(F, T, site, uidx, uniforms, sweep, tag, k0, k1);
    if (xa > 20.0) return 1;
    if (xa < -20.0) return 0;
    return xa > lg ? 1 : 0;
}
static __device__ __noinline__ double own_logit(uint32_t site, uint32_t uidx, const double* __restrict__ uniforms, uint32_t sweep, uint32_t tag,
                                                uint32_t k0, uint32_t k1) {
    const double u = uniforms ? uniforms[uidx] : dense_uniform(site, sweep, tag, k0, k1);
    return log(u) - log1p(-u);
}

// ---------------------------------------------------------------------------------------------------------------- the axpy pass
// F[r] += sum over the list entries (site j, sign) of +-J^T[j, r] for the workgroup's 64 M rows.  A lane owns a QUAD of four
// consecutive rows and one of the 4 / M entry slots of its wave: a wave-instruction loads 16 bytes per lane = the row segments of
// 4 / M consecutive entries (a "bundle"; with one dword per lane the pass was bound by instruction issue, ~15 instructions per
// entry and wave, not by memory).  Every wave takes the bundles b = wave, wave + 16, ..., U bundles in flight per lane, loads
// unconditional (entries beyond the list read the segment of site 0 with weight 0: a predicated load becomes a branch and the loads
// serial round trips).  At the end the entry slots are added up across the lanes (fixed order), the sixteen waves' partial sums go
// to LDS, and the rows' owners add them up in wave order.
//   MODE 0: every entry to every row (strips of final flips, the field pass): slots = replicas
//   MODE 1: natural order, generations: entry j only to the rows r > j (the decision's correction): slots = replicas
//   MODE 2: a caller's order (R = 1): slot 0 = rows later in the order than j (the correction), slot 1 = every row (the running field)
// NA = accumulator slots of the KERNEL (the layout of `red`).
template <typename TJ>
struct OwnQuad;
template <>
struct OwnQuad<float> {
    float4 v;
    __device__ __forceinline__ void load(const float* p) { v = *reinterpret_cast<const float4*>(p); }
    __device__ __forceinline__ double get(int m) const { return (double)(m == 0 ? v.x : m == 1 ? v.y : m == 2 ? v.z : v.w); }
};
template <>
struct OwnQuad<double> {
    double2 a, b;
    __device__ __forceinline__ void load(const double* p) {
        a = *reinterpret_cast<const double2*>(p);
        b = *reinterpret_cast<const double2*>(p + 2);
    }
    __device__ __forceinline__ double get(int m) const { return m == 0 ? a.x : m == 1 ? a.y : m == 2 ? b.x : b.y; }
};

// entries [k0, k1) of the list into acc (MASKED: per-lane masks as MODE says; else every entry to every row).  Bundles of ES consecutive
// entries, wave wv takes bundles wv, wv + 16, ..., U of them in flight per lane.  (Tried and dropped: two groups of four bundles in
// flight alternately with unconditional loads -- the dummy loads of short lists and the longer prologue cost more than the overlap
// gained: 0.29 -> 0.36 ms per sweep at n = 16384.)
template <typename TJ, int M, int R0, int RN, int MODE, bool MASKED>
static __device__ __forceinline__ void own_axpy_range(const TJ* __restrict__ JT, int n, const uint32_t* lst, int k0, int k1, int col0, int myrow,
                                                      const unsigned short* pos, const int* mypos, double (*acc)[4], int lane, int wv) {
    constexpr int QL = 16 * M, ES = 4 / M;
    constexpr int U = 4;  // bundles in flight per lane (tools/microbench_axpy: 64 workgroups x 1700 entries take 7.0 us with 4, 10.4 with 8, 8.3 with 2)
    const int t = lane / QL;                 // my entry slot
    const int nb = (k1 - k0 + ES - 1) / ES;  // bundles
    if (wv >= nb) return;
    auto consume = [&](const OwnQuad<TJ>& xq, uint32_t ev) {
        const int j = (int)(ev & 0xFFFFu);
        double v[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) v[m] = xq.get(m);
        if (MODE == 2) {
            const double sg = (double)(((int)(ev << 14)) >> 30);
            const int pj = pos[j];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                acc[1][m] += sg * v[m];
                acc[0][m] += pj < mypos[m] ? sg * v[m] : 0.0;
            }
        } else {
            if (MASKED) {
#pragma unroll
                for (int m = 0; m < 4; ++m) v[m] = j < myrow + m ? v[m] : 0.0;
            }
#pragma unroll
            for (int rho = 0; rho < RN; ++rho) {
                // (the 2-bit code, sign-extended, IS the multiplier: 01b = +1, 11b = -1, 00b = 0)
                const double sg = (double)(((int)(ev << (14 - 2 * (R0 + rho)))) >> 30);
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[rho][m] += sg * v[m];
            }
        }
    };
    for (int b = wv; b < nb; b += OWN_WAVES * U) {
        OwnQuad<TJ> x[U];
        uint32_t e[U];
        // (the bundle tests are wave-uniform -- wv is scalar: a wave issues loads for its own bundles only, short lists cost one round trip)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = k0 + (b + OWN_WAVES * u) * ES + t;
            e[u] = k < k1 ? lst[k] : 0u;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (b + OWN_WAVES * u < nb) x[u].load(JT + (size_t)(e[u] & 0xFFFFu) * n + col0);
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (b + OWN_WAVES * u < nb) consume(x[u], e[u]);
    }
}

// nlo: (MODE 1) entries [0, nlo) belong to sites before the workgroup's first row (every row takes them: no masks), [nlo, nl) to
// the workgroup's own sites (per-lane masks)
template <typename TJ, int M, int R, int MODE, int NA>
static __device__ __forceinline__ void own_axpy(const TJ* __restrict__ JT, int n, const uint32_t* lst, int nlo, int nl, int col0, int myrow,
                                                const unsigned short* pos, const int* mypos, double* red, int lane, int wv) {
    constexpr int QL = 16 * M, ES = 4 / M;
    // replicas in groups of four per pass over the list (eight replicas x four rows of f64 accumulators do not fit the registers
    // beside the loads in flight; the second pass finds the segments in the caches)
    constexpr int RG = MODE == 2 ? 1 : (R > 4 ? 4 : R);
    constexpr int NS = MODE == 2 ? 2 : RG;
#pragma unroll
    for (int r0 = 0; r0 < (MODE == 2 ? 1 : R); r0 += RG) {
        double acc[NS][4];
#pragma unroll
        for (int a = 0; a < NS; ++a)
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[a][m] = 0.0;
        if (MODE == 1) {
            // (a short list in ONE masked pass: the unmasked and the masked range would be two dependent round trips to memory)
            if (nl <= OWN_WAVES * 4 * (4 / M)) nlo = 0;
            if (r0 == 0) {
                own_axpy_range<TJ, M, 0, RG, 1, false>(JT, n, lst, 0, nlo, col0, myrow, pos, mypos, acc, lane, wv);
                own_axpy_range<TJ, M, 0, RG, 1, true>(JT, n, lst, nlo, nl, col0, myrow, pos, mypos, acc, lane, wv);
            } else {
                own_axpy_range<TJ, M, (R > 4 ? 4 : 0), RG, 1, false>(JT, n, lst, 0, nlo, col0, myrow, pos, mypos, acc, lane, wv);
                own_axpy_range<TJ, M, (R > 4 ? 4 : 0), RG, 1, true>(JT, n, lst, nlo, nl, col0, myrow, pos, mypos, acc, lane, wv);
            }
        } else if (MODE == 2) {
            own_axpy_range<TJ, M, 0, 1, 2, true>(JT, n, lst, 0, nl, col0, myrow, pos, mypos, acc, lane, wv);
        } else {
            if (r0 == 0) own_axpy_range<TJ, M, 0, RG, 0, false>(JT, n, lst, 0, nl, col0, myrow, pos, mypos, acc, lane, wv);
            else own_axpy_range<TJ, M, (R > 4 ? 4 : 0), RG, 0, false>(JT, n, lst, 0, nl, col0, myrow, pos, mypos, acc, lane, wv);
        }
        // the entry slots of a quad: lanes q, q + QL, ... (fixed order: ((0 + 1) + (2 + 3)))
#pragma unroll
        for (int a = 0; a < NS; ++a)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                double z = acc[a][m];
                if (ES >= 2) z += __shfl_xor(z, QL, 64);
                if (ES >= 4) z += __shfl_xor(z, 2 * QL, 64);
                acc[a][m] = z;
            }
        if (lane < QL) {
#pragma unroll
            for (int a = 0; a < NS; ++a)
#pragma unroll
                for (int m = 0; m < 4; ++m) red[(wv * NA + r0 + a) * (64 * M) + 4 * lane + m] = acc[a][m];
        }
    }
}

// The axpy pass for SEVERAL replicas (M = 1): one list entry per wave-instruction, lane = row (one dword per lane).  The entry and its
// per-replica codes are wave-uniform (scalar): the sign of a replica is a scalar multiplier (one multiply-add per replica and
// entry, nothing per lane to decode), and a lane holds one f64 accumulator per replica instead of four.  (The quad layout, which loads four entries per instruction, pays 8 multiply-adds per element
// whatever the codes are and needed two passes over the list for its 32 accumulators: 1.38 ms per all-replica sweep at n = 16384.)
template <typename TJ, int R, bool MASKED, int NA>
static __device__ __forceinline__ void own_axpy_rep(const TJ* __restrict__ JT, int n, const uint32_t* lst, int nl, int col, int myrow, double* red,
                                                    int lane, int wv) {
    constexpr int U = 16;  // entries (256-byte segments) in flight per wave
    double acc[R];
#pragma unroll
    for (int rho = 0; rho < R; ++rho) acc[rho] = 0.0;
    for (int k = wv; k < nl; k += OWN_WAVES * U) {
        TJ x[U];
        uint32_t e[U];
#pragma unroll
        for (int u = 0; u < U; ++u) e[u] = k + OWN_WAVES * u < nl ? lst[k + OWN_WAVES * u] : 0u;
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (k + OWN_WAVES * u < nl) {
                const uint32_t j = (uint32_t)__builtin_amdgcn_readfirstlane((int)e[u]) & 0xFFFFu;
                x[u] = JT[(size_t)j * n + col];
            }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (k + OWN_WAVES * u < nl) {
                const uint32_t ev = (uint32_t)__builtin_amdgcn_readfirstlane((int)e[u]);
                const int j = (int)(ev & 0xFFFFu);
                double v = (double)x[u];
                if (MASKED) v = j < myrow ? v : 0.0;
#pragma unroll
                for (int rho = 0; rho < R; ++rho) {
                    // (a scalar multiplier: written as a branch the compiler turns the skip into five selects per replica)
                    const double sg = (double)(((int)(ev << (14 - 2 * rho))) >> 30);  // the 2-bit code, sign-extended: +1, -1 or 0
                    acc[rho] = fma(sg, v, acc[rho]);
                }
            }
    }
#pragma unroll
    for (int rho = 0; rho < R; ++rho) red[(wv * NA + rho) * 64 + lane] = acc[rho];
}

// ------------------------------------------------------------------------------------------------------------------ the kernel
template <typename TJ, int M, int R, bool ORD>
__global__ __launch_bounds__(OWN_THREADS) void k2_own(OwnParams P) {
    static_assert(!ORD || R == 1, "replicas advance in natural order only");
    constexpr int RW = 64 * M;           // sites per workgroup
    constexpr int NA = ORD ? 2 : R;      // accumulator slots of the axpy pass
    constexpr int ND = R > 1 ? R : M;    // deciding waves: one per 64 rows (M of them), or -- several replicas, M = 1 -- one per replica
    constexpr int PW = ND;               // the wave that polls and builds the lists (the first one that decides nothing)
    extern __shared__ unsigned long long own_lds[];
    __shared__ int s_nl, s_nlo, s_nle, s_cmd, s_fail;
    __shared__ int s_nlr[2][OWN_MAX_R];  // (replicas) toggles per replica in the running generation (double-buffered by generation)
    __shared__ unsigned s_long[3];       // (replicas) bit rho: replica rho's list of this generation needs all sixteen waves
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int w = (int)blockIdx.x, W = (int)gridDim.x, G = W * M;
    const int n = P.n;
    const TJ* __restrict__ JT = (const TJ*)P.JT;
    const int NP = ORD ? G : P.sbw * M;   // groups per exchange, at most
    const int LMAX = P.lmax;
    unsigned long long* vmask = own_lds;        // [R][G]   committed values of every site, 64 per group
    unsigned long long* gm = vmask + R * G;     // [R][NP]  values of the running generation (the polled groups)
    unsigned long long* nm = gm + R * NP;       // [R][NP]  masks just polled
    double* red = (double*)(nm + R * NP);       // [OWN_WAVES][NA][M][64]
    uint32_t* lst = (uint32_t*)(red + OWN_WAVES * NA * M * 64);  // [LMAX] site | codes << 16 (per replica: bit 0 toggled, bit 1 value now 0)
    double* dF = (double*)(lst + LMAX);         // [R][M][64] the owners' state, in LDS so that it does not occupy registers of every wave:
    double* dC = dF + R * M * 64;               //   field at the start of the superblock, correction of the running generation,
    double* dL = dC + R * M * 64;               //   logit of the uniform,
    double* dA = dL + R * M * 64;               //   (ORD) the running field's change inside the current superblock  [M][64]
    unsigned short* pos = (unsigned short*)(dA + M * 64);        // (ORD) [n] position of every site in this sweep's order
    uint32_t* lsr = (uint32_t*)(dA + M * 64);                    // (replicas) [R][64] a replica's own short toggle list
#define OWN_AT(rho) (((rho) * M + (R > 1 ? 0 : wv)) * 64 + lane)
// the replicas a deciding wave handles: all of them (one replica), or its own (wave = replica)
#define OWN_FOR_RHO(rho) for (int rho = (R > 1 ? wv : 0); rho < (R > 1 ? wv + 1 : 1); ++rho)

    const int r0 = w * RW;
    const int myrow = r0 + 4 * (lane % (16 * M));       // the axpy pass: this lane's quad of rows myrow .. myrow + 3
    const int col0 = myrow + 4 <= n ? myrow : n - 4;    // (n is a multiple of 4: whole quads; lanes beyond n read the last one, unused)
    const bool decider = wv < ND;                       // wave e < M, lane l decides site r0 + 64 e + l (replicas: wave = replica, site r0 + l)
    const int drow = (decider && R == 1) ? wv : 0;      // my 64-row group inside the workgroup
    const int site = r0 + 64 * drow + lane;
    const bool site_ok = decider && site < n;
    const int mygroup = w * M + drow;
    if (threadIdx.x == 0) s_fail = 0;
    if (threadIdx.x < 3) s_long[threadIdx.x] = 0u;
    // the axpy pass in the layout that suits the replica count (see own_axpy / own_axpy_rep)
#define OWN_AXPY_ALL(nl_)                                                                                        \
    do {                                                                                                         \
        if (R > 1 && !OWN_REP_QUAD) own_axpy_rep<TJ, R, false, NA>(JT, n, lst, (nl_), rcol, r0 + lane, red, lane, wv); \
        else own_axpy<TJ, M, R, 0, NA>(JT, n, lst, 0, (nl_), col0, myrow, pos, mypos, red, lane, wv);            \
    } while (0)
#define OWN_AXPY_GEN(nlo_, nle_)                                                                                 \
    do {                                                                                                         \
        if (R > 1 && !OWN_REP_QUAD) own_axpy_rep<TJ, R, true, NA>(JT, n, lst, (nle_), rcol, r0 + lane, red, lane, wv);  \
        else own_axpy<TJ, M, R, 1, NA>(JT, n, lst, (nlo_), (nle_), col0, myrow, pos, mypos, red, lane, wv);      \
    } while (0)
    const int rcol = r0 + lane < n ? r0 + lane : n - 1;  // (replica layout: lane = row)

    const bool timing = P.timeline && (int)blockIdx.x == W - 1 && threadIdx.x == 0;
    __shared__ unsigned long long tl[12];  // (in LDS: a per-thread array would occupy registers of every wave)
    if (threadIdx.x < 12) tl[threadIdx.x] = 0ull;
    long long tl_last = wall_clock64();
#define OWN_MARK(kind)                                    \
    if (timing) {                                         \
        const long long now_ = wall_clock64();            \
        tl[kind] += (unsigned long long)(now_ - tl_last); \
        tl_last = now_;                                   \
    }

    // ---- committed values of every site as bit masks (one per 64-site group), in every workgroup
    // (eight loads in flight per lane: one at a time this loop was a chain of R G / 16 memory round trips -- 190 us of every call
    // with eight replicas at n = 16384)
    for (int q0 = wv; q0 < R * G; q0 += 8 * OWN_WAVES) {
        int b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int q = q0 + u * OWN_WAVES;
            const int rho = q / G, g = q - rho * G;
            const int sj = 64 * g + lane;
            b[u] = (q < R * G && sj < n) ? (int)P.state[(size_t)rho * n + sj] : 0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int q = q0 + u * OWN_WAVES;
            const unsigned long long mk = __ballot(b[u] != 0);
            if (lane == 0 && q < R * G) vmask[q] = mk;
        }
    }
    __syncthreads();

    // ---- the poller wave's tools -------------------------------------------------------------------------------------------
    // poll the granules of groups [g_lo, g_lo + np) of every replica in `buf` until all carry `tag`; the masks go to nm
    auto poll = [&](const unsigned long long* buf, unsigned tag, int g_lo, int np, bool sleepy) {
        constexpr int PC = R == 1 ? 4 : 1;  // chunks of 64 groups polled together (one round trip for 256 groups, not four)
        for (int k0 = 0; k0 < np; k0 += 64 * PC) {
            unsigned long long lo[PC][R], hi[PC][R];
            const long long t0 = wall_clock64();
            bool missed = false;
            for (unsigned spins = 0;; ++spins) {
                bool all = true;
#pragma unroll
                for (int c = 0; c < PC; ++c) {
                    if (k0 + 64 * c >= np) break;  // (wave-uniform)
                    const int k = k0 + 64 * c + lane;
                    const bool okk = k < np;
                    const unsigned long long* p = buf + (size_t)(g_lo + (okk ? k : 0)) * 2;
#pragma unroll
                    for (int rho = 0; rho < R; ++rho) {
                        lo[c][rho] = ld(p + (size_t)rho * G * 2);
                        hi[c][rho] = ld(p + (size_t)rho * G * 2 + 1);
                        const unsigned tl_ = (unsigned)(lo[c][rho] >> 32), th_ = (unsigned)(hi[c][rho] >> 32);
                        all = all && (!okk || (tl_ == tag && th_ == tag));
                        // (a follower that fell a whole ring behind finds a LATER generation's tag: tags only grow in a buffer)
                        if (okk && (tl_ > tag || th_ > tag)) missed = true;
                    }
                }
                if (__ballot(missed) != 0ull) {
                    st(&P.bar[BAR_ERR], 1u);
                    s_fail = 1;
                    break;
                }
                if (__ballot(!all) == 0ull) break;
                if (sleepy) __builtin_amdgcn_s_sleep(16);
                if ((spins & 63u) == 63u) {
                    if (ld(&P.bar[BAR_ERR]) || wall_clock64() - t0 > CO_TIMEOUT) {
                        st(&P.bar[BAR_ERR], 1u);
                        s_fail = 1;
                        break;
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < PC; ++c) {
                const int k = k0 + 64 * c + lane;
                if (k < np) {
#pragma unroll
                    for (int rho = 0; rho < R; ++rho) nm[rho * NP + k] = (lo[c][rho] & 0xFFFFFFFFull) | (hi[c][rho] << 32);
                }
            }
        }
    };
    // list of the sites whose value differs between newp[rho * sn + k] and oldp[rho * so + k] (oldp == nullptr: all zero), k < np,
    // group g_lo + k, ascending; with write_back the old masks become the new ones.  s_nl = entries, s_nle = entries up to and
    // including this workgroup's own groups (natural order: later sites never enter a correction).
    auto build_list = [&](const unsigned long long* newp, int sn, unsigned long long* oldp, int so, int g_lo, int np, bool write_back) {
        // lane l takes the groups [l gpl, (l + 1) gpl): one prefix scan whatever np is, entries in ascending site order
        const int gpl = (np + 63) >> 6;
        const int kb = lane * gpl, ke = kb + gpl < np ? kb + gpl : np;
        int pc = 0;
        for (int k = kb; k < ke; ++k) {
            unsigned long long un = 0ull;
#pragma unroll
            for (int rho = 0; rho < R; ++rho) un |= newp[rho * sn + k] ^ (oldp ? oldp[rho * so + k] : 0ull);
            pc += __popcll(un);
        }
        int off = pc;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) {
            const int v = __shfl_up(off, dd, 64);
            if (lane >= dd) off += v;
        }
        const int total = __shfl(off, 63, 64);
        off -= pc;
        if (lane == 0) {
            s_nle = 0;
            s_nlo = 0;
        }
        for (int k = kb; k < ke; ++k) {
            const int g = g_lo + k;
            const int sbase = 64 * g;
            if (g == w * M) s_nlo = off;
            if (R == 1) {
                const unsigned long long nw = newp[k], tg = nw ^ (oldp ? oldp[k] : 0ull);
                if (write_back) oldp[k] = nw;
                unsigned long long un = tg;
                while (un) {
                    const int b = __ffsll((long long)un) - 1;
                    un &= un - 1ull;
                    lst[off++] = (uint32_t)(sbase + b) | ((1u | (((nw >> b) & 1ull) ? 0u : 2u)) << 16);
                }
            } else {
                // (several replicas: the masks are read from LDS again per entry rather than kept in 4 R registers -- this kernel is
                // short of them; the old masks are written back after the entries are out)
                unsigned long long un = 0ull;
#pragma unroll
                for (int rho = 0; rho < R; ++rho) un |= newp[rho * sn + k] ^ (oldp ? oldp[rho * so + k] : 0ull);
                while (un) {
                    const int b = __ffsll((long long)un) - 1;
                    un &= un - 1ull;
                    uint32_t code = 0u;
#pragma unroll
                    for (int rho = 0; rho < R; ++rho) {
                        const unsigned long long nw = newp[rho * sn + k], tg = nw ^ (oldp ? oldp[rho * so + k] : 0ull);
                        if ((tg >> b) & 1ull) code |= (1u | (((nw >> b) & 1ull) ? 0u : 2u)) << (2 * rho);
                    }
                    lst[off++] = (uint32_t)(sbase + b) | (code << 16);
                }
                if (write_back) {
#pragma unroll
                    for (int rho = 0; rho < R; ++rho) oldp[rho * so + k] = newp[rho * sn + k];
                }
            }
            if (g == w * M + M - 1) s_nle = off;
        }
        if (lane == 0) s_nl = total;
    };

    // (natural order) the workgroups that decide the NEXT superblock have its predecessor's flips already (they followed its generations)
    // and now run their first, long correction pass (every first guess that flips): everybody else holds its own strip back until
    // their second generation is out (a hint only -- bounded, nothing depends on it: 30 us at most)
    auto yield_to_next = [&](unsigned seq_now, int sb_now, int sw_now) {
        const int nsb_ = (W + P.sbw - 1) / P.sbw;
        const int sbn = sb_now + 1 < nsb_ ? sb_now + 1 : 0;
        const int sbnn = sbn + 1 < nsb_ ? sbn + 1 : 0;  // its followers are on a tight schedule too (they must keep up with the generations)
        if (nsb_ < 2 || w / P.sbw == sbn || w / P.sbw == sbnn || (sbn == 0 && sw_now + 1 >= P.n_sweeps)) return;
        if (lane == 0) {
            const int gl = ((sbn + 1) * P.sbw * M < G ? (sbn + 1) * P.sbw * M : G) - 1;
            const unsigned long long* p = P.gen + ((size_t)1 * R * G + gl) * 2;  // generation buffer 1, replica 0
            const unsigned want = (seq_now + 1u) * OWN_TAG_SPAN + 2u;
            const long long t0 = wall_clock64();
            while ((unsigned)(ld(p) >> 32) < want && wall_clock64() - t0 < 3000) __builtin_amdgcn_s_sleep(32);
        }
    };



    // one replica's granules of groups [g_lo, g_lo + np), np <= 64, by wave rho = wv: lane k <-> group g_lo + k; the masks go to nm
    auto poll_one = [&](const unsigned long long* buf, unsigned tag, int g_lo, int np, bool sleepy) {
        const int rho = wv;
        const bool okk = lane < np;
        const unsigned long long* pp = buf + ((size_t)rho * G + g_lo + (okk ? lane : 0)) * 2;
        const long long t0 = wall_clock64();
        unsigned long long lo = 0ull, hi = 0ull;
        for (unsigned spins = 0;; ++spins) {
            lo = ld(pp);
            hi = ld(pp + 1);
            const bool all = !okk || ((unsigned)(lo >> 32) == tag && (unsigned)(hi >> 32) == tag);
            if (__ballot(!all) == 0ull) break;
            if (sleepy) __builtin_amdgcn_s_sleep(16);
            if ((spins & 63u) == 63u && (ld(&P.bar[BAR_ERR]) || wall_clock64() - t0 > CO_TIMEOUT)) {
                st(&P.bar[BAR_ERR], 1u);
                s_fail = 1;
                return;
            }
        }
        if (okk) nm[rho * NP + lane] = (lo & 0xFFFFFFFFull) | (hi << 32);
    };

    // ---- replicas, generation by generation WITHOUT a polling wave: deciding wave rho handles replica rho's granules itself --------
    // poll (lane k <-> group g_lo + k), toggle scan, and -- most generations move a handful of sites -- the row segments of a short list
    // in the quad layout, added straight to `dst` (dC: masked, entries before my rows; dF: every entry).  Long lists are left to all
    // sixteen waves: bit rho of s_long, toggles in nm, new values in gm (build_union below).  Eight pollers and eight list builders
    // in parallel instead of one wave doing eight replicas' worth of both.
    auto rep_step = [&](const unsigned long long* gbuf, unsigned tag, int g_lo, int np, unsigned gi, bool masked, double* dst) {
        const int rho = wv;
        const long long rp0 = (timing && gi > 0 && masked) ? wall_clock64() : 0;
        unsigned long long nw = 0ull;
        {
            const bool okk = lane < np;
            const unsigned long long* pp = gbuf + ((size_t)rho * G + g_lo + (okk ? lane : 0)) * 2;
            const long long t0 = wall_clock64();
            for (unsigned spins = 0;; ++spins) {
                const unsigned long long lo = ld(pp), hi = ld(pp + 1);
                const unsigned tl_ = (unsigned)(lo >> 32), th_ = (unsigned)(hi >> 32);
                const bool all = !okk || (tl_ == tag && th_ == tag);
                nw = (lo & 0xFFFFFFFFull) | (hi << 32);
                if (__ballot(okk && (tl_ > tag || th_ > tag)) != 0ull) {  // (a follower a whole ring behind: tags only grow in a buffer)
                    st(&P.bar[BAR_ERR], 1u);
                    s_fail = 1;
                    return;
                }
                if (__ballot(!all) == 0ull) break;
                if ((spins & 63u) == 63u && (ld(&P.bar[BAR_ERR]) || wall_clock64() - t0 > CO_TIMEOUT)) {
                    st(&P.bar[BAR_ERR], 1u);
                    s_fail = 1;
                    return;
                }
            }
        }
        const bool rtiming = timing && gi > 0 && masked;  // (wave 0 of the last workgroup: its own polling and what follows it)
        const long long rp1 = rtiming ? wall_clock64() : 0;
        if (rtiming) tl[10] += (unsigned long long)(rp1 - rp0);
        unsigned long long tg = 0ull;
        if (lane < np) {
            tg = nw ^ gm[rho * NP + lane];
            gm[rho * NP + lane] = nw;
            nm[rho * NP + lane] = tg;
        }
        const int pc = __popcll(tg);
        int off = pc;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) {
            const int x = __shfl_up(off, dd, 64);
            if (lane >= dd) off += x;
        }
        const int nl = __shfl(off, 63, 64);
        off -= pc;
        const int kme = w - g_lo;  // (M = 1: my group among the polled ones; a follower's lies outside: everything counts)
        const int cnt = masked ? __shfl(off + pc, kme, 64) : nl;
        if (lane == 0) s_nlr[gi & 1u][rho] = nl;
        if (cnt == 0) return;
        if (cnt > P.solo_max) {
            if (lane == 0) atomicOr(&s_long[gi % 3u], 1u << rho);
            return;
        }
        uint32_t* mylst = lsr + rho * 64;
        if (off < cnt) {
            unsigned long long un = tg;
            const int g = g_lo + lane;
            while (un) {
                const int b = __ffsll((long long)un) - 1;
                un &= un - 1ull;
                mylst[off++] = (uint32_t)(64 * g + b) | ((1u | (((nw >> b) & 1ull) ? 0u : 2u)) << 16);
            }
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the list is read back by other lanes of this wave)
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        const int t = lane >> 4, nb = (cnt + 3) >> 2;
        constexpr int SU = sizeof(TJ) == 4 ? 4 : 2;  // bundles in flight
        for (int b0 = 0; b0 < nb; b0 += SU) {
            OwnQuad<TJ> x[SU];
            uint32_t e[SU];
#pragma unroll
            for (int u = 0; u < SU; ++u) {
                const int k = (b0 + u) * 4 + t;
                e[u] = k < cnt ? mylst[k] : 0u;
            }
#pragma unroll
            for (int u = 0; u < SU; ++u)
                if (b0 + u < nb) x[u].load(JT + (size_t)(e[u] & 0xFFFFu) * n + col0);
#pragma unroll
            for (int u = 0; u < SU; ++u)
                if (b0 + u < nb) {
                    const int j = (int)(e[u] & 0xFFFFu);
                    const double sg = (double)(((int)(e[u] << 14)) >> 30);
#pragma unroll
                    for (int m = 0; m < 4; ++m) acc[m] += (!masked || j < myrow + m) ? sg * x[u].get(m) : 0.0;
                }
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            acc[m] += __shfl_xor(acc[m], 16, 64);
            acc[m] += __shfl_xor(acc[m], 32, 64);
        }
        double* myred = red + (size_t)wv * NA * RW;  // (this wave's own part of the partial-sum area)
        if (lane < 16) {
#pragma unroll
            for (int m = 0; m < 4; ++m) myred[4 * lane + m] = acc[m];
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        dst[OWN_AT(rho)] += myred[lane];
        __builtin_amdgcn_wave_barrier();
        if (rtiming) tl[11] += (unsigned long long)(wall_clock64() - rp1);
    };
    // The union list of the replicas in `lm`, built by ALL sixteen waves (np <= 64 groups): toggles in tgp[rho * NP + k], new values in
    // vp[rho * NP + k].  Every wave computes the groups' offsets (lane k <-> group k, one prefix scan), then takes the groups wv, wv + 16,
    // ...: lane b = site b of the group, its entry goes to offset + (toggled sites below it).  (One wave building the first
    // generation's ~4000 entries, 8 replicas' codes each, took 40 us.)
    auto build_par = [&](const unsigned long long* tgp, int ts, const unsigned long long* vp, int vs, unsigned lm, int g_lo, int np) {
        unsigned long long un = 0ull;
        if (lane < np) {
#pragma unroll
            for (int rho = 0; rho < R; ++rho)
                if ((lm >> rho) & 1u) un |= tgp[rho * ts + lane];
        }
        const int pc = __popcll(un);
        int off = pc;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) {
            const int x = __shfl_up(off, dd, 64);
            if (lane >= dd) off += x;
        }
        const int total = __shfl(off, 63, 64);
        off -= pc;
        if (wv == 0) {
            const int kme = w * M - g_lo;  // (M = 1) my group among these, if it is one of them
            const int nlo = (kme >= 0 && kme < np) ? __shfl(off, kme, 64) : 0;
            const int nle = (kme >= 0 && kme < np) ? __shfl(off + pc, kme, 64) : 0;
            if (lane == 0) {
                s_nl = total;
                s_nlo = nlo;
                s_nle = nle;
            }
        }
        for (int k = wv; k < np; k += OWN_WAVES) {
            const unsigned long long ug = __shfl(un, k, 64);  // (wave-uniform)
            if (ug == 0ull) continue;
            const int og = __shfl(off, k, 64);
            if ((ug >> lane) & 1ull) {
                uint32_t code = 0u;
#pragma unroll
                for (int rho = 0; rho < R; ++rho)
                    if (((lm >> rho) & 1u) && ((tgp[rho * ts + k] >> lane) & 1ull)) code |= (1u | (((vp[rho * vs + k] >> lane) & 1ull) ? 0u : 2u)) << (2 * rho);
                lst[og + __popcll(ug & ((1ull << lane) - 1ull))] = (uint32_t)(64 * (g_lo + k) + lane) | (code << 16);
            }
        }
    };

    // ---- the rows' owners: field, correction, logit and committed value per replica ------------------------------------------
    int vcur = 0;    // committed value of my site, bit rho
    if (decider) {
        OWN_FOR_RHO(rho) {
            dF[OWN_AT(rho)] = 0.0;
            dC[OWN_AT(rho)] = 0.0;
            dL[OWN_AT(rho)] = 0.0;
            vcur |= (int)((vmask[rho * G + mygroup] >> lane) & 1ull) << rho;
        }
        dA[drow * 64 + lane] = 0.0;
    }
    int mypos[4];  // (ORD) positions of this lane's axpy rows
#pragma unroll
    for (int m = 0; m < 4; ++m) mypos[m] = 0;
    int dpos = site;  // position of my site in the visiting order (deciders)
    // owners add the sixteen waves' partial sums of slot a in wave order
    auto reduced = [&](int a) -> double {
        double tot = 0.0;
#pragma unroll
        for (int u = 0; u < OWN_WAVES; ++u) tot += red[(u * NA + a) * RW + 64 * drow + lane];
        return tot;
    };

    const int nsb = ORD ? (n + P.sbw * RW - 1) / (P.sbw * RW) : (W + P.sbw - 1) / P.sbw;
    unsigned seq = 0;  // superblocks so far
    for (int sw = 0; sw < P.n_sweeps; ++sw) {
        if (ORD) {
            const int64_t* ord = P.order + (size_t)sw * n;
            for (int p = (int)threadIdx.x; p < n; p += OWN_THREADS) pos[ord[p]] = (unsigned short)p;
            __syncthreads();
#pragma unroll
            for (int m = 0; m < 4; ++m) mypos[m] = myrow + m < n ? (int)pos[myrow + m] : 0x7FFFFFFF;
            dpos = site_ok ? (int)pos[site] : 0x7FFFFFFF;
        }
        // a sweep starts from complete fields: computed from scratch every CO_REFRESH sweeps (counted across calls) and at the start
        // of a call -- unless the previous call left them (resume) -- and kept current by the axpy passes in between
        const bool start_phase = sw == 0 || ((sw + P.refresh_off) % CO_REFRESH) == 0;
        if (start_phase) {
            if (sw == 0 && P.resume) {
                if (decider) {
                    OWN_FOR_RHO(rho) {
                        if (site_ok) dF[OWN_AT(rho)] = P.fields_all[(size_t)(R > 1 ? P.rep[rho].src : 0) * n + site];
                    }
                }
            } else {
                if (decider) {
                    OWN_FOR_RHO(rho) dF[OWN_AT(rho)] = (site_ok && P.bias) ? P.bias[site] : 0.0;
                }
                const int lch = LMAX / 64;  // groups per list
                for (int g0 = 0; g0 < G; g0 += lch) {
                    const int np = G - g0 < lch ? G - g0 : lch;
                    if constexpr (R > 1) {
                        build_par(vmask + g0, G, vmask + g0, G, (1u << R) - 1u, g0, np);  // (every set bit: a toggle from 0 to 1)
                    } else {
                        if (wv == PW) build_list(vmask + g0, G, nullptr, 0, g0, np, false);
                    }
                    __syncthreads();
                    const int nl = s_nl;
                    OWN_AXPY_ALL(nl);
                    __syncthreads();
                    if (decider) {
                        OWN_FOR_RHO(rho) dF[OWN_AT(rho)] += reduced(rho);
                    }
                    __syncthreads();  // red and the list are rewritten
                }
            }
        }
        for (int sb = 0; sb < nsb; ++sb, ++seq) {
            if (P.fail_at >= 0 && (int)seq == P.fail_at) {  // (tests of the callers' recovery: the state is half updated here)
                if (threadIdx.x == 0) st(&P.bar[BAR_ERR], 1u);
                return;
            }
            const bool active = ORD || w / P.sbw == sb;
            const int g_lo = ORD ? 0 : sb * P.sbw * M;
            const int np = ORD ? G : (G - g_lo < NP ? G - g_lo : NP);
            const int p0 = sb * P.sbw * RW, p1 = p0 + P.sbw * RW;  // (ORD) positions of the superblock
            const unsigned slot = seq % OWN_RING;
            if (active) {
                const bool mine = site_ok && (!ORD || (dpos >= p0 && dpos < p1));  // my site is decided in this superblock
                // (per-replica parameters are read where they are used: eight replicas' worth of them hoisted here cost 200 scalar spills)
#define OWN_UNI(rho) (P.uniforms ? P.uniforms + ((size_t)(rho) * P.n_sweeps + sw) * n : nullptr)
#define OWN_TW(rho) (P.temps ? P.temps[sw] : P.rep[rho].T)
                if (decider) OWN_FOR_RHO(rho) {
                    if (mine)
                        dL[OWN_AT(rho)] = own_logit((uint32_t)site, (uint32_t)dpos, OWN_UNI(rho), P.rep[rho].sweep0 + (uint32_t)sw, P.rep[rho].tag, P.rep[rho].k0,
                                                    P.rep[rho].k1);
                    dC[OWN_AT(rho)] = 0.0;
                }
                if (decider && ORD) dA[drow * 64 + lane] = 0.0;
                // the running generation starts from the committed values
                for (int q = (int)threadIdx.x; q < R * np; q += OWN_THREADS) {
                    const int rho = q / np, k = q - rho * np;
                    gm[rho * NP + k] = vmask[rho * G + g_lo + k];
                }
                __syncthreads();
                int vnew = vcur;
                unsigned gi = 0;
                OWN_MARK(0);
                // ---- SOLO generations (natural order, one replica, 64 rows per workgroup): wave 0 runs a generation from the decision to
                // the next decision by itself -- publish, poll, toggle list, the row segments of a SHORT list (most generations move a
                // handful of sites), the sum over its entry slots, the new correction -- with no workgroup barrier, no hand-over to a
                // polling wave and no sixteen-wave reduction through LDS; the other fifteen waves wait at a barrier and are called in
                // only for long lists (the first correction pass).  Commands through s_cmd: 1 = an axpy pass over lst, 2 = converged.
                // A caller's order: the same, with every workgroup active -- the wave polls all the groups (four per lane), every toggle
                // goes to its rows twice (masked by position into the correction, unmasked into the running field).
                const bool solo = R == 1 && M == 1 && P.solo;
                if (solo && wv != 0) {
                    while (true) {
                        __syncthreads();  // (A) a command from wave 0
                        if (s_cmd != 1) break;
                        if (ORD) own_axpy<TJ, M, R, 2, NA>(JT, n, lst, 0, s_nl, col0, myrow, pos, mypos, red, lane, wv);
                        else OWN_AXPY_GEN(s_nlo, s_nle);
                        __syncthreads();  // (B) the partial sums are in red
                    }
                } else if (solo && ORD) {
                    // (the correction -- entries earlier in the order -- and the running field's change -- every entry -- stay in LDS, dC and
                    // dA, and the polled masks go straight to LDS: registers are short here)
                    while (true) {
                        const unsigned tag = seq * OWN_TAG_SPAN + gi + 1u;
                        unsigned long long* gbuf = P.gen + (size_t)(gi % OWN_NGEN) * G * 2;
                        int v = vcur & 1;
                        if (mine) {
                            const double Tw = OWN_TW(0);
                            v = own_decide(dF[OWN_AT(0)] + dC[OWN_AT(0)], dL[OWN_AT(0)], Tw, 1.0 / Tw, (uint32_t)site, (uint32_t)dpos, OWN_UNI(0), P.rep[0].sweep0 + (uint32_t)sw,
                                           P.rep[0].tag, P.rep[0].k0, P.rep[0].k1);
                        }
                        vnew = v;
                        const unsigned long long mk = __ballot(v != 0);
                        if (lane < 2) st(gbuf + (size_t)mygroup * 2 + lane, ((mk >> (32 * lane)) & 0xFFFFFFFFull) | ((unsigned long long)tag << 32));
                        // poll: lane l reads the granule pairs of groups 4 l .. 4 l + 3 until every pair carries this generation's tag
                        {
                            unsigned long long nw[4];
                            const long long t0 = wall_clock64();
                            bool failed = false;
                            for (unsigned spins = 0;; ++spins) {
                                bool all = true;
#pragma unroll
                                for (int g = 0; g < 4; ++g) {
                                    const int k = 4 * lane + g;
                                    const bool okk = k < np;
                                    const unsigned long long* pp = gbuf + (size_t)(okk ? k : 0) * 2;
                                    const unsigned long long lo = ld(pp), hi = ld(pp + 1);
                                    all = all && (!okk || ((unsigned)(lo >> 32) == tag && (unsigned)(hi >> 32) == tag));
                                    nw[g] = (lo & 0xFFFFFFFFull) | (hi << 32);
                                }
                                if (__ballot(!all) == 0ull) break;
                                if ((spins & 63u) == 63u && (ld(&P.bar[BAR_ERR]) || wall_clock64() - t0 > CO_TIMEOUT)) {
                                    st(&P.bar[BAR_ERR], 1u);
                                    failed = true;
                                    break;
                                }
                            }
                            if (failed) {
                                if (lane == 0) {
                                    s_fail = 1;
                                    s_cmd = 2;
                                }
                                __syncthreads();  // (A) releases the waiting waves
                                return;
                            }
                            // nm <- the toggles, gm <- the new values (a toggled site is now 0 iff its bit in gm is 0)
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                const int k = 4 * lane + g;
                                if (k < np) {
                                    nm[k] = nw[g] ^ gm[k];
                                    gm[k] = nw[g];
                                }
                            }
                        }
                        OWN_MARK(gi == 0 ? 9 : 1);
                        // toggle list, ascending sites: four consecutive groups per lane, one prefix scan
                        int pc = 0;
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int k = 4 * lane + g;
                            pc += k < np ? __popcll(nm[k]) : 0;
                        }
                        int off = pc;
#pragma unroll
                        for (int dd = 1; dd < 64; dd <<= 1) {
                            const int x = __shfl_up(off, dd, 64);
                            if (lane >= dd) off += x;
                        }
                        const int nl = __shfl(off, 63, 64);
                        off -= pc;
                        if (timing) {
                            tl[4] += 1;
                            tl[5] += (unsigned long long)nl;
                        }
                        if (nl == 0) break;  // nobody's value changed: the fixed point
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int k = 4 * lane + g;
                            unsigned long long un = k < np ? nm[k] : 0ull;
                            const unsigned long long val = k < np ? gm[k] : 0ull;
                            while (un) {
                                const int b = __ffsll((long long)un) - 1;
                                un &= un - 1ull;
                                lst[off++] = (uint32_t)(64 * k + b) | ((1u | (((val >> b) & 1ull) ? 0u : 2u)) << 16);
                            }
                        }
                        if (nl <= P.solo_max) {
                            __builtin_amdgcn_wave_barrier();
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the list is read back by other lanes of this wave)
                            double a0[4] = {0.0, 0.0, 0.0, 0.0}, a1[4] = {0.0, 0.0, 0.0, 0.0};
                            const int t = lane >> 4, nb = (nl + 3) >> 2;
                            constexpr int SU = sizeof(TJ) == 4 ? 4 : 2;  // bundles in flight
                            for (int b0 = 0; b0 < nb; b0 += SU) {
                                OwnQuad<TJ> x[SU];
                                uint32_t e[SU];
#pragma unroll
                                for (int u = 0; u < SU; ++u) {
                                    const int k = (b0 + u) * 4 + t;
                                    e[u] = k < nl ? lst[k] : 0u;
                                }
#pragma unroll
                                for (int u = 0; u < SU; ++u)
                                    if (b0 + u < nb) x[u].load(JT + (size_t)(e[u] & 0xFFFFu) * n + col0);
#pragma unroll
                                for (int u = 0; u < SU; ++u)
                                    if (b0 + u < nb) {
                                        const double sg = (double)(((int)(e[u] << 14)) >> 30);
                                        const int pj = pos[e[u] & 0xFFFFu];
#pragma unroll
                                        for (int m = 0; m < 4; ++m) {
                                            const double val = sg * x[u].get(m);
                                            a1[m] += val;
                                            a0[m] += pj < mypos[m] ? val : 0.0;
                                        }
                                    }
                            }
#pragma unroll
                            for (int m = 0; m < 4; ++m) {
                                a0[m] += __shfl_xor(a0[m], 16, 64);
                                a0[m] += __shfl_xor(a0[m], 32, 64);
                                a1[m] += __shfl_xor(a1[m], 16, 64);
                                a1[m] += __shfl_xor(a1[m], 32, 64);
                            }
                            if (lane < 16) {
#pragma unroll
                                for (int m = 0; m < 4; ++m) {
                                    red[4 * lane + m] = a0[m];
                                    red[64 + 4 * lane + m] = a1[m];
                                }
                            }
                            __builtin_amdgcn_wave_barrier();
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                            dC[OWN_AT(0)] += red[lane];
                            dA[drow * 64 + lane] += red[64 + lane];
                            __builtin_amdgcn_wave_barrier();
                        } else {
                            if (lane == 0) {
                                s_nl = nl;
                                s_cmd = 1;
                            }
                            __syncthreads();  // (A)
                            own_axpy<TJ, M, R, 2, NA>(JT, n, lst, 0, nl, col0, myrow, pos, mypos, red, lane, wv);
                            __syncthreads();  // (B)
                            dC[OWN_AT(0)] += reduced(0);
                            dA[drow * 64 + lane] += reduced(1);
                        }
                        OWN_MARK(gi == 0 ? 8 : 2);
                        ++gi;
                        if (gi + 2u >= OWN_TAG_SPAN) {
                            if (lane == 0) {
                                st(&P.bar[BAR_ERR + 2], 1u);
                                s_fail = 1;
                                s_cmd = 2;
                            }
                            __syncthreads();  // (A)
                            return;
                        }
                    }
                    if (lane == 0) s_cmd = 2;
                    __syncthreads();  // (A) converged: everybody goes on
                } else if (solo) {
                    const int kme = w - g_lo;  // my group's index among the polled ones (lane k <-> group g_lo + k)
                    double Cm = 0.0;  // (field and logit stay in LDS: registers are short here)
                    while (true) {
                        const unsigned tag = seq * OWN_TAG_SPAN + gi + 1u;
                        unsigned long long* gbuf = P.gen + (size_t)(gi % OWN_NGEN) * G * 2;
                        int v = vcur & 1;
                        if (mine) {
                            const double Tw = OWN_TW(0);
                            v = own_decide(dF[OWN_AT(0)] + Cm, dL[OWN_AT(0)], Tw, 1.0 / Tw, (uint32_t)site, (uint32_t)dpos, OWN_UNI(0), P.rep[0].sweep0 + (uint32_t)sw,
                                           P.rep[0].tag, P.rep[0].k0, P.rep[0].k1);
                        }
                        vnew = v;
                        const unsigned long long mk = __ballot(v != 0);
                        if (lane < 2) st(gbuf + (size_t)mygroup * 2 + lane, ((mk >> (32 * lane)) & 0xFFFFFFFFull) | ((unsigned long long)tag << 32));
                        // poll: lane k reads group g_lo + k's granule pair until every pair carries this generation's tag
                        unsigned long long nw = 0ull;
                        {
                            const bool okk = lane < np;
                            const unsigned long long* pp = gbuf + (size_t)(g_lo + (okk ? lane : 0)) * 2;
                            const long long t0 = wall_clock64();
                            bool failed = false;
                            for (unsigned spins = 0;; ++spins) {
                                const unsigned long long lo = ld(pp), hi = ld(pp + 1);
                                const bool all = !okk || ((unsigned)(lo >> 32) == tag && (unsigned)(hi >> 32) == tag);
                                nw = (lo & 0xFFFFFFFFull) | (hi << 32);
                                if (__ballot(!all) == 0ull) break;
                                if ((spins & 63u) == 63u && (ld(&P.bar[BAR_ERR]) || wall_clock64() - t0 > CO_TIMEOUT)) {
                                    st(&P.bar[BAR_ERR], 1u);
                                    failed = true;
                                    break;
                                }
                            }
                            if (failed) {
                                if (lane == 0) {
                                    s_fail = 1;
                                    s_cmd = 2;
                                }
                                __syncthreads();  // (A) releases the waiting waves
                                return;
                            }
                        }
                        OWN_MARK(gi == 0 ? 9 : 1);
                        // toggle list (ascending sites: one group per lane, one prefix scan)
                        const unsigned long long tg0 = lane < np ? nw ^ gm[lane] : 0ull;  // (gm: one group per lane, this wave's only)
                        if (lane < np) gm[lane] = nw;
                        const int pc = __popcll(tg0);
                        int off = pc;
#pragma unroll
                        for (int dd = 1; dd < 64; dd <<= 1) {
                            const int x = __shfl_up(off, dd, 64);
                            if (lane >= dd) off += x;
                        }
                        const int nl = __shfl(off, 63, 64);
                        off -= pc;
                        const int nlo = __shfl(off, kme, 64), nle = __shfl(off + pc, kme, 64);
                        if (timing) {
                            tl[4] += 1;
                            tl[5] += (unsigned long long)nl;
                        }
                        if (nl == 0) break;  // nobody's value changed: the fixed point
                        {
                            unsigned long long un = tg0;
                            const int g = g_lo + lane;
                            while (un) {
                                const int b = __ffsll((long long)un) - 1;
                                un &= un - 1ull;
                                lst[off++] = (uint32_t)(64 * g + b) | ((1u | (((nw >> b) & 1ull) ? 0u : 2u)) << 16);
                            }
                        }
                        if (nle <= P.solo_max) {
                            // the short list by this wave alone: bundles of 4 entries, SU in flight per round, per-lane masks throughout
                            __builtin_amdgcn_wave_barrier();
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the list is read back by other lanes of this wave)
                            double acc[4] = {0.0, 0.0, 0.0, 0.0};
                            const int t = lane >> 4, nb = (nle + 3) >> 2;
                            constexpr int SU = sizeof(TJ) == 4 ? 4 : 2;  // bundles in flight
                            for (int b0 = 0; b0 < nb; b0 += SU) {
                                OwnQuad<TJ> x[SU];
                                uint32_t e[SU];
#pragma unroll
                                for (int u = 0; u < SU; ++u) {
                                    const int k = (b0 + u) * 4 + t;
                                    e[u] = k < nle ? lst[k] : 0u;
                                }
#pragma unroll
                                for (int u = 0; u < SU; ++u)
                                    if (b0 + u < nb) x[u].load(JT + (size_t)(e[u] & 0xFFFFu) * n + col0);
#pragma unroll
                                for (int u = 0; u < SU; ++u)
                                    if (b0 + u < nb) {
                                        const int j = (int)(e[u] & 0xFFFFu);
                                        const double sg = (double)(((int)(e[u] << 14)) >> 30);
#pragma unroll
                                        for (int m = 0; m < 4; ++m) acc[m] += j < myrow + m ? sg * x[u].get(m) : 0.0;
                                    }
                            }
#pragma unroll
                            for (int m = 0; m < 4; ++m) {
                                acc[m] += __shfl_xor(acc[m], 16, 64);
                                acc[m] += __shfl_xor(acc[m], 32, 64);
                            }
                            // lanes 0..15 hold a quad of rows each: back to one row per lane through LDS (this wave only)
                            if (lane < 16) {
#pragma unroll
                                for (int m = 0; m < 4; ++m) red[4 * lane + m] = acc[m];
                            }
                            __builtin_amdgcn_wave_barrier();
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                            Cm += red[lane];
                            __builtin_amdgcn_wave_barrier();
                        } else {
                            if (lane == 0) {
                                s_nl = nl;
                                s_nlo = nlo;
                                s_nle = nle;
                                s_cmd = 1;
                            }
                            __syncthreads();  // (A)
                            OWN_AXPY_GEN(nlo, nle);
                            __syncthreads();  // (B)
                            Cm += reduced(0);
                        }
                        OWN_MARK(gi == 0 ? 8 : 2);
                        ++gi;
                        if (gi + 2u >= OWN_TAG_SPAN) {
                            if (lane == 0) {
                                st(&P.bar[BAR_ERR + 2], 1u);
                                s_fail = 1;
                                s_cmd = 2;
                            }
                            __syncthreads();  // (A)
                            return;
                        }
                    }
                    if (lane == 0) s_cmd = 2;
                    __syncthreads();  // (A) converged: everybody goes on
                }
                if (solo && s_fail) return;
                constexpr bool rsolo = R > 1;  // (replicas: always by rep_step; the host keeps their superblocks at <= 64 groups)
                while (rsolo) {
                    const unsigned tag = seq * OWN_TAG_SPAN + gi + 1u;
                    unsigned long long* gbuf = P.gen + (size_t)(gi % OWN_NGEN) * R * G * 2;
                    if (decider) {
                        OWN_FOR_RHO(rho) {
                            int v = (vcur >> rho) & 1;
                            if (mine) {
                                const double Tw = OWN_TW(rho);
                                v = own_decide(dF[OWN_AT(rho)] + dC[OWN_AT(rho)], dL[OWN_AT(rho)], Tw, 1.0 / Tw, (uint32_t)site, (uint32_t)dpos, OWN_UNI(rho),
                                               P.rep[rho].sweep0 + (uint32_t)sw, P.rep[rho].tag, P.rep[rho].k0, P.rep[rho].k1);
                            }
                            vnew = (vnew & ~(1 << rho)) | (v << rho);
                            const unsigned long long mk = __ballot(v != 0);
                            if (lane < 2)
                                st(gbuf + ((size_t)rho * G + mygroup) * 2 + lane, ((mk >> (32 * lane)) & 0xFFFFFFFFull) | ((unsigned long long)tag << 32));
                        }
                        rep_step(gbuf, tag, g_lo, np, gi, true, dC);
                    }
                    __syncthreads();  // (A) every replica's count and list (or its bit in s_long) is there
                    if (s_fail) return;
                    if (threadIdx.x == 0) s_long[(gi + 2u) % 3u] = 0u;
                    int total = 0;
#pragma unroll
                    for (int rho = 0; rho < R; ++rho) total += s_nlr[gi & 1u][rho];
                    OWN_MARK(gi == 0 ? 9 : 1);
                    if (timing) {
                        tl[4] += 1;
                        tl[5] += (unsigned long long)total;
                    }
                    if (total == 0) break;  // nobody's value changed in any replica: the fixed point
                    const unsigned lm = s_long[gi % 3u];
                    if (lm) {
                        build_par(nm, NP, gm, NP, lm, g_lo, np);
                        __syncthreads();
                        OWN_AXPY_GEN(s_nlo, s_nle);
                        __syncthreads();
                        if (decider && ((lm >> wv) & 1u)) dC[OWN_AT(wv)] += reduced(wv);
                        __syncthreads();  // (red is rewritten by the next generation's solo passes)
                    }
                    OWN_MARK(gi == 0 ? 8 : 2);
                    ++gi;
                    if (gi + 2u >= OWN_TAG_SPAN) {
                        if (threadIdx.x == 0) st(&P.bar[BAR_ERR + 2], 1u);
                        return;
                    }
                }
                if constexpr (R == 1) while (!solo) {
                    const unsigned tag = seq * OWN_TAG_SPAN + gi + 1u;
                    unsigned long long* gbuf = P.gen + (size_t)(gi % OWN_NGEN) * R * G * 2;
                    if (decider) {
                        OWN_FOR_RHO(rho) {
                            int v = (vcur >> rho) & 1;
                            if (mine) {
                                const double Tw = OWN_TW(rho);
                                v = own_decide(dF[OWN_AT(rho)] + dC[OWN_AT(rho)], dL[OWN_AT(rho)], Tw, 1.0 / Tw, (uint32_t)site, (uint32_t)dpos, OWN_UNI(rho),
                                               P.rep[rho].sweep0 + (uint32_t)sw, P.rep[rho].tag, P.rep[rho].k0, P.rep[rho].k1);
                            }
                            vnew = (vnew & ~(1 << rho)) | (v << rho);
                            const unsigned long long mk = __ballot(v != 0);
                            if (lane < 2)
                                st(gbuf + ((size_t)rho * G + mygroup) * 2 + lane, ((mk >> (32 * lane)) & 0xFFFFFFFFull) | ((unsigned long long)tag << 32));
                        }
                    }
                    if (wv == PW) {
                        const bool ptiming = P.timeline && w == W - 1 && lane == 0 && gi > 0;
                        const long long p0_ = ptiming ? wall_clock64() : 0;
                        poll(gbuf, tag, g_lo, np, false);
                        const long long p1_ = ptiming ? wall_clock64() : 0;
                        build_list(nm, NP, gm, NP, g_lo, np, true);
                        if (ptiming) {
                            tl[10] += (unsigned long long)(p1_ - p0_);
                            tl[11] += (unsigned long long)(wall_clock64() - p1_);
                        }
                    }
                    __syncthreads();
                    if (s_fail) return;
                    const int nl = s_nl;
                    OWN_MARK(gi == 0 ? 9 : 1);
                    if (timing) {
                        tl[4] += 1;
                        tl[5] += (unsigned long long)nl;
                    }
                    if (nl == 0) break;  // nobody's value changed: the fixed point
                    if (ORD) own_axpy<TJ, M, R, 2, NA>(JT, n, lst, 0, nl, col0, myrow, pos, mypos, red, lane, wv);
                    else OWN_AXPY_GEN(s_nlo, s_nle);
                    __syncthreads();
                    if (decider) {
                        if (ORD) {
                            dC[OWN_AT(0)] += reduced(0);
                            dA[drow * 64 + lane] += reduced(1);
                        } else {
                            OWN_FOR_RHO(rho) dC[OWN_AT(rho)] += reduced(rho);
                        }
                    }
                    OWN_MARK(gi == 0 ? 8 : 2);
                    ++gi;
                    if (gi + 2u >= OWN_TAG_SPAN) {  // cannot happen (the iteration is exact after as many generations as the superblock has positions)
                        if (threadIdx.x == 0) st(&P.bar[BAR_ERR + 2], 1u);
                        return;
                    }
                }
                __syncthreads();  // (every wave has read s_nl before the poller writes it again)
                // converged: gm holds the final values.  Commit my site, publish the final masks for the workgroups that did not follow
                int8_t* rec = nullptr;
                if (P.samples && sw + 1 > P.rec_from && (sw + 1 - P.rec_from) % P.rec_every == 0)
                    rec = P.samples + (size_t)((sw + 1 - P.rec_from) / P.rec_every - 1) * n;
                if (decider) {
                    OWN_FOR_RHO(rho) {
                        const int v = (vnew >> rho) & 1;
                        if (mine) {
                            P.state[(size_t)rho * n + site] = (int8_t)v;
                            if (rec) rec[site] = (int8_t)v;
                        }
                        if (!ORD) {
                            const unsigned long long mk = __ballot(v != 0);
                            if (lane < 2)
                                st(P.fin + (((size_t)slot * R + rho) * G + mygroup) * 2 + lane,
                                   ((mk >> (32 * lane)) & 0xFFFFFFFFull) | ((unsigned long long)(seq + 1u) << 32));
                        }
                    }
                    vcur = vnew;
                }
                if (ORD) {
                    // every workgroup followed every generation: the running field is current
                    if (decider) dF[OWN_AT(0)] += dA[drow * 64 + lane];
                    for (int q = (int)threadIdx.x; q < np; q += OWN_THREADS) vmask[q] = gm[q];
                    __syncthreads();
                } else {
                    // my rows' fields: the superblock's final flips in one pass (every entry, unmasked; the correction is dropped)
                    if (wv == PW) {
                        build_list(gm, NP, vmask + g_lo, G, g_lo, np, true);
                        yield_to_next(seq, sb, sw);
                    }
                    __syncthreads();
                    const int nl = s_nl;
                    OWN_AXPY_ALL(nl);
                    __syncthreads();
                    if (decider) {
                        OWN_FOR_RHO(rho) dF[OWN_AT(rho)] += reduced(rho);
                    }
                    __syncthreads();
                }
                OWN_MARK(3);
            } else {
                const int nsb_n = nsb;
                const bool follower = w / P.sbw == (sb + 1 < nsb_n ? sb + 1 : 0);
                if (follower) {
                    // my superblock is the NEXT one: follow this superblock's generations as its deciders do -- every generation's
                    // toggles go to all my rows at once (unmasked), in generation order: when it converges my fields are complete and
                    // my own first generation starts at once, instead of after a strip of ~1700 row segments.  Not waited for by
                    // anybody: the generation buffers are a ring, a missed generation fails the call (never seen: a follower's pass
                    // over a generation costs a third of a decider's).
                    OWN_MARK(0);
                    for (int q = (int)threadIdx.x; q < R * np; q += OWN_THREADS) {
                        const int rho = q / np, k = q - rho * np;
                        gm[rho * NP + k] = vmask[rho * G + g_lo + k];
                    }
                    __syncthreads();
                    constexpr bool rsolo = R > 1;
                    for (unsigned gi = 0; rsolo; ++gi) {
                        if (decider) rep_step(P.gen + (size_t)(gi % OWN_NGEN) * R * G * 2, seq * OWN_TAG_SPAN + gi + 1u, g_lo, np, gi, false, dF);
                        __syncthreads();  // (A)
                        if (s_fail) return;
                        if (threadIdx.x == 0) s_long[(gi + 2u) % 3u] = 0u;
                        int total = 0;
#pragma unroll
                        for (int rho = 0; rho < R; ++rho) total += s_nlr[gi & 1u][rho];
                        if (total == 0) break;
                        const unsigned lm = s_long[gi % 3u];
                        if (lm) {
                            build_par(nm, NP, gm, NP, lm, g_lo, np);
                            __syncthreads();
                            OWN_AXPY_ALL(s_nl);
                            __syncthreads();
                            if (decider && ((lm >> wv) & 1u)) dF[OWN_AT(wv)] += reduced(wv);
                            __syncthreads();
                        }
                        if (gi + 3u >= OWN_TAG_SPAN) return;  // (the deciders have failed the call)
                    }
                    if constexpr (R == 1) for (unsigned gi = 0;; ++gi) {
                        if (wv == PW) {
                            poll(P.gen + (size_t)(gi % OWN_NGEN) * R * G * 2, seq * OWN_TAG_SPAN + gi + 1u, g_lo, np, false);
                            build_list(nm, NP, gm, NP, g_lo, np, true);
                        }
                        __syncthreads();
                        if (s_fail) return;
                        const int nl = s_nl;
                        if (nl == 0) break;
                        OWN_AXPY_ALL(nl);
                        __syncthreads();
                        if (decider) {
                            OWN_FOR_RHO(rho) dF[OWN_AT(rho)] += reduced(rho);
                        }
                        if (gi + 3u >= OWN_TAG_SPAN) return;  // (the deciders have failed the call)
                    }
                    __syncthreads();
                    for (int q = (int)threadIdx.x; q < R * np; q += OWN_THREADS) {
                        const int rho = q / np, k = q - rho * np;
                        vmask[rho * G + g_lo + k] = gm[rho * NP + k];
                    }
                    __syncthreads();
                    OWN_MARK(6);
                } else {
                    // not my superblock, nor the one before mine: wait for its final values, apply the flips to my rows
                    OWN_MARK(0);
                    if constexpr (R > 1) {
                        // (eight waves poll a replica's final masks each; then one builds the union list)
                        if (wv < R) {
                            poll_one(P.fin + (size_t)slot * R * G * 2, seq + 1u, g_lo, np, true);
                            if (lane < np) {  // toggles against the committed values (gm is free here); the new values are committed
                                const unsigned long long nv = nm[wv * NP + lane];
                                gm[wv * NP + lane] = nv ^ vmask[wv * G + g_lo + lane];
                                vmask[wv * G + g_lo + lane] = nv;
                            }
                        }
                        __syncthreads();
                        if (s_fail) return;
                        build_par(gm, NP, nm, NP, (1u << R) - 1u, g_lo, np);
                        if (wv == PW) yield_to_next(seq, sb, sw);
                    } else {
                        if (wv == PW) {
                            poll(P.fin + (size_t)slot * R * G * 2, seq + 1u, g_lo, np, true);
                            build_list(nm, NP, vmask + g_lo, G, g_lo, np, true);
                            yield_to_next(seq, sb, sw);
                        }
                    }
                    __syncthreads();
                    if (s_fail) return;
                    OWN_MARK(6);
                    const int nl = s_nl;
                    OWN_AXPY_ALL(nl);
                    __syncthreads();
                    if (decider) {
                        OWN_FOR_RHO(rho) dF[OWN_AT(rho)] += reduced(rho);
                    }
                    __syncthreads();
                    OWN_MARK(7);
                }
            }
        }
    }
    if (P.persist && decider) {
        OWN_FOR_RHO(rho) {
            if (site_ok) P.fields_out[(size_t)rho * n + site] = dF[OWN_AT(rho)];
        }
    }
    if (timing)
        for (int x = 0; x < 12; ++x) P.timeline[x] = tl[x];
#undef OWN_MARK
#undef OWN_AT
#undef OWN_FOR_RHO
#undef OWN_AXPY_ALL
#undef OWN_UNI
#undef OWN_TW
#undef OWN_AXPY_GEN
}

// ---------------------------------------------------------------------------------------------------------------------- host side
typedef void (*own_kern)(OwnParams);

template <typename TJ>
static own_kern own_pick(int M, int R, bool ord) {
    if (ord) return M == 1 ? k2_own<TJ, 1, 1, true> : M == 2 ? k2_own<TJ, 2, 1, true> : k2_own<TJ, 4, 1, true>;
    if (R == 1) return M == 1 ? k2_own<TJ, 1, 1, false> : M == 2 ? k2_own<TJ, 2, 1, false> : k2_own<TJ, 4, 1, false>;
    if (M != 1) return nullptr;
    return R == 2 ? k2_own<TJ, 1, 2, false> : R == 4 ? k2_own<TJ, 1, 4, false> : R == 8 ? k2_own<TJ, 1, 8, false> : nullptr;
}

static int own_env(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

// One launch of k2_own: n_sweeps sweeps of R states.  *done = 0: the kernel does not take this call (the caller's other paths do;
// nothing was written) or it gave up half way (d->own_failed set: the caller restores the state from its backup).
//   states_dev: [R][n] device states (R == 1: d->state), uniforms_dev: [R][n_sweeps][n] or nullptr, order_dev: [n_sweeps][n] or nullptr
int tsu_dense_own_run(tsu_dense* d, int R_real, const OwnRep* reps, int8_t* states_dev, int n_sweeps, const double* uniforms_dev,
                      const int64_t* order_dev, const double* temps_dev, int8_t* samples_dev, int rec_from, int rec_every, bool fields_were_valid,
                      bool allow_persist, int* done) {
    tsu_ctx* ctx = d->ctx;
    const int n = d->n;
    *done = 0;
    static const int use_own = own_env("TSU_K2_OWN", 1);
    static const int own_min = own_env("TSU_K2_OWN_MIN", 2048);
    const int sb_env = own_env("TSU_K2_OWN_SB", 0), m_env = own_env("TSU_K2_OWN_M", 0);  // (read per call: tests vary them)
    if (!use_own || d->own_failed || n < own_min || n > 65536 || n_sweeps <= 0) return TSU_OK;
    const bool ord = order_dev != nullptr;
    if (R_real < 1 || R_real > OWN_MAX_R || (ord && R_real != 1)) return TSU_OK;
    const int R = R_real == 1 ? 1 : R_real == 2 ? 2 : R_real <= 4 ? 4 : 8;
    int M = 0;
    for (int m = (m_env == 2 || m_env == 4) ? m_env : 1; m <= 4; m *= 2)
        if ((n + 64 * m - 1) / (64 * m) <= ctx->cus) {
            M = m;
            break;
        }
    if (!M || (R > 1 && M != 1) || n % 4) return TSU_OK;  // (rows travel as quads)
    const int RW = 64 * M, W = (n + RW - 1) / RW, G = W * M;
    if (W < 2) return TSU_OK;
    // superblock: 4096 positions in natural order; a caller's order pays more per generation (every workgroup polls every group and
    // follows every toggle), so fewer, longer fixed points win there: the largest of 16384 / 8192 / 4096 whose toggle list fits the LDS
    // (n = 16384: 0.429 / 0.349 / 0.317 / 0.293 ms per sweep at 2048 / 4096 / 8192 / 16384)
    int sb = 0, sbw = 0, lmax = 0, nsb = 0, NP = 0;
    const int NA = ord ? 2 : R;
    size_t lds_bytes = 0;
    const int cands[3] = {16384, 8192, 4096};
    for (int ci = sb_env > 0 ? 2 : (ord ? 0 : 2); ci < 3; ++ci) {
        sb = sb_env > 0 ? sb_env : cands[ci];
        if (R > 1 && sb > 4096) sb = 4096;  // (replicas: a deciding wave polls one group per lane)
        if (sb < RW) sb = RW;
        if (sb > 16384) sb = 16384;
        sbw = sb / RW;
        if (sbw > W) sbw = W;
        lmax = sbw * RW;
        nsb = ord ? (n + sbw * RW - 1) / (sbw * RW) : (W + sbw - 1) / sbw;
        NP = ord ? G : sbw * M;
        lds_bytes = ((size_t)8 * (R * G + 2 * R * NP) + (size_t)8 * OWN_WAVES * NA * M * 64 + (size_t)4 * lmax + (size_t)8 * (3 * R + 1) * M * 64 +
                     (ord ? (size_t)2 * n : (R > 1 ? (size_t)R * 64 * 4 : 0)) + 15) / 16 * 16;
        if (lds_bytes <= 150 * 1024) break;
    }
    if (!ord && nsb > OWN_RING) return TSU_OK;
    if ((long long)n_sweeps * nsb >= (long long)(0xFFFFFFFFu / OWN_TAG_SPAN) - 2) return TSU_OK;  // generation tags are 32 bits
    if (lds_bytes > 150 * 1024) return TSU_OK;
    own_kern kern = d->dtype == TSU_DTYPE_F64 ? own_pick<double>(M, R, ord) : own_pick<float>(M, R, ord);
    if (!kern) return TSU_OK;
    if (tsu_func_allow_lds(ctx, (const void*)kern, (int)lds_bytes) != hipSuccess) {
        (void)hipGetLastError();
        return TSU_OK;
    }
    int per_cu = 0;
    if (tsu_func_blocks_per_cu(ctx, (const void*)kern, OWN_THREADS, lds_bytes, &per_cu) != hipSuccess || per_cu < 1) {
        (void)hipGetLastError();
        return TSU_OK;
    }
    const size_t gen_words = (size_t)OWN_NGEN * R * G * 2, fin_words = (size_t)OWN_RING * R * G * 2;
    if (d->own_cap < gen_words + fin_words) {
        if (d->own_gran) (void)hipFree(d->own_gran);
        d->own_gran = nullptr;
        d->own_cap = 0;
        TSU_HIP_TRY(ctx, hipMalloc(&d->own_gran, (gen_words + fin_words) * 8));
        d->own_cap = gen_words + fin_words;
    }
    if (!d->co_bar) TSU_HIP_TRY(ctx, hipMalloc(&d->co_bar, BAR_WORDS * sizeof(unsigned)));
    if (!d->co_fields) TSU_HIP_TRY(ctx, hipMalloc(&d->co_fields, (size_t)n * 8));
    TSU_HIP_TRY(ctx, hipMemsetAsync(d->own_gran, 0, (gen_words + fin_words) * 8, ctx->stream));
    TSU_HIP_TRY(ctx, hipMemsetAsync(d->co_bar + BAR_ERR, 0, 4 * sizeof(unsigned), ctx->stream));
    OwnParams P;
    memset(&P, 0, sizeof(P));
    P.JT = d->JT;
    P.bias = d->bias;
    P.state = states_dev;
    P.uniforms = uniforms_dev;
    P.gen = d->own_gran;
    P.fin = d->own_gran + gen_words;
    P.bar = d->co_bar;
    P.order = order_dev;
    P.temps = temps_dev;
    P.samples = samples_dev;
    P.fields_all = R_real > 1 ? d->rep_fields[d->rep_cur] : d->co_fields;
    P.fields_out = R_real > 1 ? d->rep_fields[d->rep_cur ^ 1] : d->co_fields;
    P.n = n;
    P.n_sweeps = n_sweeps;
    P.sbw = sbw;
    P.lmax = lmax;
    P.rec_from = rec_from;
    P.rec_every = rec_every > 0 ? rec_every : 1;
    P.fail_at = own_env("TSU_K2_OWN_TEST_FAIL", -1);
    // lists a deciding wave gathers by itself: up to 32 entries in natural order, 48 in a caller's order and for replicas (n = 16384:
    // natural order is flat from 16 to 32, the other two from 48 to 64); a replica's own short list has 64 slots
    P.solo_max = own_env("TSU_K2_OWN_SOLO_MAX", (ord || R > 1) ? 48 : OWN_SOLO_MAX);
    if (P.solo_max > 64) P.solo_max = 64;
    if (P.solo_max < 0) P.solo_max = 0;
    // solo generations (one chain): the deciding wave polls one group per lane in natural order, four in a caller's order.  (Replicas
    // always run their generations wave by wave: rep_step.)  Both switches are read per call: A/B measurements.
    P.solo = own_env("TSU_K2_OWN_SOLO", 1) && (ord ? NP <= 256 : NP <= 64) && (R == 1 || M == 1);
    for (int r = 0; r < R; ++r) P.rep[r] = reps[r < R_real ? r : 0];  // (padding replicas repeat replica 0 on its own state copy: see the caller)
    static const int keep_fields = own_env("TSU_K2_KEEP_FIELDS", 1);
    const bool single = R_real == 1 && allow_persist;
    P.resume = single && keep_fields && fields_were_valid && (d->since_refresh % CO_REFRESH) != 0 ? 1 : 0;
    P.refresh_off = P.resume ? d->since_refresh : 0;
    P.persist = single && keep_fields && d->pipe_streak >= 1 ? 1 : 0;
    if (R_real > 1) {  // (replicas: the caller matched the incoming states with the ones whose fields are kept: rep[].src)
        const bool rk = allow_persist && keep_fields && d->rep_fields[0] && d->rep_fields[1];
        P.resume = rk && fields_were_valid && (d->rep_since % CO_REFRESH) != 0 ? 1 : 0;
        P.refresh_off = P.resume ? d->rep_since : 0;
        P.persist = rk ? 1 : 0;
    }
    static const int verbose = own_env("TSU_K2_VERBOSE", 0);
    unsigned long long* d_tl = nullptr;
    if (verbose >= 2) {
        TSU_HIP_TRY(ctx, hipMalloc(&d_tl, 12 * sizeof(unsigned long long)));
        TSU_HIP_TRY(ctx, hipMemsetAsync(d_tl, 0, 12 * sizeof(unsigned long long), ctx->stream));
    }
    P.timeline = d_tl;
    {
        const int rcx = tsu_grid_exclusive_begin(ctx);
        if (rcx != TSU_OK) return rcx;
    }
    hipError_t e = tsu_launch_grid_sync(ctx, (const void*)kern, dim3((unsigned)W), dim3(OWN_THREADS), &P, lds_bytes, ctx->stream);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        if (d_tl) (void)hipFree(d_tl);
        return TSU_OK;
    }
    {
        const int rcx = tsu_grid_exclusive_end(ctx);
        if (rcx != TSU_OK) return rcx;
    }
    // (the error words come back into pinned memory: a copy into pageable memory is staged and costs ~10 us more per call)
    if (!d->h_flags && hipHostMalloc((void**)&d->h_flags, 4 * sizeof(unsigned), hipHostMallocDefault) != hipSuccess) {
        d->h_flags = nullptr;
        (void)hipGetLastError();
    }
    unsigned hbuf[4];
    unsigned* h = d->h_flags ? d->h_flags : hbuf;
    TSU_HIP_TRY(ctx, hipMemcpyAsync(h + 1, d->co_bar + BAR_ERR, 3 * sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
    TSU_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (d_tl) {
        unsigned long long tl[12];
        (void)hipMemcpy(tl, d_tl, sizeof(tl), hipMemcpyDeviceToHost);
        (void)hipFree(d_tl);
        const double per = (double)n_sweeps * (ord ? nsb : 1);  // the timed workgroup is active once per sweep in natural order
        fprintf(stderr, "[tsu] k2_own n=%d R=%d M=%d %s, superblocks of %d, %d sweeps; last workgroup, per active superblock: prologue %.1f us, first poll %.1f us, first axpy %.1f us, "
                        "later polls %.1f us, later axpy+decide %.1f us, commit+strip %.1f us, %.1f generations, %.0f toggles (later generations, poller wave: polling %.1f us, list building %.1f us); per sweep as a bystander: wait %.1f us, strips %.1f us\n",
                n, R, M, ord ? "caller's order" : "natural order", sbw * RW, n_sweeps, tl[0] / 100.0 / per, tl[9] / 100.0 / per, tl[8] / 100.0 / per, tl[1] / 100.0 / per, tl[2] / 100.0 / per,
                tl[3] / 100.0 / per, tl[4] / per, tl[5] / per, tl[10] / 100.0 / per, tl[11] / 100.0 / per, tl[6] / 100.0 / n_sweeps, tl[7] / 100.0 / n_sweeps);
    }
    if (h[1] || h[3]) {
        fprintf(stderr, "[tsu] dense sweep (owner kernel): %s; continuing on the other paths\n", h[1] ? "a wait timed out (GPU shared?)" : "no fixed point");
        d->own_failed = 1;
        return TSU_OK;
    }
    *done = 1;
    d->n_own += 1;
    if (R_real > 1 && P.persist) {
        d->rep_since = (P.refresh_off + n_sweeps) % CO_REFRESH;
        d->rep_cur ^= 1;
    }
    if (R_real == 1 && allow_persist) {
        d->since_refresh = (P.refresh_off + n_sweeps) % CO_REFRESH;
        d->fields_valid = P.persist;
        d->pipe_streak += 1;
    }
    return TSU_OK;
}
