// sparse.hip -- K5: heat-bath Gibbs sweeps on a sparse coupling graph, one colour class at a time (gfx950).
//
// Reference path replaced: GibbsSampler.gibbs_sweep on the dense N x N matrix of IsingChain / IsingModel
// (/root/reference/tsu/gibbs.py:79-162, tsu/models/ising.py:39-97,265-304).  The field of a site only involves its
// graph neighbours, and sites of one colour class are not coupled to each other, so a whole class is updated by one
// launch.  HBM-bound gather kernel (no MFMA): per update it reads the row extent (8 B amortised), the bias (8 B), per
// neighbour {column 4 B, coupling 8 B, neighbour bit 1 B} and writes 1 B -- algorithmic 17 + 13 deg bytes per update.
//
// Layout: everything lives in POSITION space.  Position p = rank of the site in the colour-major visiting order, so a
// launch writes a contiguous byte range (coalesced) and, for a chain, its neighbour bits are contiguous too.  The
// Philox uniform is keyed by the ORIGINAL site number (as K2 does), so results do not depend on the colouring's layout.
#include <vector>

#include "dense.h"

struct tsu_sparse {
    tsu_ctx* ctx;
    int n, n_colors;
    std::vector<int> color_off;  // host copy
    int64_t* row_ptr;   // n+1, position space
    int32_t* col;       // neighbour POSITIONS, in ascending order of the neighbours' site numbers
    double* val;
    double* bias;       // position space
    int32_t* site_of;   // position -> site
    int32_t* pos_of;    // site -> position
    int8_t* state;      // position space, {0,1}
    int8_t* staging;    // n bytes: site-order image for set/get
    int8_t* samples;    // recorded states (site order)
    size_t samples_cap;
    double* d_red;      // [energy, sum_spins as double pair] reduction target
    int64_t nnz;
    // regular colour classes (k5_stencil): see K5Stencil
    std::vector<struct K5Stencil> stencil;  // one per colour; deg < 0: the class is not regular
    unsigned long long* d_thr;              // [n_colors][K5_MAX_DEG + 1] acceptance thresholds of the current call
    uint8_t* d_code;                        // [n] (position space) decisions prepared for the second class of a PAIR, see K5Stencil::pair
};

// A REGULAR colour class (chains, rings, ladders ...: `IsingChain`, tsu/models/ising.py:265-286): apart from at most K5_EDGE rows at
// either end of its position range, every row has the same degree, the same coupling on every edge, the same bias, neighbours at
// fixed position offsets and a site number that is affine in the position.  Such a class needs no CSR streams at all: per update it
// reads its neighbours' bits (deg bytes, contiguous across the lanes) and writes one byte -- ~3 B instead of 43 for a chain -- and
// the field takes deg + 1 values only, so acceptance is one integer compare of the uniform's 53 bits with a threshold computed (on
// the device, with the generic kernel's own expressions) from the number of set neighbours.  The end rows run on the generic kernel.
#define K5_MAX_DEG 4
#define K5_EDGE 64
#define K5_THR_STRIDE (2 * (K5_MAX_DEG + 1))  // per class: the thresholds, then their leading 27 bits
struct K5Stencil {
    int deg;            // -1: not regular
    int pb, pe;         // position range of the class
    int lo, hi;         // rows [pb, pb + lo) and [pe - hi, pe) are irregular (generic kernel)
    int off[K5_MAX_DEG];
    double Jv, bias;
    int site0, site_stride;  // site of position p = site0 + site_stride * (p - pb - lo)
    // PAIRED classes.  The uniform of site i comes from the Philox block of i >> 1 (dense.h: words x, y for the even site, z, w for the
    // odd one), and in a chain the two sites of a block sit in the two colour classes at the same index: the launch of the first class
    // (pair = 1) has the second class's uniform in registers for free.  It cannot decide for that site yet -- its neighbours are being
    // updated -- but the decision is a function of the neighbour count alone: bit k of code[p'] = "the site at position p' of the other
    // class becomes 1 if k of its neighbours are set".  The second class's launch (pair = 2, k5_paired) computes no random numbers
    // at all: count, shift, store.  Philox blocks per sweep: one per PAIR of sites instead of one per site.
    int pair;                // 0: none; 1: prepares the codes of class `other`; 2: consumes them
    int other;               // the partner class
    int o_pb, o_lo, o_n;     // (pair = 1) the partner's first position, its leading irregular rows and the number of its REGULAR rows
    int o_deg;
};

namespace {

// thresholds of a regular class: thr[k] = ceil(p_k 2^53), p_k = sigmoid((k-fold sum of Jv + bias) / T) -- the field the generic kernel
// computes for a row with k set neighbours (it adds val * bit edge by edge: zeros change nothing), so that `u < p` for the 53-bit
// uniform u = m / 2^53 is `m < thr[k]`
__global__ void k5_thresholds(int deg, double Jv, double bias, double T, unsigned long long* __restrict__ thr) {
    const int k = threadIdx.x;
    if (k > K5_MAX_DEG) return;
    unsigned long long t = 0ull;
    if (k <= deg) {
        double F = 0.0;
        for (int i = 0; i < k; ++i) F += Jv * 1.0;
        F += bias;
        const double p = sigmoid_clamped(F / T);
        t = p >= 1.0 ? (1ull << 53) : (unsigned long long)ceil(ldexp(p, 53));
    }
    thr[k] = t;
    // its leading 27 bits: for the uniform's leading 27 bits a (m = a 2^26 + b), a < t >> 26 implies m < t and a > t >> 26 implies
    // m >= t; only a == t >> 26 (one draw in 2^27) needs the 64-bit compare
    thr[K5_MAX_DEG + 1 + k] = t >> 26;
}

// everything a colour-class launch is given
struct K5Args {
    int8_t* state;
    K5Stencil S;
    uint8_t* code;                        // (pairs) decisions by neighbour count, position space
    const int64_t* row_ptr;
    const int32_t* col;
    const double* val;
    const double* bias;
    const int32_t* site_of;
    double T;
    uint32_t sweep, tag, k0, k1;
    int force_tie;  // (tests) every wave of k5_stencil4 takes the exact 64-bit path
};

// the code of a partner site from its uniform's 53 bits: bit k = "becomes 1 with k neighbours set"
static __device__ __forceinline__ uint32_t k5_code(unsigned long long m2, const unsigned long long* __restrict__ thr_other, int o_deg) {
    uint32_t bits = 0u;
#pragma unroll
    for (int i = 0; i <= K5_MAX_DEG; ++i) bits |= (i <= o_deg && m2 < thr_other[i]) ? (1u << i) : 0u;
    return bits;
}

// ONE position of a class (index idx inside the class): regular rows by count and threshold, the few irregular end rows by the
// generic expression on their CSR rows; PAIR: also the code of the other site of the Philox block
template <bool PAIR>
static __device__ __forceinline__ void k5_site(const K5Args& A, const unsigned long long* __restrict__ thr, const unsigned long long* __restrict__ thr_other,
                                               int idx) {
    const K5Stencil& S = A.S;
    const int p = S.pb + idx;
    const bool regular = p >= S.pb + S.lo && p < S.pe - S.hi;
    // (a paired class: the site number is affine over the WHOLE class, checked at creation)
    const uint32_t site = (regular || PAIR) ? (uint32_t)(S.site0 + S.site_stride * (p - S.pb - S.lo)) : (uint32_t)A.site_of[p];
    const u32x4 w = tsu_philox(site >> 1, 0u, A.sweep, A.tag, A.k0, A.k1);  // dense_uniform's block and words (dense.h)
    const uint32_t a = ((site & 1) ? w.z : w.x) >> 5, b = ((site & 1) ? w.w : w.y) >> 6;
    if (PAIR) {
        const int j = idx - S.o_lo;  // the other site of the block: position idx of the partner class
        if (j >= 0 && j < S.o_n) {
            const uint32_t a2 = ((site & 1) ? w.x : w.z) >> 5, b2 = ((site & 1) ? w.y : w.w) >> 6;
            A.code[S.o_pb + idx] = (uint8_t)k5_code(((unsigned long long)a2 << 26) | b2, thr_other, S.o_deg);
        }
    }
    if (!regular) {
        double F = 0.0;
        for (int64_t e = A.row_ptr[p]; e < A.row_ptr[p + 1]; ++e) F += A.val[e] * (double)A.state[A.col[e]];
        F += A.bias[p];
        const double u = ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;  // (dense_uniform)
        A.state[p] = (u < sigmoid_clamped(F / A.T)) ? 1 : 0;
        return;
    }
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < K5_MAX_DEG; ++i)
        if (i < S.deg) cnt += A.state[p + S.off[i]];
    const unsigned long long m = ((unsigned long long)a << 26) | b;
    // (the five thresholds sit at uniform addresses: scalar loads issued at the top, selected by the count -- no dependent vector load)
    unsigned long long t = thr[0];
#pragma unroll
    for (int i = 1; i <= K5_MAX_DEG; ++i) t = cnt == i ? thr[i] : t;
    A.state[p] = m < t ? 1 : 0;
}

// ONE position of the second class of a pair: its regular rows take the decision the first class's launch prepared (bit `count` of
// the code); the few irregular end rows run the generic expression
static __device__ __forceinline__ void k5_site_paired(const K5Args& A, int idx) {
    const K5Stencil& S = A.S;
    const int p = S.pb + idx;
    if (p < S.pb + S.lo || p >= S.pe - S.hi) {
        double F = 0.0;
        for (int64_t e = A.row_ptr[p]; e < A.row_ptr[p + 1]; ++e) F += A.val[e] * (double)A.state[A.col[e]];
        F += A.bias[p];
        const double u = dense_uniform((uint32_t)A.site_of[p], A.sweep, A.tag, A.k0, A.k1);
        A.state[p] = (u < sigmoid_clamped(F / A.T)) ? 1 : 0;
        return;
    }
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < K5_MAX_DEG; ++i)
        if (i < S.deg) cnt += A.state[p + S.off[i]];
    A.state[p] = (int8_t)((A.code[p] >> cnt) & 1u);
}

static __device__ __forceinline__ uint32_t k5_ld4(const void* q) {  // four bytes at any alignment (one dword load: unaligned access mode)
    uint32_t v;
    __builtin_memcpy(&v, q, 4);
    return v;
}

// One launch per colour class, FOUR consecutive positions per thread: the neighbours' bits arrive as (unaligned) dwords and add up to
// four byte-wide counts at once, the new bits leave as one dword.  (One position per thread: 3 byte-wide memory instructions per
// update -- the launch was bound by their issue, not by Philox: 24 us per class of 2^23 sites with or without random numbers.)
// MODE 0: a class of its own; 1: the first class of a pair (codes for the second); 2: the second class of a pair (no random numbers).
// The host launches these only for classes whose first position (and, MODE 1, the partner's) is a multiple of 4.  (The thresholds
// come as __restrict__ kernel arguments: uniform, read-only -- scalar loads at the top; through the struct they became dependent
// vector loads inside the compare chain, +12 us per launch.)
template <int MODE>
__global__ __launch_bounds__(256) void k5_stencil4(K5Args A, const unsigned long long* __restrict__ thr, const unsigned long long* __restrict__ thr_other) {
    const K5Stencil& S = A.S;
    const int idx0 = 4 * (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const int len = S.pe - S.pb;
    if (idx0 >= len) return;
    const bool own4 = idx0 >= S.lo && idx0 + 4 <= len - S.hi;
    const bool code4 = MODE != 1 || (idx0 >= S.o_lo && idx0 + 4 <= S.o_lo + S.o_n);
    if (!own4 || !code4) {  // (the end rows' threads: position by position)
        for (int v = 0; v < 4 && idx0 + v < len; ++v) {
            if (MODE == 2) k5_site_paired(A, idx0 + v);
            else if (MODE == 1) k5_site<true>(A, thr, thr_other, idx0 + v);
            else k5_site<false>(A, thr, thr_other, idx0 + v);
        }
        return;
    }
    uint32_t tha[K5_MAX_DEG + 1], tho[K5_MAX_DEG + 1];  // the thresholds' leading 27 bits
#pragma unroll
    for (int i = 0; i <= K5_MAX_DEG; ++i) {
        tha[i] = (uint32_t)thr[K5_MAX_DEG + 1 + i];
        tho[i] = MODE == 1 ? (uint32_t)thr_other[K5_MAX_DEG + 1 + i] : 0u;
    }
    const int p0 = S.pb + idx0;
    uint32_t cnt4 = 0u;  // four counts, one per byte (<= K5_MAX_DEG each)
#pragma unroll
    for (int i = 0; i < K5_MAX_DEG; ++i)
        if (i < S.deg) cnt4 += k5_ld4(A.state + p0 + S.off[i]);
    uint32_t out = 0u;
    if (MODE == 2) {
        const uint32_t c4 = *reinterpret_cast<const uint32_t*>(A.code + p0);
#pragma unroll
        for (int v = 0; v < 4; ++v) out |= (((c4 >> (8 * v)) & 0xFFu) >> ((cnt4 >> (8 * v)) & 0xFFu) & 1u) << (8 * v);
    } else {
        uint32_t codes = 0u;
        // (compares on the uniforms' leading 27 bits against the thresholds' -- 32-bit operations on scalar operands; the 64-bit
        // compares of k5_site only for a wave in which some lane's leading bits EQUAL a threshold's: one draw in 2^27)
        bool tie = A.force_tie != 0;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const uint32_t site = (uint32_t)(S.site0 + S.site_stride * (idx0 + v - S.lo));
            const u32x4 w = tsu_philox(site >> 1, 0u, A.sweep, A.tag, A.k0, A.k1);
            const uint32_t a = ((site & 1) ? w.z : w.x) >> 5;
            const uint32_t cnt = (cnt4 >> (8 * v)) & 0xFFu;
            uint32_t ta = tha[0];
#pragma unroll
            for (int i = 1; i <= K5_MAX_DEG; ++i)
                if (i <= S.deg) ta = cnt == (uint32_t)i ? tha[i] : ta;
            out |= (a < ta ? 1u : 0u) << (8 * v);
            tie = tie || a == ta;
            if (MODE == 1) {
                const uint32_t a2 = ((site & 1) ? w.x : w.z) >> 5;
#pragma unroll
                for (int i = 0; i <= K5_MAX_DEG; ++i)
                    if (i <= S.o_deg) {
                        const uint32_t to = tho[i];
                        codes |= (a2 < to ? 1u : 0u) << (8 * v + i);
                        tie = tie || a2 == to;
                    }
            }
        }
        if (__builtin_expect(__any((int)tie), 0)) {
            for (int v = 0; v < 4; ++v) {
                if (MODE == 1) k5_site<true>(A, thr, thr_other, idx0 + v);
                else k5_site<false>(A, thr, thr_other, idx0 + v);
            }
            return;
        }
        if (MODE == 1) *reinterpret_cast<uint32_t*>(A.code + S.o_pb + idx0) = codes;
    }
    *reinterpret_cast<uint32_t*>(A.state + p0) = out;
}

// the same, one position per thread (classes that do not start at a multiple of 4)
template <int MODE>
__global__ __launch_bounds__(256) void k5_stencil1(K5Args A, const unsigned long long* __restrict__ thr, const unsigned long long* __restrict__ thr_other) {
    const int idx = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (idx >= A.S.pe - A.S.pb) return;
    if (MODE == 2) k5_site_paired(A, idx);
    else if (MODE == 1) k5_site<true>(A, thr, thr_other, idx);
    else k5_site<false>(A, thr, thr_other, idx);
}

// the generic update for an explicit list of rows [p_begin, p_end) -- the irregular end rows of a regular class
__global__ __launch_bounds__(256) void k5_color(const int64_t* __restrict__ row_ptr, const int32_t* __restrict__ col,
                                                const double* __restrict__ val, const double* __restrict__ bias,
                                                const int32_t* __restrict__ site_of, int8_t* __restrict__ state, int p_begin,
                                                int p_end, double T, uint32_t sweep, uint32_t tag, uint32_t k0, uint32_t k1) {
    const int p = p_begin + (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (p >= p_end) return;
    const int64_t e0 = row_ptr[p], e1 = row_ptr[p + 1];
    double F = 0.0;
    for (int64_t e = e0; e < e1; ++e) F += val[e] * (double)state[col[e]];
    F += bias[p];
    const double u = dense_uniform((uint32_t)site_of[p], sweep, tag, k0, k1);
    state[p] = (u < sigmoid_clamped(F / T)) ? 1 : 0;
}

// Small graphs (n <= K5S_MAX): the whole run in ONE launch of one workgroup, the state in LDS, colour after colour with
// a workgroup barrier in between; optional recording of the state after every `rec_every` sweeps past `rec_from`.
constexpr int K5S_MAX = 32768;
constexpr int K5S_THREADS = 1024;
__global__ __launch_bounds__(K5S_THREADS) void k5_small(const int64_t* __restrict__ row_ptr, const int32_t* __restrict__ col,
                                                        const double* __restrict__ val, const double* __restrict__ bias,
                                                        const int32_t* __restrict__ site_of, int8_t* __restrict__ state, int n,
                                                        const int* __restrict__ color_off, int n_colors, double T, int n_sweeps,
                                                        uint32_t sweep0, uint32_t tag, uint32_t k0, uint32_t k1,
                                                        int8_t* __restrict__ samples, int rec_from, int rec_every) {
    extern __shared__ int8_t s_state[];
    for (int p = threadIdx.x; p < n; p += K5S_THREADS) s_state[p] = state[p];
    __syncthreads();
    for (int s = 0; s < n_sweeps; ++s) {
        for (int c = 0; c < n_colors; ++c) {
            const int pb = color_off[c], pe = color_off[c + 1];
            for (int p = pb + (int)threadIdx.x; p < pe; p += K5S_THREADS) {
                const int64_t e0 = row_ptr[p], e1 = row_ptr[p + 1];
                double F = 0.0;
                for (int64_t e = e0; e < e1; ++e) F += val[e] * (double)s_state[col[e]];
                F += bias[p];
                const double u = dense_uniform((uint32_t)site_of[p], sweep0 + (uint32_t)s, tag, k0, k1);
                s_state[p] = (u < sigmoid_clamped(F / T)) ? 1 : 0;
            }
            __syncthreads();
        }
        if (samples && s + 1 > rec_from && (s + 1 - rec_from) % rec_every == 0) {
            int8_t* dst = samples + (size_t)((s + 1 - rec_from) / rec_every - 1) * n;
            for (int p = threadIdx.x; p < n; p += K5S_THREADS) dst[site_of[p]] = s_state[p];
            __syncthreads();  // the next sweep rewrites s_state with another thread-to-position mapping
        }
    }
    for (int p = threadIdx.x; p < n; p += K5S_THREADS) state[p] = s_state[p];
}

__global__ void k5_scatter(const int8_t* __restrict__ src_site, const int32_t* __restrict__ site_of, int8_t* __restrict__ dst_pos, int n) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) dst_pos[p] = src_site[site_of[p]];
}

__global__ void k5_gather(const int8_t* __restrict__ src_pos, const int32_t* __restrict__ site_of, int8_t* __restrict__ dst_site, int n) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) dst_site[site_of[p]] = src_pos[p];
}

// energy = -1/2 sum_p b_p (sum_e val_e b_col) - sum_p bias_p b_p ; sum of spins = sum_p (2 b_p - 1)
__global__ __launch_bounds__(256) void k5_energy(const int64_t* __restrict__ row_ptr, const int32_t* __restrict__ col,
                                                 const double* __restrict__ val, const double* __restrict__ bias,
                                                 const int8_t* __restrict__ state, int n, double* __restrict__ out) {
    double e = 0.0, m = 0.0;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {
        const double b = (double)state[p];
        double F = 0.0;
        for (int64_t q = row_ptr[p]; q < row_ptr[p + 1]; ++q) F += val[q] * (double)state[col[q]];
        e += -0.5 * b * F - bias[p] * b;
        m += 2.0 * b - 1.0;
    }
    for (int off = 32; off > 0; off >>= 1) {
        e += __shfl_down(e, off, 64);
        m += __shfl_down(m, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(out, e);
        atomicAdd(out + 1, m);
    }
}

int run_sweeps(tsu_sparse* g, double T, int n_sweeps, uint64_t seed, uint32_t sweep0, uint32_t replica, int8_t* samples, int rec_from,
               int rec_every) {
    tsu_ctx* ctx = g->ctx;
    const uint32_t tag = TSU_TAG_DENSE | (replica << 8), k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    if (n_sweeps <= 0) return TSU_OK;
    if (g->n <= K5S_MAX) {
        int* d_off = (int*)(g->site_of + g->n);  // colour offsets stored behind the site table (see create)
        TSU_HIP_TRY(ctx, tsu_func_allow_lds(ctx, (const void*)k5_small, K5S_MAX));
        hipLaunchKernelGGL(k5_small, dim3(1), dim3(K5S_THREADS), (size_t)((g->n + 15) / 16 * 16), ctx->stream, g->row_ptr, g->col, g->val, g->bias,
                           g->site_of, g->state, g->n, d_off, g->n_colors, T, n_sweeps, sweep0, tag, k0, k1, samples, rec_from, rec_every);
        TSU_HIP_TRY(ctx, hipGetLastError());
        return TSU_OK;
    }
    for (int c = 0; c < g->n_colors; ++c) {  // thresholds of the regular classes at this call's temperature
        const K5Stencil& S = g->stencil[(size_t)c];
        if (S.deg > 0) hipLaunchKernelGGL(k5_thresholds, dim3(1), dim3(64), 0, ctx->stream, S.deg, S.Jv, S.bias, T, g->d_thr + (size_t)c * K5_THR_STRIDE);
    }
    for (int s = 0; s < n_sweeps; ++s) {
        for (int c = 0; c < g->n_colors; ++c) {
            const int pb = g->color_off[c], pe = g->color_off[c + 1];
            if (pe <= pb) continue;
            const K5Stencil& S = g->stencil[(size_t)c];
            if (S.deg > 0) {
                K5Args A;
                A.state = g->state;
                A.S = S;
                const unsigned long long* thr = g->d_thr + (size_t)c * K5_THR_STRIDE;
                const unsigned long long* thr_other = S.pair == 1 ? g->d_thr + (size_t)S.other * K5_THR_STRIDE : thr;
                A.code = g->d_code;
                A.row_ptr = g->row_ptr;
                A.col = g->col;
                A.val = g->val;
                A.bias = g->bias;
                A.site_of = g->site_of;
                A.T = T;
                A.sweep = sweep0 + (uint32_t)s;
                A.tag = tag;
                A.k0 = k0;
                A.k1 = k1;
                A.force_tie = getenv("TSU_K5_TEST_TIE") && atoi(getenv("TSU_K5_TEST_TIE")) != 0;
                const bool use_v4 = !(getenv("TSU_K5_V4") && atoi(getenv("TSU_K5_V4")) == 0);
                const bool v4 = use_v4 && pb % 4 == 0 && (S.pair != 1 || S.o_pb % 4 == 0);
                const dim3 grid((unsigned)(((pe - pb + (v4 ? 3 : 0)) / (v4 ? 4 : 1) + 255) / 256));
                if (v4) {
                    if (S.pair == 2) hipLaunchKernelGGL(k5_stencil4<2>, grid, dim3(256), 0, ctx->stream, A, thr, thr_other);
                    else if (S.pair == 1) hipLaunchKernelGGL(k5_stencil4<1>, grid, dim3(256), 0, ctx->stream, A, thr, thr_other);
                    else hipLaunchKernelGGL(k5_stencil4<0>, grid, dim3(256), 0, ctx->stream, A, thr, thr_other);
                } else {
                    if (S.pair == 2) hipLaunchKernelGGL(k5_stencil1<2>, grid, dim3(256), 0, ctx->stream, A, thr, thr_other);
                    else if (S.pair == 1) hipLaunchKernelGGL(k5_stencil1<1>, grid, dim3(256), 0, ctx->stream, A, thr, thr_other);
                    else hipLaunchKernelGGL(k5_stencil1<0>, grid, dim3(256), 0, ctx->stream, A, thr, thr_other);
                }
                continue;
            }
            hipLaunchKernelGGL(k5_color, dim3((unsigned)((pe - pb + 255) / 256)), dim3(256), 0, ctx->stream, g->row_ptr, g->col, g->val, g->bias,
                               g->site_of, g->state, pb, pe, T, sweep0 + (uint32_t)s, tag, k0, k1);
        }
        if (samples && s + 1 > rec_from && (s + 1 - rec_from) % rec_every == 0) {
            int8_t* dst = samples + (size_t)((s + 1 - rec_from) / rec_every - 1) * g->n;
            hipLaunchKernelGGL(k5_gather, dim3((unsigned)((g->n + 255) / 256)), dim3(256), 0, ctx->stream, g->state, g->site_of, dst, g->n);
        }
    }
    TSU_HIP_TRY(ctx, hipGetLastError());
    return TSU_OK;
}

void free_all(tsu_sparse* g) {
    void* ptrs[] = {g->row_ptr, g->col, g->val, g->bias, g->site_of, g->pos_of, g->state, g->staging, g->samples, g->d_red, g->d_thr, g->d_code};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    delete g;
}
}  // namespace

extern "C" {

int tsu_sparse_create(tsu_ctx* ctx, int n, const int64_t* row_ptr, const int32_t* col_idx, const double* values, const double* bias_host,
                      int n_colors, const int32_t* color_offsets, const int32_t* order, tsu_sparse** out) {
    TSU_ENTER(ctx);
    if (!ctx) return TSU_E_INVALID;
    TSU_REQUIRE(ctx, out && n > 0 && row_ptr && n_colors > 0 && color_offsets && order, "tsu_sparse_create: bad arguments");
    *out = nullptr;
    const int64_t nnz = row_ptr[n];
    TSU_REQUIRE(ctx, row_ptr[0] == 0 && nnz >= 0 && (nnz == 0 || (col_idx && values)), "tsu_sparse_create: bad CSR arrays");
    TSU_REQUIRE(ctx, color_offsets[0] == 0 && color_offsets[n_colors] == n, "tsu_sparse_create: colour offsets must run from 0 to n");
    // validate: order is a permutation, rows ascending and in range, the colouring is proper
    std::vector<int32_t> pos_of((size_t)n, -1), color_of((size_t)n, -1);
    for (int c = 0; c < n_colors; ++c) {
        TSU_REQUIRE(ctx, color_offsets[c] <= color_offsets[c + 1], "tsu_sparse_create: colour offsets must not decrease");
        for (int p = color_offsets[c]; p < color_offsets[c + 1]; ++p) {
            const int32_t i = order[p];
            TSU_REQUIRE(ctx, i >= 0 && i < n && pos_of[(size_t)i] < 0, "tsu_sparse_create: order is not a permutation of 0..n-1");
            pos_of[(size_t)i] = p;
            color_of[(size_t)i] = c;
        }
    }
    for (int i = 0; i < n; ++i) {
        TSU_REQUIRE(ctx, row_ptr[i] <= row_ptr[i + 1], "tsu_sparse_create: row_ptr must not decrease");
        for (int64_t e = row_ptr[i]; e < row_ptr[i + 1]; ++e) {
            const int32_t j = col_idx[e];
            TSU_REQUIRE(ctx, j >= 0 && j < n, "tsu_sparse_create: column index out of range");
            TSU_REQUIRE(ctx, e == row_ptr[i] || col_idx[e - 1] < j, "tsu_sparse_create: columns of a row must ascend");
            TSU_REQUIRE(ctx, j == i || color_of[(size_t)j] != color_of[(size_t)i],
                        "tsu_sparse_create: sites %d and %d are coupled but have the same colour", i, (int)j);
        }
    }
    // position-space CSR
    std::vector<int64_t> rp((size_t)n + 1);
    std::vector<int32_t> cp((size_t)nnz);
    std::vector<double> vp((size_t)nnz), bp((size_t)n, 0.0);
    rp[0] = 0;
    for (int p = 0; p < n; ++p) {
        const int i = order[p];
        int64_t w = rp[(size_t)p];
        for (int64_t e = row_ptr[i]; e < row_ptr[i + 1]; ++e, ++w) {
            cp[(size_t)w] = pos_of[(size_t)col_idx[e]];
            vp[(size_t)w] = values[e];
        }
        rp[(size_t)p + 1] = w;
        if (bias_host) bp[(size_t)p] = bias_host[i];
    }
    tsu_sparse* g = new (std::nothrow) tsu_sparse();
    if (!g) return tsu_fail(ctx, TSU_E_NOMEM, "tsu_sparse_create: host allocation failed");
    g->ctx = ctx;
    g->n = n;
    g->n_colors = n_colors;
    g->nnz = nnz;
    g->d_thr = nullptr;
    g->d_code = nullptr;
    // regular colour classes: the pattern of the class's middle row must hold for every row but at most K5_EDGE at either end
    const bool use_stencil = !(getenv("TSU_K5_STENCIL") && atoi(getenv("TSU_K5_STENCIL")) == 0);
    g->stencil.assign((size_t)n_colors, K5Stencil());
    for (int c = 0; c < n_colors; ++c) {
        K5Stencil& S = g->stencil[(size_t)c];
        S.deg = -1;
        const int pb = color_offsets[c], pe = color_offsets[c + 1];
        if (!use_stencil || pe - pb < 4 * K5_EDGE + 2) continue;
        const int pm = pb + (pe - pb) / 2;
        const int deg = (int)(rp[(size_t)pm + 1] - rp[(size_t)pm]);
        if (deg < 1 || deg > K5_MAX_DEG) continue;
        K5Stencil T;
        T.deg = deg;
        T.pb = pb;
        T.pe = pe;
        T.Jv = vp[(size_t)rp[(size_t)pm]];
        T.bias = bp[(size_t)pm];
        for (int i = 0; i < K5_MAX_DEG; ++i) T.off[i] = i < deg ? cp[(size_t)rp[(size_t)pm] + i] - pm : 0;
        T.site_stride = order[pm + 1] - order[pm];
        auto fits = [&](int p) {
            if (rp[(size_t)p + 1] - rp[(size_t)p] != deg || bp[(size_t)p] != T.bias) return false;
            if ((long long)order[p] != (long long)order[pm] + (long long)T.site_stride * (p - pm)) return false;
            for (int i = 0; i < deg; ++i) {
                const int64_t e = rp[(size_t)p] + i;
                if (vp[(size_t)e] != T.Jv || cp[(size_t)e] - p != T.off[i] || cp[(size_t)e] == p) return false;  // (no self-loops: the row's own bit is rewritten)
            }
            return true;
        };
        int lo = 0, hi = 0;
        while (lo <= K5_EDGE && !fits(pb + lo)) ++lo;
        while (hi <= K5_EDGE && !fits(pe - 1 - hi)) ++hi;
        if (lo > K5_EDGE || hi > K5_EDGE) continue;
        bool ok = true;
        for (int p = pb + lo; p < pe - hi && ok; ++p) ok = fits(p);
        if (!ok) continue;
        T.lo = lo;
        T.hi = hi;
        T.site0 = order[pb + lo];
        T.pair = 0;
        T.other = -1;
        T.o_pb = T.o_lo = T.o_n = T.o_deg = 0;
        S = T;
    }
    // pairs of regular classes that share their Philox blocks index by index (K5Stencil::pair): sites ascending by 2 over the WHOLE
    // class (end rows included), the first sites of the two classes are the two sites of one block, and every regular row of the
    // second class has its partner in the first
    const bool use_pairs = !(getenv("TSU_K5_PAIR") && atoi(getenv("TSU_K5_PAIR")) == 0);
    bool any_pair = false;
    for (int c = 0; c < n_colors && use_pairs; ++c) {
        K5Stencil& A = g->stencil[(size_t)c];
        if (A.deg <= 0 || A.pair || A.site_stride != 2) continue;
        for (int c2 = c + 1; c2 < n_colors; ++c2) {
            K5Stencil& B = g->stencil[(size_t)c2];
            if (B.deg <= 0 || B.pair || B.site_stride != 2) continue;
            if ((order[A.pb] ^ 1) != order[B.pb]) continue;
            bool ok = true;
            for (int q = A.pb; q < A.pe && ok; ++q) ok = order[q] == order[A.pb] + 2 * (q - A.pb);
            for (int q = B.pb; q < B.pe && ok; ++q) ok = order[q] == order[B.pb] + 2 * (q - B.pb);
            if (!ok || (B.pe - B.hi) - B.pb > A.pe - A.pb) continue;  // (a regular row of B beyond A's last index would have no code)
            A.pair = 1;
            A.other = c2;
            A.o_pb = B.pb;
            A.o_lo = B.lo;
            A.o_n = (B.pe - B.hi) - (B.pb + B.lo);
            A.o_deg = B.deg;
            B.pair = 2;
            B.other = c;
            any_pair = true;
            break;
        }
    }
    g->color_off.assign(color_offsets, color_offsets + n_colors + 1);
    hipError_t e = hipSuccess;
    auto up = [&](void** dst, const void* src, size_t bytes) {
        if (e != hipSuccess) return;
        e = hipMalloc(dst, bytes ? bytes : 8);
        if (e == hipSuccess && bytes) e = hipMemcpyAsync(*dst, src, bytes, hipMemcpyHostToDevice, ctx->stream);
    };
    up((void**)&g->row_ptr, rp.data(), ((size_t)n + 1) * 8);
    up((void**)&g->col, cp.data(), (size_t)nnz * 4);
    up((void**)&g->val, vp.data(), (size_t)nnz * 8);
    up((void**)&g->bias, bp.data(), (size_t)n * 8);
    // site table followed by the colour offsets (read by the one-workgroup kernel)
    std::vector<int32_t> tab((size_t)n + (size_t)n_colors + 1);
    for (int p = 0; p < n; ++p) tab[(size_t)p] = order[p];
    for (int c = 0; c <= n_colors; ++c) tab[(size_t)n + (size_t)c] = color_offsets[c];
    up((void**)&g->site_of, tab.data(), tab.size() * 4);
    up((void**)&g->pos_of, pos_of.data(), (size_t)n * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&g->state, (size_t)n);
    if (e == hipSuccess) e = hipMemsetAsync(g->state, 0, (size_t)n, ctx->stream);
    if (e == hipSuccess) e = hipMalloc((void**)&g->staging, (size_t)n);
    if (e == hipSuccess) e = hipMalloc((void**)&g->d_red, 2 * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&g->d_thr, (size_t)n_colors * K5_THR_STRIDE * sizeof(unsigned long long));
    if (e == hipSuccess && any_pair) e = hipMalloc((void**)&g->d_code, (size_t)n);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);  // the host vectors go out of scope
    if (e != hipSuccess) {
        free_all(g);
        return tsu_fail(ctx, e == hipErrorOutOfMemory ? TSU_E_NOMEM : TSU_E_HIP, "tsu_sparse_create: %s", hipGetErrorString(e));
    }
    *out = g;
    return TSU_OK;
}

int tsu_sparse_destroy(tsu_sparse* g) {
    TSU_ENTER(g ? g->ctx : nullptr);
    if (!g) return TSU_OK;
    (void)hipStreamSynchronize(g->ctx->stream);
    free_all(g);
    return TSU_OK;
}

int tsu_sparse_set_state(tsu_sparse* g, const int8_t* bits_host) {
    TSU_ENTER(g ? g->ctx : nullptr);
    if (!g) return TSU_E_INVALID;
    tsu_ctx* ctx = g->ctx;
    TSU_REQUIRE(ctx, bits_host, "tsu_sparse_set_state: NULL buffer");
    for (int i = 0; i < g->n; ++i) TSU_REQUIRE(ctx, bits_host[i] == 0 || bits_host[i] == 1, "sparse_set_state: state must be 0/1");
    TSU_HIP_TRY(ctx, hipMemcpyAsync(g->staging, bits_host, (size_t)g->n, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k5_scatter, dim3((unsigned)((g->n + 255) / 256)), dim3(256), 0, ctx->stream, g->staging, g->site_of, g->state, g->n);
    TSU_HIP_TRY(ctx, hipGetLastError());
    TSU_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return TSU_OK;
}

int tsu_sparse_get_state(tsu_sparse* g, int8_t* bits_host) {
    TSU_ENTER(g ? g->ctx : nullptr);
    if (!g) return TSU_E_INVALID;
    tsu_ctx* ctx = g->ctx;
    TSU_REQUIRE(ctx, bits_host, "tsu_sparse_get_state: NULL buffer");
    hipLaunchKernelGGL(k5_gather, dim3((unsigned)((g->n + 255) / 256)), dim3(256), 0, ctx->stream, g->state, g->site_of, g->staging, g->n);
    TSU_HIP_TRY(ctx, hipGetLastError());
    TSU_HIP_TRY(ctx, hipMemcpyAsync(bits_host, g->staging, (size_t)g->n, hipMemcpyDeviceToHost, ctx->stream));
    TSU_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return TSU_OK;
}

int tsu_sparse_sweep(tsu_sparse* g, double T, int n_sweeps, uint64_t seed, uint32_t sweep0, uint32_t replica) {
    TSU_ENTER(g ? g->ctx : nullptr);
    if (!g) return TSU_E_INVALID;
    TSU_REQUIRE(g->ctx, T > 0.0, "Temperature must be positive");
    TSU_REQUIRE(g->ctx, n_sweeps >= 0, "tsu_sparse_sweep: n_sweeps must be >= 0");
    return run_sweeps(g, T, n_sweeps, seed, sweep0, replica, nullptr, 0, 1);
}

int tsu_sparse_sample(tsu_sparse* g, double T, int n_burnin, int n_sweeps, int n_samples, uint64_t seed, uint32_t sweep0, uint32_t replica,
                      int8_t* samples_host) {
    TSU_ENTER(g ? g->ctx : nullptr);
    if (!g) return TSU_E_INVALID;
    tsu_ctx* ctx = g->ctx;
    TSU_REQUIRE(ctx, T > 0.0, "Temperature must be positive");
    TSU_REQUIRE(ctx, n_burnin >= 0 && n_sweeps > 0 && n_samples >= 0 && (samples_host || n_samples == 0), "tsu_sparse_sample: bad arguments");
    const size_t need = (size_t)n_samples * (size_t)g->n;
    if (need > g->samples_cap) {
        if (g->samples) (void)hipFree(g->samples);
        g->samples = nullptr;
        g->samples_cap = 0;
        TSU_HIP_TRY(ctx, hipMalloc((void**)&g->samples, need));
        g->samples_cap = need;
    }
    const long long total = (long long)n_burnin + (long long)n_samples * n_sweeps;
    TSU_REQUIRE(ctx, total < (1ll << 31), "tsu_sparse_sample: too many sweeps in one call");
    int rc = run_sweeps(g, T, (int)total, seed, sweep0, replica, n_samples ? g->samples : nullptr, n_burnin, n_sweeps);
    if (rc != TSU_OK) return rc;
    if (n_samples) TSU_HIP_TRY(ctx, hipMemcpyAsync(samples_host, g->samples, need, hipMemcpyDeviceToHost, ctx->stream));
    TSU_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return TSU_OK;
}

int tsu_sparse_energy(tsu_sparse* g, double* energy, int64_t* sum_spins) {
    TSU_ENTER(g ? g->ctx : nullptr);
    if (!g) return TSU_E_INVALID;
    tsu_ctx* ctx = g->ctx;
    TSU_REQUIRE(ctx, energy, "tsu_sparse_energy: NULL output");
    TSU_HIP_TRY(ctx, hipMemsetAsync(g->d_red, 0, 2 * sizeof(double), ctx->stream));
    int blocks = (g->n + 255) / 256;
    if (blocks > 4 * ctx->cus) blocks = 4 * ctx->cus;
    hipLaunchKernelGGL(k5_energy, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, g->row_ptr, g->col, g->val, g->bias, g->state, g->n, g->d_red);
    TSU_HIP_TRY(ctx, hipGetLastError());
    double h[2];
    TSU_HIP_TRY(ctx, hipMemcpyAsync(h, g->d_red, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    TSU_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *energy = h[0];
    if (sum_spins) *sum_spins = (int64_t)llround(h[1]);
    return TSU_OK;
}

}  // extern "C"
