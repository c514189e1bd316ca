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
};

// A REGULAR colour class (chains, rings, ladders ...: `IsingChain`, tsu/models/ising.py:265-286): apart from at most K5_EDGE rows at
// either end of its position range, every row has the same degree, the same coupling on every edge, the same bias, neighbours at
// fixed position offsets and a site number that is affine in the position.  Such a class needs no CSR streams at all: per update it
// reads its neighbours' bits (deg bytes, contiguous across the lanes) and writes one byte -- ~3 B instead of 43 for a chain -- and
// the field takes deg + 1 values only, so acceptance is one integer compare of the uniform's 53 bits with a threshold computed (on
// the device, with the generic kernel's own expressions) from the number of set neighbours.  The end rows run on the generic kernel.
#define K5_MAX_DEG 4
#define K5_EDGE 64
struct K5Stencil {
    int deg;            // -1: not regular
    int pb, pe;         // position range of the class
    int lo, hi;         // rows [pb, pb + lo) and [pe - hi, pe) are irregular (generic kernel)
    int off[K5_MAX_DEG];
    double Jv, bias;
    int site0, site_stride;  // site of position p = site0 + site_stride * (p - pb - lo)
};

namespace {

// thresholds of a regular class: thr[k] = ceil(p_k 2^53), p_k = sigmoid((k-fold sum of Jv + bias) / T) -- the field the generic kernel
// computes for a row with k set neighbours (it adds val * bit edge by edge: zeros change nothing), so that `u < p` for the 53-bit
// uniform u = m / 2^53 is `m < thr[k]`
__global__ void k5_thresholds(int deg, double Jv, double bias, double T, unsigned long long* __restrict__ thr) {
    const int k = threadIdx.x;
    if (k > K5_MAX_DEG) return;
    if (k > deg) {
        thr[k] = 0ull;
        return;
    }
    double F = 0.0;
    for (int i = 0; i < k; ++i) F += Jv * 1.0;
    F += bias;
    const double p = sigmoid_clamped(F / T);
    thr[k] = p >= 1.0 ? (1ull << 53) : (unsigned long long)ceil(ldexp(p, 53));
}

// one launch per colour class: the regular rows by count and threshold, the few irregular end rows (first and last waves only) by the
// generic expression on their CSR rows
__global__ __launch_bounds__(256) void k5_stencil(int8_t* __restrict__ state, K5Stencil S, const unsigned long long* __restrict__ thr,
                                                  const int64_t* __restrict__ row_ptr, const int32_t* __restrict__ col, const double* __restrict__ val,
                                                  const double* __restrict__ bias, const int32_t* __restrict__ site_of, double T, uint32_t sweep,
                                                  uint32_t tag, uint32_t k0, uint32_t k1) {
    const int p = S.pb + (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (p >= S.pe) return;
    if (p < S.pb + S.lo || p >= S.pe - S.hi) {
        double F = 0.0;
        for (int64_t e = row_ptr[p]; e < row_ptr[p + 1]; ++e) F += val[e] * (double)state[col[e]];
        F += bias[p];
        const double u = dense_uniform((uint32_t)site_of[p], sweep, tag, k0, k1);
        state[p] = (u < sigmoid_clamped(F / T)) ? 1 : 0;
        return;
    }
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < K5_MAX_DEG; ++i)
        if (i < S.deg) cnt += state[p + S.off[i]];
    const uint32_t site = (uint32_t)(S.site0 + S.site_stride * (p - S.pb - S.lo));
    const u32x4 w = tsu_philox(site >> 1, 0u, sweep, tag, k0, k1);  // dense_uniform's block and words (dense.h)
    const uint32_t a = ((site & 1) ? w.z : w.x) >> 5, b = ((site & 1) ? w.w : w.y) >> 6;
    const unsigned long long m = ((unsigned long long)a << 26) | b;
    // (the five thresholds sit at uniform addresses: scalar loads issued at the top, selected by the count -- no dependent vector load)
    unsigned long long t = thr[0];
#pragma unroll
    for (int i = 1; i <= K5_MAX_DEG; ++i) t = cnt == i ? thr[i] : t;
    state[p] = m < t ? 1 : 0;
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
        if (S.deg > 0) hipLaunchKernelGGL(k5_thresholds, dim3(1), dim3(64), 0, ctx->stream, S.deg, S.Jv, S.bias, T, g->d_thr + (size_t)c * (K5_MAX_DEG + 1));
    }
    for (int s = 0; s < n_sweeps; ++s) {
        for (int c = 0; c < g->n_colors; ++c) {
            const int pb = g->color_off[c], pe = g->color_off[c + 1];
            if (pe <= pb) continue;
            const K5Stencil& S = g->stencil[(size_t)c];
            if (S.deg > 0) {
                hipLaunchKernelGGL(k5_stencil, dim3((unsigned)((pe - pb + 255) / 256)), dim3(256), 0, ctx->stream, g->state, S,
                                   g->d_thr + (size_t)c * (K5_MAX_DEG + 1), g->row_ptr, g->col, g->val, g->bias, g->site_of, T, sweep0 + (uint32_t)s, tag,
                                   k0, k1);
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
    void* ptrs[] = {g->row_ptr, g->col, g->val, g->bias, g->site_of, g->pos_of, g->state, g->staging, g->samples, g->d_red, g->d_thr};
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
    // regular colour classes: the pattern of the class's middle row must hold for every row but at most K5_EDGE at either end
    static const bool use_stencil = !(getenv("TSU_K5_STENCIL") && atoi(getenv("TSU_K5_STENCIL")) == 0);
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
        S = T;
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
    if (e == hipSuccess) e = hipMalloc((void**)&g->d_thr, (size_t)n_colors * (K5_MAX_DEG + 1) * sizeof(unsigned long long));
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
