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
};

namespace {

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
    for (int s = 0; s < n_sweeps; ++s) {
        for (int c = 0; c < g->n_colors; ++c) {
            const int pb = g->color_off[c], pe = g->color_off[c + 1];
            if (pe <= pb) continue;
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
    void* ptrs[] = {g->row_ptr, g->col, g->val, g->bias, g->site_of, g->pos_of, g->state, g->staging, g->samples, g->d_red};
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
