// langevin.hip -- K3 fused drift-diffusion step for separable quadratic energies (gfx950, fp32).
//
// Replaces ThermalSamplingUnit._langevin_step (tsu/core.py:64-80) with the gradient of
// E = 1/2 sum_i k_i (x_i - mu_i)^2 computed analytically in the same kernel (the reference's
// _numerical_gradient, core.py:82-98, calls a Python energy 2d times per step and cannot run on a GPU).
//
// HBM layout: x[n_chains][pitch] float32, pitch = dim rounded up to 4 (float4 per lane, 1 KiB per wave).
// Memory-bound when one step is one launch (8 B per element-step); with steps_per_launch > 1 the state
// stays in registers and only Philox + Box-Muller remain (reported separately as "fused").
//
// RNG stream contract (CPU twin: oracle/tsu_oracle.c ora_langevin_quadratic_f32):
//   quad q = i >> 2 of chain c at step s: W = Philox4x32-10(ctr = (q, c, s, TAG_LANGEVIN), key = seed);
//   pair p in {0,1}: u1 = ((W[2p] >> 8) + 1) / 2^24, u2 = (W[2p+1] >> 8) / 2^24,
//   xi[2p] = sqrt(-2 ln u1) cos(2 pi u2), xi[2p+1] = sqrt(-2 ln u1) sin(2 pi u2);
//   x <- fma(sqrt(2 T dt / gamma), xi, fma(-k (x - mu), dt / gamma, x)).
//
// COUPLED quadratic energies E = 1/2 x^T A x + b^T x (A symmetric; the shape of the reference's multivariate callers,
// tsu/api.py:94): k3_coupled, one launch per step, gradient g = A x + b as an axpy over the rows of A (A^T = A: row j of A
// is column j, so F[i] += A[j, i] x[j] reads contiguous segments -- the layout of the dense Gibbs kernel's axpy pass,
// dense_own.hip): a workgroup owns 64 rows of the result for CB chains, a lane a quad of rows and one of four entry slots, the
// chains' states sit transposed in LDS ([j][chain]: two 16-byte reads give eight chains' x_j).  A is read once per block of CB
// chains: n_chains = 1 is a GEMV at the HBM roofline (4 B per element of A and step); many chains (the restarts of
// sample_from_energy) reuse A from L2.  Same noise stream and update expression as the separable kernel.
// CPU twin: oracle/tsu_oracle.c ora_langevin_coupled_f32 (gradient accumulated in f64, rounded to f32: tolerance in the tests).
#include <utility>

#include "tsu_common.h"

struct tsu_langevin {
    tsu_ctx* ctx;
    int n_chains, dim, pitch;
    float* x;
    float* k;
    float* mu;
    float* xinit;
    int steps_per_launch;
    int have_energy;  // 0: none; 1: separable (k, mu); 2: coupled (A, b)
    int uniform;      // separable with ONE stiffness and ONE centre for every element: the kernel takes them as scalars
    float k0v, mu0v;
    float* A;         // [P][P], P = dim rounded up to 64, zero padded
    float* b;         // [P]
    float* x2;        // the other buffer of a coupled step (every element of the new state needs the whole old one)
    int P;
};

static __device__ __forceinline__ void box_muller4(const u32x4& w, float n[4]) {
    const float inv24 = 1.0f / 16777216.0f;
    float u1a = ((float)(w.x >> 8) + 1.0f) * inv24, u2a = (float)(w.y >> 8) * inv24;
    float u1b = ((float)(w.z >> 8) + 1.0f) * inv24, u2b = (float)(w.w >> 8) * inv24;
    // -2 ln u = -2 ln2 * log2 u; v_log_f32 / v_sqrt_f32 / v_sin_f32 / v_cos_f32 (sin/cos take revolutions)
    float ra = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1a));
    float rb = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1b));
    n[0] = ra * __builtin_amdgcn_cosf(u2a);
    n[1] = ra * __builtin_amdgcn_sinf(u2a);
    n[2] = rb * __builtin_amdgcn_cosf(u2b);
    n[3] = rb * __builtin_amdgcn_sinf(u2b);
}

// one thread = one quad of one chain; n_steps fused in registers; optional trajectory write per step
template <bool GRID2D, bool UNI>
__global__ __launch_bounds__(256) void k3_langevin(float* __restrict__ x, const float* __restrict__ k,
                                                  const float* __restrict__ mu, int n_chains, int quads, int pitch,
                                                  int n_steps, float a, float scale, uint32_t k0, uint32_t k1,
                                                  uint32_t step0, uint32_t chain0, float* __restrict__ traj, float ku, float muu) {
    int ch, q;
    if (GRID2D) {  // (blockIdx.y = chain: no 64-bit division per thread)
        ch = (int)blockIdx.y;
        q = (int)(blockIdx.x * blockDim.x + threadIdx.x);
        if (q >= quads) return;
    } else {
        long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
        long long total = (long long)n_chains * quads;
        if (t >= total) return;
        ch = (int)(t / quads);
        q = (int)(t % quads);
    }
    // (the state is streamed: read once, written once per launch -- nontemporal accesses keep it out of the caches' way:
    // tools/microbench_stream, 1 GiB read-modify-write: 5.98 -> 6.61 TB/s)
    typedef float k3_f4 __attribute__((ext_vector_type(4)));
    k3_f4* px = reinterpret_cast<k3_f4*>(x + (long long)ch * pitch) + q;
    const k3_f4 xin = __builtin_nontemporal_load(px);
    float4 xv = make_float4(xin.x, xin.y, xin.z, xin.w);
    // (UNI: one stiffness and one centre for every element -- E = k/2 sum (x - mu)^2, the README's energies: passed in place of the
    // pointers' first elements, no loads beside the state's)
    float4 kv, mv;
    if (UNI) {
        kv = make_float4(ku, ku, ku, ku);
        mv = make_float4(muu, muu, muu, muu);
    } else {
        kv = reinterpret_cast<const float4*>(k)[q];
        mv = reinterpret_cast<const float4*>(mu)[q];
    }
    for (int s = 0; s < n_steps; ++s) {
        u32x4 w = tsu_philox((uint32_t)q, chain0 + (uint32_t)ch, step0 + (uint32_t)s, TSU_TAG_LANGEVIN, k0, k1);
        float n[4];
        box_muller4(w, n);
        xv.x = __fmaf_rn(scale, n[0], __fmaf_rn(-(kv.x * (xv.x - mv.x)), a, xv.x));
        xv.y = __fmaf_rn(scale, n[1], __fmaf_rn(-(kv.y * (xv.y - mv.y)), a, xv.y));
        xv.z = __fmaf_rn(scale, n[2], __fmaf_rn(-(kv.z * (xv.z - mv.z)), a, xv.z));
        xv.w = __fmaf_rn(scale, n[3], __fmaf_rn(-(kv.w * (xv.w - mv.w)), a, xv.w));
        if (traj) reinterpret_cast<float4*>(traj + ((long long)s * n_chains + ch) * pitch)[q] = xv;
    }
    k3_f4 xo;
    xo.x = xv.x;
    xo.y = xv.y;
    xo.z = xv.z;
    xo.w = xv.w;
    __builtin_nontemporal_store(xo, px);
}


// ---- coupled quadratic energy: one step.  Grid (P / 64, ceil(n_chains / CB)), 1024 threads.
#define K3C_THREADS 1024
#define K3C_WAVES 16
template <int CB>
__global__ __launch_bounds__(K3C_THREADS) void k3_coupled(const float* __restrict__ xin, float* __restrict__ xout, const float* __restrict__ A,
                                                          const float* __restrict__ bvec, int n_chains, int dim, int pitch, int P, int JC, float a,
                                                          float scale, uint32_t k0, uint32_t k1, uint32_t step, uint32_t chain0,
                                                          float* __restrict__ traj) {
    extern __shared__ float k3c_lds[];
    float* xT = k3c_lds;                    // [JC][CB] a chunk of JC columns of the chains' states, chain fastest
    float* red = xT + (size_t)JC * CB;      // [K3C_WAVES][CB][64]
    const int rb = (int)blockIdx.x, c0 = (int)blockIdx.y * CB;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int q16 = lane & 15, t = lane >> 4;
    const int col0 = 64 * rb + 4 * q16;
    float acc[CB][4];
#pragma unroll
    for (int c = 0; c < CB; ++c)
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[c][m] = 0.0f;
    constexpr int U = 4;  // bundles (4 rows of A each) in flight per lane
    // the columns in chunks of JC (the LDS holds JC columns of CB chains: 8 chains whatever the dimension); the accumulators run on
    for (int j0 = 0; j0 < P; j0 += JC) {
        const int jc = P - j0 < JC ? P - j0 : JC;  // (a multiple of 64)
        if (j0) __syncthreads();  // (the previous chunk has been read by every wave)
        for (int q = (int)threadIdx.x; q < jc * CB; q += K3C_THREADS) {
            const int j = j0 + q / CB, c = q % CB;
            xT[q] = (j < dim && c0 + c < n_chains) ? xin[(size_t)(c0 + c) * pitch + j] : 0.0f;
        }
        __syncthreads();
        const int nb = jc / 4;  // (a multiple of 16)
        for (int bnd = wv; bnd < nb; bnd += K3C_WAVES * U) {
            float4 av[U];
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (bnd + K3C_WAVES * u < nb) av[u] = *reinterpret_cast<const float4*>(A + (size_t)(j0 + 4 * (bnd + K3C_WAVES * u) + t) * P + col0);
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (bnd + K3C_WAVES * u < nb) {
                    const float* xs = xT + (size_t)(4 * (bnd + K3C_WAVES * u) + t) * CB;
#pragma unroll
                    for (int c = 0; c < CB; ++c) {
                        const float xv = xs[c];
                        acc[c][0] = __fmaf_rn(av[u].x, xv, acc[c][0]);
                        acc[c][1] = __fmaf_rn(av[u].y, xv, acc[c][1]);
                        acc[c][2] = __fmaf_rn(av[u].z, xv, acc[c][2]);
                        acc[c][3] = __fmaf_rn(av[u].w, xv, acc[c][3]);
                    }
                }
        }
    }
    // the four entry slots of a quad (fixed order), then the sixteen waves through LDS
#pragma unroll
    for (int c = 0; c < CB; ++c)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            float z = acc[c][m];
            z += __shfl_xor(z, 16, 64);
            z += __shfl_xor(z, 32, 64);
            acc[c][m] = z;
        }
    if (lane < 16) {
#pragma unroll
        for (int c = 0; c < CB; ++c)
#pragma unroll
            for (int m = 0; m < 4; ++m) red[(wv * CB + c) * 64 + 4 * lane + m] = acc[c][m];
    }
    __syncthreads();
    for (int q = (int)threadIdx.x; q < CB * 64; q += K3C_THREADS) {
        const int c = q >> 6, r = q & 63;
        const int i = 64 * rb + r, ch = c0 + c;
        if (i >= dim || ch >= n_chains) continue;
        float g = 0.0f;
#pragma unroll
        for (int w = 0; w < K3C_WAVES; ++w) g += red[(w * CB + c) * 64 + r];
        g += bvec[i];
        const u32x4 wd = tsu_philox((uint32_t)i >> 2, chain0 + (uint32_t)ch, step, TSU_TAG_LANGEVIN, k0, k1);
        float n[4];
        box_muller4(wd, n);
        const float nz = (i & 3) == 0 ? n[0] : (i & 3) == 1 ? n[1] : (i & 3) == 2 ? n[2] : n[3];
        const float v = __fmaf_rn(scale, nz, __fmaf_rn(-g, a, xin[(size_t)ch * pitch + i]));
        xout[(size_t)ch * pitch + i] = v;
        if (traj) traj[(size_t)ch * pitch + i] = v;
    }
}

// columns per chunk: as many as the LDS holds for CB chains next to the partial sums (144 KB in all), a multiple of 64, at most P
static int k3c_chunk(int P, int cb) {
    const int budget = 144 * 1024 - K3C_WAVES * cb * 64 * 4;
    int jc = (budget / 4 / cb) / 64 * 64;
    return jc < P ? jc : P;
}

template <int CB>
static hipError_t k3c_launch(tsu_langevin* l, const float* xin, float* xout, float a, float scale, uint64_t seed, uint32_t step, uint32_t chain0,
                             float* traj) {
    tsu_ctx* ctx = l->ctx;
    const int JC = k3c_chunk(l->P, CB);
    const size_t lds = ((size_t)JC * CB + (size_t)K3C_WAVES * CB * 64) * sizeof(float);
    hipError_t e = tsu_func_allow_lds(ctx, (const void*)k3_coupled<CB>, (int)lds);
    if (e != hipSuccess) return e;
    const dim3 grid((unsigned)(l->P / 64), (unsigned)((l->n_chains + CB - 1) / CB));
    hipLaunchKernelGGL(k3_coupled<CB>, grid, dim3(K3C_THREADS), lds, ctx->stream, xin, xout, l->A, l->b, l->n_chains, l->dim, l->pitch, l->P, JC, a,
                       scale, (uint32_t)seed, (uint32_t)(seed >> 32), step, chain0, traj);
    return hipGetLastError();
}

// x[c] <- x_init + amp * N(0,1), chain id chain0 + c (core.py:142-143)
__global__ __launch_bounds__(256) void k3_restart(float* __restrict__ x, const float* __restrict__ xinit, int n_chains,
                                                 int quads, int pitch, float amp, uint32_t k0, uint32_t k1,
                                                 uint32_t chain0) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)n_chains * quads) return;
    int ch = (int)(t / quads), q = (int)(t % quads);
    u32x4 w = tsu_philox((uint32_t)q, chain0 + (uint32_t)ch, 0u, TSU_TAG_LANGEVIN_RESTART, k0, k1);
    float n[4];
    box_muller4(w, n);
    float4 xi = reinterpret_cast<const float4*>(xinit)[q];
    float4 o = make_float4(__fmaf_rn(amp, n[0], xi.x), __fmaf_rn(amp, n[1], xi.y), __fmaf_rn(amp, n[2], xi.z),
                           __fmaf_rn(amp, n[3], xi.w));
    reinterpret_cast<float4*>(x + (long long)ch * pitch)[q] = o;
}

extern "C" {

int tsu_langevin_create(tsu_ctx* ctx, int n_chains, int dim, tsu_langevin** out) {
    TSU_ENTER(ctx);
    if (!ctx || !out) return TSU_E_INVALID;
    *out = nullptr;
    TSU_REQUIRE(ctx, n_chains >= 1 && dim >= 1, "langevin_create: n_chains and dim must be positive");
    tsu_langevin* l = new (std::nothrow) tsu_langevin();
    if (!l) return tsu_fail(ctx, TSU_E_NOMEM, "langevin_create: host allocation failed");
    l->ctx = ctx;
    l->n_chains = n_chains;
    l->dim = dim;
    l->pitch = (dim + 3) / 4 * 4;
    l->x = l->k = l->mu = l->xinit = nullptr;
    l->A = l->b = l->x2 = nullptr;
    l->P = (dim + 63) / 64 * 64;
    l->steps_per_launch = 0;
    l->have_energy = 0;
    l->uniform = 0;
    l->k0v = l->mu0v = 0.0f;
    size_t xb = (size_t)n_chains * l->pitch * sizeof(float), vb = (size_t)l->pitch * sizeof(float);
    hipError_t e = hipMalloc(&l->x, xb);
    if (e == hipSuccess) e = hipMalloc(&l->k, vb);
    if (e == hipSuccess) e = hipMalloc(&l->mu, vb);
    if (e == hipSuccess) e = hipMalloc(&l->xinit, vb);
    if (e == hipSuccess) e = hipMemsetAsync(l->x, 0, xb, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(l->k, 0, vb, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(l->mu, 0, vb, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(l->xinit, 0, vb, ctx->stream);
    if (e != hipSuccess) {
        int rc = tsu_fail(ctx, e == hipErrorOutOfMemory ? TSU_E_NOMEM : TSU_E_HIP, "langevin_create: %s", hipGetErrorString(e));
        for (float* p : {l->x, l->k, l->mu, l->xinit})
            if (p) (void)hipFree(p);
        delete l;
        return rc;
    }
    *out = l;
    return TSU_OK;
}

int tsu_langevin_destroy(tsu_langevin* l) {
    TSU_ENTER(l ? l->ctx : nullptr);
    if (!l) return TSU_OK;
    (void)hipStreamSynchronize(l->ctx->stream);
    for (float* p : {l->x, l->k, l->mu, l->xinit, l->A, l->b, l->x2})
        if (p) (void)hipFree(p);
    delete l;
    return TSU_OK;
}

int tsu_langevin_set_state(tsu_langevin* l, const float* x_host) {
    TSU_ENTER(l ? l->ctx : nullptr);
    if (!l) return TSU_E_INVALID;
    TSU_REQUIRE(l->ctx, x_host != nullptr, "langevin_set_state: NULL");
    size_t w = (size_t)l->dim * sizeof(float);
    TSU_HIP_TRY(l->ctx, hipMemcpy2DAsync(l->x, (size_t)l->pitch * sizeof(float), x_host, w, w, (size_t)l->n_chains,
                                         hipMemcpyHostToDevice, l->ctx->stream));
    TSU_HIP_TRY(l->ctx, hipStreamSynchronize(l->ctx->stream));
    return TSU_OK;
}

int tsu_langevin_get_state(tsu_langevin* l, float* x_host) {
    TSU_ENTER(l ? l->ctx : nullptr);
    if (!l) return TSU_E_INVALID;
    TSU_REQUIRE(l->ctx, x_host != nullptr, "langevin_get_state: NULL");
    size_t w = (size_t)l->dim * sizeof(float);
    TSU_HIP_TRY(l->ctx, hipMemcpy2DAsync(x_host, w, l->x, (size_t)l->pitch * sizeof(float), w, (size_t)l->n_chains,
                                         hipMemcpyDeviceToHost, l->ctx->stream));
    TSU_HIP_TRY(l->ctx, hipStreamSynchronize(l->ctx->stream));
    return TSU_OK;
}

int tsu_langevin_set_energy(tsu_langevin* l, const float* k_host, const float* mu_host) {
    TSU_ENTER(l ? l->ctx : nullptr);
    if (!l) return TSU_E_INVALID;
    TSU_REQUIRE(l->ctx, k_host && mu_host, "langevin_set_energy: NULL");
    size_t w = (size_t)l->dim * sizeof(float);
    TSU_HIP_TRY(l->ctx, hipMemcpyAsync(l->k, k_host, w, hipMemcpyHostToDevice, l->ctx->stream));
    TSU_HIP_TRY(l->ctx, hipMemcpyAsync(l->mu, mu_host, w, hipMemcpyHostToDevice, l->ctx->stream));
    TSU_HIP_TRY(l->ctx, hipStreamSynchronize(l->ctx->stream));
    l->have_energy = 1;
    l->uniform = 1;
    for (int i = 1; i < l->dim && l->uniform; ++i)
        if (k_host[i] != k_host[0] || mu_host[i] != mu_host[0]) l->uniform = 0;
    l->k0v = k_host[0];
    l->mu0v = mu_host[0];
    return TSU_OK;
}

int tsu_langevin_set_coupling(tsu_langevin* l, const float* A_host, const float* b_host) {
    TSU_ENTER(l ? l->ctx : nullptr);
    if (!l) return TSU_E_INVALID;
    tsu_ctx* ctx = l->ctx;
    TSU_REQUIRE(ctx, A_host != nullptr, "langevin_set_coupling: NULL matrix");
    const int d = l->dim, P = l->P;
    TSU_REQUIRE(ctx, d <= 65536, "langevin_set_coupling: dim %d is beyond the coupled kernel's range (65536: a 16 GiB matrix)", d);
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < i; ++j)
            TSU_REQUIRE(ctx, A_host[(size_t)i * d + j] == A_host[(size_t)j * d + i], "langevin_set_coupling: the matrix must be symmetric (entry %d, %d)", i, j);
    if (!l->A) {
        TSU_HIP_TRY(ctx, hipMalloc(&l->A, (size_t)P * P * sizeof(float)));
        TSU_HIP_TRY(ctx, hipMalloc(&l->b, (size_t)P * sizeof(float)));
        TSU_HIP_TRY(ctx, hipMalloc(&l->x2, (size_t)l->n_chains * l->pitch * sizeof(float)));
        TSU_HIP_TRY(ctx, hipMemsetAsync(l->x2, 0, (size_t)l->n_chains * l->pitch * sizeof(float), ctx->stream));
    }
    TSU_HIP_TRY(ctx, hipMemsetAsync(l->A, 0, (size_t)P * P * sizeof(float), ctx->stream));
    TSU_HIP_TRY(ctx, hipMemsetAsync(l->b, 0, (size_t)P * sizeof(float), ctx->stream));
    TSU_HIP_TRY(ctx, hipMemcpy2DAsync(l->A, (size_t)P * sizeof(float), A_host, (size_t)d * sizeof(float), (size_t)d * sizeof(float), (size_t)d,
                                      hipMemcpyHostToDevice, ctx->stream));
    if (b_host) TSU_HIP_TRY(ctx, hipMemcpyAsync(l->b, b_host, (size_t)d * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    TSU_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    l->have_energy = 2;
    return TSU_OK;
}

int tsu_langevin_set_kernel(tsu_langevin* l, int steps_per_launch) {
    TSU_ENTER(l ? l->ctx : nullptr);
    if (!l) return TSU_E_INVALID;
    TSU_REQUIRE(l->ctx, steps_per_launch >= 0, "langevin_set_kernel: steps_per_launch must be >= 0");
    l->steps_per_launch = steps_per_launch;
    return TSU_OK;
}

int tsu_langevin_restart(tsu_langevin* l, const float* x_init_host, float amp, uint64_t seed, uint32_t chain0) {
    TSU_ENTER(l ? l->ctx : nullptr);
    if (!l) return TSU_E_INVALID;
    tsu_ctx* ctx = l->ctx;
    TSU_REQUIRE(ctx, x_init_host != nullptr, "langevin_restart: NULL");
    TSU_HIP_TRY(ctx, hipMemcpyAsync(l->xinit, x_init_host, (size_t)l->dim * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    TSU_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // the host buffer is the caller's
    int quads = l->pitch / 4;
    long long total = (long long)l->n_chains * quads;
    k3_restart<<<(unsigned)((total + 255) / 256), 256, 0, ctx->stream>>>(l->x, l->xinit, l->n_chains, quads, l->pitch, amp,
                                                                       (uint32_t)seed, (uint32_t)(seed >> 32), chain0);
    TSU_HIP_TRY(ctx, hipGetLastError());
    return TSU_OK;
}

int tsu_langevin_step(tsu_langevin* l, int n_steps, float dt, float gamma, float T, uint64_t seed, uint32_t step0,
                      uint32_t chain0, float* traj_host) {
    TSU_ENTER(l ? l->ctx : nullptr);
    if (!l) return TSU_E_INVALID;
    tsu_ctx* ctx = l->ctx;
    TSU_REQUIRE(ctx, n_steps >= 0, "langevin_step: n_steps must be >= 0");
    TSU_REQUIRE(ctx, T > 0.0f && dt > 0.0f && gamma > 0.0f, "langevin_step: T, dt, gamma must be positive");
    TSU_REQUIRE(ctx, l->have_energy, "langevin_step: call tsu_langevin_set_energy first");
    if (n_steps == 0) return TSU_OK;
    float a = dt / gamma;
    float scale = sqrtf(2.0f * T * dt / gamma);
    int quads = l->pitch / 4;
    long long total = (long long)l->n_chains * quads;
    unsigned grid = (unsigned)((total + 255) / 256);
    float* d_traj = nullptr;
    if (traj_host) TSU_HIP_TRY(ctx, hipMalloc(&d_traj, (size_t)n_steps * l->n_chains * l->pitch * sizeof(float)));
    hipError_t e = hipSuccess;
    if (l->have_energy == 2) {
        // chains per workgroup (one stream of A serves them all): at most 8, no more than there are chains
        int cb = 8;
        while (cb > 1 && cb / 2 >= l->n_chains) cb /= 2;
        for (int s = 0; s < n_steps && e == hipSuccess; ++s) {
            float* tr = d_traj ? d_traj + (size_t)s * l->n_chains * l->pitch : nullptr;
            const uint32_t st = step0 + (uint32_t)s;
            if (cb == 8) e = k3c_launch<8>(l, l->x, l->x2, a, scale, seed, st, chain0, tr);
            else if (cb == 4) e = k3c_launch<4>(l, l->x, l->x2, a, scale, seed, st, chain0, tr);
            else if (cb == 2) e = k3c_launch<2>(l, l->x, l->x2, a, scale, seed, st, chain0, tr);
            else e = k3c_launch<1>(l, l->x, l->x2, a, scale, seed, st, chain0, tr);
            std::swap(l->x, l->x2);
        }
    } else {
        int per = l->steps_per_launch > 0 ? l->steps_per_launch : n_steps;
        for (int s = 0; s < n_steps; s += per) {
            int ns = n_steps - s < per ? n_steps - s : per;
            float* tr = d_traj ? d_traj + (size_t)s * l->n_chains * l->pitch : nullptr;
            const dim3 g2((unsigned)((quads + 255) / 256), (unsigned)l->n_chains);
#define K3_ARGS l->x, l->k, l->mu, l->n_chains, quads, l->pitch, ns, a, scale, (uint32_t)seed, (uint32_t)(seed >> 32), step0 + (uint32_t)s, chain0, tr, l->k0v, l->mu0v
            if (l->n_chains <= 65535) {
                if (l->uniform) k3_langevin<true, true><<<g2, 256, 0, ctx->stream>>>(K3_ARGS);
                else k3_langevin<true, false><<<g2, 256, 0, ctx->stream>>>(K3_ARGS);
            } else {
                if (l->uniform) k3_langevin<false, true><<<grid, 256, 0, ctx->stream>>>(K3_ARGS);
                else k3_langevin<false, false><<<grid, 256, 0, ctx->stream>>>(K3_ARGS);
            }
#undef K3_ARGS
        }
    }
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess && traj_host) {
        size_t w = (size_t)l->dim * sizeof(float);
        e = hipMemcpy2DAsync(traj_host, w, d_traj, (size_t)l->pitch * sizeof(float), w, (size_t)n_steps * l->n_chains,
                             hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    }
    if (d_traj) (void)hipFree(d_traj);
    if (e != hipSuccess) return tsu_fail(ctx, TSU_E_HIP, "langevin_step: %s", hipGetErrorString(e));
    return TSU_OK;
}

}  // extern "C"
