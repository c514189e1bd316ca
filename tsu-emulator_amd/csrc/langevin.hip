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
#include "tsu_common.h"

struct tsu_langevin {
    tsu_ctx* ctx;
    int n_chains, dim, pitch;
    float* x;
    float* k;
    float* mu;
    float* xinit;
    int steps_per_launch;
    int have_energy;
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
__global__ __launch_bounds__(256) void k3_langevin(float* __restrict__ x, const float* __restrict__ k,
                                                  const float* __restrict__ mu, int n_chains, int quads, int pitch,
                                                  int n_steps, float a, float scale, uint32_t k0, uint32_t k1,
                                                  uint32_t step0, uint32_t chain0, float* __restrict__ traj) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long total = (long long)n_chains * quads;
    if (t >= total) return;
    int ch = (int)(t / quads), q = (int)(t % quads);
    float4* px = reinterpret_cast<float4*>(x + (long long)ch * pitch) + q;
    float4 xv = *px;
    float4 kv = reinterpret_cast<const float4*>(k)[q];
    float4 mv = reinterpret_cast<const float4*>(mu)[q];
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
    *px = xv;
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
    l->steps_per_launch = 0;
    l->have_energy = 0;
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
    for (float* p : {l->x, l->k, l->mu, l->xinit})
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
    int per = l->steps_per_launch > 0 ? l->steps_per_launch : n_steps;
    for (int s = 0; s < n_steps; s += per) {
        int ns = n_steps - s < per ? n_steps - s : per;
        k3_langevin<<<grid, 256, 0, ctx->stream>>>(l->x, l->k, l->mu, l->n_chains, quads, l->pitch, ns, a, scale, (uint32_t)seed,
                                                   (uint32_t)(seed >> 32), step0 + (uint32_t)s, chain0,
                                                   d_traj ? d_traj + (size_t)s * l->n_chains * l->pitch : nullptr);
    }
    hipError_t e = hipGetLastError();
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
