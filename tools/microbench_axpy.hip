// microbench_axpy.hip -- the axpy pass of k2_own in isolation: W workgroups x 1024 threads, each adds the 256-byte row segments
// J[j, 64 w .. 64 w + 63] of a list of nl rows j into 64 accumulators; variants of the inner loop.
//   hipcc --offload-arch=gfx950 -O3 -o tools/microbench_axpy tools/microbench_axpy.hip && tools/microbench_axpy [n] [nl] [W] [reps]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));               \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

// V0: quad layout, two-phase, U bundles in flight (what k2_own runs)
template <int U, bool NT>
__global__ __launch_bounds__(1024) void axpy_quad(const float* __restrict__ J, int n, const uint32_t* __restrict__ lists, int nl, int reps,
                                                  double* __restrict__ out) {
    __shared__ uint32_t lst[8192];
    __shared__ double red[16][64];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int col0 = 64 * blockIdx.x + 4 * (lane & 15), t = lane >> 4;
    double tot = 0.0;
    for (int r = 0; r < reps; ++r) {
        for (int k = threadIdx.x; k < nl; k += 1024) lst[k] = lists[(size_t)r * nl + k];
        __syncthreads();
        double acc[4] = {0, 0, 0, 0};
        const int nb = (nl + 3) / 4;
        for (int b = wv; b < nb; b += 16 * U) {
            float4 x[U];
            uint32_t e[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = (b + 16 * u) * 4 + t;
                e[u] = k < nl ? lst[k] : 0u;
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (b + 16 * u < nb) {
                    typedef float f4v __attribute__((ext_vector_type(4)));
                    const f4v* p = reinterpret_cast<const f4v*>(J + (size_t)(e[u] & 0xFFFFu) * n + col0);
                    const f4v q = NT ? __builtin_nontemporal_load(p) : *p;
                    x[u] = make_float4(q.x, q.y, q.z, q.w);
                }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (b + 16 * u < nb) {
                    const double sg = (e[u] >> 16) & 1u ? ((e[u] >> 17) & 1u ? -1.0 : 1.0) : 0.0;
                    acc[0] += sg * (double)x[u].x;
                    acc[1] += sg * (double)x[u].y;
                    acc[2] += sg * (double)x[u].z;
                    acc[3] += sg * (double)x[u].w;
                }
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            acc[m] += __shfl_xor(acc[m], 16, 64);
            acc[m] += __shfl_xor(acc[m], 32, 64);
        }
        if (lane < 16)
            for (int m = 0; m < 4; ++m) red[wv][4 * lane + m] = acc[m];
        __syncthreads();
        if (wv == 0) {
            double z = 0.0;
            for (int u = 0; u < 16; ++u) z += red[u][lane];
            tot += z;
        }
        __syncthreads();
    }
    if (wv == 0) out[64 * blockIdx.x + lane] = tot;
}

// V1: float accumulation inside a round (4 x float adds), one f64 add per round and element -- NOT the same sums (a bound on what
// cheaper arithmetic would buy)
template <int U>
__global__ __launch_bounds__(1024) void axpy_quad_f32(const float* __restrict__ J, int n, const uint32_t* __restrict__ lists, int nl, int reps,
                                                      double* __restrict__ out) {
    __shared__ uint32_t lst[8192];
    __shared__ double red[16][64];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int col0 = 64 * blockIdx.x + 4 * (lane & 15), t = lane >> 4;
    double tot = 0.0;
    for (int r = 0; r < reps; ++r) {
        for (int k = threadIdx.x; k < nl; k += 1024) lst[k] = lists[(size_t)r * nl + k];
        __syncthreads();
        double acc[4] = {0, 0, 0, 0};
        const int nb = (nl + 3) / 4;
        for (int b = wv; b < nb; b += 16 * U) {
            float4 x[U];
            uint32_t e[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = (b + 16 * u) * 4 + t;
                e[u] = k < nl ? lst[k] : 0u;
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (b + 16 * u < nb) x[u] = *reinterpret_cast<const float4*>(J + (size_t)(e[u] & 0xFFFFu) * n + col0);
            float f[4] = {0, 0, 0, 0};
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (b + 16 * u < nb) {
                    const float sg = (e[u] >> 16) & 1u ? ((e[u] >> 17) & 1u ? -1.0f : 1.0f) : 0.0f;
                    f[0] += sg * x[u].x;
                    f[1] += sg * x[u].y;
                    f[2] += sg * x[u].z;
                    f[3] += sg * x[u].w;
                }
            for (int m = 0; m < 4; ++m) acc[m] += (double)f[m];
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            acc[m] += __shfl_xor(acc[m], 16, 64);
            acc[m] += __shfl_xor(acc[m], 32, 64);
        }
        if (lane < 16)
            for (int m = 0; m < 4; ++m) red[wv][4 * lane + m] = acc[m];
        __syncthreads();
        if (wv == 0) {
            double z = 0.0;
            for (int u = 0; u < 16; ++u) z += red[u][lane];
            tot += z;
        }
        __syncthreads();
    }
    if (wv == 0) out[64 * blockIdx.x + lane] = tot;
}

// V2: loads only (no arithmetic beyond one xor-fold per bundle): the memory side alone
template <int U>
__global__ __launch_bounds__(1024) void axpy_loads(const float* __restrict__ J, int n, const uint32_t* __restrict__ lists, int nl, int reps,
                                                   double* __restrict__ out) {
    __shared__ uint32_t lst[8192];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int col0 = 64 * blockIdx.x + 4 * (lane & 15), t = lane >> 4;
    float z = 0.f;
    for (int r = 0; r < reps; ++r) {
        for (int k = threadIdx.x; k < nl; k += 1024) lst[k] = lists[(size_t)r * nl + k];
        __syncthreads();
        const int nb = (nl + 3) / 4;
        for (int b = wv; b < nb; b += 16 * U) {
            float4 x[U];
            uint32_t e[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = (b + 16 * u) * 4 + t;
                e[u] = k < nl ? lst[k] : 0u;
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (b + 16 * u < nb) x[u] = *reinterpret_cast<const float4*>(J + (size_t)(e[u] & 0xFFFFu) * n + col0);
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (b + 16 * u < nb) z += x[u].x + x[u].w;
        }
        __syncthreads();
    }
    if (z == 12345.f) out[threadIdx.x] = z;
}


// V3: the replica layout of k2_own (own_axpy_rep): lane = row (one dword per lane), one entry per wave-instruction, R accumulators
// per lane with scalar multipliers from the entry's 2-bit codes.  WHAT: 0 = as the kernel runs it, 1 = loads only, 2 = arithmetic
// only (the same instruction stream on a register value)
template <int R, int U, int WHAT>
__global__ __launch_bounds__(1024) void axpy_rep(const float* __restrict__ J, int n, const uint32_t* __restrict__ lists, int nl, int reps,
                                                 double* __restrict__ out) {
    __shared__ uint32_t lst[8192];
    __shared__ double red[16][R][64];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int col = 64 * blockIdx.x + lane;
    double tot = 0.0;
    float fake = (float)lane;
    for (int r = 0; r < reps; ++r) {
        for (int k = threadIdx.x; k < nl; k += 1024) lst[k] = lists[(size_t)r * nl + k] | (0x5A5Au << 18) | ((k * 2654435761u) & 0xFFFC0000u);
        __syncthreads();
        double acc[R];
#pragma unroll
        for (int rho = 0; rho < R; ++rho) acc[rho] = 0.0;
        for (int k = wv; k < nl; k += 16 * U) {
            float x[U];
            uint32_t e[U];
#pragma unroll
            for (int u = 0; u < U; ++u) e[u] = k + 16 * u < nl ? lst[k + 16 * u] : 0u;
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (k + 16 * u < nl) {
                    const uint32_t j = (uint32_t)__builtin_amdgcn_readfirstlane((int)e[u]) & 0xFFFFu;
                    x[u] = WHAT == 2 ? fake + (float)j : J[(size_t)j * n + col];
                }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (k + 16 * u < nl) {
                    const uint32_t ev = (uint32_t)__builtin_amdgcn_readfirstlane((int)e[u]);
                    const double v = (double)x[u];
                    if (WHAT == 1) {
                        acc[0] += v;
                    } else {
#pragma unroll
                        for (int rho = 0; rho < R; ++rho) {
                            const double sg = (double)(((int)(ev << (14 - 2 * rho))) >> 30);
                            acc[rho] = fma(sg, v, acc[rho]);
                        }
                    }
                }
        }
#pragma unroll
        for (int rho = 0; rho < R; ++rho) red[wv][rho][lane] = acc[rho];
        __syncthreads();
        if (wv < R) {
            double z = 0.0;
            for (int u = 0; u < 16; ++u) z += red[u][wv][lane];
            tot += z;
        }
        __syncthreads();
    }
    if (wv < R) out[(64 * blockIdx.x + lane) * 8 + wv] = tot;
}

// V4: replicas in the quad layout with PER-LANE multipliers: lane = (quad of rows q < 16, replica pair p < 4); a wave-instruction loads
// ONE entry's segment four times over (the four replica pairs read the same 256 bytes: one pass through the texture path, no extra
// memory traffic), 2 replicas x 4 rows of accumulators per lane, multipliers from the lane's own two codes
template <int U, int WHAT>
__global__ __launch_bounds__(1024) void axpy_rep_quad(const float* __restrict__ J, int n, const uint32_t* __restrict__ lists, int nl, int reps,
                                                      double* __restrict__ out) {
    __shared__ uint32_t lst[8192];
    __shared__ double red[16][8][64];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int q = lane & 15, p = lane >> 4;
    const int col0 = 64 * blockIdx.x + 4 * q;
    double tot = 0.0;
    for (int r = 0; r < reps; ++r) {
        for (int k = threadIdx.x; k < nl; k += 1024) lst[k] = lists[(size_t)r * nl + k] | (0x5A5Au << 18) | ((k * 2654435761u) & 0xFFFC0000u);
        __syncthreads();
        double acc[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
        for (int k = wv; k < nl; k += 16 * U) {
            float4 x[U];
            uint32_t e[U];
#pragma unroll
            for (int u = 0; u < U; ++u) e[u] = k + 16 * u < nl ? lst[k + 16 * u] : 0u;
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (k + 16 * u < nl) {
                    const uint32_t j = (uint32_t)__builtin_amdgcn_readfirstlane((int)e[u]) & 0xFFFFu;
                    x[u] = *reinterpret_cast<const float4*>(J + (size_t)j * n + col0);
                }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (k + 16 * u < nl) {
                    if (WHAT == 1) {
                        acc[0][0] += (double)(x[u].x + x[u].w);
                    } else {
                        const uint32_t c4 = e[u] >> (16 + 4 * p);  // my pair's two codes
                        const double s0 = (double)(((int)(c4 << 30)) >> 30), s1 = (double)(((int)(c4 << 28)) >> 30);
                        const double v0 = (double)x[u].x, v1 = (double)x[u].y, v2 = (double)x[u].z, v3 = (double)x[u].w;
                        acc[0][0] = fma(s0, v0, acc[0][0]);
                        acc[0][1] = fma(s0, v1, acc[0][1]);
                        acc[0][2] = fma(s0, v2, acc[0][2]);
                        acc[0][3] = fma(s0, v3, acc[0][3]);
                        acc[1][0] = fma(s1, v0, acc[1][0]);
                        acc[1][1] = fma(s1, v1, acc[1][1]);
                        acc[1][2] = fma(s1, v2, acc[1][2]);
                        acc[1][3] = fma(s1, v3, acc[1][3]);
                    }
                }
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int m = 0; m < 4; ++m) red[wv][2 * p + a][4 * q + m] = acc[a][m];
        __syncthreads();
        if (wv < 8) {
            double z = 0.0;
            for (int u = 0; u < 16; ++u) z += red[u][wv][lane];
            tot += z;
        }
        __syncthreads();
    }
    if (wv < 8) out[(64 * blockIdx.x + lane) * 8 + wv] = tot;
}

// V5: the replica layout with the multipliers' f64 bit patterns made once per (entry, replica) -- not by a v_cvt_f64_i32 per entry,
// replica AND wave-instruction.  ALG 1: lane l prepares the high dword for entry l / 8 and replica l % 8 of the round's entries,
// v_readlane brings it to a scalar register, the multiply-add takes the scalar pair {0, hi}.  ALG 2: a four-entry table in scalar
// registers indexed through M0 (s_movrels_b32).
template <int U, int WHAT, int ALG, int NS = 4>
__global__ __launch_bounds__(1024) void axpy_rep2(const float* __restrict__ J, int n, const uint32_t* __restrict__ lists, int nl, int reps,
                                                  double* __restrict__ out) {
    constexpr int R = 8;
    static_assert(U % 8 == 0, "eight entries per multiplier register");
    __shared__ uint32_t lst[8192 + 16 * 64];
    __shared__ double red[16][R][64];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int col = 64 * blockIdx.x + lane;
    double tot = 0.0;
    float fake = (float)lane;
    for (int r = 0; r < reps; ++r) {
        for (int k = threadIdx.x; k < nl; k += 1024) lst[k] = lists[(size_t)r * nl + k] | (0x5A5Au << 18) | ((k * 2654435761u) & 0xFFFC0000u);
        for (int k = nl + threadIdx.x; k < nl + 16 * U; k += 1024) lst[k] = 0u;  // (entries beyond the list: weight 0)
        __syncthreads();
        double acc[R];
#pragma unroll
        for (int rho = 0; rho < R; ++rho) acc[rho] = 0.0;
        for (int k = wv; k < nl; k += 16 * U) {
            float x[U];
            uint32_t e[U];
#pragma unroll
            for (int u = 0; u < U; ++u) e[u] = lst[k + 16 * u];
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (k + 16 * u < nl) {
                    const uint32_t j = (uint32_t)__builtin_amdgcn_readfirstlane((int)e[u]) & 0xFFFFu;
                    x[u] = WHAT == 2 ? fake + (float)j : J[(size_t)j * n + col];
                }
            uint32_t mh[U / 8];
            if (ALG == 1) {
#pragma unroll
                for (int h = 0; h < U / 8; ++h) {
                    const uint32_t ev = lst[k + 16 * (8 * h + (lane >> 3))];
                    const int t = ((int)(ev << (14 - 2 * (lane & 7)))) >> 30;  // +1, -1, 0
                    mh[h] = ((uint32_t)t & 0x80000000u) | ((uint32_t)(t & 1) * 0x3FF00000u);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (k + 16 * u < nl) {
                    const double v = (double)x[u];
                    if (ALG == 1) {
#pragma unroll
                        for (int rho = 0; rho < R; ++rho) {
                            const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)mh[u / 8], (u % 8) * 8 + rho);
                            acc[rho] = fma(__hiloint2double((int)hi, 0), v, acc[rho]);
                        }
                    } else if (ALG == 4) {  // (a floor: the multipliers cost nothing -- one scalar pair per entry for all replicas)
                        const uint32_t ev = (uint32_t)__builtin_amdgcn_readfirstlane((int)e[u]);
                        const double sg = __hiloint2double((int)(ev & 0xC0000000u), 0);
#pragma unroll
                        for (int rho = 0; rho < R; ++rho) acc[rho] = fma(sg, v, acc[rho]);
                    } else if (ALG == 6) {  // NS multipliers by the scalar unit, R - NS by the vector unit (the entry through an opaque VGPR)
                        const uint32_t ev = (uint32_t)__builtin_amdgcn_readfirstlane((int)e[u]);
                        uint32_t evv = e[u];
                        asm volatile("; opaque %0" : "+v"(evv));
#pragma unroll
                        for (int rho = 0; rho < R; ++rho) {
                            const uint32_t src = rho < NS ? ev : evv;
                            const uint32_t hi = (src << (14 - 2 * rho)) & 0xC0000000u;
                            acc[rho] = fma(__hiloint2double((int)hi, 0), v, acc[rho]);
                        }
                    } else if (ALG == 5) {  // half of the multipliers by the scalar unit, half by the vector unit (two 32-bit operations each)
                        const uint32_t ev = (uint32_t)__builtin_amdgcn_readfirstlane((int)e[u]);
#pragma unroll
                        for (int rho = 0; rho < R; ++rho) {
                            const uint32_t src = (rho & 1) ? e[u] : ev;
                            const uint32_t hi = (src << (14 - 2 * rho)) & 0xC0000000u;
                            acc[rho] = fma(__hiloint2double((int)hi, 0), v, acc[rho]);
                        }
                    } else if (ALG == 3) {
                        // +-2.0 and 0.0 have the high dwords 0x40000000, 0xC0000000, 0: the sign-extended 2-bit code shifted to the top
                        // IS the multiplier's high dword (sums come out doubled, exactly: halved at the end)
                        const uint32_t ev = (uint32_t)__builtin_amdgcn_readfirstlane((int)e[u]);
#pragma unroll
                        for (int rho = 0; rho < R; ++rho) {
                            const uint32_t hi = (ev << (14 - 2 * rho)) & 0xC0000000u;
                            acc[rho] = fma(__hiloint2double((int)hi, 0), v, acc[rho]);
                        }
                    } else {
                        const uint32_t ev = (uint32_t)__builtin_amdgcn_readfirstlane((int)e[u]);
                        uint32_t h0, h1, h2, h3, h4, h5, h6, h7;
                        asm volatile(
                            "s_mov_b32 s96, 0\n s_mov_b32 s97, 0x3ff00000\n s_mov_b32 s99, 0xbff00000\n"
                            "s_bfe_u32 m0, %8, 0x20010\n s_nop 0\n s_movrels_b32 %0, s96\n"
                            "s_bfe_u32 m0, %8, 0x20012\n s_nop 0\n s_movrels_b32 %1, s96\n"
                            "s_bfe_u32 m0, %8, 0x20014\n s_nop 0\n s_movrels_b32 %2, s96\n"
                            "s_bfe_u32 m0, %8, 0x20016\n s_nop 0\n s_movrels_b32 %3, s96\n"
                            "s_bfe_u32 m0, %8, 0x20018\n s_nop 0\n s_movrels_b32 %4, s96\n"
                            "s_bfe_u32 m0, %8, 0x2001a\n s_nop 0\n s_movrels_b32 %5, s96\n"
                            "s_bfe_u32 m0, %8, 0x2001c\n s_nop 0\n s_movrels_b32 %6, s96\n"
                            "s_bfe_u32 m0, %8, 0x2001e\n s_nop 0\n s_movrels_b32 %7, s96\n"
                            : "=s"(h0), "=s"(h1), "=s"(h2), "=s"(h3), "=s"(h4), "=s"(h5), "=s"(h6), "=s"(h7)
                            : "s"(ev)
                            : "m0", "s96", "s97", "s98", "s99");
                        const uint32_t hh[8] = {h0, h1, h2, h3, h4, h5, h6, h7};
#pragma unroll
                        for (int rho = 0; rho < R; ++rho) acc[rho] = fma(__hiloint2double((int)hh[rho], 0), v, acc[rho]);
                    }
                }
        }
#pragma unroll
        for (int rho = 0; rho < R; ++rho) red[wv][rho][lane] = acc[rho];
        __syncthreads();
        if (wv < R) {
            double z = 0.0;
            for (int u = 0; u < 16; ++u) z += red[u][wv][lane];
            tot += z;
        }
        __syncthreads();
    }
    if (wv < R) out[(64 * blockIdx.x + lane) * 8 + wv] = tot;
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 16384;
    const int nl = argc > 2 ? atoi(argv[2]) : 1700;
    const int W = argc > 3 ? atoi(argv[3]) : 64;
    const int reps = argc > 4 ? atoi(argv[4]) : 50;
    float* J;
    CHECK(hipMalloc(&J, (size_t)n * n * 4));
    CHECK(hipMemset(J, 0, (size_t)n * n * 4));
    std::vector<uint32_t> h((size_t)reps * nl);
    uint64_t s = 88172645463325252ull;
    for (int r = 0; r < reps; ++r) {
        // ascending random rows, like a flip list
        std::vector<int> pick;
        for (int k = 0; k < nl; ++k) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            pick.push_back((int)(s % (uint64_t)n));
        }
        std::sort(pick.begin(), pick.end());
        for (int k = 0; k < nl; ++k) h[(size_t)r * nl + k] = (uint32_t)pick[k] | (1u << 16) | ((k & 1) << 17);
        if (argc > 5 && argv[5][0] == 's' && r > 0)  // "same": every pass reads the SAME rows (cache-hot after the first)
            for (int k = 0; k < nl; ++k) h[(size_t)r * nl + k] = h[k];
    }
    uint32_t* lists;
    CHECK(hipMalloc(&lists, h.size() * 4));
    CHECK(hipMemcpy(lists, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    double* out;
    CHECK(hipMalloc(&out, (size_t)W * 64 * 8 * 8 + 8192));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    struct V { const char* name; void (*k)(const float*, int, const uint32_t*, int, int, double*); };
    V vs[] = {{"quad U=8", axpy_quad<8, false>},   {"quad U=4", axpy_quad<4, false>},     {"quad U=16", axpy_quad<16, false>},
              {"quad U=2", axpy_quad<2, false>},   {"quad U=3", axpy_quad<3, false>},     {"quad U=6", axpy_quad<6, false>}, {"quad U=1", axpy_quad<1, false>},
              {"loads only U=4", axpy_loads<4>},
              {"quad U=8 nt", axpy_quad<8, true>}, {"quad f32-round U=8", axpy_quad_f32<8>}, {"loads only U=8", axpy_loads<8>},
              {"loads only U=16", axpy_loads<16>},
              {"rep R=8 U=16", axpy_rep<8, 16, 0>}, {"rep R=8 U=16 loads only", axpy_rep<8, 16, 1>}, {"rep R=8 U=16 arithmetic only", axpy_rep<8, 16, 2>},
              {"rep R=8 U=8", axpy_rep<8, 8, 0>}, {"rep R=8 U=32", axpy_rep<8, 32, 0>}, {"rep R=2 U=16", axpy_rep<2, 16, 0>},
              {"rep-quad R=8 U=4", axpy_rep_quad<4, 0>}, {"rep-quad R=8 U=8", axpy_rep_quad<8, 0>}, {"rep-quad R=8 U=8 loads only", axpy_rep_quad<8, 1>},
              {"rep-quad R=8 U=16", axpy_rep_quad<16, 0>},
              {"rep2 readlane U=16", axpy_rep2<16, 0, 1>}, {"rep2 readlane U=16 arithmetic only", axpy_rep2<16, 2, 1>},
              {"rep2 movrels U=16", axpy_rep2<16, 0, 2>}, {"rep2 movrels U=16 arithmetic only", axpy_rep2<16, 2, 2>},
              {"rep2 pm2 U=16", axpy_rep2<16, 0, 3>}, {"rep2 pm2 U=16 arithmetic only", axpy_rep2<16, 2, 3>}, {"rep2 pm2 U=8", axpy_rep2<8, 0, 3>},
              {"rep2 pm2 U=24", axpy_rep2<24, 0, 3>},
              {"rep2 free multipliers U=16", axpy_rep2<16, 0, 4>}, {"rep2 free multipliers U=16 arithmetic only", axpy_rep2<16, 2, 4>},
              {"rep2 half scalar half vector U=16", axpy_rep2<16, 0, 5>}, {"rep2 half scalar half vector U=16 arithmetic only", axpy_rep2<16, 2, 5>},
              {"rep2 0 scalar 8 vector", axpy_rep2<16, 0, 6, 0>}, {"rep2 0 scalar 8 vector arithmetic only", axpy_rep2<16, 2, 6, 0>},
              {"rep2 2 scalar 6 vector", axpy_rep2<16, 0, 6, 2>}, {"rep2 2 scalar 6 vector arithmetic only", axpy_rep2<16, 2, 6, 2>},
              {"rep2 3 scalar 5 vector", axpy_rep2<16, 0, 6, 3>}, {"rep2 3 scalar 5 vector arithmetic only", axpy_rep2<16, 2, 6, 3>},
              {"rep2 4 scalar 4 vector", axpy_rep2<16, 0, 6, 4>}, {"rep2 4 scalar 4 vector arithmetic only", axpy_rep2<16, 2, 6, 4>},
              {"rep2 5 scalar 3 vector", axpy_rep2<16, 0, 6, 5>}, {"rep2 5 scalar 3 vector arithmetic only", axpy_rep2<16, 2, 6, 5>},
              {"rep2 readlane U=8", axpy_rep2<8, 0, 1>}, {"rep2 movrels U=8", axpy_rep2<8, 0, 2>}};
    if (argc > 5 && argv[5][0] == 'c') {  // calibration of the byte counters: ONE launch of the kernel k2_own runs, reading W x nl x reps x 256 bytes
        axpy_quad<4, false><<<W, 1024>>>(J, n, lists, nl, reps, out);
        CHECK(hipDeviceSynchronize());
        printf("calibration launch: %.0f bytes of row segments\n", (double)W * nl * reps * 256.0);
        return 0;
    }
    for (auto& v : vs) {
        v.k<<<W, 1024>>>(J, n, lists, nl, 2, out);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        v.k<<<W, 1024>>>(J, n, lists, nl, reps, out);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / reps;
        printf("%-22s n=%d nl=%d W=%d: %.2f us per pass, %.1f GB/s per CU, %.2f TB/s\n", v.name, n, nl, W, us, nl * 256.0 / us * 1e-3,
               (double)W * nl * 256.0 / us * 1e-6);
    }
    return 0;
}
