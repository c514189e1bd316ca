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
    }
    uint32_t* lists;
    CHECK(hipMalloc(&lists, h.size() * 4));
    CHECK(hipMemcpy(lists, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    double* out;
    CHECK(hipMalloc(&out, (size_t)W * 64 * 8 + 8192));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    struct V { const char* name; void (*k)(const float*, int, const uint32_t*, int, int, double*); };
    V vs[] = {{"quad U=8", axpy_quad<8, false>},   {"quad U=4", axpy_quad<4, false>},     {"quad U=16", axpy_quad<16, false>},
              {"quad U=2", axpy_quad<2, false>},   {"quad U=3", axpy_quad<3, false>},     {"quad U=6", axpy_quad<6, false>}, {"quad U=1", axpy_quad<1, false>},
              {"loads only U=4", axpy_loads<4>},
              {"quad U=8 nt", axpy_quad<8, true>}, {"quad f32-round U=8", axpy_quad_f32<8>}, {"loads only U=8", axpy_loads<8>},
              {"loads only U=16", axpy_loads<16>}};
    if (argc > 5) {  // calibration of the byte counters: ONE launch of the kernel k2_own runs, reading W x nl x reps x 256 bytes
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
