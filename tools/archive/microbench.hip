// microbench.hip -- instruction-rate probes that size the K1 kernel design (not product code).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench tools/microbench.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct u32x4 { uint32_t x, y, z, w; };
__device__ __forceinline__ u32x4 philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return u32x4{c0, c1, c2, c3};
}

template <int OP>
__global__ __launch_bounds__(256) void probe(uint32_t* out, int iters, uint32_t seed) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = t * 2654435761u + i * 40503u + seed;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) a[i] = a[i] + (a[(i + 1) & 7] ^ 0x9E3779B9u);                       // v_xad / add+xor
            if (OP == 1) a[i] = a[i] * 0xD2511F53u + 12345u;                                  // 32-bit mul lo (mad)
            if (OP == 2) { uint64_t p = (uint64_t)a[i] * 0xD2511F53u; a[i] = (uint32_t)(p >> 32) ^ (uint32_t)p; }  // mad_u64
            if (OP == 3) a[i] = __builtin_amdgcn_perm(a[i], a[(i + 1) & 7], 0x07020500u + (a[(i+2)&7] & 0x03030303u)); // v_perm
            if (OP == 4) { typedef short v2s __attribute__((ext_vector_type(2)));
                           v2s x = __builtin_bit_cast(v2s, a[i]), y = __builtin_bit_cast(v2s, a[(i + 1) & 7]);
                           v2s d = __builtin_elementwise_sub_sat(x, y); a[i] = __builtin_bit_cast(uint32_t, d); }     // v_pk_sub_i16 clamp
            if (OP == 5) a[i] = __builtin_amdgcn_alignbyte(a[i], a[(i + 1) & 7], 1);          // v_alignbyte
            if (OP == 6) a[i] = (a[i] & a[(i + 1) & 7]) | (a[(i + 2) & 7] >> 3);              // and_or / shifts
            if (OP == 7) a[i] = __umul24(a[i], a[(i+1)&7]) + a[(i+2)&7];      // v_mad_u32_u24
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) r ^= a[i];
    out[t] = r;
}

__global__ __launch_bounds__(256) void probe_philox(uint32_t* out, int iters, uint32_t seed) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        u32x4 r = philox(t, it, seed, 0, 0x1234567u, 0x89abcdefu);
        acc ^= r.x ^ r.y ^ r.z ^ r.w;
    }
    out[t] = acc;
}

// two interleaved philox streams (more ILP)
__global__ __launch_bounds__(256) void probe_philox2(uint32_t* out, int iters, uint32_t seed) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (int it = 0; it < iters; it += 2) {
        u32x4 r = philox(t, it, seed, 0, 0x1234567u, 0x89abcdefu);
        u32x4 q = philox(t, it + 1, seed, 0, 0x1234567u, 0x89abcdefu);
        acc ^= r.x ^ r.y ^ r.z ^ r.w ^ q.x ^ q.y ^ q.z ^ q.w;
    }
    out[t] = acc;
}

__global__ __launch_bounds__(256) void probe_lds(uint32_t* out, int iters) {
    __shared__ uint32_t buf[256 * 16];
    uint32_t t = threadIdx.x;
    for (int i = 0; i < 16; ++i) buf[t + 256 * i] = t * 31 + i;
    __syncthreads();
    uint4 acc = make_uint4(0, 0, 0, 0);
    const uint4* b4 = reinterpret_cast<const uint4*>(buf);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint4 v = b4[(t + 64 * i + it) & 1023];
            acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w;
        }
    }
    out[blockIdx.x * blockDim.x + t] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}

__global__ __launch_bounds__(256) void copy16(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = in[i];
}

template <typename F>
static float time_ms(F f, int reps = 5) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    f();
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s %s CUs=%d clock=%d kHz mem=%zu MiB l2=%d\n", prop.name, prop.gcnArchName, prop.multiProcessorCount,
           prop.clockRate, prop.totalGlobalMem >> 20, prop.l2CacheSize);
    const int blocks = prop.multiProcessorCount * 8, threads = 256;
    uint32_t* out;
    CHECK(hipMalloc(&out, (size_t)blocks * threads * 4));
    const int iters = 4096;
    const char* names[] = {"add+xor (2 ops)", "mul_lo+add (mad_u32?)", "mad_u64_u32 + xor", "v_perm_b32 (+and,+add)", "v_pk_sub_i16 clamp",
                           "v_alignbyte", "and, shr, or (3 ops)", "mul_u24+add"};
    double lanes = (double)blocks * threads;
#define RUN(OP) { float ms = time_ms([&] { probe<OP><<<blocks, threads>>>(out, iters, 1); }); \
        printf("%-28s %8.3f ms  %8.2f T statements/s\n", names[OP], ms, lanes * iters * 8 / ms / 1e9); }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7)
    { float ms = time_ms([&] { probe_philox<<<blocks, threads>>>(out, 1024, 1); });
      printf("%-28s %8.3f ms  %8.2f G philox calls/s  (%.2f T u32/s)\n", "philox4x32-10", ms, lanes * 1024 / ms / 1e6, lanes * 1024 * 4 / ms / 1e9); }
    { float ms = time_ms([&] { probe_philox2<<<blocks, threads>>>(out, 1024, 1); });
      printf("%-28s %8.3f ms  %8.2f G philox calls/s\n", "philox4x32-10 x2 ILP", ms, lanes * 1024 / ms / 1e6); }
    { float ms = time_ms([&] { probe_lds<<<blocks, threads>>>(out, 4096); });
      printf("%-28s %8.3f ms  %8.2f TB/s LDS ds_read_b128\n", "lds read b128", ms, lanes * 4096 * 4 * 16 / ms / 1e9); }
    for (size_t mb : {16, 64, 256, 1024, 4096}) {
        size_t n = mb * 1024 * 1024 / 16;
        uint4 *a, *b;
        CHECK(hipMalloc(&a, n * 16)); CHECK(hipMalloc(&b, n * 16));
        CHECK(hipMemset(a, 1, n * 16));
        float ms = time_ms([&] { copy16<<<prop.multiProcessorCount * 8, 256>>>(a, b, n); }, 10);
        printf("copy %5zu MiB: %8.3f ms  %7.2f TB/s (read+write)\n", mb, ms, 2.0 * n * 16 / ms / 1e9);
        hipFree(a); hipFree(b);
    }
    hipFree(out);
    return 0;
}
