// f64 multiply-add issue rate on one CU's SIMDs: 16 waves per workgroup, 8 independent chains per lane
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
template <int KIND>
__global__ __launch_bounds__(1024) void k(double* out, int iters, double a, float xf, int code) {
    double acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = threadIdx.x * 1e-3 + i;
    double v = (double)xf + threadIdx.x;
    float vf = xf + threadIdx.x;
    float accf[8];
    for (int i = 0; i < 8; ++i) accf[i] = threadIdx.x * 1e-3f + i;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {  // vector operands
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = fma(v, v, acc[i]);
        } else if (KIND == 1) {  // scalar multiplier
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = fma(a, v, acc[i]);
        } else if (KIND == 2) {  // add only
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = acc[i] + v;
        } else if (KIND == 3) {  // f32 fma
#pragma unroll
            for (int i = 0; i < 8; ++i) accf[i] = fmaf(vf, vf, accf[i]);
        } else if (KIND == 4) {  // cvt i32 -> f64 (scalar source) + fma
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const double sg = (double)(((int)((unsigned)(code + it) << (14 - 2 * i))) >> 30);
                acc[i] = fma(sg, v, acc[i]);
            }
        } else if (KIND == 5) {  // cvt f32 -> f64 only
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] += (double)(vf + (float)i);
        }
    }
    const long long t1 = clock64();
    double z = 0;
    for (int i = 0; i < 8; ++i) z += acc[i] + accf[i];
    out[blockIdx.x * 1024 + threadIdx.x] = z;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[gridDim.x * 1024] = (double)(t1 - t0);
}
int main() {
    double* d;
    const int grid = 256, iters = 4096;
    hipMalloc(&d, (grid * 1024 + 1) * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const char* names[] = {"f64 fma, vector operands", "f64 fma, scalar multiplier", "f64 add", "f32 fma", "cvt_f64_i32 + f64 fma", "f32 add + cvt_f64_f32 + f64 add"};
    void (*ks[])(double*, int, double, float, int) = {k<0>, k<1>, k<2>, k<3>, k<4>, k<5>};
    for (int kind = 0; kind < 6; ++kind) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            ks[kind]<<<grid, 1024>>>(d, iters, 1.0000001, 0.5f, 0x5A5A0000);
            hipEventRecord(e1);
            hipDeviceSynchronize();
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double ticks;
        hipMemcpy(&ticks, d + grid * 1024, 8, hipMemcpyDeviceToHost);
        // per SIMD: 4 waves x 8 ops per iteration
        printf("%-36s %.3f ms: %.2f ns per wave-instruction and SIMD (%.1f clock64 ticks per iteration of 8)\n", names[kind], ms,
               ms * 1e6 / ((double)iters * 8 * 4), ticks / iters);
    }
    return 0;
}
