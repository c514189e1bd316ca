// tsu_common.h -- shared internals of libtsu_hip.so (gfx950 only; no CUDA/HIP dual paths).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <unordered_map>

#include "../../include/tsu_hip.h"

struct tsu_ctx {
    int device;
    hipStream_t stream;
    hipEvent_t ev0, ev1;
    char err[512];
    int cus;
    // side streams for batches of independent lattices (tsu_ising2d_sweep_batch), created on first use
    hipStream_t pool[8];
    hipEvent_t pool_ev[8];
    hipEvent_t fork_ev;
    int pool_n;
    int in_batch;  // inside tsu_ising2d_sweep_batch: the batch itself bounds how many co-resident grids are in flight
    // per-DEVICE facts about kernels (a function attribute and an occupancy figure belong to the device they were set /
    // asked on): kept in the context, never in process-wide statics -- a second context on another GPU sets its own
    std::unordered_map<const void*, int> lds_attr;    // kernel -> MaxDynamicSharedMemorySize already granted
    std::unordered_map<uint64_t, int> blocks_per_cu;  // hash(kernel, threads, lds) -> occupancy answer
};

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (context, kernel); cached occupancy query
hipError_t tsu_func_allow_lds(tsu_ctx* ctx, const void* fn, int bytes);
hipError_t tsu_func_blocks_per_cu(tsu_ctx* ctx, const void* fn, int threads, size_t lds_bytes, int* per_cu);

// Every extern "C" entry point runs on its context's device whatever the caller's current device is (torch.cuda.set_device,
// a second context, ...), and leaves the caller's current device as it found it.
struct tsu_device_guard {
    int prev = -1;
    bool switched = false;
    explicit tsu_device_guard(const tsu_ctx* ctx) {
        if (!ctx) return;
        if (hipGetDevice(&prev) == hipSuccess && prev != ctx->device) switched = hipSetDevice(ctx->device) == hipSuccess;
    }
    ~tsu_device_guard() {
        if (switched) (void)hipSetDevice(prev);
    }
    tsu_device_guard(const tsu_device_guard&) = delete;
    tsu_device_guard& operator=(const tsu_device_guard&) = delete;
};
#define TSU_ENTER(ctx_expr) tsu_device_guard tsu_guard__(ctx_expr)

// Kernels that synchronise inside the grid (k1_resident, k2_coop) need all their workgroups on the chip at once.  Two
// of them launched on different streams of this process could each get part of the chip and wait for the rest, so
// such launches are chained per device: begin() makes the stream wait for the previous such launch, end() records
// this one.  (Across processes nothing can be chained: see DESIGN.md, honest gaps.)
int tsu_grid_exclusive_begin(tsu_ctx* ctx);
int tsu_grid_exclusive_end(tsu_ctx* ctx);

// Launch of a kernel that synchronises inside its grid.  Default: hipLaunchCooperativeKernel, so that co-residency of the
// whole grid is the runtime's promise (it refuses a grid that cannot be resident) instead of this library's arithmetic.
// TSU_COOP_LAUNCH=0 selects an ordinary launch of the same grid (same residency in practice: the grid was sized from the
// occupancy query); that is also what is used while the rocprofiler-sdk tool library is loaded in the process, because on
// ROCm 7.2 hsa_shut_down() crashes at process exit after a cooperative launch under rocprofv3 (DESIGN.md section 8,
// profiles/r02_exit_crash_symbolised.txt) -- TSU_COOP_LAUNCH=1 forces the cooperative API even then.  Inside a batch of
// lattices (tsu_ising2d_sweep_batch: several small grids share the chip on side streams, their number bounded by the batch
// itself) the launches are ordinary: the runtime serialises cooperative launches on one queue (measured: 32 lattices of
// 512 x 512, 2500 sweeps: 248 ms against 87 ms; profiles/r02_coop_launch_cost.txt).  The same holds for slabs whose launches
// alternate with RCCL kernels on the stream (beside_rccl): a cooperative launch after every halo exchange cost 230 us per
// exchange (4096 x 4096 slab, 128 sweeps per exchange: 2.0e12 instead of 2.55e12 upd/s), an ordinary one 12.
bool tsu_use_coop_launch();
hipError_t tsu_launch_grid_sync(tsu_ctx* ctx, const void* kernel, dim3 grid, dim3 block, void* param_struct, size_t lds_bytes, hipStream_t stream,
                                bool beside_rccl = false);

extern thread_local char g_tsu_init_err[512];

int tsu_fail(tsu_ctx* ctx, int code, const char* fmt, ...);

#define TSU_HIP_TRY(ctx, expr)                                                                      \
    do {                                                                                            \
        hipError_t e__ = (expr);                                                                    \
        if (e__ != hipSuccess)                                                                      \
            return tsu_fail((ctx), e__ == hipErrorOutOfMemory ? TSU_E_NOMEM : TSU_E_HIP, "%s: %s (%s:%d)", #expr, \
                            hipGetErrorString(e__), __FILE__, __LINE__);                            \
    } while (0)

#define TSU_REQUIRE(ctx, cond, ...)                                   \
    do {                                                              \
        if (!(cond)) return tsu_fail((ctx), TSU_E_INVALID, __VA_ARGS__); \
    } while (0)

// ------------------------------------------------------------------ Philox4x32-10 (device + host)
#define TSU_PHILOX_M0 0xD2511F53u
#define TSU_PHILOX_M1 0xCD9E8D57u
#define TSU_PHILOX_W0 0x9E3779B9u
#define TSU_PHILOX_W1 0xBB67AE85u

// ctr[3] stream tags (low byte); bits 8.. carry the replica / chain-group id.  DESIGN.md "RNG stream contract".
enum : uint32_t {
    TSU_TAG_ISING_HI = 0,
    TSU_TAG_ISING_LO = 1,
    TSU_TAG_INIT = 2,
    TSU_TAG_LANGEVIN = 3,
    TSU_TAG_DENSE = 4,
    TSU_TAG_LANGEVIN_RESTART = 5
};

struct u32x4 {
    uint32_t x, y, z, w;
};

// a ^ b ^ c: gfx950 has v_bitop3_b32 (any 3-input boolean function, truth table 0x96 = XOR3), which halves the
// bitwise work of a Philox round (one instruction instead of two v_xor_b32)
__host__ __device__ __forceinline__ uint32_t tsu_xor3(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
#else
    return a ^ b ^ c;
#endif
}

__host__ __device__ __forceinline__ u32x4 tsu_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                                     uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)TSU_PHILOX_M0 * c0;
        uint64_t p1 = (uint64_t)TSU_PHILOX_M1 * c2;
        uint32_t n0 = tsu_xor3((uint32_t)(p1 >> 32), c1, k0);
        uint32_t n2 = tsu_xor3((uint32_t)(p0 >> 32), c3, k1);
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
        k0 += TSU_PHILOX_W0;
        k1 += TSU_PHILOX_W1;
    }
    return u32x4{c0, c1, c2, c3};
}
