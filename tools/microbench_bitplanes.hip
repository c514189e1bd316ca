// microbench_bitplanes.hip -- SURVEY 8(f3), first half, settled by measurement: what would the K1 pair loop pay for colour planes of
// 1 bit per site instead of 4 (the nibble planes of csrc/ising2d_tiled.hip)?  Everything else in the loop is the same in both forms
// (Philox, the v_perm threshold look-up on a BYTE count per site, the packed 16-bit compares); what differs is
//   (a) from the planes to the per-site neighbour count as a byte index, and
//   (b) from the accept flags (one byte per site, 0 / 1) back to the plane format.
// Both forms are written out here for 32 sites of one colour in one row (4 octets: four dwords of a nibble plane, one dword of a bit
// plane) and timed in a dependent loop; the instruction counts per 32 sites come from the ISA (hipcc -S, tools/microbench_bitplanes.s)
// and are printed next to the cycles.
//   hipcc --offload-arch=gfx950 -O3 -o tools/microbench_bitplanes tools/microbench_bitplanes.hip && tools/microbench_bitplanes
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x)                                                    \
    do {                                                            \
        hipError_t e_ = (x);                                        \
        if (e_ != hipSuccess) {                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
            exit(1);                                                \
        }                                                           \
    } while (0)

// ---------------------------------------------------------------- nibble planes: site j < 4 of an octet in the low nibble of byte j,
// site j >= 4 in the high nibble of byte j - 4 (csrc/ising2d_tiled.hip, NIB form)
struct Nib32 {
    uint32_t up[4], dn[4], ctr[4];  // rows above / below (same columns, other colour plane) and the other colour's row itself
    uint32_t edge;                  // the neighbouring octet's edge nibble for the horizontal shift
};
// (a) counts as bytes: 8 dwords (sites 0..3 and 4..7 of each octet)
static __device__ __forceinline__ void nib_counts(const Nib32& in, uint32_t out[8]) {
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        // horizontal neighbours: the other colour's row (same octet = one neighbour, the octet shifted by one site = the other)
        const uint32_t nxt = o < 3 ? in.ctr[o + 1] : in.edge;
        const uint32_t sh = __builtin_amdgcn_alignbit(nxt, in.ctr[o], 4);  // sites shifted by one nibble
        const uint32_t c = in.up[o] + in.dn[o] + in.ctr[o] + sh;           // nibble-wise sums (<= 4: no carries)
        out[2 * o] = c & 0x0F0F0F0Fu;
        out[2 * o + 1] = (c >> 4) & 0x0F0F0F0Fu;
    }
}
// (b) accept flags (bytes 0 / 1, 8 dwords) back to nibble planes (4 dwords)
static __device__ __forceinline__ void nib_pack(const uint32_t acc[8], uint32_t out[4]) {
#pragma unroll
    for (int o = 0; o < 4; ++o) out[o] = acc[2 * o] | (acc[2 * o + 1] << 4);
}

// ---------------------------------------------------------------- bit planes: 32 sites of a colour in a row = one dword
struct Bit32 {
    uint32_t up, dn, ctr, edge;  // edge: the next dword of the other colour's row (its bit 0 shifts in)
};
// (a) counts as bytes: the 3-bit count bit-sliced (full adders on whole dwords), then every 4 sites spread to 4 bytes
static __device__ __forceinline__ void bit_counts(const Bit32& in, uint32_t out[8]) {
    const uint32_t sh = __builtin_amdgcn_alignbit(in.edge, in.ctr, 1);
    const uint32_t a = in.up, b = in.dn, c = in.ctr, d = sh;
    const uint32_t u = a ^ b ^ c;                  // (v_bitop3)
    const uint32_t t = (a & b) | (c & (a ^ b));    // carry of a + b + c (v_bitop3)
    const uint32_t s0 = u ^ d, c2 = u & d;
    const uint32_t s1 = t ^ c2, s2 = t & c2;       // count = s0 + 2 s1 + 4 s2
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        // 4 bits of each plane -> 4 bytes (bit i of the nibble to bit 0 of byte i): (x * 0x00204081) & 0x01010101
        const uint32_t n0 = (s0 >> (4 * q)) & 0xFu, n1 = (s1 >> (4 * q)) & 0xFu, n2 = (s2 >> (4 * q)) & 0xFu;
        const uint32_t b0 = (n0 * 0x00204081u) & 0x01010101u;
        const uint32_t b1 = (n1 * 0x00204081u) & 0x01010101u;
        const uint32_t b2 = (n2 * 0x00204081u) & 0x01010101u;
        out[q] = b0 | (b1 << 1) | (b2 << 2);
    }
}
// (b) accept flags (bytes 0 / 1, 8 dwords) back to one dword of bits
static __device__ __forceinline__ uint32_t bit_pack(const uint32_t acc[8]) {
    uint32_t r = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) r |= ((acc[q] * 0x08040201u) >> 24 & 0xFu) << (4 * q);  // bytes 0/1 -> 4 bits
    return r;
}

// the part both forms share stands in as one v_perm look-up per count dword (so that the counts are really used as byte indices)
static __device__ __forceinline__ uint32_t lookup(uint32_t cnt, uint32_t table_lo, uint32_t table_hi) {
    return __builtin_amdgcn_perm(table_hi, table_lo, cnt) & 0x01010101u;
}

__global__ void k_nib(uint32_t* out, int iters, uint32_t seed) {
    Nib32 in;
    for (int o = 0; o < 4; ++o) {
        in.up[o] = (seed * (o + 1)) & 0x11111111u;
        in.dn[o] = (seed * (o + 5)) & 0x11111111u;
        in.ctr[o] = (seed * (o + 9) + threadIdx.x) & 0x11111111u;
    }
    in.edge = seed & 0x11111111u;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        uint32_t cnt[8], acc[8], pl[4];
        nib_counts(in, cnt);
        for (int q = 0; q < 8; ++q) acc[q] = lookup(cnt[q], 0x01000100u, 0x00010001u);
        nib_pack(acc, pl);
        for (int o = 0; o < 4; ++o) in.ctr[o] = pl[o];  // the new plane is the next iteration's neighbour row
        in.edge ^= pl[0];
    }
    const long long t1 = clock64();
    out[threadIdx.x + blockIdx.x * blockDim.x] = in.ctr[0] ^ in.ctr[1] ^ in.ctr[2] ^ in.ctr[3];
    if (threadIdx.x == 0 && blockIdx.x == 0) out[gridDim.x * blockDim.x] = (uint32_t)(t1 - t0);
}

__global__ void k_bit(uint32_t* out, int iters, uint32_t seed) {
    Bit32 in;
    in.up = seed * 3u;
    in.dn = seed * 7u;
    in.ctr = seed * 11u + threadIdx.x;
    in.edge = seed;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        uint32_t cnt[8], acc[8];
        bit_counts(in, cnt);
        for (int q = 0; q < 8; ++q) acc[q] = lookup(cnt[q], 0x01000100u, 0x00010001u);
        in.ctr = bit_pack(acc);
        in.edge ^= in.ctr;
    }
    const long long t1 = clock64();
    out[threadIdx.x + blockIdx.x * blockDim.x] = in.ctr;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[gridDim.x * blockDim.x] = (uint32_t)(t1 - t0);
}

int main(int argc, char** argv) {
    const int iters = 4096, grid = 1024, block = 256;  // every SIMD full: 4 waves per SIMD
    uint32_t* d;
    CHECK(hipMalloc(&d, (size_t)(grid * block + 1) * 4));
    uint32_t h = 0;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int form = 0; form < 2; ++form) {
        for (int rep = 0; rep < 2; ++rep) {
            CHECK(hipEventRecord(e0));
            if (form == 0) k_nib<<<grid, block>>>(d, iters, 12345u);
            else k_bit<<<grid, block>>>(d, iters, 12345u);
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
        }
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        CHECK(hipMemcpy(&h, d + grid * block, 4, hipMemcpyDeviceToHost));
        // per wave and iteration (32 sites): cycles of one wave (4 waves share a SIMD) and chip-wide time per 32 sites and lane
        printf("%s planes: %.1f clock64 ticks per iteration (32 sites per lane) in wave 0; kernel %.3f ms = %.2f ps per site chip-wide\n",
               form == 0 ? "nibble" : "1-bit ", (double)h / iters, ms, ms * 1e9 / ((double)grid * block * iters * 32));
    }
    return 0;
}
