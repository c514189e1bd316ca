// ising2d.hip -- K1 checkerboard heat-bath sweep, K4 observables, lattice handle (gfx950).
//
// Replaces the reference's dense-matrix path for IsingGrid (tsu/models/ising.py:320-361 builds an N x N
// float64 J; tsu/gibbs.py:128-162 walks it site by site) with a stencil on int8 +-1 spins.
//
// HBM layout: row-major int8, one spin per byte, row pitch a multiple of 256 B, pad bytes are 0.
// A 0 byte means "no spin here" (outside an open boundary): it adds nothing to the neighbour sum and
// lowers the site's degree, which is exactly how the reference's missing bonds act (ising.py:348-361).
// Buffer rows: [ghost rows above][owned rows][ghost rows below]; `base` points at owned row 0.
//
// RNG stream contract (oracle/tsu_oracle.c ora_ising2d_sweep is the bit-exact CPU twin):
//   half-sweep hs = 2*sweep + colour; a site (R, c) of colour (R + c) & 1 has compact index j = c >> 1,
//   octet o = j >> 3, slot m = j & 7; W = Philox4x32-10(ctr = (o, R, hs, TAG_HI | replica << 8), key = seed);
//   hi16 = (half (m & 1) of W[m >> 1]) ^ 0x8000; lo16 = the same half of the TAG_LO block; u = hi16 << 16 | lo16;
//   spin <- +1 iff u < table[deg * 5 + up].  lo16 is only evaluated when hi16 ties with the threshold's
//   top 16 bits (probability 2^-16 per site), which cannot change the outcome of the 32-bit comparison.
#include <vector>

#include "ising2d.h"

struct K1Params {
    int8_t* base;        // owned row 0 of the current buffer
    int8_t* out;         // owned row 0 of the destination buffer (== base for in-place kernels)
    long long pitch;
    int rows, cols;
    int r_lo, r_hi;      // local row range to update (may reach into ghost rows)
    long long row0, total_rows;
    int periodic, wrap_rows;
    int top_rows, bot_rows;  // how many rows exist above owned row 0 / below owned row rows-1 in the buffer
    uint32_t k0, k1, hs, tag_hi, tag_lo;
};

struct K1Table {
    uint64_t t[25];
};

static __device__ __forceinline__ long long global_row(const K1Params& p, int r) {
    long long gr = p.row0 + r;
    if (p.periodic) {
        gr %= p.total_rows;
        if (gr < 0) gr += p.total_rows;
    }
    return gr;
}

// does local row r (possibly a ghost row) exist in the buffer?
static __device__ __forceinline__ bool row_exists(const K1Params& p, int r) {
    return r >= -p.top_rows && r < p.rows + p.bot_rows;
}

// ------------------------------------------------------------------ one octet of one colour
// 16 consecutive columns of one row (8 sites of the half-sweep's colour): neighbour bytes, degree and up-count per
// site, one Philox block, thresholds from s_tbl, masked 16-byte store.  `row`, `up_row`, `dn_row` (NULL = absent) and
// `out_chunk` may point into global memory (k1_generic) or LDS (k1_small): the row layout is the same.
static __device__ __forceinline__ void k1_update_octet(const K1Params& p, const uint64_t* s_tbl, const int8_t* row,
                                                       const int8_t* up_row, const int8_t* dn_row, int8_t* out_chunk, int q,
                                                       long long gr, int par, int nchunks, uint32_t hs) {
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
    uint4 cv = *reinterpret_cast<const uint4*>(row + 16 * q);
    uint4 uv = up_row ? *reinterpret_cast<const uint4*>(up_row + 16 * q) : zero4;
    uint4 dv = dn_row ? *reinterpret_cast<const uint4*>(dn_row + 16 * q) : zero4;
    uint64_t clo = (uint64_t)cv.x | ((uint64_t)cv.y << 32), chi = (uint64_t)cv.z | ((uint64_t)cv.w << 32);
    uint64_t ulo = (uint64_t)uv.x | ((uint64_t)uv.y << 32), uhi = (uint64_t)uv.z | ((uint64_t)uv.w << 32);
    uint64_t dlo = (uint64_t)dv.x | ((uint64_t)dv.y << 32), dhi = (uint64_t)dv.z | ((uint64_t)dv.w << 32);

    // bytes just outside the chunk: column 16q-1 and column 16q+16
    uint32_t prev = 0, next = 0;
    if (q > 0) prev = (uint8_t)row[16 * q - 1];
    else if (p.periodic) prev = (uint8_t)row[p.cols - 1];
    if (16 * q + 16 < p.cols) next = (uint8_t)row[16 * q + 16];
    else if (p.periodic && 16 * q + 16 == p.cols) next = (uint8_t)row[0];
    // periodic lattice whose width is not a multiple of 16: the right neighbour of the last column is
    // column 0, which sits inside the last chunk at byte (cols & 15); patch it in (and out again below)
    int patch = (p.periodic && (p.cols & 15) && q == nchunks - 1) ? (p.cols & 15) : -1;
    if (patch >= 0) {
        uint64_t b = (uint64_t)(uint8_t)row[0];
        if (patch < 8) clo |= b << (8 * patch);
        else chi |= b << (8 * (patch - 8));
    }

    // X = bytes [par-1 .. par+16] of the row around the chunk: sites at X[2m+1], left X[2m], right X[2m+2]
    uint64_t X0, X1, X2, U0, U1, D0, D1;
    if (par == 0) {
        X0 = (clo << 8) | prev;
        X1 = (chi << 8) | (clo >> 56);
        X2 = chi >> 56;
        U0 = ulo; U1 = uhi; D0 = dlo; D1 = dhi;
    } else {
        X0 = clo; X1 = chi; X2 = next;
        U0 = (ulo >> 8) | (uhi << 56); U1 = uhi >> 8;
        D0 = (dlo >> 8) | (dhi << 56); D1 = dhi >> 8;
    }

    u32x4 w = tsu_philox((uint32_t)q, (uint32_t)gr, hs, p.tag_hi, p.k0, p.k1);
    uint32_t wv[4] = {w.x, w.y, w.z, w.w};
    bool have_lo = false;
    uint32_t lv[4] = {0, 0, 0, 0};

    uint64_t N0 = 0, N1 = 0, M0 = 0, M1 = 0;  // new bytes / write mask, sites at bytes 0,2,4,6 of each half
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        int c = 16 * q + 2 * m + par;
        // shifted-window byte positions
        const int ps = 2 * m + 1, pl = 2 * m, pr = 2 * m + 2;
        int sl = (int8_t)((pl < 8 ? X0 >> (8 * pl) : X1 >> (8 * (pl - 8))) & 0xFF);
        int sr = (int8_t)((pr < 8 ? X0 >> (8 * pr) : (pr < 16 ? X1 >> (8 * (pr - 8)) : X2)) & 0xFF);
        int su = (int8_t)((m < 4 ? U0 >> (16 * m) : U1 >> (16 * (m - 4))) & 0xFF);
        int sd = (int8_t)((m < 4 ? D0 >> (16 * m) : D1 >> (16 * (m - 4))) & 0xFF);
        (void)ps;
        int deg = (sl != 0) + (sr != 0) + (su != 0) + (sd != 0);
        int up = (sl > 0) + (sr > 0) + (su > 0) + (sd > 0);
        uint64_t thr = s_tbl[deg * 5 + up];
        uint32_t hi = ((wv[m >> 1] >> (16 * (m & 1))) & 0xFFFFu) ^ 0x8000u;
        uint32_t thi = (uint32_t)(thr >> 16);  // 0..65536
        bool accept = hi < thi;
        if (hi == thi) {  // tie on the top 16 bits: evaluate the low half (rare)
            if (!have_lo) {
                u32x4 l = tsu_philox((uint32_t)q, (uint32_t)gr, hs, p.tag_lo, p.k0, p.k1);
                lv[0] = l.x; lv[1] = l.y; lv[2] = l.z; lv[3] = l.w;
                have_lo = true;
            }
            uint32_t lo = (lv[m >> 1] >> (16 * (m & 1))) & 0xFFFFu;
            accept = (((uint64_t)hi << 16) | lo) < thr;
        }
        uint64_t nb = accept ? 0x01ull : 0xFFull;
        uint64_t mk = (c < p.cols) ? 0xFFull : 0ull;
        if (m < 4) { N0 |= nb << (16 * m); M0 |= mk << (16 * m); }
        else { N1 |= nb << (16 * (m - 4)); M1 |= mk << (16 * (m - 4)); }
    }
    if (par) { N0 <<= 8; N1 <<= 8; M0 <<= 8; M1 <<= 8; }
    if (patch >= 0) {  // remove the patched-in copy of column 0 again
        if (patch < 8) clo &= ~(0xFFull << (8 * patch));
        else chi &= ~(0xFFull << (8 * (patch - 8)));
    }
    clo = (clo & ~M0) | (N0 & M0);
    chi = (chi & ~M1) | (N1 & M1);
    uint4 ov = make_uint4((uint32_t)clo, (uint32_t)(clo >> 32), (uint32_t)chi, (uint32_t)(chi >> 32));
    *reinterpret_cast<uint4*>(out_chunk) = ov;
}


// ------------------------------------------------------------------ generic kernel: one colour per launch
// thread = one octet = 16 consecutive columns of one row (8 sites of the launch colour).
__global__ __launch_bounds__(256) void k1_generic(K1Params p, K1Table tbl, int colour) {
    __shared__ uint64_t s_tbl[25];
    int tid = threadIdx.y * 64 + threadIdx.x;
    if (tid < 25) s_tbl[tid] = tbl.t[tid];
    __syncthreads();

    int q = blockIdx.x * 64 + threadIdx.x;          // octet / 16-byte chunk index
    int r = p.r_lo + blockIdx.y * 4 + threadIdx.y;  // local row
    int nchunks = (p.cols + 15) >> 4;
    if (q >= nchunks || r >= p.r_hi) return;

    long long gr = global_row(p, r);
    int par = (int)((gr + colour) & 1);  // column parity of this colour in this row
    const int8_t* row = p.base + (long long)r * p.pitch;

    // vertical neighbour rows: buffer row, wrapped row, or absent (zeros)
    const int8_t* up_row = nullptr;
    const int8_t* dn_row = nullptr;
    if (p.wrap_rows) {
        up_row = p.base + (long long)(r == 0 ? p.rows - 1 : r - 1) * p.pitch;
        dn_row = p.base + (long long)(r == p.rows - 1 ? 0 : r + 1) * p.pitch;
    } else {
        if (row_exists(p, r - 1)) up_row = row - p.pitch;
        if (row_exists(p, r + 1)) dn_row = row + p.pitch;
    }
    k1_update_octet(p, s_tbl, row, up_row, dn_row, p.out + (long long)r * p.pitch + 16 * q, q, gr, par, nchunks, p.hs);
}

// ------------------------------------------------------------------ small lattices: one workgroup, one launch
// The whole lattice (not a slab) lives in LDS with the HBM row layout; all n_sweeps run inside one launch with one
// workgroup barrier per half-sweep (in place: a half-sweep only writes its own colour, which no site of that colour
// reads).  Same octet update, same Philox counters, hence the same results as k1_generic -- without two launches per
// sweep, which is all a 32 x 32 lattice (BASELINE configs[0]) costs there.
__global__ __launch_bounds__(1024) void k1_small(K1Params p, K1Table tbl, uint32_t sweep0, int n_sweeps) {
    extern __shared__ int8_t s_lat[];
    __shared__ uint64_t s_tbl[25];
    const int tid = threadIdx.x, nt = blockDim.x;
    const int nchunks = (p.cols + 15) >> 4;
    const int lp = 16 * nchunks;  // LDS row pitch: whole chunks, pad bytes 0 like the HBM rows
    if (tid < 25) s_tbl[tid] = tbl.t[tid];
    for (int t = tid; t < p.rows * nchunks; t += nt) {
        const int r = t / nchunks, q = t - r * nchunks;
        *reinterpret_cast<uint4*>(s_lat + r * lp + 16 * q) = *reinterpret_cast<const uint4*>(p.base + (long long)r * p.pitch + 16 * q);
    }
    __syncthreads();
    for (int hsi = 0; hsi < 2 * n_sweeps; ++hsi) {
        const int colour = hsi & 1;
        const uint32_t hs = 2u * (sweep0 + (uint32_t)(hsi >> 1)) + (uint32_t)colour;
        for (int t = tid; t < p.rows * nchunks; t += nt) {
            const int r = t / nchunks, q = t - r * nchunks;
            const int par = (r + colour) & 1;
            int8_t* row = s_lat + r * lp;
            const int8_t* up_row = r > 0 ? row - lp : (p.periodic ? s_lat + (p.rows - 1) * lp : nullptr);
            const int8_t* dn_row = r < p.rows - 1 ? row + lp : (p.periodic ? s_lat : nullptr);
            k1_update_octet(p, s_tbl, row, up_row, dn_row, row + 16 * q, q, (long long)r, par, nchunks, hs);
        }
        __syncthreads();
    }
    for (int t = tid; t < p.rows * nchunks; t += nt) {
        const int r = t / nchunks, q = t - r * nchunks;
        *reinterpret_cast<uint4*>(p.out + (long long)r * p.pitch + 16 * q) = *reinterpret_cast<const uint4*>(s_lat + r * lp + 16 * q);
    }
}

// k1_small for many lattices of one shape: workgroup b sweeps lattice b (own buffer, thresholds, seed, counters)
struct K1BatchItem {
    K1Params p;
    K1Table tbl;
    uint32_t sweep0;
};

__global__ __launch_bounds__(1024) void k1_small_batch(const K1BatchItem* __restrict__ items, int n_sweeps) {
    extern __shared__ int8_t s_lat[];
    __shared__ uint64_t s_tbl[25];
    __shared__ K1Params sp;
    const K1BatchItem& it = items[blockIdx.x];
    const int tid = threadIdx.x, nt = blockDim.x;
    if (tid == 0) sp = it.p;
    if (tid < 25) s_tbl[tid] = it.tbl.t[tid];
    __syncthreads();
    const K1Params& p = sp;
    const uint32_t sweep0 = it.sweep0;
    const int nchunks = (p.cols + 15) >> 4;
    const int lp = 16 * nchunks;
    for (int t = tid; t < p.rows * nchunks; t += nt) {
        const int r = t / nchunks, q = t - r * nchunks;
        *reinterpret_cast<uint4*>(s_lat + r * lp + 16 * q) = *reinterpret_cast<const uint4*>(p.base + (long long)r * p.pitch + 16 * q);
    }
    __syncthreads();
    for (int hsi = 0; hsi < 2 * n_sweeps; ++hsi) {
        const int colour = hsi & 1;
        const uint32_t hs = 2u * (sweep0 + (uint32_t)(hsi >> 1)) + (uint32_t)colour;
        for (int t = tid; t < p.rows * nchunks; t += nt) {
            const int r = t / nchunks, q = t - r * nchunks;
            const int par = (r + colour) & 1;
            int8_t* row = s_lat + r * lp;
            const int8_t* up_row = r > 0 ? row - lp : (p.periodic ? s_lat + (p.rows - 1) * lp : nullptr);
            const int8_t* dn_row = r < p.rows - 1 ? row + lp : (p.periodic ? s_lat : nullptr);
            k1_update_octet(p, s_tbl, row, up_row, dn_row, row + 16 * q, q, (long long)r, par, nchunks, hs);
        }
        __syncthreads();
    }
    for (int t = tid; t < p.rows * nchunks; t += nt) {
        const int r = t / nchunks, q = t - r * nchunks;
        *reinterpret_cast<uint4*>(p.out + (long long)r * p.pitch + 16 * q) = *reinterpret_cast<const uint4*>(s_lat + r * lp + 16 * q);
    }
}

// ------------------------------------------------------------------ K4: sum of spins, sum over bonds
// grid-stride over (row, 16-byte chunk); per-thread int32 partials (|partial| <= 48 per chunk), wave shuffle +
// LDS block reduce, ONE atomic pair per block (a 4096^2 lattice used to issue 262144 contended atomics).
__global__ __launch_bounds__(256) void k4_observables(K1Params p, long long* __restrict__ acc) {
    const int nchunks = (p.cols + 15) >> 4;
    const long long total = (long long)p.rows * nchunks;
    long long ss = 0, sb = 0;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
        const int r = (int)(t / nchunks), q = (int)(t - (long long)r * nchunks);
        const int8_t* row = p.base + (long long)r * p.pitch;
        const int8_t* dn_row = nullptr;
        if (p.wrap_rows) dn_row = p.base + (long long)(r == p.rows - 1 ? 0 : r + 1) * p.pitch;
        else if (row_exists(p, r + 1)) dn_row = row + p.pitch;
        uint4 cv = *reinterpret_cast<const uint4*>(row + 16 * q);
        uint4 dv = dn_row ? *reinterpret_cast<const uint4*>(dn_row + 16 * q) : make_uint4(0, 0, 0, 0);
        int next = 0;
        if (16 * q + 16 < p.cols) next = row[16 * q + 16];
        else if (p.periodic && 16 * q + 16 == p.cols) next = row[0];
        uint32_t cw[4] = {cv.x, cv.y, cv.z, cv.w}, dw[4] = {dv.x, dv.y, dv.z, dv.w};
        const int last = p.cols - 16 * q;  // valid columns in this chunk (>= 16 except in a ragged last chunk)
        const int wrap0 = (p.periodic && last < 16) ? (int)row[0] : 0;  // wrap bond of the last column, ragged chunk
        int cs = 0, cb = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int si = (int8_t)((cw[i >> 2] >> (8 * (i & 3))) & 0xFF);
            int sr = (i < 15) ? (int)(int8_t)((cw[(i + 1) >> 2] >> (8 * ((i + 1) & 3))) & 0xFF) : next;
            if (i + 1 == last) sr = (last < 16) ? wrap0 : sr;
            const int d = (int8_t)((dw[i >> 2] >> (8 * (i & 3))) & 0xFF);
            if (i < last) {
                cs += si;
                cb += si * sr + si * d;
            }
        }
        ss += cs;
        sb += cb;
    }
    for (int off = 32; off > 0; off >>= 1) {
        ss += __shfl_down(ss, off, 64);
        sb += __shfl_down(sb, off, 64);
    }
    __shared__ long long part[2][4];
    if ((threadIdx.x & 63) == 0) {
        part[0][threadIdx.x >> 6] = ss;
        part[1][threadIdx.x >> 6] = sb;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(reinterpret_cast<unsigned long long*>(acc), (unsigned long long)(part[0][0] + part[0][1] + part[0][2] + part[0][3]));
        atomicAdd(reinterpret_cast<unsigned long long*>(acc + 1), (unsigned long long)(part[1][0] + part[1][1] + part[1][2] + part[1][3]));
    }
}

// ------------------------------------------------------------------ init kernels
__global__ __launch_bounds__(256) void k_randomize(K1Params p, uint32_t tag) {
    int q = blockIdx.x * 64 + threadIdx.x;
    int r = p.r_lo + blockIdx.y * 4 + threadIdx.y;
    int nchunks = (p.cols + 15) >> 4;
    if (q >= nchunks || r >= p.r_hi) return;
    long long gr = global_row(p, r);
    u32x4 w = tsu_philox((uint32_t)(q >> 3), (uint32_t)gr, 0u, tag, p.k0, p.k1);
    uint32_t wv[4] = {w.x, w.y, w.z, w.w};
    uint32_t bits = (wv[(q & 7) >> 1] >> (16 * (q & 1))) & 0xFFFFu;
    uint32_t o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint32_t v = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            int i = 4 * k + b;
            uint32_t byte = ((bits >> i) & 1u) ? 0x01u : 0xFFu;
            if (16 * q + i >= p.cols) byte = 0;
            v |= byte << (8 * b);
        }
        o[k] = v;
    }
    *reinterpret_cast<uint4*>(p.base + (long long)r * p.pitch + 16 * q) = make_uint4(o[0], o[1], o[2], o[3]);
}

__global__ __launch_bounds__(256) void k_fill(K1Params p, int value) {
    int q = blockIdx.x * 64 + threadIdx.x;
    int r = p.r_lo + blockIdx.y * 4 + threadIdx.y;
    int nchunks = (p.cols + 15) >> 4;
    if (q >= nchunks || r >= p.r_hi) return;
    uint32_t o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint32_t v = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            uint32_t byte = (16 * q + 4 * k + b < p.cols) ? (uint32_t)(uint8_t)value : 0u;
            v |= byte << (8 * b);
        }
        o[k] = v;
    }
    *reinterpret_cast<uint4*>(p.base + (long long)r * p.pitch + 16 * q) = make_uint4(o[0], o[1], o[2], o[3]);
}

// ------------------------------------------------------------------ host side
static K1Params make_params(const tsu_ising2d* L, int8_t* buf) {
    K1Params p;
    p.base = buf + (size_t)L->ghost * L->pitch;
    p.out = p.base;
    p.pitch = (long long)L->pitch;
    p.rows = L->rows;
    p.cols = L->cols;
    p.r_lo = 0;
    p.r_hi = L->rows;
    p.row0 = L->row0;
    p.total_rows = L->total_rows;
    p.periodic = L->periodic;
    p.wrap_rows = L->wrap_rows;
    // ghost rows that correspond to real lattice rows
    if (L->ghost == 0) {
        p.top_rows = p.bot_rows = 0;
    } else if (L->periodic) {
        p.top_rows = p.bot_rows = L->ghost;
    } else {
        long long above = L->row0, below = L->total_rows - (L->row0 + L->rows);
        p.top_rows = (int)(above < L->ghost ? above : L->ghost);
        p.bot_rows = (int)(below < L->ghost ? below : L->ghost);
    }
    p.k0 = p.k1 = p.hs = p.tag_hi = p.tag_lo = 0;
    return p;
}

// k1_small: a whole lattice (not a slab) with at most one octet per thread of one workgroup (measured: 32 x 32 3.2 us
// per sweep against 11 us for two generic launches; at 256 x 256 the whole chip wins: 8 us against 30 us)
static bool small_supported(const tsu_ising2d* L) {
    if (L->ghost != 0 || L->total_rows != L->rows || L->row0 != 0) return false;
    const long long nchunks = (L->cols + 15) >> 4, tasks = (long long)L->rows * nchunks;
    return tasks <= 1024;
}

static dim3 grid_for(const tsu_ising2d* L, int nrows) {
    int nchunks = (L->cols + 15) >> 4;
    return dim3((unsigned)((nchunks + 63) / 64), (unsigned)((nrows + 3) / 4), 1);
}


extern "C" {

int tsu_ising2d_thresholds(double J, double h, double T, int mode, uint64_t table[25]) {
    if (!table || !(T > 0.0) || (mode != TSU_MODE_PHYSICAL && mode != TSU_MODE_COMPAT)) return TSU_E_INVALID;
    for (int deg = 0; deg <= 4; ++deg)
        for (int up = 0; up <= 4; ++up) {
            if (up > deg) {
                table[deg * 5 + up] = 0;
                continue;
            }
            // reference bit representation: field = 4J*up + bias (ising.py:138,148), p = sigmoid(field/T)
            double bias = (mode == TSU_MODE_COMPAT) ? (-2.0 * h + 2.0 * J * (double)deg) : (2.0 * h - 2.0 * J * (double)deg);
            double x = (4.0 * J * (double)up + bias) / T;
            double prob = x > 20.0 ? 1.0 : (x < -20.0 ? 0.0 : 1.0 / (1.0 + exp(-x)));  // gibbs.py:73-77
            table[deg * 5 + up] = (uint64_t)floor(prob * 4294967296.0 + 0.5);
        }
    return TSU_OK;
}

int tsu_ising2d_create_slab(tsu_ctx* ctx, int64_t total_rows, int cols, int periodic, int64_t row0, int rows, int ghost,
                            tsu_ising2d** out) {
    TSU_ENTER(ctx);
    if (!ctx || !out) return TSU_E_INVALID;
    *out = nullptr;
    TSU_REQUIRE(ctx, total_rows >= 1 && cols >= 1 && rows >= 1, "ising2d: rows/cols must be positive");
    TSU_REQUIRE(ctx, row0 >= 0 && row0 + rows <= total_rows, "ising2d: slab [%lld, %lld) outside lattice of %lld rows",
                (long long)row0, (long long)(row0 + rows), (long long)total_rows);
    TSU_REQUIRE(ctx, total_rows < (1ll << 32) && cols <= (1 << 30), "ising2d: lattice too large for 32-bit counters");
    TSU_REQUIRE(ctx, ghost >= 0 && (ghost % 2) == 0, "ising2d: ghost depth must be even and >= 0");
    bool whole = (rows == total_rows);
    TSU_REQUIRE(ctx, whole || ghost >= 2, "ising2d: a slab needs ghost >= 2");
    TSU_REQUIRE(ctx, ghost <= rows, "ising2d: ghost depth %d exceeds slab height %d", ghost, rows);
    if (periodic && ((total_rows & 1) || (cols & 1) || total_rows < 4 || cols < 4))
        return tsu_fail(ctx, TSU_E_UNSUPPORTED,
                        "ising2d: a periodic checkerboard needs even rows and cols >= 4 (got %lld x %d); use the dense path",
                        (long long)total_rows, cols);
    tsu_ising2d* L = new (std::nothrow) tsu_ising2d();
    if (!L) return tsu_fail(ctx, TSU_E_NOMEM, "ising2d: host allocation failed");
    L->ctx = ctx;
    L->total_rows = total_rows;
    L->row0 = row0;
    L->rows = rows;
    L->cols = cols;
    L->periodic = periodic ? 1 : 0;
    L->ghost = ghost;
    L->wrap_rows = (whole && ghost == 0 && periodic) ? 1 : 0;
    L->pitch = ((size_t)cols + 255) / 256 * 256;
    L->alloc[0] = L->alloc[1] = nullptr;
    L->cur = 0;
    L->have_table = 0;
    L->kernel = TSU_KERNEL_AUTO;
    L->sweeps_per_launch = 0;
    L->d_obs = nullptr;
    L->timed = 0;
    L->timing = 0;
    L->launches = 0;
    L->d_sync = nullptr;
    L->d_xbuf = nullptr;
    L->xbuf_cap = 0;
    L->xgen = 0;
    L->xsig = 0;
    L->d_batch = nullptr;
    L->batch_cap = 0;
    L->d_obs_batch = nullptr;
    L->obs_batch_cap = 0;
    L->sync_cap = 0;
    L->h_err = nullptr;
    size_t bytes = (size_t)(rows + 2 * ghost) * L->pitch;
    hipError_t e = hipMalloc(&L->alloc[0], bytes);
    if (e == hipSuccess) e = hipMemsetAsync(L->alloc[0], 0, bytes, ctx->stream);
    if (e == hipSuccess) e = hipMalloc(&L->d_obs, 2 * sizeof(int64_t));
    if (e == hipSuccess) e = hipEventCreate(&L->ev0);
    if (e == hipSuccess) e = hipEventCreate(&L->ev1);
    if (e != hipSuccess) {
        int rc = tsu_fail(ctx, e == hipErrorOutOfMemory ? TSU_E_NOMEM : TSU_E_HIP, "ising2d_create: %s (%zu bytes)",
                          hipGetErrorString(e), bytes);
        if (L->alloc[0]) (void)hipFree(L->alloc[0]);
        if (L->d_obs) (void)hipFree(L->d_obs);
        delete L;
        return rc;
    }
    *out = L;
    return TSU_OK;
}

int tsu_ising2d_create(tsu_ctx* ctx, int rows, int cols, int periodic, tsu_ising2d** out) {
    TSU_ENTER(ctx);
    return tsu_ising2d_create_slab(ctx, rows, cols, periodic, 0, rows, 0, out);
}

int tsu_ising2d_destroy(tsu_ising2d* L) {
    TSU_ENTER(L ? L->ctx : nullptr);
    if (!L) return TSU_OK;
    (void)hipStreamSynchronize(L->ctx->stream);
    for (int i = 0; i < 2; ++i)
        if (L->alloc[i]) (void)hipFree(L->alloc[i]);
    if (L->d_obs) (void)hipFree(L->d_obs);
    if (L->d_sync) (void)hipFree(L->d_sync);
    if (L->d_xbuf) (void)hipFree(L->d_xbuf);
    if (L->d_batch) (void)hipFree(L->d_batch);
    if (L->d_obs_batch) (void)hipFree(L->d_obs_batch);
    if (L->h_err) (void)hipHostFree(L->h_err);
    (void)hipEventDestroy(L->ev0);
    (void)hipEventDestroy(L->ev1);
    delete L;
    return TSU_OK;
}

int tsu_ising2d_set_spins(tsu_ising2d* L, const int8_t* host, int row_first, int n_rows) {
    TSU_ENTER(L ? L->ctx : nullptr);
    if (!L) return TSU_E_INVALID;
    tsu_ctx* ctx = L->ctx;
    TSU_REQUIRE(ctx, host && n_rows >= 0 && row_first >= -L->ghost && row_first + n_rows <= L->rows + L->ghost,
                "ising2d_set_spins: rows [%d, %d) outside [%d, %d)", row_first, row_first + n_rows, -L->ghost,
                L->rows + L->ghost);
    if (n_rows == 0) return TSU_OK;
    int8_t* dst = L->alloc[L->cur] + (size_t)(L->ghost + row_first) * L->pitch;
    TSU_HIP_TRY(ctx, hipMemcpy2DAsync(dst, L->pitch, host, (size_t)L->cols, (size_t)L->cols, (size_t)n_rows,
                                      hipMemcpyHostToDevice, ctx->stream));
    TSU_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return TSU_OK;
}

static int check_persist_error(tsu_ising2d* L) {
    if (L->h_err && *L->h_err) {
        *L->h_err = 0;
        return tsu_fail(L->ctx, TSU_E_HIP, "ising2d: tile-resident sweep kernel timed out waiting for a neighbouring tile (GPU shared?); results invalid");
    }
    return TSU_OK;
}

int tsu_ising2d_get_spins(tsu_ising2d* L, int8_t* host, int row_first, int n_rows) {
    TSU_ENTER(L ? L->ctx : nullptr);
    if (!L) return TSU_E_INVALID;
    tsu_ctx* ctx = L->ctx;
    TSU_REQUIRE(ctx, host && n_rows >= 0 && row_first >= -L->ghost && row_first + n_rows <= L->rows + L->ghost,
                "ising2d_get_spins: rows [%d, %d) outside [%d, %d)", row_first, row_first + n_rows, -L->ghost,
                L->rows + L->ghost);
    if (n_rows == 0) return TSU_OK;
    const int8_t* src = L->alloc[L->cur] + (size_t)(L->ghost + row_first) * L->pitch;
    TSU_HIP_TRY(ctx, hipMemcpy2DAsync(host, (size_t)L->cols, src, L->pitch, (size_t)L->cols, (size_t)n_rows,
                                      hipMemcpyDeviceToHost, ctx->stream));
    TSU_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return check_persist_error(L);
}

int tsu_ising2d_randomize(tsu_ising2d* L, uint64_t seed, uint32_t replica) {
    TSU_ENTER(L ? L->ctx : nullptr);
    if (!L) return TSU_E_INVALID;
    tsu_ctx* ctx = L->ctx;
    K1Params p = make_params(L, L->alloc[L->cur]);
    p.r_lo = -p.top_rows;
    p.r_hi = L->rows + p.bot_rows;
    p.k0 = (uint32_t)seed;
    p.k1 = (uint32_t)(seed >> 32);
    k_randomize<<<grid_for(L, p.r_hi - p.r_lo), dim3(64, 4, 1), 0, ctx->stream>>>(p, TSU_TAG_INIT | (replica << 8));
    TSU_HIP_TRY(ctx, hipGetLastError());
    return TSU_OK;
}

int tsu_ising2d_fill(tsu_ising2d* L, int8_t value) {
    TSU_ENTER(L ? L->ctx : nullptr);
    if (!L) return TSU_E_INVALID;
    tsu_ctx* ctx = L->ctx;
    TSU_REQUIRE(ctx, value == 1 || value == -1, "ising2d_fill: value must be +1 or -1");
    K1Params p = make_params(L, L->alloc[L->cur]);
    p.r_lo = -p.top_rows;
    p.r_hi = L->rows + p.bot_rows;
    k_fill<<<grid_for(L, p.r_hi - p.r_lo), dim3(64, 4, 1), 0, ctx->stream>>>(p, (int)value);
    TSU_HIP_TRY(ctx, hipGetLastError());
    return TSU_OK;
}

int tsu_ising2d_set_thresholds(tsu_ising2d* L, const uint64_t table[25]) {
    TSU_ENTER(L ? L->ctx : nullptr);
    if (!L) return TSU_E_INVALID;
    TSU_REQUIRE(L->ctx, table != nullptr, "ising2d_set_thresholds: table is NULL");
    for (int i = 0; i < 25; ++i) {
        TSU_REQUIRE(L->ctx, table[i] <= (1ull << 32), "ising2d_set_thresholds: entry %d exceeds 2^32", i);
        L->table[i] = table[i];
    }
    L->have_table = 1;
    return TSU_OK;
}

int tsu_ising2d_set_model(tsu_ising2d* L, double J, double h, double T, int mode) {
    TSU_ENTER(L ? L->ctx : nullptr);
    if (!L) return TSU_E_INVALID;
    TSU_REQUIRE(L->ctx, T > 0.0, "Temperature must be positive");
    TSU_REQUIRE(L->ctx, mode == TSU_MODE_PHYSICAL || mode == TSU_MODE_COMPAT, "ising2d_set_model: bad mode %d", mode);
    uint64_t t[25];
    tsu_ising2d_thresholds(J, h, T, mode, t);
    return tsu_ising2d_set_thresholds(L, t);
}

int tsu_ising2d_set_kernel(tsu_ising2d* L, int kernel, int sweeps_per_launch) {
    TSU_ENTER(L ? L->ctx : nullptr);
    if (!L) return TSU_E_INVALID;
    TSU_REQUIRE(L->ctx, kernel >= TSU_KERNEL_AUTO && kernel <= TSU_KERNEL_SMALL, "ising2d_set_kernel: bad kernel %d", kernel);
    TSU_REQUIRE(L->ctx, sweeps_per_launch >= 0 && sweeps_per_launch <= 16, "ising2d_set_kernel: sweeps_per_launch in [0,16]");
    if (kernel == TSU_KERNEL_TILED && !tsu_ising2d_tiled_supported(L))
        return tsu_fail(L->ctx, TSU_E_UNSUPPORTED, "ising2d_set_kernel: tiled kernel does not support this lattice");
    if (kernel == TSU_KERNEL_SMALL && !small_supported(L))
        return tsu_fail(L->ctx, TSU_E_UNSUPPORTED, "ising2d_set_kernel: the one-workgroup kernel takes whole lattices of at most 1024 octets (rows x ceil(cols/16))");
    L->kernel = kernel;
    L->sweeps_per_launch = sweeps_per_launch;
    return TSU_OK;
}

int tsu_ising2d_sweep(tsu_ising2d* L, int n_sweeps, uint64_t seed, uint32_t sweep0, uint32_t replica) {
    TSU_ENTER(L ? L->ctx : nullptr);
    return tsu_ising2d_sweep_part(L, n_sweeps, seed, sweep0, replica, TSU_PART_ALL);
}

int tsu_ising2d_sweep_part(tsu_ising2d* L, int n_sweeps, uint64_t seed, uint32_t sweep0, uint32_t replica, int part) {
    TSU_ENTER(L ? L->ctx : nullptr);
    if (!L) return TSU_E_INVALID;
    tsu_ctx* ctx = L->ctx;
    TSU_REQUIRE(ctx, part >= TSU_PART_ALL && part <= TSU_PART_BOUNDARY, "ising2d_sweep_part: bad part %d", part);
    TSU_REQUIRE(ctx, n_sweeps >= 0, "ising2d_sweep: n_sweeps must be >= 0");
    TSU_REQUIRE(ctx, L->have_table, "ising2d_sweep: call tsu_ising2d_set_model / set_thresholds first");
    TSU_REQUIRE(ctx, (uint64_t)sweep0 + (uint64_t)n_sweeps <= (1ull << 31), "ising2d_sweep: sweep counter overflow");
    TSU_REQUIRE(ctx, L->ghost == 0 || 2 * n_sweeps <= L->ghost,
                "ising2d_sweep: %d sweeps need %d ghost rows, slab has %d", n_sweeps, 2 * n_sweeps, L->ghost);
    if (n_sweeps == 0) return TSU_OK;
    if (L->timing) TSU_HIP_TRY(ctx, hipEventRecord(L->ev0, ctx->stream));
    // AUTO: a lattice that fits one workgroup's LDS runs all its sweeps in one launch; larger ones take the tiled
    // kernel where it applies, the generic one otherwise
    const int use_small = part == TSU_PART_ALL && ((L->kernel == TSU_KERNEL_SMALL) || (L->kernel == TSU_KERNEL_AUTO && small_supported(L)));
    int use_tiled = !use_small && ((L->kernel == TSU_KERNEL_TILED) || (L->kernel == TSU_KERNEL_AUTO && tsu_ising2d_tiled_supported(L)));
    if (part != TSU_PART_ALL && !(use_tiled && tsu_ising2d_tiled_part_supported(L)))
        return tsu_fail(ctx, TSU_E_UNSUPPORTED, "ising2d_sweep_part: split sweeps need a slab on the tiled kernel with rows %% 64 == 0");
    if (use_small && tsu_ising2d_planes_supported(L)) {
        // colour planes in LDS, packed-byte update (ising2d_tiled.hip); k1_small below keeps the shapes it does not take
        int rc = tsu_ising2d_planes_sweep(&L, 1, n_sweeps, &seed, &sweep0, &replica);
        if (rc != TSU_OK) return rc;
    } else if (use_small) {
        K1Params p = make_params(L, L->alloc[L->cur]);
        K1Table tbl;
        memcpy(tbl.t, L->table, sizeof(tbl.t));
        p.k0 = (uint32_t)seed;
        p.k1 = (uint32_t)(seed >> 32);
        p.tag_hi = TSU_TAG_ISING_HI | (replica << 8);
        p.tag_lo = TSU_TAG_ISING_LO | (replica << 8);
        const int nchunks = (L->cols + 15) >> 4, tasks = L->rows * nchunks;
        const unsigned threads = tasks >= 1024 ? 1024u : (unsigned)((tasks + 63) / 64 * 64);
        const size_t lds_bytes = (size_t)tasks * 16;
        TSU_HIP_TRY(ctx, tsu_func_allow_lds(ctx, (const void*)k1_small, 128 * 1024));
        k1_small<<<1, threads, lds_bytes, ctx->stream>>>(p, tbl, sweep0, n_sweeps);
        L->launches += 1;
        TSU_HIP_TRY(ctx, hipGetLastError());
    } else if (use_tiled) {
        int rc = tsu_ising2d_tiled_sweep(L, n_sweeps, seed, sweep0, replica, part);
        if (rc != TSU_OK) return rc;
    } else {
        K1Params p = make_params(L, L->alloc[L->cur]);
        K1Table tbl;
        memcpy(tbl.t, L->table, sizeof(tbl.t));
        p.k0 = (uint32_t)seed;
        p.k1 = (uint32_t)(seed >> 32);
        p.tag_hi = TSU_TAG_ISING_HI | (replica << 8);
        p.tag_lo = TSU_TAG_ISING_LO | (replica << 8);
        int half = 0;  // half-sweeps since the ghost rows were fresh
        const bool top_edge = !L->periodic && L->row0 <= L->ghost;
        const bool bot_edge = !L->periodic && L->total_rows - (L->row0 + L->rows) <= L->ghost;
        for (int s = 0; s < n_sweeps; ++s)
            for (int colour = 0; colour < 2; ++colour, ++half) {
                // ghost rows that can still be updated consistently shrink by one per half-sweep, except
                // where the ghost region reaches the lattice's open edge (nothing beyond it can go stale)
                int ext_top = top_edge ? p.top_rows : p.top_rows - 1 - half;
                int ext_bot = bot_edge ? p.bot_rows : p.bot_rows - 1 - half;
                p.r_lo = -(ext_top > 0 ? ext_top : 0);
                p.r_hi = L->rows + (ext_bot > 0 ? ext_bot : 0);
                p.hs = 2u * (sweep0 + (uint32_t)s) + (uint32_t)colour;
                k1_generic<<<grid_for(L, p.r_hi - p.r_lo), dim3(64, 4, 1), 0, ctx->stream>>>(p, tbl, colour);
                L->launches += 1;
            }
        TSU_HIP_TRY(ctx, hipGetLastError());
    }
    if (L->timing) {
        TSU_HIP_TRY(ctx, hipEventRecord(L->ev1, ctx->stream));
        L->timed = 1;
    }
    return TSU_OK;
}

int tsu_ising2d_set_timing(tsu_ising2d* L, int enable) {
    TSU_ENTER(L ? L->ctx : nullptr);
    if (!L) return TSU_E_INVALID;
    L->timing = enable != 0;
    L->timed = 0;
    return TSU_OK;
}

int tsu_ising2d_launch_count(tsu_ising2d* L, uint64_t* n) {
    TSU_ENTER(L ? L->ctx : nullptr);
    if (!L || !n) return TSU_E_INVALID;
    *n = L->launches;
    return TSU_OK;
}

int tsu_ising2d_last_sweep_ms(tsu_ising2d* L, float* ms) {
    TSU_ENTER(L ? L->ctx : nullptr);
    if (!L || !ms) return TSU_E_INVALID;
    TSU_REQUIRE(L->ctx, L->timed, "ising2d_last_sweep_ms: no sweep has been timed (tsu_ising2d_set_timing(lat, 1), then sweep)");
    TSU_HIP_TRY(L->ctx, hipEventSynchronize(L->ev1));
    TSU_HIP_TRY(L->ctx, hipEventElapsedTime(ms, L->ev0, L->ev1));
    return check_persist_error(L);
}

int tsu_ising2d_observables(tsu_ising2d* L, int64_t* sum_s, int64_t* sum_bonds) {
    TSU_ENTER(L ? L->ctx : nullptr);
    if (!L) return TSU_E_INVALID;
    tsu_ctx* ctx = L->ctx;
    TSU_REQUIRE(ctx, sum_s && sum_bonds, "ising2d_observables: NULL output");
    K1Params p = make_params(L, L->alloc[L->cur]);
    TSU_HIP_TRY(ctx, hipMemsetAsync(L->d_obs, 0, 2 * sizeof(int64_t), ctx->stream));
    long long work = (long long)L->rows * ((L->cols + 15) / 16);
    unsigned blocks = (unsigned)((work + 255) / 256 < 512 ? (work + 255) / 256 : 512);  // one atomic pair per block: keep them few (14 ns each, serialised)
    k4_observables<<<blocks, 256, 0, ctx->stream>>>(p, (long long*)L->d_obs);
    TSU_HIP_TRY(ctx, hipGetLastError());
    int64_t h[2];
    TSU_HIP_TRY(ctx, hipMemcpyAsync(h, L->d_obs, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    TSU_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *sum_s = h[0];
    *sum_bonds = h[1];
    return check_persist_error(L);
}

int tsu_ising2d_sample(tsu_ising2d* L, int n_burnin, int n_sweeps, int n_samples, uint64_t seed, uint32_t sweep0, uint32_t replica,
                       int8_t* samples_host) {
    TSU_ENTER(L ? L->ctx : nullptr);
    if (!L) return TSU_E_INVALID;
    tsu_ctx* ctx = L->ctx;
    TSU_REQUIRE(ctx, n_burnin >= 0 && n_sweeps > 0 && n_samples >= 0, "ising2d_sample: need n_burnin >= 0, n_sweeps > 0, n_samples >= 0");
    TSU_REQUIRE(ctx, n_samples == 0 || samples_host, "ising2d_sample: NULL output");
    TSU_REQUIRE(ctx, L->ghost == 0, "ising2d_sample: whole lattices only (a slab's ghost rows need the exchange between sweeps)");
    const size_t site_bytes = (size_t)L->rows * (size_t)L->cols, bytes = site_bytes * (size_t)n_samples;
    int8_t* d_samples = nullptr;
    if (bytes) TSU_HIP_TRY(ctx, hipMalloc(&d_samples, bytes));
    int rc = tsu_ising2d_sweep(L, n_burnin, seed, sweep0, replica);
    uint32_t sw = sweep0 + (uint32_t)n_burnin;
    hipError_t e = hipSuccess;
    for (int k = 0; k < n_samples && rc == TSU_OK && e == hipSuccess; ++k) {
        rc = tsu_ising2d_sweep(L, n_sweeps, seed, sw, replica);
        sw += (uint32_t)n_sweeps;
        if (rc == TSU_OK)
            e = hipMemcpy2DAsync(d_samples + (size_t)k * site_bytes, (size_t)L->cols, L->alloc[L->cur] + (size_t)L->ghost * L->pitch, L->pitch,
                                 (size_t)L->cols, (size_t)L->rows, hipMemcpyDeviceToDevice, ctx->stream);
    }
    if (rc == TSU_OK && e == hipSuccess && bytes) e = hipMemcpyAsync(samples_host, d_samples, bytes, hipMemcpyDeviceToHost, ctx->stream);
    const hipError_t e2 = hipStreamSynchronize(ctx->stream);
    if (d_samples) (void)hipFree(d_samples);
    if (rc != TSU_OK) return rc;
    if (e != hipSuccess || e2 != hipSuccess) return tsu_fail(ctx, TSU_E_HIP, "ising2d_sample: %s", hipGetErrorString(e != hipSuccess ? e : e2));
    return check_persist_error(L);
}

int tsu_ising2d_sweep_batch(tsu_ising2d* const* lats, int n_lats, int n_sweeps, const uint64_t* seeds, const uint32_t* sweep0s,
                            const uint32_t* replicas) {
    TSU_ENTER((lats && n_lats > 0 && lats[0]) ? lats[0]->ctx : nullptr);
    if (!lats || n_lats < 1 || !lats[0]) return TSU_E_INVALID;
    tsu_ctx* ctx = lats[0]->ctx;
    TSU_REQUIRE(ctx, seeds && sweep0s && replicas, "ising2d_sweep_batch: seeds, sweep0s and replicas are per-lattice arrays");
    TSU_REQUIRE(ctx, n_sweeps >= 0, "ising2d_sweep: n_sweeps must be >= 0");
    bool one_launch = true;
    for (int i = 0; i < n_lats; ++i) {
        tsu_ising2d* L = lats[i];
        TSU_REQUIRE(ctx, L && L->ctx == ctx, "ising2d_sweep_batch: lattice %d is NULL or belongs to another context", i);
        TSU_REQUIRE(ctx, L->have_table, "ising2d_sweep: call tsu_ising2d_set_model / set_thresholds first");
        TSU_REQUIRE(ctx, (uint64_t)sweep0s[i] + (uint64_t)n_sweeps <= (1ull << 31), "ising2d_sweep: sweep counter overflow");
        one_launch = one_launch && small_supported(L) && L->rows == lats[0]->rows && L->cols == lats[0]->cols &&
                     (L->kernel == TSU_KERNEL_AUTO || L->kernel == TSU_KERNEL_SMALL);
    }
    if (n_sweeps == 0) return TSU_OK;
    if (!one_launch) {
        // larger lattices: each lattice's launches go to one of a few side streams, so that lattices which do not fill
        // the chip on their own (a 1024^2 lattice keeps 128 CUs busy) run side by side.  Fork / join with events
        // on the context's stream, which therefore sees the batch as one ordered operation.
        if (ctx->pool_n == 0) {
            TSU_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->fork_ev, hipEventDisableTiming));
            for (int i = 0; i < 8; ++i) {
                TSU_HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->pool[i], hipStreamNonBlocking));
                TSU_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->pool_ev[i], hipEventDisableTiming));
                ctx->pool_n = i + 1;
            }
        }
        hipStream_t main_stream = ctx->stream;
        TSU_HIP_TRY(ctx, hipEventRecord(ctx->fork_ev, main_stream));
        // Tile-resident launches wait inside the kernel for all of their workgroups: never have more of them in flight
        // than fit the chip together (one workgroup per CU counted), or two half-placed grids could wait for each other.
        int max_tiles = 0;
        ctx->in_batch = 1;  // (the tile plan of a lattice in a batch differs from that of a lattice on its own: see tile_plan)
        for (int i = 0; i < n_lats; ++i) {
            const int t = tsu_ising2d_tiled_tiles(lats[i]);
            if (t > max_tiles) max_tiles = t;
        }
        ctx->in_batch = 0;
        int used = n_lats < ctx->pool_n ? n_lats : ctx->pool_n;
        if (max_tiles > 0) {
            const int fit = ctx->cus / max_tiles;
            if (used > fit) used = fit < 1 ? 1 : fit;
        }
        for (int i = 0; i < used; ++i) TSU_HIP_TRY(ctx, hipStreamWaitEvent(ctx->pool[i], ctx->fork_ev, 0));
        int rc = TSU_OK;
        ctx->in_batch = 1;
        for (int i = 0; i < n_lats && rc == TSU_OK; ++i) {
            ctx->stream = ctx->pool[i % used];
            rc = tsu_ising2d_sweep(lats[i], n_sweeps, seeds[i], sweep0s[i], replicas[i]);
        }
        ctx->in_batch = 0;
        ctx->stream = main_stream;
        for (int i = 0; i < used; ++i) {
            hipError_t e = hipEventRecord(ctx->pool_ev[i], ctx->pool[i]);
            if (e == hipSuccess) e = hipStreamWaitEvent(main_stream, ctx->pool_ev[i], 0);
            if (e != hipSuccess && rc == TSU_OK) rc = tsu_fail(ctx, TSU_E_HIP, "ising2d_sweep_batch: %s", hipGetErrorString(e));
        }
        return rc;
    }
    bool planes = true;
    for (int i = 0; i < n_lats; ++i) planes = planes && tsu_ising2d_planes_supported(lats[i]) && lats[i]->periodic == lats[0]->periodic;
    if (planes) return tsu_ising2d_planes_sweep(lats, n_lats, n_sweeps, seeds, sweep0s, replicas);
    std::vector<K1BatchItem> items((size_t)n_lats);
    for (int i = 0; i < n_lats; ++i) {
        tsu_ising2d* L = lats[i];
        K1BatchItem& it = items[(size_t)i];
        it.p = make_params(L, L->alloc[L->cur]);
        memcpy(it.tbl.t, L->table, sizeof(it.tbl.t));
        it.p.k0 = (uint32_t)seeds[i];
        it.p.k1 = (uint32_t)(seeds[i] >> 32);
        it.p.tag_hi = TSU_TAG_ISING_HI | (replicas[i] << 8);
        it.p.tag_lo = TSU_TAG_ISING_LO | (replicas[i] << 8);
        it.sweep0 = sweep0s[i];
    }
    const size_t bytes = items.size() * sizeof(K1BatchItem);
    tsu_ising2d* L0 = lats[0];  // the staging buffer for the items lives with the first lattice of the batch
    if (L0->batch_cap < bytes) {
        if (L0->d_batch) (void)hipFree(L0->d_batch);
        L0->d_batch = nullptr;
        L0->batch_cap = 0;
        TSU_HIP_TRY(ctx, hipMalloc(&L0->d_batch, bytes));
        L0->batch_cap = bytes;
    }
    // stream order keeps a previous batch launch from still reading the buffer; the host array dies with this call, so
    // the copy is waited for (a few KB); the launch itself stays asynchronous
    TSU_HIP_TRY(ctx, hipMemcpyAsync(L0->d_batch, items.data(), bytes, hipMemcpyHostToDevice, ctx->stream));
    TSU_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const int nchunks = (lats[0]->cols + 15) >> 4, tasks = lats[0]->rows * nchunks;
    const unsigned threads = tasks >= 1024 ? 1024u : (unsigned)((tasks + 63) / 64 * 64);
    k1_small_batch<<<(unsigned)n_lats, threads, (size_t)tasks * 16, ctx->stream>>>((const K1BatchItem*)L0->d_batch, n_sweeps);
    TSU_HIP_TRY(ctx, hipGetLastError());
    for (int i = 0; i < n_lats; ++i) lats[i]->launches += 1;
    return TSU_OK;
}

int tsu_ising2d_observables_batch(tsu_ising2d* const* lats, int n_lats, int64_t* sum_s, int64_t* sum_bonds) {
    TSU_ENTER((lats && n_lats > 0 && lats[0]) ? lats[0]->ctx : nullptr);
    if (!lats || n_lats < 1 || !lats[0]) return TSU_E_INVALID;
    tsu_ctx* ctx = lats[0]->ctx;
    TSU_REQUIRE(ctx, sum_s && sum_bonds, "ising2d_observables: NULL output");
    std::vector<int64_t> h((size_t)2 * n_lats);
    // one accumulator array for the batch (with its first lattice): one memset, one launch per lattice, one copy back
    tsu_ising2d* L0 = lats[0];
    const size_t bytes = h.size() * sizeof(int64_t);
    if (L0->obs_batch_cap < bytes) {
        if (L0->d_obs_batch) (void)hipFree(L0->d_obs_batch);
        L0->d_obs_batch = nullptr;
        L0->obs_batch_cap = 0;
        TSU_HIP_TRY(ctx, hipMalloc(&L0->d_obs_batch, bytes));
        L0->obs_batch_cap = bytes;
    }
    TSU_HIP_TRY(ctx, hipMemsetAsync(L0->d_obs_batch, 0, bytes, ctx->stream));
    for (int i = 0; i < n_lats; ++i) {
        tsu_ising2d* L = lats[i];
        TSU_REQUIRE(ctx, L && L->ctx == ctx, "ising2d_observables_batch: lattice %d is NULL or belongs to another context", i);
        K1Params p = make_params(L, L->alloc[L->cur]);
        long long work = (long long)L->rows * ((L->cols + 15) / 16);
        unsigned blocks = (unsigned)((work + 255) / 256 < 512 ? (work + 255) / 256 : 512);  // one atomic pair per block: keep them few (14 ns each, serialised)
        k4_observables<<<blocks, 256, 0, ctx->stream>>>(p, (long long*)L0->d_obs_batch + 2 * i);
    }
    TSU_HIP_TRY(ctx, hipGetLastError());
    TSU_HIP_TRY(ctx, hipMemcpyAsync(h.data(), L0->d_obs_batch, bytes, hipMemcpyDeviceToHost, ctx->stream));
    TSU_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < n_lats; ++i) {
        sum_s[i] = h[(size_t)2 * i];
        sum_bonds[i] = h[(size_t)2 * i + 1];
    }
    return TSU_OK;
}

int tsu_ising2d_row_ptr(tsu_ising2d* L, int local_row, void** device_ptr, size_t* pitch_bytes) {
    TSU_ENTER(L ? L->ctx : nullptr);
    if (!L) return TSU_E_INVALID;
    TSU_REQUIRE(L->ctx, device_ptr && local_row >= -L->ghost && local_row < L->rows + L->ghost,
                "ising2d_row_ptr: row %d outside [%d, %d)", local_row, -L->ghost, L->rows + L->ghost);
    *device_ptr = L->alloc[L->cur] + (size_t)(L->ghost + local_row) * L->pitch;
    if (pitch_bytes) *pitch_bytes = L->pitch;
    return TSU_OK;
}

}  // extern "C"
