// ising2d_tiled.hip -- K1 fast path: LDS-staged 2-D halo tiles, k full sweeps per launch (gfx950).
//
// Same Markov kernel and the same Philox stream as k1_generic / oracle ora_ising2d_sweep (results are bit
// identical); what changes is the schedule:
//   * a workgroup stages a (H + 4k) x (16 WO + 32) tile of the int8 lattice into LDS, de-interleaved into two
//     colour planes of 0/1 "up" flags (8 same-colour sites of a row = one octet = one uint64),
//   * runs 2k half-sweeps entirely in LDS (the region that is still exact shrinks by one site per half-sweep:
//     2k halo rows, one halo octet = 16 columns >= 2k each side), and
//   * writes the H x 16 WO interior to the OTHER lattice buffer (neighbouring tiles read this tile's pre-launch
//     halo, so the launch is out of place; the host ping-pongs).
// When every tile of the lattice has its own workgroup on the chip (up to 2^25 sites), k1_resident keeps the tiles in
// LDS for a whole call and only exchanges boundary strips between generations of k sweeps (see ResidentParams).
// HBM traffic is 2/k bytes per spin update instead of 2 (far less when resident); the kernel is VALU bound (Philox), so per octet:
//   one Philox4x32-10 block -> 8 x 16 random bits; neighbour sums with packed byte adds; 16-bit thresholds
//   picked per site with v_perm_b32 byte look-ups; v_pk_sub_i16 (saturating) compares two sites per
//   instruction; only an exact tie of the top 16 bits (2^-16 per site) evaluates the low half.
#include "ising2d.h"

#include <type_traits>
#include <vector>

typedef short v2s __attribute__((ext_vector_type(2)));
typedef unsigned short v2us __attribute__((ext_vector_type(2)));

struct TiledParams {
    const int8_t* src;  // owned row 0 of the source buffer
    int8_t* dst;        // owned row 0 of the destination buffer
    long long pitch;
    int rows, nchunks;           // owned rows; ceil(cols / 16) (the last chunk may be ragged)
    int cols;
    int q_shift;                 // first octet of tile column 0 (periodic lattices of ragged width: keeps the ragged octet out of every halo)
    int tile_h;                  // tile height of this launch (even; the template H is the default and the LDS budget)
    int flex_ty;                 // > 0: the rows [r_begin, r_end) are cut into flex_ty tile rows of (nearly) equal even heights
                                 // <= tile_h instead of rows of tile_h each (tile-resident runs of lattices of any height)
    int r_begin, r_end;          // rows this launch computes: [0, rows) or, for a slab that will sweep again before its
                                 // next ghost refresh, [-ext, rows + ext) so that its own halo stays exact
    long long row0, total_rows;  // global row of owned row 0; global lattice height
    int wrap_rows, ghost;        // source rows wrap inside the buffer, or come from `ghost` ghost rows
    int k;                       // sweeps in this launch
    int tiles_x;
    int ty_first, ty_stride;     // tile row of block b: ty_first + (b / tiles_x) * ty_stride
    uint32_t k0, k1, sweep0, tag_hi, tag_lo;
    uint32_t tblH0, tblH1, tblL0, tblL1;  // (min(thr >> 16, 65535) ^ 0x8000) for up = 0..4, split into byte tables
    uint32_t t3H0, t3H1, t3L0, t3L1;      // open lattices: the same for degree 3 (entries 0..3) and degree 2 (entries 4..6)
    int open;                             // open boundary: sites outside the lattice count as 0 "up" and lower the degree
    uint64_t thr[25];                     // full thresholds, [deg * 5 + up], for the tie path
};

static __device__ __forceinline__ uint32_t perm(uint32_t s0, uint32_t s1, uint32_t sel) {
    return __builtin_amdgcn_perm(s0, s1, sel);
}
static __device__ __forceinline__ uint32_t subsat16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(v2s, a), __builtin_bit_cast(v2s, b)));
}
static __device__ __forceinline__ uint32_t minu16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(v2us, a), __builtin_bit_cast(v2us, b)));
}

// exact resolution of the (rare) fields whose top 16 bits tie with the threshold: d fields that are 0.
// Cold path (probability 2^-16 per site): kept out of line and register-to-register (no arrays by address).
static __device__ __noinline__ u32x4 resolve_ties(u32x4 d, u32x4 w, uint32_t cnt_lo, uint32_t cnt_hi, const uint64_t* s_thr,
                                                  uint32_t cq, uint32_t Rg, uint32_t hs, uint32_t tag_lo, uint32_t k0,
                                                  uint32_t k1, uint32_t edge = 0) {
    // edge (open lattices): bit 0 = the row is the lattice's top or bottom row, bit 1 = site 0 of the octet is in the
    // lattice's first column, bit 2 = site (edge >> 4) is in its last column; degree = 4 minus the missing neighbours
    const u32x4 l = tsu_philox(cq, Rg, hs, tag_lo, k0, k1);
#define TSU_FIX(DW, WW, LW, CNT, B0, M0)                                                                        \
    _Pragma("unroll") for (int hlf = 0; hlf < 2; ++hlf) {                                                       \
        const uint32_t sh = 16u * hlf;                                                                          \
        if (((DW >> sh) & 0xFFFFu) == 0) {                                                                      \
            const uint32_t c = (CNT >> (8 * (B0 + hlf))) & 0xFFu;                                               \
            const uint32_t m = M0 + hlf;                                                                        \
            const uint32_t deg = 4u - (edge & 1u) - ((m == 0 && (edge & 2u)) ? 1u : 0u) - ((m == (edge >> 4) && (edge & 4u)) ? 1u : 0u); \
            const uint64_t u = ((uint64_t)(((WW >> sh) & 0xFFFFu) ^ 0x8000u) << 16) | ((LW >> sh) & 0xFFFFu);   \
            const uint32_t f = (u < s_thr[deg * 5 + c]) ? 0x8000u : 0x0001u; /* negative field = accept */      \
            DW = (DW & ~(0xFFFFu << sh)) | (f << sh);                                                           \
        }                                                                                                       \
    }
    TSU_FIX(d.x, w.x, l.x, cnt_lo, 0, 0)
    TSU_FIX(d.y, w.y, l.y, cnt_lo, 2, 2)
    TSU_FIX(d.z, w.z, l.z, cnt_hi, 0, 4)
    TSU_FIX(d.w, w.w, l.w, cnt_hi, 2, 6)
#undef TSU_FIX
    return d;
}

// ================================================================== inner loop
// The half-sweep body is organised for instruction count and ILP:
//   * a thread owns ONE octet column for the whole kernel (global chunk index fixed, no div/mod in the loop) and
//     walks down the rows in pairs (r, r+1): the two rows share two of their three neighbour-row reads, and their
//     two Philox blocks are independent dependency chains that the scheduler interleaves;
//   * the ten round keys live in VGPRs (a v_xor with an SGPR operand issues at half the rate of a VGPR-only one);
//   * one tie test covers both rows.
struct PhiloxKeys {
    uint32_t a[10], b[10];
};

static __device__ __forceinline__ PhiloxKeys make_keys(uint32_t k0, uint32_t k1) {
    PhiloxKeys K;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t x = k0 + (uint32_t)r * TSU_PHILOX_W0, y = k1 + (uint32_t)r * TSU_PHILOX_W1;
        asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "s"(x));  // opaque: keep the keys in vector registers
        asm volatile("v_mov_b32 %0, %1" : "=v"(y) : "s"(y));
        K.a[r] = x;
        K.b[r] = y;
    }
    return K;
}

static __device__ __forceinline__ u32x4 philox_vk(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, const PhiloxKeys& K) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)TSU_PHILOX_M0 * c0;
        uint64_t p1 = (uint64_t)TSU_PHILOX_M1 * c2;
        uint32_t n0 = tsu_xor3((uint32_t)(p1 >> 32), c1, K.a[r]);
        uint32_t n2 = tsu_xor3((uint32_t)(p0 >> 32), c3, K.b[r]);
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
    }
    return u32x4{c0, c1, c2, c3};
}

struct Rows2Ctx {
    const void* Ps;  // source / destination colour plane (uint64 octets, or uint32 octets in the nibble form)
    void* Pd;
    const uint64_t* s_thr;
    int tr_lo, npairs;   // first tile row of the half-sweep's range, number of row pairs
    int rgf;             // global row of tile row tr_lo (wrapped)
    int total_rows;
    int rg_tile0, wrap_tr;  // EDGE tiles: global row of tile row 0 (wrapped); the tile row whose global row is 0 again
    uint32_t hs, tag_hi, tag_lo, k0, k1;
    uint32_t tblH0, tblH1, tblL0, tblL1;
    uint32_t t3H0, t3H1, t3L0, t3L1;  // open lattices: degree 3 / degree 2 byte tables
    int last_row;                     // open lattices: global index of the lattice's bottom row
    bool lft, rgt;                    // open lattices: this thread's octet holds the lattice's first / last column
    uint32_t r_par, r_slot;           // open lattices: column parity of the last column and its site within the octet
    uint32_t rm_lo, rm_hi;            // 0xFF at that site's byte
    uint64_t vm_e, vm_o;              // open lattices: 0x01 for the even / odd columns of this octet that exist
    int sh_next, sh_prev;             // periodic lattices of ragged width (SEAM): where the wrap meets this octet, see sweep_pairs
};

// thresholds for the 8 sites of one octet given their up-counts, compared with one Philox block: d < 0 = accept
static __device__ __forceinline__ u32x4 compare_octet(const u32x4& w, uint32_t cnt_lo, uint32_t cnt_hi, uint32_t tH0,
                                                      uint32_t tH1, uint32_t tL0, uint32_t tL1) {
    const uint32_t Hl = perm(tH1, tH0, cnt_lo), Ll = perm(tL1, tL0, cnt_lo);
    const uint32_t Hh = perm(tH1, tH0, cnt_hi), Lh = perm(tL1, tL0, cnt_hi);
    u32x4 d;
    d.x = subsat16(w.x, perm(Hl, Ll, 0x05010400u));
    d.y = subsat16(w.y, perm(Hl, Ll, 0x07030602u));
    d.z = subsat16(w.z, perm(Hh, Lh, 0x05010400u));
    d.w = subsat16(w.w, perm(Hh, Lh, 0x07030602u));
    return d;
}

// Open lattices: as compare_octet, then the sites with fewer than four neighbours take their thresholds from the
// degree-3 / degree-2 table instead (row_edge: the whole row is the lattice's top or bottom row; l0: site 0 of the octet
// sits in the lattice's first column; r7: site c.r_slot sits in its last column -- site 7 unless the width is ragged).  Only executed by waves that hold such a site.
static __device__ __forceinline__ u32x4 compare_octet_open(const u32x4& w, uint32_t cnt_lo, uint32_t cnt_hi, const Rows2Ctx& c,
                                                           bool row_edge, bool l0, bool r7) {
    uint32_t Hl = perm(c.tblH1, c.tblH0, cnt_lo), Ll = perm(c.tblL1, c.tblL0, cnt_lo);
    uint32_t Hh = perm(c.tblH1, c.tblH0, cnt_hi), Lh = perm(c.tblL1, c.tblL0, cnt_hi);
    if (row_edge | l0 | r7) {
        // entries 0..3 = degree 3, 4..6 = degree 2 (a corner: edge row and edge column)
        const uint32_t rl = r7 ? c.rm_lo : 0u, rh = r7 ? c.rm_hi : 0u;
        const uint32_t il = cnt_lo + ((row_edge && l0) ? 0x00000004u : 0u) + (row_edge ? (rl & 0x04040404u) : 0u);
        const uint32_t ih = cnt_hi + (row_edge ? (rh & 0x04040404u) : 0u);
        const uint32_t ml = row_edge ? 0xFFFFFFFFu : ((l0 ? 0x000000FFu : 0u) | rl), mh = row_edge ? 0xFFFFFFFFu : rh;
        Hl = (Hl & ~ml) | (perm(c.t3H1, c.t3H0, il) & ml);
        Ll = (Ll & ~ml) | (perm(c.t3L1, c.t3L0, il) & ml);
        Hh = (Hh & ~mh) | (perm(c.t3H1, c.t3H0, ih) & mh);
        Lh = (Lh & ~mh) | (perm(c.t3L1, c.t3L0, ih) & mh);
    }
    u32x4 d;
    d.x = subsat16(w.x, perm(Hl, Ll, 0x05010400u));
    d.y = subsat16(w.y, perm(Hl, Ll, 0x07030602u));
    d.z = subsat16(w.z, perm(Hh, Lh, 0x05010400u));
    d.w = subsat16(w.w, perm(Hh, Lh, 0x07030602u));
    return d;
}

static __device__ __forceinline__ bool has_zero_field(const u32x4& d) {
    const uint32_t mn = minu16(minu16(d.x, d.y), minu16(d.z, d.w));
    return ((mn & 0xFFFFu) == 0) | ((mn >> 16) == 0);
}

static __device__ __forceinline__ uint64_t pack_flags(const u32x4& d) {
    // v_perm_b32 selectors 8..11 replicate the sign bit of the four 16-bit fields of {S0,S1} (bits 15, 31, 47, 63)
    // into a whole byte: one instruction turns the four "accept" signs into 0x00 / 0xFF bytes
    const uint32_t nlo = perm(d.y, d.x, 0x0B0A0908u) & 0x01010101u;
    const uint32_t nhi = perm(d.w, d.z, 0x0B0A0908u) & 0x01010101u;
    return (uint64_t)nlo | ((uint64_t)nhi << 32);
}

// EDGE tiles: global rows of the pair at tile rows (tr, tr + 1).  The lattice's last row falls at ONE tile row (wrap_tr), so a
// wave's lanes (a few consecutive row lanes: tile rows tr_wave .. tr_wave + SPAN) lie on one side of it in all but one
// wave-iteration per half-sweep: the offset is a scalar select and the per-lane compares run in that one iteration only
// (they cost the edge tiles +4.7 % when every lane made them every time -- and the edge tiles set every tile's pace).
template <int NO>
static __device__ __forceinline__ void edge_rows(const Rows2Ctx& c, int tr_wave, int tr, int& rga, int& rgb) {
    constexpr int SPAN = 2 * ((NO - 1 + 63) / NO) + 1;  // tile rows below the first lane's that a wave's 64 lanes can reach
    const int s_off = tr_wave >= c.wrap_tr ? c.rg_tile0 - c.total_rows : c.rg_tile0;
    rga = tr + s_off;
    rgb = rga + 1;
    if (__builtin_expect(tr_wave < c.wrap_tr && tr_wave + SPAN >= c.wrap_tr, 0)) {  // wave-uniform
        rga = tr + c.rg_tile0;
        asm volatile("" : "+v"(rga));  // a real branch: as selects this would cost every iteration 9 instructions
        rgb = rga + 1;
        if (rga >= c.total_rows) rga -= c.total_rows;
        if (rgb >= c.total_rows) rgb -= c.total_rows;
    }
}

// P0 = column parity of the updated colour in row tr_lo (and 1-P0 in the row below it)
template <int NO, int P0, bool EDGE, bool OPEN, bool SEAM = false>
static __device__ __forceinline__ void sweep_pairs(const Rows2Ctx& c, const PhiloxKeys& K, int tr_first, int tr_end, int step_rows, int oct,
                                                   uint32_t cq) {
    // One byte offset into the source plane and one row counter are the only induction variables (kept opaque so
    // that the compiler does not re-derive them from a separate trip counter); everything else is an immediate.
    const char* const ps0 = reinterpret_cast<const char*>(c.Ps);
    const int d_off = (int)(reinterpret_cast<const char*>(c.Pd) - ps0);
    // this thread: the pairs at tile rows tr_first, tr_first + step_rows, ... < tr_end (tr_first has the parity of c.tr_lo)
    const int off_end = tr_end * NO * 8;
    const int off_step = step_rows * NO * 8;
    int off = (tr_first * NO + oct) * 8;
    // EDGE (the tile's window crosses the lattice's last row): rg counts tile rows and the wrap is applied per wave, see edge_rows
    int rg = EDGE ? tr_first : c.rgf + (tr_first - c.tr_lo);
    int tr_wave = EDGE ? __builtin_amdgcn_readfirstlane(rg) : 0;  // tile row of the wave's first lane (lanes ascend in rows)
#pragma unroll 1
    for (; off < off_end; off += off_step, rg += step_rows, tr_wave += step_rows) {
        asm volatile("" : "+v"(off), "+v"(rg));
        // issue the six LDS reads, run the two Philox blocks (which do not depend on them) while they are in flight,
        // and only then consume the neighbour rows
        const char* ps = ps0 + off;
        const uint64_t R0 = *reinterpret_cast<const uint64_t*>(ps - NO * 8), R1 = *reinterpret_cast<const uint64_t*>(ps);
        const uint64_t R2 = *reinterpret_cast<const uint64_t*>(ps + NO * 8), R3 = *reinterpret_cast<const uint64_t*>(ps + 2 * NO * 8);
        const uint32_t A0 = SEAM ? 0u : *reinterpret_cast<const uint32_t*>(ps + (P0 ? 8 : -4));
        const uint32_t A1 = SEAM ? 0u : *reinterpret_cast<const uint32_t*>(ps + NO * 8 + (P0 ? -4 : 8));
        // SEAM (a periodic lattice whose width is not a multiple of 16, workgroups whose window holds the wrap): the last
        // octet of a row has only v sites per colour, so the site after its last one is site 0 of octet 0, and the site
        // before site 0 of octet 0 is that last one: the whole neighbouring octet is read and the one byte that crosses
        // over is taken from / put at position v - 1 instead of 7 (sh_next / sh_prev; 56 for every other octet)
        const uint64_t Na = SEAM ? *reinterpret_cast<const uint64_t*>(ps + (P0 ? 8 : -8)) : 0ull;
        const uint64_t Nb = SEAM ? *reinterpret_cast<const uint64_t*>(ps + NO * 8 + (P0 ? -8 : 8)) : 0ull;
        __builtin_amdgcn_sched_barrier(0);  // the reads stay above the Philox blocks ...
        int rga = rg, rgb = rg + 1;
        if (EDGE) edge_rows<NO>(c, tr_wave, rg, rga, rgb);
        const u32x4 w0 = philox_vk(cq, (uint32_t)rga, c.hs, c.tag_hi, K);
        const u32x4 w1 = philox_vk(cq, (uint32_t)rgb, c.hs, c.tag_hi, K);
        __builtin_amdgcn_sched_barrier(0);  // ... and their first use stays below
        const uint32_t C0l = (uint32_t)R1, C0h = (uint32_t)(R1 >> 32), C1l = (uint32_t)R2, C1h = (uint32_t)(R2 >> 32);
        // horizontal neighbours: compact bytes (j, j+1) when the parity is 1, (j-1, j) when it is 0
        uint32_t S0l = P0 ? __builtin_amdgcn_alignbyte(C0h, C0l, 1) : __builtin_amdgcn_alignbyte(C0l, A0, 3);
        uint32_t S0h = P0 ? __builtin_amdgcn_alignbyte(A0, C0h, 1) : __builtin_amdgcn_alignbyte(C0h, C0l, 3);
        uint32_t S1l = P0 ? __builtin_amdgcn_alignbyte(C1l, A1, 3) : __builtin_amdgcn_alignbyte(C1h, C1l, 1);
        uint32_t S1h = P0 ? __builtin_amdgcn_alignbyte(C1h, C1l, 3) : __builtin_amdgcn_alignbyte(A1, C1h, 1);
        if (SEAM) {
            const uint64_t nxa = (Na & 0xFFull) << c.sh_next, pva = (Na >> c.sh_prev) & 0xFFull;
            const uint64_t nxb = (Nb & 0xFFull) << c.sh_next, pvb = (Nb >> c.sh_prev) & 0xFFull;
            const uint64_t Sa = P0 ? ((R1 >> 8) | nxa) : ((R1 << 8) | pva);
            const uint64_t Sb = P0 ? ((R2 << 8) | pvb) : ((R2 >> 8) | nxb);
            S0l = (uint32_t)Sa; S0h = (uint32_t)(Sa >> 32);
            S1l = (uint32_t)Sb; S1h = (uint32_t)(Sb >> 32);
        }
        // R1 + R2 is shared by both rows' vertical+centre sums
        const uint32_t ml = C0l + C1l, mh = C0h + C1h;
        const uint32_t cnt0l = (uint32_t)R0 + ml + S0l, cnt0h = (uint32_t)(R0 >> 32) + mh + S0h;
        const uint32_t cnt1l = (uint32_t)R3 + ml + S1l, cnt1h = (uint32_t)(R3 >> 32) + mh + S1h;
        u32x4 d0, d1;
        uint32_t edge_a = 0, edge_b = 0;
        if (OPEN) {
            // row a updates the columns of parity P0, row b those of parity 1 - P0: the lattice's first column is site 0
            // of octet 0 for parity 0, its last column site r_slot of the last octet for parity r_par
            const bool ea = rga == 0 || rga == c.last_row, eb = rgb == 0 || rgb == c.last_row;
            const bool la = !P0 && c.lft, ra = c.rgt && c.r_par == (uint32_t)P0, lb = P0 && c.lft, rb = c.rgt && c.r_par != (uint32_t)P0;
            d0 = compare_octet_open(w0, cnt0l, cnt0h, c, ea, la, ra);
            d1 = compare_octet_open(w1, cnt1l, cnt1h, c, eb, lb, rb);
            edge_a = (ea ? 1u : 0u) | (la ? 2u : 0u) | (ra ? 4u : 0u) | (c.r_slot << 4);
            edge_b = (eb ? 1u : 0u) | (lb ? 2u : 0u) | (rb ? 4u : 0u) | (c.r_slot << 4);
        } else {
            d0 = compare_octet(w0, cnt0l, cnt0h, c.tblH0, c.tblH1, c.tblL0, c.tblL1);
            d1 = compare_octet(w1, cnt1l, cnt1h, c.tblH0, c.tblH1, c.tblL0, c.tblL1);
        }
        const uint32_t mn = minu16(minu16(minu16(d0.x, d0.y), minu16(d0.z, d0.w)), minu16(minu16(d1.x, d1.y), minu16(d1.z, d1.w)));
        if (__builtin_expect(((mn & 0xFFFFu) == 0) | ((mn >> 16) == 0), 0)) {
            if (has_zero_field(d0)) d0 = resolve_ties(d0, w0, cnt0l, cnt0h, c.s_thr, cq, (uint32_t)rga, c.hs, c.tag_lo, c.k0, c.k1, edge_a);
            if (has_zero_field(d1)) d1 = resolve_ties(d1, w1, cnt1l, cnt1h, c.s_thr, cq, (uint32_t)rgb, c.hs, c.tag_lo, c.k0, c.k1, edge_b);
        }
        char* pd = const_cast<char*>(ps) + d_off;
        uint64_t n0 = pack_flags(d0), n1 = pack_flags(d1);
        if (OPEN) {  // what lies beyond the open edge stays empty (it is a neighbour of the edge sites in the next half-sweep)
            n0 = (rga < 0 || rga > c.last_row) ? 0 : (n0 & (P0 ? c.vm_o : c.vm_e));
            n1 = (rgb < 0 || rgb > c.last_row) ? 0 : (n1 & (P0 ? c.vm_e : c.vm_o));
        }
        if (SEAM) {  // the sites the last octet does not have stay empty
            n0 &= c.vm_e;
            n1 &= c.vm_e;
        }
        *reinterpret_cast<uint64_t*>(pd) = n0;
        *reinterpret_cast<uint64_t*>(pd + NO * 8) = n1;
    }
}


// ================================================================== nibble planes (NIB)
// The same update with the colour planes kept at 4 bits per site instead of 8: an octet is ONE dword, site j < 4 in the
// low nibble of byte j, site j >= 4 in the high nibble of byte j - 4 (so that `x & 0x0F0F0F0F` and `(x >> 4) & 0x0F0F0F0F`
// are the byte-per-site halves the threshold look-up wants).  Neighbour counts (<= 4) are summed nibble-wise with plain
// dword adds -- half the adds of the byte form -- and only the count is widened to bytes.  Half the LDS per site: a
// 512 x 512 tile (+ halo) fits one CU, which makes 8192^2 tile-resident and halves the halo work of the big lattices.
// Periodic lattices whose width is a multiple of 16 only (no OPEN / SEAM forms).
static __device__ __forceinline__ uint32_t nib_shift_next(uint32_t X, uint32_t N) {
    // S_j = X_{j+1}; site 3 <- site 4 (high nibble of byte 0), site 7 <- site 0 of the next octet N
    const uint32_t hi = (N << 4) | __builtin_amdgcn_ubfe(X, 4u, 4u);
    return __builtin_amdgcn_alignbit(hi, X, 8);
}
static __device__ __forceinline__ uint32_t nib_shift_prev(uint32_t X, uint32_t M) {
    // S_j = X_{j-1}; site 4 <- site 3 (low nibble of byte 3), site 0 <- site 7 of the previous octet M
    const uint32_t w = ((X << 4) & 0x10000000u) | (M >> 4);
    return __builtin_amdgcn_alignbit(X, w, 24);
}
// SEAM (periodic width not a multiple of 16, see sweep_pairs): the octet that crosses over sits at site v - 1 of the ragged
// last octet instead of site 7 -- bit position `pos` = 8 (j & 3) + 4 (j >> 2) of site j (28 for every other octet)
static __device__ __forceinline__ uint32_t nib_shift_next_seam(uint32_t X, uint32_t N, int pos) {
    return nib_shift_next(X, 0u) | ((N & 0xFu) << pos);  // (site v of a ragged octet is empty, so the slot is free)
}
static __device__ __forceinline__ uint32_t nib_shift_prev_seam(uint32_t X, uint32_t M, int pos) {
    const uint32_t w = ((X << 4) & 0x10000000u) | (((M >> pos) & 0xFu) << 24);
    return __builtin_amdgcn_alignbit(X, w, 24);
}
static __device__ __forceinline__ uint32_t nib_pack(const u32x4& d) {
    const uint64_t f = pack_flags(d);
    return (uint32_t)f | ((uint32_t)(f >> 32) << 4);
}

template <int NO, int P0, bool EDGE, bool OPEN = false, bool SEAM = false>
static __device__ __forceinline__ void sweep_pairs_nib(const Rows2Ctx& c, const PhiloxKeys& K, int tr_first, int tr_end, int step_rows, int oct,
                                                       uint32_t cq) {
    const char* const ps0 = reinterpret_cast<const char*>(c.Ps);
    const int d_off = (int)(reinterpret_cast<const char*>(c.Pd) - ps0);
    const int off_end = tr_end * NO * 4;
    const int off_step = step_rows * NO * 4;
    int off = (tr_first * NO + oct) * 4;
    int rg = EDGE ? tr_first : c.rgf + (tr_first - c.tr_lo);
    int tr_wave = EDGE ? __builtin_amdgcn_readfirstlane(rg) : 0;
#pragma unroll 1
    for (; off < off_end; off += off_step, rg += step_rows, tr_wave += step_rows) {
        asm volatile("" : "+v"(off), "+v"(rg));
        const char* ps = ps0 + off;
        const uint32_t R0 = *reinterpret_cast<const uint32_t*>(ps - NO * 4), R1 = *reinterpret_cast<const uint32_t*>(ps);
        const uint32_t R2 = *reinterpret_cast<const uint32_t*>(ps + NO * 4), R3 = *reinterpret_cast<const uint32_t*>(ps + 2 * NO * 4);
        const uint32_t A0 = *reinterpret_cast<const uint32_t*>(ps + (P0 ? 4 : -4));           // row a: next octet if P0, else the previous one
        const uint32_t A1 = *reinterpret_cast<const uint32_t*>(ps + NO * 4 + (P0 ? -4 : 4));  // row b: the other side
        __builtin_amdgcn_sched_barrier(0);
        int rga = rg, rgb = rg + 1;
        if (EDGE) edge_rows<NO>(c, tr_wave, rg, rga, rgb);
        const u32x4 w0 = philox_vk(cq, (uint32_t)rga, c.hs, c.tag_hi, K);
        const u32x4 w1 = philox_vk(cq, (uint32_t)rgb, c.hs, c.tag_hi, K);
        __builtin_amdgcn_sched_barrier(0);
        const uint32_t Sa = SEAM ? (P0 ? nib_shift_next_seam(R1, A0, c.sh_next) : nib_shift_prev_seam(R1, A0, c.sh_prev))
                                 : (P0 ? nib_shift_next(R1, A0) : nib_shift_prev(R1, A0));
        const uint32_t Sb = SEAM ? (P0 ? nib_shift_prev_seam(R2, A1, c.sh_prev) : nib_shift_next_seam(R2, A1, c.sh_next))
                                 : (P0 ? nib_shift_prev(R2, A1) : nib_shift_next(R2, A1));
        const uint32_t mid = R1 + R2;  // shared by both rows' vertical + centre sums
        const uint32_t cnt0 = R0 + mid + Sa, cnt1 = R3 + mid + Sb;
        const uint32_t cnt0l = cnt0 & 0x0F0F0F0Fu, cnt0h = (cnt0 >> 4) & 0x0F0F0F0Fu;
        const uint32_t cnt1l = cnt1 & 0x0F0F0F0Fu, cnt1h = (cnt1 >> 4) & 0x0F0F0F0Fu;
        u32x4 d0, d1;
        uint32_t edge_a = 0, edge_b = 0;
        if (OPEN) {  // as in sweep_pairs: the counts are bytes by now, the degree-3 / degree-2 patches are the byte form's
            const bool ea = rga == 0 || rga == c.last_row, eb = rgb == 0 || rgb == c.last_row;
            const bool la = !P0 && c.lft, ra = c.rgt && c.r_par == (uint32_t)P0, lb = P0 && c.lft, rb = c.rgt && c.r_par != (uint32_t)P0;
            d0 = compare_octet_open(w0, cnt0l, cnt0h, c, ea, la, ra);
            d1 = compare_octet_open(w1, cnt1l, cnt1h, c, eb, lb, rb);
            edge_a = (ea ? 1u : 0u) | (la ? 2u : 0u) | (ra ? 4u : 0u) | (c.r_slot << 4);
            edge_b = (eb ? 1u : 0u) | (lb ? 2u : 0u) | (rb ? 4u : 0u) | (c.r_slot << 4);
        } else {
            d0 = compare_octet(w0, cnt0l, cnt0h, c.tblH0, c.tblH1, c.tblL0, c.tblL1);
            d1 = compare_octet(w1, cnt1l, cnt1h, c.tblH0, c.tblH1, c.tblL0, c.tblL1);
        }
        const uint32_t mn = minu16(minu16(minu16(d0.x, d0.y), minu16(d0.z, d0.w)), minu16(minu16(d1.x, d1.y), minu16(d1.z, d1.w)));
        if (__builtin_expect(((mn & 0xFFFFu) == 0) | ((mn >> 16) == 0), 0)) {
            if (has_zero_field(d0)) d0 = resolve_ties(d0, w0, cnt0l, cnt0h, c.s_thr, cq, (uint32_t)rga, c.hs, c.tag_lo, c.k0, c.k1, edge_a);
            if (has_zero_field(d1)) d1 = resolve_ties(d1, w1, cnt1l, cnt1h, c.s_thr, cq, (uint32_t)rgb, c.hs, c.tag_lo, c.k0, c.k1, edge_b);
        }
        char* pd = const_cast<char*>(ps) + d_off;
        uint32_t n0 = nib_pack(d0), n1 = nib_pack(d1);
        if (OPEN) {  // what lies beyond the open edge stays empty (existing-site masks in the nibble layout)
            const uint64_t ma = P0 ? c.vm_o : c.vm_e, mb = P0 ? c.vm_e : c.vm_o;
            n0 = (rga < 0 || rga > c.last_row) ? 0u : (n0 & ((uint32_t)ma | ((uint32_t)(ma >> 32) << 4)));
            n1 = (rgb < 0 || rgb > c.last_row) ? 0u : (n1 & ((uint32_t)mb | ((uint32_t)(mb >> 32) << 4)));
        }
        if (SEAM) {  // the sites the last octet does not have stay empty
            const uint32_t m = (uint32_t)c.vm_e | ((uint32_t)(c.vm_e >> 32) << 4);
            n0 &= m;
            n1 &= m;
        }
        *reinterpret_cast<uint32_t*>(pd) = n0;
        *reinterpret_cast<uint32_t*>(pd + NO * 4) = n1;
    }
}

// one tile: HBM -> LDS planes, 2k half-sweeps, interior -> HBM (the other buffer)
// Tile-resident generations (RESIDENT): when every tile of the lattice has its own workgroup on the chip at the same
// time, a tile stays in LDS for many generations of k sweeps; after each generation it publishes the 2k interior rows
// at its top and bottom and its first and last interior octet column (the colour planes as they are, ~20 KB) to a
// global exchange buffer and refreshes its halo from its eight neighbours' strips, whose elements carry their
// generation number (no flags: see the exchange in tile_body).  The full-tile stage and store (87 + 64 KB per
// generation) and the launch gap happen once per call instead of once per generation.
// The strips travel through agent-scope relaxed atomic stores and loads (performed at the device's coherence point,
// past the per-CU L1 and the per-XCD L2, like dense_coop.hip's shared data), so no cache write-back / invalidate is
// needed: with release/acquire fences the publish step alone cost 5 us per generation.
static __device__ __forceinline__ void xst(uint64_t* p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
static __device__ __forceinline__ uint64_t xld(const uint64_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

struct ResidentParams {
    uint64_t* xbuf;  // [2 parities][tiles][XSTRIDE] exchange strips; every element carries its generation number (see tile_body)
    uint32_t gen0;   // number of the first generation this launch publishes (1 .. 16383, counted on from launch to launch)
    int* err;        // host-mapped: set if a bounded wait expired (results invalid)
    int n_gen, k_last, tiles_y;
    int wrap_y;      // tile rows wrap (whole periodic lattice); 0 for a slab: its outer tile rows have no neighbour there and
                     // live on the slab's deep ghost rows (the region that is still exact shrinks, as between exchanges),
                     // and for an open lattice: beyond the edge there is nothing, and nothing stays there
    int wrap_x;      // tile columns wrap (periodic); 0 for an open lattice
    long long* dbg;  // TSU_K1_VERBOSE=2: per tile, wall_clock64 ticks spent in [sweeps, publish, wait, fetch]
};

template <int HT, int WO, int THREADS, bool OPEN = false, bool RESIDENT = false, bool NIB = false>
static __device__ __forceinline__ void tile_body(const TiledParams& p, const int8_t* __restrict__ src, int8_t* __restrict__ dst,
                                                 const int k, const uint32_t sweep0, const int tx, const int ty, uint64_t* lds,
                                                 const PhiloxKeys& K, const ResidentParams* R = nullptr) {
    constexpr int NO = WO + 2;
    constexpr int RLMAX = THREADS / NO;
    // tile row ty: rows [row_off, row_off + H) of the launch's range
    int H = p.tile_h, row_off = ty * p.tile_h;
    if (p.flex_ty > 0) {
        const long long half = (p.r_end - p.r_begin + 1) / 2;  // (an open lattice of odd height: the last tile row ends one row past it)
        row_off = 2 * (int)(ty * half / p.flex_ty);
        H = 2 * (int)((ty + 1) * half / p.flex_ty) - row_off;
    }
    const int TR = H + 4 * k;
    // Row lanes: threads [0, RL*NO) sweep (thread = one octet column x every RL-th row pair); all threads load and
    // store.  What a half-sweep costs is the busiest SIMD's sum of wave-iterations: with every lane in use the last,
    // partial iteration occupies the fewest waves and those are consecutive, i.e. spread round-robin over the four
    // SIMDs (+5 % on 4096^2 against the smallest lane count that reaches the same iteration count).
    constexpr int RL = RLMAX;
    using E = typename std::conditional<NIB, uint32_t, uint64_t>::type;  // one octet of a colour plane
    E* plane0 = reinterpret_cast<E*>(lds + 1);
    E* plane1 = plane0 + TR * NO;
    // one spare row: the pair loop reads row idx + 2 NO of the last pair (rounded up to the uint64 grid)
    uint64_t* s_thr = reinterpret_cast<uint64_t*>((reinterpret_cast<uintptr_t>(plane1 + TR * NO + NO + 1) + 7u) & ~(uintptr_t)7u);
    const int tid = threadIdx.x;
    if (tid < 25) s_thr[tid] = p.thr[tid];

    const int q0 = tx * WO + p.q_shift, r0 = p.r_begin + row_off, Rb = r0 - 2 * k;

    // a thread owns one octet column (al = row lane, oct = column) in all three phases: no div/mod in any loop
    const int al = tid / NO, oct = tid - al * NO;
    int cqi = q0 - 1 + oct;
    // open lattice: octets outside the lattice hold no spins (flags 0); their (discarded) results still take a counter
    const bool col_out = OPEN && (cqi < 0 || cqi >= p.nchunks);
    if (!OPEN) {
        if (cqi < 0) cqi += p.nchunks;
        if (cqi >= p.nchunks) cqi -= p.nchunks;
        if (cqi >= p.nchunks) cqi -= p.nchunks;
    }
    const uint32_t cq = (uint32_t)cqi;
    // open lattices of ragged width: the columns of this thread's chunk that exist (all 16, fewer in the last chunk, none outside)
    int nv = p.cols - 16 * cqi;  // (cqi: the octet's own index in the lattice, wrapped for a periodic one)
    if (col_out) nv = 0;
    if (nv > 16) nv = 16;
    auto first_sites = [](int n) { return n >= 8 ? 0x0101010101010101ull : (((1ull << (8 * n)) - 1) & 0x0101010101010101ull); };
    const uint64_t vm_e = first_sites((nv + 1) >> 1), vm_o = first_sites(nv >> 1);

    if (al < RLMAX) {
        const int8_t* col = src + 16 * (long long)(OPEN ? (cqi < 0 ? 0 : (cqi >= p.nchunks ? p.nchunks - 1 : cqi)) : cqi);
        // four rows per step, all four 16-byte loads in flight before the first is converted: the stage is one
        // HBM latency per step, so fewer, wider steps
        constexpr int LB = 4;
        for (int tr0 = al; tr0 < TR; tr0 += LB * RLMAX) {
            uint4 v[LB];
#pragma unroll
            for (int b = 0; b < LB; ++b) {
                const int tr = tr0 + b * RLMAX;
                const int rl = Rb + (tr < TR ? tr : TR - 1);
                int srow;
                if (p.wrap_rows) srow = rl < 0 ? rl + p.rows : (rl >= p.rows ? rl - p.rows : rl);  // rows >= TR: one wrap
                else srow = rl < -p.ghost ? -p.ghost : (rl >= p.rows + p.ghost ? p.rows + p.ghost - 1 : rl);
                v[b] = *reinterpret_cast<const uint4*>(col + (long long)srow * p.pitch);
            }
#pragma unroll
            for (int b = 0; b < LB; ++b) {
                const int tr = tr0 + b * RLMAX;
                if (tr < TR) {
                    // de-interleave the two colours, then +1 (0x01) -> 1, -1 (0xFF) -> 0: bit 1 of the byte, inverted
                    const uint32_t e0 = perm(v[b].y, v[b].x, 0x06040200u), e1 = perm(v[b].w, v[b].z, 0x06040200u);
                    const uint32_t o0 = perm(v[b].y, v[b].x, 0x07050301u), o1 = perm(v[b].w, v[b].z, 0x07050301u);
                    uint64_t ev = (uint64_t)(~(e0 >> 1) & 0x01010101u) | ((uint64_t)(~(e1 >> 1) & 0x01010101u) << 32);
                    uint64_t od = (uint64_t)(~(o0 >> 1) & 0x01010101u) | ((uint64_t)(~(o1 >> 1) & 0x01010101u) << 32);
                    if (OPEN) {
                        const long long grow = p.row0 + Rb + tr;
                        if (grow < 0 || grow >= p.total_rows) ev = od = 0;  // beyond the open edge: nothing there
                    }
                    if (OPEN || nv < 16) {  // the pad bytes of a ragged last chunk are 0, which would read as "up"
                        ev &= vm_e;
                        od &= vm_o;
                    }
                    const int idx = tr * NO + oct;
                    const int gpar = (int)((p.row0 + Rb + tr) & 1);  // colour of the even columns of this row
                    if (NIB) {
                        (gpar ? plane1 : plane0)[idx] = (E)((uint32_t)ev | ((uint32_t)(ev >> 32) << 4));
                        (gpar ? plane0 : plane1)[idx] = (E)((uint32_t)od | ((uint32_t)(od >> 32) << 4));
                    } else {
                        (gpar ? plane1 : plane0)[idx] = (E)ev;
                        (gpar ? plane0 : plane1)[idx] = (E)od;
                    }
                }
            }
        }
    }

    long long rg0 = p.row0 + Rb;
    if (!OPEN) {
        rg0 %= p.total_rows;
        if (rg0 < 0) rg0 += p.total_rows;
    }
    const bool edge = !OPEN && (rg0 + TR > p.total_rows);  // open lattices do not wrap: rows beyond the edge are discarded

    Rows2Ctx c;
    c.s_thr = s_thr;
    c.total_rows = (int)p.total_rows;
    c.rg_tile0 = (int)rg0;
    c.wrap_tr = (int)(p.total_rows - rg0);
    c.tag_hi = p.tag_hi; c.tag_lo = p.tag_lo; c.k0 = p.k0; c.k1 = p.k1;
    // the threshold byte tables live in VGPRs: v_perm_b32 may read only one SGPR, so SGPR tables cost a v_mov per use
    c.tblH0 = p.tblH0; c.tblH1 = p.tblH1; c.tblL0 = p.tblL0; c.tblL1 = p.tblL1;
    c.t3H0 = p.t3H0; c.t3H1 = p.t3H1; c.t3L0 = p.t3L0; c.t3L1 = p.t3L1;
    c.last_row = (int)p.total_rows - 1;
    c.lft = OPEN && cqi == 0;
    c.rgt = OPEN && cqi == p.nchunks - 1;
    c.r_par = (uint32_t)(p.cols - 1) & 1u;
    c.r_slot = ((uint32_t)(p.cols - 1) & 15u) >> 1;
    c.rm_lo = c.r_slot < 4 ? 0xFFu << (8 * c.r_slot) : 0u;
    c.rm_hi = c.r_slot < 4 ? 0u : 0xFFu << (8 * (c.r_slot - 4));
    c.vm_e = vm_e;
    c.vm_o = vm_o;
    // periodic lattice of ragged width: v sites per colour in the last octet of a row (cols is even: the same for both)
    const int v_last = (p.cols >> 1) & 7;
    const bool ragged = !OPEN && v_last != 0;
    const int seam_site = v_last - 1;  // the site of the ragged octet that is followed by site 0 of octet 0
    const int seam_pos = NIB ? 8 * (seam_site & 3) + 4 * (seam_site >> 2) : 8 * seam_site, last_pos = NIB ? 28 : 56;
    c.sh_next = (ragged && cqi == p.nchunks - 1) ? seam_pos : last_pos;
    c.sh_prev = (ragged && cqi == 0) ? seam_pos : last_pos;
    // does this workgroup's window (octets q0 - 1 .. q0 + WO of the periodic extension) hold the ragged octet or octet 0?
    int w0 = (q0 - 1) % p.nchunks;
    if (w0 < 0) w0 += p.nchunks;
    const bool seam = ragged && (w0 == 0 || w0 + NO - 1 >= p.nchunks - 1);
    asm volatile("v_mov_b32 %0, %1" : "=v"(c.tblH0) : "s"(p.tblH0));
    asm volatile("v_mov_b32 %0, %1" : "=v"(c.tblH1) : "s"(p.tblH1));
    asm volatile("v_mov_b32 %0, %1" : "=v"(c.tblL0) : "s"(p.tblL0));
    asm volatile("v_mov_b32 %0, %1" : "=v"(c.tblL1) : "s"(p.tblL1));

    const int n_gen = RESIDENT ? R->n_gen : 1;
    long long tl0 = 0, tl[4] = {0, 0, 0, 0};
    const bool timing = RESIDENT && R->dbg && tid == 0;
    if (timing) tl0 = wall_clock64();
#define RES_MARK(q)                         \
    if (timing) {                           \
        const long long now_ = wall_clock64(); \
        tl[q] += now_ - tl0;                \
        tl0 = now_;                         \
    }
    // RESIDENT: "a neighbour wait expired in this workgroup" -- kept in the guard octet in front of plane 0 (only ever read
    // as "octet -1" padding): a static __shared__ word would push a 160 KB workgroup past the CU's LDS
    volatile int& s_fail = *reinterpret_cast<volatile int*>(lds);
    if (RESIDENT && tid == 0) s_fail = 0;

    // ---- the strip exchange between generations (RESIDENT).  Only colour plane 1 travels: a generation starts with the
    // half-sweep of colour 0, which rewrites every colour-0 site of the tile (halo included) from colour 1 alone.
    // Layout of a tile's slot: TOP[n_tb] BOTTOM[n_tb] LEFT[n_lr] RIGHT[n_lr].
    // Every strip element (8 up-flags, one per byte) carries the 14-bit number of its generation in the spare bits 1..7
    // of its bytes 0 and 1, so a reader validates each element by itself: no "strips complete" flag, no wait for the
    // stores to be acknowledged before raising one, no wait for the flag before fetching -- the refresh is a single
    // round trip that is repeated only for the elements that were not there yet.  (A slot is rewritten every second
    // generation and neighbours are never more than one generation apart, so a stale element always carries another
    // number; the host clears the buffer when the numbering restarts or the strip layout changes.)
    // (nibble planes: an element is the octet's dword, the number sits above it in bits 32..45)
    const int n_tb = 2 * k * WO, n_lr = H;  // elements of a top/bottom and of a left/right strip
    const int xstride = 2 * n_tb + 2 * p.tile_h;  // (tile rows of a flexible cut differ by two rows: one slot size for all)
    // The last tile column of a lattice whose width is not a multiple of the tile width holds wi < WO lattice octets: its last
    // interior octet is tile octet wi, its right halo tile octet wi + 1; what lies beyond is the lattice's periodic continuation
    // from the stage, computed along and never read by anything that counts (it decays like the far side of any halo octet).
    const int wi = (p.nchunks + p.q_shift - q0) < WO ? (p.nchunks + p.q_shift - q0) : WO;
    constexpr uint64_t TAG_MASK = NIB ? 0xFFFFFFFF00000000ull : 0xFEFEull, FLAG_MASK = NIB ? 0x11111111ull : 0x0101010101010101ull;
    auto strip_tag = [&](int gen) -> uint64_t {
        const uint32_t G = R->gen0 + (uint32_t)gen;
        return NIB ? ((uint64_t)G << 32) : (((uint64_t)(G & 0x7Fu) << 1) | ((uint64_t)((G >> 7) & 0x7Fu) << 9));
    };
    // publish generation gen's boundary strips: its first / last 2k interior rows and first / last interior octet column
    auto publish = [&](int gen) {
        RES_MARK(0);
        const int ntiles = p.tiles_x * R->tiles_y, me = ty * p.tiles_x + tx;
        uint64_t* mine = R->xbuf + ((size_t)(gen & 1) * ntiles + me) * xstride;
        const uint64_t tag = strip_tag(gen);
        for (int i = tid; i < n_tb; i += THREADS) {
            const int r = i / WO, o = i - r * WO;
            xst(mine + i, (uint64_t)plane1[(2 * k + r) * NO + 1 + o] | tag);     // TOP: first 2k interior rows
            xst(mine + n_tb + i, (uint64_t)plane1[(H + r) * NO + 1 + o] | tag);  // BOTTOM: last 2k interior rows
        }
        for (int i = tid; i < n_lr; i += THREADS) {
            xst(mine + 2 * n_tb + i, (uint64_t)plane1[(2 * k + i) * NO + 1] | tag);          // LEFT: first interior octet
            xst(mine + 2 * n_tb + n_lr + i, (uint64_t)plane1[(2 * k + i) * NO + wi] | tag);  // RIGHT: last interior octet
        }
        RES_MARK(1);
    };
    // refresh the halo of colour plane 1 from the eight neighbours' strips of generation gen
    auto fetch = [&](int gen) {
        RES_MARK(0);
        const int tiles_x = p.tiles_x, tiles_y = R->tiles_y, ntiles = tiles_x * tiles_y;
        const uint64_t tag = strip_tag(gen);
        const int txl = tx == 0 ? tiles_x - 1 : tx - 1, txr = tx == tiles_x - 1 ? 0 : tx + 1;
        const int w_l = txl == tiles_x - 1 ? p.nchunks + p.q_shift - (txl * WO + p.q_shift) : WO;  // lattice octets of the tile column to the left
        const int tyu = ty == 0 ? tiles_y - 1 : ty - 1, tyd = ty == tiles_y - 1 ? 0 : ty + 1;
        const bool has_u = R->wrap_y || ty > 0, has_d = R->wrap_y || ty < tiles_y - 1;
        const bool has_l = R->wrap_x || tx > 0, has_r = R->wrap_x || tx < tiles_x - 1;
        const uint64_t* xg = R->xbuf + (size_t)(gen & 1) * ntiles * xstride;
        const uint64_t* X_u = xg + (size_t)(tyu * tiles_x + tx) * xstride;
        const uint64_t* X_d = xg + (size_t)(tyd * tiles_x + tx) * xstride;
        const uint64_t* X_l = xg + (size_t)(ty * tiles_x + txl) * xstride;
        const uint64_t* X_r = xg + (size_t)(ty * tiles_x + txr) * xstride;
        const uint64_t* X_ul = xg + (size_t)(tyu * tiles_x + txl) * xstride;
        const uint64_t* X_ur = xg + (size_t)(tyu * tiles_x + txr) * xstride;
        const uint64_t* X_dl = xg + (size_t)(tyd * tiles_x + txl) * xstride;
        const uint64_t* X_dr = xg + (size_t)(tyd * tiles_x + txr) * xstride;
        // halo rows: top [0, 2k) from the BOTTOM strips above, bottom [2k + H, TR) from the TOP strips below; halo octets of the
        // interior rows: octet 0 from the left neighbour's RIGHT strip, octet NO-1 from the right one's LEFT.  Every load of
        // a thread is issued before the first is looked at, and unconditionally: a neighbour index always names a real
        // tile, a missing neighbour's value is neither checked nor stored.
        constexpr int FI = (2 * 8 * NO + THREADS - 1) / THREADS;   // k <= 8
        constexpr int TH_MAX = 160 * 1024 / (int)sizeof(E) / (2 * NO) - 32;  // tallest (stretched slab) tile whose planes fit the CU's LDS
        constexpr int SI = (TH_MAX + THREADS - 1) / THREADS;
        const uint64_t* au[FI];
        const uint64_t* ad[FI];
        uint64_t vu[FI], vd[FI], vl[SI], vr[SI];
        bool nu[FI], nd[FI], nl[SI], nr[SI];  // this thread needs the element
#pragma unroll
        for (int it = 0; it < FI; ++it) {
            const int i = tid + it * THREADS, ic = i < 2 * k * NO ? i : 0;
            const int r = ic / NO, o = ic - r * NO;
            const int e = r * WO;
            const int oc = o <= wi + 1 ? o : 1;  // (octets beyond the right halo: nothing is fetched for them)
            au[it] = oc == 0 ? X_ul + n_tb + e + w_l - 1 : (oc == wi + 1 ? X_ur + n_tb + e : X_u + n_tb + e + oc - 1);
            ad[it] = oc == 0 ? X_dl + e + w_l - 1 : (oc == wi + 1 ? X_dr + e : X_d + e + oc - 1);
            const bool col_ok = o == 0 ? has_l : (o == wi + 1 ? has_r : o <= wi);
            nu[it] = i < 2 * k * NO && has_u && col_ok;
            nd[it] = i < 2 * k * NO && has_d && col_ok;
            vu[it] = xld(au[it]);
            vd[it] = xld(ad[it]);
        }
#pragma unroll
        for (int it = 0; it < SI; ++it) {
            const int i = tid + it * THREADS, ic = i < n_lr ? i : 0;
            nl[it] = i < n_lr && has_l;
            nr[it] = i < n_lr && has_r;
            vl[it] = xld(X_l + 2 * n_tb + n_lr + ic);
            vr[it] = xld(X_r + 2 * n_tb + ic);
        }
        // elements that are not of this generation yet: ask again (a neighbour that finished later; rare and short)
        for (int spins = 0;; ++spins) {
            bool fresh = true;
#pragma unroll
            for (int it = 0; it < FI; ++it) {
                if (nu[it] && (vu[it] & TAG_MASK) != tag) { vu[it] = xld(au[it]); fresh = false; }
                if (nd[it] && (vd[it] & TAG_MASK) != tag) { vd[it] = xld(ad[it]); fresh = false; }
            }
#pragma unroll
            for (int it = 0; it < SI; ++it) {
                const int i = tid + it * THREADS, ic = i < n_lr ? i : 0;
                if (nl[it] && (vl[it] & TAG_MASK) != tag) { vl[it] = xld(X_l + 2 * n_tb + n_lr + ic); fresh = false; }
                if (nr[it] && (vr[it] & TAG_MASK) != tag) { vr[it] = xld(X_r + 2 * n_tb + ic); fresh = false; }
            }
            if (fresh) break;
            // the error flag lives in host memory (a PCIe round trip): look at it rarely
            if (spins > (1 << 22) || ((spins & 0xFFF) == 0xFFF && __hip_atomic_load(R->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM))) {
                __hip_atomic_store(R->err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // ~seconds: grid not co-resident
                s_fail = 1;
                break;
            }
        }
        RES_MARK(2);
#pragma unroll
        for (int it = 0; it < FI; ++it) {
            const int i = tid + it * THREADS;
            if (i < 2 * k * NO) {
                const int r = i / NO, o = i - r * NO;
                if (nu[it]) plane1[r * NO + o] = (E)(vu[it] & FLAG_MASK);
                if (nd[it]) plane1[(2 * k + H + r) * NO + o] = (E)(vd[it] & FLAG_MASK);
            }
        }
#pragma unroll
        for (int it = 0; it < SI; ++it) {
            const int i = tid + it * THREADS;
            if (i < n_lr) {
                if (nl[it]) plane1[(2 * k + i) * NO] = (E)(vl[it] & FLAG_MASK);
                if (nr[it]) plane1[(2 * k + i) * NO + wi + 1] = (E)(vr[it] & FLAG_MASK);
            }
        }
        RES_MARK(3);
    };

    // one call site for every form of the pair loop: rows tr_first, tr_first + step, ... < tr_end of octet column oc
    auto run = [&](int par0, int tr_first, int tr_end, int step, int oc, uint32_t cqq) {
        if (NIB && OPEN) {
            if (par0) sweep_pairs_nib<NO, 1, false, true>(c, K, tr_first, tr_end, step, oc, cqq);
            else sweep_pairs_nib<NO, 0, false, true>(c, K, tr_first, tr_end, step, oc, cqq);
        } else if (NIB && seam) {
            if (par0) sweep_pairs_nib<NO, 1, true, false, true>(c, K, tr_first, tr_end, step, oc, cqq);
            else sweep_pairs_nib<NO, 0, true, false, true>(c, K, tr_first, tr_end, step, oc, cqq);
        } else if (NIB) {
            if (edge) {
                if (par0) sweep_pairs_nib<NO, 1, true>(c, K, tr_first, tr_end, step, oc, cqq);
                else sweep_pairs_nib<NO, 0, true>(c, K, tr_first, tr_end, step, oc, cqq);
            } else {
                if (par0) sweep_pairs_nib<NO, 1, false>(c, K, tr_first, tr_end, step, oc, cqq);
                else sweep_pairs_nib<NO, 0, false>(c, K, tr_first, tr_end, step, oc, cqq);
            }
        } else if (OPEN) {
            if (par0) sweep_pairs<NO, 1, false, true>(c, K, tr_first, tr_end, step, oc, cqq);
            else sweep_pairs<NO, 0, false, true>(c, K, tr_first, tr_end, step, oc, cqq);
        } else if (seam) {
            if (par0) sweep_pairs<NO, 1, true, false, true>(c, K, tr_first, tr_end, step, oc, cqq);
            else sweep_pairs<NO, 0, true, false, true>(c, K, tr_first, tr_end, step, oc, cqq);
        } else if (edge) {
            if (par0) sweep_pairs<NO, 1, true, false>(c, K, tr_first, tr_end, step, oc, cqq);
            else sweep_pairs<NO, 0, true, false>(c, K, tr_first, tr_end, step, oc, cqq);
        } else {
            if (par0) sweep_pairs<NO, 1, false, false>(c, K, tr_first, tr_end, step, oc, cqq);
            else sweep_pairs<NO, 0, false, false>(c, K, tr_first, tr_end, step, oc, cqq);
        }
    };
    int par0 = 0;
    auto set_half_sweep = [&](uint32_t sweep_g, int hsi) {
        const int kappa = hsi & 1;
        c.hs = 2u * (sweep_g + (uint32_t)(hsi >> 1)) + (uint32_t)kappa;
        c.Pd = kappa ? plane1 : plane0;
        c.Ps = kappa ? plane0 : plane1;
        c.tr_lo = 1 + hsi;
        c.npairs = (TR - 2 - 2 * hsi) / 2;  // rows [1 + hsi, TR - 2 - hsi] in pairs (TR is even)
        int rgf = (int)rg0 + c.tr_lo;
        if (!OPEN && rgf >= c.total_rows) rgf -= c.total_rows;
        c.rgf = rgf;
        par0 = (int)((p.row0 + Rb + c.tr_lo + kappa) & 1);
    };
    for (int gen = 0; gen < n_gen; ++gen) {
    const int kg = (RESIDENT && gen == n_gen - 1) ? R->k_last : k;  // sweeps of this generation (the tile keeps TR = H + 4k)
    const uint32_t sweep_g = sweep0 + (uint32_t)(gen * k);
    for (int hsi = 0; hsi < 2 * kg; ++hsi) {
        __syncthreads();
        set_half_sweep(sweep_g, hsi);
        // a thread stays on its own octet column and walks every RL-th row pair of the half-sweep's range
        if (al < RL) run(par0, c.tr_lo + 2 * al, c.tr_lo + 2 * c.npairs, 2 * RL, oct, cq);
    }
    __syncthreads();
    if (RESIDENT && gen + 1 < n_gen) {
        publish(gen);
        fetch(gen);
        __syncthreads();
        if (s_fail) return;  // nothing is stored: the source buffer stays valid
    }
    RES_MARK(0);
    }  // generations
    if (timing)
        for (int q = 0; q < 4; ++q) R->dbg[4 * blockIdx.x + q] = tl[q];
#undef RES_MARK
    // interior octets are tile columns 1 .. NO-2: the same thread -> column mapping, halo columns idle
    if (al < RLMAX && oct >= 1 && oct <= WO && q0 + oct - 1 < p.nchunks + p.q_shift) {
        int8_t* col = dst + 16 * (long long)cqi;  // (= q0 + oct - 1, wrapped when the tiling starts at octet q_shift)
        int gpar = (int)((p.row0 + r0 + al) & 1);
#pragma unroll 2
        for (int hr = al; hr < H; hr += RLMAX) {
            const int rl = r0 + hr;
            if (rl < p.r_end) {
                const int li = (2 * k + hr) * NO + oct;
                uint64_t ev = (gpar ? plane1 : plane0)[li], od = (gpar ? plane0 : plane1)[li];
                if (NIB) {  // nibble octet -> byte-per-site flags
                    ev = (ev & 0x01010101ull) | (((ev >> 4) & 0x01010101ull) << 32);
                    od = (od & 0x01010101ull) | (((od >> 4) & 0x01010101ull) << 32);
                }
                // up flag -> spin byte: 1 -> 0x01, 0 -> 0xFF
                const uint32_t e0 = perm(0u, 0x000001FFu, (uint32_t)ev), e1 = perm(0u, 0x000001FFu, (uint32_t)(ev >> 32));
                const uint32_t o0 = perm(0u, 0x000001FFu, (uint32_t)od), o1 = perm(0u, 0x000001FFu, (uint32_t)(od >> 32));
                uint4 v = make_uint4(perm(o0, e0, 0x05010400u), perm(o0, e0, 0x07030602u), perm(o1, e1, 0x05010400u),
                                     perm(o1, e1, 0x07030602u));
                if (nv < 16) {  // ragged last chunk: the pad bytes of the row stay 0
                    auto bytes = [](int n) { return n >= 4 ? 0xFFFFFFFFu : (n <= 0 ? 0u : (1u << (8 * n)) - 1u); };
                    v.x &= bytes(nv);
                    v.y &= bytes(nv - 4);
                    v.z &= bytes(nv - 8);
                    v.w &= bytes(nv - 12);
                }
                *reinterpret_cast<uint4*>(col + (long long)rl * p.pitch) = v;
            }
            if (RLMAX & 1) gpar ^= 1;
        }
    }
}

template <int H, int WO, int THREADS, int MINW = 1, bool OPEN = false, bool NIB = false>
__global__ __launch_bounds__(THREADS, MINW) void k1_tiled2(TiledParams p) {
    extern __shared__ uint64_t lds[];
    const PhiloxKeys K = make_keys(p.k0, p.k1);
    tile_body<H, WO, THREADS, OPEN, false, NIB>(p, p.src, p.dst, p.k, p.sweep0, blockIdx.x % p.tiles_x,
                                    p.ty_first + (blockIdx.x / p.tiles_x) * p.ty_stride, lds, K);
}

// ------------------------------------------------------------------ tile-resident multi-generation kernel
struct ResidentLaunch {
    TiledParams t;
    ResidentParams r;
};

template <int H, int WO, int THREADS, int MINW = 1, bool OPEN = false, bool NIB = false>
__global__ __launch_bounds__(THREADS, MINW) void k1_resident(ResidentLaunch P) {
    extern __shared__ uint64_t lds[];
    const PhiloxKeys K = make_keys(P.t.k0, P.t.k1);
    tile_body<H, WO, THREADS, OPEN, true, NIB>(P.t, P.t.src, P.t.dst, P.t.k, P.t.sweep0, blockIdx.x % P.t.tiles_x, blockIdx.x / P.t.tiles_x,
                                          lds, K, &P.r);
}

// ------------------------------------------------------------------ host side
namespace {
constexpr int KMAX = 8;  // one halo octet (16 columns) covers 2k <= 16 half-sweeps

struct TileVariant {
    int H, WO, threads;
    void (*kernel)(TiledParams);
    void (*resident)(ResidentLaunch);  // tile-resident multi-generation form (nullptr: not built for this shape)
    void (*open)(TiledParams);       // open-boundary form (nullptr: not built for this shape)
    void (*resident_open)(ResidentLaunch);
    int nib;                         // colour planes at 4 bits per site (periodic lattices, width a multiple of 16)
    int per_cu;                      // workgroups of this shape one CU is meant to hold (LDS share of a stretched slab tile)
};

// LDS of one workgroup: guard octet, two colour planes of TR x NO octets (+ one spare row), 25 thresholds
size_t tile_lds_bytes(const TileVariant& tv, int TR) {
    const size_t es = tv.nib ? 4 : 8, NO = (size_t)tv.WO + 2;
    return 8 + (((size_t)2 * TR * NO + NO + 1) * es + 7) / 8 * 8 + 8 + 25 * sizeof(uint64_t);
}
// the tile shapes the chooser (pick_variant) can reach.  Round 1's table had 32 entries, 24 of them reachable only through
// the TSU_TILE_VARIANT development switch; numbers quoted in profiles/r01_* map as 6 -> 0, 8 -> 1, 9 -> 2, 22 -> 3, 23 -> 4,
// 25 -> 5, 26 -> 6, 27 -> 7
enum { V_64x512_T512 = 0, V_128x512_T512, V_128x512_T1024, V_256x512_T1024, V_128x256_T1024, V_64x512_T1024, V_64x256_T1024, V_32x256_T1024,
       V_N512x512_T1024, V_N256x512_T512 };
const TileVariant kVariants[] = {
    {64, 32, 512, k1_tiled2<64, 32, 512>, k1_resident<64, 32, 512>, k1_tiled2<64, 32, 512, 1, true>},
    {128, 32, 512, k1_tiled2<128, 32, 512>, k1_resident<128, 32, 512>, k1_tiled2<128, 32, 512, 1, true>},
    {128, 32, 1024, k1_tiled2<128, 32, 1024>, k1_resident<128, 32, 1024>, k1_tiled2<128, 32, 1024, 1, true>, k1_resident<128, 32, 1024, 1, true>},
    {256, 32, 1024, k1_tiled2<256, 32, 1024, 4>, k1_resident<256, 32, 1024, 4>, k1_tiled2<256, 32, 1024, 4, true>, k1_resident<256, 32, 1024, 4, true>},  // one 148 KB workgroup per CU
    {128, 16, 1024, k1_tiled2<128, 16, 1024>, k1_resident<128, 16, 1024>, k1_tiled2<128, 16, 1024, 1, true>, k1_resident<128, 16, 1024, 1, true>},  // narrower tiles for mid-size lattices
    {64, 32, 1024, k1_tiled2<64, 32, 1024>, k1_resident<64, 32, 1024>, k1_tiled2<64, 32, 1024, 1, true>, k1_resident<64, 32, 1024, 1, true>},
    {64, 16, 1024, k1_tiled2<64, 16, 1024>, k1_resident<64, 16, 1024>, k1_tiled2<64, 16, 1024, 1, true>, k1_resident<64, 16, 1024, 1, true>},
    {32, 16, 1024, k1_tiled2<32, 16, 1024>, k1_resident<32, 16, 1024>, k1_tiled2<32, 16, 1024, 1, true>, k1_resident<32, 16, 1024, 1, true>},
    // nibble planes: 512 x 512 sites per CU (148 KB) -- 8192^2 tile-resident; 256 x 512, two workgroups per CU, for what is larger still
    {512, 32, 1024, k1_tiled2<512, 32, 1024, 4, false, true>, k1_resident<512, 32, 1024, 4, false, true>, k1_tiled2<512, 32, 1024, 4, true, true>,
     k1_resident<512, 32, 1024, 4, true, true>, 1, 1},
    {256, 32, 512, k1_tiled2<256, 32, 512, 4, false, true>, nullptr, k1_tiled2<256, 32, 512, 4, true, true>, nullptr, 1, 2},
};
constexpr int kNumVariants = sizeof(kVariants) / sizeof(kVariants[0]);

// A periodic lattice whose width is not a multiple of 16 has a ragged last octet (v < 8 sites per colour).  As a halo
// octet it would give a tile 2v columns of history instead of 16, so the tiling starts at octet `shift` such that the
// ragged octet is never the left halo of a tile (tile start = 0 mod nch) nor the right halo of any but the last tile.
// 0: not needed; -1: no such start (the generic kernel takes the lattice).
int ragged_shift(const tsu_ising2d* L, int WO) {
    if (!L->periodic || L->cols % 16 == 0) return 0;
    const int nch = (L->cols + 15) / 16, tiles_x = (nch + WO - 1) / WO;
    for (int shift = 1; shift < WO && shift < nch - 1; ++shift) {
        bool ok = true;
        for (int t = 0; t < tiles_x && ok; ++t) {
            const int s = t * WO + shift;
            if (s % nch == 0) ok = false;
            if (t < tiles_x - 1 && (s + WO) % nch == nch - 1) ok = false;
        }
        if (ok) return shift;
    }
    return -1;
}

// Tile-resident runs of whole lattices of any shape (flexible cut; periodic ones need even height and width): the rows are
// cut into tiles_y tile rows of (nearly) equal even heights and the last tile column may hold fewer octets than the others, so
// the lattice need not divide into whole tiles.  As many tile rows as the chip has room for (every tile needs its own
// workgroup; more tiles = shorter tiles = a shorter generation), the tile shape by the generation-time model of pick_variant.
// Returns tiles_y (0: no such cut) for variant v, or searches the variants when v < 0 and returns the best through *v_out.
int flex_plan(const tsu_ising2d* L, int v, int* v_out, int* hmax_out, double* t_out = nullptr) {
    static int flexible = -1;
    if (flexible < 0) {
        const char* e = getenv("TSU_K1_FLEX_TILES");
        flexible = e ? atoi(e) : 1;
    }
    if (!flexible || L->ghost != 0 || L->total_rows != L->rows) return 0;
    if (L->periodic && ((L->cols & 1) || (L->rows & 1))) return 0;  // (odd periodic lattices are not bipartite: generic kernel)
    static int use_nib = -1;
    if (use_nib < 0) {
        const char* e = getenv("TSU_K1_NIBBLE");
        use_nib = e ? atoi(e) : 1;
    }
    static int FLEX_HMIN = -1;  // shortest tile row of a cut (the strips are 2k = 16 rows deep)
    if (FLEX_HMIN < 0) {
        const char* e = getenv("TSU_K1_FLEX_HMIN");
        FLEX_HMIN = e ? atoi(e) : 16;
        if (FLEX_HMIN < 16) FLEX_HMIN = 16;
    }
    static int max_tiles = -1;  // development / test switch: pretend the chip has room for this many tiles only
    if (max_tiles < 0) {
        const char* e = getenv("TSU_K1_FLEX_MAX_TILES");
        max_tiles = e ? atoi(e) : 0;
    }
    int cus = L->ctx->cus > 0 ? L->ctx->cus : 256;
    if (max_tiles > 0 && max_tiles < cus) cus = max_tiles;
    const int nch = (L->cols + 15) / 16;
    static const int cand[] = {V_256x512_T1024, V_128x256_T1024, V_N512x512_T1024};
    double best = 1e300;
    int best_ty = 0;
    for (int ci = 0; ci < (int)(sizeof(cand) / sizeof(cand[0])); ++ci) {
        if (v >= 0 && cand[ci] != v) continue;
        const TileVariant& c = kVariants[cand[ci]];
        if (!(L->periodic ? c.resident : c.resident_open) || (c.nib && !use_nib) || 2 * nch < c.WO || ragged_shift(L, c.WO) < 0) continue;
        // (an open lattice may have an odd number of rows: the cut is made over rows + 1, the last tile row holds one row less)
        const int tiles_x = (nch + c.WO - 1) / c.WO, half = (L->rows + 1) / 2;
        int tiles_y = cus / tiles_x;
        if (tiles_y > half / (FLEX_HMIN / 2)) tiles_y = half / (FLEX_HMIN / 2);  // a small lattice: not every CU gets a tile
        if (tiles_y < 1) continue;
        const int hmax = 2 * ((half + tiles_y - 1) / tiles_y), hmin = 2 * (half / tiles_y);
        if (hmin < FLEX_HMIN || L->total_rows < hmax + 4 * KMAX || tile_lds_bytes(c, hmax + 4 * KMAX) > 160 * 1024) continue;
        const int pairs = (hmax + 4 * 8 - 2) / 2, waves = (pairs * (c.WO + 2) + 63) / 64;
        const double t_gen = 16.0 * (0.25 + (c.nib ? 1.06 : 1.0) * 0.225 * ((waves + 3) / 4)) + 4.3;
        if (t_gen < best) {
            best = t_gen;
            best_ty = tiles_y;
            if (v_out) *v_out = cand[ci];
            if (hmax_out) *hmax_out = hmax;
        }
    }
    if (t_out) *t_out = best;
    return best_ty;
}

struct TilePlan {
    int v;        // index into kVariants, -1: none fits (generic kernel)
    int flex_ty;  // > 0: tile-resident runs use the flexible cut with this many tile rows ...
    int flex_h;   // ... whose tallest is this high
};

TilePlan tile_plan(const tsu_ising2d* L) {
    int flex_ty = 0, flex_h = 0;
    static int env = -2;
    if (env == -2) {
        const char* e = getenv("TSU_TILE_VARIANT");
        env = e ? atoi(e) : -1;
    }
    int v;
    if (env >= 0 && env < kNumVariants) {
        v = env;
    } else {
        // measured on MI355X (profiles/r01_k1_experiments.txt): 128-row tiles with two 512-thread workgroups per CU
        // when the lattice gives every CU several tiles; a lattice with at most one such tile per CU needs all 16
        // waves of the CU in one workgroup; one with ABOUT TWO per CU (2048 x 16384, 4096 x 8192: everything starts
        // and ends together, nothing hides the stage/store phases) does better with 256-row tiles, one per CU
        const int cus = L->ctx->cus > 0 ? L->ctx->cus : 256;
        const long long tx = (((L->cols + 15) / 16) + 31) / 32;
        const long long n128 = ((L->rows + 127) / 128) * tx, n256 = ((L->rows + 255) / 256) * tx;
        if (n128 <= cus) v = V_128x512_T1024;
        else if (n128 <= 2 * cus && n256 <= cus && 10 * n256 >= 7 * cus) v = V_256x512_T1024;
        else v = V_128x512_T512;
        if (L->rows < 256) v = V_64x512_T512;
        // Nibble planes (whole periodic lattices, width a multiple of 16): a lattice of exactly one 512 x 512 tile per CU
        // (8192^2 on 256 CUs) stays resident in LDS; larger lattices take 256 x 512 tiles, two workgroups per CU, 8 sweeps
        // per launch (half the halo rows of the 128-row byte tiles and fewer stage/store passes)
        static int use_nib = -1;
        if (use_nib < 0) {
            const char* e = getenv("TSU_K1_NIBBLE");
            use_nib = e ? atoi(e) : 1;
        }
        // (any width: an open lattice's ragged last octet is masked like every site beyond the edge, a periodic one's takes the SEAM form)
        const bool nib_ok = use_nib && L->ghost == 0 && L->total_rows == L->rows;
        if (nib_ok && n128 > 2 * cus) {
            const int nch_ = (L->cols + 15) / 16;
            const long long n512 = (long long)(L->rows / 512) * (nch_ / 32);
            if (L->rows % 512 == 0 && nch_ % 32 == 0 && n512 <= cus && 2 * n512 > cus) v = V_N512x512_T1024;
            else if (L->rows >= 512) v = V_N256x512_T512;
        }
        // Lattices that give every CU at most one tile: the tile shape that finishes a generation of 8 sweeps soonest.
        // Model fitted to measurements (4096^2, 2048^2, 1024^2, 4096 x 8192, open 1000^2; profiles/r01_k1_experiments.txt):
        // a half-sweep costs 0.25 us + 0.225 us per wave-iteration of the busiest SIMD; between generations the strip
        // exchange costs 4.3 us when the tiles stay resident in LDS (the lattice divides into whole tiles), the tile
        // store + launch gap + stage about 12 us when they do not.
        static const int cand[] = {V_256x512_T1024, V_128x512_T1024, V_128x256_T1024, V_64x512_T1024, V_64x256_T1024, V_32x256_T1024};
        double best = 1e30;
        const bool whole = L->ghost == 0 && L->total_rows == L->rows;
        const int nch = (L->cols + 15) / 16;
        for (int ci = 0; whole && ci < (int)(sizeof(cand) / sizeof(cand[0])); ++ci) {
            const TileVariant& c = kVariants[cand[ci]];
            if (!(L->periodic ? c.kernel : c.open)) continue;
            if (2 * nch < c.WO || L->total_rows < c.H + 4 * KMAX || ragged_shift(L, c.WO) < 0) continue;
            const long long nt = (long long)((L->rows + c.H - 1) / c.H) * ((nch + c.WO - 1) / c.WO);
            if (nt > cus) continue;
            const bool resident = (L->periodic ? c.resident : c.resident_open) && L->rows % c.H == 0 && nch % c.WO == 0;
            const int pairs = (c.H + 4 * 8 - 2) / 2, waves = (pairs * (c.WO + 2) + 63) / 64;
            const double t_gen = 16.0 * (0.25 + 0.225 * ((waves + 3) / 4)) + (resident ? 4.3 : 12.0);
            if (t_gen < best) {
                best = t_gen;
                v = cand[ci];
            }
        }
    }
    // a whole periodic lattice that the pick above cannot keep resident in LDS (it does not divide into that shape's tiles, or
    // has more of them than the chip has room for) takes the flexible cut if one exists
    // ... and so does one that the pick above keeps resident in fewer, taller tiles than the chip has room for (6144^2 in 512-row
    // nibble tiles: 144 tiles on 256 CUs) when the model gives the flexible cut a generation that is at least 3 % shorter.
    {
        const TileVariant& c = kVariants[v];
        const int nch = (L->cols + 15) / 16, cus = L->ctx->cus > 0 ? L->ctx->cus : 256;
        const long long nt = (long long)((L->rows + c.H - 1) / c.H) * ((nch + c.WO - 1) / c.WO);
        const int per_cu = c.per_cu ? c.per_cu : (c.threads >= 1024 ? 1 : 2);
        const bool divides = L->rows % c.H == 0 && nch % c.WO == 0;
        const bool standard_resident = (L->periodic ? c.resident : c.resident_open) && divides && nt <= (long long)per_cu * cus;
        const bool forced = env >= 0 && env < kNumVariants;
        int fv = -1, fh = 0;
        double t_flex = 0;
        const int fty = forced ? (divides ? 0 : flex_plan(L, v, &fv, &fh, &t_flex)) : flex_plan(L, -1, &fv, &fh, &t_flex);
        bool take = fty > 0 && !standard_resident;
        // a batch of lattices (a temperature scan on side streams) wants many lattices in flight, not the shortest generation of
        // one: lattices that leave half the chip to the others keep the tuned, compact shapes (32 temperatures at 1024^2: 120 ms;
        // with every lattice cut into 256 tiles 182 ms)
        const bool compact = L->ctx->in_batch && 2 * nt <= cus;
        if (compact) take = false;
        if (fty > 0 && standard_resident && !forced && per_cu == 1 && !compact) {
            const int pairs = (c.H + 4 * 8 - 2) / 2, waves = (pairs * (c.WO + 2) + 63) / 64;
            const double t_std = 16.0 * (0.25 + (c.nib ? 1.06 : 1.0) * 0.225 * ((waves + 3) / 4)) + 4.3;
            take = t_flex < 0.97 * t_std;
        }
        if (take) {
            v = fv;
            flex_ty = fty;
            flex_h = fh;
        }
    }
    // a variant must fit the lattice (a tile is a window on the lattice's periodic extension: its octet and row indices
    // wrap at most twice / once) and, for an open lattice, have the OPEN form built; -1 = none does (generic kernel)
    auto fits = [&](int vv) {
        return 2 * ((L->cols + 15) / 16) >= kVariants[vv].WO && L->total_rows >= kVariants[vv].H + 4 * KMAX && (L->periodic || kVariants[vv].open) &&
               ragged_shift(L, kVariants[vv].WO) >= 0;
    };
    if (!fits(v)) {
        static const int fallback[] = {V_64x512_T512, V_64x256_T1024, V_32x256_T1024};
        v = -1;
        flex_ty = flex_h = 0;
        for (int f : fallback)
            if (fits(f)) {
                v = f;
                break;
            }
    }
    return TilePlan{v, flex_ty, flex_h};
}

int pick_variant(const tsu_ising2d* L) { return tile_plan(L).v; }
// Tile height of a launch-per-k-sweeps run over a WHOLE lattice: any even height the variant's LDS share holds gives the same
// results, so it is chosen for the schedule.  The chip runs `slots` tiles at a time and a tile costs about (th + 2k + 12) row
// times (its trapezoid of halo rows, its stage and store), so a lattice costs ceil(tiles / slots) rounds of that: 10000^2 in
// 256-row nibble tiles is 800 tiles on 512 slots -- two rounds, the second 56 % full; in 198-row tiles it is 1020 tiles, two
// full rounds of shorter tiles (-21 %).  Lattices with fewer tiles than slots get shorter tiles so that every slot has one.
int schedule_tile_h(const tsu_ising2d* L, const TileVariant& tv, int k, int tiles_x) {
    static int flexible = -1;
    if (flexible < 0) {
        const char* e = getenv("TSU_K1_FLEX_TILES");
        flexible = e ? atoi(e) : 1;
    }
    if (!flexible) return tv.H;
    const int per_cu = tv.per_cu ? tv.per_cu : (tv.threads >= 1024 ? 1 : 2);
    const long long slots = (long long)per_cu * (L->ctx->cus > 0 ? L->ctx->cus : 256);
    const size_t lds_share = (size_t)(160 / per_cu) * 1024;
    int best_th = tv.H;
    double best = 1e300;
    for (int th = 32; th <= 1024; th += 2) {
        if (th + 4 * KMAX > L->total_rows || tile_lds_bytes(tv, th + 4 * k) > lds_share) continue;
        const long long nt = (long long)((L->rows + th - 1) / th) * tiles_x, rounds = (nt + slots - 1) / slots;
        const double cost = (double)rounds * (th + 2 * k + 12);
        if (cost < best * (1.0 - 1e-9) || (cost <= best * (1.0 + 1e-9) && th > best_th)) {
            best = cost;
            best_th = th;
        }
    }
    return best_th;
}
}  // namespace

int tsu_ising2d_tiled_supported(const tsu_ising2d* L) {
    if (pick_variant(L) < 0) return 0;                        // no tile shape fits (too narrow / too few rows): generic kernel
    const bool open_whole = !L->periodic && L->ghost == 0 && L->total_rows == L->rows;  // beyond its edges: nothing
    if (!L->wrap_rows && !open_whole && L->ghost < 2) return 0;
    return 1;
}

// number of tiles (= workgroups of a tile-resident launch) the lattice is cut into; 0 if the tiled kernel does not apply
int tsu_ising2d_tiled_tiles(const tsu_ising2d* L) {
    if (!tsu_ising2d_tiled_supported(L)) return 0;
    const TilePlan plan = tile_plan(L);
    const TileVariant& tv = kVariants[plan.v];
    const int tiles_x = (((L->cols + 15) / 16) + tv.WO - 1) / tv.WO;
    return (plan.flex_ty > 0 ? plan.flex_ty : (L->rows + tv.H - 1) / tv.H) * tiles_x;
}

// split (interior / boundary) launches: ghost-row slabs whose tile rows are all full
int tsu_ising2d_tiled_part_supported(const tsu_ising2d* L) {
    return tsu_ising2d_tiled_supported(L) && !L->wrap_rows && (L->rows % kVariants[pick_variant(L)].H) == 0;
}

int tsu_ising2d_tiled_sweep(tsu_ising2d* L, int n_sweeps, uint64_t seed, uint32_t sweep0, uint32_t replica, int part) {
    tsu_ctx* ctx = L->ctx;
    const TilePlan plan = tile_plan(L);
    const TileVariant& tv = kVariants[plan.v];
    const int TILE_H = tv.H, TILE_WO = tv.WO;
    if (!L->alloc[1]) {
        size_t bytes = (size_t)(L->rows + 2 * L->ghost) * L->pitch;
        TSU_HIP_TRY(ctx, hipMalloc(&L->alloc[1], bytes));
        TSU_HIP_TRY(ctx, hipMemsetAsync(L->alloc[1], 0, bytes, ctx->stream));
    }
    // sweeps per launch: more sweeps amortise the tile load/store and the launch gap, fewer carry less halo work
    // (one workgroup per CU, or a small lattice: 8; large lattices with two workgroups per CU that overlap each other's
    // stage/store phases: 5)
    int kmax = L->sweeps_per_launch > 0 ? L->sweeps_per_launch
                                        : ((tv.nib || tv.threads >= 1024 || (long long)L->rows * L->cols <= 4096ll * 4096ll) ? 8 : 5);
    if (kmax > KMAX) kmax = KMAX;
    const bool open_whole = !L->periodic && L->ghost == 0 && L->total_rows == L->rows;
    const bool slab = !L->wrap_rows && !open_whole;
    if (slab) {
        // A slab may sweep ghost/2 times between two ghost refreshes.  When that takes several launches, every launch
        // but the last also computes the ghost rows the following launches will read (2 rows per remaining sweep on
        // each side, from input that is still exact there), instead of waiting for the neighbours.
        TSU_REQUIRE(ctx, 2 * n_sweeps <= L->ghost, "ising2d_sweep (tiled, slab): %d sweeps per ghost refresh exceed ghost/2 = %d",
                    n_sweeps, L->ghost / 2);
        TSU_REQUIRE(ctx, part == TSU_PART_ALL || n_sweeps <= kmax, "ising2d_sweep_part: split sweeps take at most %d sweeps", kmax);
    }
    TiledParams p;
    p.pitch = (long long)L->pitch;
    p.rows = L->rows;
    p.nchunks = (L->cols + 15) / 16;
    p.cols = L->cols;
    p.row0 = L->row0;
    p.total_rows = L->total_rows;
    p.wrap_rows = L->wrap_rows;
    p.ghost = L->ghost;
    p.tiles_x = (p.nchunks + TILE_WO - 1) / TILE_WO;
    p.q_shift = ragged_shift(L, TILE_WO);
    p.r_begin = 0;
    p.r_end = L->rows;
    p.tile_h = TILE_H;
    p.flex_ty = 0;
    int tiles_y = (L->rows + TILE_H - 1) / TILE_H;
    p.k0 = (uint32_t)seed;
    p.k1 = (uint32_t)(seed >> 32);
    p.tag_hi = TSU_TAG_ISING_HI | (replica << 8);
    p.tag_lo = TSU_TAG_ISING_LO | (replica << 8);
    for (int c = 0; c < 25; ++c) p.thr[c] = L->table[c];
    p.open = L->periodic ? 0 : 1;
    auto top16 = [](uint64_t thr) {
        uint32_t thi = (uint32_t)(thr >> 16);
        if (thi > 65535u) thi = 65535u;
        return thi ^ 0x8000u;
    };
    uint32_t t16[5], t3[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int c = 0; c < 5; ++c) t16[c] = top16(L->table[4 * 5 + c]);
    for (int c = 0; c < 4; ++c) t3[c] = top16(L->table[3 * 5 + c]);      // degree 3: an edge
    for (int c = 0; c < 3; ++c) t3[4 + c] = top16(L->table[2 * 5 + c]);  // degree 2: a corner
    p.t3L0 = (t3[0] & 0xFF) | ((t3[1] & 0xFF) << 8) | ((t3[2] & 0xFF) << 16) | ((t3[3] & 0xFF) << 24);
    p.t3L1 = (t3[4] & 0xFF) | ((t3[5] & 0xFF) << 8) | ((t3[6] & 0xFF) << 16);
    p.t3H0 = (t3[0] >> 8) | ((t3[1] >> 8) << 8) | ((t3[2] >> 8) << 16) | ((t3[3] >> 8) << 24);
    p.t3H1 = (t3[4] >> 8) | ((t3[5] >> 8) << 8) | ((t3[6] >> 8) << 16);
    p.tblL0 = (t16[0] & 0xFF) | ((t16[1] & 0xFF) << 8) | ((t16[2] & 0xFF) << 16) | ((t16[3] & 0xFF) << 24);
    p.tblL1 = (t16[4] & 0xFF);
    p.tblH0 = (t16[0] >> 8) | ((t16[1] >> 8) << 8) | ((t16[2] >> 8) << 16) | ((t16[3] >> 8) << 24);
    p.tblH1 = (t16[4] >> 8);
    void (*const kern)(TiledParams) = L->periodic ? tv.kernel : tv.open;
    TSU_HIP_TRY(ctx, tsu_func_allow_lds(ctx, (const void*)kern, 160 * 1024));
    static int use_resident = -1;
    if (use_resident < 0) {
        const char* e = getenv("TSU_K1_RESIDENT");
        use_resident = e ? atoi(e) : 1;
    }
    int ntiles = p.tiles_x * tiles_y;
    // whole periodic lattice: tiles of TILE_H rows; periodic slab: its tiles stretched over the ghost rows that the call's
    // later generations need (uniform even height th, th * tiles_y = rows + 2 ext, 2 (n_sweeps - k) <= ext <= ghost)
    int res_th = TILE_H, res_ext = 0;
    void (*const res_kern)(ResidentLaunch) = L->periodic ? tv.resident : tv.resident_open;
    bool res_ok = use_resident && res_kern && part == TSU_PART_ALL && n_sweeps > kmax;
    // whole lattice that does not divide into TILE_H x TILE_WO tiles: the flexible cut (balanced tile rows, partial last column)
    const int flex_ty = res_ok && (L->wrap_rows || open_whole) ? plan.flex_ty : 0, flex_h = plan.flex_h;
    if (flex_ty > 0) {
        tiles_y = flex_ty;
        ntiles = p.tiles_x * tiles_y;
        res_th = flex_h;
    } else if (res_ok) res_ok = p.nchunks % TILE_WO == 0;
    if (flex_ty > 0) {
        // (res_th = the tallest tile row of the cut)
    } else if (res_ok && (L->wrap_rows || open_whole)) res_ok = L->rows % TILE_H == 0;
    else if (res_ok) {
        const int need = L->rows + 4 * (n_sweeps - kmax);
        res_th = (need + tiles_y - 1) / tiles_y;
        res_th += res_th & 1;
        if (res_th < TILE_H) res_th = TILE_H;
        const int extra = res_th * tiles_y - L->rows;
        res_ext = extra / 2;
        const size_t lds_share = (tv.per_cu ? 160u / (unsigned)tv.per_cu : (tv.threads >= 1024 ? 160u : 80u)) * 1024u;
        const size_t lds_need = tile_lds_bytes(tv, res_th + 4 * kmax);
        res_ok = (extra % 2 == 0) && res_ext <= L->ghost && lds_need <= lds_share;
    }
    if (res_ok) {
        // ---- tile-resident generations: every tile has its own workgroup on the chip for the whole call
        const int vi = (int)(&tv - kVariants);
        const size_t lds_bytes = tile_lds_bytes(tv, res_th + 4 * kmax);
        TSU_HIP_TRY(ctx, tsu_func_allow_lds(ctx, (const void*)res_kern, 160 * 1024));
        int fit_per_cu = 0;
        // (occupancy for the standard tile height; a stretched slab tile was checked against the variant's LDS share)
        TSU_HIP_TRY(ctx, tsu_func_blocks_per_cu(ctx, (const void*)res_kern, tv.threads,
                                                tile_lds_bytes(tv, (flex_ty > 0 ? res_th : TILE_H) + 4 * KMAX), &fit_per_cu));
        if ((long long)ntiles <= (long long)fit_per_cu * ctx->cus) {
            const size_t xstride = (size_t)2 * (2 * kmax * TILE_WO) + (size_t)2 * res_th;  // TOP, BOTTOM, LEFT, RIGHT of colour plane 1
            const size_t xneed = (size_t)2 * ntiles * xstride;
            if (L->xbuf_cap < xneed) {
                if (L->d_xbuf) (void)hipFree(L->d_xbuf);
                L->d_xbuf = nullptr;
                L->xbuf_cap = 0;
                TSU_HIP_TRY(ctx, hipMalloc(&L->d_xbuf, xneed * sizeof(uint64_t)));
                L->xbuf_cap = xneed;
                L->xsig = 0;  // fresh memory: cleared before its first use below
            }
            // strip layout of this call: the element numbering may only run on while it stays the same
            const uint64_t xsig = ((uint64_t)(vi + 1) << 48) ^ ((uint64_t)ntiles << 28) ^ ((uint64_t)kmax << 20) ^ ((uint64_t)res_th << 4) ^ ((uint64_t)flex_ty << 36) ^ (uint64_t)p.open;
            if (!L->h_err) {
                TSU_HIP_TRY(ctx, hipHostMalloc(&L->h_err, sizeof(int), hipHostMallocMapped));
                *L->h_err = 0;
            }
            int* d_err = nullptr;
            TSU_HIP_TRY(ctx, hipHostGetDevicePointer((void**)&d_err, L->h_err, 0));
            const char* vb = getenv("TSU_K1_VERBOSE");
            // one launch per 1024 generations at most (a whole lattice may be asked for millions of sweeps; a slab's call
            // is one refresh period anyway)
            const int chunk_max = (L->wrap_rows || open_whole) ? 1024 * kmax : n_sweeps;
            for (int done = 0; done < n_sweeps;) {
                const int chunk = n_sweeps - done < chunk_max ? n_sweeps - done : chunk_max;
                ResidentLaunch P;
                p.k = kmax;
                p.sweep0 = sweep0 + (uint32_t)done;
                p.ty_first = 0;
                p.ty_stride = 1;
                p.src = L->alloc[L->cur] + (size_t)L->ghost * L->pitch;
                p.dst = L->alloc[L->cur ^ 1] + (size_t)L->ghost * L->pitch;
                p.tile_h = res_th;
                p.flex_ty = flex_ty;
                p.r_begin = -res_ext;
                p.r_end = L->rows + res_ext;
                P.t = p;
                P.r.wrap_y = L->wrap_rows ? 1 : 0;
                P.r.wrap_x = L->periodic ? 1 : 0;
                P.r.xbuf = L->d_xbuf;
                const bool renumber = L->xsig != xsig || L->xgen + (uint32_t)((chunk + kmax - 1) / kmax) + 2u >= 16383u;
                if (renumber) {
                    L->xsig = xsig;
                    L->xgen = 0;
                }
                P.r.gen0 = L->xgen + 1u;
                L->xgen += (uint32_t)((chunk + kmax - 1) / kmax);
                P.r.err = d_err;
                P.r.n_gen = (chunk + kmax - 1) / kmax;
                P.r.k_last = chunk - (P.r.n_gen - 1) * kmax;
                P.r.tiles_y = tiles_y;
                P.r.dbg = nullptr;
                long long* d_dbg = nullptr;
                if (vb && atoi(vb) >= 2) {
                    TSU_HIP_TRY(ctx, hipMalloc(&d_dbg, (size_t)4 * ntiles * sizeof(long long)));
                    P.r.dbg = d_dbg;
                }
                if (vb)
                    fprintf(stderr, "[tsu] k1_resident variant %d: %d tiles (%d per CU fit), %d generations of %d sweeps, %zu KB of strips\n",
                            vi, ntiles, fit_per_cu, P.r.n_gen, kmax, xneed * 8 / 1024);
                {
                    const int rcx = tsu_grid_exclusive_begin(ctx);
                    if (rcx != TSU_OK) return rcx;
                }
                // the numbering restarts (or another strip layout starts): no element may look like one of the new run
                // (after the chaining above: an earlier launch of this lattice on another stream has finished with the buffer)
                if (renumber) TSU_HIP_TRY(ctx, hipMemsetAsync(L->d_xbuf, 0, L->xbuf_cap * sizeof(uint64_t), ctx->stream));
                // (a slab's launches alternate with the halo exchange's RCCL kernels on this stream: ordinary launch, see tsu_launch_grid_sync)
                TSU_HIP_TRY(ctx, tsu_launch_grid_sync(ctx, (const void*)res_kern, dim3((unsigned)ntiles), dim3((unsigned)tv.threads), &P, lds_bytes, ctx->stream,
                                                      slab));
                L->launches += 1;
                L->cur ^= 1;
                {
                    const int rcx = tsu_grid_exclusive_end(ctx);
                    if (rcx != TSU_OK) return rcx;
                }
                if (d_dbg) {
                    std::vector<long long> h((size_t)4 * ntiles);
                    (void)hipMemcpy(h.data(), d_dbg, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
                    (void)hipFree(d_dbg);
                    static const char* const what[4] = {"sweeps", "publish", "wait", "fetch"};
                    fprintf(stderr, "[tsu]   us per generation over %d tiles (mean / min / max):", ntiles);
                    for (int q = 0; q < 4; ++q) {
                        double sum = 0, lo = 1e30, hi = 0;
                        for (int t = 0; t < ntiles; ++t) {
                            const double v = h[(size_t)4 * t + q] / 100.0 / P.r.n_gen;
                            sum += v;
                            lo = v < lo ? v : lo;
                            hi = v > hi ? v : hi;
                        }
                        fprintf(stderr, " %s %.1f / %.1f / %.1f%s", what[q], sum / ntiles, lo, hi, q < 3 ? "," : "\n");
                    }
                    if (atoi(vb) >= 3) {  // by XCD (workgroup index mod 8): does one die run behind the others?
                        for (int x = 0; x < 8; ++x) {
                            double sw = 0, wt = 0;
                            int cnt = 0;
                            for (int t = x; t < ntiles; t += 8, ++cnt) {
                                sw += h[(size_t)4 * t] / 100.0 / P.r.n_gen;
                                wt += h[(size_t)4 * t + 2] / 100.0 / P.r.n_gen;
                            }
                            if (cnt) fprintf(stderr, "[tsu]   xcd %d: sweeps %.2f wait %.2f\n", x, sw / cnt, wt / cnt);
                        }
                        if (atoi(vb) >= 4)  // the whole map: sweeps / wait per tile, one line per tile row
                            for (int y = 0; y < tiles_y; ++y) {
                                fprintf(stderr, "[tsu]   row %3d:", y);
                                for (int x = 0; x < p.tiles_x; ++x)
                                    fprintf(stderr, " %.1f/%.1f", h[(size_t)4 * (y * p.tiles_x + x)] / 100.0 / P.r.n_gen,
                                            h[(size_t)4 * (y * p.tiles_x + x) + 2] / 100.0 / P.r.n_gen);
                                fprintf(stderr, "\n");
                            }
                    }
                }
                done += chunk;
            }
            return TSU_OK;
        }
    }
    for (int done = 0; done < n_sweeps;) {
        int k = n_sweeps - done < kmax ? n_sweeps - done : kmax;
        p.k = k;
        p.sweep0 = sweep0 + (uint32_t)done;
        p.src = L->alloc[L->cur] + (size_t)L->ghost * L->pitch;
        p.dst = L->alloc[L->cur ^ 1] + (size_t)L->ghost * L->pitch;
        int TR = TILE_H + 4 * k;
        size_t lds_bytes = tile_lds_bytes(tv, TR);
        if (!slab && part == TSU_PART_ALL) {
            const int th = schedule_tile_h(L, tv, k, p.tiles_x);
            p.tile_h = th;
            tiles_y = (L->rows + th - 1) / th;
            TR = th + 4 * k;
            lds_bytes = tile_lds_bytes(tv, TR);
        }
        if (slab && part == TSU_PART_ALL) {
            const int ext = 2 * (n_sweeps - done - k);  // rows of ghost the remaining sweeps of this refresh period need
            p.r_begin = -ext;
            p.r_end = L->rows + ext;
            // keep the number of tile rows (an extra, nearly empty tile row can cost a whole extra round of tiles):
            // stretch the tiles instead, if the taller tile still fits the LDS share this variant runs with
            const int base_ty = (L->rows + TILE_H - 1) / TILE_H;
            int th = (p.r_end - p.r_begin + base_ty - 1) / base_ty;
            th += th & 1;
            const size_t lds_share = (tv.per_cu ? 160u / (unsigned)tv.per_cu : (tv.threads >= 1024 ? 160u : 80u)) * 1024u;
            const size_t need = tile_lds_bytes(tv, th + 4 * k);
            if (need > lds_share || th < TILE_H) th = TILE_H;
            p.tile_h = th;
            tiles_y = (p.r_end - p.r_begin + th - 1) / th;
            TR = th + 4 * k;
            lds_bytes = tile_lds_bytes(tv, TR);
        }
        // tile rows 0 and tiles_y-1 read ghost rows (2k <= H); the others only read owned rows
        int n_ty = tiles_y;
        p.ty_first = 0;
        p.ty_stride = 1;
        if (part == TSU_PART_INTERIOR) {
            p.ty_first = 1;
            n_ty = tiles_y - 2;
        } else if (part == TSU_PART_BOUNDARY && tiles_y >= 2) {
            p.ty_stride = tiles_y - 1;
            n_ty = 2;
        }
        if (n_ty > 0) {
            hipLaunchKernelGGL(kern, dim3((unsigned)(p.tiles_x * n_ty)), dim3((unsigned)tv.threads), lds_bytes, ctx->stream, p);
            L->launches += 1;
        }
        if (part != TSU_PART_INTERIOR) L->cur ^= 1;  // INTERIOR does not publish; BOUNDARY (or ALL) does
        done += k;
    }
    TSU_HIP_TRY(ctx, hipGetLastError());
    return TSU_OK;
}

// ================================================================== one-workgroup colour-plane kernel (small lattices)
// A whole lattice of at most 1024 octets (32 x 32 ... 128 x 128: the reference's own sizes, BASELINE configs[0]) as two
// colour planes of up-flags in one workgroup's LDS for all sweeps of a call: one thread per octet, the tiled kernel's
// packed-byte update (one Philox block, 16-bit threshold compare, lazy low bits on ties) with true wrap-around
// indices instead of a halo.  Same counters (octet, row, half-sweep), hence the same results as every other lattice
// kernel.  Workgroup b of a launch sweeps lattice b of a batch (one temperature each of a scan).
struct PlanesItem {
    int8_t* buf;  // owned row 0 (updated in place: the lattice lives in LDS between the load and the store)
    long long pitch;
    int rows, cols, periodic;
    uint32_t k0, k1, tag_hi, tag_lo, sweep0;
    uint32_t tblH0, tblH1, tblL0, tblL1, t3H0, t3H1, t3L0, t3L1;
    uint64_t thr[25];
};

template <bool OPEN>
__global__ __launch_bounds__(1024) void k1_planes(const PlanesItem* __restrict__ items, PlanesItem one, int n_sweeps) {
    __shared__ uint64_t s_planes[2 * 1024];
    __shared__ uint64_t s_thr[25];
    const PlanesItem& it = items ? items[blockIdx.x] : one;
    const int tid = threadIdx.x;
    const int rows = it.rows, cols = it.cols, nch = (cols + 15) >> 4, tasks = rows * nch;
    if (tid < 25) s_thr[tid] = it.thr[tid];
    const PhiloxKeys K = make_keys(it.k0, it.k1);
    const bool active = tid < tasks;
    const int r = active ? tid / nch : 0, q = active ? tid - r * nch : 0;
    const int idx = r * nch + q;
    uint64_t* const plane0 = s_planes;
    uint64_t* const plane1 = s_planes + tasks;
    // the columns of this chunk that exist (open lattices of ragged width: fewer in the last chunk)
    int nv = cols - 16 * q;
    if (nv > 16) nv = 16;
    auto first_sites = [](int n) { return n >= 8 ? 0x0101010101010101ull : (((1ull << (8 * n)) - 1) & 0x0101010101010101ull); };
    const uint64_t vm_e = first_sites((nv + 1) >> 1), vm_o = first_sites(nv >> 1);
    int8_t* const chunk = it.buf + (long long)r * it.pitch + 16 * q;
    if (active) {
        const uint4 v = *reinterpret_cast<const uint4*>(chunk);
        const uint32_t e0 = perm(v.y, v.x, 0x06040200u), e1 = perm(v.w, v.z, 0x06040200u);
        const uint32_t o0 = perm(v.y, v.x, 0x07050301u), o1 = perm(v.w, v.z, 0x07050301u);
        uint64_t ev = (uint64_t)(~(e0 >> 1) & 0x01010101u) | ((uint64_t)(~(e1 >> 1) & 0x01010101u) << 32);
        uint64_t od = (uint64_t)(~(o0 >> 1) & 0x01010101u) | ((uint64_t)(~(o1 >> 1) & 0x01010101u) << 32);
        ev &= vm_e;  // the row's pad bytes are 0, which would read as "up"
        od &= vm_o;
        ((r & 1) ? plane1 : plane0)[idx] = ev;  // the even columns of row r have colour r & 1
        ((r & 1) ? plane0 : plane1)[idx] = od;
    }
    // neighbour octets: -1 = beyond an open edge (nothing there)
    const int up = r > 0 ? idx - nch : (OPEN ? -1 : idx + (rows - 1) * nch);
    const int dn = r < rows - 1 ? idx + nch : (OPEN ? -1 : idx - (rows - 1) * nch);
    const int lf = q > 0 ? idx - 1 : (OPEN ? -1 : idx + nch - 1);
    const int rt = q < nch - 1 ? idx + 1 : (OPEN ? -1 : idx - (nch - 1));
    // a periodic lattice of ragged width wraps inside the last octet: the site after its last valid one (v_last - 1) is
    // site 0 of octet 0, the site before site 0 of octet 0 is that last valid one
    const int v_last = ((cols >> 1) & 7) ? ((cols >> 1) & 7) : 8;
    const int shl = q == nch - 1 ? 8 * (v_last - 1) : 56, shr = q == 0 ? 8 * (v_last - 1) : 56;
    Rows2Ctx c;
    c.s_thr = s_thr;
    c.t3H0 = it.t3H0; c.t3H1 = it.t3H1; c.t3L0 = it.t3L0; c.t3L1 = it.t3L1;
    asm volatile("v_mov_b32 %0, %1" : "=v"(c.tblH0) : "s"(it.tblH0));
    asm volatile("v_mov_b32 %0, %1" : "=v"(c.tblH1) : "s"(it.tblH1));
    asm volatile("v_mov_b32 %0, %1" : "=v"(c.tblL0) : "s"(it.tblL0));
    asm volatile("v_mov_b32 %0, %1" : "=v"(c.tblL1) : "s"(it.tblL1));
    c.r_par = (uint32_t)(cols - 1) & 1u;
    c.r_slot = ((uint32_t)(cols - 1) & 15u) >> 1;
    c.rm_lo = c.r_slot < 4 ? 0xFFu << (8 * c.r_slot) : 0u;
    c.rm_hi = c.r_slot < 4 ? 0u : 0xFFu << (8 * (c.r_slot - 4));
    const bool row_edge = OPEN && (r == 0 || r == rows - 1);
    const uint32_t tag_hi = it.tag_hi, tag_lo = it.tag_lo, k0 = it.k0, k1 = it.k1, sweep0 = it.sweep0;
    for (int hsi = 0; hsi < 2 * n_sweeps; ++hsi) {
        __syncthreads();
        const int kappa = hsi & 1;
        const uint32_t hs = 2u * (sweep0 + (uint32_t)(hsi >> 1)) + (uint32_t)kappa;
        if (active) {
            const uint64_t* Ps = kappa ? plane0 : plane1;
            uint64_t* Pd = kappa ? plane1 : plane0;
            const int par = (r + kappa) & 1;  // column parity of the sites updated in this row
            const uint64_t U = (OPEN && up < 0) ? 0ull : Ps[up], D = (OPEN && dn < 0) ? 0ull : Ps[dn], C = Ps[idx];
            // horizontal neighbours: compact bytes (j, j+1) when the parity is 1, (j-1, j) when it is 0
            const int side = par ? rt : lf;
            const uint64_t A = (OPEN && side < 0) ? 0ull : Ps[side];
            const uint64_t S = par ? ((C >> 8) | ((A & 0xFFull) << shl)) : ((C << 8) | ((A >> shr) & 0xFFull));
            const uint64_t cnt = U + D + C + S;  // bytes <= 4: no carries between sites
            const uint32_t cnt_lo = (uint32_t)cnt, cnt_hi = (uint32_t)(cnt >> 32);
            const u32x4 w = philox_vk((uint32_t)q, (uint32_t)r, hs, tag_hi, K);
            u32x4 d;
            uint32_t edge = 0;
            if (OPEN) {
                const bool l0 = !par && q == 0, r7 = q == nch - 1 && c.r_par == (uint32_t)par;
                d = compare_octet_open(w, cnt_lo, cnt_hi, c, row_edge, l0, r7);
                edge = (row_edge ? 1u : 0u) | (l0 ? 2u : 0u) | (r7 ? 4u : 0u) | (c.r_slot << 4);
            } else {
                d = compare_octet(w, cnt_lo, cnt_hi, c.tblH0, c.tblH1, c.tblL0, c.tblL1);
            }
            if (__builtin_expect(has_zero_field(d), 0)) d = resolve_ties(d, w, cnt_lo, cnt_hi, s_thr, (uint32_t)q, (uint32_t)r, hs, tag_lo, k0, k1, edge);
            uint64_t n = pack_flags(d);
            n &= par ? vm_o : vm_e;
            Pd[idx] = n;
        }
    }
    __syncthreads();
    if (active) {
        const uint64_t ev = ((r & 1) ? plane1 : plane0)[idx], od = ((r & 1) ? plane0 : plane1)[idx];
        const uint32_t e0 = perm(0u, 0x000001FFu, (uint32_t)ev), e1 = perm(0u, 0x000001FFu, (uint32_t)(ev >> 32));
        const uint32_t o0 = perm(0u, 0x000001FFu, (uint32_t)od), o1 = perm(0u, 0x000001FFu, (uint32_t)(od >> 32));
        uint4 v = make_uint4(perm(o0, e0, 0x05010400u), perm(o0, e0, 0x07030602u), perm(o1, e1, 0x05010400u), perm(o1, e1, 0x07030602u));
        if (nv < 16) {  // ragged last chunk: the pad bytes of the row stay 0
            auto bytes = [](int n) { return n >= 4 ? 0xFFFFFFFFu : (n <= 0 ? 0u : (1u << (8 * n)) - 1u); };
            v.x &= bytes(nv);
            v.y &= bytes(nv - 4);
            v.z &= bytes(nv - 8);
            v.w &= bytes(nv - 12);
        }
        *reinterpret_cast<uint4*>(chunk) = v;
    }
}

// whole lattices of at most 1024 octets, any width; an open lattice needs two rows and two columns (the packed compare
// knows degrees 4, 3 and 2).  TSU_K1_PLANES=0: never (k1_small).
int tsu_ising2d_planes_supported(const tsu_ising2d* L) {
    static int enabled = -1;
    if (enabled < 0) {
        const char* e = getenv("TSU_K1_PLANES");
        enabled = e ? atoi(e) : 1;
    }
    if (!enabled || L->ghost != 0 || L->total_rows != L->rows || L->row0 != 0) return 0;
    if ((long long)L->rows * ((L->cols + 15) / 16) > 1024) return 0;
    return L->rows >= 2 && L->cols >= 2;
}

// one launch for n lattices of one shape and boundary (workgroup b sweeps lattice b); caller checked planes_supported
int tsu_ising2d_planes_sweep(tsu_ising2d* const* lats, int n, int n_sweeps, const uint64_t* seeds, const uint32_t* sweep0s,
                             const uint32_t* replicas) {
    tsu_ctx* ctx = lats[0]->ctx;
    std::vector<PlanesItem> items((size_t)n);
    for (int i = 0; i < n; ++i) {
        const tsu_ising2d* L = lats[i];
        PlanesItem& it = items[(size_t)i];
        it.buf = L->alloc[L->cur] + (size_t)L->ghost * L->pitch;
        it.pitch = (long long)L->pitch;
        it.rows = L->rows;
        it.cols = L->cols;
        it.periodic = L->periodic;
        it.k0 = (uint32_t)seeds[i];
        it.k1 = (uint32_t)(seeds[i] >> 32);
        it.tag_hi = TSU_TAG_ISING_HI | (replicas[i] << 8);
        it.tag_lo = TSU_TAG_ISING_LO | (replicas[i] << 8);
        it.sweep0 = sweep0s[i];
        for (int c = 0; c < 25; ++c) it.thr[c] = L->table[c];
        auto top16 = [](uint64_t thr) {
            uint32_t thi = (uint32_t)(thr >> 16);
            if (thi > 65535u) thi = 65535u;
            return thi ^ 0x8000u;
        };
        uint32_t t16[5], t3[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int c = 0; c < 5; ++c) t16[c] = top16(L->table[4 * 5 + c]);
        for (int c = 0; c < 4; ++c) t3[c] = top16(L->table[3 * 5 + c]);
        for (int c = 0; c < 3; ++c) t3[4 + c] = top16(L->table[2 * 5 + c]);
        it.t3L0 = (t3[0] & 0xFF) | ((t3[1] & 0xFF) << 8) | ((t3[2] & 0xFF) << 16) | ((t3[3] & 0xFF) << 24);
        it.t3L1 = (t3[4] & 0xFF) | ((t3[5] & 0xFF) << 8) | ((t3[6] & 0xFF) << 16);
        it.t3H0 = (t3[0] >> 8) | ((t3[1] >> 8) << 8) | ((t3[2] >> 8) << 16) | ((t3[3] >> 8) << 24);
        it.t3H1 = (t3[4] >> 8) | ((t3[5] >> 8) << 8) | ((t3[6] >> 8) << 16);
        it.tblL0 = (t16[0] & 0xFF) | ((t16[1] & 0xFF) << 8) | ((t16[2] & 0xFF) << 16) | ((t16[3] & 0xFF) << 24);
        it.tblL1 = (t16[4] & 0xFF);
        it.tblH0 = (t16[0] >> 8) | ((t16[1] >> 8) << 8) | ((t16[2] >> 8) << 16) | ((t16[3] >> 8) << 24);
        it.tblH1 = (t16[4] >> 8);
    }
    const tsu_ising2d* L0c = lats[0];
    const int tasks = L0c->rows * ((L0c->cols + 15) / 16);
    const unsigned threads = (unsigned)((tasks + 63) / 64 * 64);
    void (*const kern)(const PlanesItem*, PlanesItem, int) = L0c->periodic ? k1_planes<false> : k1_planes<true>;
    if (n == 1) {
        hipLaunchKernelGGL(kern, dim3(1), dim3(threads), 0, ctx->stream, (const PlanesItem*)nullptr, items[0], n_sweeps);
    } else {
        tsu_ising2d* L0 = lats[0];  // the staging buffer for the items lives with the first lattice of the batch
        const size_t bytes = items.size() * sizeof(PlanesItem);
        if (L0->batch_cap < bytes) {
            if (L0->d_batch) (void)hipFree(L0->d_batch);
            L0->d_batch = nullptr;
            L0->batch_cap = 0;
            TSU_HIP_TRY(ctx, hipMalloc(&L0->d_batch, bytes));
            L0->batch_cap = bytes;
        }
        // stream order keeps a previous batch launch from still reading the buffer; the host array dies with this call,
        // so the copy is waited for (a few KB); the launch itself stays asynchronous
        TSU_HIP_TRY(ctx, hipMemcpyAsync(L0->d_batch, items.data(), bytes, hipMemcpyHostToDevice, ctx->stream));
        TSU_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        hipLaunchKernelGGL(kern, dim3((unsigned)n), dim3(threads), 0, ctx->stream, (const PlanesItem*)L0->d_batch, items[0], n_sweeps);
    }
    TSU_HIP_TRY(ctx, hipGetLastError());
    for (int i = 0; i < n; ++i) lats[i]->launches += 1;
    return TSU_OK;
}
