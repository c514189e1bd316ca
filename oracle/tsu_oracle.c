/*
 * tsu_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the stochastic spin-update hot path of tsu-emulator, used only by
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker / reported
 * baseline.  Nothing in the shipped package (tsu-emulator_amd/) links, imports or calls it.
 *
 * Two groups of functions:
 *
 *  (A) "reference-order" restatements: the reference algorithm itself (sequential single-site
 *      heat-bath Gibbs on a dense coupling matrix; float64 Langevin step) with the random draws
 *      supplied by the caller, so that replaying NumPy's MT19937 stream reproduces the
 *      reference bit for bit.  Pinned by tests/golden/g1..g5 (generated from the reference).
 *        - ora_sigmoid              <- tsu/gibbs.py:61-77
 *        - ora_dense_sweep_replay   <- tsu/gibbs.py:97-100,124-126,150-162
 *        - ora_dense_energy         <- tsu/gibbs.py:233-236
 *        - ora_langevin_step_f64    <- tsu/core.py:71-80
 *
 *  (B) "device-order" twins: the SAME Markov kernels in the visiting order and with the
 *      counter-based Philox4x32-10 stream that the HIP kernels use (red-black checkerboard for
 *      the lattice; Philox doubles for the dense sweep; Philox + Box-Muller for Langevin).
 *      These define the bit-exact contract for the GPU path (DESIGN.md "RNG stream contract").
 *        - ora_philox4x32_10        (Salmon et al. SC'11; KATs in tests)
 *        - ora_ising2d_thresholds   <- tsu/models/ising.py:138,148 + tsu/gibbs.py:61-77,125
 *        - ora_ising2d_randomize / ora_ising2d_sweep / ora_ising2d_observables
 *        - ora_dense_sweep_philox, ora_sparse_sweep_philox (the same loop on a CSR graph, coloured visiting order)
 *        - ora_langevin_quadratic_f32
 *        - ora_langevin_coupled_f32   (analytic gradient of a coupled quadratic; tsu/core.py:82-98,146-150)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ Philox4x32-10 */
#define PHILOX_M0 0xD2511F53u
#define PHILOX_M1 0xCD9E8D57u
#define PHILOX_W0 0x9E3779B9u
#define PHILOX_W1 0xBB67AE85u

/* ctr[3] stream tags (low byte); bits 8.. carry the replica / chain-group id */
enum {
    TAG_ISING_HI = 0,
    TAG_ISING_LO = 1,
    TAG_INIT = 2,
    TAG_LANGEVIN = 3,
    TAG_DENSE = 4,
    TAG_LANGEVIN_RESTART = 5
};

void ora_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)PHILOX_M0 * c0;
        uint64_t p1 = (uint64_t)PHILOX_M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += PHILOX_W0; k1 += PHILOX_W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static inline void philox(uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint64_t seed, uint32_t out[4]) {
    uint32_t ctr[4] = {a, b, c, d};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    ora_philox4x32_10(ctr, key, out);
}

/* ------------------------------------------------------------------ sigmoid (tsu/gibbs.py:61-77) */
double ora_sigmoid(double x) {
    if (x > 20.0) return 1.0;
    if (x < -20.0) return 0.0;
    return 1.0 / (1.0 + exp(-x));
}

/* ================================================================== (A) reference order */

/*
 * n_sweeps sweeps of sequential / permuted single-site heat-bath on a dense coupling matrix.
 * state: n ints in {0,1} (updated in place; the caller copies as gibbs.py:150 does).
 * order: NULL -> range(n) each sweep, else n_sweeps*n site indices (np.random.permutation rows).
 * uniforms: n_sweeps*n doubles, consumed one per visited site in visiting order (gibbs.py:126).
 * The local field includes the diagonal term J_ii*state_i (np.dot of the full row, gibbs.py:97).
 * The dot product is accumulated left to right in double.
 */
void ora_dense_sweep_replay(int64_t *state, const double *J, const double *bias, int n, double T,
                            int n_sweeps, const int64_t *order, const double *uniforms) {
    for (int s = 0; s < n_sweeps; ++s) {
        for (int k = 0; k < n; ++k) {
            int i = order ? (int)order[(size_t)s * n + k] : k;
            const double *row = J + (size_t)i * n;
            double h = 0.0;
            for (int j = 0; j < n; ++j) h += row[j] * (double)state[j];
            if (bias) h += bias[i];
            double p = ora_sigmoid(h / T);
            state[i] = (uniforms[(size_t)s * n + k] < p) ? 1 : 0;
        }
    }
}

/* E = -1/2 s^T J s - b^T s (tsu/gibbs.py:233-236; also ising.py:112-117 with spins) */
double ora_dense_energy(const int64_t *state, const double *J, const double *bias, int n) {
    double e = 0.0;
    for (int i = 0; i < n; ++i) {
        double acc = 0.0;
        for (int j = 0; j < n; ++j) acc += J[(size_t)i * n + j] * (double)state[j];
        e += (double)state[i] * acc;
    }
    e *= -0.5;
    if (bias)
        for (int i = 0; i < n; ++i) e -= bias[i] * (double)state[i];
    return e;
}

/* x + (-g*dt/gamma) + sqrt(2*T*dt/gamma)*noise   (tsu/core.py:74-80), float64 */
void ora_langevin_step_f64(double *x, const double *grad, const double *noise, int d, double T, double dt,
                           double gamma) {
    double scale = sqrt(2.0 * T * dt / gamma);
    for (int i = 0; i < d; ++i) {
        double drift = -grad[i] * dt / gamma;
        double diff = scale * noise[i];
        x[i] = x[i] + drift + diff;
    }
}

/* ================================================================== (B) device order: lattice */

/*
 * Acceptance thresholds for the uniform nearest-neighbour lattice.
 * table[deg*5 + up], deg = number of existing neighbours (0..4), up = how many of them are +1.
 * In the reference's bit representation (ising.py:138,148; gibbs.py:97-100,125):
 *     field_bit = 4*J*up + bias,  bias = -2h + 2*J*deg  (compat: as shipped, sign bug)
 *                                 bias = +2h - 2*J*deg  (physical: corrected conversion)
 *     p(+1) = sigmoid(field_bit / T)
 * threshold = floor(p * 2^32 + 0.5) in [0, 2^32]; a site becomes +1 iff u32 < threshold.
 * mode: 0 = physical, 1 = compat.
 */
void ora_ising2d_thresholds(double J, double h, double T, int mode, uint64_t table[25]) {
    for (int deg = 0; deg <= 4; ++deg) {
        for (int up = 0; up <= 4; ++up) {
            if (up > deg) { table[deg * 5 + up] = 0; continue; }
            double bias = (mode == 1) ? (-2.0 * h + 2.0 * J * (double)deg) : (2.0 * h - 2.0 * J * (double)deg);
            double field = 4.0 * J * (double)up + bias;
            double p = ora_sigmoid(field / T);
            table[deg * 5 + up] = (uint64_t)floor(p * 4294967296.0 + 0.5);
        }
    }
}

/* i.i.d. +-1 start: bit (c&127) of philox(ctr=(c>>7, r, 0, TAG_INIT|replica<<8), key=seed) */
void ora_ising2d_randomize(int8_t *spins, int rows, int cols, int64_t row0, uint64_t seed, uint32_t replica) {
    uint32_t w[4];
    for (int r = 0; r < rows; ++r) {
        for (int c = 0; c < cols; ++c) {
            if ((c & 127) == 0 || c == 0) philox((uint32_t)(c >> 7), (uint32_t)(row0 + r), 0u, TAG_INIT | (replica << 8), seed, w);
            uint32_t bit = (w[(c & 127) >> 5] >> (c & 31)) & 1u;
            spins[(size_t)r * cols + c] = bit ? 1 : -1;
        }
    }
}

/* 32-bit uniform of site (R = global row, c) in half-sweep hs: hi16 from TAG_ISING_HI (top bit flipped, a
 * bijection that lets the device compare raw Philox halves as SIGNED 16-bit values), lo16 from TAG_ISING_LO */
static inline uint32_t site_uniform(uint32_t R, int c, uint32_t hs, uint64_t seed, uint32_t replica) {
    uint32_t j = (uint32_t)c >> 1, o = j >> 3, m = j & 7;
    uint32_t whi[4], wlo[4];
    philox(o, R, hs, TAG_ISING_HI | (replica << 8), seed, whi);
    philox(o, R, hs, TAG_ISING_LO | (replica << 8), seed, wlo);
    uint32_t hi = ((whi[m >> 1] >> (16 * (m & 1))) & 0xFFFFu) ^ 0x8000u; /* top bit flipped: see DESIGN.md */
    uint32_t lo = (wlo[m >> 1] >> (16 * (m & 1))) & 0xFFFFu;
    return (hi << 16) | lo;
}

/*
 * Red-black checkerboard heat-bath sweeps on a rows x cols lattice of +-1 int8 spins (row-major).
 * Sweep t = sweep0 + i: colour 0 ((r + c) even) then colour 1; every site of a colour sees the
 * other colour's current values.  periodic: wrap both dimensions (a proper 2-colouring needs
 * even rows/cols >= 4; the caller checks).  Open boundary: a missing neighbour lowers deg.
 * The 32-bit uniform is evaluated lazily (hi16 first, lo16 only on a tie with the threshold's
 * top 16 bits); ora_ising2d_sweep_plain below evaluates it in full and must agree bit for bit.
 */
void ora_ising2d_sweep(int8_t *spins, int rows, int cols, int periodic, const uint64_t table[25], int n_sweeps,
                       uint64_t seed, uint32_t sweep0, uint32_t replica) {
    for (int s = 0; s < n_sweeps; ++s) {
        uint32_t t = sweep0 + (uint32_t)s;
        for (int colour = 0; colour < 2; ++colour) {
            uint32_t hs = 2u * t + (uint32_t)colour;
            for (int r = 0; r < rows; ++r) {
                uint32_t whi[4] = {0, 0, 0, 0};
                int cached_o = -1;
                for (int c = (r + colour) & 1; c < cols; c += 2) {
                    int up = 0, deg = 0, rr, cc;
                    rr = r - 1; if (rr < 0 && periodic) rr = rows - 1;
                    if (rr >= 0) { deg++; up += spins[(size_t)rr * cols + c] > 0; }
                    rr = r + 1; if (rr >= rows && periodic) rr = 0;
                    if (rr < rows) { deg++; up += spins[(size_t)rr * cols + c] > 0; }
                    cc = c - 1; if (cc < 0 && periodic) cc = cols - 1;
                    if (cc >= 0) { deg++; up += spins[(size_t)r * cols + cc] > 0; }
                    cc = c + 1; if (cc >= cols && periodic) cc = 0;
                    if (cc < cols) { deg++; up += spins[(size_t)r * cols + cc] > 0; }
                    uint64_t thr = table[deg * 5 + up];
                    uint32_t j = (uint32_t)c >> 1, o = j >> 3, m = j & 7;
                    if ((int)o != cached_o) {
                        philox(o, (uint32_t)r, hs, TAG_ISING_HI | (replica << 8), seed, whi);
                        cached_o = (int)o;
                    }
                    uint64_t hi = ((whi[m >> 1] >> (16 * (m & 1))) & 0xFFFFu) ^ 0x8000u;
                    uint64_t thi = thr >> 16; /* 0..65536 */
                    int accept;
                    if (hi < thi) accept = 1;      /* hi*65536 + lo < thi*65536 <= thr */
                    else if (hi > thi) accept = 0; /* hi*65536 + lo >= (thi+1)*65536 > thr */
                    else accept = (uint64_t)site_uniform((uint32_t)r, c, hs, seed, replica) < thr;
                    spins[(size_t)r * cols + c] = accept ? 1 : -1;
                }
            }
        }
    }
}

/* Straightforward (non-lazy) version used to cross-check the lazy evaluation above. */
void ora_ising2d_sweep_plain(int8_t *spins, int rows, int cols, int periodic, const uint64_t table[25],
                             int n_sweeps, uint64_t seed, uint32_t sweep0, uint32_t replica) {
    for (int s = 0; s < n_sweeps; ++s) {
        uint32_t t = sweep0 + (uint32_t)s;
        for (int colour = 0; colour < 2; ++colour) {
            uint32_t hs = 2u * t + (uint32_t)colour;
            for (int r = 0; r < rows; ++r)
                for (int c = (r + colour) & 1; c < cols; c += 2) {
                    int up = 0, deg = 0, rr, cc;
                    rr = r - 1; if (rr < 0 && periodic) rr = rows - 1;
                    if (rr >= 0) { deg++; up += spins[(size_t)rr * cols + c] > 0; }
                    rr = r + 1; if (rr >= rows && periodic) rr = 0;
                    if (rr < rows) { deg++; up += spins[(size_t)rr * cols + c] > 0; }
                    cc = c - 1; if (cc < 0 && periodic) cc = cols - 1;
                    if (cc >= 0) { deg++; up += spins[(size_t)r * cols + cc] > 0; }
                    cc = c + 1; if (cc >= cols && periodic) cc = 0;
                    if (cc < cols) { deg++; up += spins[(size_t)r * cols + cc] > 0; }
                    uint32_t u = site_uniform((uint32_t)r, c, hs, seed, replica);
                    spins[(size_t)r * cols + c] = ((uint64_t)u < table[deg * 5 + up]) ? 1 : -1;
                }
        }
    }
}

/* sum of spins and sum over bonds s_i*s_j (right + down neighbours, wrap bonds when periodic) */
void ora_ising2d_observables(const int8_t *spins, int rows, int cols, int periodic, int64_t *sum_s, int64_t *sum_bonds) {
    int64_t ss = 0, sb = 0;
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) {
            int v = spins[(size_t)r * cols + c];
            ss += v;
            int cc = c + 1, rr = r + 1;
            if (cc < cols) sb += v * spins[(size_t)r * cols + cc];
            else if (periodic) sb += v * spins[(size_t)r * cols];
            if (rr < rows) sb += v * spins[(size_t)rr * cols + c];
            else if (periodic) sb += v * spins[c];
        }
    *sum_s = ss;
    *sum_bonds = sb;
}

/* ================================================================== (B) device order: dense */

/* 53-bit uniform of dense site i in sweep t (NumPy's double construction from two u32) */
double ora_dense_uniform(uint32_t i, uint32_t t, uint64_t seed, uint32_t replica) {
    uint32_t w[4];
    philox(i >> 1, 0u, t, TAG_DENSE | (replica << 8), seed, w);
    uint32_t a = w[2 * (i & 1)] >> 5, b = w[2 * (i & 1) + 1] >> 6;
    return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
}

/* as ora_dense_sweep_replay but with the Philox uniform keyed by (site, sweep); int8 {0,1} state */
void ora_dense_sweep_philox(int8_t *state, const double *J, const double *bias, int n, double T, int n_sweeps,
                            const int64_t *order, uint64_t seed, uint32_t sweep0, uint32_t replica) {
    for (int s = 0; s < n_sweeps; ++s) {
        for (int k = 0; k < n; ++k) {
            int i = order ? (int)order[(size_t)s * n + k] : k;
            const double *row = J + (size_t)i * n;
            double h = 0.0;
            for (int j = 0; j < n; ++j) h += row[j] * (double)state[j];
            if (bias) h += bias[i];
            double p = ora_sigmoid(h / T);
            state[i] = (ora_dense_uniform((uint32_t)i, sweep0 + (uint32_t)s, seed, replica) < p) ? 1 : 0;
        }
    }
}

/*
 * Sparse twin of ora_dense_sweep_philox (K5): the same sequential heat-bath loop (gibbs.py:128-162) with the local field
 * summed over the CSR row of the site (columns ascending, diagonal entry included when present: gibbs.py:97) instead of
 * the dense row.  order: n site indices visited each sweep (the colour-major order of a proper colouring: sites of one
 * colour do not read each other, so this sequential loop equals the device's colour-parallel update).
 */
void ora_sparse_sweep_philox(int8_t *state, const int64_t *row_ptr, const int32_t *col, const double *val, const double *bias,
                             int n, double T, int n_sweeps, const int32_t *order, uint64_t seed, uint32_t sweep0,
                             uint32_t replica) {
    for (int s = 0; s < n_sweeps; ++s) {
        for (int k = 0; k < n; ++k) {
            int i = order ? order[k] : k;
            double h = 0.0;
            for (int64_t e = row_ptr[i]; e < row_ptr[i + 1]; ++e) h += val[e] * (double)state[col[e]];
            if (bias) h += bias[i];
            double p = ora_sigmoid(h / T);
            state[i] = (ora_dense_uniform((uint32_t)i, sweep0 + (uint32_t)s, seed, replica) < p) ? 1 : 0;
        }
    }
}

/* -1/2 s^T J s - b^T s on the CSR graph (gibbs.py:215-236) */
double ora_sparse_energy(const int8_t *state, const int64_t *row_ptr, const int32_t *col, const double *val, const double *bias, int n) {
    double e = 0.0;
    for (int i = 0; i < n; ++i) {
        double h = 0.0;
        for (int64_t q = row_ptr[i]; q < row_ptr[i + 1]; ++q) h += val[q] * (double)state[col[q]];
        e += -0.5 * (double)state[i] * h - (bias ? bias[i] : 0.0) * (double)state[i];
    }
    return e;
}

/* ================================================================== (B) device order: Langevin */

/* four standard normals of (quad q, chain, step): two Box-Muller pairs from one Philox call (float32) */
void ora_langevin_normals_f32(uint32_t q, uint32_t chain, uint32_t step, uint32_t tag, uint64_t seed, float out[4]) {
    uint32_t w[4];
    philox(q, chain, step, tag, seed, w);
    for (int p = 0; p < 2; ++p) {
        /* u1 in (0,1], u2 in [0,1): 24-bit mantissas */
        float u1 = ((float)(w[2 * p] >> 8) + 1.0f) * (1.0f / 16777216.0f);
        float u2 = (float)(w[2 * p + 1] >> 8) * (1.0f / 16777216.0f);
        float rad = sqrtf(-2.0f * logf(u1));
        float ang = 6.283185307179586f * u2;
        out[2 * p] = rad * cosf(ang);
        out[2 * p + 1] = rad * sinf(ang);
    }
}

/*
 * n_steps of overdamped Langevin on the separable quadratic energy E = 1/2 sum_i k_i (x_i - mu_i)^2
 * (grad_i = k_i (x_i - mu_i)), float32, for n_chains independent chains of dimension dim.
 * x: n_chains*dim.  k, mu: dim each.  Step s uses Philox counter step0 + s.
 * Update (core.py:74-80): x <- x + (-g*(dt/gamma)) + sqrt(2 T dt/gamma) * xi, evaluated as
 * fmaf(scale, xi, fmaf(-g, dt/gamma, x)) so that host and device round identically.
 * traj (nullable): n_steps*n_chains*dim, state after every step.
 */
void ora_langevin_quadratic_f32(float *x, const float *k, const float *mu, int n_chains, int dim, int n_steps,
                                float dt, float gamma, float T, uint64_t seed, uint32_t step0, uint32_t chain0,
                                float *traj) {
    float scale = sqrtf(2.0f * T * dt / gamma);
    float a = dt / gamma;
    for (int s = 0; s < n_steps; ++s)
        for (int ch = 0; ch < n_chains; ++ch)
            for (int i = 0; i < dim; ++i) {
                float nrm[4];
                ora_langevin_normals_f32((uint32_t)i >> 2, chain0 + (uint32_t)ch, step0 + (uint32_t)s, TAG_LANGEVIN, seed, nrm);
                size_t idx = (size_t)ch * dim + i;
                float g = k[i] * (x[idx] - mu[i]);
                float v = fmaf(scale, nrm[i & 3], fmaf(-g, a, x[idx]));
                x[idx] = v;
                if (traj) traj[((size_t)s * n_chains + ch) * dim + i] = v;
            }
}


/*
 * The same for a COUPLED quadratic energy E = 1/2 x^T A x + b^T x (A symmetric, dim*dim row-major): grad = A x + b, the analytic form
 * of what the reference's central differences (tsu/core.py:82-98) return for such an energy.  Every element of the new state is
 * computed from the whole OLD state (core.py:146-150: the gradient is taken at x, then x is replaced).  The dot products are
 * accumulated in double and rounded to float once -- the device sums in float in its own order: tolerance in the tests.
 */
void ora_langevin_coupled_f32(float *x, const float *A, const float *b, int n_chains, int dim, int n_steps, float dt, float gamma,
                              float T, uint64_t seed, uint32_t step0, uint32_t chain0, float *traj) {
    float scale = sqrtf(2.0f * T * dt / gamma);
    float a = dt / gamma;
    float *xn = (float *)malloc((size_t)dim * sizeof(float));
    for (int s = 0; s < n_steps; ++s)
        for (int ch = 0; ch < n_chains; ++ch) {
            float *xc = x + (size_t)ch * dim;
            for (int i = 0; i < dim; ++i) {
                float nrm[4];
                ora_langevin_normals_f32((uint32_t)i >> 2, chain0 + (uint32_t)ch, step0 + (uint32_t)s, TAG_LANGEVIN, seed, nrm);
                double acc = 0.0;
                for (int j = 0; j < dim; ++j) acc += (double)A[(size_t)i * dim + j] * (double)xc[j];
                float g = (float)(acc + (b ? (double)b[i] : 0.0));
                xn[i] = fmaf(scale, nrm[i & 3], fmaf(-g, a, xc[i]));
            }
            memcpy(xc, xn, (size_t)dim * sizeof(float));
            if (traj) memcpy(traj + ((size_t)s * n_chains + ch) * dim, xn, (size_t)dim * sizeof(float));
        }
    free(xn);
}

/* test helper: the 32-bit uniform every site would use in half-sweep hs (row-major, rows*cols) */
void ora_ising2d_site_uniforms(uint32_t *out, int rows, int cols, uint32_t hs, uint64_t seed, uint32_t replica) {
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) out[(size_t)r * cols + c] = site_uniform((uint32_t)r, c, hs, seed, replica);
}

/*
 * Window form of ora_ising2d_sweep, for slab-decomposition tests: `spins` holds block_rows consecutive rows of a
 * total_rows x cols lattice starting at global row `row_global0` (taken modulo total_rows when periodic).  Rows
 * outside the window are unknown and treated as absent, so after h half-sweeps only window rows [h, block_rows-h)
 * are exact -- unless the window edge is the lattice's own open edge, or the window is the whole periodic lattice
 * (then rows wrap inside it).  RNG counters and colours use GLOBAL coordinates, which is what makes any row
 * decomposition reproduce the single-lattice trajectory bit for bit.
 */
void ora_ising2d_sweep_window(int8_t *spins, int block_rows, int cols, int64_t row_global0, int64_t total_rows,
                              int periodic, const uint64_t table[25], int n_sweeps, uint64_t seed, uint32_t sweep0,
                              uint32_t replica) {
    const int whole = (block_rows == total_rows);
    for (int s = 0; s < n_sweeps; ++s) {
        uint32_t t = sweep0 + (uint32_t)s;
        for (int colour = 0; colour < 2; ++colour) {
            uint32_t hs = 2u * t + (uint32_t)colour;
            for (int r = 0; r < block_rows; ++r) {
                int64_t R = row_global0 + r;
                if (periodic) { R %= total_rows; if (R < 0) R += total_rows; }
                if (R < 0 || R >= total_rows) continue; /* window row outside an open lattice */
                for (int c = (int)((R + colour) & 1); c < cols; c += 2) {
                    int up = 0, deg = 0, rr, cc;
                    rr = r - 1; if (rr < 0 && periodic && whole) rr = block_rows - 1;
                    if (rr >= 0 && (periodic || R - 1 >= 0)) { deg++; up += spins[(size_t)rr * cols + c] > 0; }
                    rr = r + 1; if (rr >= block_rows && periodic && whole) rr = 0;
                    if (rr < block_rows && (periodic || R + 1 < total_rows)) { deg++; up += spins[(size_t)rr * cols + c] > 0; }
                    cc = c - 1; if (cc < 0 && periodic) cc = cols - 1;
                    if (cc >= 0) { deg++; up += spins[(size_t)r * cols + cc] > 0; }
                    cc = c + 1; if (cc >= cols && periodic) cc = 0;
                    if (cc < cols) { deg++; up += spins[(size_t)r * cols + cc] > 0; }
                    uint32_t u = site_uniform((uint32_t)R, c, hs, seed, replica);
                    spins[(size_t)r * cols + c] = ((uint64_t)u < table[deg * 5 + up]) ? 1 : -1;
                }
            }
        }
    }
}
