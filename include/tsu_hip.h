/*
 * tsu_hip.h -- C ABI of libtsu_hip.so: the MI355X (gfx950) stochastic spin-update hot path.
 *
 * The reference (Arsham-001/tsu-emulator) is pure Python/NumPy and has NO FFI: its boundary for this
 * path is the Python method surface of tsu/gibbs.py, tsu/models/ising.py and tsu/core.py.  This header
 * is the boundary a host-language binding would bind instead; every entry point names the reference
 * interface it replaces (file:line under /root/reference).  INTEGRATION.md shows the ctypes stub.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++ / torch types.
 *   - every function returns int: TSU_OK or a negative TSU_E_*; tsu_last_error() gives the text.
 *     No exceptions or longjmp cross the ABI.
 *   - the caller owns all host buffers (never retained past the call); the library owns device
 *     memory behind opaque handles with explicit create/destroy.
 *   - handles are not thread-safe; one tsu_ctx per process per device.
 *   - all work is enqueued on the ctx stream (tsu_set_stream; default: the null stream).  Calls that
 *     copy results to host memory synchronise that stream before returning; pure device calls
 *     (sweep, step, randomize) are asynchronous.
 *   - randomness is counter-based Philox4x32-10 keyed by (seed, position, sweep/step counter): results
 *     do not depend on launch geometry, tile size, number of sweeps per call, or slab decomposition.
 */
#ifndef TSU_HIP_H
#define TSU_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TSU_OK 0
#define TSU_E_INVALID (-1)     /* bad argument (Python side raises ValueError / ConfigurationError) */
#define TSU_E_NOMEM (-2)       /* host or device allocation failed */
#define TSU_E_HIP (-3)         /* a HIP runtime call failed */
#define TSU_E_RCCL (-4)        /* RCCL not loadable / a communicator call failed */
#define TSU_E_UNSUPPORTED (-5) /* valid request this build has no kernel for */

#define TSU_MODE_PHYSICAL 0 /* corrected spin->bit bias: P(+1) = sigmoid(2 (J nsum + h) / T) */
#define TSU_MODE_COMPAT 1   /* bias exactly as tsu/models/ising.py:148 (bug-for-bug) */

#define TSU_DTYPE_F64 0
#define TSU_DTYPE_F32 1

#define TSU_KERNEL_AUTO 0    /* pick the fastest kernel that supports the lattice */
#define TSU_KERNEL_GENERIC 1 /* one colour per launch, global memory, any shape / boundary */
#define TSU_KERNEL_TILED 2   /* LDS-staged halo tiles, several sweeps per launch */
#define TSU_KERNEL_SMALL 3   /* whole (small) lattice resident in one workgroup's LDS, all sweeps in one launch */

typedef struct tsu_ctx tsu_ctx;
typedef struct tsu_ising2d tsu_ising2d;
typedef struct tsu_dense tsu_dense;
typedef struct tsu_langevin tsu_langevin;
typedef struct tsu_sparse tsu_sparse;
typedef struct tsu_comm tsu_comm;

/* ------------------------------------------------------------------ context */
int tsu_version(void);
/* device < 0: use the current HIP device.  Fails with TSU_E_HIP when no GPU is present. */
int tsu_init(int device, tsu_ctx** ctx);
int tsu_shutdown(tsu_ctx* ctx);
/* last error text of this ctx (ctx == NULL: of the last failed tsu_init on this thread) */
const char* tsu_last_error(const tsu_ctx* ctx);
/* run on the caller's hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); NULL = null stream */
int tsu_set_stream(tsu_ctx* ctx, void* hip_stream);
int tsu_synchronize(tsu_ctx* ctx);
int tsu_device_info(tsu_ctx* ctx, char* name, int name_len, int* compute_units, uint64_t* hbm_bytes);
/* hipEvent pair on the ctx stream: begin records, end records + synchronises and returns elapsed ms */
int tsu_timer_begin(tsu_ctx* ctx);
int tsu_timer_end(tsu_ctx* ctx, float* elapsed_ms);
/* n Philox4x32-10 blocks evaluated ON THE DEVICE (known-answer tests): ctrs n*4, key 2, out n*4 */
int tsu_philox4x32_10(tsu_ctx* ctx, int n, const uint32_t* ctrs, const uint32_t* key, uint32_t* out);

/* ------------------------------------------------------------------ 2-D lattice (K1, K4)
 * Replaces, for IsingGrid-shaped problems, the dense path
 *   IsingGrid.__init__           tsu/models/ising.py:320-361   (lattice instead of an N x N matrix)
 *   IsingModel.sample            tsu/models/ising.py:150-181
 *   GibbsSampler.gibbs_sweep     tsu/gibbs.py:128-162          (one call of tsu_ising2d_sweep)
 *   GibbsSampler.sample_conditional / _compute_local_field / _sigmoid   tsu/gibbs.py:61-126
 *   IsingModel.energy / magnetization   tsu/models/ising.py:99-117,183-193  (tsu_ising2d_observables)
 * Visiting order is red-black checkerboard (colour (row+col)&1 == 0 first), not raster order: the
 * same heat-bath kernel and stationary distribution, a different trajectory.
 */

/* Host helper: acceptance thresholds table[deg*5 + up] in [0, 2^32] for coupling J, uniform field h,
 * temperature T (gibbs.py:61-77,125 with ising.py:138,148 folded in).  Pure host arithmetic. */
int tsu_ising2d_thresholds(double J, double h, double T, int mode, uint64_t table[25]);

/* A whole rows x cols lattice on one GPU.  periodic needs even rows, cols >= 4 (2-colourability). */
int tsu_ising2d_create(tsu_ctx* ctx, int rows, int cols, int periodic, tsu_ising2d** out);
/* A row slab [row0, row0+rows) of a total_rows x cols lattice with `ghost` ghost rows on each side
 * that the host refreshes from the neighbouring ranks (see tsu_ising2d_row_ptr).  ghost must be even
 * and >= 2; at most ghost/2 sweeps may run between two refreshes. */
int tsu_ising2d_create_slab(tsu_ctx* ctx, int64_t total_rows, int cols, int periodic, int64_t row0, int rows,
                            int ghost, tsu_ising2d** out);
int tsu_ising2d_destroy(tsu_ising2d* lat);

/* +-1 int8 spins, row-major, `cols` bytes per row, local rows [row_first, row_first + n_rows) */
int tsu_ising2d_set_spins(tsu_ising2d* lat, const int8_t* host, int row_first, int n_rows);
int tsu_ising2d_get_spins(tsu_ising2d* lat, int8_t* host, int row_first, int n_rows);
/* i.i.d. +-1 from Philox (replaces np.random.randint(0,2,N), gibbs.py:201, for lattices too big to stage) */
int tsu_ising2d_randomize(tsu_ising2d* lat, uint64_t seed, uint32_t replica);
int tsu_ising2d_fill(tsu_ising2d* lat, int8_t value);

/* thresholds used by the following sweeps: explicit table, or J/h/T/mode through the helper above.
 * T is a per-call quantity because callers mutate config.temperature in place (gibbs.py:382). */
int tsu_ising2d_set_thresholds(tsu_ising2d* lat, const uint64_t table[25]);
int tsu_ising2d_set_model(tsu_ising2d* lat, double J, double h, double T, int mode);
int tsu_ising2d_set_kernel(tsu_ising2d* lat, int kernel, int sweeps_per_launch);

/* n_sweeps full checkerboard sweeps, sweep counters sweep0 .. sweep0+n_sweeps-1 (asynchronous).
 * For a slab: ghost rows must be fresh on entry and n_sweeps <= ghost/2. */
int tsu_ising2d_sweep(tsu_ising2d* lat, int n_sweeps, uint64_t seed, uint32_t sweep0, uint32_t replica);

/* Split form of tsu_ising2d_sweep for slabs, so the host can overlap the halo exchange with compute:
 * TSU_PART_INTERIOR updates the tile rows that do not read ghost rows (may run while ghosts are in flight and
 * does not publish anything); TSU_PART_BOUNDARY updates the remaining tile rows and publishes the new state
 * (after it, tsu_ising2d_row_ptr / get_spins see the swept lattice).  Call INTERIOR then BOUNDARY with the same
 * arguments; TSU_PART_ALL is tsu_ising2d_sweep.  TSU_E_UNSUPPORTED when the lattice is not on the tiled kernel. */
#define TSU_PART_ALL 0
#define TSU_PART_INTERIOR 1
#define TSU_PART_BOUNDARY 2
int tsu_ising2d_sweep_part(tsu_ising2d* lat, int n_sweeps, uint64_t seed, uint32_t sweep0, uint32_t replica, int part);

/* sum_s = sum of spins, sum_bonds = sum over nearest-neighbour bonds of s_i*s_j (owned rows; the bond to
 * the row below the slab is included when that row exists).  M = sum_s/N, E = -J*sum_bonds - h*sum_s. */
int tsu_ising2d_observables(tsu_ising2d* lat, int64_t* sum_s, int64_t* sum_bonds);

/* The sampling loop of IsingModel.sample / GibbsSampler.sample_boltzmann (ising.py:150-181, gibbs.py:203-211) on the
 * lattice: n_burnin sweeps, then n_samples x (n_sweeps sweeps, record).  samples_host receives n_samples * rows * cols
 * spins (row-major, no padding).  Sweep numbers count on from sweep0; the states are gathered on the device and cross
 * PCIe once. */
int tsu_ising2d_sample(tsu_ising2d* lat, int n_burnin, int n_sweeps, int n_samples, uint64_t seed, uint32_t sweep0,
                       uint32_t replica, int8_t* samples_host);

/* Many independent lattices at once (one per temperature of a scan, ising.py:424-476; replicas of a tempering
 * ladder): lattice i does n_sweeps sweeps with its own thresholds, seeds[i], sweep0s[i], replicas[i] -- the same
 * results as n calls of tsu_ising2d_sweep.  Lattices that fit the one-workgroup kernel (TSU_KERNEL_SMALL) run as ONE
 * launch, one workgroup each; larger ones go to a few side streams and run side by side as far as they fit the chip
 * together.  observables_batch: one synchronisation for all. */
int tsu_ising2d_sweep_batch(tsu_ising2d* const* lats, int n_lats, int n_sweeps, const uint64_t* seeds,
                            const uint32_t* sweep0s, const uint32_t* replicas);
int tsu_ising2d_observables_batch(tsu_ising2d* const* lats, int n_lats, int64_t* sum_s, int64_t* sum_bonds);

/* device address of local row r in [-ghost, rows+ghost) and the row pitch in bytes, for halo exchange
 * by the host (RCCL send/recv through torch.distributed on the same stream) */
int tsu_ising2d_row_ptr(tsu_ising2d* lat, int local_row, void** device_ptr, size_t* pitch_bytes);
/* Per-call timing (off by default: two event records per call cost ~2 % at 8 launches per call).  With timing on,
 * tsu_ising2d_last_sweep_ms returns the milliseconds of the most recent tsu_ising2d_sweep (HIP events; synchronises). */
int tsu_ising2d_set_timing(tsu_ising2d* lat, int enable);
int tsu_ising2d_last_sweep_ms(tsu_ising2d* lat, float* ms);
/* sweep-kernel launches issued for this lattice so far (a tile-resident launch runs many generations of sweeps) */
int tsu_ising2d_launch_count(tsu_ising2d* lat, uint64_t* n_launches);

/* ------------------------------------------------------------------ multi-GPU: RCCL below the ABI
 * One process per GPU.  A lattice that does not fit (or should not be swept by) one GPU is cut into row slabs
 * (tsu_ising2d_create_slab); these entry points refresh the ghost rows from the neighbouring ranks with RCCL send/recv over
 * xGMI and sum observables over the ranks -- no PyTorch involved (tsu/distributed.py offers the same through torch.distributed).
 * RCCL is dlopen'ed on first use.  The reference has no counterpart (it is a single Python thread). */
/* 128 opaque bytes, created on ONE rank and handed to the others by the caller (file, environment, MPI, torch.distributed ...) */
int tsu_comm_unique_id(uint8_t id[128]);
/* collective: every rank calls it with the same id; the communicator works on ctx's device and stream */
int tsu_comm_create(tsu_ctx* ctx, int nranks, int rank, const uint8_t id[128], tsu_comm** out);
int tsu_comm_destroy(tsu_comm* comm);
/* ghost rows of slab `rank` <- boundary rows of ranks rank-1 / rank+1 (wrapping for a periodic lattice), one RCCL group on the
 * ctx stream; asynchronous like the sweeps it is ordered with.  The slab must be rows * nranks == total_rows, row0 == rows * rank. */
int tsu_ising2d_halo_exchange(tsu_ising2d* lat, tsu_comm* comm);
/* values[i] <- sum over ranks (n <= 8; observables: sum of spins, sum over bonds); synchronises */
int tsu_comm_allreduce_i64(tsu_comm* comm, int64_t* values, int n);
/* Wait until everything issued on the context's stream (sweeps and halo exchanges) has finished, for at most timeout_s seconds:
 * an exchange whose peer never arrives returns TSU_E_RCCL instead of blocking for ever (the caller leaves with a non-zero exit).
 * *n_exchanges (may be NULL): halo exchanges issued by this communicator so far. */
int tsu_comm_wait(tsu_comm* comm, double timeout_s, uint64_t* n_exchanges);

/* ------------------------------------------------------------------ dense coupling matrix (K2)
 * Replaces GibbsSampler.gibbs_sweep / sample_boltzmann / compute_energy on a dense J
 *   tsu/gibbs.py:79-100 (_compute_local_field incl. the diagonal term), :102-126, :128-162, :215-236.
 * Visiting order is the reference's: range(n), or the caller's permutation (np.random.permutation).
 * State is {0,1} int8 on the device; the Python layer converts to the caller's dtype.
 */
int tsu_dense_create(tsu_ctx* ctx, int n, const void* J_host, int dtype, const double* bias_host /*nullable*/,
                     tsu_dense** out);
int tsu_dense_destroy(tsu_dense* d);
int tsu_dense_set_state(tsu_dense* d, const int8_t* bits_host);
int tsu_dense_get_state(tsu_dense* d, int8_t* bits_host);
/* order: NULL or n_sweeps*n site indices.  replay_uniforms: NULL (Philox doubles keyed by site and
 * sweep0+s) or n_sweeps*n doubles consumed in visiting order (replays np.random.rand, gibbs.py:126). */
int tsu_dense_sweep(tsu_dense* d, double T, int n_sweeps, const int64_t* order, uint64_t seed, uint32_t sweep0,
                    uint32_t replica, const double* replay_uniforms);
/* The whole sampling run of GibbsSampler.sample_boltzmann (tsu/gibbs.py:196-213) from the resident state:
 * n_burnin sweeps, then n_samples x (n_sweeps sweeps, record the state).  samples_host receives n_samples*n bits.
 * order / replay_uniforms cover all (n_burnin + n_samples*n_sweeps) sweeps, row per sweep; sweep numbers for the
 * Philox stream count on from sweep0.  Systems of n <= 192 (fp32 J; 128 for fp64 J) sites in natural order run as ONE launch of a single
 * wave (the reference's published benchmark sizes: n = 1 and n = 10); larger ones loop tsu_dense_sweep on the device
 * side and copy the samples back once. */
int tsu_dense_sample(tsu_dense* d, double T, int n_burnin, int n_sweeps, int n_samples, const int64_t* order,
                     uint64_t seed, uint32_t sweep0, uint32_t replica, const double* replay_uniforms,
                     int8_t* samples_host);
/* The loop of GibbsSampler.simulated_annealing (tsu/gibbs.py:366-391) from the resident state: step s does ONE sweep
 * at temperatures[s] and records the state; states_host receives n_steps*n bits (the caller evaluates the energies
 * and keeps the best, with the reference's own expression).  order / replay_uniforms: one row per step.  n <= 192 (fp32 J; 128 for fp64) in
 * natural order: one launch of a single wave for the whole schedule. */
int tsu_dense_anneal(tsu_dense* d, const double* temperatures, int n_steps, const int64_t* order, uint64_t seed,
                     uint32_t sweep0, uint32_t replica, const double* replay_uniforms, int8_t* states_host);
/* The replica loop of GibbsSampler.parallel_tempering (tsu/gibbs.py:300-306): replica r does n_sweeps sweeps of its own
 * state (states_host[r*n .. ], in and out) at temperatures[r] with its own seed / sweep counter / replica id;
 * replay_uniforms: NULL or n_replicas * n_sweeps * n doubles.  The handle's resident state is not used.  n <= 192 (fp32 J; 128 for fp64): one
 * launch, one wave per replica; up to 576 (448) sites one workgroup per replica; from 2048 sites on (a multiple of 4) up to eight
 * replicas advance together in one launch of the owner-computes kernel on ONE stream of J (groups of eight one after the other).
 * Up to eight replicas' fields stay on the device: a state that comes back byte for byte as one the previous call returned -- in
 * any position, as after a tempering swap -- resumes from that state's fields instead of a pass over J. */
int tsu_dense_sweep_replicas(tsu_dense* d, int n_replicas, const double* temperatures, int n_sweeps, int8_t* states_host,
                             const uint64_t* seeds, const uint32_t* sweep0s, const uint32_t* replicas,
                             const double* replay_uniforms);
/* -1/2 s^T J s - b^T s of the resident state (from the fields the last sweep call kept for it, or -- a state set with
 * tsu_dense_set_state that the last replica call returned -- from that replica's kept fields; otherwise one pass over J) */
int tsu_dense_energy(tsu_dense* d, double* energy);
/* the same for n_states given states (states_host: n_states x n bytes of 0/1; the resident state is not touched): the energies of
 * the states an annealing schedule recorded (simulated_annealing, gibbs.py:384-391, evaluates compute_energy after every step);
 * states the last replica call returned are evaluated from their kept fields */
int tsu_dense_energies(tsu_dense* d, const int8_t* states_host, int n_states, double* energies_host);
/* launches of the one-launch kernels this system has made so far (counts[0]: owner-computes kernel k2_own, counts[1]: pipeline
 * k2_pipe) -- lets a caller or a test see which path its sweeps took (no reference counterpart) */
int tsu_dense_launch_counts(tsu_dense* d, uint64_t counts[2]);

/* ------------------------------------------------------------------ sparse coupling graph, colour-parallel (K5)
 * Replaces GibbsSampler.gibbs_sweep / sample_boltzmann / compute_energy (tsu/gibbs.py:79-236) for models whose
 * coupling matrix is sparse -- IsingChain (tsu/models/ising.py:265-304) and IsingModel on an arbitrary graph
 * (:39-97) at sizes where the dense N x N matrix of the reference cannot exist (a 10^6-site chain would be 8 TB).
 * The graph comes as CSR (row i: columns col_idx[row_ptr[i] .. row_ptr[i+1]) ascending, bit couplings `values`, the
 * diagonal entry J_ii allowed: gibbs.py:97 includes it in the local field) plus a proper colouring: `order` lists the
 * sites colour by colour, color_offsets[c] .. color_offsets[c+1] delimit colour c in it, and no two sites of one colour
 * are coupled.  A sweep visits the colours in order and updates all sites of a colour at once -- the same outcome as
 * the reference's sequential loop run in the visiting order `order` (sites of one colour do not read each other).
 * Decision rule, field (float64) and the Philox uniform keyed by (site, sweep) are those of the dense path (K2).
 */
int tsu_sparse_create(tsu_ctx* ctx, int n, const int64_t* row_ptr /*n+1*/, const int32_t* col_idx, const double* values,
                      const double* bias_host /*nullable, n*/, int n_colors, const int32_t* color_offsets /*n_colors+1*/,
                      const int32_t* order /*n*/, tsu_sparse** out);
int tsu_sparse_destroy(tsu_sparse* g);
int tsu_sparse_set_state(tsu_sparse* g, const int8_t* bits_host); /* n bits {0,1}, site order */
int tsu_sparse_get_state(tsu_sparse* g, int8_t* bits_host);
int tsu_sparse_sweep(tsu_sparse* g, double T, int n_sweeps, uint64_t seed, uint32_t sweep0, uint32_t replica);
/* n_burnin sweeps, then n_samples x (n_sweeps sweeps, record): samples_host receives n_samples*n bits (site order) */
int tsu_sparse_sample(tsu_sparse* g, double T, int n_burnin, int n_sweeps, int n_samples, uint64_t seed, uint32_t sweep0,
                      uint32_t replica, int8_t* samples_host);
/* -1/2 s^T J s - b^T s of the resident state, and sum_i (2 s_i - 1) (the magnetisation numerator in spin language) */
int tsu_sparse_energy(tsu_sparse* g, double* energy, int64_t* sum_spins);

/* ------------------------------------------------------------------ Langevin (K3)
 * Replaces ThermalSamplingUnit._langevin_step (tsu/core.py:64-80) fused with the analytic gradient of a
 * separable quadratic energy E = 1/2 sum_i k_i (x_i - mu_i)^2 (replacing _numerical_gradient, :82-98),
 * for n_chains independent chains (the restarts of sample_from_energy, :140-159), float32 on the device.
 */
int tsu_langevin_create(tsu_ctx* ctx, int n_chains, int dim, tsu_langevin** out);
int tsu_langevin_destroy(tsu_langevin* l);
int tsu_langevin_set_state(tsu_langevin* l, const float* x_host);               /* n_chains*dim */
int tsu_langevin_get_state(tsu_langevin* l, float* x_host);
int tsu_langevin_set_energy(tsu_langevin* l, const float* k_host, const float* mu_host); /* dim each */
/* COUPLED quadratic energy E = 1/2 x^T A x + b^T x: A_host dim*dim row-major and SYMMETRIC (checked), b_host dim or NULL.
 * The gradient A x + b replaces _numerical_gradient (tsu/core.py:82-98) for the reference's multivariate callers
 * (tsu/api.py:94); dim <= 65536.  tsu_langevin_step then makes one launch per step (every new element needs the whole old
 * state); restart / set_state / get_state / trajectories as for the separable energy. */
int tsu_langevin_set_coupling(tsu_langevin* l, const float* A_host, const float* b_host);
/* x <- x_init + amp * N(0,1) per chain (core.py:142-143); chain c uses Philox chain id chain0+c */
int tsu_langevin_restart(tsu_langevin* l, const float* x_init_host /*dim*/, float amp, uint64_t seed,
                         uint32_t chain0);
/* n_steps fused steps, step counters step0.. ; traj_host (nullable): n_steps*n_chains*dim floats */
int tsu_langevin_step(tsu_langevin* l, int n_steps, float dt, float gamma, float T, uint64_t seed, uint32_t step0,
                      uint32_t chain0, float* traj_host);
int tsu_langevin_set_kernel(tsu_langevin* l, int steps_per_launch); /* 0 = auto (fuse in registers) */

#ifdef __cplusplus
}
#endif
#endif /* TSU_HIP_H */
