// dense.h -- state of one dense coupling system (K2) and the device helpers its kernels share.
#pragma once
#include "tsu_common.h"

#define DB 64  // block of visiting-order positions resolved by one wave

struct tsu_dense {
    tsu_ctx* ctx;
    int n, dtype;
    void* J;    // n x n row-major, f64 or f32
    void* JT;   // transpose (aliases J when J is symmetric)
    double* bias;
    int8_t* state;   // current state ({0,1})
    int8_t* state2;  // next state: a sweep reads `state` (frozen) and writes `state2`, then the two are swapped
    double* field;
    int64_t* order;  // device copy of the visiting order (n_sweeps * n) or NULL
    double* uniforms;
    size_t order_cap, uni_cap;
    double* d_energy;
    // superblock fixed-point path
    int8_t* delta[2];   // ping-pong flip vectors of the current superblock
    double* logit;      // T * logit(u) per site of the current superblock... stored as logit(u)
    int* sb_sync;       // [0 .. SB_MAX_IT): changes per iteration, [SB_MAX_IT]: converged flag, per superblock
    int8_t* backup;     // state at the start of the call (re-run on the exact path if a superblock did not converge)
    int sb_cap;         // superblocks allocated in sb_sync
    int sb_budget;      // iteration launches per superblock: slowest fixed point of the last call + 8 (16 .. SB_MAX_IT)
    // cooperative single-launch path (dense_coop.hip)
    double* co_logit;   // logit(u) per site for the current sweep
    double* co_corr;    // intra-superblock correction per site
    int8_t* co_d0;      // flips decided from the field alone
    int8_t* co_d1;      // current flips of the fixed-point iteration
    int* co_lists;      // two change lists of SB_SIZE entries
    int* co_counts;     // per (sweep, superblock): changes per iteration
    unsigned* co_bar;   // grid barrier counter, error flag, slowest fixed point, not-converged flag
    size_t co_counts_cap;
    int8_t* samples;    // device buffer of recorded states (tsu_dense_sample)
    size_t samples_cap;
    double* temps;      // device copy of an annealing schedule (tsu_dense_anneal)
    size_t temps_cap;
    void* rep_buf;      // tsu_dense_sweep_replicas: states, replayed uniforms and per-replica parameters
    size_t rep_cap;
    // k2_own hands the REPLICAS' fields from call to call as well (a tempering loop is a loop of short calls on states that come back
    // unchanged or swapped among themselves: without this every call pays a full pass over J for all replicas):
    double* rep_fields[2];  // [8][n] each: the fields of the states the last replica call returned (in rep_cur), and where the next writes
    int8_t* rep_prev;       // host copy of those states [8][n]: an incoming state is matched against them byte for byte
    int rep_prev_n;         // how many there are (0: none)
    int rep_cur;
    int rep_since;          // sweeps since the replicas' fields were last computed from scratch
    unsigned* h_flags;      // pinned host words: the owner kernel's error flags land here without a staged copy
    int8_t* h_rep;          // pinned host buffer of 8 n bytes: a group of replica states travels in ONE copy each way
    int8_t* h_stage;        // pinned host buffer (n bytes + 8): get_state / energy come back through it (a copy into the caller's
                            // pageable memory is staged by the runtime and costs ~10 us more)
    int rep_match;          // the resident state (tsu_dense_set_state) IS row rep_match - 1 of rep_prev: tsu_dense_energy takes its kept fields (0: no)
    int co_disabled;    // cooperative launch unavailable or failed once: use the multi-launch path
    unsigned long long* pp_masks;  // k2_pipe: flip-mask granules of the solver teams
    int pp_failed;      // k2_pipe ran and left the state half updated: the caller restores it, later calls skip the pipeline
    // k2_pipe hands the fields f = J s + b from call to call (a loop of one-sweep calls -- annealing, tempering -- would otherwise
    // stream J once more per call to rebuild them, and tsu_dense_energy once more again):
    double* co_fields;  // [n] fields of the state as the last pipeline call left it
    int fields_valid;   // co_fields belongs to d->state (cleared by everything else that writes the state)
    int since_refresh;  // sweeps since the fields were last computed from scratch (CO_REFRESH bounds the drift across calls too)
    int pipe_streak;    // consecutive pipeline calls on this state: the second one starts to keep the fields
    // owner-computes kernel (dense_own.hip): value-mask granules of the running generation and of the superblocks' final values
    unsigned long long* own_gran;
    size_t own_cap;     // 8-byte words allocated in own_gran
    int own_failed;     // k2_own ran and gave up half way: the caller restores the state, later calls skip it
    uint64_t n_own, n_pipe;  // successful launches of k2_own / k2_pipe (tsu_dense_launch_counts)
};

// per-replica parameters of a k2_own launch
struct OwnRep {
    double T;
    uint32_t sweep0, tag, k0, k1;
    int src;  // (several replicas, fields kept) which of the previous call's replicas this state is: row of its fields
};


#define SB_SIZE 4096  // positions per superblock of the fixed-point paths (4096 beats 2048 and 8192: profiles/r01_k2_notes.txt)

static __device__ __forceinline__ double dense_uniform(uint32_t i, uint32_t t, uint32_t tag, uint32_t k0, uint32_t k1) {
    u32x4 w = tsu_philox(i >> 1, 0u, t, tag, k0, k1);
    uint32_t a = (i & 1) ? w.z : w.x, b = (i & 1) ? w.w : w.y;
    a >>= 5;
    b >>= 6;
    return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
}

static __device__ __forceinline__ double sigmoid_clamped(double x) {
    if (x > 20.0) return 1.0;
    if (x < -20.0) return 0.0;
    return 1.0 / (1.0 + exp(-x));
}


// u < sigmoid(x) <=> x > logit(u): the logit is computed once per site and sweep, so a decision is one compare
// instead of a float64 exp.  The reference's own expression (gibbs.py:73-77,126) decides whenever x is within a
// safety margin of the logit and at the +-20 clamp, so outcomes are unchanged.
static __device__ __forceinline__ int dense_decide(double F, double lg, double T, double invT, uint32_t site,
                                                   const double* __restrict__ uniforms, uint32_t sweep, uint32_t tag,
                                                   uint32_t k0, uint32_t k1) {
    const double xa = F * invT;
    if (fabs(fabs(xa) - 20.0) < 1e-9 || fabs(xa - lg) <= 1e-9 * (1.0 + fabs(lg))) {
        const double u = uniforms ? uniforms[site] : dense_uniform(site, sweep, tag, k0, k1);
        return (u < sigmoid_clamped(F / T)) ? 1 : 0;
    }
    if (xa > 20.0) return 1;
    if (xa < -20.0) return 0;
    return xa > lg ? 1 : 0;
}

// cooperative single-launch sweep (dense_coop.hip): TSU_OK with *done = 1 when the call was carried out, *done = 0
// when the path is unavailable or a superblock did not converge (state untouched or restored by the caller)
int tsu_dense_pipe_run(tsu_dense* d, double T, const double* temps_dev, int n_total, int rec_from, int rec_every, int8_t* samples_dev,
                       uint64_t seed, uint32_t sweep0, uint32_t replica, bool have_uni, int* done, const int64_t* order_dev = nullptr);
// owner-computes kernel (dense_own.hip): n_sweeps sweeps of R states ([R][n] at states_dev; R == 1: d->state) in one launch, natural
// order or the caller's (order_dev: [n_sweeps][n]); *done as above
int tsu_dense_own_run(tsu_dense* d, int R, const OwnRep* reps, int8_t* states_dev, int n_sweeps, const double* uniforms_dev,
                      const int64_t* order_dev, const double* temps_dev, int8_t* samples_dev, int rec_from, int rec_every, bool fields_were_valid,
                      bool allow_persist, int* done);
int tsu_dense_coop_sweep(tsu_dense* d, double T, int n_sweeps, uint64_t seed, uint32_t sweep0, uint32_t replica,
                         bool have_uni, int* done, const int64_t* order_dev = nullptr);
