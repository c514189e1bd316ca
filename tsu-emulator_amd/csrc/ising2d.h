// ising2d.h -- lattice handle shared by ising2d.hip (generic kernel, K4, host API) and ising2d_tiled.hip.
#pragma once
#include "tsu_common.h"

struct tsu_ising2d {
    tsu_ctx* ctx;
    int64_t total_rows;  // global lattice height
    int64_t row0;        // global index of owned row 0
    int rows, cols;      // owned rows, columns
    int periodic;
    int ghost;           // ghost rows on each side (0 for a whole lattice)
    int wrap_rows;       // whole periodic lattice on one GPU: vertical neighbours wrap inside the buffer
    size_t pitch;
    int8_t* alloc[2];    // ping-pong buffers (second one allocated on first tiled launch)
    int cur;
    uint64_t table[25];
    int have_table;
    int kernel, sweeps_per_launch;
    int64_t* d_obs;      // 2 x int64 accumulators
    hipEvent_t ev0, ev1;
    int timed, timing;
    unsigned long long launches;  // sweep-kernel launches so far
    int* d_sync;         // tile-resident kernel: per-tile generation counters
    uint64_t* d_xbuf;    // tile-resident kernel: exchange strips
    size_t xbuf_cap;
    uint32_t xgen;       // generations numbered so far in d_xbuf (every strip element carries its generation number)
    uint64_t xsig;       // strip layout the numbering belongs to (0: buffer not cleared yet)
    void* d_batch;       // tsu_ising2d_sweep_batch: device copy of the per-lattice launch items
    size_t batch_cap;
    void* d_obs_batch;   // tsu_ising2d_observables_batch: [n][2] sums of the batch (lives with its first lattice)
    size_t obs_batch_cap;
    size_t sync_cap;     // ints allocated in d_sync
    int* h_err;          // host-mapped flag the tile-resident kernel sets if a bounded wait expires
};


// ising2d_tiled.hip
int tsu_ising2d_tiled_supported(const tsu_ising2d* L);
int tsu_ising2d_tiled_tiles(const tsu_ising2d* L);
int tsu_ising2d_tiled_sweep(tsu_ising2d* L, int n_sweeps, uint64_t seed, uint32_t sweep0, uint32_t replica, int part);
int tsu_ising2d_tiled_part_supported(const tsu_ising2d* L);
int tsu_ising2d_planes_supported(const tsu_ising2d* L);
int tsu_ising2d_planes_sweep(tsu_ising2d* const* lats, int n, int n_sweeps, const uint64_t* seeds, const uint32_t* sweep0s,
                             const uint32_t* replicas);
