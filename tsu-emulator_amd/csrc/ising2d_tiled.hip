// ising2d_tiled.hip -- K1 fast path: LDS-staged 2-D halo tiles, several sweeps per launch (gfx950).
#include "tsu_common.h"

struct tsu_ising2d;

int tsu_ising2d_tiled_supported(const tsu_ising2d*) { return 0; }

int tsu_ising2d_tiled_sweep(tsu_ising2d*, int, uint64_t, uint32_t, uint32_t) { return TSU_E_UNSUPPORTED; }
