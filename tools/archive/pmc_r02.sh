#!/bin/bash
# round-2 counter collection (run on the GPU box): one pmc_collect per kernel of interest
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
B="--steps 3 --warmup 1 --ramp-steps 2 --no-cpu-baseline --no-extra"
python3 tools/pmc_collect.py k1_resident_L4096_s256 k1_resident -- python3 bench.py $B
python3 tools/pmc_collect.py k1_resident_nib_L8192_s256 k1_resident -- python3 bench.py --L 8192 $B
python3 tools/pmc_collect.py k1_tiled_nib_L16384_k8 k1_tiled2 -- python3 bench.py --L 16384 --steps 2 --warmup 1 --ramp-steps 1 --no-cpu-baseline --no-extra
python3 tools/pmc_collect.py k2_pipe_N16384_f32 k2_pipe -- python3 tools/dense_prof.py 16384 8
python3 tools/pmc_collect.py k3_fused_d2p20_500 k3_langevin -- python3 tools/langevin_prof2.py fused
python3 tools/pmc_collect.py k3_unfused_256x2p20 k3_langevin -- python3 tools/langevin_prof2.py hbm
