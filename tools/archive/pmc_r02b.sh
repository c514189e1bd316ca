#!/bin/bash
# round 2, after the strip-exchange / edge-tile changes of K1: the three lattice kernels again (run on the GPU box)
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
B="--steps 3 --warmup 1 --ramp-steps 2 --no-cpu-baseline --no-extra"
python3 tools/pmc_collect.py k1_resident_L4096_s256 k1_resident -- python3 bench.py $B
python3 tools/pmc_collect.py k1_resident_nib_L8192_s256 k1_resident -- python3 bench.py --L 8192 $B
python3 tools/pmc_collect.py k1_tiled_nib_L16384_k8 k1_tiled2 -- python3 bench.py --L 16384 --steps 2 --warmup 1 --ramp-steps 1 --no-cpu-baseline --no-extra
