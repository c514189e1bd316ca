#!/bin/bash
# round-3 counter collection (run on the GPU box): the dense owner-computes kernel, and a calibration of FETCH_SIZE on its access
# pattern (256-byte row segments, 16 B per lane) with the axpy microbenchmark, whose bytes are known
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
python3 tools/pmc_collect.py k2_own_N16384_f32 k2_own -- python3 tools/dense_prof.py 16384 8
PMC_GROUPS=fetch python3 tools/pmc_collect.py axpy_calibration_W64_nl1700 axpy_quad -- tools/microbench_axpy 16384 1700 64 20 calib
# the coupled Langevin kernel (one chain at d = 16384: the matrix streamed once per step) and the paired chain classes of K5
PMC_GROUPS=fetch,write,grbm python3 tools/pmc_collect.py k3_coupled_d16384_one_chain k3_coupled -- python3 tools/langevin_coupled_time.py 16384 1 10
python3 tools/pmc_collect.py k5_stencil4_first_class_2p23 "k5_stencil4<1>" -- python3 tools/sparse_time.py 24 20
PMC_GROUPS=fetch,write,grbm python3 tools/pmc_collect.py k5_stencil4_second_class_2p23 "k5_stencil4<2>" -- python3 tools/sparse_time.py 24 20
