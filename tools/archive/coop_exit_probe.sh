#!/bin/bash
# runs the probe in the four combinations that matter; never stops at a failing one (exit codes are the data)
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
L=gpurun_out/coop_probe.log
: > $L
run() { echo "=== $*" >> $L; "$@" >> $L 2>&1; echo "=== exit code $?" >> $L; }
TSU_COOP_LAUNCH=1 run python3 tools/coop_exit_probe.py plain_coop noclose
export TSU_COOP_LAUNCH=0
run rocprofv3 --kernel-trace --stats -d gpurun_out/coop_probe_a -o a -- python3 tools/coop_exit_probe.py prof_ordinary noclose
export TSU_COOP_LAUNCH=1
run rocprofv3 --kernel-trace --stats -d gpurun_out/coop_probe_b -o b -- python3 tools/coop_exit_probe.py prof_coop noclose
run rocprofv3 --kernel-trace --stats -d gpurun_out/coop_probe_c -o c -- python3 tools/coop_exit_probe.py prof_coop_close close
tail -5 $L
exit 0
