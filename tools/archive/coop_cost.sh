#!/bin/bash
# cost of hipLaunchCooperativeKernel against an ordinary launch for the grid-synchronising kernels (r02)
cd "$(dirname "$0")/.."
for c in 0 1; do
  export TSU_COOP_LAUNCH=$c
  echo "=== TSU_COOP_LAUNCH=$c"
  python3 bench.py --no-cpu-baseline --no-extra | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench 4096^2: value %.4e  ms/step %.4f  launch us %.1f' % (d['value'], d['ms_per_step'], d['config']['avg_launch_us']))"
  python3 tools/scan_times.py
  python3 tools/dense_sweep_times.py 2>&1 | tail -4
done
