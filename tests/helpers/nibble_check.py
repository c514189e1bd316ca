"""Child process of tests/test_nibble_gpu.py: TSU_TILE_VARIANT is read once per process, so every forced tile shape gets
its own process.  Sweeps lattices on the tiled kernel and compares spins and observables with the oracle, bit for bit."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tsu-emulator_amd"))
import numpy as np  # noqa: E402
from oracle import oracle as ora  # noqa: E402
from tsu import _hip  # noqa: E402

cases = json.loads(sys.argv[1])
ctx = _hip.Context(0)
for case in cases:
    rows, cols, sweeps, T, k = case[:5]
    periodic = bool(case[5]) if len(case) > 5 else True
    table = ora.ising2d_thresholds(1.0, 0.05, T, 0)
    lat = _hip.Lattice(rows, cols, periodic, ctx=ctx)
    lat.set_kernel(_hip.KERNEL_TILED, k)
    lat.randomize(rows + cols)
    s0 = lat.get_spins()
    lat.set_thresholds(table)
    want = s0
    done = 0
    for n in sweeps:
        lat.sweep(n, 31, done)
        want = ora.ising2d_sweep(want, periodic, table, n, 31, done)
        done += n
        got = lat.get_spins()
        if not (got == want).all():
            bad = np.argwhere(got != want)
            print(f"MISMATCH {rows}x{cols} after {done} sweeps (k={k}): {len(bad)} sites, first {bad[:5].tolist()}")
            sys.exit(1)
        assert lat.observables() == ora.ising2d_observables(want, periodic)
    print(f"ok {rows}x{cols} sweeps={sweeps} k={k} launches={lat.launch_count()}", flush=True)
    lat.close()
print("ALL OK")
