"""pipeline kernel against the oracle (and against the barrier kernel in a second process): sizes around the superblock size"""
import sys, os
sys.path.insert(0, "tsu-emulator_amd"); sys.path.insert(0, ".")
import numpy as np
from tsu import _hip as hip
from oracle import oracle as ora
ctx = hip.Context.default()
bad = 0
for n, T, dt in ((4096, 1.0, "f32"), (6144, 1.0, "f32"), (4096, 1.0, "f64"), (5000, 0.7, "f32"), (8192, 1.0, "f32"), (8196, 1.0, "f32"), (9000, 1.3, "f64"), (12288, 1.0, "f32")):
    rng = np.random.default_rng(n)
    G = rng.standard_normal((n, n)).astype(np.float32)
    J = ((G + G.T) / 2 / np.sqrt(n)).astype(np.float32)
    np.fill_diagonal(J, 0.0)
    b = rng.normal(size=n) * 0.1
    s0 = rng.integers(0, 2, size=n).astype(np.int8)
    d = hip.DenseSystem(J if dt == "f32" else J.astype(np.float64), b, hip.DTYPE_F32 if dt == "f32" else hip.DTYPE_F64, ctx=ctx)
    d.set_state(s0)
    d.sweep(T, 3, seed=7, sweep0=2)
    got = d.get_state()
    want = ora.dense_sweep_philox(s0, J.astype(np.float64), b, T, 3, 7, sweep0=2)
    nd = int((got != want).sum())
    print(f"n={n} T={T} {dt}: {'OK' if nd == 0 else 'MISMATCH at %d sites, first %s' % (nd, np.flatnonzero(got != want)[:8])}", flush=True)
    bad += nd != 0
    d.close()
sys.exit(1 if bad else 0)
