import sys, os
sys.path.insert(0, "tsu-emulator_amd"); sys.path.insert(0, ".")
import numpy as np
from tsu import _hip as hip
from oracle import oracle as ora
ctx = hip.Context.default()
for n, T in ((4096, 1.0), (6144, 1.0)):
    rng = np.random.default_rng(n)
    G = rng.standard_normal((n, n)).astype(np.float32)
    J = ((G + G.T) / 2 / np.sqrt(n)).astype(np.float32)
    np.fill_diagonal(J, 0.0)
    s0 = rng.integers(0, 2, size=n).astype(np.int8)
    d = hip.DenseSystem(J, None, hip.DTYPE_F32, ctx=ctx)
    d.set_state(s0)
    want = s0
    done = 0
    for k in (4, 20):
        d.sweep(T, k, seed=1, sweep0=done)
        for s in range(k):
            want = ora.dense_sweep_philox(want, J.astype(np.float64), None, T, 1, 1, sweep0=done + s)
        done += k
        got = d.get_state()
        print(f"n={n} after {done} sweeps: mismatches {int((got != want).sum())}, sums {int(got.sum())} {int(want.sum())}", flush=True)
    d.close()
