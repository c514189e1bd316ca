"""k2_own on couplings with long dependency chains (ferromagnets, antiferromagnets: many generations per superblock, the ring of
generation buffers wraps for the followers) against the oracle.  TSU_K2_VERBOSE=2 prints the generation counts."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tsu-emulator_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from tsu import _hip
from oracle import oracle as ora
bad = 0
for n, scale, T in ((8192, 1.0, 0.1), (8192, -1.0, 0.5), (6144, 0.25, 1.0), (12288, 4.0, 2.0), (16384, -2.0, 0.3), (4096, 1.0, 1.0)):
    J = np.full((n, n), scale / n, dtype=np.float32)
    np.fill_diagonal(J, 0.0)
    rng = np.random.default_rng(n)
    J += (rng.standard_normal((n, n)) * 1e-3 / np.sqrt(n)).astype(np.float32)
    st = rng.integers(0, 2, size=n).astype(np.int8)
    d = _hip.DenseSystem(J, None, _hip.DTYPE_F32)
    d.set_state(st)
    d.sweep(T, 3, seed=5, sweep0=0)
    ok = (d.get_state() == ora.dense_sweep_philox(st, J.astype(np.float64), None, T, 3, 5, sweep0=0)).all()
    print("ok  " if ok else "FAIL", n, scale, T, d.launch_counts(), flush=True)
    bad += not ok
    d.close()
# strong nearest-index couplings: site i follows the NEW value of i - 1 -- cascades hundreds of generations long
for n, T in ((8192, 0.5), (16384, 0.7), (6148, 0.3)):
    rng = np.random.default_rng(n + 1)
    J = (rng.standard_normal((n, n)) * 0.2 / np.sqrt(n)).astype(np.float32)
    J = (J + J.T) / 2
    band = (3.0 * rng.choice([-1.0, 1.0], size=n - 1)).astype(np.float32)
    J[np.arange(n - 1), np.arange(1, n)] += band
    J[np.arange(1, n), np.arange(n - 1)] += band
    np.fill_diagonal(J, 0.0)
    st = rng.integers(0, 2, size=n).astype(np.int8)
    d = _hip.DenseSystem(J, None, _hip.DTYPE_F32)
    d.set_state(st)
    d.sweep(T, 2, seed=6, sweep0=1)
    ok = (d.get_state() == ora.dense_sweep_philox(st, J.astype(np.float64), None, T, 2, 6, sweep0=1)).all()
    print("ok  " if ok else "FAIL", "band", n, T, d.launch_counts(), flush=True)
    bad += not ok
    R = 3
    sts = rng.integers(0, 2, size=(R, n)).astype(np.int8)
    out = d.sweep_replicas(sts, [T, 2 * T, 0.5 * T], 1, [1, 2, 3], [0, 0, 0], replicas=[0, 1, 2])
    ok = all((out[r] == ora.dense_sweep_philox(sts[r], J.astype(np.float64), None, [T, 2 * T, 0.5 * T][r], 1, 1 + r, sweep0=0, replica=r)).all() for r in range(R))
    print("ok  " if ok else "FAIL", "band replicas", n, T, d.launch_counts(), flush=True)
    bad += not ok
    d.close()
sys.exit(bad)
