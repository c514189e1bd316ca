"""ms per sweep of the dense sweep in a caller's visiting order (update_order="random") at one size; TSU_K2_OWN_SB varies the superblock.
usage: own_order_time.py [n] [sweeps]"""
import sys, time
sys.path.insert(0, "tsu-emulator_amd"); sys.path.insert(0, ".")
import os
import numpy as np
from tsu import _hip as hip
ctx = hip.Context.default()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
k = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rng = np.random.default_rng(n)
G = rng.standard_normal((n, n)).astype(np.float32)
J = ((G + G.T) / 2 / np.sqrt(n)).astype(np.float32)
np.fill_diagonal(J, 0.0)
d = hip.DenseSystem(J.astype(np.float64) if os.environ.get('TSU_TOOL_F64') else J, None, hip.DTYPE_F64 if os.environ.get('TSU_TOOL_F64') else hip.DTYPE_F32, ctx=ctx)  # (TSU_TOOL_F64=1: the fp64 kernels)
d.set_state(rng.integers(0, 2, size=n).astype(np.int8))
order = np.array([rng.permutation(n) for _ in range(k)])
d.sweep(1.0, 2, seed=1, sweep0=0, order=order[:2])
ctx.synchronize()
best = 1e9
for rep in range(3):
    t0 = time.perf_counter()
    d.sweep(1.0, k, seed=1, sweep0=2 + rep * k, order=order)
    ctx.synchronize()
    best = min(best, (time.perf_counter() - t0) / k * 1e3)
print(f"n={n} caller's order: {best:.4f} ms/sweep (best of 3 calls of {k} sweeps, incl. the host's check and upload)  checksum {int(d.get_state().sum())}")
