"""ms per sweep of the dense sweep at one size (k2_own / k2_pipe as the switches say); TSU_K2_VERBOSE=2 prints the kernel's timeline.
usage: own_time.py [n] [T] [sweeps]"""
import sys, time
sys.path.insert(0, "tsu-emulator_amd"); sys.path.insert(0, ".")
import os
import numpy as np
from tsu import _hip as hip
ctx = hip.Context.default()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
T = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
k = int(sys.argv[3]) if len(sys.argv) > 3 else 20
rng = np.random.default_rng(n)
G = rng.standard_normal((n, n)).astype(np.float32)
J = ((G + G.T) / 2 / np.sqrt(n)).astype(np.float32)
np.fill_diagonal(J, 0.0)
d = hip.DenseSystem(J.astype(np.float64) if os.environ.get('TSU_TOOL_F64') else J, None, hip.DTYPE_F64 if os.environ.get('TSU_TOOL_F64') else hip.DTYPE_F32, ctx=ctx)  # (TSU_TOOL_F64=1: the fp64 kernels)
d.set_state(rng.integers(0, 2, size=n).astype(np.int8))
d.sweep(T, 4, seed=1, sweep0=0)
ctx.synchronize()
best = 1e9
for rep in range(3):
    t0 = time.perf_counter()
    d.sweep(T, k, seed=1, sweep0=4 + rep * k)
    ctx.synchronize()
    best = min(best, (time.perf_counter() - t0) / k * 1e3)
print(f"n={n} T={T}: {best:.4f} ms/sweep (best of 3 calls of {k} sweeps)  checksum {int(d.get_state().sum())}")
