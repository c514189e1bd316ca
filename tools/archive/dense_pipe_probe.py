import sys, time
sys.path.insert(0, "tsu-emulator_amd"); sys.path.insert(0, ".")
import numpy as np
from tsu import _hip as hip
ctx = hip.Context.default()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
T = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rng = np.random.default_rng(n)
G = rng.standard_normal((n, n)).astype(np.float32)
J = ((G + G.T) / 2 / np.sqrt(n)).astype(np.float32)
np.fill_diagonal(J, 0.0)
d = hip.DenseSystem(J, None, hip.DTYPE_F32, ctx=ctx)
d.set_state(rng.integers(0, 2, size=n).astype(np.int8))
d.sweep(T, 4, seed=1, sweep0=0)
ctx.synchronize()
t0 = time.perf_counter()
d.sweep(T, 20, seed=1, sweep0=4)
ctx.synchronize()
print(n, T, (time.perf_counter() - t0) / 20 * 1e3, "ms/sweep")
