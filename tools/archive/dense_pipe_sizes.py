import sys, time, os
sys.path.insert(0, "tsu-emulator_amd"); sys.path.insert(0, ".")
import numpy as np
from tsu import _hip as hip
ctx = hip.Context.default()
for n, T in ((1024, 1.0), (2048, 1.0), (4096, 1.0), (4096, 0.05), (6144, 1.0), (8192, 1.0), (12288, 1.0), (16384, 1.0), (16384, 0.2)):
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
    dt = (time.perf_counter() - t0) / 20
    print(f"pipe={os.environ.get('TSU_K2_PIPE','1')} n={n} T={T}: {dt*1e3:.3f} ms/sweep  {n*n*4/dt/1e9:.0f} GB/s of J  state sum {int(d.get_state().sum())}", flush=True)
    d.close()
