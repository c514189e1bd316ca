"""Dense (K2) sweep timings over sizes and temperatures; TSU_K2_VERBOSE=1 prints the fixed-point iteration counts."""
import sys, time
sys.path.insert(0, "tsu-emulator_amd"); sys.path.insert(0, ".")
import numpy as np
from tsu import _hip as hip

ctx = hip.Context.default()
out = {}
for n, T in ((1024, 1.0), (4096, 1.0), (4096, 0.05), (8192, 1.0), (16384, 1.0), (16384, 0.2)):
    rng = np.random.default_rng(n)
    G = rng.standard_normal((n, n)).astype(np.float32)
    J = ((G + G.T) / 2 / np.sqrt(n)).astype(np.float32)
    np.fill_diagonal(J, 0.0)
    d = hip.DenseSystem(J, None, hip.DTYPE_F32, ctx=ctx)
    d.set_state(rng.integers(0, 2, size=n).astype(np.int8))
    d.sweep(T, 4, seed=1, sweep0=0)
    ctx.synchronize()
    t0 = time.perf_counter()
    k = 20
    d.sweep(T, k, seed=1, sweep0=4)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / k
    out[(n, T)] = (round(dt * 1e3, 3), f"{n / dt:.3g} upd/s", f"{n * n * 4 / dt / 1e9:.0f} GB/s J")
    d.close()
    del J, G
for k_, v in out.items():
    print(k_, v)
