"""Dense (K2) sweep time over mid sizes: which kernel serves which N (development aid; TSU_K2_VERBOSE=1 names the path)."""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tsu-emulator_amd"))
import numpy as np
from tsu import _hip as hip
ctx = hip.Context.default()
for n in [int(v) for v in os.environ.get("SIZES", "65,96,128,256,512,1024,2048,4096,8192").split(",")]:
    rng = np.random.default_rng(n)
    G = rng.standard_normal((n, n)).astype(np.float32)
    J = ((G + G.T) / 2 / np.sqrt(n)).astype(np.float32)
    np.fill_diagonal(J, 0.0)
    for dt_name, dtc in (("f32", hip.DTYPE_F32), ("f64", hip.DTYPE_F64)):
        d = hip.DenseSystem(J if dtc == hip.DTYPE_F32 else J.astype(np.float64), None, dtc, ctx=ctx)
        d.set_state(rng.integers(0, 2, size=n).astype(np.int8))
        d.sweep(1.0, 4, seed=1, sweep0=0)
        ctx.synchronize()
        k = 50 if n <= 2048 else 20
        t0 = time.perf_counter()
        d.sweep(1.0, k, seed=1, sweep0=4)
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / k
        print(f"N={n:5d} {dt_name}: {dt * 1e6:9.1f} us/sweep  {dt / n * 1e9:7.1f} ns/update  {n / dt:.3g} upd/s", flush=True)
        d.close()
