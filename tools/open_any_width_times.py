"""us per sweep of open (non-periodic) square lattices of arbitrary width: tiled kernel vs generic kernel (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tsu-emulator_amd"))
from tsu import _hip
ctx = _hip.Context.default()
for L in (300, 500, 1000, 1500, 2000, 3000, 5000):
    row = [L]
    for kern in (_hip.KERNEL_AUTO, _hip.KERNEL_GENERIC):
        lat = _hip.Lattice(L, L, False)
        lat.set_kernel(kern)
        lat.randomize(1)
        lat.set_model(1.0, 0.0, 2.269185)
        lat.sweep(64, 1, 0)
        ctx.synchronize()
        n = 512
        t = time.perf_counter()
        lat.sweep(n, 1, 64)
        ctx.synchronize()
        dt = time.perf_counter() - t
        row += [dt / n * 1e6, L * L * n / dt]
        lat.close()
    print("L=%5d open: auto %8.2f us/sweep (%.3e upd/s)   generic %8.2f us/sweep (%.3e upd/s)" % tuple(row), flush=True)
