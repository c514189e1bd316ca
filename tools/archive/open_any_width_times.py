"""us per sweep of square lattices whose width is not a multiple of 16, open and periodic: tiled kernel vs generic kernel
(development aid; profiles/r01_any_width_lattices.txt)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tsu-emulator_amd"))
from tsu import _hip
ctx = _hip.Context.default()
for periodic in (False, True):
    for L in (200, 300, 500, 1000, 1500, 2000, 3000, 5000):
        row = [L, "periodic" if periodic else "open    "]
        for kern in (_hip.KERNEL_AUTO, _hip.KERNEL_GENERIC):
            lat = _hip.Lattice(L, L, periodic)
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
        print("L=%5d %s: auto %8.2f us/sweep (%.3e upd/s)   generic %8.2f us/sweep (%.3e upd/s)" % tuple(row), flush=True)
