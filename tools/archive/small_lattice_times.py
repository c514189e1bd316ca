"""Small lattices (the reference's own sizes): sweeps/s of the one-workgroup LDS kernel against the generic kernel."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tsu-emulator_amd"))
from tsu import _hip
ctx = _hip.Context.default()
for rows, cols, periodic in ((32, 32, False), (32, 32, True), (50, 50, False), (128, 128, True), (256, 256, True)):
    for name, kern in (("generic", _hip.KERNEL_GENERIC), ("small", _hip.KERNEL_SMALL)):
        lat = _hip.Lattice(rows, cols, periodic)
        try:
            lat.set_kernel(kern)
        except _hip.UnsupportedError:
            lat.close()
            continue
        lat.randomize(1); lat.set_model(1.0, 0.0, 2.5)
        lat.sweep(100, 1, 0); ctx.synchronize()
        n = 2000
        t0 = time.perf_counter(); lat.sweep(n, 1, 100); ctx.synchronize(); dt = time.perf_counter() - t0
        print(f"{rows}x{cols} periodic={periodic} {name:8s}: {dt / n * 1e6:7.2f} us/sweep  {rows * cols * n / dt:.3e} upd/s", flush=True)
        lat.close()
