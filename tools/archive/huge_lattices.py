"""Lattices of 10^9 sites and more: tiled kernel == generic kernel by observables (development aid)."""
import sys, time, zlib
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tsu-emulator_amd"))
from tsu import _hip
ctx = _hip.Context.default()
for rows, cols, periodic in ((32768, 32768, True), (30000, 50000, False), (20000, 20010, True)):
    res = []
    for kern in (_hip.KERNEL_AUTO, _hip.KERNEL_GENERIC):
        lat = _hip.Lattice(rows, cols, periodic)
        lat.set_kernel(kern)
        lat.randomize(3)
        lat.set_model(1.0, 0.0, 2.269185)
        t = time.perf_counter()
        lat.sweep(10, 3, 0)
        obs = lat.observables()
        dt = time.perf_counter() - t
        res.append((obs, dt))
        lat.close()
    print(rows, cols, "periodic" if periodic else "open", "EQUAL" if res[0][0] == res[1][0] else "DIFFER", res[0][0],
          "auto %.3e upd/s, generic %.3e upd/s" % (rows * cols * 10 / res[0][1], rows * cols * 10 / res[1][1]), flush=True)
