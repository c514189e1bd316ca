"""Spin-updates/s of whole lattices on the tile-resident kernel, periodic and open (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tsu-emulator_amd"))
from tsu import _hip
ctx = _hip.Context.default()
for rows, cols in ((512, 512), (1024, 1024), (2048, 2048), (4096, 4096), (4096, 8192)):
    out = []
    for periodic in (True, False):
        lat = _hip.Lattice(rows, cols, periodic)
        lat.randomize(1)
        lat.set_model(1.0, 0.0, 2.269185)
        lat.sweep(256, 1, 0)
        ctx.synchronize()
        best = 1e9
        for rep in range(3):
            t = time.perf_counter()
            lat.sweep(1024, 1, 256 + 1024 * rep)
            ctx.synchronize()
            best = min(best, time.perf_counter() - t)
        out.append("%s %.3e upd/s (%.2f us/sweep)" % ("periodic" if periodic else "open", rows * cols * 1024 / best, best / 1024 * 1e6))
        lat.close()
    print("%5d x %5d: " % (rows, cols) + "   ".join(out), flush=True)
