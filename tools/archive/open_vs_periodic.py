"""Open against periodic boundaries on the tile-resident kernel (the open forms carry the degree-3 / degree-2 threshold patches
and the edge masks: 126 VGPRs, no spills since the strip exchange moves one colour plane only)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tsu-emulator_amd"))
from tsu import _hip
ctx = _hip.Context.default()
for rows, cols in ((4096, 4096), (4096, 8192), (2048, 2048)):
    for periodic in (True, False):
        lat = _hip.Lattice(rows, cols, periodic)
        lat.randomize(42)
        lat.set_model(1.0, 0.0, 2.269185)
        n = 480
        lat.sweep(n, 1, 0)
        ctx.synchronize()
        best = 1e9
        for rep in range(3):
            ctx.timer_begin()
            lat.sweep(n, 1, n * (rep + 1))
            best = min(best, ctx.timer_end())
        print(f"{rows}x{cols} {'periodic' if periodic else 'open    '}: {best / n * 1e3:7.2f} us/sweep  {rows * cols * n / (best * 1e-3):.3e} upd/s", flush=True)
        lat.close()
