"""Tile-variant choice for lattices whose 128-row tiles fill the chip about once (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tsu-emulator_amd"))
from tsu import _hip
ctx = _hip.Context.default()
for rows, cols in [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]:
    for k in (5, 8):
        lat = _hip.Lattice(rows, cols, True)
        lat.randomize(1); lat.set_model(1.0, 0.0, 2.269185); lat.set_kernel(_hip.KERNEL_TILED, k)
        lat.sweep(400, 1, 0); ctx.synchronize()
        n = 400
        t0 = time.perf_counter(); lat.sweep(n, 1, 400); ctx.synchronize(); dt = time.perf_counter() - t0
        print(f"variant={os.environ.get('TSU_TILE_VARIANT', 'auto')} {rows}x{cols} k={k}: {rows * cols * n / dt:.3e} upd/s", flush=True)
        lat.close()
