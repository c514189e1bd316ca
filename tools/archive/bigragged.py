"""Launch-per-k-sweeps runs of big periodic lattices whose width is not a multiple of 16 (byte planes): sweeps per launch and tile
shape (TSU_TILE_VARIANT in the environment; OPEN=1: open boundaries).  usage: python tools/bigragged.py [L ...]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tsu-emulator_amd"))
from tsu import _hip
ctx = _hip.Context(0)
for L in [int(a) for a in sys.argv[1:]] or (7000, 9000, 11000):
    for k in (0, 5, 8):
        lat = _hip.Lattice(L, L, os.environ.get("OPEN", "0") != "1", ctx=ctx)
        lat.set_kernel(_hip.KERNEL_AUTO, k)
        lat.randomize(1); lat.set_thresholds(_hip.ising2d_thresholds(1.0, 0.0, 2.269185))
        n = 240
        lat.sweep(n, 7, 0); ctx.synchronize()
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter(); lat.sweep(n, 7, n * (1 + rep)); ctx.synchronize()
            best = min(best, (time.perf_counter() - t0) / n)
        print(f"variant {os.environ.get('TSU_TILE_VARIANT', 'auto')} L={L} k={k}: {best*1e6:.2f} us/sweep frac {2*L*L/best/8e12:.3f}", flush=True)
        lat.close()
