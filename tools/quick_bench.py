"""Ad-hoc timing of the lattice kernels through the C-ABI (development aid; bench.py is the contract).
usage: quick_bench.py [L ...]   env TSU_TILE_VARIANT selects the tile shape."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tsu-emulator_amd"))
from tsu import _hip

ctx = _hip.Context.default()
Ls = [int(a) for a in sys.argv[1:]] or [4096, 8192]
for L in Ls:
    lat = _hip.Lattice(L, L, os.environ.get("OPEN", "0") != "1")
    lat.randomize(42)
    lat.set_model(1.0, 0.0, 2.269185)
    for spl, n in [(int(x), 60) for x in os.environ.get("KS", "3,4,5,6").split(",")]:
        lat.set_kernel(_hip.KERNEL_TILED, spl)
        lat.sweep(n, 1, sweep0=0)
        ctx.synchronize()
        best = 1e9
        for rep in range(3):
            ctx.timer_begin()
            lat.sweep(n, 1, sweep0=100 * rep)
            best = min(best, ctx.timer_end())
        ups = L * L * n / (best * 1e-3)
        print(f"variant={os.environ.get('TSU_TILE_VARIANT','-')} L={L} k={spl}: {best / n * 1e3:8.1f} us/sweep  {ups:.3e} upd/s = {2 * ups / 8e12 * 100:.1f}% of 8 TB/s")
    lat.close()
