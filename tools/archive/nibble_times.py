"""Large lattices, automatic kernel choice; run with TSU_K1_NIBBLE=0 / 1 to compare byte and nibble colour planes."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tsu-emulator_amd"))
from tsu import _hip
ctx = _hip.Context.default()
shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]] or [(8192, 8192), (16384, 16384), (8192, 16384), (4096, 8192)]
for rows, cols in shapes:
    lat = _hip.Lattice(rows, cols, True)
    lat.randomize(42)
    lat.set_model(1.0, 0.0, 2.269185)
    lat.set_kernel(_hip.KERNEL_AUTO, int(os.environ.get("K", "0")))
    n = int(os.environ.get("SWEEPS", "240"))
    lat.sweep(n, 1, 0)
    ctx.synchronize()
    best = 1e9
    l0 = lat.launch_count()
    for rep in range(3):
        ctx.timer_begin()
        lat.sweep(n, 1, n * (rep + 1))
        best = min(best, ctx.timer_end())
    ups = rows * cols * n / (best * 1e-3)
    print(f"nibble={os.environ.get('TSU_K1_NIBBLE','1')} variant={os.environ.get('TSU_TILE_VARIANT','auto')} {rows}x{cols}: {best / n * 1e3:8.2f} us/sweep  {ups:.3e} upd/s = {2 * ups / 8e12 * 100:.1f}% of 8 TB/s  ({(lat.launch_count() - l0) // 3} launches per {n} sweeps)  obs {lat.observables()}", flush=True)
    lat.close()
