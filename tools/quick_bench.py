"""Ad-hoc timing of the lattice kernels through the C-ABI (development aid; bench.py is the contract)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tsu-emulator_amd"))
from tsu import _hip

ctx = _hip.Context.default()
print(ctx.device_info())
for L in (4096, 8192, 16384):
    lat = _hip.Lattice(L, L, True)
    lat.randomize(42)
    lat.set_model(1.0, 0.0, 2.269185)
    for kern, spl, n in ((_hip.KERNEL_GENERIC, 0, 8), (_hip.KERNEL_TILED, 1, 16), (_hip.KERNEL_TILED, 2, 32), (_hip.KERNEL_TILED, 4, 64),
                         (_hip.KERNEL_TILED, 8, 64)):
        lat.set_kernel(kern, spl)
        lat.sweep(n, 1, sweep0=0)
        ctx.synchronize()
        best = 1e9
        for rep in range(3):
            ctx.timer_begin()
            lat.sweep(n, 1, sweep0=100 * rep)
            best = min(best, ctx.timer_end())
        ups = L * L * n / (best * 1e-3)
        print(f"L={L} kernel={kern} spl={spl} n={n}: {best / n * 1e3:8.1f} us/sweep  {ups:.3e} upd/s  alg {2 * ups / 1e12:.2f} TB/s = {2 * ups / 8e12 * 100:.1f}% of 8 TB/s")
    print(lat.observables())
    lat.close()
