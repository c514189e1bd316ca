import os, sys, time
sys.path.insert(0, "/root/repo/tsu-emulator_amd")
from tsu import _hip
ctx = _hip.Context(0)
for L in [int(a) for a in sys.argv[1:]] or (5000, 6000, 7000, 3000, 10000, 12000):
    lat = _hip.Lattice(L, L, os.environ.get("OPEN", "0") != "1", ctx=ctx)
    lat.randomize(1); lat.set_thresholds(_hip.ising2d_thresholds(1.0, 0.0, 2.269185))
    n = 240
    for _ in range(3): lat.sweep(n, 7, 0)
    ctx.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for i in range(4): lat.sweep(n, 7, n * (1 + i + 4 * rep))
        ctx.synchronize()
        best = min(best, (time.perf_counter() - t0) / (4 * n))
    print(f"L={L} {best*1e6:.2f} us/sweep {L*L/best:.3e} upd/s frac {2*L*L/best/8e12:.3f}", flush=True)
    lat.close()
