"""K1: us per sweep of an L x L lattice at T_c for several generation lengths k (sweeps between two strip exchanges / stagings).
usage: lattice_kscan.py [L] [sweeps] [k ...]"""
import sys
sys.path.insert(0, "tsu-emulator_amd"); sys.path.insert(0, ".")
from tsu import _hip as hip
ctx = hip.Context.default()
L = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sweeps = int(sys.argv[2]) if len(sys.argv) > 2 else 256
ks = [int(x) for x in sys.argv[3:]] or [2, 4, 6, 8, 12, 16]
T_C = 2.0 / 0.881373587019543
for k in ks:
    lat = hip.Lattice(L, L, True, ctx=ctx)
    lat.randomize(42)
    lat.set_model(1.0, 0.0, T_C, hip.MODE_PHYSICAL)
    try:
        lat.set_kernel(hip.KERNEL_AUTO, k)
        for _ in range(8):
            lat.sweep(sweeps, 42, 0)
        ctx.synchronize()
        best = 1e30
        for r in range(5):
            ctx.timer_begin()
            lat.sweep(sweeps, 42, sweeps * (r + 1))
            best = min(best, ctx.timer_end())
        s, b = lat.observables()
        print(f"L={L} k={k}: {best * 1e3 / sweeps:.3f} us per sweep = {L * L * sweeps / (best * 1e-3):.4e} updates/s, frac {2 * L * L * sweeps / (best * 1e-3) / 8e12:.3f}  M={s / (L * L):+.4f}")
    except Exception as e:
        print(f"L={L} k={k}: {e}")
    lat.close()
