"""Soak of the tile-resident kernel's strip exchange: N generations in few launches == the same sweeps issued as one
8-sweep launch per call (never resident).  usage: soak_resident.py [L] [sweeps]"""
import os, sys, time, zlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tsu-emulator_amd"))
from tsu import _hip
L = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
total = int(sys.argv[2]) if len(sys.argv) > 2 else 400000
ctx = _hip.Context.default()
a = _hip.Lattice(L, L, True); a.randomize(5); a.set_model(1.0, 0.0, 2.269185); a.set_kernel(_hip.KERNEL_AUTO, 8)
b = _hip.Lattice(L, L, True); b.randomize(5); b.set_model(1.0, 0.0, 2.269185); b.set_kernel(_hip.KERNEL_TILED, 8)
t0 = time.perf_counter()
done = 0
while done < total:
    n = min(80000, total - done)
    a.sweep(n, 5, done)                      # resident: 10 launches of 1024 generations
    for s in range(0, n, 8):
        b.sweep(8, 5, done + s)              # one launch per call
    done += n
    ok = zlib.crc32(a.get_spins().tobytes()) == zlib.crc32(b.get_spins().tobytes())
    print(f"{done} sweeps: {'equal' if ok else 'DIFFERENT'}  ({time.perf_counter() - t0:.0f} s)", flush=True)
    if not ok:
        sys.exit(1)
print("soak ok", a.observables())
