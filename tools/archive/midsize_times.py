"""us per sweep of mid-size square lattices on each kernel that takes them (development aid)."""
import os, sys, time, zlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tsu-emulator_amd"))
from tsu import _hip
ctx = _hip.Context.default()
names = {_hip.KERNEL_AUTO: "auto", _hip.KERNEL_SMALL: "small", _hip.KERNEL_TILED: "tiled", _hip.KERNEL_GENERIC: "generic"}
for periodic in (True, False):
    for L in (64, 96, 128, 160, 192, 200, 256, 272, 320, 384, 512):
        out, sums = [], set()
        for kern in (_hip.KERNEL_AUTO, _hip.KERNEL_SMALL, _hip.KERNEL_TILED, _hip.KERNEL_GENERIC):
            lat = _hip.Lattice(L, L, periodic)
            try:
                lat.set_kernel(kern)
                lat.randomize(1)
                lat.set_model(1.0, 0.0, 2.269185)
                lat.sweep(64, 1, 0)
            except Exception as e:
                out.append("%s   n/a  " % names[kern])
                lat.close()
                continue
            ctx.synchronize()
            n = 1024
            t = time.perf_counter()
            lat.sweep(n, 1, 64)
            ctx.synchronize()
            dt = time.perf_counter() - t
            sums.add(zlib.crc32(lat.get_spins().tobytes()))
            out.append("%s %6.2f" % (names[kern], dt / n * 1e6))
            lat.close()
        print("L=%4d %s  us/sweep: " % (L, "periodic" if periodic else "open    ") + "  ".join(out) + ("  EQUAL" if len(sums) == 1 else "  DIFFER"), flush=True)
