"""A/B timing of lattice sweeps between builds of the library on ONE box (boxes differ by several per cent, so numbers from
different gpurun calls do not compare).  Each build runs in its own child process (TSU_HIP_LIB), alternating, several rounds.
usage: python tools/ab_lattice.py libA.so,libB.so[,libC.so ...] [L ...]      child: python tools/ab_lattice.py --child L [L ...]"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(sizes):
    sys.path.insert(0, os.path.join(ROOT, "tsu-emulator_amd"))
    from tsu import _hip
    ctx = _hip.Context(0)
    for L in sizes:
        lat = _hip.Lattice(L, L, True, ctx=ctx)
        lat.randomize(1)
        lat.set_thresholds(_hip.ising2d_thresholds(1.0, 0.0, 2.269185))
        n = 256 if L <= 8192 else 240
        for _ in range(6):
            lat.sweep(n, 7, 0)
        ctx.synchronize()
        best = 1e9
        for rep in range(5):
            t0 = time.perf_counter()
            for i in range(10):
                lat.sweep(n, 7, n * (1 + i + 10 * rep))
            ctx.synchronize()
            best = min(best, (time.perf_counter() - t0) / (10 * n))
        print(f"L={L} {best * 1e6:.3f} us/sweep {L * L / best:.4e} upd/s", flush=True)
        lat.close()


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child([int(a) for a in sys.argv[2:]])
    else:
        libs = sys.argv[1].split(",")
        sizes = sys.argv[2:] or ["4096", "8192"]
        for rnd in range(3):
            for lib in libs:
                env = dict(os.environ, TSU_HIP_LIB=os.path.abspath(lib))
                out = subprocess.run([sys.executable, __file__, "--child"] + sizes, env=env, capture_output=True, text=True)
                for line in out.stdout.splitlines():
                    print(f"round {rnd} {os.path.basename(lib):24s} {line}", flush=True)
                if out.returncode:
                    print(out.stderr[-2000:])
                    sys.exit(1)
