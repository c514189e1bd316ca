"""k2_own (csrc/dense_own.hip) against the oracle and the clock: natural order, a caller's order, replicas.
usage: own_probe.py [check|time|all] [n ...]   (TSU_K2_VERBOSE=2 prints the kernel's own timeline)"""
import os
import sys
import time

sys.path.insert(0, "tsu-emulator_amd")
sys.path.insert(0, ".")
import numpy as np
from oracle import oracle as ora
from tsu import _hip as hip

mode = sys.argv[1] if len(sys.argv) > 1 else "all"
sizes = [int(a) for a in sys.argv[2:]] or [2048, 4096, 8192, 16384]
ctx = hip.Context.default()


def system(n, f64=False):
    rng = np.random.default_rng(n)
    G = rng.standard_normal((n, n)).astype(np.float32)
    J = ((G + G.T) / 2 / np.sqrt(n)).astype(np.float32)
    np.fill_diagonal(J, 0.0)
    b = rng.normal(size=n) * 0.1
    return (J.astype(np.float64) if f64 else J), b, rng.integers(0, 2, size=n).astype(np.int8)


for n in sizes:
    J, b, s0 = system(n)
    J64 = J.astype(np.float64)
    d = hip.DenseSystem(J, b, hip.DTYPE_F32, ctx=ctx)
    if mode in ("check", "all"):
        d.set_state(s0)
        d.sweep(1.0, 3, seed=7, sweep0=2)
        got = d.get_state()
        want = ora.dense_sweep_philox(s0, J64, b, 1.0, 3, 7, sweep0=2)
        print(f"n={n} natural: {'OK' if (got == want).all() else 'MISMATCH %d' % int((got != want).sum())}", flush=True)
        rng = np.random.default_rng(1)
        order = np.array([rng.permutation(n) for _ in range(2)])
        d.set_state(s0)
        d.sweep(0.9, 2, seed=3, sweep0=5, order=order)
        got = d.get_state()
        want = ora.dense_sweep_philox(s0, J64, b, 0.9, 2, 3, sweep0=5, order=order)
        print(f"n={n} random order: {'OK' if (got == want).all() else 'MISMATCH %d' % int((got != want).sum())}", flush=True)
        R = 3
        sts = np.array([np.random.default_rng(100 + r).integers(0, 2, size=n) for r in range(R)], dtype=np.int8)
        temps = [1.0, 0.7, 1.6]
        out = d.sweep_replicas(sts, temps, 2, [11, 12, 13], [4, 9, 0], replicas=[0, 1, 2])
        ok = True
        for r in range(R):
            want = ora.dense_sweep_philox(sts[r], J64, b, temps[r], 2, 11 + r, sweep0=[4, 9, 0][r], replica=r)
            ok = ok and (out[r] == want).all()
        print(f"n={n} replicas: {'OK' if ok else 'MISMATCH'}", flush=True)
    if mode in ("time", "all"):
        d.set_state(s0)
        for label, kw in (("natural", {}), ("random", {"order": True})):
            k = 20
            order = np.array([np.random.default_rng(5).permutation(n) for _ in range(k)]) if kw else None
            d.sweep(1.0, 4, seed=1, sweep0=0, order=None if order is None else order[:4])
            ctx.synchronize()
            t0 = time.perf_counter()
            d.sweep(1.0, k, seed=1, sweep0=4, order=order)
            ctx.synchronize()
            print(f"n={n} {label}: {(time.perf_counter() - t0) / k * 1e3:.4f} ms/sweep (incl. upload of the order)", flush=True)
        R = 8
        sts = np.array([np.random.default_rng(100 + r).integers(0, 2, size=n) for r in range(R)], dtype=np.int8)
        d.sweep_replicas(sts, [1.0] * R, 2, list(range(R)), [0] * R)
        t0 = time.perf_counter()
        d.sweep_replicas(sts, [1.0] * R, 20, list(range(R)), [0] * R)
        print(f"n={n} 8 replicas: {(time.perf_counter() - t0) / 20 * 1e3:.4f} ms per all-replica sweep", flush=True)
    d.close()
