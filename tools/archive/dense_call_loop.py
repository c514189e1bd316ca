"""Loops of short dense calls (annealing / tempering pattern): one sweep per call, with and without an energy read after each.
TSU_K2_KEEP_FIELDS=0 switches off the fields kept from call to call.  usage: python tools/dense_call_loop.py [n ...]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tsu-emulator_amd"))
import numpy as np
from tsu import _hip as hip
ctx = hip.Context.default()
for n in [int(a) for a in sys.argv[1:]] or (4096, 16384):
    rng = np.random.default_rng(n)
    G = rng.standard_normal((n, n)).astype(np.float32)
    J = ((G + G.T) / 2 / np.sqrt(n)).astype(np.float32); np.fill_diagonal(J, 0.0)
    d = hip.DenseSystem(J, None, hip.DTYPE_F32, ctx=ctx)
    d.set_state(rng.integers(0, 2, size=n).astype(np.int8))
    for with_energy in (False, True):
        for k in range(4):
            d.sweep(1.0, 1, seed=1, sweep0=k)
        ctx.synchronize()
        t0 = time.perf_counter()
        for k in range(40):
            d.sweep(1.0 - 0.01 * k, 1, seed=1, sweep0=4 + k)
            if with_energy:
                d.energy()
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / 40
        print(f"keep_fields={os.environ.get('TSU_K2_KEEP_FIELDS', '1')} n={n} one sweep per call{' + energy' if with_energy else ''}: {dt * 1e3:.3f} ms per step", flush=True)
    d.close()
