"""Whole dense runs above the one-workgroup kernels: sample_boltzmann-like runs and annealing schedules (TSU_K2_RUN_ONE_LAUNCH=0: one
call per recorded state, as before round 2's end).  usage: python tools/dense_runs.py [n ...]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tsu-emulator_amd"))
import numpy as np
from tsu import _hip as hip
ctx = hip.Context.default()
for n in [int(a) for a in sys.argv[1:]] or (600, 1024, 2048, 4096, 16384):
    rng = np.random.default_rng(n)
    G = rng.standard_normal((n, n)).astype(np.float32)
    J = ((G + G.T) / 2 / np.sqrt(n)).astype(np.float32); np.fill_diagonal(J, 0.0)
    d = hip.DenseSystem(J, None, hip.DTYPE_F32, ctx=ctx)
    d.set_state(rng.integers(0, 2, size=n).astype(np.int8))
    ns = 100 if n <= 4096 else 30
    d.sample(1.0, 10, 10, 5, seed=1)
    t0 = time.perf_counter(); d.sample(1.0, 100, 10, ns, seed=1, sweep0=1000); dt = time.perf_counter() - t0
    steps = 1000 if n <= 4096 else 200
    temps = [10.0 * (0.01 / 10.0) ** (k / steps) for k in range(steps)]
    d.anneal(temps[:10], seed=2)
    t0 = time.perf_counter(); d.anneal(temps, seed=2, sweep0=50); da = time.perf_counter() - t0
    print(f"one_launch={os.environ.get('TSU_K2_RUN_ONE_LAUNCH', '1')} n={n}: sample run {100 + 10 * ns} sweeps {dt * 1e3:.1f} ms ({dt / (100 + 10 * ns) * 1e6:.1f} us/sweep); "
          f"annealing {steps} steps {da * 1e3:.1f} ms ({da / steps * 1e6:.1f} us/step)", flush=True)
    d.close()
