"""temperature_scan on small lattices: all temperatures in one launch per batch of sweeps (tsu_ising2d_sweep_batch)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tsu-emulator_amd"))
import numpy as np
from tsu.models.ising import temperature_scan, IsingModel2D
Ts = np.linspace(1.5, 3.5, 32)
for size in (32, 64, 128, 512, 1024):
    temperature_scan(size, Ts[:2], n_equilibrate=10, n_measure=2)
    t0 = time.perf_counter()
    r = temperature_scan(size, Ts, n_equilibrate=2000, n_measure=50, measure_every=10, seed=5)
    dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    for i, T in enumerate(Ts):  # the former schedule: one temperature after the other
        m = IsingModel2D(size, temperature=float(T), seed=5 + i, initial="up")
        m.equilibrate(n_sweeps=2000)
        for j in range(50):
            m.gibbs_update(10); m.magnetization(); m.energy()
    dt1 = time.perf_counter() - t0
    sweeps = len(Ts) * 2500
    print(f"{size}x{size}, 32 temperatures x 2500 sweeps: batched {dt * 1e3:.0f} ms ({size * size * sweeps / dt:.2e} upd/s), one by one {dt1 * 1e3:.0f} ms; |M|(T=1.5)={r['magnetization'][0]:.3f} |M|(T=3.5)={r['magnetization'][-1]:.3f}", flush=True)
