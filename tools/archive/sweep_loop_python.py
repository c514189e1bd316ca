"""The reference's idiom state = sampler.gibbs_sweep(state, J) in a loop (J bound once), with an energy read per step.
usage: python tools/sweep_loop_python.py [n ...]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tsu-emulator_amd"))
import numpy as np
from tsu.gibbs import GibbsConfig, GibbsSampler
for n in [int(a) for a in sys.argv[1:]] or (1024, 4096, 16384):
    rng = np.random.default_rng(n)
    J = rng.normal(size=(n, n)) / np.sqrt(n)
    J = ((J + J.T) / 2).astype(np.float32)
    s = GibbsSampler(GibbsConfig(temperature=1.0), coupling_dtype="float32")
    s.bind(J, None)
    state = rng.integers(0, 2, size=n)
    for _ in range(5):
        state = s.gibbs_sweep(state, J, n_sweeps=1)
    t0 = time.perf_counter()
    for k in range(100):
        state = s.gibbs_sweep(state, J, n_sweeps=1)
        e = s.compute_energy(state, J)
    dt = (time.perf_counter() - t0) / 100
    print(f"n={n}: gibbs_sweep + compute_energy per step {dt * 1e3:.3f} ms", flush=True)
    s.unbind()
