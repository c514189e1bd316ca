"""GibbsSampler.simulated_annealing through the Python API on large dense systems.  usage: python tools/anneal_python.py [n ...]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tsu-emulator_amd"))
import numpy as np
from tsu.gibbs import GibbsConfig, GibbsSampler
for n in [int(a) for a in sys.argv[1:]] or (1024, 4096):
    rng = np.random.default_rng(n)
    J = rng.normal(size=(n, n)) / np.sqrt(n)
    J = ((J + J.T) / 2).astype(np.float32)
    s = GibbsSampler(GibbsConfig(temperature=1.0), coupling_dtype="float32")
    s.simulated_annealing(J, T_initial=5.0, T_final=0.05, n_steps=20)
    t0 = time.perf_counter()
    best, e = s.simulated_annealing(J, T_initial=5.0, T_final=0.05, n_steps=1000)
    dt = time.perf_counter() - t0
    print(f"n={n}: 1000 annealing steps {dt * 1e3:.0f} ms ({dt * 1e3:.3f} us per step x 1000), best energy {e:.2f}", flush=True)
