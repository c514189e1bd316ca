import sys, time, cProfile, pstats
sys.path.insert(0, "tsu-emulator_amd"); sys.path.insert(0, ".")
import numpy as np
from tsu.gibbs import GibbsSampler, GibbsConfig
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rng = np.random.default_rng(0)
G = rng.standard_normal((n, n)).astype(np.float32)
J = ((G + G.T) / 2 / np.sqrt(n)).astype(np.float32)
np.fill_diagonal(J, 0.0)
J.setflags(write=False)
s = GibbsSampler(GibbsConfig(temperature=1.0), coupling_dtype="float32")
state = rng.integers(0, 2, size=n)
for _ in range(5):
    state = s.gibbs_sweep(state, J)
    e = s.compute_energy(state, J)
def loop(k):
    global state
    for _ in range(k):
        state = s.gibbs_sweep(state, J)
        e = s.compute_energy(state, J)
t0 = time.perf_counter(); loop(200); t1 = time.perf_counter()
print(f"n={n}: {(t1 - t0) / 200 * 1e3:.3f} ms per step")
pr = cProfile.Profile(); pr.enable(); loop(200); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
