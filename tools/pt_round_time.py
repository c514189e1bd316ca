"""A tempering round at the C-ABI level: one sweep of R replicas on one J (tsu_dense_sweep_replicas), then every replica's energy
(set_state + energy, as GibbsSampler.parallel_tempering asks for them); ms per round.
usage: pt_round_time.py [n] [R] [rounds]"""
import sys, time
sys.path.insert(0, "tsu-emulator_amd"); sys.path.insert(0, ".")
import numpy as np
from tsu import _hip as hip
ctx = hip.Context.default()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
R = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 20
rng = np.random.default_rng(n)
G = rng.standard_normal((n, n)).astype(np.float32)
J = ((G + G.T) / 2 / np.sqrt(n)).astype(np.float32)
np.fill_diagonal(J, 0.0)
d = hip.DenseSystem(J, None, hip.DTYPE_F32, ctx=ctx)
sts = rng.integers(0, 2, size=(R, n)).astype(np.int8)
temps = [1.0 + 0.1 * r for r in range(R)]
sts = d.sweep_replicas(sts, temps, 2, list(range(R)), [0] * R)
t_sw = t_en = 0.0
for k in range(rounds):
    t0 = time.perf_counter()
    sts = d.sweep_replicas(sts, temps, 1, list(range(R)), [2 + k] * R)
    t1 = time.perf_counter()
    es = []
    for r in range(R):
        d.set_state(sts[r])
        es.append(d.energy())
    t2 = time.perf_counter()
    t_sw += t1 - t0
    t_en += t2 - t1
    if k % 5 == 4:  # an exchange of neighbours now and then
        sts[[0, 1]] = sts[[1, 0]]
print(f"n={n} R={R}: {t_sw / rounds * 1e3:.3f} ms per one-sweep replica call + {t_en / rounds * 1e3:.3f} ms for the {R} energies = "
      f"{(t_sw + t_en) / rounds * 1e3:.3f} ms per tempering round; E0 {es[0]:.6f}")
