"""K3 coupled: us per step of Langevin chains on E = 1/2 x^T A x (k3_coupled), and the bytes of A per second.
usage: langevin_coupled_time.py [dim] [chains] [steps]"""
import sys
sys.path.insert(0, "tsu-emulator_amd"); sys.path.insert(0, ".")
import numpy as np
from tsu import _hip as hip
ctx = hip.Context.default()
d = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
chains = int(sys.argv[2]) if len(sys.argv) > 2 else 1
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
rng = np.random.default_rng(1)
A = np.zeros((d, d), np.float32)
A[np.arange(d), np.arange(d)] = 2.0
i = np.arange(d - 1)
A[i, i + 1] = A[i + 1, i] = 0.5  # (any symmetric positive definite matrix: the kernel reads all of it)
lc = hip.LangevinChains(chains, d, ctx=ctx)
lc.set_coupling(A)
lc.set_state(np.zeros((chains, d), np.float32))
lc.step(5, 0.01, 1.0, 1.0, 7)
ctx.synchronize()
best = 1e9
for rep in range(3):
    ctx.timer_begin()
    lc.step(steps, 0.01, 1.0, 1.0, 7, step0=5 + rep * steps)
    best = min(best, ctx.timer_end() / steps)
P = (d + 63) // 64 * 64
cb = 8
while cb > 1 and cb // 2 >= chains:
    cb //= 2
blocks = (chains + cb - 1) // cb
print(f"d={d} chains={chains}: {best * 1e3:.1f} us per step; A read {blocks} x per step = {blocks * P * P * 4 / (best * 1e-3) / 1e12:.2f} TB/s; "
      f"{chains * d / (best * 1e-3):.3e} element-steps/s; var {float(lc.get_state().var()):.4f}")
lc.close()
