"""K3, the HBM-bound case: 256 chains x 2^20 (1 GiB of state), ONE step per launch; us per launch and TB/s at 8 B per element-step."""
import sys
sys.path.insert(0, "tsu-emulator_amd"); sys.path.insert(0, ".")
import numpy as np
from tsu import _hip as hip
ctx = hip.Context.default()
chains, dim, steps = 256, 1 << 20, 20
lc = hip.LangevinChains(chains, dim, ctx=ctx)
lc.set_energy(2.0, 0.0)
lc.set_state(np.zeros((1, dim), np.float32))
lc.set_kernel(1)
lc.step(4, 0.01, 1.0, 1.0, 7, 0)
ctx.synchronize()
best = 1e30
for rep in range(3):
    ctx.timer_begin()
    lc.step(steps, 0.01, 1.0, 1.0, 7, 4 + rep * steps)
    best = min(best, ctx.timer_end())
print(f"{best * 1e3 / steps:.1f} us per launch = {8.0 * chains * dim * steps / (best * 1e-3) / 1e12:.2f} TB/s; var {float(lc.get_state()[:4].var()):.4f}")
lc.close()
