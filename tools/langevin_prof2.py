"""K3 profiling workloads.  argv[1]: fused (d = 2^20, 500 steps in one launch) | hbm (256 chains x 2^20, one step per launch, 20 steps)"""
import sys; sys.path.insert(0, "tsu-emulator_amd")
import numpy as np
from tsu import _hip as hip
ctx = hip.Context.default()
dim = 1 << 20
mode = sys.argv[1]
if mode == "fused":
    lc = hip.LangevinChains(1, dim, ctx=ctx)
    lc.set_energy(2.0, 0.0)
    lc.set_state(np.zeros((1, dim), np.float32))
    lc.set_kernel(0)
    for rep in range(4):
        lc.step(500, 0.01, 1.0, 1.0, 7, 500 * rep)
else:
    lc = hip.LangevinChains(256, dim, ctx=ctx)
    lc.set_energy(2.0, 0.0)
    lc.set_state(np.zeros((1, dim), np.float32))
    lc.set_kernel(1)
    lc.step(20, 0.01, 1.0, 1.0, 7, 0)
ctx.synchronize()
lc.close()
