import sys; sys.path.insert(0, "tsu-emulator_amd")
import numpy as np
from tsu import _hip as hip
ctx = hip.Context.default()
dim = 1 << 20
lc = hip.LangevinChains(1, dim, ctx=ctx)
lc.set_energy(2.0, 0.0)
lc.set_state(np.zeros((1, dim), np.float32))
for spl in (0, 1):
    lc.set_kernel(spl)
    lc.step(500, 0.01, 1.0, 1.0, 7, 0)
ctx.synchronize()
lc.close()
