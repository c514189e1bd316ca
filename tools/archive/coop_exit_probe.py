"""Exit-time crash probe: one dense sweep through hipLaunchCooperativeKernel (TSU_COOP_LAUNCH=1) or the
ordinary launch, then /proc/self/maps is written to gpurun_out/ so that the frames glog prints at a crash can be attributed
to libraries.  usage: coop_exit_probe.py TAG [close|noclose] [n]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tsu-emulator_amd"))
import numpy as np  # noqa: E402
from tsu import _hip  # noqa: E402

tag = sys.argv[1]
close = len(sys.argv) > 2 and sys.argv[2] == "close"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
rng = np.random.default_rng(1)
J = rng.standard_normal((n, n)).astype(np.float32) / np.sqrt(n)
J = (J + J.T) / 2
ctx = _hip.Context(0)
d = _hip.DenseSystem(J, None, _hip.DTYPE_F32, ctx=ctx)
d.set_state(rng.integers(0, 2, size=n).astype(np.int8))
d.sweep(1.0, 3, seed=1, sweep0=0)
ctx.synchronize()
print(tag, "state sum", int(d.get_state().sum()), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", f"coop_probe_maps_{tag}.txt"), "w") as f:
    f.write(open("/proc/self/maps").read())
if close:
    d.close()
    ctx.lib.tsu_shutdown(ctx.h)
    ctx.h = None
print(tag, "leaving", flush=True)
