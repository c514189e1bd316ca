"""Per-phase timeline of the dense pipeline (TSU_K2_VERBOSE=2 prints it from the library).  usage: python tools/k2_timeline.py [n ...]"""
import os, sys
os.environ.setdefault("TSU_K2_VERBOSE", "2")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tsu-emulator_amd"))
import numpy as np
from tsu import _hip as hip
ctx = hip.Context.default()
for n in [int(a) for a in sys.argv[1:]] or (16384,):
    rng = np.random.default_rng(n)
    G = rng.standard_normal((n, n)).astype(np.float32)
    J = ((G + G.T) / 2 / np.sqrt(n)).astype(np.float32); np.fill_diagonal(J, 0.0)
    d = hip.DenseSystem(J, None, hip.DTYPE_F32, ctx=ctx)
    d.set_state(rng.integers(0, 2, size=n).astype(np.int8))
    d.sweep(1.0, 4, seed=1, sweep0=0); ctx.synchronize()
    d.sweep(1.0, 20, seed=1, sweep0=4); ctx.synchronize()
    d.close()
