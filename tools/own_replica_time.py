"""ms per all-replica sweep of R replicas on one J (tsu_dense_sweep_replicas on k2_own); TSU_K2_OWN_SB varies the superblock.
usage: own_replica_time.py [n] [R] [sweeps]"""
import sys, time
sys.path.insert(0, "tsu-emulator_amd"); sys.path.insert(0, ".")
import os
import numpy as np
from tsu import _hip as hip
ctx = hip.Context.default()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
R = int(sys.argv[2]) if len(sys.argv) > 2 else 8
k = int(sys.argv[3]) if len(sys.argv) > 3 else 8
rng = np.random.default_rng(n)
G = rng.standard_normal((n, n)).astype(np.float32)
J = ((G + G.T) / 2 / np.sqrt(n)).astype(np.float32)
np.fill_diagonal(J, 0.0)
d = hip.DenseSystem(J.astype(np.float64) if os.environ.get('TSU_TOOL_F64') else J, None, hip.DTYPE_F64 if os.environ.get('TSU_TOOL_F64') else hip.DTYPE_F32, ctx=ctx)  # (TSU_TOOL_F64=1: the fp64 kernels)
sts = np.array([np.random.default_rng(100 + r).integers(0, 2, size=n) for r in range(R)], dtype=np.int8)
sts = d.sweep_replicas(sts, [1.0] * R, 2, list(range(R)), [0] * R)
best = 1e9
for rep in range(3):
    t0 = time.perf_counter()
    sts = d.sweep_replicas(sts, [1.0] * R, k, list(range(R)), [2 + rep * k] * R)
    best = min(best, (time.perf_counter() - t0) / k * 1e3)
print(f"n={n} R={R}: {best:.4f} ms per all-replica sweep (best of 3 calls of {k} sweeps)  checksum {int(sts.sum())}  launches {d.launch_counts()}")
