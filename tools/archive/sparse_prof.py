"""K5 profiling workload: 20 colour-parallel sweeps of a 2^24-site open chain (two k5_color launches per sweep)."""
import sys; sys.path.insert(0, "tsu-emulator_amd")
import numpy as np, scipy.sparse as sp
from tsu import _hip as hip
from tsu.graph import canonical_csr
ctx = hip.Context.default()
n = 1 << 24
A = canonical_csr(sp.diags([np.full(n - 1, 4.0), np.full(n - 1, 4.0)], [1, -1]))
order = np.concatenate([np.arange(0, n, 2), np.arange(1, n, 2)]).astype(np.int32)
g = hip.SparseSystem(A.indptr, A.indices, A.data, np.full(n, -8.0), np.array([0, (n + 1) // 2, n], np.int32), order, ctx=ctx)
g.set_state(np.random.default_rng(1).integers(0, 2, size=n).astype(np.int8))
g.sweep(1.7, 20, seed=3, sweep0=0)
ctx.synchronize()
g.close()
