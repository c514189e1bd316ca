"""K5: ms per sweep of an open chain of n sites (regular colour classes; TSU_K5_PAIR=0 / TSU_K5_STENCIL=0 switch the fast paths off).
usage: sparse_time.py [log2 n] [sweeps]"""
import sys
sys.path.insert(0, "tsu-emulator_amd"); sys.path.insert(0, ".")
import numpy as np
import scipy.sparse as sp
from tsu import _hip as hip
from tsu.graph import canonical_csr
ctx = hip.Context.default()
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 24)
k = int(sys.argv[2]) if len(sys.argv) > 2 else 50
A = canonical_csr(sp.diags([np.full(n - 1, 4.0), np.full(n - 1, 4.0)], [1, -1]))
bias = np.full(n, -8.0)
bias[0] = bias[-1] = -4.0
order = np.concatenate([np.arange(0, n, 2), np.arange(1, n, 2)]).astype(np.int32)
g = hip.SparseSystem(A.indptr, A.indices, A.data, bias, np.array([0, (n + 1) // 2, n], np.int32), order, ctx=ctx)
g.set_state(np.random.default_rng(1).integers(0, 2, size=n).astype(np.int8))
g.sweep(1.7, 5, seed=3, sweep0=0)
ctx.synchronize()
best = 1e9
for rep in range(3):
    ctx.timer_begin()
    g.sweep(1.7, k, seed=3, sweep0=5 + rep * k)
    best = min(best, ctx.timer_end() / k)
st = g.get_state()
print(f"n=2^{int(np.log2(n))}: {best * 1e3:.2f} us per sweep = {n / (best * 1e-3):.3e} updates/s  checksum {int(st.sum())} {int((st * np.arange(n) % 1000003).sum())}")
