"""Randomised check of the sparse (colour-parallel, K5) kernel against the oracle's sparse twin (development aid):
chains, rings, grids with holes, random graphs of several densities, with and without self-loops / bias, sizes on both
sides of the one-workgroup limit (32768 sites)."""
import os, sys, random
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tsu-emulator_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import scipy.sparse as sp
from tsu import _hip
from tsu.graph import canonical_csr, color_graph
from oracle import oracle as ora
random.seed(int(os.environ.get("FUZZ_SEED", "5")))
for case in range(int(os.environ.get("FUZZ_CASES", "30"))):
    kind = random.choice(["chain", "ring", "graph", "graph", "lattice", "star"])
    n = random.choice([1, 2, 3, 17, 64, 1000, 4097, 32768, 32769, 50000, 131072])
    rng = np.random.default_rng(random.getrandbits(30))
    if kind == "chain" or n < 3:
        A = sp.diags([rng.normal(size=max(n - 1, 0))] * 2, [1, -1], shape=(n, n))
    elif kind == "ring":
        v = rng.normal(size=n)
        A = sp.coo_matrix((v, (np.arange(n), (np.arange(n) + 1) % n)), shape=(n, n))
        A = A + A.T
    elif kind == "lattice":
        r = max(2, int(np.sqrt(n)))
        n = r * r
        idx = np.arange(n).reshape(r, r)
        i = np.concatenate([idx[:, :-1].ravel(), idx[:-1, :].ravel()])
        j = np.concatenate([idx[:, 1:].ravel(), idx[1:, :].ravel()])
        keep = rng.random(i.size) < 0.9
        A = sp.coo_matrix((rng.normal(size=int(keep.sum())), (i[keep], j[keep])), shape=(n, n))
        A = A + A.T
    elif kind == "star":
        A = sp.coo_matrix((rng.normal(size=n - 1), (np.zeros(n - 1, int), np.arange(1, n))), shape=(n, n))
        A = A + A.T
    else:
        m = int(n * random.choice([0.5, 2.0, 5.0]))
        i, j = rng.integers(0, n, m), rng.integers(0, n, m)
        A = sp.coo_matrix((rng.normal(size=m), (i, j)), shape=(n, n))   # includes self-loops when i == j
        A = A + A.T
    A = canonical_csr(A)
    bias = rng.normal(size=n) if random.random() < 0.6 else None
    st = rng.integers(0, 2, size=n).astype(np.int8)
    T = random.choice([0.2, 1.0, 3.0])
    sweeps = random.choice([1, 2, 7])
    seed, s0 = random.getrandbits(40), random.randrange(1000)
    off, order = color_graph(A)
    g = _hip.SparseSystem(A.indptr, A.indices, A.data, bias, off, order)
    g.set_state(st)
    g.sweep(T, sweeps, seed=seed, sweep0=s0)
    want = ora.sparse_sweep_philox(st, A.indptr, A.indices, A.data, bias, T, sweeps, seed, sweep0=s0, order=order)
    ok = (g.get_state() == want).all()
    e, m = g.energy()
    ok = ok and abs(e - ora.sparse_energy(want, A.indptr, A.indices, A.data, bias)) <= 1e-9 * max(1, n) and m == int((2 * want.astype(int) - 1).sum())
    print(("ok  " if ok else "FAIL"), kind, "n", n, "nnz", A.nnz, "colours", len(off) - 1, "T", T, "sweeps", sweeps, flush=True)
    g.close()
    if not ok:
        sys.exit(1)
print("all sparse cases agree")
