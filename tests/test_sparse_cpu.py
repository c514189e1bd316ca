"""CPU tests of the sparse (colour-parallel, K5) path's host side and of its oracle twin.

Ladder: reference loop (gibbs.py:128-162) == ora_dense_sweep_philox in a given visiting order (pinned by golden g1/g2 through
test_oracle_golden / test_oracle_twins) == ora_sparse_sweep_philox on the CSR form of the same matrix in the same order (here)
== the HIP kernel (tests/test_sparse_gpu.py)."""
import os
import sys

import numpy as np
import pytest
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tsu-emulator_amd"))
from oracle import oracle as ora  # noqa: E402
from tsu.graph import canonical_csr, color_graph  # noqa: E402


def random_graph(n, density, seed, self_loops=True):
    """Symmetric sparse coupling matrix with about density * n^2 random entries per triangle (pairs drawn directly: scipy's
    sp.random samples without replacement from n^2 slots, minutes at n = 60000) and a diagonal entry on 20 % of the sites."""
    rng = np.random.default_rng(seed)
    m = int(round(density * n * n))
    i, j = rng.integers(0, n, size=m), rng.integers(0, n, size=m)
    keep = i != j
    M = sp.coo_matrix((rng.normal(size=int(keep.sum())), (i[keep], j[keep])), shape=(n, n)).tocsr()
    M = M + M.T
    if self_loops:
        d = np.where(rng.random(n) < 0.2, rng.normal(size=n), 0.0)
        M = M + sp.diags(d)
    return canonical_csr(M)


def colors_of(offsets, order, n):
    col = np.empty(n, dtype=int)
    for c in range(len(offsets) - 1):
        col[order[offsets[c]:offsets[c + 1]]] = c
    return col


@pytest.mark.parametrize("n,density,seed", [(40, 0.1, 1), (64, 0.3, 2), (17, 0.0, 3), (200, 0.02, 4)])
def test_sparse_twin_equals_dense_oracle_in_colour_major_order(n, density, seed):
    A = random_graph(n, density, seed)
    rng = np.random.default_rng(seed)
    bias = rng.normal(size=n)
    st = rng.integers(0, 2, size=n).astype(np.int8)
    offsets, order = color_graph(A)
    col = colors_of(offsets, order, n)
    C = A.tocoo()
    assert not np.any((col[C.row] == col[C.col]) & (C.row != C.col)), "colouring is not proper"
    assert sorted(order.tolist()) == list(range(n))
    n_sweeps = 5
    dense = ora.dense_sweep_philox(st, A.toarray(), bias, 0.9, n_sweeps, 1234, sweep0=3, order=np.tile(order, (n_sweeps, 1)))
    sparse = ora.sparse_sweep_philox(st, A.indptr, A.indices, A.data, bias, 0.9, n_sweeps, 1234, sweep0=3, order=order)
    np.testing.assert_array_equal(dense, sparse)
    e_dense = ora.dense_energy(dense, A.toarray(), bias) if hasattr(ora, "dense_energy") else None
    e_sparse = ora.sparse_energy(sparse, A.indptr, A.indices, A.data, bias)
    b = sparse.astype(float)
    assert abs(e_sparse - (-0.5 * b @ A.toarray() @ b - bias @ b)) < 1e-9
    if e_dense is not None:
        assert abs(e_dense - e_sparse) < 1e-9


def test_colour_parallel_update_equals_the_sequential_loop():
    """Within one colour the sites do not read each other: updating them 'at once' from the state before the colour is
    the sequential loop.  Checked directly against a colour-parallel NumPy evaluation."""
    n = 50
    A = random_graph(n, 0.08, 7)
    rng = np.random.default_rng(7)
    bias = rng.normal(size=n)
    st = rng.integers(0, 2, size=n).astype(np.int8)
    offsets, order = color_graph(A)
    want = ora.sparse_sweep_philox(st, A.indptr, A.indices, A.data, bias, 1.3, 1, 5, order=order)
    cur = st.astype(float)
    for c in range(len(offsets) - 1):
        sites = order[offsets[c]:offsets[c + 1]]
        field = A[sites] @ cur + bias[sites]
        for i, f in zip(sites, field):
            x = f / 1.3
            p = 1.0 if x > 20 else 0.0 if x < -20 else 1.0 / (1.0 + np.exp(-x))
            cur[i] = 1.0 if ora.dense_uniform(int(i), 0, 5) < p else 0.0
    np.testing.assert_array_equal(cur.astype(np.int8), want)


def test_colouring_chain_ring_tree_and_odd_ring():
    n = 1001
    chain = canonical_csr(sp.diags([np.ones(n - 1), np.ones(n - 1)], [1, -1]))
    off, order = color_graph(chain)
    assert len(off) == 3 and off[1] == 501
    assert np.array_equal(order[:off[1]], np.arange(0, n, 2)) and np.array_equal(order[off[1]:], np.arange(1, n, 2))
    ring = chain.tolil()
    ring[0, n - 1] = ring[n - 1, 0] = 1.0           # odd ring: not bipartite
    off, order = color_graph(canonical_csr(ring))
    assert len(off) - 1 >= 3
    col = colors_of(off, order, n)
    C = canonical_csr(ring).tocoo()
    assert not np.any(col[C.row] == col[C.col])
    even = canonical_csr(sp.diags([np.ones(n), np.ones(n)], [1, -1], shape=(n + 1, n + 1))).tolil()
    even[0, n] = even[n, 0] = 1.0                   # even ring: bipartite
    off, order = color_graph(canonical_csr(even))
    assert len(off) == 3
    empty = canonical_csr(sp.csr_matrix((5, 5)))
    off, order = color_graph(empty)
    assert off.tolist() == [0, 5] and order.tolist() == [0, 1, 2, 3, 4]


def test_sparse_ising_models_host_side():
    from tsu.models.ising import IsingChain, IsingConfig, IsingModel
    big = IsingChain(100000, J=-0.5, config=IsingConfig(external_field=0.25))
    assert big.sparse and big.J_sparse.nnz == 2 * 99999
    with pytest.raises(MemoryError):
        big.J
    s = np.where(np.arange(100000) % 2 == 0, 1, -1)
    assert big.energy(s) == pytest.approx(-(-0.5 * -1.0) * 99999 - 0.25 * s.sum())
    small_sparse, small_dense = IsingChain(9, J=0.7, graph="sparse"), IsingChain(9, J=0.7)
    np.testing.assert_array_equal(small_sparse.J, small_dense.J)
    np.testing.assert_allclose(small_sparse._get_bit_bias(), small_dense._get_bit_bias())
    np.testing.assert_array_equal(small_sparse._get_bit_coupling().toarray(), small_dense._get_bit_coupling())
    m = IsingModel(6, graph="sparse")
    m.set_coupling(0, 5, 2.0)
    m.set_coupling(0, 5, -1.0)                      # assignment, not accumulation (ising.py:77-86)
    assert m.J[0, 5] == -1.0 and m.J[5, 0] == -1.0
    with pytest.raises(ValueError):
        m.J[0, 1] = 3.0                             # the dense view of a sparse model is a read-only copy
    with pytest.raises(ValueError):
        IsingModel(4, graph="banana")
