"""K5 parity (-m gpu): the colour-parallel sparse kernel through the C ABI against the oracle's sparse twin (bit-exact),
the drop-in surface on sparse models, and exact 1-D results at a size the reference's dense matrix cannot reach."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import oracle as ora
from test_sparse_cpu import colors_of, random_graph  # noqa: F401  (tests/ is on sys.path: rootdir conftest, prepend import mode)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    from tsu import _hip
    _hip.Context.default()


def _chain(n, J=1.0):
    from tsu.graph import canonical_csr
    return canonical_csr(sp.diags([np.full(n - 1, J), np.full(n - 1, J)], [1, -1]))


@pytest.mark.parametrize("name,n", [("chain", 1), ("chain", 2), ("chain", 7), ("chain", 4097), ("chain", 32768), ("chain", 32769),
                                    ("chain", 300001), ("graph", 5000), ("graph", 60000), ("empty", 1000)])
def test_sparse_sweep_matches_oracle(name, n):
    from tsu import _hip
    from tsu.graph import canonical_csr, color_graph
    rng = np.random.default_rng(n)
    if name == "chain":
        A = _chain(n, 0.8) if n > 1 else canonical_csr(sp.csr_matrix((1, 1)))
    elif name == "graph":
        A = random_graph(n, 3.0 / n, n)            # mean degree ~6 incl. symmetrisation, self-loops on 20 % of the sites
    else:
        A = canonical_csr(sp.csr_matrix((n, n)))
    bias = rng.normal(size=n)
    st = rng.integers(0, 2, size=n).astype(np.int8)
    offsets, order = color_graph(A)
    g = _hip.SparseSystem(A.indptr, A.indices, A.data, bias, offsets, order)
    g.set_state(st)
    np.testing.assert_array_equal(g.get_state(), st)
    g.sweep(0.9, 3, seed=99, sweep0=5)
    want = ora.sparse_sweep_philox(st, A.indptr, A.indices, A.data, bias, 0.9, 3, 99, sweep0=5, order=order)
    got = g.get_state()
    np.testing.assert_array_equal(got, want)
    e, m = g.energy()
    assert e == pytest.approx(ora.sparse_energy(want, A.indptr, A.indices, A.data, bias), rel=1e-12, abs=1e-9 * n)
    assert m == int((2 * want.astype(int) - 1).sum())
    # sampling run: burn-in 2, 3 samples 2 sweeps apart, counters continue
    smp = g.sample(0.9, 2, 2, 3, seed=99, sweep0=8)
    cur = want
    cur = ora.sparse_sweep_philox(cur, A.indptr, A.indices, A.data, bias, 0.9, 2, 99, sweep0=8, order=order)
    for k in range(3):
        cur = ora.sparse_sweep_philox(cur, A.indptr, A.indices, A.data, bias, 0.9, 2, 99, sweep0=10 + 2 * k, order=order)
        np.testing.assert_array_equal(smp[k], cur)
    np.testing.assert_array_equal(g.get_state(), cur)
    g.close()


def test_sparse_create_rejects_bad_input():
    from tsu import _hip
    A = _chain(6)
    off, order = np.array([0, 3, 6], np.int32), np.array([0, 2, 4, 1, 3, 5], np.int32)
    _hip.SparseSystem(A.indptr, A.indices, A.data, None, off, order).close()
    with pytest.raises(ValueError, match="same colour"):
        _hip.SparseSystem(A.indptr, A.indices, A.data, None, off, np.array([0, 1, 2, 3, 4, 5], np.int32))
    with pytest.raises(ValueError, match="permutation"):
        _hip.SparseSystem(A.indptr, A.indices, A.data, None, off, np.array([0, 2, 4, 1, 3, 3], np.int32))
    with pytest.raises(ValueError, match="offsets"):
        _hip.SparseSystem(A.indptr, A.indices, A.data, None, np.array([0, 3, 5], np.int32), order)
    g = _hip.SparseSystem(A.indptr, A.indices, A.data, None, off, order)
    with pytest.raises(ValueError, match="Temperature must be positive"):
        g.sweep(0.0, 1)
    with pytest.raises(ValueError, match="state must be 0/1"):  # like the dense path: no other int8 value may reach the fields
        g.set_state(np.array([0, 1, 2, 0, 1, 0], np.int8))
    g.close()


def test_gibbs_sampler_accepts_scipy_sparse_couplings():
    """gibbs_sweep / sample_boltzmann / compute_energy with a scipy.sparse J == the reference loop (dense oracle) run in
    the colour-major visiting order, with the Philox uniforms of (site, sweep)."""
    from tsu.gibbs import GibbsConfig, GibbsSampler
    from tsu.graph import color_graph
    n = 300
    A = random_graph(n, 0.02, 11)
    rng = np.random.default_rng(11)
    bias = rng.normal(size=n)
    st = rng.integers(0, 2, size=n)
    _, order = color_graph(A)
    s = GibbsSampler(GibbsConfig(temperature=0.7, n_burnin=3, n_sweeps=2), seed=4242)
    out = s.gibbs_sweep(st, A, bias, n_sweeps=4)
    want = ora.dense_sweep_philox(st.astype(np.int8), A.toarray(), bias, 0.7, 4, 4242, sweep0=0, order=np.tile(order, (4, 1)))
    np.testing.assert_array_equal(out, want)
    assert out.dtype == st.dtype and s._sweep_counter == 4
    smp = s.sample_boltzmann(A, bias, n_samples=5, initial_state=out)
    assert smp.shape == (5, n) and smp.dtype == np.zeros(1, dtype=int).dtype and s.sample_count == 5
    cur = ora.dense_sweep_philox(want, A.toarray(), bias, 0.7, 3, 4242, sweep0=4, order=np.tile(order, (3, 1)))
    for k in range(5):
        cur = ora.dense_sweep_philox(cur, A.toarray(), bias, 0.7, 2, 4242, sweep0=7 + 2 * k, order=np.tile(order, (2, 1)))
        np.testing.assert_array_equal(smp[k], cur)
    b = cur.astype(float)
    assert s.compute_energy(cur, A, bias) == pytest.approx(-0.5 * b @ A.toarray() @ b - bias @ b, abs=1e-9)
    with pytest.raises(ValueError):
        GibbsSampler(rng="numpy").gibbs_sweep(st, A, bias)
    with pytest.raises(ValueError, match="square"):
        s.sample_boltzmann(sp.csr_matrix((3, 4)))


def test_sparse_three_spin_model_samples_the_exact_boltzmann_distribution(golden):
    """The reference's own 3-spin fixture (g9: chain + a frustrating bond, field 0.2, T = 1.5): histogram of 40 000 samples
    of the sparse path against exact enumeration (both bias modes; compat = the reference's shipped sign)."""
    from tsu.models.ising import IsingConfig, IsingModel
    g9 = golden("g9_distribution")
    J, h, T = g9["n3_J"], g9["n3_h"], float(g9["n3_T"])
    for mode in ("physical", "compat"):
        m = IsingModel(3, config=IsingConfig(temperature=T, external_field=0.2, n_burnin=50, n_sweeps=3), bias_mode=mode, graph="sparse")
        m.set_coupling(0, 1, 1.0)
        m.set_coupling(1, 2, 1.0)
        m.set_coupling(0, 2, -0.5)
        np.testing.assert_array_equal(m.J, J)
        np.random.seed(5)
        s = m.sample(40000)
        code = ((s + 1) // 2).dot(1 << np.arange(3))
        hist = np.bincount(code, minlength=8) / 40000.0
        # exact distribution of what the sampler is fed: bits with couplings 4J and the mode's bias
        Jb, hb = 4 * J, m._get_bit_bias()
        states = np.array([[(c >> k) & 1 for k in range(3)] for c in range(8)], dtype=float)
        E = np.array([-0.5 * b @ Jb @ b - hb @ b for b in states])
        p = np.exp(-E / T)
        p /= p.sum()
        assert np.abs(hist - p).max() < 0.01, (mode, hist, p)
        ref = g9[f"n3_hist_{mode}"] / 4000.0        # the reference's own 4000-sample histogram
        assert np.abs(hist - ref).max() < 0.04, (mode, hist, ref)


def test_million_site_chain_nearest_neighbour_correlation():
    """IsingChain with 10^6 sites (the reference's dense J would need 8 TB): open chain, h = 0, physical bias:
    <s_i s_{i+1}> = tanh(J/T) exactly, <s_i s_{i+2}> = tanh^2(J/T)."""
    from tsu.models.ising import IsingChain, IsingConfig
    n, J, T = 1_000_000, 1.0, 1.7
    c = IsingChain(n, J=J, config=IsingConfig(temperature=T, n_burnin=200, n_sweeps=5), bias_mode="physical")
    assert c.sparse
    np.random.seed(2)
    s = c.sample(4, initial_state=None).astype(np.int8)
    assert s.shape == (4, n) and set(np.unique(s)) == {-1, 1}
    t = np.tanh(J / T)
    for k in range(4):
        c1 = np.mean(s[k, :-1] * s[k, 1:])
        c2 = np.mean(s[k, :-2] * s[k, 2:])
        assert abs(c1 - t) < 0.004 and abs(c2 - t * t) < 0.005, (c1, c2, t)
        assert abs(c.energy(s[k]) / n + J * t) < 0.004
    assert abs(c.magnetization(s)) < 0.01
    gs, e = c.find_ground_state(n_steps=60)
    assert e / n < -0.9                              # annealed towards the ferromagnetic ground state (domains remain)


def _regular(shape, n):
    from tsu.graph import canonical_csr
    if shape in ("chain", "chain_edge_bias"):
        A = _chain(n, 0.8)
    elif shape == "ring":
        A = canonical_csr(sp.diags([np.full(n - 1, 0.8), np.full(n - 1, 0.8), [0.8], [0.8]], [1, -1, n - 1, -(n - 1)]))
    else:  # a two-leg ladder of n / 2 rungs: site 2 i + r, legs along i, rungs between r = 0 and r = 1 (degree 3 inside)
        L = n // 2
        i = np.arange(L - 1)
        rows = np.concatenate([2 * i, 2 * i + 1, 2 * np.arange(L)])
        cols = np.concatenate([2 * i + 2, 2 * i + 3, 2 * np.arange(L) + 1])
        B = sp.coo_matrix((np.full(rows.size, -0.6), (rows, cols)), shape=(n, n))
        A = canonical_csr(B + B.T)
    bias = np.full(n, 0.3)
    if shape == "chain_edge_bias":  # the bit bias of an open chain (tsu/models/ising.py:148): degree-dependent at the two ends
        bias = np.full(n, -1.6)
        bias[0] = bias[-1] = -0.8
    return A, bias


@pytest.mark.parametrize("shape,n", [("chain", 300001), ("chain_edge_bias", 70001), ("ring", 100000), ("ladder", 80000)])
def test_regular_colour_classes_on_the_stencil_kernel_match_the_oracle(shape, n):
    """Uniform couplings and bias, neighbours at fixed position offsets (`IsingChain`, rings, ladders): the colour classes run
    k5_stencil (no CSR streams, integer thresholds from the count of set neighbours) with their few end rows on the generic
    kernel -- the same bits as the oracle's sequential loop on the CSR rows."""
    from tsu import _hip
    from tsu.graph import color_graph
    A, bias = _regular(shape, n)
    st = np.random.default_rng(n).integers(0, 2, size=n).astype(np.int8)
    offsets, order = color_graph(A)
    g = _hip.SparseSystem(A.indptr, A.indices, A.data, bias, offsets, order)
    g.set_state(st)
    g.sweep(1.1, 4, seed=17, sweep0=3)
    want = ora.sparse_sweep_philox(st, A.indptr, A.indices, A.data, bias, 1.1, 4, 17, sweep0=3, order=order)
    np.testing.assert_array_equal(g.get_state(), want)
    g.sweep(0.4, 2, seed=17, sweep0=7)  # another temperature: the thresholds are those of the call
    want = ora.sparse_sweep_philox(want, A.indptr, A.indices, A.data, bias, 0.4, 2, 17, sweep0=7, order=order)
    np.testing.assert_array_equal(g.get_state(), want)
    g.close()


@pytest.mark.parametrize("n", [300008, 300004, 262144 + 2, 4099])
@pytest.mark.parametrize("pair,v4,tie", [(1, 1, 0), (0, 1, 0), (1, 0, 0), (1, 1, 1), (0, 1, 1)])
def test_chain_classes_in_pairs_and_four_positions_per_thread(n, pair, v4, tie, monkeypatch):
    """The two colour classes of a chain share their Philox blocks index by index: the first class's launch prepares the second
    class's decisions by neighbour count (k5_stencil4<1> / <2>), four positions per thread when the classes start at multiples
    of 4 (n = 300008, 262146: both; 300004: the second class starts at 150002 and runs one position per thread on codes written as
    dwords).  Every combination of the two switches gives the oracle's bits; so does a chain just above the one-workgroup kernel's
    range (n = 4099 runs k5_small: the switches must not matter there).  tie = 1: every wave takes the 64-bit compares that a draw whose
    leading 27 bits equal a threshold's needs (one in 2^27: never met by chance in a test)."""
    from tsu import _hip
    from tsu.graph import color_graph
    monkeypatch.setenv("TSU_K5_PAIR", str(pair))
    monkeypatch.setenv("TSU_K5_V4", str(v4))
    monkeypatch.setenv("TSU_K5_TEST_TIE", str(tie))
    A, bias = _regular("chain_edge_bias", n)
    st = np.random.default_rng(n).integers(0, 2, size=n).astype(np.int8)
    offsets, order = color_graph(A)
    g = _hip.SparseSystem(A.indptr, A.indices, A.data, bias, offsets, order)
    g.set_state(st)
    g.sweep(0.9, 3, seed=23, sweep0=1)
    want = ora.sparse_sweep_philox(st, A.indptr, A.indices, A.data, bias, 0.9, 3, 23, sweep0=1, order=order)
    np.testing.assert_array_equal(g.get_state(), want)
    got = g.sample(2.5, 2, 1, 2, seed=23, sweep0=4)  # 2 sweeps of burn-in, then a sample after every sweep
    want = ora.sparse_sweep_philox(want, A.indptr, A.indices, A.data, bias, 2.5, 2, 23, sweep0=4, order=order)
    for k in range(2):
        want = ora.sparse_sweep_philox(want, A.indptr, A.indices, A.data, bias, 2.5, 1, 23, sweep0=6 + k, order=order)
        np.testing.assert_array_equal(got[k], want, err_msg=f"sample {k}")
    g.close()


def test_odd_sites_first_chain_pairs_too():
    """A colouring that visits the ODD sites first: the pair's first class holds the z, w words' sites, the codes are made from x, y."""
    from tsu import _hip
    n = 200000
    A, bias = _regular("chain", n)
    order = np.concatenate([np.arange(1, n, 2), np.arange(0, n, 2)]).astype(np.int32)
    offsets = np.array([0, n // 2, n], np.int32)
    st = np.random.default_rng(5).integers(0, 2, size=n).astype(np.int8)
    g = _hip.SparseSystem(A.indptr, A.indices, A.data, bias, offsets, order)
    g.set_state(st)
    g.sweep(1.3, 3, seed=29, sweep0=0)
    want = ora.sparse_sweep_philox(st, A.indptr, A.indices, A.data, bias, 1.3, 3, 29, sweep0=0, order=order)
    np.testing.assert_array_equal(g.get_state(), want)
    g.close()
