"""Parity ladder inside the oracle (CPU only): the device-order twins (checkerboard + Philox) are the
same Markov kernel as the reference-order restatement that the golden vectors pin.

  reference  ==(golden)==  dense sequential restatement
  dense restatement visited in checkerboard order with the Philox uniforms  ==  lattice twin
  lattice twin  ==(tests -m gpu)==  HIP kernel
"""
import numpy as np
import pytest

from oracle import oracle as ora


@pytest.mark.parametrize("rows,cols,periodic", [(4, 4, True), (6, 8, True), (5, 7, False), (1, 9, False),
                                                (2, 2, False), (8, 4, False), (3, 1, False), (1, 1, False)])
@pytest.mark.parametrize("mode", [ora.MODE_PHYSICAL, ora.MODE_COMPAT])
@pytest.mark.parametrize("Jc,h,T", [(1.0, 0.0, 2.269185), (-0.7, 0.3, 1.1), (0.4, -0.2, 0.05)])
def test_lattice_twin_equals_dense_restatement(rows, cols, periodic, mode, Jc, h, T):
    """Stencil twin == ref_gibbs_sweep on IsingGrid's dense 4J / h_bit, visiting colour 0 then colour 1,
    with uniforms u32 / 2^32 (threshold rounding can differ only for one u32 value in 2^32)."""
    seed = 1234 + rows * 17 + cols
    n = rows * cols
    J = ora.ref_grid_coupling(rows, cols, Jc, periodic)
    Jb = ora.ref_bit_coupling(J)
    hb = ora.ref_bit_bias(J, np.ones(n) * h, mode)
    table = ora.ising2d_thresholds(Jc, h, T, mode)
    spins = ora.ising2d_randomize(rows, cols, seed)
    bits = ((spins.reshape(-1).astype(np.int64) + 1) // 2)
    rr, cc = np.divmod(np.arange(n), cols)
    n_sweeps = 3
    for t in range(n_sweeps):
        for colour in (0, 1):
            order = np.nonzero(((rr + cc) & 1) == colour)[0]
            u = ora.ising2d_site_uniforms(rows, cols, 2 * t + colour, seed).reshape(-1)[order] / 4294967296.0
            # one partial "sweep" over this colour's sites only
            st = bits.copy()
            for k, i in enumerate(order):
                hloc = np.dot(Jb[i, :], st) + hb[i]
                st[i] = 1 if u[k] < ora.ref_sigmoid(float(hloc) / T) else 0
            bits = st
    twin = ora.ising2d_sweep(spins, periodic, table, n_sweeps, seed)
    np.testing.assert_array_equal(twin.reshape(-1), (2 * bits - 1).astype(np.int8))


@pytest.mark.parametrize("rows,cols,periodic", [(16, 32, True), (33, 47, False), (64, 64, True)])
def test_lazy_uniform_equals_plain(rows, cols, periodic):
    seed = 99
    spins = ora.ising2d_randomize(rows, cols, seed)
    # thresholds whose top 16 bits are hit often: force ties by using coarse tables
    for table in (ora.ising2d_thresholds(1.0, 0.0, 2.269185, 0),
                  np.array([(k * 0x0A3D) << 16 | 0x8000 for k in range(25)], dtype=np.uint64),
                  np.array([0, 1 << 32, 1, (1 << 32) - 1, 65536] * 5, dtype=np.uint64)):
        a = ora.ising2d_sweep(spins, periodic, table, 4, seed, sweep0=7)
        b = ora.ising2d_sweep(spins, periodic, table, 4, seed, sweep0=7, plain=True)
        np.testing.assert_array_equal(a, b)


def test_thresholds_c_vs_numpy():
    for J, h, T, mode in [(1.0, 0.0, 2.269185, 0), (1.0, 0.0, 2.5, 1), (-0.7, 0.3, 1.1, 0), (0.4, -0.2, 0.05, 1),
                          (1.0, 0.0, 0.01, 0), (2.0, 1.0, 100.0, 1)]:
        a = ora.ising2d_thresholds(J, h, T, mode).astype(np.int64)
        b = ora.ising2d_thresholds_numpy(J, h, T, mode).astype(np.int64)
        assert np.abs(a - b).max() <= 1  # libm exp vs NumPy exp: at most one unit of 2^-32
    t = ora.ising2d_thresholds(1.0, 0.0, 0.01, 0)
    assert t[4 * 5 + 4] == 1 << 32 and t[4 * 5 + 0] == 0  # sigmoid clamp -> exact 1.0 / 0.0


def test_sweep_is_composable_and_keyed_by_sweep_counter():
    """k sweeps in one call == k calls of one sweep with advancing sweep0 (counter-based RNG)."""
    table = ora.ising2d_thresholds(1.0, 0.1, 2.0, 0)
    s0 = ora.ising2d_randomize(12, 20, 5)
    a = ora.ising2d_sweep(s0, True, table, 5, 5, sweep0=3)
    b = s0
    for t in range(5):
        b = ora.ising2d_sweep(b, True, table, 1, 5, sweep0=3 + t)
    np.testing.assert_array_equal(a, b)
    assert (ora.ising2d_sweep(s0, True, table, 1, 5, sweep0=4) != ora.ising2d_sweep(s0, True, table, 1, 5, sweep0=3)).any()


def test_randomize_is_balanced_and_slab_consistent():
    s = ora.ising2d_randomize(64, 300, 42)
    assert set(np.unique(s)) == {-1, 1}
    assert abs(s.mean()) < 0.05
    np.testing.assert_array_equal(ora.ising2d_randomize(16, 300, 42, row0=32), s[32:48])


def test_observables():
    rng = np.random.default_rng(0)
    s = rng.choice([-1, 1], size=(6, 10)).astype(np.int8)
    for per in (False, True):
        J = ora.ref_grid_coupling(6, 10, 1.0, per)
        v = s.reshape(-1).astype(np.float64)
        ss, sb = ora.ising2d_observables(s, per)
        assert ss == v.sum() and sb == 0.5 * v.dot(J).dot(v)


def test_dense_philox_twin_matches_replay_with_same_uniforms():
    rng = np.random.default_rng(1)
    n = 24
    J = rng.normal(size=(n, n)) / 4
    b = rng.normal(size=n)
    st = rng.integers(0, 2, size=n)
    seed, T = 77, 0.9
    u = np.array([[ora.dense_uniform(i, t, seed) for i in range(n)] for t in range(5, 9)])
    assert ((u >= 0) & (u < 1)).all()
    a = ora.c_dense_sweep_replay(st, J, b, T, u)
    c = ora.dense_sweep_philox(st, J, b, T, 4, seed, sweep0=5)
    np.testing.assert_array_equal(a, c.astype(np.int64))
    # permuted order: uniforms are keyed by SITE, not by visiting position
    order = np.array([rng.permutation(n) for _ in range(4)])
    up = np.array([[ora.dense_uniform(i, 5 + t, seed) for i in order[t]] for t in range(4)])
    a = ora.c_dense_sweep_replay(st, J, b, T, up, order)
    c = ora.dense_sweep_philox(st, J, b, T, 4, seed, sweep0=5, order=order)
    np.testing.assert_array_equal(a, c.astype(np.int64))


def test_langevin_f32_twin_statistics_and_f64_agreement():
    """fp32 Philox/Box-Muller twin: normals are N(0,1); one step equals the f64 reference formula."""
    z = np.array([ora.langevin_normals_f32(q, 0, s, 9) for q in range(2048) for s in range(4)]).reshape(-1)
    assert abs(z.mean()) < 0.02 and abs(z.std() - 1.0) < 0.02
    assert abs(((z ** 4).mean()) - 3.0) < 0.2
    dim, T, dt, gamma = 64, 0.7, 0.02, 1.5
    rng = np.random.default_rng(2)
    x = rng.normal(size=(3, dim)).astype(np.float32)
    k = rng.uniform(0.5, 2.0, size=dim).astype(np.float32)
    mu = rng.normal(size=dim).astype(np.float32)
    x1 = ora.langevin_quadratic_f32(x, k, mu, 1, dt, gamma, T, 9, step0=4, chain0=10)
    noise = np.array([[ora.langevin_normals_f32(i >> 2, 10 + ch, 4, 9)[i & 3] for i in range(dim)] for ch in range(3)])
    want = np.array([ora.ref_langevin_step(x[ch].astype(np.float64), (k * (x[ch] - mu)).astype(np.float64),
                                           noise[ch].astype(np.float64), T, dt, gamma) for ch in range(3)])
    np.testing.assert_allclose(x1, want, rtol=0, atol=2e-6)
    # stationary variance of the discretised OU process: T / (k (1 - k dt / (2 gamma)))
    xs = ora.langevin_quadratic_f32(np.zeros((64, 64), np.float32), 2.0, 0.0, 600, 0.01, 1.0, 1.0, 3)
    assert abs(xs.var() - 1.0 / (2.0 * (1 - 0.01))) < 0.03
