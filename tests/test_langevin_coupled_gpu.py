"""K3, coupled quadratic energies E = 1/2 x^T A x + b^T x (csrc/langevin.hip k3_coupled) against the oracle's twin
(oracle/tsu_oracle.c ora_langevin_coupled_f32: same Philox normals, gradient A x + b accumulated in double and rounded to float once).
The device accumulates the dot products in float in its own (fixed) order, so the comparison carries a tolerance: 2e-4 (1 + |x|)
absolute after tens of steps on well-conditioned matrices -- the same figure as the separable kernel's tests."""
import numpy as np
import pytest

from oracle import oracle as ora

pytestmark = pytest.mark.gpu


def _spd(d, seed, cond=20.0):
    rng = np.random.default_rng(seed)
    Q, _ = np.linalg.qr(rng.standard_normal((d, d)))
    ev = np.linspace(1.0, cond, d)
    A = (Q * ev) @ Q.T
    A = 0.5 * (A + A.T)
    A32 = A.astype(np.float32)
    A32 = np.triu(A32) + np.triu(A32, 1).T  # exactly symmetric in float32
    return A32


@pytest.mark.parametrize("d,chains", [(5, 1), (64, 3), (100, 8), (257, 20), (1024, 2), (1500, 9), (4000, 9)])  # (4000 x 9: two column chunks)
def test_coupled_steps_match_the_twin(d, chains):
    from tsu import _hip
    A = _spd(d, d)
    b = np.random.default_rng(d + 1).standard_normal(d).astype(np.float32)
    x0 = np.random.default_rng(d + 2).standard_normal((chains, d)).astype(np.float32)
    lc = _hip.LangevinChains(chains, d)
    lc.set_coupling(A, b)
    lc.set_state(x0)
    traj = lc.step(12, 0.01, 1.0, 0.7, 99, step0=5, chain0=3, trajectory=True)
    got = lc.get_state()
    want, wtraj = ora.langevin_coupled_f32(x0, A, b, 12, 0.01, 1.0, 0.7, 99, step0=5, chain0=3, trajectory=True)
    tol = 2e-4 * (1.0 + np.abs(want))
    assert np.all(np.abs(got - want) <= tol), float(np.max(np.abs(got - want)))
    assert np.all(np.abs(traj - wtraj) <= 2e-4 * (1.0 + np.abs(wtraj)))
    # a second call continues the chains (the two buffers of a step have changed places an even or odd number of times)
    lc.step(5, 0.01, 1.0, 0.7, 99, step0=17, chain0=3)
    want2 = ora.langevin_coupled_f32(want, A, b, 5, 0.01, 1.0, 0.7, 99, step0=17, chain0=3)
    got2 = lc.get_state()
    assert np.all(np.abs(got2 - want2) <= 3e-4 * (1.0 + np.abs(want2)))
    lc.close()


def test_a_diagonal_coupling_is_the_separable_kernel():
    """A = diag(k), b = -k mu: the coupled kernel walks the separable kernel's trajectory (same noise, same update expression; the
    gradient k x - k mu against k (x - mu): a few ulp)."""
    from tsu import _hip
    d, chains = 200, 4
    k = np.linspace(0.5, 3.0, d).astype(np.float32)
    mu = np.linspace(-1.0, 1.0, d).astype(np.float32)
    x0 = np.zeros((chains, d), np.float32)
    a = _hip.LangevinChains(chains, d)
    a.set_energy(k, mu)
    a.set_state(x0)
    a.step(30, 0.01, 1.0, 1.0, 5)
    c = _hip.LangevinChains(chains, d)
    c.set_coupling(np.diag(k), -k * mu)
    c.set_state(x0)
    c.step(30, 0.01, 1.0, 1.0, 5)
    np.testing.assert_allclose(c.get_state(), a.get_state(), atol=2e-5)
    a.close()
    c.close()


def test_stationary_covariance_of_a_correlated_pair():
    """The chain samples N(-A^-1 b, T A^-1) up to the discretisation: covariance T (A (1 - dt A / 2))^-1 for the Euler scheme."""
    from tsu import _hip
    A = np.array([[2.0, 1.2], [1.2, 1.5]], np.float32)
    b = np.array([0.5, -1.0], np.float32)
    T, dt, chains = 0.8, 0.02, 1 << 15
    lc = _hip.LangevinChains(chains, 2)
    lc.set_coupling(A, b)
    lc.set_state(np.zeros((chains, 2), np.float32))
    lc.step(1500, dt, 1.0, T, 11)
    x = lc.get_state().astype(np.float64)
    lc.close()
    A64 = A.astype(np.float64)
    mean = -np.linalg.solve(A64, b.astype(np.float64))
    cov = T * np.linalg.inv(A64 @ (np.eye(2) - 0.5 * dt * A64))
    se = np.sqrt(np.diag(cov) / chains)
    assert np.all(np.abs(x.mean(0) - mean) <= 5 * se)
    got = np.cov(x.T)
    assert np.all(np.abs(got - cov) <= 0.03 * np.abs(cov).max())


def test_asymmetric_matrices_are_refused():
    from tsu import _hip
    lc = _hip.LangevinChains(1, 3)
    A = np.eye(3, dtype=np.float32)
    A[0, 1] = 0.5
    with pytest.raises(ValueError, match="symmetric"):
        lc.set_coupling(A)
    lc.close()


def test_sample_from_energy_runs_coupled_quadratics_on_the_device(monkeypatch):
    """`sample_from_energy` with a QuadraticForm descriptor at d = 512, and with a plain Python callable of a coupled quadratic at
    d = 6 (recognised by probing, tsu/core.py `_recognise_coupled`): both reach k3_coupled (no finite differences on the host:
    the host gradient is made to fail), with the reference's restart rule and shapes (core.py:100-162)."""
    from tsu import core
    from tsu.core import ThermalSamplingUnit, TSUConfig, QuadraticForm
    monkeypatch.setattr(ThermalSamplingUnit, "_numerical_gradient", lambda *a, **k: (_ for _ in ()).throw(AssertionError("host path")))
    d = 512
    A = _spd(d, 7, cond=5.0).astype(np.float64)
    tsu = ThermalSamplingUnit(TSUConfig(temperature=1.0, dt=0.01, n_burnin=50, n_steps=20), seed=3)
    s = tsu.sample_from_energy(QuadraticForm(A, 0.1), np.zeros(d), n_samples=64)
    assert s.shape == (64, d) and s.dtype == np.float64 and np.all(np.isfinite(s))
    assert tsu.sample_count == 64
    M = np.array([[2.0, 0.5, 0, 0, 0, 0.25], [0.5, 1.5, 0.5, 0, 0, 0], [0, 0.5, 1.0, 0.25, 0, 0], [0, 0, 0.25, 2.0, 0.5, 0],
                  [0, 0, 0, 0.5, 1.0, 0.25], [0.25, 0, 0, 0, 0.25, 1.5]])
    v = np.array([0.5, 0, -0.25, 0, 1.0, 0])
    energy = lambda x: float(0.5 * x @ M @ x + v @ x)  # noqa: E731
    q = core._recognise_quadratic(energy, np.zeros(6))
    assert isinstance(q, QuadraticForm)
    np.testing.assert_array_equal(q.A, M)
    np.testing.assert_array_equal(q.b, v)
    tsu2 = ThermalSamplingUnit(TSUConfig(temperature=0.5, dt=0.02, n_burnin=400, n_steps=10), seed=4)
    s2, traj = tsu2.sample_from_energy(energy, np.zeros(6), n_samples=4000, return_trajectory=True)
    assert s2.shape == (4000, 6) and len(traj) == 4000 * 10
    mean = -np.linalg.solve(M, v)
    cov = 0.5 * np.linalg.inv(M @ (np.eye(6) - 0.01 * M))
    assert np.all(np.abs(s2.mean(0) - mean) <= 5 * np.sqrt(np.diag(cov) / 4000))
    assert np.all(np.abs(np.cov(s2.T) - cov) <= 0.1 * np.abs(cov).max())
