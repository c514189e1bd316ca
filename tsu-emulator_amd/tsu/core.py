"""Thermodynamic Sampling Unit (Langevin dynamics) on the MI355X -- drop-in for the reference's ``tsu.core``.

Same classes, signatures and exceptions as the reference (file:line cited per symbol).  The overdamped
Langevin update  x <- x - grad E(x) dt/gamma + sqrt(2 T dt/gamma) xi  (core.py:74-80) runs in the fused HIP
kernel K3 of ``libtsu_hip.so`` whenever the energy is a *separable quadratic*
E(x) = sum_i 1/2 k_i (x_i - mu_i)^2 + c  -- either given as a :class:`QuadraticEnergy` descriptor or recognised
by probing the callable (the reference's README example ``(x**2).sum()`` and ``sample_gaussian`` are of this
form).  The gradient is then analytic and fused into the kernel; the reference's finite-difference gradient
(core.py:82-98) needs 2 d Python calls per step and cannot run on a device.

An arbitrary Python callable is, by construction, host code: for energies that are not separable quadratics
(``p_bit``'s clipped-linear and ``sample_categorical``'s piecewise-constant 1-D toys, core.py:186-191,250-255)
``sample_from_energy`` evaluates the reference algorithm as written -- float64, central differences,
``np.random.randn`` -- because there is nothing a GPU kernel could be given.  That path is not a fallback of
the kernel (it serves a different class of inputs) and is never taken for a quadratic energy.
"""
from dataclasses import dataclass
from typing import Callable, Optional

import os

import numpy as np

from . import _hip


class TSUError(Exception):
    """Base exception for TSU platform (reference: core.py:12-15)."""


class ConfigurationError(TSUError):
    """Invalid configuration parameters (reference: core.py:18-21)."""


class SamplingError(TSUError):
    """Error during sampling process (reference: core.py:24-27)."""


@dataclass
class TSUConfig:
    """Reference: tsu/core.py:30-51 (same fields, defaults, validation and messages)."""

    temperature: float = 1.0
    dt: float = 0.01
    friction: float = 1.0
    n_burnin: int = 100
    n_steps: int = 500

    def __post_init__(self):
        if self.temperature <= 0:
            raise ConfigurationError(f"Temperature must be positive, got {self.temperature}")
        if self.dt <= 0 or self.dt > 0.1:
            raise ConfigurationError(f"Time step dt must be in (0, 0.1], got {self.dt}")
        if self.friction <= 0:
            raise ConfigurationError(f"Friction must be positive, got {self.friction}")
        if self.n_burnin < 0:
            raise ConfigurationError(f"Burn-in steps must be non-negative, got {self.n_burnin}")
        if self.n_steps <= 0:
            raise ConfigurationError(f"Number of steps must be positive, got {self.n_steps}")


class QuadraticEnergy:
    """Energy descriptor E(x) = sum_i 1/2 k_i (x_i - mu_i)^2 + c, callable like any reference energy_fn.

    ``k`` and ``mu`` broadcast against the state.  ``QuadraticEnergy(2.0)`` is the README's ``(x**2).sum()``.
    """

    def __init__(self, k=1.0, mu=0.0, c: float = 0.0):
        self.k = np.asarray(k, dtype=np.float64)
        self.mu = np.asarray(mu, dtype=np.float64)
        self.c = float(c)
        if np.any(self.k < 0):
            raise ConfigurationError("QuadraticEnergy needs k >= 0")

    def __call__(self, x) -> float:
        x = np.asarray(x, dtype=np.float64)
        return float(np.sum(0.5 * self.k * (x - self.mu) ** 2) + self.c)

    def gradient(self, x) -> np.ndarray:
        return self.k * (np.asarray(x, dtype=np.float64) - self.mu)


class QuadraticForm:
    """Energy descriptor E(x) = 1/2 x^T A x + b^T x + c with a (dim, dim) matrix, callable like any reference energy_fn -- the
    shape of the reference's multivariate callers (tsu/api.py:94).  Only the symmetric part of ``A`` enters the energy; it is
    what the device keeps (gradient A_sym x + b: `csrc/langevin.hip` k3_coupled)."""

    def __init__(self, A, b=0.0, c: float = 0.0):
        A = np.asarray(A, dtype=np.float64)
        if A.ndim != 2 or A.shape[0] != A.shape[1]:
            raise ConfigurationError(f"QuadraticForm needs a square matrix, got shape {A.shape}")
        self.A = 0.5 * (A + A.T)
        self.b = np.broadcast_to(np.asarray(b, dtype=np.float64), (A.shape[0],)).copy()
        self.c = float(c)

    def __call__(self, x) -> float:
        x = np.atleast_1d(np.asarray(x, dtype=np.float64))
        return float(0.5 * x @ self.A @ x + self.b @ x + self.c)

    def gradient(self, x) -> np.ndarray:
        return self.A @ np.atleast_1d(np.asarray(x, dtype=np.float64)) + self.b


_PROBE_FULL_MAX = 4096  # up to this dimension every coordinate is probed (2 d + 1 evaluations of O(d) work each)


def _verify_quadratic(energy_fn: Callable, cand: QuadraticEnergy, x0: np.ndarray) -> bool:
    """The candidate must reproduce the callable on random points (every coordinate moves in each of them)."""
    rng = np.random.RandomState(12345)  # private stream: probing must not disturb the caller's np.random
    scale = 1.0 + np.abs(x0)
    for _ in range(6):
        xt = x0 + rng.normal(size=x0.size) * 3.0 * scale
        want, got = float(energy_fn(xt.copy())), cand(xt)
        if not abs(want - got) <= 1e-9 * max(1.0, abs(want)):
            return False
    return True


_PROBE_COUPLED_MAX = 64  # up to this dimension a callable is also probed for a COUPLED quadratic (d (d + 1) / 2 + 2 d + 1 evaluations)


def _recognise_coupled(energy_fn: Callable, x0: np.ndarray) -> Optional[QuadraticForm]:
    """Fit E = 1/2 x^T A x + b^T x + c by second differences around x0 and verify it on random points (as `_verify_quadratic`)."""
    d = x0.size
    if d < 2 or d > _PROBE_COUPLED_MAX:
        return None
    try:
        step = 0.5
        e0 = float(energy_fn(x0.copy()))
        ep, em = np.empty(d), np.empty(d)
        for i in range(d):
            xp, xm = x0.copy(), x0.copy()
            xp[i] += step
            xm[i] -= step
            ep[i], em[i] = float(energy_fn(xp)), float(energy_fn(xm))
        A = np.zeros((d, d))
        A[np.arange(d), np.arange(d)] = (ep - 2 * e0 + em) / step ** 2
        for i in range(d):
            for j in range(i):
                xpp = x0.copy()
                xpp[i] += step
                xpp[j] += step
                A[i, j] = A[j, i] = (float(energy_fn(xpp)) - ep[i] - ep[j] + e0) / step ** 2
        g0 = (ep - em) / (2 * step)  # gradient at x0 = A x0 + b
        if not (np.all(np.isfinite(A)) and np.all(np.isfinite(g0))):
            return None
        # (couplings people write are short binary fractions: take the exact ones if they fit the rounding of the differences)
        tol = 1e-9 * max(1.0, abs(e0)) / step ** 2
        Ar = np.round(A * 1024) / 1024
        if np.max(np.abs(Ar - A)) <= tol:
            A = Ar
        b = g0 - A @ x0
        br = np.round(b * 1024) / 1024
        if np.max(np.abs(br - b)) <= tol * max(1.0, float(np.max(np.abs(x0))) + 1.0):
            b = br
        c = e0 - float(0.5 * x0 @ A @ x0 + b @ x0)
        cand = QuadraticForm(A, b, c)
        # bounded below (a Langevin chain on an indefinite form runs away; the reference's loop would too, but slowly enough to return)
        if np.min(np.linalg.eigvalsh(cand.A)) <= 0:
            return None
        return cand if _verify_quadratic(energy_fn, cand, x0) else None
    except Exception:
        return None


def _recognise_quadratic(energy_fn: Callable, x_init: np.ndarray) -> Optional[QuadraticEnergy]:
    """Probe a callable, fit a separable quadratic and verify it on random points.  Returns None unless the fit is exact to
    ~1e-9 relative (so non-quadratic energies never take the GPU path).

    d <= 4096: every coordinate is probed (2 d + 1 evaluations).  Larger d (the README idiom ``lambda x: (x**2).sum()`` at
    dim = 2**20): a bounded probe -- 64 random coordinates; if they agree on ONE stiffness and ONE centre, the uniform quadratic
    is verified on random points, in which all d coordinates move, so a single deviating coordinate fails the check.  About 140
    evaluations whatever d is.  Non-uniform separable quadratics of that size need a ``QuadraticEnergy`` descriptor."""
    if isinstance(energy_fn, (QuadraticEnergy, QuadraticForm)):
        return energy_fn
    x0 = np.atleast_1d(np.asarray(x_init, dtype=np.float64))
    d = x0.size
    try:
        e0 = float(energy_fn(x0.copy()))
        step = 0.5

        def probe(i):
            xp, xm = x0.copy(), x0.copy()
            xp[i] += step
            xm[i] -= step
            ep, em = float(energy_fn(xp)), float(energy_fn(xm))
            return (ep - 2 * e0 + em) / step ** 2, (ep - em) / (2 * step)

        if d <= _PROBE_FULL_MAX:
            kg = np.array([probe(i) for i in range(d)])
            k, g = kg[:, 0], kg[:, 1]
            if not np.all(np.isfinite(k)) or np.any(k <= 0):
                return _recognise_coupled(energy_fn, x0)
            mu = x0 - g / k
        else:
            idx = np.random.RandomState(54321).choice(d, size=64, replace=False)
            kg = np.array([probe(int(i)) for i in idx])
            ks, gs = kg[:, 0], kg[:, 1]
            if not np.all(np.isfinite(ks)) or np.any(ks <= 0):
                return None
            mus = x0[idx] - gs / ks
            # (second differences of a sum of d terms carry rounding of the order eps * |E| / step^2: hence the loose agreement
            # test here; the verification below is what admits the candidate)
            tol = 1e-6 * max(1.0, abs(e0)) / step ** 2
            if np.ptp(ks) > tol or np.ptp(mus) > tol * max(1.0, 1.0 / float(np.min(ks))):
                return None
            kbar = float(np.median(ks))
            kr = round(kbar * 1024) / 1024  # stiffnesses people write are short binary fractions: take the exact one if it fits
            if abs(kr - kbar) <= tol and kr > 0:
                kbar = kr
            mbar = float(np.median(mus))
            mr = round(mbar * 1024) / 1024
            if abs(mr - mbar) <= tol * max(1.0, 1.0 / kbar):
                mbar = mr
            k, mu = np.float64(kbar), np.float64(mbar)
        kf, mf = np.broadcast_to(k, (d,)), np.broadcast_to(mu, (d,))
        c = e0 - float(np.sum(0.5 * kf * (x0 - mf) ** 2))
        cand = QuadraticEnergy(k, mu, c)
        if _verify_quadratic(energy_fn, cand, x0):
            return cand
    except Exception:
        pass
    return _recognise_coupled(energy_fn, x0)


class ThermalSamplingUnit:
    """Reference: tsu/core.py:54-267.  Keyword-only extra ``seed``: Philox seed of the device noise stream
    (default: drawn from ``np.random`` on first use, so ``np.random.seed`` makes runs reproducible)."""

    def __init__(self, config: Optional[TSUConfig] = None, *, seed: Optional[int] = None):
        self.config = config or TSUConfig()
        self.sample_count = 0
        self._seed = None if seed is None else int(seed)
        self._call_counter = 0

    def _philox_seed(self) -> int:
        if self._seed is None:
            self._seed = int(np.random.randint(0, 2 ** 31 - 1)) | (int(np.random.randint(0, 2 ** 31 - 1)) << 31)
        return self._seed

    # ------------------------------------------------------------------ scalar helpers (host, reference semantics)
    def _langevin_step(self, x: np.ndarray, grad_energy: np.ndarray) -> np.ndarray:
        """Reference: core.py:64-80 (float64, ``np.random.randn``): the single-step helper for a gradient the
        caller computed on the host.  The fused device step is :meth:`sample_from_energy`'s quadratic path."""
        cfg = self.config
        drift = -grad_energy * cfg.dt / cfg.friction
        noise_scale = np.sqrt(2 * cfg.temperature * cfg.dt / cfg.friction)
        diffusion = noise_scale * np.random.randn(*x.shape)
        return x + drift + diffusion

    def _numerical_gradient(self, energy_fn: Callable, x: np.ndarray, eps: float = 1e-5) -> np.ndarray:
        """Reference: core.py:82-98 (central differences)."""
        x = np.atleast_1d(x)
        grad = np.zeros_like(x)
        for i in range(len(x)):
            x_plus = x.copy()
            x_plus[i] += eps
            x_minus = x.copy()
            x_minus[i] -= eps
            grad[i] = (float(energy_fn(x_plus)) - float(energy_fn(x_minus))) / (2 * eps)
        return grad

    # ------------------------------------------------------------------ sampling
    def _sample_quadratic_device(self, q: QuadraticEnergy, x_init: np.ndarray, n_samples: int, return_trajectory: bool):
        """All ``n_samples`` restarts of core.py:140-159 as independent chains of one fused launch sequence."""
        cfg = self.config
        x0 = np.atleast_1d(np.asarray(x_init, dtype=np.float64))
        d = x0.size
        seed = self._philox_seed()
        chain0 = self._call_counter  # fresh Philox chain ids per call
        self._call_counter += n_samples
        lc = _hip.LangevinChains(n_samples, d)
        try:
            if isinstance(q, QuadraticForm):
                if q.A.shape[0] != d:
                    raise SamplingError(f"QuadraticForm of dimension {q.A.shape[0]} on a state of dimension {d}")
                lc.set_coupling(q.A.astype(np.float32), q.b.astype(np.float32))
            else:
                lc.set_energy(np.broadcast_to(q.k, (d,)).astype(np.float32), np.broadcast_to(q.mu, (d,)).astype(np.float32))
            # sample 0 starts exactly at x_init; samples s > 0 at x_init + 0.1 * N(0,1)  (core.py:142-143)
            lc.restart(x0.astype(np.float32), 0.1, seed, chain0)
            if n_samples >= 1:
                st = lc.get_state()
                st[0] = x0.astype(np.float32)
                lc.set_state(st)
            lc.step(int(cfg.n_burnin), cfg.dt, cfg.friction, cfg.temperature, seed, 0, chain0)
            traj = lc.step(int(cfg.n_steps), cfg.dt, cfg.friction, cfg.temperature, seed, int(cfg.n_burnin), chain0,
                           trajectory=return_trajectory)
            samples = lc.get_state().astype(np.float64)
        finally:
            lc.close()
        self.sample_count += n_samples
        if return_trajectory:
            # reference order: all sampling-phase states of sample 0, then of sample 1, ... (core.py:151-156)
            trajectory = [traj[s, c].astype(np.float64) for c in range(n_samples) for s in range(traj.shape[0])]
            return samples, trajectory
        return samples

    def sample_from_energy(self, energy_fn: Callable, x_init: np.ndarray, n_samples: int = 1,
                           return_trajectory: bool = False):
        """Reference: core.py:100-162.  Returns (n_samples, dim) float64 [, trajectory list]."""
        if n_samples <= 0:
            raise SamplingError(f"n_samples must be positive, got {n_samples}")
        try:
            test_energy = energy_fn(x_init)
            if not isinstance(test_energy, (int, float, np.number)):
                raise SamplingError(f"Energy function must return scalar, got {type(test_energy)}")
        except Exception as e:
            raise SamplingError(f"Energy function failed on initial state: {e}")

        q = _recognise_quadratic(energy_fn, np.atleast_1d(x_init))
        if q is not None:
            return self._sample_quadratic_device(q, x_init, n_samples, return_trajectory)

        # arbitrary Python energy: the reference algorithm as written (host by necessity, see module docstring): finite differences
        # with 2 d energy calls per step, i.e. O(d^2) work per step -- hours from a few thousand dimensions on.  Never started
        # silently there: the caller gets the descriptor to use instead (TSU_LANGEVIN_HOST=1 runs it anyway).
        d_ = int(np.atleast_1d(x_init).size)
        if d_ > _PROBE_FULL_MAX and os.environ.get("TSU_LANGEVIN_HOST", "0") != "1":
            steps = (self.config.n_burnin + self.config.n_steps) * n_samples
            raise SamplingError(
                f"energy_fn is not a uniform separable quadratic and dim = {d_}: the reference's finite-difference Langevin loop "
                f"would make {2 * d_ * steps:.3g} energy calls of O(dim) work on the host.  Pass a tsu.core.QuadraticEnergy(k, mu) "
                f"descriptor (k, mu scalars or arrays of length dim) to run on the GPU, or set TSU_LANGEVIN_HOST=1 to run the "
                f"host loop anyway.")
        cfg = self.config
        x_init = np.asarray(x_init, dtype=float)
        x = np.atleast_1d(x_init).copy()
        samples = []
        trajectory = [] if return_trajectory else None
        for sample_idx in range(n_samples):
            if sample_idx > 0:
                x = x_init + 0.1 * np.random.randn(*x_init.shape)
            for _ in range(cfg.n_burnin):
                x = self._langevin_step(x, self._numerical_gradient(energy_fn, x))
            for _ in range(cfg.n_steps):
                x = self._langevin_step(x, self._numerical_gradient(energy_fn, x))
                if return_trajectory:
                    trajectory.append(x.copy())
            samples.append(x.copy())
            self.sample_count += 1
        samples = np.array(samples)
        return (samples, trajectory) if return_trajectory else samples

    def sample_boltzmann(self, energy, n_samples: int = 1, dim: int = 1, x_init: Optional[np.ndarray] = None):
        """README name (README.md:63): ``tsu.sample_boltzmann(energy, n_samples=1000, dim=10)`` -> (n_samples, dim)."""
        if dim <= 0:
            raise SamplingError(f"dim must be positive, got {dim}")
        x0 = np.zeros(dim) if x_init is None else np.asarray(x_init, dtype=float)
        return self.sample_from_energy(energy, x0, n_samples)

    def p_bit(self, prob: float, n_samples: int = 1) -> np.ndarray:
        """Reference: core.py:164-203."""
        if not 0 <= prob <= 1:
            raise ConfigurationError(f"Probability must be in [0,1], got {prob}")
        if n_samples <= 0:
            raise ConfigurationError(f"n_samples must be positive, got {n_samples}")
        prob_clipped = float(np.clip(prob, 1e-10, 1 - 1e-10))

        def energy(x):
            x0 = float(np.atleast_1d(x)[0])
            x_clipped = float(np.clip(x0, 1e-10, 1 - 1e-10))
            return -np.log(prob_clipped) * x_clipped - np.log(1 - prob_clipped) * (1 - x_clipped)

        result = self.sample_from_energy(energy, np.array([prob_clipped]), n_samples)
        samples = result[0] if isinstance(result, tuple) else result
        return (samples.flatten() > 0.5).astype(int)

    def sample_gaussian(self, mu: float = 0.0, sigma: float = 1.0, n_samples: int = 1) -> np.ndarray:
        """Reference: core.py:205-240 -- E = 1/2 ((x - mu)/sigma)^2, i.e. k = 1/sigma^2: fused device chains."""
        if sigma <= 0:
            raise ConfigurationError(f"Sigma must be positive, got {sigma}")
        if n_samples <= 0:
            raise ConfigurationError(f"n_samples must be positive, got {n_samples}")
        result = self.sample_from_energy(QuadraticEnergy(1.0 / sigma ** 2, mu), np.array([float(mu)]), n_samples)
        samples = result[0] if isinstance(result, tuple) else result
        return samples.flatten()

    def sample_categorical(self, probs: np.ndarray, n_samples: int = 1) -> np.ndarray:
        """Reference: core.py:242-267."""
        probs = np.array(probs)
        probs = probs / probs.sum()

        def energy(x):
            x0 = int(abs(float(np.atleast_1d(x)[0])))
            return -np.log(probs[x0 % len(probs)] + 1e-10)

        result = self.sample_from_energy(energy, np.array([0.0]), n_samples)
        samples_cont = result[0] if isinstance(result, tuple) else result
        return np.abs(samples_cont.flatten()).astype(int) % len(probs)


class ProbabilisticNeuron:
    """Reference: core.py:270-294."""

    def __init__(self, tsu: ThermalSamplingUnit):
        self.tsu = tsu

    def activate(self, weights: np.ndarray, inputs: np.ndarray, bias: float = 0.0) -> int:
        logit = np.dot(weights, inputs) + bias
        prob = 1.0 / (1.0 + np.exp(-logit))
        return self.tsu.p_bit(prob, n_samples=1)[0]

    def forward_stochastic(self, weights: np.ndarray, inputs: np.ndarray, bias: float = 0.0, n_samples: int = 10) -> float:
        return float(np.mean([self.activate(weights, inputs, bias) for _ in range(n_samples)]))


def validate_distribution(samples: np.ndarray, expected_dist: str, params: dict, alpha: float = 0.05) -> dict:
    """Reference: core.py:298-327."""
    from scipy import stats

    results = {"mean": np.mean(samples), "std": np.std(samples), "n_samples": len(samples)}
    if expected_dist == "gaussian":
        mu, sigma = params.get("mu", 0), params.get("sigma", 1)
        results["expected_mean"] = mu
        results["expected_std"] = sigma
        ks_stat, p_value = stats.kstest(samples, "norm", args=(mu, sigma))
        results["ks_statistic"] = ks_stat
        results["ks_pvalue"] = p_value
        results["passes_ks_test"] = p_value > alpha
    elif expected_dist == "bernoulli":
        p = params.get("p", 0.5)
        results["expected_mean"] = p
        results["empirical_prob"] = np.mean(samples)
        results["error"] = abs(np.mean(samples) - p)
        results["passes_test"] = results["error"] < 0.05
    return results
