"""Sampling benchmark cases of the reference, timed on the HIP sampler.

Interface and record layout follow /root/reference/tsu/benchmarks/sampling.py:21-278 (``SamplingResult`` fields, the keys of
``summary()``, ``SamplingBenchmark(config, seed)`` with ``benchmark_gaussian / benchmark_boltzmann / benchmark_multimodal /
run_all_benchmarks``), so that the reference's published table (its README: samples/s for Uniform_Binary(dim=1),
Boltzmann(n=10), Ferromagnetic_Bimodal) can be regenerated on this backend.  What every field of a trial means for each
case is the reference's definition, restated per case below; the arithmetic is written for arrays (all trials' samples
are reduced with vector operations) instead of per-sample Python loops.

The three cases are one table (``_CASES``): couplings, and how a trial's samples turn into the four quality numbers.
``rng="numpy"`` replays ``np.random`` in the reference's order, which makes every quality number equal to the
reference's own (tests/golden/g11); the default ``rng="philox"`` draws on the device.
"""
import time
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np

from ..gibbs import GibbsConfig, GibbsSampler

_METRICS = ("ks_statistics", "ks_pvalues", "kl_divergences", "effective_sample_sizes", "sampling_times", "samples_per_second")


def _mean_std_median(x, scale=1.0) -> Dict[str, float]:
    a = np.asarray(x, dtype=float) * scale
    return {"mean": np.mean(a), "std": np.std(a), "median": np.median(a)}


@dataclass
class SamplingResult:
    """Per-trial numbers of one case (reference: benchmarks/sampling.py:21-75, same field names and summary keys)."""

    distribution_name: str
    n_samples: int
    n_trials: int
    ks_statistics: List[float] = field(default_factory=list)
    ks_pvalues: List[float] = field(default_factory=list)
    kl_divergences: List[float] = field(default_factory=list)
    effective_sample_sizes: List[float] = field(default_factory=list)
    sampling_times: List[float] = field(default_factory=list)
    samples_per_second: List[float] = field(default_factory=list)

    def summary(self) -> Dict:
        out = {"distribution": self.distribution_name, "n_samples": self.n_samples, "n_trials": self.n_trials,
               "ks_statistic": _mean_std_median(self.ks_statistics)}
        p = np.asarray(self.ks_pvalues, dtype=float)
        out["ks_pvalue"] = {"mean": np.mean(p), "std": np.std(p), "fraction_passed": np.mean(p > 0.05)}
        out["kl_divergence"] = _mean_std_median(self.kl_divergences)
        out["effective_sample_size"] = _mean_std_median(self.effective_sample_sizes)
        out["sampling_time_ms"] = _mean_std_median(self.sampling_times, 1000.0)
        out["throughput_samples_per_sec"] = _mean_std_median(self.samples_per_second)
        return out


def effective_sample_size(x: np.ndarray, max_lag: int = 100) -> float:
    """N / (1 + 2 sum_k rho_k), lags 1.. until |rho_k| < 0.05, NaN or max_lag (reference: sampling.py:311-350; rho_k is the
    Pearson correlation of the two shifted windows, as ``np.corrcoef`` gives it)."""
    x = np.asarray(x, dtype=float)
    n = x.size
    if np.var(x) < 1e-10:
        return float(n)
    c = x - x.mean()
    total = 0.0
    for lag in range(1, min(max_lag, n // 2)):
        a, b = c[:-lag], c[lag:]
        a, b = a - a.mean(), b - b.mean()
        den = np.sqrt(a.dot(a) * b.dot(b))
        rho = a.dot(b) / den if den > 0 else np.nan
        if np.isnan(rho) or abs(rho) < 0.05:
            break
        total += rho
    return float(max(1.0, n / (1.0 + 2.0 * total)))


# ---------------------------------------------------------------------------------------------------------------------
# the cases: name -> (couplings, quality numbers of one trial's samples)
def _chain(n: int) -> np.ndarray:
    J = np.zeros((n, n))
    i = np.arange(n - 1)
    J[i, i + 1] = J[i + 1, i] = 1.0
    return J


def _quality_uniform(samples: np.ndarray, J: np.ndarray, h: np.ndarray) -> Tuple[float, float, float, float]:
    """Independent fair bits (sampling.py:132-157): z-score of the first bit's mean as 'ks statistic', pass = within two
    standard errors, 'kl' = distance of the empirical entropy from one bit, ESS of the first bit."""
    first = samples[:, 0]
    n = first.size
    p_hat = np.mean(first)
    z = abs(p_hat - 0.5) / np.sqrt(0.25 / n)
    entropy = -(p_hat * np.log2(p_hat) + (1 - p_hat) * np.log2(1 - p_hat)) if 0 < p_hat < 1 else 0.0
    return z, (1.0 if z < 2.0 else 0.0), float(abs(1.0 - entropy)), effective_sample_size(first)


def _quality_chain(samples: np.ndarray, J: np.ndarray, h: np.ndarray) -> Tuple[float, float, float, float]:
    """Ferromagnetic chain (sampling.py:195-218): energies E = -x'Jx - h'x of the bit samples; 'ks statistic' = share of
    rising steps in the 20-bin energy histogram, pass = |mean bit| > 0.1, 'kl' = |mean energy|, ESS of the energies."""
    x = samples.astype(float)
    energies = -np.einsum("si,ij,sj->s", x, J, x) - x @ h
    hist = np.histogram(energies, bins=20)[0]
    rising = np.sum(np.diff(hist) > 0) / len(hist)
    mean_bit = np.mean(np.sum(samples, axis=1)) / samples.shape[1]
    return rising, (1.0 if abs(mean_bit) > 0.1 else 0.0), float(abs(np.mean(energies))), effective_sample_size(energies)


def _quality_bimodal(samples: np.ndarray, J: np.ndarray, h: np.ndarray) -> Tuple[float, float, float, float]:
    """All-to-all ferromagnet (sampling.py:252-268): magnetisation per sample in spin language; 'ks statistic' = twice the
    smaller of the shares beyond +-0.5 (mode balance), pass above 0.3, 'kl' = its standard deviation, ESS of it."""
    mag = np.mean(2 * samples - 1, axis=1)
    balance = 2.0 * min(np.mean(mag > 0.5), np.mean(mag < -0.5))
    return balance, (1.0 if balance > 0.3 else 0.0), float(np.std(mag)), effective_sample_size(mag)


_Case = Tuple[str, Callable[[int], np.ndarray], Callable]
_CASES: Dict[str, _Case] = {
    "uniform": ("Uniform_Binary(dim={n})", lambda n: np.zeros((n, n)), _quality_uniform),
    "chain": ("Boltzmann(n={n})", _chain, _quality_chain),
    "bimodal": ("Ferromagnetic_Bimodal", lambda n: np.ones((n, n)) - np.eye(n), _quality_bimodal),
}


class SamplingBenchmark:
    """Reference: benchmarks/sampling.py:78-96 (default config T = 1, 100 burn-in sweeps, 10 sweeps per sample; trial t is
    seeded with ``np.random.seed(seed + t)``).  ``rng`` / ``coupling_dtype`` are passed on to :class:`GibbsSampler`."""

    def __init__(self, config: Optional[GibbsConfig] = None, seed: int = 42, *, rng: str = "philox", coupling_dtype: str = "float64"):
        self.config = config or GibbsConfig(temperature=1.0, n_burnin=100, n_sweeps=10)
        self.seed = seed
        self.sampler = GibbsSampler(self.config, rng=rng, coupling_dtype=coupling_dtype)

    def _run(self, case: str, n: int, n_samples: int, n_trials: int) -> SamplingResult:
        title, couplings, quality = _CASES[case]
        result = SamplingResult(distribution_name=title.format(n=n), n_samples=n_samples, n_trials=n_trials)
        for trial in range(n_trials):
            np.random.seed(self.seed + trial)
            J, h = couplings(n), np.zeros(n)
            t0 = time.time()
            samples = self.sampler.sample_boltzmann(J, bias=h, n_samples=n_samples)
            elapsed = time.time() - t0
            numbers = quality(samples, J, h) + (elapsed, n_samples / elapsed)
            for name, value in zip(_METRICS, numbers):
                getattr(result, name).append(float(value))
        return result

    def benchmark_gaussian(self, n_samples: int = 10000, n_trials: int = 5, dim: int = 1) -> SamplingResult:
        """Reference: sampling.py:98-162 (its name; the case is independent fair bits)."""
        return self._run("uniform", dim, n_samples, n_trials)

    def benchmark_boltzmann(self, n_spins: int = 10, n_samples: int = 10000, n_trials: int = 5) -> SamplingResult:
        """Reference: sampling.py:164-220."""
        return self._run("chain", n_spins, n_samples, n_trials)

    def benchmark_multimodal(self, n_samples: int = 10000, n_trials: int = 5) -> SamplingResult:
        """Reference: sampling.py:222-274 (10 spins)."""
        return self._run("bimodal", 10, n_samples, n_trials)

    def _compute_ess(self, samples: np.ndarray, max_lag: int = 100) -> float:
        return effective_sample_size(samples, max_lag)

    def run_all_benchmarks(self, quick: bool = False, verbose: bool = True) -> Dict[str, SamplingResult]:
        """Reference: sampling.py:376-428 -- the three cases under the keys the runner expects."""
        n_samples, n_trials = (1000, 3) if quick else (10000, 5)
        plan = (("gaussian_1d", lambda: self.benchmark_gaussian(n_samples, n_trials, dim=1)),
                ("boltzmann", lambda: self.benchmark_boltzmann(10, n_samples, n_trials)),
                ("multimodal", lambda: self.benchmark_multimodal(n_samples, n_trials)))
        results = {}
        for key, run in plan:
            results[key] = r = run()
            if verbose:
                print(f"{r.distribution_name:26s} KL {np.mean(r.kl_divergences):9.4f}   ESS {np.mean(r.effective_sample_sizes):8.0f}   "
                      f"{np.mean(r.samples_per_second):10.0f} samples/s")
        return results
