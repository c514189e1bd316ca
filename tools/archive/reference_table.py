"""The reference's own published sampling table (tsu/benchmarks/README.md:57-59, CHANGELOG.md:21-22) on this backend:
same workloads through the same public call, GibbsSampler.sample_boltzmann -- GibbsConfig(T=1, n_burnin=100,
n_sweeps=10) as in benchmarks/sampling.py:98, 10 000 samples, 5 trials.  The reference's numbers are CPU numbers it
published itself (unspecified CPU, one thread); they are quoted, not re-measured here."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tsu-emulator_amd"))
import numpy as np
from tsu.gibbs import GibbsSampler, GibbsConfig

PUBLISHED = {"Uniform_Binary(dim=1)": 42928, "Boltzmann_Chain(n=10)": 4377, "Ferromagnet(n=10)": 4400}


def chain(n, J=1.0):
    A = np.zeros((n, n))
    for i in range(n - 1):
        A[i, i + 1] = A[i + 1, i] = J
    return A


cases = {"Uniform_Binary(dim=1)": (np.zeros((1, 1)), np.zeros(1)),
         "Boltzmann_Chain(n=10)": (chain(10), np.zeros(10)),
         "Ferromagnet(n=10)": (np.ones((10, 10)) - np.eye(10), np.zeros(10))}
for rng in ("philox", "numpy"):
    for name, (J, h) in cases.items():
        rates = []
        for trial in range(5):
            np.random.seed(42 + trial)
            s = GibbsSampler(GibbsConfig(temperature=1.0, n_burnin=100, n_sweeps=10), rng=rng)
            t0 = time.time()
            x = s.sample_boltzmann(J, bias=h, n_samples=10000)
            rates.append(10000 / (time.time() - t0))
        print(f"rng={rng:6s} {name:24s} {np.median(rates):12.0f} samples/s   (reference publishes {PUBLISHED[name]}; x{np.median(rates) / PUBLISHED[name]:.1f})  mean bit {x.mean():.3f}", flush=True)

# MAX-CUT by simulated annealing, n=15, 1000 steps (tsu/benchmarks/README.md:62, README.md:273: 17.6 +- 2.1 ms)
rng0 = np.random.RandomState(7)
W = np.triu((rng0.rand(15, 15) < 0.5).astype(float), 1)
W = W + W.T
for rng in ("philox", "numpy"):
    ts = []
    for trial in range(7):
        np.random.seed(100 + trial)
        s = GibbsSampler(GibbsConfig(temperature=1.0), rng=rng)
        t0 = time.time()
        best, e = s.simulated_annealing(-2.0 * W, bias=W.sum(1), n_steps=1000)  # E = -(cut weight)
        ts.append(time.time() - t0)
    print(f"rng={rng:6s} simulated_annealing n=15, 1000 steps: {np.median(ts[1:]) * 1e3:8.2f} ms   (reference publishes 17.6 ms)  cut found {-e:.0f} of {W.sum() / 2:.0f} edges", flush=True)

# parallel tempering, n=10 chain, 4 temperatures, 500 samples (not in the reference's published table; before the replica
# loop became one device call this took ~3 device round trips per replica per sample)
for rng in ("philox", "numpy"):
    np.random.seed(3)
    s = GibbsSampler(GibbsConfig(temperature=1.0, n_burnin=50, n_sweeps=5), rng=rng)
    t0 = time.time()
    samples, info = s.parallel_tempering(chain(10), [0.5, 1.0, 2.0, 4.0], n_samples=500)
    print(f"rng={rng:6s} parallel_tempering n=10, 4 replicas, 500 samples: {(time.time() - t0) * 1e3:8.1f} ms  swap acceptance {info['swap_acceptance_rate']:.2f}", flush=True)
