"""f4: the reference's sampling- and optimisation-benchmark harness on this backend (tsu.benchmarks.sampling / optimization / runner).

CPU part: the harness's metric arithmetic and record layout against the reference's own run (tests/golden/g11, produced by
/root/reference/tsu/benchmarks/sampling.py with seed 42), with the sampler replaced by an oracle-backed TEST DOUBLE that
replays np.random exactly as the reference consumes it -- every quality number must then equal the reference's.
GPU part (-m gpu): the same equality with the real HIP sampler in rng="numpy" mode, and the reference's acceptance checks
(KL / pass fractions) plus a throughput above the reference's own CPU rate with the device RNG.  The optimisation suite (callers of
simulated_annealing) is pinned the same way by tests/golden/g12."""
import json

import numpy as np
import pytest

from oracle import oracle as ora

QUALITY = ("ks_statistics", "ks_pvalues", "kl_divergences", "effective_sample_sizes")
KEYS = {"gaussian_1d": "Uniform_Binary(dim=1)", "boltzmann": "Boltzmann(n=10)", "multimodal": "Ferromagnetic_Bimodal"}


class ReplaySampler:
    """Test double for GibbsSampler: the reference's sample_boltzmann (gibbs.py:164-213) by the oracle's C replay of the
    sequential loop, consuming np.random in the reference's order (randint for the initial state, then one rand per site)."""

    def __init__(self, config):
        self.config = config
        self.sample_count = 0

    def sample_boltzmann(self, coupling, bias=None, n_samples=1000, burnin=None, initial_state=None):
        n = coupling.shape[0]
        burnin = self.config.n_burnin if burnin is None else burnin
        state = np.random.randint(0, 2, size=n)
        total = burnin + n_samples * self.config.n_sweeps
        u = np.random.rand(total, n)
        out = np.zeros((n_samples, n), dtype=int)
        state = ora.c_dense_sweep_replay(state, coupling, bias, self.config.temperature, u[:burnin]) if burnin else state
        pos = burnin
        for k in range(n_samples):
            state = ora.c_dense_sweep_replay(state, coupling, bias, self.config.temperature, u[pos:pos + self.config.n_sweeps])
            pos += self.config.n_sweeps
            out[k] = state
        self.sample_count += n_samples
        return out


class ReplayAnnealer:
    """Test double for GibbsSampler in the optimisation suite: the reference's simulated_annealing (gibbs.py:340-393) with the
    oracle's C replay of one sequential sweep per temperature, consuming np.random in the reference's order."""

    def __init__(self, config):
        self.config = config

    def simulated_annealing(self, coupling, bias=None, T_initial=10.0, T_final=0.1, n_steps=1000, cooling_schedule="exponential"):
        n = coupling.shape[0]
        state = np.random.randint(0, 2, size=n)
        best_state, best_energy = state.copy(), ora.ref_compute_energy(state, coupling, bias)
        for step in range(n_steps):
            T = T_initial * (T_final / T_initial) ** (step / n_steps)
            state = ora.c_dense_sweep_replay(state, coupling, bias, T, np.random.rand(1, n))
            energy = ora.ref_compute_energy(state, coupling, bias)
            if energy < best_energy:
                best_energy, best_state = energy, state.copy()
        return best_state, best_energy


OPT_KEYS = {"maxcut": "MAX-CUT", "coloring": "3-Coloring", "partition": "Number-Partition"}


def _check_optimization_against_reference(results, g12, mode):
    assert list(results) == list(OPT_KEYS)
    for key, title in OPT_KEYS.items():
        r = results[key]
        assert r.problem_name == title == str(g12[f"{mode}_{key}_problem"])
        assert r.problem_size == int(g12[f"{mode}_{key}_size"]) and r.n_trials == int(g12[f"{mode}_{key}_n_trials"])
        want = g12[f"{mode}_{key}_best_objectives"]
        assert np.array_equal(r.best_objectives, want) and np.array_equal(np.signbit(r.best_objectives), np.signbit(want)), (key, r.best_objectives, want)
        assert np.array_equal(r.final_objectives, g12[f"{mode}_{key}_final_objectives"])
        assert r.optimal_objective == float(g12[f"{mode}_{key}_optimal_objective"])
        assert np.array_equal(r.n_iterations, g12[f"{mode}_{key}_n_iterations"])
        s = r.summary()
        assert sorted(s.keys()) == list(g12[f"{mode}_{key}_summary_keys"])
        assert s["optimality_gap_percent"]["mean"] == pytest.approx(float(g12[f"{mode}_{key}_gap_mean"]), rel=1e-12)
        assert len(r.solution_times) == r.n_trials and all(t >= 0 for t in r.solution_times)


def _check_against_reference(results, g11, mode):
    for key, title in KEYS.items():
        r = results[key]
        assert r.distribution_name == title == str(g11[f"{mode}_{key}_distribution"])
        assert r.n_samples == int(g11[f"{mode}_{key}_n_samples"]) and r.n_trials == int(g11[f"{mode}_{key}_n_trials"])
        for q in QUALITY:
            np.testing.assert_allclose(getattr(r, q), g11[f"{mode}_{key}_{q}"], rtol=1e-9, atol=1e-12, err_msg=f"{key}.{q}")
        assert sorted(r.summary().keys()) == list(g11[f"{mode}_{key}_summary_keys"])
        assert len(r.sampling_times) == r.n_trials and all(t > 0 for t in r.sampling_times)


def test_harness_metrics_equal_the_reference_run_quick_mode(golden, monkeypatch):
    from tsu.benchmarks import sampling as sb
    monkeypatch.setattr(sb, "GibbsSampler", lambda config, rng, coupling_dtype: ReplaySampler(config))
    res = sb.SamplingBenchmark(seed=42).run_all_benchmarks(quick=True, verbose=False)
    _check_against_reference(res, golden("g11_sampling_benchmark"), "quick")


@pytest.mark.parametrize("mode", ["quick", "full"])
def test_optimization_harness_equals_the_reference_run(golden, monkeypatch, mode):
    """Graph and number draws, the greedy bound, the colouring record and -- through the replaying test double -- the annealed
    objectives: all equal to the reference's run with seed 42 (G12)."""
    from tsu.benchmarks import optimization as ob
    monkeypatch.setattr(ob, "GibbsSampler", lambda config, rng, **kw: ReplayAnnealer(config))
    res = ob.OptimizationBenchmark(seed=42).run_all_benchmarks(quick=(mode == "quick"), verbose=False)
    _check_optimization_against_reference(res, golden("g12_optimization_benchmark"), mode)


def test_optimization_inputs_and_greedy_bound_are_the_reference_draws(golden):
    from tsu.benchmarks import optimization as ob
    g12 = golden("g12_optimization_benchmark")
    for mode, n in (("quick", 15), ("full", 20)):
        np.random.seed(42)
        a = ob.random_graph(n, 0.5)
        assert np.array_equal(a, g12[f"{mode}_maxcut_adjacency"])
        assert -ob.greedy_maxcut(a) == float(g12[f"{mode}_maxcut_optimal_objective"])
        np.random.seed(42)
        assert np.array_equal(np.random.randint(1, 100, size=n), g12[f"{mode}_partition_numbers"])
    for mode, n in (("quick", 10), ("full", 15)):
        np.random.seed(42)
        assert np.array_equal(ob.random_graph(n, 0.4), g12[f"{mode}_coloring_adjacency"])
    # a cut by hand: path 0-1-2 with weights 1 and 0.5, sides (0, 1, 1) -> only the first edge is cut
    a = np.array([[0, 1, 0], [1, 0, 0.5], [0, 0.5, 0]])
    assert ob.cut_weight(a, np.array([0, 1, 1])) == 1.0 and ob.cut_weight(a, np.array([0, 1, 0])) == 1.5


def test_runner_writes_the_reference_record_layout(golden, monkeypatch, tmp_path):
    from tsu.benchmarks import optimization as ob
    from tsu.benchmarks import sampling as sb
    from tsu.benchmarks.runner import BenchmarkRunner
    monkeypatch.setattr(sb, "GibbsSampler", lambda config, rng, coupling_dtype: ReplaySampler(config))
    monkeypatch.setattr(ob, "GibbsSampler", lambda config, rng, **kw: ReplayAnnealer(config))
    runner = BenchmarkRunner(seed=42, output_dir=str(tmp_path / "visual_output"))
    runner.run_all(quick=True, verbose=False)
    data = json.loads((tmp_path / "visual_output" / "benchmark_results.json").read_text())
    assert list(data) == ["sampling", "optimization"] and list(data["sampling"]) == list(KEYS) and list(data["optimization"]) == list(OPT_KEYS)
    g12 = golden("g12_optimization_benchmark")
    for key in OPT_KEYS:
        s = data["optimization"][key]
        assert sorted(s) == list(g12[f"quick_{key}_summary_keys"])
        assert set(s["best_objective"]) == {"mean", "std", "best", "worst"} and set(s["solution_time_ms"]) == {"mean", "std", "median"}
        assert s["best_objective"]["mean"] == float(np.mean(g12[f"quick_{key}_best_objectives"]))
    g11 = golden("g11_sampling_benchmark")
    for key in KEYS:
        s = data["sampling"][key]
        assert sorted(s) == list(g11[f"quick_{key}_summary_keys"])
        assert set(s["ks_pvalue"]) == {"mean", "std", "fraction_passed"}
        for k in ("ks_statistic", "kl_divergence", "effective_sample_size", "sampling_time_ms", "throughput_samples_per_sec"):
            assert set(s[k]) == {"mean", "std", "median"}
        assert s["kl_divergence"]["mean"] == pytest.approx(float(np.mean(g11[f"quick_{key}_kl_divergences"])), rel=1e-9)
    report = (tmp_path / "visual_output" / "benchmark_report.txt").read_text()
    assert "TSU BENCHMARK REPORT" in report and "SAMPLING BENCHMARKS" in report and "OPTIMIZATION BENCHMARKS" in report


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["quick", "full"])
def test_hip_sampler_with_replayed_numpy_stream_reproduces_the_reference_benchmark(golden, mode):
    from tsu.benchmarks import SamplingBenchmark
    res = SamplingBenchmark(seed=42, rng="numpy").run_all_benchmarks(quick=(mode == "quick"), verbose=False)
    _check_against_reference(res, golden("g11_sampling_benchmark"), mode)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["quick", "full"])
def test_hip_annealer_with_replayed_numpy_stream_reproduces_the_reference_optimization_benchmark(golden, mode):
    from tsu.benchmarks import OptimizationBenchmark
    res = OptimizationBenchmark(seed=42, rng="numpy").run_all_benchmarks(quick=(mode == "quick"), verbose=False)
    _check_optimization_against_reference(res, golden("g12_optimization_benchmark"), mode)


@pytest.mark.gpu
def test_hip_annealer_device_rng_reaches_the_reference_objectives_faster(golden):
    """Device RNG: other draws, but these problems have a unique bottom the annealer must reach as the reference's does (its
    MAX-CUT couplings -A on BITS are minimised by the empty set, its partition couplings by the full set), in less time
    than the reference's CPU took for the same schedule (stored in G12 for information)."""
    from tsu.benchmarks import OptimizationBenchmark
    g12 = golden("g12_optimization_benchmark")
    res = OptimizationBenchmark(seed=42).run_all_benchmarks(quick=False, verbose=False)
    for key in ("maxcut", "partition"):
        assert np.array_equal(res[key].best_objectives, g12[f"full_{key}_best_objectives"]), key
        assert np.median(res[key].solution_times) < np.median(g12[f"full_{key}_solution_times"]), key
    assert np.array_equal(res["coloring"].best_objectives, g12["full_coloring_best_objectives"])


@pytest.mark.gpu
def test_hip_sampler_device_rng_passes_the_reference_checks_and_beats_its_rate(golden, tmp_path):
    from tsu.benchmarks import BenchmarkRunner
    g11 = golden("g11_sampling_benchmark")
    runner = BenchmarkRunner(seed=42, output_dir=str(tmp_path / "out"))
    res = runner.run_all(quick=False, verbose=False)["sampling"]
    uni, chain, bi = res["gaussian_1d"], res["boltzmann"], res["multimodal"]
    # the reference's own acceptance numbers (benchmarks/sampling.py:132-157, 195-218, 252-268)
    assert np.mean(uni.kl_divergences) < 1e-3 and np.mean(np.array(uni.ks_pvalues) > 0.05) >= 0.6
    assert np.mean(uni.effective_sample_sizes) > 9000
    # chain: |mean energy| is a property of the Boltzmann distribution (bits, couplings J, T = 1): equal to the reference's
    # value within Monte-Carlo error; all trials magnetised
    ref_e = g11["full_boltzmann_kl_divergences"]
    assert abs(np.mean(chain.kl_divergences) - ref_e.mean()) < 0.1, (np.mean(chain.kl_divergences), ref_e.mean())
    assert all(p == 1.0 for p in chain.ks_pvalues)
    # all-to-all ferromagnet at T = 1 on bits: every sample sits in the all-ones mode (the reference finds the same:
    # mode balance 0, tiny spread) -- equality of the spread within Monte-Carlo error
    assert np.mean(bi.ks_statistics) == pytest.approx(float(g11["full_multimodal_ks_statistics"].mean()), abs=0.02)
    assert abs(np.mean(bi.kl_divergences) - g11["full_multimodal_kl_divergences"].mean()) < 0.01
    for key in KEYS:
        ours = np.median(res[key].samples_per_second)
        theirs = np.median(g11[f"full_{key}_samples_per_second"])   # the reference, timed in the build container (CPU)
        assert ours > theirs, (key, ours, theirs)
    data = json.loads((tmp_path / "out" / "benchmark_results.json").read_text())
    assert list(data["sampling"]) == list(KEYS)
