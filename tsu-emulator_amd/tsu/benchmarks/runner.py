"""``BenchmarkRunner`` for the part of the reference's suite that sits on the hot path.

Reference: /root/reference/tsu/benchmarks/runner.py:16-102 (run_all), :155-191 (_save_results): results are kept as
``{category: {benchmark: result}}`` and saved as ``benchmark_results.json`` holding every result's ``summary()`` plus a
plain-text report.  Here the categories are ``"sampling"`` (times ``sample_boltzmann``) and ``"optimization"`` (times
``simulated_annealing``) -- the two suites that call the hot path; ML / framework comparison are outside the scope table
(SURVEY.md section 8).  The file names and the JSON layout are the reference's."""
import json
import time
from pathlib import Path

from .optimization import OptimizationBenchmark
from .sampling import SamplingBenchmark


def _plain(obj):
    """numpy scalars -> Python numbers (json)."""
    if isinstance(obj, dict):
        return {k: _plain(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_plain(v) for v in obj]
    return obj.item() if hasattr(obj, "item") else obj


class BenchmarkRunner:
    def __init__(self, seed: int = 42, output_dir: str = "visual_output", *, rng: str = "philox"):
        self.seed = seed
        self.output_dir = Path(output_dir)
        self.output_dir.mkdir(parents=True, exist_ok=True)
        self.rng = rng
        self.results = {}

    def run_all(self, quick: bool = False, save_results: bool = True, verbose: bool = True):
        t0 = time.time()
        self.results["sampling"] = SamplingBenchmark(seed=self.seed, rng=self.rng).run_all_benchmarks(quick=quick, verbose=verbose)
        self.results["optimization"] = OptimizationBenchmark(seed=self.seed, rng=self.rng).run_all_benchmarks(quick=quick, verbose=False)
        self.total_time = time.time() - t0
        if verbose:
            self._print_summary()
        if save_results:
            self._save_results()
        return self.results

    def serializable(self):
        return {cat: {name: _plain(res.summary()) for name, res in runs.items()} for cat, runs in self.results.items()}

    def _print_summary(self):
        print("SAMPLING BENCHMARKS:")
        for res in self.results["sampling"].values():
            s = res.summary()
            print(f"  {s['distribution']:25s}: KL={s['kl_divergence']['mean']:.4f}, ESS={s['effective_sample_size']['mean']:.0f}, "
                  f"Rate={s['throughput_samples_per_sec']['mean']:.0f}/s")
        print("OPTIMIZATION BENCHMARKS:")
        for res in self.results["optimization"].values():
            s = res.summary()
            gap = f", Gap={s['optimality_gap_percent']['mean']:.2f}%" if "optimality_gap_percent" in s else ""
            print(f"  {s['problem']:20s} (n={s['size']}): Time={s['solution_time_ms']['mean']:.1f}ms{gap}")

    def _save_results(self):
        data = self.serializable()
        (self.output_dir / "benchmark_results.json").write_text(json.dumps(data, indent=2))
        lines = ["=" * 80, "TSU BENCHMARK REPORT", "=" * 80, f"Random seed: {self.seed}",
                 f"Timestamp: {time.strftime('%Y-%m-%d %H:%M:%S')}", "=" * 80, ""]
        for category, runs in data.items():
            lines += ["", f"{category.upper()} BENCHMARKS", "-" * 80]
            for name, summary in runs.items():
                lines += ["", f"{name}:", json.dumps(summary, indent=2)]
        (self.output_dir / "benchmark_report.txt").write_text("\n".join(lines) + "\n")


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description="Run the sampling and optimisation benchmarks on the MI355X backend")
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--output-dir", default="visual_output")
    ap.add_argument("--rng", default="philox", choices=("philox", "numpy"))
    args = ap.parse_args(argv)
    BenchmarkRunner(seed=args.seed, output_dir=args.output_dir, rng=args.rng).run_all(quick=args.quick)


if __name__ == "__main__":
    main()
