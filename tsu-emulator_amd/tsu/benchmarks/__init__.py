"""The reference's benchmark harness on the MI355X backend (SURVEY.md section 8, row f4 and the callers of row f2).

The sampling suite times ``GibbsSampler.sample_boltzmann``, the optimisation suite ``GibbsSampler.simulated_annealing``; the
ML / framework-comparison suites of the reference (Bayesian-network models on top of the samplers, timings of other
frameworks) are outside the scope table."""
from .optimization import OptimizationBenchmark, OptimizationResult
from .sampling import SamplingBenchmark, SamplingResult
from .runner import BenchmarkRunner

__all__ = ["SamplingBenchmark", "SamplingResult", "OptimizationBenchmark", "OptimizationResult", "BenchmarkRunner"]
