"""The reference's sampling-benchmark harness on the MI355X backend.

Only the sampling part of ``tsu.benchmarks`` sits on the hot path (it times ``GibbsSampler.sample_boltzmann``); the
optimisation / ML / framework-comparison suites of the reference are out of scope."""
from .sampling import SamplingBenchmark, SamplingResult
from .runner import BenchmarkRunner

__all__ = ["SamplingBenchmark", "SamplingResult", "BenchmarkRunner"]
