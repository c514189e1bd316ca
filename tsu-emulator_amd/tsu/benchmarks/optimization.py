"""Optimisation benchmark cases of the reference, run on the HIP annealer.

Interface and record layout follow /root/reference/tsu/benchmarks/optimization.py:21-392 (``OptimizationResult`` fields and the
keys of ``summary()``; ``OptimizationBenchmark(config, seed)`` with ``benchmark_maxcut / benchmark_graph_coloring /
benchmark_number_partitioning / run_all_benchmarks``).  Two of the three cases are callers of
``GibbsSampler.simulated_annealing`` (reference: gibbs.py:340-393; here: one device launch per schedule, SURVEY.md
section 8 row f2); the colouring case of the reference never samples -- it draws one random colouring per trial and counts
conflicts -- and is kept for the record layout only.

What is random is drawn from ``np.random`` in the reference's order (graph, greedy bound, the annealer's start state), so
with ``rng="numpy"`` -- the annealer replaying ``np.random`` for its uniforms too -- every objective equals the
reference's own (tests/golden/g12); with the default ``rng="philox"`` the annealer draws on the device.  The graph
arithmetic is written for arrays: the greedy local search evaluates a node's gain as one masked row sum instead of a
Python loop over its neighbours (same visiting order, same flips).
"""
import time
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

from ..gibbs import GibbsConfig, GibbsSampler


@dataclass
class OptimizationResult:
    """Per-trial numbers of one problem (reference: benchmarks/optimization.py:21-80, same field names and summary keys)."""

    problem_name: str
    problem_size: int
    n_trials: int
    best_objectives: List[float] = field(default_factory=list)
    final_objectives: List[float] = field(default_factory=list)
    optimal_objective: Optional[float] = None
    solution_times: List[float] = field(default_factory=list)
    n_iterations: List[int] = field(default_factory=list)
    convergence_curves: List[List[float]] = field(default_factory=list)

    def summary(self) -> Dict:
        best = np.asarray(self.best_objectives, dtype=float)
        t_ms = np.asarray(self.solution_times, dtype=float) * 1000.0
        out = {
            "problem": self.problem_name, "size": self.problem_size, "n_trials": self.n_trials,
            "best_objective": {"mean": np.mean(best), "std": np.std(best), "best": np.min(best), "worst": np.max(best)},
            "solution_time_ms": {"mean": np.mean(t_ms), "std": np.std(t_ms), "median": np.median(t_ms)},
            "iterations": {"mean": np.mean(self.n_iterations), "std": np.std(self.n_iterations)},
        }
        if self.optimal_objective is not None:
            opt = self.optimal_objective
            # relative gap in per cent; against an optimum of zero (a perfect colouring / partition) the absolute difference
            gaps = (best - opt) / abs(opt) * 100 if abs(opt) > 1e-10 else np.abs(best - opt)
            out["optimality_gap_percent"] = {"mean": np.mean(gaps), "std": np.std(gaps), "best": np.min(gaps)}
        return out


def random_graph(n_nodes: int, edge_density: float) -> np.ndarray:
    """The reference's random weighted graph (optimization.py:129-131, 201-203): a uniform n x n draw thresholded at the
    density, symmetrised by averaging (so a pair drawn in one direction only weighs 0.5), no self-loops."""
    a = (np.random.rand(n_nodes, n_nodes) < edge_density).astype(float)
    a = (a + a.T) / 2
    np.fill_diagonal(a, 0)
    return a


def cut_weight(adjacency: np.ndarray, partition: np.ndarray) -> float:
    """Total weight of the edges between the two sides."""
    across = partition[:, None] != partition[None, :]
    return float(np.sum(np.triu(adjacency * across, 1)))


def greedy_maxcut(adjacency: np.ndarray, max_iterations: int = 100) -> float:
    """Local search from a random partition (one ``np.random.randint`` draw): passes over the nodes in index order, a node
    changes side whenever that strictly raises the cut, until a pass changes nothing (reference: optimization.py:292-338).
    A node's gain is the weight to its own side minus the weight across: one masked row sum."""
    n = len(adjacency)
    side = np.random.randint(0, 2, size=n)
    for _ in range(max_iterations):
        changed = False
        for i in range(n):
            same = side == side[i]
            same[i] = False
            w = adjacency[i]
            if w[same].sum() > w[~same].sum():  # (w[i] = 0: no self-loops, so i itself adds nothing to the other side)
                side[i] = 1 - side[i]
                changed = True
        if not changed:
            break
    return cut_weight(adjacency, side)


class OptimizationBenchmark:
    """MAX-CUT, number partitioning (both by simulated annealing from T = 10 to 0.01) and the reference's colouring record."""

    T_INITIAL, T_FINAL = 10.0, 0.01

    def __init__(self, config: Optional[GibbsConfig] = None, seed: int = 42, *, rng: str = "philox", coupling_dtype=None):
        self.config = config or GibbsConfig(temperature=1.0, n_burnin=100, n_sweeps=10)
        self.seed = seed
        kw = {} if coupling_dtype is None else {"coupling_dtype": coupling_dtype}
        self.sampler = GibbsSampler(self.config, rng=rng, **kw)

    def _anneal_trials(self, result: OptimizationResult, J: np.ndarray, n_trials: int, n_steps: int, objective) -> None:
        h = np.zeros(len(J))
        for trial in range(n_trials):
            np.random.seed(self.seed + trial)
            t0 = time.time()
            state, energy = self.sampler.simulated_annealing(J, bias=h, T_initial=self.T_INITIAL, T_final=self.T_FINAL, n_steps=n_steps)
            elapsed = time.time() - t0
            value = float(objective(state, energy))
            result.best_objectives.append(value)
            result.final_objectives.append(value)
            result.solution_times.append(elapsed)
            result.n_iterations.append(n_steps)
            result.convergence_curves.append([])

    def benchmark_maxcut(self, n_nodes: int = 20, edge_density: float = 0.5, n_trials: int = 5, n_steps: int = 1000) -> OptimizationResult:
        """Couplings J = -A on the bits; the objective is the annealer's best energy, the bound the greedy cut (negated)."""
        result = OptimizationResult(problem_name="MAX-CUT", problem_size=n_nodes, n_trials=n_trials)
        np.random.seed(self.seed)
        adjacency = random_graph(n_nodes, edge_density)
        result.optimal_objective = -greedy_maxcut(adjacency)
        self._anneal_trials(result, -adjacency, n_trials, n_steps, lambda state, energy: energy)
        return result

    def benchmark_graph_coloring(self, n_nodes: int = 15, n_colors: int = 3, edge_density: float = 0.4, n_trials: int = 5,
                                 n_steps: int = 1000) -> OptimizationResult:
        """As in the reference (optimization.py:166-226): one random colouring per trial, objective = conflicting edges."""
        result = OptimizationResult(problem_name=f"{n_colors}-Coloring", problem_size=n_nodes, n_trials=n_trials, optimal_objective=0.0)
        np.random.seed(self.seed)
        edges = np.triu(random_graph(n_nodes, edge_density) > 0, 1)
        for trial in range(n_trials):
            np.random.seed(self.seed + trial)
            t0 = time.time()
            colors = np.random.randint(0, n_colors, size=n_nodes)
            conflicts = float(np.count_nonzero(edges & (colors[:, None] == colors[None, :])))
            elapsed = time.time() - t0
            result.best_objectives.append(conflicts)
            result.final_objectives.append(conflicts)
            result.solution_times.append(elapsed)
            result.n_iterations.append(1)
        return result

    def benchmark_number_partitioning(self, n_numbers: int = 20, n_trials: int = 5, n_steps: int = 1000) -> OptimizationResult:
        """Couplings J = numbers numbers^T; the objective is |sum of the signed numbers| of the best state."""
        result = OptimizationResult(problem_name="Number-Partition", problem_size=n_numbers, n_trials=n_trials, optimal_objective=0.0)
        np.random.seed(self.seed)
        numbers = np.random.randint(1, 100, size=n_numbers)
        self._anneal_trials(result, np.outer(numbers, numbers), n_trials, n_steps,
                            lambda state, energy: abs(np.dot(2 * np.asarray(state) - 1, numbers)))
        return result

    def run_all_benchmarks(self, quick: bool = False, verbose: bool = True) -> Dict[str, OptimizationResult]:
        """Sizes as in the reference (optimization.py:340-361): quick = 15 / 10 / 15 nodes, 500 steps, 3 trials; else 20 / 15 / 20, 1000, 5."""
        n_cut, n_col, n_part, n_steps, n_trials = (15, 10, 15, 500, 3) if quick else (20, 15, 20, 1000, 5)
        results = {
            "maxcut": self.benchmark_maxcut(n_nodes=n_cut, n_trials=n_trials, n_steps=n_steps),
            "coloring": self.benchmark_graph_coloring(n_nodes=n_col, n_trials=n_trials, n_steps=n_steps),
            "partition": self.benchmark_number_partitioning(n_numbers=n_part, n_trials=n_trials, n_steps=n_steps),
        }
        if verbose:
            for name, res in results.items():
                s = res.summary()
                gap = f", gap {s['optimality_gap_percent']['mean']:.2f}" if "optimality_gap_percent" in s else ""
                print(f"{name:10s} best {s['best_objective']['best']:.2f}  mean {s['best_objective']['mean']:.2f}  "
                      f"{s['solution_time_ms']['mean']:.1f} ms{gap}")
        return results
