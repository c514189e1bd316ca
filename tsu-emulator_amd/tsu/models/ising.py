"""Ising models on the MI355X -- drop-in for the reference's ``tsu.models.ising``.

Same classes, signatures, return types and error messages as the reference (file:line cited per symbol).
What differs is where the work happens:

* :class:`IsingGrid` keeps the lattice as ``(rows, cols, J, h, periodic)`` and samples it with the hand-written
  red-black checkerboard HIP kernel (``libtsu_hip.so`` K1) on int8 spins.  The reference stores even a lattice
  as a dense N x N float64 matrix (ising.py:64,343-361) and walks it site by site; here ``.J`` is materialised
  lazily only when somebody reads it.  Checkerboard order is a different visiting order of the same heat-bath
  kernel: same stationary distribution, different trajectory.
* :class:`IsingModel` / :class:`IsingChain` (arbitrary graph) go through the dense HIP path of
  :class:`tsu.gibbs.GibbsSampler` in the reference's own sequential order.

``bias_mode``.  The reference's spin->bit bias conversion has a sign error (ising.py:148 returns
``-2h + 2 rowsum(J)``; the conversion of E = -1/2 s'Js - h's is ``+2h - 2 rowsum(J)``), so as shipped it samples
an Ising model with effective field ``h_eff = 2 rowsum(J) - h``.  ``bias_mode="compat"`` (default for the
reference-named classes, bug-for-bug drop-in) reproduces that; ``bias_mode="physical"`` uses the corrected
conversion (default of the README-named :class:`IsingModel2D`).
"""
from dataclasses import dataclass
from typing import List, Optional, Tuple

import numpy as np

from .. import _hip
from ..gibbs import GibbsConfig, GibbsSampler, _content_key


@dataclass
class IsingConfig:
    """Reference: tsu/models/ising.py:25-36."""

    temperature: float = 1.0
    external_field: float = 0.0  # uniform external field h
    n_burnin: int = 100
    n_sweeps: int = 10

    def __post_init__(self):
        if self.temperature <= 0:
            raise ValueError("Temperature must be positive")


def _mode_id(bias_mode: str) -> int:
    if bias_mode not in ("compat", "physical"):
        raise ValueError("bias_mode must be 'compat' or 'physical'")
    return _hip.MODE_COMPAT if bias_mode == "compat" else _hip.MODE_PHYSICAL


class IsingModel:
    """General Ising model on an arbitrary graph.  Reference: tsu/models/ising.py:39-262.

    ``graph="dense"`` keeps the coupling matrix as the reference does (an N x N array, sites visited in raster order on
    the dense kernel K2).  ``graph="sparse"`` keeps only the couplings that were set (``scipy.sparse``) and samples on
    the colour-parallel sparse kernel K5 -- the sequential loop of gibbs.py:128-162 in the colour-major visiting order of
    a proper colouring -- which is what makes a 10^6-site chain possible (its dense J would be 8 TB).  ``"auto"``
    (default): sparse above ``DENSE_LIMIT`` sites, dense below."""

    DENSE_LIMIT = 16384  # largest N for which a dense J is kept / may be materialised (2 GiB of float64)

    def __init__(self, n_spins: int, config: Optional[IsingConfig] = None, *, bias_mode: str = "compat", graph: str = "auto"):
        _mode_id(bias_mode)
        if graph not in ("auto", "dense", "sparse"):
            raise ValueError("graph must be 'auto', 'dense' or 'sparse'")
        self.n_spins = n_spins
        self.config = config or IsingConfig()
        self.bias_mode = bias_mode
        self.sparse = graph == "sparse" or (graph == "auto" and n_spins > self.DENSE_LIMIT)
        if self.sparse:
            import scipy.sparse as sp
            self._J = None
            self._Jsp = sp.csr_matrix((n_spins, n_spins), dtype=np.float64)
        else:
            self._J = np.zeros((n_spins, n_spins))
        self.h = np.ones(n_spins) * self.config.external_field
        gibbs_config = GibbsConfig(temperature=self.config.temperature, n_burnin=self.config.n_burnin,
                                   n_sweeps=self.config.n_sweeps)
        self.sampler = GibbsSampler(gibbs_config)

    # ``J`` is a plain attribute in the reference; a property here so that IsingGrid can build it lazily
    @property
    def J(self) -> np.ndarray:
        if getattr(self, "sparse", False):
            if self.n_spins > self.DENSE_LIMIT:
                raise MemoryError(f"dense J for {self.n_spins} spins would need {8 * self.n_spins ** 2 / 2 ** 30:.0f} GiB; "
                                  "the sparse kernel does not need it (see .J_sparse)")
            dense = self._Jsp.toarray()
            dense.setflags(write=False)  # a copy: edits would be lost -- use set_coupling()
            return dense
        return self._J

    @J.setter
    def J(self, value):
        if getattr(self, "sparse", False):
            import scipy.sparse as sp
            self._Jsp = sp.csr_matrix(value, dtype=np.float64)
            self.sampler.invalidate()
        else:
            self._J = np.asarray(value, dtype=float)

    @property
    def J_sparse(self):
        """The coupling matrix as ``scipy.sparse`` CSR (sparse models: the stored graph; dense models: a conversion)."""
        import scipy.sparse as sp
        return self._csr() if self.sparse else sp.csr_matrix(self.J)

    def _csr(self):
        """CSR form of a sparse model's couplings (converted once after the last set_coupling: energy() runs per sample)."""
        if getattr(self._Jsp, "format", "") != "csr":
            self._Jsp = self._Jsp.tocsr()
        return self._Jsp

    def set_coupling(self, i: int, j: int, strength: float):
        """Reference: ising.py:77-86 (symmetric assignment, not accumulation)."""
        if self.sparse:
            if not hasattr(self._Jsp, "rows"):       # CSR -> LIL once: cheap element assignment
                self._Jsp = self._Jsp.tolil()
            self._Jsp[i, j] = strength
            self._Jsp[j, i] = strength
        else:
            self.J[i, j] = strength
            self.J[j, i] = strength
        self.sampler.invalidate()

    def set_external_field(self, field: np.ndarray):
        """Reference: ising.py:88-97."""
        if len(field) != self.n_spins:
            raise ValueError(f"Field must have length {self.n_spins}")
        self.h = np.array(field)

    def energy(self, state: np.ndarray) -> float:
        """Reference: ising.py:99-117 -- E(s) = -1/2 s'Js - h's."""
        state = np.asarray(state)
        if self.sparse:
            interaction_energy = -0.5 * state.dot(self._csr().dot(state))
        else:
            interaction_energy = -0.5 * state.dot(self.J).dot(state)
        field_energy = -self.h.dot(state)
        return interaction_energy + field_energy

    def _spins_to_bits(self, spins: np.ndarray) -> np.ndarray:
        """Reference: ising.py:119-121."""
        return ((spins + 1) // 2).astype(int)

    def _bits_to_spins(self, bits: np.ndarray) -> np.ndarray:
        """Reference: ising.py:123-125."""
        return 2 * bits - 1

    def _get_bit_coupling(self) -> np.ndarray:
        """Reference: ising.py:127-138 -- J_bit = 4 J."""
        if self.sparse:
            return (4 * self._csr()).tocsr()
        return 4 * self.J

    def _get_bit_bias(self) -> np.ndarray:
        """Reference: ising.py:140-148 (``compat``: verbatim, including its sign), or the corrected conversion."""
        rowsum = np.asarray(self._csr().sum(axis=1)).ravel() if self.sparse else np.sum(self.J, axis=1)
        if self.bias_mode == "compat":
            return -2 * self.h + 2 * rowsum
        return 2 * self.h - 2 * rowsum

    def sample(self, n_samples: int = 1000, initial_state: Optional[np.ndarray] = None) -> np.ndarray:
        """Reference: ising.py:150-181 -- (n_samples, n_spins) array of +-1."""
        J_bit = self._get_bit_coupling()
        h_bit = self._get_bit_bias()
        initial_bits = self._spins_to_bits(np.asarray(initial_state)) if initial_state is not None else None
        bit_samples = self.sampler.sample_boltzmann(J_bit, bias=h_bit, n_samples=n_samples, initial_state=initial_bits)
        return self._bits_to_spins(bit_samples)

    def magnetization(self, samples: np.ndarray) -> float:
        """Reference: ising.py:183-193."""
        return np.mean(np.sum(samples, axis=1)) / self.n_spins

    def specific_heat(self, samples: np.ndarray) -> float:
        """Reference: ising.py:195-213."""
        energies = np.array([self.energy(s) for s in samples])
        mean_E = np.mean(energies)
        mean_E2 = np.mean(energies ** 2)
        T = self.config.temperature
        return float((mean_E2 - mean_E ** 2) / (T ** 2 * self.n_spins))

    def susceptibility(self, samples: np.ndarray) -> float:
        """Reference: ising.py:215-233."""
        magnetizations = np.sum(samples, axis=1) / self.n_spins
        mean_M = np.mean(magnetizations)
        mean_M2 = np.mean(magnetizations ** 2)
        T = self.config.temperature
        return (mean_M2 - mean_M ** 2) * self.n_spins / T

    def find_ground_state(self, n_steps: int = 1000) -> Tuple[np.ndarray, float]:
        """Reference: ising.py:235-262 (simulated annealing from 10 T down to 0.01 T)."""
        J_bit = self._get_bit_coupling()
        h_bit = self._get_bit_bias()
        best_bits, _ = self.sampler.simulated_annealing(J_bit, bias=h_bit, T_initial=10.0 * self.config.temperature,
                                                        T_final=0.01 * self.config.temperature, n_steps=n_steps)
        ground_state = self._bits_to_spins(best_bits)
        return ground_state, self.energy(ground_state)


class IsingChain(IsingModel):
    """1-D chain with open ends.  Reference: ising.py:265-304."""

    def __init__(self, n_spins: int, J: float = 1.0, config: Optional[IsingConfig] = None, *, bias_mode: str = "compat",
                 graph: str = "auto"):
        super().__init__(n_spins, config, bias_mode=bias_mode, graph=graph)
        if self.sparse:
            import scipy.sparse as sp
            off = np.full(max(n_spins - 1, 0), float(J))
            self._Jsp = sp.diags([off, off], [1, -1], shape=(n_spins, n_spins), format="csr", dtype=np.float64)
        else:
            for i in range(n_spins - 1):
                self._J[i, i + 1] = J
                self._J[i + 1, i] = J

    def visualize(self, state: np.ndarray, title: str = "Ising Chain"):
        """Reference: ising.py:288-304."""
        import matplotlib.pyplot as plt
        plt.figure(figsize=(12, 2))
        colors = ["blue" if s == 1 else "red" for s in state]
        plt.bar(range(self.n_spins), np.ones(self.n_spins), color=colors, width=1.0)
        plt.xlabel("Spin Index")
        plt.ylabel("State")
        plt.title(title)
        plt.ylim([0, 1.2])
        plt.tight_layout()
        return plt.gcf()


def _grid_coupling(rows: int, cols: int, J: float, periodic: bool) -> np.ndarray:
    """Dense coupling matrix of the square lattice, bond by bond as ising.py:343-361 builds it (bonds are SET:
    on a periodic dimension of size 2 the wrap bond coincides with the direct one, of size 1 it is J_ii)."""
    n = rows * cols
    M = np.zeros((n, n))
    for i in range(rows):
        for j in range(cols):
            idx = i * cols + j
            if j < cols - 1:
                M[idx, idx + 1] = M[idx + 1, idx] = J
            elif periodic:
                M[idx, i * cols] = M[i * cols, idx] = J
            if i < rows - 1:
                M[idx, idx + cols] = M[idx + cols, idx] = J
            elif periodic:
                M[idx, j] = M[j, idx] = J
    return M


class IsingGrid(IsingModel):
    """2-D square lattice, nearest neighbours.  Reference: ising.py:307-421.

    Sampling runs on the lattice kernel whenever the model is still the uniform lattice it was built as
    (no ``set_coupling`` edits, uniform field) and the checkerboard exists (open boundaries: any shape;
    periodic: even dimensions >= 4).  Otherwise it falls through to the dense path of :class:`IsingModel`
    on the materialised matrix -- still on the GPU, in the reference's raster order.
    """

    DENSE_LIMIT = 16384  # largest N for which .J may be materialised (2 GiB of float64)

    def __init__(self, size: Tuple[int, int], J: float = 1.0, config: Optional[IsingConfig] = None,
                 periodic: bool = False, *, bias_mode: str = "compat", seed: Optional[int] = None):
        _mode_id(bias_mode)
        self.rows, self.cols = size
        self.n_spins = self.rows * self.cols
        self.config = config or IsingConfig()
        self.bias_mode = bias_mode
        self.periodic = periodic
        self.coupling = float(J)
        self.sparse = False     # the lattice has its own kernel (K1); the dense view is built on first access
        self._J = None
        self._custom = False    # set_coupling() was called / J was edited: no longer a uniform lattice
        self._J_key = None
        self.h = np.ones(self.n_spins) * self.config.external_field
        gibbs_config = GibbsConfig(temperature=self.config.temperature, n_burnin=self.config.n_burnin,
                                   n_sweeps=self.config.n_sweeps)
        self.sampler = GibbsSampler(gibbs_config)
        self._seed = None if seed is None else int(seed)
        self._sweep_counter = 0
        self._lattice = None

    # ------------------------------------------------------------------ dense view (lazy)
    @property
    def J(self) -> np.ndarray:
        if self._J is None:
            if self.n_spins > self.DENSE_LIMIT:
                raise MemoryError(f"dense J for {self.n_spins} spins would need {8 * self.n_spins ** 2 / 2 ** 30:.0f} GiB; "
                                  "the lattice kernel does not need it")
            self._J = _grid_coupling(self.rows, self.cols, self.coupling, self.periodic)
            self._J_key = _content_key(self._J)
        return self._J

    @J.setter
    def J(self, value):
        self._J = np.asarray(value, dtype=float)
        self._custom = True

    def _J_edited_in_place(self) -> bool:
        """The reference lets callers write into ``grid.J`` and samples from whatever it holds (ising.py:150-181 reads
        ``self.J`` at call time).  Once the dense matrix has been handed out, every sampling call therefore checks
        (full content hash) that it still is the uniform lattice before taking the lattice kernel."""
        if self._J is None or self._custom:
            return False
        if _content_key(self._J) != self._J_key:
            self._custom = True
            return True
        return False

    def set_coupling(self, i: int, j: int, strength: float):
        super().set_coupling(i, j, strength)
        self._custom = True

    # ------------------------------------------------------------------ lattice kernel plumbing
    def _lattice_ok(self) -> bool:
        if self._custom or self._J_edited_in_place() or np.any(self.h != self.h[0]):
            return False
        if self.periodic and (self.rows % 2 or self.cols % 2 or self.rows < 4 or self.cols < 4):
            return False  # no 2-colouring (and the reference's size-1/2 wrap bonds are special): dense path
        return True

    def _philox_seed(self) -> int:
        if self._seed is None:
            self._seed = int(np.random.randint(0, 2 ** 31 - 1)) | (int(np.random.randint(0, 2 ** 31 - 1)) << 31)
        return self._seed

    def _device_lattice(self) -> "_hip.Lattice":
        if self._lattice is None:
            self._lattice = _hip.Lattice(self.rows, self.cols, self.periodic)
        return self._lattice

    def _set_model(self, lat):
        # temperature is read at call time from the sampler's config, which callers mutate in place
        # (ising.py:491-492, gibbs.py:382)
        lat.set_model(self.coupling, float(self.h[0]), float(self.sampler.config.temperature), _mode_id(self.bias_mode))

    def _flat_to_grid(self, flat_state: np.ndarray) -> np.ndarray:
        """Reference: ising.py:363-365."""
        return flat_state.reshape(self.rows, self.cols)

    def _grid_to_flat(self, grid_state: np.ndarray) -> np.ndarray:
        """Reference: ising.py:367-369."""
        return grid_state.flatten()

    # ------------------------------------------------------------------ sampling / observables
    def sample(self, n_samples: int = 1000, initial_state: Optional[np.ndarray] = None) -> np.ndarray:
        """Reference: ising.py:150-181 specialised to the lattice: burn-in, then ``n_samples`` x ``n_sweeps``
        checkerboard sweeps, recording the lattice after each.  Returns (n_samples, N) int array of +-1."""
        if not self._lattice_ok():
            return super().sample(n_samples, initial_state)
        lat = self._device_lattice()
        cfg = self.sampler.config
        if initial_state is not None:
            s0 = np.asarray(initial_state).reshape(self.rows, self.cols)
            if not np.all((s0 == 1) | (s0 == -1)):
                raise ValueError("initial_state must contain only +1 / -1")
            lat.set_spins(s0.astype(np.int8))
        else:
            bits = np.random.randint(0, 2, size=self.n_spins)  # the reference's draw (gibbs.py:201)
            lat.set_spins((2 * bits - 1).astype(np.int8).reshape(self.rows, self.cols))
        self._set_model(lat)
        seed = self._philox_seed()
        samples = np.zeros((n_samples, self.n_spins), dtype=int)
        # burn-in and the n_samples x n_sweeps loop in as few device calls as a 1 GiB staging buffer allows
        chunk = max(1, min(int(n_samples), (1 << 30) // max(1, self.n_spins)))
        done, burn = 0, int(cfg.n_burnin)
        if n_samples == 0:
            lat.sweep(burn, seed, self._sweep_counter)
            self._sweep_counter += burn
        while done < n_samples:
            m = min(chunk, n_samples - done)
            samples[done:done + m] = lat.sample(burn, int(cfg.n_sweeps), m, seed, self._sweep_counter).reshape(m, -1)
            self._sweep_counter += burn + m * int(cfg.n_sweeps)
            self.sampler.sample_count += m
            done += m
            burn = 0
        return samples

    def energy(self, state: np.ndarray) -> float:
        """Reference: ising.py:99-117; on the lattice path E = -J sum_bonds s_i s_j - h sum_i s_i by a device
        reduction (no dense matrix)."""
        state = np.asarray(state)
        if not self._lattice_ok():
            return super().energy(state.reshape(-1))
        lat = self._device_lattice()
        lat.set_spins(state.reshape(self.rows, self.cols).astype(np.int8))
        sum_s, sum_bonds = lat.observables()
        return -self.coupling * float(sum_bonds) - float(self.h[0]) * float(sum_s)

    def visualize(self, state: np.ndarray, title: str = "Ising Grid", cmap: str = "RdBu_r"):
        """Reference: ising.py:371-401."""
        import matplotlib.pyplot as plt
        if state.ndim == 1:
            state = self._flat_to_grid(state)
        fig, ax = plt.subplots(figsize=(8, 8))
        im = ax.imshow(state, cmap=cmap, vmin=-1, vmax=1, interpolation="nearest")
        ax.set_title(title, fontsize=14, fontweight="bold")
        ax.set_xlabel("Column Index")
        ax.set_ylabel("Row Index")
        cbar = plt.colorbar(im, ax=ax, fraction=0.046, pad=0.04)
        cbar.set_label("Spin State", rotation=270, labelpad=20)
        cbar.set_ticks([-1, 0, 1])
        cbar.set_ticklabels(["-1", "0", "+1"])
        plt.tight_layout()
        return fig

    def compute_domains(self, state: np.ndarray) -> int:
        """Reference: ising.py:403-421 (boundary count // 2 + 1; open-boundary differences only)."""
        if state.ndim == 1:
            state = self._flat_to_grid(state)
        horizontal_boundaries = np.sum(state[:, :-1] != state[:, 1:])
        vertical_boundaries = np.sum(state[:-1, :] != state[1:, :])
        return (horizontal_boundaries + vertical_boundaries) // 2 + 1


class IsingModel2D:
    """README facade (README.md:116-131): a lattice that lives on the GPU between calls.

    ``IsingModel2D(size=50, coupling=1.0, temperature=2.5)``; ``gibbs_update()`` = one checkerboard sweep;
    ``magnetization()`` / ``energy()`` = observables of the current state by a device reduction;
    ``equilibrate(T)`` sets the temperature, runs ``n_sweeps`` sweeps and returns ``self``.
    """

    def __init__(self, size, coupling: float = 1.0, temperature: float = 1.0, periodic: bool = True,
                 external_field: float = 0.0, seed: Optional[int] = None, bias_mode: str = "physical",
                 initial: str = "random"):
        if temperature <= 0:
            raise ValueError("Temperature must be positive")
        self.rows, self.cols = (size, size) if np.isscalar(size) else tuple(size)
        self.n_spins = self.rows * self.cols
        self.coupling = float(coupling)
        self.temperature = float(temperature)
        self.external_field = float(external_field)
        self.periodic = bool(periodic)
        self.bias_mode = bias_mode
        self._mode = _mode_id(bias_mode)
        self.seed = int(seed) if seed is not None else (
            int(np.random.randint(0, 2 ** 31 - 1)) | (int(np.random.randint(0, 2 ** 31 - 1)) << 31))
        self.sweep_count = 0
        self._lat = _hip.Lattice(self.rows, self.cols, self.periodic)
        if initial == "random":
            self._lat.randomize(self.seed)
        elif initial in ("up", "down"):
            self._lat.fill(1 if initial == "up" else -1)
        else:
            raise ValueError("initial must be 'random', 'up' or 'down'")

    def gibbs_update(self, n_sweeps: int = 1) -> "IsingModel2D":
        self._lat.set_model(self.coupling, self.external_field, self.temperature, self._mode)
        self._lat.sweep(int(n_sweeps), self.seed, self.sweep_count)
        self.sweep_count += int(n_sweeps)
        return self

    def equilibrate(self, temperature: Optional[float] = None, n_sweeps: int = 1000) -> "IsingModel2D":
        if temperature is not None:
            if temperature <= 0:
                raise ValueError("Temperature must be positive")
            self.temperature = float(temperature)
        return self.gibbs_update(n_sweeps)

    def magnetization(self) -> float:
        sum_s, _ = self._lat.observables()
        return sum_s / self.n_spins

    def energy(self) -> float:
        sum_s, sum_bonds = self._lat.observables()
        return -self.coupling * float(sum_bonds) - self.external_field * float(sum_s)

    @property
    def spins(self) -> np.ndarray:
        return self._lat.get_spins()

    @spins.setter
    def spins(self, value):
        v = np.asarray(value).reshape(self.rows, self.cols)
        if not np.all((v == 1) | (v == -1)):
            raise ValueError("spins must be +1 / -1")
        self._lat.set_spins(v.astype(np.int8))


def demonstrate_phase_transition(sizes: List[int] = [8, 16, 32], temperatures: Optional[np.ndarray] = None) -> dict:
    """Reference: ising.py:424-476 -- |M|, chi and C over a temperature scan for several lattice sizes."""
    if temperatures is None:
        temperatures = np.linspace(0.5, 4.0, 15)
    results = {}
    for size in sizes:
        print(f"\nSimulating {size}×{size} Ising grid...")
        magnetizations, susceptibilities, specific_heats = [], [], []
        for T in temperatures:
            model = IsingGrid((size, size), J=1.0, config=IsingConfig(temperature=T, n_burnin=200, n_sweeps=10))
            samples = model.sample(n_samples=500)
            mag = abs(model.magnetization(samples))
            chi = model.susceptibility(samples)
            C = model.specific_heat(samples)
            magnetizations.append(mag)
            susceptibilities.append(chi)
            specific_heats.append(C)
            print(f"  T={T:.2f}: |M|={mag:.3f}, χ={chi:.3f}, C={C:.3f}")
        results[size] = {
            "temperatures": temperatures,
            "magnetizations": np.array(magnetizations),
            "susceptibilities": np.array(susceptibilities),
            "specific_heats": np.array(specific_heats),
        }
    return results


def temperature_scan(size, temperatures, coupling: float = 1.0, n_equilibrate: int = 1000, n_measure: int = 50,
                     measure_every: int = 10, periodic: bool = True, seed: int = 0, bias_mode: str = "physical",
                     initial: str = "up") -> dict:
    """GPU-resident form of :func:`demonstrate_phase_transition` (reference: ising.py:424-476) for lattices far
    beyond what a samples array can hold: one :class:`IsingModel2D` per temperature stays on the device, and
    |M|, E/N, chi = (<M^2> - <M>^2) N / T and C = (<E^2> - <E>^2) / (T^2 N) come from the device reductions
    (``tsu_ising2d_observables``) -- no spin ever crosses PCIe.  Returns arrays indexed like ``temperatures``.
    """
    temperatures = np.asarray(temperatures, dtype=float)
    out = {k: np.zeros(len(temperatures)) for k in ("magnetization", "energy", "susceptibility", "specific_heat")}
    out["temperatures"] = temperatures
    # all temperatures advance together: lattices small enough for the one-workgroup kernel share ONE launch per
    # batch of sweeps (one workgroup per temperature) and one synchronisation per measurement; the streams (seed + i,
    # own sweep counter) and hence the results are those of sweeping the models one after the other
    models = [IsingModel2D(size, coupling=coupling, temperature=float(T), periodic=periodic, seed=seed + i,
                           bias_mode=bias_mode, initial=initial) for i, T in enumerate(temperatures)]

    def advance(n_sweeps):
        for m in models:
            m._lat.set_model(m.coupling, m.external_field, m.temperature, m._mode)
        _hip.sweep_batch([m._lat for m in models], n_sweeps, [m.seed for m in models], [m.sweep_count for m in models])
        for m in models:
            m.sweep_count += int(n_sweeps)

    if models:
        advance(int(n_equilibrate))
    Ms, Es = np.zeros((len(models), n_measure)), np.zeros((len(models), n_measure))
    for j in range(n_measure if models else 0):
        advance(int(measure_every))
        for i, (sum_s, sum_bonds) in enumerate(_hip.observables_batch([m._lat for m in models])):
            Ms[i, j] = sum_s / models[i].n_spins
            Es[i, j] = -models[i].coupling * float(sum_bonds) - models[i].external_field * float(sum_s)
    for i, T in enumerate(temperatures):
        N = models[i].n_spins
        out["magnetization"][i] = np.mean(np.abs(Ms[i]))
        out["energy"][i] = np.mean(Es[i]) / N
        out["susceptibility"][i] = (np.mean(Ms[i] ** 2) - np.mean(np.abs(Ms[i])) ** 2) * N / T
        out["specific_heat"][i] = (np.mean(Es[i] ** 2) - np.mean(Es[i]) ** 2) / (T ** 2 * N)
    del models
    return out
