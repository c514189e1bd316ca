"""Gibbs sampling on the MI355X -- drop-in for the reference's ``tsu.gibbs`` (same names, arguments,
return types and error messages; reference file:line cited per symbol).

Every sweep-level entry point (``gibbs_sweep``, ``sample_boltzmann``, ``compute_energy`` and the host loops
built on them) runs the hand-written HIP kernels of ``libtsu_hip.so`` through ``tsu._hip``; there is no CPU
fallback for them.  Only the scalar single-site helpers the reference's unit tests poke at (``_sigmoid``,
``_compute_local_field``, ``sample_conditional``) are plain host arithmetic.

Randomness.  The reference consumes the process-global ``np.random`` stream (one ``rand()`` per visited
site, gibbs.py:126).  Two modes are offered:

* ``rng="philox"`` (default): counter-based Philox4x32-10 on the device, keyed by (seed, site, sweep).  The
  seed is drawn from ``np.random`` on first use, so ``np.random.seed`` still makes runs reproducible.
* ``rng="numpy"``: the uniforms (and permutations) are drawn from ``np.random`` on the host in exactly the
  reference's order and replayed by the kernel, which reproduces the reference's trajectories bit for bit
  for the same ``np.random.seed`` (tests/golden/g1, g2).
"""
from dataclasses import dataclass
from typing import List, Optional, Tuple

import numpy as np

from . import _hip


def _is_sparse(a) -> bool:
    """scipy.sparse matrices / arrays select the colour-parallel sparse kernel (K5); everything else is a dense J."""
    return hasattr(a, "tocsr") and hasattr(a, "nnz")


@dataclass
class GibbsConfig:
    """Reference: tsu/gibbs.py:19-36 (same fields, defaults and ValueError messages)."""

    temperature: float = 1.0
    n_burnin: int = 100
    n_sweeps: int = 10
    update_order: str = "sequential"  # 'sequential' or 'random'

    def __post_init__(self):
        if self.temperature <= 0:
            raise ValueError("Temperature must be positive")
        if self.n_burnin < 0:
            raise ValueError("Burn-in steps must be non-negative")
        if self.n_sweeps <= 0:
            raise ValueError("Number of sweeps must be positive")
        if self.update_order not in ["sequential", "random"]:
            raise ValueError("Update order must be 'sequential' or 'random'")


_HASH_CHUNK = 1 << 24        # bytes per task of the parallel content hash
_HASH_PARALLEL_MIN = 1 << 25  # buffers from 32 MiB on are hashed by a thread pool
_hash_pool = None


def _hash_bytes(buf) -> bytes:
    try:
        import xxhash
        return xxhash.xxh3_128_digest(buf)
    except ImportError:  # pragma: no cover - xxhash ships with the image
        import hashlib
        return hashlib.blake2b(buf, digest_size=16).digest()


def _content_key(a: Optional[np.ndarray]):
    """Exact content key of a host array (shape, dtype, 128-bit hash of EVERY byte): decides whether the device copy of
    J / bias made by an earlier call may be reused.  The reference reads ``coupling`` afresh at every site
    (gibbs.py:97), so any in-place edit between two calls must be seen: the whole buffer is hashed, O(n^2) bytes once
    per call next to the O(n^2) work per sweep.  Large buffers are hashed in 16 MiB chunks on a thread pool (xxh3 releases
    the GIL; the key is the hash of the chunk digests): the 2 GiB of an N = 16384 float64 J take 0.33 s on one core and
    1 / cores of that on the pool -- still two orders of magnitude more than a sweep, which is why callers with a hot loop
    over one unchanged J either freeze it (``J.setflags(write=False)``: see :meth:`GibbsSampler._system`) or
    :meth:`GibbsSampler.bind` it."""
    if a is None:
        return None
    a = np.ascontiguousarray(a)
    buf = a.reshape(-1).view(np.uint8) if a.dtype != object else None
    if buf is None:
        raise TypeError("coupling / bias must be numeric arrays")
    if buf.nbytes >= _HASH_PARALLEL_MIN:
        global _hash_pool
        if _hash_pool is None:
            import os
            from concurrent.futures import ThreadPoolExecutor
            _hash_pool = ThreadPoolExecutor(max_workers=max(2, min(32, len(os.sched_getaffinity(0)))), thread_name_prefix="tsu-hash")
        parts = list(_hash_pool.map(_hash_bytes, [buf[o:o + _HASH_CHUNK] for o in range(0, buf.nbytes, _HASH_CHUNK)]))
        digest = _hash_bytes(b"".join(parts))
    else:
        digest = _hash_bytes(buf)
    return (a.shape, a.dtype.str, digest)


def _is_frozen(a) -> bool:
    """A NumPy array whose bytes cannot be edited in place by anybody: it is not writeable and owns its buffer (no writeable base
    or sibling view can reach it)."""
    return a is None or (isinstance(a, np.ndarray) and not a.flags.writeable and a.flags.owndata)


def _sample_key(a):
    """64 windows of 256 B spread evenly over the buffer, hashed: a tripwire for the one way a frozen array can still change (its
    owner makes it writeable, edits it and freezes it again between two calls): bulk edits are caught, single entries are the
    owner's responsibility (documented in :meth:`GibbsSampler._system`).  One strided gather of 16 KiB; what it costs is the 64
    pages it touches (windows of 4 KiB made it 37 us per call, a quarter of the unbound step at N = 4096)."""
    if a is None:
        return None
    buf = np.ascontiguousarray(a).reshape(-1).view(np.uint8)
    if buf.nbytes <= (1 << 16):
        return _hash_bytes(buf)
    words = buf[:buf.nbytes // 8 * 8].view(np.uint64)  # (copied 8 bytes at a time: a byte-wise strided copy of 16 KiB takes 25 us)
    step = words.size // 64
    return _hash_bytes(words[:64 * step].reshape(64, step)[:, :32].tobytes())  # (the first 256 B of each sixty-fourth of the buffer)


class GibbsSampler:
    """Reference: tsu/gibbs.py:39-393.  ``GibbsSampler(config)`` as in the reference; keyword-only extras
    select the random stream (``rng``), the Philox seed and the dtype J is stored in on the device."""

    def __init__(self, config: Optional[GibbsConfig] = None, *, rng: str = "philox", seed: Optional[int] = None,
                 coupling_dtype: str = "float64"):
        if rng not in ("philox", "numpy"):
            raise ValueError("rng must be 'philox' or 'numpy'")
        if coupling_dtype not in ("float64", "float32"):
            raise ValueError("coupling_dtype must be 'float64' or 'float32'")
        self.config = config or GibbsConfig()
        self.sample_count = 0
        self.rng = rng
        self._seed = None if seed is None else int(seed)
        self._sweep_counter = 0
        self._dtype = _hip.DTYPE_F64 if coupling_dtype == "float64" else _hip.DTYPE_F32
        self._bound = None  # (content key of J, content key of bias, DenseSystem)
        self._held = None   # (J object, bias object): arrays whose device copy is reused WITHOUT a content check
        self._frozen = None  # (J object, bias object, sample keys): frozen arrays the device copy was made from (see _system)
        self._bound_sparse = None  # (content key of the CSR arrays and bias, SparseSystem)

    # ------------------------------------------------------------------ scalar helpers (host)
    def _sigmoid(self, x: float) -> float:
        """Reference: tsu/gibbs.py:61-77 (hard clamp beyond +-20)."""
        if x > 20:
            return 1.0
        elif x < -20:
            return 0.0
        return 1.0 / (1.0 + np.exp(-x))

    def _compute_local_field(self, i: int, state: np.ndarray, coupling: np.ndarray,
                             bias: Optional[np.ndarray] = None) -> float:
        """Reference: tsu/gibbs.py:79-100 -- h_i = J[i,:].s (+ b_i), diagonal term included."""
        h = np.dot(coupling[i, :], state)
        if bias is not None:
            h += bias[i]
        return float(h)

    def sample_conditional(self, i: int, state: np.ndarray, coupling: np.ndarray,
                           bias: Optional[np.ndarray] = None) -> int:
        """Reference: tsu/gibbs.py:102-126 -- one site, one ``np.random.rand()`` (strict ``<``)."""
        h_i = self._compute_local_field(i, state, coupling, bias)
        prob = self._sigmoid(h_i / self.config.temperature)
        return 1 if np.random.rand() < prob else 0

    # ------------------------------------------------------------------ device plumbing
    def _philox_seed(self) -> int:
        if self._seed is None:
            self._seed = int(np.random.randint(0, 2 ** 31 - 1)) | (int(np.random.randint(0, 2 ** 31 - 1)) << 31)
        return self._seed

    def invalidate(self):
        """Drop the device copy of the coupling matrix (and any :meth:`bind`)."""
        if self._bound is not None:
            self._bound[2].close()
        self._bound = None
        self._held = None
        self._frozen = None
        if self._bound_sparse is not None:
            self._bound_sparse[1].close()
            self._bound_sparse = None

    def bind(self, coupling: np.ndarray, bias: Optional[np.ndarray] = None) -> "GibbsSampler":
        """Opt-in caching for hot loops: upload ``coupling`` / ``bias`` now and, until :meth:`unbind` (or a call with
        other array objects), reuse the device copy for calls that pass these very objects WITHOUT re-reading their
        contents.  The caller promises not to edit them in place meanwhile; without ``bind`` every call hashes the
        whole buffer and sees any edit, as the reference does by reading ``coupling`` at every site."""
        self._held = None
        self._system(coupling, bias)
        self._held = (coupling, bias)
        return self

    def unbind(self):
        self._held = None

    def _system(self, coupling: np.ndarray, bias: Optional[np.ndarray]) -> "_hip.DenseSystem":
        """The device copy of (coupling, bias) to use for this call.  Three cases, from cheap to dear:

        * :meth:`bind` named these very objects: reuse, no check (the caller's promise);
        * the arrays are FROZEN -- ``coupling.setflags(write=False)`` on an array that owns its data (bias likewise or None) -- and
          are the objects the device copy was made from: nobody can edit them in place, so the copy is current; O(1) plus a
          tripwire of 64 sampled windows.  (The owner could make the array writeable, edit and freeze it again between two calls: a bulk
          edit trips the sample, a single entry does not -- whoever thaws an array calls :meth:`invalidate`.)  This is how the
          reference idiom ``state = s.gibbs_sweep(state, J)`` loops at kernel speed without any API the reference lacks;
        * otherwise every byte is hashed (the reference reads ``coupling`` afresh at every site, gibbs.py:97: an in-place edit
          between two calls must be seen), in parallel from 32 MiB on."""
        if self._held is not None and self._bound is not None:
            if coupling is self._held[0] and bias is self._held[1]:
                return self._bound[2]
        if self._frozen is not None and self._bound is not None:
            if coupling is self._frozen[0] and bias is self._frozen[1] and _is_frozen(coupling) and _is_frozen(bias):
                if _sample_key(coupling) == self._frozen[2] and _sample_key(bias) == self._frozen[3]:
                    return self._bound[2]
        self._frozen = None
        coupling_in, bias_in = coupling, bias
        coupling = np.asarray(coupling)
        if coupling.ndim != 2 or coupling.shape[0] != coupling.shape[1]:
            raise ValueError("Coupling matrix must be square")
        fj, fb = _content_key(coupling), _content_key(None if bias is None else np.asarray(bias))
        if not (self._bound is not None and self._bound[0] == fj and self._bound[1] == fb):
            self.invalidate()
            self._bound = (fj, fb, _hip.DenseSystem(coupling, bias, self._dtype))
        if isinstance(coupling_in, np.ndarray) and _is_frozen(coupling_in) and _is_frozen(bias_in):
            self._frozen = (coupling_in, bias_in, _sample_key(coupling_in), _sample_key(bias_in))
        return self._bound[2]

    @staticmethod
    def _as_bits(state: np.ndarray, n: int) -> np.ndarray:
        s = np.asarray(state)
        if s.shape != (n,):
            raise ValueError(f"state must have shape ({n},)")
        if not np.all((s == 0) | (s == 1)):
            raise ValueError("state must be binary (0/1)")
        return s.astype(np.int8)

    def _run_sweeps(self, sys: "_hip.DenseSystem", n_sweeps: int):
        """n_sweeps sweeps of the resident state at the CURRENT config.temperature (read at call time:
        callers mutate it in place, gibbs.py:382)."""
        if n_sweeps <= 0:
            return
        n, T = sys.n, float(self.config.temperature)
        if T <= 0:
            raise ValueError("Temperature must be positive")
        random_order = self.config.update_order == "random"
        if self.rng == "numpy":
            # the reference's draw order per sweep: [permutation(n)] then one rand() per visited site
            order = np.empty((n_sweeps, n), dtype=np.int64) if random_order else None
            uni = np.empty((n_sweeps, n), dtype=np.float64)
            for s in range(n_sweeps):
                if random_order:
                    order[s] = np.random.permutation(n)
                uni[s] = np.random.rand(n)
            sys.sweep(T, n_sweeps, order=order, replay_uniforms=uni)
        else:
            order = np.array([np.random.permutation(n) for _ in range(n_sweeps)]) if random_order else None
            sys.sweep(T, n_sweeps, seed=self._philox_seed(), sweep0=self._sweep_counter, order=order)
            self._sweep_counter += n_sweeps

    # ------------------------------------------------------------------ sweep-level API (GPU)
    def gibbs_sweep(self, state: np.ndarray, coupling: np.ndarray, bias: Optional[np.ndarray] = None,
                    n_sweeps: int = 1) -> np.ndarray:
        """Reference: tsu/gibbs.py:128-162.  Returns a NEW array of the input dtype; the input is not modified.

        ``coupling`` may be a ``scipy.sparse`` matrix: the sweep then runs on the colour-parallel sparse kernel, i.e. the
        reference's sequential loop in the colour-major visiting order of a proper colouring of the graph."""
        state = np.asarray(state)
        if _is_sparse(coupling):
            sys = self._sparse_system(coupling, bias)
            sys.set_state(self._as_bits(state, sys.n))
            self._run_sparse(sys, int(n_sweeps))
            return sys.get_state().astype(state.dtype)
        sys = self._system(coupling, bias)
        sys.set_state(self._as_bits(state, sys.n))
        self._run_sweeps(sys, int(n_sweeps))
        return sys.get_state().astype(state.dtype)

    # ------------------------------------------------------------------ sparse graphs (K5)
    def _sparse_system(self, coupling, bias) -> "_hip.SparseSystem":
        from .graph import canonical_csr, color_graph
        if self.rng != "philox":
            raise ValueError("sparse couplings run in colour-parallel order: rng must be 'philox'")
        if self.config.update_order != "sequential":
            raise ValueError("sparse couplings run in colour-parallel order: update_order must be 'sequential'")
        if coupling.shape[0] != coupling.shape[1]:
            raise ValueError("Coupling matrix must be square")
        A = canonical_csr(coupling)
        b = None if bias is None else np.ascontiguousarray(bias, dtype=np.float64)
        key = (A.shape, _content_key(A.indptr), _content_key(A.indices), _content_key(A.data), _content_key(b))
        if self._bound_sparse is not None and self._bound_sparse[0] == key:
            return self._bound_sparse[1]
        if self._bound_sparse is not None:
            self._bound_sparse[1].close()
            self._bound_sparse = None
        offsets, order = color_graph(A)
        sys = _hip.SparseSystem(A.indptr, A.indices, A.data, b, offsets, order)
        self._bound_sparse = (key, sys)
        return sys

    def _run_sparse(self, sys: "_hip.SparseSystem", n_sweeps: int):
        if n_sweeps <= 0:
            return
        T = float(self.config.temperature)
        if T <= 0:
            raise ValueError("Temperature must be positive")
        sys.sweep(T, n_sweeps, seed=self._philox_seed(), sweep0=self._sweep_counter)
        self._sweep_counter += n_sweeps

    def _sample_sparse(self, coupling, bias, n_samples, burnin, initial_state, dtype=int) -> np.ndarray:
        n_bits = coupling.shape[0]
        sys = self._sparse_system(coupling, bias)
        burnin = burnin if burnin is not None else self.config.n_burnin
        state = np.asarray(initial_state).copy() if initial_state is not None else np.random.randint(0, 2, size=n_bits)
        sys.set_state(self._as_bits(state, n_bits))
        T = float(self.config.temperature)
        if T <= 0:
            raise ValueError("Temperature must be positive")
        n_sweeps = int(self.config.n_sweeps)
        samples = np.zeros((n_samples, n_bits), dtype=dtype)
        chunk = max(1, min(int(n_samples), (1 << 30) // max(1, n_bits)))  # <= 1 GiB of recorded states per call
        done, burn = 0, int(burnin)
        if n_samples == 0:
            self._run_sparse(sys, burn)
        while done < n_samples:
            m = min(chunk, n_samples - done)
            samples[done:done + m] = sys.sample(T, burn, n_sweeps, m, seed=self._philox_seed(), sweep0=self._sweep_counter)
            self._sweep_counter += burn + m * n_sweeps
            self.sample_count += m
            done += m
            burn = 0
        return samples

    def sample_boltzmann(self, coupling: np.ndarray, bias: Optional[np.ndarray] = None, n_samples: int = 1000,
                         burnin: Optional[int] = None, initial_state: Optional[np.ndarray] = None) -> np.ndarray:
        """Reference: tsu/gibbs.py:164-213.  Returns ``(n_samples, n_bits)`` int array of 0/1."""
        if _is_sparse(coupling):
            if coupling.shape[0] != coupling.shape[1]:
                raise ValueError("Coupling matrix must be square")
            return self._sample_sparse(coupling, bias, n_samples, burnin, initial_state)
        coupling = np.asarray(coupling)
        n_bits = coupling.shape[0]
        if coupling.shape != (n_bits, n_bits):
            raise ValueError("Coupling matrix must be square")
        burnin = burnin if burnin is not None else self.config.n_burnin
        if initial_state is not None:
            state = np.asarray(initial_state).copy()
        else:
            state = np.random.randint(0, 2, size=n_bits)
        sys = self._system(coupling, bias)
        sys.set_state(self._as_bits(state, n_bits))
        samples = np.zeros((n_samples, n_bits), dtype=int)
        # the whole run (burn-in, then n_samples x n_sweeps sweeps) in as few device calls as the draw buffers allow
        n_sweeps = int(self.config.n_sweeps)
        chunk = max(1, min(int(n_samples), (1 << 23) // max(1, n_sweeps * n_bits)))  # <= 64 MiB of replayed doubles per call
        done, burn = 0, int(burnin)
        if n_samples == 0:
            self._run_sweeps(sys, burn)
        while done < n_samples:
            m = min(chunk, n_samples - done)
            samples[done:done + m] = self._run_sampling(sys, burn, n_sweeps, m)
            self.sample_count += m
            done += m
            burn = 0
        return samples

    def _run_sampling(self, sys: "_hip.DenseSystem", n_burnin: int, n_sweeps: int, n_samples: int) -> np.ndarray:
        """n_burnin sweeps, then n_samples x (n_sweeps sweeps, record) at the current temperature: one C call."""
        n, T = sys.n, float(self.config.temperature)
        if T <= 0:
            raise ValueError("Temperature must be positive")
        total = n_burnin + n_samples * n_sweeps
        random_order = self.config.update_order == "random"
        if self.rng == "numpy":
            # the reference's draw order (gibbs.py:152-160): per sweep [permutation(n)] then one rand() per visited site
            if random_order:
                order = np.empty((total, n), dtype=np.int64)
                uni = np.empty((total, n), dtype=np.float64)
                for s in range(total):
                    order[s] = np.random.permutation(n)
                    uni[s] = np.random.rand(n)
            else:
                order, uni = None, np.random.rand(total, n)
            return sys.sample(T, n_burnin, n_sweeps, n_samples, order=order, replay_uniforms=uni)
        order = np.array([np.random.permutation(n) for _ in range(total)]).reshape(total, n) if random_order else None
        out = sys.sample(T, n_burnin, n_sweeps, n_samples, seed=self._philox_seed(), sweep0=self._sweep_counter, order=order)
        self._sweep_counter += total
        return out

    def sample_chains(self, coupling: np.ndarray, n_chains: int, n_samples_per_chain: int, bias: Optional[np.ndarray] = None,
                      burnin: Optional[int] = None) -> np.ndarray:
        """``n_chains`` consecutive ``sample_boltzmann(coupling, bias, n_samples_per_chain, burnin)`` calls of THIS sampler --
        the chain loop of ``HardwareEmulator.sample_parallel`` (reference: gibbs.py:470-478) -- with all chains advanced
        together on the device (one wave / workgroup per chain, ``tsu_dense_sweep_replicas``).  Same draws as the loop: every
        chain's initial state comes from ``np.random.randint`` in chain order, chain c uses the Philox sweep numbers the c-th
        call would have used.  Returns ``(n_chains, n_samples_per_chain, n_bits)``; ``sample_count`` advances as in the loop."""
        coupling = np.asarray(coupling)
        n_bits = coupling.shape[0]
        if coupling.shape != (n_bits, n_bits):
            raise ValueError("Coupling matrix must be square")
        burnin = int(self.config.n_burnin if burnin is None else burnin)
        n_sweeps, T = int(self.config.n_sweeps), float(self.config.temperature)
        if self.rng != "philox" or self.config.update_order != "sequential" or n_chains < 2:
            return np.array([self.sample_boltzmann(coupling, bias, n_samples_per_chain, burnin) for _ in range(n_chains)])
        if T <= 0:
            raise ValueError("Temperature must be positive")
        per_chain = burnin + n_samples_per_chain * n_sweeps
        if self._sweep_counter + n_chains * per_chain > 2 ** 32:  # the message of the C ABI's check on the one-by-one path
            raise ValueError("dense_sample: sweep counter overflow")
        sys = self._system(coupling, bias)
        seed = self._philox_seed()
        out = np.zeros((n_chains, n_samples_per_chain, n_bits), dtype=int)
        batch = max(1, min(n_chains, (1 << 26) // max(1, n_bits)))  # <= 64 MiB of chain states in flight
        for c0 in range(0, n_chains, batch):
            m = min(batch, n_chains - c0)
            states = np.array([np.random.randint(0, 2, size=n_bits) for _ in range(m)], dtype=np.int8)
            base = np.array([self._sweep_counter + (c0 + c) * per_chain for c in range(m)], dtype=np.uint32)
            temps, seeds = [T] * m, [seed] * m
            if burnin:
                states = sys.sweep_replicas(states, temps, burnin, seeds, base)
            for k in range(n_samples_per_chain):
                states = sys.sweep_replicas(states, temps, n_sweeps, seeds, base + np.uint32(burnin + k * n_sweeps))
                out[c0:c0 + m, k] = states
        self._sweep_counter += n_chains * per_chain
        self.sample_count += n_chains * n_samples_per_chain
        return out

    def sample(self, J: np.ndarray, n_samples: int = 1000, bias: Optional[np.ndarray] = None) -> np.ndarray:
        """README name (README.md:79) for :meth:`sample_boltzmann`."""
        return self.sample_boltzmann(J, bias=bias, n_samples=n_samples)

    def compute_energy(self, state: np.ndarray, coupling: np.ndarray, bias: Optional[np.ndarray] = None) -> float:
        """Reference: tsu/gibbs.py:215-236 -- E = -1/2 s^T J s - b^T s (device matvec + reduction)."""
        if _is_sparse(coupling):
            sys = self._sparse_system(coupling, bias)
            sys.set_state(self._as_bits(state, sys.n))
            return float(sys.energy()[0])
        sys = self._system(coupling, bias)
        sys.set_state(self._as_bits(state, sys.n))
        return float(sys.energy())

    # ------------------------------------------------------------------ host loops around the sweep
    def parallel_tempering(self, coupling: np.ndarray, temperatures: List[float], bias: Optional[np.ndarray] = None,
                           n_samples: int = 1000, swap_interval: int = 10) -> Tuple[np.ndarray, dict]:
        """Reference: tsu/gibbs.py:238-338 (replica exchange; swap rule :317-323)."""
        n_replicas = len(temperatures)
        coupling = np.asarray(coupling)
        n_bits = coupling.shape[0]
        bias_arr = None if bias is None else np.asarray(bias)
        if n_bits <= self._ANNEAL_HOST_ENERGY_MAX:
            # small systems: the reference's own expression on the host instead of a device round trip per energy
            def energy_of(s):
                return self._energy_host(np.asarray(s), coupling, bias_arr)
        else:
            def energy_of(s):
                return self.compute_energy(s, coupling, bias)
        states = [np.random.randint(0, 2, size=n_bits) for _ in range(n_replicas)]
        samplers = []
        for T in temperatures:
            cfg = GibbsConfig(temperature=T, n_burnin=self.config.n_burnin, n_sweeps=self.config.n_sweeps,
                              update_order=self.config.update_order)
            rep = GibbsSampler(cfg, rng=self.rng, seed=None if self.rng == "numpy" else self._philox_seed() + len(samplers) + 1,
                               coupling_dtype="float64" if self._dtype == _hip.DTYPE_F64 else "float32")
            samplers.append(rep)
        # J is hashed and uploaded once per sampler for the whole run: no caller code runs inside this call, so the
        # arrays cannot change under it (restored / dropped at the end)
        held_before = self._held
        self.bind(coupling, bias)
        if self.config.update_order != "sequential":
            for rep in samplers:
                rep.bind(coupling, bias)
        try:
            return self._parallel_tempering_run(samplers, states, coupling, bias, temperatures, n_samples, swap_interval,
                                                energy_of)
        finally:
            # bind() above replaced the device copy by this call's J: an earlier bind is only still served by it when it named
            # these very arrays; any other earlier bind is dropped (the next call takes the content-hash path and uploads again)
            if held_before is not None and held_before[0] is coupling and held_before[1] is bias:
                self._held = held_before
            else:
                self._held = None
            for rep in samplers:
                rep.invalidate()

    def _parallel_tempering_run(self, samplers, states, coupling, bias, temperatures, n_samples, swap_interval, energy_of):
        n_replicas = len(temperatures)
        states = self._sweep_replicas(samplers, states, coupling, bias, self.config.n_burnin)
        samples = []
        swap_attempts = 0
        swap_accepts = 0
        energies_history = [[] for _ in range(n_replicas)]
        sweep_count = 0
        while len(samples) < n_samples:
            states = self._sweep_replicas(samplers, states, coupling, bias, self.config.n_sweeps)
            for i in range(n_replicas):
                energies_history[i].append(energy_of(states[i]))
            sweep_count += 1
            if sweep_count % swap_interval == 0:
                for i in range(n_replicas - 1):
                    E_i = energy_of(states[i])
                    E_j = energy_of(states[i + 1])
                    delta = (1.0 / temperatures[i] - 1.0 / temperatures[i + 1]) * (E_j - E_i)
                    swap_attempts += 1
                    if delta >= 0 or np.random.rand() < np.exp(delta):
                        states[i], states[i + 1] = states[i + 1], states[i]
                        swap_accepts += 1
            samples.append(states[0].copy())
        samples = np.array(samples[:n_samples])
        info = {
            "swap_acceptance_rate": swap_accepts / swap_attempts if swap_attempts > 0 else 0,
            "swap_attempts": swap_attempts,
            "swap_accepts": swap_accepts,
            "energies": energies_history,
            "final_states": states,
        }
        return samples, info

    def _sweep_replicas(self, samplers, states, coupling, bias, n_sweeps):
        """The replica loop of parallel_tempering (reference: gibbs.py:300-306): every replica's sweeps in ONE device
        call in sequential order (one wave per replica up to 192 / 128 sites, one workgroup per replica up to 576 / 448, larger
        systems replica after replica on the device side, all from ONE device copy of J); random order: replica by replica.
        Draws (np.random uniforms / each replica's Philox counters) are consumed exactly as in the one-by-one loop."""
        n_sweeps = int(n_sweeps)
        n_bits = np.asarray(coupling).shape[0]
        if n_sweeps <= 0:
            return states
        if self.config.update_order != "sequential":
            return [s.gibbs_sweep(st, coupling, bias, n_sweeps=n_sweeps) for s, st in zip(samplers, states)]
        sys = self._system(coupling, bias)
        temps = [float(s.config.temperature) for s in samplers]
        for T in temps:
            if T <= 0:
                raise ValueError("Temperature must be positive")
        bits = np.array([self._as_bits(st, n_bits) for st in states], dtype=np.int8)
        if self.rng == "numpy":
            uni = np.array([np.random.rand(n_sweeps, n_bits) for _ in samplers])
            out = sys.sweep_replicas(bits, temps, n_sweeps, [0] * len(samplers), [0] * len(samplers), replay_uniforms=uni)
        else:
            out = sys.sweep_replicas(bits, temps, n_sweeps, [s._philox_seed() for s in samplers], [s._sweep_counter for s in samplers])
            for s in samplers:
                s._sweep_counter += n_sweeps
        return [out[i].astype(np.asarray(states[i]).dtype) for i in range(len(samplers))]

    def simulated_annealing(self, coupling: np.ndarray, bias: Optional[np.ndarray] = None, T_initial: float = 10.0,
                            T_final: float = 0.1, n_steps: int = 1000,
                            cooling_schedule: str = "exponential") -> Tuple[np.ndarray, float]:
        """Reference: tsu/gibbs.py:340-393 (mutates ``self.config.temperature`` in place, as the reference does)."""
        if _is_sparse(coupling):
            return self._anneal_sparse(coupling, bias, T_initial, T_final, n_steps, cooling_schedule)
        coupling = np.asarray(coupling)
        n_bits = coupling.shape[0]
        state = np.random.randint(0, 2, size=n_bits)
        if cooling_schedule == "exponential":
            temps = [T_initial * (T_final / T_initial) ** (step / n_steps) for step in range(n_steps)]
        else:  # linear
            temps = [T_initial + (T_final - T_initial) * step / n_steps for step in range(n_steps)]
        if n_bits > self._ANNEAL_HOST_ENERGY_MAX:
            return self._anneal_dense_large(coupling, bias, state, temps)
        # the whole schedule in one device call (one sweep per temperature, every state recorded); the energies and the
        # running minimum are evaluated here with the reference's own expression (gibbs.py:233-236), so that ties
        # between equal-energy states break as they do there
        bias_arr = None if bias is None else np.asarray(bias)
        best_state = state.copy()
        best_energy = self._energy_host(state, coupling, bias_arr)
        if n_steps > 0:
            for T in temps:
                if T <= 0:
                    raise ValueError("Temperature must be positive")
            sys = self._system(coupling, bias)
            sys.set_state(self._as_bits(state, n_bits))
            random_order = self.config.update_order == "random"
            if self.rng == "numpy":
                if random_order:
                    order = np.empty((n_steps, n_bits), dtype=np.int64)
                    uni = np.empty((n_steps, n_bits), dtype=np.float64)
                    for s_ in range(n_steps):
                        order[s_] = np.random.permutation(n_bits)
                        uni[s_] = np.random.rand(n_bits)
                else:
                    order, uni = None, np.random.rand(n_steps, n_bits)
                states = sys.anneal(temps, order=order, replay_uniforms=uni)
            else:
                order = np.array([np.random.permutation(n_bits) for _ in range(n_steps)]).reshape(n_steps, n_bits) if random_order else None
                states = sys.anneal(temps, seed=self._philox_seed(), sweep0=self._sweep_counter, order=order)
                self._sweep_counter += n_steps
            self.config.temperature = temps[-1]
            for k in range(n_steps):
                cand = states[k].astype(state.dtype)
                energy = self._energy_host(cand, coupling, bias_arr)
                if energy < best_energy:
                    best_energy = energy
                    best_state = cand
        return best_state, best_energy

    def _anneal_dense_large(self, coupling, bias, state, temps):
        """simulated_annealing above the size whose energies are cheap on the host: the schedule runs in device calls of up to
        2**26 / n steps (each ONE launch: a temperature per sweep, every state recorded in the kernel), the energies of the recorded
        states come from one ``tsu_dense_energies`` call per chunk, and the running minimum keeps the first state of the lowest
        energy, as the reference's ``energy < best_energy`` does (gibbs.py:384-391).  J is hashed / uploaded once."""
        n_bits = coupling.shape[0]
        for T in temps:
            if T <= 0:
                raise ValueError("Temperature must be positive")
        sys = self._system(coupling, bias)
        sys.set_state(self._as_bits(state, n_bits))
        best_state, best_energy = state.copy(), float(sys.energy())
        random_order = self.config.update_order == "random"
        chunk = max(1, min(len(temps), (1 << 26) // n_bits))
        for c0 in range(0, len(temps), chunk):
            tt = temps[c0:c0 + chunk]
            m = len(tt)
            order = np.array([np.random.permutation(n_bits) for _ in range(m)]).reshape(m, n_bits) if random_order and self.rng != "numpy" else None
            if self.rng == "numpy":
                if random_order:  # the reference draws a permutation, then the sweep's uniforms, step by step
                    order = np.empty((m, n_bits), dtype=np.int64)
                    uni = np.empty((m, n_bits), dtype=np.float64)
                    for s_ in range(m):
                        order[s_] = np.random.permutation(n_bits)
                        uni[s_] = np.random.rand(n_bits)
                else:
                    uni = np.random.rand(m, n_bits)
                states = sys.anneal(tt, order=order, replay_uniforms=uni)
            else:
                states = sys.anneal(tt, seed=self._philox_seed(), sweep0=self._sweep_counter, order=order)
                self._sweep_counter += m
            energies = sys.energies(states)
            k = int(np.argmin(energies))  # the first of equal minima, like the strict comparison step by step
            if energies[k] < best_energy:
                best_energy = float(energies[k])
                best_state = states[k].astype(state.dtype)
        if temps:
            self.config.temperature = temps[-1]
        return best_state, best_energy

    def _anneal_sparse(self, coupling, bias, T_initial, T_final, n_steps, cooling_schedule):
        """simulated_annealing (gibbs.py:340-393) on a sparse graph: one colour-parallel sweep per temperature, the
        energy of every state by the device reduction, the best state kept."""
        n_bits = coupling.shape[0]
        state = np.random.randint(0, 2, size=n_bits)
        if cooling_schedule == "exponential":
            temps = [T_initial * (T_final / T_initial) ** (step / n_steps) for step in range(n_steps)]
        else:
            temps = [T_initial + (T_final - T_initial) * step / n_steps for step in range(n_steps)]
        sys = self._sparse_system(coupling, bias)
        sys.set_state(self._as_bits(state, n_bits))
        best_state, best_energy = state.copy(), sys.energy()[0]
        for T in temps:
            self.config.temperature = T
            self._run_sparse(sys, 1)
            energy = sys.energy()[0]
            if energy < best_energy:
                best_energy = energy
                best_state = sys.get_state().astype(state.dtype)
        return best_state, float(best_energy)

    _ANNEAL_HOST_ENERGY_MAX = 512  # systems up to this size: energies of the recorded states on the host

    @staticmethod
    def _energy_host(state: np.ndarray, coupling: np.ndarray, bias: Optional[np.ndarray]) -> float:
        energy = -0.5 * state.dot(coupling).dot(state)
        if bias is not None:
            energy -= bias.dot(state)
        return float(energy)


class HardwareEmulator:
    """Reference: tsu/gibbs.py:396-487 -- closed-form timing model plus a loop of independent chains."""

    def __init__(self, n_bits: int = 100, clock_speed_ghz: float = 1.0, parallel_chains: int = 1000):
        self.n_bits = n_bits
        self.clock_speed_ghz = clock_speed_ghz
        self.parallel_chains = parallel_chains
        self.ns_per_cycle = 1.0 / clock_speed_ghz

    def estimate_hardware_time(self, n_samples: int, n_sweeps_per_sample: int) -> dict:
        time_per_sweep_ns = self.n_bits * self.ns_per_cycle
        time_per_sample_ns = n_sweeps_per_sample * time_per_sweep_ns
        batches_needed = int(np.ceil(n_samples / self.parallel_chains))
        total_time_ns = batches_needed * time_per_sample_ns
        return {
            "time_per_sweep_ns": time_per_sweep_ns,
            "time_per_sample_ns": time_per_sample_ns,
            "batches_needed": batches_needed,
            "total_time_ns": total_time_ns,
            "total_time_us": total_time_ns / 1000,
            "total_time_ms": total_time_ns / 1e6,
            "total_time_s": total_time_ns / 1e9,
            "speedup_vs_classical": None,
        }

    def sample_parallel(self, coupling: np.ndarray, n_samples: int, temperature: float = 1.0) -> Tuple[np.ndarray, dict]:
        config = GibbsConfig(temperature=temperature)
        sampler = GibbsSampler(config)
        samples_per_chain = int(np.ceil(n_samples / self.parallel_chains))
        # the reference's loop of independent chains (gibbs.py:470-478), all chains advanced together on the device
        chains = sampler.sample_chains(coupling, min(self.parallel_chains, n_samples), samples_per_chain, burnin=100)
        samples = chains.reshape(-1, chains.shape[-1])[:n_samples]
        timing = self.estimate_hardware_time(n_samples, config.n_sweeps)
        return samples, timing
