"""Row-slab domain decomposition of the 2-D lattice across the GPUs of one node (one process per GPU).

The reference has no parallelism of any kind; this is new design for the path
``IsingGrid`` / ``IsingModel2D.gibbs_update`` at sizes one GPU cannot hold or should not sweep alone.

Rank r owns rows [r*R, (r+1)*R) of a (P*R) x cols lattice plus ``ghost = 2S`` ghost rows on each side.  Every S
sweeps (``sweeps_per_exchange``) the ranks exchange their 2S boundary rows with the rank above and below
(``torch.distributed`` point-to-point, i.e. RCCL send/recv over xGMI with the ``nccl`` backend): one exchange per
S sweeps instead of two per sweep, because each half-sweep consumes one ghost row.  S spans several generations of k
sweeps: between two exchanges a slab keeps its own halo exact by also sweeping the ghost rows the later generations
will read (2 rows per remaining sweep; a few % extra work), so the message latency is paid once per S sweeps, and --
when every tile of the slab has its own workgroup on the GPU, as for 4096 x 4096 per GPU -- the whole period is ONE
launch with the tiles resident in LDS (csrc/ising2d_tiled.hip, k1_resident).  Philox counters use
GLOBAL (row, column, sweep) coordinates, so the trajectory is bit-identical for every P (tested).

Optional overlap (``overlap=True``, S <= 8): the tile rows that do not touch ghost rows are swept on the compute
stream while the ghost rows travel on a second stream; the two boundary tile rows follow once the exchange has
landed (``tsu_ising2d_sweep_part``).  It only pays when a slab has several tiles per CU: at 4096 x 4096 per GPU
every launch lasts one tile-time however few tiles it has, so the default is the deep-ghost schedule above.
There is no collective on the sweep path; observables need one all-reduce of two int64.
"""
from typing import Optional

import numpy as np

from . import _hip


class _DeviceRows:
    """Zero-copy view of lattice rows in libtsu_hip's device memory for torch (``__cuda_array_interface__``)."""

    def __init__(self, ptr: int, n_rows: int, pitch: int):
        self.__cuda_array_interface__ = {"shape": (n_rows, pitch), "typestr": "|i1", "data": (int(ptr), False),
                                         "version": 3, "strides": None}


class SlabLattice:
    """A (world_size * rows_per_rank) x cols periodic or open lattice, one row slab per rank.

    ``engine`` is the per-slab lattice class (default: the HIP lattice :class:`tsu._hip.Lattice`; the CPU-only
    tests inject a test double, the product never does).  ``group``/``backend`` come from ``torch.distributed``;
    with ``world_size == 1`` no process group is needed and the exchange is a device-to-device copy.
    """

    def __init__(self, rows_per_rank: int, cols: int, periodic: bool = True, sweeps_per_exchange: int = 4,
                 seed: int = 0, group=None, engine=None, overlap: bool = False, device: Optional[int] = None,
                 transport: str = "torch"):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.group = group
        self.distributed = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank(group) if self.distributed else 0
        self.world = dist.get_world_size(group) if self.distributed else 1
        self.backend = dist.get_backend(group) if self.distributed else "none"
        self.rows, self.cols, self.periodic = int(rows_per_rank), int(cols), bool(periodic)
        self.total_rows = self.rows * self.world
        self.k = int(sweeps_per_exchange)
        self.ghost = 2 * self.k
        if self.ghost > self.rows:
            raise ValueError(f"sweeps_per_exchange={self.k} needs {self.ghost} ghost rows but the slab has {self.rows}")
        self.seed = int(seed)
        self.sweep_count = 0
        self.n_exchanges = 0  # halo exchanges issued so far (whatever the transport)
        if overlap and transport == "rccl":
            # tsu_ising2d_halo_exchange enqueues its RCCL group on the context's own stream: behind the interior launch, not beside it
            raise ValueError("overlap=True needs transport='torch' (the library's RCCL transport runs on the compute stream: "
                             "its exchange would queue behind the interior sweep instead of overlapping it)")
        self.up = (self.rank - 1) % self.world if (self.periodic or self.rank > 0) else None
        self.down = (self.rank + 1) % self.world if (self.periodic or self.rank < self.world - 1) else None
        # ghosts staged through host memory: test doubles, and torch's gloo backend -- unless the library's own RCCL transport
        # moves them (then gloo, if a process group exists at all, only carries the communicator's id)
        self._host_staged = engine is not None or (self.backend == "gloo" and transport != "rccl")
        if engine is None:
            ctx = _hip.Context.default() if device is None else _hip.Context(device)
            self.ctx = ctx
            self.lat = _hip.Lattice(self.rows, self.cols, self.periodic, ctx=ctx, total_rows=self.total_rows,
                                    row0=self.rank * self.rows, ghost=self.ghost)
            self.device_index = torch.cuda.current_device() if device is None else device
            # the library enqueues on torch's current stream so that RCCL ops and kernels are ordered by torch
            self.compute_stream = torch.cuda.current_stream(self.device_index)
            ctx.set_stream(self.compute_stream.cuda_stream)
            self.comm_stream = torch.cuda.Stream(self.device_index) if overlap else None
        else:
            self.ctx = None
            self.lat = engine(self.rows, self.cols, self.periodic, total_rows=self.total_rows, row0=self.rank * self.rows,
                              ghost=self.ghost)
            self.comm_stream = None
        # transport of the ghost rows between GPUs: "torch" = torch.distributed point-to-point ops (RCCL under the nccl backend);
        # "rccl" = the library's own RCCL calls below the C ABI (tsu_ising2d_halo_exchange; torch.distributed, if initialised,
        # only carries the communicator's 128-byte id from rank 0 to the others)
        if transport not in ("torch", "rccl"):
            raise ValueError("transport must be 'torch' or 'rccl'")
        self.transport = transport
        self.comm = None
        if transport == "rccl":
            if engine is not None:
                raise ValueError("transport='rccl' moves device buffers: it needs the HIP lattice")
            if self.distributed:
                uid = torch.zeros(128, dtype=torch.uint8)
                if self.rank == 0:
                    uid = torch.frombuffer(bytearray(_hip.comm_unique_id()), dtype=torch.uint8).clone()
                if self.backend == "nccl":
                    uid = uid.cuda(self.device_index)
                dist.broadcast(uid, src=0, group=group)
                uid_bytes = bytes(uid.cpu().numpy().tobytes())
            else:
                uid_bytes = _hip.comm_unique_id()
            self.comm = _hip.Comm(self.world, self.rank, uid_bytes, ctx=self.ctx)
        # split (interior / boundary) launches need the tiled kernel and full tile rows: decided at the first sweep
        self._split = None if (engine is None and self.comm_stream is not None and not self._host_staged) else False
        self._views = {}

    # ------------------------------------------------------------------ state
    def randomize(self, seed: Optional[int] = None):
        self.lat.randomize(self.seed if seed is None else int(seed))

    def set_model(self, J: float, h: float, T: float, mode: int = _hip.MODE_PHYSICAL):
        self.lat.set_model(J, h, T, mode)

    def set_thresholds(self, table):
        self.lat.set_thresholds(table)

    def set_local_spins(self, spins):
        self.lat.set_spins(spins)

    def local_spins(self) -> np.ndarray:
        return self.lat.get_spins()

    # ------------------------------------------------------------------ halo exchange
    def _rows_tensor(self, local_row: int, n_rows: int):
        ptr, pitch = self.lat.row_ptr(local_row)
        key = (ptr, n_rows)
        t = self._views.get(key)
        if t is None:
            t = self.torch.as_tensor(_DeviceRows(ptr, n_rows, pitch), device=f"cuda:{self.device_index}")
            self._views[key] = t
        return t

    def _exchange_device(self):
        """Ghost refresh with device buffers (RCCL send/recv, or a local copy when this rank is its own neighbour)."""
        if self.comm is not None:  # the library's own RCCL group on the context's stream
            self.comm.halo_exchange(self.lat)
            return
        G, R = self.ghost, self.rows
        top, bot = self._rows_tensor(0, G), self._rows_tensor(R - G, G)
        gtop, gbot = self._rows_tensor(-G, G), self._rows_tensor(R, G)
        if self.world == 1:
            if self.periodic:
                gtop.copy_(bot)
                gbot.copy_(top)
            return
        dist = self.dist
        ops = []
        # order matters when up == down (two ranks): my first send (to up) pairs with the peer's first recv (from down)
        if self.up is not None:
            ops.append(dist.P2POp(dist.isend, top, self.up, self.group))
        if self.down is not None:
            ops.append(dist.P2POp(dist.isend, bot, self.down, self.group))
        if self.down is not None:
            ops.append(dist.P2POp(dist.irecv, gbot, self.down, self.group))
        if self.up is not None:
            ops.append(dist.P2POp(dist.irecv, gtop, self.up, self.group))
        for req in dist.batch_isend_irecv(ops):
            req.wait()

    def _exchange_host(self):
        """Ghost refresh staged through host memory (gloo backend / test doubles)."""
        G, R = self.ghost, self.rows
        torch, dist = self.torch, self.dist
        top = torch.from_numpy(np.ascontiguousarray(self.lat.get_spins(0, G)))
        bot = torch.from_numpy(np.ascontiguousarray(self.lat.get_spins(R - G, G)))
        if self.world == 1:
            if self.periodic:
                self.lat.set_spins(bot.numpy(), row_first=-G)
                self.lat.set_spins(top.numpy(), row_first=R)
            return
        gtop, gbot = torch.empty_like(top), torch.empty_like(bot)
        ops = []
        if self.up is not None:
            ops.append(dist.P2POp(dist.isend, top, self.up, self.group))
        if self.down is not None:
            ops.append(dist.P2POp(dist.isend, bot, self.down, self.group))
        if self.down is not None:
            ops.append(dist.P2POp(dist.irecv, gbot, self.down, self.group))
        if self.up is not None:
            ops.append(dist.P2POp(dist.irecv, gtop, self.up, self.group))
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        if self.up is not None:
            self.lat.set_spins(gtop.numpy(), row_first=-G)
        if self.down is not None:
            self.lat.set_spins(gbot.numpy(), row_first=R)

    def exchange(self):
        self.n_exchanges += 1
        if self._host_staged:
            self._exchange_host()
        else:
            self._exchange_device()

    # ------------------------------------------------------------------ sweeps
    def sweep(self, n_sweeps: int):
        """n_sweeps checkerboard sweeps of the whole distributed lattice (asynchronous on the GPU path)."""
        done = 0
        while done < n_sweeps:
            k = min(self.k, n_sweeps - done)
            if self._split is not False:
                torch = self.torch
                # boundary rows of the current state are final (previous BOUNDARY launch): start the exchange on
                # the comm stream, sweep the interior tile rows meanwhile, then the two boundary tile rows
                ready = torch.cuda.Event()
                ready.record(self.compute_stream)
                try:
                    self.lat.sweep_part(k, self.seed, self.sweep_count, _hip.PART_INTERIOR)
                    self._split = True
                except _hip.UnsupportedError:
                    if self._split is True:
                        raise
                    self._split = False
                    continue
                with torch.cuda.stream(self.comm_stream):
                    self.comm_stream.wait_event(ready)
                    self._exchange_device()
                    landed = torch.cuda.Event()
                    landed.record(self.comm_stream)
                self.compute_stream.wait_event(landed)
                self.lat.sweep_part(k, self.seed, self.sweep_count, _hip.PART_BOUNDARY)
            else:
                self.exchange()
                self.lat.sweep(k, self.seed, self.sweep_count)
            self.sweep_count += k
            done += k

    def observables(self):
        """(sum of spins, sum over bonds) of the WHOLE lattice: local reductions + one all-reduce of two int64."""
        self.exchange()  # the bond to the row below the slab needs a fresh ghost row
        s, b = self.lat.observables()
        if self.comm is not None:
            tot = self.comm.allreduce([s, b])
            return int(tot[0]), int(tot[1])
        if not self.distributed:
            return s, b
        # with a process group the sums always go through the collective (a one-rank RCCL group included: the same code
        # runs at every world size)
        t = self.torch.tensor([s, b], dtype=self.torch.int64)
        if self.backend == "nccl":
            t = t.cuda(self.device_index)
        self.dist.all_reduce(t, group=self.group)
        return int(t[0]), int(t[1])

    def gather_spins(self) -> Optional[np.ndarray]:
        """The whole lattice on rank 0 (tests / small lattices only)."""
        mine = self.torch.from_numpy(self.local_spins().copy())
        if not self.distributed:
            return mine.numpy()
        if self.backend == "nccl":
            mine = mine.cuda(self.device_index)
        parts = [self.torch.empty_like(mine) for _ in range(self.world)] if self.rank == 0 else None
        self.dist.gather(mine, parts, dst=0, group=self.group)
        if self.rank != 0:
            return None
        return np.concatenate([p.cpu().numpy() for p in parts])

    def synchronize(self, timeout_s: float = 120.0):
        """Wait for this rank's sweeps and exchanges -- never for ever: a halo exchange whose peer does not arrive raises after
        ``timeout_s`` (``TimeoutError`` / ``HipError``), so that a broken world ends with a non-zero exit instead of a hung node."""
        if self.comm is not None:
            self.comm.wait(timeout_s)
            return
        if self.ctx is None:
            return
        import time
        ev = self.torch.cuda.Event()
        ev.record(self.compute_stream)
        t0 = time.monotonic()
        while not ev.query():
            if time.monotonic() - t0 > timeout_s:
                raise TimeoutError(f"rank {self.rank} of {self.world}: sweeps / halo exchange {self.n_exchanges} did not finish within "
                                   f"{timeout_s:.0f} s (a neighbouring rank never arrived?)")
            time.sleep(0.0002)


# ------------------------------------------------------------------------------------------------------------------------------
# Replicas over ranks (second half): the sequential dense chain does not shard -- a column-sharded field would
# need an all-reduce per block of sites -- but independent replicas do.  No collective on the sweep path; a tempering swap needs
# two scalars per neighbouring pair (their energies).

class ReplicaLadder:
    """``GibbsSampler.parallel_tempering`` (reference: tsu/gibbs.py:238-338) with the replicas spread over the ranks of a process
    group: rank r owns the chains [r m, (r + 1) m) of R = len(temperatures) (m = ceil(R / world)), all of them advanced by ONE launch
    per step on that rank's GPU (``tsu_dense_sweep_replicas``: one stream of J for the rank's chains).

    What moves between ranks is never a state: a swap exchanges the SLOTS of two chains (temperature, Philox seed and sweep counter
    belong to the slot, as they belong to the reference's per-temperature sampler objects, gibbs.py:285-291), and deciding it
    needs the two energies only -- one ``all_gather`` of one float64 per chain per swap round (the "2 scalars" of the survey).
    Every rank draws the same ``np.random`` numbers (initial states, acceptance uniforms), so the slot bookkeeping is replicated
    and the trajectory is the single-process one for any world size (tested on gloo with an oracle-backed double, world 1 / 2 / 3).
    The sample of a step is the state in slot 0 (gibbs.py:326); its owner keeps it, ``samples()`` gathers them on every rank.

    ``engine(coupling, bias)`` builds the per-rank dense system (default: :class:`tsu._hip.DenseSystem`; the CPU-only tests inject
    a double with ``sweep_replicas`` / ``energies``, the product never does)."""

    def __init__(self, coupling, temperatures, bias=None, n_burnin: int = 100, n_sweeps: int = 10, seed: Optional[int] = None,
                 group=None, engine=None, coupling_dtype: str = "float64"):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.distributed = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank(group) if self.distributed else 0
        self.world = dist.get_world_size(group) if self.distributed else 1
        self.backend = dist.get_backend(group) if self.distributed else "none"
        self.temperatures = [float(T) for T in temperatures]
        for T in self.temperatures:
            if T <= 0:
                raise ValueError("Temperature must be positive")
        self.R = len(self.temperatures)
        coupling = np.asarray(coupling)
        if coupling.ndim != 2 or coupling.shape[0] != coupling.shape[1]:
            raise ValueError("Coupling matrix must be square")
        self.n = coupling.shape[0]
        self.n_burnin, self.n_sweeps = int(n_burnin), int(n_sweeps)
        m = -(-self.R // self.world)
        self.mine = list(range(min(self.R, self.rank * m), min(self.R, (self.rank + 1) * m)))  # chains this rank owns
        self.owner = [min(c // m, self.world - 1) for c in range(self.R)]
        # replicated bookkeeping (identical on every rank): the reference's draws in its order
        if seed is None:
            seed = int(np.random.randint(0, 2 ** 31 - 1)) | (int(np.random.randint(0, 2 ** 31 - 1)) << 31)
        self.seed = int(seed)
        states = [np.random.randint(0, 2, size=self.n) for _ in range(self.R)]  # gibbs.py:282 (every rank draws all, keeps its own)
        self.slot_of = list(range(self.R))          # chain c sits in slot slot_of[c]
        self.chain_in = list(range(self.R))         # slot i holds chain chain_in[i]
        self.slot_seed = [self.seed + i + 1 for i in range(self.R)]   # as GibbsSampler.parallel_tempering seeds its per-slot samplers
        self.slot_sweeps = [0] * self.R
        if engine is None:
            dt = _hip.DTYPE_F64 if coupling_dtype == "float64" else _hip.DTYPE_F32
            self.sys = _hip.DenseSystem(coupling, bias, dt)
        else:
            self.sys = engine(coupling, bias)
        self.states = np.array([states[c] for c in self.mine], dtype=np.int8).reshape(len(self.mine), self.n)
        self._samples = {}   # step -> state of slot 0 (on the rank that owned it at that step)
        self.steps = 0
        self.swap_attempts = self.swap_accepts = 0
        self.energies_history = [[] for _ in range(self.R)]

    # ---- one launch per call on every rank
    def _sweep(self, n_sweeps: int):
        if n_sweeps <= 0:
            return
        if self.mine:
            slots = [self.slot_of[c] for c in self.mine]
            self.states = np.asarray(self.sys.sweep_replicas(self.states, [self.temperatures[i] for i in slots], n_sweeps,
                                                             [self.slot_seed[i] for i in slots], [self.slot_sweeps[i] for i in slots]),
                                     dtype=np.int8).reshape(len(self.mine), self.n)
        for i in range(self.R):
            self.slot_sweeps[i] += n_sweeps

    def _all_energies(self) -> np.ndarray:
        """Energy of every chain, indexed by chain: each rank evaluates its own, one all_gather of m float64."""
        m = -(-self.R // self.world)
        local = np.zeros(m)
        if self.mine:
            local[:len(self.mine)] = self.sys.energies(self.states)
        if not self.distributed:
            return local[:self.R]
        t = self.torch.from_numpy(local)
        if self.backend == "nccl":
            t = t.cuda()
        parts = [self.torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(parts, t, group=self.group)
        return np.concatenate([p.cpu().numpy() for p in parts])[:self.R]

    def run(self, n_samples: int, swap_interval: int = 10):
        """The reference's loop (gibbs.py:293-327): burn-in, then per step n_sweeps sweeps of every replica, an energy record, every
        ``swap_interval`` steps a round of neighbour swaps, and the state of slot 0 as the step's sample."""
        self._sweep(self.n_burnin)
        for _ in range(int(n_samples)):
            self._sweep(self.n_sweeps)
            e = self._all_energies()
            for i in range(self.R):
                self.energies_history[i].append(float(e[self.chain_in[i]]))
            self.steps += 1
            if self.steps % swap_interval == 0:
                for i in range(self.R - 1):
                    ci, cj = self.chain_in[i], self.chain_in[i + 1]
                    delta = (1.0 / self.temperatures[i] - 1.0 / self.temperatures[i + 1]) * (e[cj] - e[ci])
                    self.swap_attempts += 1
                    if delta >= 0 or np.random.rand() < np.exp(delta):  # gibbs.py:317-323, the same draw on every rank
                        self.chain_in[i], self.chain_in[i + 1] = cj, ci
                        self.slot_of[ci], self.slot_of[cj] = i + 1, i
                        self.swap_accepts += 1
            c0 = self.chain_in[0]
            if c0 in self.mine:
                self._samples[self.steps - 1] = self.states[self.mine.index(c0)].copy()
        return self

    def samples(self) -> np.ndarray:
        """(n_steps, n) states of slot 0, on every rank (one all_gather_object at the end of a run, not per step)."""
        if not self.distributed:
            merged = self._samples
        else:
            parts = [None] * self.world
            self.dist.all_gather_object(parts, self._samples, group=self.group)
            merged = {}
            for p in parts:
                merged.update(p)
        return np.array([merged[k] for k in range(self.steps)], dtype=np.int64).reshape(self.steps, self.n)

    def info(self) -> dict:
        return {"swap_acceptance_rate": self.swap_accepts / self.swap_attempts if self.swap_attempts else 0,
                "swap_attempts": self.swap_attempts, "swap_accepts": self.swap_accepts, "energies": self.energies_history}

    def close(self):
        if hasattr(self.sys, "close"):
            self.sys.close()


def sample_quadratic_sharded(q, x_init, n_samples: int, config, seed: int, group=None, engine=None, chain0: int = 0) -> np.ndarray:
    """``ThermalSamplingUnit.sample_from_energy`` for a separable quadratic energy (reference: tsu/core.py:100-162: ``n_samples``
    independent restarts of one Langevin chain) with the restarts sharded over the ranks: rank r runs the chains
    [r m, (r + 1) m) on its GPU (K3, all steps fused) and one ``all_gather`` brings the samples to every rank.  No collective
    inside the chain loop.  The Philox stream is keyed by the GLOBAL chain number, so the samples do not depend on the world size
    (tested on gloo, world 1 / 2 / 3, and against the single-process path).  ``engine(n_chains, dim)``: the per-rank chain set
    (default :class:`tsu._hip.LangevinChains`; CPU-only tests inject an oracle-backed double)."""
    import torch
    import torch.distributed as dist
    distributed = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank(group) if distributed else 0
    world = dist.get_world_size(group) if distributed else 1
    backend = dist.get_backend(group) if distributed else "none"
    x0 = np.atleast_1d(np.asarray(x_init, dtype=np.float64))
    d = x0.size
    m = -(-int(n_samples) // world)
    lo, hi = min(n_samples, rank * m), min(n_samples, (rank + 1) * m)
    local = np.zeros((m, d), dtype=np.float32)
    if hi > lo:
        lc = (engine or _hip.LangevinChains)(hi - lo, d)
        try:
            lc.set_energy(np.broadcast_to(q.k, (d,)).astype(np.float32), np.broadcast_to(q.mu, (d,)).astype(np.float32))
            lc.restart(x0.astype(np.float32), 0.1, seed, chain0 + lo)   # x_init + 0.1 N(0,1) (core.py:142-143) ...
            if lo == 0:                                                # ... except sample 0, which starts exactly at x_init
                st = lc.get_state()
                st[0] = x0.astype(np.float32)
                lc.set_state(st)
            lc.step(int(config.n_burnin), config.dt, config.friction, config.temperature, seed, 0, chain0 + lo)
            lc.step(int(config.n_steps), config.dt, config.friction, config.temperature, seed, int(config.n_burnin), chain0 + lo)
            local[:hi - lo] = lc.get_state()
        finally:
            lc.close()
    if not distributed:
        return local[:n_samples].astype(np.float64)
    t = torch.from_numpy(local)
    if backend == "nccl":
        t = t.cuda()
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t, group=group)
    return np.concatenate([p.cpu().numpy() for p in parts])[:n_samples].astype(np.float64)
