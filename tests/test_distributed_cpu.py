"""Multi-process (gloo, world_size 2 and 3) tests of the slab decomposition on CPU.

The host logic under test is tsu.distributed.SlabLattice (halo pairing, ghost depth, sweep counters, all-reduce of
observables).  There is no GPU here, so the per-slab engine is a TEST DOUBLE built on the oracle's window sweep; the
product never takes this path (its default engine is the HIP lattice and it raises without a GPU)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleSlab:
    """Test double with the interface of tsu._hip.Lattice for a slab: owned rows + ghost rows, oracle arithmetic."""

    def __init__(self, rows, cols, periodic, total_rows, row0, ghost):
        from oracle import oracle as ora
        self.ora = ora
        self.rows, self.cols, self.periodic = rows, cols, periodic
        self.total_rows, self.row0, self.ghost = total_rows, row0, ghost
        self.buf = np.zeros((rows + 2 * ghost, cols), dtype=np.int8)
        self.table = None

    def randomize(self, seed):
        g = self.ghost
        for r in range(-g, self.rows + g):
            R = self.row0 + r
            if self.periodic:
                R %= self.total_rows
            if 0 <= R < self.total_rows:
                self.buf[g + r] = self.ora.ising2d_randomize(1, self.cols, seed, row0=R)[0]

    def set_model(self, J, h, T, mode=0):
        self.table = self.ora.ising2d_thresholds(J, h, T, mode)

    def set_thresholds(self, table):
        self.table = np.asarray(table, dtype=np.uint64)

    def set_spins(self, spins, row_first=0):
        s = np.asarray(spins, dtype=np.int8).reshape(-1, self.cols)
        self.buf[self.ghost + row_first:self.ghost + row_first + s.shape[0]] = s

    def get_spins(self, row_first=0, n_rows=None):
        n_rows = self.rows if n_rows is None else n_rows
        return self.buf[self.ghost + row_first:self.ghost + row_first + n_rows].copy()

    def sweep(self, n_sweeps, seed, sweep0=0, replica=0):
        assert 2 * n_sweeps <= self.ghost
        g = self.ghost
        lo = 0 if (not self.periodic and self.row0 - g < 0) else None
        # window = ghost + owned + ghost, clipped to the lattice when it is open
        top = g if (self.periodic or self.row0 - g >= 0) else self.row0
        bot = g if (self.periodic or self.row0 + self.rows + g <= self.total_rows) else self.total_rows - self.row0 - self.rows
        win = self.buf[g - top:g + self.rows + bot]
        out = self.ora.ising2d_sweep_window(win, self.row0 - top, self.total_rows, self.periodic, self.table, n_sweeps, seed,
                                            sweep0, replica)
        self.buf[g:g + self.rows] = out[top:top + self.rows]
        del lo

    def observables(self):
        g = self.ghost
        s = self.buf[g:g + self.rows].astype(np.int64)
        below_exists = self.periodic or (self.row0 + self.rows < self.total_rows)
        nxt = self.buf[g + 1:g + self.rows + 1].astype(np.int64)
        bonds = np.sum(s[:, :-1] * s[:, 1:]) + (np.sum(s[:, -1] * s[:, 0]) if self.periodic else 0)
        bonds += np.sum(s[:-1] * nxt[:-1]) + (np.sum(s[-1] * nxt[-1]) if below_exists else 0)
        return int(s.sum()), int(bonds)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, periodic, k, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tsu-emulator_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from oracle import oracle as ora
    from test_distributed_cpu import OracleSlab
    from tsu.distributed import SlabLattice
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        rows, cols, seed = 12, 24, 321
        lat = SlabLattice(rows, cols, periodic=periodic, sweeps_per_exchange=k, seed=seed, engine=OracleSlab)
        assert lat.world == world and lat.total_rows == rows * world and lat.ghost == 2 * k
        lat.randomize()
        lat.set_model(1.0, 0.05, 2.3)
        lat.sweep(2 * k + 1)  # two full exchanges + a short tail
        obs = lat.observables()
        full = lat.gather_spins()
        if rank == 0:
            table = ora.ising2d_thresholds(1.0, 0.05, 2.3, 0)
            want = ora.ising2d_sweep(ora.ising2d_randomize(rows * world, cols, seed), periodic, table, 2 * k + 1, seed)
            q.put((bool((full == want).all()), obs == ora.ising2d_observables(want, periodic), lat.sweep_count))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,periodic,k", [(2, True, 1), (2, True, 3), (2, False, 2), (3, True, 2), (3, False, 1)])
def test_slab_decomposition_is_bit_identical_to_single_lattice(world, periodic, k):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, periodic, k, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    same, obs_ok, count = q.get(timeout=5)
    assert same and obs_ok and count == 2 * k + 1


def test_single_process_slab_self_exchange():
    """world_size 1 without a process group: the slab is its own neighbour (periodic) or has none (open)."""
    from oracle import oracle as ora
    from tsu.distributed import SlabLattice
    for periodic in (True, False):
        lat = SlabLattice(16, 20, periodic=periodic, sweeps_per_exchange=2, seed=5, engine=OracleSlab)
        lat.randomize()
        lat.set_model(1.0, 0.0, 2.269185)
        lat.sweep(5)
        table = ora.ising2d_thresholds(1.0, 0.0, 2.269185, 0)
        want = ora.ising2d_sweep(ora.ising2d_randomize(16, 20, 5), periodic, table, 5, 5)
        np.testing.assert_array_equal(lat.gather_spins(), want)
        assert lat.observables() == ora.ising2d_observables(want, periodic)
