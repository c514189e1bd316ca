"""Replicas over ranks on CPU (gloo, world_size 1 / 2 / 3): the tempering ladder with one group of chains per rank and the 2-scalar
swap (tsu.distributed.ReplicaLadder; reference loop tsu/gibbs.py:293-327) and the Langevin restarts sharded over ranks
(tsu.distributed.sample_quadratic_sharded; reference tsu/core.py:140-159).  No GPU here: the per-rank engines are TEST DOUBLES on
the oracle (the product's defaults are the HIP systems, which raise without a GPU).  Checked: the result does not depend on the
world size, and world 1 is the plain sequential loop written out below."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleDense:
    """Double of tsu._hip.DenseSystem for the two calls the ladder makes."""

    def __init__(self, coupling, bias):
        from oracle import oracle as ora
        self.ora, self.J, self.b = ora, np.asarray(coupling, dtype=np.float64), None if bias is None else np.asarray(bias, dtype=np.float64)

    def sweep_replicas(self, states, temperatures, n_sweeps, seeds, sweep0s):
        return np.array([self.ora.dense_sweep_philox(s, self.J, self.b, T, n_sweeps, sd, sweep0=s0)
                         for s, T, sd, s0 in zip(np.asarray(states, dtype=np.int8), temperatures, seeds, sweep0s)], dtype=np.int8)

    def energies(self, states):
        return np.array([self.ora.ref_compute_energy(s.astype(np.int64), self.J, self.b) for s in np.asarray(states)])


class OracleChains:
    """Double of tsu._hip.LangevinChains on the oracle's fp32 twin (quadratic energy, Philox keyed by the global chain number)."""

    def __init__(self, n_chains, dim):
        from oracle import oracle as ora
        self.ora, self.nc, self.d = ora, n_chains, dim
        self.x = np.zeros((n_chains, dim), np.float32)

    def set_energy(self, k, mu):
        self.k, self.mu = np.asarray(k, np.float32), np.asarray(mu, np.float32)

    def restart(self, x_init, amp, seed, chain0=0):
        for c in range(self.nc):
            z = np.concatenate([self.ora.langevin_normals_f32(q, chain0 + c, 0, seed, tag=5) for q in range((self.d + 3) // 4)])[:self.d]
            self.x[c] = np.asarray(x_init, np.float32) + np.float32(amp) * z.astype(np.float32)

    def get_state(self):
        return self.x.copy()

    def set_state(self, x):
        self.x = np.asarray(x, np.float32).reshape(self.nc, self.d).copy()

    def step(self, n_steps, dt, gamma, T, seed, step0=0, chain0=0, trajectory=False):
        self.x = self.ora.langevin_quadratic_f32(self.x, self.k, self.mu, n_steps, dt, gamma, T, seed, step0=step0, chain0=chain0)

    def close(self):
        pass


def _problem():
    rng = np.random.default_rng(3)
    n = 24
    J = rng.normal(size=(n, n)) / 3
    J = (J + J.T) / 2
    return J, rng.normal(size=n) * 0.2, [0.6, 0.9, 1.4, 2.1, 3.0]


def _run_ladder(world_rank=None):
    from tsu.distributed import ReplicaLadder
    J, b, temps = _problem()
    np.random.seed(11)
    lad = ReplicaLadder(J, temps, b, n_burnin=2, n_sweeps=2, seed=99, engine=OracleDense)
    lad.run(7, swap_interval=2)
    return lad.samples(), lad.info()


def _sequential_reference():
    """gibbs.py:293-327 written out: every replica swept at ITS slot's temperature / seed / counter, states swapped."""
    from oracle import oracle as ora
    J, b, temps = _problem()
    np.random.seed(11)
    R, n = len(temps), J.shape[0]
    states = [np.random.randint(0, 2, size=n) for _ in range(R)]
    seeds, sweeps = [99 + i + 1 for i in range(R)], [0] * R

    def sweep_all(k):
        for i in range(R):
            states[i] = ora.dense_sweep_philox(states[i].astype(np.int8), J, b, temps[i], k, seeds[i], sweep0=sweeps[i]).astype(np.int64)
            sweeps[i] += k

    sweep_all(2)
    samples, attempts, accepts = [], 0, 0
    for step in range(1, 8):
        sweep_all(2)
        if step % 2 == 0:
            for i in range(R - 1):
                Ei, Ej = ora.ref_compute_energy(states[i], J, b), ora.ref_compute_energy(states[i + 1], J, b)
                delta = (1.0 / temps[i] - 1.0 / temps[i + 1]) * (Ej - Ei)
                attempts += 1
                if delta >= 0 or np.random.rand() < np.exp(delta):
                    states[i], states[i + 1] = states[i + 1], states[i]
                    accepts += 1
        samples.append(states[0].copy())
    return np.array(samples), attempts, accepts


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tsu-emulator_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import test_replicas_distributed_cpu as me
    from tsu.core import QuadraticEnergy, TSUConfig
    from tsu.distributed import sample_quadratic_sharded
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        samples, info = me._run_ladder()
        cfg = TSUConfig(temperature=0.8, dt=0.02, n_burnin=5, n_steps=9)
        x = sample_quadratic_sharded(QuadraticEnergy([2.0, 1.0, 4.0, 0.5, 3.0, 2.5], 0.3), np.linspace(-1, 1, 6), 7, cfg, seed=5, engine=me.OracleChains)
        if rank == world - 1:  # (any rank holds the whole result)
            q.put((samples, info["swap_attempts"], info["swap_accepts"], x))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_ladder_and_sharded_langevin_do_not_depend_on_the_world_size(world):
    sys.path.insert(0, os.path.join(ROOT, "tsu-emulator_amd"))
    from tsu.core import QuadraticEnergy, TSUConfig
    from tsu.distributed import sample_quadratic_sharded
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=170)
    for p in procs:
        p.join(60)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    samples, attempts, accepts, x = got
    want, a1, a2 = _sequential_reference()
    np.testing.assert_array_equal(samples, want)
    assert (attempts, accepts) == (a1, a2) and attempts == 3 * 4
    cfg = TSUConfig(temperature=0.8, dt=0.02, n_burnin=5, n_steps=9)
    x1 = sample_quadratic_sharded(QuadraticEnergy([2.0, 1.0, 4.0, 0.5, 3.0, 2.5], 0.3), np.linspace(-1, 1, 6), 7, cfg, seed=5, engine=OracleChains)
    np.testing.assert_array_equal(x, x1)
    np.testing.assert_array_equal(x1[0][:0], [])  # (shape check below)
    assert x1.shape == (7, 6)


def test_ladder_without_a_process_group_is_the_sequential_loop():
    sys.path.insert(0, os.path.join(ROOT, "tsu-emulator_amd"))
    samples, info = _run_ladder()
    want, a1, a2 = _sequential_reference()
    np.testing.assert_array_equal(samples, want)
    assert (info["swap_attempts"], info["swap_accepts"]) == (a1, a2)
    assert len(info["energies"]) == 5 and len(info["energies"][0]) == 7
