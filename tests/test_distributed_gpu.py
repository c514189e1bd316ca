"""Slab decomposition on the GPU (-m gpu): the product engine (HIP lattice) under tsu.distributed.SlabLattice.

One MI355X is available to the tests, so: (a) world_size 1 exercises the device halo path (zero-copy torch views of
the library's rows, self-exchange, the split interior/boundary launches and the two-stream overlap); (b) two
processes share the GPU over gloo with host-staged halos.  RCCL itself needs one GPU per rank (driver's 8-GPU run)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from oracle import oracle as ora

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("rows,cols,k,overlap", [(256, 1024, 4, True), (256, 1024, 4, False), (128, 544, 8, True),
                                                 (192, 640, 2, True), (64, 64, 2, True), (256, 1024, 32, False),
                                                 (384, 576, 20, False)])
def test_single_rank_slab_device_exchange_and_overlap(rows, cols, k, overlap):
    from tsu import _hip
    from tsu.distributed import SlabLattice
    seed = 77
    lat = SlabLattice(rows, cols, periodic=True, sweeps_per_exchange=k, seed=seed, overlap=overlap)
    lat.randomize()
    np.testing.assert_array_equal(lat.local_spins(), ora.ising2d_randomize(rows, cols, seed))
    lat.set_model(1.0, 0.0, 2.269185, _hip.MODE_PHYSICAL)
    lat.sweep(3 * k + 1)
    lat.synchronize()
    table = ora.ising2d_thresholds(1.0, 0.0, 2.269185, 0)
    want = ora.ising2d_sweep(ora.ising2d_randomize(rows, cols, seed), True, table, 3 * k + 1, seed)
    np.testing.assert_array_equal(lat.gather_spins(), want)
    assert lat.observables() == ora.ising2d_observables(want, True)
    if overlap and rows % 128 == 0 and cols >= 544 and k <= 8:
        assert lat._split is True  # the interior/boundary split was really used
    lat.sweep(k)
    want = ora.ising2d_sweep(want, True, table, k, seed, sweep0=3 * k + 1)
    np.testing.assert_array_equal(lat.gather_spins(), want)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tsu-emulator_amd"))
    import torch
    import torch.distributed as dist
    from tsu import _hip
    from tsu.distributed import SlabLattice
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        rows, cols, seed, k = 128, 576, 9, 10
        lat = SlabLattice(rows, cols, periodic=True, sweeps_per_exchange=k, seed=seed, device=0)
        lat.lat.set_kernel(_hip.KERNEL_AUTO, 3)  # several launches per ghost refresh
        lat.randomize()
        lat.set_model(1.0, 0.0, 2.269185, _hip.MODE_PHYSICAL)
        lat.sweep(2 * k + 3)
        obs = lat.observables()
        full = lat.gather_spins()
        if rank == 0:
            q.put((full, obs))
    finally:
        dist.destroy_process_group()


def test_two_ranks_sharing_one_gpu_over_gloo():
    world, k = 2, 10
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    full, obs = q.get(timeout=240)
    for p in procs:
        p.join(120)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    table = ora.ising2d_thresholds(1.0, 0.0, 2.269185, 0)
    want = ora.ising2d_sweep(ora.ising2d_randomize(256, 576, 9), True, table, 2 * k + 3, 9)
    np.testing.assert_array_equal(full, want)
    assert obs == ora.ising2d_observables(want, True)


# ------------------------------------------------------------------ RCCL (backend "nccl")
def _nccl_worker(rank, world, port, q, rows, cols, k, sweeps):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tsu-emulator_amd"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    from tsu import _hip
    from tsu.distributed import SlabLattice
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world,
                            device_id=torch.device(f"cuda:{rank}"))
    try:
        lat = SlabLattice(rows, cols, periodic=True, sweeps_per_exchange=k, seed=21, device=rank)
        assert lat.backend == "nccl" and lat.distributed and not lat._host_staged
        lat.randomize()
        lat.set_model(1.0, 0.0, 2.269185, _hip.MODE_PHYSICAL)
        lat.sweep(sweeps)
        obs = lat.observables()          # all_reduce of a cuda tensor over RCCL
        full = lat.gather_spins()        # gather of cuda tensors over RCCL
        t = torch.ones(1, device=f"cuda:{rank}")
        dist.all_reduce(t)
        dist.barrier()
        if rank == 0:
            q.put((full, obs, float(t[0])))
    finally:
        dist.destroy_process_group()


def _run_nccl(world, rows, cols, k, sweeps):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_nccl_worker, args=(r, world, port, q, rows, cols, k, sweeps)) for r in range(world)]
    for p in procs:
        p.start()
    full, obs, ones = q.get(timeout=300)
    for p in procs:
        p.join(120)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert ones == world
    table = ora.ising2d_thresholds(1.0, 0.0, 2.269185, 0)
    want = ora.ising2d_sweep(ora.ising2d_randomize(world * rows, cols, 21), True, table, sweeps, 21)
    np.testing.assert_array_equal(full, want)
    assert obs == ora.ising2d_observables(want, True)


def test_rccl_process_group_of_one_rank_runs_the_nccl_branches():
    """init_process_group("nccl") + SlabLattice on the device halo path + all_reduce/gather of cuda tensors: everything
    of the N > 1 path that one GPU can execute (the send/recv pair itself needs a second GPU: next test)."""
    _run_nccl(1, 256, 1024, 8, 19)


def test_rccl_two_ranks_send_recv_halo():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    _run_nccl(2, 256, 1024, 8, 19)


def test_bench_gpus_2_on_this_box_is_never_reported_as_one_gpu():
    """`python bench.py --gpus 2` launches its own ranks; with a single GPU it must fail loudly, with two it must say 2."""
    import json
    import subprocess
    import torch
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--ramp-steps", "1", "--no-cpu-baseline", "--no-extra"], env=env, capture_output=True, text=True,
                       timeout=600)
    if torch.cuda.device_count() < 2:
        assert r.returncode != 0, r.stdout
        assert '"n_gpus"' not in r.stdout
        assert "needs 2 GPUs" in r.stderr
    else:
        assert r.returncode == 0, r.stderr[-2000:]
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
        out = json.loads(line)
        assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["backend"] == "nccl"


# ------------------------------------------------------------------ RCCL below the C ABI (tsu_comm_*, tsu_ising2d_halo_exchange)
def test_c_abi_rccl_transport_one_rank_is_its_own_neighbour():
    """transport="rccl": the halo exchange is the library's own RCCL group (send/recv to the rank above and below; with one
    rank both are this rank) and the observables go through ncclAllReduce -- no torch.distributed anywhere."""
    from tsu import _hip
    from tsu.distributed import SlabLattice
    rows, cols, k, seed = 256, 1024, 8, 33
    lat = SlabLattice(rows, cols, periodic=True, sweeps_per_exchange=k, seed=seed, transport="rccl")
    assert lat.comm is not None and not lat.distributed
    lat.randomize()
    lat.set_model(1.0, 0.0, 2.269185, _hip.MODE_PHYSICAL)
    lat.sweep(2 * k + 3)
    lat.synchronize()
    table = ora.ising2d_thresholds(1.0, 0.0, 2.269185, 0)
    want = ora.ising2d_sweep(ora.ising2d_randomize(rows, cols, seed), True, table, 2 * k + 3, seed)
    np.testing.assert_array_equal(lat.gather_spins(), want)
    assert lat.observables() == ora.ising2d_observables(want, True)
    # an open slab has no neighbours at world size 1: the exchange is an empty group
    op = SlabLattice(128, 512, periodic=False, sweeps_per_exchange=4, seed=5, transport="rccl")
    op.randomize()
    op.set_model(1.0, 0.1, 2.0, _hip.MODE_PHYSICAL)
    op.sweep(6)
    want = ora.ising2d_sweep(ora.ising2d_randomize(128, 512, 5), False, ora.ising2d_thresholds(1.0, 0.1, 2.0, 0), 6, 5)
    np.testing.assert_array_equal(op.gather_spins(), want)


def test_c_abi_rccl_transport_two_ranks():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_nccl_worker_abi, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    full, obs = q.get(timeout=300)
    for p in procs:
        p.join(120)
    assert all(p.exitcode == 0 for p in procs)
    table = ora.ising2d_thresholds(1.0, 0.0, 2.269185, 0)
    want = ora.ising2d_sweep(ora.ising2d_randomize(512, 1024, 21), True, table, 19, 21)
    np.testing.assert_array_equal(full, want)
    assert obs == ora.ising2d_observables(want, True)


def _nccl_worker_abi(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tsu-emulator_amd"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    from tsu import _hip
    from tsu.distributed import SlabLattice
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world, device_id=torch.device(f"cuda:{rank}"))
    try:
        lat = SlabLattice(256, 1024, periodic=True, sweeps_per_exchange=8, seed=21, device=rank, transport="rccl")
        lat.randomize()
        lat.set_model(1.0, 0.0, 2.269185, _hip.MODE_PHYSICAL)
        lat.sweep(19)
        obs = lat.observables()
        full = lat.gather_spins()
        if rank == 0:
            q.put((full, obs))
    finally:
        dist.destroy_process_group()


# ------------------------------------------------------------------ replicas over ranks (second half)
def _replica_worker(rank, world, port, q, n, temps):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tsu-emulator_amd"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    from tsu.core import QuadraticEnergy, TSUConfig
    from tsu.distributed import ReplicaLadder, sample_quadratic_sharded
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world, device_id=torch.device(f"cuda:{rank}"))
    try:
        rng = np.random.default_rng(3)
        J = rng.normal(size=(n, n)) / np.sqrt(n)
        J = (J + J.T) / 2
        np.random.seed(11)
        lad = ReplicaLadder(J, temps, None, n_burnin=2, n_sweeps=2, seed=99)
        lad.run(6, swap_interval=2)
        samples, info = lad.samples(), lad.info()
        lad.close()
        cfg = TSUConfig(temperature=0.8, dt=0.02, n_burnin=20, n_steps=50)
        x = sample_quadratic_sharded(QuadraticEnergy(2.0, 0.3), np.zeros(4096), 5, cfg, seed=5)
        if rank == 0:
            q.put((samples, info["swap_attempts"], info["swap_accepts"], x))
    finally:
        dist.destroy_process_group()


def _run_replicas(world, n, temps):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_replica_worker, args=(r, world, port, q, n, temps)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(120)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    return got


@pytest.mark.parametrize("n", [24, 2304])
def test_tempering_ladder_and_sharded_langevin_on_an_rccl_group_of_one_rank(n):
    """ReplicaLadder / sample_quadratic_sharded with the product engines under init_process_group("nccl") at world size 1 (all this
    box can run: the all_gather of the energies and of the samples go through RCCL on cuda tensors) == GibbsSampler.parallel_tempering
    / ThermalSamplingUnit.sample_from_energy in this process (same seeds, same np.random draws)."""
    from tsu.core import QuadraticEnergy, ThermalSamplingUnit, TSUConfig
    from tsu.gibbs import GibbsConfig, GibbsSampler
    temps = [0.6, 0.9, 1.4, 2.1, 3.0]
    samples, attempts, accepts, x = _run_replicas(1, n, temps)
    rng = np.random.default_rng(3)
    J = rng.normal(size=(n, n)) / np.sqrt(n)
    J = (J + J.T) / 2
    np.random.seed(11)
    s = GibbsSampler(GibbsConfig(n_burnin=2, n_sweeps=2), seed=99)
    want, info = s.parallel_tempering(J, temps, n_samples=6, swap_interval=2)
    np.testing.assert_array_equal(samples, want)
    assert (attempts, accepts) == (info["swap_attempts"], info["swap_accepts"])
    t = ThermalSamplingUnit(TSUConfig(temperature=0.8, dt=0.02, n_burnin=20, n_steps=50), seed=5)
    np.testing.assert_array_equal(x, t.sample_from_energy(QuadraticEnergy(2.0, 0.3), np.zeros(4096), n_samples=5))


def test_tempering_ladder_two_ranks():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    temps = [0.6, 0.9, 1.4, 2.1, 3.0]
    a = _run_replicas(1, 2304, temps)
    b = _run_replicas(2, 2304, temps)
    np.testing.assert_array_equal(a[0], b[0])
    assert a[1:3] == b[1:3]
    np.testing.assert_array_equal(a[3], b[3])
