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
