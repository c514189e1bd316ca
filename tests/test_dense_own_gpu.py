"""K2 owner-computes kernel (k2_own, csrc/dense_own.hip) against the oracle, bit for bit: natural order over one and several
superblocks (row ranges of 64 / 128 / 256 sites per workgroup, partial last workgroup, f32 / f64 couplings, a bias), a caller's
visiting order (update_order="random", reference tsu/gibbs.py:153-160) with Philox and with replayed NumPy uniforms, replica
batches (the chain loop tsu/gibbs.py:474-479 and the tempering loop :296-306: every replica == its own single-chain sweep), whole
sampling runs and annealing schedules in one launch, and the size-independent properties at BASELINE configs[2] (N = 16384).
Every case checks through tsu_dense_launch_counts that the sweeps really ran on k2_own."""
import os

import numpy as np
import pytest

from oracle import oracle as ora

pytestmark = pytest.mark.gpu


def _system(n, seed, f64=False, bias=True, sym=True):
    rng = np.random.default_rng(seed)
    G = rng.standard_normal((n, n)).astype(np.float32)
    J = ((G + G.T) / 2 / np.sqrt(n)).astype(np.float32) if sym else (G / np.sqrt(n)).astype(np.float32)
    np.fill_diagonal(J, 0.0)
    b = rng.normal(size=n) * 0.1 if bias else None
    return (J.astype(np.float64) if f64 else J), b, rng.integers(0, 2, size=n).astype(np.int8)


def _dense(J, b, f64):
    from tsu import _hip
    return _hip.DenseSystem(J, b, _hip.DTYPE_F64 if f64 else _hip.DTYPE_F32)


class _env:
    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kw}
        for k, v in self.kw.items():
            os.environ[k] = str(v)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("n,T,f64,sym", [(2048, 1.0, False, True), (2052, 0.8, True, True), (4096, 1.0, False, True), (4100, 0.6, False, True),
                                         (6144, 1.0, True, True), (8192, 1.0, False, False), (12292, 1.3, False, True), (16384, 1.0, False, True)])
def test_natural_order_matches_oracle(n, T, f64, sym):
    J, b, s0 = _system(n, n, f64, sym=sym)
    d = _dense(J, b, f64)
    d.set_state(s0)
    d.sweep(T, 3, seed=7, sweep0=2)
    assert d.launch_counts()[0] == 1
    want = ora.dense_sweep_philox(s0, np.asarray(J, dtype=np.float64), b, T, 3, 7, sweep0=2)
    np.testing.assert_array_equal(d.get_state(), want)
    d.close()


@pytest.mark.parametrize("n,M,sb", [(4096, 2, 4096), (4100, 4, 2048), (6144, 1, 2048), (8192, 2, 8192), (8192, 1, 1024)])
def test_wider_row_ranges_and_other_superblock_sizes_give_the_same_trajectory(n, M, sb):
    """128 / 256 sites per workgroup (the layouts of systems above 16384 / 32768 sites, forced here at sizes the oracle finishes
    quickly) and superblocks of 1024 ... 8192 positions: the trajectory does not depend on either."""
    J, b, s0 = _system(n, 5 * n + M, False)
    d = _dense(J, b, False)
    d.set_state(s0)
    with _env(TSU_K2_OWN_M=M, TSU_K2_OWN_SB=sb):
        d.sweep(0.9, 3, seed=3, sweep0=1)
    assert d.launch_counts()[0] == 1
    np.testing.assert_array_equal(d.get_state(), ora.dense_sweep_philox(s0, J.astype(np.float64), b, 0.9, 3, 3, sweep0=1))
    d.close()


def test_20480_sites_two_row_groups_per_workgroup():
    n = 20480  # above 64 x 256: 128 sites per workgroup
    J, _, s0 = _system(n, 11, False, bias=False)
    d = _dense(J, None, False)
    d.set_state(s0)
    d.sweep(1.0, 2, seed=5, sweep0=0)
    assert d.launch_counts()[0] == 1
    np.testing.assert_array_equal(d.get_state(), ora.dense_sweep_philox(s0, J.astype(np.float64), None, 1.0, 2, 5, sweep0=0))
    d.close()


@pytest.mark.parametrize("n,f64", [(2048, True), (4100, False), (16384, False)])
def test_callers_visiting_order_matches_oracle(n, f64):
    """update_order="random" (gibbs.py:153-160): one permutation per sweep, on the one-launch kernel."""
    J, b, s0 = _system(n, 2 * n + 1, f64)
    rng = np.random.default_rng(n)
    order = np.array([rng.permutation(n) for _ in range(3)])
    d = _dense(J, b, f64)
    d.set_state(s0)
    d.sweep(0.9, 3, seed=3, sweep0=5, order=order)
    assert d.launch_counts()[0] == 1
    want = ora.dense_sweep_philox(s0, np.asarray(J, dtype=np.float64), b, 0.9, 3, 3, sweep0=5, order=order)
    np.testing.assert_array_equal(d.get_state(), want)
    d.close()


def test_callers_order_with_replayed_numpy_uniforms_is_the_reference_loop():
    """rng="numpy", update_order="random": permutation, then one rand() per visited site in visiting order (gibbs.py:157-160,126):
    the kernel replays the caller's draws and must return the reference loop's states (oracle: C replay of the sequential loop)."""
    n, k = 2304, 3
    J, b, s0 = _system(n, 9, f64=True)
    rng = np.random.default_rng(4)
    order = np.array([rng.permutation(n) for _ in range(k)])
    u = rng.random((k, n))
    d = _dense(J, b, True)
    d.set_state(s0)
    d.sweep(0.8, k, order=order, replay_uniforms=u)
    assert d.launch_counts()[0] == 1
    want = ora.c_dense_sweep_replay(s0.astype(np.int64), J, b, 0.8, u, order=order)
    np.testing.assert_array_equal(d.get_state(), want.astype(np.int8))
    d.close()


@pytest.mark.parametrize("n,R", [(2048, 2), (4100, 3), (6144, 5), (16384, 8)])
def test_replica_batch_every_replica_is_its_own_single_chain(n, R):
    """tsu_dense_sweep_replicas above the one-workgroup kernels: R states advance in one launch on one stream of J; replica r must
    be exactly what ora.dense_sweep_philox makes of state r with its own temperature, seed, sweep counter and stream id."""
    J, b, _ = _system(n, 3 * n, False)
    J64 = J.astype(np.float64)
    sts = np.array([np.random.default_rng(100 + r).integers(0, 2, size=n) for r in range(R)], dtype=np.int8)
    temps = [1.0 + 0.15 * r for r in range(R)]
    seeds = [11 + r for r in range(R)]
    sw0 = [4 * r for r in range(R)]
    d = _dense(J, b, False)
    out = d.sweep_replicas(sts, temps, 2, seeds, sw0, replicas=list(range(R)))
    assert d.launch_counts()[0] == 1
    for r in range(R):
        want = ora.dense_sweep_philox(sts[r], J64, b, temps[r], 2, seeds[r], sweep0=sw0[r], replica=r)
        np.testing.assert_array_equal(out[r], want, err_msg=f"replica {r}")
    d.close()


@pytest.mark.parametrize("n,R", [(2304, 3), (4100, 8)])
def test_replica_fields_are_kept_from_call_to_call(n, R):
    """A tempering loop: short calls on states that come back unchanged, swapped among themselves, or edited.  The replicas' fields
    stay on the device (dense.h rep_fields): a state that returns byte for byte resumes from its fields wherever it now sits, an
    edited one makes the call start from scratch -- every call's result is the oracle's, replica by replica."""
    J, b, _ = _system(n, 5 * n, False)
    J64 = J.astype(np.float64)
    rng = np.random.default_rng(n)
    sts = rng.integers(0, 2, size=(R, n)).astype(np.int8)
    temps = [0.8 + 0.2 * r for r in range(R)]
    d = _dense(J, b, False)
    sweep0 = 0
    for call in range(6):
        if call == 2:  # neighbours swap (as a tempering exchange does)
            sts[[0, 1]] = sts[[1, 0]]
        if call == 3:  # a rotation of all of them
            sts = np.roll(sts, 1, axis=0)
        if call == 4:  # one state is edited by the caller: nothing may be taken over for it
            sts[R - 1, 7] ^= 1
        k = 1 if call % 2 else 2
        out = d.sweep_replicas(sts, temps, k, [3] * R, [sweep0] * R, replicas=list(range(R)))
        for r in range(R):
            want = ora.dense_sweep_philox(sts[r], J64, b, temps[r], k, 3, sweep0=sweep0, replica=r)
            np.testing.assert_array_equal(out[r], want, err_msg=f"call {call}, replica {r}")
        sts = out
        sweep0 += k
    assert d.launch_counts()[0] == 6
    d.close()


def test_energy_of_a_replica_state_comes_from_its_kept_fields():
    """parallel_tempering asks for every replica's energy between its sweeps (gibbs.py:303-323): set_state + energy of a state the
    last replica call returned takes that state's kept fields (no pass over J); the value is the fresh evaluation's to 1e-9 n, an
    edited state is evaluated from scratch, and a later replica call does not leave stale rows behind."""
    n, R = 4100, 4
    J, b, _ = _system(n, 17, False)
    rng = np.random.default_rng(2)
    sts = rng.integers(0, 2, size=(R, n)).astype(np.int8)
    d = _dense(J, b, False)
    for call in range(3):
        out = d.sweep_replicas(sts, [1.0, 1.2, 1.4, 1.6], 1, [5] * R, [call] * R, replicas=list(range(R)))
        fresh = d.energies(out)
        for r in (2, 0, 3, 1):
            d.set_state(out[r])
            assert abs(d.energy() - fresh[r]) <= 1e-9 * n, (call, r)
        edited = out[1].copy()
        edited[11] ^= 1
        d.set_state(edited)
        assert abs(d.energy() - d.energies(edited[None])[0]) <= 1e-9 * n
        # tsu_dense_energies takes the kept fields too; against a system that never swept replicas (one pass over J per state)
        if call == 0:
            d2 = _dense(J, b, False)
            np.testing.assert_allclose(fresh, d2.energies(out), rtol=0, atol=1e-9 * n)
            d2.close()
        sts = out[::-1].copy()  # the next call sees them in another order
    d.close()


def test_replica_batch_of_eleven_and_replayed_uniforms():
    # eleven replicas = a launch of eight and one of three (padded to four); replayed uniforms per replica
    n, R, k = 2560, 11, 2
    J, b, _ = _system(n, 77, True)
    sts = np.array([np.random.default_rng(r).integers(0, 2, size=n) for r in range(R)], dtype=np.int8)
    u = np.random.default_rng(5).random((R, k, n))
    temps = [0.7 + 0.1 * r for r in range(R)]
    d = _dense(J, b, True)
    out = d.sweep_replicas(sts, temps, k, [0] * R, [0] * R, replay_uniforms=u)
    assert d.launch_counts()[0] == 2
    for r in range(R):
        want = ora.c_dense_sweep_replay(sts[r].astype(np.int64), J, b, temps[r], u[r])
        np.testing.assert_array_equal(out[r], want.astype(np.int8), err_msg=f"replica {r}")
    d.close()


def test_sampling_run_and_annealing_schedule_with_a_callers_order_in_one_launch():
    n = 2304
    J, b, s0 = _system(n, 31, False)
    J64 = J.astype(np.float64)
    rng = np.random.default_rng(8)
    d = _dense(J, b, False)
    d.set_state(s0)
    order = np.array([rng.permutation(n) for _ in range(2 + 3 * 2)])
    got = d.sample(0.9, 2, 2, 3, seed=21, sweep0=10, order=order)
    assert d.launch_counts()[0] == 1
    want = ora.dense_sweep_philox(s0, J64, b, 0.9, 2, 21, sweep0=10, order=order[:2])
    for k in range(3):
        want = ora.dense_sweep_philox(want, J64, b, 0.9, 2, 21, sweep0=12 + 2 * k, order=order[2 + 2 * k:4 + 2 * k])
        np.testing.assert_array_equal(got[k], want, err_msg=f"sample {k}")
    temps = [2.0 * (0.05 / 2.0) ** (k / 5) for k in range(5)]
    order = np.array([rng.permutation(n) for _ in range(5)])
    got = d.anneal(temps, seed=22, sweep0=40, order=order)
    assert d.launch_counts()[0] == 2
    for k, T in enumerate(temps):
        want = ora.dense_sweep_philox(want, J64, b, T, 1, 22, sweep0=40 + k, order=order[k:k + 1])
        np.testing.assert_array_equal(got[k], want, err_msg=f"annealing step {k}")
    d.close()


def test_config3_16384_properties_over_a_long_call():
    """BASELINE configs[2] at full size, beyond what the oracle is asked to repeat: (i) one call of 12 sweeps == 12 calls of one
    sweep == calls of 5 + 7 (fields handed from sweep to sweep and from call to call); (ii) superblocks of 4096 and 8192 agree;
    (iii) the energy of the final state from the kept fields == a fresh evaluation (tsu_dense_energies) within 1e-9 n."""
    n = 16384
    J, _, s0 = _system(n, 42, False, bias=False)
    d = _dense(J, None, False)
    finals = []
    for plan in ([12], [1] * 12, [5, 7]):
        d.set_state(s0)
        done = 0
        for k in plan:
            d.sweep(1.0, k, seed=9, sweep0=done)
            done += k
        finals.append(d.get_state())
    with _env(TSU_K2_OWN_SB=8192):
        d.set_state(s0)
        d.sweep(1.0, 12, seed=9, sweep0=0)
        finals.append(d.get_state())
    for f in finals[1:]:
        np.testing.assert_array_equal(f, finals[0])
    assert d.launch_counts() == (1 + 12 + 2 + 1, 0)
    e_kept = d.energy()
    e_fresh = d.energies(finals[0][None, :])[0]
    assert abs(e_kept - e_fresh) <= 1e-9 * n
    # the first two sweeps against the oracle (the rest is pinned by the equalities above)
    d.set_state(s0)
    d.sweep(1.0, 2, seed=9, sweep0=0)
    np.testing.assert_array_equal(d.get_state(), ora.dense_sweep_philox(s0, J.astype(np.float64), None, 1.0, 2, 9, sweep0=0))
    d.close()


def test_a_run_that_gives_up_half_way_is_redone_on_the_other_paths():
    """k2_own's waits are bounded: a lost peer (a GPU shared with another long-running kernel) makes it give up with the state half
    updated.  The callers then restore the state of the call's start and the other paths redo the call -- same result.  Forced here
    by TSU_K2_OWN_TEST_FAIL (the kernel stops at a given superblock as if a wait had expired)."""
    n = 8192
    J, b, s0 = _system(n, 19, False)
    J64 = J.astype(np.float64)
    with _env(TSU_K2_OWN_TEST_FAIL=5):
        d = _dense(J, b, False)
        d.set_state(s0)
        d.sweep(1.0, 3, seed=7, sweep0=2)  # natural order: 2 superblocks x 3 sweeps, gives up in the last one -> the pipeline
        assert d.launch_counts() == (0, 1)
        np.testing.assert_array_equal(d.get_state(), ora.dense_sweep_philox(s0, J64, b, 1.0, 3, 7, sweep0=2))
        d.sweep(0.8, 1, seed=7, sweep0=5)  # later calls skip k2_own
        assert d.launch_counts() == (0, 2)
        d.close()
        # a recorded run (sample_boltzmann in one launch)
        d = _dense(J, b, False)
        d.set_state(s0)
        got = d.sample(0.9, 1, 1, 3, seed=3, sweep0=0)
        want = ora.dense_sweep_philox(s0, J64, b, 0.9, 1, 3, sweep0=0)
        for k in range(3):
            want = ora.dense_sweep_philox(want, J64, b, 0.9, 1, 3, sweep0=1 + k)
            np.testing.assert_array_equal(got[k], want, err_msg=f"sample {k}")
        d.close()
    n = 2304
    J, b, s0 = _system(n, 23, False)
    J64 = J.astype(np.float64)
    order = np.array([np.random.default_rng(1).permutation(n) for _ in range(2)])
    with _env(TSU_K2_OWN_TEST_FAIL=1):
        d = _dense(J, b, False)
        d.set_state(s0)
        d.sweep(0.9, 2, seed=3, sweep0=5, order=order)  # a caller's order: the block-by-block path redoes it
        assert d.launch_counts() == (0, 0)
        np.testing.assert_array_equal(d.get_state(), ora.dense_sweep_philox(s0, J64, b, 0.9, 2, 3, sweep0=5, order=order))
        sts = np.array([np.random.default_rng(r).integers(0, 2, size=n) for r in range(3)], dtype=np.int8)
        out = d.sweep_replicas(sts, [1.0, 0.7, 1.4], 2, [1, 2, 3], [0, 0, 0], replicas=[0, 1, 2])  # replicas: one after the other
        for r in range(3):
            np.testing.assert_array_equal(out[r], ora.dense_sweep_philox(sts[r], J64, b, [1.0, 0.7, 1.4][r], 2, 1 + r, sweep0=0, replica=r))
        d.close()
