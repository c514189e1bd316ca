"""K2 pipeline kernel (k2_pipe: streamers + solver teams, csrc/dense_coop.hip) against the oracle, bit for bit, at sizes
around its superblock sizes (4096 / 8192): one and several superblocks, partial last superblock, width not a multiple of the
vector width's strip, f32 / f64 couplings, a bias, and long calls in which the fields are handed from sweep to sweep (a
single-superblock system once raced here: its rows are committed by all solver workgroups just before the next solve)."""
import numpy as np
import pytest

from oracle import oracle as ora

pytestmark = pytest.mark.gpu


def _system(n, seed, f64=False, bias=True):
    rng = np.random.default_rng(seed)
    G = rng.standard_normal((n, n)).astype(np.float32)
    J = ((G + G.T) / 2 / np.sqrt(n)).astype(np.float32)
    np.fill_diagonal(J, 0.0)
    b = rng.normal(size=n) * 0.1 if bias else None
    return (J.astype(np.float64) if f64 else J), b, rng.integers(0, 2, size=n).astype(np.int8)


@pytest.mark.parametrize("n,T,f64", [(452, 1.0, True), (580, 0.6, False), (768, 1.0, False), (1000, 2.0, True), (1020, 1.0, False), (1024, 1.0, False),
                                     (2048, 0.8, False), (2052, 1.0, True), (4096, 1.0, False), (4096, 1.0, True), (5000, 0.7, False), (6144, 1.0, False),
                                     (8192, 1.0, False), (8196, 1.0, False), (9000, 1.3, True), (12288, 1.0, False), (12292, 0.4, False)])
def test_pipeline_sweeps_match_oracle(n, T, f64):
    from tsu import _hip
    J, b, s0 = _system(n, n, f64)
    d = _hip.DenseSystem(J, b, _hip.DTYPE_F64 if f64 else _hip.DTYPE_F32)
    d.set_state(s0)
    d.sweep(T, 3, seed=7, sweep0=2)
    want = ora.dense_sweep_philox(s0, np.asarray(J, dtype=np.float64), b, T, 3, 7, sweep0=2)
    np.testing.assert_array_equal(d.get_state(), want)
    d.close()


@pytest.mark.parametrize("n", [4096, 6144, 8192])
def test_pipeline_long_call_hands_fields_from_sweep_to_sweep(n):
    from tsu import _hip
    J, _, s0 = _system(n, n + 1, bias=False)
    d = _hip.DenseSystem(J, None, _hip.DTYPE_F32)
    d.set_state(s0)
    want, done = s0, 0
    for k in (4, 20):
        d.sweep(1.0, k, seed=1, sweep0=done)
        want = ora.dense_sweep_philox(want, J.astype(np.float64), None, 1.0, k, 1, sweep0=done)
        done += k
        np.testing.assert_array_equal(d.get_state(), want)
    assert abs(d.energy() - ora.c_dense_energy(want.astype(np.int64), J.astype(np.float64))) < 1e-6 * n
    d.close()


def test_pipeline_replays_numpy_uniforms():
    """rng="numpy" at a size the pipeline takes: the kernel consumes the caller's np.random.rand() doubles in visiting order
    (gibbs.py:126) and must return the reference loop's states (oracle: C replay of the sequential loop)."""
    from tsu import _hip
    n, k = 2048, 3
    J, b, s0 = _system(n, 5, f64=True)
    u = np.random.default_rng(9).random((k, n))
    d = _hip.DenseSystem(J, b, _hip.DTYPE_F64)
    d.set_state(s0)
    d.sweep(0.9, k, replay_uniforms=u)
    want = ora.c_dense_sweep_replay(s0.astype(np.int64), J, b, 0.9, u)
    np.testing.assert_array_equal(d.get_state(), want.astype(np.int8))
    d.close()


@pytest.mark.parametrize("n,f64", [(1024, False), (2052, True), (4096, False), (6144, False)])
def test_fields_handed_from_call_to_call_give_the_same_trajectory_and_energy(n, f64):
    """Loops of short calls (annealing, tempering): from the second consecutive call on the pipeline keeps the fields of the final
    state and the next call (and tsu_dense_energy) starts from them instead of streaming J again.  Every call's state must equal
    the oracle's, the energies the oracle's within rounding, a set_state in between must drop the kept fields."""
    from tsu import _hip
    J, b, s0 = _system(n, 3 * n + 1, f64)
    J64 = np.asarray(J, dtype=np.float64)
    d = _hip.DenseSystem(J, b, _hip.DTYPE_F64 if f64 else _hip.DTYPE_F32)
    d.set_state(s0)
    want, sw = s0, 5
    calls = [(1, 1.0), (1, 0.9), (3, 0.8), (1, 0.7), (2, 1.3), (1, 0.6), (1, 0.5)]
    for k, (ns, T) in enumerate(calls):
        d.sweep(T, ns, seed=11, sweep0=sw)
        want = ora.dense_sweep_philox(want, J64, b, T, ns, 11, sweep0=sw)
        sw += ns
        np.testing.assert_array_equal(d.get_state(), want, err_msg=f"call {k}")
        e = d.energy()
        assert abs(e - ora.ref_compute_energy(want.astype(np.int64), J64, b)) <= 1e-9 * n, k
        if k == 4:  # another state: the kept fields are not its fields
            want = 1 - want
            d.set_state(want.astype(np.int8))
            assert abs(d.energy() - ora.ref_compute_energy(want.astype(np.int64), J64, b)) <= 1e-9 * n
    d.close()


def test_kept_fields_are_recomputed_every_64_sweeps_across_calls():
    # 70 one-sweep calls cross the refresh period of the incremental fields (CO_REFRESH = 64 sweeps, counted across calls)
    from tsu import _hip
    n = 1024
    J, b, s0 = _system(n, 77)
    J64 = np.asarray(J, dtype=np.float64)
    d = _hip.DenseSystem(J, b, _hip.DTYPE_F32)
    d.set_state(s0)
    want = s0
    for k in range(70):
        d.sweep(1.0, 1, seed=5, sweep0=k)
        want = ora.dense_sweep_philox(want, J64, b, 1.0, 1, 5, sweep0=k)
        if k % 9 == 0 or k >= 60:
            np.testing.assert_array_equal(d.get_state(), want, err_msg=f"call {k}")
    np.testing.assert_array_equal(d.get_state(), want)
    d.close()


@pytest.mark.parametrize("n,f64", [(600, False), (1024, True), (2052, False), (4100, False)])
def test_whole_sampling_run_and_annealing_schedule_in_one_pipeline_launch(n, f64):
    """tsu_dense_sample / tsu_dense_anneal above the one-workgroup kernels: burn-in, the recorded states and a temperature per sweep
    all inside ONE launch of the pipeline kernel (it used to be one call per recorded state).  Every recorded state == oracle; with
    replayed uniforms too; a following call continues from the final state."""
    from tsu import _hip
    J, b, s0 = _system(n, 7 * n + 3, f64)
    J64 = np.asarray(J, dtype=np.float64)
    d = _hip.DenseSystem(J, b, _hip.DTYPE_F64 if f64 else _hip.DTYPE_F32)
    # sample_boltzmann: 3 burn-in sweeps, then 4 x (2 sweeps, record)
    d.set_state(s0)
    got = d.sample(0.9, 3, 2, 4, seed=21, sweep0=10)
    want = ora.dense_sweep_philox(s0, J64, b, 0.9, 3, 21, sweep0=10)
    for k in range(4):
        want = ora.dense_sweep_philox(want, J64, b, 0.9, 2, 21, sweep0=13 + 2 * k)
        np.testing.assert_array_equal(got[k], want, err_msg=f"sample {k}")
    np.testing.assert_array_equal(d.get_state(), want)
    # an annealing schedule: one sweep per temperature, every state recorded; continues from the state above
    temps = [2.0 * (0.05 / 2.0) ** (k / 9) for k in range(9)]
    got = d.anneal(temps, seed=22, sweep0=40)
    for k, T in enumerate(temps):
        want = ora.dense_sweep_philox(want, J64, b, T, 1, 22, sweep0=40 + k)
        np.testing.assert_array_equal(got[k], want, err_msg=f"annealing step {k}")
    # replayed uniforms (the reference's own stream): 2 burn-in + 3 x 1 sweeps
    rng = np.random.default_rng(5)
    u = rng.random((5, n))
    d.set_state(s0)
    got = d.sample(1.1, 2, 1, 3, replay_uniforms=u)
    want = ora.c_dense_sweep_replay(s0.astype(np.int64), J64, b, 1.1, u[:2])
    for k in range(3):
        want = ora.c_dense_sweep_replay(want, J64, b, 1.1, u[2 + k:3 + k])
        np.testing.assert_array_equal(got[k], want.astype(np.int8), err_msg=f"replayed sample {k}")
    d.close()
