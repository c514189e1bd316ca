"""CPU-only tests: host logic of the drop-in layer, C-ABI library loads and exports every symbol the header
declares, and the product path fails loudly (no CPU fallback) when there is no GPU."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "tsu_hip.h")


def _declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tsu_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from tsu import _hip
    assert os.path.exists(_hip.LIB_PATH), "build libtsu_hip.so first (tsu-emulator_amd/csrc/build.sh)"
    lib = ctypes.CDLL(_hip.LIB_PATH)
    names = _declared_symbols()
    assert len(names) >= 39
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/tsu_hip.h but not exported"
    # the ctypes prototypes cover the header one to one
    assert sorted(_hip.SIGNATURES) == names
    assert lib.tsu_version() == 100


def test_threshold_helper_is_pure_host_code():
    """tsu_ising2d_thresholds needs no GPU; it must agree with the oracle's table."""
    from oracle import oracle as ora
    from tsu import _hip
    for J, h, T, mode in [(1.0, 0.0, 2.269185, 0), (1.0, 0.0, 2.5, 1), (-0.7, 0.3, 1.1, 0), (1.0, 0.0, 0.01, 1)]:
        np.testing.assert_array_equal(_hip.ising2d_thresholds(J, h, T, mode), ora.ising2d_thresholds(J, h, T, mode))
    with pytest.raises(ValueError, match="Temperature must be positive"):
        _hip.ising2d_thresholds(1.0, 0.0, 0.0, 0)


def _no_gpu():
    try:
        import torch
        return not torch.cuda.is_available()
    except Exception:
        return not os.path.exists("/dev/kfd")


@pytest.mark.skipif(not _no_gpu(), reason="needs a machine WITHOUT a GPU")
def test_product_path_fails_loudly_without_gpu():
    from tsu import _hip
    from tsu.core import ThermalSamplingUnit
    from tsu.gibbs import GibbsSampler
    from tsu.models import IsingGrid
    from tsu.models.ising import IsingModel2D
    with pytest.raises(_hip.HipUnavailableError):
        GibbsSampler().gibbs_sweep(np.array([0, 1]), np.eye(2))
    with pytest.raises(_hip.HipUnavailableError):
        GibbsSampler().sample_boltzmann(np.eye(3), n_samples=2)
    with pytest.raises(_hip.HipUnavailableError):
        GibbsSampler().compute_energy(np.array([0, 1]), np.eye(2))
    with pytest.raises(_hip.HipUnavailableError):
        IsingGrid((4, 4)).sample(2)
    with pytest.raises(_hip.HipUnavailableError):
        IsingModel2D(8)
    with pytest.raises(_hip.HipUnavailableError):
        ThermalSamplingUnit().sample_gaussian(n_samples=3)


def test_product_never_imports_the_oracle():
    # the measurement / development scripts under tools/ stay clear of the oracle too: only tests/, smoke() and bench.py's
    # cpu_baseline leg use it
    for f in os.listdir(os.path.join(ROOT, "tools")):
        if f.endswith((".py", ".sh")):
            text = open(os.path.join(ROOT, "tools", f)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M) and "libtsu_oracle" not in text, f
    pkg = os.path.join(ROOT, "tsu-emulator_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".sh")):
                text = open(os.path.join(dirpath, f)).read()
                # comments may NAME the oracle as the CPU twin; code must not import, include, link or load it
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                assert not re.search(r"#\s*include\s*[\"<][^\">]*oracle", text), f
                assert "libtsu_oracle" not in text and "oracle/_build" not in text and "oracle/_ref" not in text, f


# ----------------------------------------------------------------------------- config objects (reference semantics)
def test_gibbs_config_validation_messages():
    from tsu.gibbs import GibbsConfig
    c = GibbsConfig(temperature=1.0, n_burnin=100, n_sweeps=10)
    assert (c.temperature, c.n_burnin, c.n_sweeps, c.update_order) == (1.0, 100, 10, "sequential")
    with pytest.raises(ValueError, match="Temperature must be positive"):
        GibbsConfig(temperature=-1.0)
    with pytest.raises(ValueError, match="Temperature must be positive"):
        GibbsConfig(temperature=0)
    with pytest.raises(ValueError, match="Burn-in steps must be non-negative"):
        GibbsConfig(n_burnin=-10)
    with pytest.raises(ValueError, match="Number of sweeps must be positive"):
        GibbsConfig(n_sweeps=0)
    with pytest.raises(ValueError, match="Update order must be"):
        GibbsConfig(update_order="invalid")


def test_gibbs_scalar_helpers(golden):
    from tsu.gibbs import GibbsSampler
    s = GibbsSampler()
    assert abs(s._sigmoid(0.0) - 0.5) < 1e-12 and s._sigmoid(100) == 1.0 and s._sigmoid(-100) == 0.0
    g = golden("g7_sigmoid")
    for x, y in zip(g["x"], g["y"]):
        assert s._sigmoid(float(x)) == y
    state = np.array([1, 0, 1])
    coupling = np.array([[0, 1, 2], [1, 0, 1], [2, 1, 0]])
    assert s._compute_local_field(0, state, coupling) == 2.0
    assert s._compute_local_field(0, state, coupling, np.array([0.5, -0.5, 1.0])) == 2.5
    g6 = golden("g6_observables")
    for b, row in zip(g6["dense_bits"], g6["dense_field"]):
        for i in range(6):
            assert s._compute_local_field(i, b, g6["dense_J"], g6["dense_b"]) == row[i]
    np.random.seed(0)
    draws = [s.sample_conditional(0, np.array([1, 0, 1, 0]), np.eye(4)) for _ in range(200)]
    assert 0 in draws and 1 in draws


def test_gibbs_sampler_argument_errors():
    from tsu.gibbs import GibbsSampler
    with pytest.raises(ValueError, match="Coupling matrix must be square"):
        GibbsSampler().sample_boltzmann(np.zeros((3, 4)), n_samples=1)
    with pytest.raises(ValueError):
        GibbsSampler(rng="mt")


def test_hardware_emulator_arithmetic():
    from tsu.gibbs import HardwareEmulator
    hw = HardwareEmulator(n_bits=100, clock_speed_ghz=1.0, parallel_chains=1000)
    assert (hw.n_bits, hw.clock_speed_ghz, hw.parallel_chains, hw.ns_per_cycle) == (100, 1.0, 1000, 1.0)
    t = hw.estimate_hardware_time(n_samples=10000, n_sweeps_per_sample=10)
    assert t["time_per_sweep_ns"] == 100.0 and t["batches_needed"] == 10 and t["total_time_ns"] == 10000.0
    assert t["total_time_s"] == 1e-5 and t["speedup_vs_classical"] is None


def test_ising_config_and_model_host_parts(golden):
    from tsu.models import IsingChain, IsingGrid, IsingModel
    from tsu.models.ising import IsingConfig
    c = IsingConfig(temperature=2.0, external_field=0.5)
    assert (c.temperature, c.external_field, c.n_burnin, c.n_sweeps) == (2.0, 0.5, 100, 10)
    with pytest.raises(ValueError, match="Temperature must be positive"):
        IsingConfig(temperature=-1.0)
    m = IsingModel(n_spins=5)
    assert m.J.shape == (5, 5) and m.h.shape == (5,)
    m.set_coupling(0, 1, 2.0)
    assert m.J[0, 1] == 2.0 and m.J[1, 0] == 2.0
    m.set_external_field(np.array([1, -1, 0, 2, -2]))
    assert np.allclose(m.h, [1, -1, 0, 2, -2])
    with pytest.raises(ValueError, match="Field must have length 5"):
        m.set_external_field(np.zeros(4))
    # energies of the ferromagnetic 3-chain (reference tests/test_ising.py:47-73 values)
    m3 = IsingModel(3, IsingConfig(external_field=0))
    m3.set_coupling(0, 1, 1.0)
    m3.set_coupling(1, 2, 1.0)
    assert m3.energy(np.array([1, 1, 1])) == -2.0 and m3.energy(np.array([1, -1, 1])) == 2.0
    # conversions and the (compat / physical) bit bias
    np.testing.assert_array_equal(m3._spins_to_bits(np.array([-1, 1, -1])), [0, 1, 0])
    np.testing.assert_array_equal(m3._bits_to_spins(np.array([0, 1, 0])), [-1, 1, -1])
    g = golden("g3_grid_builder")
    for key in ("r3c4p0J-0.7h0.3", "r4c4p1J1.0h0.0", "r2c5p1J-0.7h0.3", "r1c4p1J1.0h0.0", "r5c1p1J-0.7h0.3"):
        r, cc = int(key[1]), int(key[3])
        per = bool(int(key[5]))
        Jc = float(key[key.index("J") + 1:key.index("h")])
        hf = float(key[key.index("h") + 1:])
        grid = IsingGrid((r, cc), J=Jc, config=IsingConfig(temperature=1.7, external_field=hf), periodic=per)
        np.testing.assert_array_equal(grid.J, g[key + "_J"])
        np.testing.assert_array_equal(grid._get_bit_coupling(), g[key + "_Jbit"])
        np.testing.assert_array_equal(grid._get_bit_bias(), g[key + "_hbit"])
        assert grid.rows == r and grid.cols == cc and grid.n_spins == r * cc
    grid = IsingGrid((3, 4), J=1.0, bias_mode="physical")
    np.testing.assert_array_equal(grid._get_bit_bias(), -2 * grid.J.sum(axis=1))
    flat = np.arange(12)
    assert grid._flat_to_grid(flat).shape == (3, 4) and np.array_equal(grid._grid_to_flat(grid._flat_to_grid(flat)), flat)
    chain = IsingChain(7, J=-1.2, config=IsingConfig(temperature=0.6, external_field=-0.4))
    g6 = golden("g6_observables")
    np.testing.assert_array_equal(chain.J, g6["chain_J"])
    np.testing.assert_array_equal(chain.h, g6["chain_h"])
    for s, e in zip(g6["chain_samples"], g6["chain_E"]):
        assert chain.energy(s) == e
    assert chain.magnetization(g6["chain_samples"]) == g6["chain_M"]
    # observables that take a samples array are host reductions over that array
    gg = IsingGrid((4, 6), J=0.8, config=IsingConfig(temperature=1.9, external_field=0.25), periodic=True)
    assert gg.magnetization(g6["grid_samples"]) == g6["grid_M"]
    assert abs(gg.susceptibility(g6["grid_samples"]) - g6["grid_chi"]) < 1e-12
    np.testing.assert_array_equal([gg.compute_domains(s) for s in g6["grid_samples"]], g6["grid_domains"])
    big = IsingGrid((4096, 4096), periodic=True)  # no N x N matrix is ever built
    with pytest.raises(MemoryError):
        big.J


def test_tsu_config_and_helpers(golden):
    from tsu.core import (ConfigurationError, QuadraticEnergy, SamplingError, ThermalSamplingUnit, TSUConfig, TSUError,
                          _recognise_quadratic, validate_distribution)
    assert TSUConfig(temperature=1.0, dt=0.01, n_steps=100).temperature == 1.0
    for bad in (dict(temperature=-1.0), dict(dt=-0.01), dict(dt=1.0), dict(n_steps=-10), dict(friction=0), dict(n_burnin=-1)):
        with pytest.raises(ConfigurationError):
            TSUConfig(**bad)
    assert issubclass(ConfigurationError, TSUError) and issubclass(SamplingError, TSUError)
    g = golden("g5_langevin")
    t = ThermalSamplingUnit(TSUConfig(temperature=float(g["T"]), dt=float(g["dt"]), friction=float(g["friction"]),
                                      n_burnin=int(g["n_burnin"]), n_steps=int(g["n_steps"])))
    np.random.seed(77)
    np.testing.assert_array_equal(t._langevin_step(g["x"], g["grad"]), g["x_next"])
    for p, gr in zip(g["grad_pts"], g["grads"]):
        np.testing.assert_array_equal(t._numerical_gradient(lambda v: float((v ** 2).sum()), p), gr)
    with pytest.raises(SamplingError):
        t.sample_from_energy(lambda v: 0.0, np.zeros(2), n_samples=0)
    with pytest.raises(SamplingError):
        t.sample_from_energy(lambda v: np.zeros(2), np.zeros(2))
    for bad in (lambda: t.p_bit(1.5), lambda: t.p_bit(0.5, n_samples=0), lambda: t.sample_gaussian(sigma=-1.0),
                lambda: t.sample_gaussian(n_samples=0)):
        with pytest.raises(ConfigurationError):
            bad()
    # quadratic recognition: exact for separable quadratics, None for anything else
    q = _recognise_quadratic(lambda v: float((v ** 2).sum()), np.zeros(5))
    assert q is not None and np.allclose(q.k, 2.0) and np.allclose(q.mu, 0.0) and abs(q.c) < 1e-12
    q = _recognise_quadratic(lambda v: 0.5 * ((float(np.atleast_1d(v)[0]) - 5.0) / 2.0) ** 2, np.array([5.0]))
    assert q is not None and np.allclose(q.k, 0.25) and np.allclose(q.mu, 5.0)
    assert _recognise_quadratic(lambda v: float(np.sum(v ** 4)), np.ones(3)) is None
    assert _recognise_quadratic(lambda v: float(v[0] * v[1]), np.ones(2)) is None
    assert _recognise_quadratic(lambda v: float(np.sum(np.abs(v))), np.ones(2)) is None
    qe = QuadraticEnergy(2.0, 1.0, 3.0)
    assert qe(np.array([2.0, 3.0])) == 0.5 * 2 * (1 + 4) + 3.0
    rng = np.random.default_rng(1)
    r = validate_distribution(rng.normal(size=4000), "gaussian", {"mu": 0, "sigma": 1})
    assert r["passes_ks_test"] and r["n_samples"] == 4000
    r = validate_distribution((rng.random(2000) < 0.7).astype(int), "bernoulli", {"p": 0.7})
    assert r["passes_test"] and abs(r["empirical_prob"] - 0.7) < 0.05


# ------------------------------------------------------------------ bench.py process layout (no GPU, no torch import)
def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_bench_launch_plan_spawns_one_rank_per_gpu():
    b = _bench_module()
    assert b.launch_plan(1, {}, []) == ("inline", None)
    mode, cmd = b.launch_plan(4, {}, ["--gpus", "4", "--steps", "5"], free_port=lambda: 12345)
    assert mode == "spawn"
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "12345"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "5"]
    # inside a launcher: inline when the world is the one asked for, an error (never a silent n_gpus) otherwise
    assert b.launch_plan(8, {"WORLD_SIZE": "8"}, []) == ("inline", None)
    assert b.launch_plan(8, {"WORLD_SIZE": "1"}, [])[0] == "error"
    assert b.launch_plan(1, {"WORLD_SIZE": "2"}, [])[0] == "error"
    assert b.launch_plan(0, {}, [])[0] == "error"


def test_bench_refuses_a_world_that_is_not_the_one_requested():
    import subprocess
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and not r.stdout.strip()


def test_parallel_tempering_does_not_leave_a_stale_bind_behind(monkeypatch):
    """bind(J1); parallel_tempering(J2) replaces the device copy by J2's: afterwards a call with J1 must not be served from it
    (ADVICE round 2).  Host logic only: the device system is a stand-in that remembers which matrix it was built from."""
    import numpy as np
    from tsu import gibbs, _hip

    class FakeSystem:
        def __init__(self, J, bias, dtype):
            self.J, self.n = np.array(J), np.asarray(J).shape[0]

        def close(self):
            pass

    monkeypatch.setattr(_hip, "DenseSystem", FakeSystem)
    s = gibbs.GibbsSampler(gibbs.GibbsConfig(n_burnin=0, n_sweeps=1), seed=1)
    monkeypatch.setattr(s, "_parallel_tempering_run", lambda *a, **k: (np.zeros((0, 4)), {}))
    J1, J2 = np.eye(4), 2.0 * np.eye(4)
    s.bind(J1)
    assert s._system(J1, None).J[0, 0] == 1.0
    s.parallel_tempering(J2, [1.0, 2.0], n_samples=0)
    assert s._system(J1, None).J[0, 0] == 1.0, "J1 served from J2's device copy"
    # a bind of the very arrays the tempering call uses survives it
    s.bind(J2)
    s.parallel_tempering(J2, [1.0, 2.0], n_samples=0)
    assert s._held is not None and s._held[0] is J2
    assert s._system(J2, None).J[0, 0] == 2.0


def test_sample_chains_checks_the_sweep_counter_like_the_c_abi(monkeypatch):
    import numpy as np
    import pytest
    from tsu import gibbs
    s = gibbs.GibbsSampler(gibbs.GibbsConfig(n_burnin=1, n_sweeps=1), seed=1)
    s._sweep_counter = 2 ** 32 - 5
    with pytest.raises(ValueError, match="sweep counter overflow"):
        s.sample_chains(np.eye(4), 4, 1)


def test_large_dim_energy_callables_are_probed_in_bounded_time_or_refused():
    """A callable energy with d > 4096: a uniform separable quadratic is recognised with ~140 evaluations whatever d is; anything
    else is refused with the descriptor to use instead of silently starting the O(d^2) finite-difference loop on the host."""
    import time
    import numpy as np
    import pytest
    from tsu import core
    t0 = time.perf_counter()
    q = core._recognise_quadratic(lambda x: (x ** 2).sum(), np.zeros(2 ** 18))
    assert q is not None and float(q.k) == 2.0 and float(q.mu) == 0.0 and q.c == 0.0
    q = core._recognise_quadratic(lambda x: 1.5 * ((x - 0.25) ** 2).sum() + 7.0, np.ones(2 ** 16))
    assert q is not None and float(q.k) == 3.0 and float(q.mu) == 0.25 and abs(q.c - 7.0) < 1e-6
    assert core._recognise_quadratic(lambda x: (x ** 4).sum(), np.zeros(2 ** 16)) is None
    assert core._recognise_quadratic(lambda x: (x ** 2).sum() + x[5] ** 2, np.zeros(2 ** 16)) is None  # one deviating coordinate
    assert core._recognise_quadratic(lambda x: (x ** 2).sum() + x[0] * x[1], np.zeros(2 ** 16)) is None  # coupled
    tsu = core.ThermalSamplingUnit(core.TSUConfig(n_burnin=10, n_steps=10), seed=1)
    with pytest.raises(core.SamplingError, match="QuadraticEnergy"):
        tsu.sample_boltzmann(lambda x: (x ** 4).sum(), n_samples=2, dim=2 ** 16)
    assert time.perf_counter() - t0 < 5.0


def test_frozen_arrays_skip_the_content_hash_and_thawed_ones_do_not(monkeypatch):
    """J.setflags(write=False) on a data-owning array: nobody can edit it in place, so the device copy made from it is reused
    without hashing 8 n^2 bytes per call (the reference idiom ``state = s.gibbs_sweep(state, J)`` at kernel speed); a writeable
    array is hashed on every call, an in-place edit of it is seen, and a bulk edit slipped in by thaw / refreeze trips the sample."""
    import numpy as np
    from tsu import gibbs, _hip

    class FakeSystem:
        built = 0

        def __init__(self, J, bias, dtype):
            FakeSystem.built += 1
            self.J = np.array(J)

        def close(self):
            pass

    calls = {"n": 0}
    real = gibbs._content_key

    def counting(a):
        calls["n"] += 1
        return real(a)

    monkeypatch.setattr(_hip, "DenseSystem", FakeSystem)
    monkeypatch.setattr(gibbs, "_content_key", counting)
    s = gibbs.GibbsSampler(seed=1)
    J = np.random.default_rng(0).random((300, 300))
    # writeable: hashed every call, one upload while unchanged, a new upload after an in-place edit
    s._system(J, None)
    s._system(J, None)
    assert calls["n"] == 4 and FakeSystem.built == 1  # (J and bias keys per call)
    J[3, 4] += 1.0
    assert s._system(J, None).J[3, 4] == J[3, 4] and FakeSystem.built == 2
    # frozen: hashed once, then O(1)
    J.setflags(write=False)
    n0 = calls["n"]
    s._system(J, None)
    s._system(J, None)
    s._system(J, None)
    assert calls["n"] == n0 + 2 and FakeSystem.built == 2
    # a frozen VIEW of a writeable array is not trusted (its base can be edited)
    K = np.random.default_rng(1).random((300, 300))
    V = K.view()
    V.setflags(write=False)
    n0 = calls["n"]
    s._system(V, None)
    s._system(V, None)
    assert calls["n"] == n0 + 4
    # thaw, bulk edit, refreeze between two calls: the sample trips
    s._system(J, None)
    built = FakeSystem.built
    s._system(J, None)
    assert FakeSystem.built == built
    J.setflags(write=True)
    J *= 2.0
    J.setflags(write=False)
    assert s._system(J, None).J[0, 0] == J[0, 0] and FakeSystem.built == built + 1


def test_coupled_langevin_twin_is_the_plain_formula():
    """oracle ora_langevin_coupled_f32 == x + (-(A x + b) dt / gamma) + sqrt(2 T dt / gamma) xi with the K3 normals, every element
    from the OLD state (tsu/core.py:146-150), in float32 with the gradient rounded once."""
    from oracle import oracle as ora
    rng = np.random.default_rng(3)
    d, chains = 7, 2
    A = rng.standard_normal((d, d)).astype(np.float32)
    A = np.triu(A) + np.triu(A, 1).T + 4 * np.eye(d, dtype=np.float32)
    b = rng.standard_normal(d).astype(np.float32)
    x = rng.standard_normal((chains, d)).astype(np.float32)
    got, traj = ora.langevin_coupled_f32(x, A, b, 3, 0.01, 2.0, 0.5, 77, step0=4, chain0=1, trajectory=True)
    want = x.copy()
    a, scale = np.float32(0.01) / np.float32(2.0), np.sqrt(np.float32(2.0) * np.float32(0.5) * np.float32(0.01) / np.float32(2.0))
    for s in range(3):
        for c in range(chains):
            g = (A.astype(np.float64) @ want[c].astype(np.float64) + b.astype(np.float64)).astype(np.float32)
            xi = np.concatenate([ora.langevin_normals_f32(q, 1 + c, 4 + s, 77) for q in range((d + 3) // 4)])[:d]
            drift = (want[c].astype(np.float64) + (-g.astype(np.float64)) * np.float64(a)).astype(np.float32)  # fmaf(-g, a, x)
            want[c] = (drift.astype(np.float64) + np.float64(scale) * xi.astype(np.float64)).astype(np.float32)  # fmaf(scale, xi, .)
        np.testing.assert_array_equal(traj[s], want)
    np.testing.assert_array_equal(got, want)


def test_coupled_quadratics_are_recognised_by_probing_and_others_are_not():
    from tsu import core
    M = np.array([[2.0, 0.5, 0.0], [0.5, 1.0, -0.25], [0.0, -0.25, 1.5]])
    v = np.array([1.0, 0.0, -0.5])
    q = core._recognise_quadratic(lambda x: float(0.5 * x @ M @ x + v @ x + 3.0), np.array([0.3, -0.2, 0.1]))
    assert isinstance(q, core.QuadraticForm)
    np.testing.assert_allclose(q.A, M, atol=1e-12)
    np.testing.assert_allclose(q.b, v, atol=1e-12)
    assert abs(q.c - 3.0) < 1e-9
    # separable ones stay on the separable descriptor
    assert isinstance(core._recognise_quadratic(lambda x: float((x ** 2).sum()), np.zeros(3)), core.QuadraticEnergy)
    # not quadratic, indefinite, too large for the coupled probe: host path
    assert core._recognise_quadratic(lambda x: float(np.sum(x ** 4) + x[0] * x[1]), np.zeros(3)) is None
    assert core._recognise_quadratic(lambda x: float(x[0] * x[1]), np.zeros(2)) is None
    big = np.eye(80) + 0.01
    assert core._recognise_quadratic(lambda x: float(0.5 * x @ big @ x), np.zeros(80)) is None
    # the descriptor: symmetric part, call, gradient
    f = core.QuadraticForm([[1.0, 1.0], [0.0, 2.0]], [1.0, -1.0], 0.5)
    x = np.array([2.0, 3.0])
    assert f(x) == 0.5 * (1 * 4 + 1 * 6 + 2 * 9) + (2 - 3) + 0.5
    np.testing.assert_array_equal(f.gradient(x), np.array([[1.0, 0.5], [0.5, 2.0]]) @ x + [1.0, -1.0])
    with pytest.raises(core.ConfigurationError):
        core.QuadraticForm(np.zeros((2, 3)))
