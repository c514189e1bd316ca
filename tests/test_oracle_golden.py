"""The oracle pinned against golden vectors produced by the unmodified reference
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from oracle import oracle as ora


# ----------------------------------------------------------------------------- Philox KATs
# Random123 known-answer tests (also listed in SURVEY.md section 7)
KATS = [
    ((0, 0, 0, 0), (0, 0), (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
    ((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2, (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
    ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0),
     (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)),
]


@pytest.mark.parametrize("ctr,key,expect", KATS)
def test_philox_kat(ctr, key, expect):
    assert tuple(int(v) for v in ora.philox4x32_10(ctr, key)) == expect


# ----------------------------------------------------------------------------- G7 sigmoid
def test_sigmoid_table(golden):
    g = golden("g7_sigmoid")
    for x, y in zip(g["x"], g["y"]):
        assert ora.ref_sigmoid(float(x)) == y
        # C helper uses libm exp; NumPy's exp may differ in the last ulp
        assert abs(ora.c_sigmoid(float(x)) - y) <= 2e-16 * max(1.0, abs(y))
    assert ora.c_sigmoid(20.0000001) == 1.0 and ora.c_sigmoid(-20.0000001) == 0.0
    assert 0.0 < ora.c_sigmoid(-20.0) and ora.c_sigmoid(20.0) < 1.0


# ----------------------------------------------------------------------------- G1/G2 dense Gibbs
@pytest.mark.parametrize("name", ["g1_dense_sequential", "g2_dense_random"])
def test_dense_sample_boltzmann_replay(golden, name):
    g = golden(name)
    order = g["perms"] if name.startswith("g2") else None
    args = (g["J"], g["bias"], float(g["T"]), int(g["burnin"]), int(g["n_sweeps"]), int(g["n_samples"]),
            g["init"], g["uniforms"])
    out = ora.ref_sample_boltzmann(*args, order=order)
    assert out.dtype == g["samples"].dtype
    np.testing.assert_array_equal(out, g["samples"])
    # C restatement: same trajectory
    st = g["init"].copy()
    n_total = int(g["burnin"]) + int(g["n_sweeps"]) * int(g["n_samples"])
    pos, rows = int(g["burnin"]), []
    st = ora.c_dense_sweep_replay(st, g["J"], g["bias"], float(g["T"]), g["uniforms"][:pos],
                                  None if order is None else order[:pos])
    for _ in range(int(g["n_samples"])):
        ns = int(g["n_sweeps"])
        st = ora.c_dense_sweep_replay(st, g["J"], g["bias"], float(g["T"]), g["uniforms"][pos:pos + ns],
                                      None if order is None else order[pos:pos + ns])
        pos += ns
        rows.append(st.copy())
    assert pos == n_total
    np.testing.assert_array_equal(np.array(rows), g["samples"])


def test_dense_asymmetric_no_bias(golden):
    g = golden("g1b_dense_asymmetric")
    out = ora.ref_gibbs_sweep(g["init"], g["J"], None, float(g["T"]), g["uniforms"])
    np.testing.assert_array_equal(out, g["sweep_out"])
    np.testing.assert_array_equal(ora.c_dense_sweep_replay(g["init"], g["J"], None, float(g["T"]), g["uniforms"]),
                                  g["sweep_out"])
    sb = ora.ref_sample_boltzmann(g["J"], None, float(g["T"]), 0, 1, 3, g["init"], g["sb_uniforms"])
    np.testing.assert_array_equal(sb, g["sb_samples"])


# ----------------------------------------------------------------------------- G3 lattice builder
def _g3_cases(g):
    keys = sorted({k.rsplit("_", 1)[0] for k in g.files})
    for key in keys:
        r = int(key[1:key.index("c")])
        c = int(key[key.index("c") + 1:key.index("p")])
        per = bool(int(key[key.index("p") + 1]))
        Jc = float(key[key.index("J") + 1:key.index("h")])
        hf = float(key[key.index("h") + 1:])
        yield key, r, c, per, Jc, hf


def test_grid_builder_and_conversions(golden):
    g = golden("g3_grid_builder")
    n = 0
    for key, r, c, per, Jc, hf in _g3_cases(g):
        J = ora.ref_grid_coupling(r, c, Jc, per)
        np.testing.assert_array_equal(J, g[key + "_J"])
        np.testing.assert_array_equal(ora.ref_bit_coupling(J), g[key + "_Jbit"])
        h = np.ones(r * c) * hf
        np.testing.assert_array_equal(ora.ref_bit_bias(J, h, ora.MODE_COMPAT), g[key + "_hbit"])
        for s, e in zip(g[key + "_states"], g[key + "_energy"]):
            assert -0.5 * s.dot(J).dot(s) - h.dot(s) == e
        n += 1
    assert n == 18


def test_grid_sample_replay(golden):
    """IsingGrid.sample(4) with np.random.seed(11): replay MT19937 through the restatement."""
    g = golden("g3_grid_builder")
    for key, r, c, per, Jc, hf in _g3_cases(g):
        J = ora.ref_grid_coupling(r, c, Jc, per)
        h = np.ones(r * c) * hf
        n = r * c
        burnin, n_sweeps, n_samples = 2, 1, 4
        np.random.seed(11)
        init = np.random.randint(0, 2, size=n)
        u = np.random.rand(burnin + n_sweeps * n_samples, n)
        bits = ora.ref_sample_boltzmann(ora.ref_bit_coupling(J), ora.ref_bit_bias(J, h), 1.7, burnin, n_sweeps,
                                        n_samples, init, u)
        np.testing.assert_array_equal(2 * bits - 1, g[key + "_samples"])


# ----------------------------------------------------------------------------- G4 config-1 trajectory
@pytest.mark.parametrize("per", [0, 1])
@pytest.mark.parametrize("mode", ["compat", "physical"])
def test_config1_trajectory(golden, per, mode):
    """IsingGrid 32x32, T=2.5, seed 42: states after 1/10/100/1000 sweeps (C restatement, replayed MT19937)."""
    g = golden("g4_config1_trajectory")
    key = f"p{per}_{mode}"
    J = ora.ref_grid_coupling(32, 32, 1.0, bool(per))
    h = np.zeros(1024)
    hb = ora.ref_bit_bias(J, h, ora.MODE_COMPAT if mode == "compat" else ora.MODE_PHYSICAL)
    np.testing.assert_array_equal(hb, g[key + "_hbit"])
    np.random.seed(42)
    bits = np.random.randint(0, 2, size=1024)
    np.testing.assert_array_equal(bits.astype(np.int8), g[key + "_init"])
    done = 0
    assert list(g["checkpoints"]) == [1, 10, 100, 1000]
    for cp, want, M, E in zip(g["checkpoints"], g[key + "_states"], g[key + "_M"], g[key + "_E"]):
        u = np.random.rand(int(cp) - done, 1024)
        bits = ora.c_dense_sweep_replay(bits, 4 * J, hb, 2.5, u)
        done = int(cp)
        s = 2 * bits - 1
        np.testing.assert_array_equal(s.astype(np.int8), want)
        assert s.sum() / 1024 == M
        sa, sb = ora.ising2d_observables(s.reshape(32, 32).astype(np.int8), bool(per))
        assert sa == s.sum() and -1.0 * sb / 1024 == E


# ----------------------------------------------------------------------------- G5 Langevin
def test_langevin_step_and_chain(golden):
    g = golden("g5_langevin")
    T, dt, fr = float(g["T"]), float(g["dt"]), float(g["friction"])
    x1 = ora.ref_langevin_step(g["x"], g["grad"], g["noise"], T, dt, fr)
    np.testing.assert_array_equal(x1, g["x_next"])
    np.testing.assert_array_equal(ora.c_langevin_step_f64(g["x"], g["grad"], g["noise"], T, dt, fr), g["x_next"])

    def energy(v):
        return float((v ** 2).sum())

    for p, gr in zip(g["grad_pts"], g["grads"]):
        np.testing.assert_array_equal(ora.ref_numerical_gradient(energy, p), gr)
    samples, traj = ora.ref_sample_from_energy(energy, g["x0"], 3, T, dt, fr, int(g["n_burnin"]), int(g["n_steps"]),
                                               g["draws"])
    np.testing.assert_array_equal(samples, g["samples"])
    np.testing.assert_array_equal(np.array(traj), g["trajectory"])


# ----------------------------------------------------------------------------- G6 energies
def test_grid_observables_of_a_samples_array(golden):
    """magnetization / susceptibility / specific_heat of the reference on a fixed samples array (ising.py:183-233),
    restated on the oracle's lattice builder: E(s) = -1/2 s'Js - h's, C = var(E) / (T^2 N), chi = var(M) N / T."""
    g = golden("g6_observables")
    J = ora.ref_grid_coupling(4, 6, 0.8, True)
    h = np.ones(24) * 0.25
    S = g["grid_samples"].astype(float)
    E = np.array([-0.5 * s.dot(J).dot(s) - h.dot(s) for s in S])
    np.testing.assert_allclose(E, g["grid_E"], rtol=0, atol=1e-12)
    T, N = 1.9, 24
    assert abs((np.mean(E ** 2) - np.mean(E) ** 2) / (T ** 2 * N) - float(g["grid_C"])) < 1e-12
    M = S.sum(axis=1) / N
    assert np.mean(S.sum(axis=1)) / N == float(g["grid_M"])
    assert abs((np.mean(M ** 2) - np.mean(M) ** 2) * N / T - float(g["grid_chi"])) < 1e-12


def test_dense_energy(golden):
    g = golden("g6_observables")
    for b, e, e0 in zip(g["dense_bits"], g["dense_E"], g["dense_E_nobias"]):
        assert ora.ref_compute_energy(b, g["dense_J"], g["dense_b"]) == e
        assert ora.ref_compute_energy(b, g["dense_J"]) == e0
        assert abs(ora.c_dense_energy(b, g["dense_J"], g["dense_b"]) - e) < 1e-12
