"""BASELINE.json's full-size configurations on the GPU (-m gpu): direct comparison with the oracle where the oracle
finishes in seconds, size-independent properties (kernel-to-kernel equality, decomposition invariance, stationary
moments) beyond that.  Bit-exact for spins / bits; the fp32 Langevin tolerance is the one of test_hip_parity."""
import zlib

import numpy as np
import pytest

from oracle import oracle as ora

pytestmark = pytest.mark.gpu
T_C = 2.269185314213022


@pytest.fixture(scope="module")
def hip():
    from tsu import _hip
    _hip.Context.default()
    return _hip


def test_config2_lattice_4096_matches_oracle_and_generic_kernel(hip):
    """configs[1]: 4096 x 4096 at T_c, periodic, physical mode.  Tiled kernel (8 sweeps per launch) == oracle C port
    == generic kernel after 16 sweeps; observables equal the oracle's."""
    L, seed, n = 4096, 42, 16
    table = ora.ising2d_thresholds(1.0, 0.0, T_C, ora.MODE_PHYSICAL)
    lat = hip.Lattice(L, L, True)
    lat.set_kernel(hip.KERNEL_TILED, 8)
    lat.randomize(seed)
    s0 = lat.get_spins()
    assert zlib.crc32(s0.tobytes()) == zlib.crc32(ora.ising2d_randomize(L, L, seed).tobytes())
    lat.set_thresholds(table)
    lat.sweep(n, seed, sweep0=0)
    got = lat.get_spins()
    want = ora.ising2d_sweep(s0, True, table, n, seed, sweep0=0)
    np.testing.assert_array_equal(got, want)
    assert lat.observables() == ora.ising2d_observables(want, True)
    gen = hip.Lattice(L, L, True)
    gen.set_kernel(hip.KERNEL_GENERIC)
    gen.set_spins(s0)
    gen.set_thresholds(table)
    gen.sweep(n, seed, sweep0=0)
    np.testing.assert_array_equal(gen.get_spins(), got)
    # a longer call: nine tile-resident generations (strips exchanged between neighbouring tiles) == 144 generic launches
    lat.sweep(72, seed, sweep0=n)
    gen.sweep(72, seed, sweep0=n)
    assert zlib.crc32(lat.get_spins().tobytes()) == zlib.crc32(gen.get_spins().tobytes())
    assert lat.observables() == gen.observables()
    lat.close()
    gen.close()


def test_lattice_2pow25_sites_resident_256_row_tiles(hip):
    """4096 x 8192 (the largest lattice that stays resident in LDS: 256 tiles of 256 x 512): 40 sweeps == generic kernel."""
    sums = []
    for kern in (hip.KERNEL_AUTO, hip.KERNEL_GENERIC):
        lat = hip.Lattice(4096, 8192, True)
        lat.set_kernel(kern)
        lat.randomize(9)
        lat.set_model(1.0, 0.02, T_C)
        lat.sweep(40, 9, sweep0=1)
        sums.append((zlib.crc32(lat.get_spins().tobytes()), lat.observables()))
        lat.close()
    assert sums[0] == sums[1]


def test_config2_lattice_8192_kernels_agree(hip):
    """8192 x 8192 (the roofline target's size): what AUTO picks there -- the nibble-plane tile-resident kernel the 0.91 of the roofline
    is measured on -- == tiled k=5 == tiled k=8 == generic kernel, by checksum of checksums."""
    L, seed, n = 8192, 7, 40
    sums = []
    for kern, k in ((hip.KERNEL_AUTO, 0), (hip.KERNEL_TILED, 5), (hip.KERNEL_TILED, 8), (hip.KERNEL_GENERIC, 0)):
        lat = hip.Lattice(L, L, True)
        lat.set_kernel(kern, k)
        lat.randomize(seed)
        lat.set_model(1.0, 0.0, T_C)
        lat.sweep(n, seed, sweep0=3)
        s = lat.get_spins()
        sums.append((zlib.crc32(s.tobytes()), lat.observables()))
        lat.close()
    assert sums[0] == sums[1] == sums[2] == sums[3]
    # 40 sweeps from a random start at T_c: energy per site is already near -1.35 (far from -sqrt 2, critical slowing down)
    assert -1.42 < -sums[0][1][1] / (L * L) < -1.25


@pytest.mark.parametrize("rows,cols,sweeps", [(16384, 4096, (19, 8)),    # 2^26 sites: 256 nibble-plane tiles of 512 x 512, resident
                                              (5888, 6144, (9, 3)),      # ragged tile rows: 256 x 512 nibble tiles, one launch per 8 sweeps
                                              (8192, 16384, (11,))])     # 2^27 sites, whole tiles
def test_nibble_plane_lattices_chosen_automatically_equal_the_generic_kernel(hip, rows, cols, sweeps):
    """Lattices above 2^25 sites take the nibble colour planes (csrc/ising2d_tiled.hip: pick_variant): the automatic choice
    must give the generic kernel's lattice, by checksum and observables, over several calls (sweep counters run on)."""
    lats = []
    for kern in (hip.KERNEL_AUTO, hip.KERNEL_GENERIC):
        lat = hip.Lattice(rows, cols, True)
        lat.set_kernel(kern)
        lat.randomize(13)
        lat.set_model(1.0, -0.01, T_C)
        lats.append(lat)
    done = 0
    for n in sweeps:
        res = []
        for lat in lats:
            lat.sweep(n, 13, sweep0=done)
            res.append((zlib.crc32(lat.get_spins().tobytes()), lat.observables()))
        done += n
        assert res[0] == res[1], (rows, cols, done)
    for lat in lats:
        lat.close()


def test_config4_16384_as_eight_slabs_equals_one_lattice(hip):
    """configs[3]: 16384 x 16384 cut into 8 row slabs of 2048 x 16384 (ghost rows refreshed by the test every 8 sweeps,
    the exchange RCCL performs between ranks) == the same lattice swept whole.  Row checksums compared per slab."""
    L, P, seed, k, rounds = 16384, 8, 11, 8, 2
    per, ghost = L // P, 2 * k
    whole = hip.Lattice(L, L, True)
    whole.randomize(seed)
    whole.set_model(1.0, 0.0, T_C)
    slabs = [hip.Lattice(per, L, True, total_rows=L, row0=i * per, ghost=ghost) for i in range(P)]
    for s in slabs:
        s.randomize(seed)
        s.set_model(1.0, 0.0, T_C)
    for r in range(rounds):
        tops = [s.get_spins(0, ghost) for s in slabs]
        bots = [s.get_spins(per - ghost, ghost) for s in slabs]
        for i, s in enumerate(slabs):
            s.set_spins(bots[(i - 1) % P], row_first=-ghost)
            s.set_spins(tops[(i + 1) % P], row_first=per)
        for s in slabs:
            s.sweep(k, seed, sweep0=r * k)
        whole.sweep(k, seed, sweep0=r * k)
    for i, s in enumerate(slabs):
        assert zlib.crc32(s.get_spins().tobytes()) == zlib.crc32(whole.get_spins(i * per, per).tobytes()), f"slab {i}"
        s.close()
    whole.close()


def test_config3_dense_16384_matches_oracle(hip):
    """configs[2]: dense spin glass, N = 16384, fp32 couplings, T = 1: two sweeps bit-exact against the oracle."""
    n = 16384
    rng = np.random.default_rng(42)
    G = rng.standard_normal((n, n), dtype=np.float32)
    J = (G + G.T) / np.float32(2 * np.sqrt(n))
    np.fill_diagonal(J, 0.0)
    del G
    st = rng.integers(0, 2, size=n).astype(np.int8)
    d = hip.DenseSystem(J, None, hip.DTYPE_F32)
    d.set_state(st)
    d.sweep(1.0, 2, seed=9, sweep0=0)
    want = ora.dense_sweep_philox(st, J.astype(np.float64), None, 1.0, 2, 9, sweep0=0)
    np.testing.assert_array_equal(d.get_state(), want)
    d.close()


def test_config5_langevin_2pow20_matches_oracle_and_stationary_variance(hip):
    """configs[4]: d = 2^20, E = sum x^2, fp32: 10 steps against the oracle (2e-4 absolute), then 600 steps to the
    stationary variance T / (k (1 - k dt / 2 gamma)) = 0.505051 (standard error of the estimate 7e-4)."""
    dim = 1 << 20
    lc = hip.LangevinChains(1, dim)
    lc.set_energy(2.0, 0.0)
    x0 = np.zeros((1, dim), np.float32)
    lc.set_state(x0)
    lc.step(10, 0.01, 1.0, 1.0, 7, step0=0)
    want = ora.langevin_quadratic_f32(x0, 2.0, 0.0, 10, 0.01, 1.0, 1.0, 7, step0=0)
    np.testing.assert_allclose(lc.get_state(), want, rtol=0, atol=2e-4)
    lc.step(600, 0.01, 1.0, 1.0, 7, step0=10)
    x = lc.get_state()
    assert abs(x.var() - 0.505051) < 0.004 and abs(x.mean()) < 0.004
    lc.close()


def test_onsager_values_on_8192_squared_nibble_resident_lattice():
    """Physics at the roofline target's size on the kernel that holds it (512 x 512 nibble-plane tiles resident in LDS, strips
    exchanged between the 256 tiles): IsingModel2D(8192) at T = 2.0 < T_c from a cold start reproduces Onsager's infinite-lattice
    magnetisation (1 - sinh(2/T)^-4)^(1/8) = 0.911319 and energy u = -1.745565 per site."""
    from tsu.models.ising import IsingModel2D
    m = IsingModel2D(size=8192, temperature=2.0, seed=77, initial="up")
    m.equilibrate(n_sweeps=3000)
    ms, es = [], []
    for _ in range(20):
        m.gibbs_update(100)
        ms.append(m.magnetization())
        es.append(m.energy() / m.n_spins)
    assert abs(np.mean(ms) - 0.911319) < 0.0005, np.mean(ms)
    assert abs(np.mean(es) + 1.745565) < 0.0005, np.mean(es)
