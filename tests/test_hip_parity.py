"""Parity tests proper (-m gpu): the HIP path, called through the C-ABI (ctypes), against the oracle on the
same seeded inputs.  Bit-exact for spins / bits / integer observables; float tolerance stated per test."""
import numpy as np
import pytest

from oracle import oracle as ora

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from tsu import _hip
    _hip.Context.default()  # raises HipUnavailableError (test error, not skip) when the GPU path is missing
    return _hip


# ----------------------------------------------------------------------------- K5 Philox on the device
def test_philox_kat_on_device(hip):
    ctx = hip.Context.default()
    kats = [((0, 0, 0, 0), (0, 0), (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
            ((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2, (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
            ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0),
             (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1))]
    for ctr, key, want in kats:
        got = ctx.philox4x32_10([ctr], key)[0]
        assert tuple(int(v) for v in got) == want
    rng = np.random.default_rng(0)
    ctrs = rng.integers(0, 2 ** 32, size=(1000, 4), dtype=np.uint64).astype(np.uint32)
    key = np.array([123456789, 987654321], dtype=np.uint32)
    got = ctx.philox4x32_10(ctrs, key)
    want = np.array([ora.philox4x32_10(c, key) for c in ctrs])
    np.testing.assert_array_equal(got, want)


# ----------------------------------------------------------------------------- K1 lattice sweep
LATTICES = [(8, 8, True), (32, 32, True), (64, 64, True), (4, 4, True), (6, 20, True), (130, 36, True),
            (257, 130, False), (33, 47, False), (1, 9, False), (9, 1, False), (1, 1, False), (2, 2, False),
            (5, 16, False), (16, 17, False), (64, 1024, True), (12, 2050, True)]


@pytest.mark.parametrize("kernel", ["generic", "auto", "small"])
@pytest.mark.parametrize("rows,cols,periodic", LATTICES)
def test_ising2d_sweep_bit_exact(hip, rows, cols, periodic, kernel):
    seed = 1000 + rows * 31 + cols
    for (J, h, T, mode) in [(1.0, 0.0, 2.269185, hip.MODE_PHYSICAL), (-0.7, 0.3, 1.1, hip.MODE_COMPAT)]:
        table = ora.ising2d_thresholds(J, h, T, mode)
        lat = hip.Lattice(rows, cols, periodic)
        if kernel == "small" and rows * ((cols + 15) // 16) > 1024:
            with pytest.raises(hip.UnsupportedError):
                lat.set_kernel(hip.KERNEL_SMALL)
            return
        lat.set_kernel({"generic": hip.KERNEL_GENERIC, "auto": hip.KERNEL_AUTO, "small": hip.KERNEL_SMALL}[kernel])
        lat.randomize(seed)
        s0 = lat.get_spins()
        np.testing.assert_array_equal(s0, ora.ising2d_randomize(rows, cols, seed))
        lat.set_thresholds(table)
        lat.sweep(1, seed, sweep0=0)
        want = ora.ising2d_sweep(s0, periodic, table, 1, seed, sweep0=0)
        np.testing.assert_array_equal(lat.get_spins(), want)
        lat.sweep(7, seed, sweep0=1)
        want = ora.ising2d_sweep(want, periodic, table, 7, seed, sweep0=1)
        got = lat.get_spins()
        np.testing.assert_array_equal(got, want)
        assert lat.observables() == ora.ising2d_observables(want, periodic)
        lat.close()


TILED_LATTICES = [(96, 544), (128, 1024), (130, 560), (200, 2080), (256, 4096),
                  (96, 288), (128, 320), (256, 512), (1024, 1024),  # narrower lattices: 256-column tiles
                  (200, 128), (66, 160), (256, 144)]  # narrower than a tile: the tile is a window on the periodic extension


@pytest.mark.parametrize("periodic", [True, False])
@pytest.mark.parametrize("spl", [1, 2, 3, 8])
@pytest.mark.parametrize("rows,cols", TILED_LATTICES)
def test_ising2d_tiled_kernel_bit_exact(hip, rows, cols, spl, periodic):
    """LDS-tiled multi-sweep kernel == oracle == generic kernel, for any sweeps-per-launch; periodic lattices and
    open ones (the reference's default, ising.py:320-326: edge sites have degree 3, corners 2, own thresholds)."""
    seed = 555 + rows + cols
    s0 = ora.ising2d_randomize(rows, cols, seed)
    for (J, h, T, mode) in [(1.0, 0.0, 2.269185, hip.MODE_PHYSICAL), (-0.7, 0.3, 1.1, hip.MODE_COMPAT)]:
        table = ora.ising2d_thresholds(J, h, T, mode)
        lat = hip.Lattice(rows, cols, periodic)
        lat.set_kernel(hip.KERNEL_TILED, spl)
        lat.set_spins(s0)
        lat.set_thresholds(table)
        lat.sweep(1, seed, sweep0=0)
        want = ora.ising2d_sweep(s0, periodic, table, 1, seed, sweep0=0)
        np.testing.assert_array_equal(lat.get_spins(), want)
        lat.sweep(11, seed, sweep0=1)
        want = ora.ising2d_sweep(want, periodic, table, 11, seed, sweep0=1)
        np.testing.assert_array_equal(lat.get_spins(), want)
        assert lat.observables() == ora.ising2d_observables(want, periodic)
        lat.close()


@pytest.mark.parametrize("rows,cols,spl", [(100, 289, 3), (96, 290, 8), (130, 295, 5), (128, 300, 8), (97, 303, 2), (200, 1000, 8),
                                           (256, 1017, 8), (128, 1023, 4), (1000, 1000, 0), (512, 2040, 0)])
def test_ising2d_tiled_open_lattice_of_any_width(hip, rows, cols, spl):
    """Open lattices whose width is not a multiple of 16 (the reference's default boundary, any size: IsingGrid((1000, 1000))):
    the last 16-byte chunk of a row is ragged; the sites beyond the last column do not exist, the last column has degree 3,
    the row's pad bytes stay 0.  Tiled kernel == oracle, including the tie path (coarse table)."""
    seed = 31 + rows + cols
    s0 = ora.ising2d_randomize(rows, cols, seed)
    tables = [ora.ising2d_thresholds(1.0, 0.1, 2.0, hip.MODE_PHYSICAL),
              np.array([(k * 0x0A3D) << 16 | 0x8000 for k in range(25)], dtype=np.uint64)]
    for table in tables:
        lat = hip.Lattice(rows, cols, False)
        if spl:
            lat.set_kernel(hip.KERNEL_TILED, spl)
        lat.set_spins(s0)
        lat.set_thresholds(table)
        n0 = lat.launch_count()
        lat.sweep(19, seed, sweep0=2)
        assert lat.launch_count() - n0 <= -(-19 // (spl or 8))  # the tiled kernel, also when the library chooses (generic: 38 launches)
        want = ora.ising2d_sweep(s0, False, table, 19, seed, sweep0=2)
        np.testing.assert_array_equal(lat.get_spins(), want)
        assert lat.observables() == ora.ising2d_observables(want, False)
        # pad bytes untouched: a generic-kernel sweep afterwards (it reads whole chunks) still agrees
        lat.set_kernel(hip.KERNEL_GENERIC)
        lat.sweep(2, seed, sweep0=21)
        np.testing.assert_array_equal(lat.get_spins(), ora.ising2d_sweep(want, False, table, 2, seed, sweep0=21))
        lat.close()


@pytest.mark.parametrize("rows,cols,spl", [(130, 300, 3), (96, 290, 8), (128, 1000, 5), (256, 1000, 8), (200, 130, 8), (64, 258, 2),
                                           (1000, 1000, 0), (512, 2042, 0), (256, 1016, 0),
                                           (128, 520, 8), (96, 264, 8), (128, 1032, 8), (192, 530, 0)])  # 33 / 17 / 65 / 34 chunks: tiling origins 1 and 2
def test_ising2d_tiled_periodic_lattice_of_any_even_width(hip, rows, cols, spl):
    """Periodic lattices whose width is not a multiple of 16 (IsingModel2D(1000)): the wrap falls inside the last octet of a
    row; the workgroups whose window holds it take the one byte that crosses over from / to position v - 1 instead of 7.
    Tiled kernel (also tile-resident: 1016 columns = 64 chunks) == oracle, including the tie path (coarse table)."""
    seed = 57 + rows + cols
    s0 = ora.ising2d_randomize(rows, cols, seed)
    tables = [ora.ising2d_thresholds(1.0, 0.1, 2.0, hip.MODE_PHYSICAL),
              np.array([(k * 0x0A3D) << 16 | 0x8000 for k in range(25)], dtype=np.uint64)]
    for table in tables:
        lat = hip.Lattice(rows, cols, True)
        if spl:
            lat.set_kernel(hip.KERNEL_TILED, spl)
        lat.set_spins(s0)
        lat.set_thresholds(table)
        n0 = lat.launch_count()
        lat.sweep(19, seed, sweep0=2)
        assert lat.launch_count() - n0 <= -(-19 // (spl or 8))  # the tiled kernel, also when the library chooses (generic: 38 launches)
        want = ora.ising2d_sweep(s0, True, table, 19, seed, sweep0=2)
        np.testing.assert_array_equal(lat.get_spins(), want)
        assert lat.observables() == ora.ising2d_observables(want, True)
        lat.set_kernel(hip.KERNEL_GENERIC)  # pad bytes intact: the generic kernel reads whole chunks
        lat.sweep(2, seed, sweep0=21)
        np.testing.assert_array_equal(lat.get_spins(), ora.ising2d_sweep(want, True, table, 2, seed, sweep0=21))
        lat.close()


@pytest.mark.parametrize("rows,cols,periodic", [(32, 32, True), (4, 16, True), (128, 128, True), (64, 256, True), (30, 48, True), (4, 4, True), (8, 8, True), (20, 20, True), (50, 50, True), (126, 110, True),
                                               
                                                (2, 2, False), (3, 2, False), (50, 50, False), (96, 160, False), (127, 113, False),
                                                (64, 241, False), (33, 31, False)])
def test_ising2d_colour_plane_kernel_bit_exact(hip, rows, cols, periodic):
    """One-workgroup colour-plane kernel (whole lattices of at most 1024 octets: BASELINE configs[0] and the reference's
    own sizes) == oracle: physical and compat thresholds, a coarse table that drives the tie path, several calls."""
    seed = 77 + 3 * rows + cols
    s0 = ora.ising2d_randomize(rows, cols, seed)
    tables = [ora.ising2d_thresholds(1.0, 0.0, 2.269185, hip.MODE_PHYSICAL), ora.ising2d_thresholds(-0.7, 0.3, 1.1, hip.MODE_COMPAT),
              np.array([(k * 0x0A3D) << 16 | 0x8000 for k in range(25)], dtype=np.uint64),
              np.array([0, 1 << 32, 1, (1 << 32) - 1, 65536] * 5, dtype=np.uint64)]
    for table in tables:
        lat = hip.Lattice(rows, cols, periodic)
        lat.set_kernel(hip.KERNEL_SMALL)
        lat.set_spins(s0)
        lat.set_thresholds(table)
        n0 = lat.launch_count()
        lat.sweep(1, seed, sweep0=0)
        lat.sweep(30, seed, sweep0=1)
        assert lat.launch_count() - n0 == 2
        want = ora.ising2d_sweep(s0, periodic, table, 31, seed, sweep0=0)
        np.testing.assert_array_equal(lat.get_spins(), want)
        assert lat.observables() == ora.ising2d_observables(want, periodic)
        lat.set_kernel(hip.KERNEL_GENERIC)  # pad bytes intact: the generic kernel reads whole chunks
        lat.sweep(2, seed, sweep0=31)
        np.testing.assert_array_equal(lat.get_spins(), ora.ising2d_sweep(want, periodic, table, 2, seed, sweep0=31))
        lat.close()


def test_ising2d_tiled_ties_and_clamps(hip):
    """Coarse / extreme threshold tables drive the tiled kernel's tie path (low 16 bits) and the 0 / 2^32 clamps."""
    rows, cols = 128, 1040
    s0 = ora.ising2d_randomize(rows, cols, 5)
    for table in (np.array([(k * 0x0A3D) << 16 | 0x8000 for k in range(25)], dtype=np.uint64),
                  np.array([0, 1 << 32, 1, (1 << 32) - 1, 65536] * 5, dtype=np.uint64),
                  np.array([0xFFFF0000, 0x00010000, 0xFFFFFFFF, 0x0000FFFF, 0x80000000] * 5, dtype=np.uint64)):
        for periodic in (True, False):
            lat = hip.Lattice(rows, cols, periodic)
            lat.set_kernel(hip.KERNEL_TILED, 4)
            lat.set_spins(s0)
            lat.set_thresholds(table)
            lat.sweep(9, 99, sweep0=11)
            np.testing.assert_array_equal(lat.get_spins(), ora.ising2d_sweep(s0, periodic, table, 9, 99, sweep0=11))
            lat.close()


@pytest.mark.parametrize("nslab,ghost,k,spl", [(2, 8, 4, 0), (3, 16, 8, 0), (2, 2, 1, 0), (2, 32, 16, 4), (2, 40, 19, 8),
                                                (3, 24, 12, 5)])
@pytest.mark.parametrize("periodic,cols", [(True, 576), (False, 576), (True, 1024), (True, 1000), (False, 1001)])
def test_ising2d_tiled_slabs(hip, nslab, ghost, k, spl, periodic, cols):
    """Row slabs driven through the tiled kernel (ghost rows as the vertical halo) == whole lattice; with
    k > sweeps-per-launch a slab sweeps several launches per ghost refresh, extending into its ghost rows.
    Open lattices: the outer slabs' ghost rows lie beyond the lattice and count as empty.  1024 columns = whole tiles:
    periodic slabs then keep their tiles in LDS for the whole refresh period (tile-resident generations)."""
    per, seed = 128, 31
    rows = per * nslab
    table = ora.ising2d_thresholds(1.0, 0.1, 2.269185, 0)
    full = ora.ising2d_randomize(rows, cols, seed)
    slabs = [hip.Lattice(per, cols, periodic, total_rows=rows, row0=i * per, ghost=ghost) for i in range(nslab)]
    for i, s in enumerate(slabs):
        s.set_kernel(hip.KERNEL_TILED, spl)
        s.set_spins(full[i * per:(i + 1) * per])
        s.set_thresholds(table)
    want = full
    for it in range(3):
        owned = [s.get_spins() for s in slabs]
        for i, s in enumerate(slabs):
            if periodic or i > 0:
                s.set_spins(owned[(i - 1) % nslab][-ghost:], row_first=-ghost)
            if periodic or i < nslab - 1:
                s.set_spins(owned[(i + 1) % nslab][:ghost], row_first=per)
        for s in slabs:
            s.sweep(k, seed, sweep0=it * k)
        want = ora.ising2d_sweep(want, periodic, table, k, seed, sweep0=it * k)
        np.testing.assert_array_equal(np.concatenate([s.get_spins() for s in slabs]), want)


def test_ising2d_set_model_matches_oracle_thresholds(hip):
    for (J, h, T, mode) in [(1.0, 0.0, 2.269185, 0), (1.0, 0.0, 2.5, 1), (-0.7, 0.3, 1.1, 0), (0.4, -0.2, 0.05, 1),
                            (1.0, 0.0, 0.01, 0)]:
        np.testing.assert_array_equal(hip.ising2d_thresholds(J, h, T, mode), ora.ising2d_thresholds(J, h, T, mode))
    lat = hip.Lattice(16, 16, True)
    lat.randomize(3)
    s0 = lat.get_spins()
    lat.set_model(1.0, 0.1, 2.0, hip.MODE_PHYSICAL)
    lat.sweep(3, 3)
    np.testing.assert_array_equal(lat.get_spins(), ora.ising2d_sweep(s0, True, ora.ising2d_thresholds(1.0, 0.1, 2.0, 0), 3, 3))


def test_ising2d_ties_exercise_lazy_low_bits(hip):
    """Coarse thresholds make hi16 ties frequent; thresholds 0 / 2^32 / 1 / 2^32-1 hit the clamps."""
    rows, cols = 64, 96
    s0 = ora.ising2d_randomize(rows, cols, 5)
    for table in (np.array([(k * 0x0A3D) << 16 | 0x8000 for k in range(25)], dtype=np.uint64),
                  np.array([0, 1 << 32, 1, (1 << 32) - 1, 65536] * 5, dtype=np.uint64)):
        for periodic in (True, False):
            lat = hip.Lattice(rows, cols, periodic)
            lat.set_spins(s0)
            lat.set_thresholds(table)
            lat.sweep(6, 99, sweep0=11)
            np.testing.assert_array_equal(lat.get_spins(), ora.ising2d_sweep(s0, periodic, table, 6, 99, sweep0=11))


def test_ising2d_replica_and_fill(hip):
    lat = hip.Lattice(16, 32, True)
    lat.fill(-1)
    assert (lat.get_spins() == -1).all()
    lat.set_model(1.0, 0.0, 3.0)
    lat.sweep(2, 7, replica=3)
    want = ora.ising2d_sweep(-np.ones((16, 32), np.int8), True, ora.ising2d_thresholds(1.0, 0.0, 3.0, 0), 2, 7, replica=3)
    np.testing.assert_array_equal(lat.get_spins(), want)


@pytest.mark.parametrize("rows,cols,periodic", [(32, 32, True), (20, 20, False), (64, 256, True), (128, 1024, True)])
def test_ising2d_sweep_batch_equals_one_by_one(hip, rows, cols, periodic):
    """tsu_ising2d_sweep_batch / observables_batch: lattices at different temperatures, seeds, counters and replica
    ids advance together (one launch when they fit the one-workgroup kernel) == each swept alone == the oracle."""
    Ts = [1.5, 2.0, 2.269185, 3.0, 4.5]
    lats, want = [], []
    for i, T in enumerate(Ts):
        lat = hip.Lattice(rows, cols, periodic)
        lat.randomize(100 + i, replica=i)
        lat.set_model(1.0, 0.05 * i, T)
        lats.append(lat)
        table = ora.ising2d_thresholds(1.0, 0.05 * i, T, 0)
        s0 = ora.ising2d_randomize(rows, cols, 100 + i, replica=i)
        want.append(ora.ising2d_sweep(s0, periodic, table, 9, 100 + i, sweep0=3 * i, replica=i))
    hip.sweep_batch(lats, 9, [100 + i for i in range(len(Ts))], [3 * i for i in range(len(Ts))], list(range(len(Ts))))
    obs = hip.observables_batch(lats)
    for i, lat in enumerate(lats):
        np.testing.assert_array_equal(lat.get_spins(), want[i])
        assert obs[i] == ora.ising2d_observables(want[i], periodic) == lat.observables()
        lat.close()


@pytest.mark.parametrize("rows,cols,periodic", [(32, 32, False), (130, 36, True), (128, 1024, True)])
def test_ising2d_sample_run_matches_oracle(hip, rows, cols, periodic):
    """tsu_ising2d_sample: burn-in + n_samples x n_sweeps with the states gathered on the device == the oracle's chain."""
    seed, burn, ns, m = 4711, 5, 3, 4
    table = ora.ising2d_thresholds(1.0, 0.1, 2.4, 0)
    lat = hip.Lattice(rows, cols, periodic)
    lat.randomize(seed)
    lat.set_thresholds(table)
    got = lat.sample(burn, ns, m, seed, sweep0=2)
    cur = ora.ising2d_sweep(ora.ising2d_randomize(rows, cols, seed), periodic, table, burn, seed, sweep0=2)
    for k in range(m):
        cur = ora.ising2d_sweep(cur, periodic, table, ns, seed, sweep0=2 + burn + k * ns)
        np.testing.assert_array_equal(got[k], cur)
    np.testing.assert_array_equal(lat.get_spins(), cur)
    lat.close()


def test_ising2d_argument_errors(hip):
    with pytest.raises(hip.UnsupportedError):
        hip.Lattice(5, 8, True)  # odd periodic dimension has no 2-colouring
    with pytest.raises(ValueError):
        hip.Lattice(0, 8, False)
    lat = hip.Lattice(8, 8, False)
    with pytest.raises(ValueError):
        lat.sweep(1, 0)  # no thresholds yet
    with pytest.raises(ValueError, match="Temperature must be positive"):
        lat.set_model(1.0, 0.0, -1.0)


@pytest.mark.parametrize("periodic", [True, False])
@pytest.mark.parametrize("nslab,ghost,k", [(2, 2, 1), (4, 4, 2), (3, 8, 4)])
def test_ising2d_slab_decomposition_invariance(hip, periodic, nslab, ghost, k):
    """P row slabs with host-mediated ghost refresh every k sweeps == the whole lattice (globally keyed RNG)."""
    rows, cols, seed = 24 * nslab, 40, 77
    table = ora.ising2d_thresholds(1.0, 0.05, 2.3, 0)
    full = ora.ising2d_randomize(rows, cols, seed)
    per = rows // nslab
    slabs = [hip.Lattice(per, cols, periodic, total_rows=rows, row0=i * per, ghost=ghost) for i in range(nslab)]
    for i, s in enumerate(slabs):
        s.randomize(seed)
        np.testing.assert_array_equal(s.get_spins(), full[i * per:(i + 1) * per])
        s.set_thresholds(table)
    want = full
    for it in range(3):
        # ghost refresh: rows adjacent to each slab, taken from the neighbours' owned rows
        owned = [s.get_spins() for s in slabs]
        for i, s in enumerate(slabs):
            up, dn = (i - 1) % nslab, (i + 1) % nslab
            if periodic or i > 0:
                s.set_spins(owned[up][-ghost:], row_first=-ghost)
            if periodic or i < nslab - 1:
                s.set_spins(owned[dn][:ghost], row_first=per)
        for s in slabs:
            s.sweep(k, seed, sweep0=it * k)
        want = ora.ising2d_sweep(want, periodic, table, k, seed, sweep0=it * k)
        got = np.concatenate([s.get_spins() for s in slabs])
        np.testing.assert_array_equal(got, want)
    # observables: per-slab sums add up to the whole-lattice values (ghost rows fresh)
    owned = [s.get_spins() for s in slabs]
    for i, s in enumerate(slabs):
        if periodic or i < nslab - 1:
            s.set_spins(owned[(i + 1) % nslab][:ghost], row_first=per)
    tot = np.sum([s.observables() for s in slabs], axis=0)
    assert tuple(int(v) for v in tot) == ora.ising2d_observables(want, periodic)


# ----------------------------------------------------------------------------- K2 dense sweep
@pytest.mark.parametrize("n", [1, 12, 64, 65, 100, 128, 129, 191, 192, 193, 200, 333, 448, 449, 512, 576, 577])
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_dense_sweep_matches_oracle(hip, n, dtype):
    rng = np.random.default_rng(n)
    J = rng.normal(size=(n, n)) / max(1.0, np.sqrt(n))
    if n % 2 == 0:
        J = (J + J.T) / 2  # symmetric for even n, asymmetric for odd n (exercises the J^T path)
    if dtype == "f32":
        J = J.astype(np.float32).astype(np.float64)
    b = rng.normal(size=n) * 0.3
    st = rng.integers(0, 2, size=n).astype(np.int8)
    T, seed = 0.9, 4242
    d = hip.DenseSystem(J, b, hip.DTYPE_F64 if dtype == "f64" else hip.DTYPE_F32)
    # Philox uniforms, natural order
    d.set_state(st)
    d.sweep(T, 5, seed=seed, sweep0=2)
    want = ora.dense_sweep_philox(st, J, b, T, 5, seed, sweep0=2)
    np.testing.assert_array_equal(d.get_state(), want)
    assert abs(d.energy() - ora.c_dense_energy(want, J, b)) < 1e-9 * max(1.0, n)
    # replayed uniforms + permuted order (the reference's update_order="random")
    order = np.array([rng.permutation(n) for _ in range(4)])
    u = rng.random(size=(4, n))
    d.set_state(st)
    d.sweep(T, 4, order=order, replay_uniforms=u)
    want = ora.c_dense_sweep_replay(st, J, b, T, u, order)
    np.testing.assert_array_equal(d.get_state().astype(np.int64), want)
    # replayed uniforms, natural order (one wave up to 192 / 128 sites)
    d.set_state(st)
    d.sweep(T, 4, replay_uniforms=u)
    np.testing.assert_array_equal(d.get_state().astype(np.int64), ora.c_dense_sweep_replay(st, J, b, T, u, None))
    d.close()


@pytest.mark.parametrize("n,dtype,T,sym", [(4500, "f32", 1.0, True), (2051, "f64", 0.3, False), (6144, "f32", 0.2, True)])
def test_dense_superblocks_match_oracle(hip, n, dtype, T, sym):
    """Several superblocks of 2048 positions (fixed-point iteration, change lists, strip updates, grid barriers across
    the XCDs): bit-exact against the oracle's sequential sweep; vector and scalar load paths, symmetric and not."""
    rng = np.random.default_rng(n)
    J = rng.standard_normal((n, n)) / np.sqrt(n)
    if sym:
        J = (J + J.T) / 2
    if dtype == "f32":
        J = J.astype(np.float32).astype(np.float64)
    b = rng.normal(size=n) * 0.2
    st = rng.integers(0, 2, size=n).astype(np.int8)
    d = hip.DenseSystem(J, b, hip.DTYPE_F64 if dtype == "f64" else hip.DTYPE_F32)
    d.set_state(st)
    d.sweep(T, 3, seed=77, sweep0=5)
    want = ora.dense_sweep_philox(st, J, b, T, 3, 77, sweep0=5)
    np.testing.assert_array_equal(d.get_state(), want)
    # a second call continues from the device state (odd/even buffer swap inside the single launch)
    d.sweep(T, 2, seed=77, sweep0=8)
    want = ora.dense_sweep_philox(want, J, b, T, 2, 77, sweep0=8)
    np.testing.assert_array_equal(d.get_state(), want)
    if n == 2051:  # many sweeps in one call: fields handed from sweep to sweep, recomputed every 64th
        d.set_state(st)
        d.sweep(T, 70, seed=78, sweep0=0)
        np.testing.assert_array_equal(d.get_state(), ora.dense_sweep_philox(st, J, b, T, 70, 78, sweep0=0))
    # replayed uniforms take the same path
    u = rng.random(size=(2, n))
    d.set_state(st)
    d.sweep(T, 2, replay_uniforms=u)
    np.testing.assert_array_equal(d.get_state().astype(np.int64), ora.c_dense_sweep_replay(st, J, b, T, u, None))
    d.close()


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("n", [1, 10, 64, 65, 100, 128, 150, 192, 200])
def test_dense_sample_run_matches_oracle(hip, n, dtype):
    """tsu_dense_sample (burn-in + n_samples x n_sweeps in one call; a single wave up to 192 fp32 / 128 fp64 sites) against
    the oracle's sweep-by-sweep chain: Philox uniforms, replayed uniforms in natural and (n > 64) permuted order."""
    rng = np.random.default_rng(100 + n)
    J = rng.normal(size=(n, n)) / max(1.0, np.sqrt(n))
    J = (J + J.T) / 2
    if dtype == "f32":
        J = J.astype(np.float32).astype(np.float64)
    b = rng.normal(size=n) * 0.4
    st = rng.integers(0, 2, size=n).astype(np.int8)
    T, burn, ns, m, seed = 0.8, 7, 3, 5, 31337
    d = hip.DenseSystem(J, b, hip.DTYPE_F64 if dtype == "f64" else hip.DTYPE_F32)
    d.set_state(st)
    got = d.sample(T, burn, ns, m, seed=seed, sweep0=11)
    cur = ora.dense_sweep_philox(st, J, b, T, burn, seed, sweep0=11)
    for k in range(m):
        cur = ora.dense_sweep_philox(cur, J, b, T, ns, seed, sweep0=11 + burn + k * ns)
        np.testing.assert_array_equal(got[k], cur)
    np.testing.assert_array_equal(d.get_state(), cur)  # the run leaves the last state resident
    total = burn + m * ns
    u = rng.random(size=(total, n))
    for order in ([None, np.array([rng.permutation(n) for _ in range(total)])] if n > 64 else [None]):
        d.set_state(st)
        got = d.sample(T, burn, ns, m, order=order, replay_uniforms=u)
        cur = ora.c_dense_sweep_replay(st, J, b, T, u[:burn], None if order is None else order[:burn])
        for k in range(m):
            lo = burn + k * ns
            cur = ora.c_dense_sweep_replay(cur, J, b, T, u[lo:lo + ns], None if order is None else order[lo:lo + ns])
            np.testing.assert_array_equal(got[k].astype(np.int64), cur)
    d.close()


@pytest.mark.parametrize("n,dtype", [(15, "f64"), (100, "f32"), (150, "f64"), (300, "f32"), (700, "f32")])
def test_dense_anneal_schedule_matches_oracle(hip, n, dtype):
    """tsu_dense_anneal (one sweep per temperature of a schedule, every state recorded, one call): one wave, one workgroup
    or one tsu_dense_sweep per step depending on the size -- the same chain as the oracle swept step by step."""
    rng = np.random.default_rng(7 + n)
    J = rng.normal(size=(n, n)) / np.sqrt(n)
    J = (J + J.T) / 2
    if dtype == "f32":
        J = J.astype(np.float32).astype(np.float64)
    st = rng.integers(0, 2, size=n).astype(np.int8)
    temps = 5.0 * (0.02 / 5.0) ** (np.arange(70) / 70.0)  # the reference's exponential schedule (gibbs.py:375-377)
    d = hip.DenseSystem(J, None, hip.DTYPE_F64 if dtype == "f64" else hip.DTYPE_F32)
    d.set_state(st)
    got = d.anneal(temps, seed=99, sweep0=3)
    cur = st
    for k, T in enumerate(temps):
        cur = ora.dense_sweep_philox(cur, J, None, float(T), 1, 99, sweep0=3 + k)
        np.testing.assert_array_equal(got[k], cur)
    np.testing.assert_array_equal(d.get_state(), cur)
    d.close()


@pytest.mark.parametrize("n", [10, 64, 100, 128, 130, 300, 500])
def test_dense_sweep_replicas_match_oracle(hip, n):
    """tsu_dense_sweep_replicas (a tempering ladder's replica loop in one call; one wave per replica for n <= 64):
    each replica == the oracle's chain at its temperature / seed / counter; replayed uniforms likewise."""
    rng = np.random.default_rng(n)
    J = rng.normal(size=(n, n)) / np.sqrt(n)
    J = (J + J.T) / 2
    b = rng.normal(size=n) * 0.2
    R, ns = 4, 6
    Ts = [0.5, 1.0, 2.0, 4.0]
    st = rng.integers(0, 2, size=(R, n)).astype(np.int8)
    seeds, s0s, reps = [11, 22, 33, 44], [0, 5, 9, 100], [0, 1, 2, 3]
    d = hip.DenseSystem(J, b)
    got = d.sweep_replicas(st, Ts, ns, seeds, s0s, reps)
    for r in range(R):
        np.testing.assert_array_equal(got[r], ora.dense_sweep_philox(st[r], J, b, Ts[r], ns, seeds[r], sweep0=s0s[r], replica=reps[r]))
    u = rng.random(size=(R, ns, n))
    got = d.sweep_replicas(st, Ts, ns, [0] * R, [0] * R, replay_uniforms=u)
    for r in range(R):
        np.testing.assert_array_equal(got[r].astype(np.int64), ora.c_dense_sweep_replay(st[r], J, b, Ts[r], u[r], None))
    d.close()


def test_dense_long_run_stays_on_the_oracle_chain(hip):
    """Soak for the cooperative kernel's barrier / agent-scope data path: 200 sweeps of a 5000-site glass (two
    superblocks, ~4000 grid barriers, fields handed on between sweeps) in ONE call stay bit-identical to the oracle."""
    n, T = 5000, 0.7
    rng = np.random.default_rng(5)
    J = rng.standard_normal((n, n)).astype(np.float32).astype(np.float64) / np.sqrt(n)
    J = (J + J.T) / 2
    st = rng.integers(0, 2, size=n).astype(np.int8)
    d = hip.DenseSystem(J, None, hip.DTYPE_F64)
    d.set_state(st)
    d.sweep(T, 200, seed=3, sweep0=0)
    np.testing.assert_array_equal(d.get_state(), ora.dense_sweep_philox(st, J, None, T, 200, 3, sweep0=0))
    d.close()


def test_dense_golden_replay_on_device(hip, golden):
    """The reference's own seeded run (tests/golden/g1, g2) reproduced by the HIP kernel with replayed MT19937 draws."""
    for name in ("g1_dense_sequential", "g2_dense_random"):
        g = golden(name)
        order = g["perms"] if name.startswith("g2") else None
        d = hip.DenseSystem(g["J"], g["bias"])
        d.set_state(g["init"].astype(np.int8))
        T, burnin, ns = float(g["T"]), int(g["burnin"]), int(g["n_sweeps"])
        d.sweep(T, burnin, order=None if order is None else order[:burnin], replay_uniforms=g["uniforms"][:burnin])
        pos = burnin
        for k in range(int(g["n_samples"])):
            d.sweep(T, ns, order=None if order is None else order[pos:pos + ns], replay_uniforms=g["uniforms"][pos:pos + ns])
            pos += ns
            np.testing.assert_array_equal(d.get_state(), g["samples"][k])


# ----------------------------------------------------------------------------- K3 Langevin
LANGEVIN_ATOL = 2e-4  # fp32; device log2/sin/cos are the native approximations (v_log_f32, v_sin_f32, v_cos_f32)


@pytest.mark.parametrize("n_chains,dim", [(1, 8), (3, 64), (2, 1027), (1, 5)])
def test_langevin_matches_oracle(hip, n_chains, dim):
    rng = np.random.default_rng(dim)
    x = rng.normal(size=(n_chains, dim)).astype(np.float32)
    k = rng.uniform(0.5, 2.0, size=dim).astype(np.float32)
    mu = rng.normal(size=dim).astype(np.float32)
    T, dt, gamma, seed = 0.7, 0.02, 1.5, 99
    lc = hip.LangevinChains(n_chains, dim)
    lc.set_energy(k, mu)
    lc.set_state(x)
    traj = lc.step(20, dt, gamma, T, seed, step0=3, chain0=5, trajectory=True)
    want, wtraj = ora.langevin_quadratic_f32(x, k, mu, 20, dt, gamma, T, seed, step0=3, chain0=5, trajectory=True)
    np.testing.assert_allclose(traj, wtraj, rtol=0, atol=LANGEVIN_ATOL)
    np.testing.assert_allclose(lc.get_state(), want, rtol=0, atol=LANGEVIN_ATOL)
    # one launch per step == fused steps (same counters)
    lc.set_state(x)
    lc.set_kernel(steps_per_launch=1)
    lc.step(20, dt, gamma, T, seed, step0=3, chain0=5)
    np.testing.assert_array_equal(lc.get_state(), traj[-1])
    # restart: x_init + amp * N(0,1)
    lc.restart(mu, 0.1, seed, chain0=2)
    wantr = np.array([[mu[i] + np.float32(0.1) * ora.langevin_normals_f32(i >> 2, 2 + c, 0, seed, ora.TAG_LANGEVIN_RESTART)[i & 3]
                       for i in range(dim)] for c in range(n_chains)], dtype=np.float32)
    np.testing.assert_allclose(lc.get_state(), wantr, rtol=0, atol=LANGEVIN_ATOL)


def test_langevin_uniform_energy_is_the_general_kernel_with_scalars(hip):
    """One stiffness and one centre for every element (`QuadraticEnergy(2.0)`, the README's energies) runs the kernel variant that
    takes them as scalars -- no loads beside the state's; the same arithmetic: bit for bit what the general variant makes of the same
    numbers (an array whose LAST element differs selects the general variant; all other elements must agree exactly), and the oracle's
    trajectory within the fp32 tolerance."""
    n_chains, dim = 3, 1030
    x = np.random.default_rng(1).normal(size=(n_chains, dim)).astype(np.float32)
    a = hip.LangevinChains(n_chains, dim)
    a.set_energy(1.75, 0.25)
    a.set_state(x)
    a.step(15, 0.02, 1.5, 0.7, 99, step0=2, chain0=4)
    k = np.full(dim, 1.75, np.float32)
    k[-1] = 1.5
    b = hip.LangevinChains(n_chains, dim)
    b.set_energy(k, np.full(dim, 0.25, np.float32))
    b.set_state(x)
    b.step(15, 0.02, 1.5, 0.7, 99, step0=2, chain0=4)
    np.testing.assert_array_equal(a.get_state()[:, :-1], b.get_state()[:, :-1])
    want = ora.langevin_quadratic_f32(x, 1.75, 0.25, 15, 0.02, 1.5, 0.7, 99, step0=2, chain0=4)
    np.testing.assert_allclose(a.get_state(), want, rtol=0, atol=LANGEVIN_ATOL)


def test_langevin_stationary_variance(hip):
    """dim 2^16 chains-as-elements: var -> T / (k (1 - k dt / (2 gamma))) = 0.505051 for k=2, dt=0.01."""
    lc = hip.LangevinChains(1, 1 << 16)
    lc.set_energy(2.0, 0.0)
    lc.set_state(np.zeros((1, 1 << 16), np.float32))
    lc.step(1500, 0.01, 1.0, 1.0, 7)
    x = lc.get_state()
    assert abs(x.var() - 0.505051) < 0.01 and abs(x.mean()) < 0.02
