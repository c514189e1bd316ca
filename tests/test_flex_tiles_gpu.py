"""K1 tile-resident runs of lattices that do not divide into whole tiles (flexible cut: balanced tile rows of different heights,
a last tile column with fewer octets): bit-exact against the oracle on lattices small enough for it, with the number of tiles
capped (TSU_K1_FLEX_MAX_TILES) so that a few tall tiles cover them, and at full size against the generic kernel."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(variant, max_tiles, cases):
    env = dict(os.environ, TSU_TILE_VARIANT=str(variant), TSU_K1_FLEX_MAX_TILES=str(max_tiles))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "nibble_check.py"), json.dumps(cases)], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ALL OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    return {(int(m.group(1)), int(m.group(2))): int(m.group(3)) for m in re.finditer(r"ok (\d+)x(\d+) .* launches=(\d+)", r.stdout)}


def test_nibble_tiles_with_unequal_rows_and_a_partial_last_column():
    # 512-column nibble tiles.  1120 columns = 70 octets = 32 + 32 + 6; 1040 = 65 octets: the last column holds ONE octet;
    # 1504 = 94 octets: 30.  614 rows over 2 tile rows = 306 + 308, 1000 over 2 = 500 + 500, 3 tile rows of 604 = 200 + 202 + 202.
    launches = _run(8, 6, [[1000, 1120, [3, 21, 40], 2.269185, 8],
                           [604, 1040, [17, 9, 30], 2.0, 8],
                           [614, 1504, [24, 25], 2.5, 8],
                           [604, 512, [40], 3.0, 8],          # one tile column (its own left and right neighbour), 6 tile rows
                           [600, 1120, [12, 13], 2.269185, 4]])
    # a call of more than k sweeps is ONE launch when the tiles stay resident: 3 sweeps = 1 launch, 21 = 1, 40 = 1
    assert launches[(1000, 1120)] == 3 and launches[(614, 1504)] == 2 and launches[(604, 512)] == 1


def test_byte_tiles_with_unequal_rows_and_a_partial_last_column():
    # 512-column byte tiles: 500 rows over 3 tile rows = 166 + 166 + 168
    launches = _run(3, 9, [[500, 1120, [20, 21], 2.269185, 8], [332, 1040, [33], 2.0, 8]])
    assert launches[(500, 1120)] == 2
    # 256-column byte tiles: 400 columns = 25 octets = 16 + 9
    launches = _run(4, 4, [[300, 400, [18, 40], 2.5, 8], [302, 272, [25], 2.269185, 8]])
    assert launches[(300, 400)] == 2


def test_ragged_widths_take_the_flexible_cut_on_byte_tiles():
    # widths that are not a multiple of 16: the wrap falls inside the last octet (SEAM form), the tiling starts at a shifted octet;
    # 1110 columns = 69 octets + 6 columns, 1000 = 62 octets + 8 columns, 1318 = 82 octets + 6 columns
    launches = _run(3, 9, [[500, 1110, [20, 21], 2.269185, 8], [302, 1000, [40, 9], 2.0, 8], [418, 1318, [17], 2.5, 8]])
    assert launches[(500, 1110)] == 2 and launches[(302, 1000)] == 2
    launches = _run(4, 6, [[300, 410, [18, 40], 2.5, 8], [420, 1000, [25, 16], 2.269185, 8]])
    assert launches[(300, 410)] == 2
    # the same on nibble planes (SEAM form of the nibble loop)
    launches = _run(8, 6, [[1000, 1110, [20, 21], 2.269185, 8], [604, 1000, [33, 9], 2.0, 8], [614, 1318, [24], 2.5, 8]])
    assert launches[(1000, 1110)] == 2


def test_open_lattices_of_any_shape_take_the_flexible_cut():
    # open boundaries (the reference's IsingGrid default): odd heights and widths too -- 501 rows over 3 tile rows = 168 + 168 + 165
    launches = _run(3, 9, [[501, 1111, [20, 21], 2.269185, 8, 0], [333, 1000, [40, 9], 2.0, 8, 0], [410, 530, [17], 2.5, 8, 0]])
    assert launches[(501, 1111)] == 2 and launches[(333, 1000)] == 2
    launches = _run(4, 6, [[299, 401, [18, 40], 2.5, 8, 0], [420, 1000, [25, 16], 2.269185, 4, 0]])
    assert launches[(299, 401)] == 2
    # nibble planes, open, flexible cut: 1001 rows over 2 tile rows, 1111 columns = 69 octets + 7 columns
    launches = _run(8, 6, [[1001, 1111, [20, 21], 2.269185, 8, 0], [603, 1040, [33, 9], 2.0, 8, 0]])
    assert launches[(1001, 1111)] == 2


@pytest.mark.parametrize("L", [4000, 5000, 6000, 7000])
@pytest.mark.parametrize("periodic", [True, False])
def test_full_size_lattices_that_do_not_divide_into_tiles_stay_resident_and_equal_the_generic_kernel(L, periodic):
    from tsu import _hip
    ctx = _hip.Context.default()
    table = _hip.ising2d_thresholds(1.0, 0.0, 2.269185)
    if not periodic:
        L += 1  # odd: 4001, 5001, 6001 (the last one in nibble planes: beyond what 256 CUs hold in byte planes)
    a = _hip.Lattice(L, L, periodic, ctx=ctx)
    b = _hip.Lattice(L, L, periodic, ctx=ctx)
    b.set_kernel(_hip.KERNEL_GENERIC, 0)
    for lat in (a, b):
        lat.randomize(7)
        lat.set_thresholds(table)
    n0 = a.launch_count()
    for n, s0 in ((40, 0), (19, 40)):
        a.sweep(n, 11, s0)
        b.sweep(n, 11, s0)
    assert a.launch_count() - n0 == 2, "tile-resident: one launch per call"
    assert a.observables() == b.observables()
    assert (a.get_spins() == b.get_spins()).all()
    a.close()
    b.close()
