"""K1 with nibble colour planes (4 bits per site in LDS): bit-exact against the oracle, resident and launch-per-generation
forms, forced on lattices small enough for the oracle (the automatic choice takes these shapes only from 8192^2 up, which
tests/test_full_size_gpu.py covers through size-independent properties)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(variant, cases):
    env = dict(os.environ, TSU_TILE_VARIANT=str(variant))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "nibble_check.py"), json.dumps(cases)], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ALL OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    return r.stdout


def test_nibble_512_row_tiles_resident_and_not():
    # variant 8: 512 x 512 tiles, 1024 threads.  rows % 512 == 0 and cols % 512 == 0 -> tile-resident for calls of > k sweeps
    out = _run(8, [[1024, 1024, [3, 8, 21, 40], 2.269185, 8],     # 2 x 2 tiles; 21 = 2 generations + a short one
                   [1536, 512, [9, 17], 2.0, 4],                   # 3 x 1 tiles, k = 4
                   [1000, 1024, [5, 20], 2.5, 8],                  # ragged tile rows: one launch per generation
                   [1024, 768, [12], 3.0, 8]])                     # ragged tile columns
    assert "ok 1024x1024" in out


def test_nibble_256_row_tiles_two_workgroups_per_cu():
    # variant 9: 256 x 512 tiles, 512 threads, one launch per generation (what lattices beyond 8192^2 run on)
    _run(9, [[512, 512, [3, 8, 13], 2.269185, 8], [768, 1024, [20], 2.0, 8], [300, 512, [7], 2.5, 5], [1280, 2048, [16], 2.269185, 8]])


def test_nibble_planes_on_open_lattices_of_any_width():
    # open boundaries: degree-3 / degree-2 thresholds at the edges, nothing beyond them; widths that are not a multiple of 16
    _run(9, [[512, 512, [3, 8, 13], 2.269185, 8, 0], [768, 1000, [20], 2.0, 8, 0], [301, 530, [7, 9], 2.5, 5, 0], [1280, 2047, [16], 2.269185, 8, 0]])
    out = _run(8, [[1024, 1024, [3, 21, 40], 2.269185, 8, 0],   # 2 x 2 tiles of 512 x 512: tile-resident
                   [1536, 512, [9, 17], 2.0, 4, 0],
                   [1000, 1111, [5, 20], 2.5, 8, 0]])
    assert "ok 1024x1024" in out


def test_nibble_planes_on_periodic_lattices_of_ragged_width():
    # widths that are not a multiple of 16: the wrap falls inside the last octet (SEAM form of the nibble loop); 1000 = 62 octets + 8
    # columns (4 sites per colour in the last octet), 1110 = 69 + 6 (3 sites), 530 = 33 + 2 (1 site), 2046 = 127 + 14 (7 sites)
    _run(9, [[512, 1000, [3, 8, 13], 2.269185, 8], [768, 1110, [20], 2.0, 8], [300, 530, [7, 9], 2.5, 5], [1280, 2046, [16], 2.269185, 8],
             [600, 1004, [11], 2.269185, 8], [600, 1012, [11], 3.0, 8]])
    _run(8, [[1024, 1000, [3, 21, 40], 2.269185, 8], [1000, 1110, [5, 20], 2.5, 8], [1536, 530, [9, 17], 2.0, 4]])
