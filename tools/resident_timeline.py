"""Per-tile timeline of the tile-resident lattice kernel (TSU_K1_VERBOSE=3 prints it from the library): where a generation's
time goes on every tile, and whether one XCD runs behind the others.  usage: python tools/resident_timeline.py [L ...]   (OPEN=1: open boundaries)"""
import os
import sys

os.environ.setdefault("TSU_K1_VERBOSE", "3")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tsu-emulator_amd"))
from tsu import _hip  # noqa: E402

ctx = _hip.Context(0)
for L in [int(a) for a in sys.argv[1:]] or [4096, 8192]:
    lat = _hip.Lattice(L, L, os.environ.get("OPEN", "0") != "1", ctx=ctx)
    lat.randomize(1)
    lat.set_thresholds(_hip.ising2d_thresholds(1.0, 0.0, 2.269185))
    for rep in range(3):
        print(f"--- L={L} rep {rep}", file=sys.stderr, flush=True)
        lat.sweep(256, 7, 256 * rep)
        ctx.synchronize()
    lat.close()
