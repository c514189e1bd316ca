"""One-off randomized cross-check of the tiled / tile-resident kernels against the generic kernel (development aid)."""
import os, sys, zlib, random
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tsu-emulator_amd"))
from tsu import _hip
random.seed(int(os.environ.get("FUZZ_SEED", "1")))
n_ok = 0
for case in range(int(os.environ.get("FUZZ_CASES", "60"))):
    rows = random.choice([96, 128, 192, 256, 320, 512, 1024, 2048, 3072])
    cols = random.choice([288, 320, 512, 544, 768, 1024, 1536, 2048, 4096])
    periodic = random.random() < 0.6
    if os.environ.get("FUZZ_NARROW"):  # lattices narrower than a tile (a tile is a window on the periodic extension), few rows
        rows = random.choice([64, 66, 96, 100, 128, 200, 256, 500, 1000, 2048])
        cols = random.choice([128, 144, 160, 176, 192, 208, 224, 240, 256, 272, 288, 304])
    if not periodic and random.random() < 0.6:  # open lattices: any width / height
        cols = random.randint(130, 400) if os.environ.get("FUZZ_NARROW") else random.randint(288, 2100)
        rows = random.randint(64, 1100)
    if periodic and os.environ.get("FUZZ_RAGGED"):  # periodic lattices of any even width / height: the wrap falls inside an octet
        cols = 2 * random.randint(65, 1100)
        rows = 2 * random.randint(32, 600)
    if os.environ.get("FUZZ_FLEX"):  # periodic, width a multiple of 16, any even height: with TSU_K1_FLEX_MAX_TILES=6..20 in the
        periodic = True              # environment these take the flexible tile-resident cut (unequal tile rows, partial last column)
        cols = 16 * random.randint(17, 140)
        rows = 2 * random.randint(40, 700)
    if periodic and (rows % 2 or cols % 2):
        continue
    k = random.choice([0, 1, 3, 5, 8])
    calls = [random.choice([1, 2, 7, 9, 17, 40, 65]) for _ in range(random.choice([1, 2, 3]))]
    J, h, T = random.choice([1.0, -0.8]), random.choice([0.0, 0.15]), random.choice([1.7, 2.269185, 3.1])
    seed = random.getrandbits(40)
    res = []
    for kern in (_hip.KERNEL_AUTO, _hip.KERNEL_GENERIC):
        lat = _hip.Lattice(rows, cols, periodic)
        lat.set_kernel(kern, k if kern == _hip.KERNEL_AUTO else 0)
        lat.randomize(seed)
        lat.set_model(J, h, T)
        s0 = 0
        for n in calls:
            lat.sweep(n, seed, s0)
            s0 += n
        res.append((zlib.crc32(lat.get_spins().tobytes()), lat.observables()))
        lat.close()
    ok = res[0] == res[1]
    n_ok += ok
    print(("ok  " if ok else "FAIL"), rows, cols, "periodic" if periodic else "open", "k", k, "calls", calls, flush=True)
    if not ok:
        sys.exit(1)
print("all", n_ok, "cases agree")
