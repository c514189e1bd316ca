"""One-off randomized cross-check of the one-workgroup colour-plane kernel against the generic kernel (development aid)."""
import os, sys, zlib, random
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tsu-emulator_amd"))
import numpy as np
from tsu import _hip
random.seed(int(os.environ.get("FUZZ_SEED", "1")))
n_ok = 0
for case in range(int(os.environ.get("FUZZ_CASES", "300"))):
    periodic = random.random() < 0.5
    rows = random.randint(2, 128)
    cols = random.randint(2, 512)
    if periodic:
        rows += rows & 1
        cols += cols & 1
        rows, cols = max(rows, 4), max(cols, 4)
    if rows * ((cols + 15) // 16) > 1024:
        continue
    calls = [random.choice([1, 2, 7, 33]) for _ in range(random.choice([1, 2, 3]))]
    seed = random.getrandbits(40)
    kind = random.random()
    if kind < 0.25:
        table = np.array([(random.getrandbits(6) * 0x0400) << 16 | random.getrandbits(16) for _ in range(25)], dtype=np.uint64)  # ties
    res = []
    for kern in (_hip.KERNEL_SMALL, _hip.KERNEL_GENERIC):
        lat = _hip.Lattice(rows, cols, periodic)
        lat.set_kernel(kern)
        lat.randomize(seed)
        if kind < 0.25:
            lat.set_thresholds(table)
        else:
            lat.set_model(random.Random(seed).choice([1.0, -0.8]), random.Random(seed + 1).choice([0.0, 0.15]), random.Random(seed + 2).choice([1.7, 2.269185, 3.1]))
        s0 = 5
        for n in calls:
            lat.sweep(n, seed, s0)
            s0 += n
        res.append((zlib.crc32(lat.get_spins().tobytes()), lat.observables()))
        lat.close()
    ok = res[0] == res[1]
    n_ok += ok
    print(("ok  " if ok else "FAIL"), rows, cols, "periodic" if periodic else "open", "ties" if kind < 0.25 else "model", "calls", calls, flush=True)
    if not ok:
        sys.exit(1)
print("all", n_ok, "cases agree")
