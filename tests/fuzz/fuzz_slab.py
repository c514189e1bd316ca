"""Randomized check: a slab that is its own neighbour (world size 1, device halo copies) == the plain periodic lattice."""
import os, sys, zlib, random
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tsu-emulator_amd"))
import torch
from tsu import _hip
from tsu.distributed import SlabLattice
random.seed(int(os.environ.get("FUZZ_SEED", "3")))
for case in range(int(os.environ.get("FUZZ_CASES", "30"))):
    rows = random.choice([256, 512, 1024, 2048, 4096])
    cols = random.choice([512, 544, 1024, 2048, 4096])
    if os.environ.get("FUZZ_RAGGED"):  # widths that are not a multiple of 16 (the wrap falls inside an octet), also resident ones
        cols = random.choice([290, 500, 1000, 1016, 1500, 2040, 3000])
    S = random.choice([4, 8, 16, 24, 32, 64, 128])
    k = random.choice([0, 4, 5, 8])
    if 2 * S > rows:
        continue
    n = S * random.choice([1, 2, 3]) + random.choice([0, 0, 3])
    seed = random.getrandbits(30)
    slab = SlabLattice(rows, cols, periodic=True, sweeps_per_exchange=S, seed=seed)
    slab.lat.set_kernel(_hip.KERNEL_AUTO, k)
    slab.randomize(); slab.set_model(1.0, 0.05, 2.3)
    slab.sweep(n); slab.synchronize(); torch.cuda.synchronize()
    a = (zlib.crc32(slab.local_spins().tobytes()), slab.observables())
    plain = _hip.Lattice(rows, cols, True)
    plain.set_kernel(_hip.KERNEL_GENERIC)
    plain.randomize(seed); plain.set_model(1.0, 0.05, 2.3)
    plain.sweep(n, seed, 0)
    b = (zlib.crc32(plain.get_spins().tobytes()), plain.observables())
    ok = a == b
    print(("ok  " if ok else "FAIL"), rows, cols, "S", S, "k", k, "n", n, flush=True)
    plain.close(); del slab
    if not ok:
        sys.exit(1)
print("all slab cases agree")
