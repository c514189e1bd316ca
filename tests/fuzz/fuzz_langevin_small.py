"""Randomized checks of the Langevin kernel, the one-workgroup lattice kernel and the lattice batch call (development aid)."""
import os, sys, random
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tsu-emulator_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from tsu import _hip
from oracle import oracle as ora
random.seed(int(os.environ.get("FUZZ_SEED", "9")))
for case in range(25):
    chains, dim = random.choice([1, 2, 5]), random.choice([1, 3, 4, 63, 64, 257, 1000])
    steps, spl = random.choice([1, 7, 40]), random.choice([0, 1, 3])
    rng = np.random.default_rng(random.getrandbits(30))
    x = rng.normal(size=(chains, dim)).astype(np.float32)
    k = rng.uniform(0.3, 3.0, size=dim).astype(np.float32); mu = rng.normal(size=dim).astype(np.float32)
    T, dt, gamma, seed, s0, c0 = random.choice([0.2, 1.0]), random.choice([0.005, 0.05]), random.choice([1.0, 2.5]), random.getrandbits(40), random.randrange(50), random.randrange(9)
    lc = _hip.LangevinChains(chains, dim); lc.set_energy(k, mu); lc.set_state(x); lc.set_kernel(spl)
    lc.step(steps, dt, gamma, T, seed, s0, c0)
    want = ora.langevin_quadratic_f32(x, k, mu, steps, dt, gamma, T, seed, step0=s0, chain0=c0)
    err = np.abs(lc.get_state() - want).max()
    print(("ok  " if err < 2e-4 else "FAIL"), "langevin", chains, dim, steps, spl, f"{err:.1e}", flush=True)
    lc.close()
    if err >= 2e-4: sys.exit(1)
for case in range(40):
    rows, cols = random.choice([1, 2, 3, 5, 16, 31, 32, 64, 100, 128]), random.choice([1, 2, 7, 16, 17, 33, 64, 100, 128])
    periodic = random.random() < 0.5 and rows % 2 == 0 and cols % 2 == 0 and rows >= 4 and cols >= 4
    if rows * ((cols + 15) // 16) > 1024: continue
    n = random.choice([1, 2, 9, 33])
    table = ora.ising2d_thresholds(random.choice([1.0, -1.0]), random.choice([0.0, 0.2]), random.choice([1.5, 2.3, 4.0]), random.choice([0, 1]))
    seed = random.getrandbits(40)
    lat = _hip.Lattice(rows, cols, periodic); lat.set_kernel(_hip.KERNEL_SMALL); lat.randomize(seed); lat.set_thresholds(table)
    lat.sweep(n, seed, 3)
    want = ora.ising2d_sweep(ora.ising2d_randomize(rows, cols, seed), periodic, table, n, seed, sweep0=3)
    ok = (lat.get_spins() == want).all() and lat.observables() == ora.ising2d_observables(want, periodic)
    print(("ok  " if ok else "FAIL"), "small", rows, cols, periodic, n, flush=True)
    lat.close()
    if not ok: sys.exit(1)
print("all cases agree")
