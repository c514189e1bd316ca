"""Randomized check of the dense sweep paths against the oracle (development aid)."""
import os, sys, random
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tsu-emulator_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from tsu import _hip
from oracle import oracle as ora
random.seed(int(os.environ.get("FUZZ_SEED", "5")))
for case in range(int(os.environ.get("FUZZ_CASES", "30"))):
    n = random.choice([1, 7, 63, 64, 65, 127, 128, 129, 500, 1023, 2050, 4096, 4097, 5555])
    if os.environ.get("FUZZ_SMALL"):  # the one-wave kernel: up to 192 fp32 / 128 fp64 sites
        n = random.randint(1, 200)
    if os.environ.get("FUZZ_MID"):  # the one-workgroup kernel and its hand-over to the grid-wide path
        n = random.randint(129, 640)
    if os.environ.get("FUZZ_PIPE"):  # the pipeline kernel: N a multiple of 4 from 1024, one / two / three superblocks
        n = 4 * random.randint(256, 2300)
    if os.environ.get("FUZZ_PIPE_LOW"):  # the pipeline kernel just above the one-workgroup kernel's range: a single short superblock
        n = 4 * random.randint(113, 300)
    own = bool(os.environ.get("FUZZ_OWN"))  # the owner-computes kernel: N a multiple of 4 from 2048; natural / caller's order / replicas
    if own:
        n = 4 * random.randint(512, 2400)
    f32 = random.random() < 0.5
    sym = random.random() < 0.7
    T = random.choice([0.1, 0.5, 1.0, 3.0])
    sweeps = random.choice([1, 2, 5, 13])
    rng = np.random.default_rng(random.getrandbits(30))
    J = rng.standard_normal((n, n)) / max(1.0, np.sqrt(n)) * random.choice([0.3, 1.0, 3.0])
    if sym:
        J = (J + J.T) / 2
    if (own or os.environ.get("FUZZ_SMALL") or os.environ.get("FUZZ_MID") or os.environ.get("FUZZ_PIPE") or os.environ.get("FUZZ_PIPE_LOW")) and random.random() < 0.3:  # ferromagnet / antiferromagnet: long dependency chains
        J = np.full((n, n), random.choice([1.0, -1.0, 0.25]) * random.choice([1.0, 1.0 / max(1, n)]))
        np.fill_diagonal(J, 0.0)
    if f32:
        J = J.astype(np.float32).astype(np.float64)
    b = rng.normal(size=n) * 0.3 if random.random() < 0.5 else None
    st = rng.integers(0, 2, size=n).astype(np.int8)
    d = _hip.DenseSystem(J, b, _hip.DTYPE_F32 if f32 else _hip.DTYPE_F64)
    d.set_state(st)
    seed, s0 = random.getrandbits(40), random.randrange(1000)
    what = random.choice(["natural", "natural", "order", "replicas"]) if own else "natural"
    if own:
        sweeps = min(sweeps, 3)
        os.environ["TSU_K2_OWN_SB"] = str(random.choice([1024, 2048, 4096, 8192]))
        os.environ["TSU_K2_OWN_M"] = str(random.choice([1, 1, 2, 4])) if what != "replicas" else "1"
    if what == "natural":
        d.sweep(T, sweeps, seed=seed, sweep0=s0)
        ok = (d.get_state() == ora.dense_sweep_philox(st, J, b, T, sweeps, seed, sweep0=s0)).all()
    elif what == "order":
        order = np.array([rng.permutation(n) for _ in range(sweeps)])
        d.sweep(T, sweeps, seed=seed, sweep0=s0, order=order)
        ok = (d.get_state() == ora.dense_sweep_philox(st, J, b, T, sweeps, seed, sweep0=s0, order=order)).all()
    else:
        R = random.randint(2, 9)
        sts = rng.integers(0, 2, size=(R, n)).astype(np.int8)
        temps = [T * (1 + 0.2 * r) for r in range(R)]
        out = d.sweep_replicas(sts, temps, sweeps, [seed + r for r in range(R)], [s0 + r for r in range(R)], replicas=list(range(R)))
        ok = all((out[r] == ora.dense_sweep_philox(sts[r], J, b, temps[r], sweeps, seed + r, sweep0=s0 + r, replica=r)).all() for r in range(R))
    if own and sum(d.launch_counts()) == 0:
        print("(not on a one-launch kernel)", end=" ")
    print(("ok  " if ok else "FAIL"), what, "n", n, "f32" if f32 else "f64", "sym" if sym else "asym", "T", T, "sweeps", sweeps,
          os.environ.get("TSU_K2_OWN_SB", ""), os.environ.get("TSU_K2_OWN_M", ""), d.launch_counts(), flush=True)
    d.close()
    if not ok:
        sys.exit(1)
print("all dense cases agree")
