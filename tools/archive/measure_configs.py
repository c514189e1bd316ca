"""SURVEY section 8(d) measurement table on one MI355X: C2 (hot / cold / random start, several L), C3 (dense, f32 and
f64), C4's slab with several halo depths (world size 1: a slab that is its own neighbour), C5 (single chain and 256
chains).  Writes gpurun_out/configs_table.txt; development aid, bench.py is the contract."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tsu-emulator_amd"))
import numpy as np
import torch
from tsu import _hip
from tsu.distributed import SlabLattice

T_C = 2.269185314213022
ctx = _hip.Context.default()
lines = []


def say(s):
    print(s, flush=True)
    lines.append(s)


say(f"device: {ctx.device_info()}")
say("-- C2 checkerboard sweep, periodic, physical mode, T_c, upd/s (best of 3 x 400 sweeps after 400 warm-up sweeps)")
for L, k in ((4096, 8), (8192, 5), (16384, 5)):
    for start in ("random", "cold(+1)", "hot->cold T=1.5"):
        lat = _hip.Lattice(L, L, True, ctx=ctx)
        if start == "cold(+1)":
            lat.fill(1)
        else:
            lat.randomize(42)
        lat.set_model(1.0, 0.0, 1.5 if "T=1.5" in start else T_C)
        lat.set_kernel(_hip.KERNEL_AUTO, k)
        n = 400
        lat.sweep(n, 42, 0); ctx.synchronize()
        best = 1e9
        for r in range(3):
            ctx.timer_begin(); lat.sweep(n, 42, n * (r + 1)); best = min(best, ctx.timer_end())
        s, b = lat.observables()
        ups = L * L * n / (best * 1e-3)
        say(f"L={L:6d} k={k} start={start:16s} {ups:.3e} upd/s  {2 * ups / 8e12 * 100:5.1f} % of 8 TB/s   M={s / L / L:+.4f} E/N={-b / L / L:+.4f}")
        lat.close()

say("-- C4 slab (world size 1, own neighbour): 4096 rows x 4096 cols per GPU, k=8 per launch, sweeps per exchange S (ghost 2S rows)")
for S in (8, 16, 32, 64):
    slab = SlabLattice(4096, 4096, periodic=True, sweeps_per_exchange=S, seed=1)
    slab.lat.set_kernel(_hip.KERNEL_AUTO, 8)
    slab.randomize(); slab.set_model(1.0, 0.0, T_C)
    slab.sweep(64); slab.synchronize(); torch.cuda.synchronize()
    n = 64 * 10
    t0 = time.perf_counter(); slab.sweep(n); slab.synchronize(); torch.cuda.synchronize(); t = time.perf_counter() - t0
    say(f"S={S:3d}: {4096 * 4096 * n / t:.3e} upd/s")
    del slab
slab = SlabLattice(2048, 16384, periodic=True, sweeps_per_exchange=32, seed=1)
slab.lat.set_kernel(_hip.KERNEL_AUTO, 8)
slab.randomize(); slab.set_model(1.0, 0.0, T_C)
slab.sweep(64); slab.synchronize(); torch.cuda.synchronize()
t0 = time.perf_counter(); slab.sweep(640); slab.synchronize(); torch.cuda.synchronize(); t = time.perf_counter() - t0
say(f"C4 shape, one rank's slab 2048 x 16384, k=8, S=32 (bench.py --strong): {2048 * 16384 * 640 / t:.3e} upd/s")
del slab

say("-- C3 dense Gibbs, spin-glass J = (G + G^T)/2/sqrt(N), T=1, natural order")
for n, dt_ in ((4096, _hip.DTYPE_F32), (4096, _hip.DTYPE_F64), (16384, _hip.DTYPE_F32), (16384, _hip.DTYPE_F64)):
    rng = np.random.default_rng(42)
    G = rng.standard_normal((n, n)).astype(np.float32)
    J = (G + G.T) / np.float32(2 * np.sqrt(n))
    np.fill_diagonal(J, 0.0)
    del G
    if dt_ == _hip.DTYPE_F64:
        J = J.astype(np.float64)
    d = _hip.DenseSystem(J, None, dt_, ctx=ctx)
    d.set_state(rng.integers(0, 2, size=n).astype(np.int8))
    d.sweep(1.0, 3, seed=1, sweep0=0); ctx.synchronize()
    ctx.timer_begin(); d.sweep(1.0, 10, seed=1, sweep0=3); ms = ctx.timer_end() / 10
    eb = 4 if dt_ == _hip.DTYPE_F32 else 8
    say(f"N={n:6d} {'f32' if eb == 4 else 'f64'}: {ms:.3f} ms/sweep  {n / ms * 1e3:.3e} upd/s  J stream {n * n * eb / ms / 1e6:.0f} GB/s")
    d.close(); del J

say("-- C5 Langevin, E = sum x^2 (k=2), T=1, dt=0.01, 500 steps")
for chains in (1, 256):
    dim = 1 << 20
    lc = _hip.LangevinChains(chains, dim, ctx=ctx)
    lc.set_energy(2.0, 0.0)
    lc.set_state(np.zeros((chains, dim), np.float32))
    for name, spl in (("fused", 0), ("1 step/launch", 1)):
        steps = 500 if (chains == 1 or spl == 0) else 50
        lc.set_kernel(spl)
        lc.step(steps, 0.01, 1.0, 1.0, 7, 0); ctx.synchronize()
        ctx.timer_begin(); lc.step(steps, 0.01, 1.0, 1.0, 7, steps); ms = ctx.timer_end()
        es = chains * dim * steps / (ms * 1e-3)
        say(f"chains={chains:4d} dim=2^20 {name:14s}: {es:.3e} element-steps/s   ({8 * es / 1e9:.0f} GB/s at 8 B per element-step)")
    v = lc.get_state()[0].var()
    say(f"    variance of chain 0: {v:.4f} (stationary 0.5051)")
    lc.close()
os.makedirs("gpurun_out", exist_ok=True)
open("gpurun_out/configs_table.txt", "w").write("\n".join(lines) + "\n")
