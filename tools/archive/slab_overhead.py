"""How much host time does one generation of the slab loop cost?  (development aid, world_size 1)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tsu-emulator_amd"))
import torch
from tsu import _hip
from tsu.distributed import SlabLattice
L, k = 4096, 8
for overlap, spx in ((False, 8), (True, 8), (False, 32), (False, 64)):
    lat = SlabLattice(L, L, periodic=True, sweeps_per_exchange=spx, seed=1, overlap=overlap)
    lat.lat.set_kernel(_hip.KERNEL_AUTO, k)
    lat.randomize(); lat.set_model(1.0, 0.0, 2.269185)
    lat.sweep(64); lat.synchronize(); torch.cuda.synchronize()
    n = 64 * 20
    t0 = time.perf_counter(); lat.sweep(n); t_host = time.perf_counter() - t0
    lat.synchronize(); torch.cuda.synchronize(); t_all = time.perf_counter() - t0
    gens = n // k
    print(f"overlap={overlap} sweeps/exchange={spx} split={lat._split}: host {t_host / gens * 1e6:.1f} us/generation, total {t_all / gens * 1e6:.1f} us/generation, "
          f"{L * L * n / t_all:.3e} upd/s")
plain = _hip.Lattice(L, L, True); plain.randomize(1); plain.set_model(1.0, 0.0, 2.269185); plain.set_kernel(0, k)
plain.sweep(64, 1, 0); _hip.Context.default().synchronize()
t0 = time.perf_counter(); plain.sweep(64 * 20, 1, 64); _hip.Context.default().synchronize(); t = time.perf_counter() - t0
print(f"plain lattice: {t / (64 * 20 // k) * 1e6:.1f} us/generation, {L * L * 64 * 20 / t:.3e} upd/s")
