"""One rank's slab of the weak-scaling bench on one GPU (the rank is its own neighbour: device copies instead of RCCL)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tsu-emulator_amd"))
import torch
from tsu import _hip
from tsu.distributed import SlabLattice
torch.cuda.set_device(0)
for rows, cols, S in [(int(a), int(b), int(c)) for a, b, c in (t.split("x") for t in os.environ.get("SLABS", "4096x4096x64,4096x4096x128,2048x16384x32").split(","))]:
    lat = SlabLattice(rows, cols, periodic=True, sweeps_per_exchange=S, seed=42, transport=os.environ.get("TRANSPORT", "torch"))
    lat.randomize()
    lat.set_model(1.0, 0.0, 2.269185, _hip.MODE_PHYSICAL)
    lat.sweep(4 * S)
    lat.synchronize()
    n = 16 * S
    t = time.perf_counter()
    lat.sweep(n)
    lat.synchronize()
    dt = time.perf_counter() - t
    print("transport %s: slab %5d x %5d, %3d sweeps per exchange: %.3e upd/s (%.2f us/sweep)" % (os.environ.get("TRANSPORT", "torch (device copy at world size 1)"), rows, cols, S, rows * cols * n / dt, dt / n * 1e6), flush=True)
    del lat
