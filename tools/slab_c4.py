"""One rank's slab of BASELINE configs[3] (16384^2 over 8 GPUs = 2048 x 16384 per GPU) on one GPU: (k, S) scan."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tsu-emulator_amd"))
import torch
from tsu import _hip
from tsu.distributed import SlabLattice
R, C = int(os.environ.get("ROWS", 2048)), int(os.environ.get("COLS", 16384))
for k, S in [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(5, 60), (5, 5), (5, 10), (4, 20), (4, 8), (4, 4), (8, 8)]:
    slab = SlabLattice(R, C, periodic=True, sweeps_per_exchange=S, seed=1)
    slab.lat.set_kernel(_hip.KERNEL_AUTO, k)
    slab.randomize(); slab.set_model(1.0, 0.0, 2.269185)
    slab.sweep(S * 2); slab.synchronize(); torch.cuda.synchronize()
    n = S * max(1, 400 // S)
    t0 = time.perf_counter(); slab.sweep(n); slab.synchronize(); torch.cuda.synchronize(); t = time.perf_counter() - t0
    print(f"slab {R}x{C} k={k} S={S}: {R * C * n / t:.3e} upd/s", flush=True)
    del slab
