"""Soak of the dense pipeline kernel: the same call repeated must give the same state every time (no timing-dependent
outcome), and that state must be the barrier kernel's (second process, TSU_K2_PIPE=0).  usage: pipe_soak.py [reps]"""
import os, subprocess, sys, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tsu-emulator_amd"))
import numpy as np

CASES = ((16384, 1.0, 24), (16384, 0.2, 12), (12288, 1.0, 24), (8192, 0.5, 24), (4096, 1.0, 40), (2048, 1.0, 40))


def run(n, T, sweeps, reps):
    from tsu import _hip as hip
    ctx = hip.Context.default()
    rng = np.random.default_rng(n)
    G = rng.standard_normal((n, n)).astype(np.float32)
    J = ((G + G.T) / 2 / np.sqrt(n)).astype(np.float32)
    np.fill_diagonal(J, 0.0)
    s0 = rng.integers(0, 2, size=n).astype(np.int8)
    d = hip.DenseSystem(J, None, hip.DTYPE_F32, ctx=ctx)
    out = []
    for r in range(reps):
        d.set_state(s0)
        d.sweep(T, sweeps, seed=3, sweep0=1)
        out.append(d.get_state())
    d.close()
    return out


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        res = {}
        for n, T, sweeps in CASES:
            res[f"{n}_{T}"] = run(n, T, sweeps, 1)[0]
        np.savez(sys.argv[2], **res)
        sys.exit(0)
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    tmp = os.path.join(tempfile.gettempdir(), "pipe_soak_ref.npz")
    subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", tmp], env=dict(os.environ, TSU_K2_PIPE="0"))
    ref = np.load(tmp)
    bad = 0
    for n, T, sweeps in CASES:
        outs = run(n, T, sweeps, reps)
        same = all((o == outs[0]).all() for o in outs)
        agrees = bool((outs[0] == ref[f"{n}_{T}"]).all())
        print(f"n={n} T={T} {sweeps} sweeps x {reps} repetitions: {'identical' if same else 'DIFFERENT RUNS'}; {'equals' if agrees else 'DIFFERS FROM'} the barrier kernel", flush=True)
        bad += (not same) + (not agrees)
    sys.exit(1 if bad else 0)
