#!/usr/bin/env python3
"""bench.py -- the contract benchmark: spin-updates/s of the checkerboard Gibbs sweep on an L x L Ising lattice.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--L 4096] [--sweeps-per-step 256] [--sweeps-per-launch 8]

N = 1: BASELINE.json configs[1] (IsingModel2D 4096 x 4096 at T_c, periodic, physical bias mode) on one MI355X.
N > 1 (one rank per GPU, started by torch.distributed.run -- by the caller, or by bench.py itself as a child process
when it is run plainly as `python bench.py --gpus N`): weak scaling -- every rank owns an L x L row slab of a
(N*L) x L lattice, halo exchange by RCCL send/recv, no collective on the sweep path.  `--strong --L 16384` is BASELINE
configs[3] (ONE lattice cut into N slabs).  The run refuses to report a world different from --gpus.

A step = `sweeps-per-step` full lattice sweeps (each sweep updates every spin once: colour 0, then colour 1).
Inputs are resident in HBM before the timed region.  One JSON line is printed by rank 0.
"""
import argparse
import json
import os
import sys
import time

# dmabuf IPC (the host driver of this pool supports nothing else): RCCL's cross-process buffers need it; exported by the
# image already, kept here for environments built by hand
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tsu-emulator_amd"))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

T_C = 2.269185314213022  # 2 / ln(1 + sqrt 2)
HBM_PEAK_GBS = 8000.0    # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def cpu_baseline(L, budget_s=12.0):
    """The oracle's C restatement of the SAME kernel (checkerboard + Philox), one host core, bounded sample."""
    from oracle import oracle as ora
    table = ora.ising2d_thresholds(1.0, 0.0, T_C, 0)
    s = ora.ising2d_randomize(L, L, 42)
    s = ora.ising2d_sweep(s, True, table, 1, 42, sweep0=0)  # warm
    n, t0 = 0, time.perf_counter()
    while True:
        s = ora.ising2d_sweep(s, True, table, 2, 42, sweep0=1 + n)
        n += 2
        dt = time.perf_counter() - t0
        if dt > budget_s or n >= 256:
            break
    return {"value": L * L * n / dt, "unit": "spin-updates/s", "cores": 1, "kind": "port",
            "sample": f"oracle C restatement of the same checkerboard/Philox sweep, L={L}, T_c, {n} sweeps in {dt:.1f} s, "
                      f"host has {os.cpu_count()} cores"}


def cpu_reference_order(budget_s=4.0):
    """BASELINE configs[0]: the reference's own algorithm (per-site NumPy loop on the dense 1024 x 1024 J of
    IsingGrid 32 x 32, T = 2.5), restated in oracle.ref_gibbs_sweep and pinned by tests/golden -- one core."""
    from oracle import oracle as ora
    J = ora.ref_grid_coupling(32, 32, 1.0, False)
    Jb, hb = ora.ref_bit_coupling(J), ora.ref_bit_bias(J, np.zeros(1024), ora.MODE_COMPAT)
    rng = np.random.RandomState(42)
    bits = rng.randint(0, 2, size=1024)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        bits = ora.ref_gibbs_sweep(bits, Jb, hb, 2.5, rng.rand(1, 1024))
        n += 1
    dt = time.perf_counter() - t0
    return {"value": 1024 * n / dt, "unit": "spin-updates/s", "cores": 1, "kind": "port",
            "sample": f"reference-order per-site NumPy loop, IsingGrid 32x32 dense J, T=2.5, {n} sweeps in {dt:.1f} s"}


def cpu_numpy_checkerboard(L, budget_s=3.0):
    """a vectorised NumPy checkerboard heat-bath sweep (np.random uniforms) on one core --
    NOT the reference's algorithm (which is a per-site Python loop), just what NumPy alone reaches on this host."""
    rng = np.random.RandomState(42)
    s = (2 * rng.randint(0, 2, size=(L, L)) - 1).astype(np.int8)
    ii, jj = np.indices((L, L), sparse=True)
    masks = [((ii + jj) & 1) == c for c in (0, 1)]
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        for m in masks:
            nb = np.roll(s, 1, 0) + np.roll(s, -1, 0) + np.roll(s, 1, 1) + np.roll(s, -1, 1)
            p = 1.0 / (1.0 + np.exp(-2.0 * nb / T_C))
            s = np.where(m, np.where(rng.random_sample((L, L)) < p, 1, -1), s).astype(np.int8)
        n += 1
    dt = time.perf_counter() - t0
    return {"value": L * L * n / dt, "unit": "spin-updates/s", "cores": 1, "kind": "numpy-vectorised, not the reference",
            "sample": f"L={L}, T_c, {n} sweeps in {dt:.1f} s"}


def cpu_numpy_langevin(budget_s=2.0):
    """the reference's _langevin_step arithmetic (x - g dt/gamma + sqrt(2 T dt/gamma) randn, f64)
    with the analytic gradient of E = sum x^2, d = 2^20, one core."""
    d = 1 << 20
    x = np.zeros(d)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        x = x - (2.0 * x) * 0.01 + np.sqrt(2.0 * 0.01) * np.random.randn(d)
        n += 1
    dt = time.perf_counter() - t0
    return {"value": d * n / dt, "unit": "element-steps/s", "cores": 1, "kind": "numpy f64",
            "sample": f"d=2^20, {n} steps in {dt:.1f} s, variance {x.var():.3f}"}


def load_counters():
    """profiles/counters.json: per-launch PMC figures of the kernels (collected by tools/pmc_r02.sh; see its _comment)."""
    path = os.path.join(ROOT, "profiles", "counters.json")
    try:
        return json.load(open(path))
    except Exception:
        return {}


VALU_CYCLES_PER_WAVE_INSTRUCTION = 4.0  # one wave64 VALU instruction holds its SIMD's issue port ~4 cycles (VOP3 / packed /
# v_perm / v_mad_u64_u32: 4.2-4.4 measured, plain VOP2 2.4-2.7: profiles/r01_microbench_instruction_cost.txt)
N_SIMDS = 256 * 4


def issue_roofline(entry, updates_per_launch, launch_seconds, what="spin update"):
    """The ceiling that binds a VALU-bound kernel: wave-instructions issued per second against
    1024 SIMDs x clock / 4 cycles.  Instruction count and clock come from the PMC passes (profiles/), the time is live."""
    if not entry or not entry.get("valu_insts") or not entry.get("clock_ghz"):
        return None
    insts = float(entry["valu_insts"])
    peak = N_SIMDS * entry["clock_ghz"] * 1e9 / VALU_CYCLES_PER_WAVE_INSTRUCTION
    achieved = insts / launch_seconds
    return {"bound": "valu-issue", "achieved": achieved / 1e9, "peak": peak / 1e9, "unit": "G wave-instructions/s", "frac": achieved / peak,
            f"lane_instructions_per_{what.replace(' ', '_')}": insts * 64.0 / updates_per_launch,
            "valu_wave_instructions_per_launch": insts, "clock_ghz_in_pmc_pass": entry["clock_ghz"],
            "cycles_per_wave_instruction_assumed": VALU_CYCLES_PER_WAVE_INSTRUCTION,
            "note": "SQ_INSTS_VALU and GRBM_GUI_ACTIVE from profiles/counters.json (rocprofv3 --pmc passes), launch time from this run"}


def time_lattice(hip, ctx, L, k, sweeps, reps=3, cols=None, counters=None, ckey=None, sweeps_per_launch=None):
    rows, L = L, (cols or L)
    lat = hip.Lattice(rows, L, True, ctx=ctx)
    lat.randomize(42)
    lat.set_model(1.0, 0.0, T_C, hip.MODE_PHYSICAL)
    lat.set_kernel(hip.KERNEL_AUTO, k)
    lat.sweep(sweeps, 42, 0)
    ctx.synchronize()
    best = 1e30
    for r in range(reps):
        ctx.timer_begin()
        lat.sweep(sweeps, 42, sweeps * (r + 1))
        best = min(best, ctx.timer_end())
    s, b = lat.observables()
    lat.close()
    ups = rows * L * sweeps / (best * 1e-3)
    out = {"spin_updates_per_s": ups, "us_per_sweep": best * 1e3 / sweeps, "algorithmic_GBps": 2 * ups / 1e9,
           "frac_of_8TBps": 2 * ups / 1e9 / HBM_PEAK_GBS, "M": s / (rows * L), "E_per_site": -b / (rows * L)}
    entry = (counters or {}).get(ckey) if ckey else None
    if entry:
        # the PMC figures are per launch of `sweeps_per_launch` sweeps (a tile-resident launch runs the whole call)
        spl = sweeps_per_launch or sweeps
        launch_s = best * 1e-3 * spl / sweeps
        out["counters_key"] = ckey
        out["traffic_bytes_per_launch"] = entry.get("hbm_bytes")
        out["algorithmic_bytes_per_launch"] = 2.0 * rows * L * spl
        out["issue_roofline"] = issue_roofline(entry, float(rows) * L * spl, launch_s)
    return out


def time_langevin(hip, ctx, counters):
    dim, steps = 1 << 20, 500
    out = {}
    lc = hip.LangevinChains(1, dim, ctx=ctx)
    lc.set_energy(2.0, 0.0)
    lc.set_state(np.zeros((1, dim), np.float32))
    for name, spl in (("fused_500_steps_per_launch", 0), ("one_step_per_launch", 1)):
        lc.set_kernel(spl)
        lc.step(steps, 0.01, 1.0, 1.0, 7, 0)
        ctx.synchronize()
        ctx.timer_begin()
        lc.step(steps, 0.01, 1.0, 1.0, 7, steps)
        ms = ctx.timer_end()
        es = dim * steps / (ms * 1e-3)
        out[name] = {"element_steps_per_s": es, "effective_GBps_at_8B_per_element_step": 8 * es / 1e9}
        if spl == 0:
            # the state stays in registers for all 500 steps: real traffic is 8/500 B per element-step and the kernel is bound by
            # instruction issue (Philox + Box-Muller), not HBM
            out[name]["issue_roofline"] = issue_roofline(counters.get("k3_fused_d2p20_500"), dim * steps, ms * 1e-3, "element step")
            out[name]["note"] = "effective GB/s only: the state is read and written once per launch (8.4 MB), see issue_roofline"
        else:
            out[name]["note"] = "4 MiB of state, 3.7 us per launch: launch-bound, not HBM-bound (the HBM-bound case is below)"
    out["variance"] = float(lc.get_state().var())  # -> T / (k (1 - k dt / 2)) = 0.505
    lc.close()
    # the honestly HBM-bound K3 case: 256 chains x 2^20 (1 GiB of state), ONE step per launch: 8 B per element-step really move
    chains, steps = 256, 20
    lc = hip.LangevinChains(chains, dim, ctx=ctx)
    lc.set_energy(2.0, 0.0)
    lc.set_state(np.zeros((1, dim), np.float32))
    lc.set_kernel(1)
    lc.step(4, 0.01, 1.0, 1.0, 7, 0)
    ctx.synchronize()
    ctx.timer_begin()
    lc.step(steps, 0.01, 1.0, 1.0, 7, 4)
    ms = ctx.timer_end()
    lc.close()
    gbs = 8.0 * chains * dim * steps / (ms * 1e-3) / 1e9
    tr = counters.get("k3_unfused_256x2p20", {}).get("hbm_bytes")
    out["256_chains_one_step_per_launch"] = {
        "element_steps_per_s": chains * dim * steps / (ms * 1e-3), "us_per_launch": ms * 1e3 / steps,
        "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": tr,
                     "note": "algorithmic 8 B per element-step (fp32 read + write) x 2^28 elements per launch; traffic: PMC bytes per launch"}}
    # coupled quadratic energy E = 1/2 x^T A x (k3_coupled): one chain at d = 16384 is a matrix-vector product per step, A (1 GiB)
    # streamed once = 4 B per element of A and step
    d2, steps = 16384, 30
    A = np.zeros((d2, d2), np.float32)
    A[np.arange(d2), np.arange(d2)] = 2.0
    i = np.arange(d2 - 1)
    A[i, i + 1] = A[i + 1, i] = 0.5
    lc = hip.LangevinChains(1, d2, ctx=ctx)
    lc.set_coupling(A)
    del A
    lc.set_state(np.zeros((1, d2), np.float32))
    lc.step(5, 0.01, 1.0, 1.0, 7, 0)
    ctx.synchronize()
    ms = 1e30
    for rep in range(3):
        ctx.timer_begin()
        lc.step(steps, 0.01, 1.0, 1.0, 7, 5 + rep * steps)
        ms = min(ms, ctx.timer_end())
    var = float(lc.get_state().var())
    lc.close()
    gbs = 4.0 * d2 * d2 * steps / (ms * 1e-3) / 1e9
    out["coupled_quadratic_d16384_one_chain"] = {
        "us_per_step": ms * 1e3 / steps, "element_steps_per_s": d2 * steps / (ms * 1e-3), "variance": var,
        "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                     "traffic": counters.get("k3_coupled_d16384_one_chain", {}).get("hbm_bytes"),
                     "note": "k3_coupled, one launch per step: algorithmic d^2 x 4 B per step (the symmetric matrix streamed once, gradient "
                             "A x + b as an axpy over its rows); best of 3 calls of 30 steps; traffic: PMC bytes per launch"}}
    return out


def time_dense(hip, ctx, counters, n=16384):
    """BASELINE configs[2]: dense Gibbs, N=16384 fp32 couplings (1 GiB of J): natural visiting order (k2_own), a caller's order
    (update_order="random": one permutation per sweep) and eight replicas on one J (tempering ladder / independent chains)."""
    rng = np.random.default_rng(42)
    G = rng.standard_normal((n, n)).astype(np.float32)
    J = ((G + G.T) / 2 / np.sqrt(n)).astype(np.float32)
    np.fill_diagonal(J, 0.0)
    del G
    d = hip.DenseSystem(J, None, hip.DTYPE_F32, ctx=ctx)
    d.set_state(rng.integers(0, 2, size=n).astype(np.int8))
    # a loop of calls in steady state: from the second consecutive call on the kernel hands the fields from call to call, so
    # the timed call neither rebuilds them from scratch nor is the first to leave them behind
    d.sweep(1.0, 2, seed=1, sweep0=0)
    d.sweep(1.0, 2, seed=1, sweep0=2)
    ctx.synchronize()
    ctx.timer_begin()
    d.sweep(1.0, 16, seed=1, sweep0=4)  # one call = one launch: the fields stay with their rows' owners from sweep to sweep
    ms_call = ctx.timer_end()
    ms = ms_call / 16
    own, pipe = d.launch_counts()
    gbs = n * n * 4 / (ms * 1e-3) / 1e9
    e = counters.get("k2_own_N16384_f32", {})
    out = {"N": n, "dtype": "f32", "ms_per_sweep": ms, "spin_updates_per_s": n / (ms * 1e-3), "J_stream_GBps": gbs,
           "kernel": "k2_own" if own else "k2_pipe", "launches": {"k2_own": own, "k2_pipe": pipe},
           "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                        "traffic": e.get("hbm_bytes_per_sweep_steady"),
                        "note": "per SWEEP: algorithmic N^2 x 4 B (J streamed once) / time per sweep; traffic = PMC bytes per sweep of a lone launch "
                                "of 8 sweeps without its field pass ((5.22 - 0.54) GB / 8); k2_own reads only the rows of J^T of "
                                "the sites whose value changes (~41 % flip per sweep at T = 1, plus the toggles of the generations), so "
                                "the PMC traffic per sweep is BELOW the algorithmic bytes (profiles/r03_pmc_k2_own_N16384_f32.txt)"}}
    # update_order="random": the caller's permutation per sweep (uploaded with the call; validated on the host)
    k = 8
    order = np.array([rng.permutation(n) for _ in range(k)])
    d.sweep(1.0, 2, seed=2, sweep0=0, order=order[:2])
    ctx.synchronize()
    t0 = time.perf_counter()
    d.sweep(1.0, k, seed=2, sweep0=2, order=order)
    ctx.synchronize()
    ms_r = (time.perf_counter() - t0) * 1e3 / k
    out["random_order"] = {"ms_per_sweep": ms_r, "vs_natural": ms_r / ms, "roofline_frac": n * n * 4 / (ms_r * 1e-3) / 1e9 / HBM_PEAK_GBS,
                           "note": "wall clock of the call incl. the host-side permutation check and the upload of the orders (8 B per site and sweep)"}
    # eight replicas on ONE stream of J (tsu_dense_sweep_replicas: parallel_tempering / sample_chains above the one-workgroup kernels)
    R = 8
    sts = np.array([np.random.default_rng(100 + r).integers(0, 2, size=n) for r in range(R)], dtype=np.int8)
    sts = d.sweep_replicas(sts, [1.0] * R, 2, list(range(R)), [0] * R)
    ctx.synchronize()
    t0 = time.perf_counter()
    d.sweep_replicas(sts, [1.0] * R, 8, list(range(R)), [2] * R)
    ms_8 = (time.perf_counter() - t0) * 1e3 / 8
    out["eight_replicas"] = {"ms_per_all_replica_sweep": ms_8, "vs_single_chain_sweep": ms_8 / ms, "replica_sweeps_per_s": R / (ms_8 * 1e-3),
                             "note": "8 states advanced together by one launch; wall clock incl. the states' PCIe round trip (128 KiB each way)"}
    d.close()
    # the same system with fp64 couplings (the drop-in's default dtype: the reference's J is float64): 2 GiB streamed per sweep
    d = hip.DenseSystem(J.astype(np.float64), None, hip.DTYPE_F64, ctx=ctx)
    d.set_state(np.random.default_rng(42).integers(0, 2, size=n).astype(np.int8))
    d.sweep(1.0, 2, seed=1, sweep0=0)
    d.sweep(1.0, 2, seed=1, sweep0=2)
    ms64 = 1e30
    for rep in range(2):
        ctx.synchronize()
        ctx.timer_begin()
        d.sweep(1.0, 8, seed=1, sweep0=4 + 8 * rep)
        ms64 = min(ms64, ctx.timer_end() / 8)
    d.close()
    out["f64"] = {"ms_per_sweep": ms64, "J_stream_GBps": n * n * 8 / (ms64 * 1e-3) / 1e9,
                  "roofline_frac": n * n * 8 / (ms64 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                  "note": "fp64 couplings: algorithmic N^2 x 8 B per sweep; same kernel (k2_own<double>), the rows of J^T of the sites that change"}
    return out


def time_python_surface(hip, ctx):
    """The reference idioms through the Python surface, UNBOUND (no GibbsSampler.bind): what a caller who switches libraries sees.
    ``state = s.gibbs_sweep(state, J)`` + ``s.compute_energy(state, J)`` per step with J writeable (every byte hashed per call, the
    price of seeing in-place edits as the reference does) and with J frozen (``J.setflags(write=False)``: no hash, same API)."""
    from tsu.core import QuadraticEnergy, ThermalSamplingUnit, TSUConfig
    from tsu.gibbs import GibbsConfig, GibbsSampler
    from tsu.models.ising import IsingModel2D
    out = {}
    for n in (4096, 16384):
        rng = np.random.default_rng(n)
        G = rng.standard_normal((n, n)).astype(np.float32)
        J = ((G + G.T) / 2 / np.sqrt(n)).astype(np.float32)
        np.fill_diagonal(J, 0.0)
        del G
        s = GibbsSampler(GibbsConfig(temperature=1.0), seed=5, coupling_dtype="float32")
        state = rng.integers(0, 2, size=n)
        row = {}
        for label, steps in (("writeable_J_hashed_every_call", 3), ("frozen_J", 20)):
            if label == "frozen_J":
                J.setflags(write=False)
            state = s.gibbs_sweep(state, J)
            e = s.compute_energy(state, J)
            t0 = time.perf_counter()
            for _ in range(steps):
                state = s.gibbs_sweep(state, J)
                e = s.compute_energy(state, J)
            row[label] = {"ms_per_step_sweep_plus_energy": (time.perf_counter() - t0) * 1e3 / steps}
        row["energy"] = float(e)
        s.invalidate()
        out[f"gibbs_sweep_and_compute_energy_N{n}_f32"] = row
    m = IsingModel2D(4096, temperature=T_C, seed=3)
    m.gibbs_update(8)
    t0 = time.perf_counter()
    for _ in range(20):
        m.gibbs_update(8)
    mag = m.magnetization()
    dt = time.perf_counter() - t0
    out["IsingModel2D_4096_gibbs_update_8_sweeps"] = {"ms_per_call": dt * 1e3 / 20, "spin_updates_per_s": 4096 * 4096 * 8 * 20 / dt, "M": float(mag)}
    t = ThermalSamplingUnit(TSUConfig(temperature=1.0, dt=0.01, n_burnin=100, n_steps=400), seed=3)
    t0 = time.perf_counter()
    x = t.sample_boltzmann(QuadraticEnergy(2.0), n_samples=1, dim=2 ** 20)
    dt = time.perf_counter() - t0
    out["TSU_sample_boltzmann_QuadraticEnergy_dim_2^20"] = {"ms_per_call": dt * 1e3, "element_steps_per_s": (2 ** 20) * 500 / dt, "variance": float(x.var())}
    t0 = time.perf_counter()
    x = t.sample_boltzmann(lambda v: (v ** 2).sum(), n_samples=1, dim=2 ** 20)
    dt = time.perf_counter() - t0
    out["TSU_sample_boltzmann_lambda_dim_2^20"] = {"ms_per_call": dt * 1e3, "variance": float(x.var()),
                                                   "note": "the README idiom with a Python callable: ~140 probing evaluations recognise the uniform quadratic, then K3"}
    return out


def time_sparse_chain(hip, ctx, counters, n=1 << 24):
    """K5: colour-parallel sweeps of a 2^24-site open chain (the reference's IsingChain would need a 2 PB dense J)."""
    import scipy.sparse as sp
    from tsu.graph import canonical_csr
    A = canonical_csr(sp.diags([np.full(n - 1, 4.0), np.full(n - 1, 4.0)], [1, -1]))
    bias = np.full(n, -8.0)
    bias[0] = bias[-1] = -4.0
    order = np.concatenate([np.arange(0, n, 2), np.arange(1, n, 2)]).astype(np.int32)
    g = hip.SparseSystem(A.indptr, A.indices, A.data, bias, np.array([0, (n + 1) // 2, n], np.int32), order, ctx=ctx)
    g.set_state(np.random.default_rng(1).integers(0, 2, size=n).astype(np.int8))
    g.sweep(1.7, 5, seed=3, sweep0=0)
    ctx.synchronize()
    ctx.timer_begin()
    g.sweep(1.7, 50, seed=3, sweep0=5)
    ms = ctx.timer_end() / 50
    _, m = g.energy()
    g.close()
    # a uniform chain's colour classes are REGULAR and PAIRED (csrc/sparse.hip: k5_stencil4): no CSR streams; per update two neighbour
    # bits read, one byte written, and one byte of prepared decisions written (first class) or read (second class) = 4 B (the general
    # CSR kernel moved 17 + 13 deg = 43 B per update)
    alg = 4
    gbs = alg * n / (ms * 1e-3) / 1e9
    return {"sites": n, "ms_per_sweep": ms, "spin_updates_per_s": n / (ms * 1e-3),
            "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                         "traffic": ((counters.get("k5_stencil4_first_class_2p23", {}).get("hbm_bytes") or 0) +
                                     (counters.get("k5_stencil4_second_class_2p23", {}).get("hbm_bytes") or 0)) or None,
                         "valu_lane_instructions_per_update_first_class": ((counters.get("k5_stencil4_first_class_2p23", {}).get("valu_insts") or 0) * 64.0 / (n / 2)) or None,
                         "note": "regular, paired colour classes (k5_stencil4<1> + <2>): algorithmic 4 B per update = 34 MB per launch of one "
                                 "colour class (2^23 sites), so the sweep is bound by instruction issue -- ONE Philox block per PAIR of sites "
                                 "(20 64-bit multiply-adds, ~17 us per 2^23 blocks), computed by the first class's launch, which also prepares "
                                 "the second class's decisions by neighbour count; the second launch is a 4.4 TB/s byte shuffle (7.7 us) -- "
                                 "not by HBM; the round-2 CSR kernel moved 43 B per update (0.52 of the roofline at 9.6e10 updates/s); "
                                 "traffic: PMC bytes of the two launches of a sweep (the read half a lower estimate: 4 B/lane loads)"},
            "M": m / n}


def time_strong_slab(hip, SlabLattice, dist, torch, world, local_rank, backend, L, transport, steps, spx=32, k=8):
    """BASELINE configs[3]: ONE L x L lattice (16384^2) in `world` row slabs, halo exchange every `spx` sweeps; every rank calls it,
    the returned record is the same on all of them (max over ranks of the elapsed time)."""
    rows_local = L // world
    slab = SlabLattice(rows_local, L, periodic=True, sweeps_per_exchange=spx, seed=42, transport=transport)
    slab.lat.set_kernel(hip.KERNEL_AUTO, k)
    slab.randomize()
    slab.set_model(1.0, 0.0, T_C, hip.MODE_PHYSICAL)
    sps = 4 * spx
    timeout = float(os.environ.get("TSU_COMM_TIMEOUT_S", "120"))

    def barrier():
        slab.synchronize(timeout)
        torch.cuda.synchronize()
        dist.barrier()

    for _ in range(3):
        slab.sweep(sps)
    barrier()
    ex0 = slab.n_exchanges
    t0 = time.perf_counter()
    for _ in range(steps):
        slab.sweep(sps)
    barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}" if backend == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t[0])
    s_, b_ = slab.observables()
    n_ex = slab.n_exchanges - ex0
    if slab.comm is not None:
        slab.comm.close()
    slab.lat.close()
    ups = float(L) * L * sps * steps / elapsed
    return {"value": ups, "unit": "spin-updates/s", "scaling": "strong", "n_gpus": world, "lattice": f"{L}x{L} in {world} slabs of {rows_local}x{L}",
            "halo_transport": transport, "sweeps_per_exchange": spx, "sweeps_per_step": sps, "steps": steps, "ms_per_step": elapsed * 1e3 / steps,
            "halo_exchanges": n_ex, "roofline_frac_per_gpu": 2.0 * ups / world / 1e9 / HBM_PEAK_GBS,
            "M": s_ / (float(L) * L), "E_per_site": -b_ / (float(L) * L)}


def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--L", type=int, default=4096)
    ap.add_argument("--sweeps-per-step", type=int, default=0, help="0 = auto: 4 halo-exchange periods (256 sweeps at k=8)")
    ap.add_argument("--sweeps-per-exchange", type=int, default=0, help="N>1: 0 = auto, the multiple of sweeps-per-launch nearest 64")
    ap.add_argument("--ramp-steps", type=int, default=20, help="untimed steps before the W warmup steps (GPU clock ramp, ~30 ms)")
    ap.add_argument("--sweeps-per-launch", type=int, default=0, help="0 = auto: 8 up to 4096^2 per GPU, 5 above")
    ap.add_argument("--strong", action="store_true",
                    help="N>1: ONE L x L lattice cut into N slabs of L/N rows (BASELINE configs[3]: --L 16384 --gpus 8) "
                         "instead of the default weak scaling (L x L per GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    return ap


def launch_plan(gpus, env, argv, free_port=None):
    """What `python bench.py --gpus N ...` does about its process layout, decided BEFORE torch or the GPU is touched.

    * already inside a rank of a launcher (WORLD_SIZE set): run inline; the world must equal --gpus (checked in main),
    * --gpus 1 and no launcher: run inline on one GPU,
    * --gpus N > 1 and no launcher: spawn `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD
      process (never an exec: a process that has initialised the GPU must not be replaced) and relay its output and
      exit code.
    Returns ("inline", None), ("spawn", command list) or ("error", message)."""
    if gpus < 1:
        return "error", f"--gpus must be >= 1, got {gpus}"
    ws = env.get("WORLD_SIZE")
    if ws is not None:
        if int(ws) != gpus:
            return "error", (f"bench.py --gpus {gpus} was started inside a launcher with WORLD_SIZE={ws}: refusing to "
                             f"report n_gpus={ws} for a run asked to use {gpus}")
        return "inline", None
    if gpus == 1:
        return "inline", None
    port = free_port() if free_port else 29500
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return "spawn", cmd


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def main():
    args = build_parser().parse_args()
    mode, what = launch_plan(args.gpus, os.environ, sys.argv[1:], _free_port)
    if mode == "error":
        raise SystemExit("bench.py: " + what)
    if mode == "spawn":
        # one rank per GPU: the ranks are started by torch.distributed.run; this parent never imports torch or touches a GPU
        import subprocess
        env = dict(os.environ, MASTER_ADDR="127.0.0.1")
        proc = subprocess.run(what, env=env)
        raise SystemExit(proc.returncode)

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # one rank per GPU; BENCH_BACKEND=gloo lets several ranks share a GPU (rehearsal of the N > 1 path on a 1-GPU box)
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    if world > torch.cuda.device_count():
        if backend == "nccl":
            raise SystemExit(f"bench.py: --gpus {world} needs {world} GPUs, this node shows {torch.cuda.device_count()} "
                             "(RCCL wants one GPU per rank; BENCH_BACKEND=gloo rehearses the path with ranks sharing a GPU)")
        # rehearsal with several ranks on one GPU: two processes' tile-resident grids cannot both be on the chip, and half of
        # each would wait for the other half for ever (reported as a timeout) -- one launch per generation there
        os.environ.setdefault("TSU_K1_RESIDENT", "0")
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        from datetime import timedelta
        # (a rank that never arrives ends the run with an error after this long instead of holding the node)
        pg_timeout = timedelta(seconds=float(os.environ.get("BENCH_PG_TIMEOUT_S", "300")))
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"), timeout=pg_timeout)
        else:
            dist.init_process_group(backend, timeout=pg_timeout)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py: process group has {dist.get_world_size()} ranks, --gpus asked for {args.gpus}")

    from tsu import _hip as hip
    from tsu.distributed import SlabLattice
    ctx = hip.Context(local_rank)
    hip.Context._default = ctx
    L = args.L
    rows_local = L // world if (args.strong and world > 1) else L  # rows of this rank's slab
    if args.strong and world > 1 and L % world:
        raise SystemExit("--strong needs L divisible by the number of GPUs")
    # sweeps per generation / launch: the library's own choice (8: one halo octet = 16 columns covers 2 x 8 half-sweeps) unless asked
    k = args.sweeps_per_launch or 8
    # sweeps per halo exchange (tools/slab_rate.py: a 4096^2 slab does 2.55e12 upd/s at 64, 2.71e12 at 96..128, 2.62e12 at 192:
    # fewer launches and exchanges against taller tiles on the deep ghost rows; thin strong-scaling slabs: 32)
    spx = args.sweeps_per_exchange or (32 if (args.strong and world > 1) else k * round(128 / k))
    sps = args.sweeps_per_step or (4 * spx if (args.strong and world > 1) else 2 * spx)  # one step: 256 sweeps, ~1.5 ms at 4096^2
    if world > 1 and sps % spx:
        raise SystemExit("--sweeps-per-step must be a multiple of --sweeps-per-exchange")

    if world == 1:
        lat = hip.Lattice(L, L, True, ctx=ctx)
        lat.randomize(42)
        lat.set_model(1.0, 0.0, T_C, hip.MODE_PHYSICAL)
        lat.set_kernel(hip.KERNEL_AUTO, args.sweeps_per_launch)
        state = {"sweep": 0}

        def step():
            lat.sweep(sps, 42, state["sweep"])
            state["sweep"] += sps

        def barrier():
            ctx.synchronize()
            torch.cuda.synchronize()
        parallelism = "single GPU"
    else:
        # one halo exchange per spx sweeps; the slab keeps its own halo exact in between (deep ghost rows)
        # BENCH_TRANSPORT=rccl: the halo travels through the library's own RCCL calls below the C ABI (tsu_ising2d_halo_exchange)
        # instead of torch.distributed's point-to-point ops (the default)
        slab = SlabLattice(rows_local, L, periodic=True, sweeps_per_exchange=spx, seed=42, transport=os.environ.get("BENCH_TRANSPORT", "torch"))
        slab.lat.set_kernel(hip.KERNEL_AUTO, k)
        slab.randomize()
        slab.set_model(1.0, 0.0, T_C, hip.MODE_PHYSICAL)

        def step():
            slab.sweep(sps)

        def barrier():
            slab.synchronize(float(os.environ.get("TSU_COMM_TIMEOUT_S", "120")))  # bounded: raises instead of hanging on a lost peer
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
        parallelism = (f"{world} row slabs of {rows_local}x{L}, RCCL send/recv halo ({2 * spx} rows every {spx} sweeps = "
                       f"{spx // k} launches), no collective")

    # The first ~10 ms of work after idle run ~6 % slower (clock ramp): spend them before the warmup steps.
    # (a fixed count, so that every rank issues the same halo exchanges)
    for _ in range(args.ramp_steps):
        step()
    for _ in range(args.warmup):
        step()
    barrier()
    the_lattice = lat if world == 1 else slab.lat
    launches0 = the_lattice.launch_count()
    exchanges0 = 0 if world == 1 else slab.n_exchanges
    ctx.timer_begin()  # HIP events on the stream the kernels are launched on
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ev_ms = ctx.timer_end()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed, ev_ms], dtype=torch.float64, device=f"cuda:{local_rank}" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, ev_ms = float(t[0]), float(t[1])

    updates = float(rows_local) * L * world * sps * args.steps
    value = updates / elapsed
    # kernel launches actually issued in the timed region (a tile-resident launch runs a whole step: many generations
    # of k sweeps with the tile in LDS; otherwise one launch per k sweeps)
    n_launches = int(the_lattice.launch_count() - launches0)
    avg_launch_ms = ev_ms / n_launches
    sweeps_per_launch_avg = sps * args.steps / n_launches
    # algorithmic bytes: 2 B per spin update (int8 read + write); a launch of s sweeps on this GPU's sites carries
    # 2 * sites * s bytes
    alg_bytes_per_launch = 2.0 * rows_local * L * sweeps_per_launch_avg
    achieved = alg_bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9
    counters = load_counters()
    resident = sweeps_per_launch_avg > k
    # nibble colour planes: whole periodic lattices above 2^25 sites (csrc/ising2d_tiled.hip: pick_variant)
    nib = world == 1 and L % 16 == 0 and L * L > (1 << 25) and os.environ.get("TSU_K1_NIBBLE", "1") != "0"
    ckey = (f"k1_resident{'_nib' if nib else ''}_L{L}_s{int(round(sweeps_per_launch_avg))}" if resident
            else f"k1_tiled{'_nib' if nib else ''}_L{L}_k{int(round(sweeps_per_launch_avg))}")
    centry = counters.get(ckey) if world == 1 else None
    traffic = centry.get("hbm_bytes") if centry else None
    issue = issue_roofline(centry, float(rows_local) * L * sweeps_per_launch_avg, avg_launch_ms * 1e-3) if centry else None

    if world == 1:
        s, b = lat.observables()
    else:
        s, b = slab.observables()

    multi_extra = None
    if world > 1 and not args.no_extra and not args.strong:
        # BASELINE configs[3] in the same record: ONE 16384 x 16384 lattice cut into `world` row slabs (strong scaling), with both
        # halo transports (torch.distributed point-to-point ops; the library's own RCCL calls below the C ABI)
        multi_extra = {}
        Ls = int(os.environ.get("BENCH_STRONG_L", "16384"))
        if Ls % world == 0 and (Ls // world) >= 64:
            slab.lat.close()
            for transport in (("torch", "rccl") if backend == "nccl" else ("torch",)):
                # (a side measurement must not cost the record its main line: a failure is reported in its place)
                try:
                    multi_extra[f"strong_L{Ls}_{transport}"] = time_strong_slab(hip, SlabLattice, dist, torch, world, local_rank, backend, Ls,
                                                                               transport, steps=max(2, min(args.steps, 10)))
                except Exception as e:  # noqa: BLE001
                    multi_extra[f"strong_L{Ls}_{transport}"] = {"error": f"{type(e).__name__}: {e}"[:400]}
                    break
    if rank == 0:
        out = {
            "metric": "spin-updates/sec on L×L 2D Ising Gibbs sweep; achieved HBM GB/s vs peak",
            "value": value, "unit": "spin-updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True, "scaling": "strong" if (args.strong and world > 1) else "weak", "vs_baseline": None,
            "dtype": "int8", "data": "synthetic (Philox i.i.d. +-1 start, seed 42)",
            "backend": backend if world > 1 else None, "ranks_seen": world, "gpus_requested": args.gpus,
            "halo_transport": (os.environ.get("BENCH_TRANSPORT", "torch") if world > 1 else None),
            "halo_exchanges_in_timed_region": (slab.n_exchanges - exchanges0) if world > 1 else None,
            "config": {"workload": f"IsingModel2D {rows_local}x{L} per GPU at T_c=2.269185, J=1, h=0, periodic, checkerboard Gibbs sweep "
                                   "(BASELINE.json configs[1])", "L": L, "lattice_rows": rows_local * world, "lattice_cols": L,
                       "sweeps_per_step": sps, "sweeps_per_generation": k, "sweeps_per_launch": sweeps_per_launch_avg, "sweeps_per_exchange": spx if world > 1 else None,
                       "clock_ramp_steps_before_warmup": args.ramp_steps,
                       "kernel": ("k1_resident (tiles stay in LDS across generations of k sweeps, boundary strips exchanged through HBM)"
                                  if resident else "k1_tiled2 (LDS halo tiles, row-pair inner loop)") + (", nibble colour planes" if nib else ""),
                       "counters_key": ckey if centry else None,
                       "bias_mode": "physical", "parallelism": parallelism, "launches": n_launches,
                       "avg_launch_us": avg_launch_ms * 1e3, "algorithmic_bytes_per_launch": alg_bytes_per_launch},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "note": "achieved = algorithmic 2 B/spin-update x sites x sweeps per launch / avg launch time (HIP events "
                                 "on the launch stream over the timed region); tiles live in LDS for many sweeps, so real "
                                 "HBM traffic is a small fraction of that and the kernel is VALU (Philox) bound: see issue_roofline"},
            "issue_roofline": issue,
            "observables": {"M": s / (float(rows_local) * L * world), "E_per_site": -b / (float(rows_local) * L * world),
                            "note": "physical mode, random start; u(T_c) = -sqrt(2) = -1.4142 is approached slowly (critical slowing down)"},
        }
        if multi_extra is not None:
            out["extra"] = multi_extra
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(L)
            out["cpu_reference_order"] = cpu_reference_order()
            out["cpu_numpy_checkerboard"] = cpu_numpy_checkerboard(min(L, 2048))
            out["cpu_numpy_langevin"] = cpu_numpy_langevin()
        if not args.no_extra and world == 1:
            lat.close()
            extra = {}

            def side(name, fn, *a, **kw):
                # (a side measurement must not cost the record its main line: a failure is reported in its place)
                try:
                    extra[name] = fn(*a, **kw)
                except Exception as e:  # noqa: BLE001
                    extra[name] = {"error": f"{type(e).__name__}: {e}"[:400]}

            # 2^25 sites: the largest lattice whose tiles all stay resident in LDS (256 tiles of 256 x 512)
            side("ising2d_4096x8192", time_lattice, hip, ctx, 4096, 0, 240, cols=8192)
            # 8192^2: one 512 x 512 nibble-plane tile per CU, resident in LDS; 16384^2: 256 x 512 nibble tiles, 8 sweeps per launch
            side("ising2d_L8192", time_lattice, hip, ctx, 8192, 0, 256, counters=counters, ckey="k1_resident_nib_L8192_s256")
            side("ising2d_L16384", time_lattice, hip, ctx, 16384, 0, 240, counters=counters, ckey="k1_tiled_nib_L16384_k8", sweeps_per_launch=8)
            # BASELINE configs[0], the reference's own CPU-runnable case (cpu_reference_order times its loop on the same lattice)
            side("ising2d_32x32", time_lattice, hip, ctx, 32, 0, 20000)
            # a width that is not a multiple of 16 (the wrap falls inside an octet): IsingModel2D(1000)
            side("ising2d_1000x1000", time_lattice, hip, ctx, 1000, 0, 4096)
            # lattices that do not divide into whole tiles stay tile-resident through the flexible cut (balanced tile rows of
            # different heights, a narrower last tile column): nibble planes at 6000^2, byte planes (ragged width) at 5000^2
            for L2 in (5000, 6000):
                side(f"ising2d_L{L2}", time_lattice, hip, ctx, L2, 0, 240)
            side("langevin_dim_2^20", time_langevin, hip, ctx, counters)
            side("dense_gibbs", time_dense, hip, ctx, counters)
            side("sparse_chain_2^24", time_sparse_chain, hip, ctx, counters)
            side("python_surface_unbound", time_python_surface, hip, ctx)
            out["extra"] = extra
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
