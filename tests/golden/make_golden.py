#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the UNMODIFIED reference.

Run in the build container only (the reference tree does not exist on the GPU box):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg \
        python /root/repo/tests/golden/make_golden.py

The script imports the reference package from /root/reference (read-only), drives its
public entry points with seeded inputs, and stores ONLY arrays (inputs, the RNG draws the
run consumed, and outputs) as .npz files.  No reference source text or bytecode is stored.

Fixture ids follow SURVEY.md section 8(c): G1..G9; G10 (equilibrium observables of the reference's own sampler at
T = 2.0, T_c, 2.5 on 16 x 16 and 32 x 32 lattices, ~15 minutes on 7 cores) pins the acceptance criterion of BASELINE.json;
G11 / G12: the reference's own sampling / optimisation benchmark runs (seed 42, quick and full).
"""
import os
import sys

sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")
REF = os.environ.get("TSU_REFERENCE", "/root/reference")
sys.path.insert(0, REF)

import numpy as np  # noqa: E402

import tsu  # noqa: E402  (the reference)
from tsu.core import ThermalSamplingUnit, TSUConfig  # noqa: E402
from tsu.gibbs import GibbsConfig, GibbsSampler  # noqa: E402
from tsu.models.ising import IsingChain, IsingConfig, IsingGrid, IsingModel  # noqa: E402

assert os.path.realpath(tsu.__file__).startswith(os.path.realpath(REF)), tsu.__file__
OUT = os.path.dirname(os.path.abspath(__file__))


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote", path, {k: np.asarray(v).shape for k, v in arrays.items()})


# ---------------------------------------------------------------- G1 / G2: dense Gibbs
def g1_g2():
    rng = np.random.default_rng(2024)
    n = 12
    J = rng.normal(size=(n, n))
    J = (J + J.T) / 2.0  # symmetric, non-zero diagonal kept on purpose
    bias = rng.normal(size=n) * 0.5
    T, burnin, n_sweeps, n_samples, seed = 0.8, 3, 2, 5, 123
    total_sweeps = burnin + n_sweeps * n_samples

    for order, tag in (("sequential", "g1_dense_sequential"), ("random", "g2_dense_random")):
        cfg = GibbsConfig(temperature=T, n_burnin=burnin, n_sweeps=n_sweeps, update_order=order)
        np.random.seed(seed)
        samples = GibbsSampler(cfg).sample_boltzmann(J, bias=bias, n_samples=n_samples)
        # replay the global-RNG draws the run consumed, in the order it consumed them
        np.random.seed(seed)
        init = np.random.randint(0, 2, size=n)
        perms = np.zeros((total_sweeps, n), dtype=np.int64)
        unis = np.zeros((total_sweeps, n), dtype=np.float64)
        for s in range(total_sweeps):
            perms[s] = np.random.permutation(n) if order == "random" else np.arange(n)
            unis[s] = np.random.rand(n)
        save(tag, J=J, bias=bias, T=T, burnin=burnin, n_sweeps=n_sweeps, n_samples=n_samples,
             seed=seed, init=init, perms=perms, uniforms=unis, samples=samples)

    # explicit initial_state + bias=None + asymmetric J + int8 state dtype through gibbs_sweep
    Ja = rng.normal(size=(7, 7))
    cfg = GibbsConfig(temperature=1.3, n_burnin=0, n_sweeps=1)
    st0 = rng.integers(0, 2, size=7)
    np.random.seed(7)
    out = GibbsSampler(cfg).gibbs_sweep(st0, Ja, None, n_sweeps=4)
    np.random.seed(7)
    u = np.random.rand(4, 7)
    smp = GibbsSampler(cfg)
    np.random.seed(8)
    sb = smp.sample_boltzmann(Ja, n_samples=3, burnin=0, initial_state=st0)
    np.random.seed(8)
    u2 = np.random.rand(3, 7)
    save("g1b_dense_asymmetric", J=Ja, T=1.3, init=st0, sweep_out=out, uniforms=u,
         sb_samples=sb, sb_uniforms=u2, sample_count=smp.sample_count)


# ---------------------------------------------------------------- G3: lattice builder
def g3():
    arrs = {}
    cases = [((3, 4), False), ((3, 4), True), ((4, 4), False), ((4, 4), True),
             ((2, 5), False), ((2, 5), True), ((1, 4), False), ((1, 4), True), ((5, 1), True)]
    rng = np.random.default_rng(5)
    for (r, c), per in cases:
        for Jc, hf in ((1.0, 0.0), (-0.7, 0.3)):
            key = f"r{r}c{c}p{int(per)}J{Jc}h{hf}"
            g = IsingGrid((r, c), J=Jc, config=IsingConfig(temperature=1.7, external_field=hf,
                                                           n_burnin=2, n_sweeps=1), periodic=per)
            arrs[key + "_J"] = g.J
            arrs[key + "_Jbit"] = g._get_bit_coupling()
            arrs[key + "_hbit"] = g._get_bit_bias()
            states = rng.choice([-1, 1], size=(6, r * c))
            arrs[key + "_states"] = states
            arrs[key + "_energy"] = np.array([g.energy(s) for s in states])
            np.random.seed(11)
            arrs[key + "_samples"] = g.sample(n_samples=4)
    save("g3_grid_builder", **arrs)


# ---------------------------------------------------------------- G4: config-1 trajectory
def g4():
    checkpoints = [1, 10, 100, 1000]
    arrs = {"checkpoints": np.array(checkpoints)}
    for per in (False, True):
        g = IsingGrid((32, 32), J=1.0, config=IsingConfig(temperature=2.5), periodic=per)
        Jb = g._get_bit_coupling()
        rowsum = g.J.sum(axis=1)
        for mode, hb in (("compat", g._get_bit_bias()), ("physical", 2 * g.h - 2 * rowsum)):
            key = f"p{int(per)}_{mode}"
            np.random.seed(42)
            bits = np.random.randint(0, 2, size=g.n_spins)
            arrs[key + "_init"] = bits.astype(np.int8)
            done = 0
            Ms, Es, states = [], [], []
            for cp in checkpoints:
                bits = g.sampler.gibbs_sweep(bits, Jb, hb, n_sweeps=cp - done)
                done = cp
                s = 2 * bits - 1
                Ms.append(g.magnetization(s[None, :]))
                Es.append(g.energy(s) / g.n_spins)
                states.append(s.astype(np.int8))
            arrs[key + "_M"] = np.array(Ms)
            arrs[key + "_E"] = np.array(Es)
            arrs[key + "_states"] = np.array(states)
            arrs[key + "_hbit"] = hb
            print(key, "M", Ms, "E/N", Es)
    save("g4_config1_trajectory", **arrs)


# ---------------------------------------------------------------- G5: Langevin
def g5():
    cfg = TSUConfig(temperature=0.7, dt=0.02, friction=1.5, n_burnin=3, n_steps=5)
    tsu_ = ThermalSamplingUnit(cfg)
    rng = np.random.default_rng(9)
    x = rng.normal(size=8)
    g = rng.normal(size=8)
    np.random.seed(77)
    x1 = tsu_._langevin_step(x, g)
    np.random.seed(77)
    noise = np.random.randn(8)

    def energy(v):
        return float((v ** 2).sum())

    x0 = np.array([0.3, -0.2, 0.1, 0.5])
    np.random.seed(5)
    samples, traj = tsu_.sample_from_energy(energy, x0, n_samples=3, return_trajectory=True)
    # numerical gradient of the quadratic at a few points (pins eps / central difference)
    pts = rng.normal(size=(3, 4))
    grads = np.array([tsu_._numerical_gradient(energy, p) for p in pts])
    # replay the draws: per sample (s>0: restart randn(d)), then (burnin + n_steps) x randn(d)
    np.random.seed(5)
    draws = []
    for s in range(3):
        if s > 0:
            draws.append(np.random.randn(4))
        for _ in range(cfg.n_burnin + cfg.n_steps):
            draws.append(np.random.randn(4))
    save("g5_langevin", T=cfg.temperature, dt=cfg.dt, friction=cfg.friction, n_burnin=cfg.n_burnin,
         n_steps=cfg.n_steps, x=x, grad=g, noise=noise, x_next=x1, x0=x0, samples=samples,
         trajectory=np.array(traj), draws=np.array(draws), grad_pts=pts, grads=grads,
         sample_count=tsu_.sample_count)


# ---------------------------------------------------------------- G6: observables
def g6():
    rng = np.random.default_rng(3)
    cfg = IsingConfig(temperature=1.9, external_field=0.25)
    grid = IsingGrid((4, 6), J=0.8, config=cfg, periodic=True)
    samples = rng.choice([-1, 1], size=(9, 24))
    chain = IsingChain(7, J=-1.2, config=IsingConfig(temperature=0.6, external_field=-0.4))
    csamples = rng.choice([-1, 1], size=(5, 7))
    gs = GibbsSampler()
    Jd = rng.normal(size=(6, 6))
    bd = rng.normal(size=6)
    bits = rng.integers(0, 2, size=(4, 6))
    save("g6_observables",
         grid_samples=samples,
         grid_M=grid.magnetization(samples), grid_chi=grid.susceptibility(samples),
         grid_C=grid.specific_heat(samples),
         grid_E=np.array([grid.energy(s) for s in samples]),
         grid_domains=np.array([grid.compute_domains(s) for s in samples]),
         chain_J=chain.J, chain_h=chain.h, chain_samples=csamples,
         chain_E=np.array([chain.energy(s) for s in csamples]),
         chain_M=chain.magnetization(csamples),
         dense_J=Jd, dense_b=bd, dense_bits=bits,
         dense_E=np.array([gs.compute_energy(b, Jd, bd) for b in bits]),
         dense_E_nobias=np.array([gs.compute_energy(b, Jd) for b in bits]),
         dense_field=np.array([[gs._compute_local_field(i, b, Jd, bd) for i in range(6)] for b in bits]),
         spins_to_bits=grid._spins_to_bits(samples[0]), bits_to_spins=grid._bits_to_spins(bits[0]))


# ---------------------------------------------------------------- G7: sigmoid table
def g7():
    gs = GibbsSampler()
    xs = np.concatenate([np.linspace(-25, 25, 201), [20.0, -20.0, 20.0000001, -20.0000001, 19.9999999,
                                                     -19.9999999, 0.0, 1e-300, 36.0, 745.0, -745.0]])
    ys = np.array([gs._sigmoid(float(x)) for x in xs])
    save("g7_sigmoid", x=xs, y=ys)


# ---------------------------------------------------------------- G8: SA / PT callers
def g8():
    rng = np.random.default_rng(17)
    n = 8
    J = rng.normal(size=(n, n))
    J = (J + J.T) / 2
    b = rng.normal(size=n) * 0.3
    arrs = {"J": J, "b": b}
    for sched in ("exponential", "linear"):
        s = GibbsSampler(GibbsConfig(temperature=1.0))
        np.random.seed(31)
        best, e = s.simulated_annealing(J, b, T_initial=5.0, T_final=0.1, n_steps=40, cooling_schedule=sched)
        arrs[f"sa_{sched}_state"] = best
        arrs[f"sa_{sched}_energy"] = e
        arrs[f"sa_{sched}_final_T"] = s.config.temperature
    s = GibbsSampler(GibbsConfig(temperature=1.0, n_burnin=2, n_sweeps=1))
    np.random.seed(32)
    smp, info = s.parallel_tempering(J, [0.5, 1.0, 2.0], bias=b, n_samples=12, swap_interval=3)
    arrs["pt_samples"] = smp
    arrs["pt_attempts"] = info["swap_attempts"]
    arrs["pt_accepts"] = info["swap_accepts"]
    arrs["pt_rate"] = info["swap_acceptance_rate"]
    arrs["pt_energies"] = np.array(info["energies"])
    arrs["pt_final_states"] = np.array(info["final_states"])
    m = IsingModel(6, config=IsingConfig(temperature=1.0))
    for i in range(5):
        m.set_coupling(i, i + 1, 1.0 if i % 2 == 0 else -1.0)
    np.random.seed(33)
    gstate, genergy = m.find_ground_state(n_steps=60)
    arrs["fgs_state"] = gstate
    arrs["fgs_energy"] = genergy
    arrs["fgs_J"] = m.J
    save("g8_callers", **arrs)


# ---------------------------------------------------------------- G9: distributional
def g9():
    arrs = {}
    for n, T in ((2, 1.0), (3, 1.5)):
        m = IsingModel(n, config=IsingConfig(temperature=T, external_field=0.2, n_burnin=50, n_sweeps=3))
        for i in range(n - 1):
            m.set_coupling(i, i + 1, 1.0)
        if n == 3:
            m.set_coupling(0, 2, -0.5)
        np.random.seed(100 + n)
        s = m.sample(n_samples=4000)
        code = ((s + 1) // 2).dot(1 << np.arange(n))
        arrs[f"n{n}_J"] = m.J
        arrs[f"n{n}_h"] = m.h
        arrs[f"n{n}_T"] = T
        arrs[f"n{n}_hist_compat"] = np.bincount(code, minlength=1 << n)
        # physical mode = the reference's own GibbsSampler fed the corrected bias
        hb = 2 * m.h - 2 * m.J.sum(axis=1)
        np.random.seed(200 + n)
        bits = m.sampler.sample_boltzmann(4 * m.J, bias=hb, n_samples=4000)
        arrs[f"n{n}_hist_physical"] = np.bincount(bits.dot(1 << np.arange(n)), minlength=1 << n)
    save("g9_distribution", **arrs)


# ---------------------------------------------------------------- G10: equilibrium observables (acceptance criterion)
T_C = 2.269185314213022


def _g10_chain(job):
    """One independent chain of the reference's own sampler (GibbsSampler.sample_boltzmann behind IsingGrid) on an
    L x L periodic lattice: per-sweep |m|, m, m^2 and (every `estride` sweeps) e = E/N through IsingGrid.energy."""
    L, T, mode, chain, burnin, n_sweeps, estride, nblocks = job
    g = IsingGrid((L, L), J=1.0, config=IsingConfig(temperature=T, n_burnin=0, n_sweeps=1), periodic=True)
    g.sampler.config.n_sweeps = 1
    Jb = g._get_bit_coupling()
    hb = g._get_bit_bias() if mode == "compat" else 2 * g.h - 2 * g.J.sum(axis=1)
    np.random.seed(100000 + 1000 * L + 100 * int(T * 10) + 10 * (mode == "compat") + chain)
    N = L * L
    state = np.random.randint(0, 2, size=N)
    state = g.sampler.sample_boltzmann(Jb, bias=hb, n_samples=1, burnin=burnin, initial_state=state)[-1]
    chunk = 1000
    m = np.empty(n_sweeps)
    e = []
    done = 0
    while done < n_sweeps:
        n = min(chunk, n_sweeps - done)
        smp = g.sampler.sample_boltzmann(Jb, bias=hb, n_samples=n, burnin=0, initial_state=state)
        state = smp[-1]
        spins = g._bits_to_spins(smp)
        m[done:done + n] = spins.sum(axis=1) / N
        for i in range(0, n, estride):
            e.append(g.energy(spins[i]) / N)
        done += n
    e = np.array(e)

    def blocks(x):
        nb = nblocks
        x = x[:len(x) // nb * nb]
        return x.reshape(nb, -1).mean(axis=1)
    return (L, T, mode, chain, blocks(np.abs(m)), blocks(m), blocks(m * m), blocks(m ** 4), blocks(e), blocks(e * e))


def g10():
    """<|m|>, <m>, <m^2>, <m^4>, <e>, <e^2> with blocked standard errors for IsingGrid 16^2 and 32^2 (periodic, J=1, h=0)
    at T = 2.0, T_c, 2.5 -- the reference's own sampler, shipped ("compat") bias and corrected ("physical") bias.
    Independent chains on the container's cores; block means over all chains give mean and standard error."""
    import multiprocessing as mp
    jobs = []
    nblocks = 20
    for L in (16, 32):
        for T in (2.0, T_C, 2.5):
            for mode in ("compat", "physical"):
                if mode == "compat":        # the shipped bias is a field of strength 2*J*deg: m ~ 1, decorrelates at once
                    chains, burnin, n = 2, 200, (4000 if L == 16 else 2000)
                elif abs(T - T_C) < 1e-6:
                    chains, burnin, n = 14, (2000 if L == 16 else 5000), (200000 if L == 16 else 80000)
                else:
                    chains, burnin, n = 7, (1000 if L == 16 else 2000), (60000 if L == 16 else 20000)
                estride = 1 if L == 16 else 8
                for c in range(chains):
                    jobs.append((L, T, mode, c, burnin, n, estride, nblocks))
    jobs.sort(key=lambda j: -(j[0] ** 2) * j[5])
    with mp.Pool(int(os.environ.get("G10_WORKERS", "7"))) as pool:
        results = pool.map(_g10_chain, jobs, chunksize=1)
    arrs = {"T_c": T_C}
    cases = {}
    for (L, T, mode, chain, am, m, m2, m4, e, e2) in results:
        cases.setdefault((L, T, mode), []).append((am, m, m2, m4, e, e2))
    for (L, T, mode), rs in sorted(cases.items()):
        key = f"L{L}_T{T:.4f}_{mode}"
        job = [j for j in jobs if j[:3] == (L, T, mode)][0]
        arrs[key + "_chains"] = len(rs)
        arrs[key + "_sweeps_per_chain"] = job[5]
        arrs[key + "_burnin"] = job[4]
        for idx, name in enumerate(("absm", "m", "m2", "m4", "e", "e2")):
            b = np.concatenate([r[idx] for r in rs])      # block means of all chains
            arrs[f"{key}_{name}_mean"] = b.mean()
            arrs[f"{key}_{name}_se"] = b.std(ddof=1) / np.sqrt(len(b))
            arrs[f"{key}_{name}_blocks"] = b
        print(key, "<|m|> = %.5f +- %.5f   <e> = %.5f +- %.5f" % (arrs[key + "_absm_mean"], arrs[key + "_absm_se"],
                                                                 arrs[key + "_e_mean"], arrs[key + "_e_se"]), flush=True)
    save("g10_equilibrium", **arrs)


# ---------------------------------------------------------------- G11: the reference's sampling benchmark (f4)
def g11():
    """tsu.benchmarks.sampling.SamplingBenchmark(seed=42).run_all_benchmarks(quick) of the reference, quick and full mode:
    the quality metrics of every trial (the timing fields are the reference's CPU and are stored for information only)."""
    import contextlib
    import io
    from tsu.benchmarks.sampling import SamplingBenchmark
    arrs = {}
    for quick in (True, False):
        with contextlib.redirect_stdout(io.StringIO()):
            res = SamplingBenchmark(seed=42).run_all_benchmarks(quick=quick)
        for name, r in res.items():
            key = f"{'quick' if quick else 'full'}_{name}"
            arrs[key + "_n_samples"] = r.n_samples
            arrs[key + "_n_trials"] = r.n_trials
            arrs[key + "_distribution"] = np.array(r.distribution_name)
            arrs[key + "_ks_statistics"] = np.array(r.ks_statistics)
            arrs[key + "_ks_pvalues"] = np.array(r.ks_pvalues)
            arrs[key + "_kl_divergences"] = np.array(r.kl_divergences)
            arrs[key + "_effective_sample_sizes"] = np.array(r.effective_sample_sizes)
            arrs[key + "_samples_per_second"] = np.array(r.samples_per_second)
            summ = r.summary()
            arrs[key + "_summary_keys"] = np.array(sorted(summ.keys()))
            print(key, "KL", np.mean(r.kl_divergences), "ESS", np.mean(r.effective_sample_sizes), "rate", np.mean(r.samples_per_second))
    save("g11_sampling_benchmark", **arrs)


def g12():
    """tsu.benchmarks.optimization.OptimizationBenchmark(seed=42).run_all_benchmarks(quick) of the reference, quick and full mode:
    every trial's objective and the greedy bound (the timing fields are the reference's CPU and are stored for information only),
    plus the inputs each case drew (graph / numbers) so that the host-side arithmetic can be checked without a GPU."""
    import contextlib
    import io
    from tsu.benchmarks.optimization import OptimizationBenchmark
    arrs = {}
    for quick in (True, False):
        with contextlib.redirect_stdout(io.StringIO()):
            res = OptimizationBenchmark(seed=42).run_all_benchmarks(quick=quick)
        for name, r in res.items():
            key = f"{'quick' if quick else 'full'}_{name}"
            arrs[key + "_problem"] = np.array(r.problem_name)
            arrs[key + "_size"] = r.problem_size
            arrs[key + "_n_trials"] = r.n_trials
            arrs[key + "_best_objectives"] = np.array(r.best_objectives)
            arrs[key + "_final_objectives"] = np.array(r.final_objectives)
            arrs[key + "_optimal_objective"] = np.array(np.nan if r.optimal_objective is None else r.optimal_objective)
            arrs[key + "_n_iterations"] = np.array(r.n_iterations)
            arrs[key + "_solution_times"] = np.array(r.solution_times)
            summ = r.summary()
            arrs[key + "_summary_keys"] = np.array(sorted(summ.keys()))
            arrs[key + "_gap_mean"] = np.array(summ.get("optimality_gap_percent", {}).get("mean", np.nan))
            print(key, "best", r.best_objectives, "optimal", r.optimal_objective, "ms", np.mean(r.solution_times) * 1e3)
    # the inputs of the two graph cases and of the partition case, as the seeded draws give them (quick and full sizes)
    for n, dens, tag in ((15, 0.5, "quick_maxcut"), (20, 0.5, "full_maxcut"), (10, 0.4, "quick_coloring"), (15, 0.4, "full_coloring")):
        np.random.seed(42)
        a = (np.random.rand(n, n) < dens).astype(float)
        a = (a + a.T) / 2
        np.fill_diagonal(a, 0)
        arrs[tag + "_adjacency"] = a
    for n, tag in ((15, "quick_partition"), (20, "full_partition")):
        np.random.seed(42)
        arrs[tag + "_numbers"] = np.random.randint(1, 100, size=n)
    save("g12_optimization_benchmark", **arrs)


if __name__ == "__main__":
    only = set(sys.argv[1:])
    for fn in (g1_g2, g3, g4, g5, g6, g7, g8, g9, g10, g11, g12):
        if not only or fn.__name__ in only:
            fn()
