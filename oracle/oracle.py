"""CPU ORACLE -- test infrastructure, NOT product code.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The shipped package (tsu-emulator_amd/) never does; its ops fail loudly without the HIP library.

Contents
  * NumPy restatements of the reference algorithms, in the reference's own visiting order, with
    the random draws passed in (so NumPy's legacy MT19937 stream can be replayed bit for bit):
      ref_sigmoid, ref_gibbs_sweep, ref_sample_boltzmann, ref_compute_energy,
      ref_langevin_step, ref_numerical_gradient, ref_sample_from_energy,
      ref_grid_coupling, ref_bit_coupling, ref_bit_bias
    Each cites the reference lines it follows; all are pinned by tests/golden/g*.npz, which were
    produced by running the unmodified reference (tests/golden/make_golden.py).
  * ctypes bindings of oracle/tsu_oracle.c: the same restatements in C plus the "device-order"
    twins (checkerboard + Philox) that define the bit-exact contract for the HIP kernels.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "tsu_oracle.c")
_LIB = os.path.join(_HERE, "_build", "libtsu_oracle.so")

MODE_PHYSICAL = 0
MODE_COMPAT = 1


def build(force=False):
    """Compile oracle/tsu_oracle.c with gcc into oracle/_build/ (idempotent)."""
    if not force and os.path.exists(_LIB) and os.path.getmtime(_LIB) >= os.path.getmtime(_SRC):
        return _LIB
    os.makedirs(os.path.dirname(_LIB), exist_ok=True)
    tmp = _LIB + ".tmp%d" % os.getpid()
    subprocess.check_call(["gcc", "-O2", "-fno-fast-math", "-ffp-contract=off", "-shared", "-fPIC",
                           "-o", tmp, _SRC, "-lm"])
    os.replace(tmp, _LIB)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.ora_sigmoid.restype = C.c_double
        _lib.ora_sigmoid.argtypes = [C.c_double]
        _lib.ora_dense_energy.restype = C.c_double
        _lib.ora_dense_uniform.restype = C.c_double
        _lib.ora_sparse_energy.restype = C.c_double
        _lib.ora_dense_uniform.argtypes = [C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32]
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


# ============================================================================ reference order (NumPy)

def ref_sigmoid(x):
    """tsu/gibbs.py:61-77 -- hard clamp beyond +-20, else 1/(1+exp(-x))."""
    if x > 20:
        return 1.0
    elif x < -20:
        return 0.0
    return 1.0 / (1.0 + np.exp(-x))


def ref_gibbs_sweep(state, coupling, bias, T, uniforms, order=None):
    """tsu/gibbs.py:128-162 with :97-100 and :124-126 inlined.

    uniforms: (n_sweeps, n) doubles consumed in visiting order; order: None or (n_sweeps, n).
    Returns a new state array of the input dtype (the input is not modified, gibbs.py:150).
    """
    state = state.copy()
    n = len(state)
    uniforms = np.asarray(uniforms, dtype=np.float64).reshape(-1, n)
    for s in range(uniforms.shape[0]):
        idx = range(n) if order is None else order[s]
        for k, i in enumerate(idx):
            h = np.dot(coupling[i, :], state)
            if bias is not None:
                h += bias[i]
            prob = ref_sigmoid(float(h) / T)
            state[i] = 1 if uniforms[s, k] < prob else 0
    return state


def ref_sample_boltzmann(coupling, bias, T, burnin, n_sweeps, n_samples, init, uniforms, order=None):
    """tsu/gibbs.py:164-213 given the initial state and the replayed draws.

    uniforms / order: (burnin + n_sweeps*n_samples, n).  Returns (n_samples, n) int64.
    """
    n = coupling.shape[0]
    if coupling.shape != (n, n):
        raise ValueError("Coupling matrix must be square")
    state = np.asarray(init).copy()
    o = None if order is None else order[:burnin]
    state = ref_gibbs_sweep(state, coupling, bias, T, uniforms[:burnin].reshape(-1, n), o) if burnin else state
    samples = np.zeros((n_samples, n), dtype=int)
    pos = burnin
    for k in range(n_samples):
        o = None if order is None else order[pos:pos + n_sweeps]
        state = ref_gibbs_sweep(state, coupling, bias, T, uniforms[pos:pos + n_sweeps], o)
        pos += n_sweeps
        samples[k] = state
    return samples


def ref_compute_energy(state, coupling, bias=None):
    """tsu/gibbs.py:215-236."""
    e = -0.5 * state.dot(coupling).dot(state)
    if bias is not None:
        e -= bias.dot(state)
    return float(e)


def ref_grid_coupling(rows, cols, J, periodic):
    """tsu/models/ising.py:343-361 -- dense coupling matrix of the square lattice.

    Bonds are SET (not added): on a periodic dimension of size 2 the wrap bond coincides with the
    direct bond; on a periodic dimension of size 1 the wrap bond is a self-coupling J_ii.
    """
    n = rows * cols
    M = np.zeros((n, n))
    for i in range(rows):
        for j in range(cols):
            idx = i * cols + j
            if j < cols - 1:
                M[idx, idx + 1] = M[idx + 1, idx] = J
            elif periodic:
                M[idx, i * cols] = M[i * cols, idx] = J
            if i < rows - 1:
                M[idx, idx + cols] = M[idx + cols, idx] = J
            elif periodic:
                M[idx, j] = M[j, idx] = J
    return M


def ref_bit_coupling(J):
    """tsu/models/ising.py:138."""
    return 4 * J


def ref_bit_bias(J, h, mode=MODE_COMPAT):
    """tsu/models/ising.py:148 (compat, as shipped) or the corrected conversion (physical)."""
    if mode == MODE_COMPAT:
        return -2 * h + 2 * np.sum(J, axis=1)
    return 2 * h - 2 * np.sum(J, axis=1)


def ref_langevin_step(x, grad, noise, T, dt, friction):
    """tsu/core.py:64-80 with np.random.randn(*x.shape) replaced by `noise`."""
    drift = -grad * dt / friction
    noise_scale = np.sqrt(2 * T * dt / friction)
    diffusion = noise_scale * noise
    return x + drift + diffusion


def ref_numerical_gradient(energy_fn, x, eps=1e-5):
    """tsu/core.py:82-98."""
    x = np.atleast_1d(x)
    grad = np.zeros_like(x)
    for i in range(len(x)):
        xp = x.copy()
        xp[i] += eps
        xm = x.copy()
        xm[i] -= eps
        grad[i] = (float(energy_fn(xp)) - float(energy_fn(xm))) / (2 * eps)
    return grad


def ref_sample_from_energy(energy_fn, x_init, n_samples, T, dt, friction, n_burnin, n_steps, draws):
    """tsu/core.py:135-162 with every np.random.randn draw taken from `draws` in order.

    Returns (samples (n_samples, d), trajectory list of the sampling-phase states).
    """
    draws = iter(np.asarray(draws))
    x = np.atleast_1d(x_init).copy()
    samples, traj = [], []
    for s in range(n_samples):
        if s > 0:
            x = x_init + 0.1 * next(draws)
        for _ in range(n_burnin):
            x = ref_langevin_step(x, ref_numerical_gradient(energy_fn, x), next(draws), T, dt, friction)
        for _ in range(n_steps):
            x = ref_langevin_step(x, ref_numerical_gradient(energy_fn, x), next(draws), T, dt, friction)
            traj.append(x.copy())
        samples.append(x.copy())
    return np.array(samples), traj


# ============================================================================ C restatements (ctypes)

def philox4x32_10(ctr, key):
    ctr = np.ascontiguousarray(ctr, dtype=np.uint32)
    key = np.ascontiguousarray(key, dtype=np.uint32)
    out = np.zeros(4, dtype=np.uint32)
    lib().ora_philox4x32_10(_p(ctr, C.c_uint32), _p(key, C.c_uint32), _p(out, C.c_uint32))
    return out


def c_sigmoid(x):
    return lib().ora_sigmoid(float(x))


def c_dense_sweep_replay(state, J, bias, T, uniforms, order=None):
    st = np.ascontiguousarray(state, dtype=np.int64).copy()
    n = st.size
    Jc = np.ascontiguousarray(J, dtype=np.float64)
    u = np.ascontiguousarray(uniforms, dtype=np.float64).reshape(-1, n)
    b = None if bias is None else np.ascontiguousarray(bias, dtype=np.float64)
    o = None if order is None else np.ascontiguousarray(order, dtype=np.int64).reshape(-1, n)
    lib().ora_dense_sweep_replay(_p(st, C.c_int64), _p(Jc, C.c_double), None if b is None else _p(b, C.c_double),
                                 C.c_int(n), C.c_double(T), C.c_int(u.shape[0]),
                                 None if o is None else _p(o, C.c_int64), _p(u, C.c_double))
    return st


def c_dense_energy(state, J, bias=None):
    st = np.ascontiguousarray(state, dtype=np.int64)
    Jc = np.ascontiguousarray(J, dtype=np.float64)
    b = None if bias is None else np.ascontiguousarray(bias, dtype=np.float64)
    return lib().ora_dense_energy(_p(st, C.c_int64), _p(Jc, C.c_double), None if b is None else _p(b, C.c_double),
                                  C.c_int(st.size))


def c_langevin_step_f64(x, grad, noise, T, dt, friction):
    xx = np.ascontiguousarray(x, dtype=np.float64).copy()
    g = np.ascontiguousarray(grad, dtype=np.float64)
    nz = np.ascontiguousarray(noise, dtype=np.float64)
    lib().ora_langevin_step_f64(_p(xx, C.c_double), _p(g, C.c_double), _p(nz, C.c_double), C.c_int(xx.size),
                                C.c_double(T), C.c_double(dt), C.c_double(friction))
    return xx


# ---------------------------------------------------------------------------- device-order twins

def ising2d_thresholds(J, h, T, mode=MODE_PHYSICAL):
    """table[deg*5+up] (uint64, 0..2^32): see ora_ising2d_thresholds."""
    t = np.zeros(25, dtype=np.uint64)
    lib().ora_ising2d_thresholds(C.c_double(J), C.c_double(h), C.c_double(T), C.c_int(mode), _p(t, C.c_uint64))
    return t


def ising2d_thresholds_numpy(J, h, T, mode=MODE_PHYSICAL):
    """The same table from the NumPy restatement of _sigmoid (cross-check of the C helper)."""
    t = np.zeros(25, dtype=np.uint64)
    for deg in range(5):
        for up in range(deg + 1):
            bias = (-2.0 * h + 2.0 * J * deg) if mode == MODE_COMPAT else (2.0 * h - 2.0 * J * deg)
            p = ref_sigmoid((4.0 * J * up + bias) / T)
            t[deg * 5 + up] = int(np.floor(p * 4294967296.0 + 0.5))
    return t


def ising2d_randomize(rows, cols, seed, replica=0, row0=0):
    s = np.zeros((rows, cols), dtype=np.int8)
    lib().ora_ising2d_randomize(_p(s, C.c_int8), C.c_int(rows), C.c_int(cols), C.c_int64(row0),
                                C.c_uint64(seed), C.c_uint32(replica))
    return s


def ising2d_sweep(spins, periodic, table, n_sweeps, seed, sweep0=0, replica=0, plain=False):
    """n_sweeps checkerboard heat-bath sweeps; returns a new (rows, cols) int8 array."""
    s = np.ascontiguousarray(spins, dtype=np.int8).copy()
    assert s.ndim == 2
    t = np.ascontiguousarray(table, dtype=np.uint64)
    assert t.size == 25
    fn = lib().ora_ising2d_sweep_plain if plain else lib().ora_ising2d_sweep
    fn(_p(s, C.c_int8), C.c_int(s.shape[0]), C.c_int(s.shape[1]), C.c_int(int(bool(periodic))), _p(t, C.c_uint64),
       C.c_int(n_sweeps), C.c_uint64(seed), C.c_uint32(sweep0), C.c_uint32(replica))
    return s


def ising2d_observables(spins, periodic):
    s = np.ascontiguousarray(spins, dtype=np.int8)
    a, b = C.c_int64(0), C.c_int64(0)
    lib().ora_ising2d_observables(_p(s, C.c_int8), C.c_int(s.shape[0]), C.c_int(s.shape[1]),
                                  C.c_int(int(bool(periodic))), C.byref(a), C.byref(b))
    return a.value, b.value


def dense_uniform(i, t, seed, replica=0):
    return lib().ora_dense_uniform(int(i), int(t), int(seed), int(replica))


def dense_sweep_philox(state, J, bias, T, n_sweeps, seed, sweep0=0, replica=0, order=None):
    st = np.ascontiguousarray(state, dtype=np.int8).copy()
    n = st.size
    Jc = np.ascontiguousarray(J, dtype=np.float64)
    b = None if bias is None else np.ascontiguousarray(bias, dtype=np.float64)
    o = None if order is None else np.ascontiguousarray(order, dtype=np.int64).reshape(n_sweeps, n)
    lib().ora_dense_sweep_philox(_p(st, C.c_int8), _p(Jc, C.c_double), None if b is None else _p(b, C.c_double),
                                 C.c_int(n), C.c_double(T), C.c_int(n_sweeps),
                                 None if o is None else _p(o, C.c_int64), C.c_uint64(seed), C.c_uint32(sweep0),
                                 C.c_uint32(replica))
    return st


def sparse_sweep_philox(state, row_ptr, col, val, bias, T, n_sweeps, seed, sweep0=0, replica=0, order=None):
    """K5 twin: sequential heat-bath sweeps on a CSR graph in the visiting order ``order`` (n sites, the same every sweep).
    Reference loop: tsu/gibbs.py:128-162; field incl. the diagonal entry: gibbs.py:97."""
    st = np.ascontiguousarray(state, dtype=np.int8).copy()
    n = st.size
    rp = np.ascontiguousarray(row_ptr, dtype=np.int64)
    ci = np.ascontiguousarray(col, dtype=np.int32)
    va = np.ascontiguousarray(val, dtype=np.float64)
    b = None if bias is None else np.ascontiguousarray(bias, dtype=np.float64)
    o = None if order is None else np.ascontiguousarray(order, dtype=np.int32)
    lib().ora_sparse_sweep_philox(_p(st, C.c_int8), _p(rp, C.c_int64), _p(ci, C.c_int32), _p(va, C.c_double),
                                  None if b is None else _p(b, C.c_double), C.c_int(n), C.c_double(T), C.c_int(n_sweeps),
                                  None if o is None else _p(o, C.c_int32), C.c_uint64(seed), C.c_uint32(sweep0), C.c_uint32(replica))
    return st


def sparse_energy(state, row_ptr, col, val, bias):
    st = np.ascontiguousarray(state, dtype=np.int8)
    rp = np.ascontiguousarray(row_ptr, dtype=np.int64)
    ci = np.ascontiguousarray(col, dtype=np.int32)
    va = np.ascontiguousarray(val, dtype=np.float64)
    b = None if bias is None else np.ascontiguousarray(bias, dtype=np.float64)
    return float(lib().ora_sparse_energy(_p(st, C.c_int8), _p(rp, C.c_int64), _p(ci, C.c_int32), _p(va, C.c_double),
                                         None if b is None else _p(b, C.c_double), C.c_int(st.size)))


TAG_LANGEVIN = 3
TAG_LANGEVIN_RESTART = 5


def langevin_normals_f32(q, chain, step, seed, tag=TAG_LANGEVIN):
    out = np.zeros(4, dtype=np.float32)
    lib().ora_langevin_normals_f32(C.c_uint32(q), C.c_uint32(chain), C.c_uint32(step), C.c_uint32(tag),
                                   C.c_uint64(seed), _p(out, C.c_float))
    return out


def langevin_quadratic_f32(x, k, mu, n_steps, dt, gamma, T, seed, step0=0, chain0=0, trajectory=False):
    """x: (n_chains, dim) float32.  Returns x_final or (x_final, traj (n_steps, n_chains, dim))."""
    xx = np.ascontiguousarray(x, dtype=np.float32).copy()
    if xx.ndim == 1:
        xx = xx[None, :]
    n_chains, dim = xx.shape
    kk = np.ascontiguousarray(np.broadcast_to(np.asarray(k, dtype=np.float32), (dim,)))
    mm = np.ascontiguousarray(np.broadcast_to(np.asarray(mu, dtype=np.float32), (dim,)))
    traj = np.zeros((n_steps, n_chains, dim), dtype=np.float32) if trajectory else None
    lib().ora_langevin_quadratic_f32(_p(xx, C.c_float), _p(kk, C.c_float), _p(mm, C.c_float), C.c_int(n_chains),
                                     C.c_int(dim), C.c_int(n_steps), C.c_float(dt), C.c_float(gamma), C.c_float(T),
                                     C.c_uint64(seed), C.c_uint32(step0), C.c_uint32(chain0),
                                     None if traj is None else _p(traj, C.c_float))
    return (xx, traj) if trajectory else xx


def langevin_coupled_f32(x, A, b, n_steps, dt, gamma, T, seed, step0=0, chain0=0, trajectory=False):
    """Langevin steps on E = 1/2 x^T A x + b^T x (A symmetric (dim, dim)); x: (n_chains, dim) float32.  Returns x_final or
    (x_final, traj (n_steps, n_chains, dim))."""
    xx = np.ascontiguousarray(x, dtype=np.float32).copy()
    if xx.ndim == 1:
        xx = xx[None, :]
    n_chains, dim = xx.shape
    aa = np.ascontiguousarray(A, dtype=np.float32).reshape(dim, dim)
    bb = None if b is None else np.ascontiguousarray(np.broadcast_to(np.asarray(b, dtype=np.float32), (dim,)))
    traj = np.zeros((n_steps, n_chains, dim), dtype=np.float32) if trajectory else None
    lib().ora_langevin_coupled_f32(_p(xx, C.c_float), _p(aa, C.c_float), None if bb is None else _p(bb, C.c_float), C.c_int(n_chains),
                                   C.c_int(dim), C.c_int(n_steps), C.c_float(dt), C.c_float(gamma), C.c_float(T), C.c_uint64(seed),
                                   C.c_uint32(step0), C.c_uint32(chain0), None if traj is None else _p(traj, C.c_float))
    return (xx, traj) if trajectory else xx


def ising2d_site_uniforms(rows, cols, hs, seed, replica=0):
    """(rows, cols) uint32: the uniform each site uses in half-sweep hs = 2*sweep + colour."""
    out = np.zeros((rows, cols), dtype=np.uint32)
    lib().ora_ising2d_site_uniforms(_p(out, C.c_uint32), C.c_int(rows), C.c_int(cols), C.c_uint32(hs),
                                    C.c_uint64(seed), C.c_uint32(replica))
    return out


def ising2d_sweep_window(block, row_global0, total_rows, periodic, table, n_sweeps, seed, sweep0=0, replica=0):
    """Sweeps on a window of rows of a larger lattice (see ora_ising2d_sweep_window); returns a new array."""
    s = np.ascontiguousarray(block, dtype=np.int8).copy()
    t = np.ascontiguousarray(table, dtype=np.uint64)
    lib().ora_ising2d_sweep_window(_p(s, C.c_int8), C.c_int(s.shape[0]), C.c_int(s.shape[1]), C.c_int64(row_global0),
                                   C.c_int64(total_rows), C.c_int(int(bool(periodic))), _p(t, C.c_uint64), C.c_int(n_sweeps),
                                   C.c_uint64(seed), C.c_uint32(sweep0), C.c_uint32(replica))
    return s
