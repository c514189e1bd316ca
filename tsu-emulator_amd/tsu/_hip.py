"""ctypes binding of libtsu_hip.so (include/tsu_hip.h) -- the only door to the GPU.

There is deliberately no CPU fallback in this package: if the HIP library is missing or no GPU is
present, every operation that needs it raises :class:`HipUnavailableError`.
"""
import atexit
import ctypes as C
import os
import sys
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TSU_HIP_LIB") or os.path.join(_HERE, "_lib", "libtsu_hip.so")  # (TSU_HIP_LIB: development builds)

TSU_OK = 0
TSU_E_INVALID, TSU_E_NOMEM, TSU_E_HIP, TSU_E_RCCL, TSU_E_UNSUPPORTED = -1, -2, -3, -4, -5
MODE_PHYSICAL, MODE_COMPAT = 0, 1
DTYPE_F64, DTYPE_F32 = 0, 1
KERNEL_AUTO, KERNEL_GENERIC, KERNEL_TILED, KERNEL_SMALL = 0, 1, 2, 3
PART_ALL, PART_INTERIOR, PART_BOUNDARY = 0, 1, 2


class HipUnavailableError(RuntimeError):
    """libtsu_hip.so is not built / not loadable, or there is no HIP device."""


class HipError(RuntimeError):
    """A call into libtsu_hip.so failed (TSU_E_HIP / TSU_E_NOMEM / TSU_E_RCCL)."""


class UnsupportedError(HipError):
    """TSU_E_UNSUPPORTED: valid request the HIP build has no kernel for."""


_u8p, _i8p = C.POINTER(C.c_uint8), C.POINTER(C.c_int8)
_u32p, _u64p, _i64p = C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_int64)
_i32p = C.POINTER(C.c_int32)
_f32p, _f64p = C.POINTER(C.c_float), C.POINTER(C.c_double)
_vp = C.c_void_p

# name -> (restype, argtypes): mirrors include/tsu_hip.h one to one (tests check every symbol exists)
SIGNATURES = {
    "tsu_version": (C.c_int, []),
    "tsu_init": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "tsu_shutdown": (C.c_int, [_vp]),
    "tsu_last_error": (C.c_char_p, [_vp]),
    "tsu_set_stream": (C.c_int, [_vp, _vp]),
    "tsu_synchronize": (C.c_int, [_vp]),
    "tsu_device_info": (C.c_int, [_vp, C.c_char_p, C.c_int, C.POINTER(C.c_int), _u64p]),
    "tsu_timer_begin": (C.c_int, [_vp]),
    "tsu_timer_end": (C.c_int, [_vp, _f32p]),
    "tsu_philox4x32_10": (C.c_int, [_vp, C.c_int, _u32p, _u32p, _u32p]),
    "tsu_ising2d_thresholds": (C.c_int, [C.c_double, C.c_double, C.c_double, C.c_int, _u64p]),
    "tsu_ising2d_create": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]),
    "tsu_ising2d_create_slab": (C.c_int, [_vp, C.c_int64, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int,
                                          C.POINTER(_vp)]),
    "tsu_ising2d_destroy": (C.c_int, [_vp]),
    "tsu_ising2d_set_spins": (C.c_int, [_vp, _i8p, C.c_int, C.c_int]),
    "tsu_ising2d_get_spins": (C.c_int, [_vp, _i8p, C.c_int, C.c_int]),
    "tsu_ising2d_randomize": (C.c_int, [_vp, C.c_uint64, C.c_uint32]),
    "tsu_ising2d_fill": (C.c_int, [_vp, C.c_int8]),
    "tsu_ising2d_set_thresholds": (C.c_int, [_vp, _u64p]),
    "tsu_ising2d_set_model": (C.c_int, [_vp, C.c_double, C.c_double, C.c_double, C.c_int]),
    "tsu_ising2d_set_kernel": (C.c_int, [_vp, C.c_int, C.c_int]),
    "tsu_ising2d_sweep": (C.c_int, [_vp, C.c_int, C.c_uint64, C.c_uint32, C.c_uint32]),
    "tsu_ising2d_sweep_part": (C.c_int, [_vp, C.c_int, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int]),
    "tsu_ising2d_observables": (C.c_int, [_vp, _i64p, _i64p]),
    "tsu_ising2d_sample": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint32, C.c_uint32, _i8p]),
    "tsu_ising2d_sweep_batch": (C.c_int, [C.POINTER(_vp), C.c_int, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32),
                                          C.POINTER(C.c_uint32)]),
    "tsu_ising2d_observables_batch": (C.c_int, [C.POINTER(_vp), C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "tsu_ising2d_row_ptr": (C.c_int, [_vp, C.c_int, C.POINTER(_vp), C.POINTER(C.c_size_t)]),
    "tsu_ising2d_set_timing": (C.c_int, [_vp, C.c_int]),
    "tsu_ising2d_last_sweep_ms": (C.c_int, [_vp, _f32p]),
    "tsu_ising2d_launch_count": (C.c_int, [_vp, _u64p]),
    "tsu_comm_unique_id": (C.c_int, [_u8p]),
    "tsu_comm_create": (C.c_int, [_vp, C.c_int, C.c_int, _u8p, C.POINTER(_vp)]),
    "tsu_comm_destroy": (C.c_int, [_vp]),
    "tsu_ising2d_halo_exchange": (C.c_int, [_vp, _vp]),
    "tsu_comm_wait": (C.c_int, [_vp, C.c_double, _u64p]),
    "tsu_comm_allreduce_i64": (C.c_int, [_vp, _i64p, C.c_int]),
    "tsu_dense_create": (C.c_int, [_vp, C.c_int, _vp, C.c_int, _f64p, C.POINTER(_vp)]),
    "tsu_dense_destroy": (C.c_int, [_vp]),
    "tsu_dense_set_state": (C.c_int, [_vp, _i8p]),
    "tsu_dense_get_state": (C.c_int, [_vp, _i8p]),
    "tsu_dense_sweep": (C.c_int, [_vp, C.c_double, C.c_int, _i64p, C.c_uint64, C.c_uint32, C.c_uint32, _f64p]),
    "tsu_dense_sample": (C.c_int, [_vp, C.c_double, C.c_int, C.c_int, C.c_int, _i64p, C.c_uint64, C.c_uint32, C.c_uint32, _f64p, _i8p]),
    "tsu_dense_anneal": (C.c_int, [_vp, _f64p, C.c_int, _i64p, C.c_uint64, C.c_uint32, C.c_uint32, _f64p, _i8p]),
    "tsu_dense_sweep_replicas": (C.c_int, [_vp, C.c_int, _f64p, C.c_int, _i8p, _u64p, _u32p, _u32p, _f64p]),
    "tsu_dense_energy": (C.c_int, [_vp, _f64p]),
    "tsu_dense_energies": (C.c_int, [_vp, _i8p, C.c_int, _f64p]),
    "tsu_dense_launch_counts": (C.c_int, [_vp, _u64p]),
    "tsu_sparse_create": (C.c_int, [_vp, C.c_int, _i64p, _i32p, _f64p, _f64p, C.c_int, _i32p, _i32p, C.POINTER(_vp)]),
    "tsu_sparse_destroy": (C.c_int, [_vp]),
    "tsu_sparse_set_state": (C.c_int, [_vp, _i8p]),
    "tsu_sparse_get_state": (C.c_int, [_vp, _i8p]),
    "tsu_sparse_sweep": (C.c_int, [_vp, C.c_double, C.c_int, C.c_uint64, C.c_uint32, C.c_uint32]),
    "tsu_sparse_sample": (C.c_int, [_vp, C.c_double, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint32, C.c_uint32, _i8p]),
    "tsu_sparse_energy": (C.c_int, [_vp, _f64p, _i64p]),
    "tsu_langevin_create": (C.c_int, [_vp, C.c_int, C.c_int, C.POINTER(_vp)]),
    "tsu_langevin_destroy": (C.c_int, [_vp]),
    "tsu_langevin_set_state": (C.c_int, [_vp, _f32p]),
    "tsu_langevin_get_state": (C.c_int, [_vp, _f32p]),
    "tsu_langevin_set_energy": (C.c_int, [_vp, _f32p, _f32p]),
    "tsu_langevin_set_coupling": (C.c_int, [_vp, _f32p, _f32p]),
    "tsu_langevin_restart": (C.c_int, [_vp, _f32p, C.c_float, C.c_uint64, C.c_uint32]),
    "tsu_langevin_step": (C.c_int, [_vp, C.c_int, C.c_float, C.c_float, C.c_float, C.c_uint64, C.c_uint32,
                                    C.c_uint32, _f32p]),
    "tsu_langevin_set_kernel": (C.c_int, [_vp, C.c_int]),
}

_lib = None


def load_library():
    """dlopen libtsu_hip.so and declare every prototype.  Raises HipUnavailableError if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipUnavailableError(
            f"{LIB_PATH} not found: build it with tsu-emulator_amd/csrc/build.sh (hipcc, gfx950). "
            "This package has no CPU fallback.")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # missing ROCm runtime etc.
        raise HipUnavailableError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _ptr(a, t):
    return a.ctypes.data_as(t)


_live_contexts = weakref.WeakSet()


@atexit.register
def _shutdown_contexts():
    """Deterministic teardown at interpreter exit (before the HIP runtime's own exit handlers): every context still
    alive is synchronised and its streams / events are destroyed (``__del__`` is not reliable at finalisation)."""
    for ctx in list(_live_contexts):
        try:
            if getattr(ctx, "h", None):
                ctx.lib.tsu_shutdown(ctx.h)
                ctx.h = None
        except Exception:
            pass


class Context:
    """One tsu_ctx (one GPU).  ``Context.default()`` is the per-process singleton used by the API layer."""

    _default = None

    def __init__(self, device=-1):
        self.lib = load_library()
        h = _vp()
        rc = self.lib.tsu_init(int(device), C.byref(h))
        if rc != TSU_OK:
            msg = self.lib.tsu_last_error(None).decode()
            raise HipUnavailableError(f"tsu_init failed ({rc}): {msg}")
        self.h = h
        _live_contexts.add(self)

    @classmethod
    def default(cls):
        if cls._default is None:
            dev = int(os.environ.get("TSU_HIP_DEVICE", os.environ.get("LOCAL_RANK", "-1")))
            cls._default = cls(dev)
        return cls._default

    def check(self, rc):
        if rc == TSU_OK:
            return
        msg = self.lib.tsu_last_error(self.h).decode()
        if rc == TSU_E_INVALID:
            raise ValueError(msg)
        if rc == TSU_E_UNSUPPORTED:
            raise UnsupportedError(msg)
        if rc == TSU_E_NOMEM:
            raise MemoryError(msg)
        if rc == TSU_E_RCCL:
            raise HipError(f"RCCL: {msg}")
        raise HipError(f"libtsu_hip error {rc}: {msg}")

    def set_stream(self, stream_ptr):
        self.check(self.lib.tsu_set_stream(self.h, _vp(stream_ptr or 0)))

    def synchronize(self):
        self.check(self.lib.tsu_synchronize(self.h))

    def device_info(self):
        name = C.create_string_buffer(256)
        cus, mem = C.c_int(0), C.c_uint64(0)
        self.check(self.lib.tsu_device_info(self.h, name, 256, C.byref(cus), C.byref(mem)))
        return {"name": name.value.decode(), "compute_units": cus.value, "hbm_bytes": mem.value}

    def timer_begin(self):
        self.check(self.lib.tsu_timer_begin(self.h))

    def timer_end(self):
        ms = C.c_float(0)
        self.check(self.lib.tsu_timer_end(self.h, C.byref(ms)))
        return ms.value

    def philox4x32_10(self, ctrs, key):
        ctrs = np.ascontiguousarray(ctrs, dtype=np.uint32).reshape(-1, 4)
        key = np.ascontiguousarray(key, dtype=np.uint32).reshape(2)
        out = np.zeros_like(ctrs)
        self.check(self.lib.tsu_philox4x32_10(self.h, ctrs.shape[0], _ptr(ctrs, _u32p), _ptr(key, _u32p),
                                              _ptr(out, _u32p)))
        return out

    def __del__(self, _finalizing=sys.is_finalizing):
        try:
            if getattr(self, "h", None) and not _finalizing():
                self.lib.tsu_shutdown(self.h)
                self.h = None
        except Exception:
            pass


def ising2d_thresholds(J, h, T, mode=MODE_PHYSICAL):
    """Host helper of the library (no GPU needed): table[deg*5+up] as uint64."""
    lib = load_library()
    t = np.zeros(25, dtype=np.uint64)
    rc = lib.tsu_ising2d_thresholds(float(J), float(h), float(T), int(mode), _ptr(t, _u64p))
    if rc != TSU_OK:
        raise ValueError("Temperature must be positive" if not T > 0 else "invalid threshold arguments")
    return t


class Lattice:
    """tsu_ising2d handle: a rows x cols lattice (or a row slab of one) of +-1 int8 spins on the GPU."""

    def __init__(self, rows, cols, periodic=False, ctx=None, total_rows=None, row0=0, ghost=0):
        self.ctx = ctx or Context.default()
        self.lib = self.ctx.lib
        self.rows, self.cols, self.periodic = int(rows), int(cols), bool(periodic)
        self.total_rows = int(total_rows if total_rows is not None else rows)
        self.row0, self.ghost = int(row0), int(ghost)
        h = _vp()
        self.ctx.check(self.lib.tsu_ising2d_create_slab(self.ctx.h, self.total_rows, self.cols, int(self.periodic),
                                                        self.row0, self.rows, self.ghost, C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.tsu_ising2d_destroy(self.h)
            self.h = None

    def __del__(self, _finalizing=sys.is_finalizing):
        if not _finalizing():  # at interpreter exit the process teardown frees the device memory
            self.close()

    def set_spins(self, spins, row_first=0):
        s = np.ascontiguousarray(spins, dtype=np.int8).reshape(-1, self.cols)
        self.ctx.check(self.lib.tsu_ising2d_set_spins(self.h, _ptr(s, _i8p), int(row_first), s.shape[0]))

    def get_spins(self, row_first=0, n_rows=None):
        n_rows = self.rows if n_rows is None else int(n_rows)
        out = np.empty((n_rows, self.cols), dtype=np.int8)
        self.ctx.check(self.lib.tsu_ising2d_get_spins(self.h, _ptr(out, _i8p), int(row_first), n_rows))
        return out

    def randomize(self, seed, replica=0):
        self.ctx.check(self.lib.tsu_ising2d_randomize(self.h, int(seed), int(replica)))

    def fill(self, value):
        self.ctx.check(self.lib.tsu_ising2d_fill(self.h, int(value)))

    def set_thresholds(self, table):
        t = np.ascontiguousarray(table, dtype=np.uint64)
        if t.size != 25:
            raise ValueError("threshold table must have 25 entries")
        self.ctx.check(self.lib.tsu_ising2d_set_thresholds(self.h, _ptr(t, _u64p)))

    def set_model(self, J, h, T, mode=MODE_PHYSICAL):
        self.ctx.check(self.lib.tsu_ising2d_set_model(self.h, float(J), float(h), float(T), int(mode)))

    def set_kernel(self, kernel=KERNEL_AUTO, sweeps_per_launch=0):
        self.ctx.check(self.lib.tsu_ising2d_set_kernel(self.h, int(kernel), int(sweeps_per_launch)))

    def sweep(self, n_sweeps, seed, sweep0=0, replica=0):
        self.ctx.check(self.lib.tsu_ising2d_sweep(self.h, int(n_sweeps), int(seed), int(sweep0), int(replica)))

    def sweep_part(self, n_sweeps, seed, sweep0, part, replica=0):
        self.ctx.check(self.lib.tsu_ising2d_sweep_part(self.h, int(n_sweeps), int(seed), int(sweep0), int(replica), int(part)))

    def sample(self, n_burnin, n_sweeps, n_samples, seed, sweep0=0, replica=0):
        """n_burnin sweeps, then n_samples x (n_sweeps sweeps, record): (n_samples, rows, cols) int8, one PCIe transfer."""
        out = np.empty((int(n_samples), self.rows, self.cols), dtype=np.int8)
        self.ctx.check(self.lib.tsu_ising2d_sample(self.h, int(n_burnin), int(n_sweeps), int(n_samples), int(seed), int(sweep0),
                                                   int(replica), _ptr(out, _i8p)))
        return out

    def observables(self):
        a, b = C.c_int64(0), C.c_int64(0)
        self.ctx.check(self.lib.tsu_ising2d_observables(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def row_ptr(self, local_row):
        p, pitch = _vp(), C.c_size_t(0)
        self.ctx.check(self.lib.tsu_ising2d_row_ptr(self.h, int(local_row), C.byref(p), C.byref(pitch)))
        return p.value, pitch.value

    def set_timing(self, enable=True):
        self.ctx.check(self.lib.tsu_ising2d_set_timing(self.h, int(bool(enable))))

    def launch_count(self):
        n = C.c_uint64(0)
        self.ctx.check(self.lib.tsu_ising2d_launch_count(self.h, C.byref(n)))
        return n.value

    def last_sweep_ms(self):
        ms = C.c_float(0)
        self.ctx.check(self.lib.tsu_ising2d_last_sweep_ms(self.h, C.byref(ms)))
        return ms.value


def sweep_batch(lattices, n_sweeps, seeds, sweep0s, replicas=None):
    """n_sweeps sweeps of every lattice (own thresholds, seed, sweep counter, replica id); lattices that fit the
    one-workgroup kernel run concurrently in one launch.  Same results as sweeping them one by one."""
    n = len(lattices)
    if n == 0:
        return
    ctx = lattices[0].ctx
    hs = (_vp * n)(*[l.h for l in lattices])
    sd = (C.c_uint64 * n)(*[int(v) for v in seeds])
    s0 = (C.c_uint32 * n)(*[int(v) for v in sweep0s])
    rp = (C.c_uint32 * n)(*([0] * n if replicas is None else [int(v) for v in replicas]))
    ctx.check(lattices[0].lib.tsu_ising2d_sweep_batch(hs, n, int(n_sweeps), sd, s0, rp))


def observables_batch(lattices):
    """[(sum of spins, sum over bonds)] of every lattice with one synchronisation."""
    n = len(lattices)
    if n == 0:
        return []
    hs = (_vp * n)(*[l.h for l in lattices])
    a, b = (C.c_int64 * n)(), (C.c_int64 * n)()
    lattices[0].ctx.check(lattices[0].lib.tsu_ising2d_observables_batch(hs, n, a, b))
    return [(int(a[i]), int(b[i])) for i in range(n)]


class DenseSystem:
    """tsu_dense handle: dense coupling matrix + bias resident on the GPU, {0,1} int8 state."""

    def __init__(self, J, bias=None, dtype=DTYPE_F64, ctx=None):
        self.ctx = ctx or Context.default()
        self.lib = self.ctx.lib
        J = np.asarray(J)
        if J.ndim != 2 or J.shape[0] != J.shape[1]:
            raise ValueError("Coupling matrix must be square")
        self.n = J.shape[0]
        Jc = np.ascontiguousarray(J, dtype=np.float64 if dtype == DTYPE_F64 else np.float32)
        b = None if bias is None else np.ascontiguousarray(bias, dtype=np.float64)
        if b is not None and b.size != self.n:
            raise ValueError(f"bias must have length {self.n}")
        h = _vp()
        self.ctx.check(self.lib.tsu_dense_create(self.ctx.h, self.n, Jc.ctypes.data_as(_vp), int(dtype),
                                                 None if b is None else _ptr(b, _f64p), C.byref(h)))
        self.h = h
        # host mirror of the resident state, valid right after set_state / get_state: a caller that hands back the state it was
        # just given (the functional gibbs_sweep(state, ...) -> state idiom, compute_energy of that state) does not upload it again --
        # and the library keeps the fields J s + b it holds for exactly that state (tsu_dense_set_state would drop them)
        self._mirror = None

    def close(self):
        if getattr(self, "h", None):
            self.lib.tsu_dense_destroy(self.h)
            self.h = None

    def __del__(self, _finalizing=sys.is_finalizing):
        if not _finalizing():  # at interpreter exit the process teardown frees the device memory
            self.close()

    def set_state(self, bits):
        b = np.ascontiguousarray(bits, dtype=np.int8).reshape(self.n)
        if self._mirror is not None and np.array_equal(b, self._mirror):
            return  # the device already holds exactly this state
        self.ctx.check(self.lib.tsu_dense_set_state(self.h, _ptr(b, _i8p)))
        self._mirror = b.copy()

    def get_state(self):
        out = np.empty(self.n, dtype=np.int8)
        self.ctx.check(self.lib.tsu_dense_get_state(self.h, _ptr(out, _i8p)))
        self._mirror = out.copy()
        return out

    def sweep(self, T, n_sweeps, seed=0, sweep0=0, replica=0, order=None, replay_uniforms=None):
        o = None if order is None else np.ascontiguousarray(order, dtype=np.int64).reshape(n_sweeps, self.n)
        u = None if replay_uniforms is None else np.ascontiguousarray(replay_uniforms, dtype=np.float64).reshape(
            n_sweeps, self.n)
        self._mirror = None
        self.ctx.check(self.lib.tsu_dense_sweep(self.h, float(T), int(n_sweeps), None if o is None else _ptr(o, _i64p),
                                                int(seed), int(sweep0), int(replica),
                                                None if u is None else _ptr(u, _f64p)))

    def sample(self, T, n_burnin, n_sweeps, n_samples, seed=0, sweep0=0, replica=0, order=None, replay_uniforms=None):
        """The whole sample_boltzmann run from the resident state in one C call: (n_samples, n) int8 of 0/1."""
        total = int(n_burnin) + int(n_samples) * int(n_sweeps)
        o = None if order is None else np.ascontiguousarray(order, dtype=np.int64).reshape(total, self.n)
        u = None if replay_uniforms is None else np.ascontiguousarray(replay_uniforms, dtype=np.float64).reshape(total, self.n)
        out = np.empty((int(n_samples), self.n), dtype=np.int8)
        self._mirror = None
        self.ctx.check(self.lib.tsu_dense_sample(self.h, float(T), int(n_burnin), int(n_sweeps), int(n_samples),
                                                 None if o is None else _ptr(o, _i64p), int(seed), int(sweep0), int(replica),
                                                 None if u is None else _ptr(u, _f64p), _ptr(out, _i8p)))
        return out

    def anneal(self, temperatures, seed=0, sweep0=0, replica=0, order=None, replay_uniforms=None):
        """One sweep per entry of ``temperatures`` from the resident state, every state recorded: (n_steps, n) int8."""
        t = np.ascontiguousarray(temperatures, dtype=np.float64)
        steps = t.size
        o = None if order is None else np.ascontiguousarray(order, dtype=np.int64).reshape(steps, self.n)
        u = None if replay_uniforms is None else np.ascontiguousarray(replay_uniforms, dtype=np.float64).reshape(steps, self.n)
        out = np.empty((steps, self.n), dtype=np.int8)
        self._mirror = None
        self.ctx.check(self.lib.tsu_dense_anneal(self.h, _ptr(t, _f64p), steps, None if o is None else _ptr(o, _i64p), int(seed),
                                                 int(sweep0), int(replica), None if u is None else _ptr(u, _f64p), _ptr(out, _i8p)))
        return out

    def sweep_replicas(self, states, temperatures, n_sweeps, seeds, sweep0s, replicas=None, replay_uniforms=None):
        """n_sweeps sweeps of every replica state (rows of ``states``) at its own temperature: (R, n) int8 of 0/1."""
        st = np.ascontiguousarray(states, dtype=np.int8).reshape(-1, self.n).copy()
        R = st.shape[0]
        t = np.ascontiguousarray(temperatures, dtype=np.float64).reshape(R)
        sd = np.ascontiguousarray(seeds, dtype=np.uint64).reshape(R)
        s0 = np.ascontiguousarray(sweep0s, dtype=np.uint32).reshape(R)
        rp = np.zeros(R, dtype=np.uint32) if replicas is None else np.ascontiguousarray(replicas, dtype=np.uint32).reshape(R)
        u = None if replay_uniforms is None else np.ascontiguousarray(replay_uniforms, dtype=np.float64).reshape(R, int(n_sweeps), self.n)
        self._mirror = None  # (systems beyond the one-workgroup kernels are swept replica by replica through the resident state)
        self.ctx.check(self.lib.tsu_dense_sweep_replicas(self.h, R, _ptr(t, _f64p), int(n_sweeps), _ptr(st, _i8p), _ptr(sd, _u64p),
                                                         _ptr(s0, _u32p), _ptr(rp, _u32p), None if u is None else _ptr(u, _f64p)))
        return st

    def energy(self):
        e = C.c_double(0)
        self.ctx.check(self.lib.tsu_dense_energy(self.h, C.byref(e)))
        return e.value

    def launch_counts(self):
        """(launches of the owner-computes kernel k2_own, launches of the pipeline k2_pipe) made by this system so far."""
        c = (C.c_uint64 * 2)()
        self.ctx.check(self.lib.tsu_dense_launch_counts(self.h, c))
        return int(c[0]), int(c[1])

    def energies(self, states):
        """Energies of the given states ((k, n) of 0/1), evaluated on the device in one call; the resident state stays as it is."""
        st = np.ascontiguousarray(states, dtype=np.int8).reshape(-1, self.n)
        out = np.empty(st.shape[0], dtype=np.float64)
        if st.shape[0]:
            self.ctx.check(self.lib.tsu_dense_energies(self.h, _ptr(st, _i8p), int(st.shape[0]), _ptr(out, _f64p)))
        return out


def comm_unique_id() -> bytes:
    """128 opaque bytes that identify a new RCCL communicator: create on one rank, hand to the others."""
    lib = load_library()
    buf = np.zeros(128, dtype=np.uint8)
    rc = lib.tsu_comm_unique_id(_ptr(buf, _u8p))
    if rc != TSU_OK:
        raise HipError(f"tsu_comm_unique_id failed ({rc}): {lib.tsu_last_error(None).decode()}")
    return buf.tobytes()


class Comm:
    """tsu_comm handle: an RCCL communicator below the C ABI (one rank = one process = one GPU)."""

    def __init__(self, nranks, rank, unique_id: bytes, ctx=None):
        self.ctx = ctx or Context.default()
        self.lib = self.ctx.lib
        self.nranks, self.rank = int(nranks), int(rank)
        uid = np.frombuffer(unique_id, dtype=np.uint8).copy()
        if uid.size != 128:
            raise ValueError("the unique id is 128 bytes")
        h = _vp()
        self.ctx.check(self.lib.tsu_comm_create(self.ctx.h, self.nranks, self.rank, _ptr(uid, _u8p), C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.tsu_comm_destroy(self.h)
            self.h = None

    def __del__(self, _finalizing=sys.is_finalizing):
        if not _finalizing():
            self.close()

    def halo_exchange(self, lattice: "Lattice"):
        self.ctx.check(self.lib.tsu_ising2d_halo_exchange(lattice.h, self.h))

    def wait(self, timeout_s=120.0):
        """Bounded wait for the stream's sweeps and exchanges; returns the number of halo exchanges issued so far."""
        n = C.c_uint64(0)
        self.ctx.check(self.lib.tsu_comm_wait(self.h, float(timeout_s), C.byref(n)))
        return int(n.value)

    def allreduce(self, values):
        v = np.ascontiguousarray(values, dtype=np.int64).copy()
        self.ctx.check(self.lib.tsu_comm_allreduce_i64(self.h, _ptr(v, _i64p), v.size))
        return v


class SparseSystem:
    """tsu_sparse handle: a sparse coupling graph (CSR, bit couplings incl. an optional diagonal) with a proper colouring,
    swept one colour class at a time (K5).  ``order`` lists the sites colour by colour, ``color_offsets`` delimits the
    colours in it.  State is {0,1} int8 in site order."""

    def __init__(self, row_ptr, col_idx, values, bias, color_offsets, order, ctx=None):
        self.ctx = ctx or Context.default()
        self.lib = self.ctx.lib
        rp = np.ascontiguousarray(row_ptr, dtype=np.int64)
        ci = np.ascontiguousarray(col_idx, dtype=np.int32)
        va = np.ascontiguousarray(values, dtype=np.float64)
        self.n = rp.size - 1
        b = None if bias is None else np.ascontiguousarray(bias, dtype=np.float64).reshape(self.n)
        co = np.ascontiguousarray(color_offsets, dtype=np.int32)
        od = np.ascontiguousarray(order, dtype=np.int32).reshape(self.n)
        self.n_colors = co.size - 1
        h = _vp()
        self.ctx.check(self.lib.tsu_sparse_create(self.ctx.h, self.n, _ptr(rp, _i64p), _ptr(ci, _i32p), _ptr(va, _f64p),
                                                  None if b is None else _ptr(b, _f64p), self.n_colors, _ptr(co, _i32p),
                                                  _ptr(od, _i32p), C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.tsu_sparse_destroy(self.h)
            self.h = None

    def __del__(self, _finalizing=sys.is_finalizing):
        if not _finalizing():
            self.close()

    def set_state(self, bits):
        b = np.ascontiguousarray(bits, dtype=np.int8).reshape(self.n)
        self.ctx.check(self.lib.tsu_sparse_set_state(self.h, _ptr(b, _i8p)))

    def get_state(self):
        out = np.empty(self.n, dtype=np.int8)
        self.ctx.check(self.lib.tsu_sparse_get_state(self.h, _ptr(out, _i8p)))
        return out

    def sweep(self, T, n_sweeps, seed=0, sweep0=0, replica=0):
        self.ctx.check(self.lib.tsu_sparse_sweep(self.h, float(T), int(n_sweeps), int(seed), int(sweep0), int(replica)))

    def sample(self, T, n_burnin, n_sweeps, n_samples, seed=0, sweep0=0, replica=0):
        out = np.empty((int(n_samples), self.n), dtype=np.int8)
        self.ctx.check(self.lib.tsu_sparse_sample(self.h, float(T), int(n_burnin), int(n_sweeps), int(n_samples), int(seed),
                                                  int(sweep0), int(replica), _ptr(out, _i8p)))
        return out

    def energy(self):
        """(-1/2 s'Js - b's of the resident bits, sum of the spins 2b-1)."""
        e, m = C.c_double(0), C.c_int64(0)
        self.ctx.check(self.lib.tsu_sparse_energy(self.h, C.byref(e), C.byref(m)))
        return e.value, m.value


class LangevinChains:
    """tsu_langevin handle: n_chains x dim float32 states with a separable quadratic energy."""

    def __init__(self, n_chains, dim, ctx=None):
        self.ctx = ctx or Context.default()
        self.lib = self.ctx.lib
        self.n_chains, self.dim = int(n_chains), int(dim)
        h = _vp()
        self.ctx.check(self.lib.tsu_langevin_create(self.ctx.h, self.n_chains, self.dim, C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.tsu_langevin_destroy(self.h)
            self.h = None

    def __del__(self, _finalizing=sys.is_finalizing):
        if not _finalizing():  # at interpreter exit the process teardown frees the device memory
            self.close()

    def set_state(self, x):
        xx = np.ascontiguousarray(np.broadcast_to(np.asarray(x, dtype=np.float32), (self.n_chains, self.dim)))
        self.ctx.check(self.lib.tsu_langevin_set_state(self.h, _ptr(xx, _f32p)))

    def get_state(self):
        out = np.empty((self.n_chains, self.dim), dtype=np.float32)
        self.ctx.check(self.lib.tsu_langevin_get_state(self.h, _ptr(out, _f32p)))
        return out

    def set_energy(self, k, mu):
        kk = np.ascontiguousarray(np.broadcast_to(np.asarray(k, dtype=np.float32), (self.dim,)))
        mm = np.ascontiguousarray(np.broadcast_to(np.asarray(mu, dtype=np.float32), (self.dim,)))
        self.ctx.check(self.lib.tsu_langevin_set_energy(self.h, _ptr(kk, _f32p), _ptr(mm, _f32p)))

    def set_coupling(self, A, b=None):
        """E = 1/2 x^T A x + b^T x with a symmetric (dim, dim) matrix: the steps that follow run the coupled kernel."""
        aa = np.ascontiguousarray(A, dtype=np.float32).reshape(self.dim, self.dim)
        bb = None if b is None else np.ascontiguousarray(np.broadcast_to(np.asarray(b, dtype=np.float32), (self.dim,)))
        self.ctx.check(self.lib.tsu_langevin_set_coupling(self.h, _ptr(aa, _f32p), None if bb is None else _ptr(bb, _f32p)))

    def restart(self, x_init, amp, seed, chain0=0):
        xi = np.ascontiguousarray(np.broadcast_to(np.asarray(x_init, dtype=np.float32), (self.dim,)))
        self.ctx.check(self.lib.tsu_langevin_restart(self.h, _ptr(xi, _f32p), float(amp), int(seed), int(chain0)))

    def step(self, n_steps, dt, gamma, T, seed, step0=0, chain0=0, trajectory=False):
        traj = np.empty((n_steps, self.n_chains, self.dim), dtype=np.float32) if trajectory else None
        self.ctx.check(self.lib.tsu_langevin_step(self.h, int(n_steps), float(dt), float(gamma), float(T), int(seed),
                                                  int(step0), int(chain0),
                                                  None if traj is None else _ptr(traj, _f32p)))
        return traj

    def set_kernel(self, steps_per_launch=0):
        self.ctx.check(self.lib.tsu_langevin_set_kernel(self.h, int(steps_per_launch)))
