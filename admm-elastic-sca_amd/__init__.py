"""admm-elastic-sca_amd: MI355X-native ADMM elastic solver (hot path of
mattoverby/admm-elastic-sca) -- Python plumbing over the C ABI.

The product is libadmm_hip.so (HIP kernels + C ABI, include/admm_hip.h).  This
module only binds it with ctypes for bench.py / tests and mirrors the
reference's ``admm::System`` surface (``add_nodes``, ``forces``, ``initialize``,
``step``, ``m_x``/``m_v``; reference deps/admm-elastic-sca/src/system/System.hpp:29-76)
so that scene code reads like the reference's samples.  There is no CPU
fallback: without the library or without a GPU every compute call raises.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build
from . import meshgen  # noqa: F401

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libadmm_hip.so")

KIND = dict(ANCHOR=0, SPRING=1, TET_LINEAR=2, TET_VOLUME=3, TET_NH=4, TET_STVK=5, TRI_STRAIN=6, BEND=7, COLLISION=8, TRI_AREA=9, TRI_FUNG=10)
SHARD = dict(contiguous=0, subtree=1)
KIND_NODES = [1, 2, 4, 4, 4, 4, 3, 4, 1, 3, 3]
KIND_ROWS = [3, 3, 9, 9, 9, 9, 6, 9, 3, 6, 6]
KIND_PARAMS = [2, 1, 1, 3, 3, 3, 4, 1, 1, 4, 3]
KIND_STATE = [0, 0, 0, 0, 4, 4, 0, 0, 0, 0, 4]
SHAPE = dict(FLOOR=0, SPHERE=1, CYLINDER=2)
EXPLICIT = dict(CONST=0, WIND=1)

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)

ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)
PROJECT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_double, C.c_int64, _dp, _dp, _dp)
KIND_GENERIC = 11


class Info(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("n_nodes", "n_elems_total", "n_elems_local", "rows_compact", "nnz_A", "nnz_L",
                                         "panel_bytes", "n_supernodes", "n_levels", "max_super_cols", "max_super_rows",
                                         "solve_contrib_rows")] + \
               [(n, C.c_double) for n in ("t_order_s", "t_symbolic_s", "t_numeric_s", "t_upload_s")] + \
               [(n, C.c_int32) for n in ("rank", "world", "device_id", "host_threads", "dense_solve", "device_factor")] + \
               [(n, C.c_int64) for n in ("rhs_slots", "sweep_entries_own", "sweep_entries_top", "sweep_entries_top_bwd", "nodes_own", "nodes_top",
                                         "comm_doubles_iter", "comm_doubles_frame", "factor_doubles_resident", "front_doubles",
                                         "factor_exchange_doubles")] + \
               [(n, C.c_int32) for n in ("factor_local", "dist_top")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class Timing(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("prologue_ms", "local_ms", "rhs_ms", "allreduce_ms", "solve_fwd_ms", "solve_bwd_ms",
                                         "epilogue_ms", "total_ms")] + [("iters", C.c_int32)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class AdmmHipError(RuntimeError):
    pass


_lib = None


def build(force=False, verbose=False):
    return _build.build(force=force, verbose=verbose)


def lib():
    """Loads libadmm_hip.so (rebuilding it when a source is newer: build() checks mtimes).  Raises if it cannot be loaded."""
    global _lib
    if _lib is None:
        # One HIP runtime per process: PyTorch bundles its own libamdhip64; importing it
        # first makes libadmm_hip.so bind to that copy (same SONAME) so that streams and
        # device pointers can be shared with torch.distributed (bench.py, tests).
        try:
            import torch  # noqa: F401
        except Exception:
            pass
        path = os.environ.get("ADMM_HIP_LIB", LIB_PATH)   # experimental variants (tools/ab_local.py)
        if path == LIB_PATH or not os.path.exists(path):
            build()     # no-op when the library is newer than every source (a stale library after an edit is worse than 0.1 s of stat calls)
        L = C.CDLL(path)
        L.admm_hip_last_error.restype = C.c_char_p
        L.admm_hip_last_error.argtypes = [C.c_void_p]
        L.admm_hip_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
        L.admm_hip_destroy.argtypes = [C.c_void_p]
        L.admm_hip_destroy.restype = None
        L.admm_hip_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        L.admm_hip_set_shard_mode.argtypes = [C.c_void_p, C.c_int]
        L.admm_hip_set_factor_local.argtypes = [C.c_void_p, C.c_int]
        L.admm_hip_debug_node_owner.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
        L.admm_hip_local_elements.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_int)]
        L.admm_hip_enable_residuals.argtypes = [C.c_void_p, C.c_int]
        L.admm_hip_set_tolerance.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_int]
        L.admm_hip_get_residuals.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int)]
        L.admm_hip_set_timestep.argtypes = [C.c_void_p, C.c_double]
        L.admm_hip_add_nodes.argtypes = [C.c_void_p, C.c_int, _dp, _dp, C.POINTER(C.c_int)]
        L.admm_hip_add_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, _ip, _dp, _dp, C.POINTER(C.c_int)]
        L.admm_hip_add_gravity.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double]
        L.admm_hip_set_gravity.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double]
        L.admm_hip_set_shard.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.admm_hip_add_explicit.argtypes = [C.c_void_p, C.c_int, _dp, C.c_int, _ip, C.POINTER(C.c_int)]
        L.admm_hip_set_collision_shapes.argtypes = [C.c_void_p, C.c_int, _ip, _dp]
        L.admm_hip_set_allreduce.argtypes = [C.c_void_p, ALLREDUCE_FN, C.c_void_p]
        L.admm_hip_finalize.argtypes = [C.c_void_p]
        L.admm_hip_set_weights.argtypes = [C.c_void_p, C.c_int, _dp]
        L.admm_hip_recompute_weights.argtypes = [C.c_void_p]
        L.admm_hip_update_anchors.argtypes = [C.c_void_p, C.c_int, _dp, _ip]
        L.admm_hip_step.argtypes = [C.c_void_p, C.c_int]
        L.admm_hip_sync.argtypes = [C.c_void_p]
        for n in ("get_x", "get_v"):
            getattr(L, "admm_hip_" + n).argtypes = [C.c_void_p, _dp]
        for n in ("set_x", "set_v", "local_step_only"):
            getattr(L, "admm_hip_" + n).argtypes = [C.c_void_p, _dp]
        L.admm_hip_read_local.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _dp, _ip]
        L.admm_hip_write_local.argtypes = [C.c_void_p, C.c_int, _dp, _dp]
        L.admm_hip_read_rest.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _ip]
        L.admm_hip_solve_only.argtypes = [C.c_void_p, _dp, _dp]
        L.admm_hip_local_step_dx.argtypes = [C.c_void_p, C.c_int, _dp]
        L.admm_hip_apply_A.argtypes = [C.c_void_p, _dp, _dp]
        L.admm_hip_debug_panel_solve_host.argtypes = [C.c_void_p, _dp, _dp]
        L.admm_hip_debug_math.argtypes = [C.c_void_p, C.c_int, C.c_int64, _dp, _dp]
        L.admm_hip_debug_gemm.argtypes = [C.c_void_p] + [C.c_int] * 7 + [C.c_double, C.c_double, _dp, C.c_int64, _dp, C.c_int64, _dp, C.c_int64]
        L.admm_hip_debug_potrf_inv.argtypes = [C.c_void_p, C.c_int, C.c_int, _dp, _dp]
        L.admm_hip_add_generic_batch.argtypes = [C.c_void_p, C.c_int, _ip, C.c_int64, _ip, _ip, _dp, _dp, C.POINTER(C.c_int)]
        L.admm_hip_set_project_hook.argtypes = [C.c_void_p, PROJECT_FN, C.c_void_p]
        L.admm_hip_rccl_unique_id.argtypes = [C.c_void_p]
        L.admm_hip_rccl_init.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.admm_hip_set_rccl_comm.argtypes = [C.c_void_p, C.c_void_p]
        L.admm_hip_debug_allreduce.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.admm_hip_rccl_async_error.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.admm_hip_allreduce_host.argtypes = [C.c_void_p, _dp, C.c_int64]
        L.admm_hip_debug_graph_state.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int64)]
        L.admm_hip_pin_host.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        L.admm_hip_upload_state.argtypes = [C.c_void_p, _dp, _dp]
        L.admm_hip_download_state.argtypes = [C.c_void_p, _dp, _dp]
        L.admm_hip_get_info.argtypes = [C.c_void_p, C.POINTER(Info)]
        L.admm_hip_enable_timing.argtypes = [C.c_void_p, C.c_int]
        L.admm_hip_keep_z.argtypes = [C.c_void_p, C.c_int]
        L.admm_hip_get_timing.argtypes = [C.c_void_p, C.POINTER(Timing)]
        L.admm_hip_get_timing_previous.argtypes = [C.c_void_p, C.POINTER(Timing)]
        _lib = L
    return _lib


def _d(a):
    return a.ctypes.data_as(_dp) if a is not None else None


def _i(a):
    return a.ctypes.data_as(_ip) if a is not None else None


class System:
    """Python face of the C ABI with the reference's System vocabulary.

    device_id < 0 gives a host-only context (assembly + factorization only;
    every device call raises) -- used by the CPU test-suite."""

    def __init__(self, device_id=0, stream=None):
        self.L = lib()
        h = C.c_void_p()
        rc = self.L.admm_hip_create(C.byref(h), int(device_id))
        if rc != 0:
            raise AdmmHipError("admm_hip_create(device %d) failed with code %d: no usable MI355X/HIP device "
                               "(this package has no CPU fallback)" % (device_id, rc))
        self.h = h
        self.batches = []  # (kind, n)
        self.n_nodes = 0
        self._cb = None
        if stream is not None:
            self._chk(self.L.admm_hip_set_stream(self.h, C.c_void_p(stream)))

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.L.admm_hip_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise AdmmHipError("admm_hip error %d: %s" % (rc, self.L.admm_hip_last_error(self.h).decode()))

    # ---- setup (System.hpp:36-63) ----
    def set_timestep(self, dt):
        self._chk(self.L.admm_hip_set_timestep(self.h, float(dt)))

    def add_nodes(self, x, m):
        x = np.ascontiguousarray(x, dtype=np.float64).ravel()
        m = np.ascontiguousarray(m, dtype=np.float64).ravel()
        assert x.size == m.size and x.size % 3 == 0
        tot = C.c_int()
        self._chk(self.L.admm_hip_add_nodes(self.h, x.size // 3, _d(x), _d(m), C.byref(tot)))
        self.n_nodes = tot.value
        return tot.value

    def add_forces(self, kind, idx, params, targets=None):
        idx = np.ascontiguousarray(idx, dtype=np.int32).reshape(-1, KIND_NODES[kind])
        n = idx.shape[0]
        params = np.ascontiguousarray(np.broadcast_to(np.asarray(params, dtype=np.float64), (n, KIND_PARAMS[kind])))
        tg = None if targets is None else np.ascontiguousarray(targets, dtype=np.float64).reshape(n, 3)
        b = C.c_int()
        self._chk(self.L.admm_hip_add_batch(self.h, kind, n, _i(idx), _d(params), _d(tg), C.byref(b)))
        self.batches.append((kind, n))
        return b.value

    def add_generic(self, elem_row_ptr, trip_row, trip_col, trip_val, row_weight):
        """A run of user-defined forces (admm_hip_add_generic_batch): selector rows as triplets (row relative to the batch,
        col = 3 * node + component), one weight per row; project() is the hook installed with set_project_hook."""
        erp = np.ascontiguousarray(elem_row_ptr, dtype=np.int32)
        tr = np.ascontiguousarray(trip_row, dtype=np.int32); tc = np.ascontiguousarray(trip_col, dtype=np.int32)
        tv = np.ascontiguousarray(trip_val, dtype=np.float64); rw = np.ascontiguousarray(row_weight, dtype=np.float64)
        assert tr.size == tc.size == tv.size and rw.size == erp[-1]
        b = C.c_int()
        self._chk(self.L.admm_hip_add_generic_batch(self.h, erp.size - 1, _i(erp), tr.size, _i(tr), _i(tc), _d(tv), _d(rw), C.byref(b)))
        self.batches.append((KIND_GENERIC, erp.size - 1))
        self._generic_rows = getattr(self, "_generic_rows", {})
        self._generic_rows[b.value] = np.diff(erp)
        return b.value

    def set_project_hook(self, pyfunc):
        """pyfunc(dt, Dx, u, z): numpy views over ALL generic rows; update u and z in place (Force::project, System.cpp:57-58)."""
        def tramp(user, dt, n_rows, dx, u, z):
            try:
                n = int(n_rows)
                pyfunc(float(dt), np.ctypeslib.as_array(dx, shape=(n,)), np.ctypeslib.as_array(u, shape=(n,)), np.ctypeslib.as_array(z, shape=(n,)))
                return 0
            except Exception as e:  # never let an exception cross the C boundary
                print("project hook raised:", e)
                return 1
        self._pcb = PROJECT_FN(tramp)
        self._chk(self.L.admm_hip_set_project_hook(self.h, self._pcb, None))

    # ---- RCCL inside the library (admm_hip.h "RCCL inside the library") ----
    def rccl_unique_id(self):
        buf = np.zeros(128, np.uint8)
        rc = self.L.admm_hip_rccl_unique_id(buf.ctypes.data_as(C.c_void_p))
        if rc != 0:
            raise AdmmHipError("admm_hip_rccl_unique_id failed (%d)" % rc)
        return buf

    def rccl_init(self, uid, rank, world):
        uid = np.ascontiguousarray(uid, dtype=np.uint8)
        assert uid.size == 128
        self._chk(self.L.admm_hip_rccl_init(self.h, uid.ctypes.data_as(C.c_void_p), int(rank), int(world)))

    def set_rccl_comm(self, comm_ptr):
        self._chk(self.L.admm_hip_set_rccl_comm(self.h, C.c_void_p(comm_ptr) if comm_ptr else None))

    def debug_allreduce(self, dev_ptr, count):
        self._chk(self.L.admm_hip_debug_allreduce(self.h, C.c_void_p(dev_ptr), int(count)))

    def rccl_async_error(self):
        """ncclCommGetAsyncError of the installed communicator: 0 while healthy (raises with RCCL's text otherwise)"""
        r = C.c_int(0)
        self._chk(self.L.admm_hip_rccl_async_error(self.h, C.byref(r)))
        return r.value

    def allreduce_host(self, vec):
        """a short host vector summed in place across the ranks through the installed transport (no-op at world 1)"""
        assert vec.dtype == np.float64 and vec.flags.c_contiguous
        self._chk(self.L.admm_hip_allreduce_host(self.h, _d(vec), vec.size))
        return vec

    def graph_state(self):
        a = C.c_int(0); b = C.c_int(0); n = C.c_int64(0)
        self._chk(self.L.admm_hip_debug_graph_state(self.h, C.byref(a), C.byref(b), C.byref(n)))
        return dict(iter_graph=bool(a.value), frame_graph_iters=b.value, graph_launches=n.value)

    # ---- the class API's frame boundary (host/admm/System.hpp step()) ----
    def pin_host(self, arr, on=True):
        self._chk(self.L.admm_hip_pin_host(self.h, C.c_void_p(arr.ctypes.data), arr.nbytes, 1 if on else 0))

    def upload_state(self, x=None, v=None):
        self._chk(self.L.admm_hip_upload_state(self.h, _d(x), _d(v)))

    def download_state(self, x=None, v=None):
        self._chk(self.L.admm_hip_download_state(self.h, _d(x), _d(v)))

    def add_gravity(self, g):
        self._chk(self.L.admm_hip_add_gravity(self.h, float(g[0]), float(g[1]), float(g[2])))

    def add_explicit(self, type_, direction, idx=None):
        d = np.ascontiguousarray(direction, dtype=np.float64)
        ix = None if idx is None else np.ascontiguousarray(idx, dtype=np.int32)
        n = 0 if ix is None else (ix.size // 3 if type_ == EXPLICIT["WIND"] else ix.size)
        w = C.c_int()
        self._chk(self.L.admm_hip_add_explicit(self.h, type_, _d(d), n, _i(ix), C.byref(w)))
        return w.value

    def set_collision_shapes(self, types, params):
        t = np.ascontiguousarray(types, dtype=np.int32)
        p = np.ascontiguousarray(params, dtype=np.float64).reshape(-1, 4)
        self._chk(self.L.admm_hip_set_collision_shapes(self.h, t.size, _i(t), _d(p)))

    def set_gravity(self, which, g):
        self._chk(self.L.admm_hip_set_gravity(self.h, which, float(g[0]), float(g[1]), float(g[2])))

    def set_shard(self, rank, world):
        self._chk(self.L.admm_hip_set_shard(self.h, rank, world))

    # ---- residuals / early exit (extension described at System.cpp:64-65) ----
    def enable_residuals(self, on=True):
        self._chk(self.L.admm_hip_enable_residuals(self.h, 1 if on else 0))

    def set_tolerance(self, eps_r, eps_s, check_every=1):
        self._chk(self.L.admm_hip_set_tolerance(self.h, float(eps_r), float(eps_s), int(check_every)))

    def residuals(self, capacity=256):
        r = np.zeros(capacity); s = np.zeros(capacity); n = C.c_int(0)
        self._chk(self.L.admm_hip_get_residuals(self.h, _d(r), _d(s), capacity, C.byref(n)))
        k = min(n.value, capacity)
        return r[:k], s[:k], n.value

    def set_allreduce(self, pyfunc):
        """pyfunc(dev_ptr:int, count:int, stream:int) -> 0 on success."""
        def tramp(user, buf, count, stream):
            try:
                return int(pyfunc(buf or 0, int(count), stream or 0) or 0)
            except Exception as e:  # never let an exception cross the C boundary
                print("allreduce hook raised:", e)
                return 1
        self._cb = ALLREDUCE_FN(tramp)
        self._chk(self.L.admm_hip_set_allreduce(self.h, self._cb, None))

    def initialize(self):
        self._chk(self.L.admm_hip_finalize(self.h))
        return True

    def recompute_weights(self):
        self._chk(self.L.admm_hip_recompute_weights(self.h))

    def set_weights(self, batch, w):
        w = np.ascontiguousarray(w, dtype=np.float64)
        self._chk(self.L.admm_hip_set_weights(self.h, batch, _d(w)))

    def update_anchors(self, batch, targets=None, active=None):
        tg = None if targets is None else np.ascontiguousarray(targets, dtype=np.float64)
        ac = None if active is None else np.ascontiguousarray(active, dtype=np.int32)
        self._chk(self.L.admm_hip_update_anchors(self.h, batch, _d(tg), _i(ac)))

    # ---- stepping (System.hpp:65) ----
    def step(self, admm_iters):
        self._chk(self.L.admm_hip_step(self.h, int(admm_iters)))

    def sync(self):
        self._chk(self.L.admm_hip_sync(self.h))

    # ---- state ----
    def _getn(self, fn):
        a = np.zeros(3 * self.n_nodes)
        self._chk(fn(self.h, _d(a)))
        return a

    @property
    def m_x(self):
        return self._getn(self.L.admm_hip_get_x)

    @m_x.setter
    def m_x(self, val):
        val = np.ascontiguousarray(val, dtype=np.float64).ravel()
        assert val.size == 3 * self.n_nodes
        self._chk(self.L.admm_hip_set_x(self.h, _d(val)))

    @property
    def m_v(self):
        return self._getn(self.L.admm_hip_get_v)

    @m_v.setter
    def m_v(self, val):
        val = np.ascontiguousarray(val, dtype=np.float64).ravel()
        self._chk(self.L.admm_hip_set_v(self.h, _d(val)))

    # ---- parity / introspection ----
    def info(self):
        inf = Info()
        self._chk(self.L.admm_hip_get_info(self.h, C.byref(inf)))
        return inf.as_dict()

    def local_elements(self, batch):
        """this rank's elements of a batch (reference order): the rows of read_local / write_local"""
        n = C.c_int(0)
        self._chk(self.L.admm_hip_local_elements(self.h, batch, None, 0, C.byref(n)))
        ids = np.zeros(n.value, np.int32)
        self._chk(self.L.admm_hip_local_elements(self.h, batch, _i(ids), n.value, C.byref(n)))
        return ids

    def local_range(self, batch):
        """contiguous sharding: [first, end) of this rank's elements"""
        ids = self.local_elements(batch)
        if ids.size == 0:
            kind, n = self.batches[batch]
            inf = self.info()
            return n * inf["rank"] // inf["world"], n * inf["rank"] // inf["world"]
        assert np.array_equal(ids, np.arange(ids[0], ids[0] + ids.size)), "not a contiguous shard"
        return int(ids[0]), int(ids[0]) + ids.size

    def node_owner(self):
        o = np.zeros(self.n_nodes, np.int32)
        self._chk(self.L.admm_hip_debug_node_owner(self.h, _i(o)))
        return o

    def set_shard_mode(self, mode):
        self._chk(self.L.admm_hip_set_shard_mode(self.h, SHARD[mode] if isinstance(mode, str) else int(mode)))

    def set_factor_local(self, on):
        """Rank-local factorization under subtree sharding (default on): initialize() / recompute_weights() are then collective calls."""
        self._chk(self.L.admm_hip_set_factor_local(self.h, int(bool(on))))

    def read_local(self, batch):
        kind, _ = self.batches[batch]
        if kind == KIND_GENERIC:     # u, z of this rank's user forces, element after element
            nr = int(self._generic_rows[batch][self.local_elements(batch)].sum())
            u = np.zeros(nr); z = np.zeros(nr)
            self._chk(self.L.admm_hip_read_local(self.h, batch, _d(u), _d(z), None, None))
            return dict(u=u, z=z)
        n = self.local_elements(batch).size
        rows = KIND_ROWS[kind]
        u = np.zeros((n, rows)); z = np.zeros((n, rows))
        st = np.zeros((n, 4 if KIND_STATE[kind] else 3)); it = np.zeros(n, np.int32)
        self._chk(self.L.admm_hip_read_local(self.h, batch, _d(u), _d(z), _d(st), _i(it)))
        return dict(u=u, z=z, state=st, n_iters=it)

    def write_local(self, batch, u=None, state=None):
        u = None if u is None else np.ascontiguousarray(u, dtype=np.float64)
        state = None if state is None else np.ascontiguousarray(state, dtype=np.float64)
        self._chk(self.L.admm_hip_write_local(self.h, batch, _d(u), _d(state)))

    def read_rest(self, batch):
        kind, n = self.batches[batch]
        w = np.zeros(n); rest = np.zeros((n, 12)); g = np.zeros(n, np.int32)
        self._chk(self.L.admm_hip_read_rest(self.h, batch, _d(w), _d(rest), _i(g)))
        return dict(weight=w, rest=rest, global_idx=g)

    def local_step_only(self, x_cur):
        x_cur = np.ascontiguousarray(x_cur, dtype=np.float64).ravel()
        self._chk(self.L.admm_hip_local_step_only(self.h, _d(x_cur)))

    def local_step_dx(self, batch, dx):
        dx = np.ascontiguousarray(dx, dtype=np.float64)
        self._chk(self.L.admm_hip_local_step_dx(self.h, batch, _d(dx)))

    def solve_only(self, b):
        b = np.ascontiguousarray(b, dtype=np.float64).ravel()
        x = np.zeros_like(b)
        self._chk(self.L.admm_hip_solve_only(self.h, _d(b), _d(x)))
        return x

    def apply_A(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64).ravel()
        y = np.zeros_like(x)
        self._chk(self.L.admm_hip_apply_A(self.h, _d(x), _d(y)))
        return y

    def debug_math(self, op, x):
        """the device's log (op 0) / exp (op 1) on an array (parity tests)"""
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty_like(x)
        self._chk(self.L.admm_hip_debug_math(self.h, int(op), x.size, _d(x), _d(y)))
        return y

    def debug_gemm(self, A, B, Cm, m, n, k, flags=0, alpha=1.0, beta=0.0):
        """gemm_f64_kernel on column-major (Fortran-ordered) 2-D arrays; Cm is updated in place and returned."""
        A = np.asfortranarray(A, dtype=np.float64); B = np.asfortranarray(B, dtype=np.float64)
        assert Cm.flags.f_contiguous and Cm.dtype == np.float64
        self._chk(self.L.admm_hip_debug_gemm(self.h, m, n, k, A.shape[0], B.shape[0], Cm.shape[0], flags, alpha, beta,
                                             _d(A), A.size, _d(B), B.size, _d(Cm), Cm.size))
        return Cm

    def debug_potrf_inv(self, blk):
        """potrf_inv_kernel: (L, L^-1) of a symmetric positive definite block of at most 64 rows."""
        w = blk.shape[0]
        Lm = np.asfortranarray(blk, dtype=np.float64).copy(order="F")
        X = np.zeros_like(Lm, order="F")
        self._chk(self.L.admm_hip_debug_potrf_inv(self.h, w, w, _d(Lm), _d(X)))
        return np.tril(Lm), X

    def debug_panel_solve_host(self, b):
        b = np.ascontiguousarray(b, dtype=np.float64).ravel()
        x = np.zeros_like(b)
        self._chk(self.L.admm_hip_debug_panel_solve_host(self.h, _d(b), _d(x)))
        return x

    def keep_z(self, on=True):
        """admm_hip_keep_z: whether admm_hip_step stores the tet batches' z (read_local) -- off for production frames."""
        self._chk(self.L.admm_hip_keep_z(self.h, 1 if on else 0))

    def enable_timing(self, on=True):
        self._chk(self.L.admm_hip_enable_timing(self.h, int(on)))

    def timing(self):
        t = Timing()
        self._chk(self.L.admm_hip_get_timing(self.h, C.byref(t)))
        return t.as_dict()

    def timing_previous(self):
        """phase times of the step BEFORE the last one (read after the next step has been queued: no idle GPU between frames)"""
        t = Timing()
        self._chk(self.L.admm_hip_get_timing_previous(self.h, C.byref(t)))
        return t.as_dict()


def initialize_together(systems, timeout=3600.0):
    """initialize() of several ranks' contexts living in ONE process (tests, tools/ranks_one_gpu.py), each from its own thread:
    under rank-local factorization (subtree shards, the default) admm_hip_finalize is a collective call -- the ranks meet in the
    all-reduce of the subtree roots' update matrices -- exactly like System::initialize() of N real processes."""
    import threading
    errs = []

    def run(s):
        try:
            s.initialize()
        except BaseException as e:  # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=run, args=(s,), daemon=True) for s in systems]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout)
    if errs:
        raise errs[0]
    if any(t.is_alive() for t in th):
        raise AdmmHipError("initialize_together: a rank is still inside initialize() after %g s" % timeout)


def make_bar_system(nx, ny, nz, kind=KIND["TET_NH"], mu=1e5, lam=1e5, max_iter=5, density=1000.0, h=0.05, dt=0.04,
                    gravity=(0.0, -9.8, 0.0), device_id=0, rank=0, world=1, stream=None, shard_mode=None):
    """The synthetic bar of BASELINE.md section 4 config 4: tets first, then
    StaticAnchors on the k = 0 face, gravity, lumped density-weighted mass."""
    x, tets = meshgen.bar(nx, ny, nz, h)
    m = meshgen.lumped_tet_mass(x, tets, density)
    s = System(device_id=device_id, stream=stream)
    s.set_timestep(dt)
    s.add_nodes(x.ravel(), np.repeat(m, 3))
    s.add_forces(kind, tets, [mu, lam, max_iter])
    s.add_forces(KIND["ANCHOR"], meshgen.bar_anchor_nodes(nx, ny), [-1.0, 1.0])
    s.add_gravity(gravity)
    if world > 1:
        s.set_shard(rank, world)
        if shard_mode is not None:
            s.set_shard_mode(shard_mode)
    s.n_tets = tets.shape[0]
    return s


def make_mixed_system(nx, ny, nz, cloth_w, cloth_l, device_id=0, dt=0.04, rank=0, world=1, stream=None, shard_mode=None):
    """BASELINE.md section 4 config 5 ("mixed scene"): a bar whose lower half (in z) is Neo-Hookean and
    upper half StVK (tets first), then a sym-plane cloth with LimitedTriangleStrain (k=100,
    limits .95/1.05) + BendForce (k=20) hanging from two corner anchors, bar face anchored, gravity.
    Returns (system, description dict with the arrays the oracle needs)."""
    x, tets = meshgen.bar(nx, ny, nz)
    m = meshgen.lumped_tet_mass(x, tets, 1000.0)
    half = tets.shape[0] // 2
    xc, tris = meshgen.sym_plane(cloth_w, cloth_l, size=1.0)
    xc = xc + np.array([3.0, 1.0, 0.0])
    hinges = meshgen.bend_hinges(tris)
    off = x.shape[0]
    X = np.concatenate([x, xc])
    M = np.concatenate([m, np.full(xc.shape[0], 0.5 / xc.shape[0])])
    s = System(device_id=device_id, stream=stream)
    s.set_timestep(dt)
    s.add_nodes(X.ravel(), np.repeat(M, 3))
    desc = dict(X=X, M=M, forces=[
        ("TET_NH", tets[:half], [1e5, 1e5, 5]), ("TET_STVK", tets[half:], [1e5, 1e5, 5]),
        ("TRI_STRAIN", tris + off, [100.0, 0.95, 1.05, 1.0]), ("BEND", hinges + off, [20.0]),
        ("ANCHOR", np.concatenate([meshgen.bar_anchor_nodes(nx, ny), np.array([off, off + cloth_w], dtype=np.int32)]), [-1.0, 1.0])])
    for name, idx, par in desc["forces"]:
        s.add_forces(KIND[name], idx, par)
    s.add_gravity((0.0, -9.8, 0.0))
    if world > 1:
        s.set_shard(rank, world)
        if shard_mode is not None:
            s.set_shard_mode(shard_mode)
    s.n_elements = tets.shape[0] + tris.shape[0]
    return s, desc
