"""ctypes bindings of the two TEST-ONLY checkers:

  * ``Oracle``  -- oracle/liboracle.so, our C restatement of the reference path
  * ``Ref``     -- oracle/_ref/libadmm_ref.so, the REAL reference compiled from
                   /root/reference (only exists where oracle/Makefile built it)

Nothing in the product package imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

KIND = dict(ANCHOR=0, SPRING=1, TET_LINEAR=2, TET_VOLUME=3, TET_NH=4, TET_STVK=5, TRI_STRAIN=6, BEND=7, COLLISION=8, TRI_AREA=9, TRI_FUNG=10)
KIND_NODES = [1, 2, 4, 4, 4, 4, 3, 4, 1, 3, 3]
KIND_ROWS = [3, 3, 9, 9, 9, 9, 6, 9, 3, 6, 6]
KIND_PARAMS = [2, 1, 1, 3, 3, 3, 4, 1, 1, 4, 3]

dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int)


def _d(a):
    return a.ctypes.data_as(dp)


def _i(a):
    return a.ctypes.data_as(ip)


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "liboracle.so"])
    if os.path.isdir("/root/reference"):
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "ref"])


def have_ref():
    """The compiled reference (oracle/_ref, built from /root/reference where that exists -- this container; the GPU box
    only has what was built here).  Built on first use so that the suite does not depend on __graft_entry__.build() having run."""
    lib = os.path.join(ORACLE_DIR, "_ref", "libadmm_ref.so")
    if not os.path.exists(lib) and os.path.isdir("/root/reference"):
        subprocess.call(["make", "-s", "-C", ORACLE_DIR, "ref"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return os.path.exists(lib)


def deformed_start(x0):
    """A smooth, LARGE deformation of a bar's rest positions built from + and * only (no transcendental, no division: the same
    bits on every host and every numpy build): shear, bending and twist-like terms that put the deformation gradients 5-20 %
    away from the identity.  The start of the one-iteration full-size fixture (tests/golden/traj_bar_1M_one_iter.npz): with it
    the single local step of that frame is a real Neo-Hookean prox for every tet instead of the identity."""
    p = np.asarray(x0, dtype=np.float64).reshape(-1, 3)
    q = p.copy()
    q[:, 0] = p[:, 0] + 0.02 * (p[:, 1] * p[:, 2])
    q[:, 1] = p[:, 1] - 0.004 * (p[:, 2] * p[:, 2])
    q[:, 2] = p[:, 2] + 0.03 * (p[:, 0] * p[:, 1])
    return q.ravel()


def solve_rhs(seed, x0, m3):
    """The three right-hand sides of the direct solve fixtures (tests/golden/solve_*.npz; 3n doubles each, node-major like
    System::m_x): white noise, an M x_bar-like smooth vector (mass x displaced positions, what System.cpp:61 feeds the solver),
    and one unit spike (a column of A^-1)."""
    rng = np.random.default_rng(int(seed))
    x0 = np.asarray(x0, dtype=np.float64).ravel(); m3 = np.asarray(m3, dtype=np.float64).ravel()
    n = x0.size
    b0 = rng.normal(size=n)
    b1 = m3 * (x0 + 0.01 * np.sin(7.0 * x0 + 0.3 * np.arange(n)))
    b2 = np.zeros(n); b2[int(rng.integers(0, n))] = 1.0
    return np.stack([b0, b1, b2])


class _Sys:
    """Common python face of oracle / reference systems."""

    prefix = None
    lib = None

    def _f(self, name):
        return getattr(self.lib, self.prefix + name)

    def settings(self, dt, iters):
        raise NotImplementedError

    def add_nodes(self, x, m):
        x = np.ascontiguousarray(x, dtype=np.float64).ravel()
        m = np.ascontiguousarray(m, dtype=np.float64).ravel()
        return self._f("add_nodes")(self.h, x.size, _d(x), _d(m))

    def add_forces(self, kind, idx, params):
        idx = np.ascontiguousarray(idx, dtype=np.int32).reshape(-1, KIND_NODES[kind])
        n = idx.shape[0]
        params = np.ascontiguousarray(np.broadcast_to(np.asarray(params, dtype=np.float64), (n, KIND_PARAMS[kind])))
        r = self._f("add_forces")(self.h, kind, n, _i(idx), _d(params))
        assert r >= 0
        return r

    def add_moving_anchor(self, idx, pos, active=True, weight=-1.0):
        pos = np.ascontiguousarray(pos, dtype=np.float64)
        return self._f("add_moving_anchor")(self.h, int(idx), _d(pos), int(active), float(weight))

    def set_control_point(self, handle, pos, active=True):
        pos = np.ascontiguousarray(pos, dtype=np.float64)
        self._f("set_control_point")(self.h, handle, _d(pos), int(active))

    def add_gravity(self, g):
        self._f("add_gravity")(self.h, float(g[0]), float(g[1]), float(g[2]))

    def initialize(self):
        return bool(self._f("initialize")(self.h))

    def step(self):
        return bool(self._f("step")(self.h))

    @property
    def dof(self):
        return self._f("dof")(self.h)

    @property
    def rows(self):
        return self._f("rows")(self.h)

    @property
    def n_forces(self):
        return self._f("n_forces")(self.h)

    def D_triplets(self):
        n = self._f("D_nnz")(self.h)
        r = np.zeros(n, np.int32); c = np.zeros(n, np.int32); v = np.zeros(n)
        self._f("get_D")(self.h, _i(r), _i(c), _d(v))
        return r, c, v

    def L_nnz(self):
        return self._f("L_nnz")(self.h)

    def time_steps(self, frames):
        return self._f("time_steps")(self.h, frames)


class Ref(_Sys):
    prefix = "ref_"

    @classmethod
    def load(cls):
        if cls.lib is None:
            lib = C.CDLL(os.path.join(ORACLE_DIR, "_ref", "libadmm_ref.so"))
            lib.ref_create.restype = C.c_void_p
            for n in ("destroy", "settings", "set_control_point", "add_gravity", "get_x", "set_x", "get_v", "set_v", "get_u",
                      "get_z", "get_wdiag", "get_D", "svd3", "svd32", "recompute_weights", "set_force_weight", "get_masses",
                      "get_control_point", "add_wind"):
                getattr(lib, "ref_" + n).restype = None
            lib.ref_D_nnz.restype = C.c_long
            lib.ref_L_nnz.restype = C.c_long
            lib.ref_force_weight.restype = C.c_double
            lib.ref_time_steps.restype = C.c_double
            lib.ref_elapsed.restype = C.c_double
            lib.ref_settings.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_int]
            lib.ref_add_gravity.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double]
            lib.ref_add_moving_anchor.argtypes = [C.c_void_p, C.c_int, dp, C.c_int, C.c_double]
            lib.ref_set_force_weight.argtypes = [C.c_void_p, C.c_int, C.c_double]
            lib.ref_add_wind.argtypes = [C.c_void_p, C.c_int, ip, C.c_double, C.c_double, C.c_double]
            lib.ref_add_explicit_subset.argtypes = [C.c_void_p, C.c_int, ip, C.c_double, C.c_double, C.c_double]
            lib.ref_add_explicit_subset.restype = None
            lib.ref_set_explicit_dir.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double]
            lib.ref_set_explicit_dir.restype = None
            lib.ref_add_collision.argtypes = [C.c_void_p, C.c_int, ip, dp, C.c_double]
            lib.ref_project_single.argtypes = [C.c_int, dp, dp, C.c_double, C.c_int, dp, dp, dp, dp, dp, ip, dp]
            for n in ("add_nodes", "add_forces", "set_control_point", "initialize", "step", "dof", "rows", "n_forces",
                      "get_x", "set_x", "get_v", "set_v", "get_u", "get_z", "get_wdiag", "force_global_idx", "force_weight",
                      "D_nnz", "get_D", "L_nnz", "get_hyper_state", "set_hyper_state", "time_steps", "destroy",
                      "recompute_weights", "get_masses", "get_control_point", "elapsed"):
                fn = getattr(lib, "ref_" + n)
                if fn.argtypes is None:
                    fn.argtypes = None  # first arg is the handle; set below where needed
            cls.lib = lib
        return cls.lib

    def __init__(self):
        self.load()
        self.h = C.c_void_p(self.lib.ref_create())

    def __del__(self):
        try:
            self.lib.ref_destroy(self.h)
        except Exception:
            pass

    def settings(self, dt, iters):
        self.lib.ref_settings(self.h, dt, iters, 0)

    def _vec(self, name, n):
        a = np.zeros(n)
        getattr(self.lib, "ref_get_" + name)(self.h, _d(a))
        return a

    @property
    def x(self):
        return self._vec("x", self.dof)

    @x.setter
    def x(self, val):
        val = np.ascontiguousarray(val, dtype=np.float64)
        self.lib.ref_set_x(self.h, _d(val))

    @property
    def v(self):
        return self._vec("v", self.dof)

    @property
    def masses(self):
        return self._vec("masses", self.dof)

    @property
    def u(self):
        return self._vec("u", self.rows)

    @property
    def z(self):
        return self._vec("z", self.rows)

    @property
    def wdiag(self):
        return self._vec("wdiag", self.rows)

    def global_idx(self):
        return np.array([self.lib.ref_force_global_idx(self.h, i) for i in range(self.n_forces)], np.int64)

    def weights(self):
        return np.array([self.lib.ref_force_weight(self.h, i) for i in range(self.n_forces)])

    def hyper_state(self, i):
        st = np.zeros(4)
        it = self.lib.ref_get_hyper_state(self.h, i, _d(st))
        return st, it

    def add_wind(self, tris, direction):
        tris = np.ascontiguousarray(tris, dtype=np.int32).reshape(-1, 3)
        self.lib.ref_add_wind(self.h, tris.shape[0], _i(tris), *[float(d) for d in direction])

    def add_explicit_subset(self, idx, g):
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        self.lib.ref_add_explicit_subset(self.h, idx.size, _i(idx), float(g[0]), float(g[1]), float(g[2]))

    def set_explicit_dir(self, which, g):
        self.lib.ref_set_explicit_dir(self.h, which, float(g[0]), float(g[1]), float(g[2]))

    def add_collision(self, types, params, weight=32.0):
        t = np.ascontiguousarray(types, dtype=np.int32); p = np.ascontiguousarray(params, dtype=np.float64).reshape(-1, 4)
        return self.lib.ref_add_collision(self.h, t.size, _i(t), _d(p), float(weight))

    def recompute_weights(self):
        self.lib.ref_recompute_weights(self.h)

    def set_force_weight(self, i, w):
        self.lib.ref_set_force_weight(self.h, i, w)

    def solve(self, b):
        """x = solver.solve(b) on the reference's own SimplicialLDLT of the 3n x 3n system (System.cpp:62,
        SimplicialCholesky.h:153-177) -- reached through the derived-class accessor of oracle/ref_shim.cpp"""
        b = np.ascontiguousarray(b, dtype=np.float64).ravel()
        assert b.size == self.dof
        x = np.zeros_like(b)
        self.lib.ref_solve(self.h, _d(b), _d(x))
        return x

    @classmethod
    def project_single(cls, kind, x_rest, params, Dx, u0=None, state=None, dt=0.04):
        """n_calls consecutive project() calls of one stand-alone element."""
        lib = cls.load()
        rows = KIND_ROWS[kind]
        Dx = np.ascontiguousarray(Dx, dtype=np.float64).reshape(-1, rows)
        n = Dx.shape[0]
        u = np.zeros(rows) if u0 is None else np.array(u0, dtype=np.float64)
        z_out = np.zeros((n, rows)); u_out = np.zeros((n, rows))
        st = np.array([1, 1, 1, 1], dtype=np.float64) if state is None else np.array(state, dtype=np.float64)
        iters = np.zeros(n, np.int32)
        init = np.zeros(16)
        x_rest = np.ascontiguousarray(x_rest, dtype=np.float64)
        params = np.ascontiguousarray(params, dtype=np.float64)
        r = lib.ref_project_single(kind, _d(x_rest), _d(params), dt, n, _d(Dx), _d(u), _d(z_out), _d(u_out), _d(st), _i(iters), _d(init))
        assert r == 0
        return dict(z=z_out, u=u_out, state=st, n_iters=iters, init=init)

    @classmethod
    def svd3(cls, F):
        lib = cls.load()
        F = np.ascontiguousarray(F, dtype=np.float64)
        U = np.zeros(9); S = np.zeros(3); V = np.zeros(9)
        lib.ref_svd3(_d(F), _d(U), _d(S), _d(V))
        return U, S, V

    @classmethod
    def svd32(cls, F):
        lib = cls.load()
        F = np.ascontiguousarray(F, dtype=np.float64)
        U = np.zeros(9); S = np.zeros(2); V = np.zeros(4)
        lib.ref_svd32(_d(F), _d(U), _d(S), _d(V))
        return U, S, V


class OrcForce(C.Structure):
    _fields_ = [("kind", C.c_int), ("idx", C.c_int * 4), ("params", C.c_double * 4), ("weight", C.c_double),
                ("B", C.c_double * 12), ("measure", C.c_double), ("alpha", C.c_double * 4), ("pos", C.c_double * 3),
                ("active", C.c_int), ("moving", C.c_int), ("state", C.c_double * 4), ("n_iters", C.c_int),
                ("n_fev", C.c_int), ("global_idx", C.c_int)]


class Oracle(_Sys):
    prefix = "orc_"

    @classmethod
    def load(cls):
        if cls.lib is None:
            path = os.path.join(ORACLE_DIR, "liboracle.so")
            if not os.path.exists(path):
                build_oracle()
            lib = C.CDLL(path)
            lib.orc_create.restype = C.c_void_p
            for n in ("x", "v", "u", "z", "wdiag"):
                getattr(lib, "orc_" + n).restype = dp
                getattr(lib, "orc_" + n).argtypes = [C.c_void_p]
            lib.orc_get_force.restype = C.POINTER(OrcForce)
            lib.orc_get_force.argtypes = [C.c_void_p, C.c_int]
            lib.orc_D_nnz.restype = C.c_long
            lib.orc_L_nnz.restype = C.c_long
            lib.orc_time_steps.restype = C.c_double
            lib.orc_settings.argtypes = [C.c_void_p, C.c_double, C.c_int]
            lib.orc_set_layout.argtypes = [C.c_void_p, C.c_int]
            lib.orc_add_gravity.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double]
            lib.orc_add_moving_anchor.argtypes = [C.c_void_p, C.c_int, dp, C.c_int, C.c_double]
            lib.orc_force_construct.argtypes = [C.POINTER(OrcForce), C.c_int, ip, dp]
            lib.orc_force_initialize.argtypes = [C.POINTER(OrcForce), dp]
            lib.orc_force_project.argtypes = [C.POINTER(OrcForce), C.c_double, dp, dp, dp]
            lib.orc_add_explicit.argtypes = [C.c_void_p, C.c_int, dp, C.c_int, ip]
            lib.orc_set_collision_shapes.argtypes = [C.c_void_p, C.c_int, ip, dp]
            lib.orc_track_residuals.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double]
            lib.orc_track_residuals.restype = None
            lib.orc_get_residuals.argtypes = [C.c_void_p, dp, dp, C.c_int]
            for n in ("settings", "set_layout", "add_gravity", "destroy", "get_D", "set_control_point", "svd3", "svd32", "add_explicit", "set_collision_shapes",
                      "oriented_svd", "force_construct", "force_initialize", "force_project"):
                getattr(lib, "orc_" + n).restype = None
            cls.lib = lib
        return cls.lib

    def __init__(self, ref_layout=False):
        self.load()
        self.h = C.c_void_p(self.lib.orc_create())
        self.lib.orc_set_layout(self.h, int(ref_layout))

    def __del__(self):
        try:
            self.lib.orc_destroy(self.h)
        except Exception:
            pass

    def settings(self, dt, iters):
        self.lib.orc_settings(self.h, dt, iters)

    def track_residuals(self, on=True, tol_r=0.0, tol_s=0.0):
        self.lib.orc_track_residuals(self.h, int(on), float(tol_r), float(tol_s))

    def residuals(self):
        r = np.zeros(256); s = np.zeros(256)
        n = self.lib.orc_get_residuals(self.h, _d(r), _d(s), 256)
        return r[:n], s[:n], n

    def _view(self, name, n):
        p = getattr(self.lib, "orc_" + name)(self.h)
        return np.ctypeslib.as_array(p, shape=(n,)) if n else np.zeros(0)

    @property
    def x(self):
        return self._view("x", self.dof).copy()

    @x.setter
    def x(self, val):
        self._view("x", self.dof)[:] = val

    @property
    def v(self):
        return self._view("v", self.dof).copy()

    @property
    def u(self):
        return self._view("u", self.rows).copy()

    @property
    def z(self):
        return self._view("z", self.rows).copy()

    @property
    def wdiag(self):
        return self._view("wdiag", self.rows).copy()

    def add_explicit(self, type_, direction, idx=None):
        d = np.ascontiguousarray(direction, dtype=np.float64)
        ix = np.zeros(0, np.int32) if idx is None else np.ascontiguousarray(idx, dtype=np.int32)
        n = ix.size // 3 if type_ == 1 else ix.size
        self.lib.orc_add_explicit(self.h, type_, _d(d), n, _i(ix))

    def set_collision_shapes(self, types, params):
        t = np.ascontiguousarray(types, dtype=np.int32); p = np.ascontiguousarray(params, dtype=np.float64).reshape(-1, 4)
        self.lib.orc_set_collision_shapes(self.h, t.size, _i(t), _d(p))

    def force(self, i):
        return self.lib.orc_get_force(self.h, i).contents

    def global_idx(self):
        return np.array([self.force(i).global_idx for i in range(self.n_forces)], np.int64)

    def weights(self):
        return np.array([self.force(i).weight for i in range(self.n_forces)])

    def hyper_state(self, i):
        f = self.force(i)
        return np.array(list(f.state)), f.n_iters

    def solve(self, b):
        """the oracle's LDL^T solve of the same 3n x 3n system (admm_oracle.c ldl_solve = SimplicialCholesky.h:153-177)"""
        b = np.ascontiguousarray(b, dtype=np.float64).ravel()
        assert b.size == self.dof
        x = np.zeros_like(b)
        self.lib.orc_solve(self.h, _d(b), _d(x))
        return x

    def local_step(self, xcur, dt=0.04):
        """ONE local step of the oracle on caller-supplied positions: Dx = D x_cur accumulated column-ascending like Eigen's
        column-major product (System.cpp:54), then project() of every force in list order (System.cpp:57-58) on the oracle's
        own u and warm-start state.  Returns flat (u, z) in the oracle's (compact) row layout; per force: rows
        global_idx .. global_idx + KIND_ROWS[kind]."""
        rr, cc, vv = self.D_triplets()
        Dx = np.zeros(self.rows)
        k = np.lexsort((cc, rr))
        for r_, c_, v_ in zip(rr[k], cc[k], vv[k]):
            Dx[r_] += v_ * xcur[c_]
        u = self._view("u", self.rows); z = self._view("z", self.rows)
        for i in range(self.n_forces):
            f = self.lib.orc_get_force(self.h, i)
            g = f.contents.global_idx; rows = KIND_ROWS[f.contents.kind]
            d = np.ascontiguousarray(Dx[g:g + rows]); uu = np.ascontiguousarray(u[g:g + rows]); zz = np.zeros(rows)
            self.lib.orc_force_project(f, float(dt), _d(d), _d(uu), _d(zz))
            u[g:g + rows] = uu; z[g:g + rows] = zz
        return u.copy(), z.copy()

    @classmethod
    def project_single(cls, kind, x_rest, params, Dx, u0=None, state=None, dt=0.04):
        lib = cls.load()
        rows = KIND_ROWS[kind]
        Dx = np.ascontiguousarray(Dx, dtype=np.float64).reshape(-1, rows)
        n = Dx.shape[0]
        f = OrcForce()
        idx = np.arange(4, dtype=np.int32)
        params = np.ascontiguousarray(params, dtype=np.float64)
        x_rest = np.ascontiguousarray(x_rest, dtype=np.float64)
        lib.orc_force_construct(C.byref(f), kind, _i(idx), _d(params))
        lib.orc_force_initialize(C.byref(f), _d(x_rest))
        if state is not None:
            for j in range(4):
                f.state[j] = state[j]
        u = np.zeros(rows) if u0 is None else np.array(u0, dtype=np.float64)
        z = np.zeros(rows)
        z_out = np.zeros((n, rows)); u_out = np.zeros((n, rows)); iters = np.zeros(n, np.int32); fev = np.zeros(n, np.int32)
        for c in range(n):
            d = np.ascontiguousarray(Dx[c])
            lib.orc_force_project(C.byref(f), dt, _d(d), _d(u), _d(z))
            z_out[c] = z; u_out[c] = u; iters[c] = f.n_iters; fev[c] = f.n_fev
        init = np.zeros(16)
        init[0] = f.weight
        if kind in (2, 3, 4, 5):
            init[1:13] = list(f.B); init[13] = f.measure
        elif kind in (6, 9, 10):
            init[1:7] = list(f.B)[:6]; init[7] = f.measure
        elif kind == 7:
            init[1:5] = list(f.alpha)
        elif kind == 1:
            init[1] = f.measure
        return dict(z=z_out, u=u_out, state=np.array(list(f.state)), n_iters=iters, n_fev=fev, init=init)

    @classmethod
    def svd3(cls, F):
        lib = cls.load()
        F = np.ascontiguousarray(F, dtype=np.float64)
        U = np.zeros(9); S = np.zeros(3); V = np.zeros(9)
        lib.orc_svd3(_d(F), _d(U), _d(S), _d(V))
        return U, S, V

    @classmethod
    def svd32(cls, F):
        lib = cls.load()
        F = np.ascontiguousarray(F, dtype=np.float64)
        U = np.zeros(9); S = np.zeros(2); V = np.zeros(4)
        lib.orc_svd32(_d(F), _d(U), _d(S), _d(V))
        return U, S, V


def extreme_matrices(rng, n):
    """3x3 inputs (row-major 9-vectors of D_i x in the device's row order) that walk the corners of the Jacobi SVD and the proxes: every
    power-of-ten scale the format holds, entries 300 decades apart inside one matrix, rank 0 / 1 / 2, diagonal, permutation, symmetric
    and antisymmetric 2x2 blocks (the t == 0 branch of real_2x2_jacobi_svd), off-diagonals at the rotation threshold, repeated
    singular values, inverted and reflected matrices, infinities and NaNs."""
    M = []
    base = rng.normal(size=(n, 3, 3))
    for k, e in enumerate([-305, -300, -200, -160, -154, -150, -100, -20, -8, 0, 8, 20, 100, 150, 154, 160, 200, 300, 305]):
        M.append(base[k] * 10.0 ** e)
    for k in range(40):                       # rows / columns / entries scaled by wildly different powers of ten
        A = rng.normal(size=(3, 3))
        if k % 3 == 0: A = A * 10.0 ** rng.uniform(-300, 0, size=(3, 1))
        elif k % 3 == 1: A = A * 10.0 ** rng.uniform(-300, 0, size=(1, 3))
        else: A = A * 10.0 ** rng.uniform(-160, 0, size=(3, 3))
        M.append(A)
    I = np.eye(3)
    M += [np.zeros((3, 3)), I, -I, 2.5 * I, I[[1, 2, 0]], I[[2, 1, 0]], np.diag([3.0, 2.0, 1.0]), np.diag([1.0, 1.0, 1e-300]), np.diag([1e-200, 1.0, 1e200]),
          np.diag([1.0, -1.0, 1.0]), np.diag([-2.0, -3.0, -4.0])]
    for k in range(10):                        # rank one and rank two
        a, b = rng.normal(size=3), rng.normal(size=3)
        M.append(np.outer(a, b)); M.append(np.outer(a, b) + np.outer(rng.normal(size=3), rng.normal(size=3)))
    for eps in (0.0, 1e-17, 2.2e-16, 4.5e-16, 1e-15, 1e-8, 1e-3):     # a 2x2 block at the rotation threshold, symmetric / antisymmetric / one-sided
        for sym in (1.0, -1.0, 0.0):
            A = np.diag([1.0, 1.0 + 1e-9, 0.5]); A[0, 1] = eps; A[1, 0] = sym * eps; M.append(A)
            A = np.diag([1.0, -1.0, 0.5]); A[0, 1] = eps; A[1, 0] = sym * eps; M.append(A)      # t = m00 + m11 == 0
    R = np.array([[np.cos(0.3), -np.sin(0.3), 0], [np.sin(0.3), np.cos(0.3), 0], [0, 0, 1]])
    M += [R, R @ np.diag([2.0, 2.0, 1.0]), R @ np.diag([1.0, 1.0, 1.0]) @ R.T, -R]
    for v in (np.inf, -np.inf, np.nan):
        A = rng.normal(size=(3, 3)); A[1, 2] = v; M.append(A)
    while len(M) < n:
        M.append(rng.normal(size=(3, 3)) * 10.0 ** rng.integers(-3, 4))
    return np.array(M[:n]).reshape(n, 9)
