"""ctypes binding of the parity oracle (oracle/libnirrt_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (nirrt_star_amd) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libnirrt_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "nirrt_oracle.c")
    mk = os.path.join(_HERE, "Makefile")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(src), os.path.getmtime(mk)):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


class StepResult(C.Structure):
    _fields_ = [("collided", C.c_int32), ("inserted", C.c_int32), ("nearest_idx", C.c_int64),
                ("new_idx", C.c_int64), ("n_near", C.c_int32), ("reparented", C.c_int32),
                ("n_rewired", C.c_int32), ("in_goal", C.c_int32), ("n", C.c_int64),
                ("node_new", C.c_double * 3)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        dp = C.POINTER(C.c_double)
        ip = C.POINTER(C.c_int64)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.c_int, C.c_int64, dp, dp, C.c_double, C.c_double, C.c_double, dp, dp,
                                 C.c_int, dp, C.c_int, dp]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_vertices.restype = dp
        L.orc_vertices.argtypes = [C.c_void_p]
        L.orc_parents.restype = ip
        L.orc_parents.argtypes = [C.c_void_p]
        L.orc_solutions.restype = ip
        L.orc_solutions.argtypes = [C.c_void_p]
        L.orc_num_vertices.restype = C.c_int64
        L.orc_num_vertices.argtypes = [C.c_void_p]
        L.orc_num_solutions.restype = C.c_int64
        L.orc_num_solutions.argtypes = [C.c_void_p]
        L.orc_load.argtypes = [C.c_void_p, C.c_int64, dp, ip]
        L.orc_cost.restype = C.c_double
        L.orc_cost.argtypes = [C.c_void_p, C.c_int64]
        L.orc_nearest.restype = C.c_int64
        L.orc_nearest.argtypes = [C.c_void_p, dp]
        L.orc_steer.argtypes = [C.c_void_p, dp, dp, dp]
        L.orc_near_radius.restype = C.c_double
        L.orc_near_radius.argtypes = [C.c_void_p, C.c_int64]
        L.orc_near.restype = C.c_int64
        L.orc_near.argtypes = [C.c_void_p, dp, C.c_int64, ip]
        L.orc_search_goal_parent.restype = C.c_int64
        L.orc_search_goal_parent.argtypes = [C.c_void_p]
        L.orc_in_goal_region.restype = C.c_int
        L.orc_in_goal_region.argtypes = [C.c_void_p, dp]
        L.orc_best_solution.restype = C.c_int64
        L.orc_best_solution.argtypes = [C.c_void_p, dp]
        L.orc_path_len.restype = C.c_double
        L.orc_path_len.argtypes = [C.c_void_p, C.c_int64]
        L.orc_step.argtypes = [C.c_void_p, dp, C.c_int, C.POINTER(StepResult)]
        L.orc_replay.argtypes = [C.c_void_p, dp, C.c_int64, C.c_int]
        u32 = C.POINTER(C.c_uint32)
        L.orc_run_sampling.restype = C.c_int64
        L.orc_run_sampling.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64, u32, C.c_int64, u32, C.c_int64,
                                       C.c_double, dp, dp, dp, ip, ip]
        for f in ("orc_is_collision",):
            getattr(L, f).restype = C.c_int
            getattr(L, f).argtypes = [C.c_void_p, dp, dp]
        for f in ("orc_is_inside_obs", "orc_is_in_range", "orc_is_valid"):
            getattr(L, f).restype = C.c_int
            getattr(L, f).argtypes = [C.c_void_p, dp]
        L.orc_hypot_np.restype = C.c_double
        L.orc_hypot_np.argtypes = [C.c_double, C.c_double]
        L.orc_hypot_glibc_restated.restype = C.c_double
        L.orc_hypot_glibc_restated.argtypes = [C.c_double, C.c_double]
        L.orc_hypot_py.restype = C.c_double
        L.orc_hypot_py.argtypes = [C.c_int, dp]
        L.orc_norm_1d.restype = C.c_double
        L.orc_norm_1d.argtypes = [C.c_int, dp]
        L.orc_dot_blas.restype = C.c_double
        L.orc_dot_blas.argtypes = [C.c_int, dp, dp]
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _vec(x, dim=None):
    a = np.ascontiguousarray(np.asarray(x, dtype=np.float64))
    return a


def obstacles_from_env_dict(env_dict, dim):
    """(round (n, dim+1) f64, box (n, 2*dim) f64, lo, hi) from a reference-schema env_dict."""
    if dim == 2:
        rnd = np.asarray(env_dict.get("circle_obstacles", []), dtype=np.float64).reshape(-1, 3)
        box = np.asarray(env_dict.get("rectangle_obstacles", []), dtype=np.float64).reshape(-1, 4)
        h, w = env_dict["env_dims"]
        lo, hi = np.array([0.0, 0.0]), np.array([float(w), float(h)])
    else:
        rnd = np.asarray(env_dict.get("ball_obstacles", []), dtype=np.float64).reshape(-1, 4)
        box = np.asarray(env_dict.get("box_obstacles", []), dtype=np.float64).reshape(-1, 6)
        h, w, d = env_dict["env_dims"]
        lo, hi = np.zeros(3), np.array([float(w), float(h), float(d)])
    return np.ascontiguousarray(rnd), np.ascontiguousarray(box), lo, hi


class OracleTree:
    """One reference-semantics planning tree on the CPU."""

    def __init__(self, dim, iter_max, x_start, x_goal, step_len, search_radius, clearance, env_dict):
        L = lib()
        self.dim = dim
        self.iter_max = iter_max
        rnd, box, lo, hi = obstacles_from_env_dict(env_dict, dim)
        self._keep = (rnd, box, lo, hi, _vec(x_start), _vec(x_goal))
        self.h = L.orc_create(dim, iter_max, _dp(self._keep[4]), _dp(self._keep[5]), float(step_len),
                              float(search_radius), float(clearance), _dp(lo), _dp(hi), len(rnd), _dp(rnd),
                              len(box), _dp(box))
        self.L = L

    def close(self):
        if self.h:
            self.L.orc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- state ---------------------------------------------------------------------------
    @property
    def n(self):
        return int(self.L.orc_num_vertices(self.h))

    @property
    def vertices(self):
        p = self.L.orc_vertices(self.h)
        return np.ctypeslib.as_array(p, shape=(1 + self.iter_max, self.dim))[: self.n].copy()

    @property
    def parents(self):
        p = self.L.orc_parents(self.h)
        return np.ctypeslib.as_array(p, shape=(1 + self.iter_max,))[: self.n].copy()

    @property
    def solutions(self):
        ns = int(self.L.orc_num_solutions(self.h))
        if ns == 0:
            return np.zeros(0, dtype=np.int64)
        p = self.L.orc_solutions(self.h)
        return np.ctypeslib.as_array(p, shape=(ns,)).copy()

    def load(self, vertices, parents):
        v = np.ascontiguousarray(vertices, dtype=np.float64)
        p = np.ascontiguousarray(parents, dtype=np.int64)
        self.L.orc_load(self.h, len(v), _dp(v), p.ctypes.data_as(C.POINTER(C.c_int64)))

    # --- primitives ----------------------------------------------------------------------
    def is_collision(self, a, b):
        a, b = _vec(a), _vec(b)
        return bool(self.L.orc_is_collision(self.h, _dp(a), _dp(b)))

    def is_inside_obs(self, p):
        p = _vec(p)
        return bool(self.L.orc_is_inside_obs(self.h, _dp(p)))

    def is_in_range(self, p):
        p = _vec(p)
        return bool(self.L.orc_is_in_range(self.h, _dp(p)))

    def is_valid(self, p):
        p = _vec(p)
        return bool(self.L.orc_is_valid(self.h, _dp(p)))

    def cost(self, idx):
        return float(self.L.orc_cost(self.h, int(idx)))

    def nearest(self, q):
        q = _vec(q)
        return int(self.L.orc_nearest(self.h, _dp(q)))

    def steer(self, a, b):
        a, b = _vec(a), _vec(b)
        out = np.zeros(3)
        self.L.orc_steer(self.h, _dp(a), _dp(b), _dp(out))
        return out[: self.dim].copy()

    def near_radius(self, n):
        return float(self.L.orc_near_radius(self.h, int(n)))

    def near(self, node_new, new_idx):
        q = _vec(node_new)
        out = np.zeros(4096, dtype=np.int64)
        k = int(self.L.orc_near(self.h, _dp(q), int(new_idx), out.ctypes.data_as(C.POINTER(C.c_int64))))
        return out[:k].copy()

    def search_goal_parent(self):
        return int(self.L.orc_search_goal_parent(self.h))

    def in_goal_region(self, p):
        p = _vec(p)
        return bool(self.L.orc_in_goal_region(self.h, _dp(p)))

    def best_solution(self):
        c = C.c_double(0)
        x = int(self.L.orc_best_solution(self.h, C.byref(c)))
        return c.value, x

    def path_len(self, goal_parent):
        return float(self.L.orc_path_len(self.h, int(goal_parent)))

    def step(self, node_rand, irrt=False):
        q = _vec(node_rand)
        r = StepResult()
        self.L.orc_step(self.h, _dp(q), int(irrt), C.byref(r))
        return r

    def replay(self, samples, irrt=False):
        s = np.ascontiguousarray(samples, dtype=np.float64)
        self.L.orc_replay(self.h, _dp(s), len(s), int(irrt))

    def run_sampling(self, iters, np_words, py_words=None, irrt=False, goal_scan=False, stop_first=False, frame=None, want_trace=False):
        """whole loop incl. sampling from raw MT19937 word streams -> dict(iters_done, np_used, py_used, cost_trace)"""
        u32 = C.POINTER(C.c_uint32)
        npw = np.ascontiguousarray(np_words, dtype=np.uint32)
        pyw = np.ascontiguousarray(py_words if py_words is not None else np.zeros(0), dtype=np.uint32)
        c_min, xc, Cm = frame if frame is not None else (0.0, np.zeros(3), np.eye(3))
        xc3 = np.zeros(3)
        xc3[: self.dim] = np.asarray(xc, dtype=np.float64).ravel()[: self.dim]
        Cm = np.ascontiguousarray(np.asarray(Cm, dtype=np.float64).reshape(3, 3))
        trace = np.zeros(iters, dtype=np.float64) if want_trace else None
        npu, pyu = C.c_int64(0), C.c_int64(0)
        done = self.L.orc_run_sampling(self.h, int(irrt), int(goal_scan), int(stop_first), int(iters), npw.ctypes.data_as(u32), len(npw),
                                       pyw.ctypes.data_as(u32), len(pyw), float(c_min), _dp(xc3), _dp(Cm),
                                       _dp(trace) if want_trace else None, C.byref(npu), C.byref(pyu))
        return {"iters_done": int(done), "np_used": npu.value, "py_used": pyu.value, "cost_trace": trace}
