"""ctypes binding of libnirrt_hip.so (C ABI: include/nirrt_hip.h).

There is NO fallback: if the shared library is missing, or no gfx950 device is visible when a tree
is created, this raises.  (The CPU oracle under oracle/ is test infrastructure and is never
imported from here.)
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("NIRRT_HIP_SO") or os.path.join(_HERE, "libnirrt_hip.so")   # env override: A/B builds

F_IRRT = 1
F_GOAL_SCAN = 2
F_STOP_FIRST = 4
F_PNG = 8

E_ARG, E_HIP, E_CAPACITY, E_NODEVICE, E_STREAM, E_CLOUD, E_LIBM, E_PARK = -1, -2, -3, -4, -5, -6, -7, -8
ABI_VERSION = 6   # include/nirrt_hip.h NIRRT_ABI_VERSION: the structs below mirror THAT header
MAX_OBSTACLES = 64

EXPORTS = [
    "nirrt_last_error", "nirrt_device_count", "nirrt_create", "nirrt_destroy", "nirrt_reset", "nirrt_upload",
    "nirrt_download", "nirrt_num_vertices", "nirrt_nearest", "nirrt_collision_batch", "nirrt_points_in_obs",
    "nirrt_near", "nirrt_cost", "nirrt_search_goal_parent", "nirrt_best_solution", "nirrt_solutions",
    "nirrt_step", "nirrt_extend", "nirrt_run", "nirrt_set_informed", "nirrt_debug_prof", "nirrt_set_cloud", "nirrt_reset_batch", "nirrt_pool_trim", "nirrt_set_cloud_batch",
    "nirrt_mt19937_fill", "nirrt_set_generators", "nirrt_get_generators", "nirrt_generator_words", "nirrt_abi_version",
    "nirrt_create_batch", "nirrt_set_informed_batch", "nirrt_collision_each",
]


class NirrtError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [("dim", C.c_int32), ("device_id", C.c_int32), ("iter_max", C.c_int64),
                ("x_start", C.c_double * 3), ("x_goal", C.c_double * 3),
                ("step_len", C.c_double), ("search_radius", C.c_double), ("clearance", C.c_double),
                ("range_lo", C.c_double * 3), ("range_hi", C.c_double * 3),
                ("n_round", C.c_int32), ("round_obs", C.POINTER(C.c_double)),
                ("n_box", C.c_int32), ("box_obs", C.POINTER(C.c_double))]


class StepResult(C.Structure):
    _fields_ = [("collided", C.c_int32), ("inserted", C.c_int32), ("nearest_idx", C.c_int64),
                ("new_idx", C.c_int64), ("n_near", C.c_int32), ("reparented", C.c_int32),
                ("n_rewired", C.c_int32), ("in_goal", C.c_int32), ("n", C.c_int64),
                ("node_new", C.c_double * 3), ("c_best", C.c_double), ("x_best", C.c_int64),
                ("n_solutions", C.c_int64), ("status", C.c_int32), ("reserved", C.c_int32)]


class RunArgs(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("flags", C.c_uint32), ("inputs_on_device", C.c_int32), ("park_limit", C.c_int32), ("iters", C.c_int64),
                ("samples", C.POINTER(C.c_double)),
                ("np_words", C.POINTER(C.POINTER(C.c_uint32))), ("n_np", C.POINTER(C.c_int64)),
                ("py_words", C.POINTER(C.POINTER(C.c_uint32))), ("n_py", C.POINTER(C.c_int64)),
                ("cost_trace", C.POINTER(C.c_double)), ("np_used", C.POINTER(C.c_int64)),
                ("py_used", C.POINTER(C.c_int64)), ("iters_done", C.POINTER(C.c_int64)),
                ("status", C.POINTER(C.c_int32)), ("kernel_ms", C.POINTER(C.c_double)),
                ("scan_elems", C.POINTER(C.c_int64)), ("alg_elems", C.POINTER(C.c_int64)),
                ("stats", C.POINTER(C.c_int64)), ("iters_each", C.POINTER(C.c_int64)), ("lanes_hint", C.POINTER(C.c_int32)),
                ("slice_iters", C.c_int64), ("run_ahead", C.POINTER(C.c_int32))]

N_STATS = 24
ST_ITERS, ST_T0, ST_T1, ST_CBEST, ST_BUSY = 13, 14, 15, 17, 20   # slots 14 / 15 / 17 are absolute values of a launch, the rest are deltas
STAT_NAMES = ["visited", "visit_bytes", "members", "spilled", "hop_records", "rewire_candidates", "rewired", "recosted",
              "list_entries", "inserted", "rebuilt", "revisits", "whole_tree_visits", "iterations", "t0_ticks", "t1_ticks",
              "alg_elems", "c_best_bits", "rewire_rounds", "rewired_one_by_one", "busy_ticks", "solution_entries", "r22", "r23"]


def useful_bytes(stats, dim):
    """Bytes the IMPLEMENTED algorithm has to move for what a launch did (per tree or summed), from the kernel's own
    counters (include/nirrt_hip.h, nirrt_run_args.stats): visited slot records (28 B in 2D, 36 B in 3D since round 6 - packed
    records with the vertex index inside; already in bytes), 96 B per tree record walked (the hop part: eight hops each), one
    32-byte vertex record + 20 B of its tree record (links, parent) per rewire candidate examined, 16 B written per re-costed vertex
    (its record's and its slot's cost), 12 B + one 32-byte record per re-evaluated goal-candidate list entry, 8 B per re-evaluated
    solution list entry (its cached cost + Line(v, goal): round 6; 44 B before), per inserted vertex the vertex record (32), the tree
    record (112), the slot record and two link words (8) written, per vertex of an index rebuild 32 B read + the slot record + 4
    (its slot number) + 8 (rank) written."""
    st = np.asarray(stats, dtype=np.float64).reshape(-1, N_STATS).sum(axis=0)
    slot = 36 if dim == 3 else 28
    return float(st[1] + 96 * st[4] + 52 * st[5] + 16 * st[7] + 44 * st[8] + 8 * st[21] + (32 + 112 + slot + 8) * st[9] + (32 + slot + 4 + 8) * st[10])


_lib = None


def load():
    """Load libnirrt_hip.so; raises NirrtError (never falls back) when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise NirrtError("libnirrt_hip.so is not built (%s). Run `python -m nirrt_star_amd.build` "
                         "(hipcc --offload-arch=gfx950). There is no CPU fallback." % SO_PATH)
    # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64 / libhsa-runtime64 (same SONAMEs as /opt/rocm's).
    # Loaded first, torch's copies also satisfy this library's DT_NEEDED entries; loaded second they would be a SECOND runtime,
    # whose HSA finds no GPU ("No HIP GPUs are available" from the first torch.cuda call after a tree was created).  The
    # guidance path needs torch anyway (device memory, PointNet++); a process without torch just loads /opt/rocm's runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(SO_PATH)
    dp, ip, up = C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_uint8)
    vp = C.c_void_p
    L.nirrt_last_error.restype = C.c_char_p
    L.nirrt_device_count.argtypes = [C.POINTER(C.c_int)]
    L.nirrt_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.nirrt_destroy.argtypes = [vp]
    L.nirrt_reset.argtypes = [vp]
    L.nirrt_reset_batch.argtypes = [C.POINTER(vp), C.c_int32]
    L.nirrt_pool_trim.argtypes = []
    L.nirrt_mt19937_fill.argtypes = [vp, C.POINTER(C.c_int32), C.c_int64, vp]
    L.nirrt_set_generators.argtypes = [C.POINTER(vp), C.c_int32, vp, vp, vp, vp]
    L.nirrt_get_generators.argtypes = [C.POINTER(vp), C.c_int32, vp, vp, vp, vp]
    L.nirrt_generator_words.argtypes = [C.POINTER(vp), C.c_int32, C.c_int32, C.c_int64, vp, C.c_int64, C.c_int32]
    L.nirrt_set_cloud_batch.argtypes = [C.POINTER(vp), C.c_int32, vp, C.c_int64, C.POINTER(C.c_int32), vp, C.c_int64, C.c_double,
                                        C.c_double, dp, C.POINTER(C.c_int32)]
    L.nirrt_upload.argtypes = [vp, C.c_int64, dp, ip]
    L.nirrt_download.argtypes = [vp, dp, ip, ip]
    L.nirrt_num_vertices.argtypes = [vp, ip]
    L.nirrt_nearest.argtypes = [vp, dp, ip]
    L.nirrt_collision_batch.argtypes = [vp, C.c_int64, dp, up]
    L.nirrt_points_in_obs.argtypes = [vp, C.c_int64, dp, up, up]
    L.nirrt_near.argtypes = [vp, dp, C.c_int64, ip, ip, C.c_int64]
    L.nirrt_cost.argtypes = [vp, C.c_int64, ip, dp]
    L.nirrt_search_goal_parent.argtypes = [vp, ip, dp]
    L.nirrt_best_solution.argtypes = [vp, dp, ip]
    L.nirrt_solutions.argtypes = [vp, ip, ip, C.c_int64]
    L.nirrt_step.argtypes = [vp, dp, C.c_uint32, C.POINTER(StepResult)]
    L.nirrt_extend.argtypes = [vp, C.c_int64, dp, C.c_uint32, C.POINTER(StepResult)]
    L.nirrt_run.argtypes = [C.POINTER(vp), C.c_int32, C.POINTER(RunArgs)]
    L.nirrt_set_informed.argtypes = [vp, C.c_double, dp, dp]
    L.nirrt_debug_prof.argtypes = [vp, ip]
    L.nirrt_set_cloud.argtypes = [vp, C.c_int64, dp, C.c_double, C.c_double, C.c_double]
    L.nirrt_abi_version.argtypes = []
    L.nirrt_create_batch.argtypes = [C.POINTER(Config), C.c_int32, C.POINTER(vp)]
    L.nirrt_set_informed_batch.argtypes = [C.POINTER(vp), C.c_int32, dp, dp, dp]
    L.nirrt_collision_each.argtypes = [C.POINTER(vp), C.c_int32, dp, up]
    L.nirrt_libm_probe.argtypes = [C.c_int32, C.c_int64, dp, dp, dp, C.c_int]   # (include/nirrt_pointops.h)
    L.nirrt_libm_probe.restype = C.c_int
    for name in EXPORTS:
        if name != "nirrt_last_error":
            getattr(L, name).restype = C.c_int
    if L.nirrt_abi_version() != ABI_VERSION:
        raise NirrtError("%s was built from another include/nirrt_hip.h (NIRRT_ABI_VERSION %d, this binding mirrors %d): rebuild it with "
                         "`python -m nirrt_star_amd.build --force`" % (SO_PATH, L.nirrt_abi_version(), ABI_VERSION))
    _lib = L
    return L


def _run_args():
    a = RunArgs()   # (zero-initialised)
    a.struct_size = C.sizeof(RunArgs)
    return a


# ---- does this host's libm compute what the device's restatement computes? ----------------------------------------------------
# The reference steers with math.atan2 / math.cos / math.sin (rrt_star_2d.py:67-78) and samples the 3D unit ball with np.sin /
# np.cos (irrt_star_3d.py:146-158): whatever libm its host has.  The device evaluates glibc 2.35's x86-64 FMA variants, restated
# (csrc/glibc235_libm.inc).  On a host with another libm / CPU the REFERENCE computes other last bits, and the device-resident
# loop would no longer be bit-identical to it: the planner classes then default to mode="exact" (steer / sampling with the host's
# libm between two launches).  Checked once per process, the first time a planner asks (needs a GPU).
_libm_ok = None


def libm_check(device_id=0, n=4096, force=False):
    """True if the device-side atan2 / sin / cos equal this host's math.atan2 / math.sin / math.cos bit for bit on n random steer
    arguments each (+ the quadrant boundaries); False (with ONE RuntimeWarning) otherwise.  NIRRT_LIBM_CHECK=0 skips the probe
    (returns True); the result is cached."""
    global _libm_ok
    if _libm_ok is not None and not force:
        return _libm_ok
    if os.environ.get("NIRRT_LIBM_CHECK", "1") == "0":
        _libm_ok = True
        return True
    import math
    import warnings
    L = load()
    rng = np.random.RandomState(20260930)
    dy, dx = rng.uniform(-224.0, 224.0, n), rng.uniform(-224.0, 224.0, n)
    dy[:8] = [0.0, 0.0, 1.0, -1.0, 5.0, -5.0, 1e-300, -0.0]
    dx[:8] = [1.0, -1.0, 0.0, 0.0, 5.0, -5.0, 1.0, -1.0]
    ang = np.concatenate([rng.uniform(-math.pi, math.pi, n - 64), rng.uniform(0.0, 2 * math.pi, 32), rng.uniform(-1e-6, 1e-6, 32)])
    bad = []
    for fn, name, a, b, host in ((0, "atan2", dy, dx, lambda i: math.atan2(dy[i], dx[i])),
                                 (1, "sin", ang, ang, lambda i: math.sin(ang[i])),
                                 (2, "cos", ang, ang, lambda i: math.cos(ang[i]))):
        out = np.zeros(n, dtype=np.float64)
        a_, b_ = _f64(a), _f64(b)
        rc = L.nirrt_libm_probe(fn, n, _dp(a_), _dp(b_), _dp(out), int(device_id))
        if rc != 0:
            raise NirrtError("nirrt_libm_probe failed (%d)" % rc)
        ref = np.array([host(i) for i in range(n)], dtype=np.float64)
        diff = int(np.count_nonzero(out.view(np.uint64) != ref.view(np.uint64)))
        if diff:
            bad.append("%s: %d of %d" % (name, diff, n))
    _libm_ok = not bad
    if bad:
        warnings.warn("nirrt_star_amd: this host's libm differs from the one the device code restates (glibc 2.35, x86-64, FMA) - "
                      + "; ".join(bad) + " arguments give other bits.  The reference itself computes this host's values, so the "
                      "planner classes now default to mode='exact' (steer and sampling on the host, bit-identical, slower); "
                      "scripts/libm_port/make_inc.sh regenerates the restatement from another libm.", RuntimeWarning, stacklevel=2)
    return _libm_ok


def device_count():
    n = C.c_int(0)
    load().nirrt_device_count(C.byref(n))
    return n.value


def _check(rc):
    if rc != 0:
        msg = load().nirrt_last_error().decode(errors="replace")
        raise NirrtError("libnirrt_hip error %d: %s" % (rc, msg))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def _f64(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.float64))


def obstacle_tables(env, dim):
    """(round (n,dim+1), box (n,2*dim), lo, hi) float64 from an Env / Env3D-like object
    (rrt_utils_2d.py:5-17 / rrt_utils_3d.py:7-20 build the same tables)."""
    if dim == 2:
        rnd = np.asarray(env.obs_circle, dtype=np.float64).reshape(-1, 3)
        box = np.asarray(env.obs_rectangle, dtype=np.float64).reshape(-1, 4)
        lo = np.array([env.x_range[0], env.y_range[0]], dtype=np.float64)
        hi = np.array([env.x_range[1], env.y_range[1]], dtype=np.float64)
    else:
        rnd = np.asarray(env.obs_ball, dtype=np.float64).reshape(-1, 4)
        box = np.asarray(env.obs_box, dtype=np.float64).reshape(-1, 6)
        lo = np.array([env.x_range[0], env.y_range[0], env.z_range[0]], dtype=np.float64)
        hi = np.array([env.x_range[1], env.y_range[1], env.z_range[1]], dtype=np.float64)
    return np.ascontiguousarray(rnd), np.ascontiguousarray(box), lo, hi


def _fill_config(cfg, dim, iter_max, x_start, x_goal, step_len, search_radius, clearance, env, device_id):
    """nirrt_config of one problem; returns the arrays its pointers refer to (to be kept alive until the create call returns)"""
    rnd, box, lo, hi = obstacle_tables(env, dim)
    cfg.dim = int(dim)
    cfg.device_id = int(device_id)
    cfg.iter_max = int(iter_max)
    xs, xg = _f64(x_start), _f64(x_goal)
    for k in range(dim):
        cfg.x_start[k] = xs[k]
        cfg.x_goal[k] = xg[k]
        cfg.range_lo[k] = lo[k]
        cfg.range_hi[k] = hi[k]
    cfg.step_len = float(step_len)
    cfg.search_radius = float(search_radius)
    cfg.clearance = float(clearance)
    cfg.n_round = len(rnd)
    cfg.round_obs = _dp(rnd) if len(rnd) else None
    cfg.n_box = len(box)
    cfg.box_obs = _dp(box) if len(box) else None
    return rnd, box


def create_trees(dim, iter_max, specs, device_id=0):
    """One tree per problem of `specs` = [(x_start, x_goal, step_len, search_radius, clearance, env), ...] through nirrt_create_batch:
    host work per tree, ONE device pass for the batch (8192 trees: ~1 s instead of the 5 s of 8192 nirrt_create calls)."""
    L = load()
    n = len(specs)
    if n == 0:
        return []
    cfgs = (Config * n)()
    keep = []
    for i, (xs, xg, step_len, radius, clearance, env) in enumerate(specs):
        keep.append(_fill_config(cfgs[i], int(dim), int(iter_max), xs, xg, step_len, radius, clearance, env, device_id))
    handles = (C.c_void_p * n)()
    _check(L.nirrt_create_batch(cfgs, n, handles))
    return [HipTree(dim, iter_max, None, None, None, None, None, None, device_id=device_id, _handle=C.c_void_p(handles[i])) for i in range(n)]


def set_informed_batch(trees, frames):
    """IRRTStar.init for a batch: frames[i] = (c_min, x_center, C) of sampling.informed_frame - one copy, one launch"""
    nt = len(trees)
    if nt == 0:
        return
    cm = np.zeros(nt)
    xc = np.zeros((nt, 3))
    Cm = np.zeros((nt, 9))
    for i, (c_min, x_center, Cmat) in enumerate(frames):
        cm[i] = float(c_min)
        v = np.asarray(x_center, dtype=np.float64).ravel()
        xc[i, : trees[i].dim] = v[: trees[i].dim]
        Cm[i] = np.asarray(Cmat, dtype=np.float64).reshape(9)
    handles = (C.c_void_p * nt)(*[t.h for t in trees])
    _check(load().nirrt_set_informed_batch(handles, nt, _dp(cm), _dp(xc), _dp(Cm)))


def collision_each(trees, segs):
    """Utils.is_collision of ONE segment per tree (segs: (n_trees, 2, dim)), each against its own tree's obstacles, in one launch:
    the "is the straight start-goal segment free?" probes of a batch"""
    nt = len(trees)
    if nt == 0:
        return np.zeros(0, dtype=bool)
    seg = _f64(segs).reshape(nt, 2, trees[0].dim)
    out = np.zeros(nt, dtype=np.uint8)
    handles = (C.c_void_p * nt)(*[t.h for t in trees])
    _check(load().nirrt_collision_each(handles, nt, _dp(seg), out.ctypes.data_as(C.POINTER(C.c_uint8))))
    return out.astype(bool)


class HipTree:
    """One planning tree resident in HBM (opaque nirrt_tree handle)."""

    def __init__(self, dim, iter_max, x_start, x_goal, step_len, search_radius, clearance, env, device_id=0, _handle=None):
        L = load()
        self.L = L
        self.dim = int(dim)
        self.iter_max = int(iter_max)
        self.device_id = int(device_id)
        self._res = StepResult()
        if _handle is not None:     # (create_trees: the handle comes out of nirrt_create_batch)
            self.h = _handle
            return
        cfg = Config()
        self._keep = _fill_config(cfg, self.dim, self.iter_max, x_start, x_goal, step_len, search_radius, clearance, env, device_id)
        h = C.c_void_p()
        _check(L.nirrt_create(C.byref(cfg), C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            # whoever parked generator states in this tree (batch.ProblemStreams) takes them back before the tree goes
            for hook in list(getattr(self, "_release_hooks", ())):
                try:
                    import weakref
                    fn = hook() if isinstance(hook, weakref.WeakMethod) else hook   # (batch.ProblemStreams registers weak references)
                    if fn is not None:
                        fn()
                except Exception:
                    pass
            self._release_hooks = []
            self.L.nirrt_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- state ----
    def reset(self):
        _check(self.L.nirrt_reset(self.h))

    def upload(self, vertices, parents):
        v = _f64(vertices)
        p = np.ascontiguousarray(parents, dtype=np.int64)
        _check(self.L.nirrt_upload(self.h, len(v), _dp(v), _ip(p)))

    @property
    def n(self):
        n = C.c_int64(0)
        _check(self.L.nirrt_num_vertices(self.h, C.byref(n)))
        return n.value

    def download(self):
        n = self.n
        v = np.zeros((n, self.dim), dtype=np.float64)
        p = np.zeros(n, dtype=np.int64)
        nn = C.c_int64(0)
        _check(self.L.nirrt_download(self.h, _dp(v), _ip(p), C.byref(nn)))
        return v, p

    def download_into(self, vertices, parents):
        """fill caller arrays sized (>=n, dim) / (>=n,) - the reference's preallocated buffers"""
        nn = C.c_int64(0)
        _check(self.L.nirrt_download(self.h, _dp(vertices), _ip(parents), C.byref(nn)))
        return nn.value

    @property
    def solutions(self):
        ns = C.c_int64(0)
        _check(self.L.nirrt_solutions(self.h, C.byref(ns), None, 0))
        out = np.zeros(ns.value, dtype=np.int64)
        if ns.value:
            _check(self.L.nirrt_solutions(self.h, C.byref(ns), _ip(out), len(out)))
        return out

    # ---- primitives ----
    def nearest(self, q):
        q = _f64(q)
        i = C.c_int64(0)
        _check(self.L.nirrt_nearest(self.h, _dp(q), C.byref(i)))
        return i.value

    def collision_batch(self, seg):
        seg = _f64(seg).reshape(-1, 2, self.dim)
        out = np.zeros(len(seg), dtype=np.uint8)
        _check(self.L.nirrt_collision_batch(self.h, len(seg), _dp(seg), out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out

    def is_collision(self, a, b):
        return bool(self.collision_batch(np.stack([_f64(a), _f64(b)])[None])[0])

    def points_in_obs(self, pts):
        pts = _f64(pts).reshape(-1, self.dim)
        ins = np.zeros(len(pts), dtype=np.uint8)
        val = np.zeros(len(pts), dtype=np.uint8)
        u8 = C.POINTER(C.c_uint8)
        _check(self.L.nirrt_points_in_obs(self.h, len(pts), _dp(pts), ins.ctypes.data_as(u8), val.ctypes.data_as(u8)))
        return ins, val

    def near(self, node_new, new_idx):
        q = _f64(node_new)
        k = C.c_int64(0)
        out = np.zeros(self.iter_max + 1, dtype=np.int64)   # a Near set can never exceed the tree
        _check(self.L.nirrt_near(self.h, _dp(q), int(new_idx), C.byref(k), _ip(out), len(out)))
        return out[:k.value].copy()

    def cost(self, idx):
        idx = np.ascontiguousarray(np.atleast_1d(idx), dtype=np.int64)
        out = np.zeros(len(idx), dtype=np.float64)
        _check(self.L.nirrt_cost(self.h, len(idx), _ip(idx), _dp(out)))
        return out

    def search_goal_parent(self):
        i = C.c_int64(0)
        d = C.c_double(0)
        _check(self.L.nirrt_search_goal_parent(self.h, C.byref(i), C.byref(d)))
        return i.value, d.value

    def best_solution(self):
        i = C.c_int64(0)
        d = C.c_double(0)
        _check(self.L.nirrt_best_solution(self.h, C.byref(d), C.byref(i)))
        return d.value, i.value

    def set_cloud(self, pts, sample_rate, update_cost_ratio, c_update):
        pts = _f64(pts).reshape(-1, self.dim)
        _check(self.L.nirrt_set_cloud(self.h, len(pts), _dp(pts) if len(pts) else None, float(sample_rate),
                                      float(update_cost_ratio), float(c_update)))

    def debug_prof(self):
        out = np.zeros(24, dtype=np.int64)
        _check(self.L.nirrt_debug_prof(self.h, _ip(out)))
        return out

    def set_informed(self, c_min, x_center, C):
        xc = np.zeros(3)
        xc[: self.dim] = np.asarray(x_center, dtype=np.float64).ravel()[: self.dim]
        Cm = _f64(np.asarray(C, dtype=np.float64).reshape(3, 3))
        _check(self.L.nirrt_set_informed(self.h, float(c_min), _dp(xc), _dp(Cm)))

    # ---- whole iteration ----
    def step(self, node_rand, flags=0):
        q = _f64(node_rand)
        _check(self.L.nirrt_step(self.h, _dp(q), int(flags), C.byref(self._res)))
        return self._res

    def extend(self, nearest_idx, node_new, flags=0):
        q = _f64(node_new)
        _check(self.L.nirrt_extend(self.h, int(nearest_idx), _dp(q), int(flags), C.byref(self._res)))
        return self._res


def set_cloud_batch(trees, clouds_ptr, cloud_stride, n_points, pred_ptr, pred_stride, sample_rate, update_cost_ratio, c_update):
    """path points of a batch of clouds into their trees in one launch (device pointers); returns the points kept per tree"""
    nt = len(trees)
    handles = (C.c_void_p * nt)(*[t.h for t in trees])
    npts = np.ascontiguousarray(n_points, dtype=np.int32)
    cu = np.ascontiguousarray(c_update, dtype=np.float64)
    out = np.zeros(nt, dtype=np.int32)
    _check(load().nirrt_set_cloud_batch(handles, nt, C.c_void_p(int(clouds_ptr)), int(cloud_stride), npts.ctypes.data_as(C.POINTER(C.c_int32)),
                                        C.c_void_p(int(pred_ptr)), int(pred_stride), float(sample_rate), float(update_cost_ratio),
                                        _dp(cu), out.ctypes.data_as(C.POINTER(C.c_int32))))
    return out


def mt19937_outputs(key, pos, n):
    """the next n raw outputs of an MT19937 generator in state (key[624], pos) -> (outputs uint32 (n,), key after, pos after);
    numpy's RandomState.randint(0, 2**32, dtype=uint32) produces the same words 10-30x slower (nirrt_mt19937_fill, host code)"""
    k = np.array(key, dtype=np.uint32, copy=True)
    if k.shape != (624,):
        raise ValueError("MT19937 state: 624 key words expected")
    p = C.c_int32(int(pos))
    out = np.empty(int(n), dtype=np.uint32)
    _check(load().nirrt_mt19937_fill(k.ctypes.data, C.byref(p), int(n), out.ctypes.data))
    return out, k, int(p.value)


def np_state(rs=None):
    """(key uint32[624], pos) of a numpy legacy RandomState (default: the process-global one the reference draws from)"""
    st = (rs if rs is not None else np.random).get_state()
    return np.ascontiguousarray(st[1], dtype=np.uint32), int(st[2])


def py_state(rnd=None):
    """(key uint32[624], pos) of a CPython random.Random (default: the module-level generator)"""
    import random
    st = (rnd if rnd is not None else random).getstate()[1]
    return np.array(st[:624], dtype=np.uint32), int(st[624])


def set_np_state(key, pos, rs=None):
    tgt = rs if rs is not None else np.random
    st = tgt.get_state()
    tgt.set_state((st[0], np.asarray(key, dtype=np.uint32), int(pos), st[3], st[4]))


def set_py_state(key, pos, rnd=None):
    import random
    tgt = rnd if rnd is not None else random
    st = tgt.getstate()
    tgt.setstate((st[0], tuple(int(v) for v in key) + (int(pos),), st[2]))


def set_generators(trees, np_states=None, py_states=None):
    """np.random.set_state / random.setstate for the trees' own generators: per-tree (key[624], pos) pairs; None leaves a
    stream alone"""
    nt = len(trees)
    handles = (C.c_void_p * nt)(*[t.h for t in trees])

    def tab(states):
        if states is None:
            return None, None, None, None
        k = np.ascontiguousarray(np.stack([np.asarray(s[0], dtype=np.uint32) for s in states]))
        p = np.array([int(s[1]) for s in states], dtype=np.int32)
        assert k.shape == (nt, 624)
        return k, p, k.ctypes.data, p.ctypes.data

    nk, npos, nka, npa = tab(np_states)
    pk, ppos, pka, ppa = tab(py_states)
    _check(load().nirrt_set_generators(handles, nt, nka, npa, pka, ppa))


def get_generators(trees, want_np=True, want_py=True):
    """-> (np_keys (nt, 624), np_pos (nt,), py_keys, py_pos): the generators' states as get_state() / getstate() would show
    them after what the device consumed"""
    nt = len(trees)
    handles = (C.c_void_p * nt)(*[t.h for t in trees])
    nk = np.zeros((nt, 624), dtype=np.uint32) if want_np else None
    npos = np.zeros(nt, dtype=np.int32) if want_np else None
    pk = np.zeros((nt, 624), dtype=np.uint32) if want_py else None
    ppos = np.zeros(nt, dtype=np.int32) if want_py else None
    _check(load().nirrt_get_generators(handles, nt, nk.ctypes.data if want_np else None, npos.ctypes.data if want_np else None,
                                       pk.ctypes.data if want_py else None, ppos.ctypes.data if want_py else None))
    return nk, npos, pk, ppos


def generator_words(trees, which, n_words, device_ptr=None, stride=None):
    """the next n_words raw outputs of every tree's numpy (which=0) / python (which=1) generator, produced on the device and
    consumed; returns a (nt, n_words) uint32 array, or writes to device memory (device_ptr, stride in words) and returns None"""
    nt = len(trees)
    handles = (C.c_void_p * nt)(*[t.h for t in trees])
    if device_ptr is None:
        out = np.zeros((nt, int(n_words)), dtype=np.uint32)
        _check(load().nirrt_generator_words(handles, nt, int(which), int(n_words), out.ctypes.data, int(n_words), 0))
        return out
    _check(load().nirrt_generator_words(handles, nt, int(which), int(n_words), C.c_void_p(int(device_ptr)),
                                        int(stride if stride is not None else n_words), 1))
    return None


def pool_trim():
    """device chunks that hold no live tree go back to the driver (see nirrt_pool_trim)"""
    _check(load().nirrt_pool_trim())


def reset_batch(trees):
    """all trees back to their single start vertex in one launch (one workgroup per tree)"""
    nt = len(trees)
    handles = (C.c_void_p * nt)(*[t.h for t in trees])
    _check(load().nirrt_reset_batch(handles, nt))


def run_replay(trees, samples, flags=0, want_trace=False, device_ptr=None, iters=None):
    """Device-resident loop over many trees with replayed samples (n_trees, iters, dim).
    `samples` is a host array, or pass device_ptr (int address of an (n_trees, iters, dim) f64
    buffer already in HBM, e.g. torch tensor.data_ptr()) together with iters.
    Returns dict(iters_done, status, kernel_ms, cost_trace, scan_elems)."""
    L = load()
    nt = len(trees)
    if device_ptr is None:
        samples = _f64(samples)
        assert samples.ndim == 3 and samples.shape[0] == nt and samples.shape[2] == trees[0].dim
        iters = samples.shape[1]
    handles = (C.c_void_p * nt)(*[t.h for t in trees])
    done = np.zeros(nt, dtype=np.int64)
    status = np.zeros(nt, dtype=np.int32)
    ms = C.c_double(0)
    trace = np.zeros((nt, iters), dtype=np.float64) if want_trace else None
    a = _run_args()
    a.flags = int(flags)
    a.iters = iters
    if device_ptr is None:
        a.samples = _dp(samples)
    else:
        a.inputs_on_device = 1
        a.samples = C.cast(C.c_void_p(int(device_ptr)), C.POINTER(C.c_double))
    scan = np.zeros(nt, dtype=np.int64)
    alg = np.zeros(nt, dtype=np.int64)
    stats = np.zeros((nt, N_STATS), dtype=np.int64)
    a.scan_elems = _ip(scan)
    a.alg_elems = _ip(alg)
    a.stats = _ip(stats)
    a.cost_trace = _dp(trace) if want_trace else None
    a.iters_done = _ip(done)
    a.status = status.ctypes.data_as(C.POINTER(C.c_int32))
    a.kernel_ms = C.pointer(ms)
    rc = L.nirrt_run(handles, nt, C.byref(a))
    if rc != E_CAPACITY:        # a full tree is reported per tree in status[] (the reference raises IndexError there)
        _check(rc)
    return {"iters_done": done, "status": status, "kernel_ms": ms.value, "cost_trace": trace, "scan_elems": scan,
            "alg_elems": alg, "stats": stats}


def run_sampling(trees, iters, np_words=None, py_words=None, flags=0, want_trace=False, on_device=False, iters_each=None, lanes_hint=None,
                 slice_iters=0, run_ahead=None, park_limit=0):
    """Device-resident loop with in-kernel sampling.  np_words = None (the normal case): every tree draws from its own
    generators resident in HBM (set_generators / get_generators).  Otherwise np_words / py_words: per-tree uint32 arrays of
    raw MT19937 outputs (numpy legacy global stream / python `random`); with on_device=True they are
    per-tree (device_address, n_words) pairs of buffers already resident in HBM (e.g. slices of a
    torch.cuda tensor).  Returns dict with iters_done, np_used, py_used (words consumed), status
    (0 | E_STREAM | E_CAPACITY), kernel_ms, cost_trace, scan_elems."""
    L = load()
    nt = len(trees)
    handles = (C.c_void_p * nt)(*[t.h for t in trees])
    u32p = C.POINTER(C.c_uint32)
    keep = []

    def table(words):
        if on_device:
            ptrs = (u32p * nt)(*[C.cast(C.c_void_p(int(p)), u32p) for p, _ in words])
            cnt = np.array([int(c) for _, c in words], dtype=np.int64)
        else:
            arrs = [np.ascontiguousarray(w, dtype=np.uint32) for w in words]
            keep.append(arrs)
            ptrs = (u32p * nt)(*[w.ctypes.data_as(u32p) for w in arrs])
            cnt = np.array([len(w) for w in arrs], dtype=np.int64)
        keep.append((ptrs, cnt))
        return C.cast(ptrs, C.POINTER(u32p)), _ip(cnt)

    a = _run_args()
    a.flags = int(flags)
    a.inputs_on_device = 1 if on_device else 0
    a.iters = int(iters)
    if iters_each is not None:     # per-tree budgets <= iters (trees resumed after stopping at different iterations)
        each = np.ascontiguousarray(iters_each, dtype=np.int64)
        assert len(each) == nt
        keep.append(each)
        a.iters_each = _ip(each)
    if lanes_hint is not None:     # per-tree workgroup size (0 / 64 / 128 / 256): heavy trees on wider workgroups, launched concurrently
        hint = np.ascontiguousarray(lanes_hint, dtype=np.int32)
        assert len(hint) == nt
        keep.append(hint)
        a.lanes_hint = hint.ctypes.data_as(C.POINTER(C.c_int32))
    if run_ahead is not None:      # per-tree flag: a tree known to be long never waits for its turn in a time-sliced launch
        ahead = np.ascontiguousarray(run_ahead, dtype=np.int32)
        assert len(ahead) == nt
        keep.append(ahead)
        a.run_ahead = ahead.ctypes.data_as(C.POINTER(C.c_int32))
    a.samples = None
    a.park_limit = int(park_limit)       # guided runs: the launch ends once this many trees wait for a cloud refresh (status E_PARK for the others)
    a.slice_iters = int(slice_iters)     # 0: the library decides whether / how to time-slice a batch larger than the GPU; < 0: never
    if np_words is not None:      # None: the trees' own generators (set_generators) produce the words on the device
        a.np_words, a.n_np = table(np_words)
        if py_words is not None:
            a.py_words, a.n_py = table(py_words)
    done = np.zeros(nt, dtype=np.int64)
    np_used = np.zeros(nt, dtype=np.int64)
    py_used = np.zeros(nt, dtype=np.int64)
    status = np.zeros(nt, dtype=np.int32)
    scan = np.zeros(nt, dtype=np.int64)
    alg = np.zeros(nt, dtype=np.int64)
    ms = C.c_double(0)
    stats = np.zeros((nt, N_STATS), dtype=np.int64)
    a.stats = _ip(stats)
    trace = np.zeros((nt, iters), dtype=np.float64) if want_trace else None
    a.cost_trace = _dp(trace) if want_trace else None
    a.alg_elems = _ip(alg)
    a.np_used, a.py_used, a.iters_done = _ip(np_used), _ip(py_used), _ip(done)
    a.status = status.ctypes.data_as(C.POINTER(C.c_int32))
    a.kernel_ms = C.pointer(ms)
    a.scan_elems = _ip(scan)
    rc = L.nirrt_run(handles, nt, C.byref(a))
    if rc != E_CAPACITY:        # a full tree is reported per tree in status[] (the reference raises IndexError there)
        _check(rc)
    return {"iters_done": done, "np_used": np_used, "py_used": py_used, "status": status, "kernel_ms": ms.value,
            "cost_trace": trace, "scan_elems": scan, "alg_elems": alg, "stats": stats}
