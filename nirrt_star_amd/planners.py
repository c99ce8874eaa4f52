"""Planner classes with the reference's constructor / method signatures, backed by libnirrt_hip.so.

Mirrors (same names, argument meaning, public state, return values):
    RRTStar2D / RRTStar3D         path_planning_classes{,_3d}/rrt_star_{2d,3d}.py + rrt_base_*.py
    IRRTStar2D / IRRTStar3D       path_planning_classes{,_3d}/irrt_star_{2d,3d}.py
    NIRRTStarPNG2D / ...3D        path_planning_classes{,_3d}/nirrt_star_png_{2d,3d}.py
    NIRRTStarPNGC2D / ...3D       path_planning_classes{,_3d}/nirrt_star_png_c_{2d,3d}.py
Public state after planning, like the reference: .vertices ((1+iter_max, D) f64), .vertex_parents
(int), .num_vertices, .path, .path_solutions.

Where the loop runs (``mode``):
  "resident" (default for RRT*/IRRT*): the whole `for k in range(iter_max)` runs in ONE persistent
      kernel; sampling happens in the kernel from the process-global numpy / python generators, whose
      MT19937 states move into the tree for the launch (twist and tempering run on the device) and
      back afterwards - `np.random.seed(s); random.seed(s)` therefore mean what they mean for the
      reference, and get_state() afterwards shows what the reference's own loop would have left.  NIRRT* runs the same way: its sampling policy (cloud / informed / free) is in
      the kernel too, which returns to the host only when the guidance cloud is due for a PointNet++ refresh.
  "step": host Python loop; sampling with the reference's own numpy / random calls, one fused HIP
      kernel per iteration (nearest -> steer -> ... -> rewire).
  "exact": like "step" but nearest and the rest are two launches with new_state() evaluated on the
      host by glibc in between: 2D vertices bit-identical to the reference (3D already is).
"""
import math
import random

import numpy as np

from . import _hip, sampling
from .env import Env, Env3D

_PRINT_EVERY = 1000


class _HipPlanner:
    dim = 2
    algo_flags = 0
    default_mode = "resident"

    def _base_init(self, x_start, x_goal, step_len, search_radius, iter_max, env, clearance, name, mode=None,
                   device_id=0):
        D = self.dim
        self.x_start = np.array(x_start).astype(np.float64)
        self.x_goal = np.array(x_goal).astype(np.float64)
        self.step_len = step_len
        self.search_radius = search_radius
        self.iter_max = iter_max
        self.vertices = np.zeros((1 + iter_max, D))
        self.vertex_parents = np.zeros(1 + iter_max).astype(int)
        self.vertices[0] = self.x_start
        self.num_vertices = 1
        self.path = []
        self.env = env
        self.clearance = clearance
        self.x_range = env.x_range
        self.y_range = env.y_range
        if D == 3:
            self.z_range = env.z_range
        self.path_planner_name = name
        self.device_id = device_id
        self.tree = _hip.HipTree(D, iter_max, self.x_start, self.x_goal, step_len, search_radius, clearance, env,
                                 device_id=device_id)
        # The device-resident loop evaluates glibc 2.35's atan2 / cos / sin (2D steer, 3D informed sampling); on a host with another
        # libm the reference computes other last bits, and only steer / sampling on THIS host reproduce it: mode "exact".
        # (3D RRT* needs IEEE operations only.)
        uses_libm = D == 2 or isinstance(self, _IRRTStar)
        self.mode = mode or (self.default_mode if (not uses_libm or _hip.libm_check(device_id)) else "exact")
        self.utils = _Utils(self)
        self.last_kernel_ms = 0.0

    # ---- reference helpers -------------------------------------------------------------------
    def get_path_planner_name(self):
        return self.path_planner_name

    def check_success(self, path):
        if path is None or len(path) == 0:
            return False
        return np.all(path[0] == self.x_start) and np.all(path[-1] == self.x_goal)

    def get_path_len(self, path):
        if path is None or len(path) == 0:
            return np.inf
        path = np.array(path)
        return np.linalg.norm(path[1:] - path[:-1], axis=1).sum()

    def extract_path(self, goal_parent_index):
        self._sync()
        path = [self.x_goal]
        i = int(goal_parent_index)
        while i != 0:
            path.append(self.vertices[i])
            i = int(self.vertex_parents[i])
        path.append(self.vertices[0])
        path.reverse()
        return np.stack(path, axis=0)

    def cost(self, vertex_index):
        return float(self.tree.cost([int(vertex_index)])[0])

    def Line(self, a, b):
        return math.hypot(*(np.asarray(b, dtype=np.float64) - np.asarray(a, dtype=np.float64)))

    def nearest_neighbor(self, node_list, n):
        """signature of the reference's staticmethod; node_list is ignored (the tree lives in HBM)"""
        self._sync()
        i = self.tree.nearest(n)
        return self.vertices[i], i

    def find_near_neighbors(self, node_new, node_new_index=None):
        return self.tree.near(node_new, -1 if node_new_index is None else node_new_index)

    def search_goal_parent(self):
        gp, _ = self.tree.search_goal_parent()
        return None if gp < 0 else gp

    def InGoalRegion(self, node):
        return self.Line(node, self.x_goal) < self.step_len and not self.utils.is_collision(node, self.x_goal)

    def new_state(self, node_start, node_goal):
        a = np.asarray(node_start, dtype=np.float64)
        b = np.asarray(node_goal, dtype=np.float64)
        if self.dim == 2:
            dx, dy = b - a
            dist, theta = math.hypot(dx, dy), math.atan2(dy, dx)
            dist = min(self.step_len, dist)
            return a + dist * np.array([math.cos(theta), math.sin(theta)])
        dx, dy, dz = b - a
        distance = math.hypot(dx, dy, dz)
        direction = np.zeros(3) if distance == 0 else (b - a) / distance
        return a + min(self.step_len, distance) * direction

    def SampleFree(self):
        lo = [r[0] + self.clearance for r in self._ranges()]
        hi = [r[1] - self.clearance for r in self._ranges()]
        while True:
            p = tuple(np.random.uniform(l, h) for l, h in zip(lo, hi))
            if not self.utils.is_inside_obs(p):
                return np.array(p)

    def generate_random_node(self, *a, **k):
        return self.SampleFree()

    def _ranges(self):
        return (self.x_range, self.y_range) if self.dim == 2 else (self.x_range, self.y_range, self.z_range)

    # ---- device <-> host ----------------------------------------------------------------------
    def _sync(self):
        self.num_vertices = self.tree.download_into(self.vertices, self.vertex_parents)

    def _one_step(self, node_rand, flags):
        """one loop body; returns the step result struct"""
        if self.mode == "exact":
            ni = self.tree.nearest(node_rand)
            node_new = self.new_state(self._vertex(ni), node_rand)
            r = self.tree.extend(ni, node_new, flags)
        else:
            r = self.tree.step(node_rand, flags)
        if r.inserted:
            self._vcache[int(r.new_idx)] = np.array(r.node_new[: self.dim])
        return r

    def _vertex(self, i):
        return self._vcache[i]

    def _begin_host_loop(self):
        self._vcache = {0: self.x_start.copy()}

    def _resident(self, iters, flags, want_trace=False):
        """run `iters` loop bodies in the persistent kernel.  The reference draws from the process-global generators inside
        its loop; here their states move into the tree (np.random.get_state() / random.getstate() -> nirrt_set_generators), the
        kernel draws from them where they are (twist and tempering on the device), and they move back afterwards - the
        global generators end up exactly where the reference's own loop would have left them."""
        D = self.dim
        done_total = 0
        traces = []
        ms = 0.0
        need_py = D == 2 and (flags & _hip.F_IRRT)
        _hip.set_generators([self.tree], [_hip.np_state()], [_hip.py_state()] if need_py else None)

        def hand_back():
            nk, npos, pk, ppos = _hip.get_generators([self.tree], want_py=bool(need_py))
            _hip.set_np_state(nk[0], npos[0])
            if need_py:
                _hip.set_py_state(pk[0], ppos[0])

        try:
            while done_total < iters:
                left = iters - done_total
                res = _hip.run_sampling([self.tree], left, None, None, flags=flags, want_trace=want_trace)
                d = int(res["iters_done"][0])
                ms += res["kernel_ms"]
                if want_trace:
                    traces.append(res["cost_trace"][0, :d])
                before = done_total
                done_total += d
                for kk in range(before // _PRINT_EVERY + 1, done_total // _PRINT_EVERY + 1):
                    self._progress(kk * _PRINT_EVERY)
                st = int(res["status"][0])
                if st == _hip.E_CAPACITY:
                    raise IndexError("tree capacity (1+iter_max vertices) exceeded")   # the reference raises IndexError here too
                if st == _hip.E_ARG:
                    raise ValueError("high <= 0")   # empty predicted cloud: np.random.randint(0, 0) in the reference
                if st == _hip.E_CLOUD:
                    hand_back()                     # update_point_cloud draws its candidates from the global generator
                    self._refresh_cloud_on_device()
                    _hip.set_generators([self.tree], [_hip.np_state()], None)
                    continue
                if st == _hip.E_STREAM:
                    # one draw rejected 2^22 generator outputs in a row (the reference would spin for ever in SampleFree)
                    raise _hip.NirrtError("sampling cannot make progress: one draw rejected 2^22 generator outputs (free space empty?)")
                if st == 0 and d < left:   # STOP_FIRST fired
                    break
        finally:
            hand_back()
        self.last_kernel_ms = ms
        self._sync()
        return done_total, (np.concatenate(traces) if want_trace and traces else np.zeros(0))

    def _progress(self, k):
        print(k)

    # ---- plotting (reference: rrt_star_2d.py:146-157 and the per-planner overrides) ---------
    _vis_kind = "RRTStarVisualizer"      # class of nirrt_star_amd.visualizer (3D planners append "3D")
    _vis_tag = "rrt*"

    @property
    def visualizer(self):
        """the reference builds it in __init__; here matplotlib is only touched when somebody asks"""
        if getattr(self, "_visualizer", None) is None:
            from . import visualizer as V
            self._visualizer = getattr(V, self._vis_kind + ("3D" if self.dim == 3 else ""))(self.x_start, self.x_goal, self.env)
        return self._visualizer

    def _vis_defaults(self, figure_title, img_filename):
        if figure_title is None:
            figure_title = "%s %dD, iteration %d" % (self._vis_tag, self.dim, self.iter_max)
        if img_filename is None:
            img_filename = "%s_%dd_example.png" % (self._vis_tag.replace("(c)", "_c"), self.dim)
        return figure_title, img_filename

    def visualize(self, figure_title=None, img_filename=None):
        figure_title, img_filename = self._vis_defaults(figure_title, img_filename)
        self.visualizer.animation(self.vertices[:self.num_vertices], self.vertex_parents[:self.num_vertices], self.path,
                                  figure_title, animation=False, img_filename=img_filename)


class _Utils:
    """Utils(env, clearance) adaptor of the reference (rrt_utils_2d.py / rrt_utils_3d.py) on the GPU tables."""

    def __init__(self, planner):
        self.p = planner
        self.env = planner.env
        self.clearance = planner.clearance

    def is_collision(self, start, end):
        return self.p.tree.is_collision(start, end)

    def is_inside_obs(self, node):
        return bool(self.p.tree.points_in_obs(np.asarray(node, dtype=np.float64)[None])[0][0])

    def is_valid(self, node):
        return bool(self.p.tree.points_in_obs(np.asarray(node, dtype=np.float64)[None])[1][0])

    def is_in_range(self, node):
        c = self.clearance
        return all(r[0] + c <= v <= r[1] - c for v, r in zip(node, self.p._ranges()))


# ================================================================================================
# RRT*
# ================================================================================================
class _RRTStar(_HipPlanner):
    _extra_flags = 0   # NRRT*: F_PNG

    def planning(self, visualize=False):
        """rrt_star_2d.py:32-65"""
        if self.mode == "resident":
            self._resident(self.iter_max, self._extra_flags)
        else:
            self._begin_host_loop()
            for k in range(self.iter_max):
                self._one_step(self.generate_random_node(), 0)
                if (k + 1) % _PRINT_EVERY == 0:
                    print(k + 1)
            self._sync()
        goal_parent_index = self.search_goal_parent()
        if goal_parent_index is None:
            if visualize:
                self.visualize()
            return
        self.path = self.extract_path(goal_parent_index)
        if visualize:
            self.visualize()

    def planning_random(self, iter_after_initial):
        """rrt_star_2d.py:198-268: until the first finite path length (<= iter_max iterations), then
        exactly iter_after_initial more; entry j = path length after j+1 iterations."""
        return self._planning_random(iter_after_initial, _hip.F_GOAL_SCAN | self._extra_flags)

    def planning_block_gap(self, path_len_threshold):
        """rrt_star_2d.py:159-196 (host loop; stops when the path gets shorter than the threshold)"""
        self._begin_host_loop()
        out = []
        for k in range(self.iter_max):
            r = self._one_step(self.generate_random_node(), _hip.F_GOAL_SCAN)
            out.append(float(r.c_best))
            if out[-1] < path_len_threshold:
                break
        self._sync()
        return out

    def _planning_random(self, iter_after_initial, flags):
        if self.mode == "resident":
            n1, tr1 = self._resident(self.iter_max, flags | _hip.F_STOP_FIRST, want_trace=True)
            lst = [float(v) for v in tr1]
            if not lst or lst[-1] == np.inf:
                return lst
            if iter_after_initial > 0:
                n2, tr2 = self._resident(iter_after_initial, flags, want_trace=True)
                lst += [float(v) for v in tr2]
            return lst
        self._begin_host_loop()
        lst = []
        for k in range(self.iter_max):
            r = self._one_step(self._sample_for_random(lst), flags)
            lst.append(float(r.c_best))
            if lst[-1] < np.inf:
                break
        if lst[-1] == np.inf:
            self._sync()
            return lst
        for k in range(iter_after_initial):
            r = self._one_step(self._sample_for_random(lst), flags)
            lst.append(float(r.c_best))
        self._sync()
        return lst

    def _sample_for_random(self, lst):
        return self.generate_random_node()


class RRTStar2D(_RRTStar):
    dim = 2

    def __init__(self, x_start, x_goal, step_len, search_radius, iter_max, env, clearance, mode=None, device_id=0):
        self._base_init(x_start, x_goal, step_len, search_radius, iter_max, env, clearance, "RRT* 2D", mode, device_id)


class RRTStar3D(_RRTStar):
    dim = 3

    def __init__(self, x_start, x_goal, step_len, search_radius, iter_max, env, clearance, mode=None, device_id=0):
        self._base_init(x_start, x_goal, step_len, search_radius, iter_max, env, clearance, "RRT* 3D", mode, device_id)


# ================================================================================================
# Informed RRT*
# ================================================================================================
class _IRRTStar(_RRTStar):
    def _irrt_init(self):
        self.path_solutions = []
        self._frame = sampling.informed_frame(self.x_start, self.x_goal)
        self.tree.set_informed(*self._frame)

    def init(self):
        c_min, xc, C = self._frame
        if self.dim == 2:
            theta = math.atan2(self.x_goal[1] - self.x_start[1], self.x_goal[0] - self.x_start[0])
            x_center = np.zeros((3, 1))
            x_center[:2, 0] = xc
            return theta, c_min, x_center, C
        return c_min, xc, C

    def find_best_path_solution(self):
        c, x = self.tree.best_solution()
        return c, x

    _vis_kind = "IRRTStarVisualizer"
    _vis_tag = "irrt*"

    def visualize(self, x_center=None, c_best=None, start_goal_straightline_dist=None, theta=None, figure_title=None,
                  img_filename=None):
        """irrt_star_2d.py:163-178 (3D: the rotation matrix C in place of theta, irrt_star_3d.py:176-192); called without
        arguments it draws the informed set of the current best solution"""
        if x_center is None:
            frame = self.init()
            x_center, start_goal_straightline_dist = (frame[2], frame[1]) if self.dim == 2 else (frame[1], frame[0])
            theta = frame[0] if self.dim == 2 else frame[2]
            c_best = self.find_best_path_solution()[0] if len(self.tree.solutions) else np.inf
        figure_title, img_filename = self._vis_defaults(figure_title, img_filename)
        self._vis_sync_clouds()
        self.visualizer.animation(self.vertices[:self.num_vertices], self.vertex_parents[:self.num_vertices], self.path,
                                  figure_title, x_center, c_best, start_goal_straightline_dist, theta, img_filename=img_filename)

    def _vis_sync_clouds(self):
        pass

    # --- sampling with the reference's own calls (host loop modes) -----------------------------
    def SampleUnitBall(self):
        if self.dim == 2:
            while True:
                x, y = random.uniform(-1, 1), random.uniform(-1, 1)
                if x ** 2 + y ** 2 < 1:
                    return np.array([[x], [y], [0.0]])
        r = np.random.uniform(0.0, 1.0)
        theta = np.random.uniform(0, np.pi)
        phi = np.random.uniform(0, 2 * np.pi)
        return np.array([r * np.sin(theta) * np.cos(phi), r * np.sin(theta) * np.sin(phi), r * np.cos(theta)])

    def SampleInformedSubset(self, c_max, c_min, x_center, C):
        eps = 1e-6 if c_max ** 2 - c_min ** 2 < 0 else 0
        if self.dim == 2:
            r = [c_max / 2.0, math.sqrt(c_max ** 2 - c_min ** 2 + eps) / 2.0, math.sqrt(c_max ** 2 - c_min ** 2 + eps) / 2.0]
            L = np.diag(r)
            xc = np.zeros((3, 1))
            xc[:2, 0] = np.asarray(x_center).ravel()[:2]
            while True:
                x_ball = self.SampleUnitBall()
                node_rand = np.dot(np.dot(C, L), x_ball) + xc
                if self.utils.is_valid((node_rand[0, 0], node_rand[1, 0])):
                    return node_rand[:2, 0]
        r = np.zeros(3)
        r[0] = c_max / 2
        r[1] = r[2] = np.sqrt(c_max ** 2 - c_min ** 2 + eps) / 2
        L = np.diag(r)
        while True:
            xball = self.SampleUnitBall()
            node_rand = C @ L @ xball + np.asarray(x_center).ravel()
            if self.utils.is_valid((node_rand[0], node_rand[1], node_rand[2])):
                return node_rand

    def generate_random_node(self, c_max=np.inf, c_min=None, x_center=None, C=None):
        if c_max < np.inf:
            f = self._frame
            return self.SampleInformedSubset(c_max, f[0], f[1], f[2])
        return self.SampleFree()

    def planning(self, visualize=False):
        """irrt_star_2d.py:42-82"""
        c_best = np.inf
        if self.mode == "resident":
            print(0)
            self._resident(self.iter_max, _hip.F_IRRT)
        else:
            self._begin_host_loop()
            for k in range(self.iter_max):
                if k % _PRINT_EVERY == 0:
                    print(k)
                r = self._one_step(self.generate_random_node(c_best), _hip.F_IRRT)
                c_best = float(r.c_best)
            if self.iter_max % _PRINT_EVERY == 0:
                print(self.iter_max)
            self._sync()
        self.path_solutions = [int(v) for v in self.tree.solutions]
        if len(self.path_solutions) > 0:
            c_best, x_best = self.find_best_path_solution()
            self.path = self.extract_path(x_best)
        else:
            self.path = []
        if visualize:
            self.visualize()

    def planning_random(self, iter_after_initial):
        """irrt_star_2d.py:230-316: entry j = c_best after j+1 iterations"""
        lst = self._planning_random(iter_after_initial, _hip.F_IRRT)
        self.path_solutions = [int(v) for v in self.tree.solutions]
        return lst

    def planning_block_gap(self, path_len_threshold):
        """irrt_star_2d.py:180-228"""
        self._begin_host_loop()
        out, c_best = [], np.inf
        for k in range(self.iter_max):
            r = self._one_step(self.generate_random_node(c_best), _hip.F_IRRT)
            c_best = float(r.c_best)
            out.append(c_best)
            if c_best < path_len_threshold:
                break
        self._sync()
        self.path_solutions = [int(v) for v in self.tree.solutions]
        return out

    def _sample_for_random(self, lst):
        return self.generate_random_node(lst[-1] if lst else np.inf)


class IRRTStar2D(_IRRTStar):
    dim = 2

    def __init__(self, x_start, x_goal, step_len, search_radius, iter_max, env, clearance, mode=None, device_id=0):
        self._base_init(x_start, x_goal, step_len, search_radius, iter_max, env, clearance, "IRRT* 2D", mode, device_id)
        self._irrt_init()


class IRRTStar3D(_IRRTStar):
    dim = 3

    def __init__(self, x_start, x_goal, step_len, search_radius, iter_max, env, clearance, mode=None, device_id=0):
        self._base_init(x_start, x_goal, step_len, search_radius, iter_max, env, clearance, "IRRT* 3D", mode, device_id)
        self._irrt_init()


# ================================================================================================
# Neural Informed RRT* (point-cloud guidance)
# ================================================================================================
class _NIRRTStarPNG(_IRRTStar):
    # "resident": the loop runs in the persistent kernel (sampling policy included) and returns to the host only
    # when the guidance cloud is due for a refresh (PointNet++), O(10) times per run
    default_mode = "resident"
    connect = False
    _vis_kind = "NIRRTStarVisualizer"
    _vis_tag = "nirrt*"

    def _vis_sync_clouds(self):
        self.visualizer.set_path_point_cloud_pred(self.path_point_cloud_pred)
        self.visualizer.set_path_point_cloud_other(self.path_point_cloud_other)

    def _png_init(self, png_wrapper, binary_mask, pc_n_points, pc_over_sample_scale, pc_sample_rate, pc_update_cost_ratio,
                  env_dict=None, connect_max_trial_attempts=None):
        self.png_wrapper = png_wrapper
        self.binary_mask = binary_mask
        self.pc_n_points = pc_n_points
        self.pc_over_sample_scale = pc_over_sample_scale
        self.pc_sample_rate = pc_sample_rate
        self.pc_neighbor_radius = self.step_len
        self.pc_update_cost_ratio = pc_update_cost_ratio
        self.env_dict = env_dict
        self.connect_max_trial_attempts = connect_max_trial_attempts
        self.path_point_cloud_pred = None
        self.path_point_cloud_other = None
        self.num_png_calls = 0
        self._c_update = np.inf
        self._irrt_init()

    def init_pc(self):
        self.update_point_cloud(cmax=np.inf, cmin=None)

    def _push_cloud(self):
        pc = self.path_point_cloud_pred if self.path_point_cloud_pred is not None else np.zeros((0, self.dim))
        self.tree.set_cloud(pc, self.pc_sample_rate, self.pc_update_cost_ratio, self._c_update)

    def _refresh_cloud_on_device(self):
        """the kernel stopped because c_best < pc_update_cost_ratio * c_update (nirrt_star_png_2d.py:114-116)"""
        c_best, _ = self.tree.best_solution()
        self.update_point_cloud(c_best, self._frame[0])
        self._c_update = c_best
        self._push_cloud()

    def SamplePointCloud(self):
        return self.path_point_cloud_pred[np.random.randint(0, len(self.path_point_cloud_pred))]

    def generate_random_node(self, c_curr, c_min=None, x_center=None, C=None, c_update=np.inf):
        """nirrt_star_png_2d.py:99-127 -> (node_rand, c_update)"""
        if c_curr < self.pc_update_cost_ratio * c_update:
            self.update_point_cloud(c_curr, self._frame[0])
            c_update = c_curr
        if np.random.random() < self.pc_sample_rate:
            return self.SamplePointCloud(), c_update
        if c_curr < np.inf:
            f = self._frame
            return self.SampleInformedSubset(c_curr, f[0], f[1], f[2]), c_update
        return self.SampleFree(), c_update

    def update_point_cloud(self, cmax, cmin):
        """nirrt_star_png_2d.py:132-174 / nirrt_star_png_c_2d.py:52-87 (+3D twins)"""
        from . import pointcloud as pcu
        if self.pc_sample_rate == 0:
            self.path_point_cloud_pred = None
            return
        if self.dim == 2:
            if cmax < np.inf:
                pc = pcu.ellipsoid_point_cloud_sampling(self.x_start, self.x_goal, cmax / cmin, self.binary_mask,
                                                        self.pc_n_points, n_raw_samples=self.pc_n_points * self.pc_over_sample_scale,
                                                        device_id=self.device_id)
            else:
                pc = pcu.generate_rectangle_point_cloud(self.binary_mask, self.pc_n_points, self.pc_over_sample_scale,
                                                        device_id=self.device_id)
        else:
            if cmax < np.inf:
                pc = pcu.ellipsoid_point_cloud_sampling_3d(self.x_start, self.x_goal, cmax / cmin, self.env, self.pc_n_points,
                                                           n_raw_samples=self.pc_n_points * self.pc_over_sample_scale,
                                                           device_id=self.device_id)
            else:
                pc = pcu.generate_rectangle_point_cloud_3d(self.env, self.pc_n_points, over_sample_scale=self.pc_over_sample_scale,
                                                           device_id=self.device_id)
        if self.connect:
            _, n_runs, path_pred = self.png_wrapper.generate_connected_path_points(
                pc.astype(np.float32), self.x_start, self.x_goal, self.env_dict, neighbor_radius=self.pc_neighbor_radius,
                max_trial_attempts=self.connect_max_trial_attempts)
            self.num_png_calls += n_runs
        else:
            start_mask = pcu.get_point_cloud_mask_around_points(pc, self.x_start[np.newaxis, :], self.pc_neighbor_radius)
            goal_mask = pcu.get_point_cloud_mask_around_points(pc, self.x_goal[np.newaxis, :], self.pc_neighbor_radius)
            path_pred, _ = self.png_wrapper.classify_path_points(pc.astype(np.float32), start_mask.astype(np.float32),
                                                                 goal_mask.astype(np.float32))
            self.num_png_calls += 1
        self.point_cloud = pc
        self.path_pred = np.asarray(path_pred)
        self.path_point_cloud_pred = pc[np.asarray(path_pred).nonzero()[0]]
        self.path_point_cloud_other = pc[np.nonzero(np.asarray(path_pred) == 0)[0]]

    def planning(self, visualize=False):
        """nirrt_star_png_2d.py:56-96"""
        self.init_pc()
        c_best = np.inf
        c_update = c_best
        if self.mode == "resident":
            self._c_update = np.inf
            self._push_cloud()
            print(0)
            self._resident(self.iter_max, _hip.F_IRRT | _hip.F_PNG)
        else:
            self._begin_host_loop()
            for k in range(self.iter_max):
                if k % _PRINT_EVERY == 0:
                    print(k)
                node_rand, c_update = self.generate_random_node(c_best, c_update=c_update)
                r = self._one_step(node_rand, _hip.F_IRRT)
                c_best = float(r.c_best)
            self._sync()
        self.path_solutions = [int(v) for v in self.tree.solutions]
        if len(self.path_solutions) > 0:
            c_best, x_best = self.find_best_path_solution()
            self.path = self.extract_path(x_best)
        else:
            self.path = []
        if visualize:
            self.visualize()

    def planning_random(self, iter_after_initial):
        """nirrt_star_png_2d.py:247-335"""
        self.init_pc()
        if self.mode == "resident":
            self._c_update = np.inf
            self._push_cloud()
            lst = self._planning_random(iter_after_initial, _hip.F_IRRT | _hip.F_PNG)
            self.path_solutions = [int(v) for v in self.tree.solutions]
            return lst
        self._begin_host_loop()
        lst, c_best, c_update = [], np.inf, np.inf
        for k in range(self.iter_max):
            node_rand, c_update = self.generate_random_node(c_best, c_update=c_update)
            r = self._one_step(node_rand, _hip.F_IRRT)
            c_best = float(r.c_best)
            lst.append(c_best)
            if c_best < np.inf:
                break
        if lst and lst[-1] < np.inf:
            for k in range(iter_after_initial):
                node_rand, c_update = self.generate_random_node(c_best, c_update=c_update)
                r = self._one_step(node_rand, _hip.F_IRRT)
                c_best = float(r.c_best)
                lst.append(c_best)
        self._sync()
        self.path_solutions = [int(v) for v in self.tree.solutions]
        return lst

    def planning_block_gap(self, path_len_threshold):
        self.init_pc()
        self._begin_host_loop()
        out, c_best, c_update = [], np.inf, np.inf
        for k in range(self.iter_max):
            node_rand, c_update = self.generate_random_node(c_best, c_update=c_update)
            r = self._one_step(node_rand, _hip.F_IRRT)
            c_best = float(r.c_best)
            out.append(c_best)
            if c_best < path_len_threshold:
                break
        self._sync()
        self.path_solutions = [int(v) for v in self.tree.solutions]
        return out


class NIRRTStarPNG2D(_NIRRTStarPNG):
    dim = 2

    def __init__(self, x_start, x_goal, step_len, search_radius, iter_max, env_dict, png_wrapper, binary_mask, clearance,
                 pc_n_points, pc_over_sample_scale, pc_sample_rate, pc_update_cost_ratio, mode=None, device_id=0):
        self._base_init(x_start, x_goal, step_len, search_radius, iter_max, Env(env_dict), clearance, "NIRRT*-PNG 2D", mode, device_id)
        self._png_init(png_wrapper, binary_mask, pc_n_points, pc_over_sample_scale, pc_sample_rate, pc_update_cost_ratio, env_dict)


class NIRRTStarPNGC2D(_NIRRTStarPNG):
    dim = 2
    connect = True
    _vis_tag = "nirrt*(c)"

    def __init__(self, x_start, x_goal, step_len, search_radius, iter_max, env_dict, png_wrapper_connect, binary_mask, clearance,
                 pc_n_points, pc_over_sample_scale, pc_sample_rate, pc_update_cost_ratio, connect_max_trial_attempts,
                 mode=None, device_id=0):
        self._base_init(x_start, x_goal, step_len, search_radius, iter_max, Env(env_dict), clearance, "NIRRT*-PNG(C) 2D", mode, device_id)
        self._png_init(png_wrapper_connect, binary_mask, pc_n_points, pc_over_sample_scale, pc_sample_rate, pc_update_cost_ratio,
                       env_dict, connect_max_trial_attempts)


class NIRRTStarPNG3D(_NIRRTStarPNG):
    dim = 3

    def __init__(self, x_start, x_goal, step_len, search_radius, iter_max, env_dict, png_wrapper, clearance,
                 pc_n_points, pc_over_sample_scale, pc_sample_rate, pc_update_cost_ratio, mode=None, device_id=0):
        self._base_init(x_start, x_goal, step_len, search_radius, iter_max, Env3D(env_dict), clearance, "NIRRT*-PNG 3D", mode, device_id)
        self._png_init(png_wrapper, None, pc_n_points, pc_over_sample_scale, pc_sample_rate, pc_update_cost_ratio, env_dict)


class NIRRTStarPNGC3D(_NIRRTStarPNG):
    dim = 3
    connect = True
    _vis_tag = "nirrt*(c)"

    def __init__(self, x_start, x_goal, step_len, search_radius, iter_max, env_dict, png_wrapper_connect, clearance,
                 pc_n_points, pc_over_sample_scale, pc_sample_rate, pc_update_cost_ratio, connect_max_trial_attempts,
                 mode=None, device_id=0):
        self._base_init(x_start, x_goal, step_len, search_radius, iter_max, Env3D(env_dict), clearance, "NIRRT*-PNG(C) 3D", mode, device_id)
        self._png_init(png_wrapper_connect, None, pc_n_points, pc_over_sample_scale, pc_sample_rate, pc_update_cost_ratio,
                       env_dict, connect_max_trial_attempts)


# ================================================================================================
# Neural RRT* (point-cloud guidance without the informed set): path_planning_classes{,_3d}/nrrt_star_png{,_c}_*.py
# ================================================================================================
class _NRRTStarPNG(_RRTStar):
    """RRT* whose sampler draws from the PointNet++-predicted cloud with probability pc_sample_rate
    (nrrt_star_png_2d.py:52-59); the cloud is computed once (init_pc) and never refreshed."""
    _extra_flags = _hip.F_PNG
    connect = False
    _vis_kind = "NRRTStarPNGVisualizer"
    _vis_tag = "nrrt*-png"

    def visualize(self, figure_title=None, img_filename=None):
        self.visualizer.set_path_point_cloud_pred(self.path_point_cloud_pred)
        _RRTStar.visualize(self, figure_title, img_filename)

    def _nrrt_init(self, png_wrapper, binary_mask, pc_n_points, pc_over_sample_scale, pc_sample_rate, env_dict=None,
                   connect_max_trial_attempts=None):
        self.png_wrapper = png_wrapper
        self.binary_mask = binary_mask
        self.pc_n_points = pc_n_points
        self.pc_over_sample_scale = pc_over_sample_scale
        self.pc_sample_rate = pc_sample_rate
        self.pc_neighbor_radius = self.step_len
        self.env_dict = env_dict
        self.connect_max_trial_attempts = connect_max_trial_attempts
        self.path_point_cloud_pred = None

    def init_pc(self):
        self.update_point_cloud()
        pc = self.path_point_cloud_pred if self.path_point_cloud_pred is not None else np.zeros((0, self.dim))
        self.tree.set_cloud(pc, self.pc_sample_rate, 0.0, np.inf)   # ratio 0: the kernel never asks for a refresh

    def update_point_cloud(self):
        from . import pointcloud as pcu
        if self.pc_sample_rate == 0:
            self.path_point_cloud_pred = None
            return
        if self.dim == 2:
            pc = pcu.generate_rectangle_point_cloud(self.binary_mask, self.pc_n_points, self.pc_over_sample_scale, device_id=self.device_id)
        else:
            pc = pcu.generate_rectangle_point_cloud_3d(self.env, self.pc_n_points, over_sample_scale=self.pc_over_sample_scale,
                                                       device_id=self.device_id)
        if self.connect:
            _, _, path_pred = self.png_wrapper.generate_connected_path_points(
                pc.astype(np.float32), self.x_start, self.x_goal, self.env_dict, neighbor_radius=self.pc_neighbor_radius,
                max_trial_attempts=self.connect_max_trial_attempts)
        else:
            sm = pcu.get_point_cloud_mask_around_points(pc, self.x_start[np.newaxis, :], self.pc_neighbor_radius)
            gm = pcu.get_point_cloud_mask_around_points(pc, self.x_goal[np.newaxis, :], self.pc_neighbor_radius)
            path_pred, _ = self.png_wrapper.classify_path_points(pc.astype(np.float32), sm.astype(np.float32), gm.astype(np.float32))
        self.path_point_cloud_pred = pc[np.asarray(path_pred).nonzero()[0]]

    def SamplePointCloud(self):
        return self.path_point_cloud_pred[np.random.randint(0, len(self.path_point_cloud_pred))]

    def generate_random_node(self, *a, **k):
        if np.random.random() < self.pc_sample_rate:
            return self.SamplePointCloud()
        return self.SampleFree()

    def planning(self, visualize=False):
        self.init_pc()
        _RRTStar.planning(self, visualize)

    def planning_random(self, iter_after_initial):
        self.init_pc()
        return _RRTStar.planning_random(self, iter_after_initial)

    def planning_block_gap(self, path_len_threshold):
        self.init_pc()
        return _RRTStar.planning_block_gap(self, path_len_threshold)


class NRRTStarPNG2D(_NRRTStarPNG):
    dim = 2

    def __init__(self, x_start, x_goal, step_len, search_radius, iter_max, env_dict, png_wrapper, binary_mask, clearance,
                 pc_n_points, pc_over_sample_scale, pc_sample_rate, mode=None, device_id=0):
        self._base_init(x_start, x_goal, step_len, search_radius, iter_max, Env(env_dict), clearance, "NRRT*-PNG 2D", mode, device_id)
        self._nrrt_init(png_wrapper, binary_mask, pc_n_points, pc_over_sample_scale, pc_sample_rate, env_dict)


class NRRTStarPNGC2D(_NRRTStarPNG):
    dim = 2
    connect = True
    _vis_tag = "nrrt*-png(c)"

    def __init__(self, x_start, x_goal, step_len, search_radius, iter_max, env_dict, png_wrapper_connect, binary_mask, clearance,
                 pc_n_points, pc_over_sample_scale, pc_sample_rate, connect_max_trial_attempts, mode=None, device_id=0):
        self._base_init(x_start, x_goal, step_len, search_radius, iter_max, Env(env_dict), clearance, "NRRT*-PNG(C) 2D", mode, device_id)
        self._nrrt_init(png_wrapper_connect, binary_mask, pc_n_points, pc_over_sample_scale, pc_sample_rate, env_dict,
                        connect_max_trial_attempts)


class NRRTStarPNG3D(_NRRTStarPNG):
    dim = 3

    def __init__(self, x_start, x_goal, step_len, search_radius, iter_max, env_dict, png_wrapper, clearance,
                 pc_n_points, pc_over_sample_scale, pc_sample_rate, mode=None, device_id=0):
        self._base_init(x_start, x_goal, step_len, search_radius, iter_max, Env3D(env_dict), clearance, "NRRT*-PNG 3D", mode, device_id)
        self._nrrt_init(png_wrapper, None, pc_n_points, pc_over_sample_scale, pc_sample_rate, env_dict)


class NRRTStarPNGC3D(_NRRTStarPNG):
    dim = 3
    connect = True
    _vis_tag = "nrrt*-png(c)"

    def __init__(self, x_start, x_goal, step_len, search_radius, iter_max, env_dict, png_wrapper_connect, clearance,
                 pc_n_points, pc_over_sample_scale, pc_sample_rate, connect_max_trial_attempts, mode=None, device_id=0):
        self._base_init(x_start, x_goal, step_len, search_radius, iter_max, Env3D(env_dict), clearance, "NRRT*-PNG(C) 3D", mode, device_id)
        self._nrrt_init(png_wrapper_connect, None, pc_n_points, pc_over_sample_scale, pc_sample_rate, env_dict,
                        connect_max_trial_attempts)


# ================================================================================================
# get_path_planner factories (same signature as every reference planner module)
# ================================================================================================
def get_rrt_star_2d(args, problem, neural_wrapper=None):
    return RRTStar2D(problem["x_start"], problem["x_goal"], args.step_len, problem["search_radius"], args.iter_max,
                     problem["env"], args.clearance)


def get_irrt_star_2d(args, problem, neural_wrapper=None):
    return IRRTStar2D(problem["x_start"], problem["x_goal"], args.step_len, problem["search_radius"], args.iter_max,
                      problem["env"], args.clearance)


def get_rrt_star_3d(args, problem, neural_wrapper=None):
    return RRTStar3D(problem["x_start"], problem["x_goal"], args.step_len, problem["search_radius"], args.iter_max,
                     problem["env"], args.clearance)


def get_irrt_star_3d(args, problem, neural_wrapper=None):
    return IRRTStar3D(problem["x_start"], problem["x_goal"], args.step_len, problem["search_radius"], args.iter_max,
                      problem["env"], args.clearance)


def get_nirrt_star_png_2d(args, problem, neural_wrapper):
    return NIRRTStarPNG2D(problem["x_start"], problem["x_goal"], args.step_len, problem["search_radius"], args.iter_max,
                          problem["env_dict"], neural_wrapper, problem["binary_mask"], args.clearance, args.pc_n_points,
                          args.pc_over_sample_scale, args.pc_sample_rate, args.pc_update_cost_ratio)


def get_nirrt_star_png_c_2d(args, problem, neural_wrapper):
    return NIRRTStarPNGC2D(problem["x_start"], problem["x_goal"], args.step_len, problem["search_radius"], args.iter_max,
                           problem["env_dict"], neural_wrapper, problem["binary_mask"], args.clearance, args.pc_n_points,
                           args.pc_over_sample_scale, args.pc_sample_rate, args.pc_update_cost_ratio,
                           args.connect_max_trial_attempts)


def get_nirrt_star_png_3d(args, problem, neural_wrapper):
    return NIRRTStarPNG3D(problem["x_start"], problem["x_goal"], args.step_len, problem["search_radius"], args.iter_max,
                          problem["env_dict"], neural_wrapper, args.clearance, args.pc_n_points,
                          args.pc_over_sample_scale, args.pc_sample_rate, args.pc_update_cost_ratio)


def get_nirrt_star_png_c_3d(args, problem, neural_wrapper):
    return NIRRTStarPNGC3D(problem["x_start"], problem["x_goal"], args.step_len, problem["search_radius"], args.iter_max,
                           problem["env_dict"], neural_wrapper, args.clearance, args.pc_n_points,
                           args.pc_over_sample_scale, args.pc_sample_rate, args.pc_update_cost_ratio,
                           args.connect_max_trial_attempts)


def get_nrrt_star_png_2d(args, problem, neural_wrapper):
    return NRRTStarPNG2D(problem["x_start"], problem["x_goal"], args.step_len, problem["search_radius"], args.iter_max,
                         problem["env_dict"], neural_wrapper, problem["binary_mask"], args.clearance, args.pc_n_points,
                         args.pc_over_sample_scale, args.pc_sample_rate)


def get_nrrt_star_png_c_2d(args, problem, neural_wrapper):
    return NRRTStarPNGC2D(problem["x_start"], problem["x_goal"], args.step_len, problem["search_radius"], args.iter_max,
                          problem["env_dict"], neural_wrapper, problem["binary_mask"], args.clearance, args.pc_n_points,
                          args.pc_over_sample_scale, args.pc_sample_rate, args.connect_max_trial_attempts)


def get_nrrt_star_png_3d(args, problem, neural_wrapper):
    return NRRTStarPNG3D(problem["x_start"], problem["x_goal"], args.step_len, problem["search_radius"], args.iter_max,
                         problem["env_dict"], neural_wrapper, args.clearance, args.pc_n_points, args.pc_over_sample_scale,
                         args.pc_sample_rate)


def get_nrrt_star_png_c_3d(args, problem, neural_wrapper):
    return NRRTStarPNGC3D(problem["x_start"], problem["x_goal"], args.step_len, problem["search_radius"], args.iter_max,
                          problem["env_dict"], neural_wrapper, args.clearance, args.pc_n_points, args.pc_over_sample_scale,
                          args.pc_sample_rate, args.connect_max_trial_attempts)
