#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE (read-only,
/root/reference) in this container.  Output = data only (inputs + expected outputs, .npz);
no reference source is copied.  Re-run:  python tests/golden/make_golden.py [--only NAME]

The reference has no tests or golden vectors of its own, so these fixtures
are what pins the oracle (oracle/nirrt_oracle.c) and, through it, the HIP path.

Families:
  geom2d / geom3d      segment-vs-obstacle, point-in-obstacle, point-validity known answers
  run_*                seeded whole planner runs with the recorded node_rand sequence
                       (L1 replay + L2 seeded parity), per-iteration trace for small runs
  random_*             planning_random() per-iteration best-cost lists
"""
import argparse
import contextlib
import io
import json
import os
import random
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
import refshim  # noqa: E402

refshim.install()
import numpy as np  # noqa: E402

from nirrt_star_amd import worlds  # noqa: E402

from path_planning_classes.rrt_star_2d import RRTStar2D  # noqa: E402
from path_planning_classes.irrt_star_2d import IRRTStar2D  # noqa: E402
from path_planning_classes.rrt_utils_2d import Utils as Utils2D  # noqa: E402
from path_planning_classes_3d.rrt_star_3d import RRTStar3D  # noqa: E402
from path_planning_classes_3d.irrt_star_3d import IRRTStar3D  # noqa: E402
from path_planning_classes_3d.rrt_utils_3d import Utils as Utils3D  # noqa: E402
from path_planning_utils.rrt_env import Env as RefEnv2D  # noqa: E402
from path_planning_utils_3d.rrt_env_3d import Env as RefEnv3D  # noqa: E402

STEP_LEN = 10


def quiet():
    return contextlib.redirect_stdout(io.StringIO())


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote %s (%.1f KiB)" % (path, os.path.getsize(path) / 1024))


def env_json(env_dict):
    return np.array(json.dumps(env_dict))


# ----------------------------------------------------------------------------------
# geometry known answers
# ----------------------------------------------------------------------------------
def geom2d():
    rng = np.random.default_rng(7)
    out = {}
    worlds_ = [worlds.random_world_2d(0, "ref2d"), worlds.random_world_2d(1, "b30"),
               worlds.random_world_2d(2, "ref2d")]
    # hand-made world for boundary-exact cases (integer geometry so every case is exact in f64)
    edge_world = {"env_dims": (224, 224), "rectangle_obstacles": [[100, 100, 20, 10]],
                  "circle_obstacles": [[50, 50, 10]], "start": [[5, 5]], "goal": [[200, 200]]}
    worlds_.append(edge_world)
    for wi, ed in enumerate(worlds_):
        u = Utils2D(RefEnv2D(ed), 3)
        n = 4000
        a = rng.uniform(-5, 229, size=(n, 2))
        # short segments (like tree edges, <= ~14 long) and a tail of long ones
        d = rng.normal(size=(n, 2)) * rng.choice([3.0, 10.0, 60.0], size=(n, 1), p=[0.3, 0.5, 0.2])
        b = a + d
        if wi == 3:
            special = [
                # tangent to inflated circle (r+c = 13): horizontal line y = 63
                ([30, 63], [70, 63]), ([30, 63.0000001], [70, 63.0000001]), ([30, 62.9999999], [70, 62.9999999]),
                # endpoint exactly on inflated circle boundary / zero-length segments
                ([63, 50], [80, 50]), ([63, 50], [63, 50]), ([64, 50], [64, 50]), ([50, 50], [50, 50]),
                # rectangle inflated: x in [97,123], y in [97,113]
                ([90, 97], [130, 97]),        # collinear with inflated bottom edge
                ([90, 96.9999], [130, 96.9999]),
                ([97, 90], [97, 120]),        # collinear with inflated left edge
                ([90, 90], [97, 97]),         # ends exactly on the corner
                ([90, 104], [104, 90]),       # passes exactly through corner (97,97)
                ([90, 103.9], [103.9, 90]),   # just misses the corner
                ([97, 97], [97, 97]),         # zero-length on the corner
                ([110, 105], [110, 105]),     # zero-length inside
                ([80, 105], [140, 105]),      # crosses
                ([80, 120], [140, 120]),      # parallel, outside
                ([124, 90], [124, 120]),      # parallel to right edge, outside by 1
                ([123, 90], [123, 120]),      # on right edge
            ]
            sa = np.array([s[0] for s in special], dtype=np.float64)
            sb = np.array([s[1] for s in special], dtype=np.float64)
            a = np.concatenate([sa, a[: n - len(sa)]])
            b = np.concatenate([sb, b[: n - len(sb)]])
        col = np.array([u.is_collision(a[i], b[i]) for i in range(n)], dtype=np.uint8)
        pts = rng.uniform(-5, 229, size=(n, 2))
        if wi == 3:
            sp = np.array([[63, 50], [50, 63], [62.9999999, 50], [97, 97], [123, 113], [123.0000001, 113],
                           [96.9999999, 100], [110, 105], [3, 3], [221, 221], [2.9999, 50], [221.0001, 50],
                           [3, 224], [0, 0]], dtype=np.float64)
            pts[: len(sp)] = sp
        inside = np.array([u.is_inside_obs(pts[i]) for i in range(n)], dtype=np.uint8)
        valid = np.array([u.is_valid(pts[i]) for i in range(n)], dtype=np.uint8)
        inrange = np.array([u.is_in_range(pts[i]) for i in range(n)], dtype=np.uint8)
        out["w%d_env" % wi] = env_json(ed)
        out["w%d_seg_a" % wi] = a
        out["w%d_seg_b" % wi] = b
        out["w%d_collision" % wi] = col
        out["w%d_pts" % wi] = pts
        out["w%d_inside" % wi] = inside
        out["w%d_valid" % wi] = valid
        out["w%d_inrange" % wi] = inrange
    out["n_worlds"] = np.array(len(worlds_))
    out["clearance"] = np.array(3.0)
    save("geom2d", **out)


def geom3d():
    rng = np.random.default_rng(11)
    out = {}
    worlds_ = [worlds.random_world_3d(0), worlds.random_world_3d(1)]
    edge_world = {"env_dims": [50, 50, 50], "box_obstacles": [[20, 20, 20, 10, 8, 6]],
                  "ball_obstacles": [[10, 10, 10, 5]], "start": [[3, 3, 3]], "goal": [[45, 45, 45]]}
    worlds_.append(edge_world)
    for wi, ed in enumerate(worlds_):
        u = Utils3D(RefEnv3D(ed), 2)
        n = 4000
        a = rng.uniform(-3, 53, size=(n, 3))
        d = rng.normal(size=(n, 3)) * rng.choice([2.0, 6.0, 25.0], size=(n, 1), p=[0.3, 0.5, 0.2])
        b = a + d
        if wi == 2:
            special = [
                # ball inflated radius 7: tangent line at z = 17 through (x, 10, 17)
                ([0, 10, 17], [20, 10, 17]), ([0, 10, 17.0000001], [20, 10, 17.0000001]),
                ([17, 10, 10], [30, 10, 10]), ([17, 10, 10], [17, 10, 10]), ([10, 10, 10], [10, 10, 10]),
                ([17.0000001, 10, 10], [17.0000001, 10, 10]),
                ([3, 10, 10], [1, 10, 10]),     # t<=0 branch, start exactly on the sphere
                ([0, 10, 10], [3, 10, 10]),     # t>=1 branch, end exactly on the sphere
                # box inflated: x [18,32], y [18,30], z [18,28]
                ([10, 18, 23], [40, 18, 23]),   # along a face
                ([10, 17.9999, 23], [40, 17.9999, 23]),
                ([18, 18, 10], [18, 18, 40]),   # along an edge
                ([10, 10, 10], [18, 18, 18]),   # ends on the corner
                ([18, 18, 18], [18, 18, 18]),   # zero length on corner
                ([25, 24, 23], [25, 24, 23]),   # zero length inside
                ([10, 26, 10], [26, 10, 10]),   # diagonal far below the box
                ([16, 24, 23], [34, 24, 23]),   # through
                ([33, 10, 23], [33, 40, 23]),   # parallel outside
                ([32, 10, 23], [32, 40, 23]),   # on face
            ]
            sa = np.array([s[0] for s in special], dtype=np.float64)
            sb = np.array([s[1] for s in special], dtype=np.float64)
            a = np.concatenate([sa, a[: n - len(sa)]])
            b = np.concatenate([sb, b[: n - len(sb)]])
        col = np.array([u.is_collision(a[i], b[i]) for i in range(n)], dtype=np.uint8)
        pts = rng.uniform(-3, 53, size=(n, 3))
        if wi == 2:
            sp = np.array([[17, 10, 10], [10, 17, 10], [16.9999999, 10, 10], [18, 18, 18], [32, 30, 28],
                           [32.0000001, 30, 28], [25, 24, 23], [2, 2, 2], [48, 48, 48], [1.9999, 25, 25],
                           [48.0001, 25, 25], [0, 0, 0]], dtype=np.float64)
            pts[: len(sp)] = sp
        inside = np.array([u.is_inside_obs(pts[i]) for i in range(n)], dtype=np.uint8)
        valid = np.array([u.is_valid(pts[i]) for i in range(n)], dtype=np.uint8)
        out["w%d_env" % wi] = env_json(ed)
        out["w%d_seg_a" % wi] = a
        out["w%d_seg_b" % wi] = b
        out["w%d_collision" % wi] = col
        out["w%d_pts" % wi] = pts
        out["w%d_inside" % wi] = inside
        out["w%d_valid" % wi] = valid
    out["n_worlds"] = np.array(len(worlds_))
    out["clearance"] = np.array(2.0)
    save("geom3d", **out)


# ----------------------------------------------------------------------------------
# whole runs
# ----------------------------------------------------------------------------------
class Recorder:
    """Wrap a reference planner instance to log node_rand and per-iteration decisions."""

    def __init__(self, planner, trace):
        self.p = planner
        self.samples = []
        self.trace = trace
        self.nearest = []
        self.near_off = [0]
        self.near_idx = []
        gen = planner.generate_random_node

        def rec_gen(*a, **k):
            r = gen(*a, **k)
            self.samples.append(np.array(r, dtype=np.float64))
            return r

        planner.generate_random_node = rec_gen
        if trace:
            nn = planner.nearest_neighbor

            def rec_nn(node_list, n):
                r = nn(node_list, n)
                self.nearest.append(int(r[1]))
                return r

            planner.nearest_neighbor = rec_nn
            fn = planner.find_near_neighbors

            def rec_fn(node_new, node_new_index=None):
                r = fn(node_new, node_new_index)
                # pad the offset table so that entry k belongs to iteration k (not every
                # iteration reaches find_near_neighbors)
                while len(self.near_off) < len(self.nearest):
                    self.near_off.append(self.near_off[-1])
                self.near_idx.extend(int(v) for v in r)
                self.near_off.append(len(self.near_idx))
                return r

            planner.find_near_neighbors = rec_fn

    def arrays(self):
        out = {"samples": np.array(self.samples)}
        if self.trace:
            while len(self.near_off) < len(self.nearest) + 1:
                self.near_off.append(self.near_off[-1])
            out["trace_nearest"] = np.array(self.nearest, dtype=np.int32)
            out["trace_near_off"] = np.array(self.near_off, dtype=np.int32)
            out["trace_near_idx"] = np.array(self.near_idx, dtype=np.int32)
        return out


def make_problem(dim, world_kind, world_seed, pair):
    if dim == 2:
        ed = worlds.random_world_2d(world_seed, world_kind)
        pr = worlds.problem_2d(ed, pair)
        clearance = 3
    else:
        ed = worlds.random_world_3d(world_seed)
        np.random.seed(world_seed)  # gamma estimate consumes the global RNG; pin it
        pr = worlds.problem_3d(ed)
        clearance = 2
    return pr, clearance


def run_planner(name, algo, dim, world_kind, world_seed, pair, iters, seed, trace=False, mode="planning",
                iter_after_initial=0):
    t0 = time.time()
    pr, clearance = make_problem(dim, world_kind, world_seed, pair)
    cls = {("rrt", 2): RRTStar2D, ("irrt", 2): IRRTStar2D, ("rrt", 3): RRTStar3D, ("irrt", 3): IRRTStar3D}[(algo, dim)]
    env = RefEnv2D(pr["env_dict"]) if dim == 2 else RefEnv3D(pr["env_dict"])
    planner = cls(pr["x_start"], pr["x_goal"], STEP_LEN, pr["search_radius"], iters, env, clearance)
    rec = Recorder(planner, trace)
    np.random.seed(seed)
    random.seed(seed)
    with quiet():
        if mode == "planning":
            planner.planning()
            extra = {}
        else:
            lst = planner.planning_random(iter_after_initial)
            extra = {"path_len_list": np.array(lst, dtype=np.float64),
                     "iter_after_initial": np.array(iter_after_initial)}
    n = planner.num_vertices
    path = np.array(planner.path, dtype=np.float64).reshape(-1, dim) if len(planner.path) else np.zeros((0, dim))
    arrays = dict(
        env=env_json(pr["env_dict"]), dim=np.array(dim), algo=np.array(algo), seed=np.array(seed),
        iter_max=np.array(iters), step_len=np.array(float(STEP_LEN)), clearance=np.array(float(clearance)),
        search_radius=np.array(float(pr["search_radius"])),
        x_start=np.array(pr["x_start"], dtype=np.float64), x_goal=np.array(pr["x_goal"], dtype=np.float64),
        n=np.array(n), vertices=planner.vertices[:n].copy(), parents=planner.vertex_parents[:n].astype(np.int64),
        path=path, path_len=np.array(float(planner.get_path_len(planner.path))),
        path_solutions=np.array(getattr(planner, "path_solutions", []), dtype=np.int64),
        **rec.arrays(), **extra)
    save(name, **arrays)
    print("   %s: n=%d path_len=%.6f  (%.1fs)" % (name, n, float(arrays["path_len"]), time.time() - t0))


def nirrt_fixture(name, dim, connect, world_seed, iters, seed, random_after=None, ratio=0.9):
    """L3: reference NIRRT*-PNG[(C)] whole run with a deterministic fake wrapper (tests/conftest.py FakePNG)
    and OUR farthest-point restatement behind the open3d stub: pins update rule + sampling mix + RNG use.
    ratio = pc_update_cost_ratio: 0.9 is the planner classes' default, 1.0 the default of demo_planning_3d.py:21
    (a refresh on EVERY improvement of the best cost)."""
    sys.path.insert(0, os.path.join(os.path.dirname(HERE)))
    from conftest import FakePNG
    if dim == 2:
        from path_planning_classes.nirrt_star_png_2d import NIRRTStarPNG2D as P
        from path_planning_classes.nirrt_star_png_c_2d import NIRRTStarPNGC2D as PC
    else:
        from path_planning_classes_3d.nirrt_star_png_3d import NIRRTStarPNG3D as P
        from path_planning_classes_3d.nirrt_star_png_c_3d import NIRRTStarPNGC3D as PC
    t0 = time.time()
    pr, clearance = make_problem(dim, "b30", world_seed, 0)
    w = FakePNG(pr["x_start"], pr["x_goal"], 25.0 if dim == 2 else 8.0)
    common = [pr["x_start"], pr["x_goal"], STEP_LEN, pr["search_radius"], iters, pr["env_dict"], w]
    tail = [clearance, 2048, 5, 0.5, ratio]
    if dim == 2:
        common.append(pr["binary_mask"])
    planner = (PC(*common, *tail, 5) if connect else P(*common, *tail))
    samples = []
    gen = planner.generate_random_node

    def rec(*a, **k):
        r = gen(*a, **k)
        samples.append(np.array(r[0], dtype=np.float64))
        return r

    planner.generate_random_node = rec
    np.random.seed(seed)
    random.seed(seed)
    lst = np.zeros(0)
    with quiet():
        if random_after is None:
            planner.planning()
        else:   # planning_random (nirrt_star_png_2d.py:247-335): per-iteration best cost list
            lst = np.array(planner.planning_random(random_after), dtype=np.float64)
    n = planner.num_vertices
    path = np.array(planner.path, dtype=np.float64).reshape(-1, dim) if len(planner.path) else np.zeros((0, dim))
    save(name, env=env_json(pr["env_dict"]), dim=np.array(dim), algo=np.array("nirrt_c" if connect else "nirrt"),
         path_len_list=lst, iter_after_initial=np.array(-1 if random_after is None else random_after),
         pc_update_cost_ratio=np.array(float(ratio)), seed=np.array(seed), iter_max=np.array(iters), step_len=np.array(float(STEP_LEN)), clearance=np.array(float(clearance)),
         search_radius=np.array(float(pr["search_radius"])), x_start=np.array(pr["x_start"], dtype=np.float64),
         x_goal=np.array(pr["x_goal"], dtype=np.float64), n=np.array(n), vertices=planner.vertices[:n].copy(),
         parents=planner.vertex_parents[:n].astype(np.int64), path=path, path_len=np.array(float(planner.get_path_len(planner.path))),
         path_solutions=np.array(planner.path_solutions, dtype=np.int64), samples=np.array(samples),
         png_calls=np.array(w.calls), binary_mask=(pr["binary_mask"].astype(np.uint8) if dim == 2 else np.zeros(0, np.uint8)))
    print("   %s: n=%d path_len=%.4f png calls %d (%.1fs)" % (name, n, float(planner.get_path_len(planner.path)), w.calls, time.time() - t0))


def synthetic_checkpoint_root(dim):
    """a scratch root_dir holding the seeded synthetic PointNet++ checkpoint in the reference's layout (there are no
    trained weights offline): written by nirrt_star_amd.png_wrapper.make_synthetic_checkpoint on the CPU, which the
    tests regenerate the same way on the GPU box (tests/conftest.py synthetic_checkpoint)"""
    import tempfile
    from nirrt_star_amd import pointops, png_wrapper
    from oracle import pointops_ref
    pointops_ref.patched(pointops).start()
    root = tempfile.mkdtemp(prefix="nirrt_ck_")
    png_wrapper.make_synthetic_checkpoint(png_wrapper.checkpoint_path(root, dim), seed=0, dim=dim, device="cpu")
    return root


def nirrt_real_fixture(name, dim, connect, world_seed, iters, seed):
    """BASELINE configs 3 / 4 as composed in the reference: NIRRT*-PNG[(C)] with the reference's own PNGWrapper and
    PointNet++ model (CPU, synthetic checkpoint), seeded numpy / python / torch generators.  Every network call is recorded
    (cloud, masks, path_pred, path_score), so the GPU test can (L3) inject path_pred and demand the reference's tree, and
    (L4) compare its own scores on the very same inputs."""
    import torch
    root = synthetic_checkpoint_root(dim)
    if dim == 2:
        from path_planning_classes.nirrt_star_png_2d import NIRRTStarPNG2D as P
        from path_planning_classes.nirrt_star_png_c_2d import NIRRTStarPNGC2D as PC
        from wrapper.pointnet_pointnet2.pointnet2_wrapper import PNGWrapper as W
        from wrapper.pointnet_pointnet2.pointnet2_wrapper_connect_bfs import PNGWrapper as WC
    else:
        from path_planning_classes_3d.nirrt_star_png_3d import NIRRTStarPNG3D as P
        from path_planning_classes_3d.nirrt_star_png_c_3d import NIRRTStarPNGC3D as PC
        from wrapper_3d.pointnet_pointnet2.pointnet2_wrapper import PNGWrapper as W
        from wrapper_3d.pointnet_pointnet2.pointnet2_wrapper_connect_bfs import PNGWrapper as WC
    t0 = time.time()
    pr, clearance = make_problem(dim, "b30", world_seed, 0)
    with quiet():
        w = (WC if connect else W)(root_dir=root, device="cpu")
    calls = []
    inner = w.classify_path_points

    def classify(pc, start_mask, goal_mask):
        pred, score = inner(pc, start_mask, goal_mask)
        calls.append((np.asarray(pc, dtype=np.float32).copy(), np.asarray(start_mask, dtype=np.float32).copy(),
                      np.asarray(goal_mask, dtype=np.float32).copy(), np.asarray(pred, dtype=np.int64).copy(),
                      np.asarray(score, dtype=np.float32).copy()))
        return pred, score

    w.classify_path_points = classify
    common = [pr["x_start"], pr["x_goal"], STEP_LEN, pr["search_radius"], iters, pr["env_dict"], w]
    tail = [clearance, 2048, 5, 0.5, 0.9]
    if dim == 2:
        common.append(pr["binary_mask"])
    planner = (PC(*common, *tail, 5) if connect else P(*common, *tail))
    np.random.seed(seed)
    random.seed(seed)
    torch.manual_seed(seed)
    with quiet():
        planner.planning()
    n = planner.num_vertices
    path = np.array(planner.path, dtype=np.float64).reshape(-1, dim) if len(planner.path) else np.zeros((0, dim))
    arrays = {}
    for i, (pc, sm, gm, pred, score) in enumerate(calls):
        arrays["call%d_pc" % i], arrays["call%d_start" % i], arrays["call%d_goal" % i] = pc, sm.astype(np.uint8), gm.astype(np.uint8)
        arrays["call%d_pred" % i], arrays["call%d_score" % i] = pred.astype(np.uint8), score.astype(np.float16)
    save(name, env=env_json(pr["env_dict"]), dim=np.array(dim), algo=np.array("nirrt_c" if connect else "nirrt"),
         seed=np.array(seed), iter_max=np.array(iters), step_len=np.array(float(STEP_LEN)), clearance=np.array(float(clearance)),
         search_radius=np.array(float(pr["search_radius"])), x_start=np.array(pr["x_start"], dtype=np.float64),
         x_goal=np.array(pr["x_goal"], dtype=np.float64), n=np.array(n), vertices=planner.vertices[:n].copy(),
         parents=planner.vertex_parents[:n].astype(np.int64), path=path, path_len=np.array(float(planner.get_path_len(planner.path))),
         path_solutions=np.array(planner.path_solutions, dtype=np.int64), n_calls=np.array(len(calls)),
         binary_mask=(pr["binary_mask"].astype(np.uint8) if dim == 2 else np.zeros(0, np.uint8)), **arrays)
    print("   %s: n=%d path_len=%.4f network calls %d, path points per call %s (%.1fs)"
          % (name, n, float(planner.get_path_len(planner.path)), len(calls), [int(c[3].sum()) for c in calls], time.time() - t0))


def nrrt_fixture(name, dim, world_seed, iters, seed, connect=False):
    """NRRT*-PNG[(C)] (RRT* + cloud sampling, no informed set) whole run with the fake wrapper (see nirrt_fixture)."""
    sys.path.insert(0, os.path.join(os.path.dirname(HERE)))
    from conftest import FakePNG
    if dim == 2:
        if connect:
            from path_planning_classes.nrrt_star_png_c_2d import NRRTStarPNGC2D as P
        else:
            from path_planning_classes.nrrt_star_png_2d import NRRTStarPNG2D as P
    else:
        if connect:
            from path_planning_classes_3d.nrrt_star_png_c_3d import NRRTStarPNGC3D as P
        else:
            from path_planning_classes_3d.nrrt_star_png_3d import NRRTStarPNG3D as P
    pr, clearance = make_problem(dim, "b30", world_seed, 0)
    w = FakePNG(pr["x_start"], pr["x_goal"], 25.0 if dim == 2 else 8.0)
    common = [pr["x_start"], pr["x_goal"], STEP_LEN, pr["search_radius"], iters, pr["env_dict"], w]
    if dim == 2:
        common.append(pr["binary_mask"])
    planner = P(*common, clearance, 2048, 5, 0.5, 5) if connect else P(*common, clearance, 2048, 5, 0.5)
    np.random.seed(seed)
    random.seed(seed)
    with quiet():
        planner.planning()
    n = planner.num_vertices
    path = np.array(planner.path, dtype=np.float64).reshape(-1, dim) if len(planner.path) else np.zeros((0, dim))
    save(name, env=env_json(pr["env_dict"]), dim=np.array(dim), algo=np.array("nrrt_c" if connect else "nrrt"), seed=np.array(seed), iter_max=np.array(iters),
         step_len=np.array(float(STEP_LEN)), clearance=np.array(float(clearance)), search_radius=np.array(float(pr["search_radius"])),
         x_start=np.array(pr["x_start"], dtype=np.float64), x_goal=np.array(pr["x_goal"], dtype=np.float64), n=np.array(n),
         vertices=planner.vertices[:n].copy(), parents=planner.vertex_parents[:n].astype(np.int64), path=path,
         path_len=np.array(float(planner.get_path_len(planner.path))), png_calls=np.array(w.calls),
         binary_mask=(pr["binary_mask"].astype(np.uint8) if dim == 2 else np.zeros(0, np.uint8)))
    print("   %s: n=%d path_len=%.4f png calls %d" % (name, n, float(planner.get_path_len(planner.path)), w.calls))


def guidance_fixture():
    """a18: the guidance-cloud functions of datasets/point_cloud_mask_utils.py:20-174 and
    datasets_3d/point_cloud_mask_utils_3d.py:83-200 run by the reference itself with seeded generators: the candidate
    set each call hands to open3d's farthest_point_down_sample (captured by the stub, refshim.FPS_CALLS), the returned
    cloud (down-sampled by the repo's restatement behind the stub), the generator position afterwards, and
    get_point_cloud_mask_around_points for one and for several centre points."""
    from datasets.point_cloud_mask_utils import (ellipsoid_point_cloud_sampling, generate_rectangle_point_cloud,
                                                  get_point_cloud_mask_around_points)
    from datasets_3d.point_cloud_mask_utils_3d import ellipsoid_point_cloud_sampling_3d, generate_rectangle_point_cloud_3d
    out = {}
    pr, _ = make_problem(2, "b30", 3, 0)
    mask = pr["binary_mask"]
    xs, xg = np.array(pr["x_start"], dtype=np.float64), np.array(pr["x_goal"], dtype=np.float64)
    out["env2"] = env_json(pr["env_dict"])
    out["mask2"] = mask.astype(np.uint8)
    out["xs2"], out["xg2"] = xs, xg

    def capture(tag, fn, seed):
        del refshim.FPS_CALLS[:]
        np.random.seed(seed)
        pc = fn()
        out[tag + "_seed"] = np.array(seed)
        out[tag + "_cloud"] = np.asarray(pc, dtype=np.float64)
        out[tag + "_next"] = np.array(np.random.random_sample())     # where the global generator stands afterwards
        out[tag + "_fps_calls"] = np.array(len(refshim.FPS_CALLS))
        if refshim.FPS_CALLS:
            out[tag + "_cand"] = refshim.FPS_CALLS[0][0]
            out[tag + "_nsamp"] = np.array(refshim.FPS_CALLS[0][1])
        return pc

    rect = capture("rect2", lambda: generate_rectangle_point_cloud(mask, 2048, 5), 11)
    for tag, ratio, seed in (("ell2_wide", 1.6, 12), ("ell2_mid", 1.15, 13), ("ell2_thin", 1.004, 14)):
        capture(tag, lambda r=ratio: ellipsoid_point_cloud_sampling(xs, xg, r, mask, n_points=2048, n_raw_samples=10240), seed)
        out[tag + "_ratio"] = np.array(ratio)
    out["mask_start"] = get_point_cloud_mask_around_points(rect, xs[np.newaxis, :], 10)
    out["mask_goal"] = get_point_cloud_mask_around_points(rect, xg[np.newaxis, :], 10)
    line = xs + np.linspace(0, 1, 25)[:, None] * (xg - xs)
    out["mask_line_pts"] = line
    out["mask_line"] = get_point_cloud_mask_around_points(rect, line, 12.5)
    out["mask_f32"] = get_point_cloud_mask_around_points(rect.astype(np.float32), xs[np.newaxis].astype(np.float32), 10)

    np.random.seed(5)
    pr3, _ = make_problem(3, "ref3d", 5, 0)
    env3 = RefEnv3D(pr3["env_dict"])
    xs3, xg3 = np.array(pr3["x_start"], dtype=np.float64), np.array(pr3["x_goal"], dtype=np.float64)
    out["env3"] = env_json(pr3["env_dict"])
    out["xs3"], out["xg3"] = xs3, xg3
    capture("rect3", lambda: generate_rectangle_point_cloud_3d(env3, 2048, over_sample_scale=5), 21)
    for tag, ratio, seed in (("ell3_wide", 1.5, 22), ("ell3_thin", 1.01, 23)):
        capture(tag, lambda r=ratio: ellipsoid_point_cloud_sampling_3d(xs3, xg3, r, env3, n_points=2048, n_raw_samples=10240), seed)
        out[tag + "_ratio"] = np.array(ratio)
    save("guidance_clouds", **out)
    print("   guidance_clouds: " + ", ".join("%s %s" % (k[:-6], out[k].shape) for k in out if k.endswith("_cloud")))


def connect_fixture():
    """a20: wrapper/utils/bfs_connect_heuristic.py:5-181 on saved clouds + PNGWrapper.generate_connected_path_points
    (wrapper/pointnet_pointnet2/pointnet2_wrapper_connect_bfs.py:76-240) driven by a deterministic classifier that needs
    several rounds: a point is "path" iff it lies within 28 of a point of the start or goal mask, so each round grows two
    blobs from the seeds the heuristic picked."""
    from datasets.point_cloud_mask_utils import generate_rectangle_point_cloud, get_point_cloud_mask_around_points
    from wrapper.utils.bfs_connect_heuristic import (bfs_point_cloud_visualization, get_boundary_mask,
                                                     select_heuristic_boundary_point)
    from wrapper.pointnet_pointnet2.pointnet2_wrapper_connect_bfs import PNGWrapper as RefWrapper
    out = {}
    pr, _ = make_problem(2, "b30", 7, 1)
    np.random.seed(31)
    pc = generate_rectangle_point_cloud(pr["binary_mask"], 2048, 5).astype(np.float32)
    xs, xg = np.array(pr["x_start"], dtype=np.float64), np.array(pr["x_goal"], dtype=np.float64)
    out["pc"], out["xs"], out["xg"] = pc, xs, xg
    out["env"] = env_json(pr["env_dict"])
    # (1) the helpers on two masks: a corridor that connects start and goal, and two separate blobs
    ab = (xg - xs).astype(np.float32)
    tt = np.clip(((pc - xs.astype(np.float32)) @ ab) / (ab @ ab), 0, 1)
    dline = np.linalg.norm(pc - (xs.astype(np.float32) + tt[:, None] * ab), axis=1)
    corridor = (dline < 14).astype(np.float32)
    blobs = ((np.linalg.norm(pc - xs.astype(np.float32), axis=1) < 35) | (np.linalg.norm(pc - xg.astype(np.float32), axis=1) < 35)).astype(np.float32)
    for tag, pm in (("corridor", corridor), ("blobs", blobs)):
        for dtag, a, b in (("sg", xs, xg), ("gs", xg, xs)):
            has, line, vis = bfs_point_cloud_visualization(pc, pm, a.astype(np.float32), b.astype(np.float32), STEP_LEN)
            bm = get_boundary_mask(pc, vis, 1 - pm, STEP_LEN)
            bi, bp, heur = select_heuristic_boundary_point(pc, bm, a.astype(np.float32), b.astype(np.float32))
            k = "%s_%s" % (tag, dtag)
            out[k + "_mask"] = pm
            out[k + "_has"] = np.array(bool(has))
            out[k + "_line"] = np.zeros((0, 2), np.float32) if line is None else np.asarray(line, dtype=np.float32)
            out[k + "_visited"] = np.asarray(vis, dtype=np.float32)
            out[k + "_boundary"] = np.asarray(bm, dtype=np.float32)
            out[k + "_bidx"] = np.array(-1 if bi is None else int(bi))
            out[k + "_bpoint"] = np.zeros(2, np.float32) if bp is None else np.asarray(bp, dtype=np.float32)
            out[k + "_heur"] = np.zeros(0) if heur is None else np.asarray(heur, dtype=np.float64)
    # (2) the multi-round loop of the reference's wrapper class with the stub classifier (no weights involved)
    calls = []

    def classify(pc_, start_mask, goal_mask):
        seeds = pc_[(np.asarray(start_mask) + np.asarray(goal_mask)) > 0]
        d = np.linalg.norm(pc_[:, None] - seeds[None], axis=2).min(axis=1) if len(seeds) else np.full(len(pc_), np.inf)
        calls.append((np.asarray(start_mask, dtype=np.float32).copy(), np.asarray(goal_mask, dtype=np.float32).copy()))
        return (d < 28).astype(np.int64), (1.0 / (1.0 + d)).astype(np.float32)

    w = object.__new__(RefWrapper)          # the stub replaces the network: no checkpoint is loaded
    w.classify_path_points = classify
    for tag, trials in (("loop5", 5), ("loop2", 2)):
        del calls[:]
        ok, runs, mask = w.generate_connected_path_points(pc, xs, xg, pr["env_dict"], STEP_LEN, trials)
        out[tag + "_ok"], out[tag + "_runs"], out[tag + "_mask"] = np.array(bool(ok)), np.array(int(runs)), np.asarray(mask, dtype=np.float32)
        out[tag + "_start_masks"] = np.stack([c[0] for c in calls])
        out[tag + "_goal_masks"] = np.stack([c[1] for c in calls])
        print("   connect %s: ok=%s runs=%d path points %d" % (tag, ok, runs, int(mask.sum())))
    save("connect_ref", **out)


def seed_grow_classifier(grow, calls=None):
    """deterministic stand-in for the network that needs several neural-connect rounds: a point is "path" iff it lies within
    `grow` of a point of the start or goal mask (same function in tests/conftest.py)"""
    def classify(pc_, start_mask, goal_mask):
        seeds = pc_[(np.asarray(start_mask) + np.asarray(goal_mask)) > 0]
        d = np.linalg.norm(pc_[:, None] - seeds[None], axis=2).min(axis=1) if len(seeds) else np.full(len(pc_), np.inf)
        if calls is not None:
            calls.append((np.asarray(start_mask, dtype=np.float32).copy(), np.asarray(goal_mask, dtype=np.float32).copy()))
        return (d < grow).astype(np.int64), (1.0 / (1.0 + d)).astype(np.float32)
    return classify


def connect_fixture3d():
    """a20 in 3D: the helpers of wrapper/utils/bfs_connect_heuristic.py on a 3D cloud and the multi-round loop of the 3D wrapper
    class (wrapper_3d/pointnet_pointnet2/pointnet2_wrapper_connect_bfs.py:76-...) with the seed-growing classifier."""
    from datasets_3d.point_cloud_mask_utils_3d import generate_rectangle_point_cloud_3d
    from wrapper.utils.bfs_connect_heuristic import (bfs_point_cloud_visualization, get_boundary_mask,
                                                     select_heuristic_boundary_point)
    from wrapper_3d.pointnet_pointnet2.pointnet2_wrapper_connect_bfs import PNGWrapper as RefWrapper3D
    out = {}
    np.random.seed(9)
    pr, _ = make_problem(3, "ref3d", 9, 0)
    env3 = RefEnv3D(pr["env_dict"])
    np.random.seed(41)
    pc = generate_rectangle_point_cloud_3d(env3, 2048, over_sample_scale=5).astype(np.float32)
    xs, xg = np.array(pr["x_start"], dtype=np.float64), np.array(pr["x_goal"], dtype=np.float64)
    out["pc"], out["xs"], out["xg"] = pc, xs, xg
    out["env"] = env_json(pr["env_dict"])
    RADIUS = 6      # (the planners use step_len = 10; 2048 points in a 50^3 box are ~4 apart, so the helpers get a tighter radius here)
    ab = (xg - xs).astype(np.float32)
    tt = np.clip(((pc - xs.astype(np.float32)) @ ab) / (ab @ ab), 0, 1)
    dline = np.linalg.norm(pc - (xs.astype(np.float32) + tt[:, None] * ab), axis=1)
    corridor = (dline < 10).astype(np.float32)
    blobs = ((np.linalg.norm(pc - xs.astype(np.float32), axis=1) < 13) | (np.linalg.norm(pc - xg.astype(np.float32), axis=1) < 13)).astype(np.float32)
    for tag, pm in (("corridor", corridor), ("blobs", blobs)):
        for dtag, a, b in (("sg", xs, xg), ("gs", xg, xs)):
            has, line, vis = bfs_point_cloud_visualization(pc, pm, a.astype(np.float32), b.astype(np.float32), RADIUS)
            bm = get_boundary_mask(pc, vis, 1 - pm, RADIUS)
            bi, bp, heur = select_heuristic_boundary_point(pc, bm, a.astype(np.float32), b.astype(np.float32))
            k = "%s_%s" % (tag, dtag)
            out[k + "_mask"] = pm
            out[k + "_has"] = np.array(bool(has))
            out[k + "_visited"] = np.asarray(vis, dtype=np.float32)
            out[k + "_boundary"] = np.asarray(bm, dtype=np.float32)
            out[k + "_bidx"] = np.array(-1 if bi is None else int(bi))
            print("   connect3d %s: has=%s visited %d boundary %d seed %s" % (k, has, int(vis.sum()), int(bm.sum()), bi))
    out["radius"] = np.array(float(RADIUS))
    calls = []
    w = object.__new__(RefWrapper3D)
    w.classify_path_points = seed_grow_classifier(8.0, calls)
    for tag, trials, rad in (("loop5", 5, 6), ("loop2", 2, 6)):
        del calls[:]
        ok, runs, mask = w.generate_connected_path_points(pc, xs, xg, pr["env_dict"], rad, trials)
        out[tag + "_ok"], out[tag + "_runs"], out[tag + "_mask"] = np.array(bool(ok)), np.array(int(runs)), np.asarray(mask, dtype=np.float32)
        out[tag + "_start_masks"] = np.stack([c[0] for c in calls])
        out[tag + "_goal_masks"] = np.stack([c[1] for c in calls])
        print("   connect3d %s: ok=%s runs=%d path points %d" % (tag, ok, runs, int(mask.sum())))
    save("connect_ref3d", **out)


def nirrtc_bfs_fixture(name, dim, world_seed, iters, seed, grow):
    """NIRRT*-PNG(C) whole run in which the neural-connect loop REALLY runs: the reference planner
    (path_planning_classes{,_3d}/nirrt_star_png_c_{2d,3d}.py) with the reference's own connect wrapper class
    (generate_connected_path_points and the bfs helpers are the reference's code) whose network is replaced by the seed-growing
    classifier.  Pins the 3D instantiation of the loop inside a planner run (NIRRT*-C 3D had no reference-run fixture)."""
    if dim == 2:
        from path_planning_classes.nirrt_star_png_c_2d import NIRRTStarPNGC2D as PC
        from wrapper.pointnet_pointnet2.pointnet2_wrapper_connect_bfs import PNGWrapper as WC
    else:
        from path_planning_classes_3d.nirrt_star_png_c_3d import NIRRTStarPNGC3D as PC
        from wrapper_3d.pointnet_pointnet2.pointnet2_wrapper_connect_bfs import PNGWrapper as WC
    t0 = time.time()
    pr, clearance = make_problem(dim, "b30", world_seed, 0)
    calls = []
    w = object.__new__(WC)
    w.classify_path_points = seed_grow_classifier(grow, calls)
    common = [pr["x_start"], pr["x_goal"], STEP_LEN, pr["search_radius"], iters, pr["env_dict"], w]
    if dim == 2:
        common.append(pr["binary_mask"])
    planner = PC(*common, clearance, 2048, 5, 0.5, 0.9, 5)
    np.random.seed(seed)
    random.seed(seed)
    with quiet():
        planner.planning()
    n = planner.num_vertices
    path = np.array(planner.path, dtype=np.float64).reshape(-1, dim) if len(planner.path) else np.zeros((0, dim))
    save(name, env=env_json(pr["env_dict"]), dim=np.array(dim), algo=np.array("nirrt_c"), grow=np.array(float(grow)),
         seed=np.array(seed), iter_max=np.array(iters), step_len=np.array(float(STEP_LEN)), clearance=np.array(float(clearance)),
         search_radius=np.array(float(pr["search_radius"])), x_start=np.array(pr["x_start"], dtype=np.float64),
         x_goal=np.array(pr["x_goal"], dtype=np.float64), n=np.array(n), vertices=planner.vertices[:n].copy(),
         parents=planner.vertex_parents[:n].astype(np.int64), path=path, path_len=np.array(float(planner.get_path_len(planner.path))),
         path_solutions=np.array(planner.path_solutions, dtype=np.int64), n_classifications=np.array(len(calls)),
         num_png_calls=np.array(int(planner.num_png_calls) if hasattr(planner, "num_png_calls") else -1),
         binary_mask=(pr["binary_mask"].astype(np.uint8) if dim == 2 else np.zeros(0, np.uint8)))
    print("   %s: n=%d path_len=%.4f classifications %d (%.1fs)" % (name, n, float(planner.get_path_len(planner.path)), len(calls), time.time() - t0))


def block_gap_fixture():
    """Block / gap evaluation problems: the reference's generator script run in a scratch directory with a seeded
    numpy generator (its JSON output is the fixture), and the problem dicts its own loader builds for a few of them.
    cv2 is not installed: the only call on that path, cv2.rectangle(img, p0, p1, color, -1), is stood in by an
    inclusive-corner fill (OpenCV's documented behaviour; pixel parity with real cv2 unpinned, as for random_2d)."""
    import runpy
    import tempfile
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as d:
        os.chdir(d)
        try:
            np.random.seed(7)
            with quiet():
                runpy.run_path(os.path.join(refshim.REF, "generate_block_gap_env_2d.py"))
            with open(os.path.join("data", "block_gap", "block_gap_configs.json")) as f:
                cfg = json.load(f)
        finally:
            os.chdir(cwd)
    import cv2

    def rectangle(img, p0, p1, color, thickness):
        assert thickness == -1
        img[p0[1]:p1[1] + 1, p0[0]:p1[0] + 1] = color
    cv2.rectangle = rectangle
    from datasets.planning_problem_utils_2d import get_block_problem_input, get_gap_problem_input
    picks = {"block": [0, 137, 250, 499], "gap": [0, 137, 250, 499]}
    probs = {}
    for kind, fn in (("block", get_block_problem_input), ("gap", get_gap_problem_input)):
        for i in picks[kind]:
            pr = fn(cfg[kind][i])
            probs["%s_%d" % (kind, i)] = {
                "x_start": [int(v) for v in pr["x_start"]], "x_goal": [int(v) for v in pr["x_goal"]],
                "env_dims": [int(v) for v in pr["env_dict"]["env_dims"]],
                "rectangle_obstacles": [[int(v) for v in r] for r in pr["env_dict"]["rectangle_obstacles"]],
                "free_pixels": float(pr["binary_mask"].sum()), "search_radius": float(pr["search_radius"]),
                "threshold": float(pr["best_path_len"] if kind == "block" else pr["flank_path_len"]),
            }
    with open(os.path.join(HERE, "block_gap_seed7.json"), "w") as f:
        json.dump({"seed": 7, "configs": cfg, "problems": probs}, f)
    print("block_gap_seed7.json: %d block, %d gap configs, %d problems" % (len(cfg["block"]), len(cfg["gap"]), len(probs)))


def block_gap_run(name, algo, kind, idx, seed, iter_max, percentage=0.02):
    """planning_block_gap() of the reference on one block / gap problem (problem dict from the reference's loader,
    see block_gap_fixture for the cv2 stand-in): per-iteration path lengths until the threshold is met."""
    with open(os.path.join(HERE, "block_gap_seed7.json")) as f:
        cfg = json.load(f)["configs"][kind][idx]
    import cv2

    def rectangle(img, p0, p1, color, thickness):
        img[p0[1]:p1[1] + 1, p0[0]:p1[0] + 1] = color
    cv2.rectangle = rectangle
    from datasets.planning_problem_utils_2d import get_block_problem_input, get_gap_problem_input
    pr = get_block_problem_input(cfg) if kind == "block" else get_gap_problem_input(cfg)
    thr = pr["best_path_len"] * (1 + percentage) if kind == "block" else pr["flank_path_len"]   # eval_planning_2d.py:117-121
    cls = {"rrt": RRTStar2D, "irrt": IRRTStar2D}[algo]
    planner = cls(pr["x_start"], pr["x_goal"], STEP_LEN, pr["search_radius"], iter_max, pr["env"], 3)
    np.random.seed(seed)
    random.seed(seed)
    with quiet():
        lst = planner.planning_block_gap(thr)
    n = planner.num_vertices
    ed = {k: ([list(map(int, r)) for r in v] if k.endswith("obstacles") else [list(map(int, q)) for q in v] if k in ("start", "goal") else list(map(int, v)))
          for k, v in pr["env_dict"].items()}
    save(name, env=env_json(ed), dim=np.array(2), algo=np.array(algo), seed=np.array(seed), iter_max=np.array(iter_max),
         step_len=np.array(float(STEP_LEN)), clearance=np.array(3.0), search_radius=np.array(float(pr["search_radius"])),
         x_start=np.array(pr["x_start"], dtype=np.float64), x_goal=np.array(pr["x_goal"], dtype=np.float64),
         kind=np.array(kind), config=np.array(json.dumps(cfg)), threshold=np.array(float(thr)),
         path_len_list=np.array(lst, dtype=np.float64), n=np.array(n),
         vertices=planner.vertices[:n].copy(), parents=planner.vertex_parents[:n].astype(np.int64))
    print("   %s: %d iterations, final %.4f < %.4f, n=%d" % (name, len(lst), lst[-1], thr, n))


def analysis_fixture():
    """The reference's own analysis script (result_analysis_random_world_2d.py) run on synthetic result pickles in a scratch
    directory: its path-cost-ratio means and first-solution indices are the expected values for nirrt_star_amd.analysis."""
    import pickle
    import runpy
    import tempfile
    import analysis_inputs
    n, methods, data = analysis_inputs.N_PROBLEMS, analysis_inputs.STEMS, analysis_inputs.make_inputs()
    cwd, argv = os.getcwd(), sys.argv
    with tempfile.TemporaryDirectory() as d:
        os.chdir(d)
        try:
            os.makedirs('results/evaluation/2d')
            for m in methods:
                with open('results/evaluation/2d/random_2d-%s-%d.pickle' % (m, n), 'wb') as f:
                    pickle.dump(data[m], f)
            sys.argv = ['x', '--random_dataset_len', str(n)]
            with quiet():
                g = runpy.run_path(os.path.join(refshim.REF, 'result_analysis_random_world_2d.py'))
        finally:
            os.chdir(cwd)
            sys.argv = argv
    out = {'n': n,
           'path_cost_mean': {k: [float(x) for x in v] for k, v in g['path_cost_mean'].items()},
           'first_solution': {k: [int(x) for x in v] for k, v in g['random_analysis'].items()}}
    with open(os.path.join(HERE, 'analysis_ref.json'), 'w') as f:
        json.dump(out, f)
    print('analysis_ref.json: %d methods x %d problems' % (len(methods), n))


def dataset_fixture():
    """The reference's training Dataset (pointnet_pointnet2/PathPlanDataLoader.py) on a small synthetic .npz in the
    generator's schema: items and label weights are the expected values for nirrt_star_amd.path_plan_dataset."""
    import tempfile
    from pointnet_pointnet2.PathPlanDataLoader import PathPlanDataset
    rng = np.random.default_rng(33)
    n, m = 5, 64
    cols = {"token": np.array(["test-%d_0" % i for i in range(n)]), "pc": rng.uniform(0, 224, size=(n, m, 2)).astype(np.float32)}
    for k, p in (("start", 0.05), ("goal", 0.05), ("astar", 0.3)):
        cols[k] = (rng.uniform(size=(n, m)) < p).astype(np.float32)
    cols["free"] = ((1 - cols["start"]) * (1 - cols["goal"])).astype(np.float32)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "test.npz")
        np.savez(path, **cols)
        with quiet():
            ds = PathPlanDataset(path)
        items = [ds[i] for i in (0, 3)]
    save("dataset_ref", **{"in_" + k: v for k, v in cols.items()}, labelweights=ds.labelweights, length=np.array(len(ds)),
         item0_raw=items[0][0], item0_xyz=items[0][1], item0_feat=items[0][2], item0_lab=items[0][3], item0_tok=np.array(str(items[0][4])),
         item3_raw=items[1][0], item3_xyz=items[1][1], item3_feat=items[1][2], item3_lab=items[1][3], item3_tok=np.array(str(items[1][4])))


def pointnet2_fixture():
    """L4: the reference PointNet++ (CPU, fp32) on a seeded cloud.  Weights = torch.manual_seed(seed) init
    (regenerated by the test from the same seed - identical construction order) + the BatchNorm running
    statistics saved here (calibrated by 4 train-mode forwards of the REFERENCE model)."""
    import torch
    from pointnet_pointnet2.models.pointnet2 import get_model as ref_get_model
    from pointnet_pointnet2.models import pointnet2_utils as ref_utils
    seed = 7
    torch.manual_seed(seed)
    model = ref_get_model(2)
    rs = np.random.RandomState(seed)

    def make_input(pc):
        xyz = ref_utils.pc_normalize(pc)
        s = (np.linalg.norm(pc - pc[0], axis=1) < 10).astype(np.float32)
        g = (np.linalg.norm(pc - pc[1], axis=1) < 10).astype(np.float32)
        feat = np.stack([s, g, 1 - ((s + g) > 0).astype(np.float32)], axis=-1)
        return torch.from_numpy(np.concatenate([xyz, feat], axis=1).astype(np.float32)).permute(1, 0).unsqueeze(0)

    model.train()
    with torch.no_grad():
        for _ in range(4):
            pc = rs.uniform(0, 224, size=(2048, 3)).astype(np.float32)
            pc[:, 2] = 0
            model(make_input(pc))
    model.eval()
    bn = {k: v.numpy().copy() for k, v in model.state_dict().items() if "running_" in k}
    # test cloud: a free-space cloud of world 0 (2D, z = 0), reference FPS starts recorded
    ed = worlds.random_world_2d(0, "b30")
    pr = worlds.problem_2d(ed, 0)
    rs2 = np.random.RandomState(11)
    cand = rs2.uniform(0, 224, size=(6000, 2))
    free = pr["binary_mask"][np.clip(cand[:, 1].astype(int), 0, 223), np.clip(cand[:, 0].astype(int), 0, 223)] > 0
    pc2 = cand[free][:2048].astype(np.float32)
    pc = np.concatenate([pc2, np.zeros((2048, 1), np.float32)], axis=1)
    xs, xg = np.array(pr["x_start"], np.float32), np.array(pr["x_goal"], np.float32)
    s = (np.linalg.norm(pc2 - xs, axis=1) < 10).astype(np.float32)
    g = (np.linalg.norm(pc2 - xg, axis=1) < 10).astype(np.float32)
    feat = np.stack([s, g, 1 - ((s + g) > 0).astype(np.float32)], axis=-1)
    x = torch.from_numpy(np.concatenate([ref_utils.pc_normalize(pc), feat], axis=1).astype(np.float32)).permute(1, 0).unsqueeze(0)
    fps_log = []
    orig = ref_utils.farthest_point_sample

    def rec(xyz, npoint):
        r = orig(xyz, npoint)
        fps_log.append(r[0].numpy().copy())
        return r

    ref_utils.farthest_point_sample = rec
    torch.manual_seed(123)
    with torch.no_grad():
        logp, l4 = model(x)
    ref_utils.farthest_point_sample = orig
    out = {"seed": np.array(seed), "input": x.numpy(), "logp": logp.numpy(), "l4": l4.numpy(), "pc": pc2,
           "start_mask": s, "goal_mask": g}
    for i, f in enumerate(fps_log):
        out["fps%d" % i] = f.astype(np.int32)
    for k, v in bn.items():
        out["bn::" + k] = v
    save("pointnet2_ref", **out)
    print("   pointnet2_ref: predicted path points %d / 2048" % int((logp.numpy()[0].argmax(-1) == 1).sum()))


JOBS = {
    "geom2d": geom2d,
    "geom3d": geom3d,
    # config 1 of BASELINE.json: rrt_star random_2d iter_max=500
    "run_rrt2d_500": lambda: run_planner("run_rrt2d_500", "rrt", 2, "ref2d", 0, 0, 500, 1000, trace=True),
    "run_rrt2d_3000": lambda: run_planner("run_rrt2d_3000", "rrt", 2, "ref2d", 3, 1, 3000, 1003),
    "run_rrt2d_b30_2000": lambda: run_planner("run_rrt2d_b30_2000", "rrt", 2, "b30", 5, 0, 2000, 1005, trace=True),
    "run_irrt2d_800": lambda: run_planner("run_irrt2d_800", "irrt", 2, "ref2d", 0, 0, 800, 1000, trace=True),
    "run_irrt2d_3000": lambda: run_planner("run_irrt2d_3000", "irrt", 2, "b30", 7, 2, 3000, 1007),
    "run_rrt3d_500": lambda: run_planner("run_rrt3d_500", "rrt", 3, "ref3d", 0, 0, 500, 1000, trace=True),
    "run_rrt3d_3000": lambda: run_planner("run_rrt3d_3000", "rrt", 3, "ref3d", 2, 0, 3000, 1002),
    "run_irrt3d_3000": lambda: run_planner("run_irrt3d_3000", "irrt", 3, "ref3d", 1, 0, 3000, 1001, trace=True),
    "random_rrt2d": lambda: run_planner("random_rrt2d", "rrt", 2, "ref2d", 4, 0, 5000, 1004, mode="random", iter_after_initial=300),
    "random_irrt2d": lambda: run_planner("random_irrt2d", "irrt", 2, "ref2d", 4, 0, 5000, 1004, mode="random", iter_after_initial=300),
    "random_rrt3d": lambda: run_planner("random_rrt3d", "rrt", 3, "ref3d", 3, 0, 5000, 1003, mode="random", iter_after_initial=300),
    "pointnet2_ref": pointnet2_fixture,
    # a DEGENERATE problem (free straight start-goal segment: the informed set collapses onto the segment, every rewiring pass sees
    # dozens of members within rounding of the threshold) - the class on which an ulp in the steer used to flip parents (round 5)
    "run_irrt2d_free_5000": lambda: run_planner("run_irrt2d_free_5000", "irrt", 2, "b30", 6, 2, 5000, 1031),
    "run_nirrt2d_1500": lambda: nirrt_fixture("run_nirrt2d_1500", 2, False, 9, 1500, 1009),
    "run_nirrtc2d_1500": lambda: nirrt_fixture("run_nirrtc2d_1500", 2, True, 10, 1500, 1010),
    "run_nirrt3d_1500": lambda: nirrt_fixture("run_nirrt3d_1500", 3, False, 4, 1500, 1004),
    # pc_update_cost_ratio = 1.0, the default of the reference's 3D demo (demo_planning_3d.py:21): a cloud refresh on every
    # improvement of the best cost (2D twin for the 2D code path)
    "run_nirrt3d_ratio1_1500": lambda: nirrt_fixture("run_nirrt3d_ratio1_1500", 3, False, 6, 1500, 1026, ratio=1.0),
    "run_nirrt2d_ratio1_1500": lambda: nirrt_fixture("run_nirrt2d_ratio1_1500", 2, False, 12, 1500, 1027, ratio=1.0),
    "run_nrrt2d_1500": lambda: nrrt_fixture("run_nrrt2d_1500", 2, 12, 1500, 1012),
    "run_nrrt3d_1500": lambda: nrrt_fixture("run_nrrt3d_1500", 3, 6, 1500, 1006),
    "random_irrt3d": lambda: run_planner("random_irrt3d", "irrt", 3, "ref3d", 3, 0, 5000, 1003, mode="random", iter_after_initial=300),
    "block_gap": block_gap_fixture,
    "analysis_ref": analysis_fixture,
    "dataset_ref": dataset_fixture,
    "blockgap_irrt_block": lambda: block_gap_run("blockgap_irrt_block", "irrt", "block", 137, 2001, 5000, percentage=0.1),
    "blockgap_rrt_gap": lambda: block_gap_run("blockgap_rrt_gap", "rrt", "gap", 250, 2002, 6000),
    "guidance_clouds": guidance_fixture,
    "connect_ref": connect_fixture,
    "connect_ref3d": connect_fixture3d,
    "run_nirrtc3d_bfs_1500": lambda: nirrtc_bfs_fixture("run_nirrtc3d_bfs_1500", 3, 11, 1500, 1021, 8.0),
    "run_nirrtc2d_bfs_1500": lambda: nirrtc_bfs_fixture("run_nirrtc2d_bfs_1500", 2, 16, 1500, 1022, 28.0),
    "run_nrrtc2d_1500": lambda: nrrt_fixture("run_nrrtc2d_1500", 2, 13, 1500, 1013, connect=True),
    "run_nrrtc3d_1500": lambda: nrrt_fixture("run_nrrtc3d_1500", 3, 7, 1500, 1007, connect=True),
    "random_nirrt2d": lambda: nirrt_fixture("random_nirrt2d", 2, False, 14, 4000, 1014, random_after=300),
    # BASELINE configs 3 and 4 with the reference's own wrapper + PointNet++ (CPU, synthetic checkpoint)
    "config3_nirrtc2d_real": lambda: nirrt_real_fixture("config3_nirrtc2d_real", 2, True, 15, 2500, 1015),
    "config4_nirrt3d_real": lambda: nirrt_real_fixture("config4_nirrt3d_real", 3, False, 8, 2500, 1008),
}

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*", default=None)
    args = ap.parse_args()
    for name, job in JOBS.items():
        if args.only and name not in args.only:
            continue
        job()
