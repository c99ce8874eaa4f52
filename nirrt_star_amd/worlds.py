"""Synthetic planning worlds + problem dicts (SURVEY.md §8d "Synthetic inputs").

The reference ships no data (``data/`` is git-ignored there) and its world
generators depend on cv2, so benchmark/test inputs are produced here from the
*recipe* of ``generate_random_world_env_2d.py:14-47`` /
``generate_random_world_env_3d_raw.py:15-87`` with our own seeded
``numpy.random.Generator`` (world seeds 0..999), while the *problem dict* keeps
the reference's schema (``datasets/planning_problem_utils_2d.py:145-162``,
``datasets_3d/planning_problem_utils_3d.py:62-75``):

    problem = {x_start, x_goal, env_dict, env, [binary_mask], search_radius}

Three families:
  * ``ref2d``  224x224, 8-12 rectangles + 8-12 circles, sizes 16-24, clearance 3
  * ``b30``    224x224, 30 circles (BASELINE.json wording), radius range selectable
  * ``ref3d``  50^3, 6-9 boxes (side 8-19) + 6-9 balls (r 8-11), clearance 2
"""

import math

import numpy as np

from .env import Env, Env3D

CLEARANCE_2D = 3  # demo_planning_2d.py:78-79
CLEARANCE_3D = 2  # demo_planning_3d.py:75-76


# ----------------------------------------------------------------------------
# 2D mask + gamma
# ----------------------------------------------------------------------------
def rasterize_mask_2d(env_dims, rectangle_obstacles, circle_obstacles):
    """Free-space mask (H, W) float64, 1 = free, 0 = occupied.

    Analytic stand-in for the cv2 drawing in generate_random_world_env_2d.py:31-42
    (filled rectangle with both corners inclusive; filled disk). cv2 itself is an
    un-vendored dependency, so pixel-level equality with the reference PNGs is
    unpinned; oracle and product are always fed this same mask.
    """
    h, w = env_dims
    yy, xx = np.mgrid[0:h, 0:w]
    occ = np.zeros((h, w), dtype=bool)
    for x, y, rw, rh in rectangle_obstacles:
        occ |= (xx >= x) & (xx <= x + rw) & (yy >= y) & (yy <= y + rh)
    for cx, cy, r in circle_obstacles:
        occ |= (xx - cx) ** 2 + (yy - cy) ** 2 <= r * r
    return (~occ).astype(np.float64)


def gamma_rrt_star(free_vol, dim):
    """RRT* search-radius constant, datasets/planning_problem_utils_2d.py:164-172."""
    unit_ball_vol = np.pi if dim == 2 else 4.0 / 3.0 * np.pi
    return math.ceil((2 * (1 + 1.0 / dim)) ** (1.0 / dim) * (free_vol / unit_ball_vol) ** (1.0 / dim))


def approximate_free_vol_3d(env, n_points=100000):
    """Monte-Carlo free volume; consumes 3*n_points draws of the *global* numpy legacy RNG
    in the same order as datasets_3d/planning_problem_utils_3d.py:83-97 (x, then y, then z)."""
    px = np.random.uniform(env.x_range[0], env.x_range[1], n_points)
    py = np.random.uniform(env.y_range[0], env.y_range[1], n_points)
    pz = np.random.uniform(env.z_range[0], env.z_range[1], n_points)
    in_obs = np.zeros(n_points, dtype=bool)
    for bx, by, bz, br in np.asarray(env.obs_ball, dtype=np.float64).reshape(-1, 4):
        in_obs |= (px - bx) ** 2 + (py - by) ** 2 + (pz - bz) ** 2 < br ** 2
    for x, y, z, w, h, d in np.asarray(env.obs_box, dtype=np.float64).reshape(-1, 6):
        in_obs |= (x <= px) & (px <= x + w) & (y <= py) & (py <= y + h) & (z <= pz) & (pz <= z + d)
    ratio = 1 - np.mean(in_obs.astype(np.float64))
    return (env.x_range[1] - env.x_range[0]) * (env.y_range[1] - env.y_range[0]) * \
        (env.z_range[1] - env.z_range[0]) * ratio


# ----------------------------------------------------------------------------
# world generators
# ----------------------------------------------------------------------------
def _eroded_free(mask, c):
    """pixel is 'clear' iff its (2c+1)^2 neighbourhood is entirely free (Astar_with_clearance.py:228-229)."""
    h, w = mask.shape
    pad = np.zeros((h + 2 * c, w + 2 * c))
    pad[c:c + h, c:c + w] = mask
    out = np.ones((h, w), dtype=bool)
    for dy in range(2 * c + 1):
        for dx in range(2 * c + 1):
            out &= pad[dy:dy + h, dx:dx + w] > 0
    return out


def _label4(free):
    from scipy import ndimage
    lab, _ = ndimage.label(free)
    return lab


def _draw_pairs_2d(rng, mask, clearance, n_pairs, dim_limit=50, max_attempts=2000):
    """start/goal integer pixels, |dx|>=50 and |dy|>=50, clear neighbourhoods
    (generate_start_goal_points, Astar_with_clearance.py:218-235) and - our addition so that
    every benchmark problem is solvable - both in the same 4-connected clear component."""
    h, w = mask.shape
    clear = _eroded_free(mask, clearance)
    lab = _label4(clear)
    starts, goals = [], []
    attempts = 0
    while len(starts) < n_pairs and attempts < max_attempts:
        attempts += 1
        xs, xg = rng.integers(clearance, w - clearance, size=2)
        ys, yg = rng.integers(clearance, h - clearance, size=2)
        if abs(xs - xg) < dim_limit or abs(ys - yg) < dim_limit:
            continue
        if lab[ys, xs] == 0 or lab[ys, xs] != lab[yg, xg]:
            continue
        starts.append([int(xs), int(ys)])
        goals.append([int(xg), int(yg)])
    return starts, goals


def random_world_2d(seed, kind="ref2d", n_pairs=4, circle_radius_range=None):
    """env_dict with ``n_pairs`` start/goal pairs (lists), reference schema (§8(a) a23)."""
    rng = np.random.default_rng(np.random.PCG64(seed))
    h = w = 224
    while True:
        rects, circs = [], []
        if kind == "ref2d":
            n_rect = int(rng.integers(8, 13))
            n_circ = int(rng.integers(8, 13))
            rr = circle_radius_range or (16, 24)
        elif kind == "b30":
            n_rect, n_circ = 0, 30
            rr = circle_radius_range or (8, 12)
        else:
            raise ValueError(kind)
        for _ in range(n_rect):
            rects.append([int(rng.integers(0, w + 1)), int(rng.integers(0, h + 1)),
                          int(rng.integers(16, 25)), int(rng.integers(16, 25))])
        for _ in range(n_circ):
            circs.append([int(rng.integers(0, w + 1)), int(rng.integers(0, h + 1)),
                          int(rng.integers(rr[0], rr[1] + 1))])
        mask = rasterize_mask_2d((h, w), rects, circs)
        starts, goals = _draw_pairs_2d(rng, mask, CLEARANCE_2D, n_pairs)
        if len(starts) == n_pairs:
            break  # otherwise resample the world (SURVEY §8d 2D-B30 note)
    return {
        "env_dims": (h, w),
        "rectangle_obstacles": rects,
        "circle_obstacles": circs,
        "start": starts,
        "goal": goals,
    }


def random_world_3d(seed):
    """env_dict following generate_env_3d / generate_start_goal_points_3d
    (generate_random_world_env_3d_raw.py:15-87): upper-exclusive integer ranges."""
    rng = np.random.default_rng(np.random.PCG64(seed))
    xmax = ymax = zmax = 50
    c = CLEARANCE_3D
    while True:
        boxes, balls = [], []
        for _ in range(int(rng.integers(6, 10))):
            while True:
                x, y, z = (int(v) for v in rng.integers(0, 50, size=3))
                bw, bh, bd = (int(v) for v in rng.integers(8, 20, size=3))
                if x < xmax - bw and y < ymax - bh and z < zmax - bd:
                    break
            boxes.append([x, y, z, bw, bh, bd])
        for _ in range(int(rng.integers(6, 10))):
            while True:
                x, y, z = (int(v) for v in rng.integers(0, 50, size=3))
                r = int(rng.integers(8, 12))
                if r < x < xmax - r and r < y < ymax - r and r < z < zmax - r:
                    break
            balls.append([x, y, z, r])
        bb = np.asarray(balls, dtype=np.float64)
        bx = np.asarray(boxes, dtype=np.float64)
        found = None
        for _ in range(1000):
            sg = rng.integers(0, 50, size=(2, 3))
            if ((sg[0] - sg[1]) ** 2).sum() <= 50 ** 2:
                continue
            p = sg.astype(np.float64)
            in_ball = (((p[:, None, :] - bb[None, :, :3]) ** 2).sum(-1) <= (bb[None, :, 3] + c) ** 2).any()
            in_box = ((p[:, None, :] >= bx[None, :, :3] - c) & (p[:, None, :] <= bx[None, :, :3] + bx[None, :, 3:] + c)).all(-1).any()
            inside = ((p >= c) & (p <= 50 - c)).all()
            if not in_ball and not in_box and inside:
                found = sg
                break
        if found is not None:
            break
    return {
        "env_dims": [ymax, xmax, zmax],
        "box_obstacles": boxes,
        "ball_obstacles": balls,
        "start": [[int(v) for v in found[0]]],
        "goal": [[int(v) for v in found[1]]],
    }


# ----------------------------------------------------------------------------
# problem dicts (reference schema)
# ----------------------------------------------------------------------------
def problem_2d(env_dict, pair=0, mask=None):
    """= get_random_2d_problem_input (planning_problem_utils_2d.py:145-162) minus the png read.  `mask`: the world's free-space
    mask if the caller already has it (it does not depend on the start / goal pair: batches of thousands of problems over a few
    hundred worlds rasterise each world once)."""
    ed = dict(env_dict)
    ed["start"] = [list(env_dict["start"][pair])]
    ed["goal"] = [list(env_dict["goal"][pair])]
    if mask is None:
        mask = rasterize_mask_2d(ed["env_dims"], ed["rectangle_obstacles"], ed["circle_obstacles"])
    return {
        "x_start": tuple(ed["start"][0]),
        "x_goal": tuple(ed["goal"][0]),
        "env_dict": ed,
        "env": Env(ed),
        "binary_mask": mask,
        "search_radius": gamma_rrt_star(mask.sum(), 2),
    }


def problem_3d(env_dict):
    """= get_random_3d_problem_input (planning_problem_utils_3d.py:62-75); consumes 300 000
    draws of the global numpy RNG for gamma exactly like the reference."""
    env = Env3D(env_dict)
    return {
        "x_start": tuple(env_dict["start"][0]),
        "x_goal": tuple(env_dict["goal"][0]),
        "env_dict": env_dict,
        "env": env,
        "search_radius": gamma_rrt_star(approximate_free_vol_3d(env), 3),
    }
