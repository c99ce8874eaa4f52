"""Guidance point clouds (reference: datasets/point_cloud_mask_utils.py, datasets_3d/point_cloud_mask_utils_3d.py).

Same function names / arguments / RNG consumption (global numpy legacy generator) as the reference.
The farthest-point down-sampling step is open3d's `farthest_point_down_sample` in the reference
(un-vendored dependency, open3d 0.17): restated here from its published behaviour - greedy max-min
squared distance starting at point 0, result returned in ORIGINAL index order - parity at this
boundary is unpinned; everything UP TO that call (the candidate sets) and the masks are
pinned by reference-generated fixtures (tests/golden/guidance_*.npz).
"""
import math

import numpy as np


def farthest_point_down_sample(points, num_samples, device_id=0):
    """open3d.geometry.PointCloud.farthest_point_down_sample restated (points (n, 3) f64): greedy max-min squared
    distance from point 0, first maximum on ties, survivors in original order.  HIP kernel k_fps_f64 (csrc/pointops.hip,
    float64 arithmetic) on GPU `device_id`; like open3d, asking for more samples than there are points is an error."""
    pts = np.ascontiguousarray(points, dtype=np.float64)
    n = len(pts)
    if num_samples > n:
        raise ValueError("farthest_point_down_sample: num_samples (%d) exceeds the cloud size (%d)" % (num_samples, n))
    if num_samples == n:
        return pts.copy()
    if pts.shape[1] != 3:
        raise ValueError("farthest_point_down_sample expects (n, 3) points")
    from . import pointops
    return pts[pointops.farthest_point_down_sample_f64(pts, num_samples, device_id)]


def get_point_cloud_mask_around_points(point_cloud, points, neighbor_radius=3):
    """point_cloud (n, C), points (m, C) -> bool (n,): strictly inside neighbor_radius of any point
    (point_cloud_mask_utils.py:20-31)"""
    diff = point_cloud[:, np.newaxis] - points
    dist = np.linalg.norm(diff, axis=2)
    return np.sum(dist < neighbor_radius, axis=1) > 0


def _free_pixels_2d(point_cloud, binary_mask):
    """keep points whose 4 surrounding pixels (clipped to the image) are all free (point_cloud_mask_utils.py:52-62: the
    product of binary_mask[clip(y + dy), clip(x + dx)] over dx, dy in {0, 1} is non-zero).  One table look-up per point: the
    edge-replicated mask is reduced to "this 2 x 2 block is free" once (entry [iy, ix] stands for the clipped pixel pairs
    (iy - 1, iy) x (ix - 1, ix)), which is the same set of four pixels for every integer position, inside the image or not."""
    h, w = binary_mask.shape
    P = np.empty((h + 2, w + 2), dtype=bool)
    P[1:-1, 1:-1] = binary_mask != 0
    P[0, 1:-1] = P[1, 1:-1]
    P[-1, 1:-1] = P[-2, 1:-1]
    P[:, 0] = P[:, 1]
    P[:, -1] = P[:, -2]
    block_free = P[:-1, :-1] & P[:-1, 1:] & P[1:, :-1] & P[1:, 1:]      # (h + 1, w + 1)
    pix = point_cloud.astype(int)
    ix = pix[:, 0] + 1
    np.maximum(ix, 0, out=ix)
    np.minimum(ix, w, out=ix)
    iy = pix[:, 1] + 1
    np.maximum(iy, 0, out=iy)
    np.minimum(iy, h, out=iy)
    return np.flatnonzero(block_free.ravel()[iy * (w + 1) + ix])


def free_block_table(binary_mask):
    """(h + 1, w + 1) uint8: entry [iy, ix] = 1 iff the clipped pixel pairs (iy - 1, iy) x (ix - 1, ix) are all free - the table
    _free_pixels_2d looks points up in (row-major; what the device-side candidate filter reads)"""
    h, w = binary_mask.shape
    P = np.empty((h + 2, w + 2), dtype=bool)
    P[1:-1, 1:-1] = binary_mask != 0
    P[0, 1:-1] = P[1, 1:-1]
    P[-1, 1:-1] = P[-2, 1:-1]
    P[:, 0] = P[:, 1]
    P[:, -1] = P[:, -2]
    return np.ascontiguousarray((P[:-1, :-1] & P[:-1, 1:] & P[1:, :-1] & P[1:, 1:]).astype(np.uint8))


def ellipse_frame_2d(start_point, goal_point):
    """what ellipsoid_candidates derives from the problem alone: (c_min, C, x_center) (point_cloud_mask_utils.py:118-126)"""
    dx, dy = goal_point - start_point
    c_min = math.hypot(dx, dy)
    C = _rotation_to_world_2d(start_point, goal_point, c_min)
    x_center = np.concatenate([(start_point + goal_point) / 2., np.array([0.])], axis=0)
    return c_min, C, x_center


def ellipse_transform_2d(start_point, goal_point, max_min_ratio, frame=None):
    """the constants of ellipsoid_candidates' transform: (C.L rows 0-1 / columns 0-1, x_center) exactly as the reference forms them
    (point_cloud_mask_utils.py:118-135); `frame` = ellipse_frame_2d(...) of the problem, if the caller keeps it"""
    c_min, C, x_center = frame if frame is not None else ellipse_frame_2d(start_point, goal_point)
    c_max = c_min * max_min_ratio
    eps = 1e-6 if c_max ** 2 - c_min ** 2 < 0 else 0
    r = [c_max / 2.0, math.sqrt(c_max ** 2 - c_min ** 2 + eps) / 2.0, math.sqrt(c_max ** 2 - c_min ** 2 + eps) / 2.0]
    CL = np.dot(C, np.diag(r))
    return [float(CL[0, 0]), float(CL[0, 1]), float(CL[1, 0]), float(CL[1, 1]), float(x_center[0]), float(x_center[1])]


def rectangle_candidates(binary_mask, n_points, over_sample_scale=5, rng=None):
    """the over-sampled free-space candidates the reference hands to open3d (point_cloud_mask_utils.py:35-68):
    (m, 3) with z = 0.  `rng`: a numpy RandomState (default: the process-global legacy generator, like the reference)."""
    rng = np.random if rng is None else rng
    h, w = binary_mask.shape
    # rng.uniform(low=[0, 0], high=[w, h], size=(m, 2)) of the reference = 0 + (high - low) * random_sample, element by element
    # in C order: the same doubles, without the array-argument path of the legacy generator
    pc = rng.random_sample((n_points * over_sample_scale, 2)) * np.array([float(w), float(h)])
    keep = _free_pixels_2d(pc, binary_mask)
    out = np.zeros((len(keep), 3))
    out[:, :2] = pc[keep]
    return out


def generate_rectangle_point_cloud(binary_mask, n_points, over_sample_scale=5, rng=None, device_id=0):
    """point_cloud_mask_utils.py:35-73 -> (n_points, 2)"""
    return farthest_point_down_sample(rectangle_candidates(binary_mask, n_points, over_sample_scale, rng), n_points, device_id)[:, :2]


def _rotation_to_world_2d(start_point, goal_point, L):
    a1 = (goal_point - start_point) / L
    a1 = np.concatenate([a1, np.array([0.])], axis=0)[:, np.newaxis]
    e1 = np.array([[1.0], [0.0], [0.0]])
    M = a1 @ e1.T
    U, _, V_T = np.linalg.svd(M, True, True)
    return U @ np.diag([1.0, 1.0, np.linalg.det(U) * np.linalg.det(V_T.T)]) @ V_T


def ellipsoid_candidates(start_point, goal_point, max_min_ratio, binary_mask, n_raw_samples=10000, rng=None):
    """candidates of the ellipse-restricted cloud before any down-sampling (point_cloud_mask_utils.py:104-168): (m, 2)"""
    rng = np.random if rng is None else rng
    dx, dy = goal_point - start_point
    c_min = math.hypot(dx, dy)
    C = _rotation_to_world_2d(start_point, goal_point, c_min)
    x_center = np.concatenate([(start_point + goal_point) / 2., np.array([0.])], axis=0)
    c_max = c_min * max_min_ratio
    eps = 1e-6 if c_max ** 2 - c_min ** 2 < 0 else 0
    r = [c_max / 2.0, math.sqrt(c_max ** 2 - c_min ** 2 + eps) / 2.0, math.sqrt(c_max ** 2 - c_min ** 2 + eps) / 2.0]
    L = np.diag(r)
    samples = rng.uniform(-1, 1, size=(n_raw_samples, 2))
    samples = samples[np.linalg.norm(samples, axis=1) <= 1]
    samples = np.concatenate([samples, np.zeros((len(samples), 1))], axis=1)
    x_rand = np.dot(np.dot(C, L), samples.T).T + x_center
    pc = x_rand[:, :2]
    pc = pc[_free_pixels_2d(pc, binary_mask)]
    h, w = binary_mask.shape
    in_range = (0 <= pc[:, 0]) & (pc[:, 0] <= w) & (0 <= pc[:, 1]) & (pc[:, 1] <= h)
    return pc[in_range]


def ellipsoid_point_cloud_sampling(start_point, goal_point, max_min_ratio, binary_mask, n_points=1000, n_raw_samples=10000, rng=None,
                                   device_id=0):
    """point_cloud_mask_utils.py:104-174 -> (<= n_points, 2): down-sampled only if more than n_points candidates survive"""
    pc = ellipsoid_candidates(start_point, goal_point, max_min_ratio, binary_mask, n_raw_samples, rng)
    if len(pc) > n_points:
        pc3 = np.concatenate([pc, np.zeros((pc.shape[0], 1))], axis=1)
        pc = farthest_point_down_sample(pc3, n_points, device_id)[:, :2]
    return pc


# ---------------------------------------------------------------------------------------------
# 3D
# ---------------------------------------------------------------------------------------------
def _in_obstacles_3d(points, env, clearance):
    p = np.asarray(points, dtype=np.float64)
    inside = np.zeros(len(p), dtype=bool)
    for bx, by, bz, br in np.asarray(env.obs_ball, dtype=np.float64).reshape(-1, 4):
        rc = br + clearance
        inside |= (p[:, 0] - bx) ** 2 + (p[:, 1] - by) ** 2 + (p[:, 2] - bz) ** 2 < rc ** 2
    for x, y, z, w, h, d in np.asarray(env.obs_box, dtype=np.float64).reshape(-1, 6):
        inside |= ((x - clearance <= p[:, 0]) & (p[:, 0] <= x + w + clearance) & (y - clearance <= p[:, 1]) &
                   (p[:, 1] <= y + h + clearance) & (z - clearance <= p[:, 2]) & (p[:, 2] <= z + d + clearance))
    return inside


def rectangle_candidates_3d(env, n_points, over_sample_scale=5, clearance=0, rng=None):
    """point_cloud_mask_utils_3d.py:83-107: uniform over the (shrunken) box, obstacle points dropped -> (m, 3)"""
    rng = np.random if rng is None else rng
    pc = rng.uniform(
        low=(env.x_range[0] + clearance, env.y_range[0] + clearance, env.z_range[0] + clearance),
        high=(env.x_range[1] - clearance, env.y_range[1] - clearance, env.z_range[1] - clearance),
        size=(n_points * over_sample_scale, 3))
    return pc[~_in_obstacles_3d(pc, env, clearance)]


def generate_rectangle_point_cloud_3d(env, n_points, over_sample_scale=5, use_open3d=True, clearance=0, rng=None, device_id=0):
    """point_cloud_mask_utils_3d.py:83-113"""
    pc = rectangle_candidates_3d(env, n_points, over_sample_scale, clearance, rng)
    if len(pc) > n_points:
        pc = farthest_point_down_sample(pc, n_points, device_id)
    return pc


def _rotation_to_world_3d(x_start, x_goal, L):
    a1 = (x_goal - x_start) / L
    M = np.outer(a1, [1, 0, 0])
    U, S, V = np.linalg.svd(M)
    return U @ np.diag([1, 1, np.linalg.det(U) * np.linalg.det(V)]) @ V.T


def ellipsoid_frame_3d(start_point, goal_point):
    """what ellipsoid_candidates_3d derives from the problem alone: (c_min, C, x_center) (point_cloud_mask_utils_3d.py:137-141) -
    the SVD behind C is the same for every refresh of a problem's cloud"""
    c_min = np.linalg.norm(goal_point - start_point)
    C = _rotation_to_world_3d(start_point, goal_point, c_min)
    x_center = (start_point + goal_point) / 2.
    return c_min, C, x_center


def ellipsoid_transforms_3d(c_min, C, max_min_ratio):
    """ellipsoid_transform_3d's C.L for MANY clouds at once: c_min (n,), C (n, 3, 3), max_min_ratio (n,) -> (n, 3, 3).
    C @ diag(r) scales column j by r[j] (the products with the zeros of diag(r) add exact zeros), so the entries are the single
    products C[i][j] * r[j] the reference's np.dot rounds to (tests/test_guidance_host.py compares the two)."""
    c_min = np.asarray(c_min, dtype=np.float64)
    c_max = c_min * np.asarray(max_min_ratio, dtype=np.float64)
    # the squares are the reference's SCALAR `c_max ** 2` / `c_min ** 2`: numpy evaluates a float64 scalar's power through libm's
    # pow(x, 2.0), which is not always the correctly rounded x * x an array's `** 2` (np.square) gives (82 of 100 000 arguments
    # differ by an ulp with glibc 2.35)
    sq_max = np.array([v ** 2 for v in c_max], dtype=np.float64)
    sq_min = np.array([v ** 2 for v in c_min], dtype=np.float64)
    rad = sq_max - sq_min
    eps = np.where(rad < 0, 1e-6, 0.0)
    r = np.empty((len(c_min), 3))
    r[:, 0] = c_max / 2
    r[:, 1] = r[:, 2] = np.sqrt(rad + eps) / 2
    return np.asarray(C, dtype=np.float64) * r[:, np.newaxis, :]


def ellipsoid_transform_3d(start_point, goal_point, max_min_ratio, frame=None):
    """the constants of ellipsoid_candidates_3d's transform, formed like the reference forms them (point_cloud_mask_utils_3d.py:
    137-150): (C.L (3, 3), x_center (3,)) - what the device-side candidate generation takes; `frame` = ellipsoid_frame_3d(...)
    of the problem, if the caller keeps it"""
    c_min, C, x_center = frame if frame is not None else ellipsoid_frame_3d(start_point, goal_point)
    c_max = c_min * max_min_ratio
    eps = 1e-6 if c_max ** 2 - c_min ** 2 < 0 else 0
    r = np.zeros(3)
    r[0] = c_max / 2
    r[1] = r[2] = np.sqrt(c_max ** 2 - c_min ** 2 + eps) / 2
    return np.dot(C, np.diag(r)), x_center


def ellipsoid_candidates_3d(start_point, goal_point, max_min_ratio, env, n_raw_samples=10000, clearance=0, rng=None):
    """point_cloud_mask_utils_3d.py:132-195: ellipsoid-restricted candidates before any down-sampling -> (m, 3)"""
    rng = np.random if rng is None else rng
    c_min = np.linalg.norm(goal_point - start_point)
    C = _rotation_to_world_3d(start_point, goal_point, c_min)
    x_center = (start_point + goal_point) / 2.
    c_max = c_min * max_min_ratio
    eps = 1e-6 if c_max ** 2 - c_min ** 2 < 0 else 0
    r = np.zeros(3)
    r[0] = c_max / 2
    r[1] = r[2] = np.sqrt(c_max ** 2 - c_min ** 2 + eps) / 2
    L = np.diag(r)
    radius = rng.uniform(0.0, 1.0, n_raw_samples)
    theta = rng.uniform(0, np.pi, n_raw_samples)
    phi = rng.uniform(0, 2 * np.pi, n_raw_samples)
    samples = np.array([radius * np.sin(theta) * np.cos(phi), radius * np.sin(theta) * np.sin(phi), radius * np.cos(theta)]).T
    pc = np.dot(np.dot(C, L), samples.T).T + x_center
    in_range = ((env.x_range[0] + clearance <= pc[:, 0]) & (pc[:, 0] <= env.x_range[1] - clearance) &
                (env.y_range[0] + clearance <= pc[:, 1]) & (pc[:, 1] <= env.y_range[1] - clearance) &
                (env.z_range[0] + clearance <= pc[:, 2]) & (pc[:, 2] <= env.z_range[1] - clearance))
    return pc[in_range & ~_in_obstacles_3d(pc, env, clearance)]


def ellipsoid_point_cloud_sampling_3d(start_point, goal_point, max_min_ratio, env, n_points=1000, n_raw_samples=10000,
                                      clearance=0, rng=None, device_id=0):
    """point_cloud_mask_utils_3d.py:132-200"""
    pc = ellipsoid_candidates_3d(start_point, goal_point, max_min_ratio, env, n_raw_samples, clearance, rng)
    if len(pc) > n_points:
        pc = farthest_point_down_sample(pc, n_points, device_id)
    return pc
