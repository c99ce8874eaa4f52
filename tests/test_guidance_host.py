"""Host side of the batched cloud refresh (batch.Guidance._device_jobs, round 6): the per-batch job table is filled column by
column from per-problem constants - it must hold exactly what the one-structure-at-a-time path of rounds 3-5 computed with the
reference's own formulas (point_cloud_mask_utils_3d.py:132-150: C from the SVD of the start-goal frame, C @ diag(r))."""
import ctypes as C
import types

import numpy as np
import torch

from nirrt_star_amd import batch, pointcloud as pcu, pointops


def _env(rng):
    e = types.SimpleNamespace()
    e.x_range, e.y_range, e.z_range = (0.0, 50.0), (0.0, 40.0), (0.0, 30.0 + rng.uniform(0, 5))
    e.obs_ball = [[rng.uniform(5, 45), rng.uniform(5, 35), rng.uniform(5, 25), rng.uniform(1, 4)] for _ in range(3)]
    e.obs_box = [[rng.uniform(5, 40), rng.uniform(5, 30), rng.uniform(5, 20), 3.0, 4.0, 5.0] for _ in range(2)]
    return e


def test_vectorised_CL_equals_the_reference_form():
    rng = np.random.RandomState(5)
    n = 500
    xs = rng.uniform(0, 50, (n, 3))
    xg = rng.uniform(0, 50, (n, 3))
    ratio = np.concatenate([rng.uniform(1.0, 3.0, n - 3), [1.0, 1.0 - 1e-12, 1.0 + 1e-15]])      # incl. the eps branch (c_max < c_min)
    frames = [pcu.ellipsoid_frame_3d(a, b) for a, b in zip(xs, xg)]
    CL = pcu.ellipsoid_transforms_3d([f[0] for f in frames], np.stack([f[1] for f in frames]), ratio)
    for k in range(n):
        # the reference's own sequence (np.dot(C, np.diag(r))), as ellipsoid_transform_3d restates it
        ref, xc = pcu.ellipsoid_transform_3d(xs[k], xg[k], ratio[k])
        assert np.all(CL[k] == ref), k           # (== : a zero's sign is the only thing the two may differ in)
        assert np.all(np.isfinite(CL[k]))
        # ... with or without the cached frame
        ref2, xc2 = pcu.ellipsoid_transform_3d(xs[k], xg[k], ratio[k], frame=frames[k])
        assert np.array_equal(ref, ref2) and np.array_equal(xc, xc2) and np.array_equal(xc, frames[k][2])


def test_job_table_rows_equal_the_structures_filled_one_by_one():
    rng = np.random.RandomState(11)
    n = 40
    problems = []
    for _ in range(n):
        problems.append({"env": _env(rng), "x_start": rng.uniform(1, 20, 3), "x_goal": rng.uniform(25, 29, 3)})
    c_best = np.array([np.inf if k % 4 == 0 else 80.0 + k for k in range(n)])
    frames = [(float(np.linalg.norm(np.asarray(p["x_goal"]) - np.asarray(p["x_start"]))),) for p in problems]
    idx = [k for k in range(n) if k % 5 != 1]
    g = batch.Guidance(wrapper=None, dim=3, step_len=1.0)
    dev = torch.device("cpu")
    addr = [1000 + 8 * k for k in range(len(idx))]
    for rep in range(2):      # second pass: the per-problem constants come from the problems' dicts
        jobs, n_raw, n_words = g._device_jobs(idx, problems, addr, c_best, frames, dev)
        assert n_raw == 2048 * 5 and n_words == 6 * n_raw and jobs.dtype.itemsize == C.sizeof(pointops.CloudJob)
        table = (pointops.CloudJob * len(idx)).from_buffer_copy(jobs.tobytes())
        for k, i in enumerate(idx):
            pr, j = problems[i], table[k]
            env = pr["env"]
            balls, boxes = pr["_obs_dev"]
            assert j.words == addr[k] and j.balls == balls.data_ptr() and j.boxes == boxes.data_ptr()
            assert (j.n_ball, j.n_box, j.clearance, j.free_tab, j.w, j.h, j.pad) == (3, 2, 0.0, None, 0, 0, 0)
            assert np.array_equal(balls.numpy(), np.asarray(env.obs_ball)) and np.array_equal(boxes.numpy(), np.asarray(env.obs_box))
            lo = np.array([env.x_range[0], env.y_range[0], env.z_range[0]])
            hi = np.array([env.x_range[1], env.y_range[1], env.z_range[1]])
            a = np.array(j.a[:])
            if c_best[i] < np.inf:
                CL, xc = pcu.ellipsoid_transform_3d(np.asarray(pr["x_start"]), np.asarray(pr["x_goal"]), c_best[i] / frames[i][0])
                assert j.mode == 3
                assert np.all(a[:9] == CL.reshape(9)) and np.array_equal(a[9:12], xc) and np.array_equal(a[12:15], lo) and np.array_equal(a[15:18], hi)
                assert np.all(a[18:] == 0)
            else:
                assert j.mode == 2
                assert np.array_equal(a[:3], lo) and np.array_equal(a[3:6], hi - lo) and np.all(a[6:] == 0)


def test_job_table_2d_rows():
    rng = np.random.RandomState(3)
    n = 6
    problems = []
    for k in range(n):
        m = (rng.uniform(0, 1, (64, 48)) > 0.2).astype(np.uint8)
        problems.append({"binary_mask": m, "x_start": rng.uniform(1, 20, 2), "x_goal": rng.uniform(25, 40, 2)})
    c_best = np.array([np.inf, 70.0, 55.0, np.inf, 61.5, 90.0])
    frames = [(float(np.hypot(*(np.asarray(p["x_goal"]) - np.asarray(p["x_start"])))),) for p in problems]
    g = batch.Guidance(wrapper=None, dim=2, step_len=1.0)
    idx = list(range(n))
    jobs, n_raw, n_words = g._device_jobs(idx, problems, [64 * k for k in idx], c_best, frames, torch.device("cpu"))
    assert n_words == 4 * n_raw
    table = (pointops.CloudJob * n).from_buffer_copy(jobs.tobytes())
    for k in idx:
        pr, j = problems[k], table[k]
        assert (j.w, j.h) == (48, 64) and j.free_tab == pr["_free_tab_dev"].data_ptr() and (j.words or 0) == 64 * k
        assert np.array_equal(pr["_free_tab_dev"].numpy(), pcu.free_block_table(pr["binary_mask"]))
        a = np.array(j.a[:])
        if c_best[k] < np.inf:
            want = pcu.ellipse_transform_2d(np.asarray(pr["x_start"]), np.asarray(pr["x_goal"]), c_best[k] / frames[k][0])
            assert j.mode == 1 and list(a[:6]) == want and np.all(a[6:] == 0)
        else:
            assert j.mode == 0 and list(a[:2]) == [48.0, 64.0] and np.all(a[2:] == 0)
