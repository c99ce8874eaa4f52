"""GPU: edge cases and error paths of the C ABI (empty / minimal / maximal inputs, capacity limits, bad arguments)."""
import ctypes as C
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _env2d(rects=(), circs=()):
    from nirrt_star_amd.env import Env
    return Env({"env_dims": (224, 224), "rectangle_obstacles": [list(r) for r in rects], "circle_obstacles": [list(c) for c in circs]})


def test_bad_arguments_are_error_codes_not_crashes():
    from nirrt_star_amd import _hip
    env = _env2d()
    L0 = _hip.load()
    cfg = _hip.Config()
    cfg.dim, cfg.device_id, cfg.iter_max = 4, 0, 10
    h = C.c_void_p()
    assert L0.nirrt_create(C.byref(cfg), C.byref(h)) == _hip.E_ARG and not h.value
    assert b"dim" in L0.nirrt_last_error()
    cfg.dim, cfg.iter_max = 2, -5
    assert L0.nirrt_create(C.byref(cfg), C.byref(h)) == _hip.E_ARG
    assert L0.nirrt_create(None, C.byref(h)) == _hip.E_ARG
    with pytest.raises(_hip.NirrtError, match="obstacles"):
        _hip.HipTree(2, 10, (5, 5), (200, 200), 10, 100, 3, _env2d(circs=[(10 + i, 10, 3) for i in range(65)]))
    with pytest.raises(_hip.NirrtError):
        _hip.HipTree(2, 10, (5, 5), (200, 200), 10, 100, 3, env, device_id=99)
    t = _hip.HipTree(2, 10, (5, 5), (200, 200), 10, 100, 3, env)
    with pytest.raises(_hip.NirrtError):
        t.upload(np.zeros((3, 2)), np.array([0, 5, 0]))          # parent index out of range
    with pytest.raises(_hip.NirrtError):
        t.upload(np.zeros((12, 2)), np.zeros(12, dtype=np.int64))  # more vertices than capacity
    L = _hip.load()
    assert L.nirrt_nearest(None, None, None) == _hip.E_ARG
    assert L.nirrt_destroy(None) == 0
    t.close()
    t.close()   # idempotent


def test_exactly_64_obstacles_of_each_kind_work():
    from nirrt_star_amd import _hip
    rng = np.random.default_rng(0)
    circs = [(int(rng.integers(20, 200)), int(rng.integers(20, 200)), 2) for _ in range(64)]
    rects = [(int(rng.integers(20, 200)), int(rng.integers(20, 200)), 3, 3) for _ in range(64)]
    t = _hip.HipTree(2, 300, (5, 5), (218, 218), 10, 150, 1, _env2d(rects, circs))
    from oracle import oracle as orc
    ed = {"env_dims": (224, 224), "rectangle_obstacles": [list(r) for r in rects], "circle_obstacles": [list(c) for c in circs]}
    o = orc.OracleTree(2, 300, (5, 5), (218, 218), 10, 150, 1, ed)
    seg = rng.uniform(0, 224, size=(3000, 2, 2))
    exp = np.array([o.is_collision(a, b) for a, b in seg], dtype=np.uint8)
    assert np.array_equal(t.collision_batch(seg), exp)
    qs = rng.uniform(1, 223, size=(300, 2))
    for q in qs:
        r, ro = t.step(q, _hip.F_IRRT), o.step(q, True)
        assert (r.nearest_idx, r.n, r.collided) == (ro.nearest_idx, ro.n, ro.collided)
    assert np.array_equal(t.download()[1], o.parents)
    t.close()
    o.close()


def test_minimal_batches_and_zero_length_runs():
    from nirrt_star_amd import _hip
    t = _hip.HipTree(2, 5, (5, 5), (200, 200), 10, 100, 3, _env2d(circs=[(100, 100, 20)]))
    assert list(t.collision_batch(np.array([[[90.0, 100.0], [110.0, 100.0]]]))) == [1]
    ins, val = t.points_in_obs(np.array([[100.0, 100.0]]))
    assert (ins[0], val[0]) == (1, 0)
    assert len(t.cost(np.zeros(0, dtype=np.int64))) == 0 and t.cost([0])[0] == 0.0
    res = _hip.run_replay([t], np.zeros((1, 0, 2)))
    assert res["iters_done"][0] == 0 and t.n == 1
    res = _hip.run_sampling([t], 3, [np.zeros(0, dtype=np.uint32)], None, flags=0)      # no words at all
    assert res["iters_done"][0] == 0 and res["status"][0] == _hip.E_STREAM and res["np_used"][0] == 0 and t.n == 1
    t.close()


def test_capacity_is_the_references_1_plus_iter_max():
    """the reference allocates 1+iter_max rows and raises IndexError on the next insertion; the C ABI reports
    NIRRT_E_CAPACITY and the tree is left untouched by the failing iteration"""
    from nirrt_star_amd import _hip, planners, worlds
    pr = worlds.problem_2d(worlds.random_world_2d(1, "ref2d"), 0)
    t = _hip.HipTree(2, 40, pr["x_start"], pr["x_goal"], 10, pr["search_radius"], 3, pr["env"])
    rng = np.random.default_rng(1)
    with pytest.raises(_hip.NirrtError, match="capacity"):
        for _ in range(400):
            t.step(rng.uniform(3, 221, size=2), 0)
    assert t.n == 41
    v, p = t.download()
    assert len(v) == 41 and np.all(p < 41)
    t.close()
    p2 = planners.RRTStar2D(pr["x_start"], pr["x_goal"], 10, pr["search_radius"], 60, pr["env"], 3)
    np.random.seed(3)
    random.seed(3)
    p2.planning()
    with pytest.raises(IndexError):
        p2._resident(200, 0)          # keep growing a full tree, like calling the reference's loop body again
    assert p2.num_vertices <= 61


def test_start_in_goal_region_and_immediate_solution():
    from nirrt_star_amd import _hip
    t = _hip.HipTree(2, 50, (100, 100), (104, 100), 10, 100, 3, _env2d())
    gp, ln = t.search_goal_parent()
    assert gp == 0 and ln == 4.0                       # the root itself is within step_len of the goal
    r = t.step((150.0, 100.0), _hip.F_IRRT)
    assert r.inserted == 1 and r.in_goal == 1 and r.n_solutions == 1 and abs(r.c_best - 10.0 - 6.0) < 1e-12
    t.close()


def test_3d_world_with_only_boxes_or_only_balls(oracle):
    from nirrt_star_amd import _hip
    from nirrt_star_amd.env import Env3D
    rng = np.random.default_rng(4)
    for ed in ({"env_dims": [50, 50, 50], "box_obstacles": [[10, 10, 10, 12, 9, 8], [30, 25, 5, 8, 15, 19]], "ball_obstacles": []},
               {"env_dims": [50, 50, 50], "box_obstacles": [], "ball_obstacles": [[25, 25, 25, 9], [10, 38, 12, 8]]}):
        t = _hip.HipTree(3, 400, (3, 3, 3), (46, 46, 46), 10, 40, 2, Env3D(ed))
        o = oracle.OracleTree(3, 400, (3, 3, 3), (46, 46, 46), 10, 40, 2, ed)
        for q in rng.uniform(2, 48, size=(400, 3)):
            r, ro = t.step(q, 0), o.step(q, False)
            assert (r.nearest_idx, r.n, r.n_near, r.n_rewired) == (ro.nearest_idx, ro.n, ro.n_near, ro.n_rewired)
        v, p = t.download()
        assert np.array_equal(p, o.parents) and np.array_equal(v, o.vertices)
        t.close()
        o.close()
