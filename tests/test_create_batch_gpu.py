"""GPU: nirrt_create_batch / nirrt_set_informed_batch / nirrt_collision_each (one device pass for a batch of problems - the planner
objects of an evaluation set, eval_planning_2d.py:83-136) against the per-tree entry points: identical trees, identical runs."""
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dim", [2, 3])
def test_batch_created_trees_equal_trees_created_one_by_one(dim):
    from nirrt_star_amd import _hip, sampling, worlds
    if dim == 2:
        probs = [worlds.problem_2d(worlds.random_world_2d(20 + i, "b30"), i % 4) for i in range(9)]
        clr = 3
    else:
        np.random.seed(3)
        probs = [worlds.problem_3d(worlds.random_world_3d(20 + i)) for i in range(9)]
        clr = 2
    iters = 1500
    specs = [(pr["x_start"], pr["x_goal"], 10, pr["search_radius"], clr, pr["env"]) for pr in probs]
    a = _hip.create_trees(dim, iters, specs)
    b = [_hip.HipTree(dim, iters, *sp) for sp in specs]
    frames = [sampling.informed_frame(pr["x_start"], pr["x_goal"]) for pr in probs]
    _hip.set_informed_batch(a, frames)
    for t, f in zip(b, frames):
        t.set_informed(*f)
    segs = [np.stack([np.asarray(pr["x_start"], dtype=np.float64), np.asarray(pr["x_goal"], dtype=np.float64)]) for pr in probs]
    each = _hip.collision_each(a, segs)
    assert [bool(x) for x in each] == [t.is_collision(pr["x_start"], pr["x_goal"]) for t, pr in zip(b, probs)]
    # a few segments that do collide / graze, each against ITS tree's obstacles
    rng = np.random.default_rng(5)
    lo, hi = (3.0, 221.0) if dim == 2 else (2.0, 48.0)
    more = rng.uniform(lo, hi, size=(len(a), 2, dim))
    assert [bool(x) for x in _hip.collision_each(a, more)] == [t.is_collision(s[0], s[1]) for t, s in zip(b, more)]
    st_np = [_hip.np_state(np.random.RandomState(50 + i)) for i in range(len(a))]
    st_py = [_hip.py_state(random.Random(50 + i)) for i in range(len(a))]
    out = []
    for trees in (a, b):
        _hip.set_generators(trees, st_np, st_py)
        r = _hip.run_sampling(trees, iters, flags=_hip.F_IRRT, want_trace=True)
        out.append((r, [t.download() for t in trees], [t.solutions for t in trees]))
    (ra, da, sa), (rb, db, sb) = out
    assert np.array_equal(ra["iters_done"], rb["iters_done"]) and np.array_equal(ra["np_used"], rb["np_used"]) and np.array_equal(ra["py_used"], rb["py_used"])
    assert np.array_equal(ra["cost_trace"], rb["cost_trace"], equal_nan=True)
    for (va, pa), (vb, pb), s1, s2 in zip(da, db, sa, sb):
        assert np.array_equal(pa, pb) and np.array_equal(va, vb) and np.array_equal(s1, s2)
    # destroying some trees of a batch leaves the others (and their shared result slots) usable
    for t in a[::2]:
        t.close()
    assert a[1].nearest(probs[1]["x_goal"]) == b[1].nearest(probs[1]["x_goal"])
    for t in a[1::2] + b:
        t.close()


def test_a_bad_config_in_the_batch_creates_nothing():
    import ctypes as C
    from nirrt_star_amd import _hip, worlds
    pr = worlds.problem_2d(worlds.random_world_2d(1, "b30"), 0)
    L = _hip.load()
    cfgs = (_hip.Config * 3)()
    keep = [_hip._fill_config(cfgs[i], 2, 100, pr["x_start"], pr["x_goal"], 10, pr["search_radius"], 3, pr["env"], 0) for i in range(3)]
    cfgs[2].dim = 5
    handles = (C.c_void_p * 3)()
    assert L.nirrt_create_batch(cfgs, 3, handles) == _hip.E_ARG
    assert all(h is None for h in handles)
    del keep
