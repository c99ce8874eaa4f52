"""L2 (seeded) parity of the oracle's whole loops incl. sampling: same seeds as the reference run ->
same tree, straight from the raw MT19937 word streams of numpy's legacy generator and python's random."""
import random

import numpy as np
import pytest

from conftest import load_golden, make_oracle_tree


def _streams(g, n_np, n_py):
    from nirrt_star_amd import sampling
    np.random.seed(int(g["seed"]))
    random.seed(int(g["seed"]))
    return sampling.peek_np_words(n_np), sampling.peek_py_words(n_py)


@pytest.mark.parametrize("name", ["run_rrt2d_3000", "run_rrt3d_3000", "run_irrt2d_3000", "run_irrt2d_800", "run_irrt3d_3000"])
def test_seeded_loop_reproduces_reference(oracle, name):
    from nirrt_star_amd import sampling
    g = load_golden(name)
    irrt = str(g["algo"]) == "irrt"
    dim = int(g["dim"])
    iters = int(g["iter_max"])
    t = make_oracle_tree(oracle, g)
    npw, pyw = _streams(g, iters * 6 * (60 if dim == 3 and irrt else 4), iters * 24)
    frame = sampling.informed_frame(g["x_start"], g["x_goal"])
    res = t.run_sampling(iters, npw, pyw, irrt=irrt, frame=frame)
    assert res["iters_done"] == iters
    assert t.n == int(g["n"]) and np.array_equal(t.parents, g["parents"])
    assert np.array_equal(t.vertices, g["vertices"])   # (3D informed sampling: numpy 2.2's float64 np.sin / np.cos are libm's)
    if irrt:
        assert np.array_equal(t.solutions, g["path_solutions"])


@pytest.mark.parametrize("name,irrt", [("random_rrt2d", False), ("random_irrt2d", True), ("random_rrt3d", False), ("random_irrt3d", True)])
def test_planning_random_lists(oracle, name, irrt):
    from nirrt_star_amd import sampling
    g = load_golden(name)
    dim = int(g["dim"])
    t = make_oracle_tree(oracle, g)
    npw, pyw = _streams(g, 400000, 100000)
    frame = sampling.informed_frame(g["x_start"], g["x_goal"])
    r1 = t.run_sampling(int(g["iter_max"]), npw, pyw, irrt=irrt, goal_scan=not irrt, stop_first=True, frame=frame, want_trace=True)
    k1 = r1["iters_done"]
    r2 = t.run_sampling(int(g["iter_after_initial"]), npw[r1["np_used"]:], pyw[r1["py_used"]:], irrt=irrt, goal_scan=not irrt,
                        frame=frame, want_trace=True)
    lst = np.concatenate([r1["cost_trace"][:k1], r2["cost_trace"][: r2["iters_done"]]])
    exp = g["path_len_list"]
    assert len(lst) == len(exp)
    assert np.array_equal(np.isinf(lst), np.isinf(exp))
    m = np.isfinite(exp)
    assert np.max(np.abs(lst[m] - exp[m])) <= 1e-9 * np.max(exp[m])
    assert np.array_equal(t.parents, g["parents"])
