"""GPU: the planner CLASSES (reference constructor/method signatures) against seeded golden runs.
`np.random.seed(s); random.seed(s)` then planner.planning() must give the reference's tree."""
import random
from types import SimpleNamespace

import numpy as np
import pytest

from conftest import FakePNG, load_golden

pytestmark = pytest.mark.gpu


def _make(g, mode=None, iter_max=None):
    from nirrt_star_amd import planners
    from nirrt_star_amd.env import Env, Env3D
    dim = int(g["dim"])
    algo = str(g["algo"])
    env = Env(g["env"]) if dim == 2 else Env3D(g["env"])
    cls = {("rrt", 2): planners.RRTStar2D, ("irrt", 2): planners.IRRTStar2D, ("rrt", 3): planners.RRTStar3D,
           ("irrt", 3): planners.IRRTStar3D}[(algo, dim)]
    clr = int(g["clearance"])
    return cls(tuple(g["x_start"]), tuple(g["x_goal"]), 10, float(g["search_radius"]),
               int(iter_max if iter_max is not None else g["iter_max"]), env, clr, mode=mode)


def _seed(g):
    np.random.seed(int(g["seed"]))
    random.seed(int(g["seed"]))


def _check_tree(p, g, exact):
    n = p.num_vertices
    assert n == int(g["n"])
    assert np.array_equal(p.vertex_parents[:n], g["parents"])
    assert np.array_equal(p.vertices[:n], g["vertices"])   # (the device's 2D steer and 3D informed sampler evaluate the reference's own libm functions)
    if np.isfinite(float(g["path_len"])):
        assert abs(p.get_path_len(p.path) - float(g["path_len"])) <= 1e-5
        assert p.check_success(p.path)
    else:
        assert len(p.path) == 0


@pytest.mark.parametrize("name", ["run_rrt2d_500", "run_rrt2d_3000", "run_irrt2d_3000", "run_rrt3d_3000", "run_irrt3d_3000"])
def test_planning_resident_mode(name, capsys):
    g = load_golden(name)
    p = _make(g)
    _seed(g)
    p.planning()
    # bit-exact where only IEEE ops are involved (3D steer + SampleFree); 3D informed sampling uses sin/cos
    _check_tree(p, g, exact=int(g["dim"]) == 3 and str(g["algo"]) == "rrt")
    if str(g["algo"]) == "irrt":
        assert np.array_equal(np.array(p.path_solutions), g["path_solutions"])
    # the global numpy generator was advanced exactly like the reference's loop would have: RRT* draws `dim` uniforms per
    # SampleFree attempt and nothing else, so the stream position after the run is the end of the attempt that produced
    # the fixture's last recorded sample (IRRT*: test_global_rng_state_after_resident_run_matches_host_loop)
    if str(g["algo"]) == "rrt":
        nxt_np = np.random.random_sample()
        D = int(g["dim"])
        lo = np.array([r[0] + p.clearance for r in p._ranges()], dtype=np.float64)
        hi = np.array([r[1] - p.clearance for r in p._ranges()], dtype=np.float64)
        _seed(g)
        attempts = np.random.random_sample(size=(8 * len(g["samples"]), D))
        hits = np.nonzero((lo + (hi - lo) * attempts == g["samples"][-1]).all(axis=1))[0]
        assert len(hits) >= 1
        _seed(g)
        np.random.random_sample(size=(int(hits[-1]) + 1, D))
        assert np.random.random_sample() == nxt_np


@pytest.mark.parametrize("name,mode,exact", [("run_rrt2d_500", "step", False), ("run_rrt2d_500", "exact", True),
                                             ("run_irrt2d_800", "exact", True), ("run_irrt3d_3000", "step", True)])
def test_planning_host_loop_modes(name, mode, exact):
    g = load_golden(name)
    p = _make(g, mode=mode)
    _seed(g)
    p.planning()
    _check_tree(p, g, exact=exact)


def test_global_rng_state_after_resident_run_matches_host_loop():
    g = load_golden("run_irrt2d_800")
    tails = []
    for mode in ("resident", "exact"):
        p = _make(g, mode=mode)
        _seed(g)
        p.planning()
        tails.append((np.random.random_sample(), random.random()))
    assert tails[0] == tails[1]


@pytest.mark.parametrize("name", ["random_rrt2d", "random_irrt2d", "random_rrt3d", "random_irrt3d"])
@pytest.mark.parametrize("mode", ["resident", "step"])
def test_planning_random_lists(name, mode):
    g = load_golden(name)
    p = _make(g, mode=mode)
    _seed(g)
    lst = np.array(p.planning_random(int(g["iter_after_initial"])))
    exp = g["path_len_list"]
    assert len(lst) == len(exp)
    assert np.array_equal(np.isinf(lst), np.isinf(exp))
    m = np.isfinite(exp)
    assert np.max(np.abs(lst[m] - exp[m])) <= 1e-5
    n = p.num_vertices
    assert n == int(g["n"]) and np.array_equal(p.vertex_parents[:n], g["parents"])


@pytest.mark.parametrize("name", ["run_nirrt2d_1500", "run_nirrtc2d_1500", "run_nirrt3d_1500", "run_nirrt3d_ratio1_1500", "run_nirrt2d_ratio1_1500"])
def test_nirrt_control_flow_with_fake_wrapper(name):
    """L3: guidance injected by a deterministic fake wrapper; cloud generation, refresh rule, 50/50 sampling mix
    and RNG consumption must reproduce the reference run.  The `ratio1` fixtures ran with pc_update_cost_ratio = 1.0, the
    default of demo_planning_3d.py:21: a refresh on every improvement of the best cost."""
    from nirrt_star_amd import planners
    g = load_golden(name)
    dim = int(g["dim"])
    w = FakePNG(g["x_start"], g["x_goal"], 25.0 if dim == 2 else 8.0)
    common = [tuple(g["x_start"]), tuple(g["x_goal"]), 10, float(g["search_radius"]), int(g["iter_max"]), g["env"], w]
    tail = [int(g["clearance"]), 2048, 5, 0.5, float(g["pc_update_cost_ratio"]) if "pc_update_cost_ratio" in g else 0.9]
    connect = str(g["algo"]) == "nirrt_c"
    if dim == 2:
        common.append(g["binary_mask"].astype(np.float64))
        cls = planners.NIRRTStarPNGC2D if connect else planners.NIRRTStarPNG2D
    else:
        cls = planners.NIRRTStarPNGC3D if connect else planners.NIRRTStarPNG3D
    for mode in ("exact", "resident"):
        w.calls = 0
        p = cls(*common, *tail, 5, mode=mode) if connect else cls(*common, *tail, mode=mode)
        _seed(g)
        p.planning()
        assert w.calls == int(g["png_calls"]), mode
        _check_tree(p, g, exact=(mode == "exact"))
        assert np.array_equal(np.array(p.path_solutions), g["path_solutions"])
        tail_rng = (np.random.random_sample(), random.random())
        if mode == "exact":
            ref_tail = tail_rng
        else:
            assert tail_rng == ref_tail   # both modes leave the global generators in the same state


def test_get_path_planner_factories_and_dropin_paths():
    import importlib
    import os
    import sys
    from nirrt_star_amd import worlds
    drop = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nirrt_star_amd", "dropin")
    sys.path.insert(0, drop)
    try:
        for m in list(sys.modules):
            if m == "datasets" or m.startswith("datasets."):
                del sys.modules[m]
        mod = importlib.import_module("path_planning_classes.irrt_star_2d")
        pu = importlib.import_module("datasets.planning_problem_utils_2d")
        pr = worlds.problem_2d(worlds.random_world_2d(3), 0)
        args = SimpleNamespace(step_len=10, iter_max=600, clearance=3)
        p = mod.get_path_planner(args, pr, None)
        np.random.seed(1)
        random.seed(1)
        p.planning()
        assert p.get_path_planner_name() == "IRRT* 2D" and p.num_vertices > 100
        assert pu.compute_gamma_rrt_star(pr["binary_mask"]) == pr["search_radius"]
    finally:
        sys.path.remove(drop)


def test_sharded_eval_batch_matches_planner_class():
    """eval_sharded.plan_batch (many problems per persistent launch) == planning_random of the planner class"""
    from types import SimpleNamespace as NS
    from nirrt_star_amd import eval_sharded as es, planners, worlds
    args = NS(problem="random_2d", planner="irrt_star", iter_max=4000, iter_after_initial=250, step_len=10, clearance=3)
    ed = [worlds.random_world_2d(i, "ref2d") for i in range(3)]
    probs = [worlds.problem_2d(e, 0) for e in ed]
    pids = [5, 6, 7]
    recs, _traces = es.plan_batch(probs, pids, args, 0)
    for pr, pid, rec in zip(probs, pids, recs):
        p = planners.IRRTStar2D(pr["x_start"], pr["x_goal"], 10, pr["search_radius"], 4250, pr["env"], 3)
        np.random.seed(1000 + pid)
        random.seed(1000 + pid)
        lst = np.array(p.planning_random(250))
        first = int(np.argmax(np.isfinite(lst))) + 1
        assert rec[0] == pid and rec[1] == first and rec[2] == p.num_vertices and rec[3] == len(lst)
        assert abs(rec[4] - lst[first - 1]) <= 1e-9 and abs(rec[5] - lst[first - 1 + 250]) <= 1e-9
    # the reference's result wire format: the full per-iteration lists
    for pr, pid, res in zip(probs, pids, es.result_lists("random_2d", _traces)):
        p = planners.IRRTStar2D(pr["x_start"], pr["x_goal"], 10, pr["search_radius"], 4250, pr["env"], 3)
        np.random.seed(1000 + pid)
        random.seed(1000 + pid)
        lst = np.array(p.planning_random(250))
        res = np.array(res)
        assert len(res) == len(lst) and np.array_equal(np.isinf(res), np.isinf(lst))
        assert np.max(np.abs(res[np.isfinite(lst)] - lst[np.isfinite(lst)])) <= 1e-9


@pytest.mark.parametrize("name", ["run_nrrt2d_1500", "run_nrrt3d_1500", "run_nrrtc2d_1500", "run_nrrtc3d_1500"])
def test_nrrt_png_against_reference_run(name):
    """NRRT*-PNG and NRRT*-PNG(C) (nrrt_star_png_2d.py:10-114, nrrt_star_png_c_2d.py, 3D twins):
    RRT* + point-cloud sampling, resident and host-loop modes"""
    from nirrt_star_amd import planners
    g = load_golden(name)
    dim = int(g["dim"])
    connect = str(g["algo"]) == "nrrt_c"
    w = FakePNG(g["x_start"], g["x_goal"], 25.0 if dim == 2 else 8.0)
    common = [tuple(g["x_start"]), tuple(g["x_goal"]), 10, float(g["search_radius"]), int(g["iter_max"]), g["env"], w]
    if dim == 2:
        common.append(g["binary_mask"].astype(np.float64))
    for mode in ("exact", "resident"):
        w.calls = 0
        if connect:
            cls = planners.NRRTStarPNGC2D if dim == 2 else planners.NRRTStarPNGC3D
            p = cls(*common, int(g["clearance"]), 2048, 5, 0.5, 5, mode=mode)
        else:
            cls = planners.NRRTStarPNG2D if dim == 2 else planners.NRRTStarPNG3D
            p = cls(*common, int(g["clearance"]), 2048, 5, 0.5, mode=mode)
        _seed(g)
        p.planning()
        assert w.calls == int(g["png_calls"]) == 1
        _check_tree(p, g, exact=(mode == "exact" or dim == 3))


def test_nirrt_planning_random_against_reference_run():
    """NIRRT*-PNG planning_random (nirrt_star_png_2d.py:247-335): per-iteration best cost list and final tree"""
    from nirrt_star_amd import planners
    g = load_golden("random_nirrt2d")
    w = FakePNG(g["x_start"], g["x_goal"], 25.0)
    p = planners.NIRRTStarPNG2D(tuple(g["x_start"]), tuple(g["x_goal"]), 10, float(g["search_radius"]), int(g["iter_max"]), g["env"], w,
                                g["binary_mask"].astype(np.float64), int(g["clearance"]), 2048, 5, 0.5, 0.9)
    _seed(g)
    lst = np.array(p.planning_random(int(g["iter_after_initial"])))
    exp = g["path_len_list"]
    assert len(lst) == len(exp) and np.array_equal(np.isinf(lst), np.isinf(exp))
    m = np.isfinite(exp)
    assert m.any() and np.max(np.abs(lst[m] - exp[m])) <= 1e-5
    n = p.num_vertices
    assert n == int(g["n"]) and np.array_equal(p.vertex_parents[:n], g["parents"]) and w.calls == int(g["png_calls"])


def test_sharded_eval_batch_of_rrt_star_matches_planner_class():
    """plan_batch with rrt_star (always SampleFree, in crowded worlds: r in [16, 24] circles) == the planner class; word
    windows are refilled by the batch driver, and a tree that stops abnormally raises instead of being recorded as unsolved"""
    from types import SimpleNamespace as NS
    from nirrt_star_amd import eval_sharded as es, planners, worlds
    args = NS(problem="random_2d", planner="rrt_star", iter_max=3000, iter_after_initial=150, step_len=10, clearance=3)
    probs = [worlds.problem_2d(worlds.random_world_2d(60 + i, "b30", circle_radius_range=(16, 24)), 0) for i in range(3)]
    pids = [21, 22, 23]
    recs, traces = es.plan_batch(probs, pids, args, 0)
    for pr, pid, tr in zip(probs, pids, traces):
        p = planners.RRTStar2D(pr["x_start"], pr["x_goal"], 10, pr["search_radius"], 3150, pr["env"], 3)
        np.random.seed(1000 + pid)
        random.seed(1000 + pid)
        lst = np.array(p.planning_random(150))
        tr = np.asarray(tr)
        assert len(tr) == len(lst) and np.array_equal(np.isinf(tr), np.isinf(lst))
        fin = np.isfinite(lst)
        assert np.max(np.abs(tr[fin] - lst[fin])) <= 1e-9 if fin.any() else True


@pytest.mark.parametrize("name", ["blockgap_irrt_block", "blockgap_rrt_gap"])
def test_planning_block_gap_against_reference_run(name):
    """planning_block_gap() on a block / gap problem built by OUR loader from the fixture's config: the per-iteration
    path lengths (until the threshold is met, or iter_max) and the final tree equal the reference's run."""
    from nirrt_star_amd import planners, problems
    import json
    g = load_golden(name)
    cfg = json.loads(str(g["config"]))
    pr = problems.get_block_problem_input(cfg) if str(g["kind"]) == "block" else problems.get_gap_problem_input(cfg)
    assert float(pr["search_radius"]) == float(g["search_radius"])
    cls = planners.IRRTStar2D if str(g["algo"]) == "irrt" else planners.RRTStar2D
    p = cls(pr["x_start"], pr["x_goal"], 10, pr["search_radius"], int(g["iter_max"]), pr["env"], 3)
    _seed(g)
    lst = np.array(p.planning_block_gap(float(g["threshold"])))
    exp = g["path_len_list"]
    assert len(lst) == len(exp)
    assert np.array_equal(np.isinf(lst), np.isinf(exp))
    m = np.isfinite(exp)
    if m.any():
        assert np.max(np.abs(lst[m] - exp[m])) <= 1e-5
    n = p.num_vertices
    assert n == int(g["n"]) and np.array_equal(p.vertex_parents[:n], g["parents"])
    assert np.array_equal(p.vertices[:n], g["vertices"])


def test_block_gap_batch_evaluation_is_segment_independent():
    """eval_sharded's block / gap protocol: the persistent loop in short segments with problems leaving the batch as
    they get below their threshold == one long launch truncated afterwards (same records)."""
    from nirrt_star_amd import eval_sharded as E, problems
    cfgs = problems.get_block_env_configs("/nonexistent")[:6]
    probs = [problems.get_block_problem_input(c) for c in cfgs]
    thr = [p["best_path_len"] * 1.1 for p in probs]
    out = []
    for seg in (250, 3000):
        args = SimpleNamespace(problem="block", planner="irrt_star", iter_max=3000, step_len=10, clearance=3, segment=seg)
        out.append(np.array(E.plan_batch_block_gap(probs, list(range(6)), thr, args, 0)[0]))
    assert np.array_equal(out[0][:, [0, 1, 3]], out[1][:, [0, 1, 3]])          # ids, first-solution and stop iterations
    fin = np.isfinite(out[1][:, 4:])
    assert np.array_equal(np.isfinite(out[0][:, 4:]), fin)
    assert np.array_equal(out[0][:, 4:][fin], out[1][:, 4:][fin])
    assert (out[1][:, 3] > 0).any()                                              # some problem met its threshold
