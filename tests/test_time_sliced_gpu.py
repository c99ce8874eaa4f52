"""GPU: time-sliced launches (k_run_pool, nirrt_run_args.slice_iters): a resident set of workgroups shares all trees of a batch
in slices of a few iterations, a tree's slices running on whatever workgroup (CU, XCD) is free.  Every result must equal the
launch with one workgroup per tree: trees, best-cost traces, iterations done, generator outputs consumed and final generator
states - with and without NIRRT_F_STOP_FIRST, with per-tree budgets, in 2D and 3D."""
import random

import numpy as np
import pytest

from conftest import load_golden
from test_hip_parity import make_hip_tree

pytestmark = pytest.mark.gpu


def _batch(g, B, iters, seed0):
    from nirrt_star_amd import _hip, sampling
    trees = [make_hip_tree(g, iter_max=iters) for _ in range(B)]
    frame = sampling.informed_frame(g["x_start"], g["x_goal"])
    for t in trees:
        t.set_informed(*frame)
    seeds = [int(g["seed"])] + [seed0 + i for i in range(B - 1)]
    _hip.set_generators(trees, [_hip.np_state(np.random.RandomState(s)) for s in seeds], [_hip.py_state(random.Random(s)) for s in seeds])
    return trees


@pytest.mark.parametrize("name,irrt,stop_first", [("run_irrt2d_3000", True, False), ("run_irrt2d_3000", True, True), ("run_rrt3d_3000", False, False),
                                                  ("run_irrt3d_3000", True, False)])
def test_time_sliced_launch_equals_one_workgroup_per_tree(name, irrt, stop_first, monkeypatch):
    from nirrt_star_amd import _hip
    g = load_golden(name)
    B, iters = 41, 3000
    flags = (_hip.F_IRRT if irrt else 0) | (_hip.F_STOP_FIRST if stop_first else 0)
    each = np.full(B, iters, dtype=np.int64)
    each[5], each[9], each[17], each[40] = 1, 0, 1234, 2999   # per-tree budgets (trees resumed after stopping at different iterations; 0 = not run at all)
    out = {}
    for mode in ("plain", "sliced"):
        if mode == "sliced":
            monkeypatch.setenv("NIRRT_POOL_RESIDENT", "7")     # 7 resident workgroups share the 41 trees
        else:
            monkeypatch.delenv("NIRRT_POOL_RESIDENT", raising=False)
        trees = _batch(g, B, iters, 7000)
        ahead = None if mode == "plain" else [1 if b % 5 == 0 else 0 for b in range(B)]     # (every fifth tree never waits for its turn)
        r = _hip.run_sampling(trees, iters, flags=flags, want_trace=True, iters_each=each, slice_iters=(-1 if mode == "plain" else 137),
                              run_ahead=ahead)
        st = _hip.get_generators(trees)
        out[mode] = (r, [t.download() for t in trees], [t.solutions for t in trees], st)
        for t in trees:
            t.close()
    (r0, d0, s0, g0), (r1, d1, s1, g1) = out["plain"], out["sliced"]
    assert np.array_equal(r0["iters_done"], r1["iters_done"]) and np.array_equal(r0["status"], r1["status"])
    assert np.array_equal(r0["np_used"], r1["np_used"]) and np.array_equal(r0["py_used"], r1["py_used"])
    for b in range(B):
        k = int(r0["iters_done"][b])
        assert np.array_equal(r0["cost_trace"][b, :k], r1["cost_trace"][b, :k])
        assert np.array_equal(d0[b][1], d1[b][1]) and np.array_equal(d0[b][0], d1[b][0])
        assert np.array_equal(s0[b], s1[b])
    for a, b_ in zip(g0, g1):
        assert np.array_equal(a, b_)
    if not stop_first:
        assert np.array_equal(d1[0][1], g["parents"])       # tree 0 is the reference's run
        assert r1["iters_done"][17] == 1234 and r1["iters_done"][5] == 1
    else:
        assert (r1["iters_done"] <= each).all() and (r1["iters_done"] < iters).any()
    # the counters add up over the slices; busy time is reported per tree
    assert np.array_equal(r0["stats"][:, 13], r1["stats"][:, 13]) and np.array_equal(r0["stats"][:, 9], r1["stats"][:, 9])
    assert (r1["stats"][each > 0, _hip.ST_BUSY] > 0).all() and r1["iters_done"][9] == 0 == r0["iters_done"][9]


def test_scheduled_segments_with_lane_groups_equal_one_launch():
    """batch.run_scheduled: a run as three launches with the trees re-ordered (longest first) and some of them moved to 256- /
    128-lane workgroups between the launches (concurrent lane groups on their own streams) equals the run as one launch: trees,
    iterations, generator outputs consumed - scheduling never changes a result"""
    from nirrt_star_amd import _hip, batch
    g = load_golden("run_irrt2d_3000")
    B, iters = 24, 3000
    out = {}
    for mode in ("one", "scheduled"):
        trees = _batch(g, B, iters, 9000)
        if mode == "one":
            r = batch.run_scheduled(trees, [iters], _hip.F_IRRT)
        else:
            # thresholds low enough that trees change lanes after the first segment (Near visits of a few hundred slots per iteration)
            r = batch.run_scheduled(trees, [700, 800, 1500], _hip.F_IRRT, wide_visits=400.0, narrow_visits=150.0)
            assert r["wide"] + r["narrow"] > 0
        st = _hip.get_generators(trees)
        out[mode] = (r, [t.download() for t in trees], [t.solutions for t in trees], st)
        for t in trees:
            t.close()
    (r0, d0, s0, g0), (r1, d1, s1, g1) = out["one"], out["scheduled"]
    assert np.array_equal(r0["iters_done"], r1["iters_done"]) and (r1["iters_done"] == iters).all() and not r1["status"].any()
    assert r0["words"] == r1["words"]
    for b in range(B):
        assert np.array_equal(d0[b][1], d1[b][1]) and np.array_equal(d0[b][0], d1[b][0]) and np.array_equal(s0[b], s1[b])
    for a, b_ in zip(g0, g1):
        assert np.array_equal(a, b_)
    assert np.array_equal(d1[0][1], g["parents"])
    assert (r1["seconds"] > 0).all() and np.array_equal(r0["stats"][:, 13], r1["stats"][:, 13])


def test_first_solution_on_the_last_iteration_of_a_slice_ends_the_run(monkeypatch):
    """NIRRT_F_STOP_FIRST with the first solution found on the LAST iteration of a time slice: the run has ended - a
    further slice would add an iteration (vertex, generator outputs, trace entry) that the launch with one workgroup per tree
    never runs.  The slice length is set to each of several trees' own first-solution iteration in turn."""
    from nirrt_star_amd import _hip
    g = load_golden("run_irrt2d_3000")
    B, iters = 12, 3000
    flags = _hip.F_IRRT | _hip.F_STOP_FIRST
    monkeypatch.delenv("NIRRT_POOL_RESIDENT", raising=False)
    trees = _batch(g, B, iters, 7000)
    r0 = _hip.run_sampling(trees, iters, flags=flags, want_trace=True, slice_iters=-1)
    d0 = [t.download() for t in trees]
    g0 = _hip.get_generators(trees)
    for t in trees:
        t.close()
    k_first = r0["iters_done"]
    assert (k_first < iters).sum() >= 3
    monkeypatch.setenv("NIRRT_POOL_RESIDENT", "5")
    tested = 0
    for b in np.argsort(k_first)[:6]:
        k1 = int(k_first[b])
        if k1 < 2 or k1 >= iters:
            continue
        for q in sorted({k1, k1 // 2 if k1 % 2 == 0 else k1}):      # the solving iteration = last of slice 0 (and of slice 1 when k1 is even)
            trees = _batch(g, B, iters, 7000)
            r1 = _hip.run_sampling(trees, iters, flags=flags, want_trace=True, slice_iters=q)
            assert np.array_equal(r0["iters_done"], r1["iters_done"]), (b, k1, q)
            assert np.array_equal(r0["np_used"], r1["np_used"]) and np.array_equal(r0["py_used"], r1["py_used"])
            for j in range(B):
                v, p = trees[j].download()
                assert np.array_equal(p, d0[j][1]) and np.array_equal(v, d0[j][0])
                k = int(r0["iters_done"][j])
                assert np.array_equal(r0["cost_trace"][j, :k], r1["cost_trace"][j, :k])
            g1 = _hip.get_generators(trees)
            for x, y in zip(g0, g1):
                assert np.array_equal(x, y)
            for t in trees:
                t.close()
            tested += 1
    assert tested >= 3
