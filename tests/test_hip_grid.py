"""GPU: the uniform-grid index (cell-ordered float32 twins + tail) must give the very same nearest / Near answers as
the whole scans.  The index normally starts at 2048 vertices; here it is forced on from 32 vertices with a rebuild
every 8 insertions and with several cell sizes, so every golden run (2D / 3D, RRT* / IRRT*, replayed and in-kernel
sampled) exercises cell rows, the tail, box widening and the bitmap ordering."""
import numpy as np
import pytest

from conftest import load_golden, make_oracle_tree
import test_hip_parity as P
import test_hip_sampling as S

pytestmark = pytest.mark.gpu

GRIDS = {"default-cells": None, "coarse-7": "7", "fine-40": "40"}


@pytest.fixture(params=list(GRIDS))
def forced_grid(request, monkeypatch):
    monkeypatch.setenv("NIRRT_GRID_MIN", "32")
    monkeypatch.setenv("NIRRT_GRID_REBUILD", "8")
    if GRIDS[request.param]:
        monkeypatch.setenv("NIRRT_GRID_G", GRIDS[request.param])
    return request.param


@pytest.mark.parametrize("name", P.RUNS)
def test_grid_step_replay(forced_grid, name):
    P.test_step_replay_device_steer(name)


def test_grid_resident_replay(forced_grid):
    P.test_resident_loop_replay_many_trees()


@pytest.mark.parametrize("name", ["run_rrt2d_3000", "run_rrt3d_3000"])
def test_grid_in_kernel_sampling_rrt(forced_grid, name):
    S.test_rrt_in_kernel_sample_free(name)


@pytest.mark.parametrize("name", ["run_irrt2d_800", "run_irrt2d_3000"])
def test_grid_in_kernel_sampling_irrt(forced_grid, name):
    S.test_irrt2d_in_kernel_informed_sampling(name)


@pytest.mark.parametrize("name", ["run_rrt2d_3000", "run_irrt3d_3000"])
def test_grid_primitives_after_a_run(oracle, forced_grid, name):
    """nearest / Near C-ABI primitives on a tree whose index was built by the loop itself (ordered part + tail)"""
    from nirrt_star_amd import _hip
    g = load_golden(name)
    dim = int(g["dim"])
    irrt = str(g["algo"]) == "irrt"
    t = P.make_hip_tree(g)
    if irrt:
        from nirrt_star_amd import sampling
        t.set_informed(*sampling.informed_frame(g["x_start"], g["x_goal"]))
    res = _hip.run_replay([t], g["samples"][None], flags=_hip.F_IRRT if irrt else 0)
    assert res["iters_done"][0] == len(g["samples"]) and res["status"][0] == 0
    v, p = t.download()
    n = len(v)
    assert np.array_equal(p, g["parents"])
    o = make_oracle_tree(oracle, g)
    o.load(v, p)
    rng = np.random.default_rng(11)
    hi = 224.0 if dim == 2 else 50.0
    qs = rng.uniform(-5.0, hi + 5.0, size=(300, dim))          # some queries outside the range box
    qs[:20] = 0.5 * (v[10:30] + v[40:60])                       # near-ties
    qs[20:40] = v[100:120]                                      # exact vertices
    for q in qs:
        assert t.nearest(q) == o.nearest(q)
    for q in qs[:150]:
        assert np.array_equal(t.near(q, n), o.near(q, n))
    for i in (5, 77, n - 1):
        assert np.array_equal(t.near(v[i], i), o.near(v[i], i))
    t.close()
    o.close()


def _problem(dim, seed):
    from nirrt_star_amd import worlds
    if dim == 2:
        pr = worlds.problem_2d(worlds.random_world_2d(seed, "b30"), 0)
        return pr, 3.0
    np.random.seed(seed)
    return worlds.problem_3d(worlds.random_world_3d(seed)), 2.0


@pytest.mark.parametrize("dim,irrt,iters", [(2, False, 12000), (2, True, 9000), (3, False, 12000), (2, True, 22000)])
def test_default_index_against_the_oracle_at_mid_size(oracle, dim, irrt, iters):
    """Default settings (index from 2048 vertices, rebuild every 1024, 128^2 / 16^3 cells): whole loops with in-kernel
    sampling on trees several times larger than the golden runs, HIP vs the C oracle fed with the same generator words -
    same vertex count, parents, solution list, generator words consumed; coordinates bit-equal."""
    from nirrt_star_amd import _hip, sampling
    pr, clr = _problem(dim, 11 + dim)
    t = _hip.HipTree(dim, iters, pr["x_start"], pr["x_goal"], 10, pr["search_radius"], clr, pr["env"])
    o = oracle.OracleTree(dim, iters, pr["x_start"], pr["x_goal"], 10.0, float(pr["search_radius"]), clr, pr["env_dict"])
    frame = sampling.informed_frame(pr["x_start"], pr["x_goal"])
    t.set_informed(*frame)
    rs = np.random.RandomState(77)
    npw = rs.randint(0, 1 << 32, size=iters * 40 + 4096, dtype=np.uint32)
    pyw = rs.randint(0, 1 << 32, size=iters * 24 + 4096, dtype=np.uint32) if (irrt and dim == 2) else None
    res = _hip.run_sampling([t], iters, [npw], [pyw] if pyw is not None else None, flags=_hip.F_IRRT if irrt else 0)
    ro = o.run_sampling(iters, npw, pyw, irrt=irrt, frame=frame)
    assert res["iters_done"][0] == iters == ro["iters_done"] and res["status"][0] == 0
    assert int(res["np_used"][0]) == ro["np_used"] and int(res["py_used"][0]) == ro["py_used"]
    v, p = t.download()
    assert len(v) == o.n > 4000                                   # well past the index threshold
    assert np.array_equal(p, o.parents)
    assert np.array_equal(v, o.vertices)
    if irrt:
        assert np.array_equal(t.solutions, o.solutions) and len(t.solutions) > 0
    t.close()
    o.close()
