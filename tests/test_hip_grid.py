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
