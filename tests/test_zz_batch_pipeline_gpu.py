"""GPU: the pipelined form of the batched guided planners (nirrt_star_amd/batch.py run_batch with two half-batches whose
launches are issued from a worker thread while the other half's clouds are refreshed) gives every tree the result of the
planner class run alone - the same check as tests/test_nirrt_batch_gpu.py::test_batched_guided_planners_equal_the_planner_class,
with the split forced onto a 4-problem batch (NIRRT_BATCH_GROUPS=2, NIRRT_BATCH_OVERLAP_MIN=1; by default a guided batch runs as
one group since round 3).  Reference: nirrt_star_png_2d.py:56-174 (planning / planning_random), eval_planning_2d.py:83-136.
(Named to run last: it is the one test of the suite that drives the library from two threads.)"""
import random
from types import SimpleNamespace as NS

import numpy as np
import pytest

from conftest import FakePNG

pytestmark = pytest.mark.gpu


class DiagonalFake(FakePNG):
    """deterministic, problem-independent labels (a point is "path" iff it lies near the world's main diagonal)"""

    def __init__(self):
        FakePNG.__init__(self, np.zeros(2), np.full(2, 224.0), 40.0)
        self.forwards = 0

    def classify_batch(self, clouds, start_masks, goal_masks, fps_starts=None):
        self.forwards += 1
        res = [self.classify_path_points(c, s, g_) for c, s, g_ in zip(clouds, start_masks, goal_masks)]
        return np.stack([r[0] for r in res]), np.stack([r[1] for r in res])


def test_two_pipelined_half_batches_equal_the_planner_class(monkeypatch):
    from nirrt_star_amd import batch, eval_sharded as es, planners, worlds
    monkeypatch.setenv("NIRRT_BATCH_OVERLAP_MIN", "1")
    monkeypatch.setenv("NIRRT_BATCH_GROUPS", "2")       # (round 3: one group is the default; the split is an option)
    probs = [worlds.problem_2d(worlds.random_world_2d(20 + i, "b30"), 0) for i in range(4)]
    pids = [11, 12, 13, 14]
    args = NS(problem="random_2d", planner="nirrt_star", iter_max=3000, iter_after_initial=200, step_len=10, clearance=3,
              pc_n_points=2048, pc_over_sample_scale=5, pc_sample_rate=0.5, pc_update_cost_ratio=0.9,
              connect_max_trial_attempts=5, root_dir=".", segment=1000)
    groups_seen = []
    inner = batch.Guidance.refresh

    def spying(self, due, *a, **k):
        groups_seen.append(tuple(due))
        return inner(self, due, *a, **k)

    monkeypatch.setattr(batch.Guidance, "refresh", spying)
    w = DiagonalFake()
    recs, traces = es.plan_batch(probs, pids, args, 0, wrapper=w)
    assert (0, 2) in groups_seen and (1, 3) in groups_seen      # init clouds went out per half: the split was active
    for pr, pid, rec, tr in zip(probs, pids, recs, traces):
        p = planners.NIRRTStarPNG2D(pr["x_start"], pr["x_goal"], 10, pr["search_radius"], 3200, pr["env_dict"], DiagonalFake(),
                                    pr["binary_mask"], 3, 2048, 5, 0.5, 0.9)
        np.random.seed(1000 + pid)
        random.seed(1000 + pid)
        lst = np.array(p.planning_random(200))
        tr = np.asarray(tr)
        assert len(tr) == len(lst) and np.array_equal(np.isinf(tr), np.isinf(lst)), "problem %d" % pid
        fin = np.isfinite(lst)
        assert fin.any() and np.max(np.abs(tr[fin] - lst[fin])) <= 1e-9
        assert rec[2] == p.num_vertices
