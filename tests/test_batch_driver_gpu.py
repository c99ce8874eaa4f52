"""GPU: the batch driver (nirrt_star_amd/batch.py) stops and resumes trees across launches - with launches of 40 iterations the
result still equals the uninterrupted run, and the problems' generators (resident in the trees) end at the same state."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_short_launches_resume_until_the_budget_is_spent():
    from nirrt_star_amd import _hip, batch, sampling, worlds
    probs = [worlds.problem_2d(worlds.random_world_2d(70 + i, "b30"), 0) for i in range(4)]
    out = []
    for window in (65536, 40):        # 40 iterations per launch -> fifty launches
        trees, streams = [], []
        for i, pr in enumerate(probs):
            t = _hip.HipTree(2, 2000, pr["x_start"], pr["x_goal"], 10, pr["search_radius"], 3, pr["env"])
            t.set_informed(*sampling.informed_frame(pr["x_start"], pr["x_goal"]))
            trees.append(t)
            streams.append(batch.ProblemStreams(500 + i))
        r = batch.run_batch(trees, streams, 2000, _hip.F_IRRT, 2, want_trace=True, window=window)
        assert not r["failed"] and (r["iters_done"] == 2000).all()
        out.append(([t.download() for t in trees], r["traces"], r["launches"], [s.rs.random_sample() for s in streams]))
        for t in trees:
            t.close()
    assert out[1][2] > 5 * out[0][2]
    for (va, pa), (vb, pb) in zip(out[0][0], out[1][0]):
        assert np.array_equal(pa, pb) and np.array_equal(va, vb)
    for ta, tb in zip(out[0][1], out[1][1]):
        assert np.array_equal(ta, tb)
    assert out[0][3] == out[1][3]     # the problems' generators end at the same position either way
