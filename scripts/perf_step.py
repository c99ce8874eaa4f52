"""Host-driven step API rate: python scripts/perf_step.py iters"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from nirrt_star_amd import _hip, worlds
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
pr = worlds.problem_2d(worlds.random_world_2d(0, "b30"), 0)
t = _hip.HipTree(2, iters, pr["x_start"], pr["x_goal"], 10, pr["search_radius"], 3, pr["env"])
rng = np.random.default_rng(1)
cand = rng.uniform(3, 221, size=(iters * 3, 2))
ins, _ = t.points_in_obs(cand)
s = cand[ins == 0][:iters]
t0 = time.time()
for k, q in enumerate(s):
    t.step(q, _hip.F_IRRT)
    if (k + 1) % 1000 == 0:
        print(k + 1, "%.0f it/s so far" % ((k + 1) / (time.time() - t0)))
