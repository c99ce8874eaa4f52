"""The slowest trees of the default bench batch, alone: per-tree seconds, counters per iteration and (profile build) phase shares.
    NIRRT_HIP_SO=nirrt_star_amd/libnirrt_hip_prof.so NIRRT_FORCE_VARIANT=slim python scripts/perf_tail.py 1626,5727,1363,3362,2760 [world]"""
import os, sys
from types import SimpleNamespace
sys.path.insert(0, '.')
import numpy as np
import bench
from nirrt_star_amd import _hip, sampling

# "find:B" = the whole batch of B problems, slowest 8 printed; TAIL_DIM=3 for the 3D worlds
find = sys.argv[1].startswith("find:")
pids = list(range(int(sys.argv[1][5:]))) if find else [int(x) for x in sys.argv[1].split(",")]
world = sys.argv[2] if len(sys.argv) > 2 else "b30"
iters = int(os.environ.get("TAIL_ITERS", "50000"))
DIM = int(os.environ.get("TAIL_DIM", "2"))
a = SimpleNamespace(algo="irrt", dim=DIM, world=world, iters=iters, trees=0)
cache, trees = {}, []
for pid in pids:
    pr = bench.make_problem(a, pid, cache)
    t = _hip.HipTree(DIM, iters, pr["x_start"], pr["x_goal"], 10, pr["search_radius"], pr["clearance"], pr["env"])
    t.set_informed(*sampling.informed_frame(pr["x_start"], pr["x_goal"]))
    trees.append(t)
np_st, py_st = bench.problem_generators(pids)
_hip.set_generators(trees, np_st, py_st)
res = _hip.run_sampling(trees, iters, flags=_hip.F_IRRT)
st = res["stats"].astype(float)
secs = (st[:, 15] - st[:, 14]) / 1e8
names = _hip.STAT_NAMES
show = list(np.argsort(-secs)[:8]) if find else range(len(pids))
if find:
    print("kernel %.0f ms; per-tree s: mean %.2f median %.2f p99 %.2f max %.2f; slowest: %s" % (res["kernel_ms"], secs.mean(), np.median(secs), np.percentile(secs, 99), secs.max(), ",".join(str(pids[i]) for i in show)))
for i in show:
    pid = pids[i]
    print("pid %d: %.2f s n=%d | " % (pid, secs[i], trees[i].n) + ", ".join("%s %.1f" % (names[j], st[i, j] / iters) for j in list(range(13)) + [18, 19]))
pr_ = np.array([t.debug_prof() for t in trees]).astype(float)
pn = ["nearest", "steer+edge", "query", "choose", "cost(new)", "rewire", "goal/ingoal", "report", "(R.collect)", "(R.relink C)", "(R.recost)", "(R.recost.bfs)", "rebuild", "(Q.visit)", "(Q.nearest)", "(Q.finish)", "L.draw", "L.iteration", "L.report", "L.other", "(Q.setup)", "(R.test A)", "(R.block B)", "#bfs_levels"]
for row in pr_[:0] if find else pr_:
    tot = row[16:20].sum()
    if tot > 0:
        print("phase share: " + ", ".join("%s %.1f%%" % (n, 100 * v / tot) for n, v in zip(pn, row) if v > 0), "| us/iter %.1f" % (tot / iters / 100.0))
