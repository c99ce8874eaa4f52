"""Per-phase profile of the bench batch split by problem class (free straight start-goal segment = the stragglers).
    NIRRT_HIP_SO=<profile build> python scripts/perf_heavy.py [B] [iters] [heavy|light|all] [variant-env]
Build the profile variant with:  NIRRT_EXTRA_FLAGS=-DNIRRT_PROFILE hipcc ... (scripts/build_variant.sh)"""
import sys
from types import SimpleNamespace
sys.path.insert(0, '.')
import numpy as np
import bench
from nirrt_star_amd import _hip, sampling

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
which = sys.argv[3] if len(sys.argv) > 3 else "heavy"
a = SimpleNamespace(algo="irrt", dim=2, world="b30", iters=iters, trees=B)
n_np, n_py = bench.word_budgets(a)
cache, pids, pid = {}, [], 0
trees, npw, pyw = [], [], []
while len(pids) < B and pid < 8192 * 4:
    pr = bench.make_problem(a, pid, cache)
    t = _hip.HipTree(2, iters, pr["x_start"], pr["x_goal"], 10, pr["search_radius"], 3, pr["env"])
    free = not t.is_collision(pr["x_start"], pr["x_goal"])
    if which == "all" or (which == "heavy") == free:
        t.set_informed(*sampling.informed_frame(pr["x_start"], pr["x_goal"]))
        w1, w2 = bench.problem_words(a, pid, n_np, n_py)
        trees.append(t); npw.append(w1); pyw.append(w2); pids.append(pid)
    else:
        del t
    pid += 1
print("class %s: %d problems out of the first %d" % (which, len(pids), pid))
res = _hip.run_sampling(trees, iters, npw, pyw, flags=_hip.F_IRRT)
st = res["stats"].astype(float)
secs = (st[:, 15] - st[:, 14]) / 1e8
print("kernel %.1f ms; per-tree s: mean %.2f median %.2f p90 %.2f max %.2f; %.2f M it/s" % (res["kernel_ms"], secs.mean(), np.median(secs),
      np.percentile(secs, 90), secs.max(), res["iters_done"].sum() / res["kernel_ms"] / 1e3))
tot_it = st[:, 13].sum()
print("per iteration: " + ", ".join("%s %.2f" % (n, st[:, j].sum() / tot_it) for j, n in enumerate(_hip.STAT_NAMES) if j not in (13, 14, 15, 17) and st[:, j].sum() > 0))
pr_ = np.array([t.debug_prof() for t in trees]).sum(0).astype(float)
if pr_.sum() > 0:
    names = ["nearest", "steer+edge", "query", "choose", "cost(new)", "rewire", "goal/ingoal", "report", "(R.collect)", "(R.rounds)", "(R.recost)", "", "rebuild", "(Q.visit)", "(Q.nearest)", "(Q.finish)", "L.draw", "L.iteration", "L.report", "L.other", "(Q.setup)", "(R.test)", "(R.relink)", ""]
    tot = pr_[16:20].sum() if pr_[16:20].sum() > 0 else pr_.sum()
    print("phase share: " + ", ".join("%s %.1f%%" % (n, 100 * v / tot) for n, v in zip(names, pr_) if v > 0),
          "| ticks/iter/tree %.0f (100MHz => %.1f us)" % (tot / tot_it, tot / tot_it / 100.0))
