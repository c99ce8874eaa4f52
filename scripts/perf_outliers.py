"""which trees of the bench batch are slow, and what did they do?  python scripts/perf_outliers.py [B] [iters]"""
import sys
from types import SimpleNamespace
sys.path.insert(0, '.')
import numpy as np
import bench
from nirrt_star_amd import _hip, sampling

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
if len(sys.argv) > 3 and "-" in sys.argv[3]:
    lo, hi = sys.argv[3].split("-")
    pids = list(range(int(lo), int(hi) + 1))
else:
    pids = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else list(range(B))
a = SimpleNamespace(algo="irrt", dim=2, world="b30", iters=iters, trees=len(pids))
n_np, n_py = bench.word_budgets(a)
trees, npw, pyw, cache = [], [], [], {}
for pid in pids:
    pr = bench.make_problem(a, pid, cache)
    t = _hip.HipTree(2, iters, pr["x_start"], pr["x_goal"], 10, pr["search_radius"], 3, pr["env"])
    t.set_informed(*sampling.informed_frame(pr["x_start"], pr["x_goal"]))
    w1, w2 = bench.problem_words(a, pid, n_np, n_py)
    trees.append(t); npw.append(w1); pyw.append(w2)
res = _hip.run_sampling(trees, iters, npw, pyw, flags=_hip.F_IRRT, want_trace=True)
st = res["stats"].astype(float)
secs = (st[:, 15] - st[:, 14]) / 1e8
order = np.argsort(-secs)
print("kernel %.1f ms; per-tree s: mean %.2f median %.2f p90 %.2f p99 %.2f max %.2f" % (res["kernel_ms"], secs.mean(), np.median(secs), np.percentile(secs, 90), np.percentile(secs, 99), secs.max()))
names = _hip.STAT_NAMES
med = np.median(st, axis=0)
print("median tree   : " + ", ".join("%s %.1f" % (names[j], med[j] / iters) for j in range(13)))
for i in order[:6]:
    tr = res["cost_trace"][i]
    first = int(np.argmax(np.isfinite(tr))) if np.isfinite(tr).any() else -1
    print("pid %d: %.2f s, n=%d nsol=%d first_sol_it=%d c_best=%.3f c_min=%.3f | " % (pids[i], secs[i], trees[i].n, len(trees[i].solutions), first, tr[-1],
          float(np.hypot(*(np.array(bench.make_problem(a, pids[i], cache)["x_goal"], float) - np.array(bench.make_problem(a, pids[i], cache)["x_start"], float)))))
          + ", ".join("%s %.1f" % (names[j], st[i, j] / iters) for j in range(13)))
