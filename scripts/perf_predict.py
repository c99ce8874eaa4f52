"""what predicts a slow tree?  4096 bench problems: pilot launch (first 5000 iterations) then the rest; per-tree seconds of
both + problem features -> rank correlations.  python scripts/perf_predict.py"""
import sys
from types import SimpleNamespace
sys.path.insert(0, '.')
import numpy as np
import bench
from nirrt_star_amd import _hip, sampling

B, iters, pilot = 4096, 50000, int(sys.argv[1]) if len(sys.argv) > 1 else 5000
a = SimpleNamespace(algo="irrt", dim=2, world="b30", iters=iters, trees=B)
n_np, n_py = bench.word_budgets(a)
trees, npw, pyw, cache, probs = [], [], [], {}, []
for pid in range(B):
    pr = bench.make_problem(a, pid, cache)
    t = _hip.HipTree(2, iters, pr["x_start"], pr["x_goal"], 10, pr["search_radius"], 3, pr["env"])
    t.set_informed(*sampling.informed_frame(pr["x_start"], pr["x_goal"]))
    w1, w2 = bench.problem_words(a, pid, n_np, n_py)
    trees.append(t); npw.append(w1); pyw.append(w2); probs.append(pr)
free = np.array([not t.is_collision(pr["x_start"], pr["x_goal"]) for t, pr in zip(trees, probs)])
cmin = np.array([np.hypot(*(np.array(pr["x_goal"], float) - np.array(pr["x_start"], float))) for pr in probs])
r1 = _hip.run_sampling(trees, pilot, npw, pyw, flags=_hip.F_IRRT, want_trace=True)
s1 = (r1["stats"][:, 15] - r1["stats"][:, 14]) / 1e8
cb1 = r1["cost_trace"][:, -1]
r2 = _hip.run_sampling(trees, iters - pilot, [w[u:] for w, u in zip(npw, r1["np_used"])], [w[u:] for w, u in zip(pyw, r1["py_used"])], flags=_hip.F_IRRT)
s2 = (r2["stats"][:, 15] - r2["stats"][:, 14]) / 1e8
k1 = r1["stats"][:, 2] / pilot
k2 = r2["stats"][:, 2] / (iters - pilot)
ratio1 = cb1 / cmin
def rankcorr(x, y):
    rx, ry = np.argsort(np.argsort(x)), np.argsort(np.argsort(y))
    return float(np.corrcoef(rx, ry)[0, 1])
print("pilot %d its: kernel %.0f ms, rest: kernel %.0f ms; rest per-tree s: mean %.2f median %.2f p99 %.2f max %.2f" % (pilot, r1["kernel_ms"], r2["kernel_ms"], s2.mean(), np.median(s2), np.percentile(s2, 99), s2.max()))
print("free-line share %.3f; rest-time mean free %.2f / blocked %.2f" % (free.mean(), s2[free].mean(), s2[~free].mean()))
for name, x in (("pilot seconds", s1), ("pilot members/iter", k1), ("c_best/c_min after pilot", -ratio1), ("free line", free.astype(float)), ("c_min", cmin)):
    print("rank corr(rest seconds, %s) = %.3f" % (name, rankcorr(s2, x)))
top = np.argsort(-s2)[:82]
for name, x in (("pilot seconds", s1), ("pilot members/iter", k1), ("-c_best/c_min", -ratio1)):
    pred = set(np.argsort(-x)[:164].tolist())
    print("top-82 slowest covered by top-164 of %s: %d" % (name, len(pred & set(top.tolist()))))
print("top-10 slowest: ", [(int(i), round(float(s2[i]), 1), round(float(s1[i]), 2), round(float(k1[i])), round(float(k2[i])), round(float(ratio1[i]), 4), bool(free[i])) for i in top[:10]])
