"""IRRT*/RRT* with in-kernel sampling: python scripts/perf_irrt.py B iters [dim] [algo] [words_per_iter]"""
import sys, time, random
sys.path.insert(0, '.')
import numpy as np
from nirrt_star_amd import _hip, worlds, sampling

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
dim = int(sys.argv[3]) if len(sys.argv) > 3 else 2
algo = sys.argv[4] if len(sys.argv) > 4 else "irrt"
wpi = int(sys.argv[5]) if len(sys.argv) > 5 else 16
flags = _hip.F_IRRT if algo == "irrt" else 0
trees, npw, pyw = [], [], []
t0 = time.time()
cache = {}
for b in range(B):
    if dim == 2:
        if b % 64 not in cache:
            cache[b % 64] = worlds.random_world_2d(b % 64, "b30")
        pr = worlds.problem_2d(cache[b % 64], (b // 64) % 4)
        clr = 3
    else:
        np.random.seed(b)
        pr = worlds.problem_3d(worlds.random_world_3d(b % 64))
        clr = 2
    t = _hip.HipTree(dim, iters, pr["x_start"], pr["x_goal"], 10, pr["search_radius"], clr, pr["env"])
    c_min, xc, C = sampling.informed_frame(pr["x_start"], pr["x_goal"])
    t.set_informed(c_min, xc, C)
    np.random.seed(1000 + b); random.seed(1000 + b)
    npw.append(sampling.peek_np_words(iters * wpi if dim == 3 else max(20000, iters * 2)))
    pyw.append(sampling.peek_py_words(iters * wpi) if dim == 2 else None)
    trees.append(t)
print("setup %.1fs" % (time.time() - t0))
res = _hip.run_sampling(trees, iters, npw, pyw if dim == 2 else None, flags=flags, want_trace=True)
ms = res["kernel_ms"]
done = res["iters_done"]
ns = [t.n for t in trees]
nsol = [len(t.solutions) for t in trees[:8]]
tr = res["cost_trace"]
first = [int(np.argmax(np.isfinite(tr[b]))) if np.isfinite(tr[b]).any() else -1 for b in range(B)]
print("B=%d iters=%d dim=%d %s: kernel %.1f ms -> %.0f it/s aggregate, %.0f it/s per tree; done min %d; status %s; n=%d..%d; nsol %s"
      % (B, iters, dim, algo, ms, done.sum() / (ms / 1e3), done.mean() / (ms / 1e3), done.min(), set(res["status"].tolist()), min(ns), max(ns), nsol))
print("first-solution iteration: median %s  (min %d max %d); final c_best median %.2f; np_used mean %.0f py_used mean %.0f; alg GB/s %.1f"
      % (np.median(first), min(first), max(first), np.median(tr[:, -1]), res["np_used"].mean(), res["py_used"].mean(),
         res["scan_elems"].sum() * dim * 8 / 1e9 / (ms / 1e3)))
st = res["stats"].astype(float)
tot_it = st[:, 13].sum()
print("per iteration: " + ", ".join("%s %.2f" % (n, st[:, j].sum() / tot_it) for j, n in enumerate(_hip.STAT_NAMES) if j not in (13, 14, 15) and st[:, j].sum() > 0))
ts = (st[:, 15] - st[:, 14]) / 1e8
print("per-tree seconds in the launch: mean %.3f median %.3f max %.3f; useful bytes/iter %.0f" % (ts.mean(), np.median(ts), ts.max(), _hip.useful_bytes(res["stats"], dim) / tot_it))
pr_ = np.array([t.debug_prof() for t in trees]).sum(0).astype(float)
if pr_.sum() > 0:
    names = ["nearest", "steer+edge", "query", "choose", "cost(new)", "rewire", "goal/ingoal", "report", "(R.collect)", "(R.rounds)", "(R.recost)", "", "rebuild", "(Q.visit)", "(Q.nearest)", "(Q.finish)", "L.draw", "L.iteration", "L.report", "L.other", "(Q.setup)", "", "", ""]
    tot = pr_[16:20].sum() if pr_[16:20].sum() > 0 else pr_.sum()
    print("phase share: " + ", ".join("%s %.1f%%" % (n, 100 * v / tot) for n, v in zip(names, pr_) if v > 0),
          "| ticks/iter/tree %.0f (100MHz => %.1f us)" % (tot / done.sum(), tot / done.sum() / 100.0))
