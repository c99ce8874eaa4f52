"""Which trees of the bench batch set the launch time, and where do they spend it?
    python scripts/perf_stragglers.py find [B]            -> gpurun_out/stragglers.json (slowest 32 pids of the default bench batch + stats)
    NIRRT_HIP_SO=<profile build> NIRRT_FORCE_VARIANT=slim python scripts/perf_stragglers.py prof  -> phase shares of those pids"""
import json, os, sys
from types import SimpleNamespace
sys.path.insert(0, '.')
import numpy as np
import bench
from nirrt_star_amd import _hip, sampling

mode = sys.argv[1]
iters = 50000
DIM = int(os.environ.get("STRAG_DIM", "2"))
ALGO = os.environ.get("STRAG_ALGO", "irrt")
a = SimpleNamespace(algo=ALGO, dim=DIM, world="b30", iters=iters, trees=0)
n_np, n_py = bench.word_budgets(a)
OUT = "gpurun_out/stragglers.json"


def setup(pids):
    cache, trees, npw, pyw = {}, [], [], []
    for pid in pids:
        pr = bench.make_problem(a, pid, cache)
        t = _hip.HipTree(DIM, iters, pr["x_start"], pr["x_goal"], 10, pr["search_radius"], pr["clearance"], pr["env"])
        t.set_informed(*sampling.informed_frame(pr["x_start"], pr["x_goal"]))
        w1, w2 = bench.problem_words(a, pid, n_np, n_py)
        trees.append(t); npw.append(w1); pyw.append(w2)
    return trees, npw, pyw


def report(res, pids, trees, top=32):
    st = res["stats"].astype(float)
    secs = (st[:, 15] - st[:, 14]) / 1e8
    order = np.argsort(-secs)
    print("kernel %.1f ms; per-tree s: mean %.2f median %.2f p90 %.2f p99 %.2f p99.9 %.2f max %.2f" % (res["kernel_ms"], secs.mean(), np.median(secs),
          np.percentile(secs, 90), np.percentile(secs, 99), np.percentile(secs, 99.9), secs.max()))
    names = _hip.STAT_NAMES
    for i in order[:top]:
        print("pid %d: %.2f s n=%d | " % (pids[i], secs[i], trees[i].n) + ", ".join("%s %.1f" % (names[j], st[i, j] / iters) for j in range(13)))
    return [int(pids[i]) for i in order[:top]], secs


if mode == "find":
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
    pids = list(range(B))
    trees, npw, pyw = setup(pids)
    free_line = [not t.is_collision(t_pr[0], t_pr[1]) for t, t_pr in zip(trees, [(bench.make_problem(a, p)["x_start"], bench.make_problem(a, p)["x_goal"]) for p in pids])] if False else None
    res = _hip.run_sampling(trees, iters, npw, pyw if DIM == 2 and ALGO == "irrt" else None, flags=_hip.F_IRRT if ALGO == "irrt" else 0)
    slow, secs = report(res, pids, trees)
    json.dump({"slow": slow, "secs": [float(s) for s in secs]}, open(OUT, "w"))
else:
    # pids: from the command line ("1626,5727,...") or the file a `find` run of the same gpurun call left
    slow = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 and "," in sys.argv[2] else json.load(open(OUT))["slow"]
    trees, npw, pyw = setup(slow)
    res = _hip.run_sampling(trees, iters, npw, pyw if DIM == 2 and ALGO == "irrt" else None, flags=_hip.F_IRRT if ALGO == "irrt" else 0)
    report(res, slow, trees, top=8)
    pr_ = np.array([t.debug_prof() for t in trees]).astype(float)
    names = ["nearest", "steer+edge", "query", "choose", "cost(new)", "rewire", "goal/ingoal", "report", "(R.collect)", "(R.rounds)", "(R.recost)", "", "rebuild", "(Q.visit)", "(Q.nearest)", "(Q.finish)", "L.draw", "L.iteration", "L.report", "L.other", "(Q.setup)", "(R.test)", "(R.relink)", ""]
    for row in (pr_.sum(0), pr_[0]):
        tot = row[16:20].sum()
        if tot > 0:
            print("phase share: " + ", ".join("%s %.1f%%" % (n, 100 * v / tot) for n, v in zip(names, row) if v > 0), "| us/iter %.1f" % (tot / iters / (len(trees) if row is not pr_[0] else 1) / 100.0))
