#!/usr/bin/env python3
"""Diagnostic: the headline launch (B b30r16 problems x iters, time-sliced, own generators) against the oracle for the first N trees
of the dispatch order; prints every tree whose parents / solutions / word counts differ and how (scripts/diag_headline.py B iters N)."""
import os
import sys
from concurrent.futures import ThreadPoolExecutor
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import bench  # noqa: E402
from nirrt_star_amd import _hip, sampling  # noqa: E402
from oracle import oracle as orc  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
N = int(sys.argv[3]) if len(sys.argv) > 3 else 256
sliced = (sys.argv[4] != "plain") if len(sys.argv) > 4 else True
J0 = int(sys.argv[5]) if len(sys.argv) > 5 else 0      # first launch position to check
orc.build()
a = SimpleNamespace(algo="irrt", dim=2, world="b30r16", iters=iters, trees=B, scaling="weak")
cache = {}
probs = [bench.make_problem(a, pid, cache) for pid in range(B)]
trees = []
for pr in probs:
    t = _hip.HipTree(2, iters, pr["x_start"], pr["x_goal"], 10, pr["search_radius"], pr["clearance"], pr["env"])
    t.set_informed(*sampling.informed_frame(pr["x_start"], pr["x_goal"]))
    trees.append(t)
free = [not t.is_collision(pr["x_start"], pr["x_goal"]) for t, pr in zip(trees, probs)]
order = sorted(range(B), key=lambda b: (not free[b], b))
np_st, py_st = bench.problem_generators([probs[b]["pid"] for b in order])
launch = [trees[b] for b in order]
_hip.set_generators(launch, np_st, py_st)
res = _hip.run_sampling(launch, iters, flags=_hip.F_IRRT, want_trace=True, slice_iters=0 if sliced else -1)
print("launch done: kernel %.0f ms, %d trees, sliced=%s" % (res["kernel_ms"], B, sliced), flush=True)
n_np, n_py = bench.word_budgets(a)


def check(j):
    pr = probs[order[j]]
    npw = _hip.mt19937_outputs(np_st[j][0], np_st[j][1], n_np)[0]
    pyw = _hip.mt19937_outputs(py_st[j][0], py_st[j][1], n_py)[0]
    frame = sampling.informed_frame(pr["x_start"], pr["x_goal"])
    o = orc.OracleTree(2, iters, pr["x_start"], pr["x_goal"], 10.0, float(pr["search_radius"]), float(pr["clearance"]), pr["env_dict"])
    ro = o.run_sampling(iters, npw, pyw, irrt=True, frame=frame, want_trace=True)
    v, p = launch[j].download()
    out = None
    same_n = len(v) == o.n
    bad = (not same_n) or (not np.array_equal(p, o.parents)) or int(res["np_used"][j]) != ro["np_used"] or int(res["py_used"][j]) != ro["py_used"]
    if bad:
        m = min(len(v), o.n)
        dp = np.nonzero(p[:m] != o.parents[:m])[0]
        dv = np.nonzero((v[:m] != o.vertices[:m]).any(axis=1))[0]
        tr, tro = res["cost_trace"][j], ro["cost_trace"]
        fin = np.isfinite(tr) & np.isfinite(tro)
        dt = np.nonzero(tr[fin] != tro[fin])[0]
        out = dict(j=j, pid=pr["pid"], free=free[order[j]], n_gpu=len(v), n_orc=o.n, np_used=(int(res["np_used"][j]), ro["np_used"]),
                   py_used=(int(res["py_used"][j]), ro["py_used"]), parents_differ=len(dp), first_parent_diff=int(dp[0]) if len(dp) else -1,
                   last_parent_diff=int(dp[-1]) if len(dp) else -1,
                   vertices_differ_bits=len(dv), first_vertex_diff=int(dv[0]) if len(dv) else -1,
                   max_abs_vertex_diff=float(np.max(np.abs(v[:m] - o.vertices[:m]))),
                   trace_first_diff=int(dt[0]) if len(dt) else -1, trace_max_diff=float(np.max(np.abs(tr[fin] - tro[fin]))) if fin.any() else 0.0,
                   gpu_parent=[int(x) for x in p[dp[:4]]], orc_parent=[int(x) for x in o.parents[dp[:4]]], idx=[int(x) for x in dp[:4]])
    else:
        dv = np.nonzero((v != o.vertices).any(axis=1))[0]
        out = dict(j=j, ok=True, vertices_differ_bits=len(dv))
    o.close()
    return out


with ThreadPoolExecutor(min(128, os.cpu_count() or 8)) as ex:
    outs = list(ex.map(check, range(J0, min(J0 + N, B))))
bad = [o for o in outs if not o.get("ok")]
print("%d of %d trees differ from the oracle; trees with bit-different vertices among the identical ones: %d"
      % (len(bad), len(outs), sum(1 for o in outs if o.get("ok") and o["vertices_differ_bits"])))
for o in bad:
    print(o)
