#!/usr/bin/env python3
"""Which trees end a time-sliced launch: busy time vs wall span of the slowest trees of one bench step (python scripts/perf_tail.py [bench args])"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import bench  # noqa: E402


def main():
    args = bench.parse(sys.argv[1:])
    from nirrt_star_amd import _hip, sampling
    probs = bench.make_problems(args, 0)
    D, iters = args.dim, args.iters
    flags = _hip.F_IRRT if args.algo == "irrt" else 0
    trees = []
    for pr in probs:
        t = _hip.HipTree(D, iters, pr["x_start"], pr["x_goal"], 10, pr["search_radius"], pr["clearance"], pr["env"])
        t.set_informed(*sampling.informed_frame(pr["x_start"], pr["x_goal"]))
        trees.append(t)
    free = [not t.is_collision(pr["x_start"], pr["x_goal"]) for t, pr in zip(trees, probs)]
    order = sorted(range(len(trees)), key=lambda b: (not free[b], b)) if args.algo == "irrt" else list(range(len(trees)))
    np_st, py_st = bench.problem_generators([probs[b]["pid"] for b in order])
    launch = [trees[b] for b in order]
    _hip.set_generators(launch, np_st, py_st)
    ahead = [1 if free[b] else 0 for b in order] if os.environ.get("RUN_AHEAD", "1") == "1" else None
    r = _hip.run_sampling(launch, iters, flags=flags, run_ahead=ahead)
    st = r["stats"].astype(np.float64)
    busy = st[:, _hip.ST_BUSY] / 1e8
    t0, t1 = st[:, 14] / 1e8, st[:, 15] / 1e8
    base = t0.min()
    print("kernel %.0f ms; busy mean %.3f max %.3f; last tree ends %.3f s after the first starts" % (r["kernel_ms"], busy.mean(), busy.max(), t1.max() - base))
    for j in np.argsort(-t1)[:12]:
        print("  launch pos %5d pid %5d free %d: starts %.3f ends %.3f busy %.3f  visited/it %.0f members/it %.0f rewire cand/it %.1f one-by-one %.2f"
              % (j, probs[order[j]]["pid"], free[order[j]], t0[j] - base, t1[j] - base, busy[j], st[j, 0] / iters, st[j, 2] / iters, st[j, 5] / iters, st[j, 19] / iters))


if __name__ == "__main__":
    main()
