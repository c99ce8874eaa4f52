#!/usr/bin/env python3
"""Where an iteration's time goes on the bench workload: one step of bench.py's batch with the -DNIRRT_PROFILE build of the
library (per-phase wall_clock64 ticks per tree, TreeDev::prof) at full occupancy.

    scripts/build_variant.sh prof -DNIRRT_PROFILE
    NIRRT_HIP_SO=$PWD/nirrt_star_amd/libnirrt_hip_prof.so python scripts/perf_phases.py [bench.py arguments]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import bench  # noqa: E402

NAMES = ["nearest", "steer+edge", "query", "choose", "cost(new)", "rewire", "goal/ingoal", "report", "(R.collect)", "(R.rounds)", "(R.recost)",
         "(recost levels)", "rebuild", "(Q.visit)", "(Q.nearest)", "(Q.finish)", "L.draw", "L.iteration", "L.report", "L.other", "(Q.setup)", "", "", "(recost depth)"]


def main():
    args = bench.parse(sys.argv[1:])
    from nirrt_star_amd import _hip, batch, sampling
    probs = bench.make_problems(args, 0)
    D, iters = args.dim, args.iters
    flags = _hip.F_IRRT if args.algo == "irrt" else 0
    trees = []
    for pr in probs:
        t = _hip.HipTree(D, iters, pr["x_start"], pr["x_goal"], 10, pr["search_radius"], pr["clearance"], pr["env"])
        t.set_informed(*sampling.informed_frame(pr["x_start"], pr["x_goal"]))
        trees.append(t)
    order = list(range(len(trees)))
    if args.algo == "irrt":
        free = [not t.is_collision(pr["x_start"], pr["x_goal"]) for t, pr in zip(trees, probs)]
        order = sorted(order, key=lambda b: (not free[b], b))
    np_st, py_st = bench.problem_generators([pr["pid"] for pr in probs])
    _hip.set_generators(trees, np_st, py_st)
    r = batch.run_scheduled(trees, [iters], flags, order=order)
    done = float(r["iters_done"].sum())
    print("kernel %.0f ms, %.2f M it/s" % (r["kernel_ms"], done / r["kernel_ms"] / 1e3))
    pr_ = np.array([t.debug_prof() for t in trees]).astype(float)
    tot_t = pr_[:, 16:20].sum(axis=1)
    tot = tot_t.sum()
    if tot <= 0:
        print("no phase counters: not a -DNIRRT_PROFILE build")
        return
    s = pr_.sum(axis=0)
    print("ticks per iteration and tree: %.0f (100 MHz -> %.1f us)" % (tot / done, tot / done / 100.0))
    for n, v in zip(NAMES, s):
        if v > 0 and n:
            print("  %-16s %6.2f %%   %7.2f us/iter" % (n, 100 * v / tot, v / done / 100.0))
    # the same split for the slowest decile of the trees (they bound the launch)
    slow = np.argsort(tot_t)[-max(1, len(trees) // 10):]
    s2 = pr_[slow].sum(axis=0)
    t2 = tot_t[slow].sum()
    it2 = float(r["iters_done"][slow].sum()) if len(r["iters_done"]) == len(trees) else done / 10
    print("slowest decile: %.1f us/iter: " % (t2 / it2 / 100.0) + ", ".join("%s %.1f%%" % (n, 100 * v / t2) for n, v in zip(NAMES, s2) if v > 0.01 * t2 and n))


if __name__ == "__main__":
    main()
