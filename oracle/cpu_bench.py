"""One process of bench.py's cpu_baseline leg (TEST / MEASUREMENT INFRASTRUCTURE, see oracle/oracle.py): the C restatement of
the reference loop (oracle/nirrt_oracle.c, incl. sampling from the problem's own seeded generators) on ONE host core, for
the first --iters iterations of problem --pid of bench.py's batch.  Prints one JSON line {iters, seconds, n}."""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--algo", default="irrt")
    ap.add_argument("--dim", type=int, default=2)
    ap.add_argument("--world", default="b30")
    ap.add_argument("--iters", type=int, default=20000)
    ap.add_argument("--cap", type=int, default=50000)
    ap.add_argument("--pid", type=int, default=0)
    ap.add_argument("--deadline-s", type=float, default=0.0, help="> 0: run in chunks of --chunk iterations and stop once this many seconds "
                                                                   "of loop time have passed (the iterations done so far are reported)")
    ap.add_argument("--chunk", type=int, default=2500)
    a = ap.parse_args()
    import bench
    from nirrt_star_amd import sampling
    from oracle import oracle as orc
    orc.build()
    ns = SimpleNamespace(algo=a.algo, dim=a.dim, world=a.world, iters=a.iters, trees=1)
    pr = bench.make_problem(ns, a.pid)
    n_np, n_py = bench.word_budgets(ns)
    npw, pyw = bench.problem_words(ns, a.pid, n_np, n_py)
    o = orc.OracleTree(a.dim, a.cap, pr["x_start"], pr["x_goal"], 10, pr["search_radius"], pr["clearance"], pr["env_dict"])
    frame = sampling.informed_frame(pr["x_start"], pr["x_goal"])
    t0 = time.perf_counter()
    if a.deadline_s > 0:
        # the same loop in chunks (the generators' outputs are consumed where the last chunk stopped: identical trees), so that a
        # slow problem can be cut off at the deadline and still report what it did
        done, np_at, py_at = 0, 0, 0
        while done < a.iters and time.perf_counter() - t0 < a.deadline_s:
            r = o.run_sampling(min(a.chunk, a.iters - done), npw[np_at:], pyw[py_at:] if pyw is not None else None, irrt=a.algo == "irrt", frame=frame)
            if int(r["iters_done"]) == 0:
                break
            done += int(r["iters_done"]); np_at += int(r["np_used"]); py_at += int(r["py_used"])
        dt = time.perf_counter() - t0
        print(json.dumps({"iters": done, "seconds": dt, "n": int(o.n), "solutions": int(len(o.solutions)), "complete": done >= a.iters}))
        return
    r = o.run_sampling(a.iters, npw, pyw, irrt=a.algo == "irrt", frame=frame)
    dt = time.perf_counter() - t0
    print(json.dumps({"iters": int(r["iters_done"]), "seconds": dt, "n": int(o.n), "solutions": int(len(o.solutions)), "complete": True}))


if __name__ == "__main__":
    main()
