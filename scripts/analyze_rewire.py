"""How do the members re-parented in one rewire pass depend on one another? (analysis build of the oracle: -DORC_REWIRE_STATS)
    gcc -O2 -fPIC -std=c11 -ffp-contract=off -mfma -shared -DORC_REWIRE_STATS -o /tmp/liborc_stats.so oracle/nirrt_oracle.c -lm
    python scripts/analyze_rewire.py <pid> [iters] [world]"""
import ctypes as C, os, sys, time
from types import SimpleNamespace
sys.path.insert(0, '.')
import numpy as np
import oracle.oracle as orc
orc._SO = "/tmp/liborc_stats.so"
orc.build = lambda force=False: orc._SO
import bench
from nirrt_star_amd import sampling

pid = int(sys.argv[1]); iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
world = sys.argv[3] if len(sys.argv) > 3 else "b30"
ns = SimpleNamespace(algo="irrt", dim=2, world=world, iters=iters, trees=1)
pr = bench.make_problem(ns, pid)
n_np, n_py = bench.word_budgets(ns)
npw, pyw = bench.problem_words(ns, pid, n_np, n_py)
o = orc.OracleTree(2, iters, pr["x_start"], pr["x_goal"], 10, pr["search_radius"], pr["clearance"], pr["env_dict"])
t0 = time.time()
r = o.run_sampling(iters, npw, pyw, irrt=True, frame=sampling.informed_frame(pr["x_start"], pr["x_goal"]))
st = (C.c_double * 16).in_dll(orc.lib(), "orc_rw_stats")
names = ["tested", "rewired", "flip_to_pass", "dependent", "dep_margin<1e-9", "dep_margin<1e-12", "dep_with_tie_between", "flip_to_fail", "dep_hops", "near_tie_fail", "passes_with_rewire", "passes_with_2+", "passes"]
print("pid %d, %d iterations, %.0f s, n=%d" % (pid, int(r["iters_done"]), time.time() - t0, o.n))
for n, v in zip(names, st):
    print("  %-22s %12.0f  (%.3f per iteration)" % (n, v, v / iters))
