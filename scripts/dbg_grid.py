import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, random
from conftest import load_golden
from test_hip_parity import make_hip_tree
from nirrt_star_amd import _hip, sampling
name, ntrees, mode = sys.argv[1], int(sys.argv[2]), sys.argv[3]
g = load_golden(name)
dim = int(g["dim"]); iters = int(g["iter_max"])
ts = [make_hip_tree(g) for _ in range(ntrees)]
if mode == "sample":
    np.random.seed(int(g["seed"]))
    npw = sampling.peek_np_words(iters * dim * 2 * 4)
    res = _hip.run_sampling(ts, iters, [npw] * ntrees, None, flags=0, want_trace=False)
else:
    smp = np.ascontiguousarray(np.broadcast_to(g["samples"][None], (ntrees, iters, dim)))
    res = _hip.run_replay(ts, smp, flags=0, want_trace=False)
print("done", res["iters_done"][:3], res["status"][:3], flush=True)
v, p = ts[0].download()
print("n", len(v), int(g["n"]), "parents equal", np.array_equal(p, g["parents"]), flush=True)
