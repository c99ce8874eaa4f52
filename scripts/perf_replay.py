"""Quick timing of the device-resident replay loop: python scripts/perf_replay.py B iters [dim]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from nirrt_star_amd import _hip, worlds

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
dim = int(sys.argv[3]) if len(sys.argv) > 3 else 2
kind = sys.argv[4] if len(sys.argv) > 4 else "b30"
trees, samples = [], np.zeros((B, iters, dim))
for b in range(B):
    if dim == 2:
        pr = worlds.problem_2d(worlds.random_world_2d(b % 16, kind), 0)
        clr, lo, hi = 3, 3.0, 221.0
    else:
        np.random.seed(b)
        pr = worlds.problem_3d(worlds.random_world_3d(b % 16))
        clr, lo, hi = 2, 2.0, 48.0
    t = _hip.HipTree(dim, iters, pr["x_start"], pr["x_goal"], 10, pr["search_radius"], clr, pr["env"])
    rng = np.random.default_rng(100 + b)
    # SampleFree-like: uniform then reject inside obstacles (done on the GPU primitive)
    cand = rng.uniform(lo, hi, size=(iters * 3, dim))
    ins, _ = t.points_in_obs(cand)
    samples[b] = cand[ins == 0][:iters]
    trees.append(t)
t0 = time.time()
res = _hip.run_replay(trees, samples, flags=0)
wall = time.time() - t0
ns = [t.n for t in trees]
ms = res["kernel_ms"]
tot = B * iters
bytes_alg = sum(8 * dim * n * n for n in ns)  # ~ sum_n 2*n*D*8 with n growing ~linearly to ns
print("B=%d iters=%d dim=%d kernel %.1f ms wall %.1f ms -> %.0f it/s aggregate, %.0f it/s per tree; n=%d..%d; alg %.2f GB -> %.1f GB/s"
      % (B, iters, dim, ms, wall * 1e3, tot / (ms / 1e3), iters / (ms / 1e3), min(ns), max(ns), bytes_alg / 1e9, bytes_alg / 1e9 / (ms / 1e3)))
