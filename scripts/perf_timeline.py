"""Start / end device clocks of every tree of the default bench launch -> gpurun_out/timeline.npz (analysed offline:
how many trees are resident over time, when the last ones start, who ends last)."""
import os, sys
sys.path.insert(0, '.')
import numpy as np
import bench
from nirrt_star_amd import _hip, sampling

sys.argv = [sys.argv[0]] + sys.argv[1:]
args = bench.parse()
probs = bench.make_problems(args, 0)
B, iters = len(probs), args.iters
trees = []
for pr in probs:
    t = _hip.HipTree(args.dim, iters, pr["x_start"], pr["x_goal"], 10, pr["search_radius"], pr["clearance"], pr["env"])
    t.set_informed(*sampling.informed_frame(pr["x_start"], pr["x_goal"]))
    trees.append(t)
free_line = [not t.is_collision(pr["x_start"], pr["x_goal"]) for t, pr in zip(trees, probs)]
order = sorted(range(B), key=lambda b: (not free_line[b], b)) if args.free_first else list(range(B))
np_st, py_st = bench.problem_generators([pr["pid"] for pr in probs])
_hip.set_generators(trees, np_st, py_st)
hint = np.array([args.free_lanes if free_line[b] else 0 for b in order], dtype=np.int32) if args.free_lanes else None
r = _hip.run_sampling([trees[b] for b in order], iters, flags=_hip.F_IRRT if args.algo == "irrt" else 0, lanes_hint=hint)
st = r["stats"]
os.makedirs("gpurun_out", exist_ok=True)
np.savez("gpurun_out/timeline.npz", t0=st[:, 14], t1=st[:, 15], order=np.array(order), free=np.array(free_line), hint=hint if hint is not None else np.zeros(1), kernel_ms=r["kernel_ms"], stats=st)
print("kernel_ms", r["kernel_ms"])
