import sys, time
sys.path.insert(0, '.')
import bench
from types import SimpleNamespace
from nirrt_star_amd import _hip
a = bench.parse(["--trees", "2048"])
t0 = time.perf_counter(); probs = bench.make_problems(a, 0); t1 = time.perf_counter()
trees = [_hip.HipTree(2, 50000, pr["x_start"], pr["x_goal"], 10, pr["search_radius"], pr["clearance"], pr["env"]) for pr in probs]
t2 = time.perf_counter()
fl = [t.is_collision(pr["x_start"], pr["x_goal"]) for t, pr in zip(trees, probs)]
t3 = time.perf_counter()
print("problems %.2f s, create %.2f s (%.2f ms per tree), collision probes %.2f s" % (t1 - t0, t2 - t1, (t2 - t1) / len(trees) * 1e3, t3 - t2))
