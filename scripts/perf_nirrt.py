"""Where does NIRRT*-PNG(C) time go? python scripts/perf_nirrt.py [iters]"""
import sys, time, random, os
sys.path.insert(0, '.')
import numpy as np, torch
from nirrt_star_amd import planners, png_wrapper, worlds, pointcloud
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
ck = png_wrapper.checkpoint_path('/tmp/nirrt_ck', 2)
if not os.path.exists(ck):
    png_wrapper.make_synthetic_checkpoint(ck)
w = png_wrapper.PNGWrapper(root_dir='/tmp/nirrt_ck', device='cuda')
pr = worlds.problem_2d(worlds.random_world_2d(3, "ref2d"), 0)
T = {}
def timed(name, fn):
    def f(*a, **k):
        t0 = time.perf_counter(); r = fn(*a, **k); torch.cuda.synchronize(); T[name] = T.get(name, 0) + time.perf_counter() - t0; T[name + "#"] = T.get(name + "#", 0) + 1
        return r
    return f
pointcloud.farthest_point_down_sample = timed("fps_downsample", pointcloud.farthest_point_down_sample)
w.classify_path_points = timed("classify", w.classify_path_points)
for cls, extra in ((planners.NIRRTStarPNG2D, ()), (planners.NIRRTStarPNGC2D, (5,))):
    T.clear()
    p = cls(pr["x_start"], pr["x_goal"], 10, pr["search_radius"], iters, pr["env_dict"], w, pr["binary_mask"], 3, 2048, 5, 0.5, 0.9, *extra)
    p.update_point_cloud = timed("update_point_cloud", p.update_point_cloud)
    np.random.seed(1); random.seed(1); torch.manual_seed(1)
    import io, contextlib
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        p.planning()
    tot = time.perf_counter() - t0
    print(cls.__name__, "%d iters %.2f s; path %.2f; " % (iters, tot, p.get_path_len(p.path)) + ", ".join("%s %.3f s x%d" % (k, v, T[k + "#"]) for k, v in T.items() if not k.endswith("#")))
