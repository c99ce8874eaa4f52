import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from conftest import synthetic_checkpoint_root
from nirrt_star_amd import _hip, batch, png_wrapper, pointops, sampling, worlds
from nirrt_star_amd.png_wrapper import connect_rounds_device
dim = 3
np.random.seed(6)
probs = [worlds.problem_3d(worlds.random_world_3d(90 + i)) for i in range(4)]
w = png_wrapper.PNGWrapper3D(root_dir=synthetic_checkpoint_root(3), device="cuda")
w.use_graph = False
for cb_ratio in (None, 1.5, 1.1):
    trees, streams, frames = [], [], []
    for i, pr in enumerate(probs):
        t = _hip.HipTree(3, 100, pr["x_start"], pr["x_goal"], 10, pr["search_radius"], 2, pr["env"])
        trees.append(t); streams.append(batch.ProblemStreams(1000 + i)); frames.append(sampling.informed_frame(pr["x_start"], pr["x_goal"]))
    batch.hand_over(trees, streams)
    g = batch.Guidance(w, 3, 10, connect=True)
    dev = torch.device("cuda", 0)
    n_raw, n_words = g.cloud_words()
    words = torch.empty((4, n_words), dtype=torch.int32, device=dev)
    _hip.generator_words(trees, 0, n_words, device_ptr=words.data_ptr(), stride=n_words)
    cbest = [np.inf if cb_ratio is None else cb_ratio * frames[i][0] for i in range(4)]
    jobs, n_raw, nw = g._device_jobs(list(range(4)), probs, [words.data_ptr() + 4 * n_words * k for k in range(4)], cbest, frames, dev)
    clouds_dev = torch.zeros((4, g.n_points, 3), dtype=torch.float64, device=dev)
    n_cand, n_out = pointops.guidance_clouds(jobs, n_raw, g.n_points, clouds_dev, 0)
    host = clouds_dev.cpu().numpy()
    clouds = [host[k, : n_out[k], :3] for k in range(4)]
    xs = [np.asarray(p["x_start"], dtype=np.float64) for p in probs]; xg = [np.asarray(p["x_goal"], dtype=np.float64) for p in probs]
    def mk():
        ss = [batch.ProblemStreams(7 + i) for i in range(4)]
        def f(group):
            sizes = (int(n_out[group[0]]), 1024, 256, 64)
            return [torch.cat([ss[j].fps_start(n) for j in group]) for n in sizes]
        return f
    res = w.generate_connected_path_points_batch([c.astype(np.float32) for c in clouds], xs, xg, 10, 5, mk())
    has, runs, path = connect_rounds_device(w, clouds_dev, n_out, xs, xg, 10, 5, mk(), 3, 0)
    path = path.cpu().numpy()
    print("ratio", cb_ratio, "n_out", list(n_out))
    for j in range(4):
        print("  cloud", j, "host ok/runs/pts", res[j][0], res[j][1], int(res[j][2].sum()), "| dev", bool(has[j]), int(runs[j]), int(path[j, :n_out[j]].sum()),
              "same" if np.array_equal(res[j][2] > 0, path[j, :n_out[j]] > 0) else "DIFF %d" % int(((res[j][2] > 0) != (path[j, :n_out[j]] > 0)).sum()))
    for t in trees: t.close()
