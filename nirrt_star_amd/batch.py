"""Many independent planning problems through the device-resident loop at once (the throughput path of eval_sharded and
bench.py): RRT* / IRRT* / NRRT* / NIRRT*[-C] trees advance together in persistent launches (one workgroup per tree), each fed
from ITS OWN seeded generator pair, exactly like one reference process per problem would consume
`np.random.seed(s); random.seed(s); torch.manual_seed(s)`.

What the host does between launches:
  * a tree whose word window ran dry (NIRRT_E_STREAM) gets the next window of its generators and resumes;
  * NIRRT* trees whose best cost dropped below pc_update_cost_ratio * c_update (NIRRT_E_CLOUD, nirrt_star_png_2d.py:114-116)
    get a new guidance cloud: candidates are drawn from the tree's OWN numpy stream (so the stream position is what the
    reference's single process would have), all due clouds are down-sampled in one launch (k_fps_f64, one workgroup per
    cloud) and classified in ONE batched PointNet++ forward (B = number of due trees) - or, for the -C planners, in batched
    neural-connect rounds;
  * everybody else just keeps its place.
Reference loop being batched: eval_planning_2d.py:83-136 calling planning_random / planning of nirrt_star_png_2d.py:56-174,
247-335, nirrt_star_png_c_2d.py:52-87 (+ the 3D twins).
"""
import random

import numpy as np

from . import _hip
from . import pointcloud as pcu


class ProblemStreams:
    """the generator pair of ONE problem: numpy legacy RandomState + python Random (+ torch CPU generator for the FPS
    start indices of its PointNet++ forwards), all seeded like the reference seeds its process-global ones"""

    def __init__(self, seed):
        self.seed = int(seed)
        self.rs = np.random.RandomState(self.seed)
        self.py = random.Random(self.seed)
        self._torch = None

    def peek_np(self, n):
        st = self.rs.get_state()
        w = self.rs.randint(0, 1 << 32, size=int(n), dtype=np.uint32)
        self.rs.set_state(st)
        return w

    def advance_np(self, n):
        if n:
            self.rs.randint(0, 1 << 32, size=int(n), dtype=np.uint32)

    def peek_py(self, n):
        n = int(n)
        st = self.py.getstate()
        bits = self.py.getrandbits(32 * n)
        self.py.setstate(st)
        return np.frombuffer(bits.to_bytes(4 * n, "little"), dtype="<u4").astype(np.uint32)

    def advance_py(self, n):
        if n:
            self.py.getrandbits(32 * int(n))

    def fps_start(self, n_points):
        """the reference's `torch.randint(0, N, (B,))` of one forward over ONE cloud (pointnet2_utils.py:77), from this
        problem's own torch generator"""
        import torch
        if self._torch is None:
            self._torch = torch.Generator().manual_seed(self.seed)
        return torch.randint(0, int(n_points), (1,), generator=self._torch, dtype=torch.long)


class Guidance:
    """NIRRT* / NRRT* point-cloud guidance of a batch: policy scalars + the (batched) cloud refresh"""

    def __init__(self, wrapper, dim, step_len, pc_n_points=2048, pc_over_sample_scale=5, pc_sample_rate=0.5,
                 pc_update_cost_ratio=0.9, connect=False, connect_max_trial_attempts=5, informed=True, device_id=0):
        self.wrapper = wrapper
        self.dim = dim
        self.radius = step_len              # pc_neighbor_radius = step_len (nirrt_star_png_2d.py:41)
        self.n_points = pc_n_points
        self.scale = pc_over_sample_scale
        self.rate = pc_sample_rate
        self.ratio = pc_update_cost_ratio if informed else 0.0   # NRRT*: the cloud is never refreshed
        self.connect = connect
        self.max_trials = connect_max_trial_attempts
        self.device_id = device_id
        self.calls = 0                      # PointNet++ forwards (batched ones count once)
        self.clouds_classified = 0

    def refresh(self, due, problems, trees, streams, c_best, frames):
        """new clouds for the trees `due` (indices into the batch): c_best[i] = inf draws the whole-world cloud
        (nirrt_star_png_2d.py:132-145), otherwise the ellipse / ellipsoid-restricted one (:146-160)"""
        from . import pointops
        cands = []
        for i in due:
            pr, rng = problems[i], streams[i].rs
            xs, xg = np.asarray(pr["x_start"], dtype=np.float64), np.asarray(pr["x_goal"], dtype=np.float64)
            cmax = c_best[i]
            n_raw = self.n_points * self.scale
            if self.dim == 2:
                if cmax < np.inf:
                    c = pcu.ellipsoid_candidates(xs, xg, cmax / frames[i][0], pr["binary_mask"], n_raw, rng)
                    c = np.concatenate([c, np.zeros((len(c), 1))], axis=1)
                    need_full = False
                else:
                    c = pcu.rectangle_candidates(pr["binary_mask"], self.n_points, self.scale, rng)
                    need_full = True      # the reference down-samples unconditionally here (open3d raises if it cannot)
            else:
                if cmax < np.inf:
                    c = pcu.ellipsoid_candidates_3d(xs, xg, cmax / frames[i][0], pr["env"], n_raw, 0, rng)
                else:
                    c = pcu.rectangle_candidates_3d(pr["env"], self.n_points, self.scale, 0, rng)
                need_full = False
            if need_full and len(c) < self.n_points:
                raise ValueError("farthest_point_down_sample: %d candidates for %d samples (problem %d)" % (len(c), self.n_points, i))
            cands.append(np.ascontiguousarray(c, dtype=np.float64))
        masks = pointops.farthest_point_down_sample_f64_batch(cands, self.n_points, self.device_id)
        clouds = [c[m][:, : self.dim] for c, m in zip(cands, masks)]
        xs_l = [np.asarray(problems[i]["x_start"], dtype=np.float64) for i in due]
        xg_l = [np.asarray(problems[i]["x_goal"], dtype=np.float64) for i in due]

        def fps_starts_for(group):   # group: positions inside `due`
            import torch
            sizes = (len(clouds[group[0]]), 1024, 256, 64)
            return [torch.cat([streams[due[j]].fps_start(n) for j in group]) for n in sizes]

        preds = [None] * len(due)
        if self.connect:
            res = self.wrapper.generate_connected_path_points_batch([c.astype(np.float32) for c in clouds], xs_l, xg_l, self.radius,
                                                                    self.max_trials, fps_starts_for)
            for j, (_, runs, mask) in enumerate(res):
                preds[j] = mask
            self.calls += max(r[1] for r in res) if res else 0
        else:
            sm = [pcu.get_point_cloud_mask_around_points(c, xs[np.newaxis, :], self.radius).astype(np.float32) for c, xs in zip(clouds, xs_l)]
            gm = [pcu.get_point_cloud_mask_around_points(c, xg[np.newaxis, :], self.radius).astype(np.float32) for c, xg in zip(clouds, xg_l)]
            for size in sorted(set(len(c) for c in clouds)):
                grp = [j for j, c in enumerate(clouds) if len(c) == size]
                pred, _ = self.wrapper.classify_batch([clouds[j].astype(np.float32) for j in grp], [sm[j] for j in grp],
                                                      [gm[j] for j in grp], fps_starts=fps_starts_for(grp))
                for jj, j in enumerate(grp):
                    preds[j] = pred[jj]
                self.calls += 1
        self.clouds_classified += len(due)
        out = {}
        for j, i in enumerate(due):
            path_pts = clouds[j][np.asarray(preds[j]).nonzero()[0]]
            trees[i].set_cloud(path_pts, self.rate, self.ratio, c_best[i])
            out[i] = (clouds[j], np.asarray(preds[j]))
        return out


def run_batch(trees, streams, iters, flags, dim, problems=None, guidance=None, frames=None, want_trace=True, stop_first=False,
              np_per_iter=None, py_per_iter=None, window=65536, init_clouds=True, pad=4096):
    """`iters` loop bodies for every tree of the batch (fewer for trees that stop: first solution with stop_first, full
    tree).  Returns dict(traces = per-tree best cost after each iteration, iters_done, kernel_ms, launches, stats).
    The per-tree generators in `streams` end up advanced by exactly what each tree consumed."""
    B = len(trees)
    irrt = bool(flags & _hip.F_IRRT)
    png = guidance is not None
    need_py = dim == 2 and irrt
    if np_per_iter is None:
        np_per_iter = (2 * dim * 4) if not irrt else (8 if dim == 2 else 6 * 40)
    if py_per_iter is None:
        py_per_iter = 16
    run_flags = flags | (_hip.F_STOP_FIRST if stop_first else 0) | (_hip.F_PNG if png else 0)
    remaining = np.full(B, int(iters), dtype=np.int64)
    grow = np.ones(B, dtype=np.int64)
    traces = [[] for _ in range(B)]
    c_best = np.full(B, np.inf)
    finished = np.zeros(B, dtype=bool)
    failed = {}
    kernel_ms, launches = 0.0, 0
    stats = np.zeros((B, _hip.N_STATS), dtype=np.int64)
    clouds = {}
    if png and init_clouds:   # init_pc: the whole-world cloud before the first iteration (nirrt_star_png_2d.py:58)
        clouds.update(guidance.refresh(list(range(B)), problems, trees, streams, c_best, frames))
    active = list(range(B))
    while active:
        rem = remaining[active]
        npw = [streams[i].peek_np((min(int(r), window) * np_per_iter + pad) * int(grow[i])) for i, r in zip(active, rem)]
        pyw = [streams[i].peek_py((min(int(r), window) * py_per_iter + pad) * int(grow[i])) for i, r in zip(active, rem)] if need_py else None
        r = _hip.run_sampling([trees[i] for i in active], int(rem.max()), npw, pyw, flags=run_flags, want_trace=want_trace,
                              iters_each=rem)
        kernel_ms += r["kernel_ms"]
        launches += 1
        due = []
        for j, i in enumerate(active):
            d = int(r["iters_done"][j])
            streams[i].advance_np(int(r["np_used"][j]))
            if need_py:
                streams[i].advance_py(int(r["py_used"][j]))
            if want_trace and d:
                tr = r["cost_trace"][j, :d]
                traces[i].append(tr.copy())
                c_best[i] = tr[-1]
            remaining[i] -= d
            stats[i] += r["stats"][j]
            st = int(r["status"][j])
            if st == _hip.E_CLOUD:
                c_best[i] = trees[i].best_solution()[0]   # the cost the stopped kernel compared with ratio * c_update
                due.append(i)
            elif st == _hip.E_STREAM:
                if d == 0:   # not even one draw fitted into the window: widen it (free space nearly empty ...)
                    grow[i] *= 4
                    if grow[i] > 4096:
                        failed[i] = "sampling cannot make progress (free space empty?)"
                        finished[i] = True
            elif st == _hip.E_CAPACITY:
                failed[i] = "tree capacity exceeded"
                finished[i] = True
            elif st == _hip.E_ARG:
                failed[i] = "empty predicted cloud (np.random.randint(0, 0) in the reference)"
                finished[i] = True
            elif st == 0:
                if remaining[i] <= 0 or (stop_first and d > 0 and np.isfinite(c_best[i])) or (stop_first and not want_trace):
                    finished[i] = True
            else:
                failed[i] = "device status %d" % st
                finished[i] = True
        if due:
            clouds.update(guidance.refresh(due, problems, trees, streams, c_best, frames))
        active = [i for i in active if not finished[i] and remaining[i] > 0]
    return {"traces": [np.concatenate(t) if t else np.zeros(0) for t in traces], "iters_done": int(iters) - remaining,
            "kernel_ms": kernel_ms, "launches": launches, "stats": stats, "failed": failed, "clouds": clouds}
