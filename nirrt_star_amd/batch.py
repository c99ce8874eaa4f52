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
import os
import random

import numpy as np

from . import _hip
from . import pointcloud as pcu


def _untemper(y):
    """MT19937 outputs -> the state words they were tempered from (vectorised inverse of genrand's four xor-shifts)"""
    y = np.array(y, dtype=np.uint32)
    y ^= y >> np.uint32(18)
    y ^= (y << np.uint32(15)) & np.uint32(0xEFC60000)
    t = y.copy()
    for _ in range(4):
        t = y ^ ((t << np.uint32(7)) & np.uint32(0x9D2C5680))
    y = t
    t = y.copy()
    for _ in range(2):
        t = y ^ (t >> np.uint32(11))
    return t


class ProblemStreams:
    """the generator pair of ONE problem: numpy legacy RandomState + python Random (+ torch CPU generator for the FPS
    start indices of its PointNet++ forwards), all seeded like the reference seeds its process-global ones.

    The kernels consume raw 32-bit MT19937 outputs, a launch needs a window of them ahead of the current position, and a
    batch is resumed many times (cloud refreshes): the look-ahead is therefore generated ONCE and kept - on the host and as a
    resident copy in HBM that launches (and the device-side cloud generation) read in place (`window_np/py(n, device)` ->
    (address, count)).  What the device consumed is booked with advance_np/py(): the position inside the look-ahead moves at
    once, the host generator itself follows lazily - by an O(1) state jump - when somebody looks at it (`rs` / `py`: host-side
    cloud candidates, the end of a run) or the look-ahead has to be regenerated.  `rs` / `py` always return the generator at
    the problem's true position; if a caller draws from them, the window is found again inside the look-ahead by its next
    outputs."""

    def __init__(self, seed):
        self.seed = int(seed)
        self._rs = np.random.RandomState(self.seed)
        self._pyg = random.Random(self.seed)
        self._torch = None
        self._npc = {"host": None, "dev": None, "off": 0, "pending": 0, "touched": False}
        self._pyc = {"host": None, "dev": None, "off": 0, "pending": 0, "touched": False}

    # ---- the generators themselves ----
    @property
    def rs(self):
        self._settle(self._npc, True)
        self._npc["touched"] = True      # the caller may draw from it
        return self._rs

    @property
    def py(self):
        self._settle(self._pyc, False)
        self._pyc["touched"] = True
        return self._pyg

    def _set_from_outputs(self, is_np, last624):
        """any 624 consecutive outputs determine an MT19937 generator: state rebuilt from them, position 624 = "block used up\""""
        key = _untemper(last624)
        if is_np:
            st = self._rs.get_state(legacy=True)
            self._rs.set_state((st[0], key, 624, st[3], st[4]))
        else:
            st = self._pyg.getstate()
            self._pyg.setstate((st[0], tuple(int(v) for v in key) + (624,), st[2]))

    def _settle(self, c, is_np):
        """move the host generator past the outputs the device consumed since it was last looked at"""
        n = c["pending"]
        if not n:
            return
        c["pending"] = 0
        end = c["off"]            # already counts the pending outputs
        if c["host"] is not None and 624 <= end <= len(c["host"]):
            self._set_from_outputs(is_np, c["host"][end - 624:end])
        elif is_np:
            self._rs.randint(0, 1 << 32, size=int(n), dtype=np.uint32)
        else:
            self._pyg.getrandbits(32 * int(n))

    # ---- raw outputs ahead of the current position (nothing is consumed) ----
    def _gen(self, is_np, n):
        """the next n raw outputs of the numpy / python stream; the generator itself stays where it is.  Both are MT19937
        (CPython: getrandbits(32 k) = k consecutive outputs, least significant word first): the words come from the library's
        host-side generator (nirrt_mt19937_fill) started in a copy of the stream's state."""
        if is_np:
            st = self._rs.get_state(legacy=True)
            key, pos = st[1], int(st[2])
        else:
            st = self._pyg.getstate()[1]
            key, pos = np.array(st[:624], dtype=np.uint32), int(st[624])
        return _hip.mt19937_outputs(key, pos, n)[0]

    def _window(self, c, is_np, n, device):
        n = int(n)
        if c["touched"]:
            # somebody held the generator itself (cloud candidates drawn on the host): find its next outputs in the look-ahead
            c["touched"] = False
            if c["host"] is not None:
                found = -1
                probe = self._gen(is_np, 8)
                h, o = c["host"], c["off"]
                if np.array_equal(h[o:o + 8], probe):
                    found = o
                else:
                    for k in np.flatnonzero(h[o:len(h) - 7] == probe[0]):
                        if np.array_equal(h[o + k:o + k + 8], probe):
                            found = o + int(k)
                            break
                if found < 0:
                    c["host"] = c["dev"] = None
                else:
                    c["off"] = found
        if c["host"] is None or len(c["host"]) - c["off"] < n:
            # from the current position; replaces what was left.  Rare: windows shrink as a run proceeds, and the numpy
            # look-ahead carries headroom for the candidates of the cloud refreshes in between
            self._settle(c, is_np)
            c["host"] = self._gen(is_np, n + (min(n // 2, 1 << 20) + 131072 if is_np else 0))
            c["dev"] = None
            c["off"] = 0
        if device is None:
            return c["host"][c["off"]:c["off"] + n]
        if c["dev"] is None:
            import torch
            c["dev"] = torch.from_numpy(c["host"].view(np.int32)).to(device)
        return (c["dev"].data_ptr() + 4 * c["off"], n)

    def window_np(self, n, device=None):
        """the next n outputs of the numpy stream: host array, or (device address, n) of the resident copy"""
        return self._window(self._npc, True, n, device)

    def window_py(self, n, device=None):
        """the next n 32-bit outputs of the python stream"""
        return self._window(self._pyc, False, n, device)

    def prime(self, n_np, n_py, device):
        """produce and upload the look-ahead before a run starts (bench: inputs resident before the timed region)"""
        self.window_np(n_np, device)
        if n_py:
            self.window_py(n_py, device)

    def peek_np(self, n):
        self._settle(self._npc, True)
        return self._gen(True, n)

    def peek_py(self, n):
        self._settle(self._pyc, False)
        return self._gen(False, n)

    # ---- consuming: "n outputs later" ----
    def _advance(self, c, is_np, n):
        n = int(n)
        if not n:
            return
        if c["touched"]:               # position inside the look-ahead unknown until the next window: move the generator itself
            self._settle(c, is_np)
            if is_np:
                self._rs.randint(0, 1 << 32, size=n, dtype=np.uint32)
            else:
                self._pyg.getrandbits(32 * n)
            return
        c["pending"] += n
        if c["host"] is not None:
            c["off"] += n
            if c["off"] > len(c["host"]):      # ran past the look-ahead (cannot happen for windows handed out): settle by drawing
                over = c["off"] - len(c["host"])
                c["off"] = len(c["host"])
                c["pending"] -= over
                self._settle(c, is_np)
                if is_np:
                    self._rs.randint(0, 1 << 32, size=over, dtype=np.uint32)
                else:
                    self._pyg.getrandbits(32 * over)
                c["host"] = c["dev"] = None
                c["off"] = 0

    def advance_np(self, n):
        self._advance(self._npc, True, n)

    def advance_py(self, n):
        self._advance(self._pyc, False, n)

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
        self.device_clouds = os.environ.get("NIRRT_HOST_CLOUDS", "0") != "1"   # clouds generated on the device (0: the host path of round 2)
        self.device_input = os.environ.get("NIRRT_HOST_INPUT", "0") != "1"     # network input blocks assembled on the device
        self.calls = 0                      # PointNet++ forwards (batched ones count once)
        self.clouds_classified = 0
        self.seconds = {"candidates": 0.0, "downsample": 0.0, "classify": 0.0, "set_cloud": 0.0}   # host wall time per refresh stage

    # ---- cloud generation -------------------------------------------------------------------------------------------
    def _host_clouds(self, idx, problems, streams, c_best, frames):
        """candidates from the problem's own numpy generator on the host + one batched down-sampling launch (the 3D ellipsoid
        candidates go through sin / cos and stay with the host's libm; also the path when no resident look-ahead is wanted)"""
        from . import pointops
        cands = []
        for i in idx:
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
        return [c[m] for c, m in zip(cands, masks)]          # (n_b, 3) each

    def _device_jobs(self, idx, problems, streams, c_best, frames, dev):
        """nirrt_cloud_job of every problem in idx (2D: whole image / ellipse; 3D: whole box), reading the generator outputs from
        the problem's resident look-ahead at its current position"""
        import torch
        from . import pointops
        n_raw = self.n_points * self.scale
        n_words = 2 * (2 if self.dim == 2 else 3) * n_raw
        jobs = []
        for i in idx:
            pr = problems[i]
            j = pointops.CloudJob()
            addr, _ = streams[i].window_np(n_words, dev)
            j.words = addr
            if self.dim == 2:
                if "_free_tab_dev" not in pr:
                    pr["_free_tab_dev"] = torch.from_numpy(pcu.free_block_table(pr["binary_mask"])).to(dev)
                h, w = pr["binary_mask"].shape
                j.free_tab, j.w, j.h = pr["_free_tab_dev"].data_ptr(), int(w), int(h)
                if c_best[i] < np.inf:
                    xs, xg = np.asarray(pr["x_start"], dtype=np.float64), np.asarray(pr["x_goal"], dtype=np.float64)
                    j.mode = 1
                    if "_ellipse_frame" not in pr:
                        pr["_ellipse_frame"] = pcu.ellipse_frame_2d(xs, xg)      # (the SVD behind C is the same for every refresh)
                    for k, v in enumerate(pcu.ellipse_transform_2d(xs, xg, c_best[i] / frames[i][0], pr["_ellipse_frame"])):
                        j.a[k] = v
                else:
                    j.mode = 0
                    j.a[0], j.a[1] = float(w), float(h)
            else:
                env = pr["env"]
                if "_obs_dev" not in pr:
                    pr["_obs_dev"] = (torch.from_numpy(np.ascontiguousarray(np.asarray(env.obs_ball, dtype=np.float64).reshape(-1, 4))).to(dev),
                                      torch.from_numpy(np.ascontiguousarray(np.asarray(env.obs_box, dtype=np.float64).reshape(-1, 6))).to(dev))
                balls, boxes = pr["_obs_dev"]
                j.mode, j.balls, j.boxes, j.n_ball, j.n_box = 2, balls.data_ptr(), boxes.data_ptr(), int(balls.shape[0]), int(boxes.shape[0])
                lo = np.array([env.x_range[0] + 0, env.y_range[0] + 0, env.z_range[0] + 0], dtype=np.float64)
                hi = np.array([env.x_range[1] - 0, env.y_range[1] - 0, env.z_range[1] - 0], dtype=np.float64)
                diff = hi - lo
                for k in range(3):
                    j.a[k], j.a[3 + k] = float(lo[k]), float(diff[k])
                j.clearance = 0.0
            jobs.append(j)
        return jobs, n_raw, n_words

    def refresh(self, due, problems, trees, streams, c_best, frames):
        """new clouds for the trees `due` (indices into the batch): c_best[i] = inf draws the whole-world cloud
        (nirrt_star_png_2d.py:132-145), otherwise the ellipse / ellipsoid-restricted one (:146-160).  Clouds are generated, down-
        sampled and handed to the trees ON the device (candidates from each problem's resident generator look-ahead); the host
        sees them once, for the network's input block and the caller's records."""
        import time
        import torch
        from . import pointops
        if self.rate == 0:
            # update_point_cloud returns at once when pc_sample_rate == 0 (nirrt_star_png_2d.py:121-124): no candidates are
            # drawn (the problem's generators stay where they are), no forward; the trees only get the policy scalars
            empty = np.zeros((0, self.dim))
            for i in due:
                trees[i].set_cloud(empty, 0.0, self.ratio, c_best[i])
            return {i: (empty, np.zeros(0, dtype=np.int64)) for i in due}
        t0 = time.perf_counter()
        dev = torch.device("cuda", self.device_id)
        nd = len(due)
        on_dev = [j for j, i in enumerate(due) if self.device_clouds and not (self.dim == 3 and c_best[i] < np.inf)]
        on_host = [j for j in range(nd) if j not in set(on_dev)]
        clouds_dev = torch.zeros((nd, self.n_points, 3), dtype=torch.float64, device=dev)
        n_out = np.zeros(nd, dtype=np.int32)
        clouds = [None] * nd
        if on_dev:
            idx = [due[j] for j in on_dev]
            jobs, n_raw, n_words = self._device_jobs(idx, problems, streams, c_best, frames, dev)
            sub = clouds_dev if len(on_dev) == nd else torch.zeros((len(on_dev), self.n_points, 3), dtype=torch.float64, device=dev)
            n_cand, n_o = pointops.guidance_clouds(jobs, n_raw, self.n_points, sub, self.device_id)
            for k, j in enumerate(on_dev):
                i = due[j]
                if self.dim == 2 and not (c_best[i] < np.inf) and n_cand[k] < self.n_points:
                    raise ValueError("farthest_point_down_sample: %d candidates for %d samples (problem %d)" % (n_cand[k], self.n_points, i))
                streams[i].advance_np(n_words)      # what rng.random_sample / rng.uniform would have consumed
                n_out[j] = n_o[k]
            if sub is not clouds_dev:
                clouds_dev[torch.as_tensor(on_dev, device=dev)] = sub
            host = sub.cpu().numpy()
            for k, j in enumerate(on_dev):
                clouds[j] = host[k, : n_out[j], : self.dim]
        t1 = time.perf_counter()
        if on_host:
            pts = self._host_clouds([due[j] for j in on_host], problems, streams, c_best, frames)
            for k, j in enumerate(on_host):
                clouds[j] = pts[k][:, : self.dim]
                n_out[j] = len(pts[k])
                clouds_dev[j, : len(pts[k])] = torch.from_numpy(np.ascontiguousarray(pts[k]))
        t2 = time.perf_counter()
        xs_l = [np.asarray(problems[i]["x_start"], dtype=np.float64) for i in due]
        xg_l = [np.asarray(problems[i]["x_goal"], dtype=np.float64) for i in due]

        def fps_starts_for(group):   # group: positions inside `due`
            sizes = (int(n_out[group[0]]), 1024, 256, 64)
            return [torch.cat([streams[due[j]].fps_start(n) for j in group]) for n in sizes]

        preds = [None] * nd
        pred_dev = None
        if self.connect:
            res = self.wrapper.generate_connected_path_points_batch([c.astype(np.float32) for c in clouds], xs_l, xg_l, self.radius,
                                                                    self.max_trials, fps_starts_for)
            for j, (_, runs, mask) in enumerate(res):
                preds[j] = mask
            self.calls += max(r[1] for r in res) if res else 0
        elif self.device_input and hasattr(self.wrapper, "classify_device") and not (nd == 1 and getattr(self.wrapper, "use_graph", False)):
            # input blocks assembled from the resident clouds (k_net_input, bit-equal to the numpy evaluation below), predictions
            # stay on the device for set_cloud_batch; the host gets one copy of the prediction bytes for the caller's records
            s3, g3 = np.zeros((nd, 3)), np.zeros((nd, 3))
            for j in range(nd):
                s3[j, : self.dim], g3[j, : self.dim] = xs_l[j][: self.dim], xg_l[j][: self.dim]
            pred_dev = torch.zeros((nd, self.n_points), dtype=torch.uint8, device=dev)
            for size in sorted(set(int(v) for v in n_out)):
                grp = [j for j in range(nd) if n_out[j] == size]
                x = pointops.net_input(clouds_dev, grp, size, s3[grp], g3[grp], self.radius)
                pred = self.wrapper.classify_device(x, fps_starts=fps_starts_for(grp))
                pred_dev[torch.as_tensor(grp, device=dev), :size] = (pred != 0).to(torch.uint8)
                self.calls += 1
            pred_host = pred_dev.cpu().numpy().astype(np.int64)
            for j in range(nd):
                preds[j] = pred_host[j, : n_out[j]]
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
        self.clouds_classified += nd
        t3 = time.perf_counter()
        # path points into the trees: one launch for the whole batch (prediction bytes cross once)
        if pred_dev is None:
            pb = np.zeros((nd, self.n_points), dtype=np.uint8)
            for j in range(nd):
                pb[j, : n_out[j]] = np.asarray(preds[j]) != 0
            pred_dev = torch.from_numpy(pb).to(dev)
        _hip.set_cloud_batch([trees[i] for i in due], clouds_dev.data_ptr(), self.n_points * 3, n_out, pred_dev.data_ptr(), self.n_points,
                             self.rate, self.ratio, [c_best[i] for i in due])
        out = {i: (clouds[j], np.asarray(preds[j])) for j, i in enumerate(due)}
        t4 = time.perf_counter()
        for k, v in zip(("candidates", "downsample", "classify", "set_cloud"), (t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
            self.seconds[k] += v
        return out


def run_batch(trees, streams, iters, flags, dim, problems=None, guidance=None, frames=None, want_trace=True, stop_first=False,
              np_per_iter=None, py_per_iter=None, window=65536, init_clouds=True, pad=4096, overlap_min=None):
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
    import torch
    from concurrent.futures import ThreadPoolExecutor
    dev = torch.device("cuda", trees[0].device_id)

    import time
    prof = {"windows": 0.0, "wait_launch": 0.0, "book": 0.0, "refresh": 0.0}

    def launch(act):
        """windows of generator outputs (read in place from each problem's resident look-ahead) + the arguments of one
        persistent launch over the trees `act`"""
        rem = remaining[act].copy()
        t_w = time.perf_counter()
        npw = [streams[i].window_np((min(int(r), window) * np_per_iter + pad) * int(grow[i]), dev) for i, r in zip(act, rem)]
        pyw = [streams[i].window_py((min(int(r), window) * py_per_iter + pad) * int(grow[i]), dev) for i, r in zip(act, rem)] if need_py else None
        prof["windows"] += time.perf_counter() - t_w
        return lambda: _hip.run_sampling([trees[i] for i in act], int(rem.max()), npw, pyw, flags=run_flags, want_trace=want_trace,
                                         iters_each=rem, on_device=True)

    def absorb(act, r):
        """book a finished launch, refresh the clouds that are due; returns the trees of `act` that go on"""
        nonlocal kernel_ms, launches
        kernel_ms += r["kernel_ms"]
        launches += 1
        t_b = time.perf_counter()
        due = []
        for j, i in enumerate(act):
            d = int(r["iters_done"][j])
            streams[i].advance_np(int(r["np_used"][j]))
            if need_py:
                streams[i].advance_py(int(r["py_used"][j]))
            if want_trace and d:
                tr = r["cost_trace"][j, :d]
                traces[i].append(tr.copy())
                c_best[i] = tr[-1]
            remaining[i] -= d
            # delta slots add up; T0 / T1 (device clocks) and the best-cost bit pattern are absolute values of the launch:
            # first T0, last T1, last best cost
            first_launch = stats[i, _hip.ST_ITERS] == 0 and stats[i, _hip.ST_T0] == 0
            keep_t0 = stats[i, _hip.ST_T0]
            stats[i] += r["stats"][j]
            stats[i, _hip.ST_T0] = r["stats"][j, _hip.ST_T0] if first_launch else keep_t0
            stats[i, _hip.ST_T1] = r["stats"][j, _hip.ST_T1]
            stats[i, _hip.ST_CBEST] = r["stats"][j, _hip.ST_CBEST]
            st = int(r["status"][j])
            if st == _hip.E_CLOUD:
                # the cost the stopped kernel compared with ratio * c_update (find_best_path_solution), handed back with the
                # launch's counters: no extra launch per stopped tree
                c_best[i] = float(r["stats"][j, 17:18].view(np.float64)[0])
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
        t_r = time.perf_counter()
        prof["book"] += t_r - t_b
        if due:
            clouds.update(guidance.refresh(due, problems, trees, streams, c_best, frames))
        prof["refresh"] += time.perf_counter() - t_r
        return [i for i in act if not finished[i] and remaining[i] > 0]

    # Guided runs alternate between the loop and the cloud refresh.  Round 2 split a large batch into two halves so that one
    # half's (host-side) refresh overlapped the other half's launch.  With candidates, down-sampling, network input and
    # predictions all on the device there is little host work left to hide, and the halves cost more than they gave: a half
    # batch runs at two waves per SIMD instead of four, and PointNet++'s set-abstraction kernel (151 KB of LDS per workgroup)
    # cannot share a CU with resident trees, so its forward waited for the other half's launch anyway.  4096 trees, 2D, 50000
    # iterations: two halves 10.4 M it/s; one group 11.1 M (window 4096), 16.6 M (2048), 19.6 M (1024), 18.2 M (768), 16.1 M (512).
    # NIRRT_BATCH_GROUPS / NIRRT_BATCH_INFLIGHT bring the split back (groups launched from worker threads; the C call releases
    # the GIL); trees are independent and draw from their own generators, so no grouping changes a result.
    if overlap_min is None:
        overlap_min = int(os.environ.get("NIRRT_BATCH_OVERLAP_MIN", "1024"))   # trees per group when splitting; tests set 1
    # iterations per persistent launch: a tree whose cloud is due idles in its slot until the launch ends, so guided runs take
    # short launches
    if png and window > 1024:
        window = 1024
    window = int(os.environ.get("NIRRT_BATCH_WINDOW", window))
    n_groups = max(1, int(os.environ.get("NIRRT_BATCH_GROUPS", "1"))) if (png and B >= 2 * overlap_min) else 1
    in_flight = max(1, min(max(1, n_groups - 1), int(os.environ.get("NIRRT_BATCH_INFLIGHT", "1"))))   # launches on the device at once
    groups = [list(range(g, B, n_groups)) for g in range(n_groups)]
    futures = [None] * n_groups
    with ThreadPoolExecutor(max_workers=in_flight) as pool:
        for g in range(n_groups):
            if png and init_clouds:   # init_pc: the whole-world cloud before the first iteration (nirrt_star_png_2d.py:58)
                t_r = time.perf_counter()
                clouds.update(guidance.refresh(groups[g], problems, trees, streams, c_best, frames))
                prof["refresh"] += time.perf_counter() - t_r
            futures[g] = pool.submit(launch(groups[g]))
        while any(f is not None for f in futures):
            for g in range(n_groups):
                if futures[g] is None:
                    continue
                t_w = time.perf_counter()
                r = futures[g].result()
                prof["wait_launch"] += time.perf_counter() - t_w
                futures[g] = None
                groups[g] = absorb(groups[g], r)
                if groups[g]:
                    futures[g] = pool.submit(launch(groups[g]))
    return {"traces": [np.concatenate(t) if t else np.zeros(0) for t in traces], "iters_done": int(iters) - remaining,
            "kernel_ms": kernel_ms, "launches": launches, "stats": stats, "failed": failed, "clouds": clouds,
            "host_seconds": dict(prof, **(guidance.seconds if png else {}))}
