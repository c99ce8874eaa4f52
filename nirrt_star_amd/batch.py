"""Many independent planning problems through the device-resident loop at once (the throughput path of eval_sharded and
bench.py): RRT* / IRRT* / NRRT* / NIRRT*[-C] trees advance together in persistent launches (one workgroup per tree), each fed
from ITS OWN seeded generator pair, exactly like one reference process per problem would consume
`np.random.seed(s); random.seed(s); torch.manual_seed(s)`.

The generators themselves live in the trees (nirrt_set_generators): the loop and the cloud generation draw from them in
HBM, no generator output is produced on the host.  What the host does between launches:
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
import weakref

import numpy as np

from . import _hip
from . import pointcloud as pcu


class ProblemStreams:
    """the generator pair of ONE problem: numpy legacy RandomState + python Random (+ torch CPU generator for the FPS
    start indices of its PointNet++ forwards), all seeded like the reference seeds its process-global ones.

    The numpy / python generators LIVE ON THE DEVICE while a tree plans the problem: run_batch hands their states to the tree
    (nirrt_set_generators), the persistent loop and the device-side cloud generation draw from them in HBM (twist and
    tempering in the tree's wave; no output is ever produced on the host), and the host objects fall behind.  `rs` / `py`
    return the host generator AT THE PROBLEM'S TRUE POSITION: the state is fetched from the tree first
    (nirrt_get_generators), and because the caller may draw from it, it is handed back to the tree before the next launch
    (`push_if_touched`)."""

    def __init__(self, seed):
        self.seed = int(seed)
        self._rs = np.random.RandomState(self.seed)
        self._pyg = random.Random(self.seed)
        self._fps_rng = None
        self._tree = None                 # the tree that owns the states right now (None: the host objects are current)
        self._behind = [False, False]     # the device has drawn since the host object was last synchronised
        self._touched = [False, False]    # the host object was handed out since: the tree needs its state back

    # ---- ownership ----
    def bound_to(self, tree):
        return self._tree is tree

    def bind(self, tree):
        """(run_batch) the tree takes over; the caller uploads np_state() / py_state() in one batched call.  What a previous
        owner has drawn is fetched first: a stream that moves from tree A to tree B continues, it does not replay"""
        if self._tree is not None and self._tree is not tree:
            self.release()
        if self._tree is not tree:
            hooks = getattr(tree, "_release_hooks", None)      # (HipTree.close: the states come home before the tree is destroyed)
            if hooks is None:
                hooks = []
                try:
                    tree._release_hooks = hooks
                except AttributeError:
                    hooks = None
            if hooks is not None:
                # (a weak reference: the tree keeps no stream alive and no tree <-> stream cycle keeps a closed-over arena from
                #  being freed by reference counting)
                hooks.append(weakref.WeakMethod(self.release))
        self._tree = tree
        self._behind = [False, False]
        self._touched = [False, False]

    def release(self):
        """the host objects become current and own the states again (the end of a run: the tree may be closed afterwards)"""
        if self._tree is not None:
            if getattr(self._tree, "h", True):          # (a closed tree has nothing left to fetch)
                self._pull(0)
                self._pull(1)
            self._unhook()
            self._tree = None
        self._behind = [False, False]
        self._touched = [False, False]

    def _unhook(self):
        hooks = getattr(self._tree, "_release_hooks", None)
        if hooks:
            hooks[:] = [h for h in hooks if (h() if isinstance(h, weakref.WeakMethod) else h) not in (None, self.release)]

    def np_state(self):
        return _hip.np_state(self._rs)

    def py_state(self):
        return _hip.py_state(self._pyg)

    def device_drew(self, np_too=True, py_too=True):
        if self._tree is not None:
            self._behind[0] = self._behind[0] or np_too
            self._behind[1] = self._behind[1] or py_too

    def absorb(self, np_st=None, py_st=None):
        """states fetched from the tree (batched by the caller)"""
        if np_st is not None:
            _hip.set_np_state(np_st[0], np_st[1], self._rs)
            self._behind[0] = False
        if py_st is not None:
            _hip.set_py_state(py_st[0], py_st[1], self._pyg)
            self._behind[1] = False

    def _pull(self, which):
        if self._tree is not None and self._behind[which]:
            nk, npos, pk, ppos = _hip.get_generators([self._tree], want_np=which == 0, want_py=which == 1)
            if which == 0:
                self.absorb(np_st=(nk[0], npos[0]))
            else:
                self.absorb(py_st=(pk[0], ppos[0]))

    # ---- the generators themselves ----
    @property
    def rs(self):
        self._pull(0)
        self._touched[0] = self._tree is not None     # the caller may draw from it
        return self._rs

    @property
    def py(self):
        self._pull(1)
        self._touched[1] = self._tree is not None
        return self._pyg

    def touched(self):
        return self._touched[0] or self._touched[1]

    def fps_start(self, n_points):
        """the reference's `torch.randint(0, N, (B,))` of one forward over ONE cloud (pointnet2_utils.py:77), from this
        problem's own torch generator"""
        import torch
        return torch.from_numpy(self.fps_starts((n_points,)))

    def fps_starts(self, sizes):
        """the start indices of one forward's farthest-point samplings (`torch.randint(0, N, (1,))` per set-abstraction level, N in
        `sizes`) as an int64 array.  torch's CPU generator is MT19937 seeded by init_genrand(seed), and randint over a range below
        2^32 is `next 32-bit output % N`; the same outputs come from a numpy MT19937 with the same seeding, a call of which costs
        a tenth of four torch.randint calls (at pc_update_cost_ratio = 1.0 a batch makes > 100 000 forwards' worth of them per
        step).  tests/test_batch_host_logic.py compares the two generators draw by draw."""
        if self._fps_rng is None:
            if not 0 <= self.seed < 2 ** 32:
                raise ValueError("ProblemStreams: seeds of the torch generator twin must fit 32 bits")
            self._fps_rng = np.random.RandomState(self.seed)
        raw = self._fps_rng.randint(0, 2 ** 32, size=len(sizes), dtype=np.uint32).astype(np.int64)
        return raw % np.asarray(sizes, dtype=np.int64)


def hand_over(trees, streams, only_touched=False):
    """generator states host -> trees in ONE call: every stream not yet owned by its tree (a fresh problem), and every
    stream whose host object was handed out since the last launch (host-side cloud candidates)"""
    np_i, py_i = [], []
    for i, (t, s) in enumerate(zip(trees, streams)):
        if not s.bound_to(t):
            if only_touched:
                continue
            s.bind(t)
            np_i.append(i)
            py_i.append(i)
        else:
            if s._touched[0]:
                np_i.append(i)
            if s._touched[1]:
                py_i.append(i)
            s._touched = [False, False]
    py_set = set(py_i)
    both = [i for i in np_i if i in py_set]
    both_set = set(both)
    only_np = [i for i in np_i if i not in both_set]
    only_py = [i for i in py_i if i not in both_set]
    if both:
        _hip.set_generators([trees[i] for i in both], [streams[i].np_state() for i in both], [streams[i].py_state() for i in both])
    if only_np:
        _hip.set_generators([trees[i] for i in only_np], [streams[i].np_state() for i in only_np], None)
    if only_py:
        _hip.set_generators([trees[i] for i in only_py], None, [streams[i].py_state() for i in only_py])


def release_all(trees, streams):
    """the end of a batch: every problem's generators come home in ONE call per stream kind (HipTree.close would fetch them tree by
    tree - thousands of small synchronous copies for a large batch), after which the trees can be closed"""
    for which in (0, 1):
        idx = [i for i, s in enumerate(streams) if s._tree is trees[i] and s._behind[which] and getattr(trees[i], "h", None)]
        if idx:
            nk, npos, pk, ppos = _hip.get_generators([trees[i] for i in idx], want_np=which == 0, want_py=which == 1)
            for k, i in enumerate(idx):
                if which == 0:
                    streams[i].absorb(np_st=(nk[k], npos[k]))
                else:
                    streams[i].absorb(py_st=(pk[k], ppos[k]))
    for s in streams:
        s.release()      # (nothing left to fetch: bookkeeping only)


def fetch_np(trees, streams, idx):
    """the numpy generators of the problems `idx` brought up to date on the host in one call (host-side candidates)"""
    idx = [i for i in idx if streams[i]._tree is not None and streams[i]._behind[0]]
    if idx:
        nk, npos, _, _ = _hip.get_generators([trees[i] for i in idx], want_py=False)
        for k, i in enumerate(idx):
            streams[i].absorb(np_st=(nk[k], npos[k]))


def run_scheduled(trees, seg_len, flags, order=None, hint=None, wide_visits=0.0, narrow_visits=0.0, reorder=True, ahead=None):
    """RRT* / IRRT* on a batch whose trees draw from their own generators, as len(seg_len) persistent launches of seg_len[k]
    iterations each.  Between two launches the HOST re-schedules the independent problems from what the device measured in the
    launch before (problems, generators and results are untouched: a tree resumes exactly where it stopped):
      * dispatch order = longest first by the device time of the last segment - a launch lasts as long as its last tree, and
        trees dispatched late should be the short ones;
      * a tree whose fused nearest / Near visits covered >= wide_visits slots per iteration moves to a 256-lane workgroup
        (>= narrow_visits: 128 lanes): its visit is the whole iteration and scales with the lanes (a 3D tree with 14 000 visited
        slots per iteration: 450 us per iteration on one wave, 190 us on four).  Lane groups run at the same time.
    `order` / `hint`: dispatch order / lane hints (by position in `order`) of the FIRST launch.  `ahead`: per TREE (index into
    `trees`) flag - a tree known to be long (free straight start-goal segment) never waits for its turn in a time-sliced launch
    (nirrt_run_args.run_ahead).
    Returns the sums over the segments: kernel_ms, stats (B, N_STATS), alg_elems, iters_done, seconds, status, words; wide /
    narrow = trees on 256 / 128 lanes in the last launch."""
    B = len(trees)
    order = list(range(B)) if order is None else list(order)
    tot = {"kernel_ms": 0.0, "stats": np.zeros((B, _hip.N_STATS), dtype=np.int64), "alg_elems": np.zeros(B, dtype=np.int64),
           "iters_done": np.zeros(B, dtype=np.int64), "seconds": np.zeros(B), "status": np.zeros(B, dtype=np.int32), "words": 0,
           "wide": int(np.sum(np.asarray(hint) == 256)) if hint is not None else 0,
           "narrow": int(np.sum(np.asarray(hint) == 128)) if hint is not None else 0}
    for si, n_it in enumerate(seg_len):
        live = [b for b in order if tot["status"][b] == 0]
        if not live:
            break
        if len(live) != len(order):      # (a stopped tree leaves the batch; hints follow their trees)
            keep = {b: (hint[j] if hint is not None else 0) for j, b in enumerate(order)}
            order = live
            hint = np.array([keep[b] for b in order], dtype=np.int32) if hint is not None else None
        r = _hip.run_sampling([trees[b] for b in order], int(n_it), flags=flags, lanes_hint=hint,
                              run_ahead=None if ahead is None else [1 if ahead[b] else 0 for b in order])
        idx = np.asarray(order)
        secs = r["stats"][:, _hip.ST_BUSY] / 1e8      # (device time inside the loop; a time-sliced launch idles a tree between its slices)
        tot["kernel_ms"] += r["kernel_ms"]
        tot["stats"][idx] += r["stats"]
        tot["alg_elems"][idx] += r["alg_elems"]
        tot["iters_done"][idx] += r["iters_done"]
        tot["seconds"][idx] += secs
        tot["status"][idx] = r["status"]
        tot["words"] += int(r["np_used"].sum()) + int(r["py_used"].sum())
        if si + 1 < len(seg_len) and B > 1:
            visits = r["stats"][:, 0] / np.maximum(1, r["stats"][:, _hip.ST_ITERS])
            rank = np.argsort(-secs, kind="stable") if reorder else np.arange(len(order))
            order = [order[j] for j in rank]
            h = np.zeros(len(order), dtype=np.int32)
            if wide_visits > 0:
                h[visits[rank] >= wide_visits] = 256
            if narrow_visits > 0:
                h[(visits[rank] >= narrow_visits) & (h == 0)] = 128
            hint = h if h.any() else None
            tot["wide"], tot["narrow"] = int(np.sum(h == 256)), int(np.sum(h == 128))
    return tot


class Guidance:
    """NIRRT* / NRRT* point-cloud guidance of a batch: policy scalars + the (batched) cloud refresh"""

    def __init__(self, wrapper, dim, step_len, pc_n_points=2048, pc_over_sample_scale=5, pc_sample_rate=0.5,
                 pc_update_cost_ratio=0.9, connect=False, connect_max_trial_attempts=5, informed=True, device_id=0,
                 host_ellipsoid_3d=None):
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
        # parity runs: the 3D ellipsoid candidates (mode 3 of nirrt_guidance_clouds) go through the device's sin / cos and equal the
        # host's only to a few ulp - a candidate ON an obstacle / range boundary can flip.  NIRRT_HOST_ELLIPSOID3D=1 (or
        # `host_ellipsoid_3d=True`) draws exactly those clouds with numpy / glibc on the host (bit-reproducible, slower).
        self.host_ellipsoid_3d = host_ellipsoid_3d if host_ellipsoid_3d is not None else os.environ.get("NIRRT_HOST_ELLIPSOID3D", "0") == "1"
        self.calls = 0                      # PointNet++ forwards (batched ones count once)
        self.clouds_classified = 0
        self.seconds = {"candidates": 0.0, "downsample": 0.0, "classify": 0.0, "set_cloud": 0.0}   # host wall time per refresh stage
        self.size_log = [] if os.environ.get("NIRRT_REFRESH_LOG", "0") == "1" else None
        self.ragged = os.environ.get("NIRRT_RAGGED_FORWARD", "1") == "1"      # clouds of different sizes in ONE forward (refresh)

    # ---- cloud generation -------------------------------------------------------------------------------------------
    def _host_clouds(self, idx, problems, trees, streams, c_best, frames):
        """candidates from the problem's own numpy generator on the host + one batched down-sampling launch (NIRRT_HOST_CLOUDS=1:
        the host path the reference-generated fixtures pin).  The generators come back from their trees first and return
        there before the next launch (hand_over)."""
        from . import pointops
        fetch_np(trees, streams, idx)
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

    def _side_streams(self, dev, n):
        """streams for the forwards of one refresh (NIRRT_REFRESH_STREAMS, default 1 = everything on the current stream)"""
        import torch
        want = min(n, max(0, int(os.environ.get("NIRRT_REFRESH_STREAMS", "1"))))
        if want <= 1:
            return []
        pool = self.__dict__.setdefault("_streams", {})
        key = str(dev)
        while len(pool.setdefault(key, [])) < want:
            pool[key].append(torch.cuda.Stream(device=dev))
        return pool[key][:want]

    def cloud_words(self):
        """generator outputs one cloud's candidates consume: n_raw points x dim doubles x 2 words"""
        n_raw = self.n_points * self.scale
        return n_raw, 2 * (2 if self.dim == 2 else 3) * n_raw

    def _device_jobs(self, idx, problems, word_addr, c_best, frames, dev):
        """nirrt_cloud_job of every problem in idx (2D: whole image / ellipse; 3D: whole box / ellipsoid) as one job table
        (pointops.cloud_job_table); word_addr[k] = device address of the generator outputs job k reads (nirrt_generator_words of
        the problem's tree).  What depends on the problem alone (device copies of its obstacle tables, the rotation C of its
        start-goal frame, its range) is made once per problem and kept in the problem's dict; the per-refresh part of a batch is
        filled column by column."""
        import torch
        from . import pointops
        n_raw, n_words = self.cloud_words()
        n = len(idx)
        jobs = pointops.cloud_job_table(n)
        jobs["words"] = np.asarray(word_addr, dtype=np.uint64)
        cb = np.array([c_best[i] for i in idx], dtype=np.float64)
        finite = cb < np.inf
        if self.dim == 2:
            for k, i in enumerate(idx):
                pr = problems[i]
                if "_free_tab_dev" not in pr:
                    pr["_free_tab_dev"] = torch.from_numpy(pcu.free_block_table(pr["binary_mask"])).to(dev)
                h, w = pr["binary_mask"].shape
                j = jobs[k]
                j["free_tab"], j["w"], j["h"] = pr["_free_tab_dev"].data_ptr(), int(w), int(h)
                if finite[k]:
                    xs, xg = np.asarray(pr["x_start"], dtype=np.float64), np.asarray(pr["x_goal"], dtype=np.float64)
                    j["mode"] = 1
                    if "_ellipse_frame" not in pr:
                        pr["_ellipse_frame"] = pcu.ellipse_frame_2d(xs, xg)      # (the SVD behind C is the same for every refresh)
                    j["a"][:6] = pcu.ellipse_transform_2d(xs, xg, c_best[i] / frames[i][0], pr["_ellipse_frame"])
                else:
                    j["mode"] = 0
                    j["a"][0], j["a"][1] = float(w), float(h)
        else:
            consts = []
            for i in idx:
                pr = problems[i]
                c = pr.get("_cloud_const_3d")
                if c is None:
                    env = pr["env"]
                    balls = torch.from_numpy(np.ascontiguousarray(np.asarray(env.obs_ball, dtype=np.float64).reshape(-1, 4))).to(dev)
                    boxes = torch.from_numpy(np.ascontiguousarray(np.asarray(env.obs_box, dtype=np.float64).reshape(-1, 6))).to(dev)
                    pr["_obs_dev"] = (balls, boxes)
                    lo = np.array([env.x_range[0] + 0, env.y_range[0] + 0, env.z_range[0] + 0], dtype=np.float64)
                    hi = np.array([env.x_range[1] - 0, env.y_range[1] - 0, env.z_range[1] - 0], dtype=np.float64)
                    xs, xg = np.asarray(pr["x_start"], dtype=np.float64), np.asarray(pr["x_goal"], dtype=np.float64)
                    c_min, C, xc = pcu.ellipsoid_frame_3d(xs, xg)      # (point_cloud_mask_utils_3d.py:137-141: the same for every refresh)
                    c = pr["_cloud_const_3d"] = {"balls": balls.data_ptr(), "boxes": boxes.data_ptr(), "n_ball": int(balls.shape[0]),
                                                 "n_box": int(boxes.shape[0]), "lo": lo, "hi": hi, "diff": hi - lo, "c_min": float(c_min),
                                                 "C": np.ascontiguousarray(C, dtype=np.float64), "xc": np.asarray(xc, dtype=np.float64)}
                consts.append(c)
            jobs["balls"] = np.array([c["balls"] for c in consts], dtype=np.uint64)
            jobs["boxes"] = np.array([c["boxes"] for c in consts], dtype=np.uint64)
            jobs["n_ball"] = [c["n_ball"] for c in consts]
            jobs["n_box"] = [c["n_box"] for c in consts]
            jobs["mode"] = np.where(finite, 3, 2)
            A = jobs["a"]
            lo = np.stack([c["lo"] for c in consts])
            hi = np.stack([c["hi"] for c in consts])
            whole = ~finite
            if whole.any():      # whole box: lo, hi - lo
                A[whole, 0:3] = lo[whole]
                A[whole, 3:6] = np.stack([c["diff"] for c in consts])[whole]
            if finite.any():     # ellipsoid-restricted cloud (point_cloud_mask_utils_3d.py:132-200): C.L, x_center, range
                f = np.nonzero(finite)[0]
                ratio = np.array([c_best[idx[k]] / frames[idx[k]][0] for k in f], dtype=np.float64)
                CL = pcu.ellipsoid_transforms_3d(np.array([consts[k]["c_min"] for k in f]), np.stack([consts[k]["C"] for k in f]), ratio)
                A[f, 0:9] = CL.reshape(len(f), 9)
                A[f, 9:12] = np.stack([consts[k]["xc"] for k in f])
                A[f, 12:15] = lo[f]
                A[f, 15:18] = hi[f]
            jobs["clearance"] = 0.0
        return jobs, n_raw, n_words

    def refresh(self, due, problems, trees, streams, c_best, frames):
        """new clouds for the trees `due` (indices into the batch): c_best[i] = inf draws the whole-world cloud
        (nirrt_star_png_2d.py:132-145), otherwise the ellipse / ellipsoid-restricted one (:146-160).  Clouds are generated, down-
        sampled and handed to the trees ON the device (candidates from outputs of each problem's own generator, produced in its tree); the host
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
        on_dev = [j for j, i in enumerate(due)
                  if self.device_clouds and not (self.host_ellipsoid_3d and self.dim == 3 and c_best[i] < np.inf)]
        on_dev_set = set(on_dev)
        on_host = [j for j in range(nd) if j not in on_dev_set]
        clouds_dev = torch.zeros((nd, self.n_points, 3), dtype=torch.float64, device=dev)
        n_out = np.zeros(nd, dtype=np.int32)
        clouds = [None] * nd
        if on_dev:
            idx = [due[j] for j in on_dev]
            # what rng.random_sample / rng.uniform would consume, produced by the problems' own generators where they live
            n_raw, n_words = self.cloud_words()
            words = torch.empty((len(idx), n_words), dtype=torch.int32, device=dev)
            _hip.generator_words([trees[i] for i in idx], 0, n_words, device_ptr=words.data_ptr(), stride=n_words)
            for i in idx:
                streams[i].device_drew(py_too=False)
            jobs, n_raw, n_words = self._device_jobs(idx, problems, [words.data_ptr() + 4 * n_words * k for k in range(len(idx))],
                                                     c_best, frames, dev)
            sub = clouds_dev if len(on_dev) == nd else torch.zeros((len(on_dev), self.n_points, 3), dtype=torch.float64, device=dev)
            n_cand, n_o = pointops.guidance_clouds(jobs, n_raw, self.n_points, sub, self.device_id)
            for k, j in enumerate(on_dev):
                i = due[j]
                if self.dim == 2 and not (c_best[i] < np.inf) and n_cand[k] < self.n_points:
                    raise ValueError("farthest_point_down_sample: %d candidates for %d samples (problem %d)" % (n_cand[k], self.n_points, i))
                n_out[j] = n_o[k]
            if sub is not clouds_dev:
                clouds_dev[torch.as_tensor(on_dev, device=dev)] = sub
            host = sub.cpu().numpy()
            for k, j in enumerate(on_dev):
                clouds[j] = host[k, : n_out[j], : self.dim]
        t1 = time.perf_counter()
        if on_host:
            pts = self._host_clouds([due[j] for j in on_host], problems, trees, streams, c_best, frames)
            for k, j in enumerate(on_host):
                clouds[j] = pts[k][:, : self.dim]
                n_out[j] = len(pts[k])
                clouds_dev[j, : len(pts[k])] = torch.from_numpy(np.ascontiguousarray(pts[k]))
        t2 = time.perf_counter()
        xs_l = [np.asarray(problems[i]["x_start"], dtype=np.float64) for i in due]
        xg_l = [np.asarray(problems[i]["x_goal"], dtype=np.float64) for i in due]

        def fps_starts_for(group):   # group: positions inside `due`; every cloud's first-level start is drawn with ITS size
            st = np.stack([streams[due[j]].fps_starts((int(n_out[j]), 1024, 256, 64)) for j in group])      # (len(group), 4): every problem draws from its own generator
            return [torch.from_numpy(np.ascontiguousarray(st[:, k])) for k in range(4)]

        preds = [None] * nd
        pred_dev = None
        if self.connect and self.device_input and hasattr(self.wrapper, "classify_device"):
            # neural connect on the device: every round = one forward per cloud size + ONE launch of searches / heuristics
            from .png_wrapper import connect_rounds_device
            _, runs, pred_dev = connect_rounds_device(self.wrapper, clouds_dev, n_out, xs_l, xg_l, self.radius, self.max_trials,
                                                      fps_starts_for, self.dim, self.device_id)
            pred_host = pred_dev.cpu().numpy().astype(np.float32)
            for j in range(nd):
                preds[j] = pred_host[j, : n_out[j]]
            self.calls += int(getattr(connect_rounds_device, "last_forwards", 0) or (int(runs.max()) if nd else 0))
        elif self.connect:
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
            by_size = {}
            for j in range(nd):
                by_size.setdefault(int(n_out[j]), []).append(j)
            if self.size_log is not None:      # (diagnostics: the clouds of every refresh as {cloud size: clouds})
                self.size_log.append({n_: len(v) for n_, v in by_size.items()})
            if self.ragged and len(by_size) > 1 and max(by_size) <= 2048:
                # ONE forward over all due clouds, whatever their sizes (round 6).  A refresh holds one large group (clouds that were
                # down-sampled to n_points) and a few clouds with fewer candidates than that - each of a size of its own, each a
                # forward of its own before, and a forward over one cloud costs about what one over a hundred does (its
                # farthest-point samplings are ~1400 dependent steps, its ~100 launches come from the host one by one): at
                # pc_update_cost_ratio = 1.0 two thirds of the forwards were over ONE cloud.  Only the first set-abstraction level
                # looks at a cloud as a whole; its sampling and ball queries take the clouds' own sizes (PointNet2.forward).
                n_max = max(by_size)
                grp = list(range(nd))
                nv = torch.from_numpy(np.ascontiguousarray(n_out, dtype=np.int32)).to(dev)
                starts = fps_starts_for(grp)
                x = pointops.net_input(clouds_dev, grp, n_max, s3, g3, self.radius, n_each=nv)
                pred = self.wrapper.classify_device(x, fps_starts=starts, n_valid=nv)
                pred_dev[:, :n_max] = (pred != 0).to(torch.uint8)      # (the labels behind a cloud's own points are never read)
                self.calls += 1
                by_size = {}
            # otherwise one forward per cloud size (a forward's sampling and grouping depend on N).  With NIRRT_REFRESH_STREAMS > 1
            # the groups run on streams of their own, largest first - measured (round 6, config 4 at ratio 1.0: 12.2 s of forwards
            # against 11.1 s on one stream; NIRRT* 2D: 2.3 against 1.4 s) it is a loss: the forwards are bound by the host's launch
            # rate, which streams do not raise.  Default: one stream.
            order = sorted(by_size, key=lambda n_: (-len(by_size[n_]), n_))
            cur = torch.cuda.current_stream(dev)
            side = self._side_streams(dev, len(order)) if len(order) > 1 else []
            for k, size in enumerate(order):
                grp = by_size[size]
                starts = fps_starts_for(grp)      # (host: every problem's own generator, in batch order)
                st = side[k % len(side)] if side else cur
                if side:
                    st.wait_stream(cur)
                with torch.cuda.stream(st):
                    x = pointops.net_input(clouds_dev, grp, size, s3[grp], g3[grp], self.radius)
                    pred = self.wrapper.classify_device(x, fps_starts=starts)
                    pred_dev[torch.as_tensor(grp, device=dev), :size] = (pred != 0).to(torch.uint8)
                    del x, pred
                self.calls += 1
            for st in side:
                cur.wait_stream(st)
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
              window=65536, init_clouds=True, overlap_min=None):
    """`iters` loop bodies for every tree of the batch (fewer for trees that stop: first solution with stop_first, full
    tree), in persistent launches of at most `window` iterations.  Returns dict(traces = per-tree best cost after each
    iteration, iters_done, kernel_ms, launches, stats).  Every problem's generators move into its tree for the run (streams[i]
    -> trees[i]); `streams[i].rs` / `.py` afterwards show them advanced by exactly what the tree consumed."""
    B = len(trees)
    irrt = bool(flags & _hip.F_IRRT)
    png = guidance is not None
    need_py = dim == 2 and irrt
    hand_over(trees, streams)
    run_flags = flags | (_hip.F_STOP_FIRST if stop_first else 0) | (_hip.F_PNG if png else 0)
    remaining = np.full(B, int(iters), dtype=np.int64)
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
    prof = {"generators": 0.0, "wait_launch": 0.0, "book": 0.0, "refresh": 0.0}
    pace = np.zeros(B)      # device ticks per iteration of every tree in its last launch (0: not run yet)
    paced = png and os.environ.get("NIRRT_BATCH_PACE", "1") == "1"
    pace_ref = float(os.environ.get("NIRRT_BATCH_PACE_REF", "1.25"))      # trees slower than this x the median get shorter windows
    # a guided launch ends when this share of its trees waits for a refresh (0 = off).  The -C planners keep their launches whole:
    # a refresh of theirs is up to five rounds of forwards and searches, and fewer, larger refreshes beat idle slots (config 3,
    # round 6: 13.9 M it/s without, 12.2 - 13.4 with; NIRRT* 3D: 7.6 -> 8.0 with, at pc_update_cost_ratio = 1.0 1.6 -> 2.1)
    park_frac = float(os.environ.get("NIRRT_BATCH_PARK", "0" if (png and getattr(guidance, "connect", False)) else "0.25"))
    park_min = int(os.environ.get("NIRRT_BATCH_PARK_MIN", "16"))

    def launch(act):
        """the arguments of one persistent launch over the trees `act` (each draws from its own generators, in HBM); host
        generators that were handed out since the last launch (host-side cloud candidates) go back to their trees first"""
        rem = np.minimum(remaining[act], window)
        if paced and len(act) > 1:
            # A launch lasts as long as its slowest tree.  A tree whose iterations cost more than the typical one (measured: its
            # device time per iteration in its last launch) gets proportionally fewer of them this time, so that everybody is
            # done at about the same moment; it catches up over more launches while the others refresh their clouds.  Trees are
            # independent and launch boundaries never change a result.
            p_act = pace[act]
            known = p_act[p_act > 0]
            if len(known):
                ref = pace_ref * float(np.median(known))
                scale = np.where(p_act > ref, ref / np.maximum(p_act, 1e-12), 1.0)
                rem = np.minimum(rem, np.maximum(window // 8, (window * scale).astype(np.int64)))
        t_w = time.perf_counter()
        hand_over([trees[i] for i in act], [streams[i] for i in act], only_touched=True)
        for i in act:
            streams[i].device_drew(py_too=need_py)
        prof["generators"] += time.perf_counter() - t_w
        # A tree whose cloud is due idles in its slot until the launch ends.  With park_frac > 0 the launch itself ends as soon as
        # that share of its trees is waiting (nirrt_run_args.park_limit): the others come back with status E_PARK and their
        # remaining budget.  What decides is how often clouds fall due: at pc_update_cost_ratio = 1.0 (demo_planning_3d.py:21, a
        # refresh on EVERY improvement) most trees of a window-long launch were parked most of the time.
        park = max(park_min, int(park_frac * len(act))) if (png and park_frac > 0 and len(act) > 1) else 0
        return lambda: _hip.run_sampling([trees[i] for i in act], int(rem.max()), flags=run_flags, want_trace=want_trace, iters_each=rem,
                                         park_limit=park)

    def absorb(act, r, defer_refresh=False):
        """book a finished launch, refresh the clouds that are due; returns the trees of `act` that go on (defer_refresh: the
        trees that go on WITHOUT the due ones, and the due ones - their refresh is the caller's)"""
        nonlocal kernel_ms, launches
        kernel_ms += r["kernel_ms"]
        launches += 1
        t_b = time.perf_counter()
        due = []
        for j, i in enumerate(act):
            d = int(r["iters_done"][j])
            if want_trace and d:
                tr = r["cost_trace"][j, :d]
                traces[i].append(tr.copy())
                c_best[i] = tr[-1]
            remaining[i] -= d
            if d > 0:
                pace[i] = float(r["stats"][j, _hip.ST_BUSY]) / d
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
                if frames is not None and frames[i] is not None and not (c_best[i] >= float(frames[i][0]) * (1.0 - 1e-9)):
                    # (a solution cannot be shorter than the straight start-goal distance: the device handed back something else)
                    raise RuntimeError("tree %d stopped for a cloud refresh with best cost %r < |goal - start| = %r (launch %d, %d iterations in it, "
                                       "budget %d, counters %s)" % (i, c_best[i], float(frames[i][0]), launches, d, int(remaining[i]) + d,
                                                                    r["stats"][j].tolist()))
                due.append(i)
            elif st == _hip.E_STREAM:   # one draw rejected 2^22 generator outputs in a row
                failed[i] = "sampling cannot make progress (free space empty?)"
                finished[i] = True
            elif st == _hip.E_CAPACITY:
                failed[i] = "tree capacity exceeded"
                finished[i] = True
            elif st == _hip.E_ARG:
                failed[i] = "empty predicted cloud (np.random.randint(0, 0) in the reference)"
                finished[i] = True
            elif st == 0 or st == _hip.E_PARK:   # (E_PARK: the launch ended early for the others' cloud refreshes; the tree goes on)
                if remaining[i] <= 0 or (stop_first and d > 0 and np.isfinite(c_best[i])) or (stop_first and not want_trace):
                    finished[i] = True
            else:
                failed[i] = "device status %d" % st
                finished[i] = True
        t_r = time.perf_counter()
        prof["book"] += t_r - t_b
        if defer_refresh:      # (overlapped mode: the caller refreshes `due` while the others are already running again)
            due_set = set(due)
            return [i for i in act if not finished[i] and remaining[i] > 0 and i not in due_set], due
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
    # Overlapped refresh (round 6, one group, NIRRT_BATCH_OVERLAP=1; OFF by default): the trees whose cloud is due sit a launch out -
    # the others are launched again at once, and the due trees' clouds (candidates, down-sampling, PointNet++ forward: device work
    # on other streams + host bookkeeping) are made WHILE that launch runs; they join the launch after.  Nobody's results depend
    # on launch boundaries.  Measured at pc_update_cost_ratio = 1.0 (demo_planning_3d.py:21; 2048 3D trees, profiles/README_r06.md):
    # the refreshes are hidden (35 of 36 s), but the launches share the GPU with them and take 25.6 s instead of 14.6 s, and the
    # smaller due sets mean more refreshes, each as long as its slowest cloud: 45.4 s per step against 43.5 s one after the other.
    # AND IT IS NOT SAFE ON THIS STACK: like NIRRT_BATCH_GROUPS >= 2 (a persistent launch on one thread, refresh work on another) it
    # ended full-size runs with trees whose best cost was below the straight start-goal distance, or with a GPU memory fault
    # (gpurun_out of round 6: ov_full, c4r10_p25_g2) - small runs pass.  Both stay available for whoever hunts that down; nothing
    # in the library turns them on.
    overlap = png and n_groups == 1 and B > 1 and os.environ.get("NIRRT_BATCH_OVERLAP", "0") == "1"
    if overlap:
        active = list(range(B))
        if init_clouds:
            t_r = time.perf_counter()
            clouds.update(guidance.refresh(active, problems, trees, streams, c_best, frames))
            prof["refresh"] += time.perf_counter() - t_r
        pending = []
        prof["refresh_hidden"] = 0.0
        with ThreadPoolExecutor(max_workers=1) as pool:
            while active or pending:
                fut = pool.submit(launch(active)) if active else None
                refreshed = []
                if pending:
                    t_r = time.perf_counter()
                    clouds.update(guidance.refresh(pending, problems, trees, streams, c_best, frames))
                    dt = time.perf_counter() - t_r
                    prof["refresh"] += dt
                    if fut is not None:
                        prof["refresh_hidden"] += dt
                    refreshed, pending = pending, []
                cont = []
                if fut is not None:
                    t_w = time.perf_counter()
                    r = fut.result()
                    prof["wait_launch"] += time.perf_counter() - t_w
                    cont, pending = absorb(active, r, defer_refresh=True)
                active = cont + [i for i in refreshed if not finished[i] and remaining[i] > 0]
        return {"traces": [np.concatenate(t) if t else np.zeros(0) for t in traces], "iters_done": int(iters) - remaining,
                "kernel_ms": kernel_ms, "launches": launches, "stats": stats, "failed": failed, "clouds": clouds,
                "host_seconds": dict(prof, **(guidance.seconds if png else {}))}
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
