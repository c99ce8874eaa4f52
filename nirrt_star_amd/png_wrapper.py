"""PNGWrapper — point-cloud guidance adaptor around the PointNet++ net, reference interface:
wrapper/pointnet_pointnet2/pointnet2_wrapper.py:9-63, pointnet2_wrapper_connect_bfs.py:13-240 and the
wrapper_3d twins (3D: no z padding, 3D checkpoint path).

    PNGWrapper(num_classes=2, root_dir='.', device='cuda')
        .classify_path_points(pc f32 (N,2|3), start_mask f32 (N,), goal_mask f32 (N,)) -> (path_pred int (N,), path_score f32 (N,))
        .generate_connected_path_points(pc, x_start, x_goal, env_dict, neighbor_radius, max_trial_attempts,
                                        visualize=False, vis_folderpath="", token="") -> (bool, int, f32 (N,))
"""
from os.path import join

import os

import numpy as np
import torch

from .bfs_connect import bfs_point_cloud_visualization, get_boundary_mask, select_heuristic_boundary_point
from .pointcloud import get_point_cloud_mask_around_points
from .pointnet2 import get_model, pc_normalize


def checkpoint_path(root_dir, dim):
    tag = "pointnet2_%dd" % dim
    return join(root_dir, "results/model_training/%s/checkpoints/best_%s.pth" % (tag, tag))


def make_synthetic_checkpoint(path, seed=0, num_classes=2, calib_forwards=4, n_points=2048, dim=2, device="cuda"):
    """Seeded random-init weights + BatchNorm statistics calibrated by a few train-mode forwards on random
    clouds, saved in the reference's checkpoint format (train_pointnet_pointnet2.py:266-272).  There are
    no trained weights without network access; plain random init predicts an empty
    class and the planner cannot sample from it.  The calibration forwards run on `device` (the point operators only
    exist as HIP kernels; the CPU test-suite passes "cpu" after replacing the pointops functions from tests/conftest.py)."""
    import os
    torch.manual_seed(seed)
    model = get_model(num_classes).to(device)
    model.train()
    rs = np.random.RandomState(seed)
    with torch.no_grad():
        for _ in range(calib_forwards):
            pc = rs.uniform(0, 224 if dim == 2 else 50, size=(n_points, 3)).astype(np.float32)
            if dim == 2:
                pc[:, 2] = 0
            xyz = pc_normalize(pc)
            s = (np.linalg.norm(pc - pc[0], axis=1) < 10).astype(np.float32)
            gl = (np.linalg.norm(pc - pc[1], axis=1) < 10).astype(np.float32)
            feat = np.stack([s, gl, 1 - ((s + gl) > 0).astype(np.float32)], axis=-1)
            x = torch.from_numpy(np.concatenate([xyz, feat], axis=1).astype(np.float32)).permute(1, 0).unsqueeze(0)
            model(x.to(device))
    # un-trained weights put (almost) every point in one class and the planner cannot sample from an empty
    # prediction (the reference crashes in np.random.randint(0, 0), nirrt_star_png_2d.py:130): shift the class-1
    # bias so that about a third of the calibration points are labelled "path"
    model.eval()
    gaps = []
    with torch.no_grad():
        for _ in range(2):
            pc = rs.uniform(0, 224 if dim == 2 else 50, size=(n_points, 3)).astype(np.float32)
            if dim == 2:
                pc[:, 2] = 0
            s = (np.linalg.norm(pc - pc[0], axis=1) < 10).astype(np.float32)
            gl = (np.linalg.norm(pc - pc[1], axis=1) < 10).astype(np.float32)
            feat = np.stack([s, gl, 1 - ((s + gl) > 0).astype(np.float32)], axis=-1)
            x = torch.from_numpy(np.concatenate([pc_normalize(pc), feat], axis=1).astype(np.float32)).permute(1, 0).unsqueeze(0)
            logp, _ = model(x.to(device))
            gaps.append((logp[0, :, 1] - logp[0, :, 0]).cpu().numpy())
        model.conv2.bias[1] -= float(np.percentile(np.concatenate(gaps), 65))
    os.makedirs(os.path.dirname(path), exist_ok=True)
    torch.save({"epoch": 0, "class_avg_iou": 0.0, "model_state_dict": {k: v.cpu() for k, v in model.state_dict().items()},
                "optimizer_state_dict": {}}, path)
    return path


class PNGWrapper:
    dim = 2

    def __init__(self, num_classes=2, root_dir='.', device='cuda', model_filepath=None):
        self.device = device
        self.model = get_model(num_classes).to(device)
        if model_filepath is None:
            model_filepath = checkpoint_path(root_dir, self.dim)
        # weights_only=False: reference checkpoints carry a numpy scalar ('class_avg_iou'), SURVEY Appendix B
        checkpoint = torch.load(model_filepath, map_location=torch.device(device), weights_only=False)
        self.model.load_state_dict(checkpoint['model_state_dict'])
        self.model = self.model.eval().fold()
        # one-cloud forwards are launch-bound (~150 small kernels): replayed from a HIP graph per cloud size
        self.use_graph = torch.device(device).type == "cuda"
        self._graphs = {}
        self.MAX_GRAPHS = 2   # cloud sizes that keep a captured B = 1 forward (see _graph_forward)
        print("PointNet++ wrapper%s is initialized." % ("" if self.dim == 2 else " 3d"))

    @staticmethod
    def network_input(pc, start_mask, goal_mask):
        """one cloud -> the (6, N) float32 channel block the net consumes: xyz (z = 0 for planar clouds) shifted to the
        centroid and scaled by the largest norm, then the start / goal indicator channels and the "neither" channel
        (reference: pointnet2_wrapper.py:43-58; arithmetic kept in float32 numpy like there)"""
        cloud = np.asarray(pc, dtype=np.float32)
        n = len(cloud)
        block = np.zeros((6, n), dtype=np.float32)
        block[: cloud.shape[1]] = pc_normalize(cloud).T if cloud.shape[1] == 3 else \
            pc_normalize(np.concatenate([cloud, np.zeros((n, 1), dtype=np.float32)], axis=1)).T[:2]
        s = np.asarray(start_mask, dtype=np.float32)
        g = np.asarray(goal_mask, dtype=np.float32)
        block[3], block[4] = s, g
        block[5] = ((s + g) == 0).astype(np.float32)
        return block

    def classify_batch(self, clouds, start_masks, goal_masks, fps_starts=None):
        """B clouds of equal size in ONE forward: lists / arrays of (N, 2|3) clouds and (N,) masks ->
        (path_pred int64 (B, N), path_score float32 (B, N)).  The input block is assembled once on the host, crosses to
        the device once, and only the two result rows per cloud come back."""
        x = np.stack([self.network_input(c, s, g) for c, s, g in zip(clouds, start_masks, goal_masks)], axis=0)
        with torch.no_grad():
            if x.shape[0] == 1 and getattr(self, "use_graph", False):
                logp = self._graph_forward(x, fps_starts)
            else:
                logp, _ = self.model(torch.from_numpy(x).to(self.device), fps_starts=fps_starts)   # (B, N, classes)
            pred = logp.argmax(dim=2)
            score = torch.softmax(logp, dim=2)[:, :, 1]
            return pred.cpu().numpy(), score.cpu().numpy()

    def classify_device(self, x, fps_starts=None, n_valid=None):
        """input blocks already on the device (pointops.net_input): x f32 (B, 6, N) -> path_pred int64 (B, N), left on the device.
        n_valid (B,) int32 on the device: ragged batch (clouds of different sizes, PointNet2.forward) - row b's first n_valid[b]
        labels are the cloud's"""
        with torch.no_grad():
            if n_valid is not None:
                logp, _ = self.model(x, fps_starts=fps_starts, n_valid=n_valid)
            else:
                logp, _ = self.model(x, fps_starts=fps_starts)
            return logp.argmax(dim=2)

    def _graph_forward(self, x, fps_starts):
        """B = 1: the forward of a cloud of this size captured once into a HIP graph (static input / start-index buffers) and
        replayed.  The FPS start indices are drawn here, on the CPU generator and in the order the model itself would draw them
        (pointnet2_utils.py:77: one torch.randint per set-abstraction level), and copied into the graph's buffers."""
        n = x.shape[2]
        if fps_starts is None:
            fps_starts = [torch.randint(0, m, (1,), dtype=torch.long) for m in (n, 1024, 256, 64)]
        g = self._graphs.get(n)
        if g is None and len(self._graphs) >= self.MAX_GRAPHS:
            # ellipse-restricted clouds come in arbitrary sizes: only the first few sizes seen (the full 2048-point cloud among
            # them) get a captured graph with its pinned buffers; the others take plain launches
            logp, _ = self.model(torch.from_numpy(x).to(self.device), fps_starts=fps_starts)
            return logp
        if g is None:
            g = {"x": torch.zeros(1, 6, n, device=self.device), "st": [torch.zeros(1, dtype=torch.long, device=self.device) for _ in range(4)]}
            try:
                side = torch.cuda.Stream(device=self.device)
                side.wait_stream(torch.cuda.current_stream(self.device))
                with torch.cuda.stream(side):
                    for _ in range(2):
                        self.model(g["x"], fps_starts=g["st"])
                torch.cuda.current_stream(self.device).wait_stream(side)
                g["graph"] = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g["graph"]):
                    g["out"], _ = self.model(g["x"], fps_starts=g["st"])
            except Exception as e:   # capture not possible in this environment: plain launches from now on
                print("PointNet++ wrapper: HIP graph capture failed (%s); using plain launches." % e)
                self.use_graph = False
                logp, _ = self.model(torch.from_numpy(x).to(self.device), fps_starts=fps_starts)
                return logp
            self._graphs[n] = g
        g["x"].copy_(torch.from_numpy(x))
        for dst, src in zip(g["st"], fps_starts):
            dst.copy_(src)
        g["graph"].replay()
        return g["out"]

    def classify_path_points(self, pc, start_mask, goal_mask):
        """reference signature (pointnet2_wrapper.py:43-63): one cloud -> (path_pred (N,), path_score (N,))"""
        pred, score = self.classify_batch([pc], [start_mask], [goal_mask])
        return pred[0], score[0]

    def generate_connected_path_points(self, pc, x_start, x_goal, env_dict, neighbor_radius, max_trial_attempts,
                                       visualize=False, vis_folderpath="", token=""):
        """<= max_trial_attempts rounds of classify + BFS connectivity, re-seeding the start / goal masks at the
        heuristic boundary point of the visited set, alternating start->goal and goal->start
        (pointnet2_wrapper_connect_bfs.py:76-240)."""
        state = ConnectState(pc, x_start, x_goal, neighbor_radius)
        for _ in range(max_trial_attempts):
            path_pred, _ = self.classify_path_points(pc, state.start_mask, state.goal_mask)
            if state.absorb(path_pred):
                break
        return state.has_path, state.trials, state.path_pred_mask

    def generate_connected_path_points_batch(self, clouds, starts, goals, neighbor_radius, max_trial_attempts, fps_starts_for=None):
        """neural connect for several clouds at once: round r classifies, in ONE forward per cloud size, every cloud that
        is not connected yet.  Returns a list of (connection_success, num_png_runs, path_pred_mask) like the single call.
        fps_starts_for(indices) -> the 4 FPS start tensors of a forward over those clouds (default: torch's CPU generator)."""
        states = [ConnectState(pc, xs, xg, neighbor_radius) for pc, xs, xg in zip(clouds, starts, goals)]
        for _ in range(max_trial_attempts):
            open_ = [i for i, st in enumerate(states) if not st.has_path]
            if not open_:
                break
            for size in sorted(set(len(clouds[i]) for i in open_)):
                grp = [i for i in open_ if len(clouds[i]) == size]
                pred, _ = self.classify_batch([clouds[i] for i in grp], [states[i].start_mask for i in grp],
                                              [states[i].goal_mask for i in grp],
                                              fps_starts=fps_starts_for(grp) if fps_starts_for else None)
                for j, i in enumerate(grp):
                    states[i].absorb(pred[j])
        return [(st.has_path, st.trials, st.path_pred_mask) for st in states]


def connect_rounds_device(wrapper, clouds_dev, n_out, starts, goals, radius, max_trial_attempts, fps_starts_for=None, dim=2,
                          device_id=0):
    """generate_connected_path_points for a batch of clouds that are resident on the device (pointnet2_wrapper_connect_bfs.py:
    76-240): round r classifies, in ONE forward per cloud size, every cloud that is not connected yet (input blocks assembled on
    the device from the masks the round before left there), then ONE launch runs the breadth-first searches, boundary masks and
    seed heuristics of all of them (nirrt_connect_round) and one more writes the next masks.  The host sees a few integers per
    cloud and round.  clouds_dev f64 (nd, P, 3), n_out (nd,) points per cloud, starts / goals per cloud.
    Returns (has_path bool (nd,), num_png_runs int (nd,), path_pred_mask uint8 tensor (nd, P) on the device)."""
    from . import pointops
    from .bfs_connect import select_heuristic_boundary_point
    nd, P = int(clouds_dev.shape[0]), int(clouds_dev.shape[1])
    dev = clouds_dev.device
    n_out = np.asarray(n_out, dtype=np.int64)
    path = torch.zeros((nd, P), dtype=torch.uint8, device=dev)
    smask, gmask = torch.zeros_like(path), torch.zeros_like(path)
    pred = torch.zeros_like(path)
    boundary = torch.zeros((nd, 2, P), dtype=torch.uint8, device=dev)
    jobs = []
    for j in range(nd):
        jb = pointops.ConnectJob()
        jb.cloud, jb.pred, jb.path_mask = clouds_dev[j].data_ptr(), pred[j].data_ptr(), path[j].data_ptr()
        jb.start_mask, jb.goal_mask, jb.boundary = smask[j].data_ptr(), gmask[j].data_ptr(), boundary[j].data_ptr()
        jb.n, jb.dim = int(n_out[j]), int(dim)
        for k in range(dim):
            jb.start[k], jb.goal[k] = float(starts[j][k]), float(goals[j][k])
        jobs.append(jb)
    pointops.connect_masks(jobs, radius, np.full((nd, 2), -2, dtype=np.int32), device_id)      # masks around the start / goal states
    has = np.zeros(nd, dtype=bool)
    trials = np.zeros(nd, dtype=np.int64)
    wrapper_calls = 0
    ragged = os.environ.get("NIRRT_RAGGED_FORWARD", "1") == "1"
    for _ in range(int(max_trial_attempts)):
        open_ = [j for j in range(nd) if not has[j]]
        if not open_:
            break
        sizes = sorted(set(int(n_out[j]) for j in open_))
        if ragged and len(sizes) > 1 and fps_starts_for is not None and sizes[-1] <= 2048:
            # every open cloud of the round in ONE forward, whatever their sizes (PointNet2.forward(n_valid=...)); the labels
            # behind a cloud's own points are never read (the searches take jobs[j].n points)
            nv = torch.from_numpy(np.ascontiguousarray(n_out[open_], dtype=np.int32)).to(dev)
            x = pointops.net_input_masks(clouds_dev, open_, sizes[-1], smask, gmask, n_each=nv)
            p = wrapper.classify_device(x, fps_starts=fps_starts_for(open_), n_valid=nv)
            pred[torch.as_tensor(open_, device=dev), :sizes[-1]] = (p != 0).to(torch.uint8)
            wrapper_calls += 1
            sizes = []
        for size in sizes:
            grp = [j for j in open_ if n_out[j] == size]
            x = pointops.net_input_masks(clouds_dev, grp, size, smask, gmask)
            p = wrapper.classify_device(x, fps_starts=fps_starts_for(grp) if fps_starts_for else None)
            pred[torch.as_tensor(grp, device=dev), :size] = (p != 0).to(torch.uint8)
            wrapper_calls += 1
        hp, seeds, ties = pointops.connect_round([jobs[j] for j in open_], radius, device_id)
        trials[open_] += 1
        for k, j in enumerate(open_):
            if hp[k]:
                has[j] = True
                continue
            for d in (0, 1):
                if ties[k, d]:     # equal keys among the boundary points: numpy's own (unstable) argsort decides, like in the reference
                    pc32 = clouds_dev[j, : n_out[j], :dim].cpu().numpy().astype(np.float32)
                    nj = int(n_out[j])
                    bm = boundary[j].reshape(-1)[d * nj:(d + 1) * nj].cpu().numpy().astype(np.float32)   # (2, n) packed
                    a = np.asarray(starts[j] if d == 0 else goals[j]).astype(np.float32)
                    b = np.asarray(goals[j] if d == 0 else starts[j]).astype(np.float32)
                    bi, _, _ = select_heuristic_boundary_point(pc32, bm, a, b)
                    seeds[k, d] = -1 if bi is None else int(bi)
        still = [k for k, j in enumerate(open_) if not hp[k]]
        if still:
            pointops.connect_masks([jobs[open_[k]] for k in still], radius, seeds[still], device_id)
    connect_rounds_device.last_forwards = wrapper_calls      # (forwards of this call: the caller's statistics)
    return has, trials, path


class ConnectState:
    """one cloud's progress through the neural-connect rounds: the union of the predictions so far, the start / goal masks
    the NEXT classification uses, and whether start and goal are connected through predicted points yet"""

    def __init__(self, pc, x_start, x_goal, neighbor_radius):
        self.pc = pc
        self.radius = neighbor_radius
        self.xs = np.asarray(x_start).astype(np.float32)
        self.xg = np.asarray(x_goal).astype(np.float32)
        self.path_pred_mask = np.zeros(len(pc)).astype(np.float32)
        self.start_mask = get_point_cloud_mask_around_points(pc, np.asarray(x_start)[np.newaxis].astype(np.float32), neighbor_radius)
        self.goal_mask = get_point_cloud_mask_around_points(pc, np.asarray(x_goal)[np.newaxis].astype(np.float32), neighbor_radius)
        self.has_path = False
        self.trials = 0

    def absorb(self, path_pred):
        """take one classification result; True once start and goal are connected"""
        self.trials += 1
        self.path_pred_mask = ((self.path_pred_mask + path_pred) > 0).astype(np.float32)
        seeds = []
        for a, b in ((self.xs, self.xg), (self.xg, self.xs)):
            has_path, _, visited_mask = bfs_point_cloud_visualization(self.pc, self.path_pred_mask, a, b, self.radius)
            boundary_mask = get_boundary_mask(self.pc, visited_mask, 1 - self.path_pred_mask, self.radius)
            _, boundary_point, _ = select_heuristic_boundary_point(self.pc, boundary_mask, a, b)
            if has_path:
                self.has_path = True
                return True
            seeds.append(boundary_point)
        nxt = []
        for cur, bp in zip((self.start_mask, self.goal_mask), seeds):
            nxt.append(cur if bp is None else get_point_cloud_mask_around_points(self.pc, bp, self.radius))
        self.start_mask, self.goal_mask = nxt
        return False


class PNGWrapper3D(PNGWrapper):
    dim = 3
