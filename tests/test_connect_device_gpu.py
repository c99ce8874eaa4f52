"""GPU: neural connect on the device (nirrt_connect_round / nirrt_connect_masks, png_wrapper.connect_rounds_device) against the
fixtures the REFERENCE produced (tests/golden/make_golden.py: connect_ref, connect_ref3d - wrapper/utils/bfs_connect_heuristic.py
and the multi-round loop of pointnet2_wrapper_connect_bfs.py:76-240 in 2D and 3D): reachability verdicts, boundary masks, seed
points, the masks fed to every round, the number of rounds and the final union mask must be identical."""
import numpy as np
import pytest

from conftest import load_golden, seed_grow_classifier

pytestmark = pytest.mark.gpu

FIX = [("connect_ref", 2, 10.0, 28.0), ("connect_ref3d", 3, None, 8.0)]


def _jobs(pc, masks, xs, xg, dim):
    """device buffers + ConnectJob per (cloud, prediction mask)"""
    import torch
    from nirrt_star_amd import pointops
    nd, n = len(masks), len(pc)
    cloud = np.zeros((nd, n, 3))
    cloud[:, :, :dim] = pc.astype(np.float64)
    t = {"cloud": torch.from_numpy(cloud).cuda(), "pred": torch.from_numpy(np.stack(masks).astype(np.uint8)).cuda(),
         "path": torch.zeros((nd, n), dtype=torch.uint8, device="cuda"), "sm": torch.zeros((nd, n), dtype=torch.uint8, device="cuda"),
         "gm": torch.zeros((nd, n), dtype=torch.uint8, device="cuda"), "bd": torch.zeros((nd, 2, n), dtype=torch.uint8, device="cuda")}
    jobs = []
    for j in range(nd):
        jb = pointops.ConnectJob()
        jb.cloud, jb.pred, jb.path_mask = t["cloud"][j].data_ptr(), t["pred"][j].data_ptr(), t["path"][j].data_ptr()
        jb.start_mask, jb.goal_mask, jb.boundary = t["sm"][j].data_ptr(), t["gm"][j].data_ptr(), t["bd"][j].data_ptr()
        jb.n, jb.dim = n, dim
        for k in range(dim):
            jb.start[k], jb.goal[k] = float(xs[k]), float(xg[k])
        jobs.append(jb)
    return t, jobs


@pytest.mark.parametrize("name,dim,radius,grow", FIX)
def test_search_boundary_and_seed_of_one_round(name, dim, radius, grow):
    from nirrt_star_amd import pointops
    g = load_golden(name)
    rad = float(g["radius"]) if radius is None else radius
    pc = g["pc"]
    t, jobs = _jobs(pc, [g["corridor_sg_mask"], g["blobs_sg_mask"]], g["xs"], g["xg"], dim)
    has, seed, tie = pointops.connect_round(jobs, rad)
    assert has[0] == 1 and bool(g["corridor_sg_has"]) and bool(g["corridor_gs_has"])      # connected: the rounds end
    assert has[1] == 0 and not bool(g["blobs_sg_has"])
    bd = t["bd"].cpu().numpy()
    for d, tag in enumerate(("sg", "gs")):
        assert np.array_equal(bd[1, d].astype(np.float32), g["blobs_%s_boundary" % tag]), tag
        if not tie[1, d]:
            assert seed[1, d] == int(g["blobs_%s_bidx" % tag])
    assert np.array_equal(t["path"].cpu().numpy()[1].astype(np.float32), g["blobs_sg_mask"])      # the union so far = this prediction
    # masks of the next classification: around the seeds ...
    pointops.connect_masks(jobs[1:], rad, seed[1:])
    from nirrt_star_amd.pointcloud import get_point_cloud_mask_around_points as around
    for d, (tag, key) in enumerate((("sg", "sm"), ("gs", "gm"))):
        bi = int(g["blobs_%s_bidx" % tag])
        assert np.array_equal(t[key].cpu().numpy()[1] != 0, around(pc, pc[bi], rad))
    # ... and, before the first round, around the start / goal states; -1 keeps a mask
    pointops.connect_masks(jobs[:1], rad, np.array([[-2, -2]], dtype=np.int32))
    assert np.array_equal(t["sm"].cpu().numpy()[0] != 0, around(pc, g["xs"][np.newaxis].astype(np.float32), rad))
    assert np.array_equal(t["gm"].cpu().numpy()[0] != 0, around(pc, g["xg"][np.newaxis].astype(np.float32), rad))
    keep = t["sm"].cpu().numpy()[0].copy()
    pointops.connect_masks(jobs[:1], rad, np.array([[-1, 5]], dtype=np.int32))
    assert np.array_equal(t["sm"].cpu().numpy()[0], keep)


class _GrowOnDevice:
    """classify_device of the seed-growing classifier: the masks are read from the input block the device assembled (channels
    3 / 4), the labels are numpy's (exact), the clouds are the fixture's"""

    def __init__(self, clouds32, grow):
        self.clouds, self.classify, self.grp, self.seen = clouds32, seed_grow_classifier(grow), None, []

    def fps_starts_for(self, grp):
        self.grp = list(grp)
        return None

    def classify_device(self, x, fps_starts=None):
        import torch
        xh = x.cpu().numpy()
        out = []
        for r, j in enumerate(self.grp):
            sm, gm = xh[r, 3], xh[r, 4]
            assert np.array_equal(xh[r, 5], ((sm + gm) == 0).astype(np.float32))
            self.seen.append((j, sm.copy(), gm.copy()))
            out.append(self.classify(self.clouds[j], sm, gm)[0])
        return torch.from_numpy(np.stack(out)).to(x.device)


@pytest.mark.parametrize("name,dim,radius,grow", FIX)
def test_rounds_of_a_batch_match_the_reference_loops(name, dim, radius, grow):
    """two clouds in one batch (the fixture's and the same points in reverse order), 5 and 2 rounds allowed"""
    import torch
    from nirrt_star_amd import png_wrapper
    from conftest import stub_connect_wrapper
    g = load_golden(name)
    rad = float(g["radius"]) if radius is None else radius
    pc = g["pc"]
    pcs = [pc, pc[::-1].copy()]
    cloud = np.zeros((2, len(pc), 3))
    for j in range(2):
        cloud[j, :, :dim] = pcs[j].astype(np.float64)
    clouds_dev = torch.from_numpy(cloud).cuda()
    for tag, trials in (("loop5", 5), ("loop2", 2)):
        w = _GrowOnDevice(pcs, grow)
        has, runs, path = png_wrapper.connect_rounds_device(w, clouds_dev, [len(pc)] * 2, [g["xs"]] * 2, [g["xg"]] * 2, rad, trials,
                                                            w.fps_starts_for, dim)
        path = path.cpu().numpy().astype(np.float32)
        assert bool(has[0]) == bool(g[tag + "_ok"]) and runs[0] == int(g[tag + "_runs"])
        assert np.array_equal(path[0], g[tag + "_mask"])
        mine = [(sm, gm) for j, sm, gm in w.seen if j == 0]
        assert np.array_equal(np.stack([m[0] for m in mine]), g[tag + "_start_masks"])
        assert np.array_equal(np.stack([m[1] for m in mine]), g[tag + "_goal_masks"])
        # the reversed cloud against the host loop of the package on the same points
        one = stub_connect_wrapper(dim, grow).generate_connected_path_points(pcs[1], g["xs"], g["xg"], g["env"], rad, trials)
        assert bool(has[1]) == bool(one[0]) and runs[1] == one[1] and np.array_equal(path[1], one[2])


def test_equal_keys_are_reported():
    """mirror-symmetric boundary points have equal g + h: the device reports the tie (the caller lets numpy decide)"""
    from nirrt_star_amd import pointops
    xs, xg = np.array([0.0, 0.0]), np.array([100.0, 0.0])
    pts = [(4.0, 0.0)] + [(8.0, 3.0), (8.0, -3.0)] + [(14.0, 6.0), (14.0, -6.0)] + [(60.0 + i, 40.0) for i in range(5)]
    pc = np.array(pts, dtype=np.float32)
    pred = np.zeros(len(pc), dtype=np.float32)
    pred[:3] = 1
    t, jobs = _jobs(pc, [pred], xs, xg, 2)
    has, seed, tie = pointops.connect_round(jobs, 8.0)
    assert has[0] == 0 and tie[0, 0] == 1
    bd = t["bd"].cpu().numpy()[0, 0]
    assert bd[1] == 1 and bd[2] == 1          # the mirror pair is on the boundary (each has an unpredicted point within 8)


@pytest.mark.parametrize("dim", [2, 3])
def test_batched_nirrt_c_device_connect_equals_host_connect(dim, monkeypatch):
    """eval_sharded.plan_batch --planner nirrt_star_c with the real wrapper: neural connect on the device against the same run
    with the host loop (NIRRT_HOST_INPUT=1: numpy masks, host breadth-first searches): identical cost traces"""
    from types import SimpleNamespace as NS
    from conftest import synthetic_checkpoint_root
    from nirrt_star_amd import eval_sharded as es, png_wrapper, worlds
    mk = (lambda i: worlds.problem_2d(worlds.random_world_2d(90 + i, "b30"), 0)) if dim == 2 else (lambda i: worlds.problem_3d(worlds.random_world_3d(90 + i)))
    args = NS(problem="random_2d" if dim == 2 else "random_3d", planner="nirrt_star_c", iter_max=2500, iter_after_initial=300, step_len=10,
              clearance=3 if dim == 2 else 2, pc_n_points=2048, pc_over_sample_scale=5, pc_sample_rate=0.5, pc_update_cost_ratio=0.9,
              connect_max_trial_attempts=5, root_dir=".", segment=1000)
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("NIRRT_HOST_INPUT", mode)
        if dim == 3:
            np.random.seed(6)
        probs = [mk(i) for i in range(4)]
        w = (png_wrapper.PNGWrapper if dim == 2 else png_wrapper.PNGWrapper3D)(root_dir=synthetic_checkpoint_root(dim), device="cuda")
        w.use_graph = False      # (a captured B = 1 forward may pick other GEMM algorithms than plain launches: labels of near-tie points)
        out[mode] = es.plan_batch(probs, list(range(4)), args, 0, wrapper=w)
    for a, b in zip(out["0"][1], out["1"][1]):
        assert np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)
    assert [r[1] for r in out["0"][0]] == [r[1] for r in out["1"][0]]


@pytest.mark.parametrize("name", ["run_nirrtc2d_bfs_1500", "run_nirrtc3d_bfs_1500"])
def test_nirrt_c_planner_with_a_running_connect_loop_equals_the_reference(name):
    """NIRRT*-PNG(C) 2D / 3D planner classes with the package's generate_connected_path_points (the neural-connect loop really
    runs: seed-growing classifier, several rounds per cloud) against the reference planner + the reference's own connect
    wrapper class run the same way (path_planning_classes{,_3d}/nirrt_star_png_c_{2d,3d}.py:10-121)"""
    import random
    from conftest import stub_connect_wrapper
    from nirrt_star_amd import planners
    g = load_golden(name)
    dim = int(g["dim"])
    calls = []
    w = stub_connect_wrapper(dim, float(g["grow"]), calls)
    common = [tuple(g["x_start"]), tuple(g["x_goal"]), 10, float(g["search_radius"]), int(g["iter_max"]), g["env"], w]
    if dim == 2:
        common.append(g["binary_mask"].astype(np.float64))
    cls = planners.NIRRTStarPNGC2D if dim == 2 else planners.NIRRTStarPNGC3D
    p = cls(*common, int(g["clearance"]), 2048, 5, 0.5, 0.9, 5)
    np.random.seed(int(g["seed"]))
    random.seed(int(g["seed"]))
    p.planning()
    n = p.num_vertices
    assert len(calls) == int(g["n_classifications"])
    assert n == int(g["n"]) and np.array_equal(p.vertex_parents[:n], g["parents"])
    assert np.array_equal(p.vertices[:n], g["vertices"])
    assert np.array_equal(np.array(p.path_solutions), g["path_solutions"])
    assert abs(p.get_path_len(p.path) - float(g["path_len"])) <= 1e-5
