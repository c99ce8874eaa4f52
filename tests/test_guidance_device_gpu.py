"""Guidance clouds generated on the device (nirrt_guidance_clouds) vs the host path that reference-generated fixtures pin
(tests/test_guidance_fixtures.py): same generator consumption, same candidate filtering, same down-sampling - the clouds must be
BIT-equal, for the whole-image and the ellipse-restricted 2D clouds and the whole-box 3D cloud
(point_cloud_mask_utils.py:35-73, 104-174; point_cloud_mask_utils_3d.py:83-113)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _device_side(pr, dim, seed):
    """a tree that owns the problem's generators (seeded like the host-side twin) + the host twin"""
    from nirrt_star_amd import _hip, batch
    t = _hip.HipTree(dim, 100, pr["x_start"], pr["x_goal"], 10, pr["search_radius"], 3 if dim == 2 else 2, pr["env"])
    sd, sh = batch.ProblemStreams(seed), batch.ProblemStreams(seed)
    batch.hand_over([t], [sd])
    return t, sd, sh


def _device_cloud(g, t, sd, pr, cbest, frame):
    import torch
    from nirrt_star_amd import _hip, pointops
    dev = torch.device("cuda", 0)
    n_raw, n_words = g.cloud_words()
    words = torch.empty((1, n_words), dtype=torch.int32, device=dev)
    _hip.generator_words([t], 0, n_words, device_ptr=words.data_ptr(), stride=n_words)
    sd.device_drew(py_too=False)
    jobs, n_raw, nw = g._device_jobs([0], [pr], [words.data_ptr()], [cbest], [frame], dev)
    out = torch.zeros((1, g.n_points, 3), dtype=torch.float64, device="cuda")
    n_cand, n_out = pointops.guidance_clouds(jobs, n_raw, g.n_points, out, 0)
    return out.cpu().numpy()[0, : n_out[0]]


@pytest.mark.parametrize("world,pair", [(0, 0), (3, 1), (7, 2)])
def test_device_clouds_2d_equal_host_clouds(world, pair):
    from nirrt_star_amd import _hip, batch, pointcloud as pcu, sampling, worlds
    pr = worlds.problem_2d(worlds.random_world_2d(world, "b30"), pair)
    g = batch.Guidance(None, 2, 10)
    frame = sampling.informed_frame(pr["x_start"], pr["x_goal"])
    xs, xg = np.asarray(pr["x_start"], dtype=np.float64), np.asarray(pr["x_goal"], dtype=np.float64)
    for cbest in (np.inf, 1.6 * frame[0], 1.05 * frame[0]):
        t, sd, sh = _device_side(pr, 2, 1000 + world)
        for _ in range(2):    # two clouds in a row: the generator position carries over
            dev_cloud = _device_cloud(g, t, sd, pr, cbest, frame)[:, :2]
            if cbest < np.inf:
                host_cloud = pcu.ellipsoid_point_cloud_sampling(xs, xg, cbest / frame[0], pr["binary_mask"], g.n_points, g.n_points * g.scale, sh.rs)
            else:
                host_cloud = pcu.generate_rectangle_point_cloud(pr["binary_mask"], g.n_points, g.scale, sh.rs)
            assert dev_cloud.shape == host_cloud.shape
            assert np.array_equal(dev_cloud, host_cloud)
            # the tree's generator sits where the host twin's does: same get_state()
            k_d, p_d = _hip.np_state(sd.rs)
            k_h, p_h = _hip.np_state(sh.rs)
            assert p_d == p_h and np.array_equal(k_d, k_h)
        t.close()


def test_device_cloud_3d_box_equals_host_cloud():
    from nirrt_star_amd import _hip, batch, pointcloud as pcu, sampling, worlds
    np.random.seed(4)
    pr = worlds.problem_3d(worlds.random_world_3d(4))
    g = batch.Guidance(None, 3, 10)
    frame = sampling.informed_frame(pr["x_start"], pr["x_goal"])
    t, sd, sh = _device_side(pr, 3, 77)
    dev_cloud = _device_cloud(g, t, sd, pr, np.inf, frame)
    host_cloud = pcu.generate_rectangle_point_cloud_3d(pr["env"], g.n_points, g.scale, clearance=0, rng=sh.rs)
    assert np.array_equal(dev_cloud, host_cloud)
    k_d, p_d = _hip.np_state(sd.rs)
    k_h, p_h = _hip.np_state(sh.rs)
    assert p_d == p_h and np.array_equal(k_d, k_h)
    t.close()


@pytest.mark.parametrize("world,ratio", [(4, 1.8), (9, 1.2), (15, 1.02)])
def test_device_cloud_3d_ellipsoid_equals_host_cloud_within_libm(world, ratio):
    """ellipsoid_point_cloud_sampling_3d (point_cloud_mask_utils_3d.py:132-200): the candidates go through np.sin / np.cos - for
    float64 these are libm's (numpy 2.2), which the device restates (csrc/glibc235_libm.inc): same candidates kept, same points
    selected by the down-sampling, coordinates BIT-EQUAL, generator at the same state.  (Rounds 3-4: the device's own sin / cos,
    <= 1e-9.)"""
    from nirrt_star_amd import _hip, batch, pointcloud as pcu, sampling, worlds
    np.random.seed(world)
    pr = worlds.problem_3d(worlds.random_world_3d(world))
    g = batch.Guidance(None, 3, 10)
    frame = sampling.informed_frame(pr["x_start"], pr["x_goal"])
    xs, xg = np.asarray(pr["x_start"], dtype=np.float64), np.asarray(pr["x_goal"], dtype=np.float64)
    t, sd, sh = _device_side(pr, 3, 300 + world)
    for _ in range(2):
        dev_cloud = _device_cloud(g, t, sd, pr, ratio * frame[0], frame)
        host_cloud = pcu.ellipsoid_point_cloud_sampling_3d(xs, xg, ratio, pr["env"], g.n_points, g.n_points * g.scale, clearance=0, rng=sh.rs)
        assert dev_cloud.shape == host_cloud.shape and len(dev_cloud) > 100
        assert np.array_equal(dev_cloud, host_cloud)      # (round 5: np.sin / np.cos = libm's, restated on the device: bit-equal)
        k_d, p_d = _hip.np_state(sd.rs)
        k_h, p_h = _hip.np_state(sh.rs)
        assert p_d == p_h and np.array_equal(k_d, k_h)
    t.close()


def test_generators_move_between_host_and_tree():
    """ProblemStreams: a draw on the host between two device draws is honoured (the state goes back to the tree before the
    next launch), and an untouched host object is simply behind until somebody looks at it"""
    import random
    from nirrt_star_amd import _hip, batch, worlds
    pr = worlds.problem_2d(worlds.random_world_2d(2, "b30"), 0)
    t = _hip.HipTree(2, 100, pr["x_start"], pr["x_goal"], 10, pr["search_radius"], 3, pr["env"])
    s = batch.ProblemStreams(11)
    ref_np, ref_py = np.random.RandomState(11), random.Random(11)
    batch.hand_over([t], [s])
    w = _hip.generator_words([t], 0, 1000)[0]
    s.device_drew()
    assert np.array_equal(w, ref_np.randint(0, 1 << 32, size=1000, dtype=np.uint32))
    assert s.rs.random_sample() == ref_np.random_sample()          # fetched from the tree, then drawn on the host
    assert s.touched()
    batch.hand_over([t], [s], only_touched=True)                   # ... and back
    assert not s.touched()
    w = _hip.generator_words([t], 0, 700)[0]
    s.device_drew()
    assert np.array_equal(w, ref_np.randint(0, 1 << 32, size=700, dtype=np.uint32))
    w = _hip.generator_words([t], 1, 10)[0]
    v = ref_py.getrandbits(320)
    assert [int(x) for x in w] == [(v >> (32 * i)) & 0xFFFFFFFF for i in range(10)]
    assert s.py.random() == ref_py.random() and s.rs.random_sample() == ref_np.random_sample()
    t.close()


def test_set_cloud_batch_equals_per_tree_set_cloud():
    """nirrt_set_cloud_batch: the kept points (dim-strided in the tree's arena) and the policy scalars land in the trees exactly
    like nirrt_set_cloud's - checked through what the loop DOES with them: trees set either way, seeded alike, draw the same
    samples from their clouds and grow the same tree (SamplePointCloud, nirrt_star_png_2d.py:129-130)"""
    import random
    import torch
    from nirrt_star_amd import _hip, sampling, worlds
    pr = worlds.problem_2d(worlds.random_world_2d(1, "b30"), 0)
    rs = np.random.RandomState(3)

    def trees3():
        ts = [_hip.HipTree(2, 400, pr["x_start"], pr["x_goal"], 10, pr["search_radius"], 3, pr["env"]) for _ in range(3)]
        for t in ts:
            t.set_informed(*sampling.informed_frame(pr["x_start"], pr["x_goal"]))
        _hip.set_generators(ts, [_hip.np_state(np.random.RandomState(50 + k)) for k in range(3)], [_hip.py_state(random.Random(50 + k)) for k in range(3)])
        return ts

    pts = rs.uniform(20, 200, size=(3, 64, 3))
    pts[:, :, 2] = 0.0
    clouds = torch.from_numpy(pts).cuda()
    n_pts = np.array([64, 10, 64], dtype=np.int32)
    pb = (rs.uniform(size=(3, 64)) < 0.4).astype(np.uint8)
    pb[2, :] = 0
    pb[2, 5] = 1                       # a one-point prediction: np.random.randint(0, 1) draws nothing
    pred = torch.from_numpy(pb).cuda()
    c_upd = [np.inf, np.inf, np.inf]
    a = trees3()
    kept = _hip.set_cloud_batch(a, clouds.data_ptr(), 64 * 3, n_pts, pred.data_ptr(), 64, 0.7, 0.9, c_upd)
    b = trees3()
    for k, t in enumerate(b):
        sel = pts[k, : n_pts[k], :2][pb[k, : n_pts[k]] != 0]
        assert kept[k] == len(sel)
        t.set_cloud(sel, 0.7, 0.9, c_upd[k])
    flags = _hip.F_IRRT | _hip.F_PNG
    ra = _hip.run_sampling(a, 400, flags=flags, want_trace=True)
    rb = _hip.run_sampling(b, 400, flags=flags, want_trace=True)
    assert np.array_equal(ra["status"], rb["status"]) and np.array_equal(ra["iters_done"], rb["iters_done"])
    assert np.array_equal(ra["np_used"], rb["np_used"]) and np.array_equal(ra["py_used"], rb["py_used"])
    for ta, tb in zip(a, b):
        va, pa = ta.download()
        vb, pb_ = tb.download()
        assert np.array_equal(pa, pb_) and np.array_equal(va, vb) and len(va) > 20
    for t in a + b:
        t.close()
