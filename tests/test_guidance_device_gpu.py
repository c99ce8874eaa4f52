"""Guidance clouds generated on the device (nirrt_guidance_clouds) vs the host path that reference-generated fixtures pin
(tests/test_guidance_fixtures.py): same generator consumption, same candidate filtering, same down-sampling - the clouds must be
BIT-equal, for the whole-image and the ellipse-restricted 2D clouds and the whole-box 3D cloud
(point_cloud_mask_utils.py:35-73, 104-174; point_cloud_mask_utils_3d.py:83-113)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _streams(seed, n_words):
    import torch
    from nirrt_star_amd import batch
    st = batch.ProblemStreams(seed)
    st.prime(n_words + 4096, 0, torch.device("cuda", 0))
    return st


@pytest.mark.parametrize("world,pair", [(0, 0), (3, 1), (7, 2)])
def test_device_clouds_2d_equal_host_clouds(world, pair):
    import torch
    from nirrt_star_amd import batch, pointcloud as pcu, sampling, worlds
    pr = worlds.problem_2d(worlds.random_world_2d(world, "b30"), pair)
    g = batch.Guidance(None, 2, 10)
    frame = sampling.informed_frame(pr["x_start"], pr["x_goal"])
    n_words = 2 * 2 * g.n_points * g.scale
    xs, xg = np.asarray(pr["x_start"], dtype=np.float64), np.asarray(pr["x_goal"], dtype=np.float64)
    for cbest in (np.inf, 1.6 * frame[0], 1.05 * frame[0]):
        sd, sh = _streams(1000 + world, 4 * n_words), _streams(1000 + world, 4 * n_words)
        for _ in range(2):    # two clouds in a row: the generator position carries over
            jobs, n_raw, nw = g._device_jobs([0], [pr], [sd], [cbest], [frame], torch.device("cuda", 0))
            out = torch.zeros((1, g.n_points, 3), dtype=torch.float64, device="cuda")
            from nirrt_star_amd import pointops
            n_cand, n_out = pointops.guidance_clouds(jobs, n_raw, g.n_points, out, 0)
            sd.advance_np(nw)
            dev_cloud = out.cpu().numpy()[0, : n_out[0], :2]
            if cbest < np.inf:
                host_cloud = pcu.ellipsoid_point_cloud_sampling(xs, xg, cbest / frame[0], pr["binary_mask"], g.n_points, g.n_points * g.scale, sh.rs)
                cand = None
            else:
                host_cloud = pcu.generate_rectangle_point_cloud(pr["binary_mask"], g.n_points, g.scale, sh.rs)
            assert dev_cloud.shape == host_cloud.shape
            assert np.array_equal(dev_cloud, host_cloud)
            # both generators sit at the same position afterwards
            assert np.array_equal(sd.peek_np(8), sh.peek_np(8))


def test_device_cloud_3d_box_equals_host_cloud():
    import torch
    from nirrt_star_amd import batch, pointcloud as pcu, pointops, sampling, worlds
    np.random.seed(4)
    pr = worlds.problem_3d(worlds.random_world_3d(4))
    g = batch.Guidance(None, 3, 10)
    frame = sampling.informed_frame(pr["x_start"], pr["x_goal"])
    n_words = 2 * 3 * g.n_points * g.scale
    sd, sh = _streams(77, 2 * n_words), _streams(77, 2 * n_words)
    jobs, n_raw, nw = g._device_jobs([0], [pr], [sd], [np.inf], [frame], torch.device("cuda", 0))
    out = torch.zeros((1, g.n_points, 3), dtype=torch.float64, device="cuda")
    n_cand, n_out = pointops.guidance_clouds(jobs, n_raw, g.n_points, out, 0)
    sd.advance_np(nw)
    host_cloud = pcu.generate_rectangle_point_cloud_3d(pr["env"], g.n_points, g.scale, clearance=0, rng=sh.rs)
    assert np.array_equal(out.cpu().numpy()[0, : n_out[0]], host_cloud)
    assert np.array_equal(sd.peek_np(8), sh.peek_np(8))


def test_set_cloud_batch_equals_per_tree_set_cloud():
    """nirrt_set_cloud_batch: the kept points and policy scalars land in the trees exactly like nirrt_set_cloud's"""
    import torch
    from nirrt_star_amd import _hip, worlds
    pr = worlds.problem_2d(worlds.random_world_2d(1, "b30"), 0)
    rs = np.random.RandomState(3)
    trees = [_hip.HipTree(2, 200, pr["x_start"], pr["x_goal"], 10, pr["search_radius"], 3, pr["env"]) for _ in range(3)]
    clouds = torch.from_numpy(rs.uniform(5, 200, size=(3, 64, 3))).cuda()
    n_pts = np.array([64, 10, 0], dtype=np.int32)
    pred = torch.from_numpy((rs.uniform(size=(3, 64)) < 0.4).astype(np.uint8)).cuda()
    kept = _hip.set_cloud_batch(trees, clouds.data_ptr(), 64 * 3, n_pts, pred.data_ptr(), 64, 0.5, 0.9, [70.0, 80.0, np.inf])
    pc, pb = clouds.cpu().numpy(), pred.cpu().numpy()
    for b in range(3):
        assert kept[b] == int(pb[b, : n_pts[b]].sum())
    for t in trees:
        t.close()
