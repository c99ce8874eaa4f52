"""L4 parity of the PointNet++ guidance net against the reference model's own outputs
(tests/golden/pointnet2_ref.npz: reference get_model on CPU, fp32).  Tolerances:
logits <= 1e-3 abs, argmax agreement >= 99 %, identical FPS indices when the start index is injected."""
import numpy as np
import pytest
import torch

from conftest import load_golden, synthetic_checkpoint_root


def _model(g, device):
    from nirrt_star_amd.pointnet2 import get_model
    torch.manual_seed(int(g["seed"]))
    m = get_model(2)   # same construction order as the reference -> same init for the same seed
    sd = m.state_dict()
    for k in g:
        if k.startswith("bn::"):
            sd[k[4:]] = torch.from_numpy(g[k])
    m.load_state_dict(sd)
    return m.to(device).eval()


def _check(g, m, device, tol):
    x = torch.from_numpy(g["input"]).to(device)
    starts = [torch.tensor([int(g["fps%d" % i][0])]) for i in range(4)]
    with torch.no_grad():
        logp, l4 = m(x, fps_starts=starts)
    for i in range(4):
        assert np.array_equal(m.last_fps[i][0].cpu().numpy(), g["fps%d" % i]), "FPS level %d" % i
    logp = logp.cpu().numpy()
    assert np.max(np.abs(logp - g["logp"])) <= tol
    assert np.mean(logp.argmax(-1) == g["logp"].argmax(-1)) >= 0.99
    assert np.max(np.abs(l4.cpu().numpy() - g["l4"])) <= tol * max(1.0, float(np.abs(g["l4"]).max()))


@pytest.mark.oracle_pointops
def test_cpu_unfolded_and_folded_match_reference():
    g = load_golden("pointnet2_ref")
    m = _model(g, "cpu")
    _check(g, m, "cpu", 1e-4)
    _check(g, m.fold(), "cpu", 1e-3)


@pytest.mark.oracle_pointops
def test_cpu_ragged_batch_is_one_forward_per_cloud():
    """The claim behind the ragged batches of round 6 (PointNet2.forward(n_valid=...)), checked with the reference's own torch
    operators on the CPU: only the first set-abstraction level looks at a cloud as a whole - with ITS sampling and ball queries
    restricted to the cloud's own points, a cloud inside a batch of clouds of other sizes (zeros behind its points) gets the labels
    of a forward over that cloud alone (pointnet2_wrapper.py:43-58 classifies one cloud per call)."""
    from nirrt_star_amd.pointnet2 import pc_normalize
    g = load_golden("pointnet2_ref")
    m = _model(g, "cpu").fold()
    rng = np.random.RandomState(31)
    sizes = [2048, 1500, 1333]
    n_max = max(sizes)
    blocks, fps = [], []
    x_all = torch.zeros(len(sizes), 6, n_max)
    for j, n in enumerate(sizes):
        pc = np.concatenate([rng.uniform(0, 224, (n, 2)), np.zeros((n, 1))], axis=1).astype(np.float32)
        sm = (rng.uniform(size=n) < 0.02).astype(np.float32)
        gm = (rng.uniform(size=n) < 0.02).astype(np.float32)
        blk = np.concatenate([pc_normalize(pc).T, sm[None], gm[None], ((sm + gm) == 0).astype(np.float32)[None]], axis=0)
        blocks.append(torch.from_numpy(blk.astype(np.float32)))
        x_all[j, :, :n] = blocks[-1]
        fps.append([int(rng.randint(n)), int(rng.randint(1024)), int(rng.randint(256)), int(rng.randint(64))])
    nv = torch.tensor(sizes, dtype=torch.int32)
    with torch.no_grad():
        logp_all, _ = m(x_all, fps_starts=[torch.tensor([f[k] for f in fps]) for k in range(4)], n_valid=nv)
    fps_all = [t.clone() for t in m.last_fps]
    for j, n in enumerate(sizes):
        with torch.no_grad():
            logp1, _ = m(blocks[j][None], fps_starts=[torch.tensor([v]) for v in fps[j]])
        for lvl in range(4):
            assert torch.equal(m.last_fps[lvl][0], fps_all[lvl][j]), (j, lvl)
        a, b = logp_all[j, :n].numpy(), logp1[0].numpy()
        assert np.max(np.abs(a - b)) <= 1e-4, (j, float(np.max(np.abs(a - b))))
        assert np.mean(a.argmax(-1) == b.argmax(-1)) >= 0.995


def test_state_dict_layout_is_the_reference_one():
    from nirrt_star_amd.pointnet2 import get_model
    keys = list(get_model(2).state_dict().keys())
    assert len(keys) == 240
    assert keys[0] == "sa1.conv_blocks.0.0.weight" and "fp1.mlp_bns.2.running_var" in keys and keys[-1] == "conv2.bias"


@pytest.mark.oracle_pointops
def test_wrapper_roundtrip_cpu(tmp_path):
    """PNGWrapper on CPU with a synthetic checkpoint in the reference format: classify + neural connect run end to end."""
    from nirrt_star_amd import png_wrapper
    ck = png_wrapper.make_synthetic_checkpoint(str(tmp_path / "results/model_training/pointnet2_2d/checkpoints/best_pointnet2_2d.pth"), device="cpu")
    w = png_wrapper.PNGWrapper(root_dir=str(tmp_path), device="cpu")
    g = load_golden("pointnet2_ref")
    torch.manual_seed(0)
    pred, score = w.classify_path_points(g["pc"], g["start_mask"], g["goal_mask"])
    assert pred.shape == (2048,) and score.shape == (2048,) and score.dtype == np.float32
    assert set(np.unique(pred)) <= {0, 1}
    ok, runs, mask = w.generate_connected_path_points(g["pc"], np.array([30.0, 30.0]), np.array([200.0, 200.0]),
                                                      {"env_dims": (224, 224)}, 10, 2)
    assert mask.shape == (2048,) and mask.dtype == np.float32 and 1 <= runs <= 2


@pytest.mark.gpu
def test_gpu_hip_pointops_match_reference():
    g = load_golden("pointnet2_ref")
    m = _model(g, "cuda").fold()
    _check(g, m, "cuda", 1e-3)


@pytest.mark.gpu
def test_gpu_pointops_equal_cpu_semantics():
    """HIP kernels vs the plain-torch restatement of the reference operators (oracle/pointops_ref.py)"""
    from nirrt_star_amd import pointops
    from oracle import pointops_ref as ref
    torch.manual_seed(0)
    xyz = torch.rand(2, 2048, 3)
    start = torch.tensor([5, 77])
    a = ref.farthest_point_sample(xyz, 256, start)
    b = pointops.farthest_point_sample(xyz.cuda(), 256, start).cpu()
    assert torch.equal(a, b)
    new = torch.gather(xyz, 1, a[..., None].expand(2, 256, 3))
    for r, k in ((0.1, 16), (0.2, 32)):
        ga = ref.ball_query(r, k, xyz, new)
        gb = pointops.ball_query(r, k, xyz.cuda(), new.cuda()).cpu()
        assert (ga == gb).float().mean() > 0.999   # membership on the r^2 boundary depends on the matmul rounding
    da, ia = ref.three_nn(xyz, new)
    db, ib = pointops.three_nn(xyz.cuda(), new.cuda())
    assert (ia == ib.cpu()).float().mean() > 0.999
    assert torch.allclose(da, db.cpu(), atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("n,s", [(40, 16), (64, 16), (200, 64), (256, 64), (1000, 256), (1024, 256), (2048, 1024), (5000, 512)])
def test_gpu_fps_every_kernel_variant(n, s):
    """each (points-per-lane, waves) instantiation of k_fps_wave and the LDS k_fps pick the reference's indices"""
    from nirrt_star_amd import pointops
    from oracle import pointops_ref as ref
    torch.manual_seed(n)
    xyz = torch.rand(3, n, 3)
    xyz[1, : n // 2] = xyz[1, n // 2: 2 * (n // 2)]        # duplicated points: ties resolved to the lowest index
    start = torch.tensor([0, n - 1, n // 3])
    a = ref.farthest_point_sample(xyz, s, start)
    b = pointops.farthest_point_sample(xyz.cuda(), s, start).cpu()
    assert torch.equal(a, b)
    new = torch.gather(xyz, 1, a[..., None].expand(3, s, 3))
    da, ia = ref.three_nn(xyz, new)
    db, ib = pointops.three_nn(xyz.cuda(), new.cuda())
    assert (ia == ib.cpu()).float().mean() > 0.99
    assert torch.allclose(da, db.cpu(), atol=1e-6)


@pytest.mark.gpu
def test_gpu_cloud_downsampling_equals_host_restatement():
    """k_fps_f64 == the numpy restatement of open3d's farthest_point_down_sample (oracle/pointops_ref.py, same float64
    arithmetic); asking for more samples than points is an error like in open3d"""
    from nirrt_star_amd import pointcloud
    from oracle import pointops_ref as ref
    rng = np.random.default_rng(5)
    for n, s in ((9000, 2048), (3000, 2048), (2500, 100), (10240, 2048)):
        pts = np.concatenate([rng.uniform(0, 224, size=(n, 2)), np.zeros((n, 1))], axis=1)
        got = pointcloud.farthest_point_down_sample(pts, s)            # HIP
        assert np.array_equal(got, pts[ref.farthest_point_down_sample_f64(pts, s)])
    pts3 = rng.uniform(0, 50, size=(7000, 3))
    assert np.array_equal(pointcloud.farthest_point_down_sample(pts3, 2048), pts3[ref.farthest_point_down_sample_f64(pts3, 2048)])
    with pytest.raises(ValueError):
        pointcloud.farthest_point_down_sample(pts3[:10], 11)


@pytest.mark.gpu
@pytest.mark.parametrize("reg", ["1", "0"])
def test_gpu_cloud_downsampling_both_kernels_on_ties_and_ragged_sizes(reg, monkeypatch):
    """the register-resident down-sampling kernel (k_fps_f64_reg, round 6; NIRRT_FPS64_REG=0: the L2-streaming k_fps_f64) against
    the numpy restatement of open3d's farthest_point_down_sample: clouds full of exact ties (lattice points, duplicates: the FIRST
    maximum wins), sizes around the 512-thread / 64-lane boundaries, planar and spatial, several clouds in one launch"""
    from nirrt_star_amd import pointops
    from oracle import pointops_ref as ref
    monkeypatch.setenv("NIRRT_FPS64_REG", reg)
    rng = np.random.default_rng(17)
    clouds, ns = [], []
    for n, s in ((10240, 2048), (10239, 2048), (513, 512), (512, 100), (65, 64), (2049, 2048), (7777, 2048)):
        lattice = rng.integers(0, 12, size=(n, 3)).astype(np.float64)          # many equal distances, many duplicate points
        clouds.append(lattice)
        ns.append(s)
        planar = np.concatenate([rng.integers(0, 40, size=(n, 2)).astype(np.float64) * 0.5, np.zeros((n, 1))], axis=1)
        clouds.append(planar)
        ns.append(s)
    clouds.append(rng.uniform(0, 50, size=(9000, 3)))
    ns.append(2048)
    for s in sorted(set(ns)):      # (one launch per sample count: several clouds each)
        group = [c for c, s_ in zip(clouds, ns) if s_ == s]
        masks = pointops.farthest_point_down_sample_f64_batch(group, s)
        for c, m in zip(group, masks):
            assert np.array_equal(np.asarray(m, dtype=bool), ref.farthest_point_down_sample_f64(c, s)), (len(c), s)


@pytest.mark.gpu
@pytest.mark.parametrize("level", [0, 1])
def test_gpu_fused_mfma_set_abstraction_equals_library_gemms(level):
    """k_sa_mlp (gather + 3 x (GEMM + bias + ReLU) + max over the group on v_mfma_f32_16x16x4_f32 tiles) vs the same branch as
    torch gather / addmm / max, for both radii of SA1 and SA2 with B = 3 (fp32 both ways: only the summation order differs)"""
    from nirrt_star_amd import pointops
    g = load_golden("pointnet2_ref")
    m = _model(g, "cuda").fold()
    sa = (m.sa1, m.sa2)[level]
    assert sa._fused is not None and all(p is not None for p in sa._fused)
    torch.manual_seed(level)
    B, N, C = 3, (2048, 1024)[level], (6, 96)[level]
    xyz = torch.rand(B, N, 3, device="cuda")
    feats = torch.randn(B, N, C, device="cuda")
    S = sa.npoint
    idx = pointops.farthest_point_sample(xyz, S, torch.tensor([0, 5, 9]))
    new_xyz = torch.gather(xyz, 1, idx[..., None].expand(B, S, 3))
    width = sum(layers[-1][0].shape[0] for layers in sa._folded)
    out = torch.zeros(B, S, width, device="cuda")
    off = 0
    for bi, (radius, K) in enumerate(zip(sa.radii, sa.nsamples)):
        gidx = pointops.ball_query(radius * (4 if level else 2), K, xyz, new_xyz)      # wider balls: groups with distinct members
        assert pointops.sa_mlp(feats, xyz, new_xyz, gidx, sa._fused[bi], out, off)
        flat = gidx.reshape(B, S * K)
        g_xyz = torch.gather(xyz, 1, flat[..., None].expand(B, S * K, 3)).view(B, S, K, 3) - new_xyz[:, :, None, :]
        g_feat = torch.gather(feats, 1, flat[..., None].expand(B, S * K, C)).view(B, S, K, C)
        x = torch.cat([g_feat, g_xyz], dim=-1).reshape(B * S * K, -1)
        for w, b in sa._folded[bi]:
            x = torch.relu(torch.addmm(b, x, w.t()))
        ref = x.view(B, S, K, -1).max(dim=2)[0]
        c3 = ref.shape[-1]
        got = out[:, :, off:off + c3]
        assert torch.allclose(got, ref, rtol=1e-4, atol=1e-4), (level, bi, float((got - ref).abs().max()))
        off += c3
    torch.cuda.synchronize()


@pytest.mark.gpu
def test_gpu_group_rows_equal_torch_gathers():
    """k_group_rows writes exactly what gather + subtract + cat produce (plus the zero column)"""
    from nirrt_star_amd import pointops
    torch.manual_seed(3)
    B, N, S, K, C = 3, 256, 64, 16, 256
    feats = torch.randn(B, N, C, device="cuda")
    xyz = torch.rand(B, N, 3, device="cuda")
    new_xyz = xyz[:, :S].contiguous()
    gidx = torch.randint(0, N, (B, S, K), device="cuda")
    got = pointops.group_rows(feats, xyz, new_xyz, gidx)
    flat = gidx.reshape(B, S * K)
    g_xyz = torch.gather(xyz, 1, flat[..., None].expand(B, S * K, 3)).view(B, S, K, 3) - new_xyz[:, :, None, :]
    g_feat = torch.gather(feats, 1, flat[..., None].expand(B, S * K, C)).view(B, S, K, C)
    ref = torch.cat([g_feat, g_xyz, torch.zeros(B, S, K, 1, device="cuda")], dim=-1).reshape(B * S * K, C + 4)
    assert got.shape == ref.shape and torch.equal(got, ref)
    assert pointops.group_rows(feats[:, :, :6].contiguous(), xyz, new_xyz, gidx) is None      # 6 channels: not float4 rows


@pytest.mark.gpu
@pytest.mark.parametrize("c1,c2,s", [(96, 256, 64), (0, 128, 256), (512, 1024, 16)])
def test_gpu_fp_rows_equal_torch_interpolation(c1, c2, s):
    """k_fp_rows against the reference formulation (pointnet2_utils.py:295-309) evaluated with torch ops"""
    from nirrt_star_amd import pointops
    torch.manual_seed(c2)
    B, N = 2, 4 * s
    xyz1 = torch.rand(B, N, 3, device="cuda")
    xyz2 = xyz1[:, ::4].contiguous()          # every fourth point is its own neighbour at distance 0: the 1e-8 guard matters
    feats1 = torch.randn(B, N, c1, device="cuda") if c1 else None
    feats2 = torch.randn(B, s, c2, device="cuda")
    d, idx = pointops.three_nn(xyz1, xyz2)
    got = pointops.fp_rows(feats1, feats2, d, idx)
    recip = 1.0 / (d + 1e-8)
    wgt = recip / recip.sum(dim=2, keepdim=True)
    nb = torch.gather(feats2, 1, idx.reshape(B, N * 3)[..., None].expand(B, N * 3, c2)).view(B, N, 3, c2)
    interp = (nb * wgt[..., None]).sum(dim=2)
    ref = (interp if feats1 is None else torch.cat([feats1, interp], dim=-1)).reshape(B * N, c1 + c2)
    assert got.shape == ref.shape
    if c1:
        assert torch.equal(got[:, :c1], ref[:, :c1])
    assert torch.allclose(got[:, c1:], ref[:, c1:], rtol=1e-6, atol=1e-6), float((got - ref).abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("dim", [2, 3])
def test_gpu_net_input_bit_equal_to_host_assembly(dim):
    """k_net_input against PNGWrapper.network_input + get_point_cloud_mask_around_points (numpy, the reference's arithmetic)"""
    from nirrt_star_amd import pointops, png_wrapper, pointcloud as pcu
    rng = np.random.RandomState(11 + dim)
    sizes = [2048, 2048, 1777, 1777, 2048, 100]
    stride = 2048
    block = np.zeros((len(sizes), stride, 3))
    starts, goals = [], []
    for j, n in enumerate(sizes):
        block[j, :n, :dim] = rng.uniform(0, 224, (n, dim))
        starts.append(block[j, rng.randint(n), :].copy() + 0.25)
        goals.append(block[j, rng.randint(n), :].copy() - 0.5)
    if dim == 2:
        for v in starts + goals:
            v[2] = 0.0
    dev_block = torch.from_numpy(block).cuda()
    radius = 10.0
    for n in sorted(set(sizes)):
        rows = [j for j, m in enumerate(sizes) if m == n]
        got = pointops.net_input(dev_block, rows, n, np.stack([starts[j] for j in rows]), np.stack([goals[j] for j in rows]), radius).cpu().numpy()
        for k, j in enumerate(rows):
            c = block[j, :n, :dim]
            sm = pcu.get_point_cloud_mask_around_points(c, starts[j][np.newaxis, :dim], radius).astype(np.float32)
            gm = pcu.get_point_cloud_mask_around_points(c, goals[j][np.newaxis, :dim], radius).astype(np.float32)
            ref = png_wrapper.PNGWrapper.network_input(c.astype(np.float32), sm, gm)
            assert sm.sum() > 0 and gm.sum() > 0
            assert np.array_equal(got[k], ref), (n, j, np.abs(got[k] - ref).max())


@pytest.mark.gpu
def test_gpu_ragged_batch_equals_one_forward_per_cloud():
    """Round 6: clouds of DIFFERENT sizes in one forward (PointNet2.forward(n_valid=...), the batched cloud refresh).  The
    reference classifies one cloud at a time (pointnet2_wrapper.py:43-58); against one forward per cloud, the ragged batch must
    give bit-equal input blocks (zeros behind a cloud's own points), identical sampling indices at all four levels, identical ball
    queries at the first level, and the same log-probabilities up to the library GEMMs' summation order (<= 1e-4, labels >= 99.5 %)."""
    from nirrt_star_amd import pointops
    g = load_golden("pointnet2_ref")
    m = _model(g, "cuda").fold()
    rng = np.random.RandomState(23)
    sizes = [2048, 1801, 2048, 1280, 2047, 1999, 1536]
    n_max = max(sizes)
    block = np.zeros((len(sizes), 2048, 3))
    starts, goals = np.zeros((len(sizes), 3)), np.zeros((len(sizes), 3))
    for j, n in enumerate(sizes):
        block[j, :n, :2] = rng.uniform(0, 224, (n, 2))
        starts[j] = block[j, rng.randint(n)] + 0.25
        goals[j] = block[j, rng.randint(n)] - 0.5
        starts[j, 2] = goals[j, 2] = 0.0
    dev_block = torch.from_numpy(block).cuda()
    fps = [[int(rng.randint(n)), int(rng.randint(1024)), int(rng.randint(256)), int(rng.randint(64))] for n in sizes]
    nv = torch.tensor(sizes, dtype=torch.int32).cuda()
    rows = list(range(len(sizes)))
    x_all = pointops.net_input(dev_block, rows, n_max, starts, goals, 10.0, n_each=nv)
    st_all = [torch.tensor([f[k] for f in fps]) for k in range(4)]
    with torch.no_grad():
        logp_all, _ = m(x_all, fps_starts=st_all, n_valid=nv)
    fps_all = [t.cpu().numpy() for t in m.last_fps]
    feats = x_all.permute(0, 2, 1).contiguous()
    xyz = feats[:, :, :3].contiguous()
    new_xyz = torch.gather(xyz, 1, m.last_fps[0][..., None].expand(len(sizes), 1024, 3))
    balls_all = [pointops.ball_query(r, k, xyz, new_xyz, nv).cpu().numpy() for r, k in zip(m.sa1.radii, m.sa1.nsamples)]
    agree = []
    for j, n in enumerate(sizes):
        x1 = pointops.net_input(dev_block, [j], n, starts[j:j + 1], goals[j:j + 1], 10.0)
        assert torch.equal(x_all[j, :, :n], x1[0]) and float(x_all[j, :, n:].abs().max() if n < n_max else 0.0) == 0.0
        with torch.no_grad():
            logp1, _ = m(x1, fps_starts=[torch.tensor([v]) for v in fps[j]])
        for lvl in range(4):
            assert np.array_equal(m.last_fps[lvl][0].cpu().numpy(), fps_all[lvl][j]), (j, lvl)
        xyz1 = x1.permute(0, 2, 1)[:, :, :3].contiguous()
        nx1 = torch.gather(xyz1, 1, m.last_fps[0][..., None].expand(1, 1024, 3))
        for bi, (r, k) in enumerate(zip(m.sa1.radii, m.sa1.nsamples)):
            assert np.array_equal(pointops.ball_query(r, k, xyz1, nx1).cpu().numpy()[0], balls_all[bi][j]), (j, bi)
        a, b = logp_all[j, :n].cpu().numpy(), logp1[0].cpu().numpy()
        assert np.max(np.abs(a - b)) <= 1e-4, (j, float(np.max(np.abs(a - b))))
        agree.append(np.mean(a.argmax(-1) == b.argmax(-1)))
    assert min(agree) >= 0.995, agree


def test_numpy_reduction_orders_the_device_input_assembly_relies_on():
    """k_net_input (csrc/pointops.hip) restates pc_normalize with the evaluation order numpy uses for it: np.mean over axis 0 of
    a C-contiguous (N, 3) float32 array adds the rows in index order (no pairwise blocking on the outer axis), the row norm is
    (x^2 + y^2) + z^2.  Pinned here on the CPU so that a numpy whose reductions work differently is noticed without a GPU."""
    from nirrt_star_amd.pointnet2 import pc_normalize
    rng = np.random.RandomState(5)
    for n in (7, 100, 1777, 2048):
        a = (rng.uniform(0, 224, (n, 3))).astype(np.float32)
        acc = np.zeros(3, dtype=np.float32)
        for i in range(n):
            acc = acc + a[i]
        mean = acc / np.float32(n)
        assert np.array_equal(mean, np.mean(a, axis=0))
        c = a - mean
        norm = np.sqrt((c[:, 0] * c[:, 0] + c[:, 1] * c[:, 1]) + c[:, 2] * c[:, 2])
        assert np.array_equal(norm, np.sqrt(np.sum(c ** 2, axis=1)))
        assert np.array_equal(c / np.max(norm), pc_normalize(a))


@pytest.mark.gpu
def test_fused_gemm_epilogue_and_plain_addmm_relu_give_the_same_labels(monkeypatch):
    """bias + ReLU in the hipBLASLt epilogue (torch._addmm_activation, a private entry point) against addmm followed
    by relu on the fixture clouds: scores within the L4 tolerance, labels equal on >= 99.9 % of the points"""
    import torch
    from nirrt_star_amd import png_wrapper, pointnet2
    g = load_golden("config3_nirrtc2d_real")
    w = png_wrapper.PNGWrapper(root_dir=synthetic_checkpoint_root(2), device="cuda")
    w.use_graph = False
    out = {}
    for fused in (True, False):
        monkeypatch.setattr(pointnet2, "FUSED_EPILOGUE", fused and hasattr(torch, "_addmm_activation"))
        res = []
        for i in range(min(3, int(g["n_calls"]))):
            st = [torch.tensor([3]), torch.tensor([2]), torch.tensor([1]), torch.tensor([0])]
            res.append(w.classify_batch([g["call%d_pc" % i]], [g["call%d_start" % i].astype(np.float32)], [g["call%d_goal" % i].astype(np.float32)],
                                        fps_starts=st))
        out[fused] = res
    for (p1, s1), (p0, s0) in zip(out[True], out[False]):
        assert np.max(np.abs(s1 - s0)) <= 1e-3
        assert np.mean(p1 == p0) >= 0.999
