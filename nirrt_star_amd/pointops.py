"""Point operators of the PointNet++ guidance net: hand-written HIP kernels of libnirrt_hip.so (csrc/pointops.hip),
launched on the current torch stream.

ONE implementation: tensors must live on the GPU; the library being absent, a host without a visible GPU, or a CPU
tensor is an error.  There is no hook for another implementation in this module: the CPU test-suite replaces these
functions from the outside (tests/conftest.py monkeypatches the module attributes on a host without a GPU).

Semantics (reference pointnet_pointnet2/models/pointnet2_utils.py): farthest_point_sample :65-86 with the start index
drawn by torch.randint on the CPU generator, query_ball_point :89-109 (first K indices in ascending order, padded with
the first), 3-NN of the feature propagation :295-299.
"""
import ctypes as C

import torch

def _need_cuda(name, t):
    if not t.is_cuda:
        raise RuntimeError("nirrt_star_amd.pointops.%s got a CPU tensor: the package only has the HIP kernels "
                           "(move the model / cloud to 'cuda')" % name)


def _need_device(name):
    from . import _hip
    if _hip.device_count() <= 0:
        raise RuntimeError("nirrt_star_amd.pointops.%s needs a visible MI355X: the down-sampling only exists as a HIP "
                           "kernel (k_fps_f64), there is no CPU path" % name)


def _lib():
    from . import _hip
    L = _hip.load()
    if not hasattr(L, "_pn2_ready"):
        vp = C.c_void_p
        L.nirrt_pn2_fps.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp]
        L.nirrt_pn2_ball_query.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, vp, vp]
        L.nirrt_pn2_three_nn.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp]
        L.nirrt_pn2_fps_ragged.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp]
        L.nirrt_pn2_ball_query_ragged.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, vp, vp, vp]
        L.nirrt_pn2_net_input_ragged.argtypes = [vp, C.c_int64, vp, C.c_int, C.c_int, vp, vp, vp, C.c_double, vp, vp]
        L.nirrt_pn2_net_input_masks_ragged.argtypes = [vp, C.c_int64, vp, C.c_int, C.c_int, vp, vp, vp, C.c_int64, vp, vp]
        L.nirrt_fps_f64.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int]
        L.nirrt_fps_f64_batch.argtypes = [vp, C.c_int, vp, vp, vp, C.c_int]
        L.nirrt_guidance_clouds.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_int]
        L.nirrt_guidance_clouds.restype = C.c_int
        L.nirrt_pn2_sa_mlp.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, vp, vp,
                                       C.c_int, vp, vp, C.c_int, vp, C.c_int, C.c_int, vp]
        L.nirrt_pn2_group_rows.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]
        L.nirrt_pn2_fp_rows.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]
        L.nirrt_pn2_net_input.argtypes = [vp, C.c_int64, vp, C.c_int, C.c_int, vp, vp, C.c_double, vp, vp]
        L.nirrt_pn2_net_input_masks.argtypes = [vp, C.c_int64, vp, C.c_int, C.c_int, vp, vp, C.c_int64, vp, vp]
        L.nirrt_connect_round.argtypes = [vp, C.c_int, C.c_double, vp, vp, vp, C.c_int]
        L.nirrt_connect_masks.argtypes = [vp, C.c_int, C.c_double, vp, C.c_int]
        for f in (L.nirrt_pn2_fps, L.nirrt_pn2_ball_query, L.nirrt_pn2_three_nn, L.nirrt_fps_f64, L.nirrt_fps_f64_batch,
                  L.nirrt_pn2_sa_mlp, L.nirrt_pn2_group_rows, L.nirrt_pn2_fp_rows, L.nirrt_pn2_net_input, L.nirrt_pn2_fps_ragged,
                  L.nirrt_pn2_ball_query_ragged, L.nirrt_pn2_net_input_ragged, L.nirrt_pn2_net_input_masks_ragged):
            f.restype = C.c_int
        L._pn2_ready = True
    return L


def _stream(t):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _check(rc, what):
    if rc != 0:
        raise RuntimeError("libnirrt_hip %s failed (%d)" % (what, rc))


def farthest_point_sample(xyz, npoint, start=None, n_valid=None):
    """xyz (B, N, 3) -> indices (B, npoint) long; start (B,) long or drawn with torch.randint on the CPU generator.
    n_valid (B,) int32 on the device: ragged batch - cloud b holds n_valid[b] <= N points, its picks are those of a call over that
    cloud alone (start[b] < n_valid[b] is the caller's: the reference draws it with the cloud's own size)"""
    B, N, _ = xyz.shape
    if start is None:
        if n_valid is not None:
            raise ValueError("farthest_point_sample: a ragged batch needs its start indices (one draw per cloud with the cloud's own size)")
        start = torch.randint(0, N, (B,), dtype=torch.long)
    _need_cuda("farthest_point_sample", xyz)
    start = start.to(xyz.device)
    xyz = xyz.contiguous().float()
    out = torch.empty(B, npoint, dtype=torch.long, device=xyz.device)
    if n_valid is None:
        _check(_lib().nirrt_pn2_fps(xyz.data_ptr(), B, N, npoint, start.contiguous().data_ptr(), out.data_ptr(), _stream(xyz)), "fps")
    else:
        assert n_valid.is_cuda and n_valid.dtype == torch.int32 and n_valid.numel() == B
        _check(_lib().nirrt_pn2_fps_ragged(xyz.data_ptr(), B, N, npoint, start.contiguous().data_ptr(), n_valid.contiguous().data_ptr(),
                                           out.data_ptr(), _stream(xyz)), "fps_ragged")
    return out


def ball_query(radius, nsample, xyz, new_xyz, n_valid=None):
    """first `nsample` indices in ascending order within `radius` of each query, padded with the first -> (B, S, K) long
    (n_valid: ragged batch, see farthest_point_sample)"""
    _need_cuda("ball_query", xyz)
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    out = torch.empty(B, S, nsample, dtype=torch.long, device=xyz.device)
    r2 = float(torch.tensor(radius ** 2, dtype=torch.float32))
    if n_valid is None:
        _check(_lib().nirrt_pn2_ball_query(xyz.contiguous().data_ptr(), new_xyz.contiguous().data_ptr(), B, N, S, nsample,
                                           C.c_float(r2), out.data_ptr(), _stream(xyz)), "ball_query")
    else:
        assert n_valid.is_cuda and n_valid.dtype == torch.int32 and n_valid.numel() == B
        _check(_lib().nirrt_pn2_ball_query_ragged(xyz.contiguous().data_ptr(), new_xyz.contiguous().data_ptr(), B, N, S, nsample,
                                                  C.c_float(r2), n_valid.contiguous().data_ptr(), out.data_ptr(), _stream(xyz)), "ball_query_ragged")
    return out


def three_nn(xyz1, xyz2):
    """3 nearest coarse points per fine point -> (squared distances (B, N, 3), indices (B, N, 3))"""
    _need_cuda("three_nn", xyz1)
    B, N, _ = xyz1.shape
    S = xyz2.shape[1]
    d = torch.empty(B, N, 3, dtype=torch.float32, device=xyz1.device)
    i = torch.empty(B, N, 3, dtype=torch.long, device=xyz1.device)
    _check(_lib().nirrt_pn2_three_nn(xyz1.contiguous().data_ptr(), xyz2.contiguous().data_ptr(), B, N, S, d.data_ptr(),
                                     i.data_ptr(), _stream(xyz1)), "three_nn")
    return d, i


def net_input(clouds, rows, n, starts, goals, radius, n_each=None):
    """the network's input blocks of resident clouds (k_net_input): clouds f64 (n_clouds, stride, 3) on the device, rows = the
    clouds to take (all of n points), starts / goals (len(rows), 3) f64 -> x f32 (len(rows), 6, n), bit-equal to
    PNGWrapper.network_input + get_point_cloud_mask_around_points on the host.  n_each (len(rows),) int32 on the device: ragged
    batch - cloud b has n_each[b] <= n points, its block is what a call over that cloud alone gives, followed by zeros"""
    _need_cuda("net_input", clouds)
    dev = clouds.device
    rows_t = torch.as_tensor(rows, dtype=torch.int32).to(dev)
    st = torch.as_tensor(starts, dtype=torch.float64).reshape(-1, 3).to(dev)
    gl = torch.as_tensor(goals, dtype=torch.float64).reshape(-1, 3).to(dev)
    if n_each is not None:      # ragged: blocks as wide as the largest cloud, zeros behind a cloud's own points
        assert n_each.is_cuda and n_each.dtype == torch.int32 and n_each.numel() == len(rows_t)
        out = torch.zeros(len(rows_t), 6, int(n), dtype=torch.float32, device=dev)
        _check(_lib().nirrt_pn2_net_input_ragged(clouds.data_ptr(), clouds.shape[1], rows_t.data_ptr(), len(rows_t), int(n), n_each.data_ptr(),
                                                 st.data_ptr(), gl.data_ptr(), float(radius), out.data_ptr(), _stream(clouds)), "net_input_ragged")
        return out
    out = torch.empty(len(rows_t), 6, int(n), dtype=torch.float32, device=dev)
    _check(_lib().nirrt_pn2_net_input(clouds.data_ptr(), clouds.shape[1], rows_t.data_ptr(), len(rows_t), int(n), st.data_ptr(),
                                      gl.data_ptr(), float(radius), out.data_ptr(), _stream(clouds)), "net_input")
    return out


def net_input_masks(clouds, rows, n, start_masks, goal_masks, n_each=None):
    """net_input with GIVEN indicator channels: start_masks / goal_masks uint8 (n_clouds, stride) on the device (the masks of
    the neural-connect rounds) -> x f32 (len(rows), 6, n)"""
    _need_cuda("net_input_masks", clouds)
    dev = clouds.device
    assert start_masks.dtype == torch.uint8 and goal_masks.dtype == torch.uint8 and start_masks.is_contiguous() and goal_masks.is_contiguous()
    assert start_masks.shape == goal_masks.shape and start_masks.shape[0] == clouds.shape[0]
    rows_t = torch.as_tensor(rows, dtype=torch.int32).to(dev)
    if n_each is not None:      # ragged (see net_input)
        assert n_each.is_cuda and n_each.dtype == torch.int32 and n_each.numel() == len(rows_t)
        out = torch.zeros(len(rows_t), 6, int(n), dtype=torch.float32, device=dev)
        _check(_lib().nirrt_pn2_net_input_masks_ragged(clouds.data_ptr(), clouds.shape[1], rows_t.data_ptr(), len(rows_t), int(n), n_each.data_ptr(),
                                                       start_masks.data_ptr(), goal_masks.data_ptr(), start_masks.shape[1], out.data_ptr(),
                                                       _stream(clouds)), "net_input_masks_ragged")
        return out
    out = torch.empty(len(rows_t), 6, int(n), dtype=torch.float32, device=dev)
    _check(_lib().nirrt_pn2_net_input_masks(clouds.data_ptr(), clouds.shape[1], rows_t.data_ptr(), len(rows_t), int(n),
                                            start_masks.data_ptr(), goal_masks.data_ptr(), start_masks.shape[1], out.data_ptr(),
                                            _stream(clouds)), "net_input_masks")
    return out


class ConnectJob(C.Structure):
    """nirrt_connect_job of include/nirrt_pointops.h"""
    _fields_ = [("cloud", C.c_void_p), ("pred", C.c_void_p), ("path_mask", C.c_void_p), ("start_mask", C.c_void_p),
                ("goal_mask", C.c_void_p), ("boundary", C.c_void_p), ("n", C.c_int32), ("dim", C.c_int32),
                ("start", C.c_double * 3), ("goal", C.c_double * 3)]


def connect_round(jobs, radius, device_id=0):
    """one neural-connect round of len(jobs) clouds on the device -> (has_path (n,), seed_idx (n, 2), tie (n, 2)) int32"""
    import numpy as np
    n = len(jobs)
    arr = (ConnectJob * n)(*jobs)
    has, seed, tie = np.zeros(n, dtype=np.int32), np.zeros((n, 2), dtype=np.int32), np.zeros((n, 2), dtype=np.int32)
    torch.cuda.current_stream(torch.device("cuda", device_id)).synchronize()
    _check(_lib().nirrt_connect_round(C.cast(arr, C.c_void_p), n, float(radius), has.ctypes.data, seed.ctypes.data, tie.ctypes.data, int(device_id)), "connect_round")
    return has, seed, tie


def connect_masks(jobs, radius, seed_idx, device_id=0):
    """start / goal masks of the next classification from seeds (n, 2): -2 start / goal state, -1 keep, >= 0 cloud point"""
    import numpy as np
    n = len(jobs)
    arr = (ConnectJob * n)(*jobs)
    seed = np.ascontiguousarray(seed_idx, dtype=np.int32).reshape(n, 2)
    torch.cuda.current_stream(torch.device("cuda", device_id)).synchronize()
    _check(_lib().nirrt_connect_masks(C.cast(arr, C.c_void_p), n, float(radius), seed.ctypes.data, int(device_id)), "connect_masks")


def group_rows(feats, xyz, new_xyz, gidx):
    """sample_and_group's concatenation in one pass (k_group_rows): (B * S * K, C + 4) rows [feats[b, gidx], xyz[b, gidx] -
    new_xyz[b, s], 0]; None when C is not a multiple of 4 (caller keeps the torch gathers)"""
    B, N, C = feats.shape
    S, K = gidx.shape[1], gidx.shape[2]
    if C % 4:
        return None
    _need_cuda("group_rows", feats)
    out = torch.empty(B * S * K, C + 4, dtype=torch.float32, device=feats.device)
    _check(_lib().nirrt_pn2_group_rows(feats.contiguous().data_ptr(), xyz.contiguous().data_ptr(), new_xyz.contiguous().data_ptr(),
                                       gidx.contiguous().data_ptr(), B, N, S, K, C, out.data_ptr(), _stream(feats)), "group_rows")
    return out


def fp_rows(feats1, feats2, dist, idx):
    """input rows of a feature-propagation level in one pass (k_fp_rows): (B * N, C1 + C2) = [feats1, inverse-distance
    interpolation of feats2 over the three neighbours (dist, idx)]; None when a width is not a multiple of 4"""
    B, N, _ = dist.shape
    S, C2 = feats2.shape[1], feats2.shape[2]
    C1 = 0 if feats1 is None else feats1.shape[2]
    if C1 % 4 or C2 % 4:
        return None
    _need_cuda("fp_rows", feats2)
    out = torch.empty(B * N, C1 + C2, dtype=torch.float32, device=feats2.device)
    f1 = feats1.contiguous() if feats1 is not None else None
    _check(_lib().nirrt_pn2_fp_rows(f1.data_ptr() if f1 is not None else None, feats2.contiguous().data_ptr(),
                                    dist.contiguous().data_ptr(), idx.contiguous().data_ptr(), B, N, S, C1, C2, out.data_ptr(),
                                    _stream(feats2)), "fp_rows")
    return out


def sa_mlp_pack(layers, c_in, device):
    """folded (W (C_out, C_in), b) triple of one set-abstraction branch -> what the fused kernel takes: W^T contiguous, the
    first with its input rows zero-padded to a multiple of 4; None when the widths do not fit its 16-column MFMA tiles"""
    if len(layers) != 3:
        return None
    widths = [w.shape[0] for w, _ in layers]
    if any(c % 16 for c in widths) or widths[2] > 128:
        return None
    cin_pad = (c_in + 3) // 4 * 4
    w1 = torch.zeros(cin_pad, widths[0], dtype=torch.float32, device=device)
    w1[:c_in] = layers[0][0].t()
    pack = [w1.contiguous(), layers[0][1].float().contiguous(), layers[1][0].t().float().contiguous(), layers[1][1].float().contiguous(),
            layers[2][0].t().float().contiguous(), layers[2][1].float().contiguous()]
    return {"t": pack, "cin_pad": cin_pad, "widths": widths}


def sa_mlp(feats, xyz, new_xyz, gidx, pack, out, out_off):
    """fused gather + 3 x (GEMM + bias + ReLU) + max over the K members of every group on MFMA tiles (k_sa_mlp): writes
    out[:, :, out_off : out_off + C3]; False when the level's weights do not fit the LDS (caller falls back to library GEMMs)"""
    B, N, C = feats.shape
    S, K = gidx.shape[1], gidx.shape[2]
    t = pack["t"]
    c1, c2, c3 = pack["widths"]
    rc = _lib().nirrt_pn2_sa_mlp(feats.contiguous().data_ptr(), xyz.contiguous().data_ptr(), new_xyz.contiguous().data_ptr(),
                                 gidx.contiguous().data_ptr(), B, N, S, K, C, pack["cin_pad"], t[0].data_ptr(), t[1].data_ptr(), c1,
                                 t[2].data_ptr(), t[3].data_ptr(), c2, t[4].data_ptr(), t[5].data_ptr(), c3, out.data_ptr(),
                                 out.shape[2], out_off, _stream(feats))
    if rc == -3:
        return False
    _check(rc, "sa_mlp")
    return True


def farthest_point_down_sample_f64(pts, num_samples, device_id=0):
    """float64 greedy max-min down-sampling of a host cloud (n, 3) -> bool mask (n,) of the survivors (k_fps_f64)"""
    import numpy as np
    from . import _hip
    n = len(pts)
    _need_device("farthest_point_down_sample_f64")
    sel8 = np.zeros(n, dtype=np.uint8)
    rc = _lib().nirrt_fps_f64(pts.ctypes.data, n, int(num_samples), sel8.ctypes.data, int(device_id))
    if rc != 0:
        raise _hip.NirrtError("nirrt_fps_f64 failed (%d)" % rc)
    return sel8.astype(bool)


def farthest_point_down_sample_f64_batch(clouds, num_samples, device_id=0):
    """several host clouds [(n_b, 3) f64] in ONE launch (one workgroup per cloud) -> list of bool masks; clouds that already
    have num_samples points or fewer are kept whole"""
    import numpy as np
    from . import _hip
    masks = [None] * len(clouds)
    todo = [b for b, c in enumerate(clouds) if len(c) > num_samples]
    for b, c in enumerate(clouds):
        if len(c) <= num_samples:
            masks[b] = np.ones(len(c), dtype=bool)
    if not todo:
        return masks
    _need_device("farthest_point_down_sample_f64_batch")
    pts = np.ascontiguousarray(np.concatenate([np.asarray(clouds[b], dtype=np.float64) for b in todo], axis=0))
    cnt = np.array([len(clouds[b]) for b in todo], dtype=np.int32)
    ns = np.full(len(todo), int(num_samples), dtype=np.int32)
    sel = np.zeros(len(pts), dtype=np.uint8)
    rc = _lib().nirrt_fps_f64_batch(pts.ctypes.data, len(todo), cnt.ctypes.data, ns.ctypes.data, sel.ctypes.data, int(device_id))
    if rc != 0:
        raise _hip.NirrtError("nirrt_fps_f64_batch failed (%d)" % rc)
    off = 0
    for b, c in zip(todo, cnt):
        masks[b] = sel[off:off + c].astype(bool)
        off += int(c)
    return masks


class CloudJob(C.Structure):
    """nirrt_cloud_job of include/nirrt_pointops.h"""
    _fields_ = [("words", C.c_void_p), ("free_tab", C.c_void_p), ("balls", C.c_void_p), ("boxes", C.c_void_p),
                ("mode", C.c_int32), ("w", C.c_int32), ("h", C.c_int32), ("n_ball", C.c_int32), ("n_box", C.c_int32), ("pad", C.c_int32),
                ("a", C.c_double * 20), ("clearance", C.c_double)]


def cloud_job_table(n):
    """n zeroed nirrt_cloud_job records as a numpy structured array (same layout as CloudJob): a batch's jobs are filled column by
    column instead of one ctypes structure at a time (a refresh of hundreds of clouds spent longer filling structures than the
    device spent making the clouds)"""
    import numpy as np
    dt = np.dtype([("words", "<u8"), ("free_tab", "<u8"), ("balls", "<u8"), ("boxes", "<u8"), ("mode", "<i4"), ("w", "<i4"), ("h", "<i4"),
                   ("n_ball", "<i4"), ("n_box", "<i4"), ("pad", "<i4"), ("a", "<f8", (20,)), ("clearance", "<f8")], align=False)
    assert dt.itemsize == C.sizeof(CloudJob)
    return np.zeros(n, dtype=dt)


def guidance_clouds(jobs, n_raw, n_points, clouds, device_id=0):
    """candidates -> down-sampling -> compaction of len(jobs) guidance clouds on the device (nirrt_guidance_clouds): `jobs` a
    list of CloudJob or a cloud_job_table, `clouds` a cuda float64 tensor (len(jobs), n_points, 3) that receives them ->
    (n_cand, n_out) int32 arrays"""
    import numpy as np
    n = len(jobs)
    if isinstance(jobs, np.ndarray):
        assert jobs.dtype.itemsize == C.sizeof(CloudJob) and jobs.flags.c_contiguous
        arr = C.c_void_p(jobs.ctypes.data)
    else:
        arr = (CloudJob * n)(*jobs)
    n_cand = np.zeros(n, dtype=np.int32)
    n_out = np.zeros(n, dtype=np.int32)
    assert clouds.is_cuda and clouds.dtype == torch.float64 and clouds.is_contiguous() and tuple(clouds.shape) == (n, n_points, 3)
    torch.cuda.current_stream(clouds.device).synchronize()
    _check(_lib().nirrt_guidance_clouds(C.cast(arr, C.c_void_p), n, int(n_raw), int(n_points), clouds.data_ptr(), n_cand.ctypes.data,
                                        n_out.ctypes.data, int(device_id)), "guidance_clouds")
    return n_cand, n_out
