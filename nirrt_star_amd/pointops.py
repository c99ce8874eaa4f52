"""Point operators of the PointNet++ guidance net.

GPU tensors -> hand-written HIP kernels in libnirrt_hip.so (csrc/pointops.hip) on the current torch
stream; the library being absent is an error (no silent fallback on a GPU).  CPU tensors (explicit
device='cpu', used by the CPU test-suite and fixture generation) -> the same semantics in torch ops.
"""
import ctypes as C

import torch


def _lib():
    from . import _hip
    L = _hip.load()
    if not hasattr(L, "_pn2_ready"):
        vp = C.c_void_p
        L.nirrt_pn2_fps.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp]
        L.nirrt_pn2_ball_query.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, vp, vp]
        L.nirrt_pn2_three_nn.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp]
        for f in (L.nirrt_pn2_fps, L.nirrt_pn2_ball_query, L.nirrt_pn2_three_nn):
            f.restype = C.c_int
        L._pn2_ready = True
    return L


def _stream(t):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _check(rc, what):
    if rc != 0:
        raise RuntimeError("libnirrt_hip %s failed (%d)" % (what, rc))


def square_distance(src, dst):
    """(B, N, 3), (B, M, 3) -> (B, N, M): -2 src.dst^T + |src|^2 + |dst|^2 in that order (pointnet2_utils.py:21-42)"""
    d = -2 * torch.matmul(src, dst.permute(0, 2, 1))
    d += torch.sum(src ** 2, -1)[:, :, None]
    d += torch.sum(dst ** 2, -1)[:, None, :]
    return d


def farthest_point_sample(xyz, npoint, start=None):
    """xyz (B, N, 3) -> indices (B, npoint) long; start (B,) long or drawn with torch.randint on the CPU generator"""
    B, N, _ = xyz.shape
    if start is None:
        start = torch.randint(0, N, (B,), dtype=torch.long)
    start = start.to(xyz.device)
    if xyz.is_cuda:
        xyz = xyz.contiguous().float()
        out = torch.empty(B, npoint, dtype=torch.long, device=xyz.device)
        _check(_lib().nirrt_pn2_fps(xyz.data_ptr(), B, N, npoint, start.contiguous().data_ptr(), out.data_ptr(), _stream(xyz)), "fps")
        return out
    cent = torch.zeros(B, npoint, dtype=torch.long)
    dist = torch.ones(B, N) * 1e10
    far = start.clone()
    ar = torch.arange(B)
    for i in range(npoint):
        cent[:, i] = far
        c = xyz[ar, far, :].view(B, 1, 3)
        d = torch.sum((xyz - c) ** 2, -1)
        dist = torch.minimum(dist, d)
        far = torch.max(dist, -1)[1]
    return cent


def ball_query(radius, nsample, xyz, new_xyz):
    """first `nsample` indices in ascending order within `radius` of each query, padded with the first -> (B, S, K) long"""
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    if xyz.is_cuda:
        out = torch.empty(B, S, nsample, dtype=torch.long, device=xyz.device)
        r2 = float(torch.tensor(radius ** 2, dtype=torch.float32))
        _check(_lib().nirrt_pn2_ball_query(xyz.contiguous().data_ptr(), new_xyz.contiguous().data_ptr(), B, N, S, nsample,
                                           C.c_float(r2), out.data_ptr(), _stream(xyz)), "ball_query")
        return out
    idx = torch.arange(N, dtype=torch.long).view(1, 1, N).repeat(B, S, 1)
    idx[square_distance(new_xyz, xyz) > radius ** 2] = N
    idx = idx.sort(dim=-1)[0][:, :, :nsample]
    first = idx[:, :, 0:1].expand(-1, -1, nsample)
    return torch.where(idx == N, first, idx)


def three_nn(xyz1, xyz2):
    """3 nearest coarse points per fine point -> (squared distances (B, N, 3), indices (B, N, 3))"""
    B, N, _ = xyz1.shape
    S = xyz2.shape[1]
    if xyz1.is_cuda:
        d = torch.empty(B, N, 3, dtype=torch.float32, device=xyz1.device)
        i = torch.empty(B, N, 3, dtype=torch.long, device=xyz1.device)
        _check(_lib().nirrt_pn2_three_nn(xyz1.contiguous().data_ptr(), xyz2.contiguous().data_ptr(), B, N, S, d.data_ptr(),
                                         i.data_ptr(), _stream(xyz1)), "three_nn")
        return d, i
    d, i = square_distance(xyz1, xyz2).sort(dim=-1)
    return d[:, :, :3], i[:, :, :3]
