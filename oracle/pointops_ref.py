"""CPU references of the point operators (TEST INFRASTRUCTURE ONLY - see oracle/oracle.py).

The product package (nirrt_star_amd.pointops / pointcloud) has ONE implementation of these operators, the HIP kernels
of csrc/pointops.hip, and raises on CPU tensors; it offers no hook for another one.  The CPU test-suite and the fixture
generator (tests/golden/make_golden.py) replace the package's module attributes from the OUTSIDE with `patched(pointops)`
below (tests/conftest.py: for the whole session on a host without a GPU; on a GPU box only around the CPU calibration of the
synthetic checkpoint, so a CPU tensor that reaches the operators inside a `-m gpu` test raises).

Restated from the reference, plain torch ops in fp32 / numpy in fp64:
  square_distance          pointnet_pointnet2/models/pointnet2_utils.py:21-42
  farthest_point_sample    :65-86   (start index injected by the caller)
  ball_query               :89-109  (query_ball_point: first K in ascending index order, padded with the first)
  three_nn                 :295-299 (sort of the distance matrix, first three)
  farthest_point_down_sample_f64   open3d 0.17 PointCloud.farthest_point_down_sample as recollected in SURVEY.md §8c
                           (start at point 0, greedy max-min squared distance, survivors in original order) - unpinned
"""
import numpy as np
import torch


def square_distance(src, dst):
    d = -2 * torch.matmul(src, dst.permute(0, 2, 1))
    d += torch.sum(src ** 2, -1)[:, :, None]
    d += torch.sum(dst ** 2, -1)[:, None, :]
    return d


def farthest_point_sample(xyz, npoint, start):
    B, N, _ = xyz.shape
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
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    idx = torch.arange(N, dtype=torch.long).view(1, 1, N).repeat(B, S, 1)
    idx[square_distance(new_xyz, xyz) > radius ** 2] = N
    idx = idx.sort(dim=-1)[0][:, :, :nsample]
    first = idx[:, :, 0:1].expand(-1, -1, nsample)
    return torch.where(idx == N, first, idx)


def three_nn(xyz1, xyz2):
    d, i = square_distance(xyz1, xyz2).sort(dim=-1)
    return d[:, :, :3], i[:, :, :3]


def farthest_point_down_sample_f64(pts, num_samples):
    """pts (n, 3) float64 -> bool mask (n,) of the survivors"""
    n = len(pts)
    sel = np.zeros(n, dtype=bool)
    dist = np.full(n, np.inf)
    far = 0
    for _ in range(num_samples):
        sel[far] = True
        d = ((pts - pts[far]) ** 2).sum(axis=1)
        np.minimum(dist, d, out=dist)
        far = int(np.argmax(dist))
    return sel


def farthest_point_down_sample_f64_batch(clouds, num_samples, device_id=0):
    masks = []
    for c in clouds:
        c = np.ascontiguousarray(c, dtype=np.float64)
        masks.append(np.ones(len(c), dtype=bool) if len(c) <= num_samples else farthest_point_down_sample_f64(c, num_samples))
    return masks


def cpu_operators():
    """name -> CPU stand-in with the signature of the nirrt_star_amd.pointops function of that name"""
    def fps(xyz, npoint, start=None, n_valid=None):
        if n_valid is not None:      # ragged batch: every cloud on its own first n_valid[b] points (what the reference does: one cloud per call)
            return torch.stack([farthest_point_sample(xyz[b:b + 1, : int(n_valid[b])], npoint, start[b:b + 1])[0] for b in range(xyz.shape[0])])
        if start is None:
            start = torch.randint(0, xyz.shape[1], (xyz.shape[0],), dtype=torch.long)
        return farthest_point_sample(xyz, npoint, start)

    def bq(radius, nsample, xyz, new_xyz, n_valid=None):
        if n_valid is not None:
            return torch.stack([ball_query(radius, nsample, xyz[b:b + 1, : int(n_valid[b])], new_xyz[b:b + 1])[0] for b in range(xyz.shape[0])])
        return ball_query(radius, nsample, xyz, new_xyz)
    return {"farthest_point_sample": fps, "ball_query": bq, "three_nn": three_nn,
            "farthest_point_down_sample_f64": lambda pts, num_samples, device_id=0: farthest_point_down_sample_f64(pts, num_samples),
            "farthest_point_down_sample_f64_batch": farthest_point_down_sample_f64_batch}


class patched:
    """`with patched(nirrt_star_amd.pointops):` - the module's operators are the CPU stand-ins inside the block.
    `patched(module).start()` leaves them in place (CPU-only test sessions, fixture generation)."""

    def __init__(self, module):
        self.module = module
        self.saved = None

    def start(self):
        if self.saved is None:
            ops = cpu_operators()
            self.saved = {k: getattr(self.module, k) for k in ops}
            for k, f in ops.items():
                setattr(self.module, k, f)
        return self

    def stop(self):
        if self.saved is not None:
            for k, f in self.saved.items():
                setattr(self.module, k, f)
            self.saved = None

    __enter__ = start

    def __exit__(self, *exc):
        self.stop()
