"""Training-set path of the guidance network: the `.npz` schema of generate_random_world_env_{2d,3d}_point_cloud.py:47-94
(`pc (N, n_points, D) f32, start / goal / free / astar (N, n_points) f32, token (N,) str`) and the Dataset the reference
trains and evaluates PointNet++ with (pointnet_pointnet2/PathPlanDataLoader.py:7-52), so that real ICRA-24 data drops in.
Pure numpy on the host (torch only for the Dataset base class): not part of the GPU hot path."""
import numpy as np

from . import pointcloud as pcu
from .pointnet2 import pc_normalize

try:   # the base class only matters for torch DataLoader
    from torch.utils.data import Dataset as _Base
except Exception:   # pragma: no cover
    _Base = object

KEYS = ("token", "pc", "start", "goal", "free", "astar")


def make_sample(pc, start_point, goal_point, path, start_radius, goal_radius, path_radius, token):
    """one row of the schema from a point cloud (n_points, D), the start / goal points (D,) and an optimal path (m, D):
    masks of the cloud points around start, goal and path; free = neither around start nor around goal"""
    pc = np.asarray(pc)
    around_start = pcu.get_point_cloud_mask_around_points(pc, np.asarray(start_point)[np.newaxis, :], neighbor_radius=start_radius)
    around_goal = pcu.get_point_cloud_mask_around_points(pc, np.asarray(goal_point)[np.newaxis, :], neighbor_radius=goal_radius)
    around_path = pcu.get_point_cloud_mask_around_points(pc, np.asarray(path), neighbor_radius=path_radius)
    free = (1 - around_start) * (1 - around_goal)
    return {"token": token, "pc": pc.astype(np.float32), "start": around_start.astype(np.float32),
            "goal": around_goal.astype(np.float32), "free": free.astype(np.float32), "astar": around_path.astype(np.float32)}


def save_dataset(path, samples):
    """np.savez with the reference's keys (lists of per-sample arrays, like save_raw_dataset :20-28)"""
    cols = {k: [s[k] for s in samples] for k in KEYS}
    np.savez(path, **cols)


class PathPlanDataset(_Base):
    """Indexable view of one split file ('data/random_2d/' + mode + '.npz', mode = train / val / test) with the item layout
    the reference's training and evaluation loops unpack (PathPlanDataLoader.py:7-52):
        (raw xyz (n, 3), normalised xyz (n, 3), features (n, 3) = start / goal / free, labels (n,), token)
    and `.labelweights` = (max class frequency / class frequency) ** (1/3) over the two label classes.
    Everything is float32; planar clouds get a zero z column when the file is opened."""

    _MASKS = (("start_mask", "start"), ("goal_mask", "goal"), ("free_mask", "free"), ("astar_mask", "astar"))

    def __init__(self, dataset_filepath):
        with np.load(dataset_filepath) as z:
            cloud = np.asarray(z["pc"], dtype=np.float32)
            for attr, key in self._MASKS:
                setattr(self, attr, np.asarray(z[key], dtype=np.float32))
            self.token = z["token"]
        width = cloud.shape[-1]
        if width not in (2, 3):
            raise RuntimeError("point clouds must be (count, n_points, 2 or 3), got last axis %d" % width)
        if width == 2:
            cloud = np.pad(cloud, ((0, 0), (0, 0), (0, 1)))
        self.pc = cloud
        # class balance of the path labels: bins [0, 1) and [1, 2]
        freq = np.histogram(self.astar_mask, bins=(0, 1, 2))[0].astype(np.float32)
        freq /= freq.sum()
        self.labelweights = np.power(freq.max() / freq, 1 / 3.0)

    def __len__(self):
        return self.pc.shape[0]

    def __getitem__(self, index):
        raw = self.pc[index]
        channels = [getattr(self, attr)[index] for attr, _ in self._MASKS[:3]]
        return raw, pc_normalize(raw), np.stack(channels, axis=-1), self.astar_mask[index], self.token[index]
