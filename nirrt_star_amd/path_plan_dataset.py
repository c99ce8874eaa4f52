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
    """`dataset_filepath`: 'data/random_2d/' + mode + '.npz' (mode = train / val / test).  __getitem__ returns
    (pc_xyz_raw (n, 3), pc_xyz normalised (n, 3), pc_features (n, 3) = start / goal / free, pc_labels (n,), token)."""

    def __init__(self, dataset_filepath):
        data = np.load(dataset_filepath)
        self.pc = data['pc'].astype(np.float32)
        self.start_mask = data['start'].astype(np.float32)
        self.goal_mask = data['goal'].astype(np.float32)
        self.free_mask = data['free'].astype(np.float32)
        self.astar_mask = data['astar'].astype(np.float32)
        self.token = data['token']
        if self.pc.shape[2] == 2:   # 2D clouds get a zero z column
            self.pc = np.concatenate((self.pc, np.zeros((self.pc.shape[0], self.pc.shape[1], 1)).astype(np.float32)), axis=2)
        if self.pc.shape[2] != 3:
            raise RuntimeError("Point cloud is not 3D.")
        labelweights, _ = np.histogram(self.astar_mask, range(3))
        labelweights = labelweights.astype(np.float32)
        labelweights = labelweights / np.sum(labelweights)
        self.labelweights = np.power(np.amax(labelweights) / labelweights, 1 / 3.0)
        print(self.labelweights)

    def __len__(self):
        return len(self.pc)

    def __getitem__(self, index):
        pc_xyz_raw = self.pc[index]
        pc_xyz = pc_normalize(pc_xyz_raw)
        pc_features = np.stack((self.start_mask[index], self.goal_mask[index], self.free_mask[index]), axis=-1)
        return pc_xyz_raw, pc_xyz, pc_features, self.astar_mask[index], self.token[index]
