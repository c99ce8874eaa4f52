"""Neural-connect helpers (reference: wrapper/utils/bfs_connect_heuristic.py): BFS connectivity of the
predicted path points, boundary detection and the heuristic choice of the next seed point.
Same names, inputs and outputs as the reference so PNGWrapper.generate_connected_path_points
(wrapper/pointnet_pointnet2/pointnet2_wrapper_connect_bfs.py:76-240) reads the same."""
from collections import deque

import numpy as np


def get_boundary_mask(pc, path_mask, unvisited_mask, boundary_distance_threshold):
    """visited path points that have an unvisited point strictly within the threshold (:5-29)"""
    on = np.where(path_mask.astype(bool))[0]
    un = pc[unvisited_mask.astype(bool)]
    out = np.zeros(len(pc), dtype=np.float32)
    if len(on) and len(un):
        d = np.linalg.norm(pc[on][:, np.newaxis] - un, axis=2)
        out[on[(d < boundary_distance_threshold).any(axis=1)]] = 1
    return out


def bfs_point_cloud_visualization(pc, path_mask, x_start, x_goal, step_len):
    """BFS from start (vertex 0) to goal (vertex 1) over {start, goal, predicted points}, edge iff the
    float32 distance is < step_len (:80-139) -> (has_path, path_line | None, visited_mask f32)."""
    on = np.where(path_mask.astype(bool))[0]
    verts = np.concatenate([x_start[np.newaxis], x_goal[np.newaxis], pc[on]], axis=0)
    adj = np.linalg.norm(verts[:, np.newaxis] - verts, axis=2) < step_len
    visited = set([0])      # a real python set: the reference drops "the first element" of list(visited)
    queue = deque([0])
    parents = {}
    has_path, path_line = False, None
    while queue and not has_path:
        v = queue.popleft()
        for nb in np.where(adj[v])[0]:
            if nb == 1:
                has_path = True
                chain = [1, v]
                while chain[-1] != 0:
                    chain.append(parents[chain[-1]])
                chain.reverse()
                path_line = verts[np.array(chain)]
                break
            if nb not in visited:
                queue.append(nb)
                visited.add(nb)
                parents[nb] = v
    vis = np.array(list(visited)[1:]).astype(int) - 2
    mask = np.zeros(len(pc), dtype=np.float32)
    mask[on[vis]] = 1
    return has_path, path_line, mask


def select_heuristic_boundary_point(pc, boundary_mask, x_start, x_goal, cost_from_start_rank_weight=1):
    """boundary point maximising -(rank of g+h, ascending) - w * (rank of g, descending) (:142-181)"""
    idx = np.where(boundary_mask.astype(bool))[0]
    if len(idx) == 0:
        return None, None, None
    b = pc[idx]
    g = np.linalg.norm(b - x_start, axis=1)
    h = np.linalg.norm(b - x_goal, axis=1)
    total_rank = np.empty(len(b), dtype=int)
    total_rank[np.argsort(g + h)] = np.arange(len(b))
    g_rank = np.empty(len(b), dtype=int)
    g_rank[np.flip(np.argsort(g))] = np.arange(len(b))
    heur = [-(int(total_rank[i]) + cost_from_start_rank_weight * int(g_rank[i])) for i in range(len(b))]
    best = idx[int(np.argmax(heur))]
    return best, pc[best], heur
