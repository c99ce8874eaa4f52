"""Synthetic planning_random result lists for the analysis-metric fixture: the same seeded recipe feeds the reference's
script (make_golden.py, this container) and the test of nirrt_star_amd.analysis, so only the expected outputs are stored."""
import numpy as np

N_PROBLEMS = 12
STEMS = ['rrt_star-none', 'irrt_star-none', 'nrrt_star-pointnet2', 'nrrt_star-unet', 'nrrt_star-c-bfs-pointnet2',
         'nirrt_star-pointnet2', 'nirrt_star-c-bfs-pointnet2']


def make_inputs():
    rng = np.random.default_rng(21)
    data = {}
    for m in STEMS:
        lst = []
        for i in range(N_PROBLEMS):
            first = int(rng.integers(5, 400))
            length = first + int(rng.integers(200, 3400))          # some lists end before +3000
            cost = 300.0 - np.cumsum(rng.uniform(0, 0.05, size=length - first))
            lst.append({'img_idx': i, 'result': [float('inf')] * first + [float(c) for c in cost]})
        data[m] = lst
    return data
