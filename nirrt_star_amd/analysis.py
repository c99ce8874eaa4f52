"""The reference's evaluation metrics on result pickles (result_analysis_random_world_2d.py:35-79, _3d twin), without the
plots: first-solution iteration per problem, and the mean path-cost ratio curve - best cost at +0, +250, ... +3000
iterations after a method's first solution, divided by RRT*'s initial path cost on the same problem.

    python -m nirrt_star_amd.analysis --dim 2d --random_dataset_len 500      # prints one JSON object

reads results/evaluation/<dim>/<problem>-<planner>[-c-bfs]-<net>-<n>.pickle as written by the reference's eval_planning_*.py
or by nirrt_star_amd.eval_sharded.
"""
import argparse
import json
import pickle
from os.path import exists, join

import numpy as np

ITER_AFTER_INITIAL = list(range(0, 3000 + 250, 250))

METHOD_FILES = {   # method key -> file stem (result_analysis_random_world_2d.py:15-24)
    'rrt': '{p}-rrt_star-none',
    'irrt': '{p}-irrt_star-none',
    'nrrt_png': '{p}-nrrt_star-pointnet2',
    'nrrt_gng': '{p}-nrrt_star-unet',
    'nrrt_png_c': '{p}-nrrt_star-c-bfs-pointnet2',
    'nirrt_png': '{p}-nirrt_star-pointnet2',
    'nirrt_png_c': '{p}-nirrt_star-c-bfs-pointnet2',
}


def initial_index(result):
    """index of the first finite entry of a planning_random list (the iteration that found the first solution)"""
    idx = np.where(np.asarray(result, dtype=np.float64) < np.inf)[0]
    if len(idx) == 0:
        raise ValueError("no solution in this result list")
    return int(idx[0])


def first_solution_iterations(results):
    return [initial_index(r['result']) for r in results]


def path_cost_ratios(results, results_rrt, checkpoints=ITER_AFTER_INITIAL):
    """{checkpoint: [ratio per problem]}: cost `checkpoint` iterations after the method's own first solution (the last
    entry if the list is shorter) over RRT*'s first-solution cost on the same problem"""
    out = {c: [] for c in checkpoints}
    for r, rr in zip(results, results_rrt):
        res, ref = r['result'], rr['result']
        i0 = initial_index(res)
        base = ref[initial_index(ref)]
        for c in checkpoints:
            out[c].append((res[i0 + c] if i0 + c < len(res) else res[-1]) / base)
    return out


def path_cost_mean(results, results_rrt, checkpoints=ITER_AFTER_INITIAL):
    ratios = path_cost_ratios(results, results_rrt, checkpoints)
    return [float(np.mean(ratios[c])) for c in checkpoints]


def load_results(folder, problem, n, methods=None):
    out = {}
    for m, stem in METHOD_FILES.items():
        if methods and m not in methods:
            continue
        path = join(folder, stem.format(p=problem) + '-%d.pickle' % n)
        if exists(path):
            with open(path, 'rb') as f:
                out[m] = pickle.load(f)
    return out


def analyse(random_results, n=None):
    if 'rrt' not in random_results:
        raise ValueError("the RRT* results are the normaliser of every curve: rrt_star pickle missing")
    out = {}
    for m, res in random_results.items():
        res = res[:n] if n else res
        rrt = random_results['rrt'][:len(res)]
        out[m] = {'path_cost_mean': path_cost_mean(res, rrt), 'first_solution_iterations': first_solution_iterations(res)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dim', default='2d', choices=['2d', '3d'])
    ap.add_argument('--random_dataset_len', type=int, default=500)
    args = ap.parse_args()
    res = load_results(join('results', 'evaluation', args.dim), 'random_' + args.dim, args.random_dataset_len)
    a = analyse(res, args.random_dataset_len)
    print(json.dumps({'iter_after_initial': ITER_AFTER_INITIAL,
                      'methods': {m: {'path_cost_mean': v['path_cost_mean'],
                                      'median_first_solution_iter': float(np.median(v['first_solution_iterations']))}
                                  for m, v in a.items()}}))


if __name__ == '__main__':
    main()
