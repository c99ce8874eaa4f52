"""Evaluation of a set of independent planning problems sharded over GPUs (the
reference's single-process counterpart is eval_planning_2d.py:83-136).

Problems (env x start/goal pairs) share nothing, so the partitioning is static round-robin
`problem i -> rank i mod world_size`, one process per GPU, no collective on the data path; each rank
plans its problems in batches (one persistent launch per batch, one workgroup per problem) with the
`planning_random(iter_after_initial)` protocol, and ONE gather of fixed-size result records
(RCCL over xGMI when the ranks own GPUs; gloo in the CPU tests) ends the run.

    torchrun --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -m nirrt_star_amd.eval_sharded \
        --problem random_2d --planner irrt_star --iter_after_initial 3000
"""
import argparse
import json
import os
import random
import time

import numpy as np

CHECKPOINTS = list(range(0, 3001, 250))   # cost recorded at +0, +250, ... +3000 iterations after the first solution
RECORD_LEN = 4 + len(CHECKPOINTS)         # problem_id, first_solution_iter, n_vertices, total_iters, costs...


def shard_indices(n_problems, rank, world_size, heavy=None, first=0):
    """problem ids [first, n_problems) of rank `rank`.  Without a predictor: round-robin `i -> i mod world_size`.  With `heavy`
    (one bool per problem id, the same on every rank - e.g. straight_segment_free): the heavy problems are dealt round-robin
    first, then the light ones continue the deal where the heavy ones stopped, so that every rank gets the same number of each
    (+- 1) and the same number in total (+- 1).  Every rank computes the same partition from the same inputs; no collective."""
    ids = list(range(int(first), int(n_problems)))
    if heavy is None:
        return ids[rank::world_size]
    deal = [i for i in ids if heavy[i]] + [i for i in ids if not heavy[i]]
    return sorted(deal[rank::world_size])


def straight_segment_free(env_dict, clearance, x_start=None, x_goal=None, spacing=0.5):
    """Host-side PREDICTOR of a slow problem (scheduling only - no planning result depends on it): is the straight start-goal
    segment free of the clearance-inflated obstacles?  Such problems end up with an informed set collapsed onto the segment and
    Near sets of thousands of members (2-3x the median run time in 2D, 5x+ in 3D).  Sampled every `spacing` units against
    circles / rectangles (2D) or balls / boxes (3D)."""
    a = np.asarray(x_start if x_start is not None else env_dict["start"][0], dtype=np.float64)
    b = np.asarray(x_goal if x_goal is not None else env_dict["goal"][0], dtype=np.float64)
    n = max(2, int(np.ceil(np.linalg.norm(b - a) / spacing)) + 1)
    pts = a[None, :] + np.linspace(0.0, 1.0, n)[:, None] * (b - a)[None, :]
    d = pts.shape[1]
    round_obs = env_dict.get("circle_obstacles" if d == 2 else "ball_obstacles", [])
    box_obs = env_dict.get("rectangle_obstacles" if d == 2 else "box_obstacles", [])
    for o in round_obs:
        c, r = np.asarray(o[:d], dtype=np.float64), float(o[d])
        if (np.sum((pts - c) ** 2, axis=1) <= (r + clearance) ** 2).any():
            return False
    for o in box_obs:
        lo = np.asarray(o[:d], dtype=np.float64) - clearance
        hi = np.asarray(o[:d], dtype=np.float64) + np.asarray(o[d:2 * d], dtype=np.float64) + clearance
        if np.all((pts >= lo) & (pts <= hi), axis=1).any():
            return False
    return True


def gather_records(local, world_size, rank, device="cpu"):
    """local: (m, RECORD_LEN) float64 -> rank 0 gets every rank's records sorted by problem id (None elsewhere).
    Shards differ in length by at most one, so each is padded to the common maximum with id = -1 rows."""
    import torch
    import torch.distributed as dist
    local = np.asarray(local, dtype=np.float64).reshape(-1, RECORD_LEN)
    if not (dist.is_available() and dist.is_initialized()):      # no launcher, one rank: nothing to gather from
        if world_size != 1:
            raise RuntimeError("gather_records: %d ranks but no process group" % world_size)
        return local[np.argsort(local[:, 0])]
    # (a process group of ONE rank still goes through the collectives: `torchrun --nproc-per-node 1` is how the RCCL path is
    #  exercised on a one-GPU box - RCCL refuses two ranks on one device)
    m = torch.tensor([len(local)], dtype=torch.int64, device=device)
    mx = m.clone()
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    pad = np.full((int(mx.item()), RECORD_LEN), -1.0)
    pad[: len(local)] = local
    t = torch.from_numpy(pad).to(device)
    out = [torch.empty_like(t) for _ in range(world_size)] if rank == 0 else None
    dist.gather(t, out, dst=0)
    if rank != 0:
        return None
    allr = torch.cat(out).cpu().numpy()
    allr = allr[allr[:, 0] >= 0]
    return allr[np.argsort(allr[:, 0])]


def make_record(pid, trace, n_vertices):
    """trace: best cost after each iteration (inf before the first solution)"""
    rec = np.full(RECORD_LEN, np.inf)
    rec[0] = pid
    fin = np.isfinite(trace)
    rec[1] = int(np.argmax(fin)) + 1 if fin.any() else -1
    rec[2] = n_vertices
    rec[3] = len(trace)
    if fin.any():
        k0 = int(rec[1]) - 1
        for j, c in enumerate(CHECKPOINTS):
            if k0 + c < len(trace):
                rec[4 + j] = trace[k0 + c]
    return rec


PLANNERS = ("rrt_star", "irrt_star", "nirrt_star", "nirrt_star_c", "nrrt_star", "nrrt_star_c")


def make_wrapper(args, dim, device):
    """PNGWrapper / PNGWrapper3D on `device` (the reference's `-n pointnet2`); without trained weights on disk a seeded
    synthetic checkpoint in the reference's format is written first (there is no network access to fetch real ones)"""
    import os
    from . import png_wrapper as W
    path = W.checkpoint_path(args.root_dir, dim)
    if not os.path.exists(path):
        W.make_synthetic_checkpoint(path, seed=0, dim=dim, device=device)
    return (W.PNGWrapper if dim == 2 else W.PNGWrapper3D)(root_dir=args.root_dir, device=device)


def _batch_setup(problems, pids, args, device_id, cap, wrapper):
    from . import _hip, batch, sampling
    dim = 3 if getattr(args, "problem", "random_2d") == "random_3d" else 2
    planner = args.planner
    informed = planner in ("irrt_star", "nirrt_star", "nirrt_star_c")
    flags = _hip.F_IRRT if informed else _hip.F_GOAL_SCAN
    # the whole batch in one creation call (one device pass), one launch for the informed-sampling frames, one for the probes
    trees = _hip.create_trees(dim, cap, [(pr["x_start"], pr["x_goal"], args.step_len, pr["search_radius"], args.clearance, pr["env"])
                                         for pr in problems], device_id=device_id)
    frames = [sampling.informed_frame(pr["x_start"], pr["x_goal"]) for pr in problems]
    _hip.set_informed_batch(trees, frames)
    streams = [batch.ProblemStreams(1000 + pid) for pid in pids]
    # dispatch order inside the persistent launches = longest first by the one predictor that is known before planning: a free
    # straight start-goal segment (informed set collapsed onto it, Near sets of thousands of members: 2-3x the median run time)
    if informed and len(trees) > 1:
        free = ~_hip.collision_each(trees, [np.stack([np.asarray(pr["x_start"], dtype=np.float64), np.asarray(pr["x_goal"], dtype=np.float64)])
                                            for pr in problems])
        order = sorted(range(len(trees)), key=lambda i: (not free[i], i))
        trees, streams, frames = [trees[i] for i in order], [streams[i] for i in order], [frames[i] for i in order]
    else:
        order = list(range(len(trees)))
    guidance = None
    if planner.startswith("n"):
        if wrapper is None:
            raise ValueError("planner %s needs a PointNet++ wrapper (-n pointnet2)" % planner)
        guidance = batch.Guidance(wrapper, dim, args.step_len, args.pc_n_points, args.pc_over_sample_scale, args.pc_sample_rate,
                                  args.pc_update_cost_ratio, connect=planner.endswith("_c"),
                                  connect_max_trial_attempts=args.connect_max_trial_attempts, informed=informed, device_id=device_id)
    return dim, flags, trees, streams, frames, guidance, order


def _raise_failures(res, pids):
    if res["failed"]:
        raise RuntimeError("planning stopped abnormally: " + "; ".join("problem %d: %s" % (pids[i], m) for i, m in sorted(res["failed"].items())))


def plan_batch(problems, pids, args, device_id, wrapper=None):
    """planning_random for a batch of problems: persistent launches until every tree has its first solution (or spent
    iter_max iterations), then iter_after_initial more for the solved ones.  Each problem uses its own seeded generators
    (1000 + problem id), resident in its tree; guidance clouds are refreshed between launches (nirrt_star_amd/batch.py).  Inside the
    launches the problems with a free start-goal segment are dispatched first; records and traces come back in the caller's order."""
    from . import batch
    cap = args.iter_max + args.iter_after_initial
    dim, flags, trees, streams, frames, guidance, order = _batch_setup(problems, pids, args, device_id, cap, wrapper)
    problems, pids = [problems[i] for i in order], [pids[i] for i in order]      # dispatch order from here on; undone at the end
    # phase 1 (until the first solution) in launches of 4096 iterations: the typical problem is solved within a few hundred, and a
    # launch's cost traces are (trees x launch length) doubles on both sides of the bus - 400 MB for 1000 problems at iter_max 50000
    r1 = batch.run_batch(trees, streams, args.iter_max, flags, dim, problems, guidance, frames, want_trace=True, stop_first=True,
                         window=int(os.environ.get("NIRRT_EVAL_FIRST_WINDOW", "4096")))
    _raise_failures(r1, pids)
    traces = list(r1["traces"])
    solved = [i for i in range(len(trees)) if len(traces[i]) and np.isfinite(traces[i][-1])]
    if solved and args.iter_after_initial > 0:
        r2 = batch.run_batch([trees[i] for i in solved], [streams[i] for i in solved], args.iter_after_initial, flags, dim,
                             [problems[i] for i in solved], guidance, [frames[i] for i in solved], want_trace=True, init_clouds=False)
        _raise_failures(r2, [pids[i] for i in solved])
        for j, i in enumerate(solved):
            traces[i] = np.concatenate([traces[i], r2["traces"][j]])
    recs = [make_record(pid, tr, t.n) for pid, tr, t in zip(pids, traces, trees)]
    batch.release_all(trees, streams)      # (the generators come home in one call, not tree by tree inside close())
    for t in trees:
        t.close()
    back = np.argsort(order)      # results in the caller's order
    return [recs[j] for j in back], [traces[j] for j in back]


def result_lists(kind, traces, thresholds=None):
    """The `path_len_list` the reference's planner methods return (eval_planning_2d.py:115-126): planning_random ->
    best path length after every iteration (inf before the first solution); planning_block_gap -> the same list cut
    right after the first entry below the threshold."""
    out = []
    for i, tr in enumerate(traces):
        tr = [float(v) for v in tr]
        if kind in ("block", "gap"):
            stop = first_below(tr, thresholds[i])
            if stop > 0:
                tr = tr[:stop]
        out.append(tr)
    return out


def gather_results(local, world_size, rank):
    """local: list of (problem id, path_len_list) -> rank 0 gets all of them sorted by id (variable-length lists travel
    as pickled objects: one gather at the end of the run, off the data path)"""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return sorted(local)
    out = [None] * world_size if rank == 0 else None
    dist.gather_object(local, out, dst=0)
    if rank != 0:
        return None
    return sorted(x for part in out for x in part)


def gather_rank_seconds(seconds, n_problems, world_size, rank):
    """(seconds spent planning, problems planned) of every rank -> rank 0 (how even the partition was)"""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return [(seconds, n_problems)]
    out = [None] * world_size if rank == 0 else None
    dist.gather_object((float(seconds), int(n_problems)), out, dst=0)
    return out if rank == 0 else [(seconds, n_problems)]


def write_reference_pickle(path, env_configs, results):
    """eval_planning_2d.py:100-136 wire format: a pickled list of copies of the env config dicts, each with the
    planner's path_len_list under 'result' - what result_analysis_*.py reads."""
    import pickle
    from copy import copy
    lst = []
    for pid, res in results:
        d = copy(env_configs[pid])
        d['result'] = res
        lst.append(d)
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    with open(path, "wb") as f:
        pickle.dump(lst, f)
    return lst


FINGERPRINT_KEYS = ("problem", "planner", "neural_net", "iter_max", "iter_after_initial", "step_len", "clearance", "pc_n_points",
                    "pc_over_sample_scale", "pc_sample_rate", "pc_update_cost_ratio", "connect_max_trial_attempts",
                    "path_len_threshold_percentage")


def run_fingerprint(args, n_problems, wrapper_id=None):
    """what a result list depends on besides the problem itself: a result file is only resumed by a run with the same values (the
    reference's file name holds problem / planner / network / count only - a rerun with another iter_max or clearance would silently
    mix two experiments)"""
    fp = {k: getattr(args, k, None) for k in FINGERPRINT_KEYS}
    fp["n_problems"] = int(n_problems)
    fp["weights"] = wrapper_id
    return fp


def meta_path(pickle_path):
    return pickle_path + ".meta.json"


def load_resumable(pickle_path, fingerprint, n_problems):
    """the result dicts of an earlier run of THIS experiment (eval_planning_2d.py:99-110), or [] with the reason printed: no
    fingerprint beside the file (not written by this harness / an older version) or another fingerprint -> nothing is reused"""
    import pickle
    if not os.path.exists(pickle_path):
        return []
    try:
        with open(meta_path(pickle_path)) as f:
            old = json.load(f)
    except (OSError, ValueError):
        print("eval_sharded: %s has no fingerprint file beside it (%s): not resumed, every problem is planned" % (pickle_path, meta_path(pickle_path)))
        return []
    diff = sorted(k for k in set(old) | set(fingerprint) if old.get(k) != fingerprint.get(k))
    if diff:
        print("eval_sharded: %s was written with other settings (%s): not resumed, every problem is planned"
              % (pickle_path, ", ".join("%s %r -> %r" % (k, old.get(k), fingerprint.get(k)) for k in diff)))
        return []
    with open(pickle_path, "rb") as f:
        return pickle.load(f)[:n_problems]


def write_meta(pickle_path, fingerprint):
    os.makedirs(os.path.dirname(pickle_path) or ".", exist_ok=True)
    with open(meta_path(pickle_path), "w") as f:
        json.dump(fingerprint, f)


def first_below(trace, threshold):
    """planning_block_gap's stopping rule (rrt_star_2d.py:159-196): number of iterations until the best path length is
    below the threshold (the list the reference returns has exactly that many entries), or -1."""
    hit = np.nonzero(np.asarray(trace) < threshold)[0]
    return int(hit[0]) + 1 if len(hit) else -1


def make_block_gap_record(pid, trace, threshold, n_vertices):
    """record of a block / gap problem: iterations to get below the threshold (-1: never within iter_max), the
    threshold, and the best cost at the fixed checkpoints counted from the FIRST solution like make_record"""
    stop = first_below(trace, threshold)
    rec = make_record(pid, trace[:stop] if stop > 0 else trace, n_vertices)
    rec[3] = stop
    return rec


def plan_batch_block_gap(problems, pids, thresholds, args, device_id, wrapper=None):
    """planning_block_gap for a batch: the persistent loop runs in segments of args.segment iterations; after each
    segment the problems whose best path length got below their threshold leave the batch.  An iteration's result
    only depends on the iterations before it and every problem owns its generators, so truncating the cost trace at
    the first sub-threshold entry gives exactly the list the reference's early-exit loop returns."""
    from . import batch
    dim, flags, trees, streams, frames, guidance, order = _batch_setup(problems, pids, args, device_id, args.iter_max, wrapper)
    problems, pids, thresholds = [problems[i] for i in order], [pids[i] for i in order], [thresholds[i] for i in order]
    traces = [np.zeros(0) for _ in trees]
    active = list(range(len(trees)))
    done_iters = 0
    while active and done_iters < args.iter_max:
        seg = min(args.segment, args.iter_max - done_iters)
        r = batch.run_batch([trees[i] for i in active], [streams[i] for i in active], seg, flags, dim, [problems[i] for i in active],
                            guidance, [frames[i] for i in active], want_trace=True, init_clouds=done_iters == 0)
        _raise_failures(r, [pids[i] for i in active])
        still = []
        for j, i in enumerate(active):
            traces[i] = np.concatenate([traces[i], r["traces"][j]])
            if first_below(traces[i], thresholds[i]) < 0:
                still.append(i)
        active = still
        done_iters += seg
    recs = [make_block_gap_record(pid, tr, thr, t.n) for pid, tr, thr, t in zip(pids, traces, thresholds, trees)]
    batch.release_all(trees, streams)
    for t in trees:
        t.close()
    back = np.argsort(order)      # results in the caller's order
    return [recs[j] for j in back], [traces[j] for j in back]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--problem", default="random_2d", choices=["random_2d", "random_3d", "block", "gap"])
    ap.add_argument("--path_len_threshold_percentage", type=float, default=0.02, help="block: stop below best_path_len * (1 + this)")
    ap.add_argument("--segment", type=int, default=2000, help="block / gap: iterations per persistent launch")
    ap.add_argument("--planner", "-p", default="irrt_star", choices=list(PLANNERS))
    ap.add_argument("--neural_net", "-n", default="none", choices=["none", "pointnet2"],
                    help="nirrt_star / nrrt_star planners: the guidance network (demo_planning_2d.py:13-14)")
    ap.add_argument("--connect", "-c", default="none", choices=["none", "bfs"], help="bfs = the -C planners (neural connect)")
    ap.add_argument("--root_dir", default=".", help="where results/model_training/pointnet2_{2d,3d}/checkpoints/ lives")
    ap.add_argument("--pc_n_points", type=int, default=2048)
    ap.add_argument("--pc_over_sample_scale", type=int, default=5)
    ap.add_argument("--pc_sample_rate", type=float, default=0.5)
    ap.add_argument("--pc_update_cost_ratio", type=float, default=0.9)
    ap.add_argument("--connect_max_trial_attempts", type=int, default=5)
    ap.add_argument("--iter_max", type=int, default=None, help="default: 50000 (eval_planning_2d.py:19), 30000 for random_3d (eval_planning_3d.py:19)")
    ap.add_argument("--iter_after_initial", type=int, default=3000)
    ap.add_argument("--step_len", type=float, default=10)
    ap.add_argument("--clearance", type=float, default=None)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--max_problems", "--num_problems", dest="max_problems", type=int, default=None)
    ap.add_argument("--no_resume", action="store_true", help="plan every problem even if the result pickle already holds a prefix of them")
    ap.add_argument("--no_balance", action="store_true", help="plain round-robin sharding (no free-segment predictor)")
    ap.add_argument("--out", default="results/evaluation/sharded_result.json")
    ap.add_argument("--pickle_out", default="auto",
                    help="reference-format result pickle (list of env configs + 'result'); 'auto' = the reference's "
                         "results/evaluation/{2d,3d}/<problem>-<planner>-none-<n>.pickle, 'none' = skip")
    args = ap.parse_args()
    if args.clearance is None:
        args.clearance = 2 if args.problem == "random_3d" else 3
    if args.iter_max is None:      # the reference has one script per dimension, each with its own default
        args.iter_max = 30000 if args.problem == "random_3d" else 50000
    if args.connect == "bfs" and not args.planner.endswith("_c"):
        args.planner += "_c"
    if args.planner.startswith("n") and args.neural_net == "none":
        args.neural_net = "pointnet2"
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("eval_sharded needs MI355X GPUs (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    launched = "RANK" in os.environ and "MASTER_PORT" in os.environ      # under torchrun (also with ONE rank): RCCL process group
    if world > 1 or launched:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    from . import problems as P
    if args.problem == "random_2d":
        cfgs = P.get_random_2d_env_configs()
        get = P.get_random_2d_problem_input
    elif args.problem == "block":
        cfgs, get = P.get_block_env_configs(), P.get_block_problem_input
    elif args.problem == "gap":
        cfgs, get = P.get_gap_env_configs(), P.get_gap_problem_input
    else:
        cfgs = P.get_random_3d_env_configs()
        get = P.get_random_3d_problem_input
    if args.max_problems:
        cfgs = cfgs[: args.max_problems]
    pickle_path = args.pickle_out
    if pickle_path == "auto":
        pickle_path = os.path.join("results", "evaluation", "3d" if args.problem == "random_3d" else "2d",
                                   "%s-%s%s-%s-%d.pickle" % (args.problem, args.planner[:-2] if args.planner.endswith("_c") else args.planner,
                                                             "-c-bfs" if args.planner.endswith("_c") else "", args.neural_net, len(cfgs)))
    # resume like the reference (eval_planning_2d.py:99-110): an existing result file holds the first K problems' lists; they are
    # kept as they are and only problems K.. are planned.  Only a file of the SAME experiment counts (run_fingerprint beside the
    # pickle), and rank 0 decides for everybody: ranks without a shared file system would otherwise read different K and shard
    # inconsistently.
    wrapper = make_wrapper(args, 3 if args.problem == "random_3d" else 2, "cuda:%d" % local_rank) if args.neural_net == "pointnet2" else None
    weights_id = None
    if wrapper is not None:
        from . import png_wrapper as W
        ck = W.checkpoint_path(args.root_dir, 3 if args.problem == "random_3d" else 2)
        weights_id = "%s:%d" % (os.path.basename(ck), os.path.getsize(ck)) if os.path.exists(ck) else None
    fingerprint = run_fingerprint(args, len(cfgs), weights_id)
    loaded = []
    if pickle_path != "none" and not args.no_resume and rank == 0:
        loaded = load_resumable(pickle_path, fingerprint, len(cfgs))
    if dist.is_initialized() and world > 1:
        box = [loaded]
        dist.broadcast_object_list(box, src=0)
        loaded = box[0]
    # heavy problems (free straight start-goal segment) are dealt across the ranks first, then the rest (shard_indices)
    heavy = None
    if world > 1 and args.problem in ("random_2d", "random_3d") and not args.no_balance:
        heavy = [straight_segment_free(c["env_dict"], args.clearance) for c in cfgs]
    mine = shard_indices(len(cfgs), rank, world, heavy, first=len(loaded))
    t0 = time.time()
    recs, results = [], []
    # one rank plans problems K, K + 1, ... in order: the result file is rewritten after every batch (the reference writes after
    # every problem), so a crash leaves a resumable prefix.  With several ranks the lists meet on rank 0 at the end only.
    incremental = world == 1 and pickle_path != "none"
    if incremental:
        write_meta(pickle_path, fingerprint)
    for b0 in range(0, len(mine), args.batch):
        ids = mine[b0:b0 + args.batch]
        probs = []
        for i in ids:
            if args.problem == "random_3d":
                np.random.seed(i)   # gamma estimate consumes the global generator
            probs.append(get(cfgs[i]))
        thr = None
        if args.problem == "block":   # eval_planning_2d.py:117-121
            thr = [p["best_path_len"] * (1 + args.path_len_threshold_percentage) for p in probs]
            r, traces = plan_batch_block_gap(probs, ids, thr, args, local_rank, wrapper)
        elif args.problem == "gap":
            thr = [p["flank_path_len"] for p in probs]
            r, traces = plan_batch_block_gap(probs, ids, thr, args, local_rank, wrapper)
        else:
            r, traces = plan_batch(probs, ids, args, local_rank, wrapper)
        recs += r
        results += list(zip(ids, result_lists(args.problem, traces, thr)))
        if incremental:
            write_reference_pickle(pickle_path, cfgs, [(i, d["result"]) for i, d in enumerate(loaded)] + sorted(results))
    plan_s = time.time() - t0
    # the loaded problems' records (rank 0 only: they need no planning) from their lists, like a freshly planned one's (n_vertices
    # is not in the file: -1).  block / gap: the list IS the trace cut at the first entry below the threshold
    if rank == 0 and loaded:
        if args.problem in ("random_2d", "random_3d"):
            recs = [make_record(i, np.asarray(d["result"], dtype=np.float64), -1) for i, d in enumerate(loaded)] + recs
        else:
            old = []
            for i, d in enumerate(loaded):
                pr = get(cfgs[i])
                thr_i = pr["best_path_len"] * (1 + args.path_len_threshold_percentage) if args.problem == "block" else pr["flank_path_len"]
                old.append(make_block_gap_record(i, np.asarray(d["result"], dtype=np.float64), thr_i, -1))
            recs = old + recs
    allr = gather_records(np.array(recs).reshape(-1, RECORD_LEN), world, rank, device="cuda")
    all_results = gather_results(results, world, rank) if args.pickle_out != "none" else None
    rank_seconds = gather_rank_seconds(plan_s, len(mine), world, rank)
    if rank == 0 and all_results is not None:
        all_results = [(i, d["result"]) for i, d in enumerate(loaded)] + list(all_results)
        write_reference_pickle(pickle_path, cfgs, all_results)
        write_meta(pickle_path, fingerprint)
    if rank == 0:
        solved = allr[allr[:, 1] > 0]
        summary = {"problems": int(len(allr)), "solved": int(len(solved)), "world_size": world,
                   "resumed_from": len(loaded), "planned": int(len(cfgs) - len(loaded)),
                   "rank_problems": [int(v[1]) for v in rank_seconds], "rank_seconds": [round(float(v[0]), 3) for v in rank_seconds],
                   "median_first_solution_iter": float(np.median(solved[:, 1])) if len(solved) else None,
                   "mean_cost_at": {str(c): float(np.mean(solved[:, 4 + j][np.isfinite(solved[:, 4 + j])]))
                                    for j, c in enumerate(CHECKPOINTS) if len(solved) and np.isfinite(solved[:, 4 + j]).any()},
                   "iterations_planned": int(np.maximum(allr[len(loaded):, 3], 0).sum()),
                   "seconds": time.time() - t0}
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        with open(args.out, "w") as f:
            json.dump({"summary": summary, "records": allr.tolist()}, f)
        print(json.dumps(summary))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
