#!/usr/bin/env python3
"""bench.py — RRT*-family iterations/s while growing 50k-node trees on random_2d / random_3d (BASELINE.json).

Workload (BASELINE.json configs[1]): `irrt_star random_2d, 50k iters`, batched on one MI355X:
B independent planning problems (224x224 world, 30 circle obstacles, clearance 3, step_len 10), each
planned for `--iters` iterations (default 50 000) by the device-resident loop - ONE persistent
workgroup per tree; sampling (SampleFree, then informed once a solution exists), nearest, steer,
collision, Near, choose-parent, rewire, goal bookkeeping and best-solution tracking all happen in the
kernel.  `--algo rrt` runs plain RRT* on the same problems (uniform sampling, no solution tracking),
`--dim 3` the random_3d worlds (boxes + balls), `--world b30r16` the r in [16, 24] circles of SURVEY.md §8d.

One "step" = one pass of that loop over the whole batch (B x iters iterations), starting from fresh
one-vertex trees and freshly seeded generators: each problem's numpy / python MT19937 states (2 x 2.5 KB, seeded
like the reference seeds its process-global ones) are uploaded inside the step and every output the loop
consumes is produced by the tree's own wave (twist + tempering on the device) - no generator output exists
on the host or in HBM ahead of time.  N GPUs = N processes (torch.distributed / RCCL), each
with its own B problems (weak scaling); the only collectives are the timing protocol's barrier /
max-reduce and a gather of per-rank iteration counts.  `--gpus N` without a launcher re-executes itself
under `python -m torch.distributed.run --nproc-per-node N` (the driver's own torchrun launch is used as is).

Prints ONE JSON line (rank 0): the driver's contract fields plus
  roofline     — HBM roofline of the persistent kernel.  `achieved` = bytes the IMPLEMENTED algorithm has to move
                 (counted by the kernel: visited index slots, chain records, candidate records, re-costed vertices, list
                 re-evaluations, inserts, index rebuilds - nirrt_star_amd/_hip.useful_bytes) / HIP-event kernel time;
                 `traffic` = HBM-side bytes of a separate rocprofv3 --pmc pass of this exact configuration (named in
                 `traffic_source`; null when none was collected) and `wasted_traffic_ratio` = traffic / useful bytes;
                 `reference_scan_equiv_GBps` = what the REFERENCE algorithm's two O(n) scans per iteration would have
                 had to stream in the same time (a speed-up over the scan algorithm, not a roofline figure)
  cpu_baseline — the oracle (oracle/nirrt_oracle.c, C port of the reference loop incl. sampling) on this box's host
                 cores: C independent processes, one problem each, bounded iteration count
  time_to_first_solution — measured NIRRT_F_STOP_FIRST launches: problems run ONE AT A TIME (kernel time of each
                 launch) and as one batch (per-tree device clock from loop start to the first finite c_best)
"""
import argparse
import json
import os
import random
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s peak (6.3 TB/s achievable)
REF_PY = {"rrt": 153.0, "irrt": 43.0, "nirrt": None, "nirrt_c": None}  # reference numpy path in the survey container: context only
WORLDS = {"b30": "224x224, 30 circle obstacles r in [8, 12]", "b30r16": "224x224, 30 circle obstacles r in [16, 24], start / goal in one free component",
          "ref2d": "224x224, 8-12 rectangles (16-24) + 8-12 circles r in [16, 24]"}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--trees", type=int, default=8192, help="problems per GPU per step (3072 = 12 one-wave workgroups per CU are resident; more queue up "
                                                             "behind them and even out the heavy-tailed per-tree run times)")
    ap.add_argument("--iters", type=int, default=50000, help="planner iterations per problem (tree capacity)")
    ap.add_argument("--dim", type=int, default=2)
    ap.add_argument("--algo", default="irrt", choices=["irrt", "rrt", "nirrt", "nirrt_c"],
                    help="nirrt / nirrt_c: NIRRT*-PNG[(C)] with PointNet++ guidance (BASELINE configs 3-4) through the batched driver")
    ap.add_argument("--world", default="b30r16", choices=sorted(WORLDS),
                    help="b30r16 = SURVEY.md 8(d)'s primary world (30 circles r in [16, 24], start / goal in one free component); b30 = its "
                         "lighter fallback (r in [8, 12])")
    ap.add_argument("--pc-update-cost-ratio", type=float, default=0.9,
                    help="guided lines: refresh the cloud when the best cost drops below this fraction of the cost at the last refresh "
                         "(0.9 = the planner classes' and eval scripts' default; demo_planning_3d.py:21 passes 1.0: every improvement)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=6000, help="iterations each CPU-baseline process runs (per repetition)")
    ap.add_argument("--cpu-procs", type=int, default=0, help="CPU-baseline processes (0 = every host core)")
    ap.add_argument("--no-cpu-full", action="store_true", help="skip the ONE full-length single-core oracle run reported beside the sample")
    ap.add_argument("--no-ttfs", action="store_true")
    ap.add_argument("--ttfs", action="store_true", help="(marks the secondary lines that measure time-to-first-solution; the default line always does)")
    ap.add_argument("--no-single", action="store_true", help="skip the one-problem-alone latency runs (3 full-length launches)")
    ap.add_argument("--detail-file", default=os.path.join(ROOT, "gpurun_out", "bench_detail.json"),
                    help="the full per-line records of the secondary runs go here (the JSON line keeps a compact summary of each)")
    ap.add_argument("--pilot", type=int, default=0,
                    help="a step's iterations run as two launches: this many first, then the rest re-scheduled like --segments does; 0 = off")
    ap.add_argument("--segments", type=int, default=1, help="a step's iterations run as this many launches of equal length; between them the "
                                                           "trees are re-ordered (longest first by the last segment's device time) and re-sized "
                                                           "(--wide-visits / --narrow-visits)")
    ap.add_argument("--wide-visits", type=float, default=0.0, help="trees whose visits covered at least this many index slots per iteration in the "
                                                                    "last segment continue on 256 lanes (0 = never)")
    ap.add_argument("--narrow-visits", type=float, default=0.0, help="... on 128 lanes")
    ap.add_argument("--no-reorder", action="store_true", help="keep the dispatch order between segments")
    ap.add_argument("--free-first", type=int, default=1, help="1: problems with a free start-goal segment are dispatched first in the first launch")
    ap.add_argument("--run-ahead", type=int, default=-1,
                    help="1: problems with a free start-goal segment never wait for their turn in a time-sliced launch (nirrt_run_args.run_ahead); "
                         "-1 = 2D only (measured: b30 43.1 -> 48.4 M it/s, default line +0.4 %%; 3D with its lane groups and segments: 11.4 -> 10.5)")
    ap.add_argument("--free-lanes", type=int, default=0, choices=[0, 64, 128, 256],
                    help="workgroup size for the problems with a free start-goal segment (their Near sets grow to thousands of members: the "
                         "visit is arithmetic-bound and scales with the lanes); 0 = like the others")
    ap.add_argument("--traffic-file", default=os.path.join(ROOT, "profiles", "r06_traffic.json"),
                    help="PMC traffic table written by scripts/collect_traffic.py (an entry is used only if its key names this exact configuration)")
    ap.add_argument("--dry-run", action="store_true", help="launcher / process-group / timing protocol only, no GPU work (CPU test of --gpus N)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: every rank plans --trees problems of its own; strong: ONE fixed set of --problems problems (BASELINE config 5: "
                         "the 1000-problem evaluation set) is sharded round-robin over the ranks (problem i -> rank i mod N)")
    ap.add_argument("--problems", type=int, default=1000, help="size of the fixed problem set of --scaling strong")
    ap.add_argument("--no-secondary", action="store_true", help="skip the short runs of the other BASELINE configurations (N = 1 only)")
    ap.add_argument("--cpu-reps", type=int, default=1, help="repetitions of the truncated CPU-baseline sample (the median is reported)")
    ap.add_argument("--cpu-full-deadline", type=float, default=240.0,
                    help="cpu_baseline.value: every host core plans ONE problem of the batch for the FULL iteration count (the identical loop, SURVEY.md "
                         "8(d)(i)); a process still running after this many seconds stops and reports the iterations it did (0 = skip, the truncated "
                         "sample alone is reported)")
    return ap.parse_args(argv)


def config_key(args):
    size = ("set%d" % args.problems) if getattr(args, "scaling", "weak") == "strong" else str(args.trees)
    key = "%s_%dd_%s_%sx%d" % (args.algo, args.dim, args.world if args.dim == 2 else "ref3d", size, args.iters)
    ratio = getattr(args, "pc_update_cost_ratio", 0.9)
    if args.algo.startswith("nirrt") and ratio != 0.9:      # (another refresh policy is another workload: a key of its own)
        key += "_ratio%g" % ratio
    return key


def rank_problem_ids(args, rank, world):
    """problem ids this rank plans: weak = its own block of --trees problems; strong = its round-robin share of the fixed set"""
    if getattr(args, "scaling", "weak") == "strong":
        return list(range(rank, args.problems, world))
    return [rank * args.trees + b for b in range(args.trees)]


def make_problems(args, rank):
    """B problems for this rank: worlds 0..249 x 4 start/goal pairs = the 1000-problem evaluation set of
    SURVEY.md §8d (wrapping around for larger batches); planner seed = 1000 + problem id."""
    probs, cache = [], {}
    for pid in rank_problem_ids(args, rank, int(os.environ.get("WORLD_SIZE", "1"))):
        probs.append(make_problem(args, pid, cache))
    return probs


def make_problem(args, pid, cache=None):
    from nirrt_star_amd import worlds
    cache = {} if cache is None else cache
    if args.dim == 2:
        w = pid % 250
        if w not in cache:
            kind, rr = {"b30": ("b30", None), "b30r16": ("b30", (16, 24)), "ref2d": ("ref2d", None)}[args.world]
            ed = worlds.random_world_2d(w, kind, circle_radius_range=rr)
            cache[w] = (ed, worlds.rasterize_mask_2d(ed["env_dims"], ed["rectangle_obstacles"], ed["circle_obstacles"]))
        pr = worlds.problem_2d(cache[w][0], (pid // 250) % 4, mask=cache[w][1])      # (one rasterisation per world, not per problem)
        pr["clearance"] = 3
    else:
        np.random.seed(pid)
        pr = worlds.problem_3d(worlds.random_world_3d(pid % 1000))
        pr["clearance"] = 2
    pr["pid"] = pid
    return pr


def word_budgets(args):
    """raw generator outputs the ORACLE is handed per problem (oracle/cpu_bench.py, scripts/perf_*.py, the full-size tests): the
    C restatement consumes pre-drawn words; the device draws from the trees' own generators and needs no budget"""
    D, it = args.dim, args.iters
    if args.algo == "rrt":
        return it * D * 2 * 4 + 4096, 0          # four SampleFree attempts per iteration (crowded worlds reject 2 of 3)
    if D == 2:
        # SampleFree (4 words per attempt) until the first solution, then python-random unit disk.  In the r in [16, 24] worlds
        # only ~39 % of the range is free (10 words per sample, 14 in the most crowded ones), and the informed sampler's
        # unit-disk points (5.1 python words each) are rejected at the same rate: 13 words per sample on average, 18+ in the
        # most crowded worlds
        crowded = getattr(args, "world", "b30") == "b30r16"
        return it * (24 if crowded else 12) + 4096, it * (48 if crowded else 14) + 4096
    return it * 6 * 40 + 4096, 0                 # 3D informed sampling stays on the numpy stream


def problem_words(args, pid, n_np, n_py):
    """raw generator outputs of problem `pid`: np.random.seed(1000 + pid); random.seed(1000 + pid) (oracle side only)"""
    from nirrt_star_amd import sampling
    np.random.seed(1000 + pid)
    random.seed(1000 + pid)
    return sampling.peek_np_words(n_np), (sampling.peek_py_words(n_py) if n_py else None)


def problem_generators(pids):
    """(numpy states, python states) of the problems' generators right after `np.random.seed(1000 + pid); random.seed(1000 + pid)`
    (what the reference's evaluation loop does before planning a problem): per problem a (key[624], pos) pair"""
    from nirrt_star_amd import _hip
    np_st = [_hip.np_state(np.random.RandomState(1000 + pid)) for pid in pids]
    py_st = [_hip.py_state(random.Random(1000 + pid)) for pid in pids]
    return np_st, py_st


def respawn_under_torchrun(args):
    """`python bench.py --gpus N` with no launcher around it: become N ranks (one per GPU) ourselves."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(respawn_under_torchrun(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    import torch
    import torch.distributed as dist
    have_gpu = torch.cuda.is_available()
    if not have_gpu and not args.dry_run:
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    if have_gpu:
        torch.cuda.set_device(local_rank)
    # under a launcher (RANK / MASTER_PORT in the environment) a process group exists even for ONE rank: the timing protocol's
    # barrier / all_reduce then run through RCCL as they do at N > 1 (a one-GPU box cannot host two RCCL ranks)
    grouped = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)
    if grouped:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if have_gpu:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    dev = "cuda:%d" % local_rank if have_gpu else "cpu"

    def barrier():
        if have_gpu:
            torch.cuda.synchronize()
        if grouped:
            dist.barrier()
        if have_gpu:
            torch.cuda.synchronize()

    def reduce_time_and_work(elapsed, work):
        tot = torch.tensor([elapsed, float(work)], dtype=torch.float64, device=dev)
        if grouped:
            tmax = tot.clone()
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
            return float(tmax[0].item()), float(tot[1].item())
        return elapsed, float(work)

    if args.dry_run:
        barrier()
        t0 = time.perf_counter()
        barrier()
        emax, total = reduce_time_and_work(time.perf_counter() - t0, args.trees * args.iters * args.steps)
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "work_all_ranks": total, "config": {"workload": config_key(args)}}))
        if grouped:
            dist.barrier()
            dist.destroy_process_group()
        return

    from nirrt_star_amd import _hip, batch, build, sampling
    build.build()
    if args.algo.startswith("nirrt"):
        return bench_nirrt(args, rank, world, local_rank, barrier, reduce_time_and_work)

    full_proc = start_cpu_full_run(args) if (rank == 0 and world == 1 and not args.no_cpu_baseline and not args.no_cpu_full) else None
    t_setup = time.perf_counter()
    probs = make_problems(args, rank)
    synthesis_s = time.perf_counter() - t_setup      # the BENCHMARK's own work: 250 synthetic worlds drawn and rasterised in Python
    D, B, iters = args.dim, len(probs), args.iters   # (strong scaling: this rank's share of the fixed set)
    flags = _hip.F_IRRT if args.algo == "irrt" else 0
    # every tree of the rank in ONE creation call (nirrt_create_batch: host work per tree, one device pass), the informed-sampling
    # frames in one launch
    trees = _hip.create_trees(D, iters, [(pr["x_start"], pr["x_goal"], 10, pr["search_radius"], pr["clearance"], pr["env"]) for pr in probs],
                              device_id=local_rank)
    _hip.set_informed_batch(trees, [sampling.informed_frame(pr["x_start"], pr["x_goal"]) for pr in probs])
    # Scheduling of the independent problems (host side; problems, seeds and results are untouched).  Per-tree run times are
    # heavy-tailed - a problem whose straight start-goal segment is free ends up with an informed set collapsed onto that segment,
    # Near sets of thousands of members, and takes 2-3x the median - and a persistent launch lasts as long as its slowest tree.
    # Launch order = dispatch order, so those problems go first (measured: 20.4 vs 18.4 M it/s without the ordering).
    # Optional (--pilot N, off by default: it did not beat the simple ordering): run N iterations first, then the rest with the
    # trees of the largest measured Near sets dispatched first and on 256- / 128-lane workgroups (nirrt_run_args.lanes_hint).
    first_order = list(range(B))
    first_hint = None
    free_line = None
    if args.algo == "irrt" and B > 1 and (args.free_first or args.free_lanes):
        free_line = [bool(f) for f in ~_hip.collision_each(trees, [np.stack([np.asarray(pr["x_start"], dtype=np.float64),
                                                                             np.asarray(pr["x_goal"], dtype=np.float64)]) for pr in probs])]
        if args.free_first:
            first_order = sorted(range(B), key=lambda b: (not free_line[b], b))
        if args.free_lanes:
            first_hint = np.array([args.free_lanes if free_line[b] else 0 for b in first_order], dtype=np.int32)
    # inputs: each problem's two generators seeded like the reference seeds its process-global ones (np.random.seed(s);
    # random.seed(s)).  Only the 2 x 2.5 KB states exist on the host; every output is produced on the device inside the loop.
    t_inputs = time.perf_counter()
    np_states, py_states = problem_generators([pr["pid"] for pr in probs])
    input_generation_s = time.perf_counter() - t_inputs   # host: seeding 2 B generators (outside the timed region; the upload is inside)
    setup_s = time.perf_counter() - t_setup               # this rank: worlds, B x nirrt_create, collision probes, generator seeding

    # a step's iterations as `--segments` launches of equal length (or --pilot P: P, then the rest); between launches the host
    # re-schedules from the device's own measurements (nirrt_star_amd.batch.run_scheduled)
    if 0 < args.pilot < iters and B > 1:
        seg_len = [args.pilot, iters - args.pilot]
    elif args.segments > 1 and B > 1:
        q = iters // args.segments
        seg_len = [q] * (args.segments - 1) + [iters - q * (args.segments - 1)]
    else:
        seg_len = [iters]
    n_seg = len(seg_len)

    def one_step():
        """one pass of the loop over the whole batch = n_seg launches; returns the sums the report needs"""
        _hip.reset_batch(trees)
        _hip.set_generators(trees, np_states, py_states)      # np.random.seed(s); random.seed(s) of every problem
        # (the same problems never wait for their turn in a time-sliced launch: their late slices are the long ones)
        return batch.run_scheduled(trees, seg_len, flags, order=first_order, hint=first_hint, wide_visits=args.wide_visits,
                                   narrow_visits=args.narrow_visits, reorder=not args.no_reorder,
                                   ahead=free_line if (free_line is not None and (args.run_ahead == 1 or (args.run_ahead < 0 and D == 2))) else None)

    for _ in range(args.warmup):
        one_step()
    kernel_ms, useful, alg_elems, done_iters, visit_b = [], [], [], [], []
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        r = one_step()
        kernel_ms.append(r["kernel_ms"])
        useful.append(_hip.useful_bytes(r["stats"], D))
        visit_b.append(float(r["stats"][:, 1].sum()))
        alg_elems.append(int(r["alg_elems"].sum()))
        done_iters.append(int(r["iters_done"].sum()))
    barrier()
    elapsed = time.perf_counter() - t0
    n_final = [t.n for t in trees]
    n_sol = [len(trees[b].solutions) for b in range(0, B, max(1, B // 16))]
    short = int((r["iters_done"] < iters).sum())
    st = r["stats"].astype(np.float64)
    tree_s = r["seconds"]                           # per-tree seconds inside the launches of the last step (100 MHz device wall clock)
    elapsed_max, total_iters = reduce_time_and_work(elapsed, sum(done_iters))
    setup_max, _ = reduce_time_and_work(setup_s, 0)       # (slowest rank's set-up: what an N-GPU run waits for before its first step)
    value = total_iters / elapsed_max

    if rank == 0:
        k_ms = float(np.mean(kernel_ms))
        k_s = k_ms * 1e-3
        useful_b = float(np.mean(useful))
        achieved = useful_b / k_s / 1e9
        traffic, traffic_src = measured_traffic(args)
        per_it = st.sum(axis=0) / max(1.0, st[:, 13].sum())
        out = {
            "metric": "RRT* iters/sec (50k-node tree), random_%dd" % D,
            "value": value, "unit": "iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed_max / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "process_group": (dist.get_backend() if grouped else None),
            "input_generation_s": input_generation_s, "setup_seconds_max_over_ranks": setup_max,
            # (of which: the benchmark drawing and rasterising its synthetic worlds in Python / everything the library does for the
            #  batch - nirrt_create_batch, informed frames, free-segment probes - plus seeding 2 B host generators; rank 0)
            "setup_split_s": {"problem_synthesis": synthesis_s, "trees_and_generators": setup_s - synthesis_s},
            "end_to_end_value": total_iters / (elapsed_max + input_generation_s),   # incl. seeding every problem's generators on the host
            "config": {"workload": ("%s_star random_2d (%s: %s; clearance 3, step_len 10), %d problems/GPU x %d iters, "
                                    "device-resident batched loop with in-kernel sampling" % (args.algo, args.world, WORLDS[args.world], B, iters))
                       if D == 2 else ("%s_star random_3d (50^3, 6-9 boxes + 6-9 balls; clearance 2, step_len 10), %d problems/GPU x %d iters, "
                                       "device-resident batched loop with in-kernel sampling" % (args.algo, B, iters)),
                       "key": config_key(args), "trees_per_gpu": B, "iters": iters, "dim": D, "step_len": 10,
                       "problem_set": ("fixed set of %d problems, problem i on rank i mod %d" % (args.problems, world)) if args.scaling == "strong"
                                      else "%d problems per rank" % B,
                       "clearance": probs[0]["clearance"], "mean_final_vertices": float(np.mean(n_final)),
                       "mean_solutions_per_tree": float(np.mean(n_sol)), "trees_stopped_early": short,
                       "launches_per_step": n_seg, "trees_on_256_lanes": r["wide"], "trees_on_128_lanes": r["narrow"],
                       "per_tree_seconds": {"mean": float(tree_s.mean()), "median": float(np.median(tree_s)), "max": float(tree_s.max())}},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "wasted_traffic_ratio": (traffic / useful_b) if traffic else None,
                         "kernel": "k_run_pool<%d> (time-sliced; k_run_sample<%d> when the batch fits the GPU at once)" % (D, D), "kernel_ms": k_ms, "kernel_ms_is": "sum of the step's %d launches" % n_seg if n_seg > 1 else "the step's launch",
                         "useful_bytes_per_launch": useful_b, "visited_index_bytes_per_launch": float(np.mean(visit_b)),
                         "per_iteration": {"visited_slots": per_it[0], "visit_bytes": per_it[1], "near_members": per_it[2],
                                           "members_spilled": per_it[3], "chain_records": per_it[4], "rewire_candidates": per_it[5],
                                           "rewired": per_it[6], "recosted": per_it[7], "list_entries": per_it[8] + per_it[21],
                                           "inserted": per_it[9], "rebuilt": per_it[10], "nearest_revisits": per_it[11],
                                           "whole_tree_visits": per_it[12],
                                           "useful_bytes": useful_b / max(1.0, float(np.mean(done_iters)))},
                         "reference_scan_equiv_GBps": float(np.mean(alg_elems)) * D * 8.0 / k_s / 1e9},
            "reference_python_survey_container_its": REF_PY[args.algo],
        }
        if short:
            out["warning"] = ("%d of %d trees stopped before iteration %d (a draw rejected 2^22 generator outputs, or a tree ran out of capacity): `value` counts only the iterations "
                              "that ran" % (short, B, iters))
        detail_head = {}
        if not args.no_ttfs:
            t_full = time_to_first_solution(args, trees, np_states, py_states, flags)
            detail_head["time_to_first_solution"] = t_full
            # (the line keeps the medians; the full record goes to the detail file)
            out["time_to_first_solution"] = {"single": {k: t_full["single"][k] for k in ("problems", "solved", "median_seconds", "median_iterations")},
                                             "batch": {k: t_full["batch"][k] for k in ("problems", "solved", "median_seconds", "median_iterations")},
                                             "iteration_cap": t_full["iteration_cap"]}
        if not args.no_ttfs and not args.no_single:
            out["single_tree"] = single_tree_latency(args, trees, np_states, py_states, flags)
        if not args.no_cpu_baseline and world == 1:      # (the host-core baseline belongs to the N = 1 line)
            out["cpu_baseline"] = cpu_baseline(args, full_proc)
        if world == 1 and not args.no_secondary:
            # the other BASELINE configurations as short runs in processes of their own: this one's trees and inputs go first
            for t_ in trees:
                t_.close()
            trees = []
            torch.cuda.empty_cache()
            _hip.pool_trim()
            detail = secondary_runs(args)
            detail["eval_set protocol (config 5 as the reference runs it)"] = eval_protocol_run(args)
            try:
                os.makedirs(os.path.dirname(args.detail_file), exist_ok=True)
                with open(args.detail_file, "w") as f:
                    json.dump({"headline": dict(out, **detail_head), "secondary": detail}, f, indent=1)
            except OSError:
                pass
            # LAST key of the line, compact: the driver keeps the line's tail
            out["secondary"] = {k: compact(v) for k, v in detail.items()}
        print(json.dumps(out))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def bench_nirrt(args, rank, world, local_rank, barrier, reduce_time_and_work):
    """NIRRT*-PNG[(C)] on B problems per GPU: the persistent loop returns to the host whenever trees need a new guidance cloud;
    the due clouds are generated from each problem's own generator, down-sampled in one launch and classified in ONE
    batched PointNet++ forward (nirrt_star_amd/batch.py).  The timed step therefore contains host work (cloud candidates,
    word windows) and the network; `value` is still iterations of all trees / wall time."""
    import torch
    import torch.distributed as dist
    from types import SimpleNamespace as NS
    from nirrt_star_amd import _hip, batch, eval_sharded, sampling
    D, B, iters = args.dim, args.trees, args.iters
    probs = make_problems(args, rank)
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):   # the wrapper announces itself like the reference's does; stdout carries ONE JSON line
        wrapper = eval_sharded.make_wrapper(NS(root_dir=os.path.join(ROOT, "gpurun_out", "bench_ck")), D, "cuda:%d" % local_rank)
    guidance = batch.Guidance(wrapper, D, 10, pc_update_cost_ratio=args.pc_update_cost_ratio, connect=args.algo == "nirrt_c", device_id=local_rank)
    trees = _hip.create_trees(D, iters, [(pr["x_start"], pr["x_goal"], 10, pr["search_radius"], pr["clearance"], pr["env"]) for pr in probs],
                              device_id=local_rank)
    frames = [sampling.informed_frame(pr["x_start"], pr["x_goal"]) for pr in probs]
    _hip.set_informed_batch(trees, frames)

    # inputs: every step starts from freshly seeded generators (np.random.seed(s); random.seed(s); torch.manual_seed(s) per
    # problem); run_batch hands their states to the trees inside the timed step, every output is produced on the device
    dev = torch.device("cuda", local_rank)
    primed = [[batch.ProblemStreams(1000 + pr["pid"]) for pr in probs] for _ in range(args.warmup + args.steps)]

    def one_step():
        _hip.reset_batch(trees)
        for k in guidance.seconds:
            guidance.seconds[k] = 0.0
        return batch.run_batch(trees, primed.pop(0), iters, _hip.F_IRRT, D, probs, guidance, frames, want_trace=False)

    for _ in range(args.warmup):
        one_step()
    barrier()
    t0 = time.perf_counter()
    done, k_ms, launches = 0, [], []
    f0, c0 = guidance.calls, guidance.clouds_classified
    for _ in range(args.steps):
        r = one_step()
        done += int(r["iters_done"].sum())
        k_ms.append(r["kernel_ms"])
        launches.append(r["launches"])
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed_max, total_iters = reduce_time_and_work(elapsed, done)
    if rank == 0:
        st = r["stats"].astype(np.float64)
        useful_b = _hip.useful_bytes(r["stats"], D)
        traffic, traffic_src = measured_traffic(args)
        out = {"metric": "RRT* iters/sec (50k-node tree), random_%dd" % D, "value": total_iters / elapsed_max, "unit": "iterations/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed_max / args.steps * 1e3,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64 (tree) / f32 (PointNet++)", "data": "synthetic",
               "config": {"workload": "%s_star -n pointnet2%s random_%dd, %d problems/GPU x %d iters, 2048-point guidance clouds, batched "
                                      "PointNet++ refresh (synthetic weights; generators, cloud candidates, down-sampling, network input and predictions on the device, inside the timed step)"
                                      % ("nirrt", " -c bfs" if args.algo == "nirrt_c" else "", D, B, iters),
                          "key": config_key(args), "trees_per_gpu": B, "iters": iters, "dim": D,
                          "launches_per_step": float(np.mean(launches)), "forwards_per_step": (guidance.calls - f0) / args.steps,
                          "clouds_per_step": (guidance.clouds_classified - c0) / args.steps,
                          "mean_final_vertices": float(np.mean([t.n for t in trees])), "failed": len(r["failed"]),
                          "host_seconds_last_step": {k: round(v, 3) for k, v in r["host_seconds"].items()}},
               "roofline": {"bound": "hbm", "achieved": useful_b / (k_ms[-1] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": useful_b / (k_ms[-1] * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                            "wasted_traffic_ratio": (traffic / useful_b) if traffic else None, "kernel": "k_run_sample<%d>" % D,
                            "kernel_ms": float(np.mean(k_ms)), "kernel_share_of_step": float(np.mean(k_ms)) / (elapsed_max / args.steps * 1e3),
                            "useful_bytes_per_step": useful_b, "near_members_per_iteration": float(st[:, 2].sum() / max(1.0, st[:, 13].sum()))}}
        print(json.dumps(out))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


TTFS = ["--ttfs"]   # these lines also measure time-to-first-solution - the metric's second half (the others run --no-ttfs)
SECONDARY = [   # (label, bench arguments): each runs `--steps 1 --warmup 0` in a process of its own
    ("rrt_2d", ["--algo", "rrt", "--world", "b30"] + TTFS),
    ("rrt_3d", ["--algo", "rrt", "--dim", "3"] + TTFS),
    # (trees whose visits cover thousands of index slots per iteration - the degenerate, near-straight-line class - move to 256 / 128 lanes)
    ("irrt_3d", ["--algo", "irrt", "--dim", "3", "--trees", "4096", "--segments", "3", "--wide-visits", "6000", "--narrow-visits", "2000"] + TTFS),
    ("irrt_2d_b30 (r in [8, 12])", ["--algo", "irrt", "--world", "b30"]),
    # the reference's own random_2d obstacle distribution: 8-12 rectangles + 8-12 circles (env_configs/random_2d.yml:5-6)
    ("irrt_2d_ref2d (rectangles + circles)", ["--algo", "irrt", "--world", "ref2d"] + TTFS),
    # (the guided 2D lines run on the primary world b30r16 since round 5, like the headline)
    # (one untimed warm-up step: a guided step's first run pays for the network's one-time set-up - GEMM heuristics, graph capture -
    #  7-15 % of a step; the unguided lines have nothing of the kind)
    ("nirrt_2d", ["--algo", "nirrt", "--trees", "4096", "--warmup", "1"]),
    ("nirrt_c_2d (config 3)", ["--algo", "nirrt_c", "--trees", "2048", "--warmup", "1"]),
    ("nirrt_3d (config 4)", ["--algo", "nirrt", "--dim", "3", "--trees", "2048", "--warmup", "1"]),
    # the reference's own 3D demo refreshes the cloud on EVERY improvement (demo_planning_3d.py:21: pc_update_cost_ratio = 1.0):
    # ~65 refreshes per tree instead of ~4.5 (no warm-up step here: a step is half a minute)
    ("nirrt_3d ratio 1.0 (config 4 at the 3D demo's refresh policy)", ["--algo", "nirrt", "--dim", "3", "--trees", "2048", "--pc-update-cost-ratio", "1.0"]),
    # BASELINE config 5 as written, on ONE GPU: the fixed 1000-problem evaluation set (the anchor of the strong-scaling curve)
    ("irrt_2d eval set (config 5, N = 1)", ["--algo", "irrt", "--scaling", "strong", "--problems", "1000"]),
]


def secondary_runs(args):
    """Short driver-witnessed lines of the other BASELINE configurations (RRT* 2D / 3D, IRRT* 3D, the r in [16, 24] world, configs
    3 and 4): one step each, no warm-up, same iteration count; the fields a reader needs to judge them."""
    out = {}
    for label, extra in SECONDARY:
        ttfs = "--ttfs" in extra
        cmd = [sys.executable, os.path.abspath(__file__), "--steps", "1", "--warmup", "0", "--iters", str(args.iters),
               "--no-cpu-baseline", "--no-secondary", "--no-single"] + ([] if ttfs else ["--no-ttfs"]) + list(extra)
        t0 = time.perf_counter()
        try:
            first_error = None
            for attempt in (0, 1):      # (one retry: a line that dies says so in `retried_after` and gets a second chance at a number)
                p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
                line = [l for l in p.stdout.splitlines() if l.startswith("{")]
                if p.returncode == 0 and line:
                    break
                err = "rc %d: %s" % (p.returncode, ([l for l in p.stderr.strip().splitlines() if "amdgpu.ids" not in l] or ["no output"])[-1][:160])
                first_error = first_error or err
            if p.returncode != 0 or not line:
                out[label] = {"error": first_error}
                continue
            d = json.loads(line[-1])
            cfg, rf = d["config"], d["roofline"]
            out[label] = {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "kernel_ms": rf.get("kernel_ms"),
                          "roofline_frac": rf.get("frac"), "traffic": rf.get("traffic"), "key": cfg.get("key"),
                          "trees_per_gpu": cfg.get("trees_per_gpu"), "trees_stopped_early": cfg.get("trees_stopped_early", cfg.get("failed")),
                          "per_tree_seconds": cfg.get("per_tree_seconds"), "host_seconds": cfg.get("host_seconds_last_step"),
                          "launches_per_step": cfg.get("launches_per_step"), "trees_on_256_lanes": cfg.get("trees_on_256_lanes"),
                          "problem_set": cfg.get("problem_set"), "wasted_traffic_ratio": rf.get("wasted_traffic_ratio"),
                          "kernel_share_of_step": rf.get("kernel_share_of_step"), "time_to_first_solution": d.get("time_to_first_solution"),
                          "workload": cfg.get("workload"),
                          "warning": d.get("warning"), "wall_s": time.perf_counter() - t0}
            if first_error:
                out[label]["retried_after"] = first_error
        except subprocess.TimeoutExpired:
            out[label] = {"error": "timed out after 900 s"}
    return out


def compact(v):
    """one secondary record as the few numbers the line itself carries: M it/s, roofline fraction, HBM traffic / algorithmic
    bytes, share of the step spent in the kernel, time-to-first-solution medians (ms; one problem alone / in a 256-problem launch)"""
    if "error" in v:
        return {"error": v["error"][:80]}
    c = {"Mits": round(v["value"] / 1e6, 2)}
    if v.get("roofline_frac") is not None:
        c["frac"] = round(v["roofline_frac"], 3)
    if v.get("wasted_traffic_ratio") is not None:
        c["waste"] = round(v["wasted_traffic_ratio"], 2)
    if v.get("kernel_share_of_step") is not None:
        c["kshare"] = round(v["kernel_share_of_step"], 2)
    t = v.get("time_to_first_solution")
    if t and t["single"]["median_seconds"] is not None and t["batch"]["median_seconds"] is not None:
        c["ttfs_ms"] = [round(t["single"]["median_seconds"] * 1e3, 2), round(t["batch"]["median_seconds"] * 1e3, 2)]
    for k in ("resumed_s", "problems", "retried_after"):
        if k in v:
            c[k] = v[k]
    if v.get("trees_stopped_early"):
        c["short"] = v["trees_stopped_early"]
    return c


def eval_protocol_run(args):
    """BASELINE config 5 the way the reference runs it (eval_planning_2d.py:83-136): planning_random(3000) - plan until the first
    solution, then 3000 more iterations - over the 1000-problem evaluation set, results in the reference's pickle; then the same
    command again on a result file cut back to its first 900 problems (the reference's resume: eval_planning_2d.py:99-110).
    value = iterations planned / wall seconds of the harness process's planning loop (its own summary)."""
    import pickle
    import shutil
    import tempfile
    work = tempfile.mkdtemp(prefix="nirrt_eval_")
    cmd = [sys.executable, "-m", "nirrt_star_amd.eval_sharded", "--problem", "random_2d", "--planner", "irrt_star", "--iter_after_initial", "3000",
           "--out", os.path.join(work, "summary.json")]
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    try:
        runs = []
        for k in range(2):
            t0 = time.perf_counter()
            p = subprocess.run(cmd, cwd=work, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
            line = [l for l in p.stdout.splitlines() if l.startswith("{")]
            if p.returncode != 0 or not line:
                return {"error": "rc %d: %s" % (p.returncode, p.stderr.strip().splitlines()[-1] if p.stderr.strip() else "no output")}
            d = json.loads(line[-1])
            d["wall_s"] = time.perf_counter() - t0
            runs.append(d)
            if k == 0:      # cut the result file back to its first 900 problems: the second run resumes from there
                pk = os.path.join(work, "results", "evaluation", "2d", "random_2d-irrt_star-none-%d.pickle" % d["problems"])
                with open(pk, "rb") as f:
                    lst = pickle.load(f)
                with open(pk, "wb") as f:
                    pickle.dump(lst[: int(0.9 * len(lst))], f)
        full, res = runs
        return {"value": full["iterations_planned"] / full["seconds"], "unit": "iterations/s", "problems": full["problems"], "solved": full["solved"],
                "seconds": full["seconds"], "process_wall_s": full["wall_s"], "iterations_planned": full["iterations_planned"],
                "median_first_solution_iter": full["median_first_solution_iter"],
                "resumed_from": res["resumed_from"], "resumed_planned": res["planned"], "resumed_s": round(res["seconds"], 2),
                "workload": "eval_sharded --problem random_2d --planner irrt_star --iter_after_initial 3000 (planning_random over the 1000-problem set, N = 1)"}
    except subprocess.TimeoutExpired:
        return {"error": "timed out after 900 s"}
    finally:
        shutil.rmtree(work, ignore_errors=True)


def single_tree_latency(args, trees, np_states, py_states, flags):
    """ONE problem planned alone for the full iteration count (what demo_planning_2d.py:85-90 does): the whole GPU serves one
    tree, 256-thread kernels; HIP-event time of the launch, median over the first 3 problems of the batch."""
    from nirrt_star_amd import _hip
    ms, n_fin = [], []
    for b in range(min(3, len(trees))):
        trees[b].reset()
        _hip.set_generators([trees[b]], [np_states[b]], [py_states[b]])
        r = _hip.run_sampling([trees[b]], args.iters, flags=flags)
        ms.append(r["kernel_ms"])
        n_fin.append(int(trees[b].n))
    return {"problems": len(ms), "iterations": args.iters, "median_seconds": float(np.median(ms)) * 1e-3,
            "iterations_per_second": args.iters / (float(np.median(ms)) * 1e-3), "final_vertices": n_fin}


def time_to_first_solution(args, trees, np_states, py_states, flags):
    """Measured, not interpolated: NIRRT_F_STOP_FIRST launches on the first problems of the batch.  `single` = one
    problem per launch (the whole GPU serves one tree, 256-thread kernels): HIP-event time of the launch.  `batch` =
    the same problems in ONE launch: per-tree device clock (stats[14..15]) from the start of the tree's loop to the
    iteration that produced its first finite best cost."""
    from nirrt_star_amd import _hip
    cap = min(args.iters, 20000)
    fl = flags | _hip.F_STOP_FIRST | (_hip.F_GOAL_SCAN if args.algo == "rrt" else 0)
    n_single, n_batch = min(16, len(trees)), min(256, len(trees))
    single_ms, single_it = [], []
    for b in range(n_single):
        trees[b].reset()
        _hip.set_generators([trees[b]], [np_states[b]], [py_states[b]])
        r = _hip.run_sampling([trees[b]], cap, flags=fl, want_trace=True)
        it = int(r["iters_done"][0])
        if it > 0 and np.isfinite(r["cost_trace"][0, it - 1]):
            single_ms.append(r["kernel_ms"])
            single_it.append(it)
    sub = list(range(n_batch))
    for b in sub:
        trees[b].reset()
    _hip.set_generators([trees[b] for b in sub], [np_states[b] for b in sub], [py_states[b] for b in sub])
    r = _hip.run_sampling([trees[b] for b in sub], cap, flags=fl, want_trace=True)
    its = r["iters_done"]
    found = np.array([its[j] > 0 and np.isfinite(r["cost_trace"][j, its[j] - 1]) for j in range(len(sub))])
    secs = (r["stats"][:, 15] - r["stats"][:, 14]) / 1e8
    return {"single": {"problems": n_single, "solved": len(single_ms), "median_seconds": float(np.median(single_ms)) * 1e-3 if single_ms else None,
                       "median_iterations": float(np.median(single_it)) if single_it else None},
            "batch": {"problems": n_batch, "solved": int(found.sum()), "launch_ms": r["kernel_ms"],
                      "median_seconds": float(np.median(secs[found])) if found.any() else None,
                      "max_seconds": float(secs[found].max()) if found.any() else None,
                      "median_iterations": float(np.median(its[found])) if found.any() else None},
            "iteration_cap": cap, "how": "NIRRT_F_STOP_FIRST launches; single = HIP-event time per one-problem launch, "
                                         "batch = per-tree device wall clock inside one launch"}


def measured_traffic(args):
    """HBM-side bytes per launch from the PMC passes of scripts/collect_traffic.py (collected separately, as the profiling
    guide prescribes) - only if the table holds THIS configuration; the source is named in the line."""
    try:
        with open(args.traffic_file) as f:
            tab = json.load(f)
        e = tab["entries"][config_key(args)]
        # (a flat string: the driver's record keeps scalars of the roofline object and drops nested ones)
        return float(e["traffic_bytes"]), ("%s, entry %s, collected %s by scripts/collect_traffic.py: separate rocprofv3 --pmc FETCH_SIZE / "
                                           "WRITE_SIZE passes of this exact configuration (%.3f x FETCH_SIZE + %.3f x WRITE_SIZE, factors from %s); "
                                           "a constant replayed into the line, not a counter read in this run"
                                           % (os.path.relpath(args.traffic_file, ROOT), config_key(args), e.get("collected"), e.get("f_fetch", 2.0),
                                              e.get("f_write", 1.0), e.get("calibration")))
    except Exception:
        return None, None


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def start_cpu_full_run(args):
    """ONE problem (problem 0 of the batch) for the FULL iteration count on one host core, started when the bench starts and
    collected at its end (the oracle's per-iteration cost grows with the tree: ~minutes for 50 000 iterations)"""
    cmd = [sys.executable, os.path.join(ROOT, "oracle", "cpu_bench.py"), "--algo", args.algo, "--dim", str(args.dim),
           "--world", args.world, "--iters", str(args.iters), "--cap", str(args.iters), "--pid", "0"]
    return subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)


def cpu_baseline(args, full_proc=None):
    """The oracle (C port of the reference loop, incl. sampling and the reference's un-cached cost walks) on this box's
    host cores: C independent processes (oracle/cpu_bench.py; C = every host core), process i plans problem i of the batch for
    --cpu-iters iterations from its own seeded generators.  One repetition's value = iterations of all processes / wall time of
    the slowest; --cpu-reps repetitions, the MEDIAN is reported.  Beside it: one full-length single-core run (full_run)."""
    procs = args.cpu_procs or (os.cpu_count() or 1)
    cmd = [sys.executable, os.path.join(ROOT, "oracle", "cpu_bench.py"), "--algo", args.algo, "--dim", str(args.dim),
           "--world", args.world, "--iters", str(min(args.cpu_iters, args.iters)), "--cap", str(args.iters)]
    full = None
    if full_proc is not None:      # it has had the whole GPU part of the bench to itself; now it shares the cores with the sample
        try:
            outp = full_proc.communicate(timeout=900)[0]
            if full_proc.returncode == 0:
                r = json.loads(outp.strip().splitlines()[-1])
                full = {"iterations": r["iters"], "seconds": r["seconds"], "iterations_per_second": r["iters"] / r["seconds"],
                        "final_vertices": r["n"], "what": "problem 0 of the batch, all %d iterations, one core (oracle loop only)" % r["iters"]}
        except subprocess.TimeoutExpired:
            full_proc.kill()
            full = {"error": "not finished 900 s after the GPU part"}
    reps = []
    t_all = time.perf_counter()
    for _ in range(max(1, args.cpu_reps)):
        ps = [subprocess.Popen(cmd + ["--pid", str(i)], stdout=subprocess.PIPE, text=True) for i in range(procs)]
        res = []
        for p in ps:
            outp = p.communicate()[0]
            if p.returncode == 0:
                res.append(json.loads(outp.strip().splitlines()[-1]))
        if res:
            loop_s = max(r["seconds"] for r in res)
            rates = sorted(r["iters"] / r["seconds"] for r in res)
            reps.append({"value": sum(r["iters"] for r in res) / loop_s, "cores": len(res), "loop_s": loop_s, "iters": res[0]["iters"],
                         "single_core_median": rates[len(rates) // 2], "single_core_min": rates[0], "single_core_max": rates[-1]})
    wall = time.perf_counter() - t_all
    if not reps:
        return None
    reps.sort(key=lambda r: r["value"])
    m = reps[len(reps) // 2]
    # The SAME work as a GPU tree: every host core plans one problem of the batch for the full iteration count (the oracle's
    # per-iteration cost grows with the tree, so the truncated sample above flatters the CPU).  value = iterations of all processes /
    # wall time of the slowest; a process that is not done at the deadline stops there and counts with what it did.
    same = None
    if args.cpu_full_deadline > 0:
        t_f = time.perf_counter()
        cmd_f = [sys.executable, os.path.join(ROOT, "oracle", "cpu_bench.py"), "--algo", args.algo, "--dim", str(args.dim), "--world", args.world,
                 "--iters", str(args.iters), "--cap", str(args.iters), "--deadline-s", str(args.cpu_full_deadline)]
        ps = [subprocess.Popen(cmd_f + ["--pid", str(i)], stdout=subprocess.PIPE, text=True) for i in range(procs)]
        res = []
        for p in ps:
            try:
                outp = p.communicate(timeout=args.cpu_full_deadline + 300)[0]
            except subprocess.TimeoutExpired:
                p.kill()
                continue
            if p.returncode == 0:
                res.append(json.loads(outp.strip().splitlines()[-1]))
        if res:
            loop_s = max(r["seconds"] for r in res)
            rates = sorted(r["iters"] / r["seconds"] for r in res)
            same = {"value": sum(r["iters"] for r in res) / loop_s, "cores": len(res), "loop_s": loop_s, "iterations_each": args.iters,
                    "complete": int(sum(1 for r in res if r.get("complete"))), "deadline_s": args.cpu_full_deadline,
                    "iterations_done": int(sum(r["iters"] for r in res)),
                    "single_core_median": rates[len(rates) // 2], "single_core_min": rates[0], "single_core_max": rates[-1],
                    "wall_s": time.perf_counter() - t_f}
    head = same if same else m
    return {"value": head["value"], "unit": "iterations/s", "cores": head["cores"], "host_cores": os.cpu_count(), "cpu_model": cpu_model(),
            "kind": "port", "repetitions": len(reps),
            "value_is": ("%d processes (one per host core) x ALL %d iterations of problems 0..%d of the batch: %d finished within the %.0f s "
                         "deadline, the others count with the iterations they did; %d iterations in %.1f s (slowest process)"
                         % (same["cores"], args.iters, same["cores"] - 1, same["complete"], same["deadline_s"], same["iterations_done"], same["loop_s"]))
                        if same else "the truncated sample (see `sample`)",
            "same_work": same, "truncated_sample_value": m["value"],
            "values_of_repetitions": [r["value"] for r in reps],
            "single_core_median": head["single_core_median"], "single_core_min": head["single_core_min"], "single_core_max": head["single_core_max"],
            "full_run": full,
            # (the same numbers once more as plain scalars: a record parser that keeps only flat fields still sees them)
            "full_run_iterations_per_second": (full or {}).get("iterations_per_second"), "full_run_seconds": (full or {}).get("seconds"),
            "full_run_iterations": (full or {}).get("iterations"),
            "sample": "median of %d repetitions of: %d processes (one per host core) x first %d of %d iterations of problems 0..%d of the batch "
                      "(oracle loop only, %.1f s for the slowest process of the median repetition, %.1f s for everything incl. start-up); the "
                      "per-iteration cost grows with the tree, so the truncated sample flatters the CPU - full_run is one problem at full length"
                      % (len(reps), m["cores"], m["iters"], args.iters, m["cores"] - 1, m["loop_s"], wall)}


if __name__ == "__main__":
    main()
