#!/usr/bin/env python3
"""bench.py — RRT*-family iterations/s while growing 50k-node trees on random_2d (BASELINE.json).

Workload (BASELINE.json configs[1]): `irrt_star random_2d, 50k iters`, batched on one MI355X:
B independent planning problems (224x224 world, 30 circle obstacles, clearance 3, step_len 10), each
planned for `--iters` iterations (default 50 000) by the device-resident loop - ONE persistent
workgroup per tree; sampling (SampleFree, then informed once a solution exists), nearest, steer,
collision, Near, choose-parent, rewire, goal bookkeeping and best-solution tracking all happen in the
kernel.  `--algo rrt` runs plain RRT* on the same problems (uniform sampling, no solution tracking).

One "step" = one pass of that loop over the whole batch (B x iters iterations), starting from fresh
one-vertex trees.  Inputs (the raw MT19937 outputs of each problem's seeded numpy / python generators)
are resident in HBM before the timed region.  N GPUs = N processes (torch.distributed / RCCL), each
with its own B problems (weak scaling); the only collectives are the timing protocol's barrier /
max-reduce and a gather of per-rank iteration counts.

Prints ONE JSON line (rank 0): the driver's contract fields plus
  roofline     — HBM roofline of the persistent kernel: algorithmic bytes (vertices the REFERENCE algorithm's nearest
                 + Near scans touch x dim x 8 B, counted exactly in the kernel) / HIP-event kernel time; the kernel
                 itself answers both through a grid index and visits far fewer (streamed_GBps, traffic)
  cpu_baseline — the oracle (oracle/nirrt_oracle.c, C port of the reference loop incl. sampling) on this
                 box's host, 1 core, on problem 0 of the same batch for a bounded time
  time_to_first_solution — iterations / seconds until c_best first becomes finite (median over the batch)
"""
import argparse
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s peak (6.3 TB/s achievable)
REF_PY = {"rrt": 153.0, "irrt": 43.0}  # reference numpy path in the survey container: context only


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--trees", type=int, default=4096, help="problems per GPU per step (4096 = 16 one-wave workgroups per CU, all resident)")
    ap.add_argument("--iters", type=int, default=50000, help="planner iterations per problem (tree capacity)")
    ap.add_argument("--dim", type=int, default=2)
    ap.add_argument("--algo", default="irrt", choices=["irrt", "rrt"])
    ap.add_argument("--world", default="b30")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget-s", type=float, default=25.0)
    ap.add_argument("--traffic-gb", type=float, default=None,
                    help="HBM GB per launch from a separate rocprofv3 --pmc run (default: profiles/r01_traffic.json if it has this config)")
    return ap.parse_args()


def make_problems(args, rank):
    """B problems for this rank: worlds 0..249 x 4 start/goal pairs = the 1000-problem evaluation set of
    SURVEY.md §8d (wrapping around for larger batches); planner seed = 1000 + problem id."""
    from nirrt_star_amd import worlds
    probs, cache = [], {}
    for b in range(args.trees):
        pid = rank * args.trees + b
        if args.dim == 2:
            w = pid % 250
            if w not in cache:
                cache[w] = worlds.random_world_2d(w, args.world)
            pr = worlds.problem_2d(cache[w], (pid // 250) % 4)
            pr["clearance"] = 3
        else:
            np.random.seed(pid)
            pr = worlds.problem_3d(worlds.random_world_3d(pid % 1000))
            pr["clearance"] = 2
        pr["pid"] = pid
        probs.append(pr)
    return probs


def word_budgets(args):
    D, it = args.dim, args.iters
    if args.algo == "rrt":
        return it * D * 2 * 2 + 4096, 0
    if D == 2:
        return it * 6 + 4096, it * 14 + 4096     # SampleFree until the first solution, then python-random unit disk
    return it * 6 * 40 + 4096, 0                 # 3D informed sampling stays on the numpy stream


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from nirrt_star_amd import _hip, build, sampling
    build.build()

    probs = make_problems(args, rank)
    D, B, iters = args.dim, args.trees, args.iters
    flags = _hip.F_IRRT if args.algo == "irrt" else 0
    trees = []
    for pr in probs:
        t = _hip.HipTree(D, iters, pr["x_start"], pr["x_goal"], 10, pr["search_radius"], pr["clearance"], pr["env"],
                         device_id=local_rank)
        t.set_informed(*sampling.informed_frame(pr["x_start"], pr["x_goal"]))
        trees.append(t)
    # inputs: each problem's generator outputs (np.random.seed(s); random.seed(s)), resident in HBM
    n_np, n_py = word_budgets(args)
    dev = "cuda:%d" % local_rank
    py_stride = max(n_py, 1)
    d_np = torch.empty((B, n_np), dtype=torch.int32, device=dev)
    d_py = torch.empty((B, py_stride), dtype=torch.int32, device=dev)
    cpu_np = cpu_py = None
    CH = 256   # generated and uploaded in chunks: the host never holds more than ~1 GB of the ~16 GB of words
    for c0 in range(0, B, CH):
        c1 = min(B, c0 + CH)
        h_np = np.empty((c1 - c0, n_np), dtype=np.uint32)
        h_py = np.empty((c1 - c0, py_stride), dtype=np.uint32)
        for b in range(c0, c1):
            np.random.seed(1000 + probs[b]["pid"])
            random.seed(1000 + probs[b]["pid"])
            h_np[b - c0] = sampling.peek_np_words(n_np)
            if n_py:
                h_py[b - c0] = sampling.peek_py_words(n_py)
        if c0 == 0:   # problem 0's words also feed the CPU baseline
            cpu_np, cpu_py = h_np[0].copy(), (h_py[0].copy() if n_py else None)
        d_np[c0:c1].copy_(torch.from_numpy(h_np.view(np.int32)))
        d_py[c0:c1].copy_(torch.from_numpy(h_py.view(np.int32)))
    np_tab = [(d_np.data_ptr() + 4 * n_np * b, n_np) for b in range(B)]
    py_tab = [(d_py.data_ptr() + 4 * py_stride * b, n_py) for b in range(B)] if n_py else None
    torch.cuda.synchronize()
    del h_np, h_py

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def one_step(want_trace=False):
        for t in trees:
            t.reset()
        return _hip.run_sampling(trees, iters, np_tab, py_tab, flags=flags, want_trace=want_trace, on_device=True)

    for _ in range(args.warmup):
        one_step()
    kernel_ms, scan_elems, alg_elems, done_iters = [], [], [], []
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        r = one_step()
        kernel_ms.append(r["kernel_ms"])
        scan_elems.append(int(r["scan_elems"].sum()))
        alg_elems.append(int(r["alg_elems"].sum()))
        done_iters.append(int(r["iters_done"].sum()))
    barrier()
    elapsed = time.perf_counter() - t0
    n_final = [t.n for t in trees]
    n_sol = [len(trees[b].solutions) for b in range(0, B, max(1, B // 16))]
    short = int((r["iters_done"] < iters).sum())

    tot = torch.tensor([elapsed, float(sum(done_iters))], dtype=torch.float64, device="cuda")
    if world > 1:
        tmax = tot.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        elapsed_max, total_iters = float(tmax[0].item()), float(tot[1].item())
    else:
        elapsed_max, total_iters = elapsed, float(sum(done_iters))
    value = total_iters / elapsed_max

    if rank == 0:
        k_ms = float(np.mean(kernel_ms))
        # algorithmic bytes: the reference algorithm's two O(n) coordinate passes per iteration
        # (nearest_neighbor + find_near_neighbors), n*D*8 bytes each, counted exactly by the kernel.  The kernel
        # itself streams less: the Near pass of iteration k also answers iteration k+1's nearest query.
        alg_bytes = float(np.mean(alg_elems)) * D * 8.0
        streamed_bytes = float(np.mean(scan_elems)) * D * 8.0
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        # time to first solution (untimed extra pass over a slice of the batch, with the per-iteration trace)
        sub = list(range(0, B, max(1, B // 64)))
        for b in sub:
            trees[b].reset()
        tr = _hip.run_sampling([trees[b] for b in sub], min(iters, 5000), [np_tab[b] for b in sub],
                               [py_tab[b] for b in sub] if py_tab else None, flags=flags | _hip.F_GOAL_SCAN * (args.algo == "rrt"),
                               want_trace=True, on_device=True)
        first = np.array([int(np.argmax(np.isfinite(c))) + 1 if np.isfinite(c).any() else -1 for c in tr["cost_trace"]])
        found = first[first > 0]
        ttfs_it = float(np.median(found)) if len(found) else None
        ttfs_s = (ttfs_it / min(iters, 5000)) * tr["kernel_ms"] * 1e-3 if ttfs_it else None   # resident loop, batch running concurrently
        out = {
            "metric": "RRT* iters/sec (50k-node tree), random_%dd" % D,
            "value": value, "unit": "iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed_max / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s_star random_%dd (%s: 224x224, 30 circle obstacles), %d problems/GPU x %d iters, "
                                   "device-resident batched loop with in-kernel sampling" % (args.algo, D, args.world, B, iters)
                       if D == 2 else "%s_star random_3d, %d problems/GPU x %d iters" % (args.algo, B, iters),
                       "trees_per_gpu": B, "iters": iters, "dim": D, "step_len": 10, "clearance": probs[0]["clearance"],
                       "mean_final_vertices": float(np.mean(n_final)), "mean_solutions_per_tree": float(np.mean(n_sol)),
                       "trees_stopped_early": short,
                       "per_tree_iters_per_s": iters / (k_ms * 1e-3)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": measured_traffic(args),
                         "kernel": "k_run_sample<%d>" % D, "kernel_ms": k_ms, "algorithmic_bytes_per_launch": alg_bytes,
                         "scan_bytes_streamed_per_launch": streamed_bytes,
                         "streamed_GBps": streamed_bytes / (k_ms * 1e-3) / 1e9},
            "time_to_first_solution": {"median_iterations": ttfs_it, "median_seconds_in_batch": ttfs_s,
                                       "problems": len(sub), "solved_within_%d" % min(iters, 5000): int(len(found))},
            "reference_python_survey_container_its": REF_PY[args.algo],
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, probs[0], cpu_np, cpu_py)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def measured_traffic(args):
    """HBM bytes per launch from the PMC passes committed under profiles/ (collected separately, as the
    profiling guide prescribes); None for configurations that were not profiled."""
    if args.traffic_gb:
        return args.traffic_gb * 1e9
    try:
        with open(os.path.join(ROOT, "profiles", "r01_traffic.json")) as f:
            tab = json.load(f)
        return tab["%s_%dd_%dx%d" % (args.algo, args.dim, args.trees, args.iters)]["traffic_bytes"]
    except Exception:
        return None


def cpu_baseline(args, pr, npw, pyw):
    """The oracle (C port of the reference loop, incl. sampling and the reference's un-cached
    find_best_path_solution) on problem 0 of this very batch: 1 host core, at most --cpu-budget-s."""
    from nirrt_star_amd import sampling
    from oracle import oracle as orc
    orc.build()
    o = orc.OracleTree(args.dim, args.iters, pr["x_start"], pr["x_goal"], 10, pr["search_radius"], pr["clearance"], pr["env_dict"])
    frame = sampling.informed_frame(pr["x_start"], pr["x_goal"])
    done, np_pos, py_pos, t0 = 0, 0, 0, time.perf_counter()
    chunk = 1000
    while done < args.iters and time.perf_counter() - t0 < args.cpu_budget_s:
        r = o.run_sampling(min(chunk, args.iters - done), npw[np_pos:], pyw[py_pos:] if pyw is not None else None,
                           irrt=args.algo == "irrt", frame=frame)
        if r["iters_done"] == 0:
            break
        done += r["iters_done"]
        np_pos += r["np_used"]
        py_pos += r["py_used"]
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "iterations/s", "cores": 1, "kind": "port",
            "sample": "problem 0 of the batch, first %d of %d iterations (tree grown to %d vertices, %d solutions) in %.1f s; "
                      "the CPU rate falls as the tree grows, so a truncated sample flatters the CPU"
                      % (done, args.iters, o.n, len(o.solutions), dt)}


if __name__ == "__main__":
    main()
