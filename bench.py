#!/usr/bin/env python3
"""bench.py — RRT*-family iterations/s while growing 50k-node trees on random_2d (BASELINE.json).

One "step" = one pass of the hot path over one batch: B independent planning problems
(224x224 world, 30 circle obstacles, clearance 3, step_len 10), each grown for `--iters`
iterations (default 50 000) by the device-resident loop (one persistent workgroup per tree).
N GPUs = N processes (torch.distributed / RCCL), each with its own B problems (weak scaling); the
only collective is the barrier / max-time reduction of the timing protocol and a gather of the
per-rank iteration counts.

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  roofline     — HBM roofline of the persistent kernel: algorithmic bytes (vertices streamed by the
                 nearest + Near passes x dim x 8 B, counted exactly inside the kernel) / HIP-event time
  cpu_baseline — the oracle (oracle/nirrt_oracle.c, a C port of the reference loop) timed on this
                 box's host cores on ONE of the same problems (bounded sample)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s peak (6.3 TB/s achievable)
REF_CONTAINER_ITS = 153.0  # reference numpy path, 50k iters 2D, survey container - context only


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--trees", type=int, default=1024, help="problems per GPU per step")
    ap.add_argument("--iters", type=int, default=50000, help="planner iterations per problem (tree capacity)")
    ap.add_argument("--dim", type=int, default=2)
    ap.add_argument("--algo", default="rrt", choices=["rrt"])
    ap.add_argument("--world", default="b30")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget-s", type=float, default=25.0)
    return ap.parse_args()


def make_problems(args, rank):
    """B problems for this rank: world seeds are global problem ids, planner
    seeds 1000+i drive SampleFree (rrt_base_2d.py:46-52) exactly like the reference would."""
    from nirrt_star_amd import worlds
    probs = []
    cache = {}
    for b in range(args.trees):
        pid = rank * args.trees + b
        if args.dim == 2:
            if pid % 250 not in cache:
                cache[pid % 250] = worlds.random_world_2d(pid % 250, args.world)
            ed = cache[pid % 250]
            pr = worlds.problem_2d(ed, (pid // 250) % 4)
            pr["clearance"] = 3
        else:
            ed = worlds.random_world_3d(pid % 1000)
            np.random.seed(pid)
            pr = worlds.problem_3d(ed)
            pr["clearance"] = 2
        pr["pid"] = pid
        probs.append(pr)
    return probs


def sample_free_sequence(pr, dim, iters, seed, inside_fn):
    """The node_rand sequence RRT*'s SampleFree produces for np.random.seed(seed): uniform draws in
    the clearance-shrunk range, x then y [then z] per attempt, rejected while inside an inflated
    obstacle.  Vectorised in blocks - the legacy stream is identical to scalar draws."""
    rs = np.random.RandomState(seed)
    c = pr["clearance"]
    env = pr["env"]
    lo = np.array([env.x_range[0] + c, env.y_range[0] + c] + ([env.z_range[0] + c] if dim == 3 else []), dtype=np.float64)
    hi = np.array([env.x_range[1] - c, env.y_range[1] - c] + ([env.z_range[1] - c] if dim == 3 else []), dtype=np.float64)
    out = np.zeros((iters, dim))
    got = 0
    while got < iters:
        m = int((iters - got) * 1.6) + 64
        u = rs.random_sample(m * dim).reshape(m, dim)
        cand = lo + (hi - lo) * u
        keep = cand[~inside_fn(cand)]
        take = min(len(keep), iters - got)
        out[got:got + take] = keep[:take]
        got += take
    return out


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

    from nirrt_star_amd import _hip, build
    build.build()

    probs = make_problems(args, rank)
    D, B, iters = args.dim, args.trees, args.iters
    trees = [_hip.HipTree(D, iters, pr["x_start"], pr["x_goal"], 10, pr["search_radius"], pr["clearance"], pr["env"],
                          device_id=local_rank) for pr in probs]
    # inputs: pre-drawn SampleFree sequences, resident in HBM before the timed region
    samples = np.zeros((B, iters, D))
    for b, (pr, t) in enumerate(zip(probs, trees)):
        samples[b] = sample_free_sequence(pr, D, iters, 1000 + pr["pid"], lambda p, t=t: t.points_in_obs(p)[0].astype(bool))
    d_samples = torch.from_numpy(samples).to("cuda:%d" % local_rank)
    torch.cuda.synchronize()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def one_step():
        for t in trees:
            t.reset()
        return _hip.run_replay(trees, None, flags=0, device_ptr=d_samples.data_ptr(), iters=iters)

    for _ in range(args.warmup):
        one_step()
    kernel_ms, scan_elems, n_final = [], [], []
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        r = one_step()
        kernel_ms.append(r["kernel_ms"])
        scan_elems.append(int(r["scan_elems"].sum()))
        assert not r["status"].any() and (r["iters_done"] == iters).all()
    barrier()
    elapsed = time.perf_counter() - t0
    n_final = [t.n for t in trees]

    tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed_max = float(tmax.item())
    total_iters = world * args.steps * B * iters
    value = total_iters / elapsed_max

    if rank == 0:
        k_ms = float(np.mean(kernel_ms))
        alg_bytes = float(np.mean(scan_elems)) * D * 8.0
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        out = {
            "metric": "RRT* iters/sec (50k-node tree), random_2d" if D == 2 else "RRT* iters/sec (50k-node tree), random_3d",
            "value": value, "unit": "iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed_max / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s_star random_%dd %s, %d problems/GPU x %d iters, device-resident batched loop"
                                   % (args.algo, D, args.world, B, iters),
                       "trees_per_gpu": B, "iters": iters, "dim": D, "step_len": 10, "obstacles": "30 circles r in [8,12]"
                       if args.world == "b30" else args.world,
                       "mean_final_vertices": float(np.mean(n_final)),
                       "per_tree_iters_per_s": B * iters / (k_ms * 1e-3) / B},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "k_run_replay<%d>" % D, "kernel_ms": k_ms, "algorithmic_bytes_per_launch": alg_bytes},
            "reference_python_survey_container_its": REF_CONTAINER_ITS,
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, probs[0], samples[0])
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(args, pr, samples):
    """The oracle (C port of the reference loop) on problem 0 of this very batch, 1 host core,
    for at most --cpu-budget-s seconds (chunks of 2500 iterations)."""
    from oracle import oracle as orc
    orc.build()
    o = orc.OracleTree(args.dim, args.iters, pr["x_start"], pr["x_goal"], 10, pr["search_radius"], pr["clearance"], pr["env_dict"])
    done, t0 = 0, time.perf_counter()
    while done < args.iters and time.perf_counter() - t0 < args.cpu_budget_s:
        o.replay(samples[done:done + 2500], False)
        done += 2500
    dt = time.perf_counter() - t0
    done = min(done, args.iters)
    return {"value": done / dt, "unit": "iterations/s", "cores": 1, "kind": "port",
            "sample": "problem 0 of the batch, first %d of %d iterations (tree grown to %d vertices) in %.1f s; "
                      "the CPU rate falls as the tree grows, so a truncated sample flatters the CPU" % (done, args.iters, o.n, dt)}


if __name__ == "__main__":
    main()
