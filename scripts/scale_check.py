#!/usr/bin/env python3
"""Reads the lines scripts/scale_1248.sh collected (<dir>/{weak,strong}_N.json) and prints the scaling table; exit status 1 if a
check fails: a missing line, an N > 1 line that did not run on the RCCL process group ("nccl"), n_gpus != N, or - given a BENCH
record as second argument - an N = 1 weak value more than 2 % away from that record's value.  Dry-run lines (bench.py --dry-run:
launcher / process group / timing protocol only) are checked for rank count and work only."""
import json
import os
import sys


def last_json(path):
    with open(path) as f:
        lines = [l for l in f.read().splitlines() if l.startswith("{")]
    return json.loads(lines[-1]) if lines else None


def main():
    d = sys.argv[1]
    bench = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] else None
    ok = True
    rows = {}
    for mode in ("weak", "strong"):
        for n in (1, 2, 4, 8):
            p = os.path.join(d, "%s_%d.json" % (mode, n))
            if not os.path.exists(p):
                continue
            r = last_json(p)
            if r is None:
                print("%s N=%d: no JSON line" % (mode, n))
                ok = False
                continue
            rows[(mode, n)] = r
            if r.get("dry_run"):
                if r["n_gpus"] != n:
                    print("%s N=%d: %d ranks answered" % (mode, n, r["n_gpus"]))
                    ok = False
                continue
            if r["n_gpus"] != n:
                print("%s N=%d: the line says n_gpus = %d" % (mode, n, r["n_gpus"]))
                ok = False
            if n > 1 and r.get("process_group") != "nccl":
                print("%s N=%d: process group %r, expected 'nccl' (RCCL)" % (mode, n, r.get("process_group")))
                ok = False
    for mode in ("weak", "strong"):
        base = rows.get((mode, 1))
        for n in (1, 2, 4, 8):
            r = rows.get((mode, n))
            if r is None:
                continue
            if r.get("dry_run"):
                print("%-6s N=%d  dry run: %d ranks, work of all ranks %.3g" % (mode, n, r["n_gpus"], r["work_all_ranks"]))
                continue
            eff = r["value"] / (n * base["value"]) if (base and not base.get("dry_run")) else float("nan")
            print("%-6s N=%d  %8.2f M it/s  %7.1f ms/step  efficiency vs N=1 %.3f  setup %.1f s" % (
                mode, n, r["value"] / 1e6, r["ms_per_step"], eff, r.get("setup_seconds_max_over_ranks", float("nan"))))
    if bench and ("weak", 1) in rows and not rows[("weak", 1)].get("dry_run"):
        with open(bench) as f:
            b = json.load(f)
        ref = (b.get("parsed") or b)["value"]
        got = rows[("weak", 1)]["value"]
        dev = abs(got - ref) / ref
        print("N=1 weak %.2f M it/s vs %s %.2f M it/s: %.1f %% apart" % (got / 1e6, os.path.basename(bench), ref / 1e6, 100 * dev))
        if dev > 0.02:
            ok = False
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
