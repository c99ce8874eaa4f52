#!/bin/bash
# A/B of library builds on the GPU box:  scripts/ab_bench.sh "<bench.py args>" <so> [<so> ...]   -> gpurun_out/ab/<so>.json
R=$(cd "$(dirname "$0")/.." && pwd)
args=$1; shift
mkdir -p $R/gpurun_out/ab
for so in "$@"; do
  n=$(basename $so .so)
  NIRRT_HIP_SO=$R/nirrt_star_amd/$so python $R/bench.py --no-cpu-baseline --no-ttfs --no-secondary --steps 1 --warmup 0 $args > $R/gpurun_out/ab/$n$(echo $args | tr -d ' -').json 2> $R/gpurun_out/ab/$n.err
  python3 - $R/gpurun_out/ab/$n$(echo $args | tr -d ' -').json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(sys.argv[1].split("/")[-1], "value %.2f M it/s  kernel %.0f ms  frac %.3f  per-tree %s" % (d["value"] / 1e6, r["kernel_ms"], r["frac"], d["config"]["per_tree_seconds"]))
    print("   ", {k: round(v, 2) for k, v in r["per_iteration"].items()})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
