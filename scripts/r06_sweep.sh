#!/bin/bash
# round 6: same-box A/B lines.  Each line of the here-doc: <label> | <env assignments> | <library file> | <bench.py args>
#   scripts/r06_sweep.sh <file with such lines>   -> gpurun_out/r06/<label>.json + one summary line each on stdout
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/gpurun_out/r06
while IFS='|' read -r label envs so args; do
  label=$(echo $label); [ -z "$label" ] && continue
  case "$label" in \#*) continue;; esac
  so=$(echo $so); [ -z "$so" ] && so=libnirrt_hip.so
  out=$R/gpurun_out/r06/$label.json
  env $envs NIRRT_HIP_SO=$R/nirrt_star_amd/$so timeout 900 python $R/bench.py --no-cpu-baseline --no-ttfs --no-secondary --steps 1 --warmup 0 $args > $out 2> $R/gpurun_out/r06/$label.err
  python3 - $out "$label" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    pi = r.get("per_iteration", {})
    print("%-28s %7.2f M it/s  kernel %7.0f ms  frac %.3f  per-tree %s" % (sys.argv[2], d["value"] / 1e6, r["kernel_ms"], r["frac"],
          {k: round(v, 2) for k, v in d["config"].get("per_tree_seconds", {}).items()}), flush=True)
    c = d["config"]
    if "forwards_per_step" in c:
        print("      launches %.0f forwards %.0f clouds %.0f kernel share %.2f host %s" % (c["launches_per_step"], c["forwards_per_step"], c["clouds_per_step"],
              r.get("kernel_share_of_step", 0), c.get("host_seconds_last_step")), flush=True)
    if pi:
        print("      visited %.0f members %.0f chain %.1f cand %.2f rewired %.2f recost %.1f list %.1f rebuilt %.1f" % (
            pi["visited_slots"], pi["near_members"], pi["chain_records"], pi["rewire_candidates"], pi["rewired"], pi["recosted"], pi["list_entries"], pi["rebuilt"]), flush=True)
except Exception as e:
    print(sys.argv[2], "FAILED", e, flush=True)
PY
done < "$1"
