#!/bin/bash
# scheduling of the stragglers now that they are visit-bound: lane hints after a pilot; more trees than 2x the slots
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02z
mkdir -p $OUT
cd $R
run() {
  name=$1; shift
  timeout 900 python bench.py --no-cpu-baseline --no-ttfs --steps 1 --warmup 0 "$@" > $OUT/$name.json 2> $OUT/$name.err
  python - <<PY
import json
try:
    d=json.load(open('$OUT/$name.json'))
    pi=d['roofline']['per_iteration']
    print("$name", d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config'].get('per_tree_seconds'))
except Exception as e:
    print("$name failed", e)
PY
  tail -2 $OUT/$name.err | grep -v amdgpu.ids
}
run pilot_n04 --pilot 5000 --wide-frac 0 --narrow-frac 0.04
run pilot_w01_n04 --pilot 5000 --wide-frac 0.01 --narrow-frac 0.04
run pilot_n10 --pilot 5000 --wide-frac 0 --narrow-frac 0.10
timeout 600 python scripts/perf_outliers.py 4096 50000 4096-8191 2>&1 | tail -5 | cut -c1-600
exit 0
