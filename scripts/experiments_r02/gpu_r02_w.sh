#!/bin/bash
# callee-saved saves gone (-fno-optimize-sibling-calls): parity, bench, traffic, rebuild-interval sweep
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02w
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_sampling.py tests/test_hip_edges.py tests/test_hip_grid.py tests/test_hip_variants.py tests/test_hip_fullsize.py -m gpu -x -q 2>&1 | tail -3
run() {   # name, env..., then bench args after --
  name=$1; shift
  env "$@" timeout 900 python bench.py --no-cpu-baseline --no-ttfs --steps 1 --warmup 0 > $OUT/$name.json 2> $OUT/$name.err
  python - <<PY
import json
try:
    d=json.load(open('$OUT/$name.json'))
    print("$name", d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config'].get('per_tree_seconds'), d['roofline']['per_iteration'].get('visited_slots'))
except Exception as e:
    print("$name failed", e)
PY
  tail -2 $OUT/$name.err | grep -v amdgpu.ids
}
run base NIRRT_DUMMY=1
run rebuild512 NIRRT_GRID_REBUILD=512
run rebuild256 NIRRT_GRID_REBUILD=256
run rebuild2048 NIRRT_GRID_REBUILD=2048
cd /tmp && export TMPDIR=/tmp
timeout 1200 python $R/scripts/collect_traffic.py 2>&1 | tail -3
