#!/bin/bash
# Near stash keeps only members above the straight-line floor of cost(new): parity + bench
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02y
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_sampling.py tests/test_hip_edges.py tests/test_hip_grid.py tests/test_hip_variants.py tests/test_hip_fullsize.py -m gpu -x -q 2>&1 | tail -3
run() {
  name=$1; shift
  timeout 900 python bench.py --no-cpu-baseline --no-ttfs --steps 1 --warmup 0 "$@" > $OUT/$name.json 2> $OUT/$name.err
  python - <<PY
import json
try:
    d=json.load(open('$OUT/$name.json'))
    pi=d['roofline']['per_iteration']
    print("$name", d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config'].get('per_tree_seconds'), pi.get('near_members'), pi.get('members_spilled'), pi.get('rewire_candidates'))
except Exception as e:
    print("$name failed", e)
PY
  tail -2 $OUT/$name.err | grep -v amdgpu.ids
}
run base
run rrt2d --algo rrt
