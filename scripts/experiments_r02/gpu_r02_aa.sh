#!/bin/bash
# NIRRT* batch: generator look-ahead kept resident (no per-launch regeneration / upload)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02aa
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_nirrt_batch_gpu.py tests/test_batch_driver_gpu.py tests/test_planners_gpu.py -m gpu -x -q 2>&1 | tail -3
run() {
  name=$1; shift
  timeout 900 python bench.py --no-cpu-baseline --no-ttfs --steps 1 --warmup 0 "$@" > $OUT/$name.json 2> $OUT/$name.err
  python - <<PY
import json
try:
    d=json.load(open('$OUT/$name.json'))
    print("$name", d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline'].get('kernel_share_of_step'), d['config'].get('launches_per_step'), d['config'].get('clouds_per_step'))
except Exception as e:
    print("$name failed", e)
PY
  tail -2 $OUT/$name.err | grep -v amdgpu.ids
}
run nirrt2d_1024 --algo nirrt --trees 1024
run nirrt2d_4096 --algo nirrt --trees 4096
exit 0
