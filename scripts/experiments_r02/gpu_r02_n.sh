#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02n
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_hip_variants.py -m gpu -x -q -k "lanes_hint or more_than_2048" 2>&1 | tail -3
for L in 256 128 0; do
timeout 900 python bench.py --no-cpu-baseline --no-ttfs --steps 1 --warmup 0 --heavy-lanes $L > $OUT/bench_h$L.json 2> $OUT/bench_h$L.err
python - <<PY
import json
d=json.load(open('/root/repo/gpurun_out/r02n/bench_h$L.json'))
print($L, d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['wide_workgroup_trees'], d['config']['per_tree_seconds'])
PY
done
