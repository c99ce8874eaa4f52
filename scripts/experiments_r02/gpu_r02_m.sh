#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02m
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_hip_edges.py tests/test_hip_parity.py tests/test_hip_sampling.py tests/test_hip_grid.py tests/test_hip_variants.py -m gpu -x -q 2>&1 | tail -3
bash scripts/gpu_sweep.sh m
timeout 900 python bench.py --no-cpu-baseline --no-ttfs --steps 1 --warmup 0 > $OUT/bench_8192.json 2> $OUT/bench_8192.err
python - <<'PY'
import json
d=json.load(open('/root/repo/gpurun_out/r02m/bench_8192.json'))
print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['config']['per_tree_seconds'])
PY
tail -3 $OUT/bench_8192.err
