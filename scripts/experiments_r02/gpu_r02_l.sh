#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02l
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_pointnet2.py tests/test_nirrt_batch_gpu.py -m gpu -x -q 2>&1 | tail -8
for B in 1 16 256; do timeout 300 python scripts/pn2_forward_only.py $B 10 2>&1 | tail -1; done
timeout 900 python bench.py --no-cpu-baseline --no-ttfs --steps 1 --warmup 0 > $OUT/bench_8192.json 2> $OUT/bench_8192.err
python - <<'PY'
import json
d=json.load(open('/root/repo/gpurun_out/r02l/bench_8192.json'))
print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['config']['per_tree_seconds'])
PY
