#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02q
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_hip_sampling.py tests/test_hip_edges.py -m gpu -x -q 2>&1 | tail -3
for CFG in "5000 0.01 0.03" "5000 0.02 0.05" "0 0 0" "2500 0.01 0.03"; do
set -- $CFG
timeout 900 python bench.py --no-cpu-baseline --no-ttfs --steps 1 --warmup 0 --pilot $1 --wide-frac $2 --narrow-frac $3 > $OUT/b.json 2> $OUT/b.err
python - <<PY
import json
d=json.load(open('/root/repo/gpurun_out/r02q/b.json'))
print("$CFG", d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['per_tree_seconds'])
PY
tail -2 $OUT/b.err | grep -v amdgpu.ids
done
