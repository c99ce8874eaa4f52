#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02o
mkdir -p $OUT
cd $R
for CFG in "5 0.01 0.03" "5 0.0 0.0" "8 0.01 0.03" "5 0.02 0.06"; do
set -- $CFG
timeout 900 python bench.py --no-cpu-baseline --no-ttfs --steps 1 --warmup 0 --segments $1 --wide-frac $2 --narrow-frac $3 > $OUT/b.json 2> $OUT/b.err
python - <<PY
import json
d=json.load(open('/root/repo/gpurun_out/r02o/b.json'))
print("$CFG", d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['per_tree_seconds'])
PY
tail -2 $OUT/b.err
done
