#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02ab
mkdir -p $OUT
cd $R
timeout 900 python -X faulthandler bench.py --no-cpu-baseline --no-ttfs --steps 1 --warmup 0 --algo nirrt --trees 256 > $OUT/n256.json 2> $OUT/n256.err
echo "rc=$?"
tail -30 $OUT/n256.err | cut -c1-300
cat $OUT/n256.json | cut -c1-600
exit 0
