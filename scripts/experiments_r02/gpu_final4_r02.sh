#!/bin/bash
# default bench line once more, now that profiles/r02_traffic.json holds the PMC pass of the final kernels
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_r02
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 200 python $R/bench.py --steps 1 --warmup 1 > $OUT/bench_irrt2d.json 2> $OUT/bench_irrt2d.err
tail -2 $OUT/bench_irrt2d.err
cut -c1-300 $OUT/bench_irrt2d.json
exit 0
