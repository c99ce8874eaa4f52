#!/bin/bash
# Trimmed profile collection (GPU-minute budget): IRRT* bench line + rocprofv3 kernel stats + PMC traffic, RRT* bench line.
set -x
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_r01j
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $OUT/bench_irrt.json 2> $OUT/err.log
python $R/bench.py --algo rrt --steps 1 --cpu-budget-s 10 > $OUT/bench_rrt.json 2>> $OUT/err.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o bench -- python $R/bench.py --no-cpu-baseline --steps 1 > $OUT/bench_irrt_profiled.json 2>> $OUT/err.log
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --kernel-include-regex "k_run_" --output-format csv -d $OUT/pmc_irrt_$C -o bench -- python $R/bench.py --no-cpu-baseline --steps 1 --warmup 0 > $OUT/pmc_irrt_$C.json 2>> $OUT/err.log
done
find $OUT -name '*kernel_trace.csv' -size +1M -delete
ls -R $OUT | head -30
