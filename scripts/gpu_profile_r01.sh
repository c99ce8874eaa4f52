#!/bin/bash
# Round-1 profile collection on the MI355X box (run through gpurun from the repo root):
#   bench lines, rocprofv3 --kernel-trace --stats of the same command, and separate --pmc passes for HBM traffic.
set -x
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_r01i
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $OUT/bench_irrt.json 2> $OUT/err.log
python $R/bench.py --algo rrt > $OUT/bench_rrt.json 2>> $OUT/err.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o bench -- python $R/bench.py --no-cpu-baseline > $OUT/bench_irrt_profiled.json 2>> $OUT/err.log
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --kernel-include-regex "k_run_" --output-format csv -d $OUT/pmc_irrt_$C -o bench -- python $R/bench.py --no-cpu-baseline --steps 1 --warmup 0 > $OUT/pmc_irrt_$C.json 2>> $OUT/err.log
  rocprofv3 --pmc $C --kernel-trace --kernel-include-regex "k_run_" --output-format csv -d $OUT/pmc_rrt_$C -o bench -- python $R/bench.py --algo rrt --no-cpu-baseline --steps 1 --warmup 0 > $OUT/pmc_rrt_$C.json 2>> $OUT/err.log
done
find $OUT -name '*kernel_trace.csv' -size +1M -delete
ls -R $OUT | head -50
