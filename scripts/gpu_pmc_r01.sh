#!/bin/bash
# HBM traffic of the persistent kernel: separate --pmc passes (FETCH_SIZE, WRITE_SIZE), kernel-filtered.
set -x
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_r01c
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --kernel-include-regex "k_run_" --output-format csv -d $OUT/pmc_$C -o bench -- python $R/bench.py --no-cpu-baseline --steps 1 --warmup 0 > $OUT/pmc_$C.json 2> $OUT/pmc_$C.err
  rocprofv3 --pmc $C --kernel-trace --kernel-include-regex "k_run_" --output-format csv -d $OUT/pmc_rrt_$C -o bench -- python $R/bench.py --algo rrt --no-cpu-baseline --steps 1 --warmup 0 > $OUT/pmc_rrt_$C.json 2> $OUT/pmc_rrt_$C.err
done
ls -la $OUT/*
