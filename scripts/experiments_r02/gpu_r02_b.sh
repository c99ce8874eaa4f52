#!/bin/bash
# round 2, call B: oversubscription experiment (8192 problems on 4096 wave slots) + per-phase ticks of the slim kernels
set -x
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02b
mkdir -p $OUT
cd $R
timeout 600 python bench.py --no-cpu-baseline --steps 1 --warmup 0 --trees 8192 > $OUT/bench_8192.json 2> $OUT/bench_8192.err
cat $OUT/bench_8192.json
NIRRT_HIP_SO=$R/nirrt_star_amd/libnirrt_hip_prof.so timeout 900 python scripts/perf_irrt.py 4096 50000 2 irrt 14 > $OUT/phases_slim_4096.log 2>&1
tail -4 $OUT/phases_slim_4096.log
