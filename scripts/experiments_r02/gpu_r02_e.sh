#!/bin/bash
set -x
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02e
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_hip_edges.py tests/test_hip_parity.py tests/test_hip_sampling.py tests/test_hip_grid.py -m gpu -x -q 2>&1 | tail -5
timeout 600 python scripts/perf_irrt.py 64 20000 2 irrt 14 > $OUT/stats_64.log 2>&1
tail -3 $OUT/stats_64.log
NIRRT_HIP_SO=$R/nirrt_star_amd/libnirrt_hip_prof.so timeout 600 python scripts/perf_irrt.py 4096 50000 2 irrt 14 > $OUT/phases_4096.log 2>&1
tail -4 $OUT/phases_4096.log
timeout 900 python bench.py --no-cpu-baseline --steps 1 --warmup 0 > $OUT/bench_4096.json 2> $OUT/bench_4096.err
cat $OUT/bench_4096.json
