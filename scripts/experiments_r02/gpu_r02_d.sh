#!/bin/bash
set -x
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02d
mkdir -p $OUT
cd $R
timeout 300 python -m pytest tests/test_hip_edges.py -m gpu -x -q 2>&1 | tail -5
timeout 600 python scripts/perf_irrt.py 64 20000 2 irrt 14 > $OUT/stats_64.log 2>&1
cat $OUT/stats_64.log
NIRRT_HIP_SO=$R/nirrt_star_amd/libnirrt_hip_prof.so timeout 600 python scripts/perf_irrt.py 64 20000 2 irrt 14 > $OUT/phases_64.log 2>&1
tail -3 $OUT/phases_64.log
