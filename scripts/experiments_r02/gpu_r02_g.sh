#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02g
mkdir -p $OUT
cd $R
timeout 900 python scripts/perf_outliers.py 4096 50000 > $OUT/outliers.log 2>&1
cat $OUT/outliers.log
