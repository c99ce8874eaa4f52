#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02h
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_hip_edges.py tests/test_hip_parity.py tests/test_hip_sampling.py tests/test_hip_grid.py tests/test_hip_variants.py -m gpu -x -q 2>&1 | tail -3
timeout 900 python scripts/perf_outliers.py 4096 50000 > $OUT/outliers.log 2>&1
cat $OUT/outliers.log
NIRRT_HIP_SO=$R/nirrt_star_amd/libnirrt_hip_prof.so timeout 600 python scripts/perf_irrt.py 4096 50000 2 irrt 14 > $OUT/phases_4096.log 2>&1
tail -2 $OUT/phases_4096.log
bash scripts/gpu_sweep.sh h
