#!/bin/bash
# round 2, call C: parity suite on the fused-query kernels (fail fast), then a bench line
set -x
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02c
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_sampling.py tests/test_hip_grid.py -m gpu -x -q > $OUT/pytest_core.log 2>&1
tail -15 $OUT/pytest_core.log
timeout 1500 python -m pytest tests/test_hip_variants.py tests/test_hip_fullsize.py tests/test_hip_edges.py tests/test_planners_gpu.py -m gpu -x -q > $OUT/pytest_more.log 2>&1
tail -15 $OUT/pytest_more.log
timeout 600 python bench.py --no-cpu-baseline --steps 1 --warmup 0 > $OUT/bench_4096.json 2> $OUT/bench_4096.err
cat $OUT/bench_4096.json
