#!/bin/bash
set -x
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02f
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_hip_edges.py tests/test_hip_parity.py tests/test_hip_sampling.py tests/test_hip_grid.py tests/test_hip_variants.py -m gpu -x -q 2>&1 | tail -5
NIRRT_HIP_SO=$R/nirrt_star_amd/libnirrt_hip_prof.so timeout 600 python scripts/perf_irrt.py 4096 50000 2 irrt 14 > $OUT/phases_4096.log 2>&1
tail -4 $OUT/phases_4096.log
timeout 900 python bench.py --no-cpu-baseline --steps 1 --warmup 0 --no-ttfs > $OUT/bench_4096.json 2> $OUT/bench_4096.err
cat $OUT/bench_4096.json
timeout 900 python bench.py --no-cpu-baseline --steps 1 --warmup 0 --no-ttfs --trees 8192 > $OUT/bench_8192.json 2> $OUT/bench_8192.err
cat $OUT/bench_8192.json
