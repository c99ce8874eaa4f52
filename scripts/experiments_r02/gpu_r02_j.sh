#!/bin/bash
# the GPU tests written since the kernel rewrite + a small end-to-end bench line (all code paths) + NIRRT bench + PN++ timing
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02j
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests/test_guidance_fixtures.py tests/test_pointnet2.py tests/test_batch_driver_gpu.py tests/test_planners_gpu.py tests/test_nirrt_batch_gpu.py -m gpu -q > $OUT/pytest.log 2>&1
tail -40 $OUT/pytest.log
timeout 600 python bench.py --trees 512 --iters 5000 --steps 1 --warmup 0 --cpu-iters 3000 --cpu-procs 8 > $OUT/bench_small.json 2> $OUT/bench_small.err
cat $OUT/bench_small.json; tail -3 $OUT/bench_small.err
timeout 900 python bench.py --algo nirrt --trees 256 --iters 10000 --steps 1 --warmup 0 > $OUT/bench_nirrt.json 2> $OUT/bench_nirrt.err
cat $OUT/bench_nirrt.json; tail -3 $OUT/bench_nirrt.err
for B in 1 16 256; do timeout 300 python scripts/pn2_forward_only.py $B 10; done
