#!/bin/bash
# End-of-round run: the whole -m gpu suite, smoke(), then the profile collection of the final kernels
# (same layout as scripts/gpu_profile_r02.sh -> gpurun_out/prof_r02, installed by scripts/install_profiles_r02.py).
# PointNet++ profiles are not repeated (that code did not change since scripts/gpu_profile_r02.sh ran).
set -x
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_r02
rm -rf $OUT; mkdir -p $OUT
cd $R
if [ -z "$SKIP_TESTS" ]; then
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 > $OUT/pytest_gpu.txt
cat $OUT/pytest_gpu.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 > $OUT/smoke.txt
cat $OUT/smoke.txt
rm -rf $R/gpurun_out/test_* $R/gpurun_out/bench_ck    # checkpoints written by the suite: not results
fi
cd /tmp && export TMPDIR=/tmp
timeout 900 python $R/bench.py --steps 2 --warmup 1 > $OUT/bench_irrt2d.json 2> $OUT/bench_irrt2d.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_irrt2d -o bench -- python $R/bench.py --no-cpu-baseline --no-ttfs --steps 1 --warmup 0 > $OUT/bench_irrt2d_profiled.json 2>> $OUT/err.log
timeout 1200 python $R/scripts/collect_traffic.py > $OUT/traffic_irrt2d.json 2>> $OUT/err.log
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU --kernel-trace --kernel-include-regex "k_run_" --output-format csv -d $OUT/pmc_sq -o bench -- python $R/bench.py --no-cpu-baseline --no-ttfs --steps 1 --warmup 0 > /dev/null 2>> $OUT/err.log
timeout 900 python $R/bench.py --algo rrt --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_rrt2d.json 2>> $OUT/err.log
timeout 900 python $R/bench.py --dim 3 --algo rrt --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_rrt3d.json 2>> $OUT/err.log
timeout 900 python $R/bench.py --dim 3 --algo irrt --trees 4096 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_irrt3d.json 2>> $OUT/err.log
timeout 900 python $R/bench.py --world b30r16 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_irrt2d_b30r16.json 2>> $OUT/err.log
timeout 900 python $R/bench.py --algo nirrt --trees 4096 --iters 50000 --steps 1 --warmup 0 > $OUT/bench_nirrt2d.json 2>> $OUT/err.log
timeout 900 python $R/bench.py --algo nirrt --dim 3 --trees 512 --iters 50000 --steps 1 --warmup 0 > $OUT/bench_nirrt3d.json 2>> $OUT/err.log
find $OUT -name '*kernel_trace.csv' -size +1M -delete
find $OUT -name '*.db' -delete
find $R/gpurun_out -name '*kernel_trace.csv' -size +1M -delete
du -sh $R/gpurun_out
ls $OUT | head -40
tail -5 $OUT/err.log
exit 0
