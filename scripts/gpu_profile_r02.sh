#!/bin/bash
# Round-2 profile collection (tracked summaries go to profiles/ via scripts/install_profiles_r02.py):
#   1. default bench line (irrt_star random_2d, 8192 problems x 50 000 iterations) incl. TTFS + CPU baseline
#   2. rocprofv3 --kernel-trace --stats of the same command (1 step)
#   3. PMC traffic of the same configuration (scripts/collect_traffic.py: FETCH_SIZE / WRITE_SIZE passes)
#   4. SQ activity counters of the persistent kernel
#   5. rrt_star 2D, rrt_star / irrt_star 3D bench lines (+ kernel stats of the 3D IRRT* run)
#   6. PointNet++ forward: kernel stats at B = 1 and B = 256, MFMA-busy counters
#   7. NIRRT* batched bench line
set -x
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_r02
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 python $R/bench.py --steps 2 --warmup 1 > $OUT/bench_irrt2d.json 2> $OUT/bench_irrt2d.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_irrt2d -o bench -- python $R/bench.py --no-cpu-baseline --no-ttfs --steps 1 --warmup 0 > $OUT/bench_irrt2d_profiled.json 2>> $OUT/err.log
timeout 1200 python $R/scripts/collect_traffic.py > $OUT/traffic_irrt2d.json 2>> $OUT/err.log
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU --kernel-trace --kernel-include-regex "k_run_" --output-format csv -d $OUT/pmc_sq -o bench -- python $R/bench.py --no-cpu-baseline --no-ttfs --steps 1 --warmup 0 > /dev/null 2>> $OUT/err.log
timeout 900 python $R/bench.py --algo rrt --steps 1 --warmup 1 --cpu-iters 30000 > $OUT/bench_rrt2d.json 2>> $OUT/err.log
timeout 900 python $R/bench.py --dim 3 --algo rrt --steps 1 --warmup 0 --cpu-iters 30000 > $OUT/bench_rrt3d.json 2>> $OUT/err.log
timeout 900 python $R/bench.py --dim 3 --algo irrt --trees 4096 --steps 1 --warmup 0 --cpu-iters 10000 > $OUT/bench_irrt3d.json 2>> $OUT/err.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_irrt3d -o bench -- python $R/bench.py --dim 3 --algo irrt --trees 4096 --no-cpu-baseline --no-ttfs --steps 1 --warmup 0 > $OUT/bench_irrt3d_profiled.json 2>> $OUT/err.log
timeout 900 python $R/bench.py --world b30r16 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_irrt2d_b30r16.json 2>> $OUT/err.log
for B in 1 256; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_pn2_b$B -o pn2 -- python $R/scripts/pn2_forward_only.py $B 10 > $OUT/pn2_b$B.log 2>> $OUT/err.log
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --kernel-include-regex "k_sa_mlp|Cijk" --output-format csv -d $OUT/pmc_pn2 -o pn2 -- python $R/scripts/pn2_forward_only.py 256 5 > /dev/null 2>> $OUT/err.log
timeout 900 python $R/bench.py --algo nirrt --trees 1024 --iters 50000 --steps 1 --warmup 0 > $OUT/bench_nirrt2d.json 2>> $OUT/err.log
timeout 900 python $R/bench.py --algo nirrt --dim 3 --trees 512 --iters 50000 --steps 1 --warmup 0 > $OUT/bench_nirrt3d.json 2>> $OUT/err.log
find $OUT -name '*kernel_trace.csv' -size +1M -delete
find $OUT -name '*.db' -delete
ls -R $OUT | head -80
tail -5 $OUT/err.log
