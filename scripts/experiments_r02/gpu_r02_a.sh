#!/bin/bash
# round 2, call A: the new variant parity tests + SQ counters of the default bench (is the loop ALU- or latency-bound?)
set -x
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02a
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests/test_hip_variants.py tests/test_hip_fullsize.py -m gpu -x -q > $OUT/pytest_variants.log 2>&1
tail -5 $OUT/pytest_variants.log
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU --kernel-trace --kernel-include-regex "k_run_" --output-format csv -d $OUT/pmc_sq -o bench -- python $R/bench.py --no-cpu-baseline --steps 1 --warmup 0 > $OUT/pmc_sq.json 2> $OUT/pmc_sq.err
tail -3 $OUT/pmc_sq.err
find $OUT -name '*kernel_trace.csv' -size +1M -delete
find $OUT -name '*counter_collection.csv' | head; 
for f in $(find $OUT -name '*counter_collection.csv'); do head -20 $f; done
