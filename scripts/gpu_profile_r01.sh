#!/bin/bash
# Round-1 profile collection on the MI355X box (run through gpurun from the repo root).
set -x
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_r01b
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $OUT/bench_irrt.json 2> $OUT/bench_irrt.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o bench -- python $R/bench.py --no-cpu-baseline > $OUT/bench_irrt_profiled.json 2>> $OUT/bench_irrt.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o bench -- python $R/bench.py --no-cpu-baseline --steps 1 --warmup 0 > $OUT/pmc_fetch.json 2>> $OUT/bench_irrt.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o bench -- python $R/bench.py --no-cpu-baseline --steps 1 --warmup 0 > $OUT/pmc_write.json 2>> $OUT/bench_irrt.err
python $R/bench.py --algo rrt --no-cpu-baseline > $OUT/bench_rrt.json 2>> $OUT/bench_irrt.err
ls -R $OUT | head -40
# keep only small files
find $OUT -name '*.csv' -size +2M -exec sh -c 'head -c 200000 "$1" > "$1.head"; rm "$1"' _ {} \;
