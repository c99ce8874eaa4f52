#!/bin/bash
# r in [16, 24] world with the larger SampleFree budget (no tree may run out of generator words)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_r02
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 280 python $R/bench.py --world b30r16 --steps 1 --warmup 0 --no-cpu-baseline --no-ttfs > $OUT/bench_irrt2d_b30r16.json 2> $OUT/b30r16.err
tail -2 $OUT/b30r16.err
cut -c1-400 $OUT/bench_irrt2d_b30r16.json
exit 0
