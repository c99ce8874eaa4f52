#!/bin/bash
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/gpurun_out/r06
for park in 0.25 0; do
  NIRRT_BATCH_PARK=$park timeout 600 python $R/bench.py --no-cpu-baseline --no-ttfs --no-secondary --steps 1 --warmup 0 --algo nirrt --dim 3 --trees 512 --iters 20000 --pc-update-cost-ratio 1.0 > $R/gpurun_out/r06/dbg_park_$park.json 2> $R/gpurun_out/r06/dbg_park_$park.err
  echo "park $park rc $?"; tail -c 600 $R/gpurun_out/r06/dbg_park_$park.json; tail -5 $R/gpurun_out/r06/dbg_park_$park.err
done
