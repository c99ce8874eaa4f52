#!/bin/bash
# One command from a node with 8 MI355X to a scaling record (the driver's own SCALE run needs nothing of this; it is for whoever has
# the node):   scripts/scale_1248.sh [--dry-run] [extra bench.py arguments]
#   weak line   : python bench.py --gpus N                      (8192 problems per GPU, BASELINE config 2)
#   strong line : python bench.py --gpus N --scaling strong --problems 8192   (ONE fixed set of 8192 problems sharded over the ranks)
# for N = 1, 2, 4, 8 -> gpurun_out/scale/{weak,strong}_N.json, then scripts/scale_check.py: per-N value, efficiency vs N = 1, that the
# N = 1 weak value equals a given BENCH line within 2 % (SCALE_BENCH=<file>), and that every N > 1 line ran on the RCCL process group.
R=$(cd "$(dirname "$0")/.." && pwd)
out=$R/gpurun_out/scale
mkdir -p $out
dry=""
if [ "$1" = "--dry-run" ]; then dry="--dry-run"; shift; fi
for n in ${SCALE_NS:-1 2 4 8}; do
  for mode in weak strong; do
    extra=""
    [ $mode = strong ] && extra="--scaling strong --problems ${SCALE_PROBLEMS:-8192}"
    python $R/bench.py --gpus $n --steps ${SCALE_STEPS:-2} --warmup ${SCALE_WARMUP:-1} --no-cpu-baseline --no-secondary --no-ttfs $dry $extra "$@" \
      > $out/${mode}_$n.json 2> $out/${mode}_$n.err || echo "bench.py --gpus $n ($mode) failed: see $out/${mode}_$n.err"
  done
done
python $R/scripts/scale_check.py $out ${SCALE_BENCH:-}
