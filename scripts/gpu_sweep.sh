#!/bin/bash
# tuning sweep of the slim kernels at the bench configuration: library builds (NIRRT_HIP_SO) x index knobs (environment)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/sweep_$1
mkdir -p $OUT
cd $R
run() {  # name, so, env...
  name=$1; so=$2; shift 2
  env NIRRT_HIP_SO=$R/nirrt_star_amd/$so "$@" timeout 600 python scripts/perf_irrt.py ${TREES:-4096} 50000 2 irrt 14 > $OUT/$name.log 2>&1
  echo "== $name: $(grep -o 'kernel [0-9.]* ms' $OUT/$name.log) | $(grep 'per-tree seconds' $OUT/$name.log)"
}
run base libnirrt_hip.so
run prof libnirrt_hip_prof.so
tail -1 $OUT/prof.log
