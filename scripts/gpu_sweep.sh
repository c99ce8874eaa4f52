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
run u2 libnirrt_hip_u2.so
run u8 libnirrt_hip_u8.so
run g256 libnirrt_hip.so NIRRT_GRID_G=256
run rb512 libnirrt_hip.so NIRRT_GRID_REBUILD=512
run g256rb512 libnirrt_hip.so NIRRT_GRID_G=256 NIRRT_GRID_REBUILD=512
grep "per iteration" $OUT/base.log $OUT/g256.log $OUT/rb512.log
