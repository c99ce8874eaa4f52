#!/bin/bash
# 3D IRRT* line under several grid resolutions / schedules:  gpurun -- scripts/grid3d_sweep.sh
cd $(dirname "$0")/..
B="python bench.py --no-cpu-baseline --no-ttfs --no-secondary --steps 1 --warmup 0 --algo irrt --dim 3 --trees 4096 --wide-visits 6000 --narrow-visits 2000"
run() { echo "== G=$1 G2=$2 $3"
  NIRRT_GRID_G=$1 NIRRT_GRID_G2=$2 $B $3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']
print('value %.2f M  kernel %.0f ms  per-tree %s  wide %s narrow %s visited %.0f brute %.3f' % (d['value']/1e6, r['kernel_ms'], c['per_tree_seconds'], c['trees_on_256_lanes'], c['trees_on_128_lanes'], r['per_iteration']['visited_slots'], r['per_iteration']['whole_tree_visits']))"; }
run 32 8 "--segments 3 --free-lanes 256"
run 32 8 "--segments 5"
run 32 8 "--segments 5 --free-lanes 256"
run 40 8 "--segments 3"
run 32 8 "--segments 4 --wide-visits 4000 --narrow-visits 1500 --free-lanes 256"
