#!/bin/bash
# slots per lane and trip of the 256-lane kernels (GRID_U_WIDE builds) on the lines that run them:  gpurun -- scripts/wide_u_ab.sh <so> ...
cd $(dirname "$0")/..
for so in "$@"; do
  export NIRRT_HIP_SO=$PWD/nirrt_star_amd/$so
  echo "=== $so"
  python bench.py --no-cpu-baseline --no-secondary --steps 1 --warmup 0 --scaling strong --problems 1000 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('config 5: %.2f M  per-tree %s  single tree %.3f s  ttfs single %.2f ms' % (d['value']/1e6, c['per_tree_seconds'], d['single_tree']['median_seconds'], d['time_to_first_solution']['single']['median_seconds']*1e3))"
  python bench.py --no-cpu-baseline --no-ttfs --no-secondary --steps 1 --warmup 0 --algo irrt --dim 3 --trees 4096 --segments 3 --wide-visits 6000 --narrow-visits 2000 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('irrt_3d: %.2f M  kernel %.0f ms per-tree %s' % (d['value']/1e6, d['roofline']['kernel_ms'], c['per_tree_seconds']))"
done
