#!/bin/bash
# guided lines with / without paced windows (NIRRT_BATCH_PACE):  gpurun -- scripts/nirrt_pace_ab.sh
cd $(dirname "$0")/..
run() { echo "== $1 pace $2"; NIRRT_BATCH_PACE=$2 python bench.py --no-cpu-baseline --no-ttfs --no-secondary --steps 1 --warmup 0 $3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']
print('value %.2f M  step %.0f ms kernel %.0f ms  launches %s host %s' % (d['value']/1e6, d['ms_per_step'], r['kernel_ms'], c['launches_per_step'], {k: v for k, v in c['host_seconds_last_step'].items() if k in ('wait_launch','refresh','candidates','classify')}))"; }
python -m pytest tests/test_nirrt_batch_gpu.py tests/test_batch_driver_gpu.py -m gpu -x -q 2>&1 | tail -2
for p in 0 1; do run nirrt_2d $p "--algo nirrt --trees 4096"; done
for p in 0 1; do run nirrt_c_2d $p "--algo nirrt_c --trees 2048"; done
for p in 0 1; do run nirrt_3d $p "--algo nirrt --dim 3 --trees 2048"; done
