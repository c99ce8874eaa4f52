#!/bin/bash
# guided lines under several launch windows (NIRRT_BATCH_WINDOW: iterations per persistent launch):  gpurun -- scripts/nirrt_window_sweep.sh
cd $(dirname "$0")/..
run() { echo "== $1 window $2"; NIRRT_BATCH_WINDOW=$2 python bench.py --no-cpu-baseline --no-ttfs --no-secondary --steps 1 --warmup 0 $3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']
print('value %.2f M  step %.0f ms kernel %.0f ms  launches %s host %s' % (d['value']/1e6, d['ms_per_step'], r['kernel_ms'], c['launches_per_step'], {k: v for k, v in c['host_seconds_last_step'].items() if k in ('wait_launch','refresh','candidates','classify')}))"; }
for w in 1024 2048 4096; do run nirrt_2d $w "--algo nirrt --trees 4096"; done
for w in 512 2048; do run nirrt_c_2d $w "--algo nirrt_c --trees 2048"; done
