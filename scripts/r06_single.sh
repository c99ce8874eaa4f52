#!/bin/bash
# single-problem latency / time to first solution under two builds (same box)
R=$(cd "$(dirname "$0")/.." && pwd)
for so in libnirrt_hip.so libnirrt_hip_walk.so libnirrt_hip.so libnirrt_hip_walk.so; do
  NIRRT_HIP_SO=$R/nirrt_star_amd/$so python $R/bench.py --no-cpu-baseline --no-secondary --steps 1 --warmup 0 --trees 256 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$so', 'single %.3f s' % d['single_tree']['median_seconds'], 'ttfs single %.2f ms batch %.2f ms' % (d['time_to_first_solution']['single']['median_seconds']*1e3, d['time_to_first_solution']['batch']['median_seconds']*1e3), '256 trees %.2f M it/s' % (d['value']/1e6))"
done
