#!/bin/bash
# bench.py value/kernel_ms only: scripts/bench_short.sh [bench args]
python bench.py --no-cpu-baseline --steps 1 --warmup 1 "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('%s: %.3f M it/s, kernel %.0f ms, frac %.2f, streamed %.0f GB/s' % (j['config']['workload'][:24], j['value']/1e6, j['roofline']['kernel_ms'], j['roofline']['frac'], j['roofline']['streamed_GBps']))
"
