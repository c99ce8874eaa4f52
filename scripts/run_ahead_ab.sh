cd $GRAFT_REPO_ROOT
python -m pytest tests/test_time_sliced_gpu.py -m gpu -x -q 2>&1 | tail -2
echo "== b30 run_ahead 1"; python scripts/perf_tail.py --world b30 2>&1 | grep -A3 "^kernel"
echo "== default run_ahead 0/1"; RUN_AHEAD=0 python scripts/perf_tail.py 2>&1 | grep -A2 "^kernel"; python scripts/perf_tail.py 2>&1 | grep -A2 "^kernel"
B="python bench.py --no-cpu-baseline --no-ttfs --no-secondary --steps 1 --warmup 0 --algo irrt --dim 3 --trees 4096 --segments 3 --wide-visits 6000 --narrow-visits 2000"
for ra in 0 1; do echo "== 3D run_ahead $ra"; $B --run-ahead $ra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']
print('value %.2f M  kernel %.0f ms  per-tree %s' % (d['value']/1e6, r['kernel_ms'], c['per_tree_seconds']))"; done
