cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-ttfs --no-secondary --steps 1 --warmup 0 --scaling strong --problems 1000"
run() { echo "== $1"; shift; env "$@" $B $EXTRA 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']
print('value %.2f M  kernel %.0f ms  per-tree %s  wide %s narrow %s' % (d['value']/1e6, r['kernel_ms'], c['per_tree_seconds'], c['trees_on_256_lanes'], c['trees_on_128_lanes']))"; }
EXTRA="" run "auto (wide)" X=1
EXTRA="" run "all narrow" NIRRT_FORCE_VARIANT=narrow
EXTRA="" run "all slim" NIRRT_FORCE_VARIANT=slim
EXTRA="--free-lanes 256" run "free 256 / rest slim" NIRRT_SLIM_MIN_TREES=1
EXTRA="--free-lanes 256" run "free 256 / rest narrow" NIRRT_WIDE_MAX_TREES=0
EXTRA="--free-lanes 128" run "free 128 / rest slim" NIRRT_SLIM_MIN_TREES=1
