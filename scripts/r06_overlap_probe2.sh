#!/bin/bash
R=$(cd "$(dirname "$0")/.." && pwd)
$R/scripts/r06_fps_probe.sh
O=$R/gpurun_out/r06/overlap; mkdir -p $O
run() {
  label=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 600 python $R/bench.py --no-cpu-baseline --no-ttfs --no-secondary --steps 1 --warmup 0 "$@" > $O/$label.json 2> $O/$label.err
  rc=$?
  python3 - $O/$label.json "$label" $rc <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    c, r = d["config"], d["roofline"]
    print("%-26s rc %s %7.2f M it/s  step %.1f s kernel %.1f s launches %.0f forwards %.0f clouds %.0f host %s" % (sys.argv[2], sys.argv[3], d["value"] / 1e6,
          d["ms_per_step"] / 1e3, r["kernel_ms"] / 1e3, c["launches_per_step"], c["forwards_per_step"], c["clouds_per_step"], c.get("host_seconds_last_step")), flush=True)
except Exception as e:
    print(sys.argv[2], "rc", sys.argv[3], "FAILED", e, flush=True)
PY
  if [ $rc -ne 0 ]; then tail -5 $O/$label.err; echo "stopping after $label"; exit 1; fi
}
C4="--algo nirrt --dim 3 --pc-update-cost-ratio 1.0"
run seq_full NIRRT_BATCH_OVERLAP=0 -- $C4 --trees 2048
run ov_full NIRRT_BATCH_OVERLAP=1 -- $C4 --trees 2048
run ov_full_p50 NIRRT_BATCH_OVERLAP=1 NIRRT_BATCH_PARK=0.5 -- $C4 --trees 2048
run seq_c4_09 NIRRT_BATCH_OVERLAP=0 -- --algo nirrt --dim 3 --trees 2048
run seq_c3 NIRRT_BATCH_OVERLAP=0 -- --algo nirrt_c --trees 2048
run ov_c3 NIRRT_BATCH_OVERLAP=1 -- --algo nirrt_c --trees 2048
run seq_n2 NIRRT_BATCH_OVERLAP=0 -- --algo nirrt --trees 4096
run ov_n2 NIRRT_BATCH_OVERLAP=1 -- --algo nirrt --trees 4096
