#!/bin/bash
# NIRRT* 2D, 4096 trees: launch grouping / window sweep.  usage: scripts/nirrt_groups_sweep.sh "G:I:W" ...  (groups:in-flight:window)
for cfg in "$@"; do IFS=: read G I W <<< "$cfg"
NIRRT_BATCH_GROUPS=$G NIRRT_BATCH_INFLIGHT=$I NIRRT_BATCH_WINDOW=$W python bench.py --algo nirrt --trees 4096 --steps 1 --warmup 0 --no-cpu-baseline --no-ttfs --no-secondary > gpurun_out/nx_$G_$I_$W.json 2> gpurun_out/nx_$G_$I_$W.err
python - <<PY
import json
d=json.load(open("gpurun_out/nx_$G_$I_$W.json"))
c=d["config"]
print("groups $G inflight $I window $W:", round(d["value"]/1e6,2), "M it/s step", round(d["ms_per_step"]/1e3,2), "s kernel", round(d["roofline"]["kernel_ms"]/1e3,2), "launches", c.get("launches_per_step"), c.get("host_seconds_last_step"))
PY
done
