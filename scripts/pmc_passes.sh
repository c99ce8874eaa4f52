#!/bin/bash
# PMC passes over the default bench (separate runs, --kernel-trace only, as the profiling guide prescribes).
#   scripts/pmc_passes.sh <tag> [bench args...]      -> gpurun_out/pmc_<tag>/pass*/ + gpurun_out/pmc_<tag>/summary.txt
tag=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
out=$R/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
while read -r counters; do
  [ -z "$counters" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $counters --kernel-trace --kernel-include-regex "k_run_" --output-format csv -d $out/pass$i -o p -- \
     python $R/bench.py --no-cpu-baseline --no-ttfs --no-secondary --steps 1 --warmup 0 "$@" > $out/pass$i.json 2> $out/pass$i.err
  find $out/pass$i -name "*kernel_trace.csv" -delete
done <<'LIST'
TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum
TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum
SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU
TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum TCC_BUSY_sum TCC_REQ_sum
SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_INSTS_SMEM SQ_WAIT_INST_LDS
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
LIST
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
tot = collections.OrderedDict()
for f in sorted(glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "k_run_" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
with open(out + "/summary.txt", "w") as fh:
    for k, v in tot.items():
        fh.write("%s %.6g\n" % (k, v))
print(open(out + "/summary.txt").read())
PY
