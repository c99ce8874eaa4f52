#!/bin/bash
# instruction-cache counters of the persistent kernel (own --pmc pass, kernel trace only), default line
R=$(cd "$(dirname "$0")/.." && pwd)
out=$R/gpurun_out/r06/icache; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --kernel-include-regex "k_run_" --output-format csv -d $out -o p -- python $R/bench.py --no-cpu-baseline --no-ttfs --no-secondary --steps 1 --warmup 0 "$@" > $out/bench.json 2> $out/bench.err
find $out -name "*kernel_trace.csv" -delete
python3 - $out <<'PY'
import csv, glob, sys
tot = {}
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_run_" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for k, v in sorted(tot.items()):
    print(k, "%.4g" % v)
PY
