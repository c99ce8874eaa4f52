#!/bin/bash
# PMC passes over PointNet++ forwards (k_sa_mlp only): MFMA pipe busy, LDS bank conflicts, wave/wait cycles.
#   scripts/pmc_pn2.sh <tag> [B]    -> gpurun_out/pmc_pn2_<tag>/summary.txt
tag=$1; B=${2:-256}
R=$(cd "$(dirname "$0")/.." && pwd)
out=$R/gpurun_out/pmc_pn2_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
while read -r counters; do
  [ -z "$counters" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $counters --kernel-trace --kernel-include-regex "k_sa_mlp" --output-format csv -d $out/pass$i -o p -- \
     python $R/scripts/pn2_forward_only.py $B 2 > $out/pass$i.txt 2> $out/pass$i.err
  find $out/pass$i -name "*kernel_trace.csv" -delete
done <<'LIST'
SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE
SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
LIST
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
tot = collections.OrderedDict()
for f in sorted(glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        key = (r["Kernel_Name"][:32], r["Grid_Size"], r["LDS_Block_Size"] if "LDS_Block_Size" in r else "", r["Counter_Name"])
        tot.setdefault(key, []).append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as fh:
    for k, v in tot.items():
        fh.write("%s n=%d mean %.6g\n" % (" ".join(map(str, k)), len(v), sum(v) / len(v)))
print(open(out + "/summary.txt").read())
PY
