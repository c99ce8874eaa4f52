#!/bin/bash
# A round's evidence, collected on the GPU box in ONE gpurun call (everything lands under gpurun_out/<round>/; copy what is wanted
# into profiles/):   NIRRT_ROUND=r05 gpurun --timeout 3600 -- scripts/gpu_profile.sh [stage ...]     stages: calib traffic stats sq pn2
R=$(cd "$(dirname "$0")/.." && pwd)
RD=${NIRRT_ROUND:-r05}
export NIRRT_ROUND=$RD
out=$R/gpurun_out/$RD
mkdir -p $out $R/profiles
stages=${@:-calib traffic stats sq pn2}
cd /tmp && export TMPDIR=/tmp
for st in $stages; do
case $st in
calib)
  $R/scripts/calib/run_calib.sh > $out/calib.txt 2>&1
  python3 - "$R" <<'PY'
import json, sys
R = sys.argv[1]
c = json.load(open(R + "/gpurun_out/calib/calib.json"))
# true bytes at sector granularity / counter bytes, for the dominant read pattern (rows of 32-byte slot records: 32 B per record
# are fetched although 28 are used) and the dominant write pattern (32-byte record stores)
rows, w32 = c["rows32"], c["w32"]
n_rows = rows["algorithmic_bytes"] / 28.0
f_fetch = 32.0 * n_rows / (rows["FETCH_SIZE_KiB"] * 1024)
f_write = w32["algorithmic_bytes"] / (w32["WRITE_SIZE_KiB"] * 1024)
json.dump({"patterns": c, "factors": {"fetch": f_fetch, "write": f_write},
           "how": "fetch = 32 B per visited slot record / FETCH_SIZE bytes of the rows32 pattern; write = 32 B per record / WRITE_SIZE bytes of the "
                  "w32 pattern (scripts/calib/calib_traffic.hip: known byte counts over a 16 GiB buffer)"},
          open(R + "/profiles/r03_traffic_calibration.json", "w"), indent=1)
print("factors", f_fetch, f_write)
PY
  cp $R/profiles/r03_traffic_calibration.json $out/ ;;
traffic)
  # FETCH / WRITE passes of the default line and of every secondary line of bench.py (one configuration key each)
  python3 - "$R" > $out/traffic_configs.txt <<'PY'
import sys
sys.path.insert(0, sys.argv[1])
import bench
print("")
for label, extra in bench.SECONDARY:
    print(" ".join(e for i, e in enumerate(extra) if e != "--ttfs" and e != "--warmup" and (i == 0 or extra[i - 1] != "--warmup")))
PY
  while IFS= read -r cfg; do
    python $R/scripts/collect_traffic.py $cfg > $out/traffic_$(echo $cfg | tr -d ' -').txt 2>&1
    tail -1 $out/traffic_$(echo $cfg | tr -d ' -').txt | cut -c1-200
  done < $out/traffic_configs.txt
  cp $R/profiles/${RD}_traffic.json $R/profiles/${RD}_pmc_*.csv $out/ 2>/dev/null ;;
stats)
  for cfg in "" "--algo irrt --dim 3 --trees 4096 --segments 3 --wide-visits 6000 --narrow-visits 2000"; do
    n=$(echo $cfg | tr -d ' -'); [ -z "$n" ] && n=irrt2d
    rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$n -o p -- python $R/bench.py --no-cpu-baseline --no-ttfs --no-secondary --steps 1 --warmup 0 $cfg > $out/bench_profiled_$n.json 2> $out/bench_profiled_$n.err
    find $out/stats_$n -name "*kernel_trace.csv" -delete
  done ;;
sq)
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU --kernel-trace --kernel-include-regex "k_run_" --output-format csv -d $out/sq -o p -- python $R/bench.py --no-cpu-baseline --no-ttfs --no-secondary --steps 1 --warmup 0 > $out/bench_sq.json 2> $out/bench_sq.err
  rocprofv3 --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --kernel-include-regex "k_run_" --output-format csv -d $out/mem -o p -- python $R/bench.py --no-cpu-baseline --no-ttfs --no-secondary --steps 1 --warmup 0 > $out/bench_mem.json 2> $out/bench_mem.err
  find $out/sq $out/mem -name "*kernel_trace.csv" -delete ;;
pn2)
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/pn2_b256 -o p -- python $R/scripts/pn2_forward_only.py 256 > $out/pn2_b256.txt 2>&1
  find $out/pn2_b256 -name "*kernel_trace.csv" -delete
  python $R/scripts/sa_mlp_bench.py 256 > $out/sa_mlp_bench.txt 2>&1
  $R/scripts/pmc_pn2.sh $RD > $out/pn2_pmc.txt 2>&1
  cp $R/gpurun_out/pmc_pn2_$RD/summary.txt $out/pn2_pmc_summary.txt
  ;;
esac
done
ls -R $out | head -60
