#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the calibration patterns (separate PMC passes, --kernel-trace only) next to their known byte counts.
#   scripts/calib/run_calib.sh  -> gpurun_out/calib/calib.json   (copy to profiles/r03_traffic_calibration.json)
R=$(cd "$(dirname "$0")/../.." && pwd)
out=$R/gpurun_out/calib
mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $out/calib_traffic $R/scripts/calib/calib_traffic.hip || exit 1
cd /tmp && export TMPDIR=/tmp
for pat in rows32 rec32 rec64 word4 w8 w32 w64; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/$pat-$c -o p -- $out/calib_traffic $pat 16 2000 > $out/$pat-$c.json 2> $out/$pat-$c.err
  done
done
python3 - "$out" <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
res = {}
for pat in ["rows32", "rec32", "rec64", "word4", "w8", "w32", "w64"]:
    e = json.loads(open("%s/%s-FETCH_SIZE.json" % (out, pat)).read().strip().splitlines()[-1])
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        v = 0.0
        for f in glob.glob("%s/%s-%s/**/*counter_collection.csv" % (out, pat, c), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == c and r["Kernel_Name"].startswith("void k<"):
                    v += float(r["Counter_Value"])
        e[c + "_KiB"] = v
    alg = e["algorithmic_bytes"]
    e["fetch_bytes_over_algorithmic"] = e["FETCH_SIZE_KiB"] * 1024 / alg
    e["write_bytes_over_algorithmic"] = e["WRITE_SIZE_KiB"] * 1024 / alg
    res[pat] = e
json.dump(res, open(out + "/calib.json", "w"), indent=1)
for k, e in res.items():
    print("%-7s alg %.3e B  FETCH %.3e B (x%.2f)  WRITE %.3e B (x%.2f)  %.0f GB/s" % (k, e["algorithmic_bytes"], e["FETCH_SIZE_KiB"] * 1024,
          e["fetch_bytes_over_algorithmic"], e["WRITE_SIZE_KiB"] * 1024, e["write_bytes_over_algorithmic"], e["GBps"]))
PY
