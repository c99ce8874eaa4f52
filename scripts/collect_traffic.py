#!/usr/bin/env python3
"""HBM-side traffic of the persistent kernel for ONE bench configuration, the way MI355X_MICROARCH.md prescribes:
separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; --kernel-trace only), kernel-filtered, then
    traffic_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024
(FETCH_SIZE is in KiB and on gfx950 reports half of the bytes of wide reads - doubled as the guide says; WRITE_SIZE is
uncalibrated).  Writes / updates profiles/<round>_traffic.json: one entry per configuration key (bench.py config_key) with the
raw counters, the date, the command and the kernel name, and copies the raw counter CSV rows next to it.

    python scripts/collect_traffic.py [bench.py arguments, e.g. --trees 8192 --algo irrt]      (on the GPU box)
"""
import csv
import datetime
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = os.environ.get("NIRRT_ROUND", "r05")   # prefix of the files written under profiles/
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    bench_args = sys.argv[1:]
    args = bench.parse(bench_args)
    key = bench.config_key(args)
    out_root = os.path.join(ROOT, "gpurun_out", "traffic_" + key)
    os.makedirs(out_root, exist_ok=True)
    prof_dir = os.path.join(ROOT, "profiles")
    raw = {}
    env = dict(os.environ, TMPDIR="/tmp")
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(out_root, counter)
        cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--kernel-include-regex", "k_run_", "--output-format", "csv",
               "-d", d, "-o", "bench", "--", sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-ttfs", "--no-secondary",
               "--steps", "1", "--warmup", "0"] + bench_args
        subprocess.run(cmd, cwd="/tmp", env=env, check=True, stdout=subprocess.DEVNULL)
        rows = []
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                rows += [r for r in csv.DictReader(fh) if r["Counter_Name"] == counter and "k_run_" in r["Kernel_Name"]]
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            os.remove(f)   # ~18 MB per pass (every copy / fill / init kernel of the set-up): gpurun merges back at most 64 MiB
        rows.sort(key=lambda r: -float(r["Counter_Value"]))
        # one step = several launches (segments x workgroup-size groups): the step's traffic is the sum over all of them
        raw[counter] = dict(rows[0], Counter_Value=str(sum(float(r["Counter_Value"]) for r in rows)), dispatches=str(len(rows)))
        with open(os.path.join(prof_dir, "%s_pmc_%s_%s.csv" % (ROUND, key, counter)), "w") as fh:
            keep = ["Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count",
                    "SGPR_Count", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"]
            w = csv.DictWriter(fh, fieldnames=keep, extrasaction="ignore")
            w.writeheader()
            w.writerows(rows)
    fetch, write = float(raw["FETCH_SIZE"]["Counter_Value"]), float(raw["WRITE_SIZE"]["Counter_Value"])
    # correction for THIS access pattern: scripts/calib/run_calib.sh measures FETCH_SIZE / WRITE_SIZE of known byte counts (short
    # rows of 32-byte records, scattered 32 / 64-byte records, scattered 8 / 32 / 64-byte stores); the factors below are
    # true bytes / counter bytes of the dominant read pattern (rows of slot records) and the dominant write pattern (32-byte
    # record stores), taken from profiles/r03_traffic_calibration.json when it exists (else the guide's 2 x FETCH, 1 x WRITE)
    f_fetch, f_write, calib_src = 2.0, 1.0, "MI355X_MICROARCH.md default (2 x FETCH_SIZE, WRITE_SIZE as is)"
    cpath = os.path.join(prof_dir, "r03_traffic_calibration.json")
    if os.path.exists(cpath):
        with open(cpath) as fh:
            cal = json.load(fh)
        if "factors" in cal:
            f_fetch, f_write = float(cal["factors"]["fetch"]), float(cal["factors"]["write"])
            calib_src = "profiles/r03_traffic_calibration.json"
    path = os.path.join(prof_dir, "%s_traffic.json" % ROUND)
    tab = {"formula": "traffic_bytes = (f_fetch * FETCH_SIZE_KiB + f_write * WRITE_SIZE_KiB) * 1024, factors per entry (calibrated on this access pattern)", "entries": {}}
    if os.path.exists(path):
        with open(path) as fh:
            tab = json.load(fh)
    tab["entries"][key] = {"FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write, "traffic_bytes": (f_fetch * fetch + f_write * write) * 1024,
                           "f_fetch": f_fetch, "f_write": f_write, "calibration": calib_src,
                           "kernel": "k_run_pool<%d> / k_run_sample<%d> (slim / narrow / wide instantiations)" % (args.dim, args.dim),
                           "dispatches_summed": int(raw["FETCH_SIZE"]["dispatches"]),
                           "collected": datetime.date.today().isoformat(),
                           "command": "rocprofv3 --pmc <C> --kernel-trace --kernel-include-regex k_run_ -- python bench.py "
                                      "--no-cpu-baseline --no-ttfs --steps 1 --warmup 0 " + " ".join(bench_args)}
    with open(path, "w") as fh:
        json.dump(tab, fh, indent=1)
    # gpurun merges back gpurun_out/ only: leave copies of what went into profiles/ there
    import shutil
    shutil.copy(path, out_root)
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        shutil.copy(os.path.join(prof_dir, "%s_pmc_%s_%s.csv" % (ROUND, key, counter)), out_root)
    print(json.dumps(tab["entries"][key]))


if __name__ == "__main__":
    main()
