"""Copy one profile collection (scripts/gpu_profile_r01.sh output under gpurun_out/<dir>) into profiles/ and print the
numbers the docs quote:  python scripts/install_profiles.py gpurun_out/prof_r01h"""
import csv
import json
import os
import shutil
import sys

src = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "profiles")


def line(path):
    with open(path) as f:
        return [l for l in f if l.startswith("{")][-1]


for name, out in (("bench_irrt.json", "r01_bench_irrt2d.json"), ("bench_rrt.json", "r01_bench_rrt2d.json"),
                  ("bench_irrt_profiled.json", "r01_bench_irrt2d_profiled_run.json")):
    l = line(os.path.join(src, name))
    with open(os.path.join(dst, out), "w") as f:
        f.write(l)
    j = json.loads(l)
    r = j["roofline"]
    print("%s: %.3f M it/s, kernel %.1f ms, algorithmic %.1f GB/s (frac %.3f), visited %.1f GB/s, cpu %s" % (
        name, j["value"] / 1e6, r["kernel_ms"], r["achieved"], r["frac"], r["streamed_GBps"], j.get("cpu_baseline", {}).get("value")))
with open(os.path.join(src, "kt", "bench_kernel_stats.csv")) as f:
    rows = f.readlines()[:12]
with open(os.path.join(dst, "r01_bench_irrt2d_kernel_stats.csv"), "w") as f:
    f.writelines(rows)
k = next(csv.DictReader(rows))
tot, mn, mx, calls = float(k["TotalDurationNs"]) / 1e6, float(k["MinNs"]) / 1e6, float(k["MaxNs"]) / 1e6, int(k["Calls"])
print("kernel stats (%s): %d calls, total %.1f ms, average %.1f, max %.1f, min %.1f" % (k["Name"][:40], calls, tot, tot / calls, mx, mn))
traffic = json.load(open(os.path.join(dst, "r01_traffic.json")))
for algo in ("irrt", "rrt"):
    if not os.path.exists(os.path.join(src, "pmc_%s_FETCH_SIZE" % algo)):
        continue
    vals = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        p = os.path.join(src, "pmc_%s_%s" % (algo, c), "bench_counter_collection.csv")
        shutil.copy(p, os.path.join(dst, "r01_pmc_%s2d_%s.csv" % (algo, c)))
        vals[c] = float(next(csv.DictReader(open(p)))["Counter_Value"])
    key = "%s_2d_4096x50000" % algo
    traffic[key] = {"FETCH_SIZE_KiB": vals["FETCH_SIZE"], "WRITE_SIZE_KiB": vals["WRITE_SIZE"],
                    "traffic_bytes": (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024}
    print("%s traffic %.2f TB" % (algo, traffic[key]["traffic_bytes"] / 1e12))
json.dump(traffic, open(os.path.join(dst, "r01_traffic.json"), "w"), indent=1)
