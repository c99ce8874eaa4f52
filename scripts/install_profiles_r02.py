"""Copy the summaries of one round-2 profile collection (scripts/gpu_profile_r02.sh output, gpurun_out/prof_r02) into profiles/
(tracked) and print the numbers the docs quote:  python scripts/install_profiles_r02.py [gpurun_out/prof_r02]"""
import csv
import glob
import json
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "gpurun_out", "prof_r02")
dst = os.path.join(root, "profiles")


def last_json(path):
    with open(path) as f:
        lines = [ln for ln in f if ln.startswith("{")]
    return lines[-1] if lines else None


for name in sorted(glob.glob(os.path.join(src, "bench_*.json"))):
    ln = last_json(name)
    if not ln:
        print("no JSON line in", name)
        continue
    out = os.path.join(dst, "r02_" + os.path.basename(name))
    with open(out, "w") as f:
        f.write(ln)
    j = json.loads(ln)
    r = j.get("roofline", {})
    cb = j.get("cpu_baseline") or {}
    print("%-34s %8.3f M it/s  step %8.1f ms  kernel %8.1f ms  useful %7.1f GB/s (frac %.4f)  traffic %s  waste %s  cpu %s x%s"
          % (os.path.basename(name), j["value"] / 1e6, j["ms_per_step"], r.get("kernel_ms", 0), r.get("achieved", 0), r.get("frac", 0),
             ("%.1f TB" % (r["traffic"] / 1e12)) if r.get("traffic") else "-", ("%.2f" % r["wasted_traffic_ratio"]) if r.get("wasted_traffic_ratio") else "-",
             ("%.0f" % cb["value"]) if cb else "-", cb.get("cores", "-")))
for d in sorted(glob.glob(os.path.join(src, "kt_*"))):
    tag = os.path.basename(d)[3:]
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        rows = open(f).readlines()[:16]
        with open(os.path.join(dst, "r02_%s_kernel_stats.csv" % tag), "w") as g:
            g.writelines(rows)
        for k in list(csv.DictReader(rows))[:4]:
            print("  %s: %-60s calls %4s  avg %10.3f ms  %5s %%" % (tag, k["Name"][:60], k["Calls"], float(k["AverageNs"]) / 1e6, k["Percentage"]))
for d in ("pmc_sq", "pmc_pn2"):
    for f in glob.glob(os.path.join(src, d, "**", "*counter_collection.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
        agg = {}
        for r_ in rows:
            key = (r_["Kernel_Name"][:70], r_["Counter_Name"])
            agg[key] = agg.get(key, 0.0) + float(r_["Counter_Value"])
        with open(os.path.join(dst, "r02_%s_counters.csv" % d), "w") as g:
            g.write("kernel,counter,sum_over_dispatches\n")
            for (kn, cn), v in sorted(agg.items()):
                g.write('"%s",%s,%.0f\n' % (kn, cn, v))
        print("  %s: %d counter rows -> r02_%s_counters.csv" % (d, len(rows), d))
for f in glob.glob(os.path.join(src, "pn2_b*.log")):
    with open(f) as g:
        print(" ", os.path.basename(f), [ln.strip() for ln in g if ln.startswith("B=")])
