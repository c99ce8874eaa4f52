"""Per-function summary of the device ISA (`hipcc --cuda-device-only -S`): instructions, scratch stores / loads (static sites),
registers, for the functions whose demangled name contains one of the patterns.

    python scripts/isa_summary.py /tmp/isa/nirrt_hip.s slim:: '<2'
"""
import re
import subprocess
import sys


def main():
    path, pats = sys.argv[1], sys.argv[2:]
    funcs = {}
    cur = None
    meta = {}
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
            funcs[cur] = {"ins": 0, "st": 0, "ld": 0, "call": 0, "wl": 0, "rl": 0}
            continue
        if cur is None:
            continue
        s = line.strip()
        if s.startswith(".Lfunc_end"):
            cur = None
            continue
        if not s or s.startswith((".", ";")) or s.endswith(":"):
            m = re.match(r";\s*(NumVgprs|NumSgprs|ScratchSize|Occupancy|NumAgprs|TotalNumVgprs):\s*(\d+)", s)
            continue
        f = funcs[cur]
        f["ins"] += 1
        if s.startswith("scratch_store"):
            f["st"] += 1
        elif s.startswith("scratch_load"):
            f["ld"] += 1
        elif s.startswith("s_swappc"):
            f["call"] += 1
        elif s.startswith("v_writelane"):
            f["wl"] += 1
        elif s.startswith("v_readlane"):
            f["rl"] += 1
    # register metadata comes after each function as comments: second pass
    cur = None
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
            continue
        m = re.match(r"^;\s*(NumVgprs|NumSgprs|ScratchSize|Occupancy|NumAgprs|TotalNumVgprs):\s*(\d+)", line)
        if m and cur in funcs:
            funcs[cur].setdefault(m.group(1), int(m.group(2)))
    names = list(funcs)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    print("%-90s %7s %5s %5s %5s %5s %5s %5s %5s %7s" % ("function", "ins", "st", "ld", "call", "wlane", "rlane", "vgpr", "sgpr", "scratch"))
    for n, d in zip(names, dem):
        if all(p in d for p in pats):
            f = funcs[n]
            print("%-90s %7d %5d %5d %5d %5d %5d %5s %5s %7s" % (d[:90], f["ins"], f["st"], f["ld"], f["call"], f["wl"], f["rl"],
                                                             f.get("NumVgprs", "?"), f.get("NumSgprs", "?"), f.get("ScratchSize", "?")))


if __name__ == "__main__":
    main()
