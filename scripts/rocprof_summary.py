#!/usr/bin/env python3
"""Dump the per-kernel summary (rocprofv3 --kernel-trace --stats) out of a rocpd sqlite file.
usage: python scripts/rocprof_summary.py gpurun_out/prof_r01/bench_results.db > profiles/<name>.md"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
print("| kernel | calls | total ms | avg ms | % |")
print("|---|---:|---:|---:|---:|")
for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print("| `%s` | %d | %.3f | %.4f | %.2f |" % (name, calls, total / 1e3, avg / 1e3, pct))
print()
print("| kernel | grid | workgroup | VGPRs | SGPRs | LDS B | scratch B |")
print("|---|---:|---:|---:|---:|---:|---:|")
seen = set()
for row in cur.execute("select name,grid_x,workgroup_x,vgpr_count,sgpr_count,lds_size,scratch_size from kernels"):
    if row[0] in seen or row[0].startswith("__amd"):
        continue
    seen.add(row[0])
    print("| `%s` | %d | %d | %d | %d | %d | %d |" % row)
