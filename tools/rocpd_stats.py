"""Summarise a rocprofv3 rocpd .db (kernel trace) as the familiar kernel_stats table.

    python tools/rocpd_stats.py gpurun_out/prof_r1/bench_results.db > profiles/r01_bench_kernel_stats.csv
"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by {name_col} order by 3 desc").fetchall()
tot = sum(r[2] for r in rows) or 1
print("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs")
for n, c, t, a, mn, mx in rows:
    print(f"\"{n}\",{c},{t},{a:.1f},{100.0 * t / tot:.2f},{mn},{mx}")
