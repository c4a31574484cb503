"""Per-kernel mean of every PMC counter in a rocprofv3 rocpd .db (summed over counter instances per dispatch first).

    python tools/pmc_counters.py gpurun_out/x/sq_results.db [kernel-substring]
"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
filt = sys.argv[2] if len(sys.argv) > 2 else "lg::"
cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
disp = "dispatch_id" if "dispatch_id" in cols else ("id" if "id" in cols else "start")
rows = db.execute(f"select kernel_name, counter_name, {disp}, sum(value) from counters_collection group by kernel_name, counter_name, {disp}").fetchall()
acc = {}
for k, c, _d, v in rows:
    k = re.sub(r"\(.*$", "", k).replace("void ", "").strip()
    if filt not in k:
        continue
    acc.setdefault(k, {}).setdefault(c, []).append(float(v))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print(f"    {c:28s} mean {sum(v) / len(v):16.1f}   n={len(v)}")
