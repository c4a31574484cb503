"""SQ counters of the tools/micro/valu_rate.hip launches (rocprofv3 rocpd .db) -> JSON: per (instruction class, waves per SIMD) the
wave-level instruction count, the VALU-active quad-cycles and the busy cycles, and the ratios bench.py's compute model uses.

    python tools/valu_rate_pmc.py /tmp/vr_pmc/.../vr_results.db > profiles/r03_valu_rate_pmc.json

The benchmark launches every k_rate<OP> four times in its `pmc` mode, in the order 1, 2, 4, 8 waves per SIMD."""
import json
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
disp = "dispatch_id" if "dispatch_id" in cols else ("id" if "id" in cols else "start")
rows = db.execute(f"select kernel_name, counter_name, {disp}, sum(value) from counters_collection group by kernel_name, counter_name, {disp} order by {disp}").fetchall()
per = {}
for k, c, d, v in rows:
    m = re.search(r"k_rate<(\d+)>", k)
    if not m:
        continue
    per.setdefault((int(m.group(1)), d), {})[c] = float(v)
out = {"note": "per launch; SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES count quad-cycles (MI355X_MICROARCH.md), SQ_BUSY_CYCLES is summed over the shader engines",
       "launches": []}
seen = {}
for (op, d) in sorted(per, key=lambda t: t[1]):
    c = per[(op, d)]
    i = seen.get(op, 0); seen[op] = i + 1
    wps = (1, 2, 4, 8)[i] if i < 4 else None
    e = {"op": op, "waves_per_simd": wps, **{k: round(v, 1) for k, v in c.items()}}
    if c.get("SQ_INSTS_VALU") and c.get("SQ_ACTIVE_INST_VALU"):
        e["active_quadcycles_per_valu_inst"] = round(c["SQ_ACTIVE_INST_VALU"] / c["SQ_INSTS_VALU"], 4)
    if c.get("SQ_WAVE_CYCLES") and c.get("SQ_INSTS_VALU"):
        e["wave_quadcycles_per_valu_inst"] = round(c["SQ_WAVE_CYCLES"] / c["SQ_INSTS_VALU"], 4)
    out["launches"].append(e)
print(json.dumps(out, indent=1))
