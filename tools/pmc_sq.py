"""Per-kernel mean of the SQ counters of one or more rocprofv3 PMC passes (rocpd .db files), as JSON for bench.py.

    python tools/pmc_sq.py <workload> "<command that was profiled>" <pass1.db> [<pass2.db> ...] > profiles/rNN_x_pmc_sq_<workload>.json

Values of one dispatch are summed over counter instances (XCDs / SEs) first, then averaged over the dispatches of a kernel.
Derived per launch where the inputs exist:
  waves_per_simd_resident = 4 * SQ_WAVE_CYCLES / SQ_BUSY_CYCLES / 32
      (SQ_WAVE_CYCLES counts quad-cycles of resident waves, SQ_BUSY_CYCLES is summed over the 32 shader engines of 32 SIMDs each:
       checked on k_gaussian_backward, a 256-thread streaming kernel, which comes out at 3.5 waves per SIMD; the first round-2
       profiles (r02_a, r02_e) were written with a formula that gave twice this)
  wait_frac               = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES             (share of wave cycles spent in s_waitcnt)
"""
import json
import re
import sqlite3
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lidar-gs_amd"))
import build_hip  # noqa: E402  (build_id / box_id: what bench.py checks a looked-up profile against)

workload, command, dbs = sys.argv[1], sys.argv[2], sys.argv[3:]
acc = {}
for path in dbs:
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
    disp = "dispatch_id" if "dispatch_id" in cols else ("id" if "id" in cols else "start")
    rows = db.execute(f"select kernel_name, counter_name, {disp}, sum(value) from counters_collection group by kernel_name, counter_name, {disp}").fetchall()
    for k, c, _d, v in rows:
        k = re.sub(r"\(.*$", "", k).replace("void ", "").strip()
        if not k.startswith("lg::"):
            continue
        acc.setdefault(k, {}).setdefault(c, []).append(float(v))
res = {"workload": workload, "command": command, "build_id": build_hip.build_id(), "box": build_hip.box_id(),
       "note": "rocprofv3 --kernel-trace --pmc <SQ counters>, one or more passes; per-launch means. SQ counters count wave-level events "
               "(SQ_INSTS_VALU = wave64 VALU instructions issued).", "kernels": {}}
for k in sorted(acc):
    d = {c: round(sum(v) / len(v), 1) for c, v in acc[k].items()}
    d["launches_sampled"] = max(len(v) for v in acc[k].values())
    if d.get("SQ_BUSY_CYCLES") and d.get("SQ_WAVE_CYCLES"):
        d["waves_per_simd_resident"] = round(4.0 * d["SQ_WAVE_CYCLES"] / d["SQ_BUSY_CYCLES"] / 32.0, 3)
    if d.get("SQ_WAIT_INST_ANY") and d.get("SQ_WAVE_CYCLES"):
        d["wait_frac"] = round(d["SQ_WAIT_INST_ANY"] / d["SQ_WAVE_CYCLES"], 4)
    res["kernels"][k] = d
print(json.dumps(res, indent=1))
