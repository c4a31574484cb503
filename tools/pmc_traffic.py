"""Per-kernel HBM bytes per launch from two rocprofv3 PMC passes (rocpd .db files).

    python tools/pmc_traffic.py <fetch.db> <write.db> "<command that was profiled>" [workload] > profiles/rNN_x_pmc_traffic_<workload>.json

FETCH_SIZE / WRITE_SIZE are reported in KB; 'corrected' doubles FETCH_SIZE as MI355X_MICROARCH.md prescribes for gfx950.
Values of one dispatch are summed over counter instances first, then averaged over the dispatches of a kernel.
"""
import json
import re
import sqlite3
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lidar-gs_amd"))
import build_hip  # noqa: E402  (build_id / box_id: what bench.py checks a looked-up profile against)


def per_kernel(path, counter):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
    disp = "dispatch_id" if "dispatch_id" in cols else ("id" if "id" in cols else None)
    key = f"kernel_name, {disp}" if disp else "kernel_name, start"
    rows = db.execute(f"select kernel_name, sum(value) from counters_collection where counter_name = ? group by {key}", (counter,)).fetchall()
    out = {}
    for name, v in rows:
        name = re.sub(r"\(.*$", "", name).replace("void ", "").strip()
        out.setdefault(name, []).append(float(v))
    return {k: (sum(v) / len(v), len(v)) for k, v in out.items()}


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
res = {"workload": sys.argv[4] if len(sys.argv) > 4 else "cfg3",      # bench.py picks the profile of the workload it runs
       "command": sys.argv[3] if len(sys.argv) > 3 else "", "build_id": build_hip.build_id(), "box": build_hip.box_id(),
       "note": "KB units as reported by rocprofv3; 'corrected' doubles FETCH_SIZE as MI355X_MICROARCH.md prescribes for gfx950 "
               "(calibrated there on wide coalesced streams; the blend kernels issue 16-B-per-lane gathers, so treat the corrected read "
               "side as an upper bound). WRITE_SIZE is uncalibrated.",
       "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    if not k.startswith("lg::"):
        continue
    f, nf = fetch.get(k, (0.0, 0)); w, nw = write.get(k, (0.0, 0))
    res["kernels"][k] = {"FETCH_SIZE_KB_per_launch": round(f, 1), "WRITE_SIZE_KB_per_launch": round(w, 1), "launches_sampled": max(nf, nw),
                         "hbm_bytes_per_launch_raw": int((f + w) * 1024), "hbm_bytes_per_launch_corrected": int((2 * f + w) * 1024)}
print(json.dumps(res, indent=1))
