"""One frame's kernel timeline from a rocprofv3 rocpd .db (kernel trace): every launch of a steady-state frame in dispatch order
with its duration and the gap to the previous launch's end -- averaged over the frames of the trace.

    python tools/frame_timeline.py gpurun_out/x/bench_results.db [first_kernel_substring]
"""
import sqlite3, sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
first = sys.argv[2] if len(sys.argv) > 2 else "k_preprocess"
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
frames, curf = [], None
for n, s, e in rows:
    if first in n:
        curf = []
        frames.append(curf)
    if curf is not None:
        curf.append((n, s, e))
frames = frames[len(frames) // 3:-1]                      # steady state
L = max(set(len(f) for f in frames), key=[len(f) for f in frames].count)
frames = [f for f in frames if len(f) == L]
print(f"{len(frames)} frames of {L} launches")
tot_k = tot_g = 0.0
for i in range(L):
    d = sum(f[i][2] - f[i][1] for f in frames) / len(frames) / 1e3
    g = sum((f[i][1] - f[i - 1][2]) if i else 0 for f in frames) / len(frames) / 1e3
    tot_k += d; tot_g += g
    print(f"{i:3d} {frames[0][i][0][:60]:60s} {d:8.2f} us   gap before {g:7.2f} us")
span = sum(f[-1][2] - f[0][1] for f in frames) / len(frames) / 1e3
print(f"kernels {tot_k:.1f} us + gaps {tot_g:.1f} us = {span:.1f} us from the first launch's start to the last one's end")
