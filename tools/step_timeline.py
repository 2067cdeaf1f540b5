#!/usr/bin/env python3
"""Every kernel launch of ONE steady-state step in start order (from a rocprofv3 --kernel-trace rocpd database): start offset,
duration, name -- per-layer times where several layers share a kernel symbol.  usage: step_timeline.py <results.db> [marker] [step index]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
marker = sys.argv[2] if len(sys.argv) > 2 else "topk_regress_kernel<2>"
which = int(sys.argv[3]) if len(sys.argv) > 3 else 6
rows = db.execute("select name, start, end from kernels order by start").fetchall()
marks = [e for n, s, e in rows if marker in n]
t0, t1 = marks[which], marks[which + 1]
print(f"step {which}: {(t1 - t0) / 1e3:.1f} us between two ends of the marker kernel")
prev_end = t0
for n, s, e in rows:
    if t0 < s <= t1 or (s <= t1 and e > t0 and s > t0 - 1):
        short = n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
        short = short[:short.index("(")] if "(" in short else short
        print(f"{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:7.1f} us  {short[:70]}")
