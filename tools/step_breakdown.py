#!/usr/bin/env python3
"""Per-kernel time inside the steady-state steps of a `rocprofv3 --kernel-trace` run of bench.py.

The rocpd database also holds the warm-up (MIOpen's find phase runs every candidate solver there) and
bench.py's micro-runs; this tool cuts the window between two launches of the marker kernel that runs
exactly once per step (the final top-2 soft-argmax) and reports time per step inside it.
usage: step_breakdown.py <results.db> [marker-substring] [skip-first-n]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
marker = sys.argv[2] if len(sys.argv) > 2 else "topk_regress_kernel<2>"     # runs exactly once per step, in no micro-run
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 4
rows = db.execute("select name, start, end from kernels order by start").fetchall()
marks = [s for n, s, e in rows if marker in n]
if len(marks) < skip + 2:
    sys.exit(f"only {len(marks)} launches of the marker kernel")
t0, t1, nsteps = marks[skip], marks[-1], len(marks) - 1 - skip
agg = {}
for n, s, e in rows:
    if t0 <= s < t1:
        a = agg.setdefault(n, [0, 0.0])
        a[0] += 1
        a[1] += (e - s) / 1e3
busy = sum(a[1] for a in agg.values())
# time with NO kernel running at all inside the window (dispatch gaps between dependent kernels, host stalls)
iv = sorted((s_, e_) for n_, s_, e_ in rows if t0 <= s_ < t1)
covered, cur_s, cur_e = 0, None, None
for s_, e_ in iv:
    if cur_e is None or s_ > cur_e:
        if cur_e is not None:
            covered += cur_e - cur_s
        cur_s, cur_e = s_, e_
    else:
        cur_e = max(cur_e, e_)
if cur_e is not None:
    covered += cur_e - cur_s
idle_us = ((t1 - t0) - covered) / 1e3 / nsteps
ngaps = len(iv) / nsteps
print(f"{nsteps} steps, wall {(t1 - t0) / 1e6 / nsteps:.3f} ms/step, kernels busy {busy / 1e3 / nsteps:.3f} ms/step, "
      f"no kernel running {idle_us:.1f} us/step over {ngaps:.0f} launches")
print(f"{'us/step':>9} {'calls':>6} {'avg us':>8}  kernel")
for n, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{us / nsteps:9.1f} {c / nsteps:6.1f} {us / c:8.1f}  {n[:120]}")
