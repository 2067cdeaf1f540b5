#!/usr/bin/env python3
"""Fold the records tools/pmc_bytes.sh left in gpurun_out/ (pmc_<kernel>_b<batch>.json, and the .md tables beside them) into
profiles/: the tables as profiles/<tag>_pmc_<kernel>_b<batch>.md, the byte counts into profiles/pmc_traffic.json, which
bench.py reads for its `traffic` fields.   usage: tools/collect_pmc_traffic.py <tag, e.g. r03_f>"""
import glob
import json
import os
import shutil
import sys

# FETCH_SIZE on gfx950 reports half the bytes of WIDE coalesced reads (16 B per lane: MI355X_MICROARCH.md) and is "uncalibrated"
# for other widths.  Kernels whose loads are 4 or 8 bytes per lane are taken at x1: calibrated on the stem conv, whose x2 figure
# (847 MB of activation reads) would exceed what its halo geometry can re-read at all (6x6x34 positions per 4x4x32 outputs =
# 2.39 x 201 MB = 481 MB), while the x1 figure (319 MB = 1.6x) sits inside it.
NARROW_LOADS = {"stem", "stem_gather", "stem_gather_smooth", "warp", "strength", "stem_left", "conv_s1", "conv_s2", "conv_mid", "conv_low"}

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = os.path.join(root, "profiles", "pmc_traffic.json")
table = json.load(open(path)) if os.path.exists(path) else {}
files = sorted(glob.glob(os.path.join(root, "gpurun_out", "pmc_*_b*.json")))
# gpurun_out/ accumulates over calls and rounds: only the records of the LATEST measurement run (within two hours of the newest
# file) are filed under this tag -- an older record keeps the tag it was collected under
newest = max((os.path.getmtime(f) for f in files), default=0.0)
for f in files:
    if os.path.getmtime(f) < newest - 7200:
        print("skipped (stale):", os.path.basename(f))
        continue
    rec = json.load(open(f))
    key = f"{rec['run_kernel']}_b{rec['batch']}"
    md = f[:-5] + ".md"
    dst = f"{tag}_pmc_{key}.md"
    if os.path.exists(md):
        shutil.copy(md, os.path.join(root, "profiles", dst))
    mult = 0.5 if rec["run_kernel"] in NARROW_LOADS else 1.0            # (pmc_bytes.sh doubled FETCH_SIZE)
    rd = rec["read_bytes"] * mult
    table[key] = {"read_bytes": rd, "written_bytes": rec["written_bytes"], "total_bytes": rd + rec["written_bytes"],
                  "fetch_size_multiplier": 2 * mult, "kernel": rec["kernel"], "source": dst}
    rec["total_bytes"] = rd + rec["written_bytes"]
    print(key, f"{rec['total_bytes'] / 1e6:.1f} MB", "->", dst)
json.dump(table, open(path, "w"), indent=1, sort_keys=True)
