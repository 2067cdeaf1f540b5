#!/usr/bin/env python3
"""tools/kernel_power.sh's record (gpurun_out/kernel_power.txt or a copy under profiles/) as a table of joules per launch:
time x mean of the sampled socket powers, sorted by energy.   usage: tools/energy_table.py [file]"""
import re
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/kernel_power.txt"
rows, cur = [], None
for line in open(path):
    m = re.match(r"(\S+?)(?: \[(\S+)\])?: \S+ B=(\d+): ([0-9.]+) us/launch", line)
    if m:
        cur = {"name": m.group(1) + (f" [{m.group(2)}]" if m.group(2) else ""), "us": float(m.group(4)), "w": [], "mhz": []}
        rows.append(cur)
        continue
    m = re.search(r"sclk (\d+)Mhz\s+power ([0-9.]+) W", line)
    if m and cur is not None:
        cur["mhz"].append(int(m.group(1)))
        cur["w"].append(float(m.group(2)))
rows = [r for r in rows if r["w"]]
for r in rows:
    r["watt"] = sum(r["w"]) / len(r["w"])
    r["ghz"] = sum(r["mhz"]) / len(r["mhz"]) / 1e3
    r["mj"] = r["us"] * r["watt"] / 1e3
print(f"{'kernel':28s} {'us':>8s} {'W':>7s} {'GHz':>5s} {'mJ':>7s}")
for r in sorted(rows, key=lambda r: -r["mj"]):
    print(f"{r['name']:28s} {r['us']:8.1f} {r['watt']:7.0f} {r['ghz']:5.2f} {r['mj']:7.1f}")
print(f"{'sum':28s} {sum(r['us'] for r in rows):8.1f} {'':7s} {'':5s} {sum(r['mj'] for r in rows):7.1f}")
