#!/usr/bin/env python3
"""How many top-24 picks of each fp32 evaluation of the attention branch differ from the float64 evaluation, over N seeded pairs
(bench.seeded_pairs_parity with more pairs than the bench line affords): the HIP conv engines and the fp32 CPU oracle (= the
reference's arithmetic) side by side.  Test tooling (imports oracle/).   usage: tools/flip_stats.py [pairs, default 32] [H W maxdisp]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import semstereo_amd as sa  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
H, W, maxdisp = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (1024, 1024, 128)
dev = torch.device("cuda")
seg = sa.HotSegment(maxdisp).to(dev).eval()
bench.init_unit_gain(seg, 1234)
stats, rows, secs = bench.seeded_pairs_parity(seg, sa.modules, ["f16x3", "f32", "bf16x6"], n, H, W, maxdisp, dev, min(os.cpu_count() or 1, 32))
print(json.dumps({"pairs": n, "shape": [H, W, maxdisp], "stat": "[mean, std] per pair of 65 536 pixels (at 1024^2)", "stats": stats,
                  "cpu_seconds_per_pair": sum(secs) / len(secs)}))
for e, rr in rows.items():
    print(e, [r["picks_differing_from_float64"] for r in rr])
