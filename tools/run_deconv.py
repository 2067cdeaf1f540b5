#!/usr/bin/env python3
"""Runs only the last transposed conv of hourglass2 (64 -> 32 on [B,64,12,H/8,W/8] + redir of [B,32,24,H/4,W/4]) for PMC passes
and timing ablations.  usage: run_deconv.py [engine f32|bf16x6] [iters]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semstereo_amd import _lib  # noqa: E402
if os.environ.get("SS_TOOL_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["SS_TOOL_LIB"])
from semstereo_amd import modules as M  # noqa: E402

engine = sys.argv[1] if len(sys.argv) > 1 else "bf16x6"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda")
cin, cout, d, h, cs = 64, 32, 12, 128, 32
x = torch.randn(1, cin, d, h, h, device=dev)
w = torch.randn(cin, cout, 3, 3, 3, device=dev) * (8.0 / (cin * 27)) ** 0.5
wp = M.pack_conv_weight(w, transposed=True)
sh = torch.randn(cout, device=dev) * 0.1
skip = torch.randn(1, cs, 2 * d, 2 * h, 2 * h, device=dev)
ws = torch.randn(cs, cout, device=dev) * (1.0 / cs) ** 0.5
if engine == "f32":
    fn = lambda: M.deconv3d_hip(x, wp, sh, True, skip, ws)
else:
    wds, wss = M.pack_deconv_weight_bf16s(wp), M.pack_deconv_weight_bf16s(ws)
    fn = lambda: M.deconv3d_bf16s_hip(x, wds, cout, sh, True, 6, skip, wss)
t0 = time.time()
while time.time() - t0 < 0.4:
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    fn()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
gf = 2.0 * cout * (cin * 27 + cs * 8) * d * h * h / 1e9
print(f"hg2.conv6 deconv [{engine}]: {ms*1e3:.1f} us/launch, {gf:.1f} GFLOP, {gf/ms:.1f} TFLOP/s fp32-equivalent")
