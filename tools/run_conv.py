#!/usr/bin/env python3
"""Runs only the dominant conv (concat_stem's warped half: 32->32 k3 on [B,32,24,H/4,W/4] continuing a partial sum, gated;
SS_TOOL_CIN=64: the whole stem without partial sum) so that rocprofv3 PMC passes see nothing else.
usage: run_conv.py [engine f32|bf16x6|bf16x3|f16x3] [iters] [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semstereo_amd import _lib  # noqa: E402
if os.environ.get("SS_TOOL_LIB"):          # ablation builds of the library (tools/ablate_conv.sh)
    _lib.LIB_PATH = os.path.abspath(os.environ["SS_TOOL_LIB"])
from semstereo_amd import modules as M  # noqa: E402

engine = sys.argv[1] if len(sys.argv) > 1 else "bf16x6"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dev = torch.device("cuda")
CIN = int(os.environ.get("SS_TOOL_CIN", "32"))
x = torch.randn(B, CIN, 24, 256, 256, device=dev)
w = torch.randn(32, CIN, 3, 3, 3, device=dev) * (1.0 / (CIN * 27)) ** 0.5
part = torch.randn(B, 32, 24, 256, 256, device=dev) if CIN == 32 else None
gate = torch.rand(B, 32, 256, 256, device=dev) if CIN == 32 else None
sc, sh = torch.rand(32, device=dev) + 0.5, torch.randn(32, device=dev) * 0.1
if engine == "f32":
    wp = M.pack_conv_weight(w)
    fn = lambda: M.conv3d_hip(x, wp, sc, sh, 3, 1, True)
else:
    nt = {"bf16x6": 6, "bf16x3": 3, "f16x3": 19}[engine]
    ws = M.pack_conv_weight_bf16s(w, nt)
    fn = lambda: M.conv3d_bf16s_hip(x, ws, 32, sc, sh, True, nt, None, gate, partial=part)
import time  # noqa: E402
t0 = time.time()
while time.time() - t0 < 0.4:           # warm clocks: a cold chip measures ~20 % slower
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
gf = 2.0 * B * 32 * CIN * 27 * 24 * 256 * 256 / 1e9
if engine != "f32" and os.environ.get("SS_TOOL_CHECK") and CIN != 32:
    ref = M.conv3d_hip(x, M.pack_conv_weight(w), sc, sh, 3, 1, True)
    print(f"max |bf16s - f32 engine| = {(fn() - ref).abs().max().item():.2e}  ", end="")
print(f"concat_stem conv [{engine}] B={B}: {ms*1e3:.1f} us/launch, {gf:.1f} GFLOP, {gf/ms:.1f} TFLOP/s fp32-equivalent")
