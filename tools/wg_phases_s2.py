#!/usr/bin/env python3
"""Phase times of the persistent workgroups of a STRIDE-2 conv layer from the instrumented library (tools/build_timing.sh):
one launch after a reset of the stamps; per workgroup: prologue, K-steps summed over its tiles, everything else of the K loops
(chunk boundaries: maxima, barriers, split, LDS writes) + epilogues.  usage: SS_TOOL_LIB=tools/_build/lib_timing.so python tools/wg_phases_s2.py [Cin Cout D H W]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semstereo_amd import _lib  # noqa: E402
_lib.LIB_PATH = os.path.abspath(os.environ["SS_TOOL_LIB"])
from semstereo_amd import modules as M  # noqa: E402

Cin, Cout, D, H, W = [int(a) for a in sys.argv[1:6]] if len(sys.argv) > 5 else (32, 64, 24, 256, 256)
stride = int(os.environ.get("STRIDE", "2"))
dev = torch.device("cuda")
x = torch.randn(1, Cin, D, H, W, device=dev)
w = torch.randn(Cout, Cin, 3, 3, 3, device=dev) * (1.0 / (Cin * 27)) ** 0.5
sc, sh = torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev) * 0.1
ws = M.pack_conv_weight_bf16s(w, 19)
run = lambda: M.conv3d_bf16s_hip(x, ws, Cout, sc, sh, True, 19, stride=stride)
for _ in range(10):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    run()
e1.record()
torch.cuda.synchronize()
print(f"{Cin} -> {Cout} stride {stride} on [{D},{H},{W}]: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch (instrumented build)")
lib = ctypes.CDLL(_lib.LIB_PATH)
assert lib.ss_debug_reset() == 0
run()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (8 * 16384))()
assert lib.ss_debug_read(buf, 8 * 16384) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(16384, 8).astype(np.int64)
t = t[t[:, 0] > 0]
n = len(t)
life, pro, steps, epi_last = t[:, 3] - t[:, 0], t[:, 1] - t[:, 0], t[:, 5], t[:, 3] - t[:, 2]
print(f"{n} stamped workgroups (channel tile 0): shader-clock cycles")
for name, v in (("lifetime", life), ("prologue", pro), ("K-steps (all tiles)", steps), ("last tile's epilogue", epi_last),
                ("rest: boundaries + epilogues", life - pro - steps)):
    print(f"  {name:30s} mean {v.mean():9.1f}  p10 {np.percentile(v, 10):9.1f}  p50 {np.percentile(v, 50):9.1f}  p90 {np.percentile(v, 90):9.1f}")
print(f"  kernel span {t[:, 3].max() - t[:, 0].min()} cycles")
