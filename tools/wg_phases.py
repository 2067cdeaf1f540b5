#!/usr/bin/env python3
"""Per-workgroup phase times of the dominant conv from the instrumented library tools/build_timing.sh builds
(tools/_build/lib_timing.so = the product library with tools/conv_timing.hip in place of conv3d_bf16s.hip): prologue (launch -> first chunk staged), K loop, epilogue, and how many workgroups are
resident over time.  usage: SS_TOOL_LIB=tools/_build/lib_timing.so python tools/wg_phases.py [engine]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semstereo_amd import _lib  # noqa: E402
_lib.LIB_PATH = os.path.abspath(os.environ["SS_TOOL_LIB"])
from semstereo_amd import modules as M  # noqa: E402

engine = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
mode = sys.argv[2] if len(sys.argv) > 2 else "stem"      # stem: partial sum + gate; gate; partial; plain
nt = {"bf16x6": 6, "bf16x3": 3, "f16x3": 19}[engine]
DD = int(sys.argv[3]) if len(sys.argv) > 3 else 24      # planes: 24 -> 3072 workgroups (12 per CU); 2 -> 256 (one per CU)
dev = torch.device("cuda")
x = torch.randn(1, 32, DD, 256, 256, device=dev)
w = torch.randn(32, 32, 3, 3, 3, device=dev) * (1.0 / (32 * 27)) ** 0.5
part = torch.randn(1, 32, DD, 256, 256, device=dev)
gate = torch.rand(1, 32, 256, 256, device=dev)
sc, sh = torch.rand(32, device=dev) + 0.5, torch.randn(32, device=dev) * 0.1
ws = M.pack_conv_weight_bf16s(w, nt)
for _ in range(30):
    M.conv3d_bf16s_hip(x, ws, 32, sc, sh, True, nt, None, gate if mode in ("stem", "gate") else None,
                       partial=part if mode in ("stem", "partial") else None)
torch.cuda.synchronize()
n = 8 * 32 * (DD // 2)
buf = (ctypes.c_ulonglong * (8 * n))()
lib = ctypes.CDLL(_lib.LIB_PATH)
assert lib.ss_debug_read(buf, 8 * n) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(n, 8).astype(np.int64)
t0 = t[:, 0].min()
pro, loop, epi = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
print(f"{engine} [{mode}] {n} workgroups: shader clock cycles")
steps = t[:, 5]
for name, v in (("prologue", pro), ("K loop", loop), (" K-steps", steps), (" chunk boundaries", loop - steps), ("epilogue", epi), ("lifetime", t[:, 3] - t[:, 0])):
    print(f"  {name:18s} mean {v.mean():8.1f}  p10 {np.percentile(v, 10):8.1f}  p50 {np.percentile(v, 50):8.1f}  p90 {np.percentile(v, 90):8.1f}")
print(f"  kernel span {t[:, 3].max() - t0} ticks; sum of lifetimes / span / 256 CUs = {(t[:, 3] - t[:, 0]).sum() / (t[:, 3].max() - t0) / 256:.2f} resident workgroups per CU")
