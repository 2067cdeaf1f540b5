#!/usr/bin/env python3
"""Where the workgroups of ONE conv launch ran (instrumented library, tools/build_timing.sh): every workgroup records the XCC /
SE / SH / CU of its first wave when it finishes.  Prints workgroups per CU (histogram) and CUs used per XCD.
usage: SS_TOOL_LIB=tools/_build/lib_timing.so [STRIDE=1] python tools/wg_placement.py Cin Cout D H W [batch]"""
import collections
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semstereo_amd import _lib  # noqa: E402
_lib.LIB_PATH = os.path.abspath(os.environ["SS_TOOL_LIB"])
from semstereo_amd import modules as M  # noqa: E402

Cin, Cout, D, H, W = [int(a) for a in sys.argv[1:6]]
B = int(sys.argv[6]) if len(sys.argv) > 6 else 1
stride = int(os.environ.get("STRIDE", "1"))
dev = torch.device("cuda")
x = torch.randn(B, Cin, D, H, W, device=dev)
w = torch.randn(Cout, Cin, 3, 3, 3, device=dev) * (1.0 / (Cin * 27)) ** 0.5
sc, sh = torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev) * 0.1
ws = M.pack_conv_weight_bf16s(w, 19)
run = lambda: M.conv3d_bf16s_hip(x, ws, Cout, sc, sh, True, 19, stride=stride)   # noqa: E731
for _ in range(5):
    run()
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)
assert lib.ss_debug_reset_place() == 0
run()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 65536)()
assert lib.ss_debug_read_place(buf, 65536) == 0
v = np.frombuffer(buf, dtype=np.uint64)
v = v[(v >> np.uint64(63)) == 1]
hw = (v & np.uint64(0xffffffff)).astype(np.int64)
xcc = ((v >> np.uint64(32)) & np.uint64(0xf)).astype(np.int64)
cu, shid, se = (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 0x7
key = list(zip(xcc.tolist(), se.tolist(), shid.tolist(), cu.tolist()))
per_cu = collections.Counter(key)
hist = collections.Counter(per_cu.values())
print(f"{Cin} -> {Cout} stride {stride} on [{D},{H},{W}] batch {B}: {len(v)} workgroups on {len(per_cu)} CUs; workgroups per CU: "
      + ", ".join(f"{k}: {hist[k]} CUs" for k in sorted(hist)))
per_xcd = collections.Counter(k[0] for k in per_cu)
print("  CUs used per XCD:", dict(sorted(per_xcd.items())), " first 16 workgroups ->", key[:16])
