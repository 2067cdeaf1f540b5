#!/usr/bin/env python3
"""Where a workgroup of stem_left_fused spends its cycles (instrumented library of tools/build_timing_stem_left.sh).
usage: SS_TOOL_LIB=tools/_build/lib_timing_sl.so python tools/wg_phases_stem_left.py"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semstereo_amd import _lib  # noqa: E402
_lib.LIB_PATH = os.path.abspath(os.environ["SS_TOOL_LIB"])
from semstereo_amd import modules as M  # noqa: E402

dev = torch.device("cuda")
stem = M.BasicConv(64, 32, is_3d=True, kernel_size=3, stride=1, padding=1).to(dev).eval()
cl, att = torch.randn(1, 32, 256, 256, device=dev), torch.rand(1, 1, 24, 256, 256, device=dev)
run = lambda: M.stem_broadcast_half(stem, cl, att)      # noqa: E731
with torch.no_grad():
    for _ in range(10):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    print(f"stem_left_fused: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch (instrumented build)")
    lib = ctypes.CDLL(_lib.LIB_PATH)
    assert lib.ss_debug_reset_sl() == 0
    run()
    torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (8 * 4096))()
assert lib.ss_debug_read_sl(buf, 8 * 4096) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 8).astype(np.int64)
t = t[t[:, 4] > 0]
print(f"{len(t)} workgroups, cycles summed over the 32 channels (thread 0's view)")
for name, v in (("Q reads + multiply-adds (+ MFMA)", t[:, 0]), ("stores issued + Q parked", t[:, 1]), ("wait at the barrier", t[:, 2]),
                ("whole channel loop", t[:, 4])):
    print(f"  {name:32s} mean {v.mean():9.1f}  p10 {np.percentile(v, 10):9.1f}  p90 {np.percentile(v, 90):9.1f}")
print(f"  span of loop starts {t[:, 5].max() - t[:, 5].min()} cycles")
