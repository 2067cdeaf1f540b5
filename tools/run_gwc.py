#!/usr/bin/env python3
"""Runs only the cost-volume build kernel (live shape of models/SemStereo.py:273) so that rocprofv3
kernel-trace / PMC passes see nothing else.  usage: run_gwc.py [batch] [iters] [normalize] [hold]
hold=1 keeps the previous result alive, so the caching allocator alternates between two output buffers;
hold=0 (default) writes the same buffer every launch, as a steady-state inference loop does."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import semstereo_amd as sa  # noqa: E402
from semstereo_amd import deferred as _dfr  # noqa: E402
_dfr.ENABLED = False          # these tools time / inspect each op by itself: no deferred handles

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
norm = int(sys.argv[3]) if len(sys.argv) > 3 else 1
hold = int(sys.argv[4]) if len(sys.argv) > 4 else 0
dev = torch.device("cuda")
fl, fr = torch.randn(B, 256, 128, 128, device=dev), torch.randn(B, 256, 128, 128, device=dev)
fn = sa.ops.build_gwc_volume_norm if norm else sa.ops.build_gwc_volume
warm = int(os.environ.get("SS_WARM_MS", "400"))   # clocks ramp with load: a cold chip measures ~25 % slower
for _ in range(3):
    out = fn(fl, fr, 16, 32)
torch.cuda.synchronize()
del out
import time  # noqa: E402
t0 = time.time()
while (time.time() - t0) * 1e3 < warm:
    for _ in range(20):
        fn(fl, fr, 16, 32)
    torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    if hold:
        out = fn(fl, fr, 16, 32)
    else:
        fn(fl, fr, 16, 32)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
nbytes = 4.0 * B * (2 * 256 + 32 * 32) * 128 * 128
print(f"gwc B={B} norm={norm} hold={hold} stream={os.environ.get('SS_GWC_STREAM', 'auto')}: {ms*1e3:.1f} us/launch, {nbytes/1e6:.1f} MB algorithmic, {nbytes/ms/1e6:.0f} GB/s")
