#!/usr/bin/env python3
"""Runs only the cost-volume build kernel (live shape of models/SemStereo.py:273) so that rocprofv3
kernel-trace / PMC passes see nothing else.  usage: run_gwc.py [batch] [iters] [normalize]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import semstereo_amd as sa  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
norm = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dev = torch.device("cuda")
fl, fr = torch.randn(B, 256, 128, 128, device=dev), torch.randn(B, 256, 128, 128, device=dev)
fn = sa.ops.build_gwc_volume_norm if norm else sa.ops.build_gwc_volume
for _ in range(3):
    out = fn(fl, fr, 16, 32)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    out = fn(fl, fr, 16, 32)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
nbytes = 4.0 * B * (2 * 256 + 32 * 32) * 128 * 128
print(f"gwc B={B} norm={norm}: {ms*1e3:.1f} us/launch, {nbytes/1e6:.1f} MB algorithmic, {nbytes/ms/1e6:.0f} GB/s")
