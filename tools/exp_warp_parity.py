#!/usr/bin/env python3
"""HIP SpatialTransformer_grid / sparse concat volume against the CPU oracle (= F.grid_sample, the reference's own call) at the
widths of the bench shapes: the coordinate round trip leaves ix = integer + delta with |delta| ~ W * 1e-7, so a one-ulp difference
anywhere in that arithmetic changes the bilinear weights by ~1e-5 -- invisible at the fixtures' W = 16."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import semstereo_amd as sa
from semstereo_amd import ops
from oracle import ops as oops
torch.manual_seed(0)
for (H, W, nd, m4) in ((64, 64, 24, 32), (32, 256, 24, 32), (16, 512, 24, 48), (8, 1024, 24, 96)):
    x = torch.randn(1, 8, H, W); y = torch.randn(1, 8, H, W)
    smp = torch.stack([torch.randperm(2 * m4)[:nd].sort()[0].float() - m4 for _ in range(H * W)], 1).reshape(1, nd, H, W)
    yw_o, _ = oops.SpatialTransformer_grid(x, y, smp)
    yw64, _ = oops.SpatialTransformer_grid(x.double(), y.double(), smp.double())
    yw_h, _ = ops.SpatialTransformer_grid(x.cuda(), y.cuda(), smp.cuda())
    yw_h = yw_h.cpu()
    d = (yw_h - yw_o).abs()
    print(f"W={W:5d}: hip vs cpu-fp32 max {d.max():.3e}, bitwise equal {float((yw_h == yw_o).float().mean()):.6f}; "
          f"hip vs f64 rms {float((yw_h.double() - yw64).pow(2).mean().sqrt()):.3e} max {float((yw_h.double() - yw64).abs().max()):.3e}; "
          f"cpu-fp32 vs f64 rms {float((yw_o.double() - yw64).pow(2).mean().sqrt()):.3e} max {float((yw_o.double() - yw64).abs().max()):.3e}")
