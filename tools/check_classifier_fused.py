#!/usr/bin/env python3
"""The one-pass classifier (ss_conv3d_classifier_fused_fwd) against the two-launch form and against float64 (CPU) on a ragged shape,
then both forms timed on the live shapes.  usage (GPU box): python3 tools/check_classifier_fused.py"""
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import semstereo_amd as sa  # noqa: E402

M, E = sa.modules, sa.engine
dev = torch.device("cuda")
torch.manual_seed(0)


def make(C=32):
    cl = M.Classifier(C).to(dev).eval()
    with torch.no_grad():
        bn = cl[0][1]
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.2); bn.running_mean.normal_(0, 0.2); bn.running_var.uniform_(0.5, 1.5)
        cl[0][0].weight.mul_(2.0); cl[2].weight.mul_(2.0)
    return cl


def run(cl, x, fused):
    E.CLASSIFIER_FUSED = fused
    with torch.no_grad():
        return cl(x)


for shape in ((1, 32, 12, 134, 200), (2, 32, 12, 134, 200), (1, 32, 24, 256, 256), (1, 32, 32, 128, 128)):
    cl = make()
    x = torch.relu(torch.randn(*shape, device=dev))
    assert E.classifier_fused_applies(x, 19), shape
    a, b = run(cl, x, True), run(cl, x, False)
    d = (a - b).abs().max().item()
    scale = b.abs().max().item()
    msg = f"{shape}: |fused - two-launch| max {d:.3e} of max |out| {scale:.3e}"
    if shape[0] == 2:
        one = run(cl, x[1:2].contiguous(), True)
        msg += f"; batch-invariant: {torch.equal(one, a[1:2])}"
    if shape[2] == 12 and shape[0] == 1:
        cd = cl.double().cpu()
        with torch.no_grad():
            y = F.relu(cd[0][1](F.conv3d(x.double().cpu(), cd[0][0].weight, padding=1)))
            ref = F.conv3d(y, cd[2].weight, padding=1)
        ea, eb = (a.double().cpu() - ref).abs().max().item(), (b.double().cpu() - ref).abs().max().item()
        ra, rb = (a.double().cpu() - ref).pow(2).mean().sqrt().item(), (b.double().cpu() - ref).pow(2).mean().sqrt().item()
        msg += f"; vs float64: fused max {ea:.3e} rms {ra:.3e}, two-launch max {eb:.3e} rms {rb:.3e}"
        cl = cl.float().to(dev)
    print(msg, flush=True)

for shape in ((1, 32, 24, 256, 256), (1, 32, 32, 128, 128), (1, 32, 36, 512, 512)):
    cl = make()
    x = torch.relu(torch.randn(*shape, device=dev))
    for fused in (True, False, True, False):
        for _ in range(3):
            run(cl, x, fused)
        torch.cuda.synchronize()
        t0 = time.time()
        while time.time() - t0 < 0.4:
            for _ in range(10):
                run(cl, x, fused)
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            run(cl, x, fused)
        e1.record()
        torch.cuda.synchronize()
        print(f"{shape} fused={fused}: {e0.elapsed_time(e1) / 30 * 1e3:.1f} us", flush=True)
