#!/usr/bin/env python3
"""Is the batch-1 step loop bound by the HOST?  Times, for the plain loop and for PairPipeline lanes, how long the Python loop
needs to ISSUE K steps (no synchronisation inside) against the time until the GPU has finished them.
usage: python tools/host_bound.py [steps] [lanes]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import semstereo_amd as sa  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 300
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda")
torch.manual_seed(0)
seg = sa.HotSegment(128).to(dev).eval()
sets = [(torch.randn(1, 128, 256, 256, device=dev), torch.randn(1, 128, 256, 256, device=dev),
         torch.randn(1, 256, 128, 128, device=dev), torch.randn(1, 256, 128, 128, device=dev)) for _ in range(3)]


def run(fn, n):
    for i in range(20):
        fn(*sets[i % 3])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn(*sets[i % 3])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6


with torch.no_grad():
    issue, total = run(seg, K)
    print(f"one stream : issue {issue:7.1f} us/step, done {total:7.1f} us/step  ({1e6 / total:.1f} pairs/s; host share {issue / total:.2f})")
    pipe = sa.PairPipeline(seg, lanes)
    issue, total = run(pipe, K)
    print(f"{lanes} lanes    : issue {issue:7.1f} us/step, done {total:7.1f} us/step  ({1e6 / total:.1f} pairs/s; host share {issue / total:.2f})")
