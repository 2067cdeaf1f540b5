#!/usr/bin/env python3
"""Where do the SMALL launches of the training step come from?  (r06: ~600 launches of 2-6 us per step -- fills, copies, adds,
scalar multiplies -- are 2-3 ms of a 16.6 ms step at 1024^2 / maxdisp 64, tools/profile_train.sh.)  One step under torch.profiler
with Python stacks: every aten op that launches something is attributed to the innermost frame of semstereo_amd / this tool on its
stack, or -- for ops the autograd engine runs without a Python frame -- to the backward node it runs under.

usage (GPU box): python tools/train_glue_sources.py [--batch 1] [--top 60]
"""
import argparse
import collections
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--maxdisp", type=int, default=64)
    ap.add_argument("--top", type=int, default=60)
    args = ap.parse_args()
    import torch
    import torch.nn.functional as F
    from torch.profiler import ProfilerActivity, profile
    import bench
    import semstereo_amd as sa
    assert torch.cuda.is_available(), "needs the MI355X"
    sa._lib.load()
    dev = torch.device("cuda")
    B, H, W, md = args.batch, args.height, args.width, args.maxdisp
    seg = sa.HotSegment(md).to(dev).train()
    bench.init_unit_gain(seg, 1234)
    opt = torch.optim.Adam(seg.parameters(), lr=1e-3, betas=(0.9, 0.999), fused=True)
    fl8, fr8 = bench.synth_features(B, 256, H // 8, W // 8, 6, 5100, dev)
    fl4, fr4 = bench.synth_features(B, 128, H // 4, W // 4, 12, 5200, dev)
    feats = [t.requires_grad_(True) for t in (fl4, fr4, fl8, fr8)]
    gt = (torch.rand(B, H // 4, W // 4, device=dev) * 2 - 1) * (md // 4 - 1)

    def step():
        opt.zero_grad(set_to_none=True)
        for t in feats:
            t.grad = None
        r = seg(*feats)
        loss = F.smooth_l1_loss(r["pred"].squeeze(1), gt) + F.smooth_l1_loss(r["pred_att"], gt)
        loss.backward()
        opt.step()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=False) as prof:
        step()
        torch.cuda.synchronize()
    events = list(prof.events())
    # ops that launched a kernel themselves (leaf aten ops with device time), by source
    by_src = collections.defaultdict(lambda: [0, 0.0, collections.Counter()])
    n_leaf = 0
    for e in events:
        if e.device_type != torch.autograd.DeviceType.CPU or not e.name.startswith("aten::"):
            continue
        kernels = getattr(e, "kernels", [])
        if not kernels:
            continue
        if any(c.name.startswith("aten::") and getattr(c, "kernels", []) for c in e.cpu_children):
            continue                                       # the launch belongs to a child op
        n_leaf += 1
        dt = sum(k.duration for k in kernels)
        src = None
        for fr in (e.stack or []):
            if "semstereo_amd/" in fr or "train_glue_sources.py" in fr:
                src = fr.strip()
                break
        if src is None:
            p = e.cpu_parent
            chain = []
            while p is not None:
                chain.append(p.name)
                p = p.cpu_parent
            node = next((c for c in chain if "Backward" in c or "autograd::engine" in c or "Optimizer" in c), None)
            src = f"[no Python frame] under {node or (chain[-1] if chain else '?')}"
        rec = by_src[src]
        rec[0] += 1
        rec[1] += dt
        rec[2][e.name] += 1
    # hipMemcpyAsync / hipMemsetAsync issued by the step (the C entry points' own zero-fills, torch's contiguous copy_ and clone): not
    # kernels in the profiler's eyes, but each is a 4 us launch on the stream all the same -- by the node they were issued under
    runtime = collections.defaultdict(collections.Counter)
    for e in events:
        if "Memcpy" not in e.name and "Memset" not in e.name:
            continue
        if e.device_type != torch.autograd.DeviceType.CPU:
            continue
        p = e.cpu_parent
        chain = []
        while p is not None:
            chain.append(p.name)
            p = p.cpu_parent
        node = next((c for c in chain if "Backward" in c or c.startswith("_") or "Optimizer" in c), None) or (chain[0] if chain else "?")
        inner = next((c for c in chain if c.startswith("aten::")), "")
        runtime[e.name][f"{node} {inner}".strip()] += 1
    for name, ctr in runtime.items():
        print(f"{name}: {sum(ctr.values())} calls in the step; by node:")
        for node, n in ctr.most_common(14):
            print(f"    {n:5d}  {node}")
    print(f"{n_leaf} launching aten ops in one step at batch {B}; by source (launches, device us, ops):")
    for src, (n, dt, ops) in sorted(by_src.items(), key=lambda kv: -kv[1][0])[:args.top]:
        print(f"{n:5d} {dt:9.1f} us  {src[-110:]}   {dict(ops.most_common(4))}")


if __name__ == "__main__":
    main()
