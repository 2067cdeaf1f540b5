#!/usr/bin/env python3
"""Where does the full-size EPE come from?  Runs the hot segment three ways on the same inputs --
HIP path, the fp32 oracle, the oracle in float64 ("truth") -- and compares `pred`, the 24-candidate
cost tensor in front of regression_topk (models/SemStereo.py:322-323) and the size of the top-2 /
top-3 gap that regression_topk's hard selection depends on.  Test tooling (imports oracle/).
usage: flip_probe.py [H] [maxdisp] [engine]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import semstereo_amd as sa  # noqa: E402
from semstereo_amd import modules as M  # noqa: E402
from semstereo_amd import engine as sa_engine  # noqa: E402
from semstereo_amd import ops  # noqa: E402
from oracle import hot_segment as oseg  # noqa: E402

H = W = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
maxdisp = int(sys.argv[2]) if len(sys.argv) > 2 else 128
if len(sys.argv) > 3:
    sa_engine.CONV_ENGINE = sys.argv[3]
dev = torch.device("cuda")
seg = sa.HotSegment(maxdisp).to(dev).eval()
bench.init_unit_gain(seg, 1234)
fl4, fr4 = bench.synth_features(1, 128, H // 4, W // 4, maxdisp // 8, 1, dev)
fl8, fr8 = bench.synth_features(1, 256, H // 8, W // 8, maxdisp // 16, 2, dev)

captured = {}
real_topk = ops.regression_topk


def spy(cost, samples, k):
    captured["cost"] = cost.detach().clone()
    return real_topk(cost, samples, k)


ops.regression_topk = spy
with torch.no_grad():
    out = seg(fl4, fr4, fl8, fr8)
ops.regression_topk = real_topk
hip = {"pred": out["pred"].cpu().double(), "cost": captured["cost"].cpu().double(), "samples": out["samples"].cpu()}

P = {k: v.detach().cpu() for k, v in seg.state_dict().items()}
cin = [t.cpu() for t in (fl4, fr4, fl8, fr8)]
torch.set_num_threads(min(os.cpu_count() or 1, 32))
r32 = oseg.hot_segment(P, *cin, maxdisp, keep=True)
P64 = {k: (v.double() if v.is_floating_point() else v) for k, v in P.items()}
r64 = oseg.hot_segment(P64, *[t.double() for t in cin], maxdisp, keep=True)
o32 = {"pred": r32["pred"].double(), "cost": r32["cost"].squeeze(1).double(), "samples": r32["samples"]}
o64 = {"pred": r64["pred"], "cost": r64["cost"].squeeze(1), "samples": r64["samples"]}


def epe(a, b):
    e = (a["pred"] - b["pred"]).abs()
    return f"mean {e.mean().item():.3e}  median {e.median().item():.1e}  max {e.max().item():.2f}  >1e-3: {(e > 1e-3).double().mean().item():.2e}"


print(f"{H}x{W} maxdisp {maxdisp} engine {M.CONV_ENGINE}")
print("pred  hip  vs o32 :", epe(hip, o32))
print("pred  hip  vs f64 :", epe(hip, o64))
print("pred  o32  vs f64 :", epe(o32, o64))
for n, a in (("hip", hip), ("o32", o32)):
    same = (a["samples"] == o64["samples"]).all(1)
    d = (a["cost"] - o64["cost"])[same.unsqueeze(1).expand_as(a["cost"])]
    print(f"cost  {n} vs f64 (pixels with identical candidates {same.double().mean().item():.5f}): "
          f"rms {d.pow(2).mean().sqrt().item():.3e} max {d.abs().max().item():.3e}")
c = o64["cost"]
top = c.topk(3, dim=1).values
gap = top[:, 1] - top[:, 2]
print(f"truth cost: |mean| {c.abs().mean().item():.3f}  std across the 24 candidates {c.std(dim=1).mean().item():.3f}")
print(f"truth top2-top3 gap: median {gap.median().item():.3e};  fraction of pixels with gap < 1e-6/1e-5/1e-4/1e-3: "
      + " ".join(f"{(gap < t).double().mean().item():.2e}" for t in (1e-6, 1e-5, 1e-4, 1e-3)))
