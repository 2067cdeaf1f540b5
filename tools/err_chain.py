#!/usr/bin/env python3
"""Error of every stage of the hot segment against the float64 evaluation of the same graph, for the
HIP path and for the fp32 CPU oracle side by side (rms and max, normalised by the rms of the truth).
Test tooling (imports oracle/).  usage: err_chain.py [H] [maxdisp] [engine]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import semstereo_amd as sa  # noqa: E402
from semstereo_amd import deferred as _dfr  # noqa: E402
_dfr.ENABLED = False          # these tools time / inspect each op by itself: no deferred handles
from semstereo_amd import modules as M  # noqa: E402
from semstereo_amd import engine as sa_engine  # noqa: E402
from semstereo_amd import ops  # noqa: E402
from oracle import hot_segment as oseg  # noqa: E402

H = W = int(sys.argv[1]) if len(sys.argv) > 1 else 512
maxdisp = int(sys.argv[2]) if len(sys.argv) > 2 else 128
if len(sys.argv) > 3:
    sa_engine.CONV_ENGINE = sys.argv[3]
dev = torch.device("cuda")
seg = sa.HotSegment(maxdisp).to(dev).eval()
bench.init_unit_gain(seg, 1234)
fl4, fr4 = bench.synth_features(1, 128, H // 4, W // 4, maxdisp // 8, 1, dev)
fl8, fr8 = bench.synth_features(1, 256, H // 8, W // 8, maxdisp // 16, 2, dev)

cap = {}


def hook(name):
    def f(mod, args, outp):
        cap[name] = outp.detach().clone()
    return f


seg.classif_att_.register_forward_hook(hook("cost_att"))
seg.concat_stem.register_forward_hook(hook("stem"))
seg.hourglass.register_forward_hook(hook("hourglass"))
seg.classif.register_forward_hook(hook("cost"))
seg.hourglass_att.register_forward_hook(hook("hourglass_att"))
real_half = M.stem_volume_half


def spy_half(*a, **k):            # concat_stem by halves does not go through concat_stem.forward
    r = real_half(*a, **k)
    cap["stem"] = r.detach().clone()
    return r


M.stem_volume_half = spy_half
real = {n: getattr(ops, n) for n in ("sample_strength", "build_gwc_volume_norm")}


def spy(name):
    def f(*a, **k):
        r = real[name](*a, **k)
        cap[name] = r.detach().clone()
        return r
    return f


for n in real:
    setattr(ops, n, spy(n))
with torch.no_grad():
    out = seg(fl4, fr4, fl8, fr8)
for n in real:
    setattr(ops, n, real[n])
M.stem_volume_half = real_half

P = {k: v.detach().cpu() for k, v in seg.state_dict().items()}
cin = [t.cpu() for t in (fl4, fr4, fl8, fr8)]
torch.set_num_threads(min(os.cpu_count() or 1, 32))
r32 = oseg.hot_segment(P, *cin, maxdisp, keep=True)
P64 = {k: (v.double() if v.is_floating_point() else v) for k, v in P.items()}
r64 = oseg.hot_segment(P64, *[t.double() for t in cin], maxdisp, keep=True)
same = (out["samples"].cpu() == r64["samples"]).all(1, keepdim=True) & (r32["samples"] == r64["samples"]).all(1, keepdim=True)
print(f"{H}x{W} maxdisp {maxdisp} engine {M.CONV_ENGINE}; pixels where all three pick the same 24 candidates: {same.double().mean().item():.5f}")
rows = [("pred_att0", out["pred_att0"], "pred_att0", False), ("cost_att (classif_att_)", cap["cost_att"], "cost_att", False),
        ("strength (5 samples)", cap["sample_strength"], "strength", False),
        ("att_topk", out["att_topk"], "att_topk_full", True), ("pred_att", out["pred_att"], "pred_att", False),
        ("concat_stem * gate", cap["stem"], "stem", True), ("hourglass", cap["hourglass"], "hourglass", True),
        ("cost (classif)", cap["cost"], "cost", True), ("pred", out["pred"], "pred", False)]
print(f"{'stage':26s} {'hip rms':>10s} {'hip max':>10s} {'o32 rms':>10s} {'o32 max':>10s}   (relative to rms of the float64 tensor)")
for label, h, key, mask in rows:
    t = r64[key]
    h = h.cpu().double().reshape(t.shape)
    o = r32[key].double()
    if mask:        # only pixels whose candidate sets agree are comparable
        m = same.reshape(same.shape[0], *([1] * (t.dim() - 3)), *same.shape[-2:]).expand_as(t)
        t, h, o = t[m], h[m], o[m]
    n = t.pow(2).mean().sqrt().item()
    print(f"{label:26s} {(h - t).pow(2).mean().sqrt().item() / n:10.2e} {(h - t).abs().max().item() / n:10.2e} "
          f"{(o - t).pow(2).mean().sqrt().item() / n:10.2e} {(o - t).abs().max().item() / n:10.2e}")
