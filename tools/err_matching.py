#!/usr/bin/env python3
"""Per-stage error of the MATCHING branch (models/SemStereo.py:314-323) against its float64 evaluation, HIP path and fp32 CPU
oracle side by side, every path fed the reference's candidates of a calibrated fixture (no candidate differences anywhere):
where the HIP path's distance from the exact answer comes from.  Test tooling (imports oracle/, tests/golden).
usage: err_matching.py [fixture name, default s256_md128_cal] [engine]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from semstereo_amd import _lib  # noqa: E402
if os.environ.get("SS_TOOL_LIB"):          # experimental builds of the library (tools/build_variant.sh)
    _lib.LIB_PATH = os.path.abspath(os.environ["SS_TOOL_LIB"])
import semstereo_amd as sa  # noqa: E402
from semstereo_amd import deferred as _dfr  # noqa: E402
_dfr.ENABLED = False
from semstereo_amd import modules as M  # noqa: E402
from semstereo_amd import engine as sa_engine  # noqa: E402
from golden import cases  # noqa: E402
from oracle import hot_segment as oseg  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "s256_md128_cal"
if len(sys.argv) > 2:
    sa_engine.CONV_ENGINE = sys.argv[2]
full = name in cases.SEGMENT_FULL
g = np.load(os.path.join(ROOT, "tests", "golden", "segment_full.npz" if full else "segment.npz"))
B, H, W, maxdisp = cases.segment_shape(name)
P = cases.segment_params(name, g)
fl4, fr4, fl8, fr8, _ = cases.segment_inputs(name)
if full:
    # full-size records hold the reference's candidates only at the risk pixels: take the oracle's (fp32) attention branch instead
    att, smp, _ = oseg.attention_branch(P, fl8, fr8, fl4, fr4, maxdisp)
else:
    smp = torch.as_tensor(g[f"{name}/samples"].astype(np.float32))
    att = torch.as_tensor(g[f"{name}/att_topk"]).unsqueeze(1)
seg = sa.HotSegment(maxdisp)
seg.load_state_dict(P, strict=False)
seg = seg.cuda().eval()
cap = {}
real_half = M.stem_volume_half


def spy_half(*a, **k):
    r = real_half(*a, **k)
    cap["stem"] = r.detach().clone()
    return r


M.stem_volume_half = spy_half
seg.concat_stem.register_forward_hook(lambda m, a, o: cap.__setitem__("stem_plain", o.detach().clone()))      # (engines without the by-halves form)
seg.concat_feature_att_4.register_forward_hook(lambda m, a, o: cap.__setitem__("stem_gated", o.detach().clone()))
seg.hourglass.register_forward_hook(lambda m, a, o: cap.__setitem__("hourglass", o.detach().clone()))
seg.classif.register_forward_hook(lambda m, a, o: cap.__setitem__("cost", o.detach().clone()))
with torch.no_grad():
    cap["pred"] = seg.matching_branch(fl4.cuda(), fr4.cuda(), att.cuda(), smp.cuda())
k32, k64 = {}, {}
torch.set_num_threads(min(os.cpu_count() or 1, 32))
with torch.no_grad():
    k32["pred"] = oseg.matching_branch(P, fl4, fr4, att, smp, k32)
    P64 = {k: (v.double() if v.is_floating_point() else v) for k, v in P.items()}
    k64["pred"] = oseg.matching_branch(P64, fl4.double(), fr4.double(), att.double(), smp.double(), k64)
print(f"{name} engine {M.CONV_ENGINE}: every path on the same candidates")
print(f"{'stage':14s} {'hip rms':>10s} {'hip max':>10s} {'o32 rms':>10s} {'o32 max':>10s}  hip/o32 rms   (relative to the rms of the float64 tensor)")
if "stem" not in cap:
    cap["stem"] = cap.get("stem_gated", cap.get("stem_plain"))
for key in ("stem", "hourglass", "cost", "pred"):
    t = k64[key]
    h, o = cap[key].cpu().double().reshape(t.shape), k32[key].double()
    n = t.pow(2).mean().sqrt().item()
    eh, eo = (h - t), (o - t)
    print(f"{key:14s} {eh.pow(2).mean().sqrt().item() / n:10.2e} {eh.abs().max().item() / n:10.2e} {eo.pow(2).mean().sqrt().item() / n:10.2e} "
          f"{eo.abs().max().item() / n:10.2e}  {eh.pow(2).mean().sqrt().item() / max(eo.pow(2).mean().sqrt().item(), 1e-300):6.2f}")
