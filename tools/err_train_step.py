#!/usr/bin/env python3
"""Forensics of the training step of the hot segment against the float64 oracle (tests/test_parity_gpu.py::
test_hot_segment_training_step_runs_on_the_hip_stack): per-parameter gradient differences, every conv / transposed-conv call of the
forward and backward pass against float64 on ITS OWN inputs, the top-2 picks of regression_topk, the head's data gradient, and the
ReLU mask of classif.0 against the oracle's.  Found with it (round 3): under SS_CONV_ENGINE=f32 every kernel call is within 2e-6 of
float64 and the picks are the oracle's, but ONE of 786 432 pre-activations of classif.0 lies within rounding of zero and lands on
the other side -- its ReLU mask flips and every gradient upstream moves by 2-5e-3 of its scale (f16x3 / bf16x6: no flip, 1e-6).
Test tooling (imports oracle/, tests/).  usage: SS_CONV_ENGINE=f32 python tools/err_train_step.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import semstereo_amd as sa
from golden import cases
from oracle import hot_segment as oseg, stack as ostack
import test_parity_gpu as T
name = sys.argv[1] if len(sys.argv) > 1 else "s128"          # [fixture] [notail] [nopadatt]: switch the r04 training kernels off one by one
if "notail" in sys.argv:
    sa.train.attention_tail_applies = lambda *a, **k: False
if "nopadatt" in sys.argv:
    _strict = sa.train.window_attention_applies
    sa.train.window_attention_applies = lambda x, heads, block: _strict(x, heads, block) and x.shape[3] % block[1] == 0 and x.shape[4] % block[2] == 0
fl4, fr4, fl8, fr8, maxdisp = cases.segment_inputs(name)
seg, P = T._segment(sa, maxdisp)
seg.train()
dev = lambda t: t.cuda()
r = seg(dev(fl4), dev(fr4), dev(fl8), dev(fr8))
(r["pred"].mean() + r["pred_att"].mean()).backward()
grads = {k: v.grad.detach().cpu() for k, v in seg.named_parameters() if v.grad is not None}
P64 = {k: v.double().clone().requires_grad_(v.is_floating_point() and not k.endswith(("running_mean", "running_var"))) for k, v in P.items()}
with ostack.training_mode():
    att, smp, pred_att = oseg.attention_branch(P64, fl8.double(), fr8.double(), fl4.double(), fr4.double(), maxdisp,
                                               force_samples=r["samples"].detach().cpu().double())      # like for like: the HIP pass's picks
    pred = oseg.matching_branch(P64, fl4.double(), fr4.double(), att, smp)
(pred.mean() + pred_att.mean()).backward()
print("engine", sa.modules.CONV_ENGINE, "pred diff max", float((r["pred"].detach().cpu().double() - pred.detach()).abs().max()), "pred_att diff", float((r["pred_att"].detach().cpu().double() - pred_att.detach()).abs().max()))
errs = {k: float((g.double() - P64[k].grad).abs().max()) / (float(P64[k].grad.abs().max()) + 1e-12) for k, g in grads.items() if k not in ("gamma", "beta")}
print("median", sorted(errs.values())[len(errs) // 2])
for k in sorted(errs, key=errs.get, reverse=True)[:14]: print("%-44s %.2e" % (k, errs[k]))
print("...")
for k in sorted(errs, key=errs.get)[:6]: print("%-44s %.2e" % (k, errs[k]))

# ---- which backward kernel is off?  log every _conv_k3_forward / _deconv_k3_forward / wgrad call of a second backward pass and check it in float64
import torch.nn.functional as F
M = sa.modules
log = []
real_c, real_d, real_w = M._conv_k3_forward, M._deconv_k3_forward, M.conv3d_wgrad_hip
def spy_c(x, w, stride):
    y = real_c(x, w, stride)
    ref = F.conv3d(x.double().cpu(), w.double().cpu(), None, stride, 1)
    log.append(("conv s%d %s->%s %s" % (stride, w.shape[1], w.shape[0], tuple(x.shape[2:])), float((y.double().cpu() - ref).abs().max()) / (float(ref.abs().max()) + 1e-300)))
    return y
def spy_d(x, w):
    y = real_d(x, w)
    ref = F.conv_transpose3d(x.double().cpu(), w.double().cpu(), None, stride=2, padding=1, output_padding=1)
    log.append(("deconv %s->%s %s" % (w.shape[0], w.shape[1], tuple(x.shape[2:])), float((y.double().cpu() - ref).abs().max()) / (float(ref.abs().max()) + 1e-300)))
    return y
M._conv_k3_forward, M._deconv_k3_forward = spy_c, spy_d
seg.zero_grad()
r = seg(dev(fl4), dev(fr4), dev(fl8), dev(fr8))
nfwd = len(log)
(r["pred"].mean() + r["pred_att"].mean()).backward()
print("forward calls", nfwd, "backward calls", len(log) - nfwd)
for i, (n, e) in enumerate(log):
    if e > 2e-6: print("  %s %-40s rel err %.2e" % ("fwd" if i < nfwd else "bwd", n, e))

# ---- do the top-2 picks of regression_topk agree with the oracle's?
M._conv_k3_forward, M._deconv_k3_forward = real_c, real_d
cap = {}
h = seg.classif.register_forward_hook(lambda m, a, o: cap.__setitem__("cost", o.detach().cpu().double()))
with torch.no_grad():
    seg(dev(fl4), dev(fr4), dev(fl8), dev(fr8))
h.remove()
oc = {}
with torch.no_grad(), ostack.training_mode():
    oseg.matching_branch({k: v.detach() for k, v in P64.items()}, fl4.double(), fr4.double(), att.detach(), smp.detach(), oc)
c_hip, c_or = cap["cost"].squeeze(1), oc["cost"].squeeze(1)
t_hip, t_or = c_hip.topk(2, dim=1).indices.sort(1).values, c_or.topk(2, dim=1).indices.sort(1).values
diff = (t_hip != t_or).any(1)
print("cost max diff %.2e; pixels whose top-2 SET differs from the oracle's: %d of %d" % (float((c_hip - c_or).abs().max()), int(diff.sum()), diff.numel()))
if int(diff.sum()):
    srt = c_or.sort(1, descending=True).values
    gap = (srt[:, 1] - srt[:, 2])[diff]
    print("  the oracle's 2nd-3rd cost gaps there:", [float(g_) for g_ in gap.flatten()[:8]])

# ---- the head's data gradient and the BatchNorm backward of classif.0, from the tensors of a third pass
store = {}
real_ct = M.conv3d_train
def spy_ct(conv, x):
    if conv.out_channels == 1 and x.shape[1] == 32 and x.shape[2] == 24:
        x.retain_grad(); store["y0"] = x; store["w2"] = conv.weight
        out = real_ct(conv, x); out.retain_grad(); store["out"] = out
        return out
    return real_ct(conv, x)
M.conv3d_train = spy_ct
seg.zero_grad()
r = seg(dev(fl4), dev(fr4), dev(fl8), dev(fr8))
(r["pred"].mean() + r["pred_att"].mean()).backward()
y0, out, w2 = store["y0"], store["out"], store["w2"]
g, gx = out.grad.double().cpu(), y0.grad.double().cpu()
gx64 = F.conv_transpose3d(g, w2.detach().double().cpu(), None, stride=1, padding=1)
print("head dgrad: max |gx - gx64| / max|gx64| = %.2e" % (float((gx - gx64).abs().max()) / float(gx64.abs().max())))
mask = (y0.detach().double().cpu() > 0)
gb_from_hip_gx, gb64 = (gx * mask).sum((0, 2, 3, 4)), (gx64 * mask).sum((0, 2, 3, 4))
gb_model = dict(seg.named_parameters())["classif.0.1.bias"].grad.double().cpu()
ref_gb = P64["classif.0.1.bias"].grad
sc = float(ref_gb.abs().max())
print("bias grad: model vs sum(hip gx * mask) %.2e; model vs sum(gx64 * mask) %.2e; model vs oracle %.2e; sum(gx64*mask) vs oracle %.2e" % (
    float((gb_model - gb_from_hip_gx).abs().max()) / sc, float((gb_model - gb64).abs().max()) / sc, float((gb_model - ref_gb).abs().max()) / sc, float((gb64 - ref_gb).abs().max()) / sc))
print("g (grad wrt cost): nonzeros %d of %d, max %.3e" % (int((g != 0).sum()), g.numel(), float(g.abs().max())))

# ---- the ReLU mask of classif.0 and the gradient entering the head, HIP vs oracle
with torch.no_grad(), ostack.training_mode():
    Pd = {k: v.detach() for k, v in P64.items()}
    y0_or = F.relu(ostack.convbn_3d(Pd, "classif.0", oc["hourglass"], 1, 1))
y0_hip = y0.detach().double().cpu()
mm = ((y0_hip > 0) != (y0_or > 0))
print("classif.0 output: max diff %.2e; ReLU mask mismatches %d of %d" % (float((y0_hip - y0_or).abs().max()), int(mm.sum()), mm.numel()))
# oracle's g: d(pred.mean())/d cost
c = oc["cost"].detach().clone().requires_grad_(True)
from oracle import ops as oops
oops.regression_topk(c.squeeze(1), smp.detach(), 2).mean().backward()
print("g: max |hip - oracle| / max = %.2e" % (float((g - c.grad).abs().max()) / float(c.grad.abs().max())))
