#!/usr/bin/env python3
"""Each layer's OWN error on the HIP path: every stage of the matching branch (models/SemStereo.py:314-323) and of the attention
branch (:273-285) is fed the float64 truth of its input (rounded to fp32) and compared with the float64 truth of its output --
no error is inherited from upstream, so the table says which kernels put the HIP path further from the exact answer than the
fp32 CPU oracle (the reference's arithmetic), which runs on the same fp32 inputs beside it.
Test tooling (imports oracle/, tests/golden).  usage: err_stages.py [fixture, default s256_md128_cal] [engine]
SS_TOOL_LIB=tools/_build/lib_<variant>.so selects an experimental build of the library."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from semstereo_amd import _lib  # noqa: E402
if os.environ.get("SS_TOOL_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["SS_TOOL_LIB"])
import semstereo_amd as sa  # noqa: E402
from semstereo_amd import deferred as _dfr  # noqa: E402
_dfr.ENABLED = False
from semstereo_amd import modules as M  # noqa: E402
from semstereo_amd import engine as sa_engine  # noqa: E402
from semstereo_amd import ops  # noqa: E402
from golden import cases  # noqa: E402
from oracle import hot_segment as oseg, ops as oops, stack  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "s256_md128_cal"
if len(sys.argv) > 2:
    sa_engine.CONV_ENGINE = sys.argv[2]
g = np.load(os.path.join(ROOT, "tests", "golden", "segment.npz"))
B, H, W, maxdisp = cases.segment_shape(name)
m4, m8 = maxdisp // 4, maxdisp // 8
P = cases.segment_params(name, g)
P64 = {k: (v.double() if v.is_floating_point() else v) for k, v in P.items()}
fl4, fr4, fl8, fr8, _ = cases.segment_inputs(name)
smp = torch.as_tensor(g[f"{name}/samples"].astype(np.float32))
att = torch.as_tensor(g[f"{name}/att_topk"]).unsqueeze(1)
seg = sa.HotSegment(maxdisp)
seg.load_state_dict(P, strict=False)
seg = seg.cuda().eval()
torch.set_num_threads(min(os.cpu_count() or 1, 32))
rows = []


def f32(t):
    return t.float()


def cu(t):
    return t.float().cuda().contiguous()


def report(stage, hip, o32, t64):
    n = t64.pow(2).mean().sqrt().item()
    eh = hip.detach().cpu().double().reshape(t64.shape) - t64
    eo = o32.double().reshape(t64.shape) - t64
    rh, ro = eh.pow(2).mean().sqrt().item() / n, eo.pow(2).mean().sqrt().item() / n
    rows.append((stage, rh, eh.abs().max().item() / n, ro, eo.abs().max().item() / n))
    print(f"{stage:34s} {rh:10.2e} {eh.abs().max().item() / n:10.2e} {ro:10.2e} {eo.abs().max().item() / n:10.2e}  {rh / max(ro, 1e-300):6.2f}", flush=True)


print(f"{name} engine {M.CONV_ENGINE} lib {os.path.basename(_lib.LIB_PATH)}: each stage on the float64 truth of its input")
print(f"{'stage':34s} {'hip rms':>10s} {'hip max':>10s} {'o32 rms':>10s} {'o32 max':>10s}  hip/o32   (relative to the rms of the float64 output)")
with torch.no_grad():
    # ---------------- matching branch ----------------
    cl64, cr64 = stack.concat_feature(P64, fl4.double()), stack.concat_feature(P64, fr4.double())
    y64 = stack.basic_conv(P64, "concat_feature.0", fl4.double(), is_3d=False, stride=1, pad=1)
    cf = seg.concat_feature
    y_h = M.run_conv2d(cf[0], "bc2d", cf[0].conv, cf[0].bn, fl4.cuda(), True)
    report("concat_feature.0 (2-D 128->64)", y_h, stack.basic_conv(P, "concat_feature.0", fl4, is_3d=False, stride=1, pad=1), y64)
    z_h = M.run_conv2d(cf, "cf1", cf[1], None, cu(y64), False)
    report("concat_feature.1 (2-D 64->32)", z_h, F.conv2d(f32(y64), P["concat_feature.1.weight"], None, 1, 1), cl64)
    gate64 = torch.sigmoid(F.conv2d(stack.basic_conv(P64, "concat_feature_att_4.im_att.0", fl4.double(), is_3d=False, stride=1, pad=0),
                                    P64["concat_feature_att_4.im_att.1.weight"], P64["concat_feature_att_4.im_att.1.bias"]))
    gate32 = torch.sigmoid(F.conv2d(stack.basic_conv(P, "concat_feature_att_4.im_att.0", fl4, is_3d=False, stride=1, pad=0),
                                    P["concat_feature_att_4.im_att.1.weight"], P["concat_feature_att_4.im_att.1.bias"]))
    gate_h = seg.concat_feature_att_4.logits(fl4.cuda(), sigmoid=True)
    report("channelAtt gate (sigmoid)", gate_h, gate32, gate64)
    # the stem by halves on the truth's features
    rw64, lb64 = oops.SpatialTransformer_grid(cl64, cr64, smp.double())
    w64 = P64["concat_stem.conv.weight"]
    left64 = F.conv3d(att.double() * lb64, w64[:, :32], None, 1, 1)
    right64 = F.conv3d(att.double() * rw64, w64[:, 32:], None, 1, 1)
    part_h = M.stem_broadcast_half(seg.concat_stem, cu(cl64), att.cuda())
    report("stem broadcast half (partial sum)", part_h, F.conv3d(att * f32(lb64), P["concat_stem.conv.weight"][:, :32], None, 1, 1), left64)
    right_h = ops.concat_volume_sampled(None, cu(cr64), smp.cuda(), att.cuda())
    report("warped half x att", right_h, att * oops.SpatialTransformer_grid(f32(cl64), f32(cr64), smp)[0], att.double() * rw64)
    vol64 = att.double() * torch.cat((lb64, rw64), dim=1)
    stem64 = stack.channel_att(P64, "concat_feature_att_4", stack.basic_conv(P64, "concat_stem", vol64, is_3d=True), fl4.double())
    stem_h = M.stem_volume_half(seg.concat_stem, cu(att.double() * rw64), cu(left64), cu(gate64))
    stem_o = stack.channel_att(P, "concat_feature_att_4", stack.basic_conv(P, "concat_stem", f32(vol64), is_3d=True), fl4)
    report("stem conv (right half) + BN + gate", stem_h, stem_o, stem64)

    def hourglass_stages(key, hg, x64, block):
        c1_64 = F.relu(stack.convbn_3d(P64, key + ".conv1.0", x64, 2, 1))
        report(key + ".conv1 s2", M.run_convbn(hg, "c1", hg.conv1[0][0], hg.conv1[0][1], cu(x64), relu=True),
               F.relu(stack.convbn_3d(P, key + ".conv1.0", f32(x64), 2, 1)), c1_64)
        c2_64 = F.relu(stack.convbn_3d(P64, key + ".conv2.0", c1_64, 1, 1))
        report(key + ".conv2", M.run_convbn(hg, "c2", hg.conv2[0][0], hg.conv2[0][1], cu(c1_64), relu=True),
               F.relu(stack.convbn_3d(P, key + ".conv2.0", f32(c1_64), 1, 1)), c2_64)
        c3_64 = F.relu(stack.convbn_3d(P64, key + ".conv3.0", c2_64, 2, 1))
        report(key + ".conv3 s2", M.run_convbn(hg, "c3", hg.conv3[0][0], hg.conv3[0][1], cu(c2_64), relu=True),
               F.relu(stack.convbn_3d(P, key + ".conv3.0", f32(c2_64), 2, 1)), c3_64)
        c4_64 = F.relu(stack.convbn_3d(P64, key + ".conv4.0", c3_64, 1, 1))
        report(key + ".conv4", M.run_convbn(hg, "c4", hg.conv4[0][0], hg.conv4[0][1], cu(c3_64), relu=True),
               F.relu(stack.convbn_3d(P, key + ".conv4.0", f32(c3_64), 1, 1)), c4_64)
        a64 = stack.attention_block(P64, key + ".attention_block", c4_64, block)
        report(key + ".attention_block", hg.attention_block(cu(c4_64)), stack.attention_block(P, key + ".attention_block", f32(c4_64), block), a64)
        c5_64 = F.relu(stack.deconvbn_3d(P64, key + ".conv5", a64) + stack.convbn_3d(P64, key + ".redir2", c2_64, 1, 0))
        report(key + ".conv5 deconv + redir2", hg._up("u5", hg.conv5, hg.redir2, cu(a64), cu(c2_64)),
               F.relu(stack.deconvbn_3d(P, key + ".conv5", f32(a64)) + stack.convbn_3d(P, key + ".redir2", f32(c2_64), 1, 0)), c5_64)
        c6_64 = F.relu(stack.deconvbn_3d(P64, key + ".conv6", c5_64) + stack.convbn_3d(P64, key + ".redir1", x64, 1, 0))
        report(key + ".conv6 deconv + redir1", hg._up("u6", hg.conv6, hg.redir1, cu(c5_64), cu(x64)),
               F.relu(stack.deconvbn_3d(P, key + ".conv6", f32(c5_64)) + stack.convbn_3d(P, key + ".redir1", f32(x64), 1, 0)), c6_64)
        return c6_64

    def classifier_stages(key, cls, x64):
        y64_ = F.relu(stack.convbn_3d(P64, key + ".0", x64, 1, 1))
        report(key + ".0 (32->32)", M.run_convbn(cls, "h0", cls[0][0], cls[0][1], cu(x64), relu=True),
               F.relu(stack.convbn_3d(P, key + ".0", f32(x64), 1, 1)), y64_)
        c64 = F.conv3d(y64_, P64[key + ".2.weight"], None, 1, 1)
        report(key + ".2 head (32->1)", M.run_convbn(cls, "h2", cls[2], None, cu(y64_), relu=False),
               F.conv3d(f32(y64_), P[key + ".2.weight"], None, 1, 1), c64)
        report(key + " whole (as run)", cls(cu(x64)), stack.classifier(P, key, f32(x64)), c64)
        return c64

    hg64 = hourglass_stages("hourglass", seg.hourglass, stem64, (6, 4, 4))
    report("hourglass whole (as run)", seg.hourglass(cu(stem64)), stack.hourglass(P, "hourglass", f32(stem64), (6, 4, 4)), hg64)
    cost64 = classifier_stages("classif", seg.classif, hg64)
    pred64 = oops.regression_topk(cost64.squeeze(1), smp.double(), 2)
    report("regression_topk", ops.regression_topk(cu(cost64).squeeze(1), smp.cuda(), 2), oops.regression_topk(f32(cost64).squeeze(1), smp, 2), pred64)
    # ---------------- attention branch up to the soft-max ----------------
    corr64 = stack.channel_att(P64, "corr_feature_att_8", stack.patch_conv(P64, oops.build_gwc_volume_norm(fl8.double(), fr8.double(), m8, 32)), fl8.double())
    corr_o = stack.channel_att(P, "corr_feature_att_8", stack.patch_conv(P, oops.build_gwc_volume_norm(fl8, fr8, m8, 32)), fl8)
    corr_h = ops.gwc_patch_gate(fl8.cuda(), fr8.cuda(), m8, 32, seg.patch.weight, seg.corr_feature_att_8.logits(fl8.cuda()), _range=ops.signed_range(m8))
    report("gwc volume -> patch -> gate", corr_h, corr_o, corr64)
    hga64 = hourglass_stages("hourglass_att", seg.hourglass_att, corr64, (4, 4, 4))
    ca64 = classifier_stages("classif_att_", seg.classif_att_, hga64)
    H4, W4 = fl4.shape[-2:]
    aw64 = F.interpolate(ca64, [2 * m4, H4, W4], mode="trilinear")
    p64 = F.softmax(aw64.squeeze(1), dim=1)
    aw_h, pred0_h, var_h = ops.upsample_softmax_regression(cu(ca64), m4, H4, W4, _range=ops.signed_range(m4))
    report("trilinear up-sampling (logits)", aw_h, F.interpolate(f32(ca64), [2 * m4, H4, W4], mode="trilinear"), aw64)
    report("soft-argmax of the up-sampled", pred0_h, oops.disparity_regression(F.softmax(F.interpolate(f32(ca64), [2 * m4, H4, W4], mode="trilinear").squeeze(1), dim=1), m4),
           oops.disparity_regression(p64, m4))
