#!/usr/bin/env python3
"""Runs ONE kernel of the path back to back (live shapes of the 1024 x 1024 / maxdisp 128 pair) so that rocprofv3 kernel-trace /
PMC passes see nothing else, and prints its time and algorithmic-byte rate.
usage: run_kernel.py <kernel> [batch] [iters]        kernels: gwc gwc_fused patch head head_att classif classif_cl classif_plain classif_att head_cl conv_s1_cl conv_mid conv_low conv_mid_att conv_low_att attn attn_att warp ssr ssr2048 strength topk
                                                               catt8 catt4 upsoft stem_left stem stem_gather stem_gather_smooth conv_s1 conv_s2 conv_s2_att deconv deconv5 deconv_att6 deconv_att5"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import semstereo_amd as sa  # noqa: E402

if os.environ.get("SS_TOOL_LIB"):          # experimental builds of the library (tools/_build)
    sa._lib.LIB_PATH = os.path.abspath(os.environ["SS_TOOL_LIB"])

name = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
dev = torch.device("cuda")
M, O = sa.modules, sa.ops
R = lambda *s: torch.randn(*s, device=dev)                                  # noqa: E731

if name in ("gwc", "gwc_fused", "patch"):
    fl, fr, gl = R(B, 256, 128, 128), R(B, 256, 128, 128), R(B, 32, 128, 128)
    patch = M.DepthwisePatch(32).to(dev).eval()
    from semstereo_amd import deferred as dfr              # (in inference the op hands out a deferred handle: ask for its value)
    vol = dfr.real(O.build_gwc_volume_norm(fl, fr, 16, 32))
    fn = {"gwc": lambda: dfr.real(O.build_gwc_volume_norm(fl, fr, 16, 32)),
          "gwc_fused": lambda: O.gwc_patch_gate(fl, fr, 16, 32, patch.weight, gl),
          "patch": lambda: patch(vol, gl)}[name]
    nbytes = {"gwc": 4.0 * B * (2 * 256 + 32 * 32) * 128 * 128, "gwc_fused": 4.0 * B * (2 * 256 + 32 + 32 * 32) * 128 * 128,
              "patch": 4.0 * B * (2 * 32 * 32 + 32) * 128 * 128}[name]
elif name in ("head", "head_att"):
    D, H = (24, 256) if name == "head" else (32, 128)
    x = torch.relu(R(B, 32, D, H, H))
    ws = M.pack_head_weight_bf16s(R(1, 32, 3, 3, 3))
    fn = lambda: M.conv3d_head_bf16s_hip(x, ws, None, None, False, 6)          # noqa: E731
    nbytes = 4.0 * B * 33 * D * H * H
elif name in ("classif", "classif_cl", "classif_plain", "classif_att"):   # the whole classifier: conv 32->32 + BN + ReLU, then the 32->1 head (one pass / channels-last hand-off / plain pair)
    sa.engine.CLASSIFIER_FUSED = name in ("classif", "classif_att")
    sa.engine.CLASSIFIER_CL = name != "classif_plain"
    cl = M.Classifier(32).to(dev).eval()
    x = torch.relu(R(B, 32, 24, 256, 256)) if name != "classif_att" else torch.relu(R(B, 32, 32, 128, 128))
    fn = lambda: cl(x)                                                         # noqa: E731
    nbytes = 4.0 * B * ((32 + 1) if sa.engine.CLASSIFIER_FUSED else (32 + 32 + 32 + 1)) * x[0, 0].numel()
elif name == "warp":
    cr, smp = R(B, 32, 256, 256), torch.randint(-32, 32, (B, 24, 256, 256), device=dev).float().sort(dim=1).values
    att = torch.rand(B, 24, 256, 256, device=dev)
    fn = lambda: O.concat_volume_sampled(None, cr, smp, att)                   # noqa: E731
    nbytes = 4.0 * B * (32 + 24 + 24 + 32 * 24) * 256 * 256
elif name in ("ssr", "ssr2048"):
    Hf = 1024 if name == "ssr" else 2048
    ssr = M.SSR_upsample(6).to(dev).eval()
    d, w, l = R(B, 1, Hf // 4, Hf // 4), R(B, 6, Hf, Hf), R(B, 6, Hf, Hf)
    from semstereo_amd import deferred as dfr              # (in inference the head hands out a deferred handle: ask for its value)
    fn = lambda: dfr.real(ssr(d, w, l))                                        # noqa: E731
    nbytes = 4.0 * B * (13 + 1.0 / 16) * Hf * Hf
elif name == "strength":
    fl, fr, p0, var = R(B, 128, 256, 256), R(B, 128, 256, 256), R(B, 256, 256) * 8, torch.rand(B, 1, 256, 256, device=dev)
    g, b = torch.full((1,), 0.25, device=dev), torch.full((1,), 2.0, device=dev)
    fn = lambda: O.sample_strength(fl, fr, p0, var, g, b)                      # noqa: E731
    nbytes = 4.0 * B * (2 * 128 + 2 + 5) * 256 * 256
elif name == "topk":
    aw, st = R(B, 1, 64, 256, 256), torch.softmax(R(B, 5, 256, 256), dim=1)
    fn = lambda: O.topk_candidates(aw, st, 32, 24)                             # noqa: E731
    nbytes = 4.0 * B * (64 + 5 + 24 + 24 + 1) * 256 * 256
elif name in ("catt8", "catt4"):
    c, h = (256, 128) if name == "catt8" else (128, 256)
    mod = M.channelAtt(32, c).to(dev).eval()
    im = R(B, c, h, h)
    fn = lambda: mod.logits(im)                                                # noqa: E731
    nbytes = 4.0 * B * (c + 32) * h * h
elif name == "upsoft":
    coarse = R(B, 1, 32, 128, 128)
    fn = lambda: O.upsample_softmax_regression(coarse, 32, 256, 256)           # noqa: E731
    nbytes = 4.0 * B * (32 * 128 * 128 + 66 * 256 * 256)
elif name == "stem_left":
    stem = M.BasicConv(64, 32, is_3d=True, kernel_size=3, stride=1, padding=1).to(dev).eval()
    cl, att = R(B, 32, 256, 256), torch.rand(B, 1, 24, 256, 256, device=dev)
    fn = lambda: M.stem_broadcast_half(stem, cl, att)                          # noqa: E731
    nbytes = 4.0 * B * (32 + 24 + 32 * 24) * 256 * 256
elif name in ("attn", "attn_att"):       # attention_block of hourglass2 ([128,6,64,64], windows 6x4x4) / of the attention-branch hourglass ([128,8,32,32], 4x4x4)
    blk, d, hw = ((6, 4, 4), 6, 64) if name == "attn" else ((4, 4, 4), 8, 32)
    ab = M.attention_block(128, 16, blk).to(dev).eval()
    x = R(B, 128, d, hw, hw)
    fn = lambda: ab(x)                                                         # noqa: E731
    nbytes = 4.0 * B * 2 * 128 * d * hw * hw
elif name in ("conv_mid", "conv_low", "conv_mid_att", "conv_low_att"):   # hourglass2.conv2: 64 -> 64 at [12,128,128]; conv4: 128 -> 128 at [6,64,64]; hourglass_att's: [16,64,64], [8,32,32]
    c, d, hw = {"conv_mid": (64, 12, 128), "conv_low": (128, 6, 64), "conv_mid_att": (64, 16, 64), "conv_low_att": (128, 8, 32)}[name]
    x = torch.relu(R(B, c, d, hw, hw))
    ws = M.pack_conv_weight_bf16s(R(c, c, 3, 3, 3) * 0.03, 19)
    sc, sh = torch.rand(c, device=dev) + 0.5, R(c) * 0.1
    fn = lambda: M.conv3d_bf16s_hip(x, ws, c, sc, sh, True, 19)               # noqa: E731
    nbytes = 4.0 * B * 2 * c * d * hw * hw
elif name == "head_cl":          # classif.2 reading the channels-last intermediate of its classifier
    xcl = torch.relu(R(B, 24, 256, 256, 32))
    hnt = int(os.environ.get("SS_TOOL_HEAD_NTERMS", str(M._head_nterms())))
    ws = M.pack_head_weight_bf16s(R(1, 32, 3, 3, 3), hnt)
    outh = torch.empty(B, 1, 24, 256, 256, device=dev)
    lib = sa._lib
    fn = lambda: lib.call("ss_conv3d_head_bf16s_cl_fwd", lib.ptr(xcl), lib.ptr(ws), None, None, lib.ptr(outh), B, 32, 24, 256, 256, 0, hnt)   # noqa: E731
    nbytes = 4.0 * B * 33 * 24 * 256 * 256
elif name == "conv_s1_cl":       # the stride-1 32 -> 32 conv with channels-last output (16-byte stores)
    x = torch.relu(R(B, 32, 24, 256, 256))
    ws = M.pack_conv_weight_bf16s(R(32, 32, 3, 3, 3) * 0.03, 19)
    sc, sh = torch.rand(32, device=dev) + 0.5, R(32) * 0.1
    mid = torch.empty(B, 24, 256, 256, 32, device=dev)
    lib = sa._lib
    fn = lambda: lib.call("ss_conv3d_bf16s_cl_fwd", lib.ptr(x), lib.ptr(ws), lib.ptr(sc), lib.ptr(sh), lib.ptr(mid), B, 32, 24, 256, 256, 32, 1, 19)   # noqa: E731
    nbytes = 4.0 * B * 2 * 32 * 24 * 256 * 256
elif name == "stem":             # the dominant launch: concat_stem on the warped half, continuing the broadcast half's partial sum, gated
    x = R(B, 32, 24, 256, 256)
    ws = M.pack_conv_weight_bf16s(R(32, 32, 3, 3, 3) * 0.03, 19)
    part, gate = R(B, 32, 24, 256, 256), torch.rand(B, 32, 256, 256, device=dev)
    sc, sh = torch.rand(32, device=dev) + 0.5, R(32) * 0.1
    fn = lambda: M.conv3d_bf16s_hip(x, ws, 32, sc, sh, True, 19, None, gate, partial=part)       # noqa: E731
    nbytes = 4.0 * B * (3 * 32 * 24 + 32) * 256 * 256
elif name in ("stem_gather", "stem_gather_smooth"):     # r05: the same launch with the warped half gathered inside its staging (ss_conv3d_gather_fwd)
    stem = M.BasicConv(64, 32, is_3d=True, kernel_size=3, stride=1, padding=1).to(dev).eval()
    cr = R(B, 32, 256, 256)
    if name == "stem_gather":        # 24 of 64 disparities drawn independently per pixel: the least coherent gather
        smp = torch.rand(B, 64, 256, 256, device=dev).argsort(dim=1)[:, :24].sort(dim=1).values.float() - 32.0    # (independent per pixel: many colliding lanes)
        if name == "warp_bwd_smooth":   # a band of 24 consecutive candidates around a smooth disparity map: what the top-24 of a trained attention looks like
            yy, xx = torch.meshgrid(torch.arange(256, device=dev), torch.arange(256, device=dev), indexing="ij")
            d0 = torch.round(10.0 * torch.sin(xx / 40.0) * torch.cos(yy / 55.0))
            smp = (d0[None, None] + torch.arange(-12, 12, device=dev).float()[None, :, None, None]).expand(B, 24, 256, 256).contiguous()
    else:                            # a window of 24 around a smooth disparity field: the most coherent one
        yy, xx = torch.meshgrid(torch.arange(256, device=dev), torch.arange(256, device=dev), indexing="ij")
        centre = (12.0 * torch.sin(xx / 40.0) * torch.cos(yy / 55.0)).round()
        smp = (centre.reshape(1, 1, 256, 256) + torch.arange(-12, 12, device=dev).reshape(1, 24, 1, 1)).expand(B, 24, 256, 256).contiguous().float()
    att = torch.rand(B, 1, 24, 256, 256, device=dev)
    part, gate = R(B, 32, 24, 256, 256), torch.rand(B, 32, 256, 256, device=dev)
    fn = lambda: M.stem_gather_half(stem, cr, smp, att, part, gate)           # noqa: E731
    nbytes = 4.0 * B * (32 + 24 + 24 + 2 * 32 * 24 + 32) * 256 * 256
elif name == "conv_s2_att":      # hourglass_att.conv3: 64 -> 128 stride 2 on [16,64,64] (256 workgroups: the one-tile-per-wave form)
    x = torch.relu(R(B, 64, 16, 64, 64))
    ws = M.pack_conv_weight_bf16s(R(128, 64, 3, 3, 3) * 0.03, 19)
    sc, sh = torch.rand(128, device=dev) + 0.5, R(128) * 0.1
    fn = lambda: M.conv3d_bf16s_hip(x, ws, 128, sc, sh, True, 19, stride=2)       # noqa: E731
    nbytes = 4.0 * B * (64 * 16 * 64 * 64 + 128 * 8 * 32 * 32)
elif name in ("deconv5", "deconv_att6", "deconv_att5"):     # the other three transposed convs of the step
    hg = (M.hourglass2(32) if name == "deconv5" else M.hourglass(32)).to(dev).eval()
    if name == "deconv5":        # hourglass2.conv5: 128 -> 64 from [6,64,64] + redir2 of the 64-channel [12,128,128] volume
        xin, xskip, args = torch.relu(R(B, 128, 6, 64, 64)), torch.relu(R(B, 64, 12, 128, 128)), ("u5", hg.conv5, hg.redir2)
    elif name == "deconv_att6":  # hourglass_att.conv6: 64 -> 32 from [16,64,64] + redir1 of [32,32,128,128]
        xin, xskip, args = torch.relu(R(B, 64, 16, 64, 64)), torch.relu(R(B, 32, 32, 128, 128)), ("u6", hg.conv6, hg.redir1)
    else:                        # hourglass_att.conv5: 128 -> 64 from [8,32,32] + redir2 of [64,16,64,64]
        xin, xskip, args = torch.relu(R(B, 128, 8, 32, 32)), torch.relu(R(B, 64, 16, 64, 64)), ("u5", hg.conv5, hg.redir2)
    fn = lambda: hg._up(*args, xin, xskip)                                     # noqa: E731
    nbytes = 4.0 * B * (xin[0].numel() + 2 * xskip[0].numel())
elif name in ("conv_s2", "conv_s1", "deconv"):
    if name == "deconv":            # hourglass2.conv6: 64 -> 32 to [24,256,256] with the 1x1x1 skip projection of a 32-channel volume
        hg = M.hourglass2(32).to(dev).eval()
        c5, x0 = torch.relu(R(B, 64, 12, 128, 128)), torch.relu(R(B, 32, 24, 256, 256))
        fn = lambda: hg._up("u6", hg.conv6, hg.redir1, c5, x0)               # noqa: E731
        nbytes = 4.0 * B * (64 * 12 * 128 * 128 + 2 * 32 * 24 * 256 * 256)
    else:
        stride = 2 if name == "conv_s2" else 1
        cout = 64 if stride == 2 else 32
        x = torch.relu(R(B, 32, 24, 256, 256))
        ws = M.pack_conv_weight_bf16s(R(cout, 32, 3, 3, 3) * 0.03, 19)
        sc, sh = torch.rand(cout, device=dev) + 0.5, R(cout) * 0.1
        fn = lambda: M.conv3d_bf16s_hip(x, ws, cout, sc, sh, True, 19, stride=stride)       # noqa: E731
        nbytes = 4.0 * B * (32 * 24 * 256 * 256 + cout * (24 // stride) * (256 // stride) ** 2)
elif name in ("strength_bwd", "strength_bwd_ws", "warp_bwd", "warp_bwd_smooth"):     # the two scatter kernels of the training step at 1024^2 (quarter resolution 256 x 256)
    lib = sa._lib
    if name.startswith("strength_bwd"):          # backward of the 5-candidate probe (models/SemStereo.py:286-293): 128-channel features
        fl, fr = R(B, 128, 256, 256), R(B, 128, 256, 256)
        yy, xx = torch.meshgrid(torch.arange(256, device=dev), torch.arange(256, device=dev), indexing="ij")
        p0 = (10.0 * torch.sin(xx / 40.0) * torch.cos(yy / 55.0)).reshape(1, 256, 256).expand(B, 256, 256).contiguous()
        var, g = torch.rand(B, 1, 256, 256, device=dev), R(B, 5, 256, 256)
        gm, bt = torch.full((1,), 0.25, device=dev), torch.full((1,), 2.0, device=dev)
        gl, gr, gp, gv, ggb = torch.empty_like(fl), torch.empty_like(fr), torch.empty_like(p0), torch.empty_like(var), torch.empty(2, device=dev)
        fn = lambda: lib.call("ss_sample_strength_bwd", lib.ptr(fl), lib.ptr(fr), lib.ptr(p0), lib.ptr(var), lib.ptr(gm), lib.ptr(bt), lib.ptr(g),   # noqa: E731
                              lib.ptr(gl), lib.ptr(gr), lib.ptr(gp), lib.ptr(gv), lib.ptr(ggb), B, 128, 256, 256)
        if name == "strength_bwd_ws":   # the form the training step runs (SS_SSB_TWO_LAUNCHES=1): two launches over a [B,5,H,W] scratch
            work = torch.empty(B, 5, 256, 256, device=dev)
            fn = lambda: lib.call("ss_sample_strength_bwd_ws", lib.ptr(fl), lib.ptr(fr), lib.ptr(p0), lib.ptr(var), lib.ptr(gm), lib.ptr(bt), lib.ptr(g),   # noqa: E731
                                  lib.ptr(gl), lib.ptr(gr), lib.ptr(gp), lib.ptr(gv), lib.ptr(ggb), lib.ptr(work), B, 128, 256, 256)
        nbytes = 4.0 * B * (4 * 128 + 8) * 256 * 256
    else:                               # backward of SpatialTransformer_grid at :316 (32 channels, 24 integer candidates)
        y = R(B, 32, 256, 256)
        smp = torch.rand(B, 64, 256, 256, device=dev).argsort(dim=1)[:, :24].sort(dim=1).values.float() - 32.0    # (independent per pixel: many colliding lanes)
        if name == "warp_bwd_smooth":   # a band of 24 consecutive candidates around a smooth disparity map: what the top-24 of a trained attention looks like
            yy, xx = torch.meshgrid(torch.arange(256, device=dev), torch.arange(256, device=dev), indexing="ij")
            d0 = torch.round(10.0 * torch.sin(xx / 40.0) * torch.cos(yy / 55.0))
            smp = (d0[None, None] + torch.arange(-12, 12, device=dev).float()[None, :, None, None]).expand(B, 24, 256, 256).contiguous()
        g = R(B, 32, 24, 256, 256)
        gy = torch.empty_like(y)
        fn = lambda: lib.call("ss_warp_sampled_bwd", lib.ptr(g), None, lib.ptr(y), lib.ptr(smp), None, lib.ptr(gy), None, B, 32, 256, 256, 24)   # noqa: E731
        nbytes = 4.0 * B * (32 * 24 + 24 + 32) * 256 * 256
elif name == "concat_bwd":           # backward of the gated sparse concat volume (models/SemStereo.py:316-318) at the 1024^2 training shape
    lib = sa._lib
    cl, cr = R(B, 32, 256, 256), R(B, 32, 256, 256)
    yy, xx = torch.meshgrid(torch.arange(256, device=dev), torch.arange(256, device=dev), indexing="ij")
    d0 = torch.round(10.0 * torch.sin(xx / 40.0) * torch.cos(yy / 55.0))
    smp = (d0[None, None] + torch.arange(-12, 12, device=dev).float()[None, :, None, None]).expand(B, 24, 256, 256).contiguous()
    att = torch.rand(B, 24, 256, 256, device=dev)
    g = R(B, 64, 24, 256, 256)
    gl, gr, ga = torch.empty_like(cl), torch.empty_like(cr), torch.empty_like(att)
    fn = lambda: lib.call("ss_concat_sampled_bwd", lib.ptr(g), lib.ptr(cl), lib.ptr(cr), lib.ptr(smp), lib.ptr(att), lib.ptr(gl), lib.ptr(gr), lib.ptr(ga),   # noqa: E731
                          B, 32, 256, 256, 24, 16)
    nbytes = 4.0 * B * (64 * 24 + 24 * 3 + 32 * 4) * 256 * 256
elif name.startswith("wgrad"):   # weight gradients of the training step (1024^2 / md64): wgrad = classif.0 (32 -> 32 at [24,256,256]), wgrad_stem = concat_stem (64 -> 32),
    # wgrad_mid = hourglass2.conv2 (64 -> 64 at [12,128,128]), wgrad_low = conv4 (128 -> 128 at [6,64,64]), wgrad_s2 = conv1 (32 -> 64, stride 2), wgrad_head (32 -> 1)
    from semstereo_amd import train_layers as TL
    cin, cout, d, hw, stride = {"wgrad": (32, 32, 24, 256, 1), "wgrad_stem": (64, 32, 24, 256, 1), "wgrad_mid": (64, 64, 12, 128, 1), "wgrad_low": (128, 128, 6, 64, 1),
                                "wgrad_s2": (32, 64, 24, 256, 2), "wgrad_s2_low": (64, 128, 12, 128, 2), "wgrad_head": (32, 1, 24, 256, 1)}[name]
    x = R(B, cin, d, hw, hw)
    g = R(B, cout, d // stride, hw // stride, hw // stride)
    fn = lambda: TL.conv3d_wgrad_hip(g, x, cout, cin, stride)                  # noqa: E731
    nbytes = 4.0 * (x.numel() + g.numel())
    print(f"{name}: {2.0 * 27 * cin * cout * g[0, 0].numel() * B / 1e9:.1f} GFLOP fp32-equivalent (x 6 bf16 products)")
else:
    sys.exit(__doc__)

with torch.no_grad():
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.time()
    while (time.time() - t0) * 1e3 < float(os.environ.get("SS_WARM_MS", "300")):     # loaded clocks
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
print(f"{name} B={B}: {ms * 1e3:.1f} us/launch, {nbytes / 1e6:.1f} MB algorithmic, {nbytes / ms / 1e6:.0f} GB/s = "
      f"{100 * nbytes / ms / 1e6 / 8000:.1f} % of 8 TB/s")
