#!/usr/bin/env python3
"""Error of every conv engine against a float64 convolution of the same fp32 operands, and its time, on the layer
shapes of the hot path and on inputs that stress the fp16 form's block-floating scales (tiny / huge tensors, channels
of very different magnitude, a partial sum as initial accumulator).
usage: check_engines.py [small]"""
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semstereo_amd import _lib  # noqa: E402
if os.environ.get("SS_TOOL_LIB"):          # experimental builds of the library (tools/build_variant.sh)
    _lib.LIB_PATH = os.path.abspath(os.environ["SS_TOOL_LIB"])
from semstereo_amd import modules as M  # noqa: E402

small = len(sys.argv) > 1 and sys.argv[1] == "small"
dev = torch.device("cuda")
g = torch.Generator(device="cuda").manual_seed(0)
NT = {"bf16x6": 6, "bf16x3": 3, "f16x3": 19}


def timed(fn):
    t0 = time.time()
    while time.time() - t0 < 0.3:
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10 * 1e3


def case(name, Cin, Cout, D, H, W, stride=1, in_mul=None, partial=False):
    x = torch.relu(torch.randn(1, Cin, D, H, W, device=dev, generator=g))
    if in_mul is not None:
        x = x * in_mul.to(dev).reshape(1, -1, 1, 1, 1)
    w = torch.randn(Cout, Cin, 3, 3, 3, device=dev, generator=g) * (2.0 / (Cin * 27)) ** 0.5
    sc, sh = torch.rand(Cout, device=dev, generator=g) + 0.5, torch.randn(Cout, device=dev, generator=g) * 0.1
    truth = F.conv3d(x.double(), w.double(), stride=stride, padding=1)
    part = None
    if partial:
        part = torch.randn(truth.shape, device=dev, generator=g)
        truth = truth + part.double()
    truth = torch.relu(truth * sc.double().reshape(1, -1, 1, 1, 1) + sh.double().reshape(1, -1, 1, 1, 1))
    denom = float(truth.pow(2).mean().sqrt())
    row = []
    for eng in ("f32", "bf16x6", "bf16x3", "f16x3"):
        if eng == "f32":
            if partial:
                continue
            wp = M.pack_conv_weight(w)
            fn = lambda: M.conv3d_hip(x, wp, sc, sh, 3, stride, True)
        else:
            ws = M.pack_conv_weight_bf16s(w, NT[eng])
            fn = lambda: M.conv3d_bf16s_hip(x, ws, Cout, sc, sh, True, NT[eng], partial=part, stride=stride)
        d = (fn().double() - truth)
        row.append(f"{eng} rms {float(d.pow(2).mean().sqrt()) / denom:.2e} max {float(d.abs().max()) / denom:.2e} {timed(fn):7.1f} us")
    print(f"{name:34s} " + " | ".join(row), flush=True)


def deconv_case(name, Cin, Cout, D, H, W, Cs, in_mul=1.0):
    x = torch.relu(torch.randn(1, Cin, D, H, W, device=dev, generator=g)) * in_mul
    w = torch.randn(Cin, Cout, 3, 3, 3, device=dev, generator=g) * (8.0 / (Cin * 27)) ** 0.5
    sh = torch.randn(Cout, device=dev, generator=g) * 0.1
    truth = F.conv_transpose3d(x.double(), w.double(), stride=2, padding=1, output_padding=1)
    skip = wsk = None
    if Cs:
        skip = torch.relu(torch.randn(1, Cs, 2 * D, 2 * H, 2 * W, device=dev, generator=g))
        wsk = torch.randn(Cs, Cout, device=dev, generator=g) * (1.0 / Cs) ** 0.5
        truth = truth + torch.einsum("bcdhw,co->bodhw", skip.double(), wsk.double())
    truth = torch.relu(truth + sh.double().reshape(1, -1, 1, 1, 1))
    denom = float(truth.pow(2).mean().sqrt())
    wp = M.pack_conv_weight(w, transposed=True)
    row = []
    for eng in ("f32", "bf16x6", "bf16x3", "f16x3"):
        if eng == "f32":
            fn = lambda: M.deconv3d_hip(x, wp, sh, True, skip, wsk)
        else:
            wds = M.pack_deconv_weight_bf16s(wp, NT[eng])
            wss = M.pack_deconv_weight_bf16s(wsk) if Cs else None
            fn = lambda: M.deconv3d_bf16s_hip(x, wds, Cout, sh, True, NT[eng], skip, wss)
        d = (fn().double() - truth)
        row.append(f"{eng} rms {float(d.pow(2).mean().sqrt()) / denom:.2e} max {float(d.abs().max()) / denom:.2e} {timed(fn):7.1f} us")
    print(f"{name:34s} " + " | ".join(row), flush=True)


k = 2 if small else 1
deconv_case("hg2 conv6 deconv 64->32 + skip", 64, 32, 12, 128 // k, 128 // k, 32)
deconv_case("hg2 conv5 deconv 128->64 + skip", 128, 64, 6, 64, 64, 64)
deconv_case("deconv 24->40 odd, no skip", 24, 40, 3, 9, 35, 0)
deconv_case("deconv x 1e-5 + skip O(1)", 32, 32, 3, 9, 35, 16, in_mul=1e-5)
deconv_case("deconv x 1e+5 + skip O(1)", 32, 32, 3, 9, 35, 16, in_mul=1e5)
case("stem right half 32->32", 32, 32, 24, 256 // k, 256 // k)
case("stem right half + partial", 32, 32, 24, 256 // k, 256 // k, partial=True)
case("classif 32->32", 32, 32, 24, 256 // k, 256 // k)
case("hg conv1 32->64 s2", 32, 64, 24, 256 // k, 256 // k, stride=2)
case("hg conv2 64->64", 64, 64, 12, 128 // k, 128 // k)
case("hg conv3 64->128 s2", 64, 128, 12, 128 // k, 128 // k, stride=2)
case("hg conv4 128->128", 128, 128, 6, 64, 64)
case("odd 20->40 [5,33,70]", 20, 40, 5, 33, 70)
case("tensor x 1e-6", 32, 32, 6, 64, 64, in_mul=torch.full((32,), 1e-6))
case("tensor x 1e+6", 32, 32, 6, 64, 64, in_mul=torch.full((32,), 1e6))
case("tensor x 1e-30", 32, 32, 6, 64, 64, in_mul=torch.full((32,), 1e-30))
case("channels x 10^(-6..6)", 32, 32, 6, 64, 64, in_mul=10.0 ** torch.linspace(-6, 6, 32))
case("channels x 10^(6..-6)", 32, 32, 6, 64, 64, in_mul=10.0 ** torch.linspace(6, -6, 32))
