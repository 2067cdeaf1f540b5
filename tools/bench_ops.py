#!/usr/bin/env python3
"""Per-kernel micro-benchmarks on the MI355X (development tool, not part of the product path):
every 3-D layer of the hot segment at the BASELINE shapes with its achieved TFLOP/s against the
fp32-MFMA peak, and the bandwidth kernels with achieved GB/s against HBM peak.

    python tools/bench_ops.py [--batch B] [--size 1024] [--maxdisp 128] [--iters 10] [--only conv|bw|all]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import semstereo_amd as sa  # noqa: E402
from semstereo_amd import deferred as _dfr  # noqa: E402
_dfr.ENABLED = False          # these tools time / inspect each op by itself: no deferred handles
from semstereo_amd import modules as M  # noqa: E402
from semstereo_amd import engine as sa_engine  # noqa: E402

PEAK_TF, PEAK_GBS = 157.3, 8000.0


def timeit(fn, iters):
    import time
    t0 = time.time()
    while time.time() - t0 < 0.3:          # a chip coming out of idle runs ~20 % slower: measure at loaded clocks
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters      # ms


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--maxdisp", type=int, default=128)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--only", default="all")
    ap.add_argument("--json", default=None)
    ap.add_argument("--sweep", action="store_true", help="time every conv tile candidate (SS_CONV_TILE)")
    ap.add_argument("--engine", default=None, help="f32 | bf16x6 | bf16x3 for the 3x3x3 stride-1 convs")
    args = ap.parse_args()
    B, S, md = args.batch, args.size, args.maxdisp
    if args.engine:
        sa_engine.CONV_ENGINE = args.engine
    dev = torch.device("cuda")
    rows = []

    def conv_case(name, cin, cout, d, h, w, k, stride):
        x = torch.randn(B, cin, d, h, w, device=dev)
        wt = torch.randn(cout, cin, k, k, k, device=dev) * (1.0 / (cin * k ** 3)) ** 0.5
        wp = M.pack_conv_weight(wt)
        sc, sh = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev) * 0.1
        do, ho, wo = [(n + 2 * (k // 2) - k) // stride + 1 for n in (d, h, w)]
        gf = 2.0 * B * cout * cin * k ** 3 * do * ho * wo / 1e9
        if args.sweep and cout > 1:
            for tile in range(5 if stride == 1 else 4):
                os.environ["SS_CONV_TILE"] = str(tile)
                sa._lib.load().ss_reload_tuning()
                ms = timeit(lambda: M.conv3d_hip(x, wp, sc, sh, k, stride, True), args.iters)
                rows.append(dict(name=f"{name} [tile {tile}]", ms=ms, gflop=gf, tflops=gf / ms, frac=gf / ms / PEAK_TF))
            os.environ.pop("SS_CONV_TILE", None)
            sa._lib.load().ss_reload_tuning()
            return
        if M.CONV_ENGINE != "f32" and k == 3 and stride == 1 and cout > 1:
            ws = M.pack_conv_weight_bf16s(wt)
            nterms = 6 if M.CONV_ENGINE == "bf16x6" else 3
            ms = timeit(lambda: M.conv3d_bf16s_hip(x, ws, cout, sc, sh, True, nterms), args.iters)
            rows.append(dict(name=f"{name} [{M.CONV_ENGINE}]", ms=ms, gflop=gf, tflops=gf / ms, frac=gf / ms / PEAK_TF))
            return
        if M.CONV_ENGINE != "f32" and k == 3 and stride == 1 and cout == 1 and cin in (16, 32, 64):
            ws = M.pack_head_weight_bf16s(wt)
            nterms = 6 if M.CONV_ENGINE == "bf16x6" else 3
            ms = timeit(lambda: M.conv3d_head_bf16s_hip(x, ws, sc, sh, False, nterms), args.iters)
            rows.append(dict(name=f"{name} [{M.CONV_ENGINE} head]", ms=ms, gflop=gf, tflops=gf / ms, frac=gf / ms / PEAK_TF))
            return
        ms = timeit(lambda: M.conv3d_hip(x, wp, sc, sh, k, stride, True), args.iters)
        rows.append(dict(name=name, ms=ms, gflop=gf, tflops=gf / ms, frac=gf / ms / PEAK_TF))

    def deconv_case(name, cin, cout, d, h, w, cs):
        x = torch.randn(B, cin, d, h, w, device=dev)
        wt = torch.randn(cin, cout, 3, 3, 3, device=dev) * (8.0 / (cin * 27)) ** 0.5
        wp = M.pack_conv_weight(wt, transposed=True)
        sh = torch.randn(cout, device=dev) * 0.1
        skip = torch.randn(B, cs, 2 * d, 2 * h, 2 * w, device=dev)
        ws = torch.randn(cs, cout, device=dev) * (1.0 / cs) ** 0.5
        ms = timeit(lambda: M.deconv3d_hip(x, wp, sh, True, skip, ws), args.iters)
        gf = 2.0 * B * cout * (cin * 27 * d * h * w + cs * 8 * d * h * w) / 1e9
        rows.append(dict(name=name, ms=ms, gflop=gf, tflops=gf / ms, frac=gf / ms / PEAK_TF))

    d8, h8 = 2 * (md // 8), S // 8
    k4, h4 = 24, S // 4
    if args.only in ("all", "conv"):
        for tag, d, h in (("att", d8, h8), ("hg2", k4, h4)):
            conv_case(f"{tag}.conv1 s2 32->64", 32, 64, d, h, h, 3, 2)
            conv_case(f"{tag}.conv2 64->64", 64, 64, d // 2, h // 2, h // 2, 3, 1)
            conv_case(f"{tag}.conv3 s2 64->128", 64, 128, d // 2, h // 2, h // 2, 3, 2)
            conv_case(f"{tag}.conv4 128->128", 128, 128, d // 4, h // 4, h // 4, 3, 1)
            deconv_case(f"{tag}.conv5 deconv 128->64 +redir2", 128, 64, d // 4, h // 4, h // 4, 64)
            deconv_case(f"{tag}.conv6 deconv 64->32 +redir1", 64, 32, d // 2, h // 2, h // 2, 32)
            conv_case(f"{tag}.classif.0 32->32", 32, 32, d, h, h, 3, 1)
            conv_case(f"{tag}.classif.2 32->1", 32, 1, d, h, h, 3, 1)
        conv_case("concat_stem 64->32", 64, 32, k4, h4, h4, 3, 1)
        # attention blocks
        for tag, d, h, blk in (("att", d8 // 4, h8 // 4, (4, 4, 4)), ("hg2", k4 // 4, h4 // 4, (6, 4, 4))):
            ab = M.attention_block(128, 16, blk).to(dev).eval()
            x = torch.randn(B, 128, d, h, h, device=dev)
            with torch.no_grad():
                ms = timeit(lambda: ab(x), args.iters)
            T = blk[0] * blk[1] * blk[2]
            ntok = B * d * h * h
            gf = ntok * (2 * 128 * 384 + 2 * 128 * 128 + 16 * 2 * 2 * T * 8) / 1e9
            rows.append(dict(name=f"{tag}.attention_block {blk}", ms=ms, gflop=gf, tflops=gf / ms, frac=gf / ms / PEAK_TF))

    def bw_row(name, ms, nbytes):
        rows.append(dict(name=name, ms=ms, mbytes=nbytes / 1e6, gbs=nbytes / ms / 1e6, frac=nbytes / ms / 1e6 / PEAK_GBS))

    if args.only in ("all", "bw"):
        fl8, fr8 = torch.randn(B, 256, h8, h8, device=dev), torch.randn(B, 256, h8, h8, device=dev)
        ms = timeit(lambda: sa.ops.build_gwc_volume_norm(fl8, fr8, md // 8, 32), args.iters)
        bw_row("build_gwc_volume_norm (live)", ms, 4.0 * B * (2 * 256 + 32 * d8) * h8 * h8)
        ms = timeit(lambda: sa.ops.build_gwc_volume(fl8, fr8, md // 8, 32), args.iters)
        bw_row("build_gwc_volume (live shape)", ms, 4.0 * B * (2 * 256 + 32 * d8) * h8 * h8)
        cl, cr = torch.randn(B, 32, h4, h4, device=dev), torch.randn(B, 32, h4, h4, device=dev)
        ms = timeit(lambda: sa.ops.build_concat_volume(cl, cr, md // 4), max(2, args.iters // 2))
        bw_row("build_concat_volume (dense, m4)", ms, 4.0 * B * (2 * 32 + 64 * 2 * (md // 4)) * h4 * h4)
        samples = torch.sort(torch.rand(B, 2 * (md // 4), h4, h4, device=dev).argsort(dim=1)[:, :24], dim=1)[0].float() - md // 4
        att = torch.rand(B, 1, 24, h4, h4, device=dev)
        ms = timeit(lambda: sa.ops.concat_volume_sampled(cl, cr, samples, att), args.iters)
        bw_row("concat_volume_sampled (live, fused gate)", ms, 4.0 * B * (64 + 24 + 24 + 64 * 24) * h4 * h4)
        prob = torch.softmax(torch.randn(B, 2 * (md // 4), h4, h4, device=dev), dim=1)
        ms = timeit(lambda: sa.ops.disparity_regression(prob, md // 4), args.iters)
        bw_row("disparity_regression", ms, 4.0 * B * (2 * (md // 4) + 1) * h4 * h4)
        ms = timeit(lambda: sa.ops.softmax_regression(prob, md // 4), args.iters)
        bw_row("softmax_regression (fused)", ms, 4.0 * B * (2 * (md // 4) + 2) * h4 * h4)
        cost = torch.randn(B, 24, h4, h4, device=dev)
        ms = timeit(lambda: sa.ops.regression_topk(cost, samples, 2), args.iters)
        bw_row("regression_topk k=2", ms, 4.0 * B * (2 * 24 + 1) * h4 * h4)
        fl4, fr4 = torch.randn(B, 128, h4, h4, device=dev), torch.randn(B, 128, h4, h4, device=dev)
        d5 = torch.randn(B, 5, h4, h4, device=dev) * 8
        ms = timeit(lambda: sa.ops.warp_correlation(fl4, fr4, d5), args.iters)
        bw_row("warp_correlation (5 samples, 128 ch)", ms, 4.0 * B * (2 * 128 + 5 + 5) * h4 * h4)
        vol = torch.randn(B, 32, d8, h8, h8, device=dev)
        gate = torch.randn(B, 32, h8, h8, device=dev)
        patch = M.DepthwisePatch(32).to(dev).eval()
        with torch.no_grad():
            ms = timeit(lambda: patch(vol, gate), args.iters)
        bw_row("patch (1,3,3) depthwise + gate", ms, 4.0 * B * (2 * 32 * d8 + 32) * h8 * h8)
        vol4 = torch.randn(B, 32, 24, h4, h4, device=dev)
        gate4 = torch.randn(B, 32, h4, h4, device=dev)
        ms = timeit(lambda: sa.ops.channel_gate(gate4, vol4), args.iters)
        bw_row("channel_gate [B,32,24,H4,W4]", ms, 4.0 * B * (2 * 32 * 24 + 32) * h4 * h4)

    for r in rows:
        if "tflops" in r:
            print(f"{r['name']:44s} {r['ms']*1e3:9.1f} us  {r['gflop']:8.2f} GF  {r['tflops']:7.2f} TF/s  {100*r['frac']:5.1f}% of fp32-MFMA peak")
        else:
            print(f"{r['name']:44s} {r['ms']*1e3:9.1f} us  {r['mbytes']:8.1f} MB  {r['gbs']:7.0f} GB/s  {100*r['frac']:5.1f}% of HBM peak")
    tot = sum(r["ms"] for r in rows if "tflops" in r)
    gf = sum(r["gflop"] for r in rows if "tflops" in r)
    if tot:
        print(f"3-D stack total: {tot:.3f} ms, {gf:.1f} GFLOP, {gf/tot:.1f} TF/s")
    if args.json:
        with open(args.json, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
