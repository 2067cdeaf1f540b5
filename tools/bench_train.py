#!/usr/bin/env python3
"""The TRAINING step of the hot segment, measured: forward + backward + optimizer step of
`HotSegment.train()` -- BatchNorm on batch statistics, autograd on, gradients to every parameter AND to the four feature maps (the
reference trains its backbone through them) -- at the size the reference trains at (/root/reference/main_us3d.py:54, 74, 186-222:
1024 x 1024 tiles, maxdisp 64, batch 4, Adam lr 1e-3).  Per batch size: ms per step (HIP events around `steps` steps after `warmup`),
peak allocator memory, finite-gradient and determinism checks, PATH_COUNTS["torch"] unchanged (no PyTorch layer in the 3-D stack).
Synthetic inputs (bench.py's `synth_features`), random-init weights at unit gain, smooth-L1 losses on `pred` and `pred_att` against a
synthetic ground truth as `model_loss_train` does for the two 1/4-scale outputs the segment owns.

usage: python tools/bench_train.py [--batches 1,2,4] [--height 1024 --width 1024 --maxdisp 64] [--steps 5 --warmup 2] [--out gpurun_out/bench_train.json]
       rocprofv3 --kernel-trace --stats ... -- python3 tools/bench_train.py --batches 1 --steps 3 --no-checks      (per-kernel breakdown)
With --kernel-stats <csv> (a rocprofv3 kernel_stats.csv of such a run) it only prints the breakdown by kernel family and the
weight-gradient / data-gradient kernels' rates."""
import argparse
import csv
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def conv_layers(H, W, maxdisp):
    """(name, kind, Cin, Cout, output positions) of every 3x3x3 layer of the hot segment at this size: the flops of a layer's forward
    -- 2 * 27 * Cin * Cout * positions (transposed: input positions) -- are also those of its data gradient and of its weight gradient."""
    H8, W8, H4, W4 = H // 8, W // 8, H // 4, W // 4
    D8, k = 2 * (maxdisp // 8), 24
    L = []

    def hourglass(pre, D, Hh, Ww):
        v1, v2 = (D // 2) * (Hh // 2) * (Ww // 2), (D // 4) * (Hh // 4) * (Ww // 4)
        L.extend([(pre + ".conv1", "s2", 32, 64, v1), (pre + ".conv2", "s1", 64, 64, v1), (pre + ".conv3", "s2", 64, 128, v2),
                  (pre + ".conv4", "s1", 128, 128, v2), (pre + ".conv5", "T", 128, 64, v2), (pre + ".conv6", "T", 64, 32, v1)])
    hourglass("hourglass_att", D8, H8, W8)
    L.append(("classif_att_.0", "s1", 32, 32, D8 * H8 * W8))
    L.append(("classif_att_.2", "head", 32, 1, D8 * H8 * W8))
    L.append(("concat_stem", "s1", 64, 32, k * H4 * W4))
    hourglass("hourglass", k, H4, W4)
    L.append(("classif.0", "s1", 32, 32, k * H4 * W4))
    L.append(("classif.2", "head", 32, 1, k * H4 * W4))
    return L


def family(name):
    table = (("conv3d_wgrad", "weight gradients (3x3x3)"), ("conv_wgrad_k1", "weight gradients (1x1)"), ("deconv3d", "transposed convs (fwd + dgrad of stride-2 convs)"),
             ("conv3d_bf16s<2", "stride-2 convs (fwd + dgrad of transposed convs)"), ("conv3d_bf16s", "stride-1 3-D / 2-D convs (fwd + dgrad)"),
             ("conv3d_head", "32->1 heads / 1x1x1 projections"), ("pointwise", "32->1 heads / 1x1x1 projections"), ("conv3d_mfma", "exact-fp32 convs (k = 1, fwd + bwd)"),
             ("conv3d_k", "exact-fp32 convs (k = 1, fwd + bwd)"), ("channel_reduce", "BatchNorm statistics / bias gradients"), ("bn_", "BatchNorm apply fwd / bwd"),
             ("window_attention", "windowed attention fwd / bwd"), ("warp", "warp fwd / bwd"), ("gwc", "cost volume fwd / bwd"), ("at::", "PyTorch glue (losses, optimizer, adds)"),
             ("elementwise", "PyTorch glue (losses, optimizer, adds)"), ("vectorized", "PyTorch glue (losses, optimizer, adds)"), ("multi_tensor", "PyTorch glue (losses, optimizer, adds)"),
             ("reduce_kernel", "PyTorch glue (losses, optimizer, adds)"))
    for key, fam in table:
        if key in name:
            return fam
    return "other HIP kernels of the path (tails, gates, regressions, packs)"


def kernel_stats_report(path, H, W, maxdisp, batch, steps):
    rows = list(csv.DictReader(open(path)))
    tot = {}
    for r in rows:
        ns = float(r.get("TotalDurationNs") or r.get("Total Duration (ns)") or 0)
        f = family(r["Name"])
        a = tot.setdefault(f, [0.0, 0])
        a[0] += ns
        a[1] += int(r.get("Calls") or 0)
    allns = sum(a[0] for a in tot.values())
    print(f"kernel time of the whole run {allns / 1e6:.1f} ms (warm-up, {steps} timed steps and weight packing); by family:")
    for f, (ns, calls) in sorted(tot.items(), key=lambda kv: -kv[1][0]):
        print(f"  {ns / 1e6:9.2f} ms  {100 * ns / allns:5.1f} %  {calls:6d} calls  {f}")
    L = conv_layers(H, W, maxdisp)
    flops = sum(2.0 * 27 * ci * co * v for _, kind, ci, co, v in L if kind != "head") * batch
    print(f"3x3x3 layers: {flops / 1e9:.1f} GFLOP per forward at batch {batch} (= per data-gradient pass = per weight-gradient pass)")
    for r in sorted(rows, key=lambda r_: -float(r_.get("TotalDurationNs") or 0))[:14]:
        print(f"  {float(r.get('TotalDurationNs') or 0) / 1e6:9.2f} ms {int(r.get('Calls') or 0):6d} x {float(r.get('AverageNs') or 0) / 1e3:9.1f} us  {r['Name'][:110]}")
    wg = [r for r in rows if "conv3d_wgrad" in r["Name"]]
    if wg:
        ns, calls = sum(float(r["TotalDurationNs"]) for r in wg), sum(int(r["Calls"]) for r in wg)
        passes = float(steps)          # (--steps here: ALL passes of the profiled run, warm-up included)
        rate = flops * passes / (ns * 1e-9) / 1e12
        print(f"weight-gradient kernels (3x3x3): {ns / 1e6:.2f} ms over {calls} launches in {passes:.0f} backward passes -> {rate:.1f} TFLOP/s fp32-equivalent "
              f"= {rate / 157.3:.3f} of the fp32 MFMA peak (157.3); as issued (x 6 bf16 products with SS_WGRAD_ENGINE=bf16x6) {6 * rate:.0f} TFLOP/s = {6 * rate / 2500:.3f} of the 16-bit MFMA peak")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="1,2,4")
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--maxdisp", type=int, default=64)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--warmup-seconds", type=float, default=1.5, help="keep warming up at least this long per batch size (r06: the first size "
                    "timed in a fresh process came out at 27-34 ms instead of 15 now and then -- six steps are 0.1 s, less than the part takes "
                    "to leave its idle power state)")
    ap.add_argument("--no-fused-adam", action="store_true", help="torch.optim.Adam's default (foreach) implementation instead of fused=True")
    ap.add_argument("--no-checks", action="store_true", help="skip the determinism / eval-comparison passes (profiling runs)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "bench_train.json"))
    ap.add_argument("--kernel-stats", default=None)
    args = ap.parse_args()
    if args.kernel_stats:
        kernel_stats_report(args.kernel_stats, args.height, args.width, args.maxdisp, int(args.batches.split(",")[0]), args.steps)
        return

    import torch
    import torch.nn.functional as F
    import bench
    import semstereo_amd as sa
    assert torch.cuda.is_available(), "bench_train.py needs the MI355X (no CPU path exists)"
    sa._lib.load()
    dev = torch.device("cuda")
    H, W, md = args.height, args.width, args.maxdisp
    seg = sa.HotSegment(md).to(dev)
    bench.init_unit_gain(seg, 1234)
    L = conv_layers(H, W, md)
    flops_fwd = sum(2.0 * 27 * ci * co * v for _, kind, ci, co, v in L if kind != "head")
    res = {"workload": f"{H}x{W} maxdisp={md}, HotSegment.train(): forward + backward + Adam step; features [B,128,H/4,W/4] + [B,256,H/8,W/8] with gradients",
           "reference": "main_us3d.py:54,74,186-222 (maxdisp 64, batch 4, 1024^2 tiles)", "conv_engine": sa.engine.CONV_ENGINE,
           "gflop_3x3x3_forward_per_pair": flops_fwd / 1e9, "by_batch": {}}
    free0, total = torch.cuda.mem_get_info()
    for B in [int(b) for b in args.batches.split(",")]:
        rec = {}
        try:
            seg.train()
            # (the reference: optim.Adam(model.parameters(), lr=args.lr, betas=(0.9, 0.999)), main_us3d.py; `fused`: PyTorch's one-launch
            # implementation of the same update instead of ~6 element-wise launches per parameter tensor)
            try:
                opt = torch.optim.Adam(seg.parameters(), lr=1e-3, betas=(0.9, 0.999), fused=not args.no_fused_adam)
            except (TypeError, RuntimeError):
                opt = torch.optim.Adam(seg.parameters(), lr=1e-3, betas=(0.9, 0.999))
            fl8, fr8 = bench.synth_features(B, 256, H // 8, W // 8, 6, 5100, dev)
            fl4, fr4 = bench.synth_features(B, 128, H // 4, W // 4, 12, 5200, dev)
            feats = [t.requires_grad_(True) for t in (fl4, fr4, fl8, fr8)]
            g = torch.Generator(device=dev).manual_seed(77)
            gt = (torch.rand(B, H // 4, W // 4, generator=g, device=dev) * 2 - 1) * (md // 4 - 1)
            before = dict(sa.modules.PATH_COUNTS)

            def step(update=True):
                opt.zero_grad(set_to_none=True)
                for t in feats:
                    t.grad = None
                r = seg(*feats)
                loss = F.smooth_l1_loss(r["pred"].squeeze(1), gt) + F.smooth_l1_loss(r["pred_att"], gt)
                loss.backward()
                if update:
                    opt.step()
                return loss

            torch.cuda.reset_peak_memory_stats()
            n_warm, t_w = 0, time.perf_counter()
            while n_warm < args.warmup or (time.perf_counter() - t_w) < args.warmup_seconds:
                step()
                n_warm += 1
                if n_warm % 4 == 0:
                    torch.cuda.synchronize()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record()
            for _ in range(args.steps):
                loss = step()
            e1.record()
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
            ms = e0.elapsed_time(e1) / args.steps
            assert sa.modules.PATH_COUNTS["torch"] == before["torch"], "a PyTorch layer ran inside the 3-D stack"
            rec.update({"ms_per_step": ms, "pairs_per_s": 1e3 * B / ms, "host_wall_ms_per_step": 1e3 * wall / args.steps,
                        "peak_allocated_gb": torch.cuda.max_memory_allocated() / 2 ** 30, "peak_reserved_gb": torch.cuda.max_memory_reserved() / 2 ** 30,
                        "hip_train_calls_per_step": (sa.modules.PATH_COUNTS.get("hip_train", 0) - before.get("hip_train", 0)) / (n_warm + args.steps),
                        "loss": float(loss),
                        # forward + data gradient + weight gradient of every 3x3x3 layer: 3 x the forward's flops (fp32-equivalent)
                        "fp32_equivalent_tflops_3x3x3": 3.0 * flops_fwd * B / (ms * 1e-3) / 1e12})
            grads = {k: v.grad for k, v in seg.named_parameters() if v.grad is not None}
            rec["parameters_with_gradient"] = len(grads)
            rec["all_gradients_finite"] = bool(all(bool(torch.isfinite(g_).all()) for g_ in grads.values()) and
                                               all(t.grad is not None and bool(torch.isfinite(t.grad).all()) for t in feats))
            if not args.no_checks:
                # determinism to rounding: the same step twice without a weight update (atomics in the scatter / statistics kernels
                # reorder fp32 / float64 sums between runs)
                step(update=False)
                g1 = {k: v.grad.detach().clone() for k, v in seg.named_parameters() if v.grad is not None}
                step(update=False)
                worst = 0.0
                for k, v in seg.named_parameters():
                    if v.grad is not None and k not in ("gamma", "beta"):
                        worst = max(worst, float((v.grad - g1[k]).abs().max()) / (float(g1[k].abs().max()) + 1e-30))
                rec["rerun_max_relative_gradient_difference"] = worst
                # the inference step of the same segment at the same size, for scale
                seg.eval()
                with torch.no_grad():
                    for _ in range(2):
                        seg(*[t.detach() for t in feats])
                    torch.cuda.synchronize()
                    e0.record()
                    for _ in range(args.steps):
                        seg(*[t.detach() for t in feats])
                    e1.record()
                    torch.cuda.synchronize()
                rec["inference_ms_per_step_same_shape"] = e0.elapsed_time(e1) / args.steps
            del opt, feats, grads
        except torch.OutOfMemoryError as e:       # the largest batch that fits is part of the answer
            rec["error"] = "out of memory: " + str(e)[:160]
        torch.cuda.empty_cache()
        res["by_batch"][str(B)] = rec
        print(B, json.dumps(rec), flush=True)
    res["device_memory_gb"] = total / 2 ** 30
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps({"bench_train": res["by_batch"]}))


if __name__ == "__main__":
    main()
