#!/usr/bin/env python3
"""Benchmark of the SemStereo hot segment (cost volumes + 3-D aggregation + soft-argmax) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch PAIRS_PER_GPU]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
    python bench.py --gpus 8 --batch 4                      # BASELINE.json configs[3]: 32 pairs over 8 GPUs
    python bench.py --gpus 8 --height 2048 --width 2048 --maxdisp 192     # configs[4]: 8 pairs over 8 GPUs

A step = one pass of the hot path (1/4- and 1/8-scale features -> 1/4-scale disparity,
reference models/SemStereo.py:273-323) over this rank's batch of synthetic 1024x1024, maxdisp=128
pairs, inputs resident in HBM.  Pairs are sharded over ranks (weak scaling, no collective inside
the forward); the timed region is bracketed by barrier + synchronize and the MAX over ranks is
reported.  Rank 0 prints ONE JSON line with pairs/s, the roofline of the dominant kernel (the
matrix-core conv of concat_stem: two-term fp16 operands, fp32 accumulate) and of the cost-volume
build kernel (HBM), EPE against the CPU oracle, and the oracle's own pairs/s on the host cores.

Without a launcher (`WORLD_SIZE` unset) and `--gpus N > 1` this file launches its own N ranks: the
parent makes NO GPU/HIP call, starts N fresh child processes (RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_* set, rendezvous on 127.0.0.1), relays rank 0's JSON line and exits non-zero if any child
fails -- the counterpart of the reference's `nn.DataParallel(model)` (test_us3d.py:58), one process
per GPU instead of one thread per GPU.  `--dry-launch` rehearses exactly that control flow
(launcher, rendezvous, barriers, metric reduction, teardown) on CPU over gloo with a stand-in step.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import semstereo_amd  # noqa: E402
from semstereo_amd import dist as sdist  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
MFMA_BF16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA (spec)


def synth_features(B, C, H, W, max_shift, seed, device):
    """Left features ~ N(0,1); right = left shifted per row by a smooth signed integer disparity
    plus 5 % noise, so the volumes carry a real correlation peak."""
    g = torch.Generator(device=device).manual_seed(seed)
    left = torch.randn(B, C, H, W, generator=g, device=device)
    ys = torch.arange(H, device=device, dtype=torch.float32)
    shift = torch.round(max_shift * torch.sin(2 * torch.pi * ys / H + 0.3)).long()
    cols = (torch.arange(W, device=device).reshape(1, W) + shift.reshape(H, 1)) % W      # right[x] = left[x + d]
    right = torch.gather(left, 3, cols.reshape(1, 1, H, W).expand(B, C, H, W))
    right = right + 0.05 * torch.randn(B, C, H, W, generator=g, device=device)
    return left.contiguous(), right.contiguous()


def init_unit_gain(seg, seed):
    """Random init of the segment's weights at unit gain (U(-a,a), a = sqrt(3/fan_in), the same
    spirit as the reference's own SubModule.weight_init) with non-trivial BatchNorm statistics.
    PyTorch's default init shrinks activations layer by layer until the attention logits are ~1e-3
    and softmax over D is uniform to 5 digits: top-24 membership is then decided by fp32 rounding
    noise in ANY implementation (measured: 40 % of pixels have p24/p25 within 1e-5 relative), which
    says nothing about kernels.  Unit gain keeps logits O(1), like a trained network."""
    g = torch.Generator().manual_seed(seed)

    def uni(shape, lo, hi):
        return torch.rand(shape, generator=g) * (hi - lo) + lo
    with torch.no_grad():
        for name, t in list(seg.named_parameters()) + list(seg.named_buffers()):
            if name.endswith("num_batches_tracked"):
                continue
            if name == "gamma":
                v = torch.full(t.shape, 0.25)
            elif name == "beta":
                v = torch.full(t.shape, 2.0)
            elif name.endswith("running_var") or (name.endswith(".weight") and t.dim() == 1):
                v = uni(t.shape, 0.6, 1.4)
            elif t.dim() == 1:
                v = uni(t.shape, -0.1, 0.1)
            else:
                fan_in = t[0].numel()
                if ".conv5.0." in name or ".conv6.0." in name:       # ConvTranspose3d [Cin,Cout,3,3,3]
                    fan_in = t.shape[0] * 27 // 8
                a = (3.0 / fan_in) ** 0.5
                v = uni(t.shape, -a, a)
            t.copy_(v.to(t.device))


class KernelTimer:
    """HIP-event timing of selected launches on the stream they are launched on (torch's current
    stream, which is what semstereo_amd passes through the C ABI)."""

    def __init__(self, every=4):
        self.events = {}
        self.enabled = False
        self.every = every            # a pair of events around every `every`-th launch of a wrapped kernel inside the timed
        self.count = {}               # region: an event pair costs two extra packets and ~1 % of the step on every launch

    def wrap(self, name, fn):
        def timed(*a, **k):
            if not self.enabled:
                return fn(*a, **k)
            n = self.count[name] = self.count.get(name, 0) + 1
            if (n - 1) % self.every:
                return fn(*a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(*a, **k)
            e1.record()
            self.events.setdefault(name, []).append((e0, e1))
            return r
        return timed

    def mean_ms(self, name):
        ev = self.events.get(name, [])
        return sum(a.elapsed_time(b) for a, b in ev) / len(ev) if ev else None


def steady_ms(run, iters=20, warm_ms=200.0):
    """Mean launch time of `run` between two HIP events on the current stream, after `warm_ms` of back-to-back launches of the
    same kernel (the clock the chip holds under THIS kernel's load, not the one the previous micro-run left behind: the
    issue-bound kernels read up to 10 % slower right after a memory-bound one)."""
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    t0 = time.time()
    while (time.time() - t0) * 1e3 < warm_ms:
        for _ in range(10):
            run()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes of this file and relay
    rank 0's stdout (the JSON line).  The parent has made no GPU/HIP call (nothing before this point touches
    torch.cuda) and never exec-replaces itself; children are ended by their exact PIDs if one of them fails."""
    import socket
    import subprocess
    import threading
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: RCCL between processes needs it on this host driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0) or None))

    def end_ranks(signum, _frame):                               # the launcher is being ended: do not leave ranks behind on the GPUs
        for p in procs:
            if p.poll() is None:
                p.terminate()
        sys.exit(128 + signum)
    import signal
    signal.signal(signal.SIGTERM, end_ranks)
    signal.signal(signal.SIGINT, end_ranks)

    def relay():
        for line in procs[0].stdout:
            sys.stdout.write(line)
            sys.stdout.flush()
    t = threading.Thread(target=relay, daemon=True)
    t.start()
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        for r, p in enumerate(procs):
            if p.poll() not in (None, 0):
                failed = (r, p.returncode)
        time.sleep(0.05)
    if failed is None:
        failed = next(((r, p.returncode) for r, p in enumerate(procs) if p.returncode != 0), None)
    if failed is not None:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
        t.join(timeout=5)
        print(f"bench.py: rank {failed[0]} exited with code {failed[1]}; the other ranks were ended", file=sys.stderr)
        return 1
    t.join(timeout=30)
    return 0


def dry_step_factory(device):
    """Stand-in step of `--dry-launch`: a few small CPU matmuls (the launcher rehearsal measures nothing)."""
    a = torch.randn(64, 64, device=device)

    def step():
        return {"pred": (a @ a).sum().reshape(1, 1, 1, 1)}
    return step


def parity_vs_reference(sa, fixture_path, name, device):
    """The hot segment on the fixture's input against the REFERENCE's own outputs (tests/golden/segment_full.npz).
    `epe_vs_reference_px`: the plain run, every pixel, 1/4 scale (`..._fullres_px`: x4, the scale of the model's output
    disp = 4 * SSR_upsample(pred), models/SemStereo.py:346).  `reference_picks_restored`: the strict form of
    tests/strict.py (the reference's candidates put back where the top-24 pick differs at a margin below 1e-5)."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from golden import cases
    import strict
    g = np.load(fixture_path)
    B_, H_, W_, md = cases.segment_shape(name)
    seg = sa.HotSegment(md)
    seg.load_state_dict(cases.segment_params(name, g), strict=False)
    seg = seg.to(device).eval()
    fl4, fr4, fl8, fr8, _ = cases.segment_inputs(name)
    with torch.no_grad():
        r = seg(fl4.to(device), fr4.to(device), fl8.to(device), fr8.to(device))
    err = (r["pred"].cpu().squeeze(1) - torch.as_tensor(g[f"{name}/pred_map"])).abs()
    mine = cases.candidate_set_hash(r["samples"].cpu().numpy(), md // 4)
    out = {"fixture": f"tests/golden/segment_full.npz:{name} (reference outputs; closed-form input, calibrated BatchNorm statistics)",
           "epe_vs_reference_px": float(err.mean()), "epe_vs_reference_fullres_px": 4.0 * float(err.mean()),
           "median_abs_err_px": float(err.median()), "max_abs_err_px": float(err.max()),
           "pixels_beyond_1e-3": int((err > 1e-3).sum()), "pixels": int(err.numel()),
           "pixels_with_other_candidates": int((mine != g[f"{name}/candidate_hash"]).sum()),
           "pred_att_epe_vs_reference_px": float((r["pred_att"].cpu() - torch.as_tensor(g[f"{name}/pred_att_map"])).abs().mean())}
    if strict.fixture_view(g, name) is not None:
        rep = strict.run_strict(seg, g, name, device)[0]
        out["reference_picks_restored"] = rep
        out["note"] = ("the graph picks the 24 largest of 64 attention probabilities per pixel (models/SemStereo.py:299-303); where "
                       "the reference's own 24th / 25th are within 1e-5 relative (differing_pixels lists them with the margin and "
                       "with what the float64 evaluation of the graph picks there) an fp32 implementation may pick the other one, "
                       "and with calibrated BatchNorm the 3-D stack spreads ONE such pick over ~10^3 pixels of `pred`: the plain-run "
                       "EPE is the per-pixel error (reference_picks_restored) plus that")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1, help="pairs per GPU per step")
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--maxdisp", type=int, default=128)
    ap.add_argument("--engine", default=None, help="conv engine of the timed run: f32 | bf16x6 | bf16x3 | f16x3 "
                                                   "(default: semstereo_amd.modules.CONV_ENGINE)")
    ap.add_argument("--no-other-engines", action="store_true", help="skip the extra timings of the other engines")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-f64-truth", action="store_true", help="skip the float64 run of the oracle (about 4x the "
                    "fp32 oracle's time) that tells kernel error from the reference algorithm's own conditioning")
    ap.add_argument("--cpu-threads", type=int, default=32, help="cap on host threads for the oracle run")
    ap.add_argument("--no-kernel-timers", action="store_true")
    ap.add_argument("--graph", action="store_true", help="replay the step from a captured HIP graph (one launch per step instead "
                    "of ~45 from Python); the per-kernel HIP-event timers are off in this mode")
    ap.add_argument("--dry-launch", action="store_true", help="CPU rehearsal of the N-rank control flow over gloo "
                    "(launcher, rendezvous, barriers, metric reduction, teardown) with a stand-in step; measures nothing")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: become the parent of N ranks BEFORE anything touches the GPU (not even is_available())
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    # one rank per GPU over RCCL; SS_DIST_BACKEND=gloo lets the N > 1 control flow be rehearsed with several
    # ranks on ONE GPU (RCCL refuses two ranks on a device) -- the numbers of such a run mean nothing
    dry = args.dry_launch
    backend = "gloo" if dry else os.environ.get("SS_DIST_BACKEND", "nccl")
    if not dry:
        assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU path exists)"
        if backend != "nccl":
            os.environ["LOCAL_RANK"] = str(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
    rank, world, local = sdist.init_from_env(backend)       # (SS_DIST_FORCE_INIT=1: a group even at N = 1, tests/test_nccl_world1_gpu.py)
    grouped = dist.is_available() and dist.is_initialized()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if dry:
        device = torch.device("cpu")
        sync = lambda: None                                          # noqa: E731
        if os.environ.get("SS_DRY_FAIL_RANK") == str(rank):          # test hook: a rank that dies after the rendezvous
            sys.exit(7)
    else:
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)
        sync = torch.cuda.synchronize
        semstereo_amd._lib.load()
    M = semstereo_amd.modules
    if args.engine:
        M.CONV_ENGINE = args.engine
    engine = M.CONV_ENGINE

    H, W, maxdisp, B = args.height, args.width, args.maxdisp, args.batch
    if dry:
        seg = torch.nn.Linear(8, 8).to(device).eval()               # something to broadcast
        feats = ()
    else:
        seg = semstereo_amd.HotSegment(maxdisp).to(device).eval()
        init_unit_gain(seg, 1234)                    # same random-init weights on every rank
    sdist.broadcast_module(seg, src=0)
    if not dry:
        fl8, fr8 = synth_features(B, 256, H // 8, W // 8, 6, 100 + rank, device)
        fl4, fr4 = synth_features(B, 128, H // 4, W // 4, 12, 200 + rank, device)
        feats = (fl4, fr4, fl8, fr8)

    timer = KernelTimer()
    if not args.no_kernel_timers and not dry:
        if M.CONV_ENGINE == "f32":
            seg.concat_stem.forward = timer.wrap("concat_stem", seg.concat_stem.forward)
        else:       # the gated launch of the split-bf16 conv is concat_stem's (the right half of the volume)
            plain, timed = M.conv3d_bf16s_hip, timer.wrap("concat_stem", M.conv3d_bf16s_hip)
            M.conv3d_bf16s_hip = lambda *a, **k: (timed if (k.get("gate") is not None or (len(a) > 8 and a[8] is not None))
                                                  else plain)(*a, **k)
            M.stem_volume_half_presplit = timer.wrap("concat_stem_presplit", M.stem_volume_half_presplit)
        # the cost-volume kernel of the step: build_gwc_volume_norm fused with `patch` and the channelAtt gate
        # (models/SemStereo.py:273-276, ss_gwc_patch_gate_fwd); the volume kernel alone when that fusion is off
        semstereo_amd.segment.ops.build_gwc_volume_norm = timer.wrap("gwc", semstereo_amd.ops.build_gwc_volume_norm)
        semstereo_amd.segment.ops.gwc_patch_gate = timer.wrap("gwc_fused", semstereo_amd.ops.gwc_patch_gate)

    def step():
        with torch.no_grad():
            return seg(*feats)
    if dry:
        step = dry_step_factory(device)
    graphed = False
    if args.graph and not dry:
        # capture one step (both streams of the segment join the capture through their event waits) and replay it
        args.no_kernel_timers = True
        # the graph replays the kernels captured NOW: timing "other engines" or the unfused composition through it would
        # report this engine's rate under their labels -- those legs are skipped and marked so in the line
        args.no_other_engines = True
        gseg = semstereo_amd.GraphedSegment(seg, *feats)

        def step():                                                  # noqa: F811
            gseg.graph.replay()                                      # (inputs already resident in the captured buffers)
            return gseg.outputs
        graphed = True

    def timed_run(nsteps, nwarm, kernel_timers=False):
        """W untimed + exactly K timed steps, barrier + synchronize on both sides, MAX over ranks.
        The per-kernel HIP events are recorded only inside the timed region."""
        for _ in range(nwarm):
            o = step()
        sync()
        if grouped:
            dist.barrier()
        sync()
        timer.enabled = kernel_timers
        t0 = time.perf_counter()
        for _ in range(nsteps):
            o = step()
        sync()
        if grouped:
            dist.barrier()
        dt = time.perf_counter() - t0
        timer.enabled = False
        pairs, _, _, tmax = sdist.reduce_metrics(B * nsteps, 0.0, 0, dt, device)
        return o, pairs, tmax

    out, pairs, tmax = timed_run(args.steps, args.warmup, kernel_timers=True)
    assert M.PATH_COUNTS["torch"] == 0, "a PyTorch fallback ran inside the timed region"

    by_engine, outs = {engine: pairs / tmax}, {engine: out}
    opbyop_rate, fired = None, None
    if dry:
        unfused_rate = None
    elif not args.no_other_engines:
        for e in ("f32", "bf16x6", "bf16x3", "f16x3"):
            if e != engine:
                M.CONV_ENGINE = e
                o, p_, t_ = timed_run(max(3, args.steps // 2), 2)
                by_engine[e], outs[e] = p_ / t_, o
        M.CONV_ENGINE = engine
        # what install() + accelerate() give a reference model whose forward() is left untouched: the reference's
        # statements one by one, in its order (HotSegment's FUSED = False composition), on the reference-named ops and the
        # twins -- which in inference hand out deferred handles, so that the same fused kernels run (semstereo_amd/deferred.py);
        # and the same with deferral off: every op and module its own launch, PyTorch glue in between
        from semstereo_amd import deferred as dfr
        seg.FUSED = False
        dfr.STATS["fused"].clear()
        _, p_, t_ = timed_run(max(3, args.steps // 2), 2)
        unfused_rate = p_ / t_
        fired = dict(dfr.STATS["fused"])
        dfr.ENABLED = False
        _, p_, t_ = timed_run(max(3, args.steps // 2), 2)
        opbyop_rate = p_ / t_
        dfr.ENABLED = True
        seg.FUSED = True
    else:
        unfused_rate = None

    if rank != 0:
        # rank 0 still runs the CPU baseline; meet it at a last barrier so the group is torn down together
        dist.barrier()
        dist.destroy_process_group()
        return
    if dry:
        print(json.dumps({"metric": "dry launch (CPU rehearsal of the N-rank control flow; measures nothing)", "dry_launch": True,
                          "value": pairs / tmax, "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": 1e3 * tmax / args.steps, "pairs_counted": pairs, "backend": backend,
                          "data": "synthetic"}), flush=True)
        if grouped:
            dist.barrier()
            dist.destroy_process_group()
        return
    D8, k = 2 * (maxdisp // 8), 24
    H8, W8, H4, W4 = H // 8, W // 8, H // 4, W // 4
    cfg_name = {(1024, 1024, 128, 1, 1): "configs[1]", (1024, 1024, 128, 8, 1): "configs[2]", (1024, 1024, 128, 4, 8): "configs[3]",
                (2048, 2048, 192, 1, 8): "configs[4]"}.get((H, W, maxdisp, B, world), "shape of configs[1] at another batch / rank count"
                                                           if (H, W, maxdisp) == (1024, 1024, 128) else "custom shape")
    engine_note = {
        "f32": "exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) for every 3-D layer",
        "bf16x6": "3x3x3 stride-1 convs: fp32 operands split into 3 bf16 terms, 6 cross products on v_mfma_f32_32x32x16_bf16, "
                  "fp32 accumulate (measured error vs fp64 below the exact-fp32 MFMA's); other layers exact-fp32 MFMA",
        "bf16x3": "as bf16x6 with 3 cross products (hi*hi + hi*mid + mid*hi)",
        "f16x3": "3x3x3 convs (stride 1, 2) and transposed convs: fp32 operands as TWO fp16 terms with block-floating power-of-two "
                 "scales (per output channel for weights, per staged tile chunk for activations), 3 cross products on "
                 "v_mfma_f32_32x32x16_f16, fp32 accumulate (measured error vs fp64 at or below the exact-fp32 MFMA's, "
                 "tools/check_engines.py); heads, 1x1x1 projections, broadcast half of the stem: bf16x6",
    }[engine]
    res = {
        "metric": "stereo pairs/sec, hot segment (gwc+concat volumes, 3-D hourglass stack, soft-argmax), "
                  f"{H}x{W} maxdisp={maxdisp}",
        "value": pairs / tmax, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * tmax / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {"f32": "f32", "f16x3": "f32 (2xfp16 split operands, fp32 accumulate)",
                  "bf16x6": "f32 (3xbf16 split operands, 6 products, fp32 accumulate)",
                  "bf16x3": "f32 (3xbf16 split operands, 3 products, fp32 accumulate)"}[engine], "data": "synthetic",
        "config": {"workload": f"BASELINE.json {cfg_name}: {H}x{W} tile, maxdisp={maxdisp}, batch={B} per GPU x {world} GPU(s), "
                               "features [B,128,H/4,W/4]+[B,256,H/8,W/8] -> disparity [B,1,H/4,W/4]",
                   "pairs_per_gpu_per_step": B, "parallelism": f"pairs sharded over {world} rank(s), no collective in the forward",
                   "weights": "random init at unit gain (see init_unit_gain), BatchNorm eval",
                   "conv_engine": engine, "conv_engine_note": engine_note},
        "hip_graph": graphed, "dist_backend_initialised": dist.get_backend() if grouped else None,
        "pairs_per_s_by_conv_engine": by_engine if not graphed else {engine: by_engine[engine], "others": "skipped under --graph"},
        "pairs_per_s_reference_forward_untouched": unfused_rate if not graphed else "skipped under --graph",
        "pairs_per_s_reference_forward_untouched_no_deferral": opbyop_rate,
        "deferred_rules_fired_per_run": fired,
    }
    ms, presplit = timer.mean_ms("concat_stem"), False
    if not ms:
        ms, presplit = timer.mean_ms("concat_stem_presplit"), True
    if ms:
        halves = engine != "f32" and semstereo_amd.HotSegment.STEM_BY_HALVES
        cin_stem = 32 if halves else 64     # by linearity only the warped right half of the volume is convolved
        flops = 2.0 * 32 * cin_stem * 27 * k * H4 * W4 * B     # concat_stem: Conv3d k3 on [B,cin_stem,24,H4,W4] -> 32 channels
        eq = flops / (ms * 1e-3) / 1e12                        # fp32-equivalent rate of the flops this launch performs
        if engine == "f32":
            res["roofline"] = {"kernel": "conv3d_mfma<3,1,1,4,2,8,4> (concat_stem, 64->32 k3 on [B,64,24,H/4,W/4])",
                               "bound": "mfma", "achieved": eq, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": eq / MFMA_F32_PEAK_TFLOPS, "traffic": None, "launch_ms": ms,
                               "algorithmic_flop_per_launch": flops}
        else:
            nterms = 6 if engine == "bf16x6" else 3
            code = 19 if engine == "f16x3" else nterms         # the kernel's NTERMS template argument
            typ = "fp16" if engine == "f16x3" else "bf16"
            ex = nterms * eq                                   # 16-bit MFMA flops actually issued per second
            # the symbol rocprofv3 shows: 4 x 4 x 32 tiles where the depth is a multiple of 4 (conv3d_bf16s.hip's tile choice)
            sym = ("conv3d_pre<true>" if presplit else
                   f"conv3d_bf16s<1, 4, 4, 4, {code}, true, 1, 3>" if k % 4 == 0 else f"conv3d_bf16s<1, 4, 2, 8, {code}, true, 1, 3>")
            res["roofline"] = {"kernel": f"{sym} (concat_stem: {cin_stem}->32 k3 on [B,{cin_stem},24,H/4,W/4]"
                                         + (", the warped half of the volume; + residual (the broadcast half, by linearity) + ReLU + channelAtt gate)" if halves
                                            else " + ReLU + channelAtt gate)"),
                               "bound": "mfma", "achieved": ex, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": ex / MFMA_BF16_PEAK_TFLOPS, "traffic": None, "launch_ms": ms,
                               "algorithmic_flop_per_launch": nterms * flops,
                               "note": f"{nterms} {typ} products per fp32 product (fp16 and bf16 MFMA peaks are equal); fp32-equivalent rate {eq:.1f} TFLOP/s = "
                                       f"{eq / MFMA_F32_PEAK_TFLOPS:.2f} x the {MFMA_F32_PEAK_TFLOPS} TFLOP/s fp32-MFMA peak",
                               "fp32_equivalent_tflops": eq}
    ms, fused_gwc = timer.mean_ms("gwc"), False
    if not ms:
        ms, fused_gwc = timer.mean_ms("gwc_fused"), True
    if ms:
        # algorithmic bytes of SURVEY.md section 8(d): both feature maps in, the [B,32,D8,H8,W8] volume out (the fused
        # kernel also reads the [B,32,H8,W8] gate logits and writes the volume AFTER `patch` and the gate: same size)
        nbytes = 4.0 * (2 * 256 * H8 * W8 + 32 * D8 * H8 * W8) * B
        ach = nbytes / (ms * 1e-3) / 1e9
        res["roofline_cost_volume"] = {"kernel": ("gwc_patch_gate_v4<8,true> (build_gwc_volume_norm + patch + channelAtt gate, "
                                                  "models/SemStereo.py:273-276, live shape)" if fused_gwc
                                                  else "gwc_volume_v4<8,true> (build_gwc_volume_norm, live shape)"),
                                       "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": ach / HBM_PEAK_GBS, "traffic": None, "launch_ms": ms,
                                       "algorithmic_bytes_per_launch": nbytes,
                                       "note": "inside the timed region at this batch (a partly filled chip, the 2-D convolutions of "
                                               "the matching branch running beside it on the second stream); BASELINE.json "
                                               "configs[2] (batch 8) is below"}
        # the cost-volume kernel at BASELINE.json configs[2] (batch 8, the HBM-roofline configuration),
        # 20 back-to-back launches between two HIP events on the launch stream
        g8 = torch.Generator(device=device).manual_seed(7)
        a8 = torch.randn(8, 256, H8, W8, generator=g8, device=device)
        b8 = torch.randn(8, 256, H8, W8, generator=g8, device=device)
        ms8 = steady_ms(lambda: semstereo_amd.ops.build_gwc_volume_norm(a8, b8, maxdisp // 8, 32))
        nb8 = 8 * nbytes / B
        res["roofline_cost_volume_b8"] = {"kernel": "gwc_volume_v4<8,true,stream>, batch 8 (BASELINE.json configs[2])", "bound": "hbm",
                                          "achieved": nb8 / (ms8 * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                          "frac": nb8 / (ms8 * 1e-3) / 1e9 / HBM_PEAK_GBS, "launch_ms": ms8,
                                          "algorithmic_bytes_per_launch": nb8,
                                          "traffic": 806.6e6,
                                          "traffic_note": "HBM bytes per launch from PMC passes of tools/pmc_bytes.sh gwc 8 (profiles/"
                                                          "r02_e_pmc_gwc_b8.md; r01: 806.1e6): 2 x FETCH_SIZE (268.6 MB) + WRITE_SIZE "
                                                          "(538.0 MB), gfx950 correction applied; algorithmic 805.3e6"}
        del a8, b8
        # calibration SURVEY.md section 8(d) asks for: what a plain device-to-device copy reaches on this box
        # (read + write bytes over time, 1 GiB, 10 back-to-back copies), to read the fractions against
        src = torch.empty(256 << 20, dtype=torch.float32, device=device)
        dst = torch.empty_like(src)
        for _ in range(3):
            dst.copy_(src)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            dst.copy_(src)
        e1.record()
        torch.cuda.synchronize()
        copy_gbs = 2.0 * src.numel() * 4 * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        res["hbm_copy_measured_gbs"] = copy_gbs
        res["roofline_cost_volume_b8"]["frac_of_measured_copy"] = res["roofline_cost_volume_b8"]["achieved"] / copy_gbs
        del src, dst
        # the fused form of the step (volume -> patch -> gate in one launch) at the same batch 8
        g8 = torch.Generator(device=device).manual_seed(7)
        a8 = torch.randn(8, 256, H8, W8, generator=g8, device=device)
        b8 = torch.randn(8, 256, H8, W8, generator=g8, device=device)
        gl8 = torch.randn(8, 32, H8, W8, generator=g8, device=device)
        if semstereo_amd.ops.gwc_patch_gate_applies(a8, maxdisp // 8, 32):
            run = lambda: semstereo_amd.ops.gwc_patch_gate(a8, b8, maxdisp // 8, 32, seg.patch.weight, gl8)     # noqa: E731
            msf = steady_ms(run)
            res["roofline_cost_volume_fused_b8"] = {
                "kernel": "gwc_patch_gate_v4<8,true,stream>, batch 8: volume + patch + gate in one launch", "bound": "hbm",
                "achieved": nb8 / (msf * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": nb8 / (msf * 1e-3) / 1e9 / HBM_PEAK_GBS, "launch_ms": msf, "algorithmic_bytes_per_launch": nb8,
                "traffic": 910.6e6,
                "traffic_note": "HBM bytes per launch from PMC passes of tools/pmc_bytes.sh gwc_fused 8 (profiles/r02_e_pmc_gwc_fused_b8.md): "
                                "2 x FETCH_SIZE (373.7 MB: the 8-rows-for-6 halo re-reads) + WRITE_SIZE (536.9 MB)"}
        del a8, b8, gl8
        # ... and alone at the bench batch: inside the step it shares the chip with the matching branch's 2-D convolutions
        # on the second stream (roofline_cost_volume above is that concurrent figure)
        gB = torch.Generator(device=device).manual_seed(8)
        aB, bB = torch.randn(B, 256, H8, W8, generator=gB, device=device), torch.randn(B, 256, H8, W8, generator=gB, device=device)
        glB = torch.randn(B, 32, H8, W8, generator=gB, device=device)
        if fused_gwc and semstereo_amd.ops.gwc_patch_gate_applies(aB, maxdisp // 8, 32):
            run = lambda: semstereo_amd.ops.gwc_patch_gate(aB, bB, maxdisp // 8, 32, seg.patch.weight, glB)     # noqa: E731
            msa = steady_ms(run)
            res["roofline_cost_volume_alone"] = {
                "kernel": res["roofline_cost_volume"]["kernel"] + ", launched alone", "bound": "hbm",
                "achieved": nbytes / (msa * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": nbytes / (msa * 1e-3) / 1e9 / HBM_PEAK_GBS, "launch_ms": msa, "algorithmic_bytes_per_launch": nbytes,
                "traffic": None,
                "note": "the two kernels it replaces (volume, then patch + gate) move 2.35x these bytes: 100.7 + 136.3 MB per pair"}
        del aB, bB, glB
    # the semantic-guided refinement head that follows the segment in the model (SSR_upsample, models/submodule.py:412-431;
    # BASELINE.json configs[4] names it): one launch per pair at full resolution, outside the timed segment
    try:
        ssr = M.SSR_upsample(6).to(device).eval()
        gs = torch.Generator(device=device).manual_seed(9)
        d_low = torch.randn(B, 1, H4, W4, generator=gs, device=device) * 8
        wts, lab = torch.randn(B, 6, H, W, generator=gs, device=device), torch.randn(B, 6, H, W, generator=gs, device=device)
        prm = ssr._params()
        out_ssr = torch.empty(B, H, W, device=device)
        lib = semstereo_amd._lib
        run = lambda: lib.call("ss_ssr_upsample_fwd", lib.ptr(d_low), lib.ptr(wts), lib.ptr(lab), lib.ptr(prm), lib.ptr(out_ssr), B, H4, W4, 6)   # noqa: E731
        mss = steady_ms(run)
        nbs = 4.0 * B * (13 * H * W + H4 * W4)
        res["roofline_ssr_upsample"] = {"kernel": "ssr_upsample_tiled (SSR_upsample: 4x bilinear + 6-class gated residual, one launch)",
                                        "bound": "hbm", "achieved": nbs / (mss * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "frac": nbs / (mss * 1e-3) / 1e9 / HBM_PEAK_GBS, "launch_ms": mss,
                                        "algorithmic_bytes_per_launch": nbs, "traffic": None}
        del d_low, wts, lab, out_ssr
    except Exception as e:       # noqa: BLE001  (never let the side measurement take the line down)
        res["roofline_ssr_upsample"] = {"error": repr(e)}
    # the 32 -> 1 head of `classif` (models/SemStereo.py:228-234) reading its classifier's channels-last intermediate: the
    # bandwidth kernel VERDICT r1 #6 named (268 MB in 83 us then), alone at the bench batch
    try:
        xcl = torch.relu(torch.randn(B, 24, H4, W4, 32, generator=torch.Generator(device=device).manual_seed(10), device=device))
        wsh = M.pack_head_weight_bf16s(torch.randn(1, 32, 3, 3, 3, device=device) * 0.03)
        outh = torch.empty(B, 1, xcl.shape[1], H4, W4, device=device)
        lib = semstereo_amd._lib
        runh = lambda: lib.call("ss_conv3d_head_bf16s_cl_fwd", lib.ptr(xcl), lib.ptr(wsh), None, None, lib.ptr(outh), B, 32, xcl.shape[1], H4, W4, 0, 6)   # noqa: E731
        msh = steady_ms(runh)
        nbh = 4.0 * B * 33 * xcl.shape[1] * H4 * W4
        res["roofline_classifier_head"] = {
            "kernel": "conv3d_head_bf16s<4,8,2,6,CL> (classif.2: Conv3d(32,1,3) over [B,32,24,H/4,W/4], channels-last input), launched alone",
            "bound": "hbm", "achieved": nbh / (msh * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": nbh / (msh * 1e-3) / 1e9 / HBM_PEAK_GBS, "launch_ms": msh, "algorithmic_bytes_per_launch": nbh,
            "traffic": 291.6e6 * B if (H, W, maxdisp) == (1024, 1024, 128) else None,
            "traffic_note": "HBM bytes per pair from PMC passes of tools/pmc_bytes.sh head_cl 1 (profiles/r02_e_pmc_head_cl_b1.md): "
                            "2 x FETCH_SIZE (284.1 MB: the 6x10-rows-for-4x8 halo re-reads that miss L2) + WRITE_SIZE (7.5 MB)"}
        del xcl, outh
    except Exception as e:       # noqa: BLE001
        res["roofline_classifier_head"] = {"error": repr(e)}
    if not args.no_cpu_baseline and world == 1:        # CPU baseline and EPE: rank 0 at N = 1 only
        # The oracle (this repo's CPU restatement of the reference algorithm) on ONE pair of the
        # same workload: about 10-30 s of CPU work.  ATen's CPU kernels stop scaling (the slice
        # loop of build_gwc_volume_norm anti-scales) beyond a few dozen threads, so cap them.
        from oracle import hot_segment as oseg
        from oracle import ops as oops
        P = {k_: v.detach().cpu() for k_, v in seg.state_dict().items()}
        cpu_in = [t[:1].cpu() for t in feats]
        nthreads = min(os.cpu_count() or 1, args.cpu_threads)
        torch.set_num_threads(nthreads)
        c0 = time.perf_counter()
        ref = oseg.hot_segment(P, cpu_in[0], cpu_in[1], cpu_in[2], cpu_in[3], maxdisp)
        cdt = time.perf_counter() - c0
        res["cpu_baseline"] = {"value": 1.0 / cdt, "unit": "pairs/s", "cores": nthreads, "kind": "port",
                               "sample": f"1 pair {H}x{W} maxdisp={maxdisp} through oracle.hot_segment "
                                         f"(PyTorch CPU fp32 restatement of the reference), {cdt:.1f} s, "
                                         f"{nthreads} of {os.cpu_count()} host threads"}

        def parity(o):
            pred, rpred = o["pred"][:1].cpu(), ref["pred"]
            err = (pred - rpred).abs()
            same_px = (o["samples"][:1].cpu() == ref["samples"]).all(dim=1, keepdim=True)
            return {"epe_px": oops.epe(pred, rpred), "epe_fullres_px": 4.0 * oops.epe(pred, rpred),
                    "pred_att_epe_px": oops.epe(o["pred_att"][:1].cpu(), ref["pred_att"]),
                    "median_abs_err_px": err.median().item(), "max_abs_err_px": err.max().item(),
                    "pixels_abs_err_gt_1e-3": (err > 1e-3).float().mean().item(),
                    "pixels_with_identical_top24_candidates": same_px.float().mean().item()}
        par = parity(out)
        if not args.no_f64_truth:
            # The same oracle in float64 = the mathematically exact answer of the reference graph for
            # these weights.  regression_topk's hard top-2 pick (models/submodule.py:436-437) makes
            # `pred` discontinuous in the costs, so any two fp32 implementations differ by whole
            # candidates wherever the 2nd/3rd largest cost are closer than their rounding error:
            # measure both fp32 paths against the truth, and the EPE where the truth's gap is not tiny.
            P64 = {k_: (v.double() if v.is_floating_point() else v) for k_, v in P.items()}
            c1 = time.perf_counter()
            cpu64 = [t.double() for t in cpu_in]
            tru = oseg.hot_segment(P64, *cpu64, maxdisp, keep=True)
            e_hip = (out["pred"][:1].cpu().double() - tru["pred"]).abs()
            e_o32 = (ref["pred"].double() - tru["pred"]).abs()
            # matching branch alone, every path fed the truth's candidates (no top-24 differences, whose
            # effect spreads over the 3-D stack's receptive field): cost error and EPE of each fp32 path
            att32, smp = tru["att_topk"].float(), tru["samples"].float()
            keep32, cap = {}, {}
            p32 = oseg.matching_branch(P, cpu_in[0], cpu_in[1], att32, smp, keep32)
            hk = seg.classif.register_forward_hook(lambda m_, a_, o_: cap.__setitem__("cost", o_.detach()))
            with torch.no_grad():
                ph = seg.matching_branch(feats[0][:1], feats[1][:1], att32.to(device), smp.to(device))
            hk.remove()
            cost64 = tru["cost"].squeeze(1)
            top3 = cost64.topk(3, dim=1).values
            gap = (top3[:, 1] - top3[:, 2]).unsqueeze(1)
            ok = gap > 1e-4
            g_hip = (ph.cpu().double() - tru["pred"]).abs()
            g_o32 = (p32.double() - tru["pred"]).abs()

            def rms(x):
                return x.double().pow(2).mean().sqrt().item()
            par["vs_float64_truth"] = {
                "whole_path_hip_epe_px": e_hip.mean().item(), "whole_path_oracle_fp32_epe_px": e_o32.mean().item(),
                "given_truth_candidates": {
                    "hip_cost_rms_err": rms(cap["cost"].cpu().squeeze(1).double() - cost64),
                    "oracle_fp32_cost_rms_err": rms(keep32["cost"].squeeze(1).double() - cost64),
                    "hip_epe_px": g_hip.mean().item(), "oracle_fp32_epe_px": g_o32.mean().item(),
                    "hip_epe_px_where_truth_top2_gap_gt_1e-4": g_hip[ok].mean().item(),
                    "hip_max_err_px_there": g_hip[ok].max().item(),
                    "oracle_fp32_epe_px_there": g_o32[ok].mean().item(),
                    "oracle_fp32_max_err_px_there": g_o32[ok].max().item(),
                    "fraction_of_pixels_there": ok.double().mean().item()},
                "truth_cost_std_over_candidates": cost64.std(dim=1).mean().item(),
                "truth_top2_gap_median": gap.median().item(),
                "fraction_of_pixels_with_gap_lt_1e-5": (gap < 1e-5).double().mean().item(),
                "seconds": time.perf_counter() - c1}
        res["epe_vs_oracle_px"] = par["epe_px"]
        # EPE against the REFERENCE itself (mean |disp - disp_ref| over all pixels): the committed
        # full-size record tests/golden/segment_full.npz holds the reference's own `pred` map, candidate hashes and margins
        # for its closed-form 1024 x 1024 / maxdisp 128 input with calibrated BatchNorm statistics (made by
        # tests/golden/make_golden.py from /root/reference in the build container; nothing of the reference is read here)
        fx = os.path.join(ROOT, "tests", "golden", "segment_full.npz")
        if (H, W, maxdisp) == (1024, 1024, 128) and os.path.exists(fx):
            try:
                res["parity_vs_reference"] = parity_vs_reference(semstereo_amd, fx, "f1024_md128_cal", device)
                pv = res["parity_vs_reference"]
                # top level: the plain run (SURVEY.md section 8d's definition, every pixel) and, beside it, the same run with
                # the reference's own candidates put back at the pixels whose top-24 pick differs (reference margin < 1e-5)
                for k_ in ("epe_vs_reference_px", "epe_vs_reference_fullres_px", "pixels_with_other_candidates"):
                    res[k_] = pv[k_]
                if "reference_picks_restored" in pv:
                    res["epe_vs_reference_fullres_px_reference_picks_restored"] = pv["reference_picks_restored"]["epe_vs_reference_fullres_px"]
            except Exception as e:       # noqa: BLE001
                res["parity_vs_reference"] = {"error": repr(e)}
        res["parity_vs_oracle"] = par
        res["parity_vs_oracle_by_conv_engine"] = {e: parity(o) for e, o in outs.items() if e != engine}
    print(json.dumps(res), flush=True)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
