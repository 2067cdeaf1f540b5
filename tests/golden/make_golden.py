#!/usr/bin/env python3
"""Generate the golden fixtures by running the REFERENCE itself.

Runs ONLY in the build container, where /root/reference is mounted read-only:
it imports the reference's Python by path (nothing is copied), feeds it the
closed-form inputs of `cases.py`, and stores the reference's OUTPUTS as small
.npz files next to this script.  The GPU box has no /root/reference; tests
there only read the committed .npz files.

    python tests/golden/make_golden.py            # writes tests/golden/*.npz
"""
import importlib.util
import os
import sys
import types
import warnings

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from golden import cases  # noqa: E402
from oracle import hot_segment as oseg  # noqa: E402  (parameter table + deterministic fill only)

REF = "/root/reference"
warnings.filterwarnings("ignore")
torch.set_num_threads(8)


def load_ref_oplib():
    spec = importlib.util.spec_from_file_location("ref_submodule", os.path.join(REF, "models/submodule.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_ref_model_module():
    """Import models.SemStereo without models/__init__.py (needs torchvision)
    and with a stand-in `timm` whose backbone has the attribute surface
    Feature() reads (models/SemStereo.py:37-45).  The backbone is never run:
    the generator replays closed-form feature maps instead."""
    class _Backbone(nn.Module):
        def __init__(self):
            super().__init__()
            mk = lambda i, o, s: nn.Sequential(nn.Conv2d(i, o, 3, s, 1, bias=False))
            self.stem = mk(3, 32, 2)
            self.stages_0 = nn.Sequential(mk(32, 64, 1)); self.stages_1 = nn.Sequential(mk(64, 128, 2))
            self.stages_2 = nn.Sequential(mk(128, 256, 2)); self.stages_3 = nn.Sequential(mk(256, 384, 2))
            self.stages_4 = nn.Sequential(mk(384, 512, 2))
    timm = types.ModuleType("timm")
    timm.create_model = lambda *a, **k: _Backbone()
    sys.modules["timm"] = timm
    pkg = types.ModuleType("models")
    pkg.__path__ = [os.path.join(REF, "models")]
    sys.modules["models"] = pkg
    sys.path.insert(0, REF)
    import models.SemStereo as ms
    return ms


class Replay(nn.Module):
    """Stands in for an out-of-scope module: returns queued tensors in call order."""
    def __init__(self, *items):
        super().__init__()
        self.items = list(items)

    def forward(self, *a, **k):
        return self.items.pop(0)


def f32(t):
    return t.detach().cpu().numpy().astype(np.float32)


def summary(t, salt):
    """checksum record for a big tensor: sum, sum of squares, 64 sampled voxels."""
    a = t.detach().double().reshape(-1)
    idx = cases.sample_index(a.numel(), 64, salt)
    return np.concatenate([[a.sum().item(), (a * a).sum().item()], a[idx].numpy()]).astype(np.float64)


def gen_ops(ref):
    out = {}
    for n in cases.GWC:
        a, b, m, G = cases.gwc_inputs(n)
        out[f"gwc/{n}"] = f32(ref.build_gwc_volume(a, b, m, G))
        out[f"gwc_norm/{n}"] = f32(ref.build_gwc_volume_norm(a, b, m, G))
        out[f"gcorr/{n}"] = f32(ref.groupwise_correlation(a, b, G))
        out[f"gcorr_norm/{n}"] = f32(ref.groupwise_correlation_norm(a, b, G))
    for n in cases.CONCAT:
        a, b, m = cases.concat_inputs(n)
        out[f"concat/{n}"] = f32(ref.build_concat_volume(a, b, m))
    for n in cases.REGRESSION:
        p, m, d = cases.regression_inputs(n)
        out[f"regression/{n}"] = f32(ref.disparity_regression(p, m))
        out[f"variance/{n}"] = f32(ref.disparity_variance(p, m, d))
    for n in cases.WARP:
        x, y, d = cases.warp_inputs(n)
        yw, xw = ref.SpatialTransformer_grid(x, y, d)
        out[f"warp_y/{n}"] = f32(yw)
        out[f"warp_x/{n}"] = f32(xw)
    for n in cases.TOPK:
        c, s, k = cases.topk_inputs(n)
        out[f"topk/{n}"] = f32(ref.regression_topk(c, s, k))
    for n in cases.PROP:
        out[f"prop/{n}"] = f32(ref.Propagation()(cases.prop_inputs(n)))
    for n in cases.PROP_PROB:
        out[f"prop_prob/{n}"] = f32(ref.Propagation_prob()(cases.prop_prob_inputs(n)))
    np.savez_compressed(os.path.join(HERE, "ops.npz"), **out)
    print("ops.npz:", len(out), "arrays")


def load_ref_unsigned_oplib():
    spec = importlib.util.spec_from_file_location("ref_submodule_unsigned", os.path.join(REF, "models/submodule_.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def gen_ops_unsigned(uref):
    """The unsigned-range op set, models/submodule_.py (what models/SemStereo_WHU.py is written for)."""
    out = {}
    for n in cases.UGWC:
        a, b, m, G = cases.ugwc_inputs(n)
        if m <= a.shape[3]:                       # beyond the width the reference's own slice assignment raises
            out[f"ugwc/{n}"] = f32(uref.build_gwc_volume(a, b, m, G))
            out[f"ugwc_norm/{n}"] = f32(uref.build_gwc_volume_norm(a, b, m, G))
    for n in cases.UCONCAT:
        a, b, m = cases.uconcat_inputs(n)
        if m <= a.shape[3]:
            out[f"uconcat/{n}"] = f32(uref.build_concat_volume(a, b, m))
    for n in cases.UREGRESSION:
        p, m, d = cases.uregression_inputs(n)
        out[f"uregression/{n}"] = f32(uref.disparity_regression(p, m))
        out[f"uvariance/{n}"] = f32(uref.disparity_variance(p, m, d))
    np.savez_compressed(os.path.join(HERE, "ops_unsigned.npz"), **out)
    print("ops_unsigned.npz:", len(out), "arrays")


def load_ref_whu_module(uref):
    """models.SemStereo_WHU with the unsigned op set bound in ITS globals (install()'s own mechanism): as shipped it
    star-imports the signed models/submodule.py and fails at disparity_regression."""
    import models.SemStereo_WHU as mw
    for name in ("build_gwc_volume", "build_gwc_volume_norm", "build_concat_volume", "disparity_regression", "disparity_variance"):
        setattr(mw, name, getattr(uref, name))
    return mw


def gen_segment_whu(ms, uref):
    mw = load_ref_whu_module(uref)
    out = {}
    for n, (B, H, W, maxdisp) in cases.SEGMENT_WHU.items():
        net = mw.SemStereo_WHU(maxdisp, False, True, True, 6).eval()
        P = oseg.deterministic_params()
        res = net.load_state_dict(P, strict=False)
        assert not res.unexpected_keys
        fl4, fr4, fl8, fr8, _ = cases.segment_inputs(n)
        cap = run_reference_segment(mw, net, B, H, W, (fl4, fr4, fl8, fr8))
        (cost_sq, samples, k), pred = cap["regression_topk"][0]
        out[f"{n}/pred"] = f32(pred)
        out[f"{n}/samples"] = samples.numpy().astype(np.int16)
        out[f"{n}/pred_att0"] = f32(cap["disparity_regression"][0][1])
        out[f"{n}/pred_att"] = f32(cap["ssr_in"][0].squeeze(1))
        out[f"{n}/cost_att"] = f32(cap["classif_att_"])
        p = cap["aw_prob"].squeeze(1).sort(dim=1, descending=True).values
        c = cost_sq.sort(dim=1, descending=True).values
        out[f"{n}/gap24_rel"], out[f"{n}/gap2"] = f32((p[:, 23] - p[:, 24]) / p[:, 23]), f32(c[:, 1] - c[:, 2])
        out[f"{n}/att_topk"] = f32(torch.gather(cap["aw_prob"], 2, samples.long().unsqueeze(1)).squeeze(1))
        print(n, "pred", tuple(pred.shape), "range", float(pred.min()), float(pred.max()), "samples", int(samples.min()), int(samples.max()))
    np.savez_compressed(os.path.join(HERE, "segment_whu.npz"), **out)
    print("segment_whu.npz:", len(out), "arrays")


def build_ref_net(ms, maxdisp):
    net = ms.SemStereo(maxdisp, False, True, True, 6).eval()
    P = oseg.deterministic_params()
    res = net.load_state_dict(P, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys          # every oracle key name exists in the reference
    ours = set(P)
    for k in res.missing_keys:                                   # and we cover every key of the owned modules
        owned = k.split(".")[0] in {"patch", "corr_feature_att_8", "hourglass_att", "classif_att_", "gamma", "beta",
                                    "concat_feature", "concat_stem", "concat_feature_att_4", "hourglass", "classif"}
        assert (not owned) or k.endswith("num_batches_tracked"), k
    return net, P


def gen_stack(ms):
    net, P = build_ref_net(ms, 64)
    out = {}
    with torch.no_grad():
        for n, (kind, shape, _) in cases.STACK.items():
            mod = net
            for part in kind.split("."):
                mod = getattr(mod, part)
            y = mod(cases.stack_input(n))
            out[f"stack/{n}"] = f32(y)
    np.savez_compressed(os.path.join(HERE, "stack.npz"), **out)
    print("stack.npz:", {k: v.shape for k, v in out.items()})


OWNED = ("patch", "corr_feature_att_8", "hourglass_att", "classif_att_", "gamma", "beta", "concat_feature",
         "concat_stem", "concat_feature_att_4", "hourglass", "classif")
STAGES = ("build_gwc_volume_norm", "patch", "hourglass_att", "classif_att_", "concat_stem", "hourglass", "classif")


class _FProxy:
    """models.SemStereo's `F` (torch.nn.functional) with softmax recorded: the attention probabilities the reference
    sorts (models/SemStereo.py:298-299) are an anonymous intermediate of forward()."""

    def __init__(self, real, log):
        self._real, self._log = real, log

    def __getattr__(self, name):
        return getattr(self._real, name)

    def softmax(self, x, dim=None, **k):
        y = self._real.softmax(x, dim=dim, **k)
        self._log.append((tuple(x.shape), dim, y))
        return y


def run_reference_segment(ms, net, B, H, W, feats):
    """One eval-mode pass of the reference's forward() with the out-of-scope producers replaced by replays of the
    closed-form feature maps; returns the captured intermediates."""
    fl4, fr4, fl8, fr8 = feats
    z = lambda c, s: torch.zeros(B, c, H // s, W // s)
    saved = {k: getattr(net, k) for k in ("feature", "feature_up", "head_l", "head_r", "chal_0", "chal_1", "chal_2", "chal_3", "chal_4")}
    net.feature = Replay([z(1, 2)] * 5, [z(1, 2)] * 5)
    net.feature_up = Replay(([z(1, 2)] * 5, [z(1, 2)] * 5))
    net.head_l = Replay(torch.zeros(B, 6, H, W)); net.head_r = Replay(torch.zeros(B, 6, H, W))
    net.chal_0 = Replay(z(64, 2)); net.chal_3 = Replay(z(384, 16)); net.chal_4 = Replay(z(256, 32))
    net.chal_1 = Replay(fl4, fr4)
    net.chal_2 = Replay(fl8, fr8)
    cap, originals, softmaxes = {}, {}, []
    for name in ("build_gwc_volume_norm", "regression_topk", "disparity_regression"):
        orig = originals[name] = getattr(ms, name)
        def wrap(*a, _o=orig, _n=name, **k):
            r = _o(*a, **k)
            cap.setdefault(_n, []).append((a, r))
            return r
        setattr(ms, name, wrap)
    realF = ms.F
    ms.F = _FProxy(realF, softmaxes)
    hooks = []
    for name in ("patch", "hourglass_att", "classif_att_", "concat_stem", "hourglass", "classif"):
        hooks.append(getattr(net, name).register_forward_hook(lambda m_, i_, o_, _n=name: cap.__setitem__(_n, o_)))
    hooks.append(net.ssr_upsample.register_forward_pre_hook(lambda m_, i_: cap.setdefault("ssr_in", []).append(i_[0])))
    try:
        with torch.no_grad():
            net(torch.zeros(B, 3, H, W), torch.zeros(B, 3, H, W))
    finally:
        for h in hooks:
            h.remove()
        for name, orig in originals.items():
            setattr(ms, name, orig)
        ms.F = realF
        for k, v in saved.items():
            setattr(net, k, v)
    probs = [y for shp, dim, y in softmaxes if len(shp) == 5 and dim == 2]
    assert len(probs) == 1, [(shp, dim) for shp, dim, _ in softmaxes]
    cap["aw_prob"] = probs[0]                                       # [B,1,D4,H4,W4], models/SemStereo.py:298
    return cap


def calibrate_batchnorm(ms, net, B, H, W, feats):
    """Running statistics := the statistics of this input (one pass with batch statistics, momentum 1), for every
    BatchNorm of the hot segment -- what training leaves behind.  Returns {state_dict key: tensor}."""
    bns = {}
    for prefix in OWNED:
        mod = getattr(net, prefix)
        if isinstance(mod, nn.Module):
            for name, m in mod.named_modules():
                if isinstance(m, nn.modules.batchnorm._BatchNorm):
                    bns[prefix + ("." + name if name else "")] = m
    for m in bns.values():
        m.momentum = 1.0
        m.train()
    run_reference_segment(ms, net, B, H, W, feats)
    out = {}
    for key, m in bns.items():
        m.eval()
        m.momentum = 0.1
        out[key + ".running_mean"] = m.running_mean.detach().clone()
        out[key + ".running_var"] = m.running_var.detach().clone()
    return out


def decision_gaps(cap):
    """The reference's own margins at its two hard picks, per pixel [B,H4,W4]:
    gap24_rel = (p24 - p25) / p24 of the sorted attention probabilities (the top-24 cut, :299-303);
    gap2 = 2nd - 3rd largest of the 24 matching costs (the top-2 cut of regression_topk, models/submodule.py:436-437)."""
    p = cap["aw_prob"].squeeze(1).sort(dim=1, descending=True).values
    gap24 = (p[:, 23] - p[:, 24]) / p[:, 23]
    (cost_sq, samples, k), pred = cap["regression_topk"][0]
    c = cost_sq.sort(dim=1, descending=True).values
    return gap24, c[:, 1] - c[:, 2]


def gen_segment(ms):
    out = {}
    table = dict(cases.SEGMENT)
    table.update(cases.SEGMENT_CAL)
    for n, (B, H, W, maxdisp) in table.items():
        net, P = build_ref_net(ms, maxdisp)
        fl4, fr4, fl8, fr8, _ = cases.segment_inputs(n)
        feats = (fl4, fr4, fl8, fr8)
        if n.endswith("_cal"):
            for key, v in calibrate_batchnorm(ms, net, B, H, W, feats).items():
                assert key in P, key
                out[f"{n}/bn/{key}"] = f32(v)
        cap = run_reference_segment(ms, net, B, H, W, feats)
        (cost_sq, samples, k), pred = cap["regression_topk"][0]
        assert k == 2
        out[f"{n}/pred"] = f32(pred)
        out[f"{n}/samples"] = samples.numpy().astype(np.int16)
        out[f"{n}/pred_att0"] = f32(cap["disparity_regression"][0][1])
        out[f"{n}/pred_att"] = f32(cap["ssr_in"][0].squeeze(1))
        assert torch.equal(cap["ssr_in"][1], pred)
        for i, key in enumerate(STAGES):
            t = cap[key][0][1] if key == "build_gwc_volume_norm" else cap[key]
            out[f"{n}/sum/{key}"] = summary(t, i)
        out[f"{n}/cost_att"] = f32(cap["classif_att_"])       # [B,1,D8,H8,W8]: small
        gap24, gap2 = decision_gaps(cap)
        out[f"{n}/gap24_rel"], out[f"{n}/gap2"] = f32(gap24), f32(gap2)
        # the reference's 24 attention weights per pixel: lets the matching branch be checked on the reference's OWN
        # candidates (no top-24 difference upstream), where the only hard pick left is the top-2 of the costs
        ind = (samples + maxdisp // 4).long().unsqueeze(1)
        out[f"{n}/att_topk"] = f32(torch.gather(cap["aw_prob"], 2, ind).squeeze(1))
        # float64 answer of the matching branch on the reference's candidates (see gen_segment_full)
        Pn = P_cal(P, out, n) if n.endswith("_cal") else P
        out[f"{n}/pred_truth"] = f32(oseg.matching_truth_tiled(Pn, fl4, fr4, torch.gather(cap["aw_prob"], 2, ind), samples))
        print(n, "pred", tuple(pred.shape), "range", float(pred.min()), float(pred.max()),
              "| min gap24_rel %.2e  min gap2 %.2e  median gap2 %.2e  cost std over candidates %.3f" %
              (float(gap24.min()), float(gap2.min()), float(gap2.median()), float(cost_sq.std(dim=1).mean())))
    np.savez_compressed(os.path.join(HERE, "segment.npz"), **out)
    print("segment.npz:", len(out), "arrays")


def P_cal(P, out, n):
    """The case's parameter dict with the calibrated BatchNorm statistics just stored under out[n/bn/...]."""
    Q = dict(P)
    pre = f"{n}/bn/"
    for k, v in out.items():
        if k.startswith(pre):
            Q[k[len(pre):]] = torch.from_numpy(np.asarray(v, dtype=np.float32).copy())
    return Q


def gen_segment_full(ms, only=None):
    """Checksum records at the full sizes of BASELINE.json configs[1] / configs[4]."""
    path = os.path.join(HERE, "segment_full.npz")
    out = dict(np.load(path)) if (only and os.path.exists(path)) else {}
    for n, (B, H, W, maxdisp) in cases.SEGMENT_FULL.items():
        if only and n not in only:
            continue
        import time
        t0 = time.time()
        net, P = build_ref_net(ms, maxdisp)
        fl4, fr4, fl8, fr8, _ = cases.segment_inputs(n)
        feats = (fl4, fr4, fl8, fr8)
        if "_cal" in n:                        # (a name without "_cal": the default statistics of random-init weights, r06)
            for key, v in calibrate_batchnorm(ms, net, B, H, W, feats).items():
                out[f"{n}/bn/{key}"] = f32(v)
        cap = run_reference_segment(ms, net, B, H, W, feats)
        (cost_sq, samples, k), pred = cap["regression_topk"][0]
        H4, W4 = H // 4, W // 4
        idx = torch.from_numpy(cases.sample_index(B * H4 * W4, cases.FULL_SAMPLES, salt=77))
        flat = lambda t: t.reshape(B, -1, H4 * W4).permute(0, 2, 1).reshape(B * H4 * W4, -1)[idx]     # [n, channels]
        gap24, gap2 = decision_gaps(cap)
        out[f"{n}/pixels"] = idx.numpy().astype(np.int64)
        out[f"{n}/pred"] = f32(flat(pred)[:, 0])
        out[f"{n}/pred_att"] = f32(flat(cap["ssr_in"][0])[:, 0])
        out[f"{n}/pred_att0"] = f32(flat(cap["disparity_regression"][0][1])[:, 0])
        out[f"{n}/samples"] = flat(samples).numpy().astype(np.int16)
        out[f"{n}/gap24_rel"], out[f"{n}/gap2"] = f32(flat(gap24)[:, 0]), f32(flat(gap2)[:, 0])
        out[f"{n}/cost"] = f32(flat(cost_sq))
        for i, key in enumerate(STAGES):
            t = cap[key][0][1] if key == "build_gwc_volume_norm" else cap[key]
            out[f"{n}/sum/{key}"] = summary(t, i)
        # whole-map statistics of the outputs and of the margins (how many pixels of the map sit near a tie)
        out[f"{n}/pred_sum"] = summary(pred, 50)
        out[f"{n}/pred_att_sum"] = summary(cap["ssr_in"][0], 51)
        # the whole maps, compactly: pred itself (the north-star output), a 16-bit hash of every pixel's candidate list,
        # and the pixels where the reference's own margins are below the tests' DELTA (the only places where another
        # fp32 evaluation may legitimately pick differently)
        out[f"{n}/pred_map"] = f32(pred.squeeze(1))
        out[f"{n}/pred_att_map"] = f32(cap["ssr_in"][0].squeeze(1))
        out[f"{n}/candidate_hash"] = cases.candidate_set_hash(samples.numpy(), maxdisp // 4)
        out[f"{n}/risk24"] = np.flatnonzero((gap24 < 1e-4).reshape(-1).numpy()).astype(np.int32)
        out[f"{n}/risk2"] = np.flatnonzero((gap2 < 1e-4).reshape(-1).numpy()).astype(np.int32)
        out[f"{n}/gap_stats"] = np.array([float(gap24.min()), float(gap24.median()), float((gap24 < 1e-4).float().mean()),
                                          float(gap2.min()), float(gap2.median()), float((gap2 < 1e-4).float().mean())])
        # round 3 (strict full-size parity): at the risk24 pixels -- the only ones where another fp32 evaluation may select
        # other candidates -- the REFERENCE's own 24 candidates, attention weights and margin, so that a test can put
        # the reference's pick back wherever the HIP path chose differently and then hold EVERY pixel of `pred` to the bound
        # with no receptive-field excuse
        r24 = torch.from_numpy(out[f"{n}/risk24"].astype(np.int64))
        per_px = lambda t: t.reshape(B, -1, H4 * W4).permute(0, 2, 1).reshape(B * H4 * W4, -1)                 # [pixels, channels]
        ind = (samples + maxdisp // 4).long().unsqueeze(1)
        att_ref = torch.gather(cap["aw_prob"], 2, ind).squeeze(1)                                            # [B,24,H4,W4], :304
        out[f"{n}/risk24_samples"] = per_px(samples)[r24].numpy().astype(np.int16)
        out[f"{n}/risk24_att_topk"] = f32(per_px(att_ref)[r24])
        out[f"{n}/risk24_gap24_rel"] = f32(per_px(gap24)[r24][:, 0])
        # ... and what the EXACT (float64) evaluation of the attention branch selects at those pixels (the pinned oracle in
        # double precision): where an fp32 implementation and the reference disagree at such a margin, this says whose
        # rounding it is
        Pc = P_cal(P, out, n)
        P64 = {k: (v.double() if v.is_floating_point() else v) for k, v in Pc.items()}
        with torch.no_grad():
            _, smp64, _ = oseg.attention_branch(P64, fl8.double(), fr8.double(), fl4.double(), fr4.double(), maxdisp)
        out[f"{n}/risk24_truth_samples"] = per_px(smp64)[r24].numpy().astype(np.int16)
        print(n, "float64 attention branch: its candidates equal the reference's on", int((per_px(smp64)[r24] == per_px(samples)[r24]).all(dim=1).sum()),
              "of", r24.numel(), "risk24 pixels;", int((smp64 != samples).any(dim=1).sum()), "pixels of the whole map differ")
        del smp64, P64
        # ... and the float64 answer of the matching branch (models/SemStereo.py:314-323) on the reference's candidates
        # ("truth"; the pinned oracle in double precision -- the reference's forward() cannot be split): what an fp32
        # evaluation of this graph -- the reference's own included -- can be held to at this depth of the soft-argmax
        out[f"{n}/pred_truth_map"] = f32(oseg.matching_truth_tiled(Pc, fl4, fr4, att_ref.unsqueeze(1), samples).squeeze(1))
        print(n, "pred range", float(pred.min()), float(pred.max()), "gap stats", out[f"{n}/gap_stats"], "%.0f s" % (time.time() - t0))
        del cap, net
    np.savez_compressed(path, **out)
    print("segment_full.npz:", len(out), "arrays,", os.path.getsize(path) // 1024, "KiB")


def gen_ssr(ref):
    from oracle import ssr as ossr
    P = ossr.deterministic_ssr_params()
    mod = ref.SSR_upsample(6).eval()
    res = mod.load_state_dict({k[len("ssr_upsample."):]: v for k, v in P.items()}, strict=False)
    assert not res.unexpected_keys and all(k.endswith("num_batches_tracked") for k in res.missing_keys), res
    out = {}
    with torch.no_grad():
        for n in cases.SSR:
            d, w, l = cases.ssr_inputs(n)
            out[f"ssr/{n}"] = f32(mod(d, w, l))
    np.savez_compressed(os.path.join(HERE, "ssr.npz"), **out)
    print("ssr.npz:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    assert os.path.isdir(REF), "the reference is only mounted in the build container"
    what = sys.argv[1:] or ["ssr", "ops", "stack", "segment", "full", "whu"]
    if "whu" in what:
        gen_ops_unsigned(load_ref_unsigned_oplib())
    if "ssr" in what:
        gen_ssr(load_ref_oplib())
    if "ops" in what:
        gen_ops(load_ref_oplib())
    ms = load_ref_model_module()
    if "stack" in what:
        gen_stack(ms)
    if "segment" in what:
        gen_segment(ms)
    if "whu" in what:
        gen_segment_whu(ms, load_ref_unsigned_oplib())
    if "full" in what or any(w in cases.SEGMENT_FULL for w in what):      # ~1 min and ~5 min of CPU, ~25 GB at 2048^2
        gen_segment_full(ms, only=[w for w in what if w in cases.SEGMENT_FULL] or None)
    for f in ("ops.npz", "stack.npz", "segment.npz", "segment_full.npz", "ops_unsigned.npz", "segment_whu.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")
