"""Strict hot-segment parity against the REFERENCE's fixtures (test infrastructure; also used by bench.py's parity leg).

The graph has two hard picks (24 of D4 attention probabilities, models/SemStereo.py:299-303; 2 of 24 matching costs,
models/submodule.py:436-437).  The attention branch runs on the HIP path; wherever it selected other candidates than the
reference -- allowed ONLY where the reference's own 24th / 25th probabilities are within cases.DELTA24_REL -- the
reference's own candidates and attention weights (stored in the fixture for exactly those pixels) are put back, and the
matching branch runs on the result.  No candidate difference is then left upstream, so EVERY pixel of `pred` is held to
the bound unless the reference's own 2nd / 3rd largest costs are within cases.DELTA2: no receptive-field excuse, no
percentage.  The fixtures also hold the float64 answer of the matching branch on the reference's candidates
(`pred_truth`, oracle.hot_segment.matching_truth_tiled), which says how far an fp32 evaluation of this graph -- the
reference's own -- is from the exact answer at that depth of the soft-argmax.
"""
import numpy as np
import torch

from golden import cases


def fixture_view(g, name):
    """The arrays the strict check needs, in one form for the small fixtures (segment.npz: whole maps of everything) and the
    full-size ones (segment_full.npz: whole maps of pred / hash, candidate lists only at the risk24 pixels).
    -> dict of torch tensors (flat pixel order b*H4*W4 + y*W4 + x) or None when the fixture predates round 3."""
    B, H, W, maxdisp = cases.segment_shape(name)
    H4, W4, m4 = H // 4, W // 4, maxdisp // 4
    n = B * H4 * W4
    files = set(g.files)
    v = dict(B=B, H4=H4, W4=W4, m4=m4, n=n)
    if f"{name}/pred_map" in files:                                   # full-size record
        if f"{name}/risk24_samples" not in files:
            return None
        v["pred"] = torch.as_tensor(g[f"{name}/pred_map"]).reshape(n)
        v["truth"] = torch.as_tensor(g[f"{name}/pred_truth_map"]).reshape(n)
        v["hash"] = torch.from_numpy(g[f"{name}/candidate_hash"].astype(np.int64)).reshape(n)
        v["risk_idx"] = torch.from_numpy(g[f"{name}/risk24"].astype(np.int64))
        v["risk_samples"] = torch.from_numpy(g[f"{name}/risk24_samples"].astype(np.float32))
        v["risk_att"] = torch.as_tensor(g[f"{name}/risk24_att_topk"])
        v["risk_gap24"] = torch.as_tensor(g[f"{name}/risk24_gap24_rel"])
        if f"{name}/risk24_truth_samples" in files:          # what the float64 evaluation of the attention branch selects there
            v["risk_truth_samples"] = torch.from_numpy(g[f"{name}/risk24_truth_samples"].astype(np.float32))
        tie = torch.zeros(n, dtype=torch.bool)
        tie[torch.from_numpy(g[f"{name}/risk2"].astype(np.int64))] = True
        v["tie"] = tie
    else:
        if f"{name}/pred_truth" not in files:
            return None
        per_px = lambda a: torch.as_tensor(a).reshape(B, -1, H4 * W4).permute(0, 2, 1).reshape(n, -1)   # noqa: E731
        v["pred"] = torch.as_tensor(g[f"{name}/pred"]).reshape(n)
        v["truth"] = torch.as_tensor(g[f"{name}/pred_truth"]).reshape(n)
        v["hash"] = torch.from_numpy(cases.candidate_set_hash(g[f"{name}/samples"], m4).astype(np.int64)).reshape(n)
        v["risk_idx"] = torch.arange(n)
        v["risk_samples"] = per_px(g[f"{name}/samples"].astype(np.float32))
        v["risk_att"] = per_px(g[f"{name}/att_topk"])
        v["risk_gap24"] = torch.as_tensor(g[f"{name}/gap24_rel"]).reshape(n)
        v["tie"] = torch.as_tensor(g[f"{name}/gap2"]).reshape(n) < cases.DELTA2
    return v


def restore_reference_picks(v, att_topk, samples):
    """att_topk [B,1,24,H4,W4], samples [B,24,H4,W4] of the HIP attention branch (device tensors, modified IN PLACE): at
    every pixel whose candidate list differs from the reference's, write the reference's candidates and weights.
    -> (differs [n] bool, unexplained [n] bool: differing although the reference's margin is >= DELTA24_REL)."""
    B, H4, W4, m4, n = v["B"], v["H4"], v["W4"], v["m4"], v["n"]
    mine = torch.from_numpy(cases.candidate_set_hash(samples.detach().cpu().numpy(), m4).astype(np.int64)).reshape(n)
    differs = mine != v["hash"]
    row_of = torch.full((n,), -1, dtype=torch.long)
    row_of[v["risk_idx"]] = torch.arange(v["risk_idx"].numel())
    explained = torch.zeros(n, dtype=torch.bool)
    explained[v["risk_idx"][v["risk_gap24"] < cases.DELTA24_REL]] = True
    unexplained = differs & ~explained
    px = torch.nonzero(differs & (row_of >= 0)).reshape(-1)
    v["last_differing"] = {"pixels": px.tolist(), "reference_margin_rel": [float(v["risk_gap24"][r]) for r in row_of[px]]}
    if px.numel():
        rows = row_of[px]
        b, y, x = px // (H4 * W4), (px % (H4 * W4)) // W4, px % W4
        dev = samples.device
        if "risk_truth_samples" in v:                       # whose rounding is it?  compare both picks with the exact evaluation's
            mine = samples[b.to(dev), :, y.to(dev), x.to(dev)].cpu()
            v["last_differing"]["hip_pick_equals_float64_truth"] = int((mine == v["risk_truth_samples"][rows]).all(dim=1).sum())
            v["last_differing"]["reference_pick_equals_float64_truth"] = int((v["risk_samples"][rows] == v["risk_truth_samples"][rows]).all(dim=1).sum())
        samples[b.to(dev), :, y.to(dev), x.to(dev)] = v["risk_samples"][rows].to(dev)
        att_topk[b.to(dev), 0, :, y.to(dev), x.to(dev)] = v["risk_att"][rows].to(dev)
    return differs, unexplained


def strict_report(v, pred, differs):
    """Statistics of `pred` [B,1,H4,W4] (computed after restore_reference_picks) against the reference and the truth."""
    p = pred.detach().cpu().reshape(v["n"]).double()
    ref, tru, tie = v["pred"].double(), v["truth"].double(), v["tie"]
    err, e_hip, e_ref = (p - ref).abs(), (p - tru).abs(), (ref - tru).abs()
    free = ~tie
    return {
        "pixels": int(v["n"]), "pixels_with_other_candidates": int(differs.sum()), "pixels_at_top2_ties": int(tie.sum()),
        "epe_vs_reference_px": float(err.mean()), "epe_vs_reference_fullres_px": 4.0 * float(err.mean()),
        "epe_vs_reference_off_ties_px": float(err[free].mean()), "median_abs_err_px": float(err.median()),
        "max_err_off_ties_px": float(err[free].max()), "pixels_beyond_1e-3": int((err > 1e-3).sum()),
        "pixels_beyond_1e-3_off_ties": int(((err > 1e-3) & free).sum()),
        "hip_vs_truth_epe_px": float(e_hip.mean()), "reference_vs_truth_epe_px": float(e_ref.mean()),
        "hip_vs_truth_max_off_ties_px": float(e_hip[free].max()), "reference_vs_truth_max_off_ties_px": float(e_ref[free].max()),
        "hip_vs_truth_epe_off_ties_px": float(e_hip[free].mean()), "reference_vs_truth_epe_off_ties_px": float(e_ref[free].mean()),
    }


def run_strict(seg, g, name, device="cuda"):
    """HIP attention branch -> reference picks restored where they differ -> HIP matching branch.  -> (report, unexplained)."""
    v = fixture_view(g, name)
    assert v is not None, f"{name}: the fixture holds no reference candidate lists / truth (regenerate with make_golden.py)"
    fl4, fr4, fl8, fr8, _ = cases.segment_inputs(name)
    fl4, fr4, fl8, fr8 = [t.to(device) for t in (fl4, fr4, fl8, fr8)]
    with torch.no_grad():
        att_topk, samples, pred_att, _ = seg.attention_branch(fl4, fr4, fl8, fr8)
        att_topk, samples = att_topk.clone(), samples.clone()
        differs, unexplained = restore_reference_picks(v, att_topk, samples)
        pred = seg.matching_branch(fl4, fr4, att_topk, samples)
    rep = strict_report(v, pred, differs)
    rep["unexplained_candidate_differences"] = int(unexplained.sum())
    rep["differing_pixels"] = v.get("last_differing")
    return rep, v, pred, differs, unexplained
