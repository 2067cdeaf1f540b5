"""GPU parity at BASELINE.json's FULL sizes through size-independent properties (the oracle cannot run these
sizes in seconds, so here it is applied to sampled slices, and the rest are identities the reference
algorithm satisfies at any size): linearity / cross-op consistency of the volume builders, closed-form
values, determinism, batch (= shard) invariance, candidate-set invariants of the hot segment.
Run on the MI355X box: pytest -m gpu."""
import numpy as np
import pytest
import torch

from golden import cases
from oracle import ops as oops

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sa():
    import semstereo_amd
    assert torch.cuda.is_available(), "these tests need the MI355X"
    semstereo_amd._lib.load()
    return semstereo_amd


def _feat(shape, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return torch.randn(shape, generator=g, device="cuda")


def test_gwc_volume_batch8_properties(sa):
    """configs[2]: 2 x [8,256,128,128] -> [8,32,32,128,128] (maxdisp//8 = 16, 32 groups)."""
    B, C, H, W, m, G = 8, 256, 128, 128, 16, 32
    fl, fr = _feat((B, C, H, W), 1), _feat((B, C, H, W), 2)
    v = sa.ops.build_gwc_volume(fl, fr, m, G)
    assert v.shape == (B, G, 2 * m, H, W)
    # linearity in each argument: powers of two are exact in fp32 -> bit-identical
    assert torch.equal(sa.ops.build_gwc_volume(fl * 4.0, fr, m, G), v * 4.0)
    assert torch.equal(sa.ops.build_gwc_volume(fl, fr * 0.5, m, G), v * 0.5)
    # the zero-disparity plane is the plain group-wise correlation (models/submodule.py:190-196 vs :198-211)
    assert torch.equal(v[:, :, m], sa.ops.groupwise_correlation(fl, fr, G))
    # plane d is the correlation with the right image shifted by d - m, zero where the partner leaves the image
    for d in (0, 5, 2 * m - 1):
        s = d - m
        sh = torch.zeros_like(fr)
        if s >= 0:
            sh[..., s:] = fr[..., :W - s]
        else:
            sh[..., :W + s] = fr[..., -s:]
        ref = sa.ops.groupwise_correlation(fl, sh, G)
        valid = torch.zeros(W, dtype=torch.bool, device="cuda")
        valid[max(s, 0):W + min(s, 0)] = True
        assert torch.equal(v[:, :, d][..., valid], ref[..., valid])
        assert float(v[:, :, d][..., ~valid].abs().max()) == 0.0 if (~valid).any() else True
    # the oracle on sampled (batch, row) slices, both variants
    vn = sa.ops.build_gwc_volume_norm(fl, fr, m, G)
    for b, y in ((0, 0), (3, 77), (7, 127)):
        a, c = fl[b:b + 1, :, y:y + 1].cpu(), fr[b:b + 1, :, y:y + 1].cpu()
        assert float((v[b:b + 1, :, :, y:y + 1].cpu() - oops.build_gwc_volume(a, c, m, G)).abs().max()) <= 1e-6
        assert float((vn[b:b + 1, :, :, y:y + 1].cpu() - oops.build_gwc_volume_norm(a, c, m, G)).abs().max()) <= 2e-6
    # normalised volume: |value| <= 1 (Cauchy-Schwarz on unit vectors, mean over 8 channels) and scale invariance
    assert float(vn.abs().max()) <= 1.0 + 1e-6
    assert float((sa.ops.build_gwc_volume_norm(fl * 8.0, fr * 0.25, m, G) - vn).abs().max()) <= 2e-5


def test_concat_volume_full_size_is_pure_data_movement(sa):
    """natural shape of build_concat_volume at 1024^2 / maxdisp 128: 2 x [1,32,256,256] -> [1,64,64,256,256] (1.07 GB)."""
    C, H, W, m = 32, 256, 256, 32
    fl, fr = _feat((1, C, H, W), 3), _feat((1, C, H, W), 4)
    v = sa.ops.build_concat_volume(fl, fr, m)
    assert v.shape == (1, 2 * C, 2 * m, H, W)
    xs = torch.arange(W, device="cuda")
    for d in (0, 31, 32, 63):
        s = d - m
        valid = (xs - s >= 0) & (xs - s < W)
        assert torch.equal(v[0, :C, d][..., valid], fl[0][..., valid])                    # left half: the left image
        assert torch.equal(v[0, C:, d][..., valid], fr[0][..., (xs - s)[valid]])          # right half: shifted right image
        assert float(v[0, :, d][..., ~valid].abs().sum()) == 0.0                            # both halves zero outside
    # checksum of checksums: every valid column of the left half repeats the left image once per disparity
    n_valid = sum(int(((xs - (d - m) >= 0) & (xs - (d - m) < W)).sum()) for d in range(2 * m))
    assert n_valid == sum(W - abs(d - m) for d in range(2 * m))
    y = 100
    assert torch.equal(v[:, :, :, y:y + 1].cpu(), oops.build_concat_volume(fl[:, :, y:y + 1].cpu(), fr[:, :, y:y + 1].cpu(), m))


def test_regressions_closed_forms_full_size(sa):
    m, H, W = 32, 256, 256
    # uniform probabilities: E[d] over [-m, m) is exactly -0.5; a one-hot plane returns its disparity
    p = torch.full((2, 2 * m, H, W), 1.0 / (2 * m), device="cuda")
    assert float((sa.ops.disparity_regression(p, m) + 0.5).abs().max()) <= 1e-6
    onehot = torch.zeros((1, 2 * m, H, W), device="cuda")
    onehot[:, 7] = 1.0
    assert torch.equal(sa.ops.disparity_regression(onehot, m), torch.full((1, H, W), 7.0 - m, device="cuda"))
    # regression_topk: with a single dominant cost the answer is that candidate, whatever the rest is
    cost = _feat((1, 24, H, W), 5)
    base = torch.randperm(2 * m, generator=torch.Generator().manual_seed(6))[:24].sort().values.float() - m
    samples = base.reshape(1, 24, 1, 1).expand(1, 24, H, W).contiguous().cuda()
    big = cost.clone()
    big[:, 11] += 100.0
    assert float((sa.ops.regression_topk(big, samples, 2)[:, 0] - samples[:, 11]).abs().max()) <= 1e-5
    # permutation invariance over the candidate axis (sort is part of the op)
    perm = torch.randperm(24, device="cuda")
    assert float((sa.ops.regression_topk(cost[:, perm].contiguous(), samples[:, perm].contiguous(), 2)
                  - sa.ops.regression_topk(cost, samples, 2)).abs().max()) <= 1e-5


@pytest.mark.parametrize("size", [(1024, 128), (2048, 192)])
def test_hot_segment_full_size_invariants(sa, size):
    """configs[1] / configs[4] shapes: determinism, batch (= shard) invariance, candidate-set invariants."""
    import bench
    Hf, maxdisp = size
    m4 = maxdisp // 4
    seg = sa.HotSegment(maxdisp).cuda().eval()
    bench.init_unit_gain(seg, 4321)
    fl4a, fr4a = bench.synth_features(1, 128, Hf // 4, Hf // 4, maxdisp // 8, 11, "cuda")
    fl8a, fr8a = bench.synth_features(1, 256, Hf // 8, Hf // 8, maxdisp // 16, 12, "cuda")
    before = dict(sa.modules.PATH_COUNTS)
    with torch.no_grad():
        r1 = seg(fl4a, fr4a, fl8a, fr8a)
        r2 = seg(fl4a, fr4a, fl8a, fr8a)
    assert sa.modules.PATH_COUNTS["torch"] == before["torch"], "a PyTorch fallback ran"
    for k in ("pred", "pred_att", "samples", "att_topk"):
        assert torch.equal(r1[k], r2[k]), f"{k} differs between two identical launches"
    s = r1["samples"]
    assert s.shape == (1, 24, Hf // 4, Hf // 4)
    assert bool((s[:, 1:] > s[:, :-1]).all()), "candidates must be strictly ascending (24 distinct disparities)"
    assert float(s.min()) >= -m4 and float(s.max()) <= m4 - 1 and bool((s == s.round()).all())
    a = r1["att_topk"]
    assert float(a.min()) > 0.0 and float(a.sum(dim=2).max()) <= 1.0 + 1e-5      # 24 of the D4 softmax probabilities
    # pred_att / pred are convex combinations of the pixel's candidates
    for k in ("pred_att", "pred"):
        v = r1[k].reshape(1, 1, Hf // 4, Hf // 4)
        assert bool((v >= s[:, :1] - 1e-4).all()) and bool((v <= s[:, -1:] + 1e-4).all()), k
    assert bool(torch.isfinite(r1["pred"]).all())
    if Hf == 1024:
        # two different pairs in one batch == each pair alone: nothing couples batch elements, which is
        # what lets the path shard over GPUs with no collective
        fl4b, fr4b = bench.synth_features(1, 128, Hf // 4, Hf // 4, maxdisp // 8, 21, "cuda")
        fl8b, fr8b = bench.synth_features(1, 256, Hf // 8, Hf // 8, maxdisp // 16, 22, "cuda")
        with torch.no_grad():
            rb = seg(fl4b, fr4b, fl8b, fr8b)
            rab = seg(torch.cat((fl4a, fl4b)), torch.cat((fr4a, fr4b)), torch.cat((fl8a, fl8b)), torch.cat((fr8a, fr8b)))
        # (the 2-D convolutions / GEMMs of the gates are MIOpen / rocBLAS calls whose algorithm may change
        # with the batch size, so agreement is to rounding, not bitwise)
        for one, idx in ((r1, 0), (rb, 1)):
            assert float((rab["samples"][idx:idx + 1] == one["samples"]).float().mean()) >= 0.9999
            assert float((rab["pred_att"][idx:idx + 1] - one["pred_att"]).abs().median()) <= 1e-5
            d = (rab["pred"][idx:idx + 1] - one["pred"]).abs()
            assert float(d.median()) <= 1e-5 and float((d <= 1e-3).float().mean()) >= 0.99


def _batch_of_pairs(bench, n, Hf, maxdisp):
    fl4, fr4, fl8, fr8 = [], [], [], []
    for i in range(n):
        a, b = bench.synth_features(1, 128, Hf // 4, Hf // 4, maxdisp // 8, 31 + 2 * i, "cuda")
        c, d = bench.synth_features(1, 256, Hf // 8, Hf // 8, maxdisp // 16, 32 + 2 * i, "cuda")
        fl4.append(a); fr4.append(b); fl8.append(c); fr8.append(d)
    return [torch.cat(t) for t in (fl4, fr4, fl8, fr8)]


@pytest.mark.parametrize("batch", [4, 8])
def test_hot_segment_batch_invariance_at_the_sharded_batch_sizes(sa, batch):
    """BASELINE.json configs[3] gives every GPU 4 pairs of 1024 x 1024 / maxdisp 128 (32 over 8 GPUs), configs[2] runs 8 on
    one: the hot segment on a batch must give each pair exactly what it gets alone (nothing couples batch elements; this
    is what makes the shard-by-pairs multi-GPU form of SURVEY.md section 8e valid), with no PyTorch fallback."""
    if sa.modules.CONV_ENGINE == "bf16x3":
        pytest.skip("SS_CONV_ENGINE=bf16x3: a batch moves the smallest transposed conv from the exact-fp32 kernel to the split "
                    "engine (DECONV_MIN_WORKGROUPS); with the 3-product form that is a ~1e-5 change, beyond the margins below")
    import bench
    Hf, maxdisp = 1024, 128
    seg = sa.HotSegment(maxdisp).cuda().eval()
    bench.init_unit_gain(seg, 4321)
    feats = _batch_of_pairs(bench, batch, Hf, maxdisp)
    before = dict(sa.modules.PATH_COUNTS)
    with torch.no_grad():
        rb = seg(*feats)
        rb2 = seg(*feats)
    assert sa.modules.PATH_COUNTS["torch"] == before["torch"], "a PyTorch fallback ran"
    assert rb["pred"].shape == (batch, 1, Hf // 4, Hf // 4) and bool(torch.isfinite(rb["pred"]).all())
    for k in ("pred", "pred_att", "samples"):
        assert torch.equal(rb[k], rb2[k]), f"{k} differs between two identical launches at batch {batch}"
    for i in (0, batch // 2, batch - 1):
        with torch.no_grad():
            one = seg(*[t[i:i + 1].contiguous() for t in feats])
        # every kernel of the path is this repo's and batch-invariant by construction (the batch is a grid dimension);
        # kernels that switch their store policy or tile with the volume size stay bit-identical per element
        same = (rb["samples"][i:i + 1] == one["samples"]).all(dim=1)
        assert float(same.float().mean()) >= 0.9999, f"pair {i}: candidate sets differ on {int((~same).sum())} pixels"
        d_att = (rb["pred_att"][i:i + 1] - one["pred_att"]).abs()
        assert float(d_att[same].max()) <= 1e-4
        d = (rb["pred"][i:i + 1] - one["pred"]).abs().squeeze(1)
        assert float(d.median()) <= 1e-6 and float((d <= 1e-3).float().mean()) >= 0.999, (float(d.median()), float(d.max()))


DELTA24_REL, DELTA2, RF_RADIUS = cases.DELTA24_REL, cases.DELTA2, 36      # ONE definition (tests/golden/cases.py) for every hot-segment test


@pytest.mark.parametrize("name", sorted(cases.SEGMENT_FULL))
def test_hot_segment_full_size_vs_reference_checksums(sa, golden, name):
    """The REFERENCE's checksum record at the full sizes of BASELINE.json configs[1] (1024^2 / 128) and configs[4]
    (2048^2 / 192) -- tests/golden/segment_full.npz, made by make_golden.py from /root/reference: per-stage sums and 64
    sampled voxels, the whole `pred` / `pred_att` maps, a 16-bit hash of every pixel's 24 candidates, and the pixels where
    the reference's own margins at its two hard picks are below DELTA.  EVERY pixel must have the reference's candidates
    unless it is such a pixel; EVERY pixel must be within 1e-3 px unless its top-2 margin is below DELTA2 or a pixel
    with other candidates lies within the receptive field.  EPE over the whole map is reported (north star: < 1e-3)."""
    if sa.modules.CONV_ENGINE == "bf16x3":
        pytest.skip("SS_CONV_ENGINE=bf16x3: these bounds are for the fp32-accurate engines")
    import torch.nn.functional as F
    if "segment_full" not in golden:
        pytest.skip("no full-size fixture file")
    g = golden["segment_full"]
    if f"{name}/pred_map" not in g.files:
        pytest.skip(f"{name}: no fixture")
    B, H, W, maxdisp = cases.segment_shape(name)
    H4, W4, m4 = H // 4, W // 4, maxdisp // 4
    seg = sa.HotSegment(maxdisp)
    res = seg.load_state_dict(cases.segment_params(name, g), strict=False)
    assert not res.unexpected_keys and all(k.endswith("num_batches_tracked") for k in res.missing_keys)
    seg = seg.cuda().eval()
    fl4, fr4, fl8, fr8, _ = cases.segment_inputs(name)
    cap = {}
    hooks = [getattr(seg, k).register_forward_hook(lambda m_, i_, o_, _k=k: cap.__setitem__(_k, o_.detach()))
             for k in ("hourglass_att", "classif_att_", "hourglass", "classif")]
    before = dict(sa.modules.PATH_COUNTS)
    with torch.no_grad():
        r = seg(fl4.cuda(), fr4.cuda(), fl8.cuda(), fr8.cuda())
        cap["build_gwc_volume_norm"] = sa.ops.build_gwc_volume_norm(fl8.cuda(), fr8.cuda(), maxdisp // 8, 32)
    for h in hooks:
        h.remove()
    assert sa.modules.PATH_COUNTS["torch"] == before["torch"], "a PyTorch fallback ran"

    # candidate sets, every pixel
    mine = cases.candidate_set_hash(r["samples"].cpu().numpy(), m4).reshape(-1)
    differs = torch.from_numpy(mine != g[f"{name}/candidate_hash"].reshape(-1))
    risk24 = torch.zeros(B * H4 * W4, dtype=torch.bool)
    risk24[torch.as_tensor(g[f"{name}/risk24"].astype(np.int64))] = True
    risk2 = torch.zeros(B * H4 * W4, dtype=torch.bool)
    risk2[torch.as_tensor(g[f"{name}/risk2"].astype(np.int64))] = True
    # disparities, every pixel
    err_att = (r["pred_att"].cpu() - torch.as_tensor(g[f"{name}/pred_att_map"])).abs().reshape(-1)
    err = (r["pred"].cpu().squeeze(1) - torch.as_tensor(g[f"{name}/pred_map"])).abs().reshape(-1)
    near = F.max_pool2d(differs.reshape(B, 1, H4, W4).float(), 2 * RF_RADIUS + 1, stride=1, padding=RF_RADIUS).reshape(-1) > 0
    clean = ~risk2 & ~near
    rep = {"pixels": int(err.numel()), "pixels_with_other_candidates": int(differs.sum()), "at_risk24": int(risk24.sum()),
           "at_risk2": int(risk2.sum()), "epe_vs_reference_px": float(err.mean()), "epe_median": float(err.median()),
           "pixels_beyond_1e-3": int((err > 1e-3).sum()), "beyond_1e-3_at_top2_ties": int(((err > 1e-3) & risk2).sum()),
           "beyond_1e-3_near_other_candidates": int(((err > 1e-3) & ~risk2 & near).sum()),
           "max_err_where_no_excuse": float(err[clean].max()) if bool(clean.any()) else None,
           "fraction_no_excuse": float(clean.float().mean()), "pred_att_epe": float(err_att.mean())}
    rep["mean_err_where_no_excuse"] = float(err[clean].mean())
    # stages: sum of squares and the sampled voxels.  The attention branch's stages carry no hard pick: strict.
    stage_checks = []
    for i, key in enumerate(("build_gwc_volume_norm", "patch", "hourglass_att", "classif_att_", "concat_stem", "hourglass", "classif")):
        if key not in cap:
            continue
        rec = g[f"{name}/sum/{key}"]
        a = cap[key].reshape(-1)
        vox = a[torch.as_tensor(cases.sample_index(a.numel(), 64, i)).cuda()].double().cpu().numpy()
        ssq = float((a.double() * a.double()).sum())
        rep[f"stage/{key}/voxel_max_err"] = float(np.abs(vox - rec[2:]).max())
        rep[f"stage/{key}/sumsq_rel_err"] = abs(ssq - rec[1]) / rec[1]
        stage_checks.append((key, key in ("build_gwc_volume_norm", "hourglass_att", "classif_att_")))
    import json
    import os
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/fullsize_{name}.json", "w") as f:
        json.dump(rep, f, indent=1)
    for key, strict in stage_checks:
        if strict:
            assert rep[f"stage/{key}/voxel_max_err"] <= 2e-4 and rep[f"stage/{key}/sumsq_rel_err"] <= 1e-5, (key, rep)
        else:                        # downstream of the top-24 pick: voxels inside a differing pixel's receptive field may move
            assert rep[f"stage/{key}/sumsq_rel_err"] <= 1e-3, (key, rep)
    # (i) candidate sets: every pixel has the reference's 24 candidates unless the reference's own margin is below DELTA24_REL
    # (the fixture lists the pixels below 1e-4 with their margins)
    if f"{name}/risk24_gap24_rel" in g.files:
        risk24 = torch.zeros(B * H4 * W4, dtype=torch.bool)
        risk24[torch.as_tensor(g[f"{name}/risk24"].astype(np.int64))[torch.as_tensor(g[f"{name}/risk24_gap24_rel"]) < DELTA24_REL]] = True
    bad = differs & ~risk24
    assert not bool(bad.any()), f"{int(bad.sum())} pixel(s) select other candidates where the reference's margin is >= {DELTA24_REL}"
    bad = (err_att > 1e-3) & ~differs
    assert not bool(bad.any()), f"pred_att off by up to {float(err_att[bad].max()):.2e} on {int(bad.sum())} pixel(s) with the reference's candidates"
    # (ii) pred: outside the receptive field of a differing pixel and away from ties of the reference's costs, every pixel
    # within 1e-3 px at BOTH sizes and the mean far inside it.  (Until r04 the bound was 3e-3 at D4 = 96 and the measured worst
    # pixel 1.6e-3: that was the warp kernels' contracted coordinate arithmetic, not the soft-argmax's conditioning -- with it
    # fixed the default engine measures 1.1e-4 / 1.9e-4 at 1024^2 / 2048^2.)
    # (r05: the gathered stem computes the warped half EXACTLY where the reference's F.grid_sample carries its coordinate rounding,
    # so the distance to the reference is now the reference's own distance from the exact answer -- 3.3e-4 / 8.0e-4 px on its worst
    # pixel at the two sizes, tests/golden/make_golden.py -- and the bound against it is the north star's 1e-3.)
    default_engine = sa.modules.CONV_ENGINE == "f16x3"
    gathered = default_engine and sa.engine.STEM_GATHER
    bound = 1e-3 if (gathered or not default_engine) else 5e-4
    if "_cal" not in name and float(clean.float().mean()) < 0.05:
        # r06: the record with DEFAULT BatchNorm statistics has 13 % of its pixels within 1e-4 of a top-24 tie in the reference's own
        # evaluation; the receptive fields of the few that fall the other way cover the map, so this test's "no excuse" set is (nearly)
        # empty there -- the strict test below, which puts the reference's picks back, is the one that holds every pixel of that record
        return
    bad = (err > bound) & clean
    assert not bool(bad.any()), (f"pred off by up to {float(err[bad].max()):.2e} px on {int(bad.sum())} pixel(s) with no tie in the "
                                 f"reference's costs and no differing candidate set within {RF_RADIUS} px")
    assert float(err[clean].mean()) <= ((4e-5 if gathered else 3e-5) if default_engine else 1e-4) and float(err.median()) <= 1e-4


def test_hot_segment_full_size_strict_without_the_gathered_stem(sa, golden, monkeypatch):
    """the three-launch stem (SS_STEM_GATHER=0: warp kernel -> volume -> conv, the reference's own coordinate arithmetic
    restated operation by operation) keeps ROUND 4's bounds against the reference -- every pixel off its cost ties within 5e-4 px,
    HIP vs truth <= 1.25x / 1.3x the reference's own distance -- so that path cannot regress unnoticed behind the gathered default,
    whose distance to the reference is the reference's own error."""
    if sa.modules.CONV_ENGINE != "f16x3":
        pytest.skip("bounds of the default engine")
    import strict
    name = "f1024_md128_cal"
    if "segment_full" not in golden or strict.fixture_view(golden["segment_full"], name) is None:
        pytest.skip(f"{name}: no round-3 fixture")
    monkeypatch.setattr(sa.engine, "STEM_GATHER", False)
    g = golden["segment_full"]
    seg = sa.HotSegment(cases.segment_shape(name)[3])
    seg.load_state_dict(cases.segment_params(name, g), strict=False)
    seg = seg.cuda().eval()
    rep, v, pred, differs, unexplained = strict.run_strict(seg, g, name)
    ref_self, ref_mean = rep["reference_vs_truth_max_off_ties_px"], rep["reference_vs_truth_epe_off_ties_px"]
    assert not bool(unexplained.any())
    assert rep["max_err_off_ties_px"] <= 5e-4, rep
    assert rep["hip_vs_truth_max_off_ties_px"] <= 1.25 * ref_self and rep["hip_vs_truth_epe_off_ties_px"] <= 1.3 * ref_mean, rep
    assert rep["epe_vs_reference_off_ties_px"] <= 3e-5 and rep["epe_vs_reference_fullres_px"] <= 1e-3, rep


@pytest.mark.parametrize("name", sorted(cases.SEGMENT_FULL))
def test_hot_segment_full_size_strict_on_the_reference_picks(sa, golden, name):
    """Round 3: the full sizes held to the bound on EVERY pixel, with no receptive-field excuse.  The HIP
    attention branch runs; a pixel may select other candidates only where the reference's own margin is below
    cases.DELTA24_REL, and there the reference's own candidates and attention weights (fixture: `risk24_samples`,
    `risk24_att_topk`) are put back; the HIP matching branch then runs on a candidate map identical to the reference's, so
    every pixel of `pred` must be within the bound of the reference's unless the reference's own 2nd / 3rd largest costs
    are within cases.DELTA2 (tests/strict.py).  The bounds come from the fixture's float64 truth: with r = the largest
    distance of the REFERENCE's own fp32 evaluation from the exact answer over the same pixels (make_golden.py: 3.3e-4 px at
    1024^2 / D4 = 64, 8.0e-4 px at 2048^2 / D4 = 96 -- two kept candidates up to 95 disparities apart), the HIP path must be
    within max(1e-3, 2 r) of the truth and within max(1e-3, 3 r) of the reference, never more than 3e-3.  Measured r03: the HIP
    path is 1.9x as far from the truth as the reference on the worst pixel and 3x on average (6.3e-4 / 3.6e-5 px at 1024^2,
    1.5e-3 / 9.9e-5 px at 2048^2): the reference's MKL-DNN convolutions accumulate in blocks, the matrix core in one chain
    of K = 864 x 3 products -- both far inside the 1e-3 px target at the size it is stated for."""
    if sa.modules.CONV_ENGINE == "bf16x3":
        pytest.skip("SS_CONV_ENGINE=bf16x3: these bounds are for the fp32-accurate engines")
    import json
    import os
    import strict
    if "segment_full" not in golden or strict.fixture_view(golden["segment_full"], name) is None:
        pytest.skip(f"{name}: no round-3 fixture")
    g = golden["segment_full"]
    B, H, W, maxdisp = cases.segment_shape(name)
    seg = sa.HotSegment(maxdisp)
    res = seg.load_state_dict(cases.segment_params(name, g), strict=False)
    assert not res.unexpected_keys and all(k.endswith("num_batches_tracked") for k in res.missing_keys)
    seg = seg.cuda().eval()
    before = dict(sa.modules.PATH_COUNTS)
    rep, v, pred, differs, unexplained = strict.run_strict(seg, g, name)
    assert sa.modules.PATH_COUNTS["torch"] == before["torch"], "a PyTorch fallback ran"
    ref_self = rep["reference_vs_truth_max_off_ties_px"]     # the REFERENCE's own fp32 evaluation against the float64 truth, worst pixel
    ref_mean = rep["reference_vs_truth_epe_off_ties_px"]     # ... and on average
    default_engine = sa.modules.CONV_ENGINE == "f16x3"
    # Round 4: with the warp kernels following the reference's coordinate arithmetic operation by operation and
    # the chunk-blocked accumulation of the small-tile convolutions, the default engine is as close to the float64 answer as the
    # reference's own CPU arithmetic (measured: 1.015x / 1.017x its mean distance at 1024^2 / 2048^2, 0.98x / 1.03x on the worst
    # pixel; r03: 3x / 1.9x).  Bounds: worst pixel <= 1.25x the reference's, mean <= 1.3x, and against the REFERENCE itself every
    # pixel off its own cost ties within 5e-4 px at both sizes (r03: max(1e-3, 3 r) = 2.4e-3 at 2048^2; measured 7.6e-5 / 1.6e-4).
    # The other engines (single accumulation chains everywhere) keep r03's bounds.
    # Round 5: concat_stem gathers the warped half inside its staging (ss_conv3d_gather_fwd) -- for the integer
    # candidates of the live call that IS the exact value of the bilinear sample, where the reference's F.grid_sample (and r04's
    # operation-by-operation restatement of it) carries the coordinate round trip's error on a quarter of the columns and rows.
    # Measured: the HIP path is now 0.42x / 0.29x the reference's own mean distance from the float64 truth at 1024^2 / 2048^2 and
    # 0.21x / 0.19x on the worst pixel; the distance between the two fp32 evaluations is then the REFERENCE's own error
    # (3.4e-4 / 8.0e-4 px worst pixel against its 3.3e-4 / 8.0e-4 from the truth).  Bounds with the gathered stem: worst pixel and
    # mean vs truth <= 0.6x the reference's, every pixel off the reference's cost ties within min(1e-3, 1.25 r) px of it.
    gathered = default_engine and sa.engine.STEM_GATHER
    bound_truth = (0.6 if gathered else 1.25) * ref_self if default_engine else max(1e-3, 2.0 * ref_self)
    bound = (min(1e-3, 1.25 * ref_self) if gathered else 5e-4) if default_engine else max(1e-3, 3.0 * ref_self)
    rep["bound_px"], rep["bound_vs_truth_px"] = bound, bound_truth
    rep["mean_ratio_to_reference"] = rep["hip_vs_truth_epe_off_ties_px"] / ref_mean
    rep["max_ratio_to_reference"] = rep["hip_vs_truth_max_off_ties_px"] / ref_self
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/fullsize_strict_{name}.json", "w") as f:
        json.dump(rep, f, indent=1)
    assert not bool(unexplained.any()), (f"{int(unexplained.sum())} pixel(s) select other candidates where the reference's margin "
                                         f"is >= {DELTA24_REL}")
    if "_cal" not in name:
        # r06: DEFAULT (uncalibrated) BatchNorm statistics, the state of random-init weights.  The reference's own
        # evaluation has 8 404 of 65 536 pixels within 1e-4 (relative) of a top-24 tie -- exact ties among them -- and 4 434 within 1e-4 of a
        # top-2 tie; the candidates' costs spread over ~0.05 instead of O(1).  Held here: no unexplained candidate difference (above), every
        # pixel off the reference's own cost ties within the north star's 1e-3 px of the reference, the HIP path no further from the
        # float64 truth than the reference's own arithmetic, and the full-resolution EPE off ties far inside 1e-3.  The whole-map EPE
        # INCLUDING the reference's tie pixels is reported (`epe_vs_reference_fullres_px`), not bounded: a top-2 flip at a tie moves a
        # pixel by whole candidates in any fp32 evaluation, the reference's included.
        assert rep["max_err_off_ties_px"] <= 1e-3, rep
        assert rep["hip_vs_truth_max_off_ties_px"] <= max(1.25 * ref_self, 2e-4), rep
        assert rep["hip_vs_truth_epe_off_ties_px"] <= 1.3 * ref_mean, rep
        assert 4.0 * rep["epe_vs_reference_off_ties_px"] <= 1e-3 and rep["median_abs_err_px"] <= 1e-4, rep
        return
    assert bound <= 3e-3, rep
    assert rep["max_err_off_ties_px"] <= bound, rep                         # EVERY pixel away from the reference's own cost ties
    assert rep["hip_vs_truth_max_off_ties_px"] <= bound_truth, rep
    assert rep["hip_vs_truth_epe_off_ties_px"] <= ((0.6 if gathered else 1.3) * ref_mean if default_engine else max(1e-4, 4.0 * ref_mean)), rep
    assert rep["epe_vs_reference_off_ties_px"] <= (1.5 * ref_mean if default_engine else max(1e-4, 5.0 * ref_mean)) and rep["median_abs_err_px"] <= 1e-4, rep
    if 2 * (maxdisp // 4) <= 64 or default_engine:
        # whole map INCLUDING the reference's own cost ties (where a top-2 flip moves a pixel by whole candidates), full-resolution
        # EPE (x4) < 1e-3: the north star's figure at the size it is stated for -- and, since r04, at 2048^2 / 192 too
        assert rep["epe_vs_reference_fullres_px"] <= 1e-3, rep
