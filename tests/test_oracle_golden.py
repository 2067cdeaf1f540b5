"""CPU: the oracle (oracle/*.py) against the golden fixtures produced by the
reference itself (tests/golden/make_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest
import torch

from golden import cases
from oracle import hot_segment as oseg
from oracle import ops as oops
from oracle import stack as ostack


def _eq(a, ref, atol=0.0, rtol=0.0):
    a = a.detach().cpu().numpy()
    assert a.shape == ref.shape, (a.shape, ref.shape)
    if atol == 0.0 and rtol == 0.0:
        assert np.array_equal(a, ref, equal_nan=True), float(np.nanmax(np.abs(a - ref)))
    else:
        np.testing.assert_allclose(a, ref, atol=atol, rtol=rtol)


@pytest.mark.parametrize("name", sorted(cases.GWC))
def test_gwc(golden, name):
    a, b, m, G = cases.gwc_inputs(name)
    g = golden["ops"]
    _eq(oops.build_gwc_volume(a, b, m, G), g[f"gwc/{name}"])
    _eq(oops.build_gwc_volume_norm(a, b, m, G), g[f"gwc_norm/{name}"])
    _eq(oops.groupwise_correlation(a, b, G), g[f"gcorr/{name}"])
    _eq(oops.groupwise_correlation_norm(a, b, G), g[f"gcorr_norm/{name}"])


@pytest.mark.parametrize("name", sorted(cases.GWC))
def test_closed_form_gwc_norm_is_the_slice_loop_bit_for_bit(golden, name):
    """oracle.ops.build_gwc_volume_norm_closed_form (both maps normalised once; used by bench.py's many-pair parity leg) against
    the reference's fixture AND the slice-loop restatement: the same bits."""
    a, b, m, G = cases.gwc_inputs(name)
    v = oops.build_gwc_volume_norm_closed_form(a, b, m, G)
    assert torch.equal(v, oops.build_gwc_volume_norm(a, b, m, G))
    np.testing.assert_array_equal(v.numpy(), golden["ops"][f"gwc_norm/{name}"])


@pytest.mark.parametrize("name", sorted(cases.CONCAT))
def test_concat(golden, name):
    a, b, m = cases.concat_inputs(name)
    _eq(oops.build_concat_volume(a, b, m), golden["ops"][f"concat/{name}"])


@pytest.mark.parametrize("name", sorted(cases.REGRESSION))
def test_regression(golden, name):
    p, m, d = cases.regression_inputs(name)
    _eq(oops.disparity_regression(p, m), golden["ops"][f"regression/{name}"])
    _eq(oops.disparity_variance(p, m, d), golden["ops"][f"variance/{name}"])
    with pytest.raises(AssertionError):
        oops.disparity_regression(p.unsqueeze(0), m)


@pytest.mark.parametrize("name", sorted(cases.WARP))
def test_warp(golden, name):
    x, y, d = cases.warp_inputs(name)
    yw, xw = oops.SpatialTransformer_grid(x, y, d)
    _eq(yw, golden["ops"][f"warp_y/{name}"])
    _eq(xw, golden["ops"][f"warp_x/{name}"])


@pytest.mark.parametrize("name", sorted(cases.TOPK))
def test_topk(golden, name):
    c, s, k = cases.topk_inputs(name)
    _eq(oops.regression_topk(c, s, k), golden["ops"][f"topk/{name}"])


def test_propagation(golden):
    for n in cases.PROP:
        _eq(oops.propagation(cases.prop_inputs(n)), golden["ops"][f"prop/{n}"])
    for n in cases.PROP_PROB:
        _eq(oops.propagation_prob(cases.prop_prob_inputs(n)), golden["ops"][f"prop_prob/{n}"])


def run_stack_oracle(P, name):
    kind, shape, block = cases.STACK[name]
    x = cases.stack_input(name)
    if kind in ("hourglass_att", "hourglass"):
        return ostack.hourglass(P, kind, x, block)
    if kind == "classif":
        return ostack.classifier(P, "classif", x)
    if kind == "concat_stem":
        return ostack.basic_conv(P, "concat_stem", x, is_3d=True)
    return ostack.attention_block(P, kind, x, block)


@pytest.mark.parametrize("name", sorted(cases.STACK))
def test_stack(golden, name):
    P = oseg.deterministic_params()
    with torch.no_grad():
        y = run_stack_oracle(P, name)
    # same ATen kernels in a different composition: allow a few ulp
    _eq(y, golden["stack"][f"stack/{name}"], atol=2e-6, rtol=1e-5)


@pytest.mark.parametrize("name", sorted(cases.SEGMENT) + sorted(cases.SEGMENT_CAL))
def test_hot_segment(golden, name):
    g = golden["segment"]
    P = cases.segment_params(name, g)            # "_cal": BatchNorm statistics calibrated by the reference run (fixture)
    fl4, fr4, fl8, fr8, maxdisp = cases.segment_inputs(name)
    r = oseg.hot_segment(P, fl4, fr4, fl8, fr8, maxdisp, keep=True)
    assert np.array_equal(r["samples"].numpy().astype(np.int16), g[f"{name}/samples"])
    # the reference's margins at its two hard picks, restated from the oracle's intermediates (what the GPU tests'
    # explained-deviation criterion rests on)
    p = torch.softmax(r["att_weights"], dim=2).squeeze(1).sort(dim=1, descending=True).values
    _eq((p[:, 23] - p[:, 24]) / p[:, 23], g[f"{name}/gap24_rel"], atol=2e-6)
    c = r["cost"].squeeze(1).sort(dim=1, descending=True).values
    _eq(c[:, 1] - c[:, 2], g[f"{name}/gap2"], atol=2e-5)
    _eq(r["att_topk"].squeeze(1), g[f"{name}/att_topk"], atol=1e-7, rtol=1e-5)
    _eq(r["pred_att0"], g[f"{name}/pred_att0"], atol=1e-5)
    _eq(r["pred_att"], g[f"{name}/pred_att"], atol=1e-5)
    _eq(r["pred"], g[f"{name}/pred"], atol=1e-5)
    _eq(r["cost_att"], g[f"{name}/cost_att"], atol=1e-5, rtol=1e-5)
    for i, (key, t) in enumerate((("build_gwc_volume_norm", None), ("patch", r["corr_volume"]))):
        if t is None:
            continue
        rec = g[f"{name}/sum/{key}"]
        a = t.double().reshape(-1)
        idx = cases.sample_index(a.numel(), 64, i)
        np.testing.assert_allclose(a[idx].numpy(), rec[2:], atol=1e-6)
        np.testing.assert_allclose(a.sum().item(), rec[0], rtol=1e-6, atol=1e-3)


def test_param_table_matches_reference_names():
    """The key table is validated against the reference in make_golden.py
    (load_state_dict: no unexpected keys); here: it is self-consistent."""
    S = oseg.segment_param_shapes()
    P = oseg.deterministic_params()
    assert set(S) == set(P)
    assert P["hourglass_att.conv5.0.weight"].shape == (128, 64, 3, 3, 3)
    assert P["hourglass.attention_block.qkv_3d.weight"].shape == (384, 128)
    assert P["concat_stem.conv.weight"].shape == (32, 64, 3, 3, 3)
    assert P["patch.weight"].shape == (32, 1, 1, 3, 3)
    assert all(v.dtype == torch.float32 for v in P.values())


@pytest.mark.parametrize("name", ["f1024_md128_cal", "f1024_md128"])
def test_hot_segment_full_size_checksums(golden, name):
    """BASELINE.json configs[1]'s size (1024 x 1024, maxdisp 128): the oracle against the reference's checksum record
    (per-stage sums and sampled voxels; pred, pred_att, candidates and margins at 2048 sampled pixels).  ~25 s of CPU each:
    with calibrated BatchNorm statistics and (r06) with the default ones of random-init weights."""
    g = golden["segment_full"]
    B, H, W, maxdisp = cases.segment_shape(name)
    P = cases.segment_params(name, g)
    fl4, fr4, fl8, fr8, _ = cases.segment_inputs(name)
    r = oseg.hot_segment(P, fl4, fr4, fl8, fr8, maxdisp, keep=True)
    H4, W4 = H // 4, W // 4
    idx = torch.as_tensor(g[f"{name}/pixels"])
    flat = lambda t: t.reshape(B, -1, H4 * W4).permute(0, 2, 1).reshape(B * H4 * W4, -1)[idx]       # noqa: E731
    same = (flat(r["samples"]).numpy().astype(np.int16) == g[f"{name}/samples"]).all(axis=1)
    gap24, gap2 = g[f"{name}/gap24_rel"], g[f"{name}/gap2"]
    assert not (~same & (gap24 >= 1e-5)).any()                   # same ATen kernels: only exact-tie pixels could differ
    err = np.abs(flat(r["pred"])[:, 0].numpy() - g[f"{name}/pred"])
    assert not ((err > 1e-3) & (gap2 >= 1e-5) & same).any() and np.median(err) <= 1e-6
    _eq(flat(r["pred_att"].unsqueeze(1))[:, 0][torch.as_tensor(same)], g[f"{name}/pred_att"][same], atol=1e-4)
    for i, (key, t) in enumerate((("build_gwc_volume_norm", None), ("patch", r["corr_volume"]), ("hourglass_att", None),
                                  ("classif_att_", r["cost_att"]))):
        if t is None:
            continue
        rec = g[f"{name}/sum/{key}"]
        a = t.double().reshape(-1)
        np.testing.assert_allclose(a[cases.sample_index(a.numel(), 64, i)].numpy(), rec[2:], atol=2e-5)
        np.testing.assert_allclose((a * a).sum().item(), rec[1], rtol=1e-5)


# ---- the unsigned-range op set (models/submodule_.py) and the SemStereo_WHU graph ----------------------------------

def test_unsigned_op_set(golden):
    from oracle import ops_unsigned as uops
    g = golden["ops_unsigned"]
    for n in cases.UGWC:
        a, b, m, G = cases.ugwc_inputs(n)
        if f"ugwc/{n}" in g.files:
            _eq(uops.build_gwc_volume(a, b, m, G), g[f"ugwc/{n}"])
            _eq(uops.build_gwc_volume_norm(a, b, m, G), g[f"ugwc_norm/{n}"])
            # the unsigned volume is the non-negative half of the signed one (plane d of [0, m) = plane m + d of [-m, m))
            _eq(oops.build_gwc_volume(a, b, m, G)[:, :, m:], g[f"ugwc/{n}"])
    for n in cases.UCONCAT:
        a, b, m = cases.uconcat_inputs(n)
        if f"uconcat/{n}" in g.files:
            _eq(uops.build_concat_volume(a, b, m), g[f"uconcat/{n}"])
    for n in cases.UREGRESSION:
        p, m, d = cases.uregression_inputs(n)
        _eq(uops.disparity_regression(p, m), g[f"uregression/{n}"])
        _eq(uops.disparity_variance(p, m, d), g[f"uvariance/{n}"])


@pytest.mark.parametrize("name", sorted(cases.SEGMENT_WHU))
def test_hot_segment_whu(golden, name):
    """models/SemStereo_WHU.py (two lines off models/SemStereo.py: :279, :305) on the unsigned op set."""
    g = golden["segment_whu"]
    P = oseg.deterministic_params()
    fl4, fr4, fl8, fr8, maxdisp = cases.segment_inputs(name)
    r = oseg.hot_segment(P, fl4, fr4, fl8, fr8, maxdisp, keep=True, unsigned=True)
    assert np.array_equal(r["samples"].numpy().astype(np.int16), g[f"{name}/samples"])
    assert int(r["samples"].min()) >= 0 and int(r["samples"].max()) < maxdisp // 4
    _eq(r["pred_att0"], g[f"{name}/pred_att0"], atol=1e-5)
    _eq(r["pred_att"], g[f"{name}/pred_att"], atol=1e-5)
    _eq(r["pred"], g[f"{name}/pred"], atol=1e-5)
    _eq(r["cost_att"], g[f"{name}/cost_att"], atol=1e-5, rtol=1e-5)


# ---- round 3: strict parity machinery (tests/strict.py) and the float64 truth of the fixtures ----------------------

def test_truth_tiles_equal_the_whole_map():
    """oracle.hot_segment.matching_truth_tiled (the float64 `pred_truth` of the fixtures) evaluates the 3-D stack tile by
    tile: with a 48-pixel halo and cuts at multiples of 16 it must equal the whole-map float64 evaluation (receptive field
    of concat_stem + hourglass2 + classif <= 36 quarter-resolution pixels, attention windows of 16 stay whole)."""
    from oracle import detdata as dd
    H4, W4, maxdisp = 96, 112, 64
    P = oseg.deterministic_params()
    fl4, fr4 = dd.stereo_features(1, 128, H4, W4, 12, 6)
    smp = dd.distinct_sorted_candidates(1, 24, H4, W4, maxdisp // 4, 13)
    att = torch.softmax(dd.t_normalish((1, 1, 24, H4, W4), 14), dim=2)
    P64 = {k: (v.double() if v.is_floating_point() else v) for k, v in P.items()}
    with torch.no_grad():
        whole = oseg.matching_branch(P64, fl4.double(), fr4.double(), att.double(), smp.double())
    tiled = oseg.matching_truth_tiled(P, fl4, fr4, att, smp, tile=32, halo=48)
    assert tiled.dtype == torch.float64 and tiled.shape == whole.shape
    assert float((tiled - whole).abs().max()) <= 1e-12


@pytest.mark.parametrize("name", ["s128", "s256_md128_cal"])
def test_strict_machinery_on_the_oracle(golden, name):
    """tests/strict.py driven by the ORACLE instead of the HIP path (the same code the -m gpu tests and bench.py run):
    the fixture's float64 truth is reproduced by the oracle, the reference's picks are restored where a candidate list
    is made to differ, and the fp32 oracle passes the strict criterion on every pixel."""
    import strict
    g = golden["segment"]
    v = strict.fixture_view(g, name)
    assert v is not None, "segment.npz predates round 3: regenerate with tests/golden/make_golden.py segment"
    B, H, W, maxdisp = cases.segment_shape(name)
    P = cases.segment_params(name, g)
    fl4, fr4, fl8, fr8, _ = cases.segment_inputs(name)
    # the stored truth is the oracle's float64 matching branch on the reference's candidates
    att_ref = torch.as_tensor(g[f"{name}/att_topk"]).unsqueeze(1)
    smp_ref = torch.as_tensor(g[f"{name}/samples"].astype(np.float32))
    tru = oseg.matching_truth_tiled(P, fl4, fr4, att_ref, smp_ref)
    assert float((tru.reshape(-1) - v["truth"].double()).abs().max()) <= 1e-5

    class OracleSegment:                       # the two methods strict.run_strict calls on a HotSegment
        def attention_branch(self, fl4, fr4, fl8, fr8):
            att, smp, pred_att = oseg.attention_branch(P, fl8, fr8, fl4, fr4, maxdisp)
            smp = smp.clone()
            smp[0, 23, 0, 0] += 1 if float(smp[0, 23, 0, 0]) < maxdisp // 4 - 1 else -1        # one pixel made to differ
            return att, smp, pred_att, None

        def matching_branch(self, fl4, fr4, att, smp):
            return oseg.matching_branch(P, fl4, fr4, att, smp)

    rep, v, pred, differs, unexplained = strict.run_strict(OracleSegment(), g, name, device="cpu")
    assert int(differs.sum()) == 1 and bool(differs[0])                   # ... detected; restored from the fixture:
    assert int(unexplained.sum()) == (0 if float(v["risk_gap24"][0]) < cases.DELTA24_REL else 1)
    assert rep["max_err_off_ties_px"] <= 1e-3 and rep["median_abs_err_px"] <= 1e-6, rep
    assert rep["reference_vs_truth_epe_off_ties_px"] <= 1e-4, rep
