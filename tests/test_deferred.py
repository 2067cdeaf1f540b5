"""semstereo_amd.deferred on the CPU: the handle protocol, and the matching of the reference's statement sequences.

No GPU here, so the fused KERNELS are replaced by torch compositions of the same statements (`_cpu_fused_kernels`);
what is under test is everything around them: that ops and twins hand out handles, that `torch` / `F` calls and tensor
methods on handles are recorded, that each fused rule recognises its statement sequence in a forward() written in the
reference's order (tests/standin_model.py here; the REAL reference forward() in tests/test_reference_dropin.py), that
any deviation replays the recorded calls, and that the results equal the plain op-by-op execution."""
import pytest
import torch
import torch.nn.functional as F

import semstereo_amd as sa
from semstereo_amd import deferred as dfr
from oracle import ops as oops


def _cpu_fused_kernels(monkeypatch):
    """Handles on CPU tensors; twins compute on their PyTorch path; fused entry points as torch compositions."""
    M, ops = sa.modules, sa.ops
    monkeypatch.setattr(dfr, "on", lambda module, *ts: dfr.ENABLED and not getattr(dfr._TLS, "depth", 0))
    monkeypatch.setattr(M, "_inference", lambda module, *ts: False)
    monkeypatch.setattr(sa.engine, "CONV_ENGINE", "f32")                 # stem_of: volume kernel + concat_stem(volume, gate) form

    def gwc_patch_gate(fl, fr, maxdisp, groups, patch_weight, gate_logits=None, normalize=True, _range=None):
        assert _range == (-maxdisp, 2 * maxdisp)
        vol = oops.build_gwc_volume_norm(fl, fr, maxdisp, groups)
        vol = F.conv3d(vol, patch_weight, None, 1, (0, 1, 1), 1, patch_weight.shape[0])
        return vol if gate_logits is None else torch.sigmoid(gate_logits).unsqueeze(2) * vol

    def upsample_softmax_regression(coarse, maxdisp, H, W, _range=None):
        dmin, nd = _range
        assert dmin == -nd // 2
        up = F.interpolate(coarse, [nd, H, W], mode="trilinear")
        prob = F.softmax(up.squeeze(1), dim=1)
        pred0 = oops.disparity_regression(prob, nd // 2)
        return up, pred0, oops.disparity_variance(prob, nd // 2, pred0.unsqueeze(1))

    def sample_strength(left, right, pred0, var, gamma, beta):
        v = torch.sigmoid(beta + gamma * var)
        rw, lb = oops.SpatialTransformer_grid(left, right, oops.propagation(pred0.unsqueeze(1)))
        return torch.softmax((lb * rw).mean(dim=1) * oops.propagation(v), dim=1)

    def topk_candidates(att_weights, strength, maxdisp, k, _range=None):
        dmin, nd = _range
        aw = (oops.propagation_prob(att_weights) * strength.unsqueeze(2)).sum(dim=1, keepdim=True)
        awp = F.softmax(aw, dim=2)
        ind_k = awp.sort(dim=2, descending=True, stable=True)[1][:, :, :k].sort(2, False)[0]
        samples = ind_k.squeeze(1).float() + dmin
        pred_att = (F.softmax(torch.gather(aw, 2, ind_k).squeeze(1), dim=1) * samples).sum(dim=1)
        return torch.gather(awp, 2, ind_k), samples, pred_att

    def concat_volume_sampled(left, right, samples, att=None):
        rw, lb = oops.SpatialTransformer_grid(right if left is None else left, right, samples)
        vol = rw if left is None else torch.cat((lb, rw), dim=1)
        return vol if att is None else att.reshape(att.shape[0], 1, *att.shape[-3:]) * vol

    for name, fn in (("gwc_patch_gate", gwc_patch_gate), ("upsample_softmax_regression", upsample_softmax_regression),
                     ("sample_strength", sample_strength), ("topk_candidates", topk_candidates),
                     ("concat_volume_sampled", concat_volume_sampled)):
        monkeypatch.setattr(ops, name, fn)
    # ops that are not deferred compute on the CPU through the oracle's definitions
    monkeypatch.setattr(ops._WarpSampled, "apply", staticmethod(lambda x, y, d: oops.SpatialTransformer_grid(x, y, d)))
    monkeypatch.setattr(ops._RegressionTopk, "apply", staticmethod(lambda c, s, k: oops.regression_topk(c, s, k)))
    monkeypatch.setattr(ops, "_gwc_forward", lambda a, b, rng, g, norm: (oops.build_gwc_volume_norm if norm else oops.build_gwc_volume)(a, b, -rng[0], g))
    monkeypatch.setattr(dfr, "STATS", {"fused": {}, "replayed": 0})


ALL_RULES = {"gwc_patch_gate", "upsample_softmax_regression", "sample_strength", "topk_candidates", "stem_by_halves"}


# ---- the handle protocol ------------------------------------------------------------------------------------------------

def test_handles_record_replay_and_force():
    t = torch.arange(24.0).reshape(2, 3, 4)
    h = dfr.Deferred.leaf(t)
    p = torch.nn.Parameter(torch.tensor([0.5]))
    e = torch.sigmoid(p + p * h)                                   # Parameter.__mul__ / __add__ with a handle on the right
    assert isinstance(e, dfr.Deferred) and e.op == "sigmoid" and not e.done
    assert torch.equal(e.value(), torch.sigmoid(p + p * t))
    vals, idx = F.softmax(h, dim=2).sort(2, True)                  # tuple-valued method -> two handles
    top = idx[:, :, :2]
    assert isinstance(top, dfr.Deferred) and top.op == "getitem"
    want = F.softmax(t, dim=2).sort(2, True)
    assert torch.equal(top.value(), want[1][:, :, :2]) and torch.equal(vals.value(), want[0])
    g = torch.gather(h, 2, top)                                    # function with two handles
    assert torch.equal(g.value(), torch.gather(t, 2, want[1][:, :, :2]))
    s = torch.sum(h * h, dim=1, keepdim=True).squeeze(1).float() - 3
    assert isinstance(s, dfr.Deferred) and torch.equal(s.value(), (t * t).sum(dim=1, keepdim=True).squeeze(1) - 3)
    c = torch.cat((h, h * 2), dim=1)
    assert isinstance(c, dfr.Deferred) and torch.equal(c.value(), torch.cat((t, t * 2), dim=1))
    # functions that are not recorded get the values, attributes are the tensor's
    assert torch.equal(torch.relu(h - 5), torch.relu(t - 5)) and isinstance(torch.relu(h), torch.Tensor)
    assert h.shape == t.shape and h.dim() == 3 and (h * 2).size()[2] == 4 and float((h * 2).max()) == 46.0
    assert torch.equal(torch.as_tensor(3.0) - h, 3.0 - t) and torch.equal(h / 2, t / 2)
    walked = dfr.real({"a": [h, (h * 2,)], "b": 1})                                             # containers are walked
    assert walked["b"] == 1 and torch.equal(walked["a"][0], t) and torch.equal(walked["a"][1][0], t * 2)


def test_suspended_and_disabled(monkeypatch):
    monkeypatch.setattr(dfr, "ENABLED", True)
    assert dfr.on(None, torch.zeros(2)) is False                    # CPU tensors: no handles
    with dfr.suspended():
        assert dfr.on(None) is False
    prop = sa.modules.Propagation()
    x = torch.randn(1, 1, 4, 5)
    assert isinstance(prop(x), torch.Tensor)                        # (CPU) the plain shifted views
    assert torch.equal(prop(dfr.Deferred.leaf(x)), sa.modules.propagation(x))


# ---- the statement sequences of the reference -------------------------------------------------------------------------------

def _standin(att_only=False):
    import standin_model
    from oracle import detdata as dd
    net = standin_model.StandInSemStereo(64, sa.modules, att_weights_only=att_only)
    with torch.no_grad():
        for i, (name, t) in enumerate(sorted(list(net.named_parameters()) + list(net.named_buffers()))):
            if name.endswith("num_batches_tracked") or name in ("gamma", "beta"):
                continue
            if name.endswith("running_var") or (name.endswith(".weight") and t.dim() == 1):
                t.copy_(dd.t_uniform(tuple(t.shape), 900 + i, 0.6, 1.4))
            elif t.dim() == 1:
                t.copy_(dd.t_uniform(tuple(t.shape), 900 + i, -0.1, 0.1))
            else:
                fan_in = t.shape[0] * 27 // 8 if (".conv5.0." in name or ".conv6.0." in name) else t[0].numel()
                t.copy_(dd.t_uniform(tuple(t.shape), 900 + i, -(3.0 / fan_in) ** 0.5, (3.0 / fan_in) ** 0.5))
    left = dd.t_normalish((1, 3, 128, 160), 951)
    right = torch.roll(left, shifts=-3, dims=3) + 0.05 * dd.t_normalish((1, 3, 128, 160), 952)
    return net.eval(), standin_model, left, right


@pytest.mark.parametrize("att_only", [False, True])
def test_every_fused_rule_fires_on_a_forward_in_the_reference_order(monkeypatch, att_only):
    net, module, left, right = _standin(att_only)
    _cpu_fused_kernels(monkeypatch)
    monkeypatch.setattr(dfr, "ENABLED", False)
    with torch.no_grad():
        (want,), lab = net(left, right)                              # oracle op library, PyTorch layers, no handles: plain execution
    monkeypatch.setattr(dfr, "ENABLED", True)
    previous = sa.install(module)
    try:
        with torch.no_grad():
            (got,), lab2 = net(left, right)
    finally:
        sa.uninstall(module, previous)
    fired = dfr.STATS["fused"]
    expect = ALL_RULES - ({"stem_by_halves"} if att_only else set())
    assert set(fired) == expect and all(v == 1 for v in fired.values()), fired
    assert isinstance(got, torch.Tensor) and torch.equal(lab, lab2)
    assert float((got - want).abs().max()) <= 1e-4, float((got - want).abs().max())


def test_deviations_from_the_reference_text_replay_the_recorded_calls(monkeypatch):
    """The same graph with statements the reference does not have (another soft-max dim order, an extra op, a different
    k): the rules must NOT fire where the text differs, and the result must still be the plain execution's."""
    _cpu_fused_kernels(monkeypatch)
    P = __import__("oracle.hot_segment", fromlist=["x"]).deterministic_params()
    seg = sa.HotSegment(64)
    seg.load_state_dict(P, strict=False)
    seg.eval()
    seg.FUSED = False
    from golden import cases
    fl4, fr4, fl8, fr8, maxdisp = cases.segment_inputs("s128")
    with torch.no_grad():
        r = seg(fl4, fr4, fl8, fr8)                                  # the reference-order composition: every rule fires
    assert set(dfr.STATS["fused"]) == ALL_RULES, dfr.STATS
    from oracle import hot_segment as oseg
    ref = oseg.hot_segment(P, fl4, fr4, fl8, fr8, maxdisp)
    assert torch.equal(r["samples"], ref["samples"])
    assert float((r["pred"] - ref["pred"]).abs().max()) <= 1e-4 and float((r["pred_att"] - ref["pred_att"]).abs().max()) <= 1e-4
    assert all(isinstance(v, torch.Tensor) for v in r.values())
    # -- deviation 1: the probe's soft-max over another dim -> no sample_strength, no topk (its input is no longer :293's)
    monkeypatch.setattr(dfr, "STATS", {"fused": {}, "replayed": 0})
    with torch.no_grad():
        att = seg.classif_att_(seg.hourglass_att(seg.corr_feature_att_8(seg.patch(sa.ops.build_gwc_volume_norm(fl8, fr8, 8, 32)), fl8)))
        up = F.interpolate(att, [32, 32, 32], mode="trilinear")
        prob = F.softmax(torch.squeeze(up, 1), dim=1)
        pred0 = sa.ops.disparity_regression(prob, 16)
        var = torch.sigmoid(seg.beta + seg.gamma * sa.ops.disparity_variance(prob, 16, pred0.unsqueeze(1)))
        rw, lb = sa.ops.SpatialTransformer_grid(fl4, fr4, seg.propagation(pred0.unsqueeze(1)))
        odd = torch.softmax((lb * rw).mean(dim=1) * seg.propagation(var), dim=2)          # dim=2: not the reference's text
        assert isinstance(odd, dfr.Deferred)
        val = odd.value()
    assert "sample_strength" not in dfr.STATS["fused"] and dfr.STATS["replayed"] > 0
    v_ = torch.sigmoid(P["beta"] + P["gamma"] * oops.disparity_variance(F.softmax(up.value().squeeze(1), 1), 16, pred0.unsqueeze(1)))
    rw_, lb_ = oops.SpatialTransformer_grid(fl4, fr4, oops.propagation(pred0.unsqueeze(1)))
    assert float((val - torch.softmax((lb_ * rw_).mean(dim=1) * oops.propagation(v_), dim=2)).abs().max()) <= 1e-5
    # -- deviation 2: k = 7 candidates (the kernel is built for 6 / 24 / 32) and an extra `* 1.0` before the stem
    monkeypatch.setattr(dfr, "STATS", {"fused": {}, "replayed": 0})
    with torch.no_grad():
        strength = torch.softmax((lb * rw).mean(dim=1) * seg.propagation(var), dim=1)
        aw = torch.sum(seg.propagation_prob(up) * strength.unsqueeze(2), dim=1, keepdim=True)
        awp = F.softmax(aw, dim=2)
        ind_k = awp.sort(2, True)[1][:, :, :7].sort(2, False)[0]
        att_topk, samples = torch.gather(awp, 2, ind_k), ind_k.squeeze(1).float() - 16
        cl, cr = seg.concat_feature(fl4), seg.concat_feature(fr4)
        rw2, lb2 = sa.ops.SpatialTransformer_grid(cl, cr, samples)
        vol = seg.concat_feature_att_4(seg.concat_stem(att_topk * torch.cat((lb2, rw2), dim=1) * 1.0), fl4)
    assert set(dfr.STATS["fused"]) == {"sample_strength"}, dfr.STATS         # only the probe matched
    assert isinstance(vol, torch.Tensor) and vol.shape == (1, 32, 7, 32, 32)
    s_ = dfr.real(strength)
    aw_ = (oops.propagation_prob(up.value()) * s_.unsqueeze(2)).sum(dim=1, keepdim=True)
    ik_ = F.softmax(aw_, 2).sort(2, True)[1][:, :, :7].sort(2, False)[0]
    assert torch.equal(dfr.real(samples), ik_.squeeze(1).float() - 16)
