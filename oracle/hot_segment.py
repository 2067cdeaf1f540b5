"""Oracle restatement of the hot segment of SemStereo.forward
(models/SemStereo.py:273-323): from the 1/8- and 1/4-scale feature maps (after
the chal_* projections) to the 1/4-scale disparities `pred_att` and `pred`.
TEST INFRASTRUCTURE ONLY.

`P` is a flat dict with the reference's state_dict keys for the modules the
segment owns: patch, corr_feature_att_8, hourglass_att, classif_att_, gamma,
beta, concat_feature, concat_stem, concat_feature_att_4, hourglass, classif.
"""
import torch
import torch.nn.functional as F

from . import ops, stack

TOPK = 24  # models/SemStereo.py:301


GWC_CLOSED_FORM = False      # True: ops.build_gwc_volume_norm_closed_form (bit-identical, 200x faster on the CPU) at :273


def attention_branch(P, fl8, fr8, fl4, fr4, maxdisp, out=None, unsigned=False, force_samples=None):
    """models/SemStereo.py:273-310.  Returns (att_topk [B,1,k,H4,W4],
    disparity_sample_topk [B,k,H4,W4], pred_att [B,H4,W4]).
    unsigned: models/SemStereo_WHU.py's two differing lines (:279 maxdisp//4 planes, :305 no offset) on the unsigned op
    set it needs (models/submodule_.py, oracle/ops_unsigned.py)."""
    B, C8, H8, W8 = fl8.shape
    H4, W4 = fl4.shape[2], fl4.shape[3]
    m4 = maxdisp // 4
    if unsigned:
        from . import ops_unsigned as uops
        corr = uops.build_gwc_volume_norm(fl8, fr8, maxdisp // 8, C8 // 8)
        corr = stack.patch_conv(P, corr)
        cost_att = stack.channel_att(P, "corr_feature_att_8", corr, fl8)
        cost_att = stack.hourglass(P, "hourglass_att", cost_att, (4, 4, 4))
        cost_att = stack.classifier(P, "classif_att_", cost_att)
        att_weights = F.interpolate(cost_att, [m4, H4, W4], mode="trilinear")              # SemStereo_WHU.py:279
        prob0 = F.softmax(att_weights.squeeze(1), dim=1)
        pred0 = uops.disparity_regression(prob0, m4)
        var = uops.disparity_variance(prob0, m4, pred0.unsqueeze(1))
        return _attention_tail(P, fl4, fr4, att_weights, pred0, var, 0, out, corr, cost_att, force_samples)
    gwc = ops.build_gwc_volume_norm_closed_form if GWC_CLOSED_FORM else ops.build_gwc_volume_norm
    corr = gwc(fl8, fr8, maxdisp // 8, C8 // 8)                                        # :273
    corr = stack.patch_conv(P, corr)                                                   # :274
    cost_att = stack.channel_att(P, "corr_feature_att_8", corr, fl8)                   # :276
    cost_att = stack.hourglass(P, "hourglass_att", cost_att, (4, 4, 4))                # :277
    cost_att = stack.classifier(P, "classif_att_", cost_att)                           # :278
    att_weights = F.interpolate(cost_att, [m4 * 2, H4, W4], mode="trilinear")          # :279
    prob0 = F.softmax(att_weights.squeeze(1), dim=1)                                   # :281-282
    pred0 = ops.disparity_regression(prob0, m4)                                        # :283
    var = ops.disparity_variance(prob0, m4, pred0.unsqueeze(1))                        # :285
    return _attention_tail(P, fl4, fr4, att_weights, pred0, var, -m4, out, corr, cost_att, force_samples)


def _attention_tail(P, fl4, fr4, att_weights, pred0, var, dmin, out, corr, cost_att, force_samples=None):
    """models/SemStereo.py:286-310; `dmin`: the disparity of plane 0 (-maxdisp//4, or 0 in SemStereo_WHU.py:305).
    force_samples [B,k,H4,W4]: take THESE candidates instead of the graph's own top-24 (gradient checks of another
    evaluation whose picks differ at near-ties: everything downstream of the pick is then compared like for like)."""
    var = torch.sigmoid(P["beta"] + P["gamma"] * var)                                  # :286-287
    var_samples = ops.propagation(var)                                                 # :288
    disp_samples = ops.propagation(pred0.unsqueeze(1))                                 # :289
    right_w, left_b = ops.SpatialTransformer_grid(fl4, fr4, disp_samples)              # :291
    strength = (left_b * right_w).mean(dim=1)                                          # :292
    strength = torch.softmax(strength * var_samples, dim=1)                            # :293
    aw = ops.propagation_prob(att_weights)                                             # :295
    aw = (aw * strength.unsqueeze(2)).sum(dim=1, keepdim=True)                         # :296-297
    aw_prob = F.softmax(aw, dim=2)                                                     # :298
    _, ind = aw_prob.sort(2, True)                                                     # :299
    ind_k = ind[:, :, :TOPK].sort(2, False)[0]                                         # :302-303
    if out is not None:
        srt = aw_prob.sort(2, True)[0]
        out.update(gap24_rel=((srt[:, :, TOPK - 1] - srt[:, :, TOPK]) / srt[:, :, TOPK - 1]).squeeze(1))
    if force_samples is not None:
        ind_k = (force_samples - dmin).round().long().unsqueeze(1)
    att_topk = torch.gather(aw_prob, 2, ind_k)                                         # :304
    samples = ind_k.squeeze(1).float() + dmin                                          # :305
    att_prob = F.softmax(torch.gather(aw, 2, ind_k).squeeze(1), dim=1)                 # :307-308
    pred_att = (att_prob * samples).sum(dim=1)                                         # :309-310
    if out is not None:
        out.update(corr_volume=corr, cost_att=cost_att, att_weights=aw, pred_att0=pred0, strength=strength,
                   att_topk_full=att_topk)
    return att_topk, samples, pred_att


def matching_branch(P, fl4, fr4, att_topk, samples, out=None):
    """models/SemStereo.py:314-323.  Returns pred [B,1,H4,W4]."""
    cl = stack.concat_feature(P, fl4)                                                  # :314
    cr = stack.concat_feature(P, fr4)                                                  # :315
    right_w, left_b = ops.SpatialTransformer_grid(cl, cr, samples)                     # :241-242
    volume = att_topk * torch.cat((left_b, right_w), dim=1)                            # :243, :318
    volume = stack.basic_conv(P, "concat_stem", volume, is_3d=True)                    # :319
    volume = stack.channel_att(P, "concat_feature_att_4", volume, fl4)                 # :320
    if out is not None:
        out.update(concat_l=cl, stem=volume)
    cost = stack.hourglass(P, "hourglass", volume, (6, 4, 4))                          # :321
    if out is not None:
        out.update(hourglass=cost)
    cost = stack.classifier(P, "classif", cost)                                        # :322
    pred = ops.regression_topk(cost.squeeze(1), samples, 2)                            # :323
    if out is not None:
        out.update(cost=cost)
    return pred


@torch.no_grad()
def matching_truth_tiled(P, fl4, fr4, att_topk, samples, tile=128, halo=48):
    """models/SemStereo.py:314-323 (the lines of `matching_branch` above) evaluated in FLOAT64 over the whole
    quarter-resolution map, given 24 candidates and attention weights per pixel -> pred [B,1,H4,W4] float64: the exact
    answer of the reference graph for these inputs ("truth"), which any fp32 evaluation -- the reference's own
    included -- can be held against.  The 2-D features and the warp are evaluated on the whole image; the 3-D stack runs
    tile by tile (float64 convolutions go through im2col: 87 GB for the whole map at 2048^2).  A tile of `tile` pixels plus
    `halo` on every side that is not an image edge is exact on the tile: the receptive field of concat_stem + hourglass2
    + classif is <= 36 quarter-resolution pixels and the attention windows of hourglass2 (16 such pixels) stay whole
    because every cut is a multiple of 16 (tests/test_oracle_golden.py::test_truth_tiles_equal_the_whole_map)."""
    assert tile % 16 == 0 and halo % 16 == 0 and halo >= 48
    P64 = {k: (v.double() if v.is_floating_point() else v) for k, v in P.items()}
    fl4d, smp, att = fl4.double(), samples.double(), att_topk.double()
    cl = stack.concat_feature(P64, fl4d)
    cr = stack.concat_feature(P64, fr4.double())
    right_w, left_b = ops.SpatialTransformer_grid(cl, cr, smp)
    B, _, H4, W4 = fl4.shape
    pred = torch.empty(B, 1, H4, W4, dtype=torch.float64)
    for ty in range(0, H4, tile):
        for tx in range(0, W4, tile):
            y0, y1 = max(ty - halo, 0), min(ty + tile + halo, H4)
            x0, x1 = max(tx - halo, 0), min(tx + tile + halo, W4)
            sl = (Ellipsis, slice(y0, y1), slice(x0, x1))
            vol = att[sl] * torch.cat((left_b[sl], right_w[sl]), dim=1)
            vol = stack.basic_conv(P64, "concat_stem", vol, is_3d=True)
            vol = stack.channel_att(P64, "concat_feature_att_4", vol, fl4d[sl])
            cost = stack.classifier(P64, "classif", stack.hourglass(P64, "hourglass", vol, (6, 4, 4)))
            p = ops.regression_topk(cost.squeeze(1), smp[sl], 2)
            ey, ex = min(ty + tile, H4), min(tx + tile, W4)
            pred[..., ty:ey, tx:ex] = p[..., ty - y0:ey - y0, tx - x0:ex - x0]
    return pred


@torch.no_grad()
def hot_segment(P, fl4, fr4, fl8, fr8, maxdisp, keep=False, unsigned=False):
    """features_left[1], features_right[1] ([B,128,H/4,W/4]) and
    features_left[2], features_right[2] ([B,256,H/8,W/8]) -> dict with
    `pred_att` [B,H4,W4], `pred` [B,1,H4,W4], `samples`, `att_topk` (+ the
    intermediates when keep=True)."""
    out = {} if keep else None
    att_topk, samples, pred_att = attention_branch(P, fl8, fr8, fl4, fr4, maxdisp, out, unsigned)
    pred = matching_branch(P, fl4, fr4, att_topk, samples, out)
    res = dict(pred_att=pred_att, pred=pred, samples=samples, att_topk=att_topk)
    if keep:
        res.update(out)
    return res


# ---------------------------------------------------------------------------
# deterministic parameters (closed-form fill per state_dict key; no RNG)
# ---------------------------------------------------------------------------

def segment_param_shapes(c8=256, c4=128):
    """Shapes of every parameter/buffer the segment owns, keyed as in the
    reference's state_dict (dumped from the reference)."""
    g = c8 // 8          # 32 groups = volume channels
    cc = c4 // 4         # 32 concat channels
    S = {}

    def bn(key, c):
        S[key + ".weight"] = (c,); S[key + ".bias"] = (c,)
        S[key + ".running_mean"] = (c,); S[key + ".running_var"] = (c,)

    def convbn(key, co, ci, k):
        S[key + ".0.weight"] = (co, ci, k, k, k); bn(key + ".1", co)

    def hg(key, c):
        convbn(key + ".conv1.0", 2 * c, c, 3); convbn(key + ".conv2.0", 2 * c, 2 * c, 3)
        convbn(key + ".conv3.0", 4 * c, 2 * c, 3); convbn(key + ".conv4.0", 4 * c, 4 * c, 3)
        S[key + ".attention_block.qkv_3d.weight"] = (12 * c, 4 * c)
        S[key + ".attention_block.qkv_3d.bias"] = (12 * c,)
        S[key + ".attention_block.final1x1.weight"] = (4 * c, 4 * c, 1, 1, 1)
        S[key + ".attention_block.final1x1.bias"] = (4 * c,)
        S[key + ".conv5.0.weight"] = (4 * c, 2 * c, 3, 3, 3); bn(key + ".conv5.1", 2 * c)
        S[key + ".conv6.0.weight"] = (2 * c, c, 3, 3, 3); bn(key + ".conv6.1", c)
        convbn(key + ".redir1", c, c, 1); convbn(key + ".redir2", 2 * c, 2 * c, 1)

    def catt(key, cv, im):
        S[key + ".im_att.0.conv.weight"] = (im // 2, im, 1, 1); bn(key + ".im_att.0.bn", im // 2)
        S[key + ".im_att.1.weight"] = (cv, im // 2, 1, 1); S[key + ".im_att.1.bias"] = (cv,)

    S["gamma"] = (1,); S["beta"] = (1,)
    S["patch.weight"] = (g, 1, 1, 3, 3)
    S["concat_feature.0.conv.weight"] = (c4 // 2, c4, 3, 3); bn("concat_feature.0.bn", c4 // 2)
    S["concat_feature.1.weight"] = (cc, c4 // 2, 3, 3)
    catt("corr_feature_att_8", cc, c8); catt("concat_feature_att_4", cc, c4)
    hg("hourglass_att", g); hg("hourglass", cc)
    for key in ("classif_att_", "classif"):
        convbn(key + ".0", 32, 32, 3); S[key + ".2.weight"] = (1, 32, 3, 3, 3)
    S["concat_stem.conv.weight"] = (cc, 2 * cc, 3, 3, 3); bn("concat_stem.bn", cc)
    return S


def deterministic_params(c8=256, c4=128, salt=7):
    """Closed-form fill of every key (sorted-key order, hash-based values):
    conv/linear weights ~ U(-a, a) with a = sqrt(3 / fan_in) (unit gain), BN
    weight/var ~ U(0.6, 1.4), BN bias/mean and conv bias ~ U(-0.1, 0.1),
    gamma = 0.25, beta = 2 (the reference initialises gamma=0, beta=2;
    0.25 keeps the variance path live)."""
    from . import detdata
    P = {}
    for i, (key, shape) in enumerate(sorted(segment_param_shapes(c8, c4).items())):
        s = salt * 1000 + i
        if key == "gamma":
            v = torch.full(shape, 0.25)
        elif key == "beta":
            v = torch.full(shape, 2.0)
        elif key.endswith("running_var") or (key.endswith(".weight") and len(shape) == 1):
            v = detdata.t_uniform(shape, s, 0.6, 1.4)
        elif len(shape) == 1:
            v = detdata.t_uniform(shape, s, -0.1, 0.1)
        else:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            if ".conv5.0." in key or ".conv6.0." in key:      # ConvTranspose3d: [Cin, Cout, k,k,k]
                fan_in = shape[0] * 27 // 8                  # ~27/8 taps reach each output
            a = (3.0 / fan_in) ** 0.5
            v = detdata.t_uniform(shape, s, -a, a)
        P[key] = v.float()
    return P
