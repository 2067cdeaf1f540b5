"""Drop-in replacements for the hot-path callables of the reference's op library
(/root/reference/models/submodule.py), same names and positional signatures,
backed by the gfx950 kernels of libsemstereo_hip.so.

Each public function documents the reference callable it replaces.  Inputs must
be fp32 tensors on an MI355X; CPU tensors raise (there is no fallback path).
Preconditions the reference asserts raise AssertionError here too.
"""
import torch
import torch.nn.functional as F

from . import _lib
from . import deferred as dfr
from ._lib import call, ptr


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def _needs_grad(*ts):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in ts)


# --------------------------------------------------------------------------------------
# group-wise correlation volume
# --------------------------------------------------------------------------------------

def signed_range(maxdisp):
    """(dmin, ndisp) of the reference's signed op set (models/submodule.py): disparities [-maxdisp, maxdisp)."""
    return -int(maxdisp), 2 * int(maxdisp)


def unsigned_range(maxdisp):
    """(dmin, ndisp) of the unsigned op set (models/submodule_.py, the one models/SemStereo_WHU.py is written for)."""
    return 0, int(maxdisp)


def _gwc_forward(ref, tgt, rng, groups, normalize):
    dev = _lib.require_device(ref, tgt)
    B, C, H, W = ref.shape
    dmin, nd = rng
    out = torch.empty((B, groups, nd, H, W), dtype=ref.dtype, device=ref.device)
    if out.numel():
        with torch.cuda.device(dev):
            call("ss_gwc_volume_fwd", ptr(ref), ptr(tgt), ptr(out), B, C, H, W, dmin, nd, groups, int(normalize))
    return out


class _GwcVolume(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ref, tgt, rng, groups):
        ref, tgt = _c(ref), _c(tgt)
        ctx.save_for_backward(ref, tgt)
        ctx.cfg = (rng, groups)
        return _gwc_forward(ref, tgt, rng, groups, False)

    @staticmethod
    def backward(ctx, g):
        ref, tgt = ctx.saved_tensors
        (dmin, nd), groups = ctx.cfg
        g = _c(g)
        B, C, H, W = ref.shape
        gref, gtgt = torch.empty_like(ref), torch.empty_like(tgt)
        with torch.cuda.device(ref.device):
            call("ss_gwc_volume_bwd", ptr(g), ptr(ref), ptr(tgt), ptr(gref), ptr(gtgt), B, C, H, W, dmin, nd, groups)
        return gref, gtgt, None, None


def _check_pair(a, b, groups=None):
    assert a.dim() == 4 and a.shape == b.shape, "feature maps must be [B,C,H,W] of equal shape"
    if groups is not None:
        assert a.shape[1] % groups == 0


class _GroupNormalise(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, groups):
        x = _c(x)
        B, C, H, W = x.shape
        y = torch.empty_like(x)
        with torch.cuda.device(x.device):
            call("ss_group_normalise_fwd", ptr(x), ptr(y), B, C, H, W, groups, 1e-05)
        ctx.save_for_backward(x)
        ctx.groups = groups
        return y

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        g = _c(g)
        B, C, H, W = x.shape
        gx = torch.empty_like(x)
        with torch.cuda.device(x.device):
            call("ss_group_normalise_bwd", ptr(g), ptr(x), ptr(gx), B, C, H, W, ctx.groups, 1e-05)
        return gx, None


def _group_normalise(x, groups):
    """x / (||x||_2 over each group's channels + 1e-5), differentiable (training path): one HIP launch each way for fp32 device maps
    (ss_group_normalise_fwd / _bwd), torch ops otherwise."""
    if x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and GROUP_NORMALISE_HIP:
        return _GroupNormalise.apply(x, int(groups))
    B, C, H, W = x.shape
    v = x.reshape(B, groups, C // groups, H, W)
    return (v / (torch.linalg.vector_norm(v, 2, dim=2, keepdim=True) + 1e-05)).reshape(B, C, H, W)


GROUP_NORMALISE_HIP = True


@dfr.realising
def _build_gwc_volume(refimg_fea, targetimg_fea, maxdisp, num_groups, _range=None):
    """models/submodule.py:198-211 -> [B, G, 2*maxdisp, H, W] (signed disparity range)."""
    _check_pair(refimg_fea, targetimg_fea, num_groups)
    rng = _range or signed_range(maxdisp)
    if _needs_grad(refimg_fea, targetimg_fea):
        return _GwcVolume.apply(refimg_fea, targetimg_fea, rng, int(num_groups))
    return _gwc_forward(_c(refimg_fea), _c(targetimg_fea), rng, int(num_groups), False)


def _build_gwc_volume_norm(refimg_fea, targetimg_fea, maxdisp, num_groups, _range=None, _defer=True):
    """models/submodule.py:224-238 (the live call, models/SemStereo.py:273).  In inference the result is a deferred handle
    (deferred.py): the caller's next two statements, `patch` and the channelAtt gate (:274-276), run with it in ONE kernel."""
    refimg_fea, targetimg_fea = dfr.real(refimg_fea), dfr.real(targetimg_fea)
    _check_pair(refimg_fea, targetimg_fea, num_groups)
    rng = _range or signed_range(maxdisp)
    if (_defer and dfr.on(None, refimg_fea, targetimg_fea) and refimg_fea.dtype == torch.float32
            and gwc_patch_gate_applies(refimg_fea, maxdisp, num_groups, rng, targetimg_fea)):
        return dfr.Deferred.call("gwc_norm", lambda a, b, m, g, r: _build_gwc_volume_norm(a, b, m, g, r, _defer=False),
                               refimg_fea, targetimg_fea, maxdisp, int(num_groups), rng)
    if _needs_grad(refimg_fea, targetimg_fea):
        # normalise once with autograd-visible ops, then the volume kernel with its HIP backward
        return _GwcVolume.apply(_group_normalise(refimg_fea, num_groups), _group_normalise(targetimg_fea, num_groups),
                                rng, int(num_groups))
    return _gwc_forward(_c(refimg_fea), _c(targetimg_fea), rng, int(num_groups), True)


def gwc_patch_gate_applies(fea, maxdisp, num_groups, _range=None, *others):
    """Shapes (and operand alignment: 16-byte loads) ss_gwc_patch_gate_fwd is built for (otherwise: the volume kernel + the
    patch kernel).  `others`: the remaining device operands of the call (right features, gate logits), checked for alignment."""
    B, C, H, W = fea.shape
    for t in (fea,) + others:
        if t is not None and t.is_cuda and (t.data_ptr() % 16 != 0 or not t.is_contiguous()):
            return False
    cg = C // num_groups
    dmin, nd = _range or signed_range(maxdisp)
    halo = (max(-dmin, dmin + nd - 1, 0) + 3) // 4 * 4
    lds = (cg * 8 * (128 + 2 * halo + 8) + 8 * 8 * 136 + 16 * cg + 6 * 128) * 4
    return W % 4 == 0 and dmin % 4 == 0 and nd % 8 == 0 and cg in (4, 8) and lds <= 150 * 1024 and B * num_groups <= 65535


def gwc_patch_gate(refimg_fea, targetimg_fea, maxdisp, num_groups, patch_weight, gate_logits=None, normalize=True, _range=None):
    """Fused models/SemStereo.py:273-276: build_gwc_volume_norm -> patch (depthwise (1,3,3)) -> channelAtt gate in one
    kernel; bit-identical to the two-kernel form.  patch_weight [G,1,1,3,3]; gate_logits [B,G,H,W] or None.  Inference only."""
    _check_pair(refimg_fea, targetimg_fea, num_groups)
    ref, tgt, w = _c(refimg_fea), _c(targetimg_fea), _c(patch_weight.detach())
    g = None if gate_logits is None else _c(gate_logits)
    dev = _lib.require_device(ref, tgt, w, g)
    B, C, H, W = ref.shape
    assert w.numel() == num_groups * 9 and (g is None or g.shape == (B, num_groups, H, W))
    dmin, nd = _range or signed_range(maxdisp)
    out = torch.empty((B, num_groups, nd, H, W), dtype=ref.dtype, device=ref.device)
    with torch.cuda.device(dev):
        call("ss_gwc_patch_gate_fwd", ptr(ref), ptr(tgt), ptr(w), ptr(g), ptr(out), B, C, H, W, dmin, nd, int(num_groups),
             int(normalize))
    return out


def _group_corr(fea1, fea2, groups, normalize):
    _check_pair(fea1, fea2, groups)
    if _needs_grad(fea1, fea2):
        if normalize:
            fea1, fea2 = _group_normalise(fea1, groups), _group_normalise(fea2, groups)
        B, C, H, W = fea1.shape
        return (fea1 * fea2).reshape(B, groups, C // groups, H, W).mean(dim=2)
    fea1, fea2 = _c(fea1), _c(fea2)
    dev = _lib.require_device(fea1, fea2)
    B, C, H, W = fea1.shape
    out = torch.empty((B, groups, H, W), dtype=fea1.dtype, device=fea1.device)
    if out.numel():
        with torch.cuda.device(dev):
            call("ss_groupwise_correlation_fwd", ptr(fea1), ptr(fea2), ptr(out), B, C, H, W, groups, int(normalize))
    return out


@dfr.realising
def groupwise_correlation(fea1, fea2, num_groups):
    """models/submodule.py:190-196 -> [B, G, H, W]."""
    return _group_corr(fea1, fea2, int(num_groups), False)


@dfr.realising
def groupwise_correlation_norm(fea1, fea2, num_groups):
    """models/submodule.py:213-221 -> [B, G, H, W]."""
    return _group_corr(fea1, fea2, int(num_groups), True)


# --------------------------------------------------------------------------------------
# dense concat volume
# --------------------------------------------------------------------------------------

class _ConcatVolume(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ref, tgt, rng, mask_left):
        ref, tgt = _c(ref), _c(tgt)
        dev = _lib.require_device(ref, tgt)
        B, C, H, W = ref.shape
        dmin, nd = rng
        ctx.cfg = (B, C, H, W, dmin, nd, int(mask_left))
        out = torch.empty((B, 2 * C, nd, H, W), dtype=ref.dtype, device=ref.device)
        if out.numel():
            with torch.cuda.device(dev):
                call("ss_concat_volume_fwd", ptr(ref), ptr(tgt), ptr(out), B, C, H, W, dmin, nd, int(mask_left))
        return out

    @staticmethod
    def backward(ctx, g):
        B, C, H, W, dmin, nd, mask_left = ctx.cfg
        g = _c(g)
        gref = torch.empty((B, C, H, W), dtype=g.dtype, device=g.device)
        gtgt = torch.empty_like(gref)
        with torch.cuda.device(g.device):
            call("ss_concat_volume_bwd", ptr(g), ptr(gref), ptr(gtgt), B, C, H, W, dmin, nd, mask_left)
        return gref, gtgt, None, None


@dfr.realising
def _build_concat_volume(refimg_fea, targetimg_fea, maxdisp, _range=None, _mask_left=True):
    """models/submodule.py:173-187 -> [B, 2C, 2*maxdisp, H, W]."""
    _check_pair(refimg_fea, targetimg_fea)
    return _ConcatVolume.apply(refimg_fea, targetimg_fea, _range or signed_range(maxdisp), bool(_mask_left))


# --------------------------------------------------------------------------------------
# regressions
# --------------------------------------------------------------------------------------

class _DisparityRegression(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, rng):
        x = _c(x)
        dev = _lib.require_device(x)
        B, D, H, W = x.shape
        dmin, nd = rng
        ctx.cfg = (B, H, W, dmin, nd)
        out = torch.empty((B, H, W), dtype=x.dtype, device=x.device)
        if out.numel():
            with torch.cuda.device(dev):
                call("ss_disparity_regression_fwd", ptr(x), ptr(out), B, dmin, nd, H, W)
        return out

    @staticmethod
    def backward(ctx, g):
        B, H, W, dmin, nd = ctx.cfg
        g = _c(g)
        gx = torch.empty((B, nd, H, W), dtype=g.dtype, device=g.device)
        with torch.cuda.device(g.device):
            call("ss_disparity_regression_bwd", ptr(g), ptr(gx), B, dmin, nd, H, W)
        return gx, None


def _disparity_regression(x, maxdisp, _range=None):
    """models/submodule.py:164-170: [B, 2*maxdisp, H, W] -> [B, H, W].  Given the deferred handle of
    `F.softmax(torch.squeeze(F.interpolate(cost, size, mode='trilinear'), 1), dim=1)` (models/SemStereo.py:279-282), the
    up-sampling, the soft-max, this regression and the variance of :285 are one kernel."""
    rng = _range or signed_range(maxdisp)
    if isinstance(x, dfr.Deferred):
        fused = dfr.regression_of(x, rng)
        if fused is not None:
            return fused
        x = x.value()
    assert len(x.shape) == 4
    assert x.shape[1] == rng[1], "the disparity axis must span the whole range"
    return _DisparityRegression.apply(x, rng)


class _DisparityVariance(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, disparity, rng):
        x, disparity = _c(x), _c(disparity)
        dev = _lib.require_device(x, disparity)
        B, D, H, W = x.shape
        ctx.save_for_backward(x, disparity)
        ctx.rng = rng
        out = torch.empty((B, 1, H, W), dtype=x.dtype, device=x.device)
        if out.numel():
            with torch.cuda.device(dev):
                call("ss_disparity_variance_fwd", ptr(x), ptr(disparity), ptr(out), B, rng[0], rng[1], H, W)
        return out

    @staticmethod
    def backward(ctx, g):
        x, disparity = ctx.saved_tensors
        dmin, nd = ctx.rng
        dv = torch.arange(dmin, dmin + nd, dtype=x.dtype, device=x.device).reshape(1, nd, 1, 1) - disparity
        gx = g * dv * dv
        gd = (g * (x * dv).sum(dim=1, keepdim=True)) * -2.0
        return gx, gd, None


def _disparity_variance(x, maxdisp, disparity, _range=None):
    """models/submodule.py:257-263: x [B,2m,H,W], disparity [B,1,H,W] -> [B,1,H,W]."""
    rng = _range or signed_range(maxdisp)
    disparity = dfr.real(disparity)
    if isinstance(x, dfr.Deferred):
        fused = dfr.variance_of(x, rng, disparity)          # computed together with the regression of the same handle
        if fused is not None:
            return fused
        x = x.value()
    assert len(x.shape) == 4
    assert x.shape[1] == rng[1] and disparity.shape == (x.shape[0], 1, x.shape[2], x.shape[3])
    return _DisparityVariance.apply(x, disparity, rng)


def softmax_regression(logits, maxdisp, want_prob=False, _range=None):
    """Fused models/SemStereo.py:281-285: softmax over the disparity axis, its expectation and its
    variance in one kernel.  logits [B,2m,H,W] -> (disp [B,H,W], var [B,1,H,W], prob or None).
    Inference only."""
    logits = _c(logits)
    dev = _lib.require_device(logits)
    B, D, H, W = logits.shape
    dmin, nd = _range or signed_range(maxdisp)
    assert D == nd
    disp = torch.empty((B, H, W), dtype=logits.dtype, device=logits.device)
    var = torch.empty((B, 1, H, W), dtype=logits.dtype, device=logits.device)
    prob = torch.empty_like(logits) if want_prob else None
    with torch.cuda.device(dev):
        call("ss_softmax_regression_fwd", ptr(logits), ptr(prob), ptr(disp), ptr(var), B, dmin, nd, H, W)
    return disp, var, prob


def upsample_softmax_regression(coarse, maxdisp, H, W, _range=None):
    """Fused models/SemStereo.py:279-285: trilinear 2x up-sampling of the classifier output `coarse` [B,1,maxdisp,H/2,W/2]
    to [B,1,2*maxdisp,H,W], softmax over the disparity axis, its expectation and variance -- one kernel.
    -> (att_weights [B,1,2m,H,W], disp [B,H,W], var [B,1,H,W]).  Inference only; needs exact 2x and 2*maxdisp <= 128."""
    coarse = _c(coarse)
    dev = _lib.require_device(coarse)
    B = coarse.shape[0]
    dmin, nd = _range or signed_range(maxdisp)
    assert tuple(coarse.shape[1:]) == (1, nd // 2, H // 2, W // 2) and H % 2 == 0 and W % 2 == 0 and nd % 2 == 0
    up = torch.empty((B, 1, nd, H, W), dtype=coarse.dtype, device=coarse.device)
    disp = torch.empty((B, H, W), dtype=coarse.dtype, device=coarse.device)
    var = torch.empty((B, 1, H, W), dtype=coarse.dtype, device=coarse.device)
    with torch.cuda.device(dev):
        call("ss_upsample_softmax_regression_fwd", ptr(coarse), ptr(up), ptr(disp), ptr(var), B, dmin, nd, H, W)
    return up, disp, var


def upsample_softmax_regression_applies(coarse, maxdisp, H, W, _range=None):
    dmin, nd = _range or signed_range(maxdisp)
    return (coarse.dim() == 5 and nd % 2 == 0 and tuple(coarse.shape[1:]) == (1, nd // 2, H // 2, W // 2) and H % 2 == 0
            and W % 2 == 0 and nd <= 128)


class _RegressionTopk(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cost, samples, k):
        cost, samples = _c(cost), _c(samples)
        dev = _lib.require_device(cost, samples)
        B, nd, H, W = cost.shape
        ctx.save_for_backward(cost, samples)
        ctx.k = k
        out = torch.empty((B, 1, H, W), dtype=cost.dtype, device=cost.device)
        if out.numel():
            with torch.cuda.device(dev):
                call("ss_regression_topk_fwd", ptr(cost), ptr(samples), ptr(out), B, nd, H, W, k)
        return out

    @staticmethod
    def backward(ctx, g):
        cost, samples = ctx.saved_tensors
        g = _c(g)
        B, nd, H, W = cost.shape
        gc, gs = torch.empty_like(cost), torch.empty_like(samples)
        with torch.cuda.device(cost.device):
            call("ss_regression_topk_bwd", ptr(g), ptr(cost), ptr(samples), ptr(gc), ptr(gs), B, nd, H, W, ctx.k)
        return gc, gs, None


def regression_topk(cost, disparity_samples, k):
    """models/submodule.py:434-442: cost, samples [B,nd,H,W] -> [B,1,H,W].  Ties between equal
    costs resolve to the lower candidate index (the reference's unstable sort leaves them open)."""
    cost, disparity_samples = dfr.real(cost), dfr.real(disparity_samples)
    assert cost.dim() == 4 and cost.shape == disparity_samples.shape
    k = int(k)
    assert 1 <= k <= cost.shape[1]
    if k > 32:
        raise NotImplementedError("regression_topk: k > 32 is not built (the model uses k = 2)")
    return _RegressionTopk.apply(cost, disparity_samples, k)


def regression_topk_patched(pc, disparity_samples, k):
    """regression_topk(cost.squeeze(1), samples, k) where the cost is still the patches of a one-pass classifier (engine.PatchedCost:
    ss_regression_topk_patched_fwd, r06) -- bit-identical to summing the patches first.  Inference only; k = 2, 24 candidates."""
    samples = _c(dfr.real(disparity_samples))
    B, D, H, W = pc.shape
    assert samples.shape == (B, D, H, W) and int(k) == 2 and D == 24
    dev = _lib.require_device(pc.patches, samples)
    out = torch.empty((B, 1, H, W), dtype=samples.dtype, device=samples.device)
    with torch.cuda.device(dev):
        call("ss_regression_topk_patched_fwd", ptr(pc.patches), ptr(samples), ptr(out), B, D, H, W, int(k))
    return out


# --------------------------------------------------------------------------------------
# candidate warping
# --------------------------------------------------------------------------------------

class _WarpSampled(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, disp):
        x, y, disp = _c(x), _c(y), _c(disp)
        dev = _lib.require_device(x, y, disp)
        B, C, H, W = y.shape
        nd = disp.shape[1]
        ctx.save_for_backward(x, y, disp)
        yw = torch.empty((B, C, nd, H, W), dtype=y.dtype, device=y.device)
        xw = torch.empty_like(yw)
        if yw.numel():
            with torch.cuda.device(dev):
                call("ss_warp_sampled_fwd", ptr(x), ptr(y), ptr(disp), ptr(yw), ptr(xw), B, C, H, W, nd)
        return yw, xw

    @staticmethod
    def backward(ctx, gyw, gxw):
        x, y, disp = ctx.saved_tensors
        B, C, H, W = y.shape
        nd = disp.shape[1]
        need_x, need_y, need_d = ctx.needs_input_grad
        gyw = _c(gyw) if (need_y or need_d) else None
        gxw = _c(gxw) if need_x else None
        gx = torch.empty_like(x) if need_x else None
        gy = torch.empty_like(y) if need_y else None
        gd = torch.empty_like(disp) if need_d else None
        with torch.cuda.device(y.device):
            call("ss_warp_sampled_bwd", ptr(gyw), ptr(gxw), ptr(y), ptr(disp), ptr(gx), ptr(gy), ptr(gd), B, C, H, W, nd)
        return gx, gy, gd


def SpatialTransformer_grid(x, y, disp_range_samples):
    """models/submodule.py:265-288: x, y [B,C,H,W], disp [B,nd,H,W] -> (y_warped, x_warped), both
    [B,C,nd,H,W]."""
    x, y = dfr.real(x), dfr.real(y)
    if dfr.on(None, x, y, disp_range_samples):
        # inference: what the caller does with the two results decides the kernel (:291-293 the 5-candidate probe fed by the
        # propagated regression of :289; :316-320 the sparse concat volume -> x att_topk -> concat_stem -> gate); any other
        # use replays this very call
        assert x.dim() == 4 and x.shape == y.shape
        return dfr.Deferred.pair(dfr.Deferred.call("stn", lambda a, b, d: SpatialTransformer_grid(a, b, d), x, y, disp_range_samples))
    disp_range_samples = dfr.real(disp_range_samples)
    assert x.dim() == 4 and x.shape == y.shape and disp_range_samples.dim() == 4
    assert disp_range_samples.shape[0] == y.shape[0] and disp_range_samples.shape[2:] == y.shape[2:]
    return _WarpSampled.apply(x, y, disp_range_samples)


def concat_volume_sampled(left, right, disparity_samples, att=None):
    """Fused SemStereo.concat_volume_generator + `att_topk * volume` (models/SemStereo.py:241-244,
    316-318): -> [B, 2C, nd, H, W] = att * cat(left broadcast, warp(right)).  att is [B,1,nd,H,W],
    [B,nd,H,W] or None.  Inference only."""
    right, disp = _c(right), _c(disparity_samples)
    left = None if left is None else _c(left)            # None: only the right half, [B, C, nd, H, W]
    if att is not None:
        att = _c(att.reshape(att.shape[0], att.shape[-3], att.shape[-2], att.shape[-1]))
    dev = _lib.require_device(left, right, disp, att)
    B, C, H, W = right.shape
    nd = disp.shape[1]
    out = torch.empty((B, (2 if left is not None else 1) * C, nd, H, W), dtype=right.dtype, device=right.device)
    with torch.cuda.device(dev):
        call("ss_concat_sampled_fwd", ptr(left), ptr(right), ptr(disp), ptr(att), ptr(out), B, C, H, W, nd)
    return out


def concat_volume_sampled_presplit(right, disparity_samples, att=None):
    """The warped half of `att * concat_volume` (models/SemStereo.py:241-244, 316-318) in the pre-split operand form of the
    matrix-core stem (ss_conv3d_presplit_fwd): -> (xs int16 [B, C/8, 2, nd, H, W, 8], xexp int32 [3B]).  Inference only."""
    right, disp = _c(right), _c(disparity_samples)
    if att is not None:
        att = _c(att.reshape(att.shape[0], att.shape[-3], att.shape[-2], att.shape[-1]))
    dev = _lib.require_device(right, disp, att)
    B, C, H, W = right.shape
    nd = disp.shape[1]
    assert C % 8 == 0
    xs = torch.empty((B, C // 8, 2, nd, H, W, 8), dtype=torch.int16, device=right.device)
    xexp = torch.empty(3 * B, dtype=torch.int32, device=right.device)
    with torch.cuda.device(dev):
        call("ss_concat_sampled_presplit_fwd", ptr(right), ptr(disp), ptr(att), ptr(xs), ptr(xexp), B, C, H, W, nd)
    return xs, xexp


def stem_left(q, att):
    """Left (broadcast) half of concat_stem by linearity (stem_left.hip): q [B, 27*Cout, H, W] = the 1x1
    projections of the 2-D left features onto the stem's left-half weights, att [B,1,nd,H,W] or [B,nd,H,W]
    -> [B, Cout, nd, H, W] = sum_tap att[pos + tap] * q[tap, :, (pos + tap).hw].  Inference only."""
    q = _c(q)
    att = _c(att.reshape(att.shape[0], att.shape[-3], att.shape[-2], att.shape[-1]))
    dev = _lib.require_device(q, att)
    B, nd, H, W = att.shape
    assert q.shape[0] == B and q.shape[1] % 27 == 0 and tuple(q.shape[2:]) == (H, W)
    Cout = q.shape[1] // 27
    out = torch.empty((B, Cout, nd, H, W), dtype=q.dtype, device=q.device)
    with torch.cuda.device(dev):
        call("ss_stem_left_fwd", ptr(q), ptr(att), ptr(out), B, Cout, nd, H, W)
    return out


def stem_left_fused(left, wsplit, att, Cout, nterms=6):
    """stem_left with q never materialised: left [B,32,H,W], wsplit = the pair-major packed left-half weights
    (modules.stem_of_broadcast_and_volume builds them), att [B,1,nd,H,W] or [B,nd,H,W] -> [B,Cout,nd,H,W]."""
    left = _c(left)
    att = _c(att.reshape(att.shape[0], att.shape[-3], att.shape[-2], att.shape[-1]))
    dev = _lib.require_device(left, att)
    B, nd, H, W = att.shape
    out = torch.empty((B, Cout, nd, H, W), dtype=left.dtype, device=left.device)
    with torch.cuda.device(dev):
        call("ss_stem_left_fused_fwd", ptr(left), ptr(wsplit), ptr(att), ptr(out), B, left.shape[1], Cout, nd, H, W, int(nterms))
    return out


def warp_correlation(x, y, disparity_samples):
    """Fused models/SemStereo.py:291-292: mean over channels of x * warp(y) -> [B, nd, H, W].
    Inference only."""
    x, y, disp = _c(x), _c(y), _c(disparity_samples)
    dev = _lib.require_device(x, y, disp)
    B, C, H, W = x.shape
    nd = disp.shape[1]
    out = torch.empty((B, nd, H, W), dtype=x.dtype, device=x.device)
    with torch.cuda.device(dev):
        call("ss_warp_correlation_fwd", ptr(x), ptr(y), ptr(disp), ptr(out), B, C, H, W, nd)
    return out


def sample_strength(left, right, pred0, var, gamma, beta):
    """Fused models/SemStereo.py:286-293: -> disparity_sample_strength [B,5,H,W] (softmaxed over the five
    propagated candidates).  left, right [B,C,H,W]; pred0 [B,H,W]; var [B,1,H,W]; gamma, beta 1-element
    tensors.  Inference only."""
    left, right, pred0, var = _c(left), _c(right), _c(pred0), _c(var)
    gamma, beta = _c(gamma.detach().reshape(1)), _c(beta.detach().reshape(1))
    dev = _lib.require_device(left, right, pred0, var, gamma, beta)
    B, C, H, W = left.shape
    assert pred0.shape == (B, H, W) and var.shape == (B, 1, H, W)
    out = torch.empty((B, 5, H, W), dtype=left.dtype, device=left.device)
    with torch.cuda.device(dev):
        call("ss_sample_strength_fwd", ptr(left), ptr(right), ptr(pred0), ptr(var), ptr(gamma), ptr(beta), ptr(out),
             B, C, H, W)
    return out


#: the fused selection keeps a pixel's probabilities in registers: D = 2 * (maxdisp // 4) <= 128 (attention_tail.hip)
TOPK_CANDIDATES_MAX_D = 128


def topk_candidates(att_weights, strength, maxdisp, k, _range=None):
    """Fused models/SemStereo.py:295-310: att_weights [B,1,2m,H,W] (up-sampled logits), strength [B,5,H,W]
    -> (att_topk [B,1,k,H,W], disparity_sample_topk [B,k,H,W], pred_att [B,H,W]).  Inference only."""
    att_weights, strength = _c(att_weights), _c(strength)
    dev = _lib.require_device(att_weights, strength)
    B, one, D, H, W = att_weights.shape
    dmin, nd = _range or signed_range(maxdisp)
    assert one == 1 and D == nd and strength.shape == (B, 5, H, W)
    samples = torch.empty((B, k, H, W), dtype=att_weights.dtype, device=att_weights.device)
    att_topk = torch.empty((B, 1, k, H, W), dtype=att_weights.dtype, device=att_weights.device)
    pred_att = torch.empty((B, H, W), dtype=att_weights.dtype, device=att_weights.device)
    with torch.cuda.device(dev):
        call("ss_topk_candidates_fwd", ptr(att_weights), ptr(strength), ptr(samples), ptr(att_topk), ptr(pred_att),
             B, dmin, nd, H, W, int(k))
    _mark_integer(samples, True)         # (index - offset) of :305: integer-valued, see integer_candidates()
    return att_topk, samples, pred_att


def _mark_integer(samples, verdict):
    # the verdict is tied to the tensor's version counter: an in-place update afterwards (samples.add_(0.5)) voids it
    samples._ss_integer = (bool(verdict), samples._version)


def integer_candidates(samples):
    """True when every disparity candidate of `samples` is an integer -- the precondition of the gathered stem
    (ss_conv3d_gather_fwd).  The candidates of topk_candidates carry a mark and cost nothing; any other tensor is checked on
    the device once (one reduction + a host sync) and then carries the verdict for as long as it is not modified in place."""
    mark = getattr(samples, "_ss_integer", None)
    if mark is not None and mark[1] == samples._version:
        return mark[0]
    if samples.is_cuda and torch.cuda.is_current_stream_capturing():
        return False            #  no host sync inside a HIP-graph capture: an unmarked tensor takes the warp launch
    verdict = bool(torch.equal(samples, torch.trunc(samples)))
    try:
        _mark_integer(samples, verdict)
    except Exception:       # noqa: BLE001  (an object that takes no attributes: checked again next time)
        pass
    return verdict


def channel_gate(att_logits, cv):
    """channelAtt's gating (models/SemStereo.py:101-102): sigmoid(att)[:, :, None] * cv.  Inference only."""
    att_logits, cv = _c(att_logits), _c(cv)
    dev = _lib.require_device(att_logits, cv)
    B, C, D, H, W = cv.shape
    assert att_logits.shape == (B, C, H, W)
    out = torch.empty_like(cv)
    with torch.cuda.device(dev):
        call("ss_channel_gate_fwd", ptr(att_logits), ptr(cv), ptr(out), B, C, D, H, W)
    return out


# ---- the reference's exact signatures (models/submodule.py: signed range); the `_name` forms above also take a range ----

def build_gwc_volume(refimg_fea, targetimg_fea, maxdisp, num_groups):
    """models/submodule.py:198-211 -> [B, G, 2*maxdisp, H, W] (signed disparity range)."""
    return _build_gwc_volume(refimg_fea, targetimg_fea, maxdisp, num_groups)


def build_gwc_volume_norm(refimg_fea, targetimg_fea, maxdisp, num_groups):
    """models/submodule.py:224-238 (the live call, models/SemStereo.py:273)."""
    return _build_gwc_volume_norm(refimg_fea, targetimg_fea, maxdisp, num_groups)


def build_concat_volume(refimg_fea, targetimg_fea, maxdisp):
    """models/submodule.py:173-187 -> [B, 2C, 2*maxdisp, H, W]."""
    return _build_concat_volume(refimg_fea, targetimg_fea, maxdisp)


def disparity_regression(x, maxdisp):
    """models/submodule.py:164-170: [B, 2*maxdisp, H, W] -> [B, H, W]."""
    return _disparity_regression(x, maxdisp)


def disparity_variance(x, maxdisp, disparity):
    """models/submodule.py:257-263: x [B,2m,H,W], disparity [B,1,H,W] -> [B,1,H,W]."""
    return _disparity_variance(x, maxdisp, disparity)


#: names the reference model module resolves by bare global
REFERENCE_NAMES = (
    "build_gwc_volume", "build_gwc_volume_norm", "groupwise_correlation", "groupwise_correlation_norm",
    "build_concat_volume", "disparity_regression", "disparity_variance", "SpatialTransformer_grid",
    "regression_topk",
)
