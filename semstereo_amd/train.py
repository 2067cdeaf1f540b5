"""Training side of the 3-D aggregation stack on the HIP kernels (main_us3d.py:186-222 back-propagates through every
module of models/SemStereo.py:273-323): torch.autograd.Function wrappers whose forward AND backward are this repo's
kernels, for what the 3x3x3 conv / transposed-conv functions of modules.py (`_Conv3dK3`, `_Deconv3dK3`) leave over:

  batchnorm_train      BatchNorm2d / 3d with BATCH statistics (+ fused ReLU)     convbn_3d, BasicConv in train()
  conv_k1              1x1(x1) convolutions with optional bias                    redir1/2, attention_block.qkv_3d / final1x1,
                                                                                  channelAtt.im_att
  conv2d_k3            3x3 Conv2d                                                 concat_feature
  depthwise_patch      the depthwise (1,3,3) `patch` Conv3d                       models/SemStereo.py:219
  channel_gate         sigmoid(att)[:, :, None] * cv                              channelAtt, models/SemStereo.py:101-102
  window_attention     the windowed attention core                                models/submodule_other.py:805-834

Each falls back to the stock PyTorch layer (and counts it in modules.PATH_COUNTS["torch"]) for shapes the kernels are not
built for.  `engine.TRAIN_HIP = False` (SS_TRAIN_HIP=0) sends everything to PyTorch.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from ._lib import call, ptr


def _M():
    from . import modules
    return modules


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def _on(x):
    return _M().TRAIN_HIP and x.is_cuda and x.dtype == torch.float32


def _count(kind):
    pc = _M().PATH_COUNTS
    pc[kind] = pc.get(kind, 0) + 1


# ---- BatchNorm with batch statistics (+ ReLU) ---------------------------------------------------------------------------------

class _BatchNormTrain(torch.autograd.Function):
    """y = bn(x) [+ residual] [-> ReLU] with batch statistics; `residual` (same shape as x, or None) joins behind the
    normalisation and before the ReLU (hourglass.forward's `F.relu(self.conv5(conv4) + self.redir2(conv2))`,
    models/SemStereo.py:141-142) in the same pass.  `stats` = (running_mean, running_var, num_batches_tracked, momentum) or None: the
    module's bookkeeping done by the statistics kernel (ss_batchnorm_train_fwd_rs) -- a tuple, so that autograd does not see the buffers."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, relu, residual, stats=None):
        x = _c(x)
        B, C = x.shape[0], x.shape[1]
        N = x[0, 0].numel()
        y = torch.empty_like(x)
        mean = torch.empty(C, dtype=torch.float32, device=x.device)
        invstd, var_u = torch.empty_like(mean), torch.empty_like(mean)
        work = torch.empty(2 * C, dtype=torch.float64, device=x.device)
        if residual is not None:
            residual = _c(residual)
            assert residual.shape == x.shape
        rm, rv, nbt, mom = stats if stats is not None else (None, None, None, 0.0)
        with torch.cuda.device(x.device):
            call("ss_batchnorm_train_fwd_rs", ptr(x), ptr(residual), ptr(weight), ptr(bias), ptr(y), ptr(mean), ptr(invstd), ptr(var_u), ptr(work),
                 ptr(rm), ptr(rv), ptr(nbt), float(mom), B, C, N, float(eps), int(relu))
        for t in (rm, rv, nbt):                              # (written through raw pointers: the packed-weight caches key on versions)
            if t is not None:
                torch.autograd.graph.increment_version(t)
        # (y is saved only where a residual joined before the ReLU: without one the backward recomputes the mask from x)
        ctx.save_for_backward(x, y if (relu and residual is not None) else None, mean, invstd, weight, bias)
        ctx.relu, ctx.has_bias, ctx.has_res = bool(relu), bias is not None, residual is not None
        ctx.mark_non_differentiable(mean, var_u)
        ctx.set_materialize_grads(False)                     # (no zero-filled gradients for `mean` / `var_u`: two launches per layer)
        return y, mean, var_u

    @staticmethod
    def backward(ctx, g, _gm, _gv):
        x, y, mean, invstd, weight, bias = ctx.saved_tensors
        if g is None:
            return (None,) * 7
        g = _c(g)
        B, C = x.shape[0], x.shape[1]
        N = x[0, 0].numel()
        gx = torch.empty_like(x)
        gres = torch.empty_like(x) if ctx.has_res else None
        work = torch.empty(2 * C, dtype=torch.float64, device=x.device)
        gw = torch.empty(C, dtype=torch.float32, device=x.device) if weight is not None else None
        gb = torch.empty(C, dtype=torch.float32, device=x.device) if ctx.has_bias else None
        with torch.cuda.device(x.device):
            call("ss_batchnorm_bwd_pg", ptr(g), ptr(x), ptr(y), ptr(mean), ptr(invstd), ptr(weight), ptr(bias), ptr(gx), ptr(gres), ptr(work),
                 ptr(gw), ptr(gb), 1, B, C, N, int(ctx.relu))
        return gx, gw, gb, None, None, gres, None


class _BatchNormEval(torch.autograd.Function):
    """y = bn(x) [+ residual] [-> ReLU] on the RUNNING statistics with autograd on (a module in eval() whose inputs or parameters
    need gradients): the statistics are constants of the graph."""

    @staticmethod
    def forward(ctx, x, weight, bias, mean, invstd, relu, residual):
        x = _c(x)
        B, C = x.shape[0], x.shape[1]
        N = x[0, 0].numel()
        y = torch.empty_like(x)
        residual = None if residual is None else _c(residual)
        with torch.cuda.device(x.device):
            call("ss_batchnorm_eval_fwd", ptr(x), ptr(residual), ptr(mean), ptr(invstd), ptr(weight), ptr(bias), ptr(y), B, C, N, int(relu))
        ctx.save_for_backward(x, y if (relu and residual is not None) else None, mean, invstd, weight, bias)
        ctx.relu, ctx.has_bias, ctx.has_res = bool(relu), bias is not None, residual is not None
        return y

    @staticmethod
    def backward(ctx, g):
        x, y, mean, invstd, weight, bias = ctx.saved_tensors
        g = _c(g)
        B, C = x.shape[0], x.shape[1]
        N = x[0, 0].numel()
        gx = torch.empty_like(x)
        gres = torch.empty_like(x) if ctx.has_res else None
        work = torch.empty(2 * C, dtype=torch.float64, device=x.device)
        gw = torch.empty(C, dtype=torch.float32, device=x.device) if weight is not None else None
        gb = torch.empty(C, dtype=torch.float32, device=x.device) if ctx.has_bias else None
        with torch.cuda.device(x.device):
            call("ss_batchnorm_bwd_pg", ptr(g), ptr(x), ptr(y), ptr(mean), ptr(invstd), ptr(weight), ptr(bias), ptr(gx), ptr(gres), ptr(work),
                 ptr(gw), ptr(gb), 0, B, C, N, int(ctx.relu))
        return gx, gw, gb, None, None, None, gres


def batchnorm_train(bn, x, relu=False, residual=None):
    """bn(x) [+ residual] [-> ReLU] for a BatchNorm2d / 3d under autograd: in train() on batch statistics, running statistics updated
    as F.batch_norm does; in eval() (frozen statistics, r05) on the running statistics as constants of the graph."""
    if (_on(x) and isinstance(bn, nn.modules.batchnorm._BatchNorm) and not bn.training and bn.running_mean is not None
            and x.shape[1] <= 65535):
        _count("hip_train")
        with torch.no_grad():
            invstd = torch.rsqrt(bn.running_var.float() + bn.eps)
        return _BatchNormEval.apply(x, bn.weight, bn.bias, bn.running_mean.float(), invstd, relu, residual)
    if not (_on(x) and isinstance(bn, nn.modules.batchnorm._BatchNorm) and bn.training and x.shape[1] <= 65535):
        _count("torch")
        y = bn(x)
        if residual is not None:
            y = y + residual
        return F.relu(y) if relu else y
    _count("hip_train")
    tracked = bn.track_running_stats and bn.running_mean is not None
    if (tracked and bn.momentum is not None and bn.running_mean.dtype == torch.float32 and bn.running_var.dtype == torch.float32
            and bn.running_mean.is_contiguous() and bn.running_var.is_contiguous() and bn.running_mean.device == x.device
            and (bn.num_batches_tracked is None or (bn.num_batches_tracked.dtype == torch.int64 and bn.num_batches_tracked.device == x.device))):
        # the running statistics and the batch counter move inside the statistics kernel (five element-wise launches per layer otherwise)
        return _BatchNormTrain.apply(x, bn.weight, bn.bias, bn.eps, relu, residual,
                                     (bn.running_mean, bn.running_var, bn.num_batches_tracked, float(bn.momentum)))[0]
    y, mean, var_u = _BatchNormTrain.apply(x, bn.weight, bn.bias, bn.eps, relu, residual, None)
    if tracked:
        with torch.no_grad():
            if bn.num_batches_tracked is not None:
                bn.num_batches_tracked += 1
            m = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
            bn.running_mean.mul_(1.0 - m).add_(mean, alpha=m)
            bn.running_var.mul_(1.0 - m).add_(var_u, alpha=m)
    return y


# ---- 1x1(x1) convolutions -----------------------------------------------------------------------------------------------------

def _as5d(x):
    return x if x.dim() == 5 else x.unsqueeze(2)             # [B,C,H,W] -> [B,C,1,H,W]


def _k1_forward(x5, w2d, bias):
    """y[b,co,pos] = sum_ci w[co,ci] * x[b,ci,pos] (+ bias) on the exact-fp32 matrix-core kernel (ss_conv3d_fwd, k = 1)."""
    M = _M()
    wp = M.pack_conv_weight(w2d.reshape(w2d.shape[0], w2d.shape[1], 1, 1, 1))
    return M.conv3d_hip(x5, wp, None, None if bias is None else _c(bias.detach().float()), 1, 1, False)


class _ConvK1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias):
        x5 = _c(_as5d(x))
        w2d = w.detach().reshape(w.shape[0], w.shape[1])
        ctx.save_for_backward(x5, w)
        ctx.was4d, ctx.has_bias = x.dim() == 4, bias is not None
        y = _k1_forward(x5, w2d, bias)
        return y.squeeze(2) if ctx.was4d else y

    @staticmethod
    def backward(ctx, g):
        x5, w = ctx.saved_tensors
        M = _M()
        g5 = _c(_as5d(g))
        Cout, Cin = w.shape[0], w.shape[1]
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = _k1_forward(g5, w.detach().reshape(Cout, Cin).t().contiguous(), None)
            gx = gx.squeeze(2) if ctx.was4d else gx
        if ctx.needs_input_grad[1]:
            gw = torch.empty((Cout, Cin), dtype=torch.float32, device=g5.device)
            with torch.cuda.device(g5.device):
                call("ss_conv_k1_wgrad_fwd", ptr(g5), ptr(x5), ptr(gw), g5.shape[0], Cin, Cout, g5[0, 0].numel())
            gw = gw.reshape(w.shape)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            sums = torch.empty(Cout, dtype=torch.float64, device=g5.device)
            with torch.cuda.device(g5.device):
                call("ss_channel_sum_fwd", ptr(g5), ptr(sums), g5.shape[0], Cout, g5[0, 0].numel())
            gb = sums.float()
        return gx, gw, gb


def conv_k1(x, weight, bias=None):
    """1x1 Conv2d / 1x1x1 Conv3d / Linear over the channel axis of [B,C,*spatial]."""
    if _on(x) and x.dim() in (4, 5):
        _count("hip_train")
        return _ConvK1.apply(x, weight, bias)
    _count("torch")
    w = weight.reshape(weight.shape[0], weight.shape[1], *([1] * (x.dim() - 2)))
    return (F.conv3d if x.dim() == 5 else F.conv2d)(x, w, bias)


def is_k1(conv):
    k = conv.kernel_size
    return (all(v == 1 for v in k) and all(v == 1 for v in conv.stride) and all(v == 0 for v in conv.padding) and conv.groups == 1
            and all(v == 1 for v in conv.dilation))


# ---- 3x3 Conv2d (concat_feature) ----------------------------------------------------------------------------------------------

class _Conv2dK3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        M = _M()
        x = _c(x)
        ctx.save_for_backward(x, w)
        nt = M._tiled_nterms() if M.CONV_ENGINE != "f32" else 6
        return M.conv2d_bf16s_hip(x, M.pack_conv2d_weight_bf16s(w.detach(), nt), w.shape[0], None, None, False, nt)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        M = _M()
        g = _c(g)
        gx = gw = None
        nt = M._tiled_nterms() if M.CONV_ENGINE != "f32" else 6
        if ctx.needs_input_grad[0]:
            wt = w.detach().transpose(0, 1).flip(2, 3).contiguous()
            gx = M.conv2d_bf16s_hip(g, M.pack_conv2d_weight_bf16s(wt, nt), w.shape[1], None, None, False, nt)
        if ctx.needs_input_grad[1]:
            # depth-1 volumes through the 3x3x3 weight-gradient kernel: its kd = 1 plane is the 3x3 gradient
            gw = M.conv3d_wgrad_hip(g.unsqueeze(2), x.unsqueeze(2), w.shape[0], w.shape[1], 1)[:, :, 1].contiguous()
        return gx, gw


def conv2d_k3(conv, x):
    M = _M()
    if _on(x) and M._is_plain_3x3(conv):
        _count("hip_train")
        return _Conv2dK3.apply(x, conv.weight)
    _count("torch")
    return conv(x)


# ---- depthwise `patch` --------------------------------------------------------------------------------------------------------

def _patch_forward(x, w):
    x = _c(x)
    B, C, D, H, W = x.shape
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        call("ss_depthwise_patch_fwd", ptr(x), ptr(_c(w)), None, ptr(out), B, C, D, H, W)
    return out


class _DepthwisePatch(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        x = _c(x)
        ctx.save_for_backward(x, w)
        return _patch_forward(x, w.detach())

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g = _c(g)
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = _patch_forward(g, w.detach().flip(3, 4).contiguous())          # correlation with the flipped 3x3 taps
        if ctx.needs_input_grad[1]:
            B, C, D, H, W = x.shape
            gw = torch.empty_like(w)
            with torch.cuda.device(x.device):
                call("ss_depthwise_patch_wgrad_fwd", ptr(g), ptr(x), ptr(gw), B, C, D, H, W)
        return gx, gw


def depthwise_patch(conv, x):
    if _on(x) and x.dim() == 5:
        _count("hip_train")
        return _DepthwisePatch.apply(x, conv.weight)
    _count("torch")
    return nn.Conv3d.forward(conv, x)


# ---- channelAtt gate ----------------------------------------------------------------------------------------------------------

class _ChannelGate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, att, cv):
        from . import ops
        att, cv = _c(att), _c(cv)
        ctx.save_for_backward(att, cv)
        return ops.channel_gate(att, cv)

    @staticmethod
    def backward(ctx, g):
        from . import ops
        att, cv = ctx.saved_tensors
        g = _c(g)
        ga = gc = None
        if ctx.needs_input_grad[0]:
            B, C, D, H, W = cv.shape
            ga = torch.empty_like(att)
            with torch.cuda.device(cv.device):
                call("ss_channel_gate_bwd_logits", ptr(g), ptr(cv), ptr(att), ptr(ga), B, C, D, H, W)
        if ctx.needs_input_grad[1]:
            gc = ops.channel_gate(att, g)
        return ga, gc


def channel_gate(att_logits, cv):
    if _on(cv):
        _count("hip_train")
        return _ChannelGate.apply(att_logits, cv)
    _count("torch")
    return torch.sigmoid(att_logits).unsqueeze(2) * cv


# ---- windowed attention core --------------------------------------------------------------------------------------------------

class _WindowAttentionCore(torch.autograd.Function):
    """y = softmax(q k^T / sqrt(8) [+ pad mask]) v per (window, head).  `bqkv`: the qkv Linear's bias = the q / k / v of the pad
    tokens of volumes whose H, W are not window multiples (the reference pads the volume before the Linear); its gradient output
    is what reaches those tokens."""

    @staticmethod
    def forward(ctx, qkv, bqkv, heads, block):
        qkv = _c(qkv)
        B, C3, D, H, W = qkv.shape
        C = C3 // 3
        y = torch.empty((B, C, D, H, W), dtype=qkv.dtype, device=qkv.device)
        bq = _c(bqkv.detach().float())
        with torch.cuda.device(qkv.device):
            call("ss_window_attention_core_fwd", ptr(qkv), ptr(bq), ptr(y), B, C, D, H, W, heads, block[0], block[1], block[2])
        ctx.save_for_backward(qkv, bq)
        ctx.cfg = (heads, tuple(block))
        return y

    @staticmethod
    def backward(ctx, g):
        qkv, bq = ctx.saved_tensors
        heads, block = ctx.cfg
        g = _c(g)
        B, C3, D, H, W = qkv.shape
        gq = torch.empty_like(qkv)
        gbias = torch.empty(C3, dtype=torch.float32, device=qkv.device)
        with torch.cuda.device(qkv.device):
            call("ss_window_attention_core_pad_bwd", ptr(qkv), ptr(bq), ptr(g), ptr(gq), ptr(gbias), B, C3 // 3, D, H, W, heads,
                 block[0], block[1], block[2])
        return gq, (gbias if ctx.needs_input_grad[1] else None), None, None


def window_attention_applies(x, heads, block):
    B, C, D, H, W = x.shape
    return _on(x) and C == heads * 8 and D % block[0] == 0 and block[0] * block[1] * block[2] in (64, 96)


def window_attention(x, qkv_linear, final_conv, heads, block):
    """attention_block.forward (models/submodule_other.py:790-837): the qkv projection and final1x1 as 1x1x1 convolutions with bias
    over the REAL positions, the attention core per (window, head) -- pad tokens of volumes whose H, W are not window multiples
    carry the Linear's bias as their q / k / v (r04: forward and backward)."""
    _count("hip_train")
    qkv = conv_k1(x, qkv_linear.weight, qkv_linear.bias)
    y = _WindowAttentionCore.apply(qkv, qkv_linear.bias, heads, tuple(block))
    return conv_k1(y, final_conv.weight, final_conv.bias)


# ---- the attention tail (models/SemStereo.py:279-310) under autograd: the fused forward kernels + their backward kernels --------------

class _UpsampleSoftmaxRegression(torch.autograd.Function):
    """:279-285: trilinear 2x up-sampling of the classifier's output -> softmax over D -> expectation and variance."""

    @staticmethod
    def forward(ctx, coarse, H, W, rng):
        from . import ops
        up, disp, var = ops.upsample_softmax_regression(coarse, rng[1] // 2 if rng[0] else rng[1], H, W, _range=rng)
        ctx.save_for_backward(up)
        ctx.rng = rng
        return up, disp, var

    @staticmethod
    def backward(ctx, g_up, g_disp, g_var):
        (up,) = ctx.saved_tensors
        B, _, D, H, W = up.shape
        g_up = None if g_up is None else _c(g_up)
        g_disp = None if g_disp is None else _c(g_disp)
        g_var = None if g_var is None else _c(g_var)
        gc = torch.empty((B, 1, D // 2, H // 2, W // 2), dtype=up.dtype, device=up.device)
        work = torch.empty_like(up)
        with torch.cuda.device(up.device):
            call("ss_upsample_softmax_regression_bwd", ptr(up), ptr(g_up), ptr(g_disp), ptr(g_var), ptr(gc), ptr(work), B, ctx.rng[0], D, H, W)
        return gc, None, None, None


#: SS_SSB_TWO_LAUNCHES=0: the one-launch backward of the probe (rounds 4-5) instead of the two-launch form over a scratch (r06)
SSB_TWO_LAUNCHES = os.environ.get("SS_SSB_TWO_LAUNCHES", "1") != "0"


class _SampleStrength(torch.autograd.Function):
    """:286-293: the 5-candidate matching-strength probe."""

    @staticmethod
    def forward(ctx, left, right, pred0, var, gamma, beta):
        from . import ops
        left, right, pred0, var = _c(left), _c(right), _c(pred0), _c(var)
        ctx.save_for_backward(left, right, pred0, var, gamma, beta)
        return ops.sample_strength(left, right, pred0, var, gamma, beta)

    @staticmethod
    def backward(ctx, g):
        left, right, pred0, var, gamma, beta = ctx.saved_tensors
        g = _c(g)
        B, C, H, W = left.shape
        need = ctx.needs_input_grad
        gl = torch.empty_like(left) if need[0] else None
        gr = torch.empty_like(right) if need[1] else None
        gp = torch.empty_like(pred0) if need[2] else None
        gv = torch.empty_like(var) if need[3] else None
        ggb = torch.empty(2, dtype=torch.float32, device=left.device) if (need[4] or need[5]) else None
        gm, bt = _c(gamma.detach().reshape(1)), _c(beta.detach().reshape(1))
        with torch.cuda.device(left.device):
            if SSB_TWO_LAUNCHES:        # (r06) through a [B,5,H,W] scratch: the channels on the second launch's grid
                work = torch.empty((B, 5, H, W), dtype=torch.float32, device=left.device)
                call("ss_sample_strength_bwd_ws", ptr(left), ptr(right), ptr(pred0), ptr(var), ptr(gm), ptr(bt), ptr(g), ptr(gl), ptr(gr), ptr(gp),
                     ptr(gv), ptr(ggb), ptr(work), B, C, H, W)
            else:
                call("ss_sample_strength_bwd", ptr(left), ptr(right), ptr(pred0), ptr(var), ptr(gm), ptr(bt), ptr(g), ptr(gl), ptr(gr), ptr(gp),
                     ptr(gv), ptr(ggb), B, C, H, W)
        return (gl, gr, gp, gv, ggb[0:1].reshape(gamma.shape) if need[4] else None, ggb[1:2].reshape(beta.shape) if need[5] else None)


class _TopkCandidates(torch.autograd.Function):
    """:295-310: strength-weighted propagation of the logits, softmax, the 24 most probable disparities, soft-argmax over them."""

    @staticmethod
    def forward(ctx, logits, strength, k, rng):
        from . import ops
        logits, strength = _c(logits), _c(strength)
        att_topk, samples, pred_att = ops.topk_candidates(logits, strength, rng[1] // 2 if rng[0] else rng[1], k, _range=rng)
        ctx.save_for_backward(logits, strength, samples)
        ctx.cfg = (k, rng)
        ctx.mark_non_differentiable(samples)
        return att_topk, samples, pred_att

    @staticmethod
    def backward(ctx, g_att, _g_samples, g_pred):
        logits, strength, samples = ctx.saved_tensors
        k, rng = ctx.cfg
        B, _, D, H, W = logits.shape
        g_att = None if g_att is None else _c(g_att)
        g_pred = None if g_pred is None else _c(g_pred)
        gl = torch.empty_like(logits) if ctx.needs_input_grad[0] else None
        gs = torch.empty_like(strength) if ctx.needs_input_grad[1] else None
        with torch.cuda.device(logits.device):
            call("ss_topk_candidates_bwd", ptr(logits), ptr(strength), ptr(samples), ptr(g_att), ptr(g_pred), ptr(gl), ptr(gs), B, rng[0], D, H, W, k)
        return gl, gs, None, None


def attention_tail_applies(cost_att, rng, H, W, left=None, right=None, k=None):
    """The fused tail under autograd: same shape conditions as the inference kernels (exact 2x up-sampling, D <= 128), fp32 HIP
    feature maps of the output's size, and a candidate count the selection kernel is built for (anything else -- half
    tensors under autocast, k > D -- takes the PyTorch statements instead of surfacing as a C-ABI error)."""
    from . import ops
    for t in (left, right):
        if t is not None and not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.dim() == 4
                                  and tuple(t.shape[-2:]) == (H, W)):
            return False
    if k is not None and not (k in (6, 24, 32) and k <= rng[1]):
        return False
    return (_on(cost_att) and ops.upsample_softmax_regression_applies(cost_att, rng[1] // 2 if rng[0] else rng[1], H, W, _range=rng)
            and rng[1] <= ops.TOPK_CANDIDATES_MAX_D)


def attention_tail(cost_att, left, right, gamma, beta, rng, H, W, k):
    """models/SemStereo.py:279-310 with autograd on: three HIP forward launches, three HIP backward launches.
    -> (att_topk [B,1,k,H,W], samples [B,k,H,W], pred_att [B,H,W], pred0 [B,H,W])."""
    _count("hip_train")
    att_weights, pred0, var = _UpsampleSoftmaxRegression.apply(cost_att, H, W, tuple(rng))
    strength = _SampleStrength.apply(left, right, pred0, var, gamma, beta)
    att_topk, samples, pred_att = _TopkCandidates.apply(att_weights, strength, k, tuple(rng))
    return att_topk, samples, pred_att, pred0


# ---- the sparse concat volume (models/SemStereo.py:316-318) -------------------------------------------------------------------------

CONCAT_VOLUME_FUSED = os.environ.get("SS_TRAIN_CONCAT_FUSED", "1") != "0"


class _ConcatVolumeSampled(torch.autograd.Function):
    """att_topk * cat(left broadcast over the candidates, warp(right) at the candidates) -- concat_volume_generator + the multiply of
    models/SemStereo.py:241-244, 316-318 -- as one forward launch (ss_concat_sampled_fwd, the inference kernel) and one backward launch
    (ss_concat_sampled_bwd): gradients to left, right and att; the candidates are indices (models/SemStereo.py:299-305: no gradient)."""

    @staticmethod
    def forward(ctx, left, right, samples, att, margin):
        left, right, samples = _c(left), _c(right), _c(samples)
        B, C, H, W = right.shape
        nd = samples.shape[1]
        att4 = _c(att.reshape(B, nd, H, W))
        out = torch.empty((B, 2 * C, nd, H, W), dtype=torch.float32, device=right.device)
        with torch.cuda.device(right.device):
            call("ss_concat_sampled_fwd", ptr(left), ptr(right), ptr(samples), ptr(att4), ptr(out), B, C, H, W, nd)
        ctx.save_for_backward(left, right, samples, att4)
        ctx.att_shape, ctx.margin = tuple(att.shape), int(margin)
        return out

    @staticmethod
    def backward(ctx, g):
        left, right, samples, att4 = ctx.saved_tensors
        g = _c(g)
        B, C, H, W = right.shape
        nd = samples.shape[1]
        gl = torch.empty_like(left) if ctx.needs_input_grad[0] else None
        gr = torch.empty_like(right) if ctx.needs_input_grad[1] else None
        ga = torch.empty_like(att4) if ctx.needs_input_grad[3] else None
        with torch.cuda.device(right.device):
            call("ss_concat_sampled_bwd", ptr(g), ptr(left), ptr(right), ptr(samples), ptr(att4), ptr(gl), ptr(gr), ptr(ga), B, C, H, W, nd, ctx.margin)
        return gl, gr, None, None if ga is None else ga.reshape(ctx.att_shape), None


def concat_volume_applies(left, right, samples, att):
    """The one-launch form under autograd: fp32 HIP maps whose rows are whole 64-pixel blocks (every quarter-resolution width of the
    path), candidates that need no gradient, one gate per (candidate, pixel)."""
    if not (CONCAT_VOLUME_FUSED and _on(left) and _on(right) and _on(samples) and _on(att)):
        return False
    if samples.requires_grad or left.dim() != 4 or left.shape != right.shape or samples.dim() != 4:
        return False
    B, C, H, W = right.shape
    nd = samples.shape[1]
    return (W % 64 == 0 and tuple(samples.shape) == (B, nd, H, W) and att.numel() == B * nd * H * W and att.shape[0] == B
            and tuple(att.shape[-3:]) == (nd, H, W) and (B * H * W) // 64 < 2 ** 31)


def concat_volume_sampled(left, right, samples, att, margin=64):
    """models/SemStereo.py:316-318 under autograd -> [B, 2C, nd, H, W].  `margin`: the |candidate| the backward's LDS windows cover
    (maxdisp / 4 of the caller: a hint, larger shifts are still summed, one atomic at a time)."""
    _count("hip_train")
    return _ConcatVolumeSampled.apply(left, right, samples, att, margin)
