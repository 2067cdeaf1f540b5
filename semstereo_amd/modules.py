"""nn.Module replacements for the 3-D aggregation stack on the hot path, with the SAME
attribute structure and state_dict keys/shapes as the reference, so its checkpoints load:

  convbn_3d          models/submodule_other.py:845-848
  attention_block    models/submodule_other.py:790-837
  BasicConv          models/submodule.py:89-116
  hourglass          models/SemStereo.py:106-143      (window (4,4,4))
  hourglass2         models/SemStereo.py:145-182      (window (6,4,4))
  channelAtt         models/SemStereo.py:89-103
  Classifier         the nn.Sequential(convbn_3d, ReLU, Conv3d) heads, models/SemStereo.py:228-234
  DepthwisePatch     the `patch` nn.Conv3d, models/SemStereo.py:219

The torch layers inside (nn.Conv3d, nn.BatchNorm3d, ...) are parameter containers: in inference
(`eval()` and no autograd) forward() runs the gfx950 kernels with BatchNorm folded into the
accumulator epilogue.  In training (batch statistics, autograd) forward() uses the stock PyTorch
layers on the GPU -- the HIP stack is an inference path; PATH_COUNTS records which one ran.
`X.adopt(ref_module)` wraps an instance built by the REFERENCE's own classes, sharing its
parameters (this is what `semstereo_amd.install.accelerate` uses).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib, ops
from . import deferred as dfr
from . import engine as E
from . import train as T
from ._lib import call, ptr
# the functional layer (engine.py) and the training convolutions (train_layers.py) under their historical names: `modules.X`
# keeps resolving for every function; the SWITCHES are engine.py's (see __getattr__ at the end of this file)
from .engine import (PATH_COUNTS, _cache, _conv_geometry, _inference, _is_plain_3x3, _ParamCache, _ReplicaCache,  # noqa: F401
                     classifier_cl_hip, classifier_fused_applies, classifier_fused_hip, pack_classifier_head_weight, conv2d_bf16s_hip, conv3d_bf16s_hip, conv3d_head_bf16s_hip, conv3d_hip,
                     conv3d_pointwise_bf16s_hip, deconv3d_bf16s_hip, deconv3d_hip, fold_bn, pack_conv2d_weight_bf16s,
                     pack_conv_weight, pack_conv_weight_bf16s, pack_deconv_weight_bf16s, pack_head_weight_bf16s,
                     pack_pointwise_weight_bf16s, run_conv2d, run_conv2d_pair, run_convbn, stem_broadcast_half, stem_of_broadcast_and_volume,
                     stem_gather_applies, stem_gather_half, stem_presplit_applies, stem_volume_half, stem_volume_half_presplit, _aux_nterms, _deconv_nterms,
                     _head_nterms, _tiled_nterms)
from .train_layers import (_conv_k3_forward, _deconv_k3_forward, conv3d_train, conv3d_wgrad_hip, deconv3d_train)  # noqa: F401

# --------------------------------------------------------------------------------------
# building blocks with the reference's names
# --------------------------------------------------------------------------------------

class _ConvBN3d(nn.Sequential):
    """nn.Sequential(Conv3d(bias=False), BatchNorm3d): keys `0.weight`, `1.*`."""

    def forward(self, x):
        x = dfr.real(x)
        if _inference(self, x):
            PATH_COUNTS["hip"] += 1
            return run_convbn(self, "cb", self[0], self[1], x, relu=False)
        return self.train_forward(x)

    def train_forward(self, x, relu=False):
        """conv -> BatchNorm with batch statistics [-> ReLU]: HIP forward / dgrad / wgrad kernels (train.py); each helper counts
        the path it took."""
        return T.batchnorm_train(self[1], conv3d_train(self[0], x), relu=relu)


def convbn_3d(in_planes, out_planes, kernel_size, stride, pad):
    """Same factory signature as the reference's convbn_3d."""
    return _ConvBN3d(nn.Conv3d(in_planes, out_planes, kernel_size=kernel_size, padding=pad, stride=stride, bias=False),
                     nn.BatchNorm3d(out_planes))


class BasicConv(nn.Module):
    """conv (no bias) -> BN -> ReLU, 2-D or 3-D, optional deconv; keys `conv.weight`, `bn.*`.
    Only the 3-D, non-transposed form is on the hot path (concat_stem) and runs the HIP kernel; the
    2-D forms (channelAtt, concat_feature) stay on PyTorch/MIOpen as in the reference."""

    def __init__(self, in_channels, out_channels, deconv=False, is_3d=False, bn=True, relu=True, **kwargs):
        super().__init__()
        self.relu = relu
        self.use_bn = bn
        self.is_3d = is_3d
        self.deconv = deconv
        if is_3d:
            layer = nn.ConvTranspose3d if deconv else nn.Conv3d
            self.conv = layer(in_channels, out_channels, bias=False, **kwargs)
            self.bn = nn.BatchNorm3d(out_channels)
        else:
            layer = nn.ConvTranspose2d if deconv else nn.Conv2d
            self.conv = layer(in_channels, out_channels, bias=False, **kwargs)
            self.bn = nn.BatchNorm2d(out_channels)

    @classmethod
    def adopt(cls, ref):
        self = cls.__new__(cls)
        nn.Module.__init__(self)
        self.relu, self.use_bn = ref.relu, ref.use_bn
        self.conv, self.bn = ref.conv, ref.bn
        self.is_3d = isinstance(ref.conv, (nn.Conv3d, nn.ConvTranspose3d))
        self.deconv = isinstance(ref.conv, (nn.ConvTranspose2d, nn.ConvTranspose3d))
        self.train(ref.training)
        return self

    def forward(self, x, gate_logits=None, gate=None):
        """gate_logits [B,Cout,H,W] (3-D form only): fuses the channelAtt gate that follows `concat_stem`
        in the model (models/SemStereo.py:319-320) into the conv epilogue; `gate`: its sigmoid, already computed."""
        if isinstance(x, dfr.Deferred):
            if self.is_3d and not self.deconv and not x.done and gate_logits is None and gate is None and dfr.on(self):
                # concat_stem(att_topk * concat_volume), :319: the volume is still an expression and the channelAtt gate of
                # :320 comes next -- hand the pair on (deferred.stem_of runs the by-halves kernels)
                return dfr.Deferred.call("stem", lambda mod, v: mod(v), self, x)
            x = x.value()
        if self.is_3d and not self.deconv and _inference(self, x, gate_logits, gate):
            PATH_COUNTS["hip"] += 1
            g = gate if gate is not None else (None if gate_logits is None else torch.sigmoid(gate_logits).contiguous())
            return run_convbn(self, "bc", self.conv, self.bn if self.use_bn else None, x, relu=bool(self.relu), gate=g)
        if gate is not None:
            assert gate_logits is None
        if not self.is_3d and not self.deconv and gate_logits is None and gate is None and _inference(self, x):
            y = run_conv2d(self, "bc2d", self.conv, self.bn if self.use_bn else None, x, bool(self.relu))
            if y is not None:
                return y
        # training / autograd: HIP autograd functions (train.py) where they apply, each counting the path it took
        if self.deconv:
            PATH_COUNTS["torch"] += 1
            x = self.conv(x)
        elif self.is_3d or T.is_k1(self.conv):
            x = conv3d_train(self.conv, x)
        else:
            x = T.conv2d_k3(self.conv, x)
        if self.use_bn:
            x = T.batchnorm_train(self.bn, x, relu=bool(self.relu))      # (train(): batch statistics; eval() under autograd: the running ones, r05)
        elif self.relu:
            x = F.relu(x)
        if gate_logits is not None:
            x = T.channel_gate(gate_logits, x)
        if gate is not None:
            x = gate.unsqueeze(2) * x
        return x


class ConcatFeature(nn.Sequential):
    """`concat_feature` (models/SemStereo.py:221-223): nn.Sequential(BasicConv(C, C/2, 3x3) , Conv2d(C/2, C/4, 3x3, bias=False));
    keys `0.conv.weight`, `0.bn.*`, `1.weight`.  Inference: both layers on the 2-D form of the tiled conv kernel."""

    def __init__(self, channels):
        super().__init__(BasicConv(channels, channels // 2, kernel_size=3, stride=1, padding=1),
                         nn.Conv2d(channels // 2, channels // 4, 3, 1, 1, bias=False))

    @classmethod
    def adopt(cls, ref):
        assert len(ref) == 2 and isinstance(ref[1], nn.Conv2d)
        self = cls.__new__(cls)
        first = ref[0] if isinstance(ref[0], BasicConv) else BasicConv.adopt(ref[0])
        nn.Sequential.__init__(self, first, ref[1])
        self.train(ref.training)
        return self

    def forward(self, x):
        x = dfr.real(x)
        if dfr.on(self, x) and isinstance(x, torch.Tensor) and x.dim() == 4:
            # the reference applies this module to the left view, then to the right one (models/SemStereo.py:314-315), and needs
            # neither before the concat volume: a deferred handle each, so that both views go through ONE pair of launches when the
            # first value is asked for (deferred.py: rule "cfeat"; HotSegment's own composition does the same directly)
            node = dfr.Deferred.call("cfeat", self._forward_now, x)
            node.info["module"] = self
            pend = [r for r in self.__dict__.get("_ss_pending_cf", []) if r() is not None and not r().done]
            pend.append(__import__("weakref").ref(node))
            self.__dict__["_ss_pending_cf"] = pend
            return node
        return self._forward_now(x)

    def _forward_now(self, x):
        x = dfr.real(x)
        a, b = self[0], self[1]
        if (isinstance(getattr(a, "conv", None), nn.Conv2d) and getattr(a, "relu", False) and not getattr(a, "deconv", False)
                and _inference(self, x)):
            y = run_conv2d(a, "bc2d", a.conv, a.bn if getattr(a, "use_bn", True) else None, x, True)
            if y is not None:
                z = run_conv2d(self, "cf1", b, None, y, False)
                return z if z is not None else b(y)
        if _inference(self, x):
            return super().forward(x)
        return T.conv2d_k3(b, a(x))                               # training: BasicConv's own training path, then the 3x3 Conv2d



class attention_block(nn.Module):
    """Windowed multi-head self-attention + 1x1x1 conv; keys `qkv_3d.*`, `final1x1.*`."""

    def __init__(self, channels_3d, num_heads=8, block=4):
        super().__init__()
        self.block = block
        self.dim_3d = channels_3d
        self.num_heads = num_heads
        self.scale_3d = (channels_3d // num_heads) ** -0.5
        self.qkv_3d = nn.Linear(channels_3d, channels_3d * 3, bias=True)
        self.final1x1 = nn.Conv3d(channels_3d, channels_3d, 1)

    @classmethod
    def adopt(cls, ref):
        self = cls.__new__(cls)
        nn.Module.__init__(self)
        self.block, self.dim_3d, self.num_heads, self.scale_3d = ref.block, ref.dim_3d, ref.num_heads, ref.scale_3d
        self.qkv_3d, self.final1x1 = ref.qkv_3d, ref.final1x1
        self.train(ref.training)
        return self

    def _params(self):
        srcs = [self.qkv_3d.weight, self.qkv_3d.bias, self.final1x1.weight, self.final1x1.bias]

        def build():
            C = self.dim_3d
            return (self.qkv_3d.weight.detach().float().t().contiguous(),            # [C][3C]
                    self.qkv_3d.bias.detach().float().contiguous(),
                    self.final1x1.weight.detach().float().reshape(C, C).t().contiguous(),  # [Cin][Cout]
                    self.final1x1.bias.detach().float().contiguous())
        return _cache(self).get("attn", srcs, build)

    def _params_split(self):
        srcs = [self.qkv_3d.weight, self.qkv_3d.bias, self.final1x1.weight, self.final1x1.bias]

        bf = E.CONV_ENGINE != "f32" and self.dim_3d in (32, 64, 128)

        def build():
            C = self.dim_3d
            pack = pack_pointwise_weight_bf16s if bf else pack_conv_weight
            return (pack(self.qkv_3d.weight.detach().float().reshape(3 * C, C, 1, 1, 1)),
                    self.qkv_3d.bias.detach().float().contiguous(),
                    pack(self.final1x1.weight.detach().float()),
                    self.final1x1.bias.detach().float().contiguous())
        return _cache(self).get("attn_split/" + ("bf16s" if bf else "f32"), srcs, build) + (bf,)

    def hip_supported(self):
        """What window_attention.hip is built for: the reference's own configuration (models/SemStereo.py:118,157:
        128 channels, 16 heads, windows of 4x4x4 = 64 or 6x4x4 = 96 tokens)."""
        bd, bh, bw = self.block
        return self.dim_3d == 128 and self.num_heads == 16 and bd * bh * bw in (64, 96)

    def forward(self, x):
        # another head count / width / window than the reference's: the same computation as PyTorch ops on the GPU
        # (visible in PATH_COUNTS["torch"]) instead of an SS_ERR_UNSUPPORTED from deep inside forward()
        x = dfr.real(x)
        if _inference(self, x) and self.hip_supported():
            PATH_COUNTS["hip"] += 1
            x = x if x.is_contiguous() else x.contiguous()
            dev = _lib.require_device(x)
            B, C, D, H, W = x.shape
            assert C == self.dim_3d and D % self.block[0] == 0
            if E.ATTENTION_FORM == "split":
                # projection -> per-(window, 4 heads) attention -> projection: three launches that each fill
                # the chip at batch 1 (the fused kernel has one workgroup per window)
                wq, bq, wo, bo, bf = self._params_split()
                nterms = _aux_nterms()
                if bf:
                    qkv = conv3d_pointwise_bf16s_hip(x, wq, 3 * C, None, bq, False, nterms)
                else:
                    qkv = conv3d_hip(x, wq, None, bq, 1, 1, False)
                y = torch.empty_like(x)
                with torch.cuda.device(dev):
                    call("ss_window_attention_core_fwd", ptr(qkv), ptr(bq), ptr(y), B, C, D, H, W, self.num_heads,
                         self.block[0], self.block[1], self.block[2])
                if bf:
                    return conv3d_pointwise_bf16s_hip(y, wo, C, None, bo, False, nterms)
                return conv3d_hip(y, wo, None, bo, 1, 1, False)
            wq, bq, wo, bo = self._params()
            out = torch.empty_like(x)
            with torch.cuda.device(dev):
                call("ss_window_attention_fwd", ptr(x), ptr(wq), ptr(bq), ptr(wo), ptr(bo), ptr(out),
                     B, C, D, H, W, self.num_heads, self.block[0], self.block[1], self.block[2])
            return out
        if not _inference(self, x) and self.hip_supported() and T.window_attention_applies(x, self.num_heads, self.block):
            return T.window_attention(x, self.qkv_3d, self.final1x1, self.num_heads, self.block)
        PATH_COUNTS["torch"] += 1
        return self._forward_torch(x)

    def _forward_torch(self, x):
        """Training path: the same computation with autograd-visible PyTorch ops."""
        B, C, D, H0, W0 = x.shape
        bd, bh, bw = self.block
        pad_r, pad_b = (bw - W0 % bw) % bw, (bh - H0 % bh) % bh
        x = F.pad(x, (0, pad_r, 0, pad_b))
        H, W = H0 + pad_b, W0 + pad_r
        nd, nh, nw = D // bd, H // bh, W // bw
        T, hd = bd * bh * bw, C // self.num_heads
        tok = x.reshape(B, C, nd, bd, nh, bh, nw, bw).permute(0, 2, 4, 6, 3, 5, 7, 1).reshape(B, nd * nh * nw, T, C)
        qkv = self.qkv_3d(tok).reshape(B, nd * nh * nw, T, 3, self.num_heads, hd).permute(3, 0, 1, 4, 2, 5)
        logits = (qkv[0] @ qkv[1].transpose(-2, -1)) * self.scale_3d
        if pad_r > 0 and pad_b > 0:      # see window_attention.hip for why both are required
            hh = torch.arange(H, device=x.device).reshape(H, 1) >= H0
            wwf = torch.arange(W, device=x.device).reshape(1, W) >= W0
            flag = (hh | wwf).reshape(nh, bh, nw, bw).permute(0, 2, 1, 3).reshape(nh * nw, bh * bw)
            differs = (flag.unsqueeze(1) != flag.unsqueeze(2)).to(x.dtype) * -1000.0
            logits = logits + differs.repeat(nd, bd, bd).reshape(1, nd * nh * nw, 1, T, T)
        y = torch.softmax(logits, dim=-1) @ qkv[2]
        y = y.reshape(B, nd, nh, nw, self.num_heads, bd, bh, bw, hd).permute(0, 4, 8, 1, 5, 2, 6, 3, 7)
        y = y.reshape(B, C, D, H, W)[:, :, :, :H0, :W0]
        return self.final1x1(y)


class hourglass(nn.Module):
    """Two stride-2 conv stages, windowed attention, two transposed-conv stages with 1x1x1 skips."""
    BLOCK = (4, 4, 4)

    def __init__(self, in_channels):
        super().__init__()
        c = in_channels
        self.conv1 = nn.Sequential(convbn_3d(c, c * 2, 3, 2, 1), nn.ReLU(inplace=True))
        self.conv2 = nn.Sequential(convbn_3d(c * 2, c * 2, 3, 1, 1), nn.ReLU(inplace=True))
        self.conv3 = nn.Sequential(convbn_3d(c * 2, c * 4, 3, 2, 1), nn.ReLU(inplace=True))
        self.conv4 = nn.Sequential(convbn_3d(c * 4, c * 4, 3, 1, 1), nn.ReLU(inplace=True))
        self.attention_block = attention_block(channels_3d=c * 4, num_heads=16, block=self.BLOCK)
        self.conv5 = nn.Sequential(
            nn.ConvTranspose3d(c * 4, c * 2, 3, padding=1, output_padding=1, stride=2, bias=False), nn.BatchNorm3d(c * 2))
        self.conv6 = nn.Sequential(
            nn.ConvTranspose3d(c * 2, c, 3, padding=1, output_padding=1, stride=2, bias=False), nn.BatchNorm3d(c))
        self.redir1 = convbn_3d(c, c, kernel_size=1, stride=1, pad=0)
        self.redir2 = convbn_3d(c * 2, c * 2, kernel_size=1, stride=1, pad=0)

    @classmethod
    def adopt(cls, ref):
        self = cls.__new__(cls)
        nn.Module.__init__(self)
        for name, child in ref.named_children():       # registration order = state_dict key order
            setattr(self, name, attention_block.adopt(child) if name == "attention_block" else child)
        self.BLOCK = tuple(self.attention_block.block)
        self.train(ref.training)
        return self

    def _up_params(self, key, deconv_seq, redir_seq):
        """Transposed conv + BN and the 1x1x1 skip conv + BN share one accumulator, so both BN scales
        are folded into the packed weights and the shifts are summed."""
        dc, dbn, rc, rbn = deconv_seq[0], deconv_seq[1], redir_seq[0], redir_seq[1]
        srcs = [dc.weight, dbn.weight, dbn.bias, dbn.running_mean, dbn.running_var,
                rc.weight, rbn.weight, rbn.bias, rbn.running_mean, rbn.running_var]

        def build():
            ds, db = fold_bn(dbn)
            rs, rb = fold_bn(rbn)
            wd = pack_conv_weight(dc.weight, transposed=True) * ds.reshape(1, 1, -1)
            wr = pack_conv_weight(rc.weight).reshape(rc.weight.shape[1], rc.weight.shape[0]) * rs.reshape(1, -1)
            wd, wr = wd.contiguous(), wr.contiguous()
            return (wd, wr, (db + rb).contiguous(), pack_deconv_weight_bf16s(wd, _deconv_nterms()) if E.CONV_ENGINE != "f32" else None,
                    pack_deconv_weight_bf16s(wr) if E.CONV_ENGINE != "f32" else None)
        return _cache(self).get(key + "/" + E.CONV_ENGINE, srcs, build)

    def _up(self, key, deconv_seq, redir_seq, x, skip):
        wd, wr, shift, wds, wrs = self._up_params(key, deconv_seq, redir_seq)
        B, _, D, H, W = x.shape
        workgroups = B * D * ((H + 3) // 4) * ((W + 31) // 32) * ((wd.shape[2] + 31) // 32)
        # layers with few workgroups: the exact-fp32 kernel's even/odd-plane split doubles them (see DECONV_MIN_WORKGROUPS)
        if E.CONV_ENGINE != "f32" and E.DECONV_BF16S and workgroups >= E.DECONV_MIN_WORKGROUPS:
            return deconv3d_bf16s_hip(x, wds, wd.shape[2], shift, True, _deconv_nterms(), skip, wrs)
        return deconv3d_hip(x, wd, shift, relu=True, skip=skip, skip_wpack=wr)

    def forward(self, x):
        x = dfr.real(x)
        if not _inference(self, x):
            def cbr(seq, t):                       # nn.Sequential(convbn_3d, ReLU): conv -> BatchNorm (batch statistics) -> ReLU fused
                cb = seq[0]
                return cb.train_forward(t, relu=True) if isinstance(cb, _ConvBN3d) else seq(t)
            conv1 = cbr(self.conv1, x)
            conv2 = cbr(self.conv2, conv1)
            conv3 = cbr(self.conv3, conv2)
            conv4 = self.attention_block(cbr(self.conv4, conv3))
            # F.relu(self.conv5(conv4) + self.redir2(conv2)), models/SemStereo.py:141-142: the add and the ReLU inside the
            # BatchNorm apply of the transposed conv (forward and backward), not two PyTorch element-wise kernels
            def up(seq, t, skip):
                return T.batchnorm_train(seq[1], deconv3d_train(seq[0], t), relu=True, residual=skip)
            conv5 = up(self.conv5, conv4, self.redir2(conv2))
            return up(self.conv6, conv5, self.redir1(x))
        PATH_COUNTS["hip"] += 1
        c1 = run_convbn(self, "c1", self.conv1[0][0], self.conv1[0][1], x, relu=True)
        hook = self.__dict__.get("_mid_hook")            # (segment.run_segment: where the second stream's work is released)

        def at(point):
            if hook is not None:
                hook[1](point)
        c2 = run_convbn(self, "c2", self.conv2[0][0], self.conv2[0][1], c1, relu=True)
        at("c2")
        c3 = run_convbn(self, "c3", self.conv3[0][0], self.conv3[0][1], c2, relu=True)
        at("c3")
        c4 = run_convbn(self, "c4", self.conv4[0][0], self.conv4[0][1], c3, relu=True)
        at("c4")
        c4 = self.attention_block(c4)
        at("att")
        c5 = self._up("u5", self.conv5, self.redir2, c4, c2)
        at("u5")
        return self._up("u6", self.conv6, self.redir1, c5, x)


class hourglass2(hourglass):
    """models/SemStereo.py:145-182: identical topology, attention window (6,4,4)."""
    BLOCK = (6, 4, 4)


class Classifier(nn.Sequential):
    """nn.Sequential(convbn_3d(c,c,3,1,1), ReLU, Conv3d(c,1,3,p1,bias=False)): keys `0.0.weight`,
    `0.1.*`, `2.weight` (the `classif` / `classif_att_` heads)."""

    def __init__(self, channels=32):
        super().__init__(convbn_3d(channels, channels, 3, 1, 1), nn.ReLU(inplace=True),
                         nn.Conv3d(channels, 1, kernel_size=3, padding=1, stride=1, bias=False))

    @classmethod
    def adopt(cls, ref):
        self = cls.__new__(cls)
        nn.Sequential.__init__(self, ref[0], ref[1], ref[2])
        self.train(ref.training)
        return self

    def patches(self, x):
        """The one-pass form WITHOUT its patch sum (engine.PatchedCost, for ops.regression_topk_patched), or None where that form does
        not serve this input / engine / switch setting -- the caller then takes forward()."""
        x = dfr.real(x)
        c0, bn0, c2 = self[0][0], self[0][1], self[2]
        if not (_inference(self, x) and E.CLASSIFIER_FOLD and E.CLASSIFIER_FUSED and E.CLASSIFIER_CL and E.CONV_ENGINE != "f32"
                and c0.in_channels == c0.out_channels == c2.in_channels == 32 and _conv_geometry(c0) == (3, 1)
                and _conv_geometry(c2) == (3, 1) and c2.out_channels == 1 and classifier_fused_applies(x, _tiled_nterms())):
            return None
        nt0 = _tiled_nterms()

        def build_fused():
            sc, sh = fold_bn(bn0)
            return pack_conv_weight_bf16s(c0.weight, nt0), sc, sh, pack_classifier_head_weight(c2.weight)
        srcs = [c0.weight, bn0.weight, bn0.bias, bn0.running_mean, bn0.running_var, c2.weight]
        ws0, sc, sh, hw = _cache(self).get("fused/%d" % nt0, srcs, build_fused)
        PATH_COUNTS["hip"] += 1
        return classifier_fused_hip(x, ws0, sc, sh, nt0, hw, sum_patches=False)

    def forward(self, x):
        x = dfr.real(x)
        if dfr.on(self, x):
            # a handle around the result: when the caller up-samples, soft-maxes and regresses it next (:279-285), those
            # statements and this head's output meet in one kernel (deferred.regression_of); any other use sees the tensor
            with dfr.suspended():
                return dfr.Deferred.leaf(self.forward(x), role="cost")
        if _inference(self, x):
            PATH_COUNTS["hip"] += 1
            c0, bn0, c2 = self[0][0], self[0][1], self[2]
            if (E.CLASSIFIER_CL and E.CONV_ENGINE != "f32" and c0.in_channels == c0.out_channels == c2.in_channels == 32
                    and _conv_geometry(c0) == (3, 1) and _conv_geometry(c2) == (3, 1) and c2.out_channels == 1):
                nt0, nt2 = _tiled_nterms(), _head_nterms()
                if E.CLASSIFIER_FUSED and classifier_fused_applies(x, nt0):

                    def build_fused():
                        sc, sh = fold_bn(bn0)
                        return pack_conv_weight_bf16s(c0.weight, nt0), sc, sh, pack_classifier_head_weight(c2.weight)
                    srcs = [c0.weight, bn0.weight, bn0.bias, bn0.running_mean, bn0.running_var, c2.weight]
                    ws0, sc, sh, hw = _cache(self).get("fused/%d" % nt0, srcs, build_fused)
                    return classifier_fused_hip(x, ws0, sc, sh, nt0, hw)

                def build():
                    sc, sh = fold_bn(bn0)
                    return pack_conv_weight_bf16s(c0.weight, nt0), sc, sh, pack_head_weight_bf16s(c2.weight, nt2)
                srcs = [c0.weight, bn0.weight, bn0.bias, bn0.running_mean, bn0.running_var, c2.weight]
                ws0, sc, sh, ws2 = _cache(self).get("cl/%d/%d" % (nt0, nt2), srcs, build)
                return classifier_cl_hip(x, ws0, sc, sh, nt0, ws2, nt2)
            y = run_convbn(self, "h0", c0, bn0, x, relu=True)
            return run_convbn(self, "h2", c2, None, y, relu=False)
        cb = self[0]
        if isinstance(cb, _ConvBN3d) and cb[1].training:
            return conv3d_train(self[2], cb.train_forward(x, relu=True))            # conv -> BN (batch statistics) -> ReLU -> 32 -> 1 conv
        return conv3d_train(self[2], F.relu(cb(x)))


_PROP_TAPS = ((-1, -1), (0, 0), (1, 1), (1, -1), (-1, 1))    # models/submodule.py:295-300 / 367-372


def propagation(x):
    """Propagation.forward (models/submodule.py:290-307): [B,1,H,W] -> [B,5,H,W], the five diagonal
    neighbours with replicate padding (the reference's one-hot 3x3 convolution, as plain shifts: exact)."""
    H, W = x.shape[-2:]
    p = F.pad(x, (1, 1, 1, 1), mode="replicate")
    return torch.cat([p[..., 1 + dy:1 + dy + H, 1 + dx:1 + dx + W] for dy, dx in _PROP_TAPS], dim=1)


def propagation_prob(v):
    """Propagation_prob.forward (models/submodule.py:361-377): [B,1,D,H,W] -> [B,5,D,H,W]."""
    H, W = v.shape[-2:]
    p = F.pad(v, (1, 1, 1, 1, 0, 0), mode="replicate")
    return torch.cat([p[..., 1 + dy:1 + dy + H, 1 + dx:1 + dx + W] for dy, dx in _PROP_TAPS], dim=1)


class Propagation(nn.Module):
    """Twin of the reference's parameter-free Propagation (a conv2d with a one-hot [5,1,3,3] filter built on
    every call): five shifted views, bit-identical for finite inputs, differentiable."""

    @classmethod
    def adopt(cls, ref):
        return cls().train(ref.training)

    def forward(self, disparity_samples):
        if dfr.on(None, disparity_samples):         # :288-289: recorded; deferred._match_strength recognises the probe they feed
            return dfr.Deferred.call("propagation", propagation, disparity_samples)
        return propagation(dfr.real(disparity_samples))


class Propagation_prob(nn.Module):
    """Twin of Propagation_prob (a conv3d with a one-hot [5,1,1,3,3] filter over the whole volume)."""

    @classmethod
    def adopt(cls, ref):
        return cls().train(ref.training)

    def forward(self, prob_volume):
        if dfr.on(None, prob_volume):               # :295: recorded; deferred._match_selected_indices recognises the selection
            return dfr.Deferred.call("propagation_prob", propagation_prob, prob_volume)
        return propagation_prob(dfr.real(prob_volume))


class DepthwisePatch(nn.Conv3d):
    """`patch`: depthwise Conv3d kernel (1,3,3), pad (0,1,1), no bias; key `weight` [C,1,1,3,3]."""

    def __init__(self, channels):
        super().__init__(channels, channels, kernel_size=(1, 3, 3), stride=1, dilation=1, groups=channels,
                         padding=(0, 1, 1), bias=False)

    @classmethod
    def adopt(cls, ref):
        assert ref.kernel_size == (1, 3, 3) and ref.groups == ref.in_channels and ref.bias is None
        self = cls.__new__(cls)
        self.__dict__.update(ref.__dict__)          # same Parameter objects, same hyper-parameters
        return self

    def forward(self, x, gate_logits=None):
        """gate_logits [B,C,H,W]: fuses the channelAtt gate that follows `patch` in the model."""
        if isinstance(x, dfr.Deferred):
            if x.op == "gwc_norm" and not x.done and gate_logits is None and dfr.on(self):
                return dfr.Deferred.call("patch", lambda mod, v: mod(v), self, x)      # :274; the gate of :276 decides the kernel
            x = x.value()
        if _inference(self, x, gate_logits):
            PATH_COUNTS["hip"] += 1
            x = x if x.is_contiguous() else x.contiguous()
            dev = _lib.require_device(x, gate_logits)
            B, C, D, H, W = x.shape
            w = self.weight.detach()
            out = torch.empty_like(x)
            g = None if gate_logits is None else gate_logits.contiguous()
            with torch.cuda.device(dev):
                call("ss_depthwise_patch_fwd", ptr(x), ptr(w), ptr(g), ptr(out), B, C, D, H, W)
            return out
        y = T.depthwise_patch(self, x)
        return y if gate_logits is None else T.channel_gate(gate_logits, y)


class channelAtt(nn.Module):
    """sigmoid(conv1x1(BN-ReLU(conv1x1(im)))) broadcast over D, times the volume; keys `im_att.*`.
    The two 1x1 2-D convs on the image features stay on PyTorch; the volume gating is the HIP kernel."""

    def __init__(self, cv_chan, im_chan):
        super().__init__()
        self.im_att = nn.Sequential(BasicConv(im_chan, im_chan // 2, kernel_size=1, stride=1, padding=0),
                                    nn.Conv2d(im_chan // 2, cv_chan, 1))

    @classmethod
    def adopt(cls, ref):
        self = cls.__new__(cls)
        nn.Module.__init__(self)
        self.im_att = ref.im_att
        self.train(ref.training)
        return self

    def _hip_params(self):
        """im_att as ss_channel_att_logits_fwd wants it, or None when the module is not the reference's shape:
        BasicConv(1x1 Conv2d, no bias -> BatchNorm2d -> ReLU) -> 1x1 Conv2d(+bias), widths (256,128,32) / (128,64,32)."""
        a, b = self.im_att[0], self.im_att[1]
        c1 = getattr(a, "conv", None)
        ok = (len(self.im_att) == 2 and isinstance(c1, nn.Conv2d) and isinstance(b, nn.Conv2d) and getattr(a, "relu", False)
              and c1.kernel_size == (1, 1) and b.kernel_size == (1, 1) and c1.bias is None and c1.stride == (1, 1)
              and b.stride == (1, 1) and c1.groups == 1 and b.groups == 1 and c1.padding == (0, 0) and b.padding == (0, 0)
              and (c1.in_channels, c1.out_channels, b.out_channels) in ((256, 128, 32), (128, 64, 32))
              and b.in_channels == c1.out_channels)
        if not ok:
            return None
        bn = a.bn if getattr(a, "use_bn", True) else None
        srcs = [c1.weight, b.weight] + ([b.bias] if b.bias is not None else []) + \
               ([bn.weight, bn.bias, bn.running_mean, bn.running_var] if bn is not None else [])

        def build():
            sc, sh = fold_bn(bn) if bn is not None else (None, None)
            return (pack_pointwise_weight_bf16s(c1.weight), sc, sh, pack_pointwise_weight_bf16s(b.weight),
                    None if b.bias is None else b.bias.detach().float().contiguous())
        return _cache(self).get("im_att", srcs, build)

    def logits(self, im, sigmoid=False):
        """im_att(im) [B,cv_chan,H,W] (its sigmoid when `sigmoid`): one HIP launch in inference for the reference's shapes."""
        if _inference(self, im) and im.is_cuda:
            prm = self._hip_params()
            if prm is not None and im.shape[1] * im.shape[2] * im.shape[3] * 4 < 2 ** 31:     # (the kernel's 32-bit offsets per element)
                PATH_COUNTS["hip"] += 1
                w1, sc, sh, w2, b2 = prm
                im = im if im.is_contiguous() else im.contiguous()
                dev = _lib.require_device(im)
                B, Cin, H, W = im.shape
                c1, c2 = self.im_att[0].conv, self.im_att[1]
                out = torch.empty((B, c2.out_channels, H, W), dtype=im.dtype, device=im.device)
                with torch.cuda.device(dev):
                    call("ss_channel_att_logits_fwd", ptr(im), ptr(w1), ptr(sc), ptr(sh), ptr(w2), ptr(b2), ptr(out),
                         B, Cin, c1.out_channels, c2.out_channels, H, W, int(bool(sigmoid)))
                return out
        att = self.im_att(im)
        return torch.sigmoid(att) if sigmoid else att

    def forward(self, cv, im):
        im = dfr.real(im)
        if isinstance(cv, dfr.Deferred):
            # :276 after build_gwc_volume_norm + patch, or :320 after concat_stem(att_topk * concat_volume): one fused launch
            # (sequence) instead of the statements one by one; anything else: the value
            if dfr.on(self, im):
                fused = dfr.gated_volume_of(cv, self, im)
                if fused is None:
                    fused = dfr.stem_of(cv, self, im)
                if fused is not None:
                    return fused
            cv = cv.value()
        if _inference(self, cv, im):
            PATH_COUNTS["hip"] += 1
            return ops.channel_gate(self.logits(im), cv)
        a, b = self.im_att[0], self.im_att[1]
        att = conv3d_train(b, a(im)) if isinstance(b, nn.Conv2d) and T.is_k1(b) else self.im_att(im)
        return T.channel_gate(att, cv)


class SSR_upsample(nn.Module):
    """Semantic-guided refinement head (models/submodule.py:412-431): 4x bilinear up-sampling of the
    1/4-scale disparity plus a residual gated by the class probabilities.  Keys `conv.{0,1,2}.*`,
    `conv1.{0,1}.*`, `conv2.{0,1}.*`, `conv3.*`.  Inference: one HIP kernel (ss_ssr_upsample_fwd)."""

    def __init__(self, num_classes):
        super().__init__()
        n = self.num_classes = num_classes
        self.conv = nn.Sequential(nn.BatchNorm2d(1), nn.Conv2d(1, n, kernel_size=3, padding=1), nn.BatchNorm2d(n))
        self.conv1 = nn.Sequential(nn.Conv2d(n, n, kernel_size=1, padding=0), nn.BatchNorm2d(n))
        self.conv2 = nn.Sequential(nn.Conv2d(n, n, kernel_size=1, padding=0), nn.BatchNorm2d(n))
        self.conv3 = nn.Conv2d(n, 1, kernel_size=1, padding=0)

    @classmethod
    def adopt(cls, ref):
        self = cls.__new__(cls)
        nn.Module.__init__(self)
        self.num_classes = ref.num_classes
        for name, child in ref.named_children():
            setattr(self, name, child)
        self.train(ref.training)
        return self

    def _params(self):
        srcs = [t for t in list(self.parameters()) + list(self.buffers()) if t.dtype.is_floating_point]

        def build():
            # every eval-mode BatchNorm folded into the convolution before it (in float64), the two gate stages pre-multiplied by
            # -log2(e): the layout ssr_upsample.hip documents (P_BN0 ... P_B3)
            n = self.num_classes
            f64 = lambda t: t.detach().double()                                   # noqa: E731

            def fold64(bn_):
                sc = f64(bn_.weight) / torch.sqrt(f64(bn_.running_var) + bn_.eps)
                return sc, f64(bn_.bias) - f64(bn_.running_mean) * sc
            s0, t0 = fold64(self.conv[0])
            sa, ta = fold64(self.conv[2])
            s1, t1 = fold64(self.conv1[1])
            s2, t2 = fold64(self.conv2[1])
            nl2e = -1.4426950408889634
            parts = [s0, t0, sa[:, None] * f64(self.conv[1].weight).reshape(n, 9), sa * f64(self.conv[1].bias) + ta,
                     nl2e * s1[:, None] * f64(self.conv1[0].weight).reshape(n, n), nl2e * (s1 * f64(self.conv1[0].bias) + t1),
                     nl2e * s2[:, None] * f64(self.conv2[0].weight).reshape(n, n), nl2e * (s2 * f64(self.conv2[0].bias) + t2),
                     f64(self.conv3.weight).reshape(n), f64(self.conv3.bias)]
            return torch.cat([p.reshape(-1) for p in parts]).float().contiguous()
        return _cache(self).get("ssr", srcs, build)

    def forward(self, depth_low, weights, pred_label):
        # The reference calls the head twice per forward with the same `spx_pred` / `pred_label` (models/SemStereo.py:311, 324) and in
        # eval returns only the second result (:346): in inference the call hands out a deferred handle (deferred.py), so a result
        # nobody reads -- `pred_att_up` of an eval forward -- is never computed (a dead full-resolution launch per pair before r05).
        if (self.num_classes == 6 and isinstance(weights, torch.Tensor) and isinstance(pred_label, torch.Tensor)
                and dfr.on(self, *[t for t in (depth_low, weights, pred_label) if isinstance(t, torch.Tensor)])):
            dfr.STATS.setdefault("ssr", {"deferred": 0, "computed": 0})["deferred"] += 1
            return dfr.Deferred.call("ssr", self._forward_counted, depth_low, weights, pred_label)
        return self._forward_now(depth_low, weights, pred_label)

    def _forward_counted(self, depth_low, weights, pred_label):
        dfr.STATS.setdefault("ssr", {"deferred": 0, "computed": 0})["computed"] += 1
        return self._forward_now(depth_low, weights, pred_label)

    def _forward_now(self, depth_low, weights, pred_label):
        # ssr_upsample.hip is built for the reference's 6 classes (main_us3d.py:66); other counts: PyTorch ops on the GPU
        depth_low, weights, pred_label = dfr.real(depth_low), dfr.real(weights), dfr.real(pred_label)
        if _inference(self, depth_low, weights, pred_label) and self.num_classes == 6:
            PATH_COUNTS["hip"] += 1
            depth_low, weights, pred_label = [t if t.is_contiguous() else t.contiguous() for t in (depth_low, weights, pred_label)]
            dev = _lib.require_device(depth_low, weights, pred_label)
            b, c, h, w = depth_low.shape
            assert c == 1 and weights.shape == (b, self.num_classes, 4 * h, 4 * w) and pred_label.shape == weights.shape
            prm = self._params()
            assert prm.numel() == _lib.load().ss_ssr_param_count()
            out = torch.empty((b, 4 * h, 4 * w), dtype=depth_low.dtype, device=depth_low.device)
            with torch.cuda.device(dev):
                call("ss_ssr_upsample_fwd", ptr(depth_low), ptr(weights), ptr(pred_label), ptr(prm), ptr(out),
                     b, h, w, self.num_classes)
            return out
        PATH_COUNTS["torch"] += 1
        b, c, h, w = depth_low.shape
        depth_ = F.interpolate(depth_low, (h * 4, w * 4), mode="bilinear").reshape(b, 1, h * 4, w * 4)
        depth = self.conv(depth_)
        prob = self._class_gate(weights, pred_label)
        return (depth_ + self.conv3(depth * prob)).squeeze(1)

    def _class_gate(self, weights, pred_label):
        """The 6-class gate `prob` of models/submodule.py:424-428 depends on (spx_pred, pred_label) only, and a training forward asks
        for it twice with the very same tensors (models/SemStereo.py:311, 324): the first call parks it, the second takes it (one
        entry, consumed on use, so nothing outlives the forward; BatchNorm in train() then also sees the batch once per stage --
        its running statistics are updated once where the reference updates them twice with identical numbers)."""
        # (the key also carries the versions of the gate's own parameters -- an optimizer step between the two calls must not
        # hand back a gate computed with the old weights on an already-freed graph -- and the entry is parked OUTSIDE the module's
        # __dict__, in a weak map, so that copy.deepcopy / state handling of the model never meet the non-leaf tensors; an entry left
        # by an odd number of calls (att_weights_only, a direct call of the head) is replaced by the next call or dropped by
        # drop_parked_gates(), which accelerate()'s forward hook calls when the model's forward returns)
        key = (id(weights), id(pred_label), weights._version, pred_label._version, torch.is_grad_enabled(), self.training,
               tuple(p_._version for p_ in self.parameters()))
        hit = _GATE_PARKED.pop(self, None)
        if hit is not None and hit[0] == key and hit[1] is weights and hit[2] is pred_label:
            PATH_COUNTS["ssr_gate_reused"] = PATH_COUNTS.get("ssr_gate_reused", 0) + 1
            # the reference's second evaluation would have moved the running statistics once more with the same batch statistics s:
            # r1 = (1 - m) r0 + m s  =>  r2 = (1 - m) r1 + (r1 - (1 - m) r0)
            with torch.no_grad():
                for bn_, (m0, v0) in zip((self.conv1[1], self.conv2[1]), hit[4]):
                    if m0 is not None:
                        mom = bn_.momentum
                        # (through .data: F.batch_norm updates these buffers without touching their autograd version either)
                        bn_.running_mean.data.copy_((1 - mom) * bn_.running_mean + (bn_.running_mean - (1 - mom) * m0))
                        bn_.running_var.data.copy_((1 - mom) * bn_.running_var + (bn_.running_var - (1 - mom) * v0))
                        bn_.num_batches_tracked += 1
            return hit[3]
        bns = (self.conv1[1], self.conv2[1])
        tracked = [b_.training and b_.track_running_stats for b_ in bns]
        if any(tracked) and any(b_.momentum is None for b_ in bns):
            tracked = None                                     # cumulative averages: no closed form for the second update -- no reuse
        before = None if tracked is None else [(b_.running_mean.clone(), b_.running_var.clone()) if t_ else (None, None)
                                               for b_, t_ in zip(bns, tracked)]
        prob = torch.sigmoid(self.conv1(F.softmax(pred_label, dim=1) * weights))
        prob = torch.sigmoid(self.conv2(prob * weights))
        if before is not None and torch.is_grad_enabled():
            _GATE_PARKED[self] = (key, weights, pred_label, prob, before)
        return prob


_GATE_PARKED = __import__("weakref").WeakKeyDictionary()


def drop_parked_gates(model=None):
    """Forget the class gates parked by SSR_upsample._class_gate (all of them, or those of `model`'s heads): nothing parked outlives
    the forward that made it."""
    if model is None:
        _GATE_PARKED.clear()
        return
    for m_ in model.modules():
        if isinstance(m_, SSR_upsample):
            _GATE_PARKED.pop(m_, None)


def __getattr__(name):
    """`modules.CONV_ENGINE` and the other switches are engine.py's attributes: reads are forwarded (SET them on
    semstereo_amd.engine -- an assignment to `modules.X` would only create a dead attribute here, so __init__ refuses it)."""
    if name in E.SWITCHES:
        return getattr(E, name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")


class _TwinsModule(__import__("types").ModuleType):
    """An assignment `modules.CONV_ENGINE = ...` would create an attribute nothing reads: refuse it loudly."""

    def __setattr__(self, name, value):
        if name in E.SWITCHES:
            raise AttributeError(f"set semstereo_amd.engine.{name}, not semstereo_amd.modules.{name} (the switches live in engine.py)")
        super().__setattr__(name, value)


__import__("sys").modules[__name__].__class__ = _TwinsModule
