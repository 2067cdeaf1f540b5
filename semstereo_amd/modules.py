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
import os
import weakref

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib, ops
from . import deferred as dfr
from . import train as T
from ._lib import call, ptr

PATH_COUNTS = {"hip": 0, "torch": 0}


def _inference(module, *tensors):
    """True when the folded-BN HIP path is valid: eval mode and nothing needs autograd."""
    if module.training:
        return False
    if torch.is_grad_enabled():
        if any(t is not None and t.requires_grad for t in tensors):
            return False
        if any(p.requires_grad for p in module.parameters()):
            return False
        if module.__dict__.get("_is_replica", False):
            # an nn.DataParallel replica holds its weights as plain attributes (parameters() is empty);
            # torch/nn/parallel/replicate.py keeps them reachable in _former_parameters
            for m in module.modules():
                if any(p is not None and p.requires_grad for p in getattr(m, "_former_parameters", {}).values()):
                    return False
    return True


class _ParamCache:
    """Derived device tensors (packed weights, folded affines), rebuilt when a source tensor changes."""

    def __init__(self):
        self._store = {}
        self.owner = None                    # weakref to the module the cache belongs to (set by _cache)

    def get(self, key, sources, build):
        stamp = tuple((t.data_ptr(), t._version, str(t.device)) for t in sources)
        hit = self._store.get(key)
        if hit is None or hit[0] != stamp:
            with torch.no_grad():
                hit = (stamp, build())
            self._store[key] = hit
        return hit[1]


class _ReplicaCache:
    """The cache view of an nn.DataParallel replica: entries live on the ORIGINAL module (whose `_ss_cache` object the
    replica's shallow-copied __dict__ shares), keyed by the replica's device, and are valid while the original's
    parameters and buffers are unchanged -- the replica's own tensors are fresh broadcast copies on every forward that
    the caching allocator tends to hand the same address with version 0, so their (data_ptr, version) says nothing.
    Packed weights are built from the replica's device-local copies, once per device and weight update instead of once per
    forward (~50 pack launches per GPU and step)."""

    def __init__(self, shared, owner, device):
        self.shared, self.owner, self.device = shared, owner, str(device)

    def get(self, key, sources, build):
        stamp = tuple((t.data_ptr(), t._version) for t in list(self.owner.parameters()) + list(self.owner.buffers()))
        k = ("replica", self.device, key)
        hit = self.shared._store.get(k)
        if hit is None or hit[0] != stamp:
            with torch.no_grad():
                hit = (stamp, build())
            self.shared._store[k] = hit
        return hit[1]


def _cache(module):
    """Per-module cache of derived tensors (packed weights, folded affines)."""
    c = module.__dict__.get("_ss_cache")
    if module.__dict__.get("_is_replica", False):
        owner = c.owner() if c is not None and c.owner is not None else None
        if owner is None:
            return _ParamCache()              # a replica of a module that never ran on its own: nothing to validate against
        dev_ = next((t.device for t in list(module.__dict__.get("_former_parameters", {}).values()) + list(module.buffers()) if t is not None), "?")
        for m in module.modules():
            fp = [t for t in getattr(m, "_former_parameters", {}).values() if t is not None]
            if fp:
                dev_ = fp[0].device
                break
        return _ReplicaCache(c, owner, dev_)
    if c is None:
        c = module.__dict__["_ss_cache"] = _ParamCache()
        c.owner = weakref.ref(module)
    return c


def fold_bn(bn):
    """eval-mode BatchNorm as y = x*scale + shift (the same two-step form ATen's inference path uses)."""
    invstd = 1.0 / torch.sqrt(bn.running_var + bn.eps)
    scale = (bn.weight * invstd) if bn.weight is not None else invstd
    shift = (bn.bias if bn.bias is not None else 0.0) - bn.running_mean * scale
    return scale.float().contiguous(), shift.float().contiguous()


def pack_conv_weight(w, transposed=False):
    """[Cout,Cin,k,k,k] (or ConvTranspose3d's [Cin,Cout,k,k,k]) -> [Cin][k^3][Cout] on the device."""
    w = w.detach().float().contiguous()
    _lib.require_device(w)
    if transposed:
        Cin, Cout, k = w.shape[0], w.shape[1], w.shape[2]
    else:
        Cout, Cin, k = w.shape[0], w.shape[1], w.shape[2]
    assert w.shape[2] == w.shape[3] == w.shape[4], "cubic kernels only"
    out = torch.empty((Cin, k * k * k, Cout), dtype=torch.float32, device=w.device)
    with torch.cuda.device(w.device):
        call("ss_pack_conv3d_weights", ptr(w), ptr(out), Cout, Cin, k, int(transposed))
    return out


def conv3d_hip(x, wpack, scale, shift, k, stride, relu, residual=None, gate=None):
    """Conv3d(bias=False, pad=k//2) + per-channel affine + optional residual + optional ReLU + optional
    channelAtt gate (sigmoid(gate[b,co,h,w]) broadcast over D)."""
    x = x if x.is_contiguous() else x.contiguous()
    dev = _lib.require_device(x, wpack, scale, shift, residual, gate)
    B, Cin, D, H, W = x.shape
    assert wpack.shape[0] == Cin and wpack.shape[1] == k ** 3
    Cout = wpack.shape[2]
    pad = k // 2
    Do, Ho, Wo = [(n + 2 * pad - k) // stride + 1 for n in (D, H, W)]
    out = torch.empty((B, Cout, Do, Ho, Wo), dtype=x.dtype, device=x.device)
    if residual is not None:
        assert residual.shape == out.shape and residual.is_contiguous()
    if gate is not None:
        assert gate.shape == (B, Cout, Ho, Wo) and gate.is_contiguous()
    with torch.cuda.device(dev):
        call("ss_conv3d_fwd", ptr(x), ptr(wpack), ptr(scale), ptr(shift), ptr(residual), ptr(gate), ptr(out),
             B, Cin, D, H, W, Cout, k, stride, int(relu))
    return out


def deconv3d_hip(x, wpack, shift, relu, skip=None, skip_wpack=None):
    """ConvTranspose3d(k3,s2,p1,op1) [+ 1x1x1 projection of `skip`] + shift + optional ReLU.
    Per-branch BN scales are expected to be folded into the packed weights already."""
    x = x if x.is_contiguous() else x.contiguous()
    dev = _lib.require_device(x, wpack, shift, skip, skip_wpack)
    B, Cin, D, H, W = x.shape
    Cout = wpack.shape[2]
    out = torch.empty((B, Cout, 2 * D, 2 * H, 2 * W), dtype=x.dtype, device=x.device)
    Cs = 0
    if skip is not None:
        skip = skip if skip.is_contiguous() else skip.contiguous()
        Cs = skip.shape[1]
        assert skip.shape == (B, Cs, 2 * D, 2 * H, 2 * W) and skip_wpack.shape == (Cs, Cout)
    with torch.cuda.device(dev):
        call("ss_deconv3d_fwd", ptr(x), ptr(wpack), None, ptr(shift), ptr(skip), ptr(skip_wpack), None, None,
             ptr(out), B, Cin, D, H, W, Cout, Cs, int(relu))
    return out


def pack_deconv_weight_bf16s(wpack, nterms=6):
    """fp32 pack [Cin][ntaps][Cout] (ntaps 27: transposed conv, BN scale folded; or [Cs][Cout]: the skip projection)
    -> split fragments for ss_deconv3d_bf16s_fwd: three bf16 terms, or (nterms 19, main weights only) two scaled fp16
    terms + the per-channel inverse scales."""
    wpack = wpack.detach().float().contiguous()
    _lib.require_device(wpack)
    Cin, Cout = wpack.shape[0], wpack.shape[-1]
    ntaps = 1 if wpack.dim() == 2 else wpack.shape[1]
    with torch.cuda.device(wpack.device):
        if nterms == 19:
            assert ntaps == 27
            out = torch.empty(((Cin + 15) // 16) * 27 * 2 * 2 * Cout * 8 + 2 * Cout, dtype=torch.int16, device=wpack.device)
            call("ss_pack_deconv3d_weights_f16s", ptr(wpack), ptr(out), Cin, Cout)
        else:
            out = torch.empty(((Cin + 15) // 16) * ntaps * 3 * 2 * Cout * 8, dtype=torch.int16, device=wpack.device)
            call("ss_pack_deconv3d_weights_bf16s", ptr(wpack), ptr(out), Cin, Cout, ntaps)
    return out


def deconv3d_bf16s_hip(x, wsplit, Cout, shift, relu, nterms, skip=None, skip_wsplit=None):
    """deconv3d_hip on the split-bf16 engine."""
    x = x if x.is_contiguous() else x.contiguous()
    dev = _lib.require_device(x, shift, skip)
    B, Cin, D, H, W = x.shape
    out = torch.empty((B, Cout, 2 * D, 2 * H, 2 * W), dtype=x.dtype, device=x.device)
    Cs = 0
    if skip is not None:
        skip = skip if skip.is_contiguous() else skip.contiguous()
        Cs = skip.shape[1]
        assert skip.shape == (B, Cs, 2 * D, 2 * H, 2 * W)
    with torch.cuda.device(dev):
        call("ss_deconv3d_bf16s_fwd", ptr(x), ptr(wsplit), ptr(shift), ptr(skip), ptr(skip_wsplit), ptr(out),
             B, Cin, D, H, W, Cout, Cs, int(relu), int(nterms))
    return out


#: matrix-core engine of the 3x3x3 stride-1 convolutions: "f32" = exact-fp32 MFMA (conv3d.hip);
#: "bf16x6" / "bf16x3" = split-bf16 (conv3d_bf16s.hip, fp32 operands as 3 bf16 terms, 6 or 3 cross
#: products) for every 3x3x3 conv (stride 1 and 2), the transposed convs, the 32 -> 1 heads and the 1x1x1
#: projections of the attention blocks.
#: Default bf16x6: its measured error against fp64 is BELOW the exact-fp32 MFMA's (1.1e-7 vs 1.8e-7 of
#: sum|a*b|, tools/exp_split_bf16.hip) at ~1.5x its speed; SS_CONV_ENGINE=f32 selects the exact engine.
#: "f16x3": the tiled 3x3x3 convs (stride 1 and 2) on TWO fp16 terms and three products with block-floating operands
#: (same accuracy class as bf16x6, half its matrix-core time: conv3d_bf16s.hip); the other kernels stay on bf16x6.
CONV_ENGINE = os.environ.get("SS_CONV_ENGINE", "f16x3")
_NTERMS_TILED = {"bf16x6": 6, "bf16x3": 3, "f16x3": 19}          # `nterms` codes of ss_conv3d_bf16s_fwd
_NTERMS_AUX = {"bf16x6": 6, "bf16x3": 3, "f16x3": 6, "f32": 6}   # kernels without an fp16 form (f32: unused)


def _tiled_nterms():
    return _NTERMS_TILED[CONV_ENGINE]


def _aux_nterms():
    return _NTERMS_AUX[CONV_ENGINE]


DECONV_F16 = os.environ.get("SS_DECONV_F16", "1") != "0"        # f16x3 engine: the transposed convs' main loop on fp16 terms too


def _deconv_nterms():
    return _NTERMS_TILED[CONV_ENGINE] if DECONV_F16 else _NTERMS_AUX[CONV_ENGINE]
#: transposed convs with fewer workgroups than this run on the exact-fp32 kernel, whose even/odd-plane split doubles them.
#: 0 since r03: on the one layer of the bench shape below 256 workgroups (hourglass_att.conv5, 128) the split engine's kernel has
#: overtaken it (step 2.117 -> 2.093 ms, 2.119 -> 2.085 on a second box), and the engine of a layer no longer depends on the
#: batch size (r01: bf16x6 at 128 workgroups 88 vs 67 us, the fp16 form 66 vs 72 us)
DECONV_MIN_WORKGROUPS = int(os.environ.get("SS_DECONV_MIN_WGS", "0"))
DECONV_BF16S = os.environ.get("SS_DECONV_BF16S", "1") != "0"     # transposed convs on the split engine too (else exact fp32 MFMA)


def pack_conv_weight_bf16s(w, nterms=6):
    """[Cout,Cin,3,3,3] fp32 -> split fragments for ss_conv3d_bf16s_fwd (int16 tensor, 16-B aligned): three bf16 terms
    (nterms 6 / 3) or two scaled fp16 terms + the per-channel inverse scales (nterms 19)."""
    w = w.detach().float().contiguous()
    _lib.require_device(w)
    Cout, Cin = w.shape[0], w.shape[1]
    assert tuple(w.shape[2:]) == (3, 3, 3)
    with torch.cuda.device(w.device):
        if nterms == 19:
            out = torch.empty(((Cin + 7) // 8) * 14 * 2 * 2 * Cout * 8 + 2 * Cout, dtype=torch.int16, device=w.device)
            call("ss_pack_conv3d_weights_f16s", ptr(w), ptr(out), Cout, Cin)
        else:
            out = torch.empty(((Cin + 7) // 8) * 14 * 3 * 2 * Cout * 8, dtype=torch.int16, device=w.device)
            call("ss_pack_conv3d_weights_bf16s", ptr(w), ptr(out), Cout, Cin)
    return out


def conv3d_bf16s_hip(x, wsplit, Cout, scale, shift, relu, nterms, residual=None, gate=None, partial=None, stride=1):
    """3x3x3 Conv3d (stride 1 or 2) + affine + optional residual / ReLU on the split-bf16 engine.  `partial`
    [B,Cout,D,H,W]: a partial sum of the same convolution (other input channels), added BEFORE the affine."""
    x = x if x.is_contiguous() else x.contiguous()
    dev = _lib.require_device(x, scale, shift, residual, gate, partial)
    B, Cin, D, H, W = x.shape
    Do, Ho, Wo = [(n - 1) // stride + 1 for n in (D, H, W)]
    out = torch.empty((B, Cout, Do, Ho, Wo), dtype=x.dtype, device=x.device)
    if gate is not None:
        assert gate.shape == (B, Cout, Ho, Wo) and gate.is_contiguous()
    with torch.cuda.device(dev):
        if partial is not None:
            assert residual is None and stride == 1 and partial.shape == out.shape and partial.is_contiguous()
            call("ss_conv3d_bf16s_partial_fwd", ptr(x), ptr(wsplit), ptr(partial), ptr(scale), ptr(shift), ptr(gate), ptr(out),
                 B, Cin, D, H, W, Cout, int(relu), int(nterms))
        else:
            call("ss_conv3d_bf16s_fwd", ptr(x), ptr(wsplit), ptr(scale), ptr(shift), ptr(residual), ptr(gate), ptr(out),
                 B, Cin, D, H, W, Cout, int(stride), int(relu), int(nterms))
    return out


def classifier_cl_hip(x, ws0, scale0, shift0, nterms0, ws2, nterms2):
    """nn.Sequential(convbn_3d(C,C,3,1,1), ReLU, Conv3d(C,1,3,p1)) (models/SemStereo.py:228-234) as two launches whose
    intermediate is channels-last [B,D,H,W,C] (private to the pair: 16-byte stores in the first, 16-byte loads in the head)."""
    x = x if x.is_contiguous() else x.contiguous()
    dev = _lib.require_device(x, scale0, shift0)
    B, C, D, H, W = x.shape
    mid = torch.empty((B, D, H, W, C), dtype=x.dtype, device=x.device)
    out = torch.empty((B, 1, D, H, W), dtype=x.dtype, device=x.device)
    with torch.cuda.device(dev):
        call("ss_conv3d_bf16s_cl_fwd", ptr(x), ptr(ws0), ptr(scale0), ptr(shift0), ptr(mid), B, C, D, H, W, C, 1, int(nterms0))
        call("ss_conv3d_head_bf16s_cl_fwd", ptr(mid), ptr(ws2), None, None, ptr(out), B, C, D, H, W, 0, int(nterms2))
    return out


def pack_conv2d_weight_bf16s(w, nterms=6):
    """[Cout,Cin,3,3] fp32 -> split fragments for ss_conv2d_bf16s_fwd (three bf16 terms, or two scaled fp16 terms: nterms 19)."""
    w = w.detach().float().contiguous()
    _lib.require_device(w)
    Cout, Cin = w.shape[0], w.shape[1]
    assert tuple(w.shape[2:]) == (3, 3)
    with torch.cuda.device(w.device):
        if nterms == 19:
            out = torch.empty(((Cin + 7) // 8) * 5 * 2 * 2 * Cout * 8 + 2 * Cout, dtype=torch.int16, device=w.device)
            call("ss_pack_conv2d_weights_f16s", ptr(w), ptr(out), Cout, Cin)
        else:
            out = torch.empty(((Cin + 7) // 8) * 5 * 3 * 2 * Cout * 8, dtype=torch.int16, device=w.device)
            call("ss_pack_conv2d_weights_bf16s", ptr(w), ptr(out), Cout, Cin)
    return out


def conv2d_bf16s_hip(x, wsplit, Cout, scale, shift, relu, nterms, residual=None):
    """Conv2d(k3, s1, p1, bias=False) + affine + optional residual / ReLU on the split-bf16 engine."""
    x = x if x.is_contiguous() else x.contiguous()
    dev = _lib.require_device(x, scale, shift, residual)
    B, Cin, H, W = x.shape
    out = torch.empty((B, Cout, H, W), dtype=x.dtype, device=x.device)
    with torch.cuda.device(dev):
        call("ss_conv2d_bf16s_fwd", ptr(x), ptr(wsplit), ptr(scale), ptr(shift), ptr(residual), ptr(out), B, Cin, H, W, Cout,
             int(relu), int(nterms))
    return out


def _is_plain_3x3(conv):
    return (isinstance(conv, nn.Conv2d) and conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1)
            and conv.dilation == (1, 1) and conv.groups == 1 and conv.bias is None and conv.padding_mode == "zeros")


#: concat_feature's 3x3 2-D convs on the split engine instead of MIOpen (which also takes MIOpen's per-box algorithm choice
#: out of the matching branch).  They run on the second stream UNDER the attention branch and compete with it for the
#: matrix pipe: with three-term bf16 operands (110 + 38 us against 170 + 58 us of Winograd + BatchNorm + clamp) the step
#: was 1 % slower at batch 1, with the fp16 form (half the matrix-core time again) it is 2.2 % faster at batch 1 and 4.
#: "auto": on for the f16x3 engine; SS_CONV2D_HIP=0 / 1 forces it.
_c2d = os.environ.get("SS_CONV2D_HIP", "auto")
CONV2D_HIP = "auto" if _c2d == "auto" else (_c2d != "0")


def _conv2d_hip_on():
    return CONV_ENGINE == "f16x3" if CONV2D_HIP == "auto" else bool(CONV2D_HIP)


def run_conv2d(owner, key, conv, bn, x, relu):
    """Conv2d(3x3, s1, p1, no bias) [+ BN(eval)] [+ ReLU] of a 2-D map on the split-bf16 engine; None when it does not apply."""
    if not (_conv2d_hip_on() and CONV_ENGINE != "f32" and _is_plain_3x3(conv) and x.is_cuda):
        return None
    nterms = _tiled_nterms()
    srcs = [conv.weight] + ([bn.weight, bn.bias, bn.running_mean, bn.running_var] if bn is not None else [])

    def build():
        sc, sh = fold_bn(bn) if bn is not None else (None, None)
        return pack_conv2d_weight_bf16s(conv.weight, nterms), sc, sh
    ws, scale, shift = _cache(owner).get(key + "/2d_" + CONV_ENGINE, srcs, build)
    return conv2d_bf16s_hip(x, ws, conv.out_channels, scale, shift, relu, nterms)


def pack_head_weight_bf16s(w, nterms=6):
    """[1,Cin,3,3,3] fp32 -> split fragments (taps as matrix rows) for ss_conv3d_head_bf16s_fwd: three bf16 terms (nterms 6 / 3)
    or two scaled fp16 terms + the inverse scale (nterms 19)."""
    w = w.detach().float().contiguous()
    _lib.require_device(w)
    Cin = w.shape[1]
    assert w.shape[0] == 1 and tuple(w.shape[2:]) == (3, 3, 3) and Cin % 16 == 0
    with torch.cuda.device(w.device):
        if nterms == 19:
            out = torch.empty((Cin // 16) * 2 * 2 * 32 * 8 + 8, dtype=torch.int16, device=w.device)
            call("ss_pack_conv3d_head_weights_f16s", ptr(w), ptr(out), Cin)
        else:
            out = torch.empty((Cin // 16) * 3 * 2 * 32 * 8, dtype=torch.int16, device=w.device)
            call("ss_pack_conv3d_head_weights_bf16s", ptr(w), ptr(out), Cin)
    return out


#: SS_HEAD_F16=1: the 32 -> 1 classifier heads on two fp16 terms (3 products, a block exponent per input row) instead of three
#: bf16 terms (6 products).  Off by default: since the channels-last hand-off (r02) the head is bound by its loads, not by its
#: matrix work -- measured r03_i: 51.1 vs 52.7 us alone, 462.0 vs 461.6 pairs/s for the step (`profiles/r03_i_*`)
HEAD_F16 = os.environ.get("SS_HEAD_F16", "0") != "0"


def _head_nterms():
    return 19 if (CONV_ENGINE == "f16x3" and HEAD_F16) else _aux_nterms()


def conv3d_head_bf16s_hip(x, wsplit, scale, shift, relu, nterms):
    """Conv3d(C, 1, 3, padding=1) + affine (+ReLU) on the split-bf16 engine (conv3d_head.hip)."""
    x = x if x.is_contiguous() else x.contiguous()
    dev = _lib.require_device(x, scale, shift)
    B, Cin, D, H, W = x.shape
    out = torch.empty((B, 1, D, H, W), dtype=x.dtype, device=x.device)
    with torch.cuda.device(dev):
        call("ss_conv3d_head_bf16s_fwd", ptr(x), ptr(wsplit), ptr(scale), ptr(shift), ptr(out), B, Cin, D, H, W,
             int(relu), int(nterms))
    return out


def pack_pointwise_weight_bf16s(w):
    """[Cout,Cin] (or [Cout,Cin,1,1,1]) fp32 -> split-bf16 fragments for ss_conv3d_pointwise_bf16s_fwd."""
    w = w.detach().float().reshape(w.shape[0], w.shape[1]).contiguous()
    _lib.require_device(w)
    Cout, Cin = w.shape
    assert Cin % 16 == 0
    out = torch.empty(((Cout + 31) // 32) * (Cin // 16) * 3 * 2 * 32 * 8, dtype=torch.int16, device=w.device)
    with torch.cuda.device(w.device):
        call("ss_pack_pointwise_weights_bf16s", ptr(w), ptr(out), Cout, Cin)
    return out


def conv3d_pointwise_bf16s_hip(x, wsplit, Cout, scale, shift, relu, nterms):
    """1x1x1 Conv3d / Linear over channels + affine (+ReLU) on the split-bf16 engine; x [B,Cin,*spatial]."""
    x = x if x.is_contiguous() else x.contiguous()
    dev = _lib.require_device(x, scale, shift)
    B, Cin = x.shape[0], x.shape[1]
    out = torch.empty((B, Cout) + tuple(x.shape[2:]), dtype=x.dtype, device=x.device)
    with torch.cuda.device(dev):
        call("ss_conv3d_pointwise_bf16s_fwd", ptr(x), ptr(wsplit), ptr(scale), ptr(shift), ptr(out), B, Cin, Cout,
             x[0, 0].numel(), int(relu), int(nterms))
    return out


def _convbn_params(owner, key, conv, bn):
    """(wpack, scale, shift) of a Conv3d(+BN) pair, cached on `owner`."""
    srcs = [conv.weight] + ([bn.weight, bn.bias, bn.running_mean, bn.running_var] if bn is not None else [])

    def build():
        wp = pack_conv_weight(conv.weight)
        if bn is None:
            return wp, None, None
        s, b = fold_bn(bn)
        return wp, s, b
    return _cache(owner).get(key, srcs, build)


def _conv_geometry(conv):
    k, s, p = conv.kernel_size, conv.stride, conv.padding
    assert k[0] == k[1] == k[2] and s[0] == s[1] == s[2] and p[0] == p[1] == p[2] == k[0] // 2
    assert conv.bias is None and conv.groups == 1 and conv.dilation == (1, 1, 1)
    return k[0], s[0]


def run_convbn(owner, key, conv, bn, x, relu, residual=None, gate=None):
    """Fused Conv3d -> BN(eval) [-> +residual] [-> ReLU] [-> * sigmoid(gate)] on the selected engine."""
    k, s = _conv_geometry(conv)
    if CONV_ENGINE != "f32" and k == 3 and s in (1, 2) and conv.out_channels > 1:
        nterms = _tiled_nterms()
        srcs = [conv.weight] + ([bn.weight, bn.bias, bn.running_mean, bn.running_var] if bn is not None else [])

        def build():
            sc, sh = fold_bn(bn) if bn is not None else (None, None)
            return pack_conv_weight_bf16s(conv.weight, nterms), sc, sh
        ws, scale, shift = _cache(owner).get(key + ("/f16s" if nterms == 19 else "/bf16s"), srcs, build)
        return conv3d_bf16s_hip(x, ws, conv.out_channels, scale, shift, relu, nterms, residual, gate, stride=s)
    if (CONV_ENGINE != "f32" and k == 3 and s == 1 and conv.out_channels == 1 and conv.in_channels in (16, 32, 64)
            and residual is None and gate is None):
        nterms = _head_nterms()
        srcs = [conv.weight] + ([bn.weight, bn.bias, bn.running_mean, bn.running_var] if bn is not None else [])

        def build_head():
            sc, sh = fold_bn(bn) if bn is not None else (None, None)
            return pack_head_weight_bf16s(conv.weight, nterms), sc, sh
        ws, scale, shift = _cache(owner).get(key + "/head_%d" % nterms, srcs, build_head)
        return conv3d_head_bf16s_hip(x, ws, scale, shift, relu, nterms)
    wp, scale, shift = _convbn_params(owner, key, conv, bn)
    return conv3d_hip(x, wp, scale, shift, k, s, relu, residual, gate)


# --------------------------------------------------------------------------------------
# training: 3x3x3 Conv3d / ConvTranspose3d with HIP forward, data gradient and weight gradient
# (main_us3d.py:186-222 back-propagates through the whole stack; BatchNorm with batch statistics and ReLU stay PyTorch)
# --------------------------------------------------------------------------------------

CLASSIFIER_CL = os.environ.get("SS_CLASSIFIER_CL", "1") != "0"    # 0: plain-layout intermediate inside the classifiers (two generic launches)
TRAIN_HIP = os.environ.get("SS_TRAIN_HIP", "1") != "0"      # 0: the stock PyTorch layers whenever autograd / batch statistics are needed


def conv3d_wgrad_hip(grad_out, x, Cout, Cin, stride):
    """dW [Cout,Cin,3,3,3] of a 3x3x3, padding-1 Conv3d: grad_out [B,Cout,Do,Ho,Wo], x [B,Cin,D,H,W] (conv3d_wgrad.hip)."""
    grad_out = grad_out if grad_out.is_contiguous() else grad_out.contiguous()
    x = x if x.is_contiguous() else x.contiguous()
    dev = _lib.require_device(grad_out, x)
    B, _, D, H, W = x.shape
    gw = torch.empty((Cout, Cin, 3, 3, 3), dtype=x.dtype, device=x.device)
    with torch.cuda.device(dev):
        call("ss_conv3d_wgrad_fwd", ptr(grad_out), ptr(x), ptr(gw), B, Cin, D, H, W, Cout, int(stride))
    return gw


def _conv_k3_forward(x, w, stride):
    """Conv3d(k3, p1, stride, no bias) on the selected engine, weights packed on the fly (they change every step)."""
    if CONV_ENGINE != "f32" and w.shape[0] == 1 and stride == 1 and w.shape[1] in (16, 32, 64):
        return conv3d_head_bf16s_hip(x, pack_head_weight_bf16s(w, _head_nterms()), None, None, False, _head_nterms())      # the 32 -> 1 classifier heads
    if CONV_ENGINE == "f32":
        return conv3d_hip(x, pack_conv_weight(w), None, None, 3, stride, False)
    nterms = _tiled_nterms()
    return conv3d_bf16s_hip(x, pack_conv_weight_bf16s(w, nterms), w.shape[0], None, None, False, nterms, stride=stride)


def _deconv_k3_forward(x, w):
    """ConvTranspose3d(k3, s2, p1, op1, no bias), weight [Cin,Cout,3,3,3]."""
    wp = pack_conv_weight(w, transposed=True)
    zero = torch.zeros(w.shape[1], dtype=x.dtype, device=x.device)
    B, _, D, H, W = x.shape
    workgroups = B * D * ((H + 3) // 4) * ((W + 31) // 32) * ((w.shape[1] + 31) // 32)
    if CONV_ENGINE != "f32" and DECONV_BF16S and workgroups >= DECONV_MIN_WORKGROUPS:
        return deconv3d_bf16s_hip(x, pack_deconv_weight_bf16s(wp, _deconv_nterms()), w.shape[1], zero, False, _deconv_nterms())
    return deconv3d_hip(x, wp, zero, relu=False)


class _Conv3dK3(torch.autograd.Function):
    """y = conv3d(x, w, stride, padding=1).  dx: stride 1 = the same convolution with the taps flipped and the channel axes
    swapped; stride 2 = the transposed convolution (the deconv kernels).  dw: conv3d_wgrad.hip."""

    @staticmethod
    def forward(ctx, x, w, stride):
        x = x if x.is_contiguous() else x.contiguous()
        ctx.save_for_backward(x, w)
        ctx.stride = stride
        return _conv_k3_forward(x, w.detach(), stride)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g = g if g.is_contiguous() else g.contiguous()
        gx = gw = None
        if ctx.needs_input_grad[0]:
            if ctx.stride == 1:
                gx = _conv_k3_forward(g, w.detach().transpose(0, 1).flip(2, 3, 4).contiguous(), 1)
            else:
                gx = _deconv_k3_forward(g, w.detach())          # w [Cout,Cin,...] read as ConvTranspose3d's [in,out,...]
        if ctx.needs_input_grad[1]:
            gw = conv3d_wgrad_hip(g, x, w.shape[0], w.shape[1], ctx.stride)
        return gx, gw, None


class _Deconv3dK3(torch.autograd.Function):
    """y = conv_transpose3d(x, w, stride 2, padding 1, output_padding 1), w [Cin,Cout,3,3,3].  dx = the stride-2 convolution of
    the output gradient with the same tensor read as a Conv3d weight [out=Cin, in=Cout]; dw = the stride-2 weight gradient
    with the roles of input and output gradient swapped."""

    @staticmethod
    def forward(ctx, x, w):
        x = x if x.is_contiguous() else x.contiguous()
        ctx.save_for_backward(x, w)
        return _deconv_k3_forward(x, w.detach())

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g = g if g.is_contiguous() else g.contiguous()
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = _conv_k3_forward(g, w.detach(), 2)
        if ctx.needs_input_grad[1]:
            gw = conv3d_wgrad_hip(x, g, w.shape[0], w.shape[1], 2)
        return gx, gw


def _is_k3(conv, stride_ok=(1, 2)):
    return (conv.kernel_size == (3, 3, 3) and conv.padding == (1, 1, 1) and conv.stride[0] == conv.stride[1] == conv.stride[2]
            and conv.stride[0] in stride_ok and conv.dilation == (1, 1, 1) and conv.groups == 1 and conv.bias is None
            and conv.padding_mode == "zeros")


def conv3d_train(conv, x):
    """nn.Conv3d's forward for the training path: the HIP autograd function for 3x3x3 layers on the GPU (stride 2 needs even
    sizes: its data gradient is the k3-s2-p1-op1 transposed convolution), the stock layer otherwise."""
    if (TRAIN_HIP and x.is_cuda and x.dtype == torch.float32 and isinstance(conv, nn.Conv3d) and _is_k3(conv)
            and (conv.stride[0] == 1 or all(n % 2 == 0 for n in x.shape[2:]))):
        PATH_COUNTS["hip_train"] = PATH_COUNTS.get("hip_train", 0) + 1
        return _Conv3dK3.apply(x, conv.weight, conv.stride[0])
    if isinstance(conv, (nn.Conv3d, nn.Conv2d)) and T.is_k1(conv):
        return T.conv_k1(x, conv.weight, conv.bias)                       # redir1 / redir2, channelAtt.im_att (counts its own path)
    PATH_COUNTS["torch"] += 1
    return conv(x)


def deconv3d_train(deconv, x):
    if (TRAIN_HIP and x.is_cuda and x.dtype == torch.float32 and isinstance(deconv, nn.ConvTranspose3d) and deconv.kernel_size == (3, 3, 3)
            and deconv.stride == (2, 2, 2) and deconv.padding == (1, 1, 1) and deconv.output_padding == (1, 1, 1)
            and deconv.dilation == (1, 1, 1) and deconv.groups == 1 and deconv.bias is None):
        PATH_COUNTS["hip_train"] = PATH_COUNTS.get("hip_train", 0) + 1
        return _Deconv3dK3.apply(x, deconv.weight)
    PATH_COUNTS["torch"] += 1
    return deconv(x)


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
            x = T.batchnorm_train(self.bn, x, relu=bool(self.relu)) if self.bn.training else (F.relu(self.bn(x)) if self.relu else self.bn(x))
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


STEM_LEFT_FUSED = os.environ.get("SS_STEM_LEFT_FUSED", "1") != "0"     # Q of the broadcast half on the fly (one launch) or through HBM (two)


def _stem_halves_params(stem, C):
    conv, bn = stem.conv, stem.bn if stem.use_bn else None
    Cout = conv.out_channels
    assert conv.in_channels == 2 * C and _conv_geometry(conv) == (3, 1)
    srcs = [conv.weight] + ([bn.weight, bn.bias, bn.running_mean, bn.running_var] if bn is not None else [])

    def build():
        sc, sh = fold_bn(bn) if bn is not None else (None, None)
        w = conv.weight.detach().float()
        wl = w[:, :C].reshape(Cout, C, 27)           # the left half's weights, unscaled: its sum joins the accumulator
        wq = wl.permute(2, 0, 1).reshape(27 * Cout, C)                           # row tap*Cout + co (two-launch form)
        # fused form: per pair of output channels 64 rows, row tap*2 + c = channel 2*pair + c, rows 54-63 zero
        wf = torch.zeros(Cout // 2, 64, C, dtype=w.dtype, device=w.device)
        wf[:, :54] = wl.reshape(Cout // 2, 2, C, 27).permute(0, 3, 1, 2).reshape(Cout // 2, 54, C)
        return (pack_pointwise_weight_bf16s(wq), pack_pointwise_weight_bf16s(wf.reshape(Cout // 2 * 64, C)),
                pack_conv_weight_bf16s(w[:, C:].contiguous(), _tiled_nterms()), sc, sh)
    return _cache(stem).get("bc/halves/" + CONV_ENGINE, srcs, build)


def stem_broadcast_half(stem, left, att):
    """Partial sum of `stem` over its first C input channels when they are att * (the 2-D map `left` [B,C,H,W]
    broadcast over the candidates): sum_tap att[pos+tap] * Q[tap](pos+tap), Q = a 1x1 projection of `left`
    (3.6 instead of 87 GFLOP on the bench shape).  -> [B,Cout,nd,H,W], no BatchNorm / ReLU applied."""
    assert stem.is_3d and not stem.deconv and CONV_ENGINE != "f32" and _inference(stem, left, att)
    C, Cout = left.shape[1], stem.conv.out_channels
    nterms = _aux_nterms()
    wq, wf, _, _, _ = _stem_halves_params(stem, C)
    PATH_COUNTS["hip"] += 1
    if C == 32 and Cout % 2 == 0 and STEM_LEFT_FUSED:
        return ops.stem_left_fused(left, wf, att, Cout, nterms)
    q = conv3d_pointwise_bf16s_hip(left, wq, 27 * Cout, None, None, False, nterms)               # [B, 27*Cout, H, W]
    return ops.stem_left(q, att)


def stem_volume_half(stem, right_vol, partial, gate=None):
    """`stem` over its last C input channels (`right_vol` [B,C,nd,H,W]) continuing `partial`, then BatchNorm, ReLU
    and the optional channelAtt gate (`gate` [B,Cout,H,W]: the SIGMOID of the gate's logits) on the total."""
    assert stem.is_3d and not stem.deconv and CONV_ENGINE != "f32" and _inference(stem, right_vol, partial, gate)
    nterms = _tiled_nterms()
    _, _, wr, scale, shift = _stem_halves_params(stem, right_vol.shape[1])
    g = None if gate is None else gate.contiguous()
    return conv3d_bf16s_hip(right_vol, wr, stem.conv.out_channels, scale, shift, bool(stem.relu), nterms, None, g, partial=partial)


#: SS_STEM_PRESPLIT=1: the warped half handed to the stem PRE-SPLIT (ss_concat_sampled_presplit_fwd -> ss_conv3d_presplit_fwd:
#: LDS-DMA staging, no conversion, no per-chunk maximum in the conv; f16x3 engine only).  OFF by default: measured r03_d / r03_e
#: (profiles/r03_e_bench_b1*.json) the stem launch takes 313-317 us in that form against 297-298 us with the fp32 volume and the
#: on-the-fly split, the step 442.9 vs 452.3 pairs/s -- removing ALL staging arithmetic beside the matrix pipe does not speed
#: the kernel up (nor did removing 16 % of it: 4.2 -> 3.55 VALU per MFMA at unchanged time, profiles/r03_b_pmc_conv_stem.txt)
STEM_PRESPLIT = os.environ.get("SS_STEM_PRESPLIT", "0") != "0"


def stem_presplit_applies(stem, right):
    return (STEM_PRESPLIT and CONV_ENGINE == "f16x3" and right.shape[1] % 8 == 0 and stem.conv.in_channels == 2 * right.shape[1]
            and _conv_geometry(stem.conv) == (3, 1))


def stem_volume_half_presplit(stem, xs, xexp, partial, gate=None):
    """stem_volume_half on the pre-split warped half (xs, xexp of ops.concat_volume_sampled_presplit)."""
    assert stem.is_3d and not stem.deconv and CONV_ENGINE == "f16x3" and _inference(stem, partial, gate)
    B, nchunks, _, D, H, W, _ = xs.shape
    _, _, wr, scale, shift = _stem_halves_params(stem, nchunks * 8)
    Cout = stem.conv.out_channels
    g = None if gate is None else gate.contiguous()
    dev = _lib.require_device(partial, scale, shift, g)
    out = torch.empty((B, Cout, D, H, W), dtype=torch.float32, device=xs.device)
    if partial is not None:
        assert partial.shape == out.shape and partial.is_contiguous()
    with torch.cuda.device(dev if dev is not None else xs.device):
        call("ss_conv3d_presplit_fwd", ptr(xs), ptr(xexp), ptr(wr), ptr(partial), ptr(scale), ptr(shift), ptr(g), ptr(out),
             B, nchunks * 8, D, H, W, Cout, int(bool(stem.relu)))
    return out


def stem_of_broadcast_and_volume(stem, left, att, right_vol, gate=None):
    """`stem` (a 3x3x3 stride-1 BasicConv with 2C input channels) applied to cat(att * left broadcast over the
    candidates, right_vol) WITHOUT building the left half of that volume or convolving it: by linearity its
    contribution (stem_broadcast_half) initialises the accumulators of the right half's convolution
    (stem_volume_half) (models/SemStereo.py:241-244, 316-320).  Split-bf16 engines, inference only."""
    return stem_volume_half(stem, right_vol, stem_broadcast_half(stem, left, att), gate)


ATTENTION_FORM = os.environ.get("SS_ATTENTION", "split")      # "split" (3 launches) | "fused" (one kernel per window)


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

        bf = CONV_ENGINE != "f32" and self.dim_3d in (32, 64, 128)

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
            if ATTENTION_FORM == "split":
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
            return (wd, wr, (db + rb).contiguous(), pack_deconv_weight_bf16s(wd, _deconv_nterms()) if CONV_ENGINE != "f32" else None,
                    pack_deconv_weight_bf16s(wr) if CONV_ENGINE != "f32" else None)
        return _cache(self).get(key + "/" + CONV_ENGINE, srcs, build)

    def _up(self, key, deconv_seq, redir_seq, x, skip):
        wd, wr, shift, wds, wrs = self._up_params(key, deconv_seq, redir_seq)
        B, _, D, H, W = x.shape
        workgroups = B * D * ((H + 3) // 4) * ((W + 31) // 32) * ((wd.shape[2] + 31) // 32)
        # layers with few workgroups: the exact-fp32 kernel's even/odd-plane split doubles them (see DECONV_MIN_WORKGROUPS)
        if CONV_ENGINE != "f32" and DECONV_BF16S and workgroups >= DECONV_MIN_WORKGROUPS:
            return deconv3d_bf16s_hip(x, wds, wd.shape[2], shift, True, _deconv_nterms(), skip, wrs)
        return deconv3d_hip(x, wd, shift, relu=True, skip=skip, skip_wpack=wr)

    def forward(self, x):
        x = dfr.real(x)
        if not _inference(self, x):
            def cbr(seq, t):                       # nn.Sequential(convbn_3d, ReLU): conv -> BatchNorm (batch statistics) -> ReLU fused
                cb = seq[0]
                return cb.train_forward(t, relu=True) if isinstance(cb, _ConvBN3d) and cb[1].training else seq(t)
            conv1 = cbr(self.conv1, x)
            conv2 = cbr(self.conv2, conv1)
            conv3 = cbr(self.conv3, conv2)
            conv4 = self.attention_block(cbr(self.conv4, conv3))
            conv5 = F.relu(T.batchnorm_train(self.conv5[1], deconv3d_train(self.conv5[0], conv4)) + self.redir2(conv2))
            return F.relu(T.batchnorm_train(self.conv6[1], deconv3d_train(self.conv6[0], conv5)) + self.redir1(x))
        PATH_COUNTS["hip"] += 1
        c1 = run_convbn(self, "c1", self.conv1[0][0], self.conv1[0][1], x, relu=True)
        c2 = run_convbn(self, "c2", self.conv2[0][0], self.conv2[0][1], c1, relu=True)
        c3 = run_convbn(self, "c3", self.conv3[0][0], self.conv3[0][1], c2, relu=True)
        c4 = run_convbn(self, "c4", self.conv4[0][0], self.conv4[0][1], c3, relu=True)
        c4 = self.attention_block(c4)
        c5 = self._up("u5", self.conv5, self.redir2, c4, c2)
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
            if (CLASSIFIER_CL and CONV_ENGINE != "f32" and c0.in_channels == c0.out_channels == c2.in_channels == 32
                    and _conv_geometry(c0) == (3, 1) and _conv_geometry(c2) == (3, 1) and c2.out_channels == 1):
                nt0, nt2 = _tiled_nterms(), _head_nterms()

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
            n = self.num_classes
            s0, t0 = fold_bn(self.conv[0])
            sa, ta = fold_bn(self.conv[2])
            s1, t1 = fold_bn(self.conv1[1])
            s2, t2 = fold_bn(self.conv2[1])
            parts = [s0, t0, self.conv[1].weight.reshape(n * 9), self.conv[1].bias, sa, ta,
                     self.conv1[0].weight.reshape(n * n), self.conv1[0].bias, s1, t1,
                     self.conv2[0].weight.reshape(n * n), self.conv2[0].bias, s2, t2,
                     self.conv3.weight.reshape(n), self.conv3.bias]
            return torch.cat([p.detach().float().reshape(-1) for p in parts]).contiguous()
        return _cache(self).get("ssr", srcs, build)

    def forward(self, depth_low, weights, pred_label):
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
        pred_label = F.softmax(pred_label, dim=1)
        depth_ = F.interpolate(depth_low, (h * 4, w * 4), mode="bilinear").reshape(b, 1, h * 4, w * 4)
        depth = self.conv(depth_)
        prob = torch.sigmoid(self.conv1(pred_label * weights))
        prob = torch.sigmoid(self.conv2(prob * weights))
        return (depth_ + self.conv3(depth * prob)).squeeze(1)
