"""The functional layer under the module twins: which matrix-core engine runs (switches), the per-module cache of derived
tensors (packed weights, folded BatchNorm affines), weight packing, and one thin launch wrapper per C-ABI entry point of the 3-D
stack -- plus the two compositions that are functions rather than modules: a Conv3d(+BN) pair on the selected engine
(`run_convbn`) and `concat_stem` by halves (models/SemStereo.py:241-244, 316-320).  `modules.py` holds the nn.Module twins that
call into this file, `train_layers.py` the autograd functions of the training path.

The switches below are module attributes of THIS file (tests and tools set `engine.CONV_ENGINE = ...`); `modules.X` reads of the
same names are forwarded here (modules.__getattr__).
"""
import os
import weakref

import torch
import torch.nn as nn

from . import _lib, ops
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


#: Every build of a cache entry (first use, a weight / BatchNorm update, an engine switch, a new shape's key) bumps this counter,
#: and -- while a multi-stream caller has asked for it (retain_replaced) -- the entry it REPLACES is parked in `_RETIRED` instead
#: of being freed: a caller that issues consecutive calls on several HIP streams (segment.PairPipeline) compares the counter
#: around a call; when it moved, the packing kernels ran on that call's stream only and pairs in flight on the other streams may
#: still be reading the replaced tensors: the caller drains its streams, then drops the retired entries.
_CACHE_GENERATION = [0]
_RETIRED = []
_RETAIN = [0]


def cache_generation():
    return _CACHE_GENERATION[0]


def retain_replaced(on):
    """Multi-stream callers: keep replaced cache entries alive until drop_retired() (counted: several callers may ask)."""
    _RETAIN[0] = max(0, _RETAIN[0] + (1 if on else -1))


def drop_retired():
    del _RETIRED[:]


#: at most this many replaced entries are kept (a primed but idle PairPipeline while plain calls follow weight updates would
#: otherwise park the whole model's packed weights once per update, without bound); beyond it the oldest go -- they are older than any
#: pair a pipeline can still have in flight by the time that many rebuilds (each a full pack launch sequence) have been issued
RETIRED_CAP = 256


def _note_build(old):
    _CACHE_GENERATION[0] += 1
    if old is not None and _RETAIN[0]:
        _RETIRED.append(old)
        if len(_RETIRED) > RETIRED_CAP:
            del _RETIRED[:len(_RETIRED) - RETIRED_CAP]


class _ParamCache:
    """Derived device tensors (packed weights, folded affines), rebuilt when a source tensor changes."""

    def __init__(self):
        self._store = {}
        self.owner = None                    # weakref to the module the cache belongs to (set by _cache)

    def get(self, key, sources, build):
        stamp = tuple((t.data_ptr(), t._version, str(t.device)) for t in sources)
        hit = self._store.get(key)
        if hit is None or hit[0] != stamp:
            _note_build(hit)
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
            _note_build(hit)
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


def classifier_fused_applies(x, nterms0):
    """The one-pass form of a classifier (ss_conv3d_classifier_fused_fwd) serves this input: decided by the LAYER (what one pair of
    it offers the chip), never by the batch, so that a pair gets the same bits alone and in a batch."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 5 and nterms0 == 19):
        return False
    B, C, D, H, W = x.shape
    return (C % 8 == 0 and D % 4 == 0 and ((W + 31) // 32) * ((H + 7) // 8) * ((D + 1) // 2) >= 512 and C * D * H * W * 4 < 0x7fffffff
            and H <= 65535 and B * D <= 65535)


def pack_classifier_head_weight(w2):
    """[1,32,3,3,3] fp32 -> the head's fragments for ss_conv3d_classifier_fused_fwd (three bf16 terms, 6144 bytes)."""
    w2 = w2.detach().float().contiguous()
    _lib.require_device(w2)
    assert tuple(w2.shape) == (1, 32, 3, 3, 3)
    out = torch.empty(3 * 2 * 64 * 8, dtype=torch.int16, device=w2.device)
    with torch.cuda.device(w2.device):
        call("ss_pack_classifier_head_weights", ptr(w2), ptr(out))
    return out


class PatchedCost:
    """What a one-pass classifier leaves when its patch sum is folded into the consumer (r06): the tiles' patches of a [B,1,D,H,W] cost."""

    def __init__(self, patches, shape):
        self.patches, self.shape = patches, tuple(shape)        # shape = (B, D, H, W)


#: SS_CLASSIFIER_FOLD=1 (r06): `classif` hands its patches straight to the top-2 soft-argmax that follows it
#: (models/SemStereo.py:322-323: ss_regression_topk_patched_fwd) -- no patch-sum launch, no cost tensor.  Bit-identical, one launch fewer, and
#: SLOWER: the folded kernel gathers 2.4 patch values per cost in a latency-bound launch -- 571.6 / 571.5 / 570.6 -> 566.9 / 566.2 / 567.0
#: pairs/s on one stream, 613.5 / 615.1 / 614.7 -> 613.5 / 613.3 / 614.0 pipelined (profiles/r06_z_ab_classifier_fold.txt).  Off.
CLASSIFIER_FOLD = os.environ.get("SS_CLASSIFIER_FOLD", "0") != "0"


def classifier_fused_hip(x, ws0, scale0, shift0, nterms0, head_w, sum_patches=True):
    """nn.Sequential(convbn_3d(C,32,3,1,1), ReLU, Conv3d(32,1,3,p1)) (models/SemStereo.py:228-234) in one pass over the volume: the
    32-channel intermediate stays in the accumulators, each tile writes a 6 x 6 x 34 patch of head outputs, a second small launch
    adds the patches (conv3d_classifier.hip)."""
    x = x if x.is_contiguous() else x.contiguous()
    dev = _lib.require_device(x, scale0, shift0)
    B, C, D, H, W = x.shape
    ntiles = ((W + 31) // 32) * ((H + 3) // 4) * (D // 4)
    patches = torch.empty((B, ntiles, 6 * 6 * 34), dtype=x.dtype, device=x.device)
    out = torch.empty((B, 1, D, H, W), dtype=x.dtype, device=x.device) if sum_patches else None
    with torch.cuda.device(dev):
        call("ss_conv3d_classifier_fused_fwd", ptr(x), ptr(ws0), ptr(scale0), ptr(shift0), ptr(head_w), ptr(patches), ptr(out),
             B, C, D, H, W, int(nterms0))
    return out if sum_patches else PatchedCost(patches, (B, D, H, W))


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


def run_conv2d_pair(owner, key, conv, bn, xa, xb, relu):
    """run_conv2d on two inputs of one shape in ONE launch (ss_conv2d_bf16s_pair_fwd): -> [2B,Cout,H,W] (first B: xa's), or None."""
    if not (_conv2d_hip_on() and CONV_ENGINE != "f32" and _is_plain_3x3(conv) and xa.is_cuda and xb.is_cuda
            and xa.shape == xb.shape and xa.dtype == xb.dtype == torch.float32):
        return None
    nterms = _tiled_nterms()
    srcs = [conv.weight] + ([bn.weight, bn.bias, bn.running_mean, bn.running_var] if bn is not None else [])

    def build():
        sc, sh = fold_bn(bn) if bn is not None else (None, None)
        return pack_conv2d_weight_bf16s(conv.weight, nterms), sc, sh
    ws, scale, shift = _cache(owner).get(key + "/2d_" + CONV_ENGINE, srcs, build)
    xa, xb = xa.contiguous(), xb.contiguous()
    dev = _lib.require_device(xa, xb, scale, shift)
    B, Cin, H, W = xa.shape
    out = torch.empty((2 * B, conv.out_channels, H, W), dtype=xa.dtype, device=xa.device)
    with torch.cuda.device(dev):
        call("ss_conv2d_bf16s_pair_fwd", ptr(xa), ptr(xb), ptr(ws), ptr(scale), ptr(shift), ptr(out), B, Cin, H, W,
             conv.out_channels, int(relu), int(nterms))
    return out


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



TRAIN_HIP = os.environ.get("SS_TRAIN_HIP", "1") != "0"      # 0: the stock PyTorch layers whenever autograd / batch statistics are needed
CLASSIFIER_CL = os.environ.get("SS_CLASSIFIER_CL", "1") != "0"    # 0: plain-layout intermediate inside the classifiers (two generic launches)
#: the classifiers in ONE pass over the volume (the 32-channel intermediate never leaves the CU: conv3d_classifier.hip) where the
#: layer is large enough for the 4-row tile at batch 1; SS_CLASSIFIER_FUSED=0: the two-launch forms above
CLASSIFIER_FUSED = os.environ.get("SS_CLASSIFIER_FUSED", "1") != "0"


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


#: SS_STEM_GATHER (r05): concat_stem's warped half GATHERED inside the conv's staging from the 2-D right
#: feature map (ss_conv3d_gather_fwd) -- no warp launch, no 201 MB volume.  For INTEGER candidates only (what the reference's
#: top-24 selection produces, models/SemStereo.py:299-305): callers pass `integer_candidates=True` when they know the provenance.
STEM_GATHER = os.environ.get("SS_STEM_GATHER", "1") != "0"
#: SS_STEM_INPLACE (r06): the gathered conv writes its result over the partial sum it continues (callers hand it a partial sum nobody else holds)
STEM_INPLACE = os.environ.get("SS_STEM_INPLACE", "1") != "0"


def stem_gather_applies(stem, right, samples):
    return (STEM_GATHER and CONV_ENGINE == "f16x3" and right.shape[1] % 8 == 0 and right.shape[1] >= 16
            and stem.conv.in_channels == 2 * right.shape[1] and _conv_geometry(stem.conv) == (3, 1)
            and right.dtype == torch.float32 and samples.dtype == torch.float32
            # (a one-row or one-column map: the reference's coordinate normalisation divides by (H - 1) / 2 = 0 and its warp is NaN,
            # models/submodule.py:274-279 -- the warp kernel reproduces that, the gather would not)
            and right.shape[-1] > 1 and right.shape[-2] > 1)


def stem_gather_half(stem, right, samples, att, partial, gate=None, consume_partial=False):
    """`stem` over its last C input channels when they are att * (the 2-D map `right` [B,C,H,W] warped by the INTEGER candidates
    `samples` [B,nd,H,W]) -- models/SemStereo.py:241-244, 316-320 -- continuing `partial`, then BatchNorm, ReLU and the gate:
    one launch, the operand gathered while the conv stages its tiles.  consume_partial=True (callers that made `partial` themselves and
    hold no other reference, STEM_INPLACE): the result is written over `partial`."""
    assert stem.is_3d and not stem.deconv and CONV_ENGINE == "f16x3" and _inference(stem, right, partial, gate)
    right, samples = right.contiguous(), samples.contiguous()
    att = att.reshape(att.shape[0], att.shape[-3], att.shape[-2], att.shape[-1]).contiguous()
    B, C, H, W = right.shape
    nd = samples.shape[1]
    assert samples.shape == (B, nd, H, W) and att.shape == samples.shape
    _, _, wr, scale, shift = _stem_halves_params(stem, C)
    Cout = stem.conv.out_channels
    g = None if gate is None else gate.contiguous()
    dev = _lib.require_device(right, samples, att, partial, scale, shift, g)
    # r06: the result may be written IN PLACE over the partial sum (every element is read once, by the lane that then writes it): the pair
    # of launches then touches one 201 MB tensor instead of two, which is what the 256 MB Infinity Cache can still hold between them
    inplace = consume_partial and STEM_INPLACE and partial is not None and partial.is_contiguous() and tuple(partial.shape) == (B, Cout, nd, H, W) and not partial.requires_grad
    out = partial if inplace else torch.empty((B, Cout, nd, H, W), dtype=torch.float32, device=right.device)
    if partial is not None:
        assert partial.shape == out.shape and partial.is_contiguous()
    if g is not None:
        assert g.shape == (B, Cout, H, W)
    PATH_COUNTS["hip"] += 1
    with torch.cuda.device(dev):
        call("ss_conv3d_gather_fwd", ptr(right), ptr(samples), ptr(att), ptr(wr), ptr(partial), ptr(scale), ptr(shift), ptr(g),
             ptr(out), B, C, nd, H, W, Cout, int(bool(stem.relu)), _tiled_nterms())
    return out


def stem_of_broadcast_and_volume(stem, left, att, right_vol, gate=None):
    """`stem` (a 3x3x3 stride-1 BasicConv with 2C input channels) applied to cat(att * left broadcast over the
    candidates, right_vol) WITHOUT building the left half of that volume or convolving it: by linearity its
    contribution (stem_broadcast_half) initialises the accumulators of the right half's convolution
    (stem_volume_half) (models/SemStereo.py:241-244, 316-320).  Split-bf16 engines, inference only."""
    return stem_volume_half(stem, right_vol, stem_broadcast_half(stem, left, att), gate)

ATTENTION_FORM = os.environ.get("SS_ATTENTION", "split")      # "split" (3 launches) | "fused" (one kernel per window)

#: the names tests / tools may SET on this module; `modules.X` forwards reads of them here
SWITCHES = ("CONV_ENGINE", "DECONV_F16", "DECONV_MIN_WORKGROUPS", "DECONV_BF16S", "CLASSIFIER_CL", "CLASSIFIER_FUSED", "CLASSIFIER_FOLD", "TRAIN_HIP", "ATTENTION_FORM",
            "STEM_LEFT_FUSED", "STEM_PRESPLIT", "STEM_GATHER", "STEM_INPLACE", "HEAD_F16", "CONV2D_HIP")
