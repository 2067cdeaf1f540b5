"""3x3x3 Conv3d / ConvTranspose3d of the training path (main_us3d.py:186-222 back-propagates through the whole stack): HIP
forward on the selected engine with weights packed on the fly, data gradient on the SAME forward kernels (stride 1: flipped taps
and swapped channel axes; stride 2: the transposed-conv kernels; transposed conv: the stride-2 conv), weight gradient
conv3d_wgrad.hip.  BatchNorm with batch statistics, the 1x1 layers, the gates and the attention core live in train.py.
"""
import os

import torch
import torch.nn as nn

from . import _lib
from . import engine as E
from . import train as T
from ._lib import call, ptr
from .engine import (PATH_COUNTS, conv3d_bf16s_hip, conv3d_head_bf16s_hip, conv3d_hip, deconv3d_bf16s_hip, deconv3d_hip,
                     pack_conv_weight, pack_conv_weight_bf16s, pack_deconv_weight_bf16s, pack_head_weight_bf16s)



#: weight-gradient engine of the 3x3x3 layers: "bf16x6" (default, r06: three bf16 terms, six products on the bf16 matrix core,
#: conv3d_wgrad_bf16s.hip) or "f32" (the exact-fp32 MFMA kernel of rounds 2-5, conv3d_wgrad.hip: 13 TFLOP/s, two thirds of the
#: training step).  SS_WGRAD_ENGINE; read at call time like the other Python-level switches.
WGRAD_ENGINE = os.environ.get("SS_WGRAD_ENGINE", "bf16x6")


def conv3d_wgrad_hip(grad_out, x, Cout, Cin, stride):
    """dW [Cout,Cin,3,3,3] of a 3x3x3, padding-1 Conv3d: grad_out [B,Cout,Do,Ho,Wo], x [B,Cin,D,H,W] (conv3d_wgrad_bf16s.hip;
    WGRAD_ENGINE = "f32": conv3d_wgrad.hip)."""
    grad_out = grad_out if grad_out.is_contiguous() else grad_out.contiguous()
    x = x if x.is_contiguous() else x.contiguous()
    dev = _lib.require_device(grad_out, x)
    B, _, D, H, W = x.shape
    gw = torch.empty((Cout, Cin, 3, 3, 3), dtype=x.dtype, device=x.device)
    with torch.cuda.device(dev):
        if WGRAD_ENGINE == "f32":
            call("ss_conv3d_wgrad_fwd", ptr(grad_out), ptr(x), ptr(gw), B, Cin, D, H, W, Cout, int(stride))
        else:
            ws = torch.empty(((Cout + 31) // 32) * ((Cin + 31) // 32) * 27 * 1024, dtype=x.dtype, device=x.device)
            call("ss_conv3d_wgrad_bf16s_fwd", ptr(grad_out), ptr(x), ptr(gw), ptr(ws), B, Cin, D, H, W, Cout, int(stride))
    return gw


def _conv_k3_forward(x, w, stride):
    """Conv3d(k3, p1, stride, no bias) on the selected engine, weights packed on the fly (they change every step)."""
    if E.CONV_ENGINE != "f32" and w.shape[0] == 1 and stride == 1 and w.shape[1] in (16, 32, 64):
        return conv3d_head_bf16s_hip(x, pack_head_weight_bf16s(w, E._head_nterms()), None, None, False, E._head_nterms())      # the 32 -> 1 classifier heads
    if E.CONV_ENGINE == "f32":
        return conv3d_hip(x, pack_conv_weight(w), None, None, 3, stride, False)
    nterms = E._tiled_nterms()
    return conv3d_bf16s_hip(x, pack_conv_weight_bf16s(w, nterms), w.shape[0], None, None, False, nterms, stride=stride)


def _deconv_k3_forward(x, w):
    """ConvTranspose3d(k3, s2, p1, op1, no bias), weight [Cin,Cout,3,3,3]."""
    wp = pack_conv_weight(w, transposed=True)
    zero = torch.zeros(w.shape[1], dtype=x.dtype, device=x.device)
    B, _, D, H, W = x.shape
    workgroups = B * D * ((H + 3) // 4) * ((W + 31) // 32) * ((w.shape[1] + 31) // 32)
    if E.CONV_ENGINE != "f32" and E.DECONV_BF16S and workgroups >= E.DECONV_MIN_WORKGROUPS:
        return deconv3d_bf16s_hip(x, pack_deconv_weight_bf16s(wp, E._deconv_nterms()), w.shape[1], zero, False, E._deconv_nterms())
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
    if (E.TRAIN_HIP and x.is_cuda and x.dtype == torch.float32 and isinstance(conv, nn.Conv3d) and _is_k3(conv)
            and (conv.stride[0] == 1 or all(n % 2 == 0 for n in x.shape[2:]))):
        PATH_COUNTS["hip_train"] = PATH_COUNTS.get("hip_train", 0) + 1
        return _Conv3dK3.apply(x, conv.weight, conv.stride[0])
    if isinstance(conv, (nn.Conv3d, nn.Conv2d)) and T.is_k1(conv):
        return T.conv_k1(x, conv.weight, conv.bias)                       # redir1 / redir2, channelAtt.im_att (counts its own path)
    PATH_COUNTS["torch"] += 1
    return conv(x)


def deconv3d_train(deconv, x):
    if (E.TRAIN_HIP and x.is_cuda and x.dtype == torch.float32 and isinstance(deconv, nn.ConvTranspose3d) and deconv.kernel_size == (3, 3, 3)
            and deconv.stride == (2, 2, 2) and deconv.padding == (1, 1, 1) and deconv.output_padding == (1, 1, 1)
            and deconv.dilation == (1, 1, 1) and deconv.groups == 1 and deconv.bias is None):
        PATH_COUNTS["hip_train"] = PATH_COUNTS.get("hip_train", 0) + 1
        return _Deconv3dK3.apply(x, deconv.weight)
    PATH_COUNTS["torch"] += 1
    return deconv(x)

