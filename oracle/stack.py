"""Oracle restatement of the 3-D aggregation stack (TEST INFRASTRUCTURE ONLY).

Functional form over a flat parameter dict `P` whose keys are the reference's
state_dict keys, e.g. ``P["hourglass_att.conv1.0.0.weight"]``.
Inference semantics only (BatchNorm uses running statistics, eps = 1e-5).

Reference: convbn_3d models/submodule_other.py:845-848; attention_block
models/submodule_other.py:790-837; hourglass / hourglass2
models/SemStereo.py:106-182; channelAtt models/SemStereo.py:89-103;
BasicConv models/submodule.py:89-116.
"""
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
TRAINING = False        # True: BatchNorm normalises with the BATCH statistics (model.train(), main_us3d.py:186-222)


class training_mode:
    """with stack.training_mode(): the functional stack behaves as the reference's modules do in train() -- used with autograd
    on the CPU as the gradient oracle of the HIP training path (tests/test_parity_gpu.py)."""

    def __enter__(self):
        global TRAINING
        self.prev, TRAINING = TRAINING, True

    def __exit__(self, *a):
        global TRAINING
        TRAINING = self.prev


def bn(P, key, x):
    if TRAINING:
        return F.batch_norm(x, None, None, P[key + ".weight"], P[key + ".bias"], True, 0.0, BN_EPS)
    return F.batch_norm(x, P[key + ".running_mean"], P[key + ".running_var"],
                        P[key + ".weight"], P[key + ".bias"], False, 0.0, BN_EPS)


def convbn_3d(P, key, x, stride, pad):
    """Conv3d(bias=False) -> BatchNorm3d; `key` names the Sequential."""
    return bn(P, key + ".1", F.conv3d(x, P[key + ".0.weight"], None, stride, pad))


def deconvbn_3d(P, key, x):
    """ConvTranspose3d(k3, s2, p1, output_padding 1, bias=False) -> BatchNorm3d."""
    y = F.conv_transpose3d(x, P[key + ".0.weight"], None, stride=2, padding=1, output_padding=1)
    return bn(P, key + ".1", y)


def basic_conv(P, key, x, is_3d, stride=1, pad=1, relu=True):
    """BasicConv: conv(bias=False) -> BN -> ReLU (keys `.conv.weight`, `.bn.*`)."""
    conv = F.conv3d if is_3d else F.conv2d
    y = bn(P, key + ".bn", conv(x, P[key + ".conv.weight"], None, stride, pad))
    return F.relu(y) if relu else y


def attention_block(P, key, x, block, num_heads=16):
    """Windowed multi-head self-attention over (bd,bh,bw) windows of a
    [B,C,D,H,W] volume, then a 1x1x1 conv with bias.  H and W are zero-padded
    up to window multiples (pad tokens are separated from real ones by a -1000
    logit); D must already be a multiple of bd."""
    B, C, D, H0, W0 = x.shape
    bd, bh, bw = block
    pad_r = (bw - W0 % bw) % bw
    pad_b = (bh - H0 % bh) % bh
    x = F.pad(x, (0, pad_r, 0, pad_b))
    H, W = H0 + pad_b, W0 + pad_r
    nd, nh, nw = D // bd, H // bh, W // bw
    T, hd = bd * bh * bw, C // num_heads
    # tokens of each window, channel last: [B, windows, T, C]
    tok = x.reshape(B, C, nd, bd, nh, bh, nw, bw).permute(0, 2, 4, 6, 3, 5, 7, 1).reshape(B, nd * nh * nw, T, C)
    qkv = F.linear(tok, P[key + ".qkv_3d.weight"], P[key + ".qkv_3d.bias"])
    qkv = qkv.reshape(B, nd * nh * nw, T, 3, num_heads, hd).permute(3, 0, 1, 4, 2, 5)
    q, k, v = qkv[0], qkv[1], qkv[2]                           # [B, windows, heads, T, hd]
    logits = torch.matmul(q, k.transpose(-2, -1)) * (hd ** -0.5)
    if pad_r > 0 or pad_b > 0:
        # Reference quirk kept on purpose (models/submodule_other.py:822-823): the
        # fills are `mask[:, -pad_b:, :]` and `mask[:, :, -pad_r:]`, and "-0:"
        # selects EVERYTHING, so when only one of H/W needs padding the whole
        # flag map is 1 and no logit is masked.
        is_pad = torch.zeros((H, W), dtype=x.dtype)
        is_pad[(H - pad_b) if pad_b > 0 else 0:, :] = 1
        is_pad[:, (W - pad_r) if pad_r > 0 else 0:] = 1
        # per (h,w) window: pad flag of each of its bh*bw pixel columns
        flag = is_pad.reshape(nh, bh, nw, bw).permute(0, 2, 1, 3).reshape(nh * nw, bh * bw)
        differs = (flag.unsqueeze(1) != flag.unsqueeze(2)).to(x.dtype) * -1000.0   # [nh*nw, bh*bw, bh*bw]
        differs = differs.repeat(nd, bd, bd)                                       # [windows, T, T]
        logits = logits + differs.reshape(1, nd * nh * nw, 1, T, T)
    att = torch.softmax(logits, dim=-1)
    y = torch.matmul(att, v)                                   # [B, windows, heads, T, hd]
    y = y.reshape(B, nd, nh, nw, num_heads, bd, bh, bw, hd).permute(0, 4, 8, 1, 5, 2, 6, 3, 7)
    y = y.reshape(B, C, D, H, W)[:, :, :, :H0, :W0]
    return F.conv3d(y, P[key + ".final1x1.weight"], P[key + ".final1x1.bias"])


def hourglass(P, key, x, block):
    """hourglass (block (4,4,4)) / hourglass2 (block (6,4,4)): two stride-2
    conv stages, windowed attention at 1/4 resolution, two transposed-conv
    stages with 1x1x1 skip projections."""
    c1 = F.relu(convbn_3d(P, key + ".conv1.0", x, 2, 1))
    c2 = F.relu(convbn_3d(P, key + ".conv2.0", c1, 1, 1))
    c3 = F.relu(convbn_3d(P, key + ".conv3.0", c2, 2, 1))
    c4 = F.relu(convbn_3d(P, key + ".conv4.0", c3, 1, 1))
    c4 = attention_block(P, key + ".attention_block", c4, block)
    c5 = F.relu(deconvbn_3d(P, key + ".conv5", c4) + convbn_3d(P, key + ".redir2", c2, 1, 0))
    c6 = F.relu(deconvbn_3d(P, key + ".conv6", c5) + convbn_3d(P, key + ".redir1", x, 1, 0))
    return c6


def classifier(P, key, x):
    """classif / classif_att_: convbn_3d(32,32) -> ReLU -> Conv3d(32,1,k3,p1,bias=False)."""
    y = F.relu(convbn_3d(P, key + ".0", x, 1, 1))
    return F.conv3d(y, P[key + ".2.weight"], None, 1, 1)


def channel_att(P, key, cv, im):
    """channelAtt: sigmoid(conv1x1(BN-ReLU(conv1x1(im)))) broadcast over D, times cv."""
    a = basic_conv(P, key + ".im_att.0", im, is_3d=False, stride=1, pad=0)
    a = F.conv2d(a, P[key + ".im_att.1.weight"], P[key + ".im_att.1.bias"])
    return torch.sigmoid(a).unsqueeze(2) * cv


def patch_conv(P, cv):
    """`patch`: depthwise Conv3d kernel (1,3,3), pad (0,1,1), groups=C, bias=False."""
    w = P["patch.weight"]
    return F.conv3d(cv, w, None, 1, (0, 1, 1), 1, w.shape[0])


def concat_feature(P, x):
    """`concat_feature`: BasicConv 2-D 3x3 (C -> C/2) then Conv2d 3x3 (C/2 -> C/4, bias=False)."""
    y = basic_conv(P, "concat_feature.0", x, is_3d=False, stride=1, pad=1)
    return F.conv2d(y, P["concat_feature.1.weight"], None, 1, 1)
