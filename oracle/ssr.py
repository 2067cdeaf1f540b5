"""Oracle restatement of the semantic-guided refinement head `SSR_upsample`
(reference models/submodule.py:412-431; call sites models/SemStereo.py:311, 324).
TEST INFRASTRUCTURE ONLY.  Functional form over a flat dict with the reference's
state_dict keys under `ssr_upsample.`; inference semantics (BatchNorm running stats).
"""
import torch
import torch.nn.functional as F

from . import detdata
from .stack import bn


def ssr_param_shapes(num_classes=6):
    n = num_classes
    S = {}

    def bnk(key, c):
        for s in ("weight", "bias", "running_mean", "running_var"):
            S[f"{key}.{s}"] = (c,)
    bnk("ssr_upsample.conv.0", 1)
    S["ssr_upsample.conv.1.weight"] = (n, 1, 3, 3); S["ssr_upsample.conv.1.bias"] = (n,)
    bnk("ssr_upsample.conv.2", n)
    for k in ("conv1", "conv2"):
        S[f"ssr_upsample.{k}.0.weight"] = (n, n, 1, 1); S[f"ssr_upsample.{k}.0.bias"] = (n,)
        bnk(f"ssr_upsample.{k}.1", n)
    S["ssr_upsample.conv3.weight"] = (1, n, 1, 1); S["ssr_upsample.conv3.bias"] = (1,)
    return S


def deterministic_ssr_params(num_classes=6, salt=11):
    P = {}
    for i, (key, shape) in enumerate(sorted(ssr_param_shapes(num_classes).items())):
        s = salt * 1000 + i
        if key.endswith("running_var") or (key.endswith(".weight") and len(shape) == 1):
            v = detdata.t_uniform(shape, s, 0.6, 1.4)
        elif len(shape) == 1:
            v = detdata.t_uniform(shape, s, -0.2, 0.2)
        else:
            fan_in = shape[1] * shape[2] * shape[3]
            v = detdata.t_uniform(shape, s, -1.0, 1.0) * (3.0 / fan_in) ** 0.5
        P[key] = v.float()
    return P


def ssr_upsample(P, depth_low, weights, pred_label, key="ssr_upsample"):
    """depth_low [B,1,h,w], weights (spx_pred) [B,n,4h,4w], pred_label [B,n,4h,4w] -> [B,4h,4w]:
    4x bilinear up-sampling of the 1/4-scale disparity plus a residual gated by the class
    probabilities (two 1x1 conv + sigmoid stages modulated by `weights`)."""
    b, c, h, w = depth_low.shape
    label = F.softmax(pred_label, dim=1)
    depth_ = F.interpolate(depth_low, (h * 4, w * 4), mode="bilinear").reshape(b, 1, h * 4, w * 4)
    d = bn(P, key + ".conv.0", depth_)
    d = F.conv2d(d, P[key + ".conv.1.weight"], P[key + ".conv.1.bias"], 1, 1)
    d = bn(P, key + ".conv.2", d)
    prob = torch.sigmoid(bn(P, key + ".conv1.1", F.conv2d(label * weights, P[key + ".conv1.0.weight"], P[key + ".conv1.0.bias"])))
    prob = torch.sigmoid(bn(P, key + ".conv2.1", F.conv2d(prob * weights, P[key + ".conv2.0.weight"], P[key + ".conv2.0.bias"])))
    res = F.conv2d(d * prob, P[key + ".conv3.weight"], P[key + ".conv3.bias"])
    return (depth_ + res).squeeze(1)
