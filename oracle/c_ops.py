"""ctypes loader of the oracle's plain-C restatement (oracle_ops.c).  TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle_ops.so")
_lib = None
_F = ctypes.POINTER(ctypes.c_float)


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.check_call(["make", "-C", _HERE, "liboracle_ops.so"], stdout=subprocess.DEVNULL)
        _lib = ctypes.CDLL(_SO)
    return _lib


def _p(t):
    return None if t is None else t.numpy().ctypes.data_as(_F)


def _f(t):
    return t.detach().contiguous().float()


def gwc_volume(ref, tgt, m, G, normalize):
    ref, tgt = _f(ref), _f(tgt)
    B, C, H, W = ref.shape
    out = torch.empty(B, G, 2 * m, H, W)
    load().orc_gwc_volume(_p(ref), _p(tgt), _p(out), B, C, H, W, m, G, int(normalize))
    return out


def concat_volume(ref, tgt, m):
    ref, tgt = _f(ref), _f(tgt)
    B, C, H, W = ref.shape
    out = torch.empty(B, 2 * C, 2 * m, H, W)
    load().orc_concat_volume(_p(ref), _p(tgt), _p(out), B, C, H, W, m)
    return out


def disparity_regression(prob, m, disp=None):
    prob = _f(prob)
    B, D, H, W = prob.shape
    disp = None if disp is None else _f(disp)
    out = torch.empty(B, H, W)
    load().orc_disparity_regression(_p(prob), _p(disp), _p(out), B, m, H, W)
    return out


def regression_topk(cost, samples, k):
    cost, samples = _f(cost), _f(samples)
    B, nd, H, W = cost.shape
    out = torch.empty(B, 1, H, W)
    load().orc_regression_topk(_p(cost), _p(samples), _p(out), B, nd, H, W, k)
    return out


def warp_sampled(x, y, disp):
    x, y, disp = _f(x), _f(y), _f(disp)
    B, C, H, W = y.shape
    nd = disp.shape[1]
    yw, xw = torch.empty(B, C, nd, H, W), torch.empty(B, C, nd, H, W)
    load().orc_warp_sampled(_p(x), _p(y), _p(disp), _p(yw), _p(xw), B, C, H, W, nd)
    return yw, xw
