"""Oracle restatements of the UNSIGNED-range op set (reference: /root/reference/models/submodule_.py -- the older
op library whose definitions models/SemStereo_WHU.py:279,305 is written for).  TEST INFRASTRUCTURE ONLY.

Disparities span [0, maxdisp): plane d of a volume pairs left column x with right column x - d.
"""
import torch

from .ops import groupwise_correlation, groupwise_correlation_norm  # noqa: F401  (range independent: submodule_.py:180-186, 200-209)


def _gwc_volume(ref, tgt, maxdisp, num_groups, corr):
    B, C, H, W = ref.shape
    vol = ref.new_zeros([B, num_groups, maxdisp, H, W])
    for d in range(maxdisp):                     # models/submodule_.py:191-196 / 214-219
        if d >= W:
            continue                             # (the reference's slice assignment is empty there)
        if d > 0:
            vol[:, :, d, :, d:] = corr(ref[:, :, :, d:], tgt[:, :, :, :W - d], num_groups)
        else:
            vol[:, :, d] = corr(ref, tgt, num_groups)
    return vol.contiguous()


def build_gwc_volume(refimg_fea, targetimg_fea, maxdisp, num_groups):
    """models/submodule_.py:188-198."""
    return _gwc_volume(refimg_fea, targetimg_fea, maxdisp, num_groups, groupwise_correlation)


def build_gwc_volume_norm(refimg_fea, targetimg_fea, maxdisp, num_groups):
    """models/submodule_.py:211-221."""
    return _gwc_volume(refimg_fea, targetimg_fea, maxdisp, num_groups, groupwise_correlation_norm)


def build_concat_volume(refimg_fea, targetimg_fea, maxdisp):
    """models/submodule_.py:166-177: the left half is the left image on EVERY plane (unmasked); the right half is the
    right image shifted by d, zero for x < d."""
    B, C, H, W = refimg_fea.shape
    vol = refimg_fea.new_zeros([B, 2 * C, maxdisp, H, W])
    for d in range(maxdisp):
        vol[:, :C, d] = refimg_fea
        if d == 0:
            vol[:, C:, d] = targetimg_fea
        elif d < W:
            vol[:, C:, d, :, d:] = targetimg_fea[:, :, :, :W - d]
    return vol.contiguous()


def disparity_regression(x, maxdisp):
    """models/submodule_.py:159-163."""
    assert len(x.shape) == 4
    values = torch.arange(0, maxdisp, dtype=x.dtype, device=x.device).reshape(1, maxdisp, 1, 1)
    return torch.sum(x * values, 1, keepdim=False)


def disparity_variance(x, maxdisp, disparity):
    """models/submodule_.py:239-245."""
    assert len(x.shape) == 4
    values = torch.arange(0, maxdisp, dtype=x.dtype, device=x.device).reshape(1, maxdisp, 1, 1)
    return torch.sum(x * (values - disparity) ** 2, 1, keepdim=True)
