"""The UNSIGNED-range op set: drop-in replacements for the callables of the reference's other op library,
/root/reference/models/submodule_.py (disparities [0, maxdisp), `maxdisp` planes) -- the one
models/SemStereo_WHU.py is written for: it interpolates the attention logits to maxdisp//4 planes (:279) and takes the
candidate indices as disparities without an offset (:305), which only type-checks against these definitions
(with models/submodule.py's signed ones, which it star-imports as shipped, disparity_regression fails on the first call).

Same kernels as `semstereo_amd.ops`, launched over the range (dmin = 0, ndisp = maxdisp).  Range-independent callables
(groupwise_correlation[_norm], SpatialTransformer_grid, regression_topk) are the same objects as in `ops`.
"""
from . import ops
from .ops import (SpatialTransformer_grid, groupwise_correlation, groupwise_correlation_norm,  # noqa: F401
                  regression_topk, unsigned_range)


def build_gwc_volume(refimg_fea, targetimg_fea, maxdisp, num_groups):
    """models/submodule_.py:188-198 -> [B, G, maxdisp, H, W]."""
    return ops._build_gwc_volume(refimg_fea, targetimg_fea, maxdisp, num_groups, _range=unsigned_range(maxdisp))


def build_gwc_volume_norm(refimg_fea, targetimg_fea, maxdisp, num_groups):
    """models/submodule_.py:211-221 -> [B, G, maxdisp, H, W]."""
    return ops._build_gwc_volume_norm(refimg_fea, targetimg_fea, maxdisp, num_groups, _range=unsigned_range(maxdisp))


def build_concat_volume(refimg_fea, targetimg_fea, maxdisp):
    """models/submodule_.py:166-177 -> [B, 2C, maxdisp, H, W]; the left half is copied UNMASKED, only the shifted right
    half is zero where its partner column leaves the image."""
    return ops._build_concat_volume(refimg_fea, targetimg_fea, maxdisp, _range=unsigned_range(maxdisp), _mask_left=False)


def disparity_regression(x, maxdisp):
    """models/submodule_.py:159-163: [B, maxdisp, H, W] -> [B, H, W], disparity values 0 .. maxdisp-1."""
    return ops._disparity_regression(x, maxdisp, _range=unsigned_range(maxdisp))


def disparity_variance(x, maxdisp, disparity):
    """models/submodule_.py:239-245."""
    return ops._disparity_variance(x, maxdisp, disparity, _range=unsigned_range(maxdisp))


#: names models/SemStereo_WHU.py resolves by bare global
REFERENCE_NAMES = ops.REFERENCE_NAMES
