"""Drop-in installation into the reference's own model code.

The reference has no plugin registry: `models/SemStereo.py` star-imports the op library and looks
the functions up BY BARE NAME in its module globals at call time (models/SemStereo.py:7-8, 273,
283, 285, 291, 316, 323), and holds the 3-D stack as nn.Module attributes created in __init__
(:219-236).  So a drop-in is two steps, both leaving `forward()` untouched:

    import models.SemStereo as ms                  # the reference's module
    import semstereo_amd
    semstereo_amd.install(ms)                      # rebind the op-library names in ITS globals
    net = ms.SemStereo(...); net.load_state_dict(ckpt); net.cuda().eval()
    semstereo_amd.accelerate(net)                  # swap hourglass/classifier/... for HIP-backed twins
"""
import torch.nn as nn

from . import modules as M
from . import ops


def install(model_module, names=ops.REFERENCE_NAMES):
    """setattr(model_module, name, hip_op) for every hot-path callable.  Rebinding
    `models.submodule.X` alone would not be enough: the star-import copied the binding.
    Returns {name: previous object} so `uninstall` can restore it."""
    previous = {}
    for name in names:
        previous[name] = getattr(model_module, name, None)
        setattr(model_module, name, getattr(ops, name))
    return previous


def uninstall(model_module, previous):
    for name, obj in previous.items():
        if obj is None:
            if hasattr(model_module, name):
                delattr(model_module, name)
        else:
            setattr(model_module, name, obj)


# attribute of the reference SemStereo instance -> adopting class
_SWAPS = {
    "hourglass_att": M.hourglass,
    "hourglass": M.hourglass2,
    "classif_att_": M.Classifier,
    "classif": M.Classifier,
    "concat_stem": M.BasicConv,
    "patch": M.DepthwisePatch,
    "corr_feature_att_8": M.channelAtt,
    "concat_feature_att_4": M.channelAtt,
    "ssr_upsample": M.SSR_upsample,
}


def accelerate(model):
    """Replace the hot-path sub-modules of a reference `SemStereo` instance (or of the .module of
    its nn.DataParallel wrapper) by HIP-backed twins that SHARE its parameters; state_dict keys and
    values are unchanged.  Returns the list of swapped attribute names."""
    target = model.module if isinstance(model, nn.DataParallel) else model
    done = []
    for name, cls in _SWAPS.items():
        sub = getattr(target, name, None)
        if sub is None or isinstance(sub, cls):
            continue
        before = list(sub.state_dict().keys())
        new = cls.adopt(sub)
        assert list(new.state_dict().keys()) == before, f"state_dict keys of {name} changed"
        setattr(target, name, new)
        done.append(name)
    return done
