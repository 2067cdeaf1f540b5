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

That alone runs every hot op and module of the reference's forward() on the HIP kernels (measured on the
bench shape: 190 pairs/s for the hot segment).  The fusions that cross the lines of forward() -- patch+gate,
softmax+regression+variance, the two attention-tail kernels, warp+concat+gate, stem+gate, the two-stream
overlap (260 pairs/s) -- need the inference call path itself:

    semstereo_amd.accelerate(net, fuse_forward=True)   # eval()/no_grad calls take the fused path; training,
                                                       # autograd and seg-only calls still run the reference forward()
"""
import torch
import torch.nn as nn

from . import modules as M
from . import ops
from . import segment


def install(model_module, names=ops.REFERENCE_NAMES, unsigned=False):
    """setattr(model_module, name, hip_op) for every hot-path callable.  Rebinding
    `models.submodule.X` alone would not be enough: the star-import copied the binding.
    `unsigned`: the unsigned-range op set (`ops_unsigned`, models/submodule_.py's definitions) -- what
    models/SemStereo_WHU.py needs in its globals to run at all (see ops_unsigned's docstring).
    Returns {name: previous object} so `uninstall` can restore it."""
    from . import ops_unsigned
    lib = ops_unsigned if unsigned else ops
    previous = {}
    for name in names:
        previous[name] = getattr(model_module, name, None)
        setattr(model_module, name, getattr(lib, name))
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
    "concat_feature": M.ConcatFeature,
    "patch": M.DepthwisePatch,
    "corr_feature_att_8": M.channelAtt,
    "concat_feature_att_4": M.channelAtt,
    "ssr_upsample": M.SSR_upsample,
    "propagation": M.Propagation,              # parameter-free: one-hot convolutions -> shifted views
    "propagation_prob": M.Propagation_prob,
}


def accelerate(model, fuse_forward=False):
    """Replace the hot-path sub-modules of a reference `SemStereo` instance (or of the .module of
    its nn.DataParallel wrapper) by HIP-backed twins that SHARE its parameters; state_dict keys and
    values are unchanged.  With `fuse_forward` the instance's forward is additionally routed through
    `fused_inference_forward` for inference calls.  Returns the list of swapped attribute names."""
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
    if fuse_forward and not getattr(type(target), "_ss_fused_forward", False):
        target.__class__ = _fused_class(type(target))
    if not target.__dict__.get("_ss_output_hook"):
        # deferred handles (deferred.py) never leave the model: whatever forward() returns is realised on the way out -- since r05
        # also the SSR head's result (`pred_up * 4`, models/SemStereo.py:346); a handle that is NOT part of the output, like the
        # `pred_att_up` of an eval forward (:311), is simply never computed
        target.register_forward_hook(_realise_outputs)
        target.__dict__["_ss_output_hook"] = True
    return done


def _realise_outputs(module, inputs, output):
    from . import deferred as dfr
    M.drop_parked_gates(module)       #  a class gate parked by an odd number of SSR_upsample calls does not outlive the forward
    return dfr.real(output)


_FUSED_CLASSES = {}


def _fused_class(base):
    """A subclass of the model's own class whose forward() is `fused_inference_forward`; the reference's forward()
    stays reachable as the base class's.  The routing lives on the CLASS, not in the instance __dict__:
    nn.DataParallel builds its per-GPU replicas with `replica = cls.__new__(cls); replica.__dict__ =
    module.__dict__.copy()` (torch/nn/modules/module.py: _replicate_for_data_parallel), so a bound method stored in
    the instance would keep pointing every replica at the cuda:0 original (main_us3d.py:100, test_us3d.py:58)."""
    cls = _FUSED_CLASSES.get(base)
    if cls is None:
        def forward(self, left, right):
            return fused_inference_forward(self, left, right, reference_forward=super(cls, self).forward)
        def reduce_ex(self, protocol):
            # whole-model pickling (torch.save(model), multiprocessing spawn): as an instance of the reference's own class --
            # this subclass exists only in the running process (accelerate(..., fuse_forward=True) again after loading)
            return (_rebuild_as, (base, self.__dict__.copy()))
        cls = type(base.__name__, (base,), {"forward": forward, "_ss_fused_forward": True, "_ss_reference_class": base,
                                            "__reduce_ex__": reduce_ex,
                                            "__module__": base.__module__, "__qualname__": base.__qualname__})
        _FUSED_CLASSES[base] = cls
    return cls


def _rebuild_as(base, state):
    obj = base.__new__(base)
    obj.__dict__.update(state)
    return obj


def restore_forward(model):
    """Undo `accelerate(..., fuse_forward=True)`'s forward routing (the swapped sub-modules stay)."""
    target = model.module if isinstance(model, nn.DataParallel) else model
    if getattr(type(target), "_ss_fused_forward", False):
        target.__class__ = type(target)._ss_reference_class


def fused_inference_forward(self, left, right, reference_forward=None):
    """The caller side of the hot path, models/SemStereo.py:246-346, for eval-mode / no-autograd calls: the
    same sub-module calls in the same order as the reference's forward() around `segment.run_segment`
    (:273-323 fused).  Returns exactly what the reference returns in eval mode: `[disp_full_res]` or
    `([disp_full_res], label_logits)`, disp = 4 * SSR_upsample(pred).  Everything else (training, autograd,
    segmentation-only models) is handed to the reference's own forward()."""
    if reference_forward is None:
        ref_cls = getattr(type(self), "_ss_reference_class", None)
        assert ref_cls is not None or type(self).forward is not fused_inference_forward, "no reference forward() to hand the call to"
        reference_forward = (ref_cls or type(self)).forward.__get__(self)
    if (self.training or not self.stereo_if or not left.is_cuda
            or (torch.is_grad_enabled() and (left.requires_grad or right.requires_grad
                                             or any(p.requires_grad for p in self.parameters())))):
        return reference_forward(left, right)
    fl, fr = self.feature(left), self.feature(right)                                         # :248-249
    fl, fr = self.feature_up(fl, fr)                                                         # :251
    fl, fr = list(fl), list(fr)
    # :253-255; without seg_if the reference itself fails at ssr_upsample (pred_label undefined), so the
    # label logits are computed whenever the stereo branch runs
    pred_label = self.head_l(fl[0])
    for i, name in enumerate(("chal_0", "chal_1", "chal_2", "chal_3", "chal_4")):            # :258-262
        fl[i] = getattr(self, name)(fl[i])
    fr[1], fr[2] = self.chal_1(fr[1]), self.chal_2(fr[2])                                    # :264-265
    xspx = self.spx32_16(fl[4], fl[3])                                                       # :267-271
    xspx = self.spx16_8(xspx, fl[2])
    xspx = self.spx8_4(xspx, fl[1])
    xspx = self.spx4_2(xspx, fl[0])
    spx_pred = self.spx2(xspx)
    r = segment.run_segment(self, fl[1], fr[1], fl[2], fr[2], matching=not self.att_weights_only)   # :273-323
    if self.att_weights_only:
        disp = self.ssr_upsample(r["pred_att"].unsqueeze(1), spx_pred, pred_label)           # :311
    else:
        disp = self.ssr_upsample(r["pred"], spx_pred, pred_label)                            # :324
    from . import deferred as dfr
    return dfr.real(([disp * 4], pred_label) if self.seg_if else [disp * 4])                 # :340-346
