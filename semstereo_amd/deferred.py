"""Deferred evaluation at the op / module boundary: the cross-line fusions of the hot segment with the reference's
`forward()` left exactly as it is.

`models/SemStereo.py:273-323` is ~40 statements: calls of the op library (looked up by bare name: `install()` binds
this repo's ops), calls of sub-modules (`accelerate()` swaps in this repo's twins) and plain `torch` / `F` glue between
them.  The fused HIP kernels each cover several of those statements (`:273-276`, `:279-285`, `:286-293`, `:295-310`,
`:316-320`), so an op-by-op drop-in cannot use them -- unless the ops and twins do not compute right away.  In
inference they return a `Deferred`: a tensor-LIKE handle (the `__torch_function__` protocol, as NumPy-style duck
arrays use it) that records the call instead of executing it.  `torch` / `F` functions and tensor methods applied to a
handle give further handles; nothing runs until a value is needed -- by one of this repo's ops or twins, by an
untraced function, or by attribute access.  At that point the recorded expression is compared with the statement
sequences of the reference; where one matches, the fused kernel produces the value (and those of its sibling
expressions), otherwise the recorded calls are replayed one by one with the very functions and arguments the caller
used.  Every deviation from the reference's text -- another order, another dim, an extra op -- therefore falls back to
the line-by-line result: matching is an optimisation, never a semantic decision.

Handles are created only when `on(...)` holds (eval mode, no autograd, HIP tensors, `SS_DEFER` != 0): training and
autograd calls see ordinary tensors throughout.
"""
import contextlib
import os
import threading

import torch

ENABLED = os.environ.get("SS_DEFER", "1") != "0"
#: which rules fired (tests and bench.py read this); "ssr": SSR_upsample calls that handed out a handle / handles that were ever
#: computed (the `pred_att_up` of an eval forward, models/SemStereo.py:311 vs :346, is handed out and never computed)
STATS = {"fused": {}, "replayed": 0, "ssr": {"deferred": 0, "computed": 0}}


_TLS = threading.local()


@contextlib.contextmanager
def suspended():
    """No handles inside this block (this thread): HotSegment's own fused composition calls the same ops and twins and
    wants their tensors."""
    _TLS.depth = getattr(_TLS, "depth", 0) + 1
    try:
        yield
    finally:
        _TLS.depth -= 1


def on(module, *tensors):
    """True when ops / twins may hand out handles: deferral enabled and the folded-BN inference path valid."""
    if not ENABLED or getattr(_TLS, "depth", 0):
        return False
    from . import modules as M
    real_ts = [t for t in tensors if isinstance(t, torch.Tensor)]
    if not all(t.is_cuda for t in real_ts):
        return False
    if module is not None:
        return M._inference(module, *real_ts)
    return not (torch.is_grad_enabled() and any(t.requires_grad for t in real_ts))


# ---- normalised names of the functions that are recorded (everything else forces the values) -------------------------
_NAMES = {
    "mul": "mul", "__mul__": "mul", "__rmul__": "mul", "multiply": "mul",
    "add": "add", "__add__": "add", "__radd__": "add",
    "sub": "sub", "__sub__": "sub", "subtract": "sub",
    "sigmoid": "sigmoid", "softmax": "softmax", "sum": "sum", "mean": "mean", "unsqueeze": "unsqueeze", "squeeze": "squeeze",
    "sort": "sort", "__getitem__": "getitem", "gather": "gather", "float": "float", "interpolate": "interpolate", "cat": "cat",
}
_METHODS = ("mean", "sum", "unsqueeze", "squeeze", "sort", "float", "softmax", "sigmoid", "gather")


def _fname(func):
    return _NAMES.get(getattr(func, "__name__", ""), None)


_STATS_LOCK = threading.Lock()


def _fused(rule):
    with _STATS_LOCK:                       # (nn.DataParallel: one thread per replica)
        STATS["fused"][rule] = STATS["fused"].get(rule, 0) + 1


class Deferred:
    """A value that has not been computed: op name + the recorded call (func, args, kwargs)."""
    __slots__ = ("op", "func", "args", "kwargs", "_value", "info", "__weakref__")

    def __init__(self, op, func, args, kwargs=None, value=None):
        self.op, self.func, self.args, self.kwargs = op, func, tuple(args), dict(kwargs or {})
        self._value = value
        self.info = {}
        if op == "sub":
            _note_offset(self)

    # -- construction helpers ------------------------------------------------------------------------------------------
    @staticmethod
    def leaf(t, role=None):
        """A handle around a tensor that already exists (so that what is done with it next gets recorded)."""
        d = Deferred("leaf", None, (), None, t)
        if role:
            d.info["role"] = role
        return d

    @staticmethod
    def call(op, func, *args, **kwargs):
        """A handle for `func(*args, **kwargs)` (func: any callable taking the realised arguments)."""
        return Deferred(op, func, args, kwargs)

    @staticmethod
    def pair(node):
        """Two handles for the two results of a tuple-valued call."""
        return Deferred("item0", None, (node,)), Deferred("item1", None, (node,))

    # -- evaluation -----------------------------------------------------------------------------------------------------
    @property
    def done(self):
        return self._value is not None

    def value(self):
        if self._value is None:
            # handles exist only where nothing needs autograd (on()); the value may be asked for later, outside the caller's
            # no_grad block -- it is computed as it would have been at the call
            with torch.no_grad():
                for rule in _VALUE_RULES.get(self.op, ()):
                    if rule(self) and self._value is not None:
                        break
                if self._value is None:
                    self._value = self._replay()
        return self._value

    def _replay(self):
        if self.op in ("item0", "item1"):
            return self.args[0].value()[int(self.op[-1])]
        STATS["replayed"] += 1
        args, kwargs = real(self.args), real(self.kwargs)
        with suspended():                                # the recorded call itself, computing (no handles out of a replay)
            value = self.func(*args, **kwargs)
        #  `ind_k.squeeze(1).float() - maxdisp // 4` replayed statement by statement is a fresh tensor: its provenance --
        # integer indices minus an integer -- is known HERE, so it is marked as integer candidates and the gathered stem does not pay
        # a device reduction and a host sync to find that out (ops.integer_candidates)
        if self.op in ("sub", "float") and isinstance(value, torch.Tensor):
            m = _samples_index_node(self)
            if m is not None and float(m[1]) == int(m[1]):
                from . import ops
                try:
                    ops._mark_integer(value, True)
                except Exception:       # noqa: BLE001
                    pass
        return value

    # -- the tensor-like protocol --------------------------------------------------------------------------------------
    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        op = _fname(func)
        # not a recorded function: it gets the values.  Nor is anything recorded on top of the SSR head's result: that handle exists
        # so that a result nobody touches is never computed (models/SemStereo.py:311 vs :346); whatever IS done with it -- the
        # `* 4` of the return statement -- computes, so that forward() returns tensors
        if op is None or any(_is(a, "ssr") for a in args):
            return func(*real(args), **real(kwargs))
        node = Deferred(op, func, args, kwargs)
        return Deferred.pair(node) if op == "sort" else node

    def _method(self, name):
        func = getattr(torch.Tensor, name)

        def bound(*a, **k):
            node = Deferred(_NAMES[name], func, (self,) + a, k)
            return Deferred.pair(node) if name == "sort" else node
        return bound

    def __getattr__(self, name):
        # (only reached for names not defined on the class: tensor methods and attributes)
        if name in _METHODS and self.op != "ssr":
            return self._method(name)
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return getattr(self.value(), name)               # .shape, .size(), .cpu(), .item(), ...: the real tensor's

    def __mul__(self, other):
        if self.op == "ssr":
            return torch.mul(*real((self, other)))
        return Deferred("mul", torch.mul, (self, other))

    def __rmul__(self, other):
        if self.op == "ssr":
            return torch.mul(*real((other, self)))
        return Deferred("mul", torch.mul, (other, self))

    def __add__(self, other):
        if self.op == "ssr":
            return torch.add(*real((self, other)))
        return Deferred("add", torch.add, (self, other))

    def __radd__(self, other):
        return Deferred("add", torch.add, (other, self))

    def __sub__(self, other):
        return Deferred("sub", torch.sub, (self, other))

    def __rsub__(self, other):
        return torch.sub(*real((other, self)))

    def __truediv__(self, other):
        return torch.div(*real((self, other)))

    def __getitem__(self, idx):
        return Deferred("getitem", torch.Tensor.__getitem__, (self, idx))

    def __len__(self):
        return len(self.value())

    def __bool__(self):
        return bool(self.value())

    def __repr__(self):
        return f"Deferred<{self.op}{'' if self._value is None else ' =' + str(tuple(self._value.shape))}>"


def real(x):
    """Realise every handle inside x (a handle, a tensor, or a tuple / list / dict of those)."""
    if isinstance(x, Deferred):
        return x.value()
    if isinstance(x, tuple):
        return tuple(real(v) for v in x)
    if isinstance(x, list):
        return [real(v) for v in x]
    if isinstance(x, dict):
        return {k: real(v) for k, v in x.items()}
    return x


def realising(fn):
    """Decorator for ops that have no deferral rule of their own: handles in the arguments are realised first."""
    import functools

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        return fn(*real(args), **real(kwargs))
    return wrapper


# ---- small matching helpers ---------------------------------------------------------------------------------------------

def _is(x, op):
    return isinstance(x, Deferred) and x.op == op


def _arg(node, pos, key, default=None):
    """Argument `key` of the recorded call, given positionally at `pos` or by keyword."""
    if key in node.kwargs:
        return node.kwargs[key]
    return node.args[pos] if len(node.args) > pos else default


_NEUTRAL = {"dtype": (None,), "_stacklevel": None, "recompute_scale_factor": (None,), "antialias": (False, None), "out": (None,),
            "align_corners": (None, False), "scale_factor": (None,), "stable": None, "alpha": (1,)}


def _plain(node, *used):
    """True when the recorded call carries nothing beyond the arguments in `used` that could change its meaning: F.softmax /
    F.interpolate forward ALL their parameters as keywords (dtype=None, _stacklevel=3, antialias=False, ...)."""
    for k, v in node.kwargs.items():
        if k in used:
            continue
        ok = _NEUTRAL.get(k, ())
        if ok is not None and not any(v is o or v == o for o in ok):
            return False
    return True


def _is_softmax(node, dim):
    return _is(node, "softmax") and _arg(node, 1, "dim") == dim and len(node.args) <= 2 and _plain(node, "dim", "input")


def _binary(node, op):
    """(a, b) of a recorded binary call of kind `op`, else None."""
    if not _is(node, op) or len(node.args) != 2 or node.kwargs:
        return None
    return node.args


def _either(pair, pred_a, pred_b):
    """Order the two operands of a commutative call so that pred_a(a) and pred_b(b); else None."""
    if pair is None:
        return None
    a, b = pair
    if pred_a(a) and pred_b(b):
        return a, b
    if pred_a(b) and pred_b(a):
        return b, a
    return None


def _tensorish(x):
    return isinstance(x, (torch.Tensor, Deferred))


def _scalar_param(x):
    return isinstance(x, torch.Tensor) and x.numel() == 1


def _note_offset(node):
    """`float(squeeze(IND, 1)) - c`: remember c on IND (the candidate indices become disparities, models/SemStereo.py:305)."""
    a, c = node.args[0], node.args[1] if len(node.args) > 1 else None
    if isinstance(c, (int, float)) and _is(a, "float") and _is(a.args[0], "squeeze") and _arg(a.args[0], 1, "dim") == 1:
        ind = a.args[0].args[0]
        if isinstance(ind, Deferred):
            ind.info["offset"] = c


# ---- rule: F.softmax(torch.squeeze(F.interpolate(COARSE, size, mode='trilinear'), 1), dim=1)   (:279-282) ------------------

def match_upsampled_prob(node):
    """-> (interpolate node, coarse tensor, size) when `node` is the soft-max over the disparity axis of the squeezed
    trilinear up-sampling of a [B,1,D,H,W] tensor, else None."""
    if not _is_softmax(node, 1):
        return None
    sq = node.args[0]
    if not _is(sq, "squeeze") or _arg(sq, 1, "dim") != 1:
        return None
    up = sq.args[0]
    if not _is(up, "interpolate"):
        return None
    if _arg(up, 3, "mode", "nearest") != "trilinear" or _arg(up, 2, "scale_factor") is not None:
        return None
    if _arg(up, 4, "align_corners") not in (None, False) or len(up.args) > 5 or not _plain(up, "size", "mode", "input"):
        return None
    size = _arg(up, 1, "size")
    coarse = real(up.args[0])
    if size is None or len(size) != 3 or coarse.dim() != 5 or coarse.shape[1] != 1:
        return None
    return up, coarse, tuple(int(s) for s in size)


def regression_of(prob, rng):
    """Called by the `disparity_regression` op when its argument is a handle: models/SemStereo.py:279-283 fused.
    -> pred0 or None (no match / shape not built)."""
    from . import ops
    m = match_upsampled_prob(prob)
    if m is None:
        return None
    up, coarse, size = m
    if size[0] != rng[1] or not ops.upsample_softmax_regression_applies(coarse, None, size[1], size[2], rng):
        return None
    if "fused" not in prob.info:
        att_weights, pred0, var = ops.upsample_softmax_regression(coarse, None, size[1], size[2], _range=rng)
        up._value = att_weights                                   # the up-sampled logits are wanted again at :295
        prob.info["fused"] = (pred0, var, rng)
        _fused("upsample_softmax_regression")
    return prob.info["fused"][0]


def variance_of(prob, rng, disparity):
    """`disparity_variance(prob, m, pred0.unsqueeze(1))` after regression_of(prob): the fused kernel's variance (as a leaf
    handle: what is done with it next is :286-293), or None."""
    hit = prob.info.get("fused") if isinstance(prob, Deferred) else None
    if hit is None or hit[2] != rng or not isinstance(disparity, torch.Tensor):
        return None
    pred0, var, _ = hit
    if disparity.data_ptr() != pred0.data_ptr() or disparity.numel() != pred0.numel():
        return None
    return Deferred.leaf(var, role="variance")


# ---- rule: the 5-candidate probe, :286-293 -------------------------------------------------------------------------------

def _match_strength(node):
    """softmax(mul(mean(mul(STN.left, STN.right), dim=1), propagation(sigmoid(add(BETA, mul(GAMMA, VAR))))), dim=1) with
    STN = SpatialTransformer_grid(FL, FR, propagation(unsqueeze(PRED0, 1)))  ->  (fl, fr, pred0, var, gamma, beta)."""
    if not _is_softmax(node, 1):
        return None
    ab = _either(_binary(node.args[0], "mul"), lambda x: _is(x, "mean"), lambda x: _is(x, "propagation"))
    if ab is None:
        return None
    mean, vs = ab
    if _arg(mean, 1, "dim") != 1 or _arg(mean, 2, "keepdim", False) or len(mean.args) > 3 or not _plain(mean, "dim", "keepdim"):
        return None
    lr = _either(_binary(mean.args[0], "mul"), lambda x: _is(x, "item1"), lambda x: _is(x, "item0"))
    if lr is None or lr[0].args[0] is not lr[1].args[0] or not _is(lr[0].args[0], "stn"):
        return None
    fl, fr, disp = lr[0].args[0].args
    if not _is(disp, "propagation"):
        return None
    p0 = disp.args[0]
    if _is(p0, "unsqueeze") and _arg(p0, 1, "dim") == 1:
        pred0 = real(p0.args[0])
    elif isinstance(p0, torch.Tensor) and p0.dim() == 4 and p0.shape[1] == 1:
        pred0 = p0.squeeze(1)
    else:
        return None
    sig = vs.args[0]
    if not _is(sig, "sigmoid") or len(sig.args) != 1:
        return None
    bg = _either(_binary(sig.args[0], "add"), _scalar_param, lambda x: _is(x, "mul"))
    if bg is None:
        return None
    beta, gv = bg
    gvar = _either(_binary(gv, "mul"), _scalar_param, _tensorish)
    if gvar is None:
        return None
    gamma, var = gvar[0], real(gvar[1])
    fl, fr = real(fl), real(fr)
    if not (isinstance(var, torch.Tensor) and var.dim() == 4 and var.shape[1] == 1 and pred0.dim() == 3
            and fl.dim() == 4 and fl.shape == fr.shape and pred0.shape == (fl.shape[0],) + tuple(fl.shape[2:])
            and var.shape[2:] == fl.shape[2:]):
        return None
    return fl, fr, pred0, var, gamma, beta


def _rule_strength(node):
    from . import ops
    m = _match_strength(node)
    if m is None:
        return False
    node._value = ops.sample_strength(*m)
    _fused("sample_strength")
    return True


# ---- rule: the top-24 selection, :295-310 --------------------------------------------------------------------------------

def _match_selected_indices(ik):
    """`ik` = sort(getitem(sort(softmax(AW, 2), 2, True).indices, [:, :, :k]), 2, False).values with
    AW = sum(mul(propagation_prob(UP), unsqueeze(STRENGTH, 2)), dim=1, keepdim=True) -> (aw node, awp node, up, strength, k)."""
    if not _is(ik, "item0") or not _is(ik.args[0], "sort"):
        return None
    s2 = ik.args[0]
    if _arg(s2, 1, "dim") != 2 or _arg(s2, 2, "descending", False) not in (False, 0):
        return None
    it = s2.args[0]
    if not _is(it, "getitem"):
        return None
    idx = it.args[1]
    full = slice(None, None, None)
    if not (isinstance(idx, tuple) and len(idx) == 3 and idx[0] == full and idx[1] == full and isinstance(idx[2], slice)
            and idx[2].start is None and idx[2].step is None and isinstance(idx[2].stop, int) and idx[2].stop > 0):
        return None
    k = idx[2].stop
    si = it.args[0]
    if not _is(si, "item1") or not _is(si.args[0], "sort"):
        return None
    s1 = si.args[0]
    if _arg(s1, 1, "dim") != 2 or _arg(s1, 2, "descending", False) not in (True, 1):      # (stable or not: ties -> lower index)
        return None
    awp = s1.args[0]
    if not _is_softmax(awp, 2):
        return None
    aw = awp.args[0]
    if not _is(aw, "sum") or _arg(aw, 1, "dim") != 1 or _arg(aw, 2, "keepdim", False) is not True or not _plain(aw, "dim", "keepdim"):
        return None
    ps = _either(_binary(aw.args[0], "mul"), lambda x: _is(x, "propagation_prob"), lambda x: _is(x, "unsqueeze"))
    if ps is None or _arg(ps[1], 1, "dim") != 2:
        return None
    return aw, awp, ps[0].args[0], ps[1].args[0], k


def _topk_group(ik):
    """Run ss_topk_candidates_fwd once for the selection `ik` belongs to; -> dict(att_topk, samples, pred_att, dmin) or None."""
    from . import ops
    if "topk" in ik.info:
        return ik.info["topk"]
    ik.info["topk"] = None
    m = _match_selected_indices(ik)
    if m is None:
        return None
    aw, awp, up, strength, k = m
    up, strength = real(up), real(strength)
    if not (isinstance(up, torch.Tensor) and up.dim() == 5 and up.shape[1] == 1 and strength.dim() == 4 and strength.shape[1] == 5
            and up.shape[2] <= ops.TOPK_CANDIDATES_MAX_D and k <= up.shape[2] and k in (6, 24, 32)):
        return None
    dmin = -int(ik.info.get("offset", 0))
    att_topk, samples, pred_att = ops.topk_candidates(up, strength, None, k, _range=(dmin, up.shape[2]))
    _fused("topk_candidates")
    ik.info["topk"] = dict(att_topk=att_topk, samples=samples, pred_att=pred_att, dmin=dmin, aw=aw, awp=awp)
    return ik.info["topk"]


def _rule_att_topk(node):                                   # torch.gather(AWP, 2, IK)                       (:304)
    if _arg(node, 1, "dim") != 2 or len(node.args) < 3 or not isinstance(node.args[2], Deferred):
        return False
    g = _topk_group(node.args[2])
    if g is None or node.args[0] is not g["awp"]:
        return False
    node._value = g["att_topk"]
    return True


def _samples_index_node(node):
    """IK when `node` is sub(float(squeeze(IK, 1)), c) or float(squeeze(IK, 1)), with the offset it implies; else None."""
    off = 0
    if _is(node, "sub"):
        if len(node.args) != 2 or not isinstance(node.args[1], (int, float)) or node.kwargs:
            return None
        off, node = node.args[1], node.args[0]
    if not _is(node, "float") or not _is(node.args[0], "squeeze") or _arg(node.args[0], 1, "dim") != 1:
        return None
    ik = node.args[0].args[0]
    return (ik, off) if isinstance(ik, Deferred) else None


def _rule_samples(node):                                    # IK.squeeze(1).float() - maxdisp // 4             (:305)
    m = _samples_index_node(node)
    if m is None:
        return False
    ik, off = m
    g = _topk_group(ik)
    if g is None or g["dmin"] != -int(off) or off != int(off):
        return False
    node._value = g["samples"]
    return True


def _rule_pred_att(node):                                   # sum(softmax(squeeze(gather(AW, 2, IK), 1), 1) * SAMPLES, dim=1)   (:307-310)
    if _arg(node, 1, "dim") != 1 or _arg(node, 2, "keepdim", False) or len(node.args) > 3 or not _plain(node, "dim", "keepdim"):
        return False
    pq = _either(_binary(node.args[0], "mul"), lambda x: _is(x, "softmax"), lambda x: _samples_index_node(x) is not None)
    if pq is None:
        return False
    prob, smp = pq
    ik, off = _samples_index_node(smp)
    if not _is_softmax(prob, 1) or not _is(prob.args[0], "squeeze") or _arg(prob.args[0], 1, "dim") != 1:
        return False
    ga = prob.args[0].args[0]
    if not _is(ga, "gather") or _arg(ga, 1, "dim") != 2 or len(ga.args) < 3 or ga.args[2] is not ik:
        return False
    g = _topk_group(ik)
    if g is None or ga.args[0] is not g["aw"] or g["dmin"] != -int(off):
        return False
    node._value = g["pred_att"]
    return True


# ---- rule: gwc volume -> patch -> gate, :273-276 ---------------------------------------------------------------------------

def gated_volume_of(cv, gate_module, im):
    """Called by the channelAtt twin when its volume argument is a handle: patch(build_gwc_volume_norm(...)) -> one kernel."""
    from . import modules as M, ops
    if not _is(cv, "patch") or cv.done:
        return None
    patch, vol = cv.args
    if not _is(vol, "gwc_norm") or vol.done or not isinstance(patch, M.DepthwisePatch):
        return None
    fl, fr, maxdisp, groups, rng = vol.args
    logits = gate_module.logits(im)
    if not ops.gwc_patch_gate_applies(fl, maxdisp, groups, rng, fr, logits):
        return None
    M.PATH_COUNTS["hip"] += 1
    _fused("gwc_patch_gate")
    return ops.gwc_patch_gate(fl, fr, maxdisp, groups, patch.weight, logits, _range=rng)


# ---- rule: sparse concat volume -> x att_topk -> concat_stem -> gate, :316-320 ----------------------------------------------

def _match_stem_input(vol):
    """mul(ATT, cat((STN.left, STN.right), dim=1)) with STN = SpatialTransformer_grid(CL, CR, SAMPLES) -> (cl, cr, samples, att)."""
    ac = _either(_binary(vol, "mul"), lambda x: not _is(x, "cat") and _tensorish(x), lambda x: _is(x, "cat"))
    if ac is None:
        return None
    att, cat = ac
    parts = cat.args[0]
    if _arg(cat, 1, "dim") != 1 or not isinstance(parts, (tuple, list)) or len(parts) != 2:
        return None
    left, right = parts
    if not (_is(left, "item1") and _is(right, "item0") and left.args[0] is right.args[0] and _is(left.args[0], "stn")):
        return None
    cl, cr, samples = left.args[0].args
    return cl, cr, samples, att


def stem_of(stem_node, gate_module=None, im=None):
    """Called by the channelAtt twin (handle made by the BasicConv twin for `concat_stem(volume)`) or by value(): the warped
    half of the volume, the broadcast half by linearity, the stem convolution and the gate -- the kernels of
    HotSegment.matching_branch.  -> tensor or None."""
    from . import modules as M, ops
    from .segment import HotSegment
    if not _is(stem_node, "stem") or stem_node.done:
        return None
    stem, vol = stem_node.args
    m = _match_stem_input(vol) if isinstance(vol, Deferred) and not vol.done else None
    if m is None or not isinstance(stem, M.BasicConv):
        return None
    cl, cr, samples, att = real(m)
    if not (cl.dim() == 4 and cl.shape == cr.shape and samples.dim() == 4 and att.dim() == 5 and att.shape[1] == 1
            and att.shape[2:] == samples.shape[1:] and samples.shape[2:] == cl.shape[2:]):
        return None
    gate = None if gate_module is None else gate_module.logits(im, sigmoid=True)
    _fused("stem_by_halves" if gate_module is not None else "stem_by_halves_ungated")
    if (M.CONV_ENGINE != "f32" and samples.shape[1] in (6, 24, 32) and HotSegment.STEM_BY_HALVES
            and stem.conv.in_channels == 2 * cl.shape[1] and M._conv_geometry(stem.conv) == (3, 1)):
        partial = M.stem_broadcast_half(stem, cl, att)
        if M.stem_gather_applies(stem, cr, samples) and ops.integer_candidates(samples):
            return M.stem_gather_half(stem, cr, samples, att, partial, gate, consume_partial=True)
        if M.stem_presplit_applies(stem, cr):
            xs, xexp = ops.concat_volume_sampled_presplit(cr, samples, att)
            return M.stem_volume_half_presplit(stem, xs, xexp, partial, gate)
        right = ops.concat_volume_sampled(None, cr, samples, att)
        return M.stem_volume_half(stem, right, partial, gate)
    volume = ops.concat_volume_sampled(cl, cr, samples, att)
    return stem(volume, gate=gate)


def _rule_stem(node):
    v = stem_of(node)
    if v is None:
        return False
    node._value = v
    return True


# ---- rule: concat_feature(left view), concat_feature(right view), :314-315 -> one pair of launches ---------------------------------

def _rule_cfeat(node):
    """The value of one view's concat features is asked for while the other view's handle of the SAME module is still pending:
    both through ss_conv2d_bf16s_pair_fwd + one batch-2B launch of the second layer.  Anything else: the recorded call."""
    from . import modules as M
    from .segment import HotSegment
    mod = node.info.get("module")
    if mod is None or not HotSegment.PAIR_VIEWS or len(mod) != 2:
        return False
    x = real(node.args[0])
    for ref in mod.__dict__.get("_ss_pending_cf", []):
        other = ref()
        if other is None or other is node or other.done or other.info.get("module") is not mod:
            continue
        y = real(other.args[0])
        if not (isinstance(y, torch.Tensor) and y.shape == x.shape and y.device == x.device and y.dtype == x.dtype):
            continue
        a, b = mod[0], mod[1]
        if not (isinstance(getattr(a, "conv", None), torch.nn.Conv2d) and getattr(a, "relu", False) and isinstance(b, torch.nn.Conv2d)
                and M._inference(mod, x, y)):
            return False
        with suspended():
            bn = a.bn if getattr(a, "use_bn", True) else None
            h = M.run_conv2d_pair(a, "bc2d", a.conv, bn, x, y, True)
            z = M.run_conv2d(mod, "cf1", b, None, h, False) if h is not None else None
        if z is None:
            return False
        n = x.shape[0]
        node._value, other._value = z[:n], z[n:]
        _fused("concat_feature_pair")
        return True
    return False


_VALUE_RULES = {
    "softmax": (_rule_strength,),
    "gather": (_rule_att_topk,),
    "sub": (_rule_samples,),
    "float": (_rule_samples,),
    "sum": (_rule_pred_att,),
    "stem": (_rule_stem,),
    "cfeat": (_rule_cfeat,),
}
