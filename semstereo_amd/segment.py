"""The hot segment of SemStereo.forward (reference models/SemStereo.py:273-323) as one module:
1/8- and 1/4-scale feature maps (after the chal_* projections) -> 1/4-scale disparities.

Attribute names equal the reference model's (patch, corr_feature_att_8, hourglass_att,
classif_att_, gamma, beta, concat_feature, concat_stem, concat_feature_att_4, hourglass, classif),
so the matching slice of a reference checkpoint loads with `load_reference_state_dict`.
In inference every line of that segment runs in hand-written HIP kernels, with the fusions SURVEY.md
section 8(f) asks for (volume+patch+gate, up-sampling+softmax+regression+variance, the 5-candidate probe,
the top-24 selection, warp+concat+gate, stem by halves + gate); with autograd on (training) the same
graph runs line by line on the reference-named ops (HIP forward / backward) and the module twins.
"""
import contextlib
import os
import threading

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import deferred as dfr
from . import modules as M
from . import ops, ops_unsigned
from . import train as T

TOPK = 24                                   # models/SemStereo.py:301

OWNED_PREFIXES = ("patch", "corr_feature_att_8", "hourglass_att", "classif_att_", "gamma", "beta",
                  "concat_feature", "concat_stem", "concat_feature_att_4", "hourglass", "classif")


_SIDE_STREAMS = {}
_TLS = threading.local()


@contextlib.contextmanager
def overlap_override(value):
    """The within-pair second stream forced on / off for the calls of THIS thread inside the block, whatever the module's or the
    class's OVERLAP says -- callers that bring their own concurrency (PairPipeline) pass their choice down this way instead of
    writing to the shared module (two pipelines, or a pipeline beside nn.DataParallel's replica threads, raced on that write)."""
    prev = getattr(_TLS, "overlap", None)
    _TLS.overlap = value
    try:
        yield
    finally:
        _TLS.overlap = prev


def _side_stream(device):
    """The second stream of the CALLING stream (callers that run consecutive pairs on several streams get one each)."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    st = _SIDE_STREAMS.get(key)
    if st is None:
        st = _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return st


propagation, propagation_prob = M.propagation, M.propagation_prob


class _Pending:
    """A tensor produced on the second stream: `value()` makes the consuming stream wait for the producing launch (once)."""

    def __init__(self, tensor, event, consumer):
        self.tensor, self.event, self.consumer, self.joined = tensor, event, consumer, False

    def value(self):
        if not self.joined:
            self.consumer.wait_event(self.event)
            self.tensor.record_stream(self.consumer)
            self.joined = True
        return self.tensor


def _ready(x):
    return x.value() if isinstance(x, _Pending) else x


class HotSegment(nn.Module):
    #: two-stream overlap of the branches (inference): "auto" = at batch <= 2 only.  Measured (profiles/r03_a_bench_b*_ov*.json):
    #: batch 1: 466.6 vs 462.4 pairs/s with / without; batch 4: 529.3 vs 531.1; batch 8: 538.7 vs 538.5 -- from batch 4 on the
    #: attention branch fills the chip by itself and the volume kernel runs 4x slower beside the 2-D convolutions (381 vs 107 us)
    OVERLAP = {"0": False, "1": True}.get(os.environ.get("SS_OVERLAP", "auto"), "auto")
    #: False: run the line-by-line composition of the reference's forward() (reference-named ops and modules,
    #: PyTorch glue in between) also in inference -- what a reference model gets from `install()` +
    #: `accelerate()` alone, with its forward() untouched.  True: the fused kernels of this file.
    FUSED = os.environ.get("SS_FUSED", "1") != "0"
    STEM_BY_HALVES = os.environ.get("SS_STEM_HALVES", "1") != "0"      # concat_stem's broadcast half by linearity
    PAIR_VIEWS = os.environ.get("SS_PAIR_VIEWS", "1") != "0"           # concat_feature on both views in one pair of launches
    GWC_PATCH_FUSED = os.environ.get("SS_GWC_PATCH_FUSED", "1") != "0"  # gwc volume -> patch -> gate in one kernel
    #: where the second stream's work (the matching branch's 2-D convolutions and gate) is released: "start" = with the attention
    #: branch (r01-r03), or after a layer of hourglass_att ("c2", "c3", "c4", "att", "u5").  r04: released at the start it ran beside
    #: the volume kernel and the first two convs -- kernels that fill the chip by themselves (the volume kernel took 98 us beside it,
    #: 36 alone); released after conv4 it runs beside the coarsest layers (128-256 workgroups each).  Interleaved A/B, three rounds
    #: (tools/ab_env.sh SS_PRELUDE_AT): start 474.1, c3 480.8, c4 482.0, att 481.0, u5 480.5 pairs/s.  "a,b,c" gives the left-view
    #: convs, the right-view convs and the gate their own points (also "cls" / "up" / "st": after classif_att_, the soft-argmax,
    #: the probe); each result is joined where the matching branch first needs it.  Releasing the later two later did not pay
    #: (another box): c4 464.8, "c4,c4,st" 464.8, "c4,cls,st" 461.7, "c4,cls,cls" 462.1, "c4,up,st" 449.5, "att,st,st" 448.3.
    #: r05 (the two views in one pair of launches, 110 instead of 173 us of side work; profiles/r05_o_prelude_at.txt, one stream, two sweeps):
    #: start 519.1 / 519.1, c2 528.1 / 527.3, c3 523.8 / 523.9, c4 523.4 / 523.1, att 521.5 / 520.6, u5 522.3 / 522.5 -> c2.
    PRELUDE_AT = os.environ.get("SS_PRELUDE_AT", "c2")

    def __init__(self, maxdisp, c8=256, c4=128, unsigned=False):
        """unsigned=False: models/SemStereo.py (disparities [-maxdisp, maxdisp), op set models/submodule.py);
        unsigned=True: models/SemStereo_WHU.py (:279 interpolates to maxdisp//4 planes, :305 takes the candidate indices
        as disparities) on the unsigned op set it is written for, models/submodule_.py (disparities [0, maxdisp))."""
        super().__init__()
        need = 128 if unsigned else 64
        assert maxdisp % need == 0, f"the reference graph needs maxdisp % {need} == 0 (SURVEY.md section 0.4)"
        self.maxdisp = maxdisp
        self.unsigned = bool(unsigned)
        self.gamma = nn.Parameter(torch.zeros(1))
        self.beta = nn.Parameter(2 * torch.ones(1))
        self.patch = M.DepthwisePatch(c8 // 8)
        self.concat_feature = M.ConcatFeature(c4)
        self.corr_feature_att_8 = M.channelAtt(c4 // 4, c8)
        self.concat_feature_att_4 = M.channelAtt(c4 // 4, c4)
        self.hourglass_att = M.hourglass(32)
        self.classif_att_ = M.Classifier(32)
        self.hourglass = M.hourglass2(32)
        self.classif = M.Classifier(32)
        self.concat_stem = M.BasicConv(c4 // 2, c4 // 4, is_3d=True, kernel_size=3, stride=1, padding=1)
        self.propagation = M.Propagation()                  # parameter-free (models/SemStereo.py:237-238)
        self.propagation_prob = M.Propagation_prob()

    def load_reference_state_dict(self, state_dict, strict=True):
        """Load the slice of a reference SemStereo state_dict (optionally `module.`-prefixed, as
        nn.DataParallel checkpoints are) that belongs to the hot segment."""
        sd = {}
        for k, v in state_dict.items():
            k = k[7:] if k.startswith("module.") else k
            if k.split(".")[0] in OWNED_PREFIXES:
                sd[k] = v
        return self.load_state_dict(sd, strict=strict)

    # ---- models/SemStereo.py:273-310 ---------------------------------------------------
    def attention_branch(self, fl4, fr4, fl8, fr8):
        """-> (att_topk [B,1,24,H4,W4], samples [B,24,H4,W4], pred_att [B,H4,W4], pred_att0 [B,H4,W4]).  FUSED: this repo's
        own composition of the fused kernels.  Otherwise the reference's statements one by one, in its order, on the
        reference-named ops and the twins -- what an untouched forward() executes; in inference those hand out deferred
        handles (deferred.py) and the same fused kernels run, SS_DEFER=0 gives the plain op-by-op execution."""
        fast = getattr(self, "FUSED", HotSegment.FUSED) and M._inference(self, fl4, fr4, fl8, fr8)
        if fast:
            with dfr.suspended():
                return HotSegment._attention_branch(self, fl4, fr4, fl8, fr8, True)
        return HotSegment._attention_branch(self, fl4, fr4, fl8, fr8, False)

    def _attention_branch(self, fl4, fr4, fl8, fr8, fast):
        m8, m4 = self.maxdisp // 8, self.maxdisp // 4
        H4, W4 = fl4.shape[-2:]
        groups = fl8.shape[1] // 8
        # the disparity ranges at 1/8 and 1/4 scale, and the op set whose reference-named callables the line-by-line form uses
        unsigned = getattr(self, "unsigned", False)
        lib = ops_unsigned if unsigned else ops
        r8 = ops.unsigned_range(m8) if unsigned else ops.signed_range(m8)
        r4 = ops.unsigned_range(m4) if unsigned else ops.signed_range(m4)
        if fast and HotSegment.GWC_PATCH_FUSED and isinstance(self.patch, M.DepthwisePatch) and ops.gwc_patch_gate_applies(fl8, m8, groups, r8, fr8):
            M.PATH_COUNTS["hip"] += 1
            cost_att = ops.gwc_patch_gate(fl8, fr8, m8, groups, self.patch.weight, self.corr_feature_att_8.logits(fl8), _range=r8)   # :273-276 fused
        elif fast:
            corr = lib.build_gwc_volume_norm(fl8, fr8, m8, groups)                             # :273
            cost_att = self.patch(corr, self.corr_feature_att_8.logits(fl8))                   # :274 + :276 fused
        else:
            corr = lib.build_gwc_volume_norm(fl8, fr8, m8, groups)                             # :273
            cost_att = self.corr_feature_att_8(self.patch(corr), fl8)
        rel = self.__dict__.get("_release") if fast else None
        cost_att = self.classif_att_(self.hourglass_att(cost_att))                             # :277-278
        if rel is not None:
            rel("cls")
        if (not fast and getattr(self, "FUSED", HotSegment.FUSED) and not M._inference(self, fl4, fr4, fl8, fr8)
                and isinstance(cost_att, torch.Tensor) and T.attention_tail_applies(cost_att, r4, H4, W4, fl4, fr4, TOPK)):
            # training / autograd (main_us3d.py:186-222): :279-310 as the three fused launches with their backward kernels (train.py)
            return T.attention_tail(cost_att, fl4, fr4, self.gamma, self.beta, r4, H4, W4, TOPK)
        if fast and ops.upsample_softmax_regression_applies(cost_att, m4, H4, W4, r4):
            att_weights, pred0, var = ops.upsample_softmax_regression(cost_att, m4, H4, W4, _range=r4)    # :279-285 fused
        elif fast:
            att_weights = F.interpolate(cost_att, [r4[1], H4, W4], mode="trilinear")           # :279
            pred0, var, _ = ops.softmax_regression(att_weights.squeeze(1), m4, _range=r4)      # :281-285 fused
        else:
            att_weights = F.interpolate(cost_att, [r4[1], H4, W4], mode="trilinear")           # :279
            prob0 = F.softmax(att_weights.squeeze(1), dim=1)
            pred0 = lib.disparity_regression(prob0, m4)
            var = lib.disparity_variance(prob0, m4, pred0.unsqueeze(1))
        if fast and r4[1] <= ops.TOPK_CANDIDATES_MAX_D:         # beyond (maxdisp >= 320): the line-by-line form below
            if rel is not None:
                rel("up")
            strength = ops.sample_strength(fl4, fr4, pred0, var, self.gamma, self.beta)        # :286-293 fused
            if rel is not None:
                rel("st")
            att_topk, samples, pred_att = ops.topk_candidates(att_weights, strength, m4, TOPK, _range=r4)  # :295-310 fused
            return att_topk, samples, pred_att, pred0
        var = self.beta + self.gamma * var                                                     # :286
        var = torch.sigmoid(var)                                                               # :287
        var_samples = self.propagation(var)                                                    # :288
        disp_samples = self.propagation(pred0.unsqueeze(1))                                    # :289
        right_w, left_b = ops.SpatialTransformer_grid(fl4, fr4, disp_samples)                  # :291
        strength = (left_b * right_w).mean(dim=1)                                              # :292
        strength = torch.softmax(strength * var_samples, dim=1)                                # :293
        aw = self.propagation_prob(att_weights)                                                # :295
        aw = aw * strength.unsqueeze(2)                                                        # :296
        aw = torch.sum(aw, dim=1, keepdim=True)                                                # :297
        aw_prob = F.softmax(aw, dim=2)                                                         # :298
        _, ind = aw_prob.sort(2, True)                                                         # :299
        ind_k = ind[:, :, :TOPK]                                                               # :302
        ind_k = ind_k.sort(2, False)[0]                                                        # :303
        att_topk = torch.gather(aw_prob, 2, ind_k)                                             # :304
        samples = ind_k.squeeze(1).float() - (-r4[0]) if r4[0] else ind_k.squeeze(1).float()   # :305 (SemStereo_WHU.py:305: no offset)
        att_prob = torch.gather(aw, 2, ind_k).squeeze(1)                                       # :307
        att_prob = F.softmax(att_prob, dim=1)                                                  # :308
        pred_att = torch.sum(att_prob * samples, dim=1)                                        # :309-310
        return att_topk, samples, pred_att, pred0

    # ---- models/SemStereo.py:314-323 ---------------------------------------------------
    def prelude_jobs(self, fl4, fr4):
        """:314-315 and the image half of :320 -- the 2-D convolutions that do not depend on the attention branch (fast path only)
        as three independent jobs: concat features of the left view, of the right view, and the concat_feature_att_4 gate
        (sigmoid applied).  -> dict name -> callable."""
        cf = self.concat_feature

        def one_view(x):
            # conv3x3 + BN + ReLU, conv3x3 (models/SemStereo.py:222-226), reference-built or twin: both on the split engine
            if (len(cf) == 2 and isinstance(cf[1], nn.Conv2d) and isinstance(getattr(cf[0], "conv", None), nn.Conv2d)
                    and getattr(cf[0], "relu", False) and M._inference(cf, x)):
                bn = cf[0].bn if getattr(cf[0], "use_bn", True) else None
                y = M.run_conv2d(cf[0], "bc2d", cf[0].conv, bn, x, True)
                if y is not None:
                    z = M.run_conv2d(cf, "cf1", cf[1], None, y, False)
                    return z if z is not None else cf[1](y)
            return cf(x)
        def both_views():
            # r05: the two views in ONE pair of launches -- the first layer reads both inputs through two pointers
            # (ss_conv2d_bf16s_pair_fwd: no torch.cat in front, which was 28 us of ATen copy), the second is a plain batch-2B call
            if (len(cf) == 2 and isinstance(cf[1], nn.Conv2d) and isinstance(getattr(cf[0], "conv", None), nn.Conv2d)
                    and getattr(cf[0], "relu", False) and M._inference(cf, fl4, fr4) and HotSegment.PAIR_VIEWS):
                bn = cf[0].bn if getattr(cf[0], "use_bn", True) else None
                y = M.run_conv2d_pair(cf[0], "bc2d", cf[0].conv, bn, fl4, fr4, True)
                if y is not None:
                    z = M.run_conv2d(cf, "cf1", cf[1], None, y, False)
                    if z is not None:
                        n = fl4.shape[0]
                        return z[:n], z[n:]
            return None
        return {"cl": lambda: one_view(fl4), "cr": lambda: one_view(fr4), "clr": both_views,
                "gate": lambda: self.concat_feature_att_4.logits(fl4, sigmoid=True)}

    def matching_prelude(self, fl4, fr4):
        """The three jobs of prelude_jobs, in order, on the current stream -> (cl, cr, gate4)."""
        jobs = HotSegment.prelude_jobs(self, fl4, fr4)
        both = jobs["clr"]()
        cl, cr = both if both is not None else (jobs["cl"](), jobs["cr"]())
        return cl, cr, jobs["gate"]()

    def matching_branch(self, fl4, fr4, att_topk, samples, prelude=None):
        fast = getattr(self, "FUSED", HotSegment.FUSED) and M._inference(self, fl4, fr4, dfr.real(att_topk))
        if fast:
            with dfr.suspended():
                return HotSegment._matching_branch(self, fl4, fr4, dfr.real(att_topk), dfr.real(samples), prelude, True)
        return dfr.real(HotSegment._matching_branch(self, fl4, fr4, att_topk, samples, prelude, False))

    def _matching_branch(self, fl4, fr4, att_topk, samples, prelude, fast):
        if fast:
            cl, cr, gate4 = prelude if prelude is not None else HotSegment.matching_prelude(self, fl4, fr4)
            cl, cr = _ready(cl), _ready(cr)              # (gate4: joined only where the stem conv needs it)
            if M.CONV_ENGINE != "f32" and samples.shape[1] in (6, 24, 32) and HotSegment.STEM_BY_HALVES:
                # the left half of the volume is the 2-D map `cl` broadcast over the candidates: neither built
                # nor convolved (modules.stem_of_broadcast_and_volume); only the warped right half is a volume
                partial = M.stem_broadcast_half(self.concat_stem, cl, att_topk)                # :319, broadcast half
                if M.stem_gather_applies(self.concat_stem, cr, samples) and ops.integer_candidates(samples):
                    # `samples` are the integer candidates of ops.topk_candidates (:299-305): the warped half is gathered inside
                    # the conv's staging -- no warp launch, no volume (ss_conv3d_gather_fwd)
                    volume = M.stem_gather_half(self.concat_stem, cr, samples, att_topk, partial, _ready(gate4), consume_partial=True)   # :316-320
                elif M.stem_presplit_applies(self.concat_stem, cr):
                    xs, xexp = ops.concat_volume_sampled_presplit(cr, samples, att_topk)       # :316 + :318, warped half, pre-split
                    volume = M.stem_volume_half_presplit(self.concat_stem, xs, xexp, partial, _ready(gate4))   # :319 + :320
                else:
                    right = ops.concat_volume_sampled(None, cr, samples, att_topk)             # :316 + :318, warped half
                    volume = M.stem_volume_half(self.concat_stem, right, partial, _ready(gate4))       # :319 + :320 (gate4: sigmoid done)
            else:
                volume = ops.concat_volume_sampled(cl, cr, samples, att_topk)                  # :316 + :318 fused
                volume = self.concat_stem(volume, gate=_ready(gate4))                          # :319 + :320 fused
        else:
            cl = self.concat_feature(fl4)                                                      # :314
            cr = self.concat_feature(fr4)                                                      # :315
            if (getattr(self, "FUSED", HotSegment.FUSED) and torch.is_grad_enabled() and isinstance(att_topk, torch.Tensor)
                    and isinstance(samples, torch.Tensor) and isinstance(cl, torch.Tensor) and T.concat_volume_applies(cl, cr, samples, att_topk)):
                # training (r06): :316-318 as one launch each way instead of a warp, a cat and a multiply and their three backwards
                volume = T.concat_volume_sampled(cl, cr, samples, att_topk, margin=max(self.maxdisp // 4, 1))
            else:
                right_w, left_b = ops.SpatialTransformer_grid(cl, cr, samples)                 # :241-242 (concat_volume_generator)
                volume = torch.cat((left_b, right_w), dim=1)                                   # :243
                volume = att_topk * volume                                                     # :318
            volume = self.concat_stem(volume)                                                  # :319
            volume = self.concat_feature_att_4(volume, fl4)                                    # :320
        cost = self.hourglass(volume)                                                          # :321
        if fast and samples.shape[1] == 24 and isinstance(self.classif, M.Classifier):
            pc = self.classif.patches(cost)              # :322-323 in two launches instead of three: the one-pass classifier's patches
            if pc is not None:                           # are summed by the soft-argmax that reads them (r06)
                return ops.regression_topk_patched(pc, samples, 2)
        cost = self.classif(cost)                                                              # :322
        return ops.regression_topk(cost.squeeze(1), samples, 2)                                # :323

    def forward(self, fl4, fr4, fl8, fr8):
        """features_left[1], features_right[1] [B,128,H/4,W/4]; features_left[2], features_right[2]
        [B,256,H/8,W/8]  ->  dict(pred [B,1,H/4,W/4], pred_att [B,H/4,W/4], samples, att_topk, pred_att0)."""
        return run_segment(self, fl4, fr4, fl8, fr8)


def run_segment(owner, fl4, fr4, fl8, fr8, matching=True):
    """models/SemStereo.py:273-323 on `owner`: a HotSegment, or a reference SemStereo instance after
    `install.accelerate` -- the attribute names are the same, so the methods above run unbound on it.
    `matching=False` stops after the attention branch (the reference's att_weights_only mode)."""
    fused = getattr(owner, "FUSED", HotSegment.FUSED)
    overlap = getattr(owner, "OVERLAP", HotSegment.OVERLAP)
    if getattr(_TLS, "overlap", None) is not None:       # a caller's overlap_override(...) for this thread
        overlap = _TLS.overlap
    if overlap == "auto":
        overlap = fl4.shape[0] <= 2
    prelude = None
    released = {}
    if matching and fused and overlap and fl4.is_cuda and M._inference(owner, fl4, fr4, fl8, fr8):
        # The attention branch works at 1/8 scale: at small batch most of its kernels cannot fill
        # 256 CUs.  The matching branch's 2-D convolutions are independent of it, so they run on a
        # second HIP stream underneath; each result is joined where the matching branch first needs it.
        cur, side = torch.cuda.current_stream(fl4.device), _side_stream(fl4.device)
        jobs = HotSegment.prelude_jobs(owner, fl4, fr4)
        where = str(getattr(owner, "PRELUDE_AT", HotSegment.PRELUDE_AT)).split(",")
        where = dict(zip(("cl", "cr", "gate"), where + [where[-1]] * (3 - len(where))))

        def release(point):
            """Start, on the second stream, the jobs released at `point` (behind everything the main stream has issued so far)."""
            todo = [k for k in ("cl", "cr", "gate") if k not in released and (point is None or where[k] == point)]
            if not todo:
                return
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                if "cl" in todo and "cr" in todo:              # both views released together: one pair of launches for the two
                    both = jobs["clr"]()
                    if both is not None:
                        ev = torch.cuda.Event()
                        ev.record(side)
                        released["cl"], released["cr"] = _Pending(both[0], ev, cur), _Pending(both[1], ev, cur)
                        todo = [k for k in todo if k not in ("cl", "cr")]
                for k in todo:
                    t = jobs[k]()
                    ev = torch.cuda.Event()
                    ev.record(side)
                    released[k] = _Pending(t, ev, cur)
        owner.__dict__["_release"] = release
        if isinstance(owner.hourglass_att, M.hourglass):
            owner.hourglass_att.__dict__["_mid_hook"] = ("*", release)
        release("start")
    try:
        att_topk, samples, pred_att, pred0 = HotSegment.attention_branch(owner, fl4, fr4, fl8, fr8)
    finally:
        if "_release" in owner.__dict__:
            owner.__dict__.pop("_release", None)
            owner.hourglass_att.__dict__.pop("_mid_hook", None)
    if released or (matching and fused and overlap and fl4.is_cuda and M._inference(owner, fl4, fr4, fl8, fr8)):
        release(None)                                  # whatever has not been released yet (unknown point names included)
        prelude = (released["cl"], released["cr"], released["gate"])
    pred = HotSegment.matching_branch(owner, fl4, fr4, att_topk, samples, prelude) if matching else None
    return dfr.real(dict(pred=pred, pred_att=pred_att, samples=samples, att_topk=att_topk, pred_att0=pred0))


class GraphedSegment:
    """The hot segment captured ONCE into a HIP graph and replayed: one graph launch per call instead of ~40 kernel launches
    from Python (both streams of the segment join the capture through their event waits).  Shapes are fixed at capture;
    inputs are copied into the captured buffers, outputs are the captured buffers (overwritten by the next call --
    clone what must outlive it).  Inference only.  On the bench shape the step is GPU-bound (a no-kernel-running gap of 42 us
    per step shrinks to 18 us: +0.8 %); the form matters for small images and for callers with a slow host thread."""

    def __init__(self, segment, fl4, fr4, fl8, fr8, warmup=3):
        assert fl4.is_cuda and not segment.training
        self.segment = segment
        self.inputs = [t.detach().clone().contiguous() for t in (fl4, fr4, fl8, fr8)]
        with torch.no_grad():
            for _ in range(warmup):                     # packs weights, sizes the persistent grids, allocates the side stream
                segment(*self.inputs)
            torch.cuda.synchronize(fl4.device)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.outputs = segment(*self.inputs)

    def __call__(self, fl4, fr4, fl8, fr8):
        for dst, src in zip(self.inputs, (fl4, fr4, fl8, fr8)):
            assert dst.shape == src.shape, "GraphedSegment replays the shapes it was captured with"
            dst.copy_(src)
        self.graph.replay()
        return self.outputs


class PairPipeline:
    """Consecutive calls of a segment issued round-robin on `lanes` HIP streams, each with its own second stream: the kernels of
    pair i + 1 fill the compute units that the kernels of pair i leave idle -- most of the attention branch at small batch (1/8-scale
    layers of 128-512 workgroups on a 256-CU chip) and the tail of EVERY launch (the last workgroups of a persistent grid).  No
    batching, no change to what a call computes: every pair runs the same kernels on its own buffers, so its outputs are bit-identical
    to a plain call (tests/test_parity_gpu.py::test_pair_pipeline_is_bit_identical_to_sequential_calls).  Measured r04 (bench.py
    --streams): batch 1: 497 -> 528 / 531 / 542 / 538 pairs/s on 2 / 3 / 4 / 6 lanes; batch 4: 535 -> 561 (3 lanes); batch 8: 543 -> 560;
    2048^2 / 192: 120 -> 126.  r05 (gathered stem): 2 / 3 / 4 / 5 / 6 / 8 lanes 567 / 551 / 570 / 578 / 580 / 575; interleaved 4 vs 6:
    567.2 / 562.5 / 563.5 vs 572.9 / 571.8 / 574.1 (batch 8: 589.7 / 588.9 vs 592.7 / 590.4) -> default 6.  Inference only.

    `pipe(*inputs)` returns the module's output at once; the tensors are valid after `pipe.synchronize()`, or on another stream
    after `pipe.join(outputs, stream)` (waits for the call's event AND records the consumer stream on the outputs: they live in the
    lane's allocator pool).  Inputs produced on the calling stream are waited for.  `segment` is any inference
    module -- a HotSegment (`pipe(fl4, fr4, fl8, fr8)`) or a whole reference model after `install` + `accelerate`
    (`pipe(imgL, imgR)`: its backbone's kernels ride the lanes too)."""

    #: calls issued on EVERY lane when the pipeline is first used, before anything is returned to the caller: a lane's stream has its
    #: own pool in PyTorch's caching allocator, and the first calls on a cold lane pay a hipMalloc per temporary (~40 per pair)
    #: -- measured r05 with bench.py --steps 20 --warmup 5 on 6 lanes (the sixth lane saw its first call inside the timed steps):
    #: 577-582 pairs/s, 589 with 6 warm-up steps, 596 with 12, 602-609 with 30; with 2 / 4 throw-away calls per lane here the 5-step
    #: warm-up reads 595-597 / 598-600 (`profiles/r05_ab_warmup_k20.txt`).  One-time set-up like the weight packing of the priming
    #: call; what it costs is 4 x lanes forward passes at construction.
    LANE_WARM_CALLS = 4

    def __init__(self, segment, lanes=6):
        assert lanes >= 1 and not segment.training
        self.segment, self.nlanes = segment, int(lanes)
        self.lanes, self.turn, self.last_event, self._primed = None, 0, None, False
        self.rebuilds = 0                       # calls during which a per-module cache entry was (re)built (see __call__)
        self.warmed_lanes = 0                   # lanes whose allocator pools were filled at first use (all of them unless memory is short)

    def _prime(self, inputs, dev):
        # weight packing and every other per-module cache are filled by a call ON THE CALLING STREAM, and drained, before several
        # streams read them
        torch.cuda.reset_peak_memory_stats(dev)
        self._alloc_before = torch.cuda.memory_allocated(dev)
        with torch.no_grad():
            self.segment(*inputs)
        torch.cuda.synchronize(dev)
        if self.lanes is None:
            self.lanes = [torch.cuda.Stream(device=dev) for _ in range(self.nlanes)]
            if self.nlanes > 1:
                M.E.retain_replaced(True)
        M.E.drop_retired()
        if self.nlanes > 1:             # every lane's allocator pool filled by throw-away calls (see LANE_WARM_CALLS)
            cur = torch.cuda.current_stream(dev)
            #  each lane's pool ends up holding one call's working set: `lanes` times the memory of a plain call.  The
            # priming call above measured that working set; lanes whose pools would not fit in what is free now are not warmed (their
            # first calls then allocate -- or fail -- inside the caller's own steps, where the caller sees it)
            try:
                free_b, _total = torch.cuda.mem_get_info(dev)
                per_lane = max(torch.cuda.max_memory_allocated(dev) - self._alloc_before, 1)
                self.warmed_lanes = int(min(self.nlanes, free_b * 0.8 // per_lane))
            except Exception:       # noqa: BLE001
                self.warmed_lanes = self.nlanes
            for _ in range(self.LANE_WARM_CALLS if self.warmed_lanes == self.nlanes else 0):
                for lane in self.lanes:
                    lane.wait_stream(cur)
                    with torch.cuda.stream(lane), torch.no_grad(), overlap_override(False):
                        self.segment(*inputs)
                    for t in inputs:
                        if isinstance(t, torch.Tensor):
                            t.record_stream(lane)
            torch.cuda.synchronize(dev)
        self._primed = True

    def close(self):
        """Drain the lanes and stop retaining replaced cache entries (also called when the pipeline is collected)."""
        if self.lanes is not None:
            self.synchronize()
            if self.nlanes > 1:
                M.E.retain_replaced(False)
            M.E.drop_retired()
            self.lanes = None
            self._primed = False

    def __del__(self):
        try:
            self.close()
        except Exception:       # noqa: BLE001  (interpreter shutdown)
            pass

    def __call__(self, *inputs):
        dev = inputs[0].device
        if not self._primed:
            self._prime(inputs, dev)
        lane = self.lanes[self.turn % self.nlanes]
        self.turn += 1
        lane.wait_stream(torch.cuda.current_stream(dev))
        if M.E._RETIRED and all(st.query() for st in self.lanes):
            M.E.drop_retired()              # entries replaced by calls OUTSIDE the pipeline (a plain call between two pipelined ones): nothing in flight reads them
        gen = M.E.cache_generation()
        # the lanes ARE the concurrency: the within-pair second stream on top of them costs 0.8 % (4 lanes: 525.7 vs 521.4 pairs/s);
        # passed down per thread (overlap_override), the shared module is not written to
        with torch.cuda.stream(lane), torch.no_grad(), overlap_override(False if self.nlanes > 1 else None):
            out = self.segment(*inputs)
        if M.E.cache_generation() != gen and self.nlanes > 1:
            # A cache entry was (re)built during this call -- load_state_dict / an in-place weight update, an engine switch, a shape
            # whose key had not been built: its packing kernels ran on THIS lane only, and the entries they replace may still be read
            # by pairs in flight on the other lanes (they are parked, not freed: engine.retain_replaced).  Rare and not worth
            # anything finer: drain every lane, then let the replaced tensors go.  (ADVICE r4, medium.)
            self.rebuilds += 1
            for st in self.lanes:
                st.synchronize()
            M.E.drop_retired()
        with torch.cuda.stream(lane):
            for t in inputs:
                if isinstance(t, torch.Tensor):
                    t.record_stream(lane)
            self.last_event = torch.cuda.Event()
            self.last_event.record(lane)
        return out

    def join(self, outputs=None, stream=None):
        """Make `stream` (default: the current stream) wait for the last issued call and tell the allocator that `outputs` (a
        tensor or any nesting of dicts / lists of tensors: what that call returned) are used there -- the outputs live in the
        lane's allocator pool, and without record_stream the lane could reuse their memory while the consumer still reads it."""
        stream = stream or torch.cuda.current_stream()
        if self.last_event is not None:
            stream.wait_event(self.last_event)

        def mark(x):
            if isinstance(x, torch.Tensor):
                if x.is_cuda:
                    x.record_stream(stream)
            elif isinstance(x, dict):
                for v in x.values():
                    mark(v)
            elif isinstance(x, (list, tuple)):
                for v in x:
                    mark(v)
        mark(outputs)
        return outputs

    def synchronize(self):
        for st in self.lanes or ():
            st.synchronize()

