"""Build-container only (skipped where /root/reference is absent, e.g. on the GPU box): the
drop-in helpers against the REFERENCE's own classes -- `accelerate()` adopts a real
`models.SemStereo.SemStereo` instance, keeps its state_dict, shares its parameters, and the
reference's unchanged forward() still runs through the adopted twins (PyTorch path on CPU) to the
same output; `install()` rebinds the op-library names in the reference module's globals."""
import os
import sys
import types

import pytest
import torch
import torch.nn as nn

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "models")), reason="reference not mounted")


@pytest.fixture(scope="module")
def ref_module():
    class _Backbone(nn.Module):            # attribute surface Feature() reads (models/SemStereo.py:37-45)
        def __init__(self):
            super().__init__()
            mk = lambda i, o, s: nn.Sequential(nn.Conv2d(i, o, 3, s, 1, bias=False), nn.BatchNorm2d(o), nn.SiLU())
            self.stem = mk(3, 32, 2)
            self.stages_0 = nn.Sequential(mk(32, 64, 1)); self.stages_1 = nn.Sequential(mk(64, 128, 2))
            self.stages_2 = nn.Sequential(mk(128, 256, 2)); self.stages_3 = nn.Sequential(mk(256, 384, 2))
            self.stages_4 = nn.Sequential(mk(384, 512, 2))
    saved = {k: sys.modules.get(k) for k in ("timm", "models")}
    timm = types.ModuleType("timm")
    timm.create_model = lambda *a, **k: _Backbone()
    sys.modules["timm"] = timm
    pkg = types.ModuleType("models")
    pkg.__path__ = [os.path.join(REF, "models")]
    sys.modules["models"] = pkg
    sys.path.insert(0, REF)
    import importlib
    ms = importlib.import_module("models.SemStereo")
    yield ms
    sys.path.remove(REF)
    for k in [k for k in sys.modules if k.startswith("models.")]:
        sys.modules.pop(k)
    for k, v in saved.items():
        if v is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = v


def _cpu_replicate(network, detach=True):
    """What torch.nn.parallel.replicate(network, devices, detach=True) builds per device (torch/nn/parallel/replicate.py),
    on CPU: every module through _replicate_for_data_parallel(), children re-linked, parameters as plain detached
    tensors set as NON-parameter attributes (so replica.parameters() is empty), buffers copied."""
    from collections import OrderedDict
    modules = list(network.modules())
    index = {m: i for i, m in enumerate(modules)}
    copies = []
    for m in modules:
        r = m._replicate_for_data_parallel()
        r._former_parameters = OrderedDict()
        copies.append(r)
    for i, m in enumerate(modules):
        for key, child in m._modules.items():
            if child is None:
                copies[i]._modules[key] = None
            else:
                setattr(copies[i], key, copies[index[child]])
        for key, p in m._parameters.items():
            if p is None:
                copies[i]._parameters[key] = None
            else:
                c = p.detach().clone().requires_grad_(p.requires_grad and not detach)
                setattr(copies[i], key, c)
                copies[i]._former_parameters[key] = c
        for key, b in m._buffers.items():
            copies[i]._buffers[key] = None if b is None else b.detach().clone()
    return copies[0]


def test_accelerate_on_the_real_reference_model(ref_module):
    import semstereo_amd as sa
    torch.manual_seed(0)
    net = ref_module.SemStereo(64, False, True, True, 6).eval()
    keys = list(net.state_dict().keys())
    imgL, imgR = torch.randn(1, 3, 128, 128), torch.randn(1, 3, 128, 128)
    with torch.no_grad():
        (want,), _ = net(imgL, imgR)
    w = net.hourglass.conv1[0][0].weight
    done = sa.accelerate(net)
    assert sorted(done) == sorted(["hourglass_att", "hourglass", "classif_att_", "classif", "concat_stem", "patch", "concat_feature",
                                   "corr_feature_att_8", "concat_feature_att_4", "ssr_upsample", "propagation",
                                   "propagation_prob"])
    assert list(net.state_dict().keys()) == keys                      # checkpoint compatibility
    assert net.hourglass.conv1[0][0].weight is w                      # shared, not copied
    assert isinstance(net.hourglass, sa.modules.hourglass2) and net.hourglass.attention_block.block == (6, 4, 4)
    # the reference's forward(), unchanged, now calling the twins; with autograd on they take their
    # stock-PyTorch path, so this runs on CPU and must reproduce the original output
    before = sa.modules.PATH_COUNTS["torch"]
    (got,), _ = net(imgL, imgR)
    assert sa.modules.PATH_COUNTS["torch"] > before
    assert torch.allclose(got.detach(), want, atol=1e-4, rtol=1e-4)
    # fuse_forward: inference calls on GPU tensors would take the fused path; on this GPU-less box the call is
    # handed to the reference's own forward() (CPU tensors, autograd on), the attribute names it needs exist,
    # and the routing can be undone
    sa.accelerate(net, fuse_forward=True)
    assert type(net)._ss_fused_forward and isinstance(net, ref_module.SemStereo) and "forward" not in net.__dict__
    for name in ("feature", "feature_up", "head_l", "chal_0", "chal_4", "spx32_16", "spx16_8", "spx8_4", "spx4_2", "spx2",
                 "ssr_upsample", "stereo_if", "seg_if", "att_weights_only", "maxdisp", "gamma", "beta", "concat_feature"):
        assert hasattr(net, name), name
    (again,), _ = net(imgL, imgR)
    assert torch.allclose(again.detach(), want, atol=1e-4, rtol=1e-4)
    assert list(net.state_dict().keys()) == keys
    # nn.DataParallel's replicas (main_us3d.py:100, test_us3d.py:58) must run THEIR OWN forward on THEIR OWN weights:
    # replicas are built by copying the instance __dict__, so the routing has to live on the class
    replica = _cpu_replicate(net)
    assert replica.forward.__self__ is replica and type(replica) is type(net)
    assert replica.hourglass is not net.hourglass and replica.hourglass.forward.__self__ is replica.hourglass
    assert replica.hourglass.conv1[0][0].weight is not net.hourglass.conv1[0][0].weight and not list(replica.parameters())
    from semstereo_amd.modules import _cache, _inference
    assert _cache(net.hourglass) is _cache(net.hourglass)                 # the original keeps its packed weights ...
    assert _cache(replica.hourglass) is not _cache(net.hourglass)         # ... a replica never sees or reuses them
    assert _cache(replica.hourglass) is not _cache(replica.hourglass)
    with torch.no_grad():
        assert _inference(replica.hourglass, imgL)
    replica.hourglass.conv1[0][0].weight.requires_grad_(True)            # non-detached replicas (autograd on) are not inference
    assert not _inference(replica.hourglass, imgL)
    replica.hourglass.conv1[0][0].weight.requires_grad_(False)
    # with autograd on, DataParallel's replicas carry differentiable copies: the twins take their PyTorch path (CPU-runnable)
    (rep_out,), _ = _cpu_replicate(net, detach=False)(imgL, imgR)
    assert torch.allclose(rep_out.detach(), want, atol=1e-4, rtol=1e-4)
    sa.restore_forward(net)
    assert type(net) is ref_module.SemStereo and "forward" not in net.__dict__


def test_install_into_the_real_reference_module(ref_module):
    import semstereo_amd as sa
    orig = ref_module.build_gwc_volume_norm
    prev = sa.install(ref_module)
    try:
        for name in sa.ops.REFERENCE_NAMES:
            assert getattr(ref_module, name) is getattr(sa.ops, name)
        # the reference's call sites now reach the HIP ops: on a CPU tensor that is a loud error
        net = ref_module.SemStereo(64, False, True, True, 6).eval()
        with pytest.raises(sa._lib.SemStereoHipError), torch.no_grad():
            net(torch.randn(1, 3, 64, 64), torch.randn(1, 3, 64, 64))
    finally:
        sa.uninstall(ref_module, prev)
    assert ref_module.build_gwc_volume_norm is orig


def test_install_unsigned_op_set_into_the_reference_whu_module(ref_module):
    """models/SemStereo_WHU.py star-imports the SIGNED op library and cannot run as shipped; `install(module, unsigned=True)`
    binds the unsigned op set it is written for (models/submodule_.py's definitions) in ITS globals."""
    import importlib
    import semstereo_amd as sa
    mw = importlib.import_module("models.SemStereo_WHU")
    net = mw.SemStereo_WHU(128, False, True, True, 6).eval()
    with pytest.raises(RuntimeError), torch.no_grad():                     # as shipped: size mismatch in disparity_regression
        net(torch.randn(1, 3, 128, 128), torch.randn(1, 3, 128, 128))
    prev = sa.install(mw, unsigned=True)
    try:
        for name in sa.ops.REFERENCE_NAMES:
            assert getattr(mw, name) is getattr(sa.ops_unsigned, name)
        assert mw.disparity_regression is not sa.ops.disparity_regression
        with pytest.raises(sa._lib.SemStereoHipError), torch.no_grad():    # the HIP ops are reached (CPU tensors: a loud error)
            net(torch.randn(1, 3, 128, 128), torch.randn(1, 3, 128, 128))
    finally:
        sa.uninstall(mw, prev)


def test_deferred_fusion_matches_the_real_reference_forward(ref_module, monkeypatch):
    """The statement sequences semstereo_amd.deferred recognises are the REFERENCE's: its own, untouched forward()
    (models/SemStereo.py:246-346) on an installed + accelerated model fires every fused rule exactly once -- volume + patch +
    gate (:273-276), up-sampling + soft-max + regression + variance (:279-285), the 5-candidate probe (:286-293), the top-24
    selection (:295-310), sparse concat volume + stem + gate (:316-320) -- and returns what the plain execution returns.
    (CPU: the fused kernels are stood in for by torch compositions of the same statements, tests/test_deferred.py.)"""
    import semstereo_amd as sa
    from semstereo_amd import deferred as dfr
    from test_deferred import ALL_RULES, _cpu_fused_kernels
    torch.manual_seed(0)
    net = ref_module.SemStereo(64, False, True, True, 6).eval()
    with torch.no_grad():
        for name, t in net.named_parameters():                 # gamma = 0 (the reference's init) would leave the variance path dead
            if name == "gamma":
                t.fill_(0.25)
    imgL, imgR = torch.randn(1, 3, 128, 128), torch.randn(1, 3, 128, 128)
    with torch.no_grad():
        (want,), lab = net(imgL, imgR)                          # the reference alone
        net.att_weights_only = True
        (want_att,), _ = net(imgL, imgR)
        net.att_weights_only = False
    _cpu_fused_kernels(monkeypatch)
    prev = sa.install(ref_module)
    try:
        sa.accelerate(net)
        with torch.no_grad():
            (got,), lab2 = net(imgL, imgR)                      # its forward(), untouched, on the ops + twins with deferral
        fired = dict(dfr.STATS["fused"])
        assert set(fired) == ALL_RULES and all(v == 1 for v in fired.values()), fired
        assert isinstance(got, torch.Tensor) and torch.equal(lab, lab2)
        # (random PyTorch-default weights: nearly uniform attention, so a hard pick may flip on isolated pixels between two fp32
        # evaluation orders; everything else agrees to rounding)
        err = (got - want).abs()
        assert float(err.median()) <= 1e-4 and float((err <= 4e-3).float().mean()) >= 0.99, (float(err.median()), float(err.max()))
        # att_weights_only mode (:311, no matching branch)
        net.att_weights_only = True
        monkeypatch.setattr(dfr, "STATS", {"fused": {}, "replayed": 0})
        with torch.no_grad():
            (got_att,), _ = net(imgL, imgR)
        assert set(dfr.STATS["fused"]) == ALL_RULES - {"stem_by_halves"}
        err = (got_att - want_att).abs()
        assert float(err.median()) <= 1e-4 and float((err <= 4e-3).float().mean()) >= 0.99, (float(err.median()), float(err.max()))
    finally:
        sa.uninstall(ref_module, prev)
