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
    assert sorted(done) == sorted(["hourglass_att", "hourglass", "classif_att_", "classif", "concat_stem", "patch",
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
    assert net.forward.__func__ is sys.modules["semstereo_amd.install"].fused_inference_forward
    for name in ("feature", "feature_up", "head_l", "chal_0", "chal_4", "spx32_16", "spx16_8", "spx8_4", "spx4_2", "spx2",
                 "ssr_upsample", "stereo_if", "seg_if", "att_weights_only", "maxdisp", "gamma", "beta", "concat_feature"):
        assert hasattr(net, name), name
    (again,), _ = net(imgL, imgR)
    assert torch.allclose(again.detach(), want, atol=1e-4, rtol=1e-4)
    assert list(net.state_dict().keys()) == keys
    sa.restore_forward(net)
    assert "forward" not in net.__dict__ and net.forward.__func__ is ref_module.SemStereo.forward


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
