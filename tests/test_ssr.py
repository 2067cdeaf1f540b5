"""SSR_upsample head (reference models/submodule.py:412-431):
oracle and module twin against the reference's fixture (CPU), HIP kernel against it (GPU)."""
import numpy as np
import pytest
import torch

from golden import cases
from oracle import ssr as ossr


@pytest.fixture(scope="module")
def ssr_golden():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ssr.npz"))


def _twin(sa):
    mod = sa.modules.SSR_upsample(6)
    P = ossr.deterministic_ssr_params()
    res = mod.load_state_dict({k[len("ssr_upsample."):]: v for k, v in P.items()}, strict=False)
    assert not res.unexpected_keys and all(k.endswith("num_batches_tracked") for k in res.missing_keys)
    return mod.eval(), P


@pytest.mark.parametrize("name", sorted(cases.SSR))
def test_oracle_matches_reference_fixture(ssr_golden, name):
    d, w, l = cases.ssr_inputs(name)
    with torch.no_grad():
        y = ossr.ssr_upsample(ossr.deterministic_ssr_params(), d, w, l)
    assert np.array_equal(y.numpy(), ssr_golden[f"ssr/{name}"])


def test_twin_keys_and_torch_path(ssr_golden):
    import semstereo_amd as sa
    mod, P = _twin(sa)
    ours = {k for k in mod.state_dict() if not k.endswith("num_batches_tracked")}
    assert ours == {k[len("ssr_upsample."):] for k in ossr.ssr_param_shapes()}
    d, w, l = cases.ssr_inputs("a")
    y = mod(d.requires_grad_(True), w, l)              # autograd on -> stock PyTorch path, runs on CPU
    assert torch.allclose(y.detach(), torch.as_tensor(ssr_golden["ssr/a"]), atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(cases.SSR))
def test_hip_kernel_matches_reference_fixture(ssr_golden, name):
    import semstereo_amd as sa
    mod, P = _twin(sa)
    mod = mod.cuda()
    d, w, l = cases.ssr_inputs(name)
    before = dict(sa.modules.PATH_COUNTS)
    with torch.no_grad():
        y = mod(d.cuda(), w.cuda(), l.cuda())
        assert isinstance(y, sa.deferred.Deferred) and sa.modules.PATH_COUNTS["hip"] == before["hip"]     # a handle: nothing has run
        y = sa.deferred.real(y)
    assert sa.modules.PATH_COUNTS["torch"] == before["torch"] and sa.modules.PATH_COUNTS["hip"] > before["hip"]
    err = float((y.cpu() - torch.as_tensor(ssr_golden[f"ssr/{name}"])).abs().max())
    assert err <= 2e-5, err          # disparities up to +-12 px: ~1e-6 relative


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 37, 70), (1, 64, 64)])
def test_hip_kernel_on_several_tiles_matches_the_oracle(shape):
    """Sizes that span several 8 x 128 tiles of the tiled kernel (seams, ragged edges) against the CPU oracle."""
    import semstereo_amd as sa
    from oracle import detdata as dd
    mod, P = _twin(sa)
    mod = mod.cuda()
    B, h, w = shape
    d = dd.t_uniform((B, 1, h, w), 950, -12.0, 12.0)
    wt = dd.t_normalish((B, 6, 4 * h, 4 * w), 951)
    lab = dd.t_normalish((B, 6, 4 * h, 4 * w), 952) * 2.0
    with torch.no_grad():
        ref = ossr.ssr_upsample(ossr.deterministic_ssr_params(), d, wt, lab)
        y = sa.deferred.real(mod(d.cuda(), wt.cuda(), lab.cuda()))
    assert float((y.cpu() - ref).abs().max()) <= 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(8, 256, 256, 32.0), (1, 512, 512, 48.0)])
def test_hip_kernel_at_full_size_matches_the_oracle(shape):
    """BASELINE.json configs[2] / configs[4]: the head at 1024 x 1024, batch 8, and at 2048 x 2048 (disparities over the
    1/4-scale range of maxdisp 128 / 192) against the CPU oracle (pinned bit-exactly to the reference's fixture above) --
    every pixel, plus the properties the head has at any size: where the guidance weights are zero the gate is a constant
    per class, and a constant disparity map comes back as that constant plus a position-independent residual away from
    the border (the 3x3 conv sees BN0(constant) everywhere)."""
    import semstereo_amd as sa
    from oracle import detdata as dd
    mod, P = _twin(sa)
    mod = mod.cuda()
    B, h, w, rng = shape
    d = dd.t_uniform((B, 1, h, w), 960, -rng, rng)
    wt = dd.t_normalish((B, 6, 4 * h, 4 * w), 961)
    lab = dd.t_normalish((B, 6, 4 * h, 4 * w), 962) * 2.0
    torch.set_num_threads(min(32, __import__("os").cpu_count() or 1))
    with torch.no_grad():
        ref = ossr.ssr_upsample(ossr.deterministic_ssr_params(), d, wt, lab)
        before = dict(sa.modules.PATH_COUNTS)
        y = sa.deferred.real(mod(d.cuda(), wt.cuda(), lab.cuda()))
        assert sa.modules.PATH_COUNTS["torch"] == before["torch"]
    assert y.shape == (B, 4 * h, 4 * w)
    err = (y.cpu() - ref).abs()
    # (the output is the up-sampled disparity, up to +-rng px, plus an O(1) residual: 1e-6 relative to the range -- 8 fp32 ulps at 48)
    assert float(err.max()) <= 1e-6 * rng and float(err.mean()) <= 3e-6, (float(err.max()), float(err.mean()))
    # constant disparity, zero guidance: out = c + r with ONE value of r over the interior
    with torch.no_grad():
        c = torch.full((1, 1, h, w), 7.25)
        y0 = sa.deferred.real(mod(c.cuda(), torch.zeros(1, 6, 4 * h, 4 * w).cuda(), lab[:1].cuda())).cpu()
    inner = y0[0, 1:-1, 1:-1] - 7.25
    assert float(inner.max() - inner.min()) <= 2e-6


def test_training_computes_the_class_gate_once_and_keeps_the_reference_semantics():
    """models/SemStereo.py:311, 324 call the head twice with the same (spx_pred, pred_label): in training the 6-class gate is
    computed once (VERDICT r4 #4b).  Same outputs and gradients as two independent evaluations, and the BatchNorm running
    statistics of the gate's two stages end where the reference's two updates put them."""
    import copy
    import semstereo_amd as sa
    a, _ = _twin(sa)
    b = copy.deepcopy(a)
    a.train(); b.train()
    d1, w, l = cases.ssr_inputs("a")
    d2 = d1 * 0.5 + 1.0
    w1, l1 = w.clone().requires_grad_(True), l.clone().requires_grad_(True)
    before = sa.modules.PATH_COUNTS.get("ssr_gate_reused", 0)
    ya = a(d1, w1, l1) + a(d2, w1, l1)                          # same tensors twice: the second call takes the parked gate
    assert sa.modules.PATH_COUNTS.get("ssr_gate_reused", 0) == before + 1 and a not in sa.modules._GATE_PARKED
    ya.sum().backward()
    #  an ODD number of calls leaves an entry behind: it lives outside the module (deepcopy still works), is dropped on
    # request (accelerate()'s forward hook does so when the model's forward returns), and a parameter update between two calls
    # invalidates it (the key carries the parameters' versions)
    a(d1, w1, l1)
    assert a in sa.modules._GATE_PARKED
    copy.deepcopy(a)
    with torch.no_grad():
        next(a.parameters()).add_(0.0)                         # what optimizer.step() does to the version counter
    reused = sa.modules.PATH_COUNTS.get("ssr_gate_reused", 0)
    a(d2, w1, l1)
    assert sa.modules.PATH_COUNTS.get("ssr_gate_reused", 0) == reused, "a gate computed with the old weights was handed back"
    sa.modules.drop_parked_gates(a)
    assert a not in sa.modules._GATE_PARKED
    for p_ in a.parameters():
        p_.grad = None
    a.load_state_dict(b.state_dict())
    a.zero_grad(); b.zero_grad()
    w1, l1 = w.clone().requires_grad_(True), l.clone().requires_grad_(True)
    ya = a(d1, w1, l1) + a(d2, w1, l1)
    ya.sum().backward()
    w2, l2 = w.clone().requires_grad_(True), l.clone().requires_grad_(True)
    yb = b(d1, w2, l2) + b(d2, w2.clone(), l2.clone())          # other tensor objects: two full evaluations (the reference's form)
    yb.sum().backward()
    assert torch.allclose(ya, yb, atol=1e-6) and torch.allclose(w1.grad, w2.grad, atol=1e-6) and torch.allclose(l1.grad, l2.grad, atol=1e-6)
    for (ka, va), (kb, vb) in zip(sorted(a.state_dict().items()), sorted(b.state_dict().items())):
        assert ka == kb and torch.allclose(va.float(), vb.float(), atol=1e-6), ka
    for p, q in zip(a.parameters(), b.parameters()):
        assert torch.allclose(p.grad, q.grad, atol=1e-5)
