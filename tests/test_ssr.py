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
        y = mod(d.cuda(), wt.cuda(), lab.cuda())
    assert float((y.cpu() - ref).abs().max()) <= 2e-5
