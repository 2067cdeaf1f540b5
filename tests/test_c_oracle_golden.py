"""CPU: the plain-C restatement (oracle/oracle_ops.c) against the reference's golden fixtures.
A second, torch-free statement of the same algorithms; fp32 summation order differs from ATen's
vectorised kernels, hence a few-ulp tolerance instead of bit equality."""
import numpy as np
import pytest

from golden import cases
from oracle import c_ops


def _close(a, ref, atol, rtol=0.0):
    a = a.numpy()
    assert a.shape == ref.shape
    m = ~(np.isnan(a) & np.isnan(ref))
    if m.any():
        scale = np.abs(ref[m]).max() if rtol else 0.0
        assert np.abs(a[m] - ref[m]).max() <= atol + rtol * scale


@pytest.mark.parametrize("name", sorted(cases.GWC))
def test_gwc(golden, name):
    a, b, m, G = cases.gwc_inputs(name)
    _close(c_ops.gwc_volume(a, b, m, G, False), golden["ops"][f"gwc/{name}"], 1e-6)
    _close(c_ops.gwc_volume(a, b, m, G, True), golden["ops"][f"gwc_norm/{name}"], 1e-6)


@pytest.mark.parametrize("name", sorted(cases.CONCAT))
def test_concat(golden, name):
    a, b, m = cases.concat_inputs(name)
    _close(c_ops.concat_volume(a, b, m), golden["ops"][f"concat/{name}"], 0.0)


@pytest.mark.parametrize("name", sorted(cases.REGRESSION))
def test_regression(golden, name):
    p, m, d = cases.regression_inputs(name)
    _close(c_ops.disparity_regression(p, m), golden["ops"][f"regression/{name}"], 1e-6, 1e-6)
    _close(c_ops.disparity_regression(p, m, d).unsqueeze(1), golden["ops"][f"variance/{name}"], 1e-6, 2e-6)


@pytest.mark.parametrize("name", sorted(cases.TOPK))
def test_topk(golden, name):
    c, s, k = cases.topk_inputs(name)
    _close(c_ops.regression_topk(c, s, k), golden["ops"][f"topk/{name}"], 1e-6, 1e-6)


@pytest.mark.parametrize("name", sorted(cases.WARP))
def test_warp(golden, name):
    x, y, d = cases.warp_inputs(name)
    yw, xw = c_ops.warp_sampled(x, y, d)
    _close(yw, golden["ops"][f"warp_y/{name}"], 1e-6, 1e-6)
    _close(xw, golden["ops"][f"warp_x/{name}"], 0.0)
