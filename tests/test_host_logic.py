"""CPU: host-side logic of the drop-in boundary -- name rebinding, module adoption, state_dict
keys, error behaviour, the PyTorch training path of the module twins, batch sharding maths."""
import types

import pytest
import torch
import torch.nn as nn

import semstereo_amd as sa
from golden import cases
from oracle import hot_segment as oseg
from oracle import stack as ostack


def test_install_rebinds_and_uninstall_restores():
    fake = types.ModuleType("fake_model_module")
    sentinel = object()
    fake.build_gwc_volume_norm = sentinel
    prev = sa.install(fake)
    for name in sa.ops.REFERENCE_NAMES:
        assert getattr(fake, name) is getattr(sa.ops, name)
    sa.uninstall(fake, prev)
    assert fake.build_gwc_volume_norm is sentinel and not hasattr(fake, "regression_topk")


def test_reference_signatures():
    import inspect
    want = {
        "build_gwc_volume": ["refimg_fea", "targetimg_fea", "maxdisp", "num_groups"],
        "build_gwc_volume_norm": ["refimg_fea", "targetimg_fea", "maxdisp", "num_groups"],
        "groupwise_correlation": ["fea1", "fea2", "num_groups"],
        "build_concat_volume": ["refimg_fea", "targetimg_fea", "maxdisp"],
        "disparity_regression": ["x", "maxdisp"],
        "disparity_variance": ["x", "maxdisp", "disparity"],
        "SpatialTransformer_grid": ["x", "y", "disp_range_samples"],
        "regression_topk": ["cost", "disparity_samples", "k"],
    }
    for name, params in want.items():
        assert list(inspect.signature(getattr(sa.ops, name)).parameters) == params


def test_cpu_tensors_raise_instead_of_falling_back():
    a, b, m, G = cases.gwc_inputs("odd")
    for call in (lambda: sa.ops.build_gwc_volume(a, b, m, G), lambda: sa.ops.build_concat_volume(a, b, m),
                 lambda: sa.ops.disparity_regression(torch.rand(1, 2 * m, 3, 3), m)):
        with pytest.raises(sa._lib.SemStereoHipError):
            call()


def test_reference_assertions():
    a, b, m, G = cases.gwc_inputs("odd")
    with pytest.raises(AssertionError):
        sa.ops.build_gwc_volume(a, b, m, 5)
    with pytest.raises(AssertionError):
        sa.ops.disparity_regression(torch.rand(1, 1, 6, 3, 3), 3)
    with pytest.raises(AssertionError):
        sa.ops.disparity_variance(torch.rand(6, 3, 3), 3, torch.rand(1, 1, 3, 3))


def test_segment_state_dict_keys_are_the_references():
    seg = sa.HotSegment(64)
    ours = {k for k in seg.state_dict() if not k.endswith("num_batches_tracked")}
    table = oseg.segment_param_shapes()          # validated against the reference in make_golden.py
    assert ours == set(table)
    for k, v in seg.state_dict().items():
        if k in table:
            assert tuple(v.shape) == tuple(table[k]), k


def test_load_reference_state_dict_with_dataparallel_prefix():
    seg = sa.HotSegment(64)
    P = oseg.deterministic_params()
    sd = {"module." + k: v for k, v in P.items()}
    sd["module.feature.conv_stem.weight"] = torch.zeros(1)       # a key outside the segment is ignored
    res = seg.load_reference_state_dict(sd, strict=False)
    assert not res.unexpected_keys
    assert torch.equal(seg.hourglass_att.conv5[0].weight, P["hourglass_att.conv5.0.weight"])


def test_accelerate_shares_parameters_and_keeps_keys():
    class RefLikeHourglass(nn.Module):       # attribute tree of the reference's hourglass class
        def __init__(self):
            super().__init__()
            hg = sa.modules.hourglass(32)
            for n in ("conv1", "conv2", "conv3", "conv4", "conv5", "conv6", "redir1", "redir2"):
                setattr(self, n, getattr(hg, n))
            ab = nn.Module()
            ab.block, ab.dim_3d, ab.num_heads, ab.scale_3d = (4, 4, 4), 128, 16, 8 ** -0.5
            ab.qkv_3d, ab.final1x1 = hg.attention_block.qkv_3d, hg.attention_block.final1x1
            self.attention_block = ab

    holder = nn.Module()
    holder.hourglass_att = RefLikeHourglass()
    holder.patch = nn.Conv3d(32, 32, kernel_size=(1, 3, 3), groups=32, padding=(0, 1, 1), bias=False)
    keys = list(holder.state_dict().keys())
    w = holder.hourglass_att.conv1[0][0].weight
    done = sa.accelerate(holder)
    assert sorted(done) == ["hourglass_att", "patch"]
    assert list(holder.state_dict().keys()) == keys
    assert isinstance(holder.hourglass_att, sa.modules.hourglass) and isinstance(holder.patch, sa.modules.DepthwisePatch)
    assert holder.hourglass_att.conv1[0][0].weight is w            # shared Parameter, not a copy
    assert sa.accelerate(holder) == []                              # idempotent


@pytest.mark.parametrize("name", ["attn_pad", "attn_pad_w", "hourglass_att"])
def test_training_path_of_module_twins_matches_oracle(golden, name):
    """With autograd on, the twins run stock PyTorch layers (BatchNorm in eval here): same numbers
    as the oracle, on CPU."""
    seg = sa.HotSegment(64)
    seg.load_state_dict(oseg.deterministic_params(), strict=False)
    seg.eval()
    kind, shape, block = cases.STACK[name]
    mod = seg
    for part in kind.split("."):
        mod = getattr(mod, part)
    before = sa.modules.PATH_COUNTS["torch"]
    y = mod(cases.stack_input(name).requires_grad_(True))
    assert sa.modules.PATH_COUNTS["torch"] > before
    ref = torch.as_tensor(golden["stack"][f"stack/{name}"])
    assert torch.allclose(y.detach(), ref, atol=2e-5, rtol=1e-5)
    y.sum().backward()


def test_fold_bn_matches_batch_norm():
    bn = nn.BatchNorm3d(7).eval()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-1, 1)
        bn.running_mean.uniform_(-1, 1); bn.running_var.uniform_(0.5, 2)
    x = torch.randn(2, 7, 3, 4, 5)
    s, b = sa.modules.fold_bn(bn)
    assert torch.allclose(x * s.reshape(1, -1, 1, 1, 1) + b.reshape(1, -1, 1, 1, 1), bn(x), atol=1e-6)


def test_shard_bounds_cover_the_batch_exactly_once():
    for n in (1, 7, 8, 32, 33):
        for world in (1, 2, 3, 8):
            spans = [sa.dist.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_two_term_fp16_block_floating_scheme_in_numpy():
    """The arithmetic behind the f16x3 engine (semstereo_amd/csrc/split_f16.h), restated in numpy: a power-of-two scale taken
    from the biased exponent of the block maximum puts that maximum into [2^14, 2^15); hi = fp16(x s), lo = fp16(x s - hi)
    then reconstruct every element within 2^-17 of the maximum to one fp32 ulp (2^-23 relative), and anything smaller to
    2^-39 of the maximum; the three products kept (hh, hl, lh) miss the exact product by at most the dropped lo*lo."""
    import numpy as np
    rng = np.random.default_rng(7)
    E_ONE = 141
    for mag in (1e-30, 1e-6, 1.0, 3e4, 1e20):
        x = (rng.standard_normal(4096) * mag).astype(np.float32)
        x[::7] *= np.float32(2.0 ** -20)                                   # elements far below the block maximum
        m = np.abs(x).max()
        e = int(np.float32(m).view(np.uint32) >> 23)                       # biased exponent of the maximum
        s = np.float32(2.0) ** np.float32(E_ONE - e)
        assert 2.0 ** 14 <= float(m) * float(s) < 2.0 ** 15
        xs = x * s                                                          # exact: a power of two
        hi = xs.astype(np.float16)
        lo = (xs - hi.astype(np.float32)).astype(np.float16)
        assert np.isfinite(hi).all() and np.isfinite(lo).all()
        err = np.abs(xs.astype(np.float64) - hi.astype(np.float64) - lo.astype(np.float64))
        big = np.abs(xs) >= 2.0 ** -3                                       # lo still a normal fp16 number
        assert (err[big] <= np.abs(xs[big]).astype(np.float64) * 2.0 ** -23).all()
        assert (err[~big] <= 2.0 ** -25).all() and 2.0 ** -25 <= float(m) * float(s) * 2.0 ** -39
        # products: (hi + lo)(hi' + lo') - (hh + hl + lh) = lo * lo'
        w = (rng.standard_normal(4096)).astype(np.float32)
        ew = int(np.float32(np.abs(w).max()).view(np.uint32) >> 23)
        ws = w * np.float32(2.0) ** np.float32(E_ONE - ew)
        wh = ws.astype(np.float16); wl = (ws - wh.astype(np.float32)).astype(np.float16)
        h64, l64, wh64, wl64 = (a.astype(np.float64) for a in (hi, lo, wh, wl))
        kept = h64 * wh64 + h64 * wl64 + l64 * wh64
        exact = (h64 + l64) * (wh64 + wl64)
        assert (np.abs(exact - kept) <= np.abs(h64 * wh64) * 2.0 ** -21 + 2.0 ** -46).all()


def test_unfused_helpers_stay_unfused_in_the_isa(tmp_path):
    """Round 4's root cause, pinned without a GPU: hipcc's __fmul_rn / __fadd_rn are plain operators and HIP compiles with
    -ffp-contract=fast, so a product written through them was still fused into the subtraction that consumed it (warp kernels:
    `ix - floor(ix)` on the UNROUNDED product).  ss::mul_rn / add_rn / sub_rn (csrc/common.h) carry `#pragma clang fp contract(off)`:
    cross-compile a probe for gfx950 and read the instructions -- the round trip must be v_mul, v_floor, v_sub (no fma on the
    product), while the same expression written with plain operators IS contracted (the compiler's default has not changed under us
    unnoticed: if this half ever fails the pragma may no longer be needed, not the other way round)."""
    import os
    import shutil
    import subprocess
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    src = tmp_path / "probe.hip"
    src.write_text('''
#include "%s/semstereo_amd/csrc/common.h"
extern "C" __global__ void probe_rn(const float* t, float half_w, float* out) {
    const float ix = ss::mul_rn(t[threadIdx.x] + 1.0f, half_w);
    out[threadIdx.x] = ss::sub_rn(ix, floorf(ix));
}
extern "C" __global__ void probe_plain(const float* t, float half_w, float* out) {
    const float ix = (t[threadIdx.x] + 1.0f) * half_w;
    out[threadIdx.x] = ix - floorf(ix);
}
''' % ROOT)
    asm = tmp_path / "probe.s"
    r = subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", str(src), "-o", str(asm)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    text = asm.read_text()

    def body(name):
        a = text.index(name + ":")
        return text[a:text.index("s_endpgm", a)]
    rn, plain = body("probe_rn"), body("probe_plain")
    assert "v_mul_f32" in rn and "v_floor_f32" in rn and "v_sub_f32" in rn, rn
    assert "v_fma_f32" not in rn and "v_fmac_f32" not in rn, rn
    assert "v_fma_f32" in plain or "v_fmac_f32" in plain, plain


def test_cache_generation_and_retained_entries():
    """engine._ParamCache: every build bumps cache_generation(); while a multi-stream caller retains replaced entries
    they are parked, not dropped, until drop_retired() -- what PairPipeline relies on to notice a rebuild inside a lane and to
    keep the replaced tensors alive for the pairs in flight on the other lanes."""
    import torch
    from semstereo_amd import engine as E
    c = E._ParamCache()
    w = torch.ones(3)
    g0 = E.cache_generation()
    a = c.get("k", [w], lambda: w * 2)
    assert E.cache_generation() == g0 + 1 and c.get("k", [w], lambda: None) is a and E.cache_generation() == g0 + 1
    w.mul_(2.0)                                     # in-place update: the version stamp changes
    E.retain_replaced(True)
    try:
        b = c.get("k", [w], lambda: w * 2)
        assert E.cache_generation() == g0 + 2 and b is not a
        assert any(old[1] is a for old in E._RETIRED), "the replaced entry must stay alive while retained"
        E.drop_retired()
        assert not E._RETIRED
    finally:
        E.retain_replaced(False)
    w.mul_(2.0)
    c.get("k", [w], lambda: w * 2)
    assert not E._RETIRED                           # nobody retains: replaced entries are simply dropped


def test_overlap_override_is_per_thread_and_never_touches_the_module():
    import threading
    from semstereo_amd import segment
    seen = {}

    def other():
        seen["other"] = getattr(segment._TLS, "overlap", None)
    with segment.overlap_override(False):
        t = threading.Thread(target=other); t.start(); t.join()
        seen["mine"] = segment._TLS.overlap
    assert seen == {"other": None, "mine": False} and getattr(segment._TLS, "overlap", None) is None
    assert "OVERLAP" not in segment.HotSegment(64).__dict__
