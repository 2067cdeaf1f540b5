"""GPU parity at BASELINE.json's FULL sizes through size-independent properties (the oracle cannot run these
sizes in seconds, so here it is applied to sampled slices, and the rest are identities the reference
algorithm satisfies at any size): linearity / cross-op consistency of the volume builders, closed-form
values, determinism, batch (= shard) invariance, candidate-set invariants of the hot segment.
Run on the MI355X box: pytest -m gpu."""
import pytest
import torch

from oracle import ops as oops

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sa():
    import semstereo_amd
    assert torch.cuda.is_available(), "these tests need the MI355X"
    semstereo_amd._lib.load()
    return semstereo_amd


def _feat(shape, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return torch.randn(shape, generator=g, device="cuda")


def test_gwc_volume_batch8_properties(sa):
    """configs[2]: 2 x [8,256,128,128] -> [8,32,32,128,128] (maxdisp//8 = 16, 32 groups)."""
    B, C, H, W, m, G = 8, 256, 128, 128, 16, 32
    fl, fr = _feat((B, C, H, W), 1), _feat((B, C, H, W), 2)
    v = sa.ops.build_gwc_volume(fl, fr, m, G)
    assert v.shape == (B, G, 2 * m, H, W)
    # linearity in each argument: powers of two are exact in fp32 -> bit-identical
    assert torch.equal(sa.ops.build_gwc_volume(fl * 4.0, fr, m, G), v * 4.0)
    assert torch.equal(sa.ops.build_gwc_volume(fl, fr * 0.5, m, G), v * 0.5)
    # the zero-disparity plane is the plain group-wise correlation (models/submodule.py:190-196 vs :198-211)
    assert torch.equal(v[:, :, m], sa.ops.groupwise_correlation(fl, fr, G))
    # plane d is the correlation with the right image shifted by d - m, zero where the partner leaves the image
    for d in (0, 5, 2 * m - 1):
        s = d - m
        sh = torch.zeros_like(fr)
        if s >= 0:
            sh[..., s:] = fr[..., :W - s]
        else:
            sh[..., :W + s] = fr[..., -s:]
        ref = sa.ops.groupwise_correlation(fl, sh, G)
        valid = torch.zeros(W, dtype=torch.bool, device="cuda")
        valid[max(s, 0):W + min(s, 0)] = True
        assert torch.equal(v[:, :, d][..., valid], ref[..., valid])
        assert float(v[:, :, d][..., ~valid].abs().max()) == 0.0 if (~valid).any() else True
    # the oracle on sampled (batch, row) slices, both variants
    vn = sa.ops.build_gwc_volume_norm(fl, fr, m, G)
    for b, y in ((0, 0), (3, 77), (7, 127)):
        a, c = fl[b:b + 1, :, y:y + 1].cpu(), fr[b:b + 1, :, y:y + 1].cpu()
        assert float((v[b:b + 1, :, :, y:y + 1].cpu() - oops.build_gwc_volume(a, c, m, G)).abs().max()) <= 1e-6
        assert float((vn[b:b + 1, :, :, y:y + 1].cpu() - oops.build_gwc_volume_norm(a, c, m, G)).abs().max()) <= 2e-6
    # normalised volume: |value| <= 1 (Cauchy-Schwarz on unit vectors, mean over 8 channels) and scale invariance
    assert float(vn.abs().max()) <= 1.0 + 1e-6
    assert float((sa.ops.build_gwc_volume_norm(fl * 8.0, fr * 0.25, m, G) - vn).abs().max()) <= 2e-5


def test_concat_volume_full_size_is_pure_data_movement(sa):
    """natural shape of build_concat_volume at 1024^2 / maxdisp 128: 2 x [1,32,256,256] -> [1,64,64,256,256] (1.07 GB)."""
    C, H, W, m = 32, 256, 256, 32
    fl, fr = _feat((1, C, H, W), 3), _feat((1, C, H, W), 4)
    v = sa.ops.build_concat_volume(fl, fr, m)
    assert v.shape == (1, 2 * C, 2 * m, H, W)
    xs = torch.arange(W, device="cuda")
    for d in (0, 31, 32, 63):
        s = d - m
        valid = (xs - s >= 0) & (xs - s < W)
        assert torch.equal(v[0, :C, d][..., valid], fl[0][..., valid])                    # left half: the left image
        assert torch.equal(v[0, C:, d][..., valid], fr[0][..., (xs - s)[valid]])          # right half: shifted right image
        assert float(v[0, :, d][..., ~valid].abs().sum()) == 0.0                            # both halves zero outside
    # checksum of checksums: every valid column of the left half repeats the left image once per disparity
    n_valid = sum(int(((xs - (d - m) >= 0) & (xs - (d - m) < W)).sum()) for d in range(2 * m))
    assert n_valid == sum(W - abs(d - m) for d in range(2 * m))
    y = 100
    assert torch.equal(v[:, :, :, y:y + 1].cpu(), oops.build_concat_volume(fl[:, :, y:y + 1].cpu(), fr[:, :, y:y + 1].cpu(), m))


def test_regressions_closed_forms_full_size(sa):
    m, H, W = 32, 256, 256
    # uniform probabilities: E[d] over [-m, m) is exactly -0.5; a one-hot plane returns its disparity
    p = torch.full((2, 2 * m, H, W), 1.0 / (2 * m), device="cuda")
    assert float((sa.ops.disparity_regression(p, m) + 0.5).abs().max()) <= 1e-6
    onehot = torch.zeros((1, 2 * m, H, W), device="cuda")
    onehot[:, 7] = 1.0
    assert torch.equal(sa.ops.disparity_regression(onehot, m), torch.full((1, H, W), 7.0 - m, device="cuda"))
    # regression_topk: with a single dominant cost the answer is that candidate, whatever the rest is
    cost = _feat((1, 24, H, W), 5)
    base = torch.randperm(2 * m, generator=torch.Generator().manual_seed(6))[:24].sort().values.float() - m
    samples = base.reshape(1, 24, 1, 1).expand(1, 24, H, W).contiguous().cuda()
    big = cost.clone()
    big[:, 11] += 100.0
    assert float((sa.ops.regression_topk(big, samples, 2)[:, 0] - samples[:, 11]).abs().max()) <= 1e-5
    # permutation invariance over the candidate axis (sort is part of the op)
    perm = torch.randperm(24, device="cuda")
    assert float((sa.ops.regression_topk(cost[:, perm].contiguous(), samples[:, perm].contiguous(), 2)
                  - sa.ops.regression_topk(cost, samples, 2)).abs().max()) <= 1e-5


@pytest.mark.parametrize("size", [(1024, 128), (2048, 192)])
def test_hot_segment_full_size_invariants(sa, size):
    """configs[1] / configs[4] shapes: determinism, batch (= shard) invariance, candidate-set invariants."""
    import bench
    Hf, maxdisp = size
    m4 = maxdisp // 4
    seg = sa.HotSegment(maxdisp).cuda().eval()
    bench.init_unit_gain(seg, 4321)
    fl4a, fr4a = bench.synth_features(1, 128, Hf // 4, Hf // 4, maxdisp // 8, 11, "cuda")
    fl8a, fr8a = bench.synth_features(1, 256, Hf // 8, Hf // 8, maxdisp // 16, 12, "cuda")
    before = dict(sa.modules.PATH_COUNTS)
    with torch.no_grad():
        r1 = seg(fl4a, fr4a, fl8a, fr8a)
        r2 = seg(fl4a, fr4a, fl8a, fr8a)
    assert sa.modules.PATH_COUNTS["torch"] == before["torch"], "a PyTorch fallback ran"
    for k in ("pred", "pred_att", "samples", "att_topk"):
        assert torch.equal(r1[k], r2[k]), f"{k} differs between two identical launches"
    s = r1["samples"]
    assert s.shape == (1, 24, Hf // 4, Hf // 4)
    assert bool((s[:, 1:] > s[:, :-1]).all()), "candidates must be strictly ascending (24 distinct disparities)"
    assert float(s.min()) >= -m4 and float(s.max()) <= m4 - 1 and bool((s == s.round()).all())
    a = r1["att_topk"]
    assert float(a.min()) > 0.0 and float(a.sum(dim=2).max()) <= 1.0 + 1e-5      # 24 of the D4 softmax probabilities
    # pred_att / pred are convex combinations of the pixel's candidates
    for k in ("pred_att", "pred"):
        v = r1[k].reshape(1, 1, Hf // 4, Hf // 4)
        assert bool((v >= s[:, :1] - 1e-4).all()) and bool((v <= s[:, -1:] + 1e-4).all()), k
    assert bool(torch.isfinite(r1["pred"]).all())
    if Hf == 1024:
        # two different pairs in one batch == each pair alone: nothing couples batch elements, which is
        # what lets the path shard over GPUs with no collective
        fl4b, fr4b = bench.synth_features(1, 128, Hf // 4, Hf // 4, maxdisp // 8, 21, "cuda")
        fl8b, fr8b = bench.synth_features(1, 256, Hf // 8, Hf // 8, maxdisp // 16, 22, "cuda")
        with torch.no_grad():
            rb = seg(fl4b, fr4b, fl8b, fr8b)
            rab = seg(torch.cat((fl4a, fl4b)), torch.cat((fr4a, fr4b)), torch.cat((fl8a, fl8b)), torch.cat((fr8a, fr8b)))
        # (the 2-D convolutions / GEMMs of the gates are MIOpen / rocBLAS calls whose algorithm may change
        # with the batch size, so agreement is to rounding, not bitwise)
        for one, idx in ((r1, 0), (rb, 1)):
            assert float((rab["samples"][idx:idx + 1] == one["samples"]).float().mean()) >= 0.9999
            assert float((rab["pred_att"][idx:idx + 1] - one["pred_att"]).abs().median()) <= 1e-5
            d = (rab["pred"][idx:idx + 1] - one["pred"]).abs()
            assert float(d.median()) <= 1e-5 and float((d <= 1e-3).float().mean()) >= 0.99
