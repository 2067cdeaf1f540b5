"""Drop-in on a model shaped like the reference (tests/standin_model.py): `install()` rebinds the op library in
the model module's globals, `accelerate()` leaves forward() untouched, `accelerate(fuse_forward=True)` routes
inference calls through the fused hot segment -- all three must agree.  Run on the MI355X box: pytest -m gpu."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sa():
    import semstereo_amd
    assert torch.cuda.is_available(), "these tests need the MI355X"
    semstereo_amd._lib.load()
    return semstereo_amd


def _build(sa, **kw):
    import standin_model
    from oracle import detdata as dd
    net = standin_model.StandInSemStereo(64, sa.modules, **kw)
    with torch.no_grad():
        for i, (name, t) in enumerate(sorted(list(net.named_parameters()) + list(net.named_buffers()))):
            if name.endswith("num_batches_tracked") or name in ("gamma", "beta"):
                continue
            if name.endswith("running_var") or (name.endswith(".weight") and t.dim() == 1):
                t.copy_(dd.t_uniform(tuple(t.shape), 900 + i, 0.6, 1.4))
            elif t.dim() == 1:
                t.copy_(dd.t_uniform(tuple(t.shape), 900 + i, -0.1, 0.1))
            else:
                fan_in = t.shape[0] * 27 // 8 if (".conv5.0." in name or ".conv6.0." in name) else t[0].numel()
                a = (3.0 / fan_in) ** 0.5
                t.copy_(dd.t_uniform(tuple(t.shape), 900 + i, -a, a))
    return net.cuda().eval(), standin_model


def _images(B=1, H=128, W=160):
    from oracle import detdata as dd
    left = dd.t_normalish((B, 3, H, W), 951)
    right = torch.roll(left, shifts=-3, dims=3) + 0.05 * dd.t_normalish((B, 3, H, W), 952)
    return left.cuda(), right.cuda()


@pytest.mark.parametrize("att_only", [False, True])
def test_fused_forward_equals_untouched_forward(sa, att_only, deferral_on):
    net, module = _build(sa, att_weights_only=att_only)
    left, right = _images()
    previous = sa.install(module)
    try:
        assert sa.accelerate(net) == []                       # the stand-in is already built from the twins
        from semstereo_amd import deferred as dfr
        dfr.STATS["fused"].clear()
        ssr0 = dict(dfr.STATS["ssr"])
        before = dict(sa.modules.PATH_COUNTS)
        with torch.no_grad():
            (d0,), lab0 = net(left, right)                     # forward() untouched: HIP ops + HIP modules; deferred handles fuse
        assert isinstance(d0, torch.Tensor) and isinstance(lab0, torch.Tensor), "a deferred handle left the model"
        assert sa.modules.PATH_COUNTS["torch"] == before["torch"]
        if deferral_on:
            # models/SemStereo.py:311 vs :324 / :346: the head is called twice, an eval forward returns one result -- the other
            # handle (`pred_att_up` when the matching branch runs) is never computed: one launch, not two
            calls = 1 if att_only else 2
            assert dfr.STATS["ssr"]["deferred"] - ssr0["deferred"] == calls and dfr.STATS["ssr"]["computed"] - ssr0["computed"] == 1, dfr.STATS
        # (the two views of concat_feature share a launch pair only where its 2-D convolutions run on the HIP kernel: the f16x3 engine,
        # engine.conv2d_on_hip -- under the other engines that rule does not exist)
        pair = {"concat_feature_pair"} if sa.engine._conv2d_hip_on() else set()
        want_rules = {"gwc_patch_gate", "upsample_softmax_regression", "sample_strength", "topk_candidates"} | (set() if att_only else {"stem_by_halves"} | pair)
        assert set(dfr.STATS["fused"]) == want_rules and all(v == 1 for v in dfr.STATS["fused"].values()), dfr.STATS
        dfr.ENABLED = False
        try:
            with torch.no_grad():
                (d00,), _ = net(left, right)                   # ... and with deferral off: every statement its own launch(es)
        finally:
            dfr.ENABLED = True
        e00 = (d00 - d0).abs()
        assert float(e00.median()) <= 1e-4 and float((e00 <= 4e-3).float().mean()) >= 0.995, (float(e00.median()), float(e00.max()))
        calls = net.calls
        sa.accelerate(net, fuse_forward=True)
        before = dict(sa.modules.PATH_COUNTS)
        with torch.no_grad():
            (d1,), lab1 = net(left, right)
        assert net.calls == calls, "the reference-shaped forward ran although the fused one applies"
        assert sa.modules.PATH_COUNTS["torch"] == before["torch"]
        assert d1.shape == d0.shape == (1, 128, 160) and torch.equal(lab0, lab1)
        err = (d1 - d0).abs()
        # full-resolution disparities (x4): 1e-3 px at 1/4 scale = 4e-3 here; a top-2 flip on an isolated pixel is tolerated
        assert float(err.median()) <= 1e-4 and float((err <= 4e-3).float().mean()) >= 0.995, (float(err.median()), float(err.max()))
        # autograd / training calls are handed to the model's own forward()
        net.train()
        out = net(left, right)
        assert net.calls == calls + 1 and len(out) == 3 and len(out[0]) == (2 if att_only else 4)
        net.eval()
        sa.restore_forward(net)
        with torch.no_grad():
            net(left, right)
        assert net.calls == calls + 2
    finally:
        sa.uninstall(module, previous)


def test_fused_forward_matches_the_cpu_composition(sa):
    """The stand-in with the ORACLE op library and PyTorch layers on the CPU (no HIP anywhere) against the fused GPU path."""
    net, module = _build(sa)
    left, right = _images()
    import copy
    cpu_net = copy.deepcopy(net).cpu().eval()
    for p in cpu_net.parameters():           # eval() (running BN statistics) with autograd on: every twin takes its PyTorch path
        p.requires_grad_(True)
    (dc,), labc = cpu_net(left.cpu(), right.cpu())
    sa.accelerate(net, fuse_forward=True)
    with torch.no_grad():
        (dg,), labg = net(left, right)
    assert float((labg.cpu() - labc.detach()).abs().max()) <= 1e-4
    err = (dg.cpu() - dc.detach()).abs()
    assert float(err.median()) <= 1e-4 and float((err <= 4e-3).float().mean()) >= 0.99, (float(err.median()), float(err.max()))


def test_data_parallel_replicas_run_their_own_forward_in_two_threads(sa):
    """nn.DataParallel (main_us3d.py:100, test_us3d.py:58): one Python thread per replica, replicas built by copying the
    instance __dict__.  One GPU is enough to exercise the contract of SURVEY.md section 8(b) -- device_ids [0, 0] gives two
    replicas with their own broadcast weight copies that run the fused forward CONCURRENTLY from two threads on two
    streams' worth of launches: the routing must bind each replica (not the original), replicas must not share packed
    weights, and the C ABI must be re-entrant.  The result must equal the plain single-thread call."""
    import torch.nn as nn
    net, module = _build(sa)
    left, right = _images(B=2)
    sa.accelerate(net, fuse_forward=True)
    with torch.no_grad():
        (want,), lab = net(left, right)
    try:
        dp = nn.DataParallel(net, device_ids=[0, 0])
    except Exception as e:                                   # a torch build that refuses duplicate ids
        pytest.skip(f"DataParallel(device_ids=[0, 0]) not accepted: {e}")
    before = dict(sa.modules.PATH_COUNTS)
    for _ in range(3):
        with torch.no_grad():
            (got,), lab2 = dp(left, right)
        assert sa.modules.PATH_COUNTS["torch"] == before["torch"], "a replica fell back to PyTorch layers"
        assert got.shape == want.shape and torch.equal(lab2, lab)
        err = (got - want).abs()
        assert float(err.median()) <= 1e-4 and float((err <= 4e-3).float().mean()) >= 0.995, (float(err.median()), float(err.max()))
    sa.restore_forward(net)


def test_data_parallel_replicas_with_the_untouched_forward(sa, deferral_on):
    """The reference's own multi-GPU form (nn.DataParallel, test_us3d.py:58) on a model that was only install()ed and
    accelerate()d -- forward() untouched, no fuse_forward: each replica's thread builds its own deferred handles (they are
    per-call objects; deferral state is thread-local), every fused rule fires once per replica and call, the replicas pack
    their weights once per device and weight version (modules._ReplicaCache), and the result equals the single-thread call."""
    import torch.nn as nn
    from semstereo_amd import deferred as dfr
    net, module = _build(sa)
    left, right = _images(B=2)
    previous = sa.install(module)
    try:
        with torch.no_grad():
            (want,), lab = net(left, right)
        try:
            dp = nn.DataParallel(net, device_ids=[0, 0])
        except Exception as e:
            pytest.skip(f"DataParallel(device_ids=[0, 0]) not accepted: {e}")
        before = dict(sa.modules.PATH_COUNTS)
        for it in range(3):
            dfr.STATS["fused"].clear()
            with torch.no_grad():
                (got,), lab2 = dp(left, right)
            assert sa.modules.PATH_COUNTS["torch"] == before["torch"], "a replica fell back to PyTorch layers"
            assert set(dfr.STATS["fused"]) == ({"gwc_patch_gate", "upsample_softmax_regression", "sample_strength", "topk_candidates", "stem_by_halves"}
                                               | ({"concat_feature_pair"} if sa.engine._conv2d_hip_on() else set())) and all(v == 2 for v in dfr.STATS["fused"].values()), dfr.STATS
            assert got.shape == want.shape and torch.equal(lab2, lab)
            err = (got - want).abs()
            assert float(err.median()) <= 1e-4 and float((err <= 4e-3).float().mean()) >= 0.995, (float(err.median()), float(err.max()))
        # the replicas' packed weights live on the original's cache, keyed by device: built in the first call, reused after
        store = net.hourglass.__dict__["_ss_cache"]._store
        assert any(isinstance(k, tuple) and k[0] == "replica" for k in store), list(store)[:5]
    finally:
        sa.uninstall(module, previous)


def test_graphed_segment_replays_the_eager_result(sa):
    """GraphedSegment: the step captured into a HIP graph (both streams) gives bit-identical outputs to the eager call, also
    for new inputs copied into the captured buffers."""
    from golden import cases
    from oracle import hot_segment as oseg
    fl4, fr4, fl8, fr8, maxdisp = cases.segment_inputs("s128")
    seg = sa.HotSegment(maxdisp)
    seg.load_state_dict(oseg.deterministic_params(), strict=False)
    seg = seg.cuda().eval()
    ins = [t.cuda() for t in (fl4, fr4, fl8, fr8)]
    with torch.no_grad():
        want = {k: v.clone() for k, v in seg(*ins).items()}
    g = sa.GraphedSegment(seg, *ins)
    got = g(*ins)
    for k in ("pred", "pred_att", "samples", "att_topk"):
        assert torch.equal(got[k], want[k]), k
    ins2 = [torch.roll(t, shifts=3, dims=-1).contiguous() for t in ins]
    with torch.no_grad():
        want2 = {k: v.clone() for k, v in seg(*ins2).items()}
    got2 = g(*ins2)
    for k in ("pred", "pred_att", "samples"):
        assert torch.equal(got2[k], want2[k]), k


def test_pair_pipeline_around_a_whole_model_with_the_untouched_forward(sa, deferral_on):
    """semstereo_amd.PairPipeline around a model shaped like the reference (backbone + hot segment + SSR head, forward() untouched,
    install() + accelerate() only): consecutive image pairs on three HIP streams give, bit for bit, what plain calls give."""
    net, module = _build(sa)
    previous = sa.install(module)
    try:
        sa.accelerate(net)
        pairs = []
        for i in range(5):
            left, right = _images()
            pairs.append((left + 0.01 * i, torch.roll(right, shifts=i, dims=2)))
        with torch.no_grad():
            want = [net(l, r)[0][0].clone() for l, r in pairs]
        torch.cuda.synchronize()
        pipe = sa.PairPipeline(net, lanes=3)
        got = [pipe(l, r) for l, r in pairs]
        pipe.synchronize()
        for (g,), w in zip([o[0] for o in got], want):
            assert torch.equal(g, w), float((g - w).abs().max())
    finally:
        sa.uninstall(module, previous)
