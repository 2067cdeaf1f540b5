"""A miniature model with the reference SemStereo's attribute names and the reference forward()'s call
order (models/SemStereo.py:184-346), for testing the drop-in on a box where the reference itself is
absent.  TEST INFRASTRUCTURE: the out-of-scope producers (backbone, FeatUp, segmentation heads, spx_*
up-sampling chain) are single layers with the right shapes; the hot-path sub-modules are this repo's
twins; the op library is looked up BY BARE NAME in this module's globals, exactly like the reference's
`from models.submodule import *`, so `semstereo_amd.install(standin_model)` rebinds it the same way.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle.ops import (SpatialTransformer_grid, build_gwc_volume_norm, disparity_regression,  # noqa: F401
                        disparity_variance, regression_topk)


class _Pyramid(nn.Module):
    """stand-in for Feature: five maps at 1/2 .. 1/32 with 32, 64, 128, 256, 384 channels"""

    def __init__(self):
        super().__init__()
        self.convs = nn.ModuleList([nn.Conv2d(3, c, 3, padding=1) for c in (32, 64, 128, 256, 384)])

    def forward(self, x):
        return [conv(F.avg_pool2d(x, 2 ** (i + 1))) for i, conv in enumerate(self.convs)]


class _Pair(nn.Module):
    def forward(self, a, b):
        return a, b


class _Head(nn.Module):
    def __init__(self, cin, cout, up):
        super().__init__()
        self.conv, self.up = nn.Conv2d(cin, cout, 1), up

    def forward(self, x):
        return F.interpolate(self.conv(x), scale_factor=self.up, mode="bilinear")


class _Merge(nn.Module):
    """stand-in for Conv2x: up-sample `x` by 2 and mix with `skip`"""

    def __init__(self, cx, cs, cout):
        super().__init__()
        self.conv = nn.Conv2d(cx + cs, cout, 1)

    def forward(self, x, skip):
        return self.conv(torch.cat((F.interpolate(x, scale_factor=2.0, mode="nearest"), skip), dim=1))


class StandInSemStereo(nn.Module):
    def __init__(self, maxdisp, M, seg_if=True, stereo_if=True, att_weights_only=False, num_classes=6):
        super().__init__()
        self.maxdisp, self.seg_if, self.stereo_if, self.att_weights_only = maxdisp, seg_if, stereo_if, att_weights_only
        self.chans2 = [32, 128, 256, 384, 256]
        self.feature, self.feature_up = _Pyramid(), _Pair()
        self.head_l, self.head_r = _Head(32, num_classes, 2), _Head(32, num_classes, 2)
        self.chal_0, self.chal_1, self.chal_2 = nn.Conv2d(32, 32, 1), nn.Conv2d(64, 128, 1), nn.Conv2d(128, 256, 1)
        self.chal_3, self.chal_4 = nn.Conv2d(256, 384, 1), nn.Conv2d(384, 256, 1)
        self.spx32_16, self.spx16_8 = _Merge(256, 384, 64), _Merge(64, 256, 64)
        self.spx8_4, self.spx4_2 = _Merge(64, 128, 32), _Merge(32, 32, 32)
        self.spx2 = _Head(32, num_classes, 2)
        self.gamma, self.beta = nn.Parameter(torch.full((1,), 0.25)), nn.Parameter(2 * torch.ones(1))
        self.patch = M.DepthwisePatch(32)
        self.corr_feature_att_8, self.concat_feature_att_4 = M.channelAtt(32, 256), M.channelAtt(32, 128)
        self.hourglass_att, self.hourglass = M.hourglass(32), M.hourglass2(32)
        self.classif_att_, self.classif = M.Classifier(32), M.Classifier(32)
        self.concat_feature = M.ConcatFeature(128)
        self.concat_stem = M.BasicConv(64, 32, is_3d=True, kernel_size=3, stride=1, padding=1)
        self.ssr_upsample = M.SSR_upsample(num_classes)
        self.propagation, self.propagation_prob = M.Propagation(), M.Propagation_prob()
        self.calls = 0

    def concat_volume_generator(self, left_input, right_input, disparity_samples):
        right_w, left_b = SpatialTransformer_grid(left_input, right_input, disparity_samples)
        return torch.cat((left_b, right_w), dim=1)

    def forward(self, left, right):
        self.calls += 1
        fl, fr = self.feature(left), self.feature(right)
        fl, fr = self.feature_up(fl, fr)
        pred_label = self.head_l(fl[0])
        pred_label_r = self.head_r(fr[0])
        for i, name in enumerate(("chal_0", "chal_1", "chal_2", "chal_3", "chal_4")):
            fl[i] = getattr(self, name)(fl[i])
        fr[1], fr[2] = self.chal_1(fr[1]), self.chal_2(fr[2])
        xspx = self.spx32_16(fl[4], fl[3])
        xspx = self.spx16_8(xspx, fl[2])
        xspx = self.spx8_4(xspx, fl[1])
        xspx = self.spx4_2(xspx, fl[0])
        spx_pred = self.spx2(xspx)
        # the hot segment, statement by statement in the reference's order (models/SemStereo.py:273-324): an op-by-op drop-in
        # sees exactly this call sequence, and semstereo_amd.deferred recognises exactly this text
        m4 = self.maxdisp // 4
        volume8 = build_gwc_volume_norm(fl[2], fr[2], self.maxdisp // 8, self.chans2[2] // 8)
        volume8 = self.patch(volume8)
        logits8 = self.corr_feature_att_8(volume8, fl[2])
        logits8 = self.hourglass_att(logits8)
        logits8 = self.classif_att_(logits8)
        logits4 = F.interpolate(logits8, [m4 * 2, left.size()[2] // 4, left.size()[3] // 4], mode="trilinear")
        p0 = torch.squeeze(logits4, 1)
        p0 = F.softmax(p0, dim=1)
        d0 = disparity_regression(p0, m4)
        v0 = disparity_variance(p0, m4, d0.unsqueeze(1))
        v0 = self.beta + self.gamma * v0
        v0 = torch.sigmoid(v0)
        v5 = self.propagation(v0)
        d5 = self.propagation(d0.unsqueeze(1))
        right5, left5 = SpatialTransformer_grid(fl[1], fr[1], d5)
        w5 = (left5 * right5).mean(dim=1)
        w5 = torch.softmax(w5 * v5, dim=1)
        aw = self.propagation_prob(logits4)
        aw = aw * w5.unsqueeze(2)
        aw = torch.sum(aw, dim=1, keepdim=True)
        aw_prob = F.softmax(aw, dim=2)
        _, order = aw_prob.sort(2, True)
        ind_k = order[:, :, :24]
        ind_k = ind_k.sort(2, False)[0]
        att_topk = torch.gather(aw_prob, 2, ind_k)
        samples = ind_k.squeeze(1).float() - self.maxdisp // 4
        pk = torch.gather(aw, 2, ind_k).squeeze(1)
        pk = F.softmax(pk, dim=1)
        pred_att = pk * samples
        pred_att = torch.sum(pred_att, dim=1)
        pred_att_up = self.ssr_upsample(pred_att.unsqueeze(1), spx_pred, pred_label)
        if not self.att_weights_only:
            cl = self.concat_feature(fl[1])
            cr = self.concat_feature(fr[1])
            volume = self.concat_volume_generator(cl, cr, samples)
            volume = att_topk * volume
            volume = self.concat_stem(volume)
            volume = self.concat_feature_att_4(volume, fl[1])
            cost = self.hourglass(volume)
            cost = self.classif(cost)
            pred = regression_topk(cost.squeeze(1), samples, 2)
            pred_up = self.ssr_upsample(pred, spx_pred, pred_label)
        if self.training:
            outs = [pred_att_up * 4, pred_att * 4] if self.att_weights_only else \
                [pred_up * 4, pred.squeeze(1) * 4, pred_att_up * 4, pred_att * 4]
            return (outs, pred_label, pred_label_r) if self.seg_if else outs
        outs = [pred_att_up * 4] if self.att_weights_only else [pred_up * 4]
        return (outs, pred_label) if self.seg_if else outs
