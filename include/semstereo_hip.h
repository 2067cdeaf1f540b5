/*
 * semstereo_hip.h -- C ABI of libsemstereo_hip.so: the MI355X (gfx950) kernels for the
 * SemStereo cost-volume + 3-D aggregation hot path.
 *
 * The reference (chenchen235/SemStereo) is pure Python: its "operator interface" for this
 * path is a set of module-level callables in models/submodule.py that models/SemStereo.py
 * star-imports and calls by bare name.  Each entry point below is the
 * device-side replacement of one of those callables (or of one nn.Module.forward on the
 * path); the Python wrappers in semstereo_amd/ops.py keep the reference names and positional
 * signatures and call these through ctypes.
 *
 * Conventions
 *  - all tensors are fp32, contiguous, in the reference's layouts: NCHW feature maps,
 *    NCDHW volumes;
 *  - pointers are BORROWED device pointers (owned by the caller's allocator); outputs are
 *    fully written by the kernels (no pre-zeroing needed) unless stated;
 *  - `stream` is a hipStream_t (NULL = the null stream); launches are asynchronous;
 *  - re-entrant, no global mutable state, current device of the calling thread;
 *  - return value: SS_OK (0) or a negative ss_status; nothing throws across the ABI.
 */
#ifndef SEMSTEREO_HIP_H
#define SEMSTEREO_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef void* ss_stream_t; /* hipStream_t */

enum ss_status {
    SS_OK = 0,
    SS_ERR_INVALID = -1,     /* null pointer, non-positive size, C % groups != 0 ... */
    SS_ERR_UNSUPPORTED = -2, /* shape outside what the kernels are built for */
    SS_ERR_LAUNCH = -3       /* hipGetLastError() != hipSuccess after the launch */
};

/* ABI version (19 since round 6 (13-18 earlier in it); 12 since round 5; ss_abi_version() is authoritative and semstereo_amd/_lib.py checks it at load): bumped on any
 * signature change or new entry point the Python binding requires. */
int ss_abi_version(void);
/* The library's tuning switches (SS_CONV_TILE, SS_GWC_STREAM, SS_WARP_STREAM, SS_WARP_VEC, SS_DECONV_SPLIT, ...; A/B
 * measurement aids) are read from the environment ONCE per process; call this after changing
 * one of them.  No reference counterpart (the reference has no tuning surface). */
int ss_reload_tuning(void);
/* Static, human-readable text for an ss_status. */
const char* ss_status_string(int status);
/* hipGetErrorString of the last SS_ERR_LAUNCH on this thread ("" if none). */
const char* ss_last_hip_error(void);

/* ---- disparity ranges -------------------------------------------------------------------
 * Every volume / regression entry point takes (dmin, ndisp): plane p of a volume holds disparity dmin + p.
 *   signed op set    models/submodule.py   (models/SemStereo.py):      dmin = -maxdisp, ndisp = 2 * maxdisp
 *   unsigned op set  models/submodule_.py  (what models/SemStereo_WHU.py:279,305 is written for): dmin = 0, ndisp = maxdisp */

/* ---- group-wise correlation ----------------------------------------------------------
 * groupwise_correlation(fea1, fea2, num_groups)       models/submodule.py:190-196
 * groupwise_correlation_norm(...)                     models/submodule.py:213-221
 * out[b,g,y,x] = mean_c f1[b,g*Cg+c,y,x] * f2[b,g*Cg+c,y,x]; with normalize != 0 each group
 * vector is first divided by (its L2 norm over the Cg channels + 1e-5).  [B,C,H,W]^2 -> [B,G,H,W] */
int ss_groupwise_correlation_fwd(const float* fea1, const float* fea2, float* out,
                                 int B, int C, int H, int W, int groups, int normalize,
                                 ss_stream_t stream);

/* build_gwc_volume(ref, tgt, maxdisp, num_groups)      models/submodule.py:198-211   (unsigned: models/submodule_.py:188-198)
 * build_gwc_volume_norm(...)                           models/submodule.py:224-238 (live: SemStereo.py:273; unsigned :211-221)
 * out[b,g,p,y,x] = corr(ref[...,x], tgt[...,x-(dmin+p)]) for p in [0, ndisp), 0 where
 * the partner column leaves the image.  [B,C,H,W]^2 -> [B,G,ndisp,H,W] */
int ss_gwc_volume_fwd(const float* ref, const float* tgt, float* out,
                      int B, int C, int H, int W, int dmin, int ndisp, int groups, int normalize,
                      ss_stream_t stream);
/* models/SemStereo.py:273-276 in one launch: build_gwc_volume[_norm] -> `patch` (depthwise Conv3d, kernel (1,3,3),
 * padding (0,1,1), no bias; patch_w [G,1,1,3,3]) -> channelAtt gate (sigmoid(gate_logits[b,g,y,x]) broadcast over the
 * disparities; NULL: no gate).  Bit-identical to ss_gwc_volume_fwd + ss_depthwise_patch_fwd.  Needs W % 4 == 0,
 * dmin % 4 == 0, ndisp % 8 == 0, C / groups in {4, 8}, 16-byte aligned pointers: SS_ERR_UNSUPPORTED otherwise (use the two calls). */
int ss_gwc_patch_gate_fwd(const float* ref, const float* tgt, const float* patch_w, const float* gate_logits, float* out,
                          int B, int C, int H, int W, int dmin, int ndisp, int groups, int normalize, ss_stream_t stream);
/* gradients of the UN-normalised volume w.r.t. ref and tgt (both fully written). */
int ss_gwc_volume_bwd(const float* grad_out, const float* ref, const float* tgt,
                      float* grad_ref, float* grad_tgt,
                      int B, int C, int H, int W, int dmin, int ndisp, int groups,
                      ss_stream_t stream);

/* build_concat_volume(ref, tgt, maxdisp)               models/submodule.py:173-187   (unsigned: models/submodule_.py:166-177)
 * out[b,0:C,p,y,x] = ref[b,:,y,x], out[b,C:2C,p,y,x] = tgt[b,:,y,x-(dmin+p)], 0 where the partner column leaves
 * the image -- in BOTH halves with mask_left != 0 (the signed form), in the right half only with mask_left == 0 (the
 * unsigned form copies the left image unmasked).  [B,C,H,W]^2 -> [B,2C,ndisp,H,W] */
int ss_concat_volume_fwd(const float* ref, const float* tgt, float* out,
                         int B, int C, int H, int W, int dmin, int ndisp, int mask_left, ss_stream_t stream);
int ss_concat_volume_bwd(const float* grad_out, float* grad_ref, float* grad_tgt,
                         int B, int C, int H, int W, int dmin, int ndisp, int mask_left, ss_stream_t stream);

/* SpatialTransformer_grid(x, y, disp_range_samples)    models/submodule.py:265-288
 * y_warped[b,c,j,h,w] = bilinear(y[b,c], row h, col w - disp[b,j,h,w]) with zeros padding and
 * align_corners=True, coordinates taken through the reference's normalise/unnormalise fp32
 * round trip; x_warped[b,c,j,h,w] = x[b,c,h,w].  x_warped may be NULL (not written).
 * x,y [B,C,H,W], disp [B,nd,H,W] -> [B,C,nd,H,W] each. */
int ss_warp_sampled_fwd(const float* x, const float* y, const float* disp,
                        float* y_warped, float* x_warped,
                        int B, int C, int H, int W, int nd, ss_stream_t stream);
/* its backward (autograd of the reference's meshgrid -> normalise -> F.grid_sample composition): grad_y_warped,
 * grad_x_warped [B,C,nd,H,W] (either may be NULL) -> grad_x = sum over the candidates of grad_x_warped, grad_y (scatter of
 * the bilinear weights; zeroed here, accumulated with fp32 atomics), grad_disp [B,nd,H,W] (the live use at
 * models/SemStereo.py:291, where the candidates derive from pred_att).  Outputs may be NULL when not needed. */
int ss_warp_sampled_bwd(const float* grad_y_warped, const float* grad_x_warped, const float* y, const float* disp,
                        float* grad_x, float* grad_y, float* grad_disp, int B, int C, int H, int W, int nd,
                        ss_stream_t stream);
/* Fused form of SemStereo.concat_volume_generator + `att_topk * volume`
 * (models/SemStereo.py:241-244, 318): out[b,0:C,j] = att[b,j] * left[b,:],
 * out[b,C:2C,j] = att[b,j] * warp(right)[b,:,j].  att [B,nd,H,W] may be NULL (no gating).
 * -> [B,2C,nd,H,W].  left == NULL: only the right half, -> [B,C,nd,H,W] (its left half then enters
 * concat_stem through ss_stem_left_fwd). */
int ss_concat_sampled_fwd(const float* left, const float* right, const float* disp, const float* att,
                          float* out, int B, int C, int H, int W, int nd, ss_stream_t stream);
/* (r06, training) Backward of ss_concat_sampled_fwd with both halves: grad_out [B,2C,nd,H,W] -> grad_left [B,C,H,W] (= sum_j att *
 * grad of the broadcast half), grad_right [B,C,H,W] (the warp's backward on att * grad of the warped half: bilinear taps of
 * F.grid_sample, zeros padding, align_corners=True as the forward) and grad_att [B,nd,H,W] (= sum_c grad . the ungated volume); any of
 * the three may be NULL; att may be NULL when grad_att is.  The candidates `disp` are constants (the indices of
 * models/SemStereo.py:299-305): no gradient.  `margin`: a bound on |disp| the caller expects (a performance hint: taps further away are
 * added one atomic at a time instead of through the LDS windows; any value gives the same result).  W % 64 == 0 (every
 * quarter-resolution width of the path), else SS_ERR_UNSUPPORTED: the caller composes ss_warp_sampled_bwd with its own cat / multiply. */
int ss_concat_sampled_bwd(const float* grad_out, const float* left, const float* right, const float* disp, const float* att,
                          float* grad_left, float* grad_right, float* grad_att, int B, int C, int H, int W, int nd, int margin,
                          ss_stream_t stream);
/* The right (warped) half of the same volume, att[b,j] * warp(right)[b,:,j] (models/SemStereo.py:241-244, 316-318), written
 * PRE-SPLIT for the matrix core: every value as the two fp16 terms of x * 2^(141 - e), 8 channels per 16-byte slot,
 *   xs [B][C/8][2 terms][nd][H][W][8] fp16 (C % 8 == 0, 4 bytes per value: the footprint of the fp32 volume),
 *   xexp int[3*B]: [0,B) the block exponent e of each batch element (from the bound max|right[b]| * max|att[b]|),
 *                  [B,3B) scratch for the two maxima.
 * Consumed by ss_conv3d_presplit_fwd, which stages it by LDS-DMA with no conversion.  att may be NULL. */
int ss_concat_sampled_presplit_fwd(const float* right, const float* disp, const float* att, void* xs, int* xexp,
                                   int B, int C, int H, int W, int nd, ss_stream_t stream);
/* The left (broadcast) half of concat_stem (models/SemStereo.py:319) by linearity:
 *   out[b,co,j,h,w] = sum_{kd,kh,kw} att[b, j+kd-1, h+kh-1, w+kw-1] * q[b, tap*Cout + co, h+kh-1, w+kw-1]
 * (zero outside), tap = kd*9+kh*3+kw, q [B,27*Cout,H,W] = the 1x1 convolution of the 2-D left concat
 * features with concat_stem's left-half weights (ss_conv3d_pointwise_bf16s_fwd), att [B,nd,H,W] the
 * top-k attention weights, out [B,Cout,nd,H,W]; nd in {6, 24, 32}, Cout % 4 == 0.  The result is the
 * `residual` operand of the right half's ss_conv3d[_bf16s]_fwd. */
int ss_stem_left_fwd(const float* q, const float* att, float* out, int B, int Cout, int nd, int H, int W,
                     ss_stream_t stream);
/* The same, with q computed on the fly on the matrix core and never written: left [B,C,H,W] (C = 32) is the
 * 2-D left concat feature map, wsplit = ss_pack_pointwise_weights_bf16s of the [Cout/2 * 64, C] matrix whose
 * row pair*64 + tap*2 + c is (scale *) W[2*pair + c, :C, tap] (rows 54-63 of every pair zero). */
int ss_stem_left_fused_fwd(const float* left, const void* wsplit, const float* att, float* out, int B, int C,
                           int Cout, int nd, int H, int W, int nterms, ss_stream_t stream);
/* Fused form of models/SemStereo.py:291-292: mean over channels of left * warp(right):
 * x,y [B,C,H,W], disp [B,nd,H,W] -> [B,nd,H,W] */
int ss_warp_correlation_fwd(const float* x, const float* y, const float* disp, float* out,
                            int B, int C, int H, int W, int nd, ss_stream_t stream);

/* disparity_regression(x, maxdisp)                     models/submodule.py:164-170   (unsigned: models/submodule_.py:159-163)
 * out[b,y,x] = sum_p prob[b,p,y,x] * (dmin + p), p in [0, ndisp).  [B,ndisp,H,W] -> [B,H,W] */
int ss_disparity_regression_fwd(const float* prob, float* out, int B, int dmin, int ndisp, int H, int W,
                                ss_stream_t stream);
int ss_disparity_regression_bwd(const float* grad_out, float* grad_prob, int B, int dmin, int ndisp, int H, int W,
                                ss_stream_t stream);
/* disparity_variance(x, maxdisp, disparity)            models/submodule.py:257-263   (unsigned: models/submodule_.py:239-245)
 * out[b,0,y,x] = sum_p prob[b,p,y,x] * ((dmin + p) - disparity[b,0,y,x])^2 */
int ss_disparity_variance_fwd(const float* prob, const float* disparity, float* out,
                              int B, int dmin, int ndisp, int H, int W, ss_stream_t stream);
/* Fused softmax over D + regression + variance of models/SemStereo.py:281-285:
 * logits [B,ndisp,H,W] -> prob (nullable) [B,ndisp,H,W], disp [B,H,W], var [B,1,H,W] */
int ss_softmax_regression_fwd(const float* logits, float* prob, float* disp, float* var,
                              int B, int dmin, int ndisp, int H, int W, ss_stream_t stream);
/* The same with the trilinear up-sampling in front of it fused in (models/SemStereo.py:279-285):
 * coarse [B,1,ndisp/2,H/2,W/2] = classif_att_'s output; up [B,ndisp,H,W] receives
 * F.interpolate(coarse, [ndisp,H,W], mode='trilinear') (align_corners=False, exact 2x in every dimension: H, W, ndisp
 * even), disp [B,H,W] and var [B,1,H,W] the soft-argmax and variance of softmax(up) over the disparity axis.  ndisp <= 128. */
int ss_upsample_softmax_regression_fwd(const float* coarse, float* up, float* disp, float* var,
                                       int B, int dmin, int ndisp, int H, int W, ss_stream_t stream);

/* Fused models/SemStereo.py:286-293 (variance gate, Propagation x2 [models/submodule.py:290-307],
 * 5-sample SpatialTransformer_grid, channel-mean correlation, softmax over the 5 samples):
 *   strength[b,t,y,x] = softmax_t( mean_c left[b,c,y,x] * warp(right, pred0[b,nb_t(y,x)])[c]
 *                                  * sigmoid(beta + gamma * var[b,0,nb_t(y,x)]) )
 * with nb_t the five replicate-padded diagonal neighbours.  left,right [B,C,H,W]; pred0 [B,H,W];
 * var [B,1,H,W]; gamma, beta: 1-element device arrays -> strength [B,5,H,W] */
int ss_sample_strength_fwd(const float* left, const float* right, const float* pred0, const float* var,
                           const float* gamma, const float* beta, float* strength,
                           int B, int C, int H, int W, ss_stream_t stream);
/* Fused models/SemStereo.py:295-310 (Propagation_prob [models/submodule.py:361-377] weighted by
 * strength, softmax over D, descending stable sort, top-k, ascending re-sort, gathers, softmax over
 * the k, expectation):  logits [B,1,ndisp,H,W], strength [B,5,H,W] ->
 *   samples [B,k,H,W] (candidate disparities dmin + index, ascending, as floats), att_topk [B,1,k,H,W] (their
 *   probabilities), pred_att [B,H,W].  ndisp <= 128. */
int ss_topk_candidates_fwd(const float* logits, const float* strength, float* samples, float* att_topk,
                           float* pred_att, int B, int dmin, int ndisp, int H, int W, int k, ss_stream_t stream);

/* regression_topk(cost, disparity_samples, k)         models/submodule.py:434-442
 * per pixel: the k largest costs (ties: lower index first), softmax over them, expectation of
 * the matching candidates.  cost, samples [B,nd,H,W] -> [B,1,H,W].  1 <= k <= min(nd, 32). */
int ss_regression_topk_fwd(const float* cost, const float* samples, float* out,
                           int B, int nd, int H, int W, int k, ss_stream_t stream);
/* its backward: grad_out [B,1,H,W] -> grad_cost, grad_samples [B,nd,H,W] (fully written; zero outside the k selected
 * candidates; no gradient through the selection itself, as autograd of the reference's sort/gather composition gives). */
int ss_regression_topk_bwd(const float* grad_out, const float* cost, const float* samples, float* grad_cost,
                           float* grad_samples, int B, int nd, int H, int W, int k, ss_stream_t stream);

/* SSR_upsample.forward(depth_low, weights, pred_label)   models/submodule.py:412-431 (calls: SemStereo.py:311, 324)
 * 4x bilinear up-sampling of the 1/4-scale disparity + class-probability-gated residual, one kernel.
 *   depth_low [B,1,h,w]; weights (spx_pred), pred_label [B,n,4h,4w]; out [B,4h,4w]; n = 6.
 *   params: ss_ssr_param_count() floats (153 since ABI 12) = the module's parameters with every eval-mode BatchNorm2d
 *   FOLDED into the convolution before it and the two gate stages pre-multiplied by -log2(e), packed as laid out at the top of
 *   csrc/ssr_upsample.hip (semstereo_amd/modules.py: SSR_upsample._params packs it, in float64). */
int ss_ssr_upsample_fwd(const float* depth_low, const float* weights, const float* pred_label,
                        const float* params, float* out, int B, int h, int w, int num_classes,
                        ss_stream_t stream);
int ss_ssr_param_count(void);

/* channelAtt gating (models/SemStereo.py:101-102): out[b,c,d,y,x] = sigmoid(att[b,c,y,x]) * cv[b,c,d,y,x] */
int ss_channel_gate_fwd(const float* att_logits, const float* cv, float* out,
                        int B, int C, int D, int H, int W, ss_stream_t stream);

/* channelAtt.im_att (models/SemStereo.py:89-100; BasicConv, models/submodule.py:89-116) in one launch:
 *   out[b,cv,y,x] = W2[cv,:] . relu(scale1 * (W1 . im[b,:,y,x]) + shift1) + bias2[cv]     (sigmoid of it when `sigmoid`)
 * im [B,Cin,H,W]; w1_split / w2_split: W1 [Cmid,Cin], W2 [Cout,Cmid] packed by ss_pack_pointwise_weights_bf16s;
 * scale1 / shift1: the eval-mode BatchNorm2d folded to an affine (NULL: identity); bias2 may be NULL.
 * Built for the reference's two gates: (Cin, Cmid, Cout) = (256, 128, 32) and (128, 64, 32). */
int ss_channel_att_logits_fwd(const float* im, const void* w1_split, const float* scale1, const float* shift1,
                              const void* w2_split, const float* bias2, float* out,
                              int B, int Cin, int Cmid, int Cout, int H, int W, int sigmoid, ss_stream_t stream);

/* ---- 3-D aggregation stack -------------------------------------------------------------
 * Conv3d(bias=False) [+ BatchNorm3d in eval] [+ residual] [+ ReLU]: convbn_3d
 * (models/submodule_other.py:845-848), BasicConv(is_3d) (models/submodule.py:89-116), the
 * classifier heads (models/SemStereo.py:228-234).
 *   in      [B,Cin,D,H,W]
 *   wpack   weights re-laid out as [Cin][kd*kh*kw][Cout] (see ss_pack_conv3d_weights)
 *   scale, shift   per-Cout affine applied to the accumulator (folded BN; NULL = 1 / 0)
 *   residual       [B,Cout,Do,Ho,Wo] added after the affine (NULL = none)
 *   relu           != 0 -> max(.,0)
 *   gate           [B,Cout,Ho,Wo] channelAtt gate, already sigmoid-activated (models/SemStereo.py:101-102):
 *                  the result is multiplied by it, broadcast over D, last (NULL = none)
 *   kernel k in {1,3} (cubic), stride in {1,2}, pad = k/2.  out [B,Cout,Do,Ho,Wo],
 *   Do = (D + 2*pad - k)/stride + 1 ... */
int ss_conv3d_fwd(const float* in, const float* wpack, const float* scale, const float* shift,
                  const float* residual, const float* gate, float* out,
                  int B, int Cin, int D, int H, int W, int Cout, int k, int stride, int relu,
                  ss_stream_t stream);
/* Same contract as ss_conv3d_fwd for k = 3, stride 1, computed on the bf16 matrix core with every fp32
 * operand split exactly into three bf16 terms ("split-bf16"): nterms = 6 keeps all cross terms down to
 * 2^-24 (measured error below the exact-fp32 MFMA's, tools/exp_split_bf16.hip), nterms = 3 keeps
 * hi*hi + hi*mid + mid*hi.  wsplit comes from ss_pack_conv3d_weights_bf16s (16-byte aligned).
 * stride in {1, 2}. */
int ss_conv3d_bf16s_fwd(const float* in, const void* wsplit, const float* scale, const float* shift,
                        const float* residual, const float* gate, float* out,
                        int B, int Cin, int D, int H, int W, int Cout, int stride, int relu, int nterms,
                        ss_stream_t stream);
/* The same convolution continuing a PARTIAL SUM: out = gate * relu?(scale * (partial + conv(in)) + shift), partial
 * [B,Cout,D,H,W] = the contribution of input channels that are not in `in` (ss_stem_left_fused_fwd: the broadcast
 * half of concat_stem's input).  It initialises the accumulators, read under the first chunk's staging. */
int ss_conv3d_bf16s_partial_fwd(const float* in, const void* wsplit, const float* partial, const float* scale,
                                const float* shift, const float* gate, float* out,
                                int B, int Cin, int D, int H, int W, int Cout, int relu, int nterms,
                                ss_stream_t stream);
/* concat_stem with the warped half of the sparse concat volume formed INSIDE the convolution's staging (round 5; SURVEY.md
 * section 8 f1, second half).  Replaces, in one launch and without a [B,Cin,nd,H,W] volume,
 *   models/SemStereo.py:241-244 (concat_volume_generator: SpatialTransformer_grid, models/submodule.py:265-288),
 *   :318 (att_topk * volume) on the warped half, :319 (concat_stem: Conv3d k3 s1 p1 + BatchNorm + ReLU over THESE Cin channels,
 *   continuing `partial`) and :320 (the channelAtt gate):
 *   out[b,co,j,h,w] = gate[b,co,h,w] * relu?(scale[co] * (partial[b,co,j,h,w] + sum_{ci,taps} w * x[b,ci,j',h',w']) + shift[co]),
 *   x[b,ci,j,h,w] = att[b,j,h,w] * right[b,ci,h,w - cand[b,j,h,w]]  (0 where that column lies outside [0,W)).
 * right [B,Cin,H,W]; cand, att [B,nd,H,W]; partial [B,Cout,nd,H,W] / scale / shift [Cout] / gate [B,Cout,H,W] may be NULL.
 * CONTRACT: the candidates are INTEGER-valued floats -- what models/SemStereo.py:299-305 produces (indices - maxdisp/4) --
 * for which the reference's bilinear F.grid_sample is a gather up to its own coordinate rounding (<= 1e-5 relative on the
 * columns / rows whose normalise -> unnormalise round trip is inexact; the gather is the float64-exact value).  Fractional
 * candidates are truncated toward zero: use ss_concat_sampled_fwd + ss_conv3d_bf16s_partial_fwd for those.
 * Cin % 8 == 0, Cin >= 16; nterms = 19 only (wsplit from ss_pack_conv3d_weights_f16s); SS_ERR_UNSUPPORTED otherwise. */
int ss_conv3d_gather_fwd(const float* right, const float* cand, const float* att, const void* wsplit, const float* partial,
                         const float* scale, const float* shift, const float* gate, float* out,
                         int B, int Cin, int nd, int H, int W, int Cout, int relu, int nterms, ss_stream_t stream);
/* concat_stem on the pre-split warped half (models/SemStereo.py:319-320): Conv3d(k3, s1, p1, bias=False) over
 * xs / xexp of ss_concat_sampled_presplit_fwd (Cin % 8 == 0), out = gate * relu?(scale * (partial + conv) + shift);
 * wsplit from ss_pack_conv3d_weights_f16s; partial [B,Cout,D,H,W], scale / shift [Cout], gate [B,Cout,H,W] may be NULL.
 * Same results contract as ss_conv3d_bf16s_partial_fwd with nterms = 19 (the block exponent is per batch element
 * instead of per staged tile). */
int ss_conv3d_presplit_fwd(const void* xs, const int* xexp, const void* wsplit, const float* partial, const float* scale,
                           const float* shift, const float* gate, float* out, int B, int Cin, int D, int H, int W,
                           int Cout, int relu, ss_stream_t stream);
/* Conv3d weight [Cout,Cin,3,3,3] fp32 -> split/packed bf16 fragments
 * [ceil(Cin/8)][14 tap pairs][3 terms][2 halves][Cout][8] (ceil(Cin/8)*14*3*2*Cout*16 bytes). */
int ss_pack_conv3d_weights_bf16s(const float* w, void* wsplit, int Cout, int Cin, ss_stream_t stream);
/* The two-term fp16 form of the same layers (same reference lines): ss_conv3d_bf16s_fwd / _partial_fwd with
 * nterms = 19 ("f16x3": hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16, half the matrix-core time of nterms = 6).
 * fp16 has 5 exponent bits, so the operands are block floating point: this packer scales each output channel's weights
 * by a power of two (stored behind the terms and undone in the epilogue), the kernel scales every staged 8-channel chunk
 * of its halo tile by a power of two taken from the tile's running |max| and rescales its fp32 accumulators when it
 * changes.  Error vs exact: representation <= 2^-23 per operand + the dropped lo*lo <= 2^-22 per product, i.e. below
 * the fp32 accumulation error of any K >= 64 (measured in DESIGN.md section 4).  wsplit:
 * [ceil(Cin/8)][14][2 terms][2 halves][Cout][8] fp16 + float[Cout]  (ceil(Cin/8)*14*2*2*Cout*16 + 4*Cout bytes). */
int ss_pack_conv3d_weights_f16s(const float* w, void* wsplit, int Cout, int Cin, ss_stream_t stream);
/* The 2-D form of the split-bf16 conv: Conv2d(k3, s1, p1, bias=False) + affine (+residual) + ReLU on [B,Cin,H,W] maps
 * (concat_feature, models/SemStereo.py:222-226): out [B,Cout,H,W]; w [Cout,Cin,3,3] -> wsplit
 * [ceil(Cin/8)][5 tap pairs][3 terms][2 halves][Cout][8] (ceil(Cin/8)*5*3*2*Cout*16 bytes). */
int ss_conv2d_bf16s_fwd(const float* in, const void* wsplit, const float* scale, const float* shift,
                        const float* residual, float* out, int B, int Cin, int H, int W, int Cout, int relu,
                        int nterms, ss_stream_t stream);
/* The same layer on TWO input tensors in one launch: out [2B,Cout,H,W], elements 0..B-1 from in_a, B..2B-1 from in_b (both
 * [B,Cin,H,W]) -- `concat_feature` applied to the left and to the right view (models/SemStereo.py:314-315) without a torch.cat
 * in front and with twice the workgroups per launch (round 5).  No residual operand. */
int ss_conv2d_bf16s_pair_fwd(const float* in_a, const float* in_b, const void* wsplit, const float* scale, const float* shift,
                             float* out, int B, int Cin, int H, int W, int Cout, int relu, int nterms, ss_stream_t stream);
int ss_pack_conv2d_weights_bf16s(const float* w, void* wsplit, int Cout, int Cin, ss_stream_t stream);
/* two-term fp16 form of the 2-D weights (nterms = 19 of ss_conv2d_bf16s_fwd; see ss_pack_conv3d_weights_f16s):
 * ceil(Cin/8)*5*2*2*Cout*16 + 4*Cout bytes. */
int ss_pack_conv2d_weights_f16s(const float* w, void* wsplit, int Cout, int Cin, ss_stream_t stream);
/* The single-output-channel classifier heads, nn.Conv3d(C, 1, 3, padding=1, bias=False)
 * (models/SemStereo.py:228-234, classif.2 / classif_att_.2), on the split-bf16 engine with the 27 taps as
 * the matrix rows:  out [B,1,D,H,W] = relu?(scale[0] * conv(in [B,Cin,D,H,W]) + shift[0]);
 * Cin in {16, 32, 64}; nterms 6 / 3: wsplit from ss_pack_conv3d_head_weights_bf16s ((Cin/16)*3*2*32*16 bytes); nterms 19: from
 * ss_pack_conv3d_head_weights_f16s. */
int ss_conv3d_head_bf16s_fwd(const float* in, const void* wsplit, const float* scale, const float* shift,
                             float* out, int B, int Cin, int D, int H, int W, int relu, int nterms,
                             ss_stream_t stream);
int ss_pack_conv3d_head_weights_bf16s(const float* w, void* wsplit, int Cin, ss_stream_t stream);
/* nterms = 19 of the two head entry points: two fp16 terms, three products, block floating point (one power-of-two scale
 * for the weights, one per input row from a wave-wide maximum) -- half the matrix work of nterms = 6 at the same accuracy
 * class.  wsplit: (Cin/16)*2*2*32*16 bytes of terms + one float (2^-scale). */
int ss_pack_conv3d_head_weights_f16s(const float* w, void* wsplit, int Cin, ss_stream_t stream);
/* The two layers of a classifier (nn.Sequential(convbn_3d(32,32,3,1,1), ReLU, Conv3d(32,1,3,p1)), models/SemStereo.py:228-234)
 * hand their intermediate over CHANNELS-LAST, [B][D][H][W][C]: it is private to the Sequential, and with a position's
 * channels contiguous the first layer stores 16 bytes per lane and instruction and the head loads 16 (the head spends half
 * of its time issuing 4-byte loads otherwise).  ss_conv3d_bf16s_cl_fwd = ss_conv3d_bf16s_fwd (stride 1, no residual, no
 * gate, Cout % 8 == 0) with that output layout; ss_conv3d_head_bf16s_cl_fwd = ss_conv3d_head_bf16s_fwd reading it
 * (Cin = 32, `in` 16-byte aligned).  Same arithmetic, same results as the plain-layout pair. */
int ss_conv3d_bf16s_cl_fwd(const float* in, const void* wsplit, const float* scale, const float* shift, float* out,
                           int B, int Cin, int D, int H, int W, int Cout, int relu, int nterms, ss_stream_t stream);
int ss_conv3d_head_bf16s_cl_fwd(const float* in, const void* wsplit, const float* scale, const float* shift,
                                float* out, int B, int Cin, int D, int H, int W, int relu, int nterms,
                                ss_stream_t stream);
/* The same classifier in ONE pass over the volume (r05): the 32-channel intermediate never leaves the CU.  The first layer's
 * tile (4 planes x 4 rows x 32 columns x 32 channels) contracts its ReLU(BN(conv)) with the head's 27 x 32 weights on the matrix
 * core as soon as its K loop ends, sums the 27 shifted tap images over the tile in a fixed order and writes the tile's
 * contribution to the 6 x 6 x 34 output positions it touches -- a PATCH -- into `patches`
 * ([B][ceil(W/32) * ceil(H/4) * D/4][6*6*34] floats, caller-allocated scratch); a second small launch adds the <= 8 patches that
 * hold an output position, again in a fixed order: deterministic, and a pair gets the same bits at every batch size.
 *   head_w: ss_pack_classifier_head_weights(w2 [1,32,3,3,3]) -> 6144 bytes, 16-byte aligned.
 * Cin % 8 == 0, 32 intermediate channels, nterms = 19, D % 4 == 0 and a layer of at least 512 (2 x 8 x 32) tiles per pair;
 * anything else returns SS_ERR_UNSUPPORTED (the caller keeps the two-launch form above).  y is the two-launch form's y bit for
 * bit; the head's 864 products per output are summed in another order (within fp32 rounding of it). */
int ss_pack_classifier_head_weights(const float* w2, void* out, ss_stream_t stream);
int ss_conv3d_classifier_fused_fwd(const float* in, const void* wsplit, const float* scale, const float* shift,
                                   const void* head_w, float* patches, float* out, int B, int Cin, int D, int H, int W,
                                   int nterms, ss_stream_t stream);
/* regression_topk(cost.squeeze(1), samples, 2) (models/SemStereo.py:322-323, models/submodule.py:434-442) reading the patches of
 * ss_conv3d_classifier_fused_fwd directly (call that with out = NULL: no patch-sum launch, no [B,1,nd,H,W] cost tensor): patches
 * [B][tiles][1224], samples [B,nd,H,W] -> out [B,1,H,W], bit-identical to the two-launch form.  nd = 24 and k = 2 (the model's);
 * anything else returns SS_ERR_UNSUPPORTED. */
int ss_regression_topk_patched_fwd(const float* patches, const float* samples, float* out, int B, int nd, int H, int W, int k,
                                   ss_stream_t stream);
/* 1x1x1 Conv3d / per-position Linear (+ per-channel affine: bias; ReLU) on the split-bf16 engine:
 * qkv_3d and final1x1 of attention_block (models/submodule_other.py:804, 835).
 *   out [B,Cout,npos] = relu?(scale * (W in) + shift), in [B,Cin,npos], npos = D*H*W, W [Cout,Cin];
 * Cin in {32, 64, 128}; wsplit from ss_pack_pointwise_weights_bf16s
 * (ceil(Cout/32)*(Cin/16)*3*2*32*16 bytes). */
int ss_conv3d_pointwise_bf16s_fwd(const float* in, const void* wsplit, const float* scale, const float* shift,
                                  float* out, int B, int Cin, int Cout, long long npos, int relu, int nterms,
                                  ss_stream_t stream);
int ss_pack_pointwise_weights_bf16s(const float* w, void* wsplit, int Cout, int Cin, ss_stream_t stream);

/* ConvTranspose3d(k3, s2, p1, output_padding 1, bias=False) [+BN] fused with the 1x1x1 skip
 * projection of hourglass.forward (models/SemStereo.py:141-142):
 *   out = relu?( scale*deconv(in) + shift + skip_scale*conv1x1(skip) + skip_shift )
 *   in [B,Cin,D,H,W] -> out [B,Cout,2D,2H,2W]; wpack [Cin][27][Cout] (tap = kd*9+kh*3+kw of
 *   the ConvTranspose3d weight [Cin,Cout,3,3,3]); skip [B,Cs,2D,2H,2W] with skip_wpack
 *   [Cs][Cout] (NULL skip = plain deconv+affine). */
int ss_deconv3d_fwd(const float* in, const float* wpack, const float* scale, const float* shift,
                    const float* skip, const float* skip_wpack, const float* skip_scale, const float* skip_shift,
                    float* out, int B, int Cin, int D, int H, int W, int Cout, int Cs, int relu,
                    ss_stream_t stream);
/* The same layer on the split-bf16 engine (deconv3d_bf16s.hip): wsplit / skip_wsplit come from
 * ss_pack_deconv3d_weights_bf16s applied to the fp32 packs of ss_deconv3d_fwd (wpack [Cin][27][Cout] with the
 * BatchNorm scale folded, ntaps = 27; skip_wpack [Cs][Cout], ntaps = 1); only a common shift is applied.
 * wsplit: ceil(Cin/16)*ntaps*3*2*Cout*16 bytes. */
int ss_deconv3d_bf16s_fwd(const float* in, const void* wsplit, const float* shift, const float* skip,
                          const void* skip_wsplit, float* out, int B, int Cin, int D, int H, int W, int Cout,
                          int Cs, int relu, int nterms, ss_stream_t stream);
int ss_pack_deconv3d_weights_bf16s(const float* wpack, void* wsplit, int Cin, int Cout, int ntaps, ss_stream_t stream);
/* Two-term fp16 form of the MAIN weights (ntaps = 27) for nterms = 19 of ss_deconv3d_bf16s_fwd -- the block-floating
 * scheme described at ss_pack_conv3d_weights_f16s; the skip projection keeps its bf16 pack.
 * wsplit: ceil(Cin/16)*27*2*2*Cout*16 + 4*Cout bytes. */
int ss_pack_deconv3d_weights_f16s(const float* wpack, void* wsplit, int Cin, int Cout, ss_stream_t stream);
/* Weight re-layout helpers (device to device): Conv3d weight [Cout,Cin,k,k,k] or
 * ConvTranspose3d weight [Cin,Cout,k,k,k] (transposed != 0) -> [Cin][k^3][Cout]. */
int ss_pack_conv3d_weights(const float* w, float* wpack, int Cout, int Cin, int k, int transposed,
                           ss_stream_t stream);
/* Weight gradient of Conv3d(k3, padding 1, stride 1 | 2, bias=False) -- the training path, main_us3d.py:186-222:
 *   grad_w[co,ci,kd,kh,kw] = sum_{b,od,oh,ow} grad_out[b,co,od,oh,ow] * in[b,ci,od*s+kd-1,oh*s+kh-1,ow*s+kw-1]
 * grad_out [B,Cout,Do,Ho,Wo], in [B,Cin,D,H,W] -> grad_w [Cout,Cin,3,3,3] (zeroed here, accumulated with fp32 atomics).
 * Also the weight gradient of ConvTranspose3d(k3,s2,p1,op1): call it with grad_out := the layer's input, in := the
 * gradient of its output, stride 2; the result is in that weight's [Cin,Cout,3,3,3] layout.  (The data gradients are the
 * forward entry points: ss_conv3d_bf16s_fwd on flipped weights / ss_deconv3d_fwd / the stride-2 convolution.) */
int ss_conv3d_wgrad_fwd(const float* grad_out, const float* in, float* grad_w, int B, int Cin, int D, int H, int W,
                        int Cout, int stride, ss_stream_t stream);
/* The same weight gradient on the bf16 matrix core (r06; conv3d_wgrad_bf16s.hip): both operands split exactly into three bf16
 * terms, six cross products on v_mfma_f32_32x32x16_bf16, fp32 accumulation (error below the exact-fp32 MFMA's, any magnitude
 * of gradient: bf16 has fp32's exponent range).  `workspace`: ceil(Cout/32) * ceil(Cin/32) * 27 * 1024 floats of scratch
 * (zeroed here; the waves' accumulator tiles are summed into it with coalesced fp32 atomics and re-ordered into grad_w, which is
 * written, not accumulated).  Same arguments and the same ConvTranspose3d convention as ss_conv3d_wgrad_fwd. */
int ss_conv3d_wgrad_bf16s_fwd(const float* grad_out, const float* in, float* grad_w, float* workspace, int B, int Cin, int D, int H,
                              int W, int Cout, int stride, ss_stream_t stream);

/* `patch` (models/SemStereo.py:219, 274): depthwise Conv3d kernel (1,3,3), pad (0,1,1), no bias,
 * optionally fused with the channelAtt gate that follows it (:276):
 *   out[b,c,d] = sigmoid?(gate[b,c]) * conv2d_3x3(in[b,c,d], w[c])       w [C,1,1,3,3], gate [B,C,H,W] or NULL */
int ss_depthwise_patch_fwd(const float* in, const float* w, const float* gate, float* out,
                           int B, int C, int D, int H, int W, ss_stream_t stream);
/* attention_block.forward (models/submodule_other.py:790-837): windowed multi-head
 * self-attention over (bd,bh,bw) windows + 1x1x1 conv with bias.
 *   x [B,C,D,H,W] (C = heads * 8), wqkv_t [C][3C] (= qkv_3d.weight transposed), bqkv [3C],
 *   wout_t [C][C] (= final1x1.weight[:, :, 0,0,0] transposed), bout [C] -> out [B,C,D,H,W].
 *   D % bd == 0; H, W are padded virtually to window multiples with the reference's masking. */
int ss_window_attention_fwd(const float* x, const float* wqkv_t, const float* bqkv,
                            const float* wout_t, const float* bout, float* out,
                            int B, int C, int D, int H, int W, int heads, int bd, int bh, int bw,
                            ss_stream_t stream);
/* The attention core alone, for the three-launch form of the same block: qkv [B,3C,D,H,W] is the
 * qkv_3d projection (+bias) of the UNPADDED volume (a 1x1x1 ss_conv3d_fwd), bqkv [3C] the value a
 * zero-padded token projects to; y [B,C,D,H,W] = softmax(q k^T / sqrt(hd) [+ pad mask]) v per window and
 * head, un-partitioned and cropped (models/submodule_other.py:805-834); final1x1 is another
 * ss_conv3d_fwd.  One workgroup per (window, 4 heads). */
int ss_window_attention_core_fwd(const float* qkv, const float* bqkv, float* y, int B, int C, int D, int H,
                                 int W, int heads, int bd, int bh, int bw, ss_stream_t stream);

/* ---- training side of the 3-D stack (main_us3d.py:186-222): what the matrix-core forward / dgrad / wgrad kernels leave over ---- */
/* BatchNorm with BATCH statistics + optional ReLU (convbn_3d, models/submodule_other.py:845-848; BasicConv,
 * models/submodule.py:89-116, in train()): x [B,C,N] (N = D*H*W or H*W), weight / bias [C] or NULL -> y, and mean / invstd /
 * var_unbiased [C] (for the backward and the running statistics).  work: 2*C doubles of scratch. */
int ss_batchnorm_train_fwd(const float* x, const float* weight, const float* bias, float* y, float* mean, float* invstd,
                           float* var_unbiased, double* work, int B, int C, long long N, float eps, int relu,
                           ss_stream_t stream);
/* ... backward: grad_x [B,C,N]; grad_bias[c] = work[2c], grad_weight[c] = work[2c+1] (doubles).  y is read only when relu. */
int ss_batchnorm_train_bwd(const float* grad_y, const float* x, const float* y, const float* mean, const float* invstd,
                           const float* weight, float* grad_x, double* work, int B, int C, long long N, int relu,
                           ss_stream_t stream);
/* The same pair with a residual joining behind the normalisation and before the ReLU: y = relu(bn(x) + residual) -- the
 * `F.relu(self.conv5(conv4) + self.redir2(conv2))` of hourglass.forward in train() (models/SemStereo.py:141-142) in one pass;
 * backward: also grad_residual [B,C,N] = grad_y behind the ReLU mask. */
int ss_batchnorm_train_res_fwd(const float* x, const float* residual, const float* weight, const float* bias, float* y, float* mean,
                               float* invstd, float* var_unbiased, double* work, int B, int C, long long N, float eps, int relu,
                               ss_stream_t stream);
int ss_batchnorm_train_res_bwd(const float* grad_y, const float* x, const float* y, const float* mean, const float* invstd,
                               const float* weight, float* grad_x, float* grad_residual, double* work, int B, int C, long long N,
                               int relu, ss_stream_t stream);
/* BatchNorm on its RUNNING statistics under autograd (a module in eval() whose inputs / parameters require gradients; the reference's
 * training loop never freezes BatchNorm, main_us3d.py:186-222, PyTorch allows it): forward with the caller's per-channel mean and
 * invstd = 1 / sqrt(running_var + eps), optional residual (joins before the ReLU) and ReLU; backward: grad_x = w * invstd * g'
 * (g' = grad_y behind the ReLU mask), grad_residual = g' when not NULL, grad_bias[c] = work[2c], grad_weight[c] = work[2c + 1]. */
int ss_batchnorm_eval_fwd(const float* x, const float* residual, const float* mean, const float* invstd, const float* weight,
                          const float* bias, float* y, int B, int C, long long N, int relu, ss_stream_t stream);
int ss_batchnorm_eval_bwd(const float* grad_y, const float* x, const float* y, const float* mean, const float* invstd,
                          const float* weight, float* grad_x, float* grad_residual, double* work, int B, int C, long long N,
                          int relu, ss_stream_t stream);
/* (r06) The batch-statistics forward, either form (residual may be NULL), that also does nn.BatchNorm's bookkeeping in its statistics
 * kernel: running_mean / running_var [C] (NULL = leave alone) become running * (1 - momentum) + momentum * batch value (the variance
 * unbiased) as F.batch_norm moves them (torch/nn/modules/batchnorm.py: the reference's layers keep the default momentum 0.1), and
 * num_batches_tracked (int64[1] or NULL) counts the batch.  0 <= momentum <= 1; a module with momentum=None (cumulative average) is not
 * this entry's case. */
int ss_batchnorm_train_fwd_rs(const float* x, const float* residual, const float* weight, const float* bias, float* y, float* mean,
                              float* invstd, float* var_unbiased, double* work, float* running_mean, float* running_var,
                              long long* num_batches_tracked, double momentum, int B, int C, long long N, float eps, int relu,
                              ss_stream_t stream);
/* (r06) Both backward forms with the parameter gradients also as floats, grad_weight / grad_bias [C] (either may be NULL; the doubles
 * stay in `work`); grad_residual may be NULL; batch_statistics = 1: the backward of ss_batchnorm_train_fwd / _res_fwd / _fwd_rs,
 * 0: of ss_batchnorm_eval_fwd.  `bias`: the forward's (NULL = none).  y may be NULL where the forward had a ReLU and no residual: the
 * mask is recomputed from x with the forward's own expression, and y is not read (two of the backward's seven tensor passes). */
int ss_batchnorm_bwd_pg(const float* grad_y, const float* x, const float* y, const float* mean, const float* invstd,
                        const float* weight, const float* bias, float* grad_x, float* grad_residual, double* work, float* grad_weight,
                        float* grad_bias, int batch_statistics, int B, int C, long long N, int relu, ss_stream_t stream);
/* (r06, training) The normalisation of groupwise_correlation_norm (models/submodule.py:213-222), once per feature map:
 * y = x / (||x||_2 over each group's C / groups channels + eps), x [B,C,H,W]; and its backward (grad_x from grad_y and x).  The
 * volume is then built from the normalised maps by ss_gwc_volume_fwd / _bwd. */
int ss_group_normalise_fwd(const float* x, float* y, int B, int C, int H, int W, int groups, float eps, ss_stream_t stream);
int ss_group_normalise_bwd(const float* grad_y, const float* x, float* grad_x, int B, int C, int H, int W, int groups, float eps,
                           ss_stream_t stream);
/* Weight gradient of the 1x1(x1) convolutions (redir1 / redir2 `models/SemStereo.py:131-132`, attention_block.qkv_3d /
 * final1x1 `models/submodule_other.py:799-800`, channelAtt.im_att `models/SemStereo.py:92-95`):
 * grad_out [B,Cout,npos], in [B,Cin,npos] -> grad_w [Cout,Cin]. */
int ss_conv_k1_wgrad_fwd(const float* grad_out, const float* in, float* grad_w, int B, int Cin, int Cout, long long npos,
                         ss_stream_t stream);
/* sums[c] (double) = sum over batch and positions of a[b,c,:]: bias gradients of the 1x1x1 projections of attention_block
 * (models/submodule_other.py:799-800). */
int ss_channel_sum_fwd(const float* a, double* sums, int B, int C, long long N, ss_stream_t stream);
/* Weight gradient of `patch` (depthwise Conv3d (1,3,3), models/SemStereo.py:219): grad_w [C,1,1,3,3]. */
int ss_depthwise_patch_wgrad_fwd(const float* grad_out, const float* in, float* grad_w, int B, int C, int D, int H, int W,
                                 ss_stream_t stream);
/* channelAtt's gate (models/SemStereo.py:101-102), gradient of the logits:
 * grad_att [B,C,H,W] = s (1 - s) * sum_d grad_out[b,c,d] * cv[b,c,d], s = sigmoid(att_logits).  (The gradient of cv is
 * ss_channel_gate_fwd applied to grad_out.) */
int ss_channel_gate_bwd_logits(const float* grad_out, const float* cv, const float* att_logits, float* grad_att, int B, int C,
                               int D, int H, int W, ss_stream_t stream);
/* Backward of ss_window_attention_core_fwd (models/submodule_other.py:805-834) for volumes whose H, W are multiples of
 * the window (no pad tokens): qkv [B,3C,D,H,W], grad_y [B,C,D,H,W] -> grad_qkv [B,3C,D,H,W]. */
int ss_window_attention_core_bwd(const float* qkv, const float* grad_y, float* grad_qkv, int B, int C, int D, int H, int W,
                                 int heads, int bd, int bh, int bw, ss_stream_t stream);
/* ... for any H, W (r04): the reference zero-pads the volume BEFORE the qkv Linear (models/submodule_other.py:808-813), so a pad
 * token's q / k / v are that Linear's bias `bqkv` [3C] (the qkv tensor holds real positions only), its output is cropped, and a logit
 * between a pad and a real token carries -1000 when BOTH H and W were padded (:822-829).  grad_bias [3C] (zeroed by the call) receives
 * what reaches the pad tokens' q / k / v: the extra term of the Linear's bias gradient. */
int ss_window_attention_core_pad_bwd(const float* qkv, const float* bqkv, const float* grad_y, float* grad_qkv, float* grad_bias, int B,
                                     int C, int D, int H, int W, int heads, int bd, int bh, int bw, ss_stream_t stream);
/* Backward of the three fused attention-tail launches (training; models/SemStereo.py:279-310 under autograd; r04).  Each recomputes
 * its forward quantities from the saved inputs.  Gradient outputs that are scattered into neighbouring pixels (fp32 atomics) are
 * zeroed by the call; any gradient input / output pointer may be NULL unless said otherwise.
 *   ss_upsample_softmax_regression_bwd  :279-285  up [B,1,D,H,W] (the forward's output), grad_up / grad_disp [B,H,W] / grad_var [B,1,H,W]
 *                                                 -> grad_coarse [B,1,D/2,H/2,W/2] (required); work: B*D*H*W floats of scratch
 *   ss_sample_strength_bwd              :286-293  grad_strength [B,5,H,W] (required) -> grad_left / grad_right [B,C,H,W], grad_pred0 [B,H,W],
 *                                                 grad_var [B,1,H,W], grad_gamma_beta [2]
 *   ss_topk_candidates_bwd              :295-310  samples [B,K,H,W] = the forward's selection; grad_att_topk [B,1,K,H,W], grad_pred_att [B,H,W]
 *                                                 -> grad_logits [B,1,D,H,W], grad_strength [B,5,H,W] */
int ss_upsample_softmax_regression_bwd(const float* up, const float* grad_up, const float* grad_disp, const float* grad_var,
                                       float* grad_coarse, float* work, int B, int dmin, int ndisp, int H, int W, ss_stream_t stream);
int ss_sample_strength_bwd(const float* left, const float* right, const float* pred0, const float* var, const float* gamma,
                           const float* beta, const float* grad_strength, float* grad_left, float* grad_right, float* grad_pred0,
                           float* grad_var, float* grad_gamma_beta, int B, int C, int H, int W, ss_stream_t stream);
/* ... through a scratch `work` of B * 5 * H * W floats: two launches, the channels on the second one's grid (r06: 3x faster at 128 channels) */
int ss_sample_strength_bwd_ws(const float* left, const float* right, const float* pred0, const float* var, const float* gamma,
                              const float* beta, const float* grad_strength, float* grad_left, float* grad_right, float* grad_pred0,
                              float* grad_var, float* grad_gamma_beta, float* work, int B, int C, int H, int W, ss_stream_t stream);
int ss_topk_candidates_bwd(const float* logits, const float* strength, const float* samples, const float* grad_att_topk,
                           const float* grad_pred_att, float* grad_logits, float* grad_strength, int B, int dmin, int ndisp, int H, int W,
                           int k, ss_stream_t stream);
/* Measurement aid (bench.py): a plain device copy, 16 bytes per lane, nontemporal -- the HBM rate a streaming kernel can
 * reach on this box, which SURVEY.md section 8(d) asks the bandwidth fractions to be read against.  bytes % 16 == 0. */
int ss_tool_copy_fwd(const void* src, void* dst, long long bytes, ss_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SEMSTEREO_HIP_H */
