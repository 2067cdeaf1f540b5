// The classifiers of the model -- nn.Sequential(convbn_3d(32,32,3,1,1), ReLU, Conv3d(32,1,3,p1)): `classif` and `classif_att_`, reference
// models/SemStereo.py:228-234, called at :278 and :322 -- as ONE pass over the volume: the HEAD form of conv3d_bf16s (see the kernel's
// header comment in conv3d_bf16s.hip), instantiated in its own translation unit, plus the packing of the head's weights and the
// sum of the tiles' patches.
#define SS_CONV_GATHER_TU 1
#include "conv3d_bf16s.hip"

namespace {

// matrix row m of the head's A operand -> tap (kd * 9 + kh * 3 + kw), or -1: a lane half of the MFMA's result holds rows
// m = (r & 3) + 8 * (r >> 2) + 4 * half in register r; register 3 * ps + kh is tap row kh of the (kd, kw) pair ps + 5 * half, so that
// the sum over kh happens inside a lane (conv3d_bf16s.hip, HEAD)
__host__ __device__ inline int head_tap_of_row(int m) {
    const int hf = (m >> 2) & 1, r = (m & 3) + 4 * (m >> 3);
    const int ps = r / 3, kh = r % 3, pair = ps + 5 * hf;
    if (r >= 15 || pair >= 9) return -1;
    return (pair / 3) * 9 + kh * 3 + (pair % 3);
}

// w2 [1,32,3,3,3] fp32 -> [3 bf16 terms][2 K-steps][64 lanes][8]: lane (m, hf) of K-step ks holds channels
// (j & 3) + 4 hf + 8 (j >> 2) + 16 ks, j = 0..7 -- the order in which the conv's accumulators hold them
__global__ void pack_classifier_head_kernel(const float* __restrict__ w2, unsigned short* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= HEAD_WSLOTS * 8) return;
    const int j = i % 8, lane = (i / 8) % 64, ks = (i / 512) % 2, term = i / 1024;
    const int m = lane & 31, hf = lane >> 5;
    const int ch = (j & 3) + 4 * hf + 8 * (j >> 2) + 16 * ks;
    const int tap = head_tap_of_row(m);
    const float x = tap >= 0 ? w2[ch * 27 + tap] : 0.f;
    unsigned h, mm, l;
    split3(x, h, mm, l);
    out[i] = (unsigned short)(term == 0 ? h : (term == 1 ? mm : l));
}

// out[b][d][h][w] = the sum of the (at most 8) tiles' patches that hold that position, in a fixed order (planes, rows, columns
// ascending).  Per dimension a position lies in its own tile's patch and, on a tile's first / last plane, row or column, in the
// previous / next tile's ring: one or two candidates per dimension, 2.4 reads per output on average.
__device__ __forceinline__ int patch_candidates(int x, int ts, int ntile, int (&t)[2], int (&p)[2]) {
    const int t0 = x / ts, r = x - t0 * ts;
    int n = 0;
    if (r == 0 && t0 > 0) { t[n] = t0 - 1; p[n] = ts + 1; ++n; }
    t[n] = t0; p[n] = r + 1; ++n;
    if (r == ts - 1 && t0 + 1 < ntile) { t[n] = t0 + 1; p[n] = 0; ++n; }
    return n;
}
// grid (ceil(W / 256), H, B * D): the plane and row candidates are wave-uniform (scalar arithmetic), a lane owns one column
__global__ __launch_bounds__(256) void classifier_patch_sum_kernel(const float* __restrict__ patches, float* __restrict__ out, int D, int H,
                                                                    int W, int tiles_w, int tiles_h, int tiles_d) {
    const int w = blockIdx.x * 256 + threadIdx.x;
    const int h = blockIdx.y;
    const int d = blockIdx.z % D, b = blockIdx.z / D;
    if (w >= W) return;
    const float* pb = patches + (size_t)b * tiles_w * tiles_h * tiles_d * HEAD_PATCH;
    int td[2], pd[2], th[2], ph[2], tw[2], pw[2];
    const int nd = patch_candidates(d, 4, tiles_d, td, pd), nh = patch_candidates(h, 4, tiles_h, th, ph), nw = patch_candidates(w, 32, tiles_w, tw, pw);
    float v = 0.f;
    for (int a = 0; a < nd; ++a)
        for (int c = 0; c < nh; ++c) {
            const float* row = pb + (size_t)((td[a] * tiles_h + th[c]) * tiles_w) * HEAD_PATCH + (pd[a] * 6 + ph[c]) * 34;
            for (int e = 0; e < nw; ++e) v = ss::add_rn(v, row[tw[e] * HEAD_PATCH + pw[e]]);
        }
    out[(((size_t)b * D + d) * H + h) * W + w] = v;
}

// ---- the patch sum folded into the head's CONSUMER (r06): regression_topk(cost.squeeze(1), samples, 2) of models/SemStereo.py:322-323
// (models/submodule.py:434-442) reading the tiles' patches directly -- the ND costs of a pixel are summed from their <= 8 patches each in
// the order of classifier_patch_sum_kernel (same bits), kept in registers, and the top-2 soft-argmax of topk_regress_kernel<2>
// (regression.hip: value descending, index ascending; the same separately rounded products) follows: one launch and the [B,1,D,H,W]
// cost tensor fewer.  grid (ceil(W / 256), H, B): plane and row candidates are scalar arithmetic, a lane owns one column.
template <int ND>
__global__ __launch_bounds__(256) void topk2_regress_patched_kernel(const float* __restrict__ patches, const float* __restrict__ samples,
                                                                     float* __restrict__ out, int H, int W, int tiles_w, int tiles_h) {
    constexpr int tiles_d = ND / 4;
    const int w = blockIdx.x * 256 + threadIdx.x;
    const int h = blockIdx.y, b = blockIdx.z;
    if (w >= W) return;
    const float* pb = patches + (size_t)b * tiles_w * tiles_h * tiles_d * HEAD_PATCH;
    int th[2], ph[2], tw[2], pw[2];
    const int nh = patch_candidates(h, 4, tiles_h, th, ph), nw = patch_candidates(w, 32, tiles_w, tw, pw);
    float cost[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) {
        int td[2], pd[2];
        const int nd = patch_candidates(d, 4, tiles_d, td, pd);
        float v = 0.f;
        for (int a = 0; a < nd; ++a)
            for (int c = 0; c < nh; ++c) {
                const float* row = pb + (size_t)((td[a] * tiles_h + th[c]) * tiles_w) * HEAD_PATCH + (pd[a] * 6 + ph[c]) * 34;
                for (int e = 0; e < nw; ++e) v = ss::add_rn(v, row[tw[e] * HEAD_PATCH + pw[e]]);
            }
        cost[d] = v;
    }
    // the two largest costs in the order (value descending, index ascending)
    float v0 = cost[0]; int i0 = 0;
#pragma unroll
    for (int d = 1; d < ND; ++d)
        if (cost[d] > v0) { v0 = cost[d]; i0 = d; }
    float v1 = -INFINITY; int i1 = INT_MAX;
#pragma unroll
    for (int d = 0; d < ND; ++d) {
        const float v = cost[d];
        const bool after = (v < v0) || (v == v0 && d > i0);
        const bool beats = (v > v1) || (v == v1 && d < i1);
        if (after && beats) { v1 = v; i1 = d; }
    }
    if (i1 == INT_MAX) i1 = 0;
    const size_t plane = (size_t)H * W, pix = (size_t)h * W + w;
    const float* sp = samples + (size_t)b * ND * plane + pix;
    const float e0 = expf(v0 - v0), e1 = expf(v1 - v0);
    const float sum = ss::add_rn(ss::add_rn(0.f, e0), e1);
    float acc = ss::add_rn(0.f, ss::mul_rn(sp[(size_t)i0 * plane], e0 / sum));
    acc = ss::add_rn(acc, ss::mul_rn(sp[(size_t)i1 * plane], e1 / sum));
    out[(size_t)b * plane + pix] = acc;
}

}  // namespace

extern "C" int ss_pack_classifier_head_weights(const float* w2, void* out, ss_stream_t stream) {
    SS_REQUIRE(w2 && out);
    hipLaunchKernelGGL(pack_classifier_head_kernel, dim3(ss::ceil_div(HEAD_WSLOTS * 8, 256)), dim3(256), 0, ss::as_stream(stream), w2,
                       reinterpret_cast<unsigned short*>(out));
    return ss::check_launch();
}

extern "C" int ss_conv3d_classifier_fused_fwd(const float* in, const void* wsplit, const float* scale, const float* shift,
                                              const void* head_w, float* patches, float* out, int B, int Cin, int D, int H, int W,
                                              int nterms, ss_stream_t stream) {
    SS_REQUIRE(in && wsplit && head_w && patches);          // out == NULL: the patches only (ss_regression_topk_patched_fwd reads them)
    SS_REQUIRE(B > 0 && Cin > 0 && D > 0 && H > 0 && W > 0);
    SS_REQUIRE((reinterpret_cast<uintptr_t>(wsplit) & 15) == 0 && (reinterpret_cast<uintptr_t>(head_w) & 15) == 0);
    if (nterms != F16X3 || D % 4 != 0) return SS_ERR_UNSUPPORTED;
    if ((long long)Cin * D * H * W * 4 >= 0x7fffffffLL) return SS_ERR_UNSUPPORTED;
    // the form of the LAYER, whatever the batch (a pair gets the same bits alone and in a batch): layers too small for the
    // 4-row tile at batch 1 keep the two-launch form
    const long long per_pair = (long long)ss::ceil_div(W, 32) * ss::ceil_div(H, 8) * ss::ceil_div(D, 2);
    if (per_pair < 512) return SS_ERR_UNSUPPORTED;
    if (H > 65535 || (long long)B * D > 65535) return SS_ERR_UNSUPPORTED;     // (the patch sum's grid)
    hipStream_t st = ss::as_stream(stream);
    const int rc = launch_bgm<1, 4, 4, 4, F16X3, false, 1, 3, 1, false, false, true>(
        in, wsplit, scale, shift, nullptr, nullptr, patches, B, Cin, D, H, W, 32, 1, st, reinterpret_cast<const float*>(head_w));
    if (rc != SS_OK || out == nullptr) return rc;
    const int tiles_w = ss::ceil_div(W, 32), tiles_h = ss::ceil_div(H, 4), tiles_d = D / 4;
    hipLaunchKernelGGL(classifier_patch_sum_kernel, dim3(ss::ceil_div(W, 256), H, B * D), dim3(256), 0, st, patches, out, D, H, W, tiles_w,
                       tiles_h, tiles_d);
    return ss::check_launch();
}

// regression_topk(cost, samples, 2) (models/submodule.py:434-442) on the PATCHES of ss_conv3d_classifier_fused_fwd(out = NULL): patches
// [B][tiles][6*6*34] of a [B,1,nd,H,W] cost, samples [B,nd,H,W] -> out [B,1,H,W]; bit-identical to the patch sum followed by
// ss_regression_topk_fwd.  Built for the model's nd = 24 candidates and k = 2 (anything else: SS_ERR_UNSUPPORTED, take the two launches).
extern "C" int ss_regression_topk_patched_fwd(const float* patches, const float* samples, float* out, int B, int nd, int H, int W, int k,
                                              ss_stream_t stream) {
    SS_REQUIRE(patches && samples && out);
    SS_REQUIRE(B > 0 && nd > 0 && H > 0 && W > 0);
    if (nd != 24 || k != 2 || H > 65535 || B > 65535) return SS_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(topk2_regress_patched_kernel<24>, dim3(ss::ceil_div(W, 256), H, B), dim3(256), 0, ss::as_stream(stream), patches, samples,
                       out, H, W, ss::ceil_div(W, 32), ss::ceil_div(H, 4));
    return ss::check_launch();
}
