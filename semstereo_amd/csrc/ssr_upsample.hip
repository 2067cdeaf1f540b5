// Semantic-guided refinement head (SSR_upsample, reference models/submodule.py:412-431; call sites
// models/SemStereo.py:311, 324) as ONE kernel (gfx950):
//
//   label = softmax_c(pred_label);  up = bilinear x4 (align_corners=False) of the 1/4-scale disparity
//   depth = BN(Conv3x3_{1->n}(BN(up)))            (zero padding applies AFTER the first BN)
//   prob  = sigmoid(BN(Conv1x1(label * weights)));  prob = sigmoid(BN(Conv1x1(prob * weights)))
//   out   = up + Conv1x1_{n->1}(depth * prob)
//
// The reference runs ~14 ATen kernels over full-resolution [B,n,H,W] tensors.  Everything is
// per-pixel except the 3x3 conv on the up-sampled disparity, whose 9 taps are re-interpolated from
// the (L2-resident) 1/4-scale map, so one thread per output pixel reads 2n floats and writes one:
// pure HBM streaming, 4*(2n+1)*H*W bytes per pair, lanes along W.
#include <algorithm>

#include "common.h"

namespace {

constexpr int NCLS = 6;
// offsets into the packed parameter block (floats); BN layers are folded to scale/shift on the host
constexpr int P_BN0 = 0;                       // scale, shift
constexpr int P_CW = 2;                        // conv 1->n weights [n][9]
constexpr int P_CB = P_CW + NCLS * 9;          // conv bias [n]
constexpr int P_BNA = P_CB + NCLS;             // scale[n], shift[n]
constexpr int P_W1 = P_BNA + 2 * NCLS;         // conv1 [n][n] (out, in)
constexpr int P_B1 = P_W1 + NCLS * NCLS;
constexpr int P_BN1 = P_B1 + NCLS;
constexpr int P_W2 = P_BN1 + 2 * NCLS;
constexpr int P_B2 = P_W2 + NCLS * NCLS;
constexpr int P_BN2 = P_B2 + NCLS;
constexpr int P_W3 = P_BN2 + 2 * NCLS;         // conv3 [n]
constexpr int P_B3 = P_W3 + NCLS;
constexpr int P_TOTAL = P_B3 + 1;              // 189

// ATen upsample_bilinear2d, align_corners=False, scale = in/out = 0.25
__device__ __forceinline__ void src_index(int dst, int in_size, int& i0, int& i1, float& l0, float& l1) {
    float s = 0.25f * ((float)dst + 0.5f) - 0.5f;
    s = s < 0.f ? 0.f : s;
    i0 = min((int)s, in_size - 1);
    i1 = min(i0 + 1, in_size - 1);
    l1 = fminf(fmaxf(s - (float)i0, 0.f), 1.f);
    l0 = 1.0f - l1;
}

__device__ __forceinline__ float upsampled(const float* __restrict__ low, int h, int w, int Y, int X) {
    int y0, y1, x0, x1;
    float ly0, ly1, lx0, lx1;
    src_index(Y, h, y0, y1, ly0, ly1);
    src_index(X, w, x0, x1, lx0, lx1);
    const float t0 = ss::add_rn(ss::mul_rn(lx0, low[y0 * w + x0]), ss::mul_rn(lx1, low[y0 * w + x1]));
    const float t1 = ss::add_rn(ss::mul_rn(lx0, low[y1 * w + x0]), ss::mul_rn(lx1, low[y1 * w + x1]));
    return ss::add_rn(ss::mul_rn(ly0, t0), ss::mul_rn(ly1, t1));
}

__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ __launch_bounds__(256) void ssr_upsample_kernel(const float* __restrict__ depth_low, const float* __restrict__ weights,
                                                            const float* __restrict__ label, const float* __restrict__ prm,
                                                            float* __restrict__ out, int h, int w, long long total) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int H = 4 * h, W = 4 * w;
    const long long plane = (long long)H * W;
    const int X = (int)(i % W), Y = (int)((i / W) % H);
    const long long b = i / plane;
    const long long pix = (long long)Y * W + X;
    const float* low = depth_low + b * h * w;

    // 3x3 neighbourhood of BN0(up), zero outside the image
    float nb[9], centre = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int yy = Y + ky - 1, xx = X + kx - 1;
            float v = 0.f;
            if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) {
                const float u = upsampled(low, h, w, yy, xx);
                if (ky == 1 && kx == 1) centre = u;
                v = ss::add_rn(ss::mul_rn(u, prm[P_BN0]), prm[P_BN0 + 1]);
            }
            nb[ky * 3 + kx] = v;
        }
    float depth[NCLS];
#pragma unroll
    for (int c = 0; c < NCLS; ++c) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) a = fmaf(prm[P_CW + c * 9 + k], nb[k], a);
        a = ss::add_rn(a, prm[P_CB + c]);
        depth[c] = ss::add_rn(ss::mul_rn(a, prm[P_BNA + c]), prm[P_BNA + NCLS + c]);
    }
    // class probabilities and the two gated 1x1 stages
    float lab[NCLS], wt[NCLS], mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < NCLS; ++c) {
        lab[c] = label[(b * NCLS + c) * plane + pix];
        wt[c] = weights[(b * NCLS + c) * plane + pix];
        mx = fmaxf(mx, lab[c]);
    }
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < NCLS; ++c) { lab[c] = expf(lab[c] - mx); sum = ss::add_rn(sum, lab[c]); }
    float z[NCLS];
#pragma unroll
    for (int c = 0; c < NCLS; ++c) z[c] = ss::mul_rn(lab[c] / sum, wt[c]);
    float p1[NCLS];
#pragma unroll
    for (int o = 0; o < NCLS; ++o) {
        float a = 0.f;
#pragma unroll
        for (int c = 0; c < NCLS; ++c) a = fmaf(prm[P_W1 + o * NCLS + c], z[c], a);
        a = ss::add_rn(a, prm[P_B1 + o]);
        p1[o] = sigmoidf(ss::add_rn(ss::mul_rn(a, prm[P_BN1 + o]), prm[P_BN1 + NCLS + o]));
    }
#pragma unroll
    for (int c = 0; c < NCLS; ++c) z[c] = ss::mul_rn(p1[c], wt[c]);
    float res = 0.f;
#pragma unroll
    for (int o = 0; o < NCLS; ++o) {
        float a = 0.f;
#pragma unroll
        for (int c = 0; c < NCLS; ++c) a = fmaf(prm[P_W2 + o * NCLS + c], z[c], a);
        a = ss::add_rn(a, prm[P_B2 + o]);
        const float p2 = sigmoidf(ss::add_rn(ss::mul_rn(a, prm[P_BN2 + o]), prm[P_BN2 + NCLS + o]));
        res = fmaf(prm[P_W3 + o], ss::mul_rn(depth[o], p2), res);
    }
    out[i] = ss::add_rn(centre, ss::add_rn(res, prm[P_B3]));
}

}  // namespace

extern "C" int ss_ssr_upsample_fwd(const float* depth_low, const float* weights, const float* pred_label,
                                   const float* params, float* out, int B, int h, int w, int num_classes,
                                   ss_stream_t stream) {
    SS_REQUIRE(depth_low && weights && pred_label && params && out);
    SS_REQUIRE(B > 0 && h > 0 && w > 0);
    if (num_classes != NCLS) return SS_ERR_UNSUPPORTED;
    const long long total = (long long)B * 16 * h * w;
    const long long blocks = ss::ceil_div_ll(total, 256);
    if (blocks > 0x7fffffffLL) return SS_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(ssr_upsample_kernel, dim3((unsigned)blocks), dim3(256), 0, ss::as_stream(stream), depth_low, weights,
                       pred_label, params, out, h, w, total);
    return ss::check_launch();
}

extern "C" int ss_ssr_param_count(void) { return P_TOTAL; }
