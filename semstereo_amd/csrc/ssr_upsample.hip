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
// The packed parameter block (floats), built on the host (semstereo_amd/modules.py: SSR_upsample._params) with every eval-mode
// BatchNorm FOLDED into the convolution it follows, in float64, and the two gate stages pre-multiplied by -log2(e) so that their
// sigmoid is 1 / (1 + exp2(a)) on the accumulated value (r05: the head was instruction-bound -- 1450 VALU instructions per 4
// pixels, 124 of them quarter-rate, ~117 us of issue at batch 8 against 67 us of HBM; compensated exponentials, Newton steps and
// separately rounded BatchNorm affines are not what the 2e-5 px contract of the head needs):
constexpr int P_BN0 = 0;                       // scale, shift of the BatchNorm on the up-sampled disparity (zero padding follows it)
constexpr int P_CW = 2;                        // [n][9]  sa[c] * conv(1->n) weights
constexpr int P_CB = P_CW + NCLS * 9;          // [n]     sa[c] * bias + ta[c]
constexpr int P_W1 = P_CB + NCLS;              // [n][n]  -log2e * s1[o] * conv1 (out, in)
constexpr int P_B1 = P_W1 + NCLS * NCLS;       // [n]     -log2e * (s1 * b1 + t1)
constexpr int P_W2 = P_B1 + NCLS;
constexpr int P_B2 = P_W2 + NCLS * NCLS;
constexpr int P_W3 = P_B2 + NCLS;              // conv3 [n]
constexpr int P_B3 = P_W3 + NCLS;
constexpr int P_TOTAL = P_B3 + 1;              // 153

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

// 1 / (1 + 2^a): v_exp_f32 and v_rcp_f32 are 1 ulp each; a -> +inf gives 0, a -> -inf gives 1, as the sigmoid's limits
__device__ __forceinline__ float gate2(float a) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(a)); }

// one pixel: everything after the 3x3 neighbourhood `nb` of BN0(up) and the raw centre value are known
__device__ __forceinline__ float ssr_pixel(const float (&nb)[9], float centre, const float (&labv)[NCLS], const float (&wt)[NCLS],
                                           const float* __restrict__ prm) {
    float depth[NCLS];
#pragma unroll
    for (int c = 0; c < NCLS; ++c) {
        float a = prm[P_CB + c];
#pragma unroll
        for (int k = 0; k < 9; ++k) a = fmaf(prm[P_CW + c * 9 + k], nb[k], a);
        depth[c] = a;
    }
    // class probabilities (soft-max over the 6 logits) times the guidance weights; the 1 / sum joins the first stage's accumulator
    float mx = labv[0];
#pragma unroll
    for (int c = 1; c < NCLS; ++c) mx = fmaxf(mx, labv[c]);
    float z[NCLS], sum = 0.f;
#pragma unroll
    for (int c = 0; c < NCLS; ++c) {
        const float e = __builtin_amdgcn_exp2f((labv[c] - mx) * 1.44269504088896340736f);
        sum += e;
        z[c] = e * wt[c];
    }
    const float rsum = __builtin_amdgcn_rcpf(sum);             // sum in [1, 6]
    float p1[NCLS];
#pragma unroll
    for (int o = 0; o < NCLS; ++o) {
        float a = 0.f;
#pragma unroll
        for (int c = 0; c < NCLS; ++c) a = fmaf(prm[P_W1 + o * NCLS + c], z[c], a);
        p1[o] = gate2(fmaf(a, rsum, prm[P_B1 + o]));
    }
#pragma unroll
    for (int c = 0; c < NCLS; ++c) z[c] = p1[c] * wt[c];
    float res = prm[P_B3];
#pragma unroll
    for (int o = 0; o < NCLS; ++o) {
        float a = prm[P_B2 + o];
#pragma unroll
        for (int c = 0; c < NCLS; ++c) a = fmaf(prm[P_W2 + o * NCLS + c], z[c], a);
        res = fmaf(prm[P_W3 + o] * depth[o], gate2(a), res);
    }
    return centre + res;
}

// Any size: one thread per output pixel, the 9 taps of the 3x3 conv re-interpolated from the 1/4-scale map.
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
    float lab[NCLS], wt[NCLS];
#pragma unroll
    for (int c = 0; c < NCLS; ++c) {
        lab[c] = label[(b * NCLS + c) * plane + pix];
        wt[c] = weights[(b * NCLS + c) * plane + pix];
    }
    out[i] = ssr_pixel(nb, centre, lab, wt, prm);
}

// W % 4 == 0 (always: W = 4 w): a workgroup owns 8 rows x 128 columns.  The up-sampled disparity of the tile and its one-pixel
// halo is interpolated ONCE into LDS (raw, and through the first BatchNorm with the conv's zero padding applied) -- the
// per-pixel form above re-interpolates each value nine times and was bound by that arithmetic, not by HBM -- and every
// thread then finishes 4 consecutive pixels with 16-byte loads of the class logits / guidance weights and a 16-byte store.
constexpr int SST_H = 8, SST_W = 128, SSP = SST_W + 8;          // LDS pitch: halo column at index 3, tile from index 4

__global__ __launch_bounds__(256) void ssr_upsample_tiled(const float* __restrict__ depth_low, const float* __restrict__ weights,
                                                           const float* __restrict__ label, const float* __restrict__ prm,
                                                           float* __restrict__ out, int h, int w) {
    __shared__ __attribute__((aligned(16))) float raw[(SST_H + 2) * SSP];
    __shared__ __attribute__((aligned(16))) float bn[(SST_H + 2) * SSP];
    const int H = 4 * h, W = 4 * w;
    const int X0 = blockIdx.x * SST_W, Y0 = blockIdx.y * SST_H;
    const long long b = blockIdx.z;
    const long long plane = (long long)H * W;
    const float* low = depth_low + b * h * w;
    const float s0 = prm[P_BN0], t0 = prm[P_BN0 + 1];
    for (int q = threadIdx.x; q < (SST_H + 2) * (SST_W + 2); q += 256) {
        const int r = q / (SST_W + 2), c = q - r * (SST_W + 2);
        const int yy = Y0 - 1 + r, xx = X0 - 1 + c;
        float u = 0.f, v = 0.f;
        if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) {
            u = upsampled(low, h, w, yy, xx);
            v = ss::add_rn(ss::mul_rn(u, s0), t0);
        }
        raw[r * SSP + 3 + c] = u;
        bn[r * SSP + 3 + c] = v;
    }
    __syncthreads();
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int X = X0 + tx * 4, Y = Y0 + ty;
    if (X >= W || Y >= H) return;
    const long long pix = (long long)Y * W + X;
    float4 lv[NCLS], wv[NCLS];
#pragma unroll
    for (int c = 0; c < NCLS; ++c) {
        lv[c] = *reinterpret_cast<const float4*>(label + (b * NCLS + c) * plane + pix);
        wv[c] = *reinterpret_cast<const float4*>(weights + (b * NCLS + c) * plane + pix);
    }
    float win[3][6];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const float* p = &bn[(ty + ky) * SSP + 4 + tx * 4];
        const float4 m = *reinterpret_cast<const float4*>(p);
        win[ky][0] = p[-1]; win[ky][1] = m.x; win[ky][2] = m.y; win[ky][3] = m.z; win[ky][4] = m.w; win[ky][5] = p[4];
    }
    const float4 cen = *reinterpret_cast<const float4*>(&raw[(ty + 1) * SSP + 4 + tx * 4]);
    const float cv[4] = {cen.x, cen.y, cen.z, cen.w};
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float nb[9], lab[NCLS], wt[NCLS];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) nb[ky * 3 + kx] = win[ky][j + kx];
#pragma unroll
        for (int c = 0; c < NCLS; ++c) {
            lab[c] = j == 0 ? lv[c].x : (j == 1 ? lv[c].y : (j == 2 ? lv[c].z : lv[c].w));
            wt[c] = j == 0 ? wv[c].x : (j == 1 ? wv[c].y : (j == 2 ? wv[c].z : wv[c].w));
        }
        o[j] = ssr_pixel(nb, cv[j], lab, wt, prm);
    }
    *reinterpret_cast<float4*>(out + b * plane + pix) = make_float4(o[0], o[1], o[2], o[3]);
}

}  // namespace

extern "C" int ss_ssr_upsample_fwd(const float* depth_low, const float* weights, const float* pred_label,
                                   const float* params, float* out, int B, int h, int w, int num_classes,
                                   ss_stream_t stream) {
    SS_REQUIRE(depth_low && weights && pred_label && params && out);
    SS_REQUIRE(B > 0 && h > 0 && w > 0);
    if (num_classes != NCLS) return SS_ERR_UNSUPPORTED;
    const uintptr_t bits = reinterpret_cast<uintptr_t>(weights) | reinterpret_cast<uintptr_t>(pred_label) | reinterpret_cast<uintptr_t>(out);
    if ((bits & 15) == 0 && B <= 65535 && ss::ceil_div(4 * h, SST_H) <= 65535) {
        const dim3 grid(ss::ceil_div(4 * w, SST_W), ss::ceil_div(4 * h, SST_H), B);
        hipLaunchKernelGGL(ssr_upsample_tiled, grid, dim3(256), 0, ss::as_stream(stream), depth_low, weights, pred_label, params, out, h, w);
        return ss::check_launch();
    }
    const long long total = (long long)B * 16 * h * w;
    const long long blocks = ss::ceil_div_ll(total, 256);
    if (blocks > 0x7fffffffLL) return SS_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(ssr_upsample_kernel, dim3((unsigned)blocks), dim3(256), 0, ss::as_stream(stream), depth_low, weights,
                       pred_label, params, out, h, w, total);
    return ss::check_launch();
}

extern "C" int ss_ssr_param_count(void) { return P_TOTAL; }
