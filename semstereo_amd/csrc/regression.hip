// Soft-argmax regressions over the disparity axis and the channelAtt gate (gfx950).
//
//   disparity_regression   out[b,y,x]   = sum_d p[b,d,y,x] * (d - m)                 models/submodule.py:164-170
//   disparity_variance     out[b,0,y,x] = sum_d p[b,d,y,x] * ((d - m) - disp)^2      models/submodule.py:257-263
//   regression_topk        top-k costs -> softmax -> expectation of candidates       models/submodule.py:434-442
//   channel gate           out = sigmoid(att)[:, :, None] * cv                       models/SemStereo.py:101-102
//
// All HBM/L2-bound streaming reductions.  Tensors are [B,D,H,W] with D the slow axis, so lanes
// run along W (16 B per lane) and the D reduction is split over the 4 waves of a workgroup in
// 16-plane chunks that are combined in chunk order through LDS (deterministic; this is also the
// summation tree ATen's CPU sum uses for an outer reduction of this size).
#include <algorithm>
#include <limits.h>

#include "common.h"

namespace {

constexpr int CH = 16;   // planes per partial sum

enum { RG_MEAN = 0, RG_VAR = 1 };

template <int KIND, int VEC>
__global__ __launch_bounds__(256) void regress_kernel(const float* __restrict__ prob, const float* __restrict__ disp,
                                                       float* __restrict__ out, int D, int m, long long plane,
                                                       long long nvec_per_b) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [nchunks][64][VEC]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long b = blockIdx.y;
    const long long v = blockIdx.x * 64LL + lane;      // vector index inside one image plane
    const bool active = v < nvec_per_b;
    const long long pix = v * VEC;
    const float* pb = prob + b * D * plane + pix;
    float center[VEC];
#pragma unroll
    for (int p = 0; p < VEC; ++p) center[p] = 0.f;
    if (KIND == RG_VAR && active) {
        if (VEC == 4) {
            const float4 q = *reinterpret_cast<const float4*>(disp + b * plane + pix);
            center[0] = q.x; center[1 % VEC] = q.y; center[2 % VEC] = q.z; center[3 % VEC] = q.w;
        } else {
            center[0] = disp[b * plane + pix];
        }
    }
    const int nch = ss::ceil_div(D, CH);
    for (int ch = wave; ch < nch; ch += 4) {
        float acc[VEC];
#pragma unroll
        for (int p = 0; p < VEC; ++p) acc[p] = 0.f;
        if (active) {
            const int d1 = min(D, (ch + 1) * CH);
            for (int d = ch * CH; d < d1; ++d) {
                float x[VEC];
                if (VEC == 4) {
                    const float4 q = *reinterpret_cast<const float4*>(pb + d * plane);
                    x[0] = q.x; x[1 % VEC] = q.y; x[2 % VEC] = q.z; x[3 % VEC] = q.w;
                } else {
                    x[0] = pb[d * plane];
                }
                const float dv = (float)(d - m);
#pragma unroll
                for (int p = 0; p < VEC; ++p) {
                    float wgt = dv;
                    if (KIND == RG_VAR) { const float t = dv - center[p]; wgt = ss::mul_rn(t, t); }
                    acc[p] = ss::add_rn(acc[p], ss::mul_rn(x[p], wgt));
                }
            }
        }
#pragma unroll
        for (int p = 0; p < VEC; ++p) lds[(ch * 64 + lane) * VEC + p] = acc[p];
    }
    __syncthreads();
    if (wave == 0 && active) {
        float tot[VEC];
#pragma unroll
        for (int p = 0; p < VEC; ++p) tot[p] = 0.f;
        for (int ch = 0; ch < nch; ++ch)
#pragma unroll
            for (int p = 0; p < VEC; ++p) tot[p] = ss::add_rn(tot[p], lds[(ch * 64 + lane) * VEC + p]);
        if (VEC == 4)
            *reinterpret_cast<float4*>(out + b * plane + pix) = make_float4(tot[0], tot[1 % VEC], tot[2 % VEC], tot[3 % VEC]);
        else
            out[b * plane + pix] = tot[0];
    }
}

// grad_prob[b,d,y,x] = grad_out[b,y,x] * (d - m)
__global__ void regress_bwd_kernel(const float* __restrict__ gout, float* __restrict__ gprob, int D, int m,
                                   long long plane, long long total) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long pix = i % plane;
        const long long bd = i / plane;
        const int d = (int)(bd % D);
        const long long b = bd / D;
        gprob[i] = gout[b * plane + pix] * (float)(d - m);
    }
}

// Fused softmax over D + expectation + variance (models/SemStereo.py:281-285), one pixel per lane.
__global__ __launch_bounds__(256) void softmax_regress_kernel(const float* __restrict__ logits, float* __restrict__ prob,
                                                               float* __restrict__ disp, float* __restrict__ var,
                                                               int D, int m, long long plane, long long total) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= total) return;
    const long long pix = i % plane, b = i / plane;
    const float* lp = logits + b * D * plane + pix;
    float mx = -INFINITY;
    for (int d = 0; d < D; ++d) mx = fmaxf(mx, lp[d * plane]);
    float sum = 0.f;
    for (int d = 0; d < D; ++d) sum = ss::add_rn(sum, expf(lp[d * plane] - mx));
    float mean = 0.f, part = 0.f;
    for (int d = 0; d < D; ++d) {
        const float p = expf(lp[d * plane] - mx) / sum;
        if (prob) prob[(b * D + d) * plane + pix] = p;
        part = ss::add_rn(part, ss::mul_rn(p, (float)(d - m)));
        if ((d % CH) == CH - 1) { mean = ss::add_rn(mean, part); part = 0.f; }
    }
    mean = ss::add_rn(mean, part);
    float v = 0.f; part = 0.f;
    for (int d = 0; d < D; ++d) {
        const float p = expf(lp[d * plane] - mx) / sum;
        const float t = (float)(d - m) - mean;
        part = ss::add_rn(part, ss::mul_rn(p, ss::mul_rn(t, t)));
        if ((d % CH) == CH - 1) { v = ss::add_rn(v, part); part = 0.f; }
    }
    v = ss::add_rn(v, part);
    disp[i] = mean;
    var[i] = v;
}

// "j comes after (pv,pi)" / "j beats (bv,bi)" in the order (value descending, index ascending)
__device__ __forceinline__ bool after(float v, int j, float pv, int pi) { return (v < pv) || (v == pv && j > pi); }
__device__ __forceinline__ bool beats(float v, int j, float bv, int bi) { return (v > bv) || (v == bv && j < bi); }

__device__ __forceinline__ void select_next(const float* __restrict__ cp, int nd, long long plane, float pv, int pi,
                                            float& bv, int& bi) {
    bv = -INFINITY; bi = INT_MAX;
    for (int j = 0; j < nd; ++j) {
        const float v = cp[j * plane];
        if (after(v, j, pv, pi) && beats(v, j, bv, bi)) { bv = v; bi = j; }
    }
    if (bi == INT_MAX) { bi = 0; }
}

template <int K>
__global__ __launch_bounds__(256) void topk_regress_kernel(const float* __restrict__ cost, const float* __restrict__ samples,
                                                            float* __restrict__ out, int nd, int k, long long plane,
                                                            long long total) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= total) return;
    const long long pix = i % plane, b = i / plane;
    const float* cp = cost + b * nd * plane + pix;
    const float* sp = samples + b * nd * plane + pix;
    if (K > 0) {
        float sv[K > 0 ? K : 1]; int si[K > 0 ? K : 1];
        float pv = INFINITY; int pi = -1;
#pragma unroll
        for (int s = 0; s < K; ++s) { select_next(cp, nd, plane, pv, pi, sv[s], si[s]); pv = sv[s]; pi = si[s]; }
        float e[K > 0 ? K : 1], sum = 0.f;
#pragma unroll
        for (int s = 0; s < K; ++s) { e[s] = expf(sv[s] - sv[0]); sum = ss::add_rn(sum, e[s]); }
        float acc = 0.f;
#pragma unroll
        for (int s = 0; s < K; ++s) acc = ss::add_rn(acc, ss::mul_rn(sp[si[s] * plane], e[s] / sum));
        out[i] = acc;
    } else {
        // generic k: two selection sweeps (normaliser first, then the expectation)
        float pv = INFINITY; int pi = -1; float top = 0.f, sum = 0.f;
        for (int s = 0; s < k; ++s) {
            float bv; int bi; select_next(cp, nd, plane, pv, pi, bv, bi);
            if (s == 0) top = bv;
            sum = ss::add_rn(sum, expf(bv - top));
            pv = bv; pi = bi;
        }
        pv = INFINITY; pi = -1; float acc = 0.f;
        for (int s = 0; s < k; ++s) {
            float bv; int bi; select_next(cp, nd, plane, pv, pi, bv, bi);
            acc = ss::add_rn(acc, ss::mul_rn(sp[bi * plane], expf(bv - top) / sum));
            pv = bv; pi = bi;
        }
        out[i] = acc;
    }
}

// backward of regression_topk (models/submodule.py:434-442): with pool = the k selected candidates, p = softmax(cost[pool]),
// y = sum_j p_j s_j:   dL/dcost[pool_j] = g * p_j * (s_j - y),   dL/dsamples[pool_j] = g * p_j,   0 for the others
// (the sort is piecewise constant: no gradient through the selection, as in autograd of the reference's composition).
__global__ __launch_bounds__(256) void topk_regress_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ cost,
                                                                const float* __restrict__ samples, float* __restrict__ gcost,
                                                                float* __restrict__ gsamples, int nd, int k, long long plane,
                                                                long long total) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= total) return;
    const long long pix = i % plane, b = i / plane;
    const float* cp = cost + b * nd * plane + pix;
    const float* sp = samples + b * nd * plane + pix;
    float* gc = gcost + b * nd * plane + pix;
    float* gs = gsamples + b * nd * plane + pix;
    for (int j = 0; j < nd; ++j) { gc[j * plane] = 0.f; gs[j * plane] = 0.f; }
    const float g = gout[i];
    float pv = INFINITY; int pi = -1; float top = 0.f, sum = 0.f;
    for (int s = 0; s < k; ++s) {
        float bv; int bi; select_next(cp, nd, plane, pv, pi, bv, bi);
        if (s == 0) top = bv;
        sum = ss::add_rn(sum, expf(bv - top));
        pv = bv; pi = bi;
    }
    pv = INFINITY; pi = -1; float y = 0.f;
    for (int s = 0; s < k; ++s) {
        float bv; int bi; select_next(cp, nd, plane, pv, pi, bv, bi);
        y = ss::add_rn(y, ss::mul_rn(sp[bi * plane], expf(bv - top) / sum));
        pv = bv; pi = bi;
    }
    pv = INFINITY; pi = -1;
    for (int s = 0; s < k; ++s) {
        float bv; int bi; select_next(cp, nd, plane, pv, pi, bv, bi);
        const float p = expf(bv - top) / sum;
        gs[bi * plane] = g * p;
        gc[bi * plane] = g * p * (sp[bi * plane] - y);
        pv = bv; pi = bi;
    }
}

template <int VEC>
__global__ __launch_bounds__(256) void gate_kernel(const float* __restrict__ att, const float* __restrict__ cv,
                                                    float* __restrict__ out, int D, long long plane, long long nvec) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;   // over B*C*(plane/VEC)
    if (i >= nvec) return;
    const long long pv = plane / VEC;
    const long long bc = i / pv, pix = (i % pv) * VEC;
    float g[VEC];
    if (VEC == 4) {
        const float4 q = *reinterpret_cast<const float4*>(att + bc * plane + pix);
        g[0] = q.x; g[1 % VEC] = q.y; g[2 % VEC] = q.z; g[3 % VEC] = q.w;
    } else {
        g[0] = att[bc * plane + pix];
    }
#pragma unroll
    for (int p = 0; p < VEC; ++p) g[p] = 1.0f / (1.0f + expf(-g[p]));
    const float* cp = cv + bc * D * plane + pix;
    float* op = out + bc * D * plane + pix;
    for (int d = 0; d < D; ++d) {
        if (VEC == 4) {
            float4 q = *reinterpret_cast<const float4*>(cp + d * plane);
            q.x = ss::mul_rn(g[0], q.x); q.y = ss::mul_rn(g[1 % VEC], q.y);
            q.z = ss::mul_rn(g[2 % VEC], q.z); q.w = ss::mul_rn(g[3 % VEC], q.w);
            *reinterpret_cast<float4*>(op + d * plane) = q;
        } else {
            op[d * plane] = ss::mul_rn(g[0], cp[d * plane]);
        }
    }
}

template <int KIND>
int launch_regress(const float* prob, const float* disp, float* out, int B, int dmin, int D, int H, int W, hipStream_t st) {
    const int m = -dmin;                                       // plane d holds disparity d - m
    const long long plane = (long long)H * W;
    uintptr_t bits = reinterpret_cast<uintptr_t>(prob) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(disp);
    const bool v4 = (plane % 4 == 0) && ((bits & 15) == 0);
    const long long nvec = v4 ? plane / 4 : plane;
    const int nch = ss::ceil_div(D, CH);
    const size_t lds = (size_t)nch * 64 * (v4 ? 4 : 1) * sizeof(float);
    if (lds > 64 * 1024 || B > 65535) return SS_ERR_UNSUPPORTED;
    dim3 grid((unsigned)ss::ceil_div_ll(nvec, 64), B);
    if (v4)
        hipLaunchKernelGGL((regress_kernel<KIND, 4>), grid, dim3(256), lds, st, prob, disp, out, D, m, plane, nvec);
    else
        hipLaunchKernelGGL((regress_kernel<KIND, 1>), grid, dim3(256), lds, st, prob, disp, out, D, m, plane, nvec);
    return ss::check_launch();
}

}  // namespace

extern "C" int ss_disparity_regression_fwd(const float* prob, float* out, int B, int dmin, int ndisp, int H, int W,
                                           ss_stream_t stream) {
    SS_REQUIRE(prob && out);
    SS_REQUIRE(B > 0 && ndisp > 0 && H > 0 && W > 0);
    return launch_regress<RG_MEAN>(prob, nullptr, out, B, dmin, ndisp, H, W, ss::as_stream(stream));
}

extern "C" int ss_disparity_variance_fwd(const float* prob, const float* disparity, float* out, int B, int dmin, int ndisp,
                                         int H, int W, ss_stream_t stream) {
    SS_REQUIRE(prob && disparity && out);
    SS_REQUIRE(B > 0 && ndisp > 0 && H > 0 && W > 0);
    return launch_regress<RG_VAR>(prob, disparity, out, B, dmin, ndisp, H, W, ss::as_stream(stream));
}

extern "C" int ss_disparity_regression_bwd(const float* grad_out, float* grad_prob, int B, int dmin, int ndisp, int H, int W,
                                           ss_stream_t stream) {
    SS_REQUIRE(grad_out && grad_prob);
    SS_REQUIRE(B > 0 && ndisp > 0 && H > 0 && W > 0);
    const long long plane = (long long)H * W, total = (long long)B * ndisp * plane;
    const int blocks = (int)std::min<long long>(ss::ceil_div_ll(total, 256), 256 * 32);
    hipLaunchKernelGGL(regress_bwd_kernel, dim3(blocks), dim3(256), 0, ss::as_stream(stream), grad_out, grad_prob,
                       ndisp, -dmin, plane, total);
    return ss::check_launch();
}

extern "C" int ss_softmax_regression_fwd(const float* logits, float* prob, float* disp, float* var, int B, int dmin, int ndisp,
                                         int H, int W, ss_stream_t stream) {
    SS_REQUIRE(logits && disp && var);
    SS_REQUIRE(B > 0 && ndisp > 0 && H > 0 && W > 0);
    if (ss_softmax_regress_split_launch(logits, prob, disp, var, B, dmin, ndisp, H, W, ss::as_stream(stream)) == 0)
        return ss::check_launch();
    const long long plane = (long long)H * W, total = (long long)B * plane;
    hipLaunchKernelGGL(softmax_regress_kernel,dim3((unsigned)ss::ceil_div_ll(total, 256)), dim3(256), 0,
                       ss::as_stream(stream), logits, prob, disp, var, ndisp, -dmin, plane, total);
    return ss::check_launch();
}

extern "C" int ss_regression_topk_fwd(const float* cost, const float* samples, float* out, int B, int nd, int H, int W,
                                      int k, ss_stream_t stream) {
    SS_REQUIRE(cost && samples && out);
    SS_REQUIRE(B > 0 && nd > 0 && H > 0 && W > 0);
    SS_REQUIRE(k >= 1 && k <= nd && k <= 32);
    const long long plane = (long long)H * W, total = (long long)B * plane;
    dim3 grid((unsigned)ss::ceil_div_ll(total, 256)), block(256);
    hipStream_t st = ss::as_stream(stream);
    switch (k) {
        case 1: hipLaunchKernelGGL(topk_regress_kernel<1>, grid, block, 0, st, cost, samples, out, nd, k, plane, total); break;
        case 2: hipLaunchKernelGGL(topk_regress_kernel<2>, grid, block, 0, st, cost, samples, out, nd, k, plane, total); break;
        case 3: hipLaunchKernelGGL(topk_regress_kernel<3>, grid, block, 0, st, cost, samples, out, nd, k, plane, total); break;
        case 4: hipLaunchKernelGGL(topk_regress_kernel<4>, grid, block, 0, st, cost, samples, out, nd, k, plane, total); break;
        default: hipLaunchKernelGGL(topk_regress_kernel<0>, grid, block, 0, st, cost, samples, out, nd, k, plane, total); break;
    }
    return ss::check_launch();
}

extern "C" int ss_regression_topk_bwd(const float* grad_out, const float* cost, const float* samples, float* grad_cost,
                                      float* grad_samples, int B, int nd, int H, int W, int k, ss_stream_t stream) {
    SS_REQUIRE(grad_out && cost && samples && grad_cost && grad_samples);
    SS_REQUIRE(B > 0 && nd > 0 && H > 0 && W > 0);
    SS_REQUIRE(k >= 1 && k <= nd && k <= 32);
    const long long plane = (long long)H * W, total = (long long)B * plane;
    hipLaunchKernelGGL(topk_regress_bwd_kernel, dim3((unsigned)ss::ceil_div_ll(total, 256)), dim3(256), 0, ss::as_stream(stream),
                       grad_out, cost, samples, grad_cost, grad_samples, nd, k, plane, total);
    return ss::check_launch();
}

extern "C" int ss_channel_gate_fwd(const float* att_logits, const float* cv, float* out, int B, int C, int D, int H,
                                   int W, ss_stream_t stream) {
    SS_REQUIRE(att_logits && cv && out);
    SS_REQUIRE(B > 0 && C > 0 && D > 0 && H > 0 && W > 0);
    const long long plane = (long long)H * W;
    uintptr_t bits = reinterpret_cast<uintptr_t>(att_logits) | reinterpret_cast<uintptr_t>(cv) | reinterpret_cast<uintptr_t>(out);
    const bool v4 = (plane % 4 == 0) && ((bits & 15) == 0);
    const long long nvec = (long long)B * C * (v4 ? plane / 4 : plane);
    dim3 grid((unsigned)ss::ceil_div_ll(nvec, 256));
    if (v4) hipLaunchKernelGGL(gate_kernel<4>, grid, dim3(256), 0, ss::as_stream(stream), att_logits, cv, out, D, plane, nvec);
    else hipLaunchKernelGGL(gate_kernel<1>, grid, dim3(256), 0, ss::as_stream(stream), att_logits, cv, out, D, plane, nvec);
    return ss::check_launch();
}
