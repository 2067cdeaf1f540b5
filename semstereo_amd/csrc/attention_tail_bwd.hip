// Backward of the three fused "attention tail" launches of attention_tail.hip (training, main_us3d.py:186-222 back-propagates
// through models/SemStereo.py:279-310): until round 4 the tail ran as ~25 PyTorch statements whenever autograd was on.
//
//   ss_upsample_softmax_regression_bwd   :279-285   trilinear 2x up-sampling -> softmax over D -> expectation, variance
//   ss_sample_strength_bwd               :286-293   sigmoid gate, 5-tap propagations, 5-candidate warp, channel mean, softmax
//   ss_topk_candidates_bwd               :295-310   strength-weighted 5-tap propagation of the logits, softmax, top-24 gather,
//                                                   soft-argmax over the 24
//
// Each kernel recomputes its forward quantities from the saved inputs (the forward kernels keep nothing but their outputs) with
// the SAME coordinate / weight arithmetic as the forward, one thread per quarter-resolution pixel; scatters into neighbouring
// pixels (the replicate-padded propagation taps, the bilinear taps of the right features) are fp32 atomics, like the backward
// of SpatialTransformer_grid (warp.hip).  These are HBM / latency kernels on [B,D,H/4,W/4] volumes: a few MB per pair.
#include "common.h"

namespace {

__device__ __constant__ int kTapDy[5] = {-1, 0, 1, 1, -1};     // models/submodule.py:295-300 / 367-372
__device__ __constant__ int kTapDx[5] = {-1, 0, 1, -1, 1};

// ---- :279-285 ---------------------------------------------------------------------------------------------------------------------
// du[k] = g_up[k] + p[k] * (G[k] - sum_j p[j] G[j]),  G[k] = g_disp * v[k] + g_var * (v[k] - disp)^2,  p = softmax(up), v[k] = dmin + k
// (the variance's dependence on `disp` vanishes: sum_k p[k] (v[k] - disp) = 0).  One thread per fine pixel, two passes over D.
__global__ __launch_bounds__(256) void upsoft_bwd_fine_kernel(const float* __restrict__ up, const float* __restrict__ g_up,
                                                               const float* __restrict__ g_disp, const float* __restrict__ g_var,
                                                               float* __restrict__ du, int D, int dmin, long long plane, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long long b = i / plane, pix = i - b * plane;
    const float* u = up + b * D * plane + pix;
    float mx = -INFINITY;
    for (int k = 0; k < D; ++k) mx = fmaxf(mx, u[k * plane]);
    float sum = 0.f, e1 = 0.f;
    for (int k = 0; k < D; ++k) {
        const float e = expf(u[k * plane] - mx);
        sum += e;
        e1 += e * (float)(dmin + k);
    }
    const float disp = e1 / sum;
    const float gd = g_disp ? g_disp[i] : 0.f, gv = g_var ? g_var[i] : 0.f;
    float dot = 0.f;
    for (int k = 0; k < D; ++k) {
        const float p = expf(u[k * plane] - mx) / sum, v = (float)(dmin + k) - disp;
        dot += p * (gd * (float)(dmin + k) + gv * v * v);
    }
    float* o = du + b * D * plane + pix;
    const float* gu = g_up ? g_up + b * D * plane + pix : nullptr;
    for (int k = 0; k < D; ++k) {
        const float p = expf(u[k * plane] - mx) / sum, v = (float)(dmin + k) - disp;
        o[k * plane] = (gu ? gu[k * plane] : 0.f) + p * (gd * (float)(dmin + k) + gv * v * v - dot);
    }
}

// transposed trilinear 2x up-sampling (align_corners = False, exact factor 2 in D, H, W): coarse voxel m gathers, per axis, from
// the fine indices 2m - 1 (x 0.25), 2m (x 0.75), 2m + 1 (x 0.75), 2m + 2 (x 0.25); the clamped ends fold onto the edge voxels
// (fine 0 gives all of itself to coarse 0, fine 2n - 1 to coarse n - 1).
__device__ __forceinline__ void axis_taps(int m, int n, int (&f)[4], float (&w)[4]) {
    f[0] = 2 * m - 1; w[0] = 0.25f;
    f[1] = 2 * m;     w[1] = (m == 0) ? 1.0f : 0.75f;
    f[2] = 2 * m + 1; w[2] = (m == n - 1) ? 1.0f : 0.75f;
    f[3] = 2 * m + 2; w[3] = 0.25f;
    if (f[0] < 0) w[0] = 0.f;
    if (f[3] > 2 * n - 1) w[3] = 0.f;
    f[0] = max(f[0], 0); f[3] = min(f[3], 2 * n - 1);
}

__global__ __launch_bounds__(256) void upsoft_bwd_coarse_kernel(const float* __restrict__ du, float* __restrict__ g_coarse, int Dc, int Hc,
                                                                 int Wc, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int x = (int)(i % Wc), y = (int)((i / Wc) % Hc), d = (int)((i / ((long long)Wc * Hc)) % Dc);
    const long long b = i / ((long long)Wc * Hc * Dc);
    int fd[4], fy[4], fx[4];
    float wd[4], wy[4], wx[4];
    axis_taps(d, Dc, fd, wd); axis_taps(y, Hc, fy, wy); axis_taps(x, Wc, fx, wx);
    const int H = 2 * Hc, W = 2 * Wc;
    const float* base = du + b * (long long)(2 * Dc) * H * W;
    float s = 0.f;
    for (int a = 0; a < 4; ++a) {
        if (wd[a] == 0.f) continue;
        for (int c = 0; c < 4; ++c) {
            if (wy[c] == 0.f) continue;
            const float* row = base + ((long long)fd[a] * H + fy[c]) * W;
            float r = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) r += wx[e] * row[fx[e]];
            s += wd[a] * wy[c] * r;
        }
    }
    g_coarse[i] = s;
}

// ---- :286-293 ---------------------------------------------------------------------------------------------------------------------
struct Taps4 {
    int o_nw, o_ne, o_sw, o_se;
    float w_nw, w_ne, w_sw, w_se, fw, fs, fn;
    int ix, iy;                       // column of the west taps, row of the north taps
};

// the forward's arithmetic (attention_tail.hip::bilinear_taps / warp.hip::make_taps), plus the fractions the x-derivative needs
__device__ __forceinline__ Taps4 bilinear_taps(float disp, int h, int w, int H, int W, float half_w, float half_h) {
    const float gx = ((float)w - disp) / half_w - 1.0f;
    const float gy = (float)h / half_h - 1.0f;
    const float ix = ss::mul_rn(gx + 1.0f, half_w);
    const float iy = ss::mul_rn(gy + 1.0f, half_h);
    const float xw = floorf(ix), yn = floorf(iy);
    const float fw = ss::sub_rn(ix, xw), fe = 1.0f - fw, fn = ss::sub_rn(iy, yn), fs = 1.0f - fn;
    const float xe = xw + 1.0f, ys = yn + 1.0f;
    const bool mw = (xw > -1.0f) && (xw < (float)W), me = (xe > -1.0f) && (xe < (float)W);
    const bool mn = (yn > -1.0f) && (yn < (float)H), ms = (ys > -1.0f) && (ys < (float)H);
    const int ixw = (int)xw, iyn = (int)yn;
    Taps4 t;
    t.w_nw = ss::mul_rn(fs, fe); t.w_ne = ss::mul_rn(fs, fw);
    t.w_sw = ss::mul_rn(fn, fe); t.w_se = ss::mul_rn(fn, fw);
    t.fw = fw; t.fs = fs; t.fn = fn;
    t.ix = ixw; t.iy = iyn;
    t.o_nw = (mn && mw) ? iyn * W + ixw : -1;
    t.o_ne = (mn && me) ? iyn * W + ixw + 1 : -1;
    t.o_sw = (ms && mw) ? (iyn + 1) * W + ixw : -1;
    t.o_se = (ms && me) ? (iyn + 1) * W + ixw + 1 : -1;
    return t;
}

// A workgroup = 64 pixels (lanes along x) x NW waves that split the channels (wave w owns channels w, w + NW, ...).  Pass 1
// recomputes corr[t] = mean_c left * warp_t(right) -- each wave its channels, summed through LDS in a fixed order -- and every wave
// redoes the small softmax backward for its lanes' pixels; pass 2 walks the wave's channels again for the feature gradients.  g_left is
// owned by (pixel, channel); g_right, g_pred0, g_var are scattered with atomics (zeroed by the launcher; taps of weight zero -- the
// south row wherever the row coordinate is exact, three rows in four -- are skipped); g_gamma / g_beta are reduced over wave 0, one
// atomic pair per workgroup.  (Rounds 4-5: ONE thread per pixel walked all C channels twice -- 1 024 waves on the whole chip, 168 M
// unconditional atomics: 3.6 ms of the 1024^2 training step, measured r06.)
constexpr int SSB_NW = 8;
// ... and (r06, second step) the scatter into g_right goes through LDS: a workgroup's 64 pixels lie in one row y (W % 64 == 0) and the
// five candidates' taps land in that same row within +-SSB_M columns of them (|disparity| <= maxdisp / 4), so each wave sums its
// channel's contributions in a private row buffer (tagged read-add-write, ss::lds_owned_add) and adds the touched part to memory once: ~1 global atomic per
// (pixel, channel) instead of ~12.  Taps in another row (the south taps where the row coordinate is not exact) or beyond the
// margin go to memory directly, as before.
constexpr int SSB_M = 64, SSB_RB = 64 + 2 * SSB_M + 2;
// (four waves per SIMD -- 128 registers, 212 bytes of scratch per lane -- instead of the 176 registers the compiler takes when left alone:
// the kernel waits for memory and LDS 80 % of its life (SQ counters, r06), and two waves per SIMD did not cover that: 664 -> 586 us alone;
// unrolling the channel loops to put more gathers in flight spilled further and lost: 828 us)
__global__ __launch_bounds__(64 * SSB_NW, 4) void sample_strength_bwd_kernel(const float* __restrict__ left, const float* __restrict__ right,
                                                                   const float* __restrict__ pred0, const float* __restrict__ var,
                                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                   const float* __restrict__ g_strength, float* __restrict__ g_left,
                                                                   float* __restrict__ g_right, float* __restrict__ g_pred0,
                                                                   float* __restrict__ g_var, float* __restrict__ g_gb, int C, int H, int W,
                                                                   float half_w, float half_h, long long total) {
    constexpr int NW = SSB_NW;
    __shared__ float red[NW][5][64], gred[NW][5][64], rowbuf[NW][SSB_RB + 62];
    __shared__ int rowtag[NW][SSB_RB + 62];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long i = (long long)blockIdx.x * 64 + lane;
    const bool active = i < total;
    const long long plane = (long long)H * W;
    const long long ii = active ? i : 0;
    const int x = (int)(ii % W), y = (int)((ii / W) % H);
    const long long b = ii / plane, pix = (long long)y * W + x;
    Taps4 tp[5];
    long long nbs[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        const int yy = min(max(y + kTapDy[t], 0), H - 1), xx = min(max(x + kTapDx[t], 0), W - 1);
        nbs[t] = b * plane + (long long)yy * W + xx;
        tp[t] = bilinear_taps(pred0[nbs[t]], y, x, H, W, half_w, half_h);
    }
    auto sample4 = [&](const float* rp, const Taps4& t, float& a, float& bq, float& c, float& d) {
        a = (t.o_nw >= 0) ? rp[t.o_nw] : 0.f; bq = (t.o_ne >= 0) ? rp[t.o_ne] : 0.f;
        c = (t.o_sw >= 0) ? rp[t.o_sw] : 0.f; d = (t.o_se >= 0) ? rp[t.o_se] : 0.f;
    };
    float corr[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (int c = wave; c < C; c += NW) {
        const float l = left[(b * C + c) * plane + pix];
        const float* rp = right + (b * C + c) * plane;
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            float a, bq, cq, d;
            sample4(rp, tp[t], a, bq, cq, d);
            corr[t] += l * (a * tp[t].w_nw + bq * tp[t].w_ne + cq * tp[t].w_sw + d * tp[t].w_se);
        }
    }
#pragma unroll
    for (int t = 0; t < 5; ++t) red[wave][t][lane] = corr[t];
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) s += red[w][t][lane];
        corr[t] = s;
    }
    const float g = gamma[0], bt = beta[0];
    float gate[5], z[5], mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        corr[t] /= (float)C;
        gate[t] = 1.0f / (1.0f + expf(-(bt + g * var[nbs[t]])));
        z[t] = corr[t] * gate[t];
        mx = fmaxf(mx, z[t]);
    }
    float sum = 0.f, dot = 0.f;
#pragma unroll
    for (int t = 0; t < 5; ++t) { z[t] = expf(z[t] - mx); sum += z[t]; }
    float gs[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) { z[t] /= sum; gs[t] = g_strength[(b * 5 + t) * plane + pix]; dot += z[t] * gs[t]; }
    float dcorr[5], dgamma = 0.f, dbeta = 0.f;
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        const float dz = z[t] * (gs[t] - dot);
        dcorr[t] = dz * gate[t] / (float)C;
        if (wave == 0 && active) {                           // (wave-uniform) the per-pixel scalars: once per pixel
            const float dv = dz * corr[t] * gate[t] * (1.0f - gate[t]);
            dbeta += dv;
            dgamma += dv * var[nbs[t]];
            if (g_var) unsafeAtomicAdd(&g_var[nbs[t]], dv * g);
        }
    }
    float gix[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    // (a workgroup whose 64 pixels share a row: every lane active, same y -- wave-uniform)
    const bool rowblock = (W % 64 == 0) && g_right != nullptr;
    const int xb = (int)((((long long)blockIdx.x * 64) % W)) - SSB_M;      // column of rowbuf[.][0]
    float* rb = rowbuf[wave];
    if (active) {
        for (int c = wave; c < C; c += NW) {
            const float l = left[(b * C + c) * plane + pix];
            const float* rp = right + (b * C + c) * plane;
            float* grp = g_right ? g_right + (b * C + c) * plane : nullptr;
            float gl = 0.f;
            if (rowblock) {
#pragma unroll
                for (int k = 0; k < 4; ++k) ss::lds_put(&rb[lane + 64 * k], 0.f);
                __builtin_amdgcn_wave_barrier();
            }
            auto scatter = [&](int o, int row, int col, float v) {
                if (o < 0 || v == 0.f) return;
                const unsigned k = (unsigned)(col - xb);
                const bool in_row = rowblock && row == y && k < (unsigned)SSB_RB;
                if (!in_row) unsafeAtomicAdd(&grp[o], v);
                ss::lds_owned_add(rowtag[wave], k, in_row, rb, v);                                  // (the wave's own row buffer: common.h)
            };
#pragma unroll
            for (int t = 0; t < 5; ++t) {
                float a, bq, cq, d;
                sample4(rp, tp[t], a, bq, cq, d);
                gl += dcorr[t] * (a * tp[t].w_nw + bq * tp[t].w_ne + cq * tp[t].w_sw + d * tp[t].w_se);
                const float gr = dcorr[t] * l;
                if (grp) {
                    if (tp[t].w_nw != 0.f) scatter(tp[t].o_nw, tp[t].iy, tp[t].ix, gr * tp[t].w_nw);
                    if (tp[t].w_ne != 0.f) scatter(tp[t].o_ne, tp[t].iy, tp[t].ix + 1, gr * tp[t].w_ne);
                    if (tp[t].w_sw != 0.f) scatter(tp[t].o_sw, tp[t].iy + 1, tp[t].ix, gr * tp[t].w_sw);
                    if (tp[t].w_se != 0.f) scatter(tp[t].o_se, tp[t].iy + 1, tp[t].ix + 1, gr * tp[t].w_se);
                }
                // d(sample)/d(ix) = (b - a) * fs + (d - c) * fn     (ATen's grid_sampler backward: gix)
                gix[t] += gr * ((bq - a) * tp[t].fs + (d - cq) * tp[t].fn);
            }
            if (g_left) g_left[(b * C + c) * plane + pix] = gl;
            if (rowblock) {                                  // the touched part of the row -> memory
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int idx = lane + 64 * k;
                    const float v = ss::lds_get(&rb[idx]);
                    if (idx < SSB_RB && v != 0.f) unsafeAtomicAdd(&grp[(long long)y * W + xb + idx], v);
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 5; ++t) gred[wave][t][lane] = gix[t];
    __syncthreads();
    if (wave == 0) {
        if (g_pred0 && active) {
#pragma unroll
            for (int t = 0; t < 5; ++t) {    // ix = ((w - disp)/half_w - 1 + 1) * half_w: d ix / d disp = -1, by ATen's chain -(half_w * gix) / half_w
                float s = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) s += gred[w][t][lane];
                unsafeAtomicAdd(&g_pred0[nbs[t]], -((half_w * s) / half_w));
            }
        }
        // gamma, beta: one atomic pair per workgroup
        for (int o = 32; o > 0; o >>= 1) { dgamma += __shfl_xor(dgamma, o); dbeta += __shfl_xor(dbeta, o); }
        if (lane == 0 && g_gb) {
            unsafeAtomicAdd(&g_gb[0], dgamma);
            unsafeAtomicAdd(&g_gb[1], dbeta);
        }
    }
}

// ---- the same backward as TWO launches over a [B,5,H,W] scratch (r06, ss_sample_strength_bwd_ws): (1) the correlation, the soft-max
// backward and the per-pixel scalars -> dcorr; (2) the feature gradients with the CHANNELS on the grid (one channel per wave, C / 8 times the
// workgroups of the one-launch kernel above, whose waves walk 16 channels one after the other at two to four waves per SIMD and wait for memory
// and LDS 80 % of their life: 664 us alone at 1024^2 / 128 channels).
__global__ __launch_bounds__(64 * SSB_NW) void ssb_dcorr_kernel(const float* __restrict__ left, const float* __restrict__ right,
                                                              const float* __restrict__ pred0, const float* __restrict__ var,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              const float* __restrict__ g_strength, float* __restrict__ dcorr_out,
                                                              float* __restrict__ g_var, float* __restrict__ g_gb, int C, int H, int W,
                                                              float half_w, float half_h, long long total) {
    constexpr int NW = SSB_NW;
    __shared__ float red[NW][5][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long i = (long long)blockIdx.x * 64 + lane;
    const bool active = i < total;
    const long long plane = (long long)H * W;
    const long long ii = active ? i : 0;
    const int x = (int)(ii % W), y = (int)((ii / W) % H);
    const long long b = ii / plane, pix = (long long)y * W + x;
    Taps4 tp[5];
    long long nbs[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        const int yy = min(max(y + kTapDy[t], 0), H - 1), xx = min(max(x + kTapDx[t], 0), W - 1);
        nbs[t] = b * plane + (long long)yy * W + xx;
        tp[t] = bilinear_taps(pred0[nbs[t]], y, x, H, W, half_w, half_h);
    }
    float corr[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (int c = wave; c < C; c += NW) {
        const float l = left[(b * C + c) * plane + pix];
        const float* rp = right + (b * C + c) * plane;
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            const float a = (tp[t].o_nw >= 0) ? rp[tp[t].o_nw] : 0.f, bq = (tp[t].o_ne >= 0) ? rp[tp[t].o_ne] : 0.f;
            const float cq = (tp[t].o_sw >= 0) ? rp[tp[t].o_sw] : 0.f, d = (tp[t].o_se >= 0) ? rp[tp[t].o_se] : 0.f;
            corr[t] += l * (a * tp[t].w_nw + bq * tp[t].w_ne + cq * tp[t].w_sw + d * tp[t].w_se);
        }
    }
#pragma unroll
    for (int t = 0; t < 5; ++t) red[wave][t][lane] = corr[t];
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) s += red[w][t][lane];
        corr[t] = s;
    }
    const float g = gamma[0], bt = beta[0];
    float gate[5], z[5], mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        corr[t] /= (float)C;
        gate[t] = 1.0f / (1.0f + expf(-(bt + g * var[nbs[t]])));
        z[t] = corr[t] * gate[t];
        mx = fmaxf(mx, z[t]);
    }
    float sum = 0.f, dot = 0.f;
#pragma unroll
    for (int t = 0; t < 5; ++t) { z[t] = expf(z[t] - mx); sum += z[t]; }
    float gs[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) { z[t] /= sum; gs[t] = g_strength[(b * 5 + t) * plane + pix]; dot += z[t] * gs[t]; }
    float dgamma = 0.f, dbeta = 0.f;
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        const float dz = z[t] * (gs[t] - dot);
        if (active) {
            dcorr_out[(b * 5 + t) * plane + pix] = dz * gate[t] / (float)C;
            const float dv = dz * corr[t] * gate[t] * (1.0f - gate[t]);
            dbeta += dv;
            dgamma += dv * var[nbs[t]];
            if (g_var) unsafeAtomicAdd(&g_var[nbs[t]], dv * g);
        }
    }
    for (int o = 32; o > 0; o >>= 1) { dgamma += __shfl_xor(dgamma, o); dbeta += __shfl_xor(dbeta, o); }
    if (lane == 0 && g_gb) {
        unsafeAtomicAdd(&g_gb[0], dgamma);
        unsafeAtomicAdd(&g_gb[1], dbeta);
    }
}

// grid (pixel blocks of 64, ceil(C / 16)): wave w of a workgroup owns the channel PAIR 2 (8 blockIdx.y + w) + {0, 1}.  (A wave spends its
// life waiting -- 57 % of its cycles on memory, 15 % issuing, SQ counters r06 -- so it carries two channels through the same chain of
// loads: the taps are formed once, the gathers of both channels are in flight together and one claim of a row-buffer slot serves both.)
constexpr int SSB_CH = 2;
__global__ __launch_bounds__(64 * SSB_NW) void ssb_scatter_kernel(const float* __restrict__ left, const float* __restrict__ right,
                                                                const float* __restrict__ pred0, const float* __restrict__ dcorr_in,
                                                                float* __restrict__ g_left, float* __restrict__ g_right,
                                                                float* __restrict__ g_pred0, int C, int H, int W, float half_w,
                                                                float half_h, long long total) {
    constexpr int NW = SSB_NW;
    constexpr int RS = SSB_RB + 62;
    __shared__ float gred[NW][5][64], rowbuf[NW][2][SSB_CH][RS];        // [wave][row y / the other row][channel][column - xb]
    __shared__ int rowtag[NW][2][RS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long i = (long long)blockIdx.x * 64 + lane;
    const bool active = i < total;
    const long long plane = (long long)H * W;
    const long long ii = active ? i : 0;
    const int x = (int)(ii % W), y = (int)((ii / W) % H);
    const long long b = ii / plane, pix = (long long)y * W + x;
    const int c0 = (blockIdx.y * NW + wave) * SSB_CH;
    const bool cok = c0 < C, two = c0 + 1 < C;               // (wave-uniform)
    float gix[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    const bool rowblock = (W % 64 == 0) && g_right != nullptr;
    const int xb = (int)((((long long)blockIdx.x * 64) % W)) - SSB_M;
    float* rows = &rowbuf[wave][0][0][0];
    int* tags = &rowtag[wave][0][0];
    // the rows of the taps (the same for every candidate and lane: the probe shifts along x only): floor(iy) and the one below; the
    // coordinate round trip leaves iy = y -+ ~1e-5 on most rows, so the second row is the rule, with weights ~1e-5 (warp.hip, r06)
    int other = -1;
    if (cok) {
        if (rowblock) {
#pragma unroll
            for (int n = 0; n < 2 * SSB_CH; ++n)
#pragma unroll
                for (int k = 0; k < 4; ++k) ss::lds_put(&rows[n * RS + lane + 64 * k], 0.f);
        }
        const long long ca = (b * C + c0) * plane, cb = (b * C + (two ? c0 + 1 : c0)) * plane;
        const float la = left[ca + pix], lb = two ? left[cb + pix] : 0.f;
        const float* rpa = right + ca;
        const float* rpb = right + cb;
        float* gra_p = g_right ? g_right + ca : nullptr;
        float* grb_p = g_right ? g_right + cb : nullptr;
        float gla = 0.f, glb = 0.f;
        auto scatter = [&](int o, int row, int col, float va, float vb) {
            if (o < 0 || !active || (va == 0.f && vb == 0.f)) return;
            const unsigned k = (unsigned)(col - xb);
            const int r = (row == y) ? 0 : 1;
            const bool in_row = rowblock && (row == y || row == other) && k < (unsigned)SSB_RB;
            if (!in_row) {
                unsafeAtomicAdd(&gra_p[o], va);
                if (two) unsafeAtomicAdd(&grb_p[o], vb);
            }
            float* ra = rows + r * (SSB_CH * RS);                            // (the wave's own row buffers: common.h)
            ss::lds_owned_add2(tags + r * RS, k, in_row, ra, va, ra + RS, vb, two);
        };
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            const int yy = min(max(y + kTapDy[t], 0), H - 1), xx = min(max(x + kTapDx[t], 0), W - 1);
            const Taps4 tp = bilinear_taps(pred0[b * plane + (long long)yy * W + xx], y, x, H, W, half_w, half_h);
            if (t == 0) other = (tp.iy == y) ? y + 1 : tp.iy;
            const float dc = dcorr_in[(b * 5 + t) * plane + pix];
            const float a0 = (tp.o_nw >= 0) ? rpa[tp.o_nw] : 0.f, b0 = (tp.o_ne >= 0) ? rpa[tp.o_ne] : 0.f;
            const float c0v = (tp.o_sw >= 0) ? rpa[tp.o_sw] : 0.f, d0 = (tp.o_se >= 0) ? rpa[tp.o_se] : 0.f;
            const float a1 = (two && tp.o_nw >= 0) ? rpb[tp.o_nw] : 0.f, b1 = (two && tp.o_ne >= 0) ? rpb[tp.o_ne] : 0.f;
            const float c1v = (two && tp.o_sw >= 0) ? rpb[tp.o_sw] : 0.f, d1 = (two && tp.o_se >= 0) ? rpb[tp.o_se] : 0.f;
            gla += dc * (a0 * tp.w_nw + b0 * tp.w_ne + c0v * tp.w_sw + d0 * tp.w_se);
            glb += dc * (a1 * tp.w_nw + b1 * tp.w_ne + c1v * tp.w_sw + d1 * tp.w_se);
            const float ga = dc * la, gb = dc * lb;
            if (g_right) {
                if (tp.w_nw != 0.f) scatter(tp.o_nw, tp.iy, tp.ix, ga * tp.w_nw, gb * tp.w_nw);
                if (tp.w_ne != 0.f) scatter(tp.o_ne, tp.iy, tp.ix + 1, ga * tp.w_ne, gb * tp.w_ne);
                if (tp.w_sw != 0.f) scatter(tp.o_sw, tp.iy + 1, tp.ix, ga * tp.w_sw, gb * tp.w_sw);
                if (tp.w_se != 0.f) scatter(tp.o_se, tp.iy + 1, tp.ix + 1, ga * tp.w_se, gb * tp.w_se);
            }
            gix[t] = ga * ((b0 - a0) * tp.fs + (d0 - c0v) * tp.fn) + gb * ((b1 - a1) * tp.fs + (d1 - c1v) * tp.fn);
        }
        if (g_left && active) {
            g_left[ca + pix] = gla;
            if (two) g_left[cb + pix] = glb;
        }
        if (rowblock) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int row = r == 0 ? y : other;                          // (both the same in every lane of the wave)
                if (row < 0 || row >= H) continue;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int idx = lane + 64 * k, col = xb + idx;
                    const float va = ss::lds_get(&rows[(r * SSB_CH) * RS + idx]), vb = ss::lds_get(&rows[(r * SSB_CH + 1) * RS + idx]);
                    if (idx < SSB_RB && col >= 0 && col < W) {
                        if (va != 0.f) unsafeAtomicAdd(&gra_p[(long long)row * W + col], va);
                        if (two && vb != 0.f) unsafeAtomicAdd(&grb_p[(long long)row * W + col], vb);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 5; ++t) gred[wave][t][lane] = gix[t];
    __syncthreads();
    if (wave == 0 && g_pred0 && active) {
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            const int yy = min(max(y + kTapDy[t], 0), H - 1), xx = min(max(x + kTapDx[t], 0), W - 1);
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) s += gred[w][t][lane];
            unsafeAtomicAdd(&g_pred0[b * plane + (long long)yy * W + xx], -((half_w * s) / half_w));
        }
    }
}

// DETERMINISM: the three backward kernels of this file scatter into neighbouring pixels (the 5 replicate-padded taps
// of Propagation / Propagation_prob, the bilinear taps of the probe) and into per-block partial sums with fp32 hardware atomics
// (unsafeAtomicAdd), like the backward of SpatialTransformer_grid (warp.hip) and ATen's own grid_sampler / replication_pad
// backward on GPUs: TRAINING gradients of the attention tail are therefore reproducible to rounding (~1e-7 relative), not bit
// for bit from run to run -- unlike every forward of this library, which is bit-exact and batch-invariant.  A gather form
// (each pixel reading its inverse neighbours) would be deterministic at ~5x the loads; not built.
// ---- :295-310 ---------------------------------------------------------------------------------------------------------------------
// One thread per pixel; LDS: aw[D][T] (the strength-weighted propagated logits, then their gradient).  `samples` [B,K,H,W] are the
// forward's selected disparities (ascending), so the selection itself is not repeated.
__global__ void topk_candidates_bwd_kernel(const float* __restrict__ logits, const float* __restrict__ strength,
                                           const float* __restrict__ samples, const float* __restrict__ g_att,
                                           const float* __restrict__ g_pred, float* __restrict__ g_logits, float* __restrict__ g_strength,
                                           int D, int H, int W, int K, int dmin, long long total) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int T = blockDim.x, tid = threadIdx.x;
    float* aw = lds;                    // [D][T]
    const long long i = blockIdx.x * (long long)T + tid;
    if (i >= total) return;             // no barriers below: threads are independent
    const long long plane = (long long)H * W;
    const int x = (int)(i % W), y = (int)((i / W) % H);
    const long long b = i / plane, pix = (long long)y * W + x;
    long long nb[5];
    float st[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        const int yy = min(max(y + kTapDy[t], 0), H - 1), xx = min(max(x + kTapDx[t], 0), W - 1);
        nb[t] = (long long)yy * W + xx;
        st[t] = strength[(b * 5 + t) * plane + pix];
    }
    const float* lg = logits + b * D * plane;
    float mx = -INFINITY;
    for (int k = 0; k < D; ++k) {
        float a = 0.f;
#pragma unroll
        for (int t = 0; t < 5; ++t) a += st[t] * lg[k * plane + nb[t]];
        aw[k * T + tid] = a;
        mx = fmaxf(mx, a);
    }
    float sum = 0.f;
    for (int k = 0; k < D; ++k) sum += expf(aw[k * T + tid] - mx);
    // the selected planes: probabilities, the soft-argmax over them, and the two dot products of the softmax backward
    float mx2 = -INFINITY;
    for (int j = 0; j < K; ++j) mx2 = fmaxf(mx2, aw[min(max((int)samples[(b * K + j) * plane + pix] - dmin, 0), D - 1) * T + tid]);
    float sum2 = 0.f, e1 = 0.f, dotP = 0.f;
    for (int j = 0; j < K; ++j) {
        const int k = min(max((int)samples[(b * K + j) * plane + pix] - dmin, 0), D - 1);    // (clamped: `samples` is caller data, the index goes into LDS)
        const float e = expf(aw[k * T + tid] - mx2);
        sum2 += e;
        e1 += e * (float)(dmin + k);
        if (g_att) dotP += expf(aw[k * T + tid] - mx) / sum * g_att[(b * K + j) * plane + pix];
    }
    const float pred = e1 / sum2, gp = g_pred ? g_pred[i] : 0.f;
    // d aw[k] = -prob[k] * dotP for every plane; the selected ones add prob * g_att and att_prob * (v - pred) * g_pred
    for (int k = 0; k < D; ++k) {
        const float a = aw[k * T + tid];
        aw[k * T + tid] = -(expf(a - mx) / sum) * dotP;          // (a is lost: the selected planes are handled from `keep` below)
    }
    // second walk over the selected planes needs their ORIGINAL aw: recompute from the logits (24 x 5 loads)
    for (int j = 0; j < K; ++j) {
        const int k = min(max((int)samples[(b * K + j) * plane + pix] - dmin, 0), D - 1);    // (clamped: `samples` is caller data, the index goes into LDS)
        float a = 0.f;
#pragma unroll
        for (int t = 0; t < 5; ++t) a += st[t] * lg[k * plane + nb[t]];
        const float prob = expf(a - mx) / sum, ap = expf(a - mx2) / sum2;
        aw[k * T + tid] += (g_att ? prob * g_att[(b * K + j) * plane + pix] : 0.f) + ap * ((float)(dmin + k) - pred) * gp;
    }
    float gst[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    float* gl = g_logits ? g_logits + b * D * plane : nullptr;
    for (int k = 0; k < D; ++k) {
        const float da = aw[k * T + tid];
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            gst[t] += da * lg[k * plane + nb[t]];
            if (gl) unsafeAtomicAdd(&gl[k * plane + nb[t]], da * st[t]);
        }
    }
    if (g_strength) {
#pragma unroll
        for (int t = 0; t < 5; ++t) g_strength[(b * 5 + t) * plane + pix] = gst[t];
    }
}

}  // namespace

// Backward of ss_upsample_softmax_regression_fwd (models/SemStereo.py:279-285): up [B,1,D,H,W] (the forward's up-sampled logits),
// grad_up [B,1,D,H,W] / grad_disp [B,H,W] / grad_var [B,1,H,W] (each may be NULL) -> grad_coarse [B,1,D/2,H/2,W/2].
// work: B*D*H*W floats of scratch (the gradient of the up-sampled logits).
extern "C" int ss_upsample_softmax_regression_bwd(const float* up, const float* grad_up, const float* grad_disp, const float* grad_var,
                                                  float* grad_coarse, float* work, int B, int dmin, int ndisp, int H, int W,
                                                  ss_stream_t stream) {
    SS_REQUIRE(up && grad_coarse && work && B > 0 && ndisp > 0 && H > 0 && W > 0);
    if ((H & 1) || (W & 1) || (ndisp & 1)) return SS_ERR_UNSUPPORTED;               // exact 2x in every dimension, as the forward
    hipStream_t st = ss::as_stream(stream);
    const long long plane = (long long)H * W, total = (long long)B * plane;
    hipLaunchKernelGGL(upsoft_bwd_fine_kernel, dim3((unsigned)ss::ceil_div_ll(total, 256)), dim3(256), 0, st, up, grad_up, grad_disp, grad_var,
                       work, ndisp, dmin, plane, total);
    const long long ctotal = (long long)B * (ndisp / 2) * (H / 2) * (W / 2);
    hipLaunchKernelGGL(upsoft_bwd_coarse_kernel, dim3((unsigned)ss::ceil_div_ll(ctotal, 256)), dim3(256), 0, st, work, grad_coarse, ndisp / 2,
                       H / 2, W / 2, ctotal);
    return ss::check_launch();
}

// Backward of ss_sample_strength_fwd (models/SemStereo.py:286-293): grad_strength [B,5,H,W] -> grad_left / grad_right [B,C,H,W],
// grad_pred0 [B,H,W], grad_var [B,1,H,W], grad_gamma_beta [2] (any may be NULL; the scattered ones are zeroed here).
extern "C" int ss_sample_strength_bwd(const float* left, const float* right, const float* pred0, const float* var, const float* gamma,
                                      const float* beta, const float* grad_strength, float* grad_left, float* grad_right,
                                      float* grad_pred0, float* grad_var, float* grad_gamma_beta, int B, int C, int H, int W,
                                      ss_stream_t stream) {
    SS_REQUIRE(left && right && pred0 && var && gamma && beta && grad_strength && B > 0 && C > 0 && H > 0 && W > 0);
    hipStream_t st = ss::as_stream(stream);
    const long long plane = (long long)H * W, total = (long long)B * plane;
    if (grad_right && hipMemsetAsync(grad_right, 0, (size_t)B * C * plane * sizeof(float), st) != hipSuccess) return SS_ERR_LAUNCH;
    if (grad_pred0 && hipMemsetAsync(grad_pred0, 0, (size_t)total * sizeof(float), st) != hipSuccess) return SS_ERR_LAUNCH;
    if (grad_var && hipMemsetAsync(grad_var, 0, (size_t)total * sizeof(float), st) != hipSuccess) return SS_ERR_LAUNCH;
    if (grad_gamma_beta && hipMemsetAsync(grad_gamma_beta, 0, 2 * sizeof(float), st) != hipSuccess) return SS_ERR_LAUNCH;
    const float half_w = (float)((W - 1.0) / 2.0), half_h = (float)((H - 1.0) / 2.0);
    hipLaunchKernelGGL(sample_strength_bwd_kernel, dim3((unsigned)ss::ceil_div_ll(total, 64)), dim3(64 * SSB_NW), 0, st, left, right, pred0, var, gamma,
                       beta, grad_strength, grad_left, grad_right, grad_pred0, grad_var, grad_gamma_beta, C, H, W, half_w, half_h, total);
    return ss::check_launch();
}

// Backward of ss_topk_candidates_fwd (models/SemStereo.py:295-310): logits [B,1,D,H,W], strength [B,5,H,W], samples [B,K,H,W] (the
// forward's selection), grad_att_topk [B,1,K,H,W] / grad_pred_att [B,H,W] (may be NULL) -> grad_logits [B,1,D,H,W] (zeroed here),
// grad_strength [B,5,H,W].
extern "C" int ss_topk_candidates_bwd(const float* logits, const float* strength, const float* samples, const float* grad_att_topk,
                                      const float* grad_pred_att, float* grad_logits, float* grad_strength, int B, int dmin, int ndisp,
                                      int H, int W, int k, ss_stream_t stream) {
    SS_REQUIRE(logits && strength && samples && B > 0 && ndisp > 0 && H > 0 && W > 0 && k > 0 && k <= ndisp);
    if (ndisp > 192) return SS_ERR_UNSUPPORTED;
    hipStream_t st = ss::as_stream(stream);
    const long long plane = (long long)H * W, total = (long long)B * plane;
    if (grad_logits && hipMemsetAsync(grad_logits, 0, (size_t)B * ndisp * plane * sizeof(float), st) != hipSuccess) return SS_ERR_LAUNCH;
    const int T = 64;
    const size_t lds = (size_t)ndisp * T * sizeof(float);
    hipLaunchKernelGGL(topk_candidates_bwd_kernel, dim3((unsigned)ss::ceil_div_ll(total, T)), dim3(T), lds, st, logits, strength, samples,
                       grad_att_topk, grad_pred_att, grad_logits, grad_strength, ndisp, H, W, k, dmin, total);
    return ss::check_launch();
}

// The same backward through a scratch of B * 5 * H * W floats (`work`): two launches with the channels on the second one's grid (r06).
extern "C" int ss_sample_strength_bwd_ws(const float* left, const float* right, const float* pred0, const float* var, const float* gamma,
                                         const float* beta, const float* grad_strength, float* grad_left, float* grad_right,
                                         float* grad_pred0, float* grad_var, float* grad_gamma_beta, float* work, int B, int C, int H, int W,
                                         ss_stream_t stream) {
    SS_REQUIRE(left && right && pred0 && var && gamma && beta && grad_strength && work && B > 0 && C > 0 && H > 0 && W > 0);
    hipStream_t st = ss::as_stream(stream);
    const long long plane = (long long)H * W, total = (long long)B * plane;
    if (ss::ceil_div(C, SSB_NW * SSB_CH) > 65535) return SS_ERR_UNSUPPORTED;
    if (grad_right && hipMemsetAsync(grad_right, 0, (size_t)B * C * plane * sizeof(float), st) != hipSuccess) return SS_ERR_LAUNCH;
    if (grad_pred0 && hipMemsetAsync(grad_pred0, 0, (size_t)total * sizeof(float), st) != hipSuccess) return SS_ERR_LAUNCH;
    if (grad_var && hipMemsetAsync(grad_var, 0, (size_t)total * sizeof(float), st) != hipSuccess) return SS_ERR_LAUNCH;
    if (grad_gamma_beta && hipMemsetAsync(grad_gamma_beta, 0, 2 * sizeof(float), st) != hipSuccess) return SS_ERR_LAUNCH;
    const float half_w = (float)((W - 1.0) / 2.0), half_h = (float)((H - 1.0) / 2.0);
    const unsigned gx = (unsigned)ss::ceil_div_ll(total, 64);
    hipLaunchKernelGGL(ssb_dcorr_kernel, dim3(gx), dim3(64 * SSB_NW), 0, st, left, right, pred0, var, gamma, beta, grad_strength, work, grad_var,
                       grad_gamma_beta, C, H, W, half_w, half_h, total);
    if (grad_left || grad_right || grad_pred0)
        hipLaunchKernelGGL(ssb_scatter_kernel, dim3(gx, ss::ceil_div(C, SSB_NW * SSB_CH)), dim3(64 * SSB_NW), 0, st, left, right, pred0, work, grad_left,
                           grad_right, grad_pred0, C, H, W, half_w, half_h, total);
    return ss::check_launch();
}

