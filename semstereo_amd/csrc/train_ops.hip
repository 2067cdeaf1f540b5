// Training-side kernels of the 3-D aggregation stack (main_us3d.py:186-222 back-propagates through every module of
// models/SemStereo.py:273-323): what the matrix-core forward / dgrad / wgrad kernels (conv3d_bf16s.hip, deconv3d_bf16s.hip,
// conv3d_wgrad.hip) leave over -- BatchNorm with BATCH statistics (+ ReLU) forward and backward, the weight gradient of the
// depthwise `patch` convolution, the logits gradient of the channelAtt gate, per-channel sums (bias gradients), and the
// backward of the windowed attention core.  All HBM-bound element-wise / reduction kernels in fp32 with float64 reduction
// totals (hardware f64 atomics), laid out for coalesced 16-byte accesses along W.
#include <algorithm>

#include "common.h"

namespace {

// ---- per-channel reductions over [B, C, N] (N = D*H*W) -----------------------------------------------------------------------
// block (chunk, c): sums of a slice of the channel's B*N elements; 256 threads x float4; totals in double through atomics
// sums[c*NS + k]; MODE 0: (x, x^2)   MODE 1: (g', g' * xhat) with g' = g * (y > 0 if relu), xhat = (x - mean) * invstd
//                MODE 2: (g)         MODE 3: (g * cv summed over D for the gate: see gate_bwd_logits instead)
template <int MODE>
__global__ __launch_bounds__(256) void channel_reduce_kernel(const float* __restrict__ a, const float* __restrict__ x,
                                                              const float* __restrict__ y, const float* __restrict__ mean,
                                                              const float* __restrict__ invstd, double* __restrict__ sums, int C,
                                                              long long N, long long per_block, int relu,
                                                              const float* __restrict__ w = nullptr, const float* __restrict__ bias = nullptr) {
    const int c = blockIdx.y;
    const long long total = (long long)gridDim.z * N;       // (unused: batch is blockIdx.z)
    (void)total;
    const int b = blockIdx.z;
    const long long base = ((long long)b * C + c) * N;
    const long long i0 = (long long)blockIdx.x * per_block, i1 = min(i0 + per_block, N);
    float s0 = 0.f, s1 = 0.f;
    const float mu = (MODE == 1) ? mean[c] : 0.f, is = (MODE == 1) ? invstd[c] : 0.f;
    // (r06) y == NULL with relu: the ReLU mask from x itself -- the forward's own fmaf(x, sc, sh) > 0 -- instead of a third tensor read
    const bool remask = MODE == 1 && relu && y == nullptr;
    const float msc = remask ? is * (w ? w[c] : 1.f) : 0.f, msh = remask ? fmaf(-mu, msc, bias ? bias[c] : 0.f) : 0.f;
    // (r06) 16 bytes per lane where the channel rows allow it (N % 4 == 0; per_block is then a multiple of 4): the scalar loop ran at
    // ~2 TB/s of traffic, 4.4 ms of the 1024^2 training step in the four BatchNorm passes
    const uintptr_t bits = reinterpret_cast<uintptr_t>(a) | (MODE == 1 ? reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) : 0);   // (NULL y: no bits)
    if ((N & 3) == 0 && (per_block & 3) == 0 && (bits & 15) == 0) {
        for (long long i = i0 + 4 * threadIdx.x; i < i1; i += 1024) {
            const float4 av = *reinterpret_cast<const float4*>(a + base + i);
            const float ae[4] = {av.x, av.y, av.z, av.w};
            if (MODE == 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { s0 += ae[e]; s1 += ae[e] * ae[e]; }
            } else if (MODE == 1) {
                const float4 xv = *reinterpret_cast<const float4*>(x + base + i);
                const float xe[4] = {xv.x, xv.y, xv.z, xv.w};
                float ye[4] = {1.f, 1.f, 1.f, 1.f};
                if (remask) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) ye[e] = fmaf(xe[e], msc, msh);
                } else if (relu) {
                    const float4 yv = *reinterpret_cast<const float4*>(y + base + i);
                    ye[0] = yv.x; ye[1] = yv.y; ye[2] = yv.z; ye[3] = yv.w;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float g = (relu && !(ye[e] > 0.f)) ? 0.f : ae[e];
                    s0 += g; s1 += g * ((xe[e] - mu) * is);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) s0 += ae[e];
            }
        }
    } else
    for (long long i = i0 + threadIdx.x; i < i1; i += 256) {
        if (MODE == 0) {
            const float v = a[base + i];
            s0 += v; s1 += v * v;
        } else if (MODE == 1) {
            float g = a[base + i];
            if (relu && !((remask ? fmaf(x[base + i], msc, msh) : y[base + i]) > 0.f)) g = 0.f;
            s0 += g; s1 += g * ((x[base + i] - mu) * is);
        } else {
            s0 += a[base + i];
        }
    }
    __shared__ double red[2][4];
    double d0 = s0, d1 = s1;
    for (int o = 32; o > 0; o >>= 1) {
        d0 += __shfl_xor(d0, o);
        d1 += __shfl_xor(d1, o);
    }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = d0; red[1][threadIdx.x >> 6] = d1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        constexpr int NS = (MODE == 2) ? 1 : 2;
        atomicAdd(&sums[c * NS], red[0][0] + red[0][1] + red[0][2] + red[0][3]);
        if (NS == 2) atomicAdd(&sums[c * NS + 1], red[1][0] + red[1][1] + red[1][2] + red[1][3]);
    }
}

// sums (x, x^2) -> mean, invstd (biased variance, as F.batch_norm normalises with), var_unbiased for the running statistic
// ... and, when given, the module's running statistics as F.batch_norm moves them -- running = running * (1 - momentum) + momentum *
// batch value, the variance unbiased -- and its num_batches_tracked (r06: five element-wise PyTorch launches per layer and step)
__global__ void bn_finish_stats_kernel(const double* __restrict__ sums, float* __restrict__ mean, float* __restrict__ invstd,
                                       float* __restrict__ var_unbiased, int C, double count, float eps, float* __restrict__ run_mean,
                                       float* __restrict__ run_var, long long* __restrict__ batches_tracked, float mom, float keep) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double m = sums[2 * c] / count;
    const double v = fmax(sums[2 * c + 1] / count - m * m, 0.0);
    const float mf = (float)m, vu = (float)(count > 1.0 ? v * count / (count - 1.0) : v);
    mean[c] = mf;
    invstd[c] = (float)(1.0 / sqrt(v + (double)eps));
    var_unbiased[c] = vu;
    if (run_mean) run_mean[c] = fmaf(mom, mf, __fmul_rn(run_mean[c], keep));
    if (run_var) run_var[c] = fmaf(mom, vu, __fmul_rn(run_var[c], keep));
    if (c == 0 && batches_tracked) batches_tracked[0] += 1;
}

// y = (x - mean) * invstd * w + b  [+ res]  [ReLU]      (res: the skip branch of the hourglass, models/SemStereo.py:141-142)
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                        const float* __restrict__ mean,
                                                        const float* __restrict__ invstd, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ y, int C, long long N,
                                                        long long total, int relu) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)((i / N) % C);
    const float sc = invstd[c] * (w ? w[c] : 1.f), sh = fmaf(-mean[c], sc, bias ? bias[c] : 0.f);
    float v = fmaf(x[i], sc, sh);             // (the backward recomputes this very expression for the ReLU mask when it is not given y)
    if (res) v += res[i];
    if (relu) v = fmaxf(v, 0.f);
    y[i] = v;
}

// the same with 16 bytes per lane and the channel from the grid (blockIdx.y = channel, blockIdx.z = batch element; N % 4 == 0)
__global__ __launch_bounds__(256) void bn_apply_v4_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ w, const float* __restrict__ bias,
                                                           float* __restrict__ y, int C, long long N, int relu) {
    const int c = blockIdx.y;
    const long long base = ((long long)blockIdx.z * C + c) * N;
    const float sc = invstd[c] * (w ? w[c] : 1.f), sh = fmaf(-mean[c], sc, bias ? bias[c] : 0.f);
    for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < N; i += (long long)gridDim.x * 1024) {
        const float4 xv = *reinterpret_cast<const float4*>(x + base + i);
        float v[4] = {fmaf(xv.x, sc, sh), fmaf(xv.y, sc, sh), fmaf(xv.z, sc, sh), fmaf(xv.w, sc, sh)};
        if (res) {
            const float4 rv = *reinterpret_cast<const float4*>(res + base + i);
            v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
        }
        if (relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        *reinterpret_cast<float4*>(y + base + i) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_v4_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                               const float* __restrict__ y, const float* __restrict__ mean,
                                                               const float* __restrict__ invstd, const float* __restrict__ w,
                                                               const double* __restrict__ sums, float* __restrict__ dx,
                                                               float* __restrict__ dres, int C, long long N, double count, int relu,
                                                               float* __restrict__ gw, float* __restrict__ gb, const float* __restrict__ bias) {
    const int c = blockIdx.y;
    const long long base = ((long long)blockIdx.z * C + c) * N;
    const float mu = mean[c], is = invstd[c], ws = (w ? w[c] : 1.f) * is;
    const bool remask = relu && y == nullptr;                // (the mask from x: see channel_reduce_kernel)
    const float msh = fmaf(-mu, ws, bias ? bias[c] : 0.f);
    const float mg = (float)(sums[2 * c] / count), mgx = (float)(sums[2 * c + 1] / count);
    if (blockIdx.x == 0 && blockIdx.z == 0 && threadIdx.x == 0) {      // the parameter gradients as floats (r06: two strided copies per layer)
        if (gb) gb[c] = (float)sums[2 * c];
        if (gw) gw[c] = (float)sums[2 * c + 1];
    }
    for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < N; i += (long long)gridDim.x * 1024) {
        const float4 gq = *reinterpret_cast<const float4*>(g + base + i);
        const float4 xq = *reinterpret_cast<const float4*>(x + base + i);
        float gv[4] = {gq.x, gq.y, gq.z, gq.w};
        const float xe[4] = {xq.x, xq.y, xq.z, xq.w};
        if (remask) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (!(fmaf(xe[e], ws, msh) > 0.f)) gv[e] = 0.f;
        } else if (relu) {
            const float4 yq = *reinterpret_cast<const float4*>(y + base + i);
            const float ye[4] = {yq.x, yq.y, yq.z, yq.w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (!(ye[e] > 0.f)) gv[e] = 0.f;
        }
        if (dres) *reinterpret_cast<float4*>(dres + base + i) = make_float4(gv[0], gv[1], gv[2], gv[3]);
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xh = (xe[e] - mu) * is;
            o[e] = ws * (gv[e] - mg - xh * mgx);
        }
        *reinterpret_cast<float4*>(dx + base + i) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// dx = w * invstd * (g' - sum_g / M - xhat * sum_gx / M),  g' = g * (y > 0 if relu);  M = elements per channel
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                            const float* __restrict__ y, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, const float* __restrict__ w,
                                                            const double* __restrict__ sums, float* __restrict__ dx,
                                                            float* __restrict__ dres, int C,
                                                            long long N, long long total, double count, int relu,
                                                            float* __restrict__ gw, float* __restrict__ gb, const float* __restrict__ bias) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    if (i < C) {                                             // (total = B * C * N >= C)
        if (gb) gb[i] = (float)sums[2 * i];
        if (gw) gw[i] = (float)sums[2 * i + 1];
    }
    const int c = (int)((i / N) % C);
    float gv = g[i];
    if (relu) {
        const float sc = invstd[c] * (w ? w[c] : 1.f);
        const float pre = (y != nullptr) ? y[i] : fmaf(x[i], sc, fmaf(-mean[c], sc, bias ? bias[c] : 0.f));
        if (!(pre > 0.f)) gv = 0.f;
    }
    if (dres) dres[i] = gv;                   // the residual's gradient: the incoming one behind the ReLU mask
    const float xh = (x[i] - mean[c]) * invstd[c];
    const float mg = (float)(sums[2 * c] / count), mgx = (float)(sums[2 * c + 1] / count);
    dx[i] = (w ? w[c] : 1.f) * invstd[c] * (gv - mg - xh * mgx);
}

// ---- depthwise (1,3,3) `patch` convolution: weight gradient dW[c, kh, kw] = sum g[b,c,d,h,w] * x[b,c,d,h+kh-1,w+kw-1] ----------
__global__ __launch_bounds__(256) void depthwise_patch_wgrad_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                                     float* __restrict__ dw, int C, int D, int H, int W,
                                                                     int rows_per_block) {
    const int c = blockIdx.y, b = blockIdx.z;
    const long long base = ((long long)b * C + c) * D * H * W;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(r0 + rows_per_block, D * H);
    float acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = 0.f;
    // (the threads run over the block's rows x W positions together: with one row at a time and W = 128 -- the 1/8-scale volume of the
    // 1024^2 pair -- half of them had nothing to do, r06)
    const int npos = (r1 - r0) * W;
    for (int idx = threadIdx.x; idx < npos; idx += 256) {
        const int r = r0 + idx / W, w = idx % W;
        const int d = r / H, h = r - d * H;
        const float gv = g[base + ((long long)d * H + h) * W + w];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int hh = h + kh - 1;
            if ((unsigned)hh >= (unsigned)H) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int ww = w + kw - 1;
                if ((unsigned)ww < (unsigned)W) acc[kh * 3 + kw] += gv * x[base + ((long long)d * H + hh) * W + ww];
            }
        }
    }
    __shared__ float red[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        float v = acc[t];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if ((threadIdx.x & 63) == 0) red[t][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x < 9) unsafeAtomicAdd(&dw[c * 9 + threadIdx.x], red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3]);
}

// ---- channelAtt gate out = sigmoid(a)[b,c,h,w] * cv[b,c,d,h,w]: d_a = s (1 - s) * sum_d g * cv ----------------------------------
__global__ __launch_bounds__(256) void gate_bwd_logits_kernel(const float* __restrict__ g, const float* __restrict__ cv,
                                                               const float* __restrict__ a, float* __restrict__ da, int D,
                                                               long long plane, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;          // over [B*C, H*W]
    if (i >= total) return;
    const long long bc = i / plane, p = i - bc * plane;
    const float* gp = g + bc * D * plane + p;
    const float* cp = cv + bc * D * plane + p;
    float s = 0.f;
    for (int d = 0; d < D; ++d) s += gp[d * plane] * cp[d * plane];
    const float sg = 1.0f / (1.0f + expf(-a[i]));
    da[i] = sg * (1.0f - sg) * s;
}

// ---- windowed attention core, backward (models/submodule_other.py:805-834) ------------------------------------------------------
// qkv [B,3C,D,H,W] (C = heads * 8), gy [B,C,D,H,W] = gradient of y = softmax(q k^T * scale [+ pad mask]) v per (window, head),
// un-partitioned.  -> gqkv [B,3C,D,H,W].  One workgroup per (window, head): q, k, v, gy tiles [T][8] and the probabilities
// [T][T] (then, in place, the logits' gradient) in LDS.
// Volumes whose H, W are not window multiples (r04): the reference zero-pads the volume BEFORE the qkv Linear
// (models/submodule_other.py:808-813), so a pad token's q / k / v are the Linear's bias -- `bqkv` [3C] here, the qkv tensor
// holds real positions only -- its output is cropped (gradient 0), and a logit between a pad and a real token gets -1000 when
// BOTH H and W were padded (`mask_on`; the reference's `-0:` slices mark every token when only one was, :822-823: no mask).
// What reaches a pad token's q / k / v is the bias's gradient: accumulated into gbias [3C] (fp32 atomics).
// (r06: one [T][T] array instead of two -- 51 KB of LDS, three workgroups per CU instead of one -- and the soft-max rows spread over the
// lanes: a half-wave owns a row, its lanes the columns j = l, l + 32, ...; the row's probabilities, dP = go v^T and their dot stay in
// registers between the passes.  The first form walked each row with ONE thread, 96 of 256 threads, four serial passes: 509 us at the
// 1024^2 / md64 training shape against 46 for the forward.)
template <int T>
__global__ __launch_bounds__(256) void window_attention_core_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ bqkv,
                                                                         const float* __restrict__ gy, float* __restrict__ gqkv,
                                                                         float* __restrict__ gbias, int C, int D, int H, int W,
                                                                         int heads, int bd, int bh, int bw, float scale, int mask_on) {
    constexpr int HD = 8, NJ = T / 32, ROWS = T / 8;          // columns per lane of a row; rows per half-wave (8 half-waves)
    static_assert(T % 32 == 0, "a row is whole 32-lane passes");
    extern __shared__ __attribute__((aligned(16))) float sm_att[];          // 4 x [T][9] + [T][T + 1] floats (51 KB at T = 96)
    float (*q)[HD + 1] = reinterpret_cast<float (*)[HD + 1]>(sm_att);
    float (*k)[HD + 1] = q + T;
    float (*v)[HD + 1] = k + T;
    float (*go)[HD + 1] = v + T;
    float (*A)[T + 1] = reinterpret_cast<float (*)[T + 1]>(sm_att + 4 * T * (HD + 1));     // P, then dS in place
    __shared__ unsigned char is_pad[T];
    const int head = blockIdx.y, b = blockIdx.z;
    const int nw = (W + bw - 1) / bw, nh = (H + bh - 1) / bh;
    int win = blockIdx.x;
    const int ww = win % nw; win /= nw;
    const int wh = win % nh; win /= nh;
    const int wd = win;
    const long long plane = (long long)H * W, vol = (long long)D * plane;
    auto pos = [&](int t, bool& inside) {         // token t of the window -> flat spatial offset (d, h, w order as the partition)
        const int tw = t % bw, th = (t / bw) % bh, td = t / (bw * bh);
        const int gh = wh * bh + th, gw = ww * bw + tw;
        inside = gh < H && gw < W;
        return (long long)(wd * bd + td) * plane + (long long)gh * W + gw;
    };
    const float* qb = qkv + (long long)b * 3 * C * vol;
    // (token-major over the lanes: consecutive lanes read consecutive tokens of one channel plane)
    for (int e = threadIdx.x; e < T * HD; e += 256) {
        const int t = e % T, j = e / T;
        bool in;
        const long long p = pos(t, in);
        const int ch = head * HD + j;
        q[t][j] = in ? qb[(long long)ch * vol + p] : bqkv[ch];
        k[t][j] = in ? qb[(long long)(C + ch) * vol + p] : bqkv[C + ch];
        v[t][j] = in ? qb[(long long)(2 * C + ch) * vol + p] : bqkv[2 * C + ch];
        go[t][j] = in ? gy[((long long)b * C + ch) * vol + p] : 0.f;
        if (j == 0) is_pad[t] = in ? 0 : 1;
    }
    __syncthreads();
    // pass A, row by row: P = softmax(q k^T * scale + mask), dP = go v^T, dot = sum_j P dP
    const int l = threadIdx.x & 31, hw = threadIdx.x >> 5;
    float pr[ROWS][NJ], dpr[ROWS][NJ], dot[ROWS];
#pragma unroll
    for (int it = 0; it < ROWS; ++it) {
        const int i = hw + 8 * it;
        float qi[HD], gi[HD];
#pragma unroll
        for (int c = 0; c < HD; ++c) { qi[c] = q[i][c]; gi[c] = go[i][c]; }
        float sv[NJ], mx = -INFINITY;
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
            const int j = l + 32 * jj;
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int c = 0; c < HD; ++c) { s += qi[c] * k[j][c]; dp += gi[c] * v[j][c]; }
            sv[jj] = s * scale + ((mask_on && is_pad[i] != is_pad[j]) ? -1000.0f : 0.f);
            dpr[it][jj] = dp;
            mx = fmaxf(mx, sv[jj]);
        }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        float sum = 0.f;
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) { sv[jj] = expf(sv[jj] - mx); sum += sv[jj]; }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        float dt = 0.f;
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
            const float pv = sv[jj] / sum;
            pr[it][jj] = pv;
            A[i][l + 32 * jj] = pv;
            dt += pv * dpr[it][jj];
        }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) dt += __shfl_xor(dt, o);
        dot[it] = dt;
    }
    __syncthreads();
    // dv = P^T go (P column-wise)
    float* gb = gqkv + (long long)b * 3 * C * vol;
    for (int e = threadIdx.x; e < T * HD; e += 256) {
        const int t = e % T, c = e / T;
        float dv = 0.f;
        for (int j = 0; j < T; ++j) dv += A[j][t] * go[j][c];
        bool in;
        const long long p = pos(t, in);
        const int ch = head * HD + c;
        if (in) gb[(long long)(2 * C + ch) * vol + p] = dv;
        else if (gbias != nullptr) unsafeAtomicAdd(&gbias[2 * C + ch], dv);
    }
    __syncthreads();
    // pass B, in place: dS = P (dP - dot) * scale        (the gradient of the logits before the scale)
#pragma unroll
    for (int it = 0; it < ROWS; ++it) {
        const int i = hw + 8 * it;
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) A[i][l + 32 * jj] = pr[it][jj] * (dpr[it][jj] - dot[it]) * scale;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < T * HD; e += 256) {
        const int t = e % T, c = e / T;
        float dq = 0.f, dk = 0.f;
        for (int j = 0; j < T; ++j) {
            dq += A[t][j] * k[j][c];
            dk += A[j][t] * q[j][c];
        }
        bool in;
        const long long p = pos(t, in);
        const int ch = head * HD + c;
        if (in) {
            gb[(long long)ch * vol + p] = dq;
            gb[(long long)(C + ch) * vol + p] = dk;
        } else if (gbias != nullptr) {
            unsafeAtomicAdd(&gbias[ch], dq);
            unsafeAtomicAdd(&gbias[C + ch], dk);
        }
    }
}

int reduce_grid(long long N, long long& per_block) {
    long long blocks = (N + 16383) / 16384;                  // ~64 elements per thread
    if (blocks > 65535) blocks = 65535;
    per_block = (N + blocks - 1) / blocks;
    per_block = (per_block + 3) / 4 * 4;                     // (whole 16-byte words: the vector path of the kernels)
    return (int)((N + per_block - 1) / per_block);
}
// element-wise BatchNorm passes with 16 bytes per lane: N % 4 == 0 (every tensor's base is 16-byte aligned: a caching-allocator block)
inline bool vec4_ok(long long N, const void* a, const void* b_, const void* c_, const void* d, const void* e) {
    return (N & 3) == 0 && ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b_) | reinterpret_cast<uintptr_t>(c_) |
                             reinterpret_cast<uintptr_t>(d) | reinterpret_cast<uintptr_t>(e)) & 15) == 0;
}
inline int apply_grid(long long N) { return (int)std::min<long long>(std::max<long long>(1, (N + 4095) / 4096), 65535); }    // ~16 elements per thread

}  // namespace

// BatchNorm with batch statistics (training, models/submodule_other.py:845-848 / models/submodule.py:89-116) + optional ReLU.
// x [B,C,N] (N = D*H*W or H*W), weight / bias [C] (may be NULL) -> y, and mean / invstd / var_unbiased [C] for the backward and
// the running statistics; work: 2*C doubles of scratch.
static int batchnorm_train_fwd_impl(const float* x, const float* residual, const float* weight, const float* bias, float* y, float* mean,
                                    float* invstd, float* var_unbiased, double* work, int B, int C, long long N, float eps, int relu,
                                    ss_stream_t stream, float* run_mean = nullptr, float* run_var = nullptr,
                                    long long* batches_tracked = nullptr, double momentum = 0.0) {
    SS_REQUIRE(x && y && mean && invstd && var_unbiased && work && B > 0 && C > 0 && N > 0 && C <= 65535 && B <= 65535);
    hipStream_t st = ss::as_stream(stream);
    if (hipMemsetAsync(work, 0, (size_t)2 * C * sizeof(double), st) != hipSuccess) return SS_ERR_LAUNCH;
    long long per_block;
    const int gx = reduce_grid(N, per_block);
    hipLaunchKernelGGL(channel_reduce_kernel<0>, dim3(gx, C, B), dim3(256), 0, st, x, nullptr, nullptr, nullptr, nullptr, work, C, N, per_block, 0);
    // (the factor of the old value is 1 - momentum formed in double and rounded once, as `running.mul_(1.0 - momentum)` has it)
    hipLaunchKernelGGL(bn_finish_stats_kernel, dim3(ss::ceil_div(C, 64)), dim3(64), 0, st, work, mean, invstd, var_unbiased, C,
                       (double)B * (double)N, eps, run_mean, run_var, batches_tracked, (float)momentum, (float)(1.0 - momentum));
    const long long total = (long long)B * C * N;
    const long long blocks = ss::ceil_div_ll(total, 256);
    if (blocks > 0x7fffffffLL) return SS_ERR_UNSUPPORTED;
    if (vec4_ok(N, x, residual, y, nullptr, nullptr))
        hipLaunchKernelGGL(bn_apply_v4_kernel, dim3(apply_grid(N), C, B), dim3(256), 0, st, x, residual, mean, invstd, weight, bias, y, C, N, relu);
    else
        hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, residual, mean, invstd, weight, bias, y, C, N, total, relu);
    return ss::check_launch();
}

extern "C" int ss_batchnorm_train_fwd(const float* x, const float* weight, const float* bias, float* y, float* mean, float* invstd,
                                      float* var_unbiased, double* work, int B, int C, long long N, float eps, int relu,
                                      ss_stream_t stream) {
    return batchnorm_train_fwd_impl(x, nullptr, weight, bias, y, mean, invstd, var_unbiased, work, B, C, N, eps, relu, stream);
}

// ... with a residual joining BEHIND the normalisation and before the ReLU: y = relu(bn(x) + residual) in the same pass
// (hourglass.forward, models/SemStereo.py:141-142: F.relu(self.conv5(conv4) + self.redir2(conv2)) in train())
extern "C" int ss_batchnorm_train_res_fwd(const float* x, const float* residual, const float* weight, const float* bias, float* y,
                                          float* mean, float* invstd, float* var_unbiased, double* work, int B, int C, long long N,
                                          float eps, int relu, ss_stream_t stream) {
    SS_REQUIRE(residual != nullptr);
    return batchnorm_train_fwd_impl(x, residual, weight, bias, y, mean, invstd, var_unbiased, work, B, C, N, eps, relu, stream);
}

// ... either form (residual may be NULL) that also moves the module's running statistics (running_mean / running_var [C], NULL = leave
// alone) with a FIXED momentum and counts the batch in num_batches_tracked (int64[1] or NULL): nn.BatchNorm's own bookkeeping in the
// statistics kernel instead of five element-wise launches per layer
extern "C" int ss_batchnorm_train_fwd_rs(const float* x, const float* residual, const float* weight, const float* bias, float* y,
                                         float* mean, float* invstd, float* var_unbiased, double* work, float* running_mean,
                                         float* running_var, long long* num_batches_tracked, double momentum, int B, int C,
                                         long long N, float eps, int relu, ss_stream_t stream) {
    SS_REQUIRE(momentum >= 0.0 && momentum <= 1.0);
    return batchnorm_train_fwd_impl(x, residual, weight, bias, y, mean, invstd, var_unbiased, work, B, C, N, eps, relu, stream,
                                    running_mean, running_var, num_batches_tracked, momentum);
}

// Its backward: grad_y, x, y (needed only with relu), mean, invstd, weight -> grad_x [B,C,N], grad_weight / grad_bias [C] as
// doubles in work[2c + 1] / work[2c] (sum g' * xhat, sum g').
static int batchnorm_train_bwd_impl(const float* grad_y, const float* x, const float* y, const float* mean, const float* invstd,
                                    const float* weight, float* grad_x, float* grad_res, double* work, int B, int C, long long N, int relu,
                                    ss_stream_t stream, float* grad_weight = nullptr, float* grad_bias = nullptr,
                                    bool batch_statistics = true, const float* bias = nullptr, bool mask_from_x = false) {
    // (y == NULL with relu: only where the caller says the mask can be recomputed from x -- ss_batchnorm_bwd_pg without a residual)
    SS_REQUIRE(grad_y && x && mean && invstd && grad_x && work && B > 0 && C > 0 && N > 0 && (y || !relu || (mask_from_x && !grad_res)) &&
               C <= 65535 && B <= 65535);
    hipStream_t st = ss::as_stream(stream);
    if (hipMemsetAsync(work, 0, (size_t)2 * C * sizeof(double), st) != hipSuccess) return SS_ERR_LAUNCH;
    long long per_block;
    const int gx = reduce_grid(N, per_block);
    hipLaunchKernelGGL(channel_reduce_kernel<1>, dim3(gx, C, B), dim3(256), 0, st, grad_y, x, y, mean, invstd, work, C, N, per_block, relu, weight,
                       bias);
    const long long total = (long long)B * C * N;
    const long long blocks = ss::ceil_div_ll(total, 256);
    if (blocks > 0x7fffffffLL) return SS_ERR_UNSUPPORTED;
    // (on running statistics the two mean terms of the apply kernel vanish: an infinite element count makes sum / count == 0)
    const double count = batch_statistics ? (double)B * (double)N : (double)INFINITY;
    if (vec4_ok(N, grad_y, x, y, grad_x, grad_res))
        hipLaunchKernelGGL(bn_bwd_apply_v4_kernel, dim3(apply_grid(N), C, B), dim3(256), 0, st, grad_y, x, y, mean, invstd, weight, work, grad_x,
                           grad_res, C, N, count, relu, grad_weight, grad_bias, bias);
    else
        hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, st, grad_y, x, y, mean, invstd, weight, work, grad_x,
                           grad_res, C, N, total, count, relu, grad_weight, grad_bias, bias);
    return ss::check_launch();
}

extern "C" int ss_batchnorm_train_bwd(const float* grad_y, const float* x, const float* y, const float* mean, const float* invstd,
                                      const float* weight, float* grad_x, double* work, int B, int C, long long N, int relu,
                                      ss_stream_t stream) {
    return batchnorm_train_bwd_impl(grad_y, x, y, mean, invstd, weight, grad_x, nullptr, work, B, C, N, relu, stream);
}

// ... of the residual form: also grad_residual [B,C,N] = grad_y behind the ReLU mask
extern "C" int ss_batchnorm_train_res_bwd(const float* grad_y, const float* x, const float* y, const float* mean, const float* invstd,
                                          const float* weight, float* grad_x, float* grad_residual, double* work, int B, int C,
                                          long long N, int relu, ss_stream_t stream) {
    SS_REQUIRE(grad_residual != nullptr);
    return batchnorm_train_bwd_impl(grad_y, x, y, mean, invstd, weight, grad_x, grad_residual, work, B, C, N, relu, stream);
}

// ---- BatchNorm on its RUNNING statistics under autograd (a module in eval() whose inputs or parameters need gradients: fine-tuning
// with frozen statistics; main_us3d.py never does this, PyTorch allows it) ----
// forward: y = (x - mean) * invstd * w + b [+ residual] [ReLU] with the caller's per-channel mean / invstd (= 1 / sqrt(running_var + eps))
extern "C" int ss_batchnorm_eval_fwd(const float* x, const float* residual, const float* mean, const float* invstd, const float* weight,
                                     const float* bias, float* y, int B, int C, long long N, int relu, ss_stream_t stream) {
    SS_REQUIRE(x && y && mean && invstd && B > 0 && C > 0 && N > 0);
    const long long total = (long long)B * C * N;
    const long long blocks = ss::ceil_div_ll(total, 256);
    if (blocks > 0x7fffffffLL) return SS_ERR_UNSUPPORTED;
    if (vec4_ok(N, x, residual, y, nullptr, nullptr) && C <= 65535 && B <= 65535)
        hipLaunchKernelGGL(bn_apply_v4_kernel, dim3(apply_grid(N), C, B), dim3(256), 0, ss::as_stream(stream), x, residual, mean, invstd, weight, bias,
                           y, C, N, relu);
    else
        hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, ss::as_stream(stream), x, residual, mean, invstd, weight, bias, y,
                           C, N, total, relu);
    return ss::check_launch();
}
// backward: the statistics are constants, so grad_x = w * invstd * g' (g' = grad_y behind the ReLU mask; also grad_residual when asked for),
// grad_bias[c] = work[2c] = sum g', grad_weight[c] = work[2c + 1] = sum g' * xhat -- the batch-statistics backward without its two mean terms
extern "C" int ss_batchnorm_eval_bwd(const float* grad_y, const float* x, const float* y, const float* mean, const float* invstd,
                                     const float* weight, float* grad_x, float* grad_residual, double* work, int B, int C, long long N,
                                     int relu, ss_stream_t stream) {
    return batchnorm_train_bwd_impl(grad_y, x, y, mean, invstd, weight, grad_x, grad_residual, work, B, C, N, relu, stream, nullptr, nullptr,
                                    false);
}

// Both backward forms with the parameter gradients ALSO as floats: grad_weight / grad_bias [C] (either may be NULL) beside the doubles in
// `work`; grad_residual may be NULL; batch_statistics = 1: the backward of ss_batchnorm_train_fwd / _res_fwd, 0: of ss_batchnorm_eval_fwd.
// y may be NULL where the forward had a ReLU and NO residual: the mask is then recomputed from x -- fmaf(x, invstd * w, bias - mean *
// invstd * w) > 0, the forward's own expression, with `bias` the forward's (NULL = none) -- and two of the backward's seven tensor
// passes (y in the statistics pass and in the apply pass) are not read
extern "C" int ss_batchnorm_bwd_pg(const float* grad_y, const float* x, const float* y, const float* mean, const float* invstd,
                                   const float* weight, const float* bias, float* grad_x, float* grad_residual, double* work,
                                   float* grad_weight, float* grad_bias, int batch_statistics, int B, int C, long long N, int relu,
                                   ss_stream_t stream) {
    return batchnorm_train_bwd_impl(grad_y, x, y, mean, invstd, weight, grad_x, grad_residual, work, B, C, N, relu, stream, grad_weight,
                                    grad_bias, batch_statistics != 0, bias, true);
}

extern "C" int ss_channel_sum_fwd(const float* a, double* sums, int B, int C, long long N, ss_stream_t stream) {
    SS_REQUIRE(a && sums && B > 0 && C > 0 && N > 0 && C <= 65535 && B <= 65535);
    hipStream_t st = ss::as_stream(stream);
    if (hipMemsetAsync(sums, 0, (size_t)C * sizeof(double), st) != hipSuccess) return SS_ERR_LAUNCH;
    long long per_block;
    const int gx = reduce_grid(N, per_block);
    hipLaunchKernelGGL(channel_reduce_kernel<2>, dim3(gx, C, B), dim3(256), 0, st, a, nullptr, nullptr, nullptr, nullptr, sums, C, N, per_block, 0);
    return ss::check_launch();
}

// weight gradient of the depthwise `patch` Conv3d (kernel (1,3,3), padding (0,1,1), models/SemStereo.py:219): grad_w [C,1,1,3,3]
extern "C" int ss_depthwise_patch_wgrad_fwd(const float* grad_out, const float* in, float* grad_w, int B, int C, int D, int H, int W,
                                            ss_stream_t stream) {
    SS_REQUIRE(grad_out && in && grad_w && B > 0 && C > 0 && D > 0 && H > 0 && W > 0 && C <= 65535 && B <= 65535);
    hipStream_t st = ss::as_stream(stream);
    if (hipMemsetAsync(grad_w, 0, (size_t)C * 9 * sizeof(float), st) != hipSuccess) return SS_ERR_LAUNCH;
    const int rows = D * H, rows_per_block = max(1, ss::ceil_div(rows, 64));
    hipLaunchKernelGGL(depthwise_patch_wgrad_kernel, dim3(ss::ceil_div(rows, rows_per_block), C, B), dim3(256), 0, st, grad_out, in, grad_w,
                       C, D, H, W, rows_per_block);
    return ss::check_launch();
}

// ---- the group normalisation of groupwise_correlation_norm (models/submodule.py:213-222: fea / (torch.norm(fea, 2, 2, True) + 1e-05) over
// each group's channels), once per feature map, forward and backward (training: the volume kernel then runs un-normalised with its own
// backward; r06: torch's norm / add / div and their three backward nodes were 12 launches per step) ----
// one thread per (b, group, pixel): y_c = x_c / (n + eps), n = sqrt(sum_c x_c^2);
// backward: gx_k = gy_k / (n + eps) - x_k * (sum_c gy_c x_c) / ((n + eps)^2 n)   (0 for the second term where n == 0, as torch's norm backward)
template <bool BWD>
__global__ __launch_bounds__(256) void group_normalise_kernel(const float* __restrict__ x, const float* __restrict__ gy, float* __restrict__ out,
                                                               int C, int cg, long long plane, long long total, float eps) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;       // over B * groups * plane
    if (i >= total) return;
    const long long pix = i % plane, bg = i / plane;                      // bg = b * groups + g: channel base bg * cg
    const float* xp = x + bg * cg * plane + pix;
    float ss2 = 0.f, dot = 0.f;
    for (int c = 0; c < cg; ++c) {
        const float v = xp[c * plane];
        ss2 = fmaf(v, v, ss2);
        if (BWD) dot = fmaf(gy[bg * cg * plane + pix + c * plane], v, dot);
    }
    const float n = sqrtf(ss2), inv = 1.0f / (n + eps);
    const float k2 = (BWD && n > 0.f) ? dot * inv * inv / n : 0.f;
    float* op = out + bg * cg * plane + pix;
    for (int c = 0; c < cg; ++c) {
        const float v = xp[c * plane];
        op[c * plane] = BWD ? gy[bg * cg * plane + pix + c * plane] * inv - v * k2 : v * inv;
    }
}

extern "C" int ss_group_normalise_fwd(const float* x, float* y, int B, int C, int H, int W, int groups, float eps, ss_stream_t stream) {
    SS_REQUIRE(x && y && B > 0 && C > 0 && H > 0 && W > 0 && groups > 0 && C % groups == 0);
    const long long plane = (long long)H * W, total = (long long)B * groups * plane, blocks = ss::ceil_div_ll(total, 256);
    if (blocks > 0x7fffffffLL) return SS_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(group_normalise_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, ss::as_stream(stream), x, nullptr, y, C, C / groups, plane,
                       total, eps);
    return ss::check_launch();
}

extern "C" int ss_group_normalise_bwd(const float* grad_y, const float* x, float* grad_x, int B, int C, int H, int W, int groups, float eps,
                                      ss_stream_t stream) {
    SS_REQUIRE(grad_y && x && grad_x && B > 0 && C > 0 && H > 0 && W > 0 && groups > 0 && C % groups == 0);
    const long long plane = (long long)H * W, total = (long long)B * groups * plane, blocks = ss::ceil_div_ll(total, 256);
    if (blocks > 0x7fffffffLL) return SS_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(group_normalise_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, ss::as_stream(stream), x, grad_y, grad_x, C, C / groups,
                       plane, total, eps);
    return ss::check_launch();
}

// channelAtt gate (models/SemStereo.py:101-102), gradient of the logits: grad_att [B,C,H,W] = s (1 - s) * sum_d grad_out * cv
extern "C" int ss_channel_gate_bwd_logits(const float* grad_out, const float* cv, const float* att_logits, float* grad_att, int B, int C,
                                          int D, int H, int W, ss_stream_t stream) {
    SS_REQUIRE(grad_out && cv && att_logits && grad_att && B > 0 && C > 0 && D > 0 && H > 0 && W > 0);
    const long long plane = (long long)H * W, total = (long long)B * C * plane;
    const long long blocks = ss::ceil_div_ll(total, 256);
    if (blocks > 0x7fffffffLL) return SS_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(gate_bwd_logits_kernel, dim3((unsigned)blocks), dim3(256), 0, ss::as_stream(stream), grad_out, cv, att_logits,
                       grad_att, D, plane, total);
    return ss::check_launch();
}

// Backward of ss_window_attention_core_fwd: qkv [B,3C,D,H,W], grad_y [B,C,D,H,W] -> grad_qkv [B,3C,D,H,W].  Windows of 64 or 96
// tokens, 8 channels per head.  H, W not multiples of the window: `bqkv` [3C] (the qkv Linear's bias = the pad tokens' q / k / v)
// and `grad_bias` [3C] (zeroed here; what reaches the pad tokens) are required.
static int window_attention_core_bwd_impl(const float* qkv, const float* bqkv, const float* grad_y, float* grad_qkv, float* grad_bias, int B,
                                          int C, int D, int H, int W, int heads, int bd, int bh, int bw, ss_stream_t stream) {
    SS_REQUIRE(qkv && grad_y && grad_qkv && B > 0 && C > 0 && D > 0 && H > 0 && W > 0 && heads > 0 && bd > 0 && bh > 0 && bw > 0);
    const bool padded = (H % bh) != 0 || (W % bw) != 0;
    SS_REQUIRE(!padded || (bqkv != nullptr && grad_bias != nullptr));
    if (C != heads * 8 || D % bd || B > 65535 || heads > 65535) return SS_ERR_UNSUPPORTED;
    const int T = bd * bh * bw;
    const long long windows = (long long)(D / bd) * ss::ceil_div(H, bh) * ss::ceil_div(W, bw);
    if (windows > 0x7fffffffLL) return SS_ERR_UNSUPPORTED;
    const float scale = 1.0f / sqrtf(8.0f);
    hipStream_t st = ss::as_stream(stream);
    if (grad_bias != nullptr && hipMemsetAsync(grad_bias, 0, (size_t)3 * C * sizeof(float), st) != hipSuccess) return SS_ERR_LAUNCH;
    const int mask_on = ((H % bh) != 0 && (W % bw) != 0) ? 1 : 0;       // the reference masks only when BOTH were padded (see the kernel)
    const dim3 grid((unsigned)windows, heads, B);
    const size_t lds = (size_t)(4 * T * 9 + T * (T + 1)) * sizeof(float);
    if (T == 64) {
        auto kern = window_attention_core_bwd_kernel<64>;
        if (ss::ensure_dynamic_lds(reinterpret_cast<const void*>(kern), (int)lds) != SS_OK) return SS_ERR_LAUNCH;
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, qkv, bqkv, grad_y, grad_qkv, grad_bias, C, D, H, W, heads, bd, bh, bw, scale, mask_on);
    } else if (T == 96) {
        auto kern = window_attention_core_bwd_kernel<96>;
        if (ss::ensure_dynamic_lds(reinterpret_cast<const void*>(kern), (int)lds) != SS_OK) return SS_ERR_LAUNCH;
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, qkv, bqkv, grad_y, grad_qkv, grad_bias, C, D, H, W, heads, bd, bh, bw, scale, mask_on);
    } else {
        return SS_ERR_UNSUPPORTED;
    }
    return ss::check_launch();
}

extern "C" int ss_window_attention_core_bwd(const float* qkv, const float* grad_y, float* grad_qkv, int B, int C, int D, int H, int W,
                                            int heads, int bd, int bh, int bw, ss_stream_t stream) {
    if (H > 0 && W > 0 && bh > 0 && bw > 0 && (H % bh || W % bw)) return SS_ERR_UNSUPPORTED;      // pad tokens: the _pad form
    return window_attention_core_bwd_impl(qkv, nullptr, grad_y, grad_qkv, nullptr, B, C, D, H, W, heads, bd, bh, bw, stream);
}

extern "C" int ss_window_attention_core_pad_bwd(const float* qkv, const float* bqkv, const float* grad_y, float* grad_qkv, float* grad_bias,
                                                int B, int C, int D, int H, int W, int heads, int bd, int bh, int bw, ss_stream_t stream) {
    SS_REQUIRE(bqkv && grad_bias);
    return window_attention_core_bwd_impl(qkv, bqkv, grad_y, grad_qkv, grad_bias, B, C, D, H, W, heads, bd, bh, bw, stream);
}
