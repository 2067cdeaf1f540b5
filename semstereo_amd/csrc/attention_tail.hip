// The "attention tail" of SemStereo.forward between the two cost volumes (reference
// models/SemStereo.py:286-310), fused into two kernels:
//
//  ss_sample_strength_fwd   :286-293  variance gate, the two 5-tap propagations, the 5-candidate
//                                     warp of the 1/4-scale right features, their channel-mean
//                                     correlation with the left features, softmax over the 5
//  ss_topk_candidates_fwd   :295-310  5-tap propagation of the logits volume weighted by that
//                                     strength, softmax over D, the 24 most probable disparities
//                                     (descending sort, stable), re-sorted ascending, their
//                                     probabilities, and the soft-argmax over them
//
// The reference runs ~25 ATen kernels here, materialises a [B,5,D,H,W] volume and fully sorts D
// values per pixel to keep 24.  Both kernels put 64 consecutive pixels on the lanes (every plane access is
// a coalesced row segment) and split the per-pixel work -- channels, candidates -- over the 4 waves of
// the workgroup, so that a single 1024x1024 pair still gives every SIMD 4 waves; the candidate kernel
// parks the pixel's D probabilities in LDS ([D][64], conflict-free) and ranks them by counting.
#include <algorithm>
#include <limits.h>

#include "common.h"

namespace {

// (dy, dx) of the five propagation taps, models/submodule.py:295-300 / 367-372
__device__ __constant__ int kTapDy[5] = {-1, 0, 1, 1, -1};
__device__ __constant__ int kTapDx[5] = {-1, 0, 1, -1, 1};

struct Taps4 {
    int o_nw, o_ne, o_sw, o_se;
    float w_nw, w_ne, w_sw, w_se;
};

// identical arithmetic to warp.hip::make_taps (kept in sync by tests/test_parity_gpu.py)
__device__ __forceinline__ Taps4 bilinear_taps(float disp, int h, int w, int H, int W, float half_w, float half_h) {
    const float gx = ((float)w - disp) / half_w - 1.0f;
    const float gy = (float)h / half_h - 1.0f;
    const float ix = ss::mul_rn(gx + 1.0f, half_w);
    const float iy = ss::mul_rn(gy + 1.0f, half_h);
    const float xw = floorf(ix), yn = floorf(iy);
    const float fw = ix - xw, fe = 1.0f - fw, fn = iy - yn, fs = 1.0f - fn;
    const float xe = xw + 1.0f, ys = yn + 1.0f;
    const bool mw = (xw > -1.0f) && (xw < (float)W), me = (xe > -1.0f) && (xe < (float)W);
    const bool mn = (yn > -1.0f) && (yn < (float)H), ms = (ys > -1.0f) && (ys < (float)H);
    const int ixw = (int)xw, iyn = (int)yn;
    Taps4 t;
    t.w_nw = ss::mul_rn(fs, fe); t.w_ne = ss::mul_rn(fs, fw);
    t.w_sw = ss::mul_rn(fn, fe); t.w_se = ss::mul_rn(fn, fw);
    t.o_nw = (mn && mw) ? iyn * W + ixw : -1;
    t.o_ne = (mn && me) ? iyn * W + ixw + 1 : -1;
    t.o_sw = (ms && mw) ? (iyn + 1) * W + ixw : -1;
    t.o_se = (ms && me) ? (iyn + 1) * W + ixw + 1 : -1;
    return t;
}

__device__ __forceinline__ float bilinear(const float* __restrict__ plane, const Taps4& t) {
    const float a = (t.o_nw >= 0) ? plane[t.o_nw] : 0.f;
    const float b = (t.o_ne >= 0) ? plane[t.o_ne] : 0.f;
    const float c = (t.o_sw >= 0) ? plane[t.o_sw] : 0.f;
    const float d = (t.o_se >= 0) ? plane[t.o_se] : 0.f;
    float r = ss::mul_rn(a, t.w_nw);
    r = ss::add_rn(r, ss::mul_rn(b, t.w_ne));
    r = ss::add_rn(r, ss::mul_rn(c, t.w_sw));
    r = ss::add_rn(r, ss::mul_rn(d, t.w_se));
    return r;
}

// The west / east taps of a row are neighbours in memory: ONE 8-byte load (4-byte aligned; gfx950 takes it) instead of two
// 4-byte gathers -- the probe below is bound by its gather instructions (21 -> 11 per channel and pixel; 70 MB algorithmic at
// 1.1 TB/s).  `off` is where the pair is fetched, `mode` which of its halves are the taps: bit 0 / 1 = west / east tap valid,
// bit 2 = the west tap is the pair's SECOND element (east neighbour outside the image: the pair is fetched one column to the
// left so that it stays inside the plane), bit 3 = the east tap is the pair's FIRST element (west neighbour outside).
typedef float f32x2_u __attribute__((ext_vector_type(2)));
typedef f32x2_u f32x2_a4 __attribute__((aligned(4)));
struct PairTap {
    int off, mode;
};
__device__ __forceinline__ PairTap pair_tap(int o_w, int o_e) {
    PairTap p;
    if (o_w >= 0 && o_e >= 0) { p.off = o_w; p.mode = 3; }
    else if (o_w >= 0) { p.off = o_w - 1; p.mode = 1 | 4; }
    else if (o_e >= 0) { p.off = o_e; p.mode = 2 | 8; }
    else { p.off = 0; p.mode = 0; }
    return p;
}
__device__ __forceinline__ void pair_fetch(const float* __restrict__ plane, const PairTap& p, float& west, float& east) {
    const f32x2_u v = *reinterpret_cast<const f32x2_a4*>(plane + p.off);
    west = (p.mode & 1) ? ((p.mode & 4) ? v.y : v.x) : 0.f;
    east = (p.mode & 2) ? ((p.mode & 8) ? v.x : v.y) : 0.f;
}

// strength[b,t,y,x] = softmax_t( mean_c left[c] * warp(right, pred0[nb_t])[c]  *  sigmoid(beta + gamma * var[nb_t]) )
// 64 pixels x 4 waves: wave q owns channels [q*C/4, (q+1)*C/4) of the same 64 pixels (a pixel per thread
// would leave a 1024x1024 pair with one wave per SIMD and 1408 dependent L2 reads each); the four partial
// correlations meet in LDS and wave 0 finishes the softmax.
template <bool PAIRS>      // PAIRS: west / east taps fetched as one 8-byte pair (W >= 2)
__global__ __launch_bounds__(256, 3) void sample_strength_kernel(const float* __restrict__ left, const float* __restrict__ right,
                                                               const float* __restrict__ pred0, const float* __restrict__ var,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               float* __restrict__ strength, int C, int H, int W,
                                                               float half_w, float half_h, long long total) {
    __shared__ float part[4][5][64];
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const long long i = blockIdx.x * 64LL + lane;
    const bool active = i < total;
    const long long plane = (long long)H * W;
    const long long ii = active ? i : 0;
    const int x = (int)(ii % W), y = (int)((ii / W) % H);
    const long long b = ii / plane;
    Taps4 tp[5];
    long long nbs[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        const int yy = min(max(y + kTapDy[t], 0), H - 1), xx = min(max(x + kTapDx[t], 0), W - 1);   // replicate pad
        nbs[t] = b * plane + (long long)yy * W + xx;
        tp[t] = bilinear_taps(pred0[nbs[t]], y, x, H, W, half_w, half_h);
    }
    float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    const long long pix = (long long)y * W + x;
    const int cq = (C + 3) / 4, c0 = q * cq, c1 = min(C, c0 + cq);
    // The row coordinate of all five candidates is the pixel's own row through the fp32 round trip: on 3 rows in 4 it lands exactly
    // on the integer and both south weights are 0.  When that holds for every lane of the wave (a wave is 64 consecutive pixels:
    // one row whenever W % 64 == 0) the south taps are not fetched: half of the kernel's gathers (same value for finite features).
    bool south_live = false;
#pragma unroll
    for (int t = 0; t < 5; ++t) south_live |= (tp[t].w_sw != 0.f) || (tp[t].w_se != 0.f);
    const bool north_only = __builtin_amdgcn_ballot_w64(active && south_live) == 0;
    if constexpr (PAIRS) {     // (a 1-column image has no pair to fetch: the scalar taps below)
        PairTap pn[5], ps[5];
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            pn[t] = pair_tap(tp[t].o_nw, tp[t].o_ne);
            ps[t] = pair_tap(tp[t].o_sw, tp[t].o_se);
        }
        if (north_only) {
            for (int c = c0; c < c1; ++c) {
                const float l = left[(b * C + c) * plane + pix];
                const float* rp = right + (b * C + c) * plane;
#pragma unroll
                for (int t = 0; t < 5; ++t) {
                    float a, b2;
                    pair_fetch(rp, pn[t], a, b2);
                    acc[t] = ss::add_rn(acc[t], ss::mul_rn(l, ss::add_rn(ss::mul_rn(a, tp[t].w_nw), ss::mul_rn(b2, tp[t].w_ne))));
                }
            }
        } else {
            for (int c = c0; c < c1; ++c) {
                const float l = left[(b * C + c) * plane + pix];
                const float* rp = right + (b * C + c) * plane;
#pragma unroll
                for (int t = 0; t < 5; ++t) {
                    float a, b2, c2, d2;
                    pair_fetch(rp, pn[t], a, b2);
                    pair_fetch(rp, ps[t], c2, d2);
                    float r = ss::mul_rn(a, tp[t].w_nw);                 // the order of bilinear()
                    r = ss::add_rn(r, ss::mul_rn(b2, tp[t].w_ne));
                    r = ss::add_rn(r, ss::mul_rn(c2, tp[t].w_sw));
                    r = ss::add_rn(r, ss::mul_rn(d2, tp[t].w_se));
                    acc[t] = ss::add_rn(acc[t], ss::mul_rn(l, r));
                }
            }
        }
    } else if (north_only) {
        for (int c = c0; c < c1; ++c) {
            const float l = left[(b * C + c) * plane + pix];
            const float* rp = right + (b * C + c) * plane;
#pragma unroll
            for (int t = 0; t < 5; ++t) {
                const float a = (tp[t].o_nw >= 0) ? rp[tp[t].o_nw] : 0.f, b2 = (tp[t].o_ne >= 0) ? rp[tp[t].o_ne] : 0.f;
                acc[t] = ss::add_rn(acc[t], ss::mul_rn(l, ss::add_rn(ss::mul_rn(a, tp[t].w_nw), ss::mul_rn(b2, tp[t].w_ne))));
            }
        }
    } else {
        for (int c = c0; c < c1; ++c) {
            const float l = left[(b * C + c) * plane + pix];
            const float* rp = right + (b * C + c) * plane;
#pragma unroll
            for (int t = 0; t < 5; ++t) acc[t] = ss::add_rn(acc[t], ss::mul_rn(l, bilinear(rp, tp[t])));
        }
    }
#pragma unroll
    for (int t = 0; t < 5; ++t) part[q][t][lane] = acc[t];
    __syncthreads();
    if (q != 0 || !active) return;
    const float g = gamma[0], bt = beta[0];
    float z[5], mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        const float v = ss::add_rn(bt, ss::mul_rn(g, var[nbs[t]]));
        const float gate = 1.0f / (1.0f + expf(-v));
        const float corr = ss::add_rn(ss::add_rn(ss::add_rn(part[0][t][lane], part[1][t][lane]), part[2][t][lane]), part[3][t][lane]);
        z[t] = ss::mul_rn(corr / (float)C, gate);
        mx = fmaxf(mx, z[t]);
    }
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < 5; ++t) { z[t] = expf(z[t] - mx); sum = ss::add_rn(sum, z[t]); }
#pragma unroll
    for (int t = 0; t < 5; ++t) strength[(b * 5 + t) * plane + pix] = z[t] / sum;
}

// One thread per pixel; LDS: aw[D][T] and p[D][T] (T = blockDim.x threads).
__global__ void topk_candidates_kernel(const float* __restrict__ logits, const float* __restrict__ strength,
                                       float* __restrict__ samples, float* __restrict__ att_topk,
                                       float* __restrict__ pred_att, int D, int H, int W, int K, int m,
                                       long long total) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int T = blockDim.x, tid = threadIdx.x;
    float* aw = lds;                    // [D][T]
    float* pr = lds + (size_t)D * T;    // [D][T]
    const long long i = blockIdx.x * (long long)T + tid;
    if (i >= total) return;             // no barriers below: threads are independent
    const long long plane = (long long)H * W;
    const int x = (int)(i % W), y = (int)((i / W) % H);
    const long long b = i / plane;
    const long long pix = (long long)y * W + x;
    int nb[5];
    float st[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        const int yy = min(max(y + kTapDy[t], 0), H - 1), xx = min(max(x + kTapDx[t], 0), W - 1);
        nb[t] = yy * W + xx;
        st[t] = strength[(b * 5 + t) * plane + pix];
    }
    // aw[d] = sum_t logits[d, nb_t] * strength[t]   (:295-297), running max for the softmax (:298)
    const float* lb = logits + b * D * plane;
    float mx = -INFINITY;
    for (int d = 0; d < D; ++d) {
        const float* lp = lb + d * plane;
        float a = 0.f;
#pragma unroll
        for (int t = 0; t < 5; ++t) a = ss::add_rn(a, ss::mul_rn(lp[nb[t]], st[t]));
        aw[d * T + tid] = a;
        mx = fmaxf(mx, a);
    }
    float sum = 0.f;
    for (int d = 0; d < D; ++d) { const float e = expf(aw[d * T + tid] - mx); pr[d * T + tid] = e; sum = ss::add_rn(sum, e); }
    for (int d = 0; d < D; ++d) pr[d * T + tid] = pr[d * T + tid] / sum;

    // rank by probability, descending, ties to the lower index (stable sort): selected iff rank < K.
    // Pass 1 finds the max logit among the selected (for the :308 softmax); passes 2, 3 emit in
    // ascending index order, which is exactly ind_k.sort(2, False) of :303.
    float smax = -INFINITY;
    unsigned long long sel_lo = 0ull, sel_hi = 0ull;     // D <= 128
    for (int c = 0; c < D; ++c) {
        const float v = pr[c * T + tid];
        int rank = 0;
        for (int j = 0; j < D; ++j) {
            const float u = pr[j * T + tid];
            rank += (u > v) || (u == v && j < c);
        }
        if (rank < K) {
            if (c < 64) sel_lo |= 1ull << c; else sel_hi |= 1ull << (c - 64);
            smax = fmaxf(smax, aw[c * T + tid]);
        }
    }
    float esum = 0.f;
    int cnt = 0;
    for (int c = 0; c < D; ++c) {
        const bool s = (c < 64) ? ((sel_lo >> c) & 1ull) : ((sel_hi >> (c - 64)) & 1ull);
        if (!s) continue;
        samples[(b * K + cnt) * plane + pix] = (float)(c - m);
        att_topk[(b * K + cnt) * plane + pix] = pr[c * T + tid];
        esum = ss::add_rn(esum, expf(aw[c * T + tid] - smax));
        ++cnt;
    }
    float acc = 0.f;
    for (int c = 0; c < D; ++c) {
        const bool s = (c < 64) ? ((sel_lo >> c) & 1ull) : ((sel_hi >> (c - 64)) & 1ull);
        if (!s) continue;
        acc = ss::add_rn(acc, ss::mul_rn(expf(aw[c * T + tid] - smax) / esum, (float)(c - m)));
    }
    pred_att[b * plane + pix] = acc;
}

// Same contract, D known at compile time.  64 pixels x 4 waves: wave q owns the candidates
// [q*D/4, (q+1)*D/4) of the same 64 pixels (one thread per pixel leaves a 1024x1024 pair with a single
// wave per SIMD).  Ranking is a count of strictly greater values: each thread keeps its D/4
// probabilities in registers and streams all D of the pixel past them from LDS ([D][64], conflict-
// free).  Stable-sort semantics from the counts alone: values of one tie group share their count G
// and occupy ranks G .. G+E-1 in index order; with g* = max{G < K} every group with G < g* is selected
// whole and the group G == g* contributes its first K - g* members in index order (the per-wave
// quotas follow from the waves' tie counts).
template <int D>
__global__ __launch_bounds__(256) void topk_candidates_split(const float* __restrict__ logits, const float* __restrict__ strength,
                                                              float* __restrict__ samples, float* __restrict__ att_topk,
                                                              float* __restrict__ pred_att, int H, int W, int K, int m,
                                                              long long total) {
    constexpr int DQ = D / 4;
    static_assert(D % 4 == 0, "D splits over 4 waves");
    __shared__ float pr[D][64];
    __shared__ float redf[4][64];
    __shared__ int redi[2][4][64];
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const long long i = blockIdx.x * 64LL + lane;
    const bool active = i < total;
    const long long ii = active ? i : 0;
    const long long plane = (long long)H * W;
    const int x = (int)(ii % W), y = (int)((ii / W) % H);
    const long long b = ii / plane;
    const long long pix = (long long)y * W + x;
    int nb[5];
    float st[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        const int yy = min(max(y + kTapDy[t], 0), H - 1), xx = min(max(x + kTapDx[t], 0), W - 1);
        nb[t] = yy * W + xx;
        st[t] = strength[(b * 5 + t) * plane + pix];
    }
    // aw[d] = sum_t logits[d, nb_t] * strength[t]   (:295-297)
    const float* lb = logits + (b * D + q * DQ) * plane;
    float aw[DQ], p[DQ];
    float mx = -INFINITY;
    // the 5 * DQ loads are issued in batches of 40 BEFORE their sums: left to the compiler each load was followed by its
    // own s_waitcnt vmcnt(0) (74 of them for D = 64), i.e. one exposed L2 round trip per value
    constexpr int KB = (DQ % 8 == 0) ? 8 : DQ;
#pragma unroll
    for (int k0 = 0; k0 < DQ; k0 += KB) {
        float lv[KB][5];
#pragma unroll
        for (int k = 0; k < KB; ++k)
#pragma unroll
            for (int t = 0; t < 5; ++t) lv[k][t] = lb[(k0 + k) * plane + nb[t]];
#pragma unroll
        for (int k = 0; k < KB; ++k) {
            float a = 0.f;
#pragma unroll
            for (int t = 0; t < 5; ++t) a = ss::add_rn(a, ss::mul_rn(lv[k][t], st[t]));
            aw[k0 + k] = a;
            mx = fmaxf(mx, a);
        }
    }
    redf[q][lane] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(redf[0][lane], redf[1][lane]), fmaxf(redf[2][lane], redf[3][lane]));
    __syncthreads();
    float sum = 0.f;                                            // softmax over D (:298)
#pragma unroll
    for (int k = 0; k < DQ; ++k) { p[k] = expf(aw[k] - mx); sum = ss::add_rn(sum, p[k]); }
    redf[q][lane] = sum;
    __syncthreads();
    sum = ss::add_rn(ss::add_rn(ss::add_rn(redf[0][lane], redf[1][lane]), redf[2][lane]), redf[3][lane]);
#pragma unroll
    for (int k = 0; k < DQ; ++k) { p[k] = p[k] / sum; pr[q * DQ + k][lane] = p[k]; }
    __syncthreads();

    int g[DQ];
#pragma unroll
    for (int k = 0; k < DQ; ++k) g[k] = 0;
#pragma unroll 4
    for (int j = 0; j < D; ++j) {
        const float u = pr[j][lane];
#pragma unroll
        for (int k = 0; k < DQ; ++k) g[k] += (u > p[k]) ? 1 : 0;
    }
    int gstar = -1;
#pragma unroll
    for (int k = 0; k < DQ; ++k)
        if (g[k] < K) gstar = max(gstar, g[k]);
    redi[0][q][lane] = gstar;
    __syncthreads();
    gstar = max(max(redi[0][0][lane], redi[0][1][lane]), max(redi[0][2][lane], redi[0][3][lane]));
    int nsel = 0, ntie = 0;                                     // this wave's whole-group members / boundary-group members
#pragma unroll
    for (int k = 0; k < DQ; ++k) { nsel += (g[k] < gstar) ? 1 : 0; ntie += (g[k] == gstar) ? 1 : 0; }
    redi[1][q][lane] = nsel | (ntie << 16);
    __syncthreads();
    int quota = K - gstar, cnt = 0;                             // ties still to take when this wave starts, outputs before it
    for (int qq = 0; qq < q; ++qq) {
        const int v = redi[1][qq][lane];
        const int take = min(v >> 16, quota);
        quota -= take;
        cnt += (v & 0xffff) + take;
    }
    unsigned sel = 0u;
    float smax = -INFINITY;
#pragma unroll
    for (int k = 0; k < DQ; ++k) {
        bool s = g[k] < gstar;
        if (g[k] == gstar && quota > 0) { s = true; --quota; }
        if (s) { sel |= 1u << k; smax = fmaxf(smax, aw[k]); }
    }
    redf[q][lane] = smax;
    __syncthreads();
    smax = fmaxf(fmaxf(redf[0][lane], redf[1][lane]), fmaxf(redf[2][lane], redf[3][lane]));
    __syncthreads();
    float esum = 0.f, wsum = 0.f;                               // :307-310 over the selected, ascending index
#pragma unroll
    for (int k = 0; k < DQ; ++k) {
        if (!((sel >> k) & 1u)) continue;
        const int c = q * DQ + k;
        if (active) {
            samples[(b * K + cnt) * plane + pix] = (float)(c - m);
            att_topk[(b * K + cnt) * plane + pix] = p[k];
        }
        const float e = expf(aw[k] - smax);
        esum = ss::add_rn(esum, e);
        ++cnt;
    }
    redf[q][lane] = esum;
    __syncthreads();
    esum = ss::add_rn(ss::add_rn(ss::add_rn(redf[0][lane], redf[1][lane]), redf[2][lane]), redf[3][lane]);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < DQ; ++k) {
        if (!((sel >> k) & 1u)) continue;
        wsum = ss::add_rn(wsum, ss::mul_rn(expf(aw[k] - smax) / esum, (float)(q * DQ + k - m)));
    }
    redf[q][lane] = wsum;
    __syncthreads();
    if (q == 0 && active)
        pred_att[b * plane + pix] = ss::add_rn(ss::add_rn(ss::add_rn(redf[0][lane], redf[1][lane]), redf[2][lane]), redf[3][lane]);
}

// softmax over D + expectation + variance (models/SemStereo.py:281-285): 64 pixels x 4 waves, the
// D axis split over the waves in 16-plane chunks whose exponentials stay in registers.
// UP: the logits are the 2x trilinear up-sampling (F.interpolate(..., mode='trilinear'), align_corners=False,
// models/SemStereo.py:279) of `logits` = the 1/8-scale classifier output [B,1,D/2,H/2,W/2], computed here on the fly in
// ATen's own nesting -- per coarse plane the bilinear (h, w) value, then the blend along D -- and also written to `up`
// [B,D,H,W] for the candidate selection that follows: one pass instead of upsample_trilinear3d + this kernel.
template <int NCH, bool UP>   // 16-plane chunks per wave: D <= 64*NCH
__global__ __launch_bounds__(256) void softmax_regress_split(const float* __restrict__ logits, float* __restrict__ prob,
                                                              float* __restrict__ disp, float* __restrict__ var,
                                                              float* __restrict__ up, int D, int m, int W,
                                                              long long plane, long long total) {
    __shared__ float red[4][64];
    __shared__ float part[4 * NCH][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long i = blockIdx.x * 64LL + lane;
    const bool active = i < total;
    const long long pix = active ? i % plane : 0, b = active ? i / plane : 0;
    const float* lp = logits + b * D * plane + pix;
    float e[NCH][16];
    float mx = -INFINITY;
    if constexpr (UP) {
        // source coordinates of ATen's area_pixel_compute_source_index(scale = 0.5, align_corners = false)
        const int H = (int)(plane / W), W8 = W / 2, H8 = H / 2, D8 = D / 2;
        const int x = (int)(pix % W), y = (int)(pix / W);
        const float sx = fmaxf(0.5f * (x + 0.5f) - 0.5f, 0.f), sy = fmaxf(0.5f * (y + 0.5f) - 0.5f, 0.f);
        const int x0 = (int)sx, y0 = (int)sy;
        const int x1 = x0 + (x0 < W8 - 1), y1 = y0 + (y0 < H8 - 1);
        const float lx1 = sx - x0, lx0 = 1.f - lx1, ly1 = sy - y0, ly0 = 1.f - ly1;
        const long long plane8 = (long long)H8 * W8;
        const float* cp = logits + b * D8 * plane8;
        const int o00 = y0 * W8 + x0, o01 = y0 * W8 + x1, o10 = y1 * W8 + x0, o11 = y1 * W8 + x1;
#pragma unroll
        for (int n = 0; n < NCH; ++n) {
            const int dA = (n * 4 + wave) * 16;                   // first fine plane of this chunk (wave-uniform)
            float P[10];                                          // bilinear values of coarse planes dA/2 - 1 .. dA/2 + 8
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                const int c = min(max(dA / 2 - 1 + j, 0), D8 - 1);
                const float* q = cp + c * plane8;
                const float v00 = q[o00], v01 = q[o01], v10 = q[o10], v11 = q[o11];
                P[j] = ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11);
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int d = dA + k;
                // even d = 2c: planes c - 1 (0.25) and c (0.75), d = 0: plane 0 alone; odd d = 2c + 1: planes c (0.75), c + 1 (0.25)
                float v;
                if (k & 1) v = 0.75f * P[k / 2 + 1] + 0.25f * P[k / 2 + 2];
                else v = (d == 0) ? (1.f * P[1] + 0.f * P[1]) : (0.25f * P[k / 2] + 0.75f * P[k / 2 + 1]);
                const bool ok = active && d < D;
                if (ok) up[(b * D + d) * plane + pix] = v;
                e[n][k] = ok ? v : -INFINITY;
                mx = fmaxf(mx, e[n][k]);
            }
        }
    } else {
#pragma unroll
        for (int n = 0; n < NCH; ++n)
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int d = (n * 4 + wave) * 16 + k;
                e[n][k] = (active && d < D) ? lp[d * plane] : -INFINITY;
                mx = fmaxf(mx, e[n][k]);
            }
    }
    red[wave][lane] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0][lane], red[1][lane]), fmaxf(red[2][lane], red[3][lane]));
    __syncthreads();
    // sum of exponentials in plane order d = 0..D-1 (sequential like ATen's softmax accumulation)
#pragma unroll
    for (int n = 0; n < NCH; ++n) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int d = (n * 4 + wave) * 16 + k;
            e[n][k] = (d < D) ? expf(e[n][k] - mx) : 0.f;
            s = ss::add_rn(s, e[n][k]);
        }
        part[n * 4 + wave][lane] = s;
    }
    __syncthreads();
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < 4 * NCH; ++c) sum = ss::add_rn(sum, part[c][lane]);
    __syncthreads();
#pragma unroll
    for (int n = 0; n < NCH; ++n) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int d = (n * 4 + wave) * 16 + k;
            e[n][k] = e[n][k] / sum;
            if (prob && active && d < D) prob[(b * D + d) * plane + pix] = e[n][k];
            s = ss::add_rn(s, ss::mul_rn(e[n][k], (float)(d - m)));
        }
        part[n * 4 + wave][lane] = s;
    }
    __syncthreads();
    float mean = 0.f;
#pragma unroll
    for (int c = 0; c < 4 * NCH; ++c) mean = ss::add_rn(mean, part[c][lane]);
    __syncthreads();
#pragma unroll
    for (int n = 0; n < NCH; ++n) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int d = (n * 4 + wave) * 16 + k;
            const float t = (float)(d - m) - mean;
            s = ss::add_rn(s, ss::mul_rn(e[n][k], ss::mul_rn(t, t)));
        }
        part[n * 4 + wave][lane] = s;
    }
    __syncthreads();
    if (wave == 0 && active) {
        float v = 0.f;
#pragma unroll
        for (int c = 0; c < 4 * NCH; ++c) v = ss::add_rn(v, part[c][lane]);
        disp[i] = mean;
        var[i] = v;
    }
}

}  // namespace

extern "C" int ss_sample_strength_fwd(const float* left, const float* right, const float* pred0, const float* var,
                                      const float* gamma, const float* beta, float* strength, int B, int C, int H, int W,
                                      ss_stream_t stream) {
    SS_REQUIRE(left && right && pred0 && var && gamma && beta && strength);
    SS_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0);
    const long long total = (long long)B * H * W;
    const float half_w = (float)((W - 1.0) / 2.0), half_h = (float)((H - 1.0) / 2.0);
    const dim3 grid((unsigned)ss::ceil_div_ll(total, 64));
    if (W >= 2)
        hipLaunchKernelGGL(sample_strength_kernel<true>, grid, dim3(256), 0, ss::as_stream(stream), left, right, pred0, var,
                           gamma, beta, strength, C, H, W, half_w, half_h, total);
    else
        hipLaunchKernelGGL(sample_strength_kernel<false>, grid, dim3(256), 0, ss::as_stream(stream), left, right, pred0, var,
                           gamma, beta, strength, C, H, W, half_w, half_h, total);
    return ss::check_launch();
}

extern "C" int ss_topk_candidates_fwd(const float* logits, const float* strength, float* samples, float* att_topk,
                                      float* pred_att, int B, int dmin, int ndisp, int H, int W, int k, ss_stream_t stream) {
    SS_REQUIRE(logits && strength && samples && att_topk && pred_att);
    SS_REQUIRE(B > 0 && ndisp > 0 && H > 0 && W > 0 && k > 0);
    const int D = ndisp, maxdisp = -dmin;                       // candidate value of plane d: d - maxdisp = dmin + d
    SS_REQUIRE(k <= D);
    if (D > 128) return SS_ERR_UNSUPPORTED;
    const int T = 128;
    const long long npix = (long long)B * H * W;
#define SS_TOPK_REG(DD)                                                                                              \
    if (D == DD) {                                                                                                   \
        hipLaunchKernelGGL(topk_candidates_split<DD>, dim3((unsigned)ss::ceil_div_ll(npix, 64)), dim3(256), 0,       \
                           ss::as_stream(stream), logits, strength, samples, att_topk, pred_att, H, W, k, maxdisp,   \
                           npix);                                                                                    \
        return ss::check_launch();                                                                                   \
    }
    SS_TOPK_REG(32)
    SS_TOPK_REG(64)
    SS_TOPK_REG(96)
#undef SS_TOPK_REG
    const size_t lds = (size_t)2 * D * T * sizeof(float);
    auto kern = topk_candidates_kernel;
    if (lds > 64 * 1024) {
        if (ss::ensure_dynamic_lds(reinterpret_cast<const void*>(kern), (int)lds) != SS_OK) return SS_ERR_LAUNCH;
    }
    const long long total = (long long)B * H * W;
    hipLaunchKernelGGL(kern, dim3((unsigned)ss::ceil_div_ll(total, T)), dim3(T), lds, ss::as_stream(stream), logits,
                       strength, samples, att_topk, pred_att, D, H, W, k, maxdisp, total);
    return ss::check_launch();
}

// exported for regression.hip's ss_softmax_regression_fwd
int ss_softmax_regress_split_launch(const float* logits, float* prob, float* disp, float* var, int B, int dmin, int ndisp,
                                    int H, int W, hipStream_t st) {
    const int D = ndisp, maxdisp = -dmin;
    const long long plane = (long long)H * W, total = (long long)B * plane;
    dim3 grid((unsigned)ss::ceil_div_ll(total, 64));
    if (D <= 64)
        hipLaunchKernelGGL((softmax_regress_split<1, false>), grid, dim3(256), 0, st, logits, prob, disp, var, nullptr, D, maxdisp, W, plane, total);
    else if (D <= 128)
        hipLaunchKernelGGL((softmax_regress_split<2, false>), grid, dim3(256), 0, st, logits, prob, disp, var, nullptr, D, maxdisp, W, plane, total);
    else
        return 1;   // caller falls back to the one-thread-per-pixel kernel
    return 0;
}

extern "C" int ss_upsample_softmax_regression_fwd(const float* coarse, float* up, float* disp, float* var, int B, int dmin,
                                                  int ndisp, int H, int W, ss_stream_t stream) {
    SS_REQUIRE(coarse && up && disp && var);
    SS_REQUIRE(B > 0 && ndisp > 0 && H > 0 && W > 0);
    const int D = ndisp, maxdisp = -dmin;
    if ((H & 1) || (W & 1) || D > 128 || (D & 1)) return SS_ERR_UNSUPPORTED;       // exact 2x in every dimension
    const long long plane = (long long)H * W, total = (long long)B * plane;
    dim3 grid((unsigned)ss::ceil_div_ll(total, 64));
    hipStream_t st = ss::as_stream(stream);
    if (D <= 64)
        hipLaunchKernelGGL((softmax_regress_split<1, true>), grid, dim3(256), 0, st, coarse, nullptr, disp, var, up, D, maxdisp, W, plane, total);
    else
        hipLaunchKernelGGL((softmax_regress_split<2, true>), grid, dim3(256), 0, st, coarse, nullptr, disp, var, up, D, maxdisp, W, plane, total);
    return ss::check_launch();
}
