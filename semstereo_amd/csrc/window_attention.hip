// Windowed multi-head self-attention of the hourglass bottleneck (gfx950).
//
// Replaces attention_block.forward (reference models/submodule_other.py:790-837): zero-pad H, W
// to window multiples, partition [B,C,D,H,W] into (bd,bh,bw) windows of T tokens, qkv =
// Linear(C -> 3C), 16 heads x 8 dims, softmax(q k^T / sqrt(8) [+ pad mask]) v, un-partition, crop,
// 1x1x1 conv C -> C with bias.  The reference does this with 8-D permute copies, a batched GEMM
// per step and a materialised [windows, heads, T, T] logits tensor.
//
// Here one workgroup owns one window and never leaves the CU: the window's tokens are parked
// channel-major in LDS ([C][T], the natural NCDHW order); per group of 4 heads the q/k/v slab
// [96][T] is produced by fp32 MFMA (M = features, weights streamed from L2, N = tokens from LDS),
// the 4 heads' attention runs on the VALU with one (head, query) pair per lane and the softmax
// row in registers, and the output projection is accumulated across head groups directly in MFMA
// accumulators (out += Wout[:, group] * y_group), so the attended tokens never go back to HBM.
#include <algorithm>

#include "common.h"
#include "split_f16.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int HD = 8;      // head dim (C / heads)
constexpr int HG = 4;      // heads per group -> 32 channels = one MFMA M-tile per q / k / v

template <int T, int C>
struct ACfg {
    static constexpr int NTL = (T + 31) / 32;     // 32-token N tiles
    static constexpr int TP = NTL * 32;           // padded token count
    static constexpr int XS = C * TP;             // window tokens, channel-major
    static constexpr int QKV = 3 * 32 * TP;       // q,k,v slab of one head group
    static constexpr int YG = 32 * TP;            // attended channels of one head group
    static constexpr int FLAGS = TP;              // pad flag per token
    static constexpr int WL = C * 96;             // q/k/v weight slab of one head group, [k][96 features]
    static constexpr size_t LDS_BYTES = (size_t)(XS + QKV + YG + FLAGS + WL) * 4;
};

template <int T, int C>
__global__ __launch_bounds__(256) void window_attention_kernel(
    const float* __restrict__ x, const float* __restrict__ wqkv_t, const float* __restrict__ bqkv,
    const float* __restrict__ wout_t, const float* __restrict__ bout, float* __restrict__ out, int D, int H, int W,
    int bd, int bh, int bw, int nwh, int nww, int use_mask) {
    using A = ACfg<T, C>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* xs = lds;                 // [C][TP]
    float* qkv = xs + A::XS;         // [3*32][TP]   rows 0-31 q, 32-63 k, 64-95 v of the current head group
    float* yg = qkv + A::QKV;        // [32][TP]
    float* flag = yg + A::YG;        // [TP]  1 = padded position (reference mask semantics)
    float* wl = flag + A::FLAGS;     // [C][96]  qkv_3d weight rows of the current head group, k-major

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    int wi = blockIdx.x;
    const int ww = wi % nww; wi /= nww;
    const int wh = wi % nwh; wi /= nwh;
    const int wd = wi;
    const int b = blockIdx.y;
    const size_t plane = (size_t)H * W, vol = (size_t)D * plane;
    const float* xb = x + (size_t)b * C * vol;

    // ---- phase 0: window tokens -> LDS (zeros at padded positions and in the N-tile tail) ----
    if (bw == 4 && (W & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) & 15) == 0)) {
        // a window row is exactly one aligned float4: C * T/4 quads, loaded in branch-free batches
        constexpr int NQ = C * (A::TP / 4) / 256;          // quads per thread (12 or 8)
        static_assert(C * (A::TP / 4) % 256 == 0, "whole batches");
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int e = tid + 256 * i;
            const int tq = e % (A::TP / 4), c = e / (A::TP / 4);
            const int ih = tq % bh, id = tq / bh;
            const int gh = wh * bh + ih, gd = wd * bd + id, gw = ww * 4;
            const bool ok = gh < H && gw < W;
            const float4 q = *reinterpret_cast<const float4*>(xb + (size_t)c * vol + (size_t)gd * plane +
                                                              (size_t)min(gh, H - 1) * W + min(gw, W - 4));
            *reinterpret_cast<float4*>(&xs[e * 4]) = ok ? q : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    } else {
        for (int e = tid; e < C * A::TP; e += 256) {
            const int t = e % A::TP, c = e / A::TP;
            float v = 0.f;
            if (t < T) {
                const int iw = t % bw, ih = (t / bw) % bh, id = t / (bw * bh);
                const int gw = ww * bw + iw, gh = wh * bh + ih, gd = wd * bd + id;
                if (gh < H && gw < W) v = xb[(size_t)c * vol + (size_t)gd * plane + (size_t)gh * W + gw];
            }
            xs[e] = v;
        }
    }
    for (int t = tid; t < A::TP; t += 256) {
        float f = 0.f;
        if (t < T) {
            const int iw = t % bw, ih = (t / bw) % bh;
            f = ((wh * bh + ih >= H) || (ww * bw + iw >= W)) ? 1.f : 0.f;
        }
        flag[t] = f;
    }
    __syncthreads();

    // output-projection accumulators: this wave owns output channels [32*wave, 32*wave+32), all tokens
    static_assert(C == 128, "4 waves x 32 output channels");
    f32x16 oacc[A::NTL];
#pragma unroll
    for (int n = 0; n < A::NTL; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[n][r] = 0.f;

    const float scale = 0.35355339059327379f;    // 8 ** -0.5, rounded to fp32 like the reference's Python float
    constexpr int NGROUPS = C / (HG * HD);

    // the group's 96 weight rows (48 KB) go through LDS, fetched one group ahead into registers:
    // 4 waves share them, and the MFMA loop never waits on a global load
    constexpr int NWQ = C * 24 / 256;                 // float4 per thread
    float4 wreg[NWQ];
    const float* wsrc[NWQ];                           // this thread's NWQ quads of group 0; group g is + g*32 floats
#pragma unroll
    for (int i = 0; i < NWQ; ++i) {
        const int e = tid + 256 * i;
        wsrc[i] = wqkv_t + (size_t)(e / 24) * 3 * C + ((e % 24) / 8) * C + ((e % 24) % 8) * 4;
        wreg[i] = *reinterpret_cast<const float4*>(wsrc[i]);
    }

    for (int g = 0; g < NGROUPS; ++g) {
#pragma unroll
        for (int i = 0; i < NWQ; ++i) {
            const int e = tid + 256 * i;
            *reinterpret_cast<float4*>(&wl[(e / 24) * 96 + (e % 24) * 4]) = wreg[i];
        }
        __syncthreads();
        if (g + 1 < NGROUPS) {
#pragma unroll
            for (int i = 0; i < NWQ; ++i) wreg[i] = *reinterpret_cast<const float4*>(wsrc[i] + (g + 1) * 32);
        }
        // ---- phase 1: q/k/v slab of this head group = W[rows] * X + bias, by MFMA ----
        for (int u = wave; u < 3 * A::NTL; u += 4) {
            const int mt = u / A::NTL, nt = u % A::NTL;        // mt: 0 = q, 1 = k, 2 = v
            const int f0 = mt * C + g * 32;                    // first feature row of this tile
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const float* ap = wl + half * 96 + mt * 32 + l31;
            const float* bp = xs + half * A::TP + nt * 32 + l31;
#pragma unroll 8
            for (int kk = 0; kk < C; kk += 2)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[kk * 96], bp[kk * A::TP], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                qkv[(mt * 32 + row) * A::TP + nt * 32 + l31] = ss::add_rn(acc[r], bqkv[f0 + row]);
            }
        }
        __syncthreads();

        // ---- phase 2: attention of the group's 4 heads on the matrix core ----
        // One unit = (head hh, tile of 32 queries).  Scores are computed TRANSPOSED,
        //   St[key][query] = sum_d K[d][key] * Q[d][query]          (A = K^T, B = Q, 4 K-steps of 2 dims),
        // so a lane holds ONE query (column) and its keys run over the accumulator registers and the two
        // lane halves: the softmax over keys is in-register plus one cross-half exchange, and each
        // accumulator register of the normalised tile is, as it stands, the B operand of a K-step of
        //   Ot[dim][query] = sum_key V[dim][key] * Pt[key][query]   (A = V rows 0-7, rows 8-31 zero):
        // register r of key tile kt pairs keys kt*32 + (r&3) + 8*(r>>2) (+4 in the upper half).
        static_assert(A::TP == T, "windows of 64 / 96 tokens are whole 32-token tiles");
        for (int u = wave; u < HG * A::NTL; u += 4) {
            const int hh = u / A::NTL, qt = u % A::NTL;
            f32x16 st[A::NTL];
#pragma unroll
            for (int kt = 0; kt < A::NTL; ++kt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) st[kt][r] = 0.f;
#pragma unroll
                for (int sd = 0; sd < HD / 2; ++sd)
                    st[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                        qkv[(32 + hh * HD + 2 * sd + half) * A::TP + kt * 32 + l31],     // K[dim][key]
                        qkv[(hh * HD + 2 * sd + half) * A::TP + qt * 32 + l31],          // Q[dim][query]
                        st[kt], 0, 0, 0);
            }
            const float fq = flag[qt * 32 + l31];
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < A::NTL; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float d = ss::mul_rn(st[kt][r], scale);
                    if (use_mask && flag[kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half] != fq) d = ss::add_rn(d, -1000.0f);
                    st[kt][r] = d;
                    mx = fmaxf(mx, d);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < A::NTL; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) { st[kt][r] = ss::exp_fast(st[kt][r] - mx); sum = ss::add_rn(sum, st[kt][r]); }
            sum = ss::add_rn(sum, __shfl_xor(sum, 32));
            const float rsum = 1.0f / sum;            // one division per query; p = e * (1/sum) is within 1 ulp of e / sum
            f32x16 ot;
#pragma unroll
            for (int r = 0; r < 16; ++r) ot[r] = 0.f;
            const float* vrow = qkv + (64 + hh * HD + (l31 & 7)) * A::TP + 4 * half;     // V[dim = lane][...]
#pragma unroll
            for (int kt = 0; kt < A::NTL; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float vv = (l31 < HD) ? vrow[kt * 32 + (r & 3) + 8 * (r >> 2)] : 0.f;
                    ot = __builtin_amdgcn_mfma_f32_32x32x2f32(vv, ss::mul_rn(st[kt][r], rsum), ot, 0, 0, 0);
                }
            // rows 0-3 of Ot sit in registers 0-3 of the lower half, rows 4-7 in those of the upper half
#pragma unroll
            for (int r = 0; r < 4; ++r) yg[(hh * HD + 4 * half + r) * A::TP + qt * 32 + l31] = ot[r];
        }
        __syncthreads();

        // ---- phase 3: out += Wout[:, group channels] * y_group ----
        {
            const float* ap = wout_t + (size_t)(g * 32 + half) * C + wave * 32 + l31;
            const float* bp = yg + half * A::TP + l31;
#pragma unroll
            for (int kk = 0; kk < 32; kk += 2) {
                const float a = ap[(size_t)kk * C];
#pragma unroll
                for (int n = 0; n < A::NTL; ++n)
                    oacc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bp[kk * A::TP + n * 32], oacc[n], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // ---- epilogue: + bias, un-partition, crop ----
    float* ob = out + (size_t)b * C * vol;
#pragma unroll
    for (int n = 0; n < A::NTL; ++n) {
        const int t = n * 32 + l31;
        if (t >= T) continue;
        const int iw = t % bw, ih = (t / bw) % bh, id = t / (bw * bh);
        const int gw = ww * bw + iw, gh = wh * bh + ih, gd = wd * bd + id;
        if (gh >= H || gw >= W) continue;
        const size_t pos = (size_t)gd * plane + (size_t)gh * W + gw;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            ob[(size_t)co * vol + pos] = ss::add_rn(oacc[n][r], bout[co]);
        }
    }
}

// The same attention, split in three launches (ss_window_attention_core_fwd below): the q/k/v projection
// and the output projection are plain 1x1x1 convolutions over the unpadded volume (ss_conv3d_fwd, k = 1),
// and this kernel is only softmax(q k^T) v for ONE (window, group of 4 heads): 37 KB of LDS instead of
// 148 KB, 4x the workgroups -- a single 1024x1024 pair then fills the chip (the fused kernel above has one
// workgroup per window: 128-256 of them, one wave per SIMD).  A padded token's q/k/v is the Linear bias
// (the reference pads with zeros before qkv_3d, models/submodule_other.py:797-803).
template <int T, int C>
__global__ __launch_bounds__(256) void window_attention_core(const float* __restrict__ qkv_in, const float* __restrict__ bqkv,
                                                              float* __restrict__ y, int D, int H, int W, int bd, int bh,
                                                              int bw, int nwh, int nww, int use_mask) {
    using A = ACfg<T, C>;
    static_assert(A::TP == T, "windows of 64 / 96 tokens are whole 32-token tiles");
    __shared__ __attribute__((aligned(16))) float qkv[3 * 32 * A::TP];     // rows 0-31 q, 32-63 k, 64-95 v of this head group
    __shared__ float flag[A::TP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    constexpr int NGROUPS = C / (HG * HD);
    int wi = blockIdx.x;
    const int g = wi % NGROUPS; wi /= NGROUPS;
    const int ww = wi % nww; wi /= nww;
    const int wh = wi % nwh; wi /= nwh;
    const int wd = wi;
    const int b = blockIdx.y;
    const size_t plane = (size_t)H * W, vol = (size_t)D * plane;
    const float* qb = qkv_in + (size_t)b * 3 * C * vol;

    __shared__ unsigned vmax_w[4];
    float vm = 0.f;                                           // |max| of this thread's share of the V rows (64-95)
    if (bw == 4 && (W & 3) == 0 && ((reinterpret_cast<uintptr_t>(qkv_in) & 15) == 0)) {
        constexpr int NQ = 96 * (A::TP / 4) / 256;           // quads per thread (9 or 6)
        static_assert(96 * (A::TP / 4) % 256 == 0, "whole batches");
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int e = tid + 256 * i;
            const int tq = e % (A::TP / 4), row = e / (A::TP / 4);
            const int f = (row >> 5) * C + g * 32 + (row & 31);
            const int ih = tq % bh, id = tq / bh;
            const int gh = wh * bh + ih, gd = wd * bd + id, gw = ww * 4;
            const bool ok = gh < H && gw < W;
            const float4 q = *reinterpret_cast<const float4*>(qb + (size_t)f * vol + (size_t)gd * plane +
                                                              (size_t)min(gh, H - 1) * W + min(gw, W - 4));
            const float bf = bqkv[f];
            const float4 v4 = ok ? q : make_float4(bf, bf, bf, bf);
            *reinterpret_cast<float4*>(&qkv[e * 4]) = v4;
            if (row >= 64) vm = fmaxf(vm, fmaxf(fmaxf(fabsf(v4.x), fabsf(v4.y)), fmaxf(fabsf(v4.z), fabsf(v4.w))));
        }
    } else {
        for (int e = tid; e < 96 * A::TP; e += 256) {
            const int t = e % A::TP, row = e / A::TP;
            const int f = (row >> 5) * C + g * 32 + (row & 31);
            const int iw = t % bw, ih = (t / bw) % bh, id = t / (bw * bh);
            const int gw = ww * bw + iw, gh = wh * bh + ih, gd = wd * bd + id;
            const float v1 = (gh < H && gw < W) ? qb[(size_t)f * vol + (size_t)gd * plane + (size_t)gh * W + gw] : bqkv[f];
            qkv[e] = v1;
            if (row >= 64) vm = fmaxf(vm, fabsf(v1));
        }
    }
    {
        const unsigned wm = wave_max_bits(__float_as_uint(vm));
        if (lane == 0) vmax_w[wave] = wm;
    }
    for (int t = tid; t < A::TP; t += 256) {
        const int iw = t % bw, ih = (t / bw) % bh;
        flag[t] = ((wh * bh + ih >= H) || (ww * bw + iw >= W)) ? 1.f : 0.f;
    }
    __syncthreads();

    const float scale = 0.35355339059327379f;    // 8 ** -0.5
    // block exponent of V (see the P V product below): max -> [2^14, 2^15); a non-finite V makes the scale non-finite
    const int e_v = max((int)(max(max(vmax_w[0], vmax_w[1]), max(vmax_w[2], vmax_w[3])) >> 23), E_MIN);
    const float v_scale = __uint_as_float((unsigned)(127 + E_ONE - e_v) << 23);
    const float pv_unscale = __uint_as_float((unsigned)(127 - E_ONE + e_v) << 23) * (1.0f / 16384.0f);
    // V -> two scaled fp16 terms, IN PLACE and in the order the P V product reads it: the 16 tokens of a K-step (64 bytes of a
    // V row) become [half 0: hi, lo][half 1: hi, lo], 8 fp16 each, lane half h holding tokens {4 h .. +3, 8 + 4 h .. +3} of
    // the step.  Once per workgroup instead of once per (head, query tile) unit.
    for (int wk = tid; wk < 32 * (A::TP / 16); wk += 256) {
        float* base = qkv + (64 + wk / (A::TP / 16)) * A::TP + (wk % (A::TP / 16)) * 16;
        float f[16];
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) ss::lds_read16(base + 4 * k4, &f[4 * k4]);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            unsigned hi[4], lo[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int t0 = (j < 2) ? 4 * h + 2 * j : 8 + 4 * h + 2 * (j - 2);
                split2_pk_f16(f[t0] * v_scale, f[t0 + 1] * v_scale, hi[j], lo[j]);
            }
            *reinterpret_cast<uint4*>(base + 8 * h) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
            *reinterpret_cast<uint4*>(base + 8 * h + 4) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
        }
    }
    __syncthreads();
    float* yb = y + (size_t)b * C * vol;
    // one unit = (head hh, tile of 32 queries); see the fused kernel's phase 2 for the operand layout
    for (int u = wave; u < HG * A::NTL; u += 4) {
        const int hh = u / A::NTL, qt = u % A::NTL;
        f32x16 st[A::NTL];
#pragma unroll
        for (int kt = 0; kt < A::NTL; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[kt][r] = 0.f;
#pragma unroll
            for (int sd = 0; sd < HD / 2; ++sd)
                st[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                    qkv[(32 + hh * HD + 2 * sd + half) * A::TP + kt * 32 + l31],     // K[dim][key]
                    qkv[(hh * HD + 2 * sd + half) * A::TP + qt * 32 + l31],          // Q[dim][query]
                    st[kt], 0, 0, 0);
        }
        const float fq = flag[qt * 32 + l31];
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < A::NTL; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float d = ss::mul_rn(st[kt][r], scale);
                if (use_mask && flag[kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half] != fq) d = ss::add_rn(d, -1000.0f);
                st[kt][r] = d;
                mx = fmaxf(mx, d);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < A::NTL; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { st[kt][r] = ss::exp_fast(st[kt][r] - mx); sum = ss::add_rn(sum, st[kt][r]); }
        sum = ss::add_rn(sum, __shfl_xor(sum, 32));
        const float rsum = 1.0f / sum;                // one division per query; p = e * (1/sum) is within 1 ulp of e / sum
        f32x16 ot;
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[r] = 0.f;
        // P V on the fp16 matrix core, fp32-accurate: both operands as two fp16 terms, three products (hi*hi, hi*lo, lo*hi;
        // split_f16.h) -- K = 16 keys per instruction instead of the exact-fp32 MFMA's 2 (this product was 80 % of the kernel's
        // matrix time).  P is scaled by 2^14 (p <= 1), V by the power of two that brings the head group's |max| into
        // [2^14, 2^15); both are undone on the accumulator.  A 16-key step = registers 8 s .. 8 s + 7 of the logits tile: keys
        // {4 half .. +3, 8 + 4 half .. +3} of the step, and V is read in that order.
        const float* vrow = qkv + (64 + hh * HD + (l31 & 7)) * A::TP + 8 * half;     // split V[dim = lane & 7] (rows 8-31 of the tile are not stored)
        const float ps = rsum * 16384.0f;
#pragma unroll
        for (int kt = 0; kt < A::NTL; ++kt)
#pragma unroll
            for (int s8 = 0; s8 < 2; ++s8) {
                float vh[4], vl[4];
                ss::lds_read16(vrow + kt * 32 + 16 * s8, vh);
                ss::lds_read16(vrow + kt * 32 + 16 * s8 + 4, vl);
                unsigned bh_[4], bl_[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    split2_pk_f16(st[kt][8 * s8 + 2 * j] * ps, st[kt][8 * s8 + 2 * j + 1] * ps, bh_[j], bl_[j]);
                const f16x8 a_hi = __builtin_bit_cast(f16x8, make_float4(vh[0], vh[1], vh[2], vh[3]));
                const f16x8 a_lo = __builtin_bit_cast(f16x8, make_float4(vl[0], vl[1], vl[2], vl[3]));
                const f16x8 b_hi = __builtin_bit_cast(f16x8, make_uint4(bh_[0], bh_[1], bh_[2], bh_[3]));
                const f16x8 b_lo = __builtin_bit_cast(f16x8, make_uint4(bl_[0], bl_[1], bl_[2], bl_[3]));
                ot = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, b_lo, ot, 0, 0, 0);
                ot = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo, b_hi, ot, 0, 0, 0);
                ot = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, b_hi, ot, 0, 0, 0);
            }
#pragma unroll
        for (int r = 0; r < 4; ++r) ot[r] *= pv_unscale;
        // rows 0-3 of Ot sit in registers 0-3 of the lower half, rows 4-7 in those of the upper half
        const int t = qt * 32 + l31;
        const int iw = t % bw, ih = (t / bw) % bh, id = t / (bw * bh);
        const int gw = ww * bw + iw, gh = wh * bh + ih, gd = wd * bd + id;
        if (gh < H && gw < W) {
            const size_t pos = (size_t)gd * plane + (size_t)gh * W + gw;
#pragma unroll
            for (int r = 0; r < 4; ++r) yb[(size_t)(g * 32 + hh * HD + 4 * half + r) * vol + pos] = ot[r];
        }
    }
}

template <int T>
int launch_attn_core(const float* qkv, const float* bqkv, float* y, int B, int D, int H, int W, int bd, int bh, int bw,
                     hipStream_t st) {
    const int nwd = D / bd, nwh = ss::ceil_div(H, bh), nww = ss::ceil_div(W, bw);
    const int use_mask = (H % bh != 0) && (W % bw != 0);      // the reference's "-0:" quirk, see launch_attn
    const long long nblk = (long long)nwd * nwh * nww * (128 / (HG * HD));
    if (nblk > 0x7fffffffLL || B > 65535) return SS_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((window_attention_core<T, 128>), dim3((unsigned)nblk, B), dim3(256), 0, st, qkv, bqkv, y, D, H, W, bd,
                       bh, bw, nwh, nww, use_mask);
    return ss::check_launch();
}

template <int T>
int launch_attn(const float* x, const float* wqkv_t, const float* bqkv, const float* wout_t, const float* bout,
                float* out, int B, int D, int H, int W, int bd, int bh, int bw, hipStream_t st) {
    using A = ACfg<T, 128>;
    const int nwd = D / bd, nwh = ss::ceil_div(H, bh), nww = ss::ceil_div(W, bw);
    // reference quirk (models/submodule_other.py:822-823): the mask only separates pad from real
    // tokens when BOTH H and W need padding ("-0:" selects everything otherwise).
    const int use_mask = (H % bh != 0) && (W % bw != 0);
    auto kern = window_attention_kernel<T, 128>;
    if (A::LDS_BYTES > 64 * 1024) {
        if (ss::ensure_dynamic_lds(reinterpret_cast<const void*>(kern), (int)A::LDS_BYTES) != SS_OK) return SS_ERR_LAUNCH;
    }
    const long long nwin = (long long)nwd * nwh * nww;
    if (nwin > 0x7fffffffLL || B > 65535) return SS_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(kern, dim3((unsigned)nwin, B), dim3(256), A::LDS_BYTES, st, x, wqkv_t, bqkv, wout_t, bout, out, D,
                       H, W, bd, bh, bw, nwh, nww, use_mask);
    return ss::check_launch();
}

}  // namespace

extern "C" int ss_window_attention_fwd(const float* x, const float* wqkv_t, const float* bqkv, const float* wout_t,
                                       const float* bout, float* out, int B, int C, int D, int H, int W, int heads,
                                       int bd, int bh, int bw, ss_stream_t stream) {
    SS_REQUIRE(x && wqkv_t && bqkv && wout_t && bout && out);
    SS_REQUIRE(B > 0 && C > 0 && D > 0 && H > 0 && W > 0 && heads > 0 && bd > 0 && bh > 0 && bw > 0);
    SS_REQUIRE(D % bd == 0);
    if (C != 128 || heads != 16) return SS_ERR_UNSUPPORTED;
    const int T = bd * bh * bw;
    hipStream_t st = ss::as_stream(stream);
    if (T == 64) return launch_attn<64>(x, wqkv_t, bqkv, wout_t, bout, out, B, D, H, W, bd, bh, bw, st);
    if (T == 96) return launch_attn<96>(x, wqkv_t, bqkv, wout_t, bout, out, B, D, H, W, bd, bh, bw, st);
    return SS_ERR_UNSUPPORTED;
}

extern "C" int ss_window_attention_core_fwd(const float* qkv, const float* bqkv, float* y, int B, int C, int D, int H,
                                            int W, int heads, int bd, int bh, int bw, ss_stream_t stream) {
    SS_REQUIRE(qkv && bqkv && y);
    SS_REQUIRE(B > 0 && C > 0 && D > 0 && H > 0 && W > 0 && heads > 0 && bd > 0 && bh > 0 && bw > 0);
    SS_REQUIRE(D % bd == 0);
    if (C != 128 || heads != 16) return SS_ERR_UNSUPPORTED;
    const int T = bd * bh * bw;
    hipStream_t st = ss::as_stream(stream);
    if (T == 64) return launch_attn_core<64>(qkv, bqkv, y, B, D, H, W, bd, bh, bw, st);
    if (T == 96) return launch_attn_core<96>(qkv, bqkv, y, B, D, H, W, bd, bh, bw, st);
    return SS_ERR_UNSUPPORTED;
}
