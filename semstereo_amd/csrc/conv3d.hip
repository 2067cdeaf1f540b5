// 3-D convolution stack of the aggregation network as implicit GEMM on the fp32 matrix cores
// (gfx950 v_mfma_f32_32x32x2_f32: exact fp32 products and accumulation, 64 FLOP/clk/SIMD).
//
// Replaces convbn_3d (reference models/submodule_other.py:845-848), BasicConv(is_3d)
// (models/submodule.py:89-116) and the classifier heads (models/SemStereo.py:228-234): Conv3d
// without bias, then the eval-mode BatchNorm3d folded to a per-channel affine, an optional
// residual and an optional ReLU, all in the accumulator epilogue.
//
//   out[co, p] = sum_{ci,tap} W[co, ci, tap] * in[ci, p*stride + tap - pad]
//
// GEMM view: M = Cout (rows of the MFMA tile, from packed weights [Cin][taps][Cout]),
// N = 32 consecutive output columns of one (d,h) row (lanes), K = (ci, tap) with the two k of one
// 32x32x2 step being channels ci, ci+1 of the same tap (lanes 0-31 / 32-63).  NCDHW is the natural
// layout for this: the B operand of a tap is a run of 32 consecutive floats in an LDS halo tile, the
// D fragment stores 128-B row segments.  One workgroup = 4 waves sharing MT*32 output channels;
// each wave owns NT rows of 32 columns.  Per CIT-channel chunk the halo tile and the weight slab
// are staged in LDS once and reused by every tap: (MT+NT) ds_read_b32 per MT*NT MFMAs.
//
// Roofline: 86-850 flop/byte => bound by the fp32 MFMA rate (157 TFLOP/s).
#include <algorithm>
#include <stdlib.h>

#include "common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int KS, int S, int MT, int NT, int TD, int TH, int CIT>
struct Cfg {
    static constexpr int KT = KS * KS * KS;
    static constexpr int PAD = KS / 2;
    static constexpr int ID = (TD - 1) * S + KS;
    static constexpr int IH = (TH - 1) * S + KS;
    static constexpr int IW = 31 * S + KS;
    static constexpr int CS = ID * IH * IW;           // floats per staged input channel
    static constexpr int CO_T = MT * 32;              // output channels per workgroup
    static constexpr int IN_FLOATS = CIT * CS;
    static constexpr int W_FLOATS = CIT * KT * CO_T;
    static constexpr size_t LDS_BYTES = (size_t)(IN_FLOATS + W_FLOATS) * 4;
    static_assert(TD * TH == 4 * NT, "4 waves x NT rows must tile TD x TH");
    static_assert(TH % NT == 0, "a wave's rows stay inside one depth plane");
    static_assert(CIT % 2 == 0, "channels are consumed in pairs (one 32x32x2 k-step)");
};

template <int KS, int S, int MT, int NT, int TD, int TH, int CIT>
__global__ __launch_bounds__(256, 2) void conv3d_mfma(const float* __restrict__ in, const float* __restrict__ wpack,
                                                    const float* __restrict__ scale, const float* __restrict__ shift,
                                                    const float* __restrict__ residual, const float* __restrict__ gate,
                                                    float* __restrict__ out,
                                                    int Cin, int D, int H, int W, int Cout, int Do, int Ho, int Wo,
                                                    int tiles_w, int tiles_h, int relu) {
    using C = Cfg<KS, S, MT, NT, TD, TH, CIT>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* ilds = lds;                    // [CIT][ID][IH][IW]
    float* wlds = lds + C::IN_FLOATS;     // [CIT][KT][CO_T]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    // spatial tile of this workgroup (output coordinates)
    int t = blockIdx.x;
    const int tw = t % tiles_w; t /= tiles_w;
    const int th = t % tiles_h; t /= tiles_h;
    const int td = t;
    const int ow0 = tw * 32, oh0 = th * TH, od0 = td * TD;
    const int co0 = blockIdx.y * C::CO_T;
    const int b = blockIdx.z;
    // input coordinates of the halo tile origin
    const int iw0 = ow0 * S - C::PAD, ih0 = oh0 * S - C::PAD, id0 = od0 * S - C::PAD;

    // this wave's NT rows: n = wave*NT + i -> plane dzw, rows hy0 + i
    const int dzw = (wave * NT) / TH, hy0 = (wave * NT) % TH;
    const int lane_b = half * C::CS + (dzw * S * C::IH + hy0 * S) * C::IW + l31 * S;
    const int lane_a = half * C::KT * C::CO_T + l31;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][i][r] = 0.f;

    const size_t in_plane = (size_t)H * W;
    const float* inb = in + (size_t)b * Cin * D * in_plane;

    // ---- per-thread staging plan, computed once: byte offsets (relative to the chunk base) of the
    // NIN halo-tile elements and NWQ weight quads this thread moves per chunk, plus validity bits.
    // Element e = tid + 256*i of the LDS image; out-of-volume elements load offset 0 and are zeroed.
    constexpr int NIN = (C::IN_FLOATS + 255) / 256;
    constexpr int NWQ = (C::W_FLOATS / 4 + 255) / 256;
    static_assert(NIN <= 64 && NWQ <= 32, "validity masks are 64/32 bits");
    unsigned ioff[NIN], woff[NWQ];
    unsigned long long imask = 0ull;
    unsigned wmask = 0u;
#pragma unroll
    for (int i = 0; i < NIN; ++i) {
        const int e = tid + 256 * i;
        const int wx = e % C::IW;
        int r = e / C::IW;
        const int hy = r % C::IH; r /= C::IH;
        const int dz = r % C::ID;
        const int ci = r / C::ID;
        const int gw = iw0 + wx, gh = ih0 + hy, gd = id0 + dz;
        const bool ok = (e < C::IN_FLOATS) && (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H &&
                        (unsigned)gw < (unsigned)W;
        ioff[i] = ok ? (unsigned)((((size_t)ci * D + gd) * in_plane + (size_t)gh * W + gw) * 4) : 0u;
        imask |= (unsigned long long)ok << i;
    }
#pragma unroll
    for (int j = 0; j < NWQ; ++j) {
        const int e = tid + 256 * j;
        const int q = e % (C::CO_T / 4);
        const int row = e / (C::CO_T / 4);              // ci * KT + tap
        const int co = co0 + q * 4;
        const bool ok = (e < C::W_FLOATS / 4) && (co + 3 < Cout);
        woff[j] = ok ? (unsigned)(((size_t)row * Cout + co) * 4) : 0u;
        wmask |= (unsigned)ok << j;
    }

    float rin[NIN];
    float4 rw[NWQ];
    unsigned long long ilive = imask;
    unsigned wlive = wmask;
    // issue the global loads of chunk ci0 into registers (they complete under the MFMA loop)
    auto prefetch = [&](int ci0) {
        ilive = imask;
        wlive = wmask;
        if (ci0 + CIT > Cin) {        // ragged last chunk: channels >= Cin are zero and must not be read
#pragma unroll
            for (int i = 0; i < NIN; ++i)
                if (ci0 + (tid + 256 * i) / C::CS >= Cin) ilive &= ~(1ull << i);
#pragma unroll
            for (int j = 0; j < NWQ; ++j)
                if (ci0 + (tid + 256 * j) / (C::KT * C::CO_T / 4) >= Cin) wlive &= ~(1u << j);
        }
        const char* ib = reinterpret_cast<const char*>(inb + (size_t)ci0 * D * in_plane);
        const char* wb = reinterpret_cast<const char*>(wpack + (size_t)ci0 * C::KT * Cout);
#pragma unroll
        for (int i = 0; i < NIN; ++i)
            rin[i] = *reinterpret_cast<const float*>(ib + (((ilive >> i) & 1ull) ? ioff[i] : 0u));
#pragma unroll
        for (int j = 0; j < NWQ; ++j)
            rw[j] = *reinterpret_cast<const float4*>(wb + (((wlive >> j) & 1u) ? woff[j] : 0u));
    };

    prefetch(0);
    for (int ci0 = 0; ci0 < Cin; ci0 += CIT) {
        // ---- registers -> LDS: halo tile [CIT][ID][IH][IW] and weight slab [CIT][KT][CO_T] ----
#pragma unroll
        for (int i = 0; i < NIN; ++i) {
            const int e = tid + 256 * i;
            if (e < C::IN_FLOATS) ilds[e] = ((ilive >> i) & 1ull) ? rin[i] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < NWQ; ++j) {
            const int e = tid + 256 * j;
            if (e < C::W_FLOATS / 4)
                *reinterpret_cast<float4*>(&wlds[e * 4]) = ((wlive >> j) & 1u) ? rw[j] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();
        if (ci0 + CIT < Cin) prefetch(ci0 + CIT);

        // ---- CIT/2 x KT k-steps of 32x32x2 ----
#pragma unroll 1
        for (int cp = 0; cp < CIT / 2; ++cp) {
            const float* ap = wlds + lane_a + cp * 2 * C::KT * C::CO_T;
            const float* bp = ilds + lane_b + cp * 2 * C::CS;
#pragma unroll
            for (int kd = 0; kd < KS; ++kd)
#pragma unroll
                for (int kh = 0; kh < KS; ++kh)
#pragma unroll
                    for (int kw = 0; kw < KS; ++kw) {
                        const int tap = (kd * KS + kh) * KS + kw;
                        float a[MT], bv[NT];
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) a[mt] = ap[tap * C::CO_T + mt * 32];
#pragma unroll
                        for (int i = 0; i < NT; ++i) bv[i] = bp[((kd * C::IH) + kh + i * S) * C::IW + kw];
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                            for (int i = 0; i < NT; ++i)
                                acc[mt][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt], bv[i], acc[mt][i], 0, 0, 0);
                    }
        }
        __syncthreads();
    }

    // ---- epilogue: affine (folded BN), residual, ReLU; D fragment -> NCDHW ----
    const int ow = ow0 + l31;
    const int od = od0 + dzw;
    if (ow >= Wo || od >= Do) return;
    const size_t out_plane = (size_t)Ho * Wo;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        // side inputs of 4 fragment rows first (clamped addresses, no branches): their latencies overlap
#pragma unroll
        for (int r0 = 0; r0 < 16; r0 += 4) {
            float sc[4], sh[4], gv[4][NT], rv[4][NT];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = r0 + q;
                const int co = min(co0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, Cout - 1);
                sc[q] = scale ? scale[co] : 1.0f;
                sh[q] = shift ? shift[co] : 0.0f;
#pragma unroll
                for (int i = 0; i < NT; ++i) {
                    const int oh = min(oh0 + hy0 + i, Ho - 1);
                    gv[q][i] = gate ? gate[(((size_t)b * Cout + co) * Ho + oh) * Wo + ow] : 1.0f;
                    rv[q][i] = residual ? residual[(((size_t)b * Cout + co) * Do + od) * out_plane + (size_t)oh * Wo + ow] : 0.0f;
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = r0 + q;
                const int co = co0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (co >= Cout) continue;
#pragma unroll
                for (int i = 0; i < NT; ++i) {
                    const int oh = oh0 + hy0 + i;
                    if (oh >= Ho) continue;
                    float v = ss::add_rn(ss::mul_rn(acc[mt][i][r], sc[q]), sh[q]);
                    if (residual) v = ss::add_rn(v, rv[q][i]);
                    if (relu) v = fmaxf(v, 0.f);
                    if (gate) v = ss::mul_rn(gv[q][i], v);      // channelAtt gate, broadcast over D
                    out[(((size_t)b * Cout + co) * Do + od) * out_plane + (size_t)oh * Wo + ow] = v;
                }
            }
        }
    }
}

// Cout == 1 head (classif.2 / classif_att_.2): too thin for a 32-row MFMA tile, so plain VALU:
// one thread = 4 consecutive output columns, input halo tile in LDS, weights through the scalar path.
template <int TD, int TH, int CIT>
__global__ __launch_bounds__(256) void conv3d_k3_cout1(const float* __restrict__ in, const float* __restrict__ wpack,
                                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                                        float* __restrict__ out, int Cin, int D, int H, int W,
                                                        int tiles_w, int tiles_h, int relu) {
    constexpr int TWC = 32;                      // output columns per tile (8 threads x 4)
    constexpr int ID = TD + 2, IH = TH + 2, IW = TWC + 2;
    constexpr int CS = ID * IH * IW;
    static_assert(TD * TH * (TWC / 4) == 256, "one thread per 4 outputs");
    __shared__ float ilds[CIT * CS];
    const int tid = threadIdx.x;
    int t = blockIdx.x;
    const int tw = t % tiles_w; t /= tiles_w;
    const int th = t % tiles_h; t /= tiles_h;
    const int ow0 = tw * TWC, oh0 = th * TH, od0 = t * TD;
    const int b = blockIdx.z;
    const int q = tid % (TWC / 4), hy = (tid / (TWC / 4)) % TH, dz = tid / ((TWC / 4) * TH);
    const size_t plane = (size_t)H * W;
    const float* inb = in + (size_t)b * Cin * D * plane;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    // staging plan as in conv3d_mfma: offsets once, loads issued one chunk ahead into registers
    constexpr int NIN = (CIT * CS + 255) / 256;
    static_assert(NIN <= 64, "validity mask is 64 bits");
    unsigned ioff[NIN];
    unsigned long long imask = 0ull;
#pragma unroll
    for (int i = 0; i < NIN; ++i) {
        const int e = tid + 256 * i;
        const int wx = e % IW;
        int r = e / IW;
        const int y = r % IH; r /= IH;
        const int z = r % ID;
        const int ci = r / ID;
        const int gw = ow0 - 1 + wx, gh = oh0 - 1 + y, gd = od0 - 1 + z;
        const bool ok = (e < CIT * CS) && (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W;
        ioff[i] = ok ? (unsigned)((((size_t)ci * D + gd) * plane + (size_t)gh * W + gw) * 4) : 0u;
        imask |= (unsigned long long)ok << i;
    }
    float rin[NIN];
    unsigned long long ilive = imask;
    auto prefetch = [&](int ci0) {
        ilive = imask;
        if (ci0 + CIT > Cin) {
#pragma unroll
            for (int i = 0; i < NIN; ++i)
                if (ci0 + (tid + 256 * i) / CS >= Cin) ilive &= ~(1ull << i);
        }
        const char* ib = reinterpret_cast<const char*>(inb + (size_t)ci0 * D * plane);
#pragma unroll
        for (int i = 0; i < NIN; ++i)
            rin[i] = *reinterpret_cast<const float*>(ib + (((ilive >> i) & 1ull) ? ioff[i] : 0u));
    };
    prefetch(0);
    for (int ci0 = 0; ci0 < Cin; ci0 += CIT) {
#pragma unroll
        for (int i = 0; i < NIN; ++i) {
            const int e = tid + 256 * i;
            if (e < CIT * CS) ilds[e] = ((ilive >> i) & 1ull) ? rin[i] : 0.f;
        }
        __syncthreads();
        if (ci0 + CIT < Cin) prefetch(ci0 + CIT);
        const int nci = min(CIT, Cin - ci0);
        for (int ci = 0; ci < nci; ++ci) {
            const float* wp = wpack + (size_t)(ci0 + ci) * 27;      // [Cin][27][1]: uniform -> scalar loads
#pragma unroll
            for (int kd = 0; kd < 3; ++kd)
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    const float* lp = &ilds[((ci * ID + dz + kd) * IH + hy + kh) * IW + q * 4];
                    float x[6];
#pragma unroll
                    for (int j = 0; j < 6; ++j) x[j] = lp[j];
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        const float wv = wp[(kd * 3 + kh) * 3 + kw];
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[j] = fmaf(wv, x[j + kw], acc[j]);
                    }
                }
        }
        __syncthreads();
    }
    const int od = od0 + dz, oh = oh0 + hy, ow = ow0 + q * 4;
    if (od >= D || oh >= H) return;
    const float sc = scale ? scale[0] : 1.f, sh = shift ? shift[0] : 0.f;
    float* op = out + ((size_t)b * D + od) * plane + (size_t)oh * W + ow;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (ow + j >= W) break;
        float v = ss::add_rn(ss::mul_rn(acc[j], sc), sh);
        if (relu) v = fmaxf(v, 0.f);
        op[j] = v;
    }
}

// `patch`: depthwise (1,3,3) stencil, optionally gated by sigmoid(gate[b,c,y,x]).
__global__ __launch_bounds__(256) void depthwise_patch_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                               const float* __restrict__ gate, float* __restrict__ out,
                                                               int C, int D, int H, int W, long long total) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;    // over B*C*D*H*W
    if (i >= total) return;
    const int x = (int)(i % W);
    long long t = i / W;
    const int y = (int)(t % H); t /= H;
    const long long bcd = t;                       // (b*C + c)*D + d
    const long long bc = bcd / D;
    const int c = (int)(bc % C);
    const float* ip = in + bcd * H * W;
    const float* wp = w + c * 9;
    float acc = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int yy = y + ky - 1;
        if ((unsigned)yy >= (unsigned)H) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int xx = x + kx - 1;
            if ((unsigned)xx >= (unsigned)W) continue;
            acc = fmaf(wp[ky * 3 + kx], ip[(long long)yy * W + xx], acc);
        }
    }
    if (gate) {
        const float gl = gate[bc * H * W + (long long)y * W + x];
        acc = ss::mul_rn(1.0f / (1.0f + expf(-gl)), acc);
    }
    out[i] = acc;
}

// W % 4 == 0 form: one thread = 4 consecutive columns (16-B loads/stores); the 3 x 6 input window
// is read as float4 + the two neighbouring scalars per row.
__global__ __launch_bounds__(256) void depthwise_patch_v4(const float* __restrict__ in, const float* __restrict__ w,
                                                           const float* __restrict__ gate, float* __restrict__ out,
                                                           int C, int D, int H, int W, long long nquads) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;    // over B*C*D*H*(W/4)
    if (i >= nquads) return;
    const int WQ = W / 4;
    const int x0 = (int)(i % WQ) * 4;
    long long t = i / WQ;
    const int y = (int)(t % H); t /= H;
    const long long bcd = t;
    const long long bc = bcd / D;
    const int c = (int)(bc % C);
    float wv[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wv[k] = w[c * 9 + k];
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    // all nine input loads unconditional, through a buffer descriptor over this (b, c, d) plane: rows and neighbours
    // outside it get an offset beyond the buffer and read 0 (under a branch every load is followed by its own
    // s_waitcnt vmcnt(0): three exposed round trips per thread instead of one)
    const __amdgpu_buffer_rsrc_t ires = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in + bcd * H * W), 0, H * W * 4, 0x00020000);
    float4 m[3];
    float xl[3], xr[3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int yy = y + ky - 1;
        const bool rok = (unsigned)yy < (unsigned)H;
        const unsigned ro = (unsigned)((yy * W + x0) * 4);
        m[ky] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ires, (int)(rok ? ro : 0x80000000u), 0, 0));
        xl[ky] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ires, (int)((rok && x0 > 0) ? ro - 4 : 0x80000000u), 0, 0));
        xr[ky] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ires, (int)((rok && x0 + 4 < W) ? ro + 16 : 0x80000000u), 0, 0));
    }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        if ((unsigned)(y + ky - 1) >= (unsigned)H) continue;      // (the reference skips the row: no fma with zeros, -0.0 stays -0.0)
        const float x[6] = {xl[ky], m[ky].x, m[ky].y, m[ky].z, m[ky].w, xr[ky]};
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = fmaf(wv[ky * 3 + kx], x[j + kx], acc[j]);
    }
    if (gate) {
        const float4 g = *reinterpret_cast<const float4*>(gate + bc * H * W + (long long)y * W + x0);
        acc[0] = ss::mul_rn(1.0f / (1.0f + expf(-g.x)), acc[0]);
        acc[1] = ss::mul_rn(1.0f / (1.0f + expf(-g.y)), acc[1]);
        acc[2] = ss::mul_rn(1.0f / (1.0f + expf(-g.z)), acc[2]);
        acc[3] = ss::mul_rn(1.0f / (1.0f + expf(-g.w)), acc[3]);
    }
    *reinterpret_cast<float4*>(out + bcd * H * W + (long long)y * W + x0) = make_float4(acc[0], acc[1], acc[2], acc[3]);
}

__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ wpack, int Cout, int Cin, int KT,
                                    int transposed, long long total) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;   // over [Cin][KT][Cout]
    if (i >= total) return;
    const int co = (int)(i % Cout);
    const long long r = i / Cout;
    const int tap = (int)(r % KT);
    const int ci = (int)(r / KT);
    const long long src = transposed ? ((long long)ci * Cout + co) * KT + tap : ((long long)co * Cin + ci) * KT + tap;
    wpack[i] = w[src];
}

template <int KS, int S, int MT, int NT, int TD, int TH, int CIT>
int launch_conv(const float* in, const float* wpack, const float* scale, const float* shift, const float* residual,
                const float* gate, float* out, int B, int Cin, int D, int H, int W, int Cout, int relu, hipStream_t st) {
    using C = Cfg<KS, S, MT, NT, TD, TH, CIT>;
    const int Do = (D + 2 * C::PAD - KS) / S + 1, Ho = (H + 2 * C::PAD - KS) / S + 1, Wo = (W + 2 * C::PAD - KS) / S + 1;
    const int tiles_w = ss::ceil_div(Wo, 32), tiles_h = ss::ceil_div(Ho, TH), tiles_d = ss::ceil_div(Do, TD);
    const long long nt = (long long)tiles_w * tiles_h * tiles_d;
    if (nt > 0x7fffffffLL || B > 65535) return SS_ERR_UNSUPPORTED;
    auto kern = conv3d_mfma<KS, S, MT, NT, TD, TH, CIT>;
    if (C::LDS_BYTES > 64 * 1024) {
        if (ss::ensure_dynamic_lds(reinterpret_cast<const void*>(kern), (int)C::LDS_BYTES) != SS_OK) return SS_ERR_LAUNCH;
    }
    dim3 grid((unsigned)nt, ss::ceil_div(Cout, C::CO_T), B);
    hipLaunchKernelGGL(kern, grid, dim3(256), C::LDS_BYTES, st, in, wpack, scale, shift, residual, gate, out, Cin, D, H, W,
                       Cout, Do, Ho, Wo, tiles_w, tiles_h, relu);
    return ss::check_launch();
}

}  // namespace

extern "C" int ss_conv3d_fwd(const float* in, const float* wpack, const float* scale, const float* shift,
                             const float* residual, const float* gate, float* out, int B, int Cin, int D, int H, int W,
                             int Cout, int k,
                             int stride, int relu, ss_stream_t stream) {
    SS_REQUIRE(in && wpack && out);
    SS_REQUIRE(B > 0 && Cin > 0 && D > 0 && H > 0 && W > 0 && Cout > 0);
    SS_REQUIRE((k == 1 || k == 3) && (stride == 1 || stride == 2));
    hipStream_t st = ss::as_stream(stream);
    if (Cout == 1 && k == 3 && stride == 1 && residual == nullptr && gate == nullptr) {
        constexpr int TD = 4, TH = 8;
        const int tiles_w = ss::ceil_div(W, 32), tiles_h = ss::ceil_div(H, TH), tiles_d = ss::ceil_div(D, TD);
        const long long nt = (long long)tiles_w * tiles_h * tiles_d;
        if (nt > 0x7fffffffLL || B > 65535) return SS_ERR_UNSUPPORTED;
        hipLaunchKernelGGL((conv3d_k3_cout1<TD, TH, 4>), dim3((unsigned)nt, 1, B), dim3(256), 0, st, in, wpack, scale,
                           shift, out, Cin, D, H, W, tiles_w, tiles_h, relu);
        return ss::check_launch();
    }
    if (Cout % 4 != 0) return SS_ERR_UNSUPPORTED;     // weight slabs are moved as float4
    // Tile selection.  A workgroup covers MT*32 output channels x (TD x TH rows of 32 columns).
    // Candidates are ordered from the biggest tile (most operand reuse) to the smallest; the first
    // one that still gives every CU at least two workgroups wins (one resident workgroup cannot
    // overlap its own staging with its MFMAs), otherwise the one with the most workgroups.
    // SS_CONV_TILE=<index> forces a candidate (tuning aid; results are identical for every tile).
    const int pad = k / 2;
    const int Do = (D + 2 * pad - k) / stride + 1, Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    struct Cand { int mt, nt, td, th; };
    auto blocks_of = [&](const Cand& c) {
        return (long long)ss::ceil_div(Wo, 32) * ss::ceil_div(Ho, c.th) * ss::ceil_div(Do, c.td) *
               ss::ceil_div(Cout, c.mt * 32) * B;
    };
    auto pick = [&](const Cand* cands, int n) {
        const int forced = ss::tuning().conv_tile;
        if (forced >= 0 && forced < n) return forced;
        int best = 0;
        long long best_blocks = -1;
        for (int i = 0; i < n; ++i) {
            if (cands[i].mt * 32 > Cout && cands[i].mt > 1) continue;      // would waste MFMA rows
            const long long nb = blocks_of(cands[i]);
            if (nb >= 2 * 256) return i;
            if (nb > best_blocks) { best_blocks = nb; best = i; }
        }
        return best;
    };
#define SS_CONV(KS, S, MT, NT, TD, TH, CIT) \
    return launch_conv<KS, S, MT, NT, TD, TH, CIT>(in, wpack, scale, shift, residual, gate, out, B, Cin, D, H, W, Cout, relu, st)
    if (k == 3 && stride == 1) {
        static const Cand c[] = {{1, 4, 2, 8}, {2, 2, 1, 8}, {2, 1, 1, 4}, {1, 2, 1, 8}, {1, 1, 1, 4}};
        switch (pick(c, 5)) {
            case 0: SS_CONV(3, 1, 1, 4, 2, 8, 4);
            case 1: SS_CONV(3, 1, 2, 2, 1, 8, 4);
            case 2: SS_CONV(3, 1, 2, 1, 1, 4, 4);
            case 3: SS_CONV(3, 1, 1, 2, 1, 8, 4);
            default: SS_CONV(3, 1, 1, 1, 1, 4, 8);
        }
    }
    if (k == 3 && stride == 2) {
        static const Cand c[] = {{2, 2, 1, 8}, {2, 1, 1, 4}, {1, 2, 1, 8}, {1, 1, 1, 4}};
        switch (pick(c, 4)) {
            case 0: SS_CONV(3, 2, 2, 2, 1, 8, 2);
            case 1: SS_CONV(3, 2, 2, 1, 1, 4, 4);
            case 2: SS_CONV(3, 2, 1, 2, 1, 8, 2);
            default: SS_CONV(3, 2, 1, 1, 1, 4, 4);
        }
    }
    if (k == 1 && stride == 1) {
        static const Cand c[] = {{2, 2, 1, 8}, {1, 4, 2, 8}, {1, 1, 1, 4}};
        switch (pick(c, 3)) {
            case 0: SS_CONV(1, 1, 2, 2, 1, 8, 16);
            case 1: SS_CONV(1, 1, 1, 4, 2, 8, 8);
            default: SS_CONV(1, 1, 1, 1, 1, 4, 16);
        }
    }
#undef SS_CONV
    return SS_ERR_UNSUPPORTED;
}

extern "C" int ss_pack_conv3d_weights(const float* w, float* wpack, int Cout, int Cin, int k, int transposed,
                                      ss_stream_t stream) {
    SS_REQUIRE(w && wpack && Cout > 0 && Cin > 0 && k > 0);
    const int KT = k * k * k;
    const long long total = (long long)Cin * KT * Cout;
    hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)ss::ceil_div_ll(total, 256)), dim3(256), 0,
                       ss::as_stream(stream), w, wpack, Cout, Cin, KT, transposed, total);
    return ss::check_launch();
}

extern "C" int ss_depthwise_patch_fwd(const float* in, const float* w, const float* gate, float* out, int B, int C,
                                      int D, int H, int W, ss_stream_t stream) {
    SS_REQUIRE(in && w && out);
    SS_REQUIRE(B > 0 && C > 0 && D > 0 && H > 0 && W > 0);
    const long long total = (long long)B * C * D * H * W;
    const uintptr_t bits = reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(gate);
    if (W % 4 == 0 && (bits & 15) == 0) {
        const long long nq = total / 4, qb = ss::ceil_div_ll(nq, 256);
        if (qb > 0x7fffffffLL) return SS_ERR_UNSUPPORTED;
        hipLaunchKernelGGL(depthwise_patch_v4, dim3((unsigned)qb), dim3(256), 0, ss::as_stream(stream), in, w, gate, out, C,
                           D, H, W, nq);
        return ss::check_launch();
    }
    const long long blocks = ss::ceil_div_ll(total, 256);
    if (blocks > 0x7fffffffLL) return SS_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(depthwise_patch_kernel, dim3((unsigned)blocks), dim3(256), 0, ss::as_stream(stream), in, w, gate,
                       out, C, D, H, W, total);
    return ss::check_launch();
}
