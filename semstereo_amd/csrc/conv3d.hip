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
__global__ __launch_bounds__(256) void conv3d_mfma(const float* __restrict__ in, const float* __restrict__ wpack,
                                                    const float* __restrict__ scale, const float* __restrict__ shift,
                                                    const float* __restrict__ residual, float* __restrict__ out,
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

    for (int ci0 = 0; ci0 < Cin; ci0 += CIT) {
        // ---- stage the input halo tile (zero outside the volume / beyond Cin) ----
        for (int e = tid; e < C::IN_FLOATS; e += 256) {
            const int wx = e % C::IW;
            int r = e / C::IW;
            const int hy = r % C::IH; r /= C::IH;
            const int dz = r % C::ID;
            const int ci = r / C::ID;
            const int gw = iw0 + wx, gh = ih0 + hy, gd = id0 + dz, gc = ci0 + ci;
            float v = 0.f;
            if (gc < Cin && (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W)
                v = inb[((size_t)gc * D + gd) * in_plane + (size_t)gh * W + gw];
            ilds[e] = v;
        }
        // ---- stage the weight slab [CIT][KT][CO_T] (zero beyond Cin / Cout) ----
        for (int e = tid; e < C::W_FLOATS / 4; e += 256) {
            const int q = e % (C::CO_T / 4);
            const int row = e / (C::CO_T / 4);          // ci * KT + tap
            const int ci = row / C::KT;
            const int co = co0 + q * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ci0 + ci < Cin) {
                const float* wp = wpack + ((size_t)(ci0 + ci) * C::KT + (row - ci * C::KT)) * Cout + co;
                if (co + 3 < Cout && (Cout & 3) == 0) {
                    v = *reinterpret_cast<const float4*>(wp);
                } else {
                    if (co + 0 < Cout) v.x = wp[0];
                    if (co + 1 < Cout) v.y = wp[1];
                    if (co + 2 < Cout) v.z = wp[2];
                    if (co + 3 < Cout) v.w = wp[3];
                }
            }
            *reinterpret_cast<float4*>(&wlds[e * 4]) = v;
        }
        __syncthreads();

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
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (co >= Cout) continue;
            const float sc = scale ? scale[co] : 1.0f;
            const float sh = shift ? shift[co] : 0.0f;
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const int oh = oh0 + hy0 + i;
                if (oh >= Ho) continue;
                const size_t o = (((size_t)b * Cout + co) * Do + od) * out_plane + (size_t)oh * Wo + ow;
                float v = acc[mt][i][r];
                v = ss::add_rn(ss::mul_rn(v, sc), sh);
                if (residual) v = ss::add_rn(v, residual[o]);
                if (relu) v = fmaxf(v, 0.f);
                out[o] = v;
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
    for (int ci0 = 0; ci0 < Cin; ci0 += CIT) {
        for (int e = tid; e < CIT * CS; e += 256) {
            const int wx = e % IW;
            int r = e / IW;
            const int y = r % IH; r /= IH;
            const int z = r % ID;
            const int ci = r / ID;
            const int gw = ow0 - 1 + wx, gh = oh0 - 1 + y, gd = od0 - 1 + z, gc = ci0 + ci;
            float v = 0.f;
            if (gc < Cin && (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W)
                v = inb[((size_t)gc * D + gd) * plane + (size_t)gh * W + gw];
            ilds[e] = v;
        }
        __syncthreads();
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
                float* out, int B, int Cin, int D, int H, int W, int Cout, int relu, hipStream_t st) {
    using C = Cfg<KS, S, MT, NT, TD, TH, CIT>;
    const int Do = (D + 2 * C::PAD - KS) / S + 1, Ho = (H + 2 * C::PAD - KS) / S + 1, Wo = (W + 2 * C::PAD - KS) / S + 1;
    const int tiles_w = ss::ceil_div(Wo, 32), tiles_h = ss::ceil_div(Ho, TH), tiles_d = ss::ceil_div(Do, TD);
    const long long nt = (long long)tiles_w * tiles_h * tiles_d;
    if (nt > 0x7fffffffLL || B > 65535) return SS_ERR_UNSUPPORTED;
    auto kern = conv3d_mfma<KS, S, MT, NT, TD, TH, CIT>;
    if (C::LDS_BYTES > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)C::LDS_BYTES);
        if (e != hipSuccess) { ss::note_hip_error(e); return SS_ERR_LAUNCH; }
    }
    dim3 grid((unsigned)nt, ss::ceil_div(Cout, C::CO_T), B);
    hipLaunchKernelGGL(kern, grid, dim3(256), C::LDS_BYTES, st, in, wpack, scale, shift, residual, out, Cin, D, H, W,
                       Cout, Do, Ho, Wo, tiles_w, tiles_h, relu);
    return ss::check_launch();
}

}  // namespace

extern "C" int ss_conv3d_fwd(const float* in, const float* wpack, const float* scale, const float* shift,
                             const float* residual, float* out, int B, int Cin, int D, int H, int W, int Cout, int k,
                             int stride, int relu, ss_stream_t stream) {
    SS_REQUIRE(in && wpack && out);
    SS_REQUIRE(B > 0 && Cin > 0 && D > 0 && H > 0 && W > 0 && Cout > 0);
    SS_REQUIRE((k == 1 || k == 3) && (stride == 1 || stride == 2));
    hipStream_t st = ss::as_stream(stream);
    if (Cout == 1 && k == 3 && stride == 1 && residual == nullptr) {
        constexpr int TD = 4, TH = 8;
        const int tiles_w = ss::ceil_div(W, 32), tiles_h = ss::ceil_div(H, TH), tiles_d = ss::ceil_div(D, TD);
        const long long nt = (long long)tiles_w * tiles_h * tiles_d;
        if (nt > 0x7fffffffLL || B > 65535) return SS_ERR_UNSUPPORTED;
        hipLaunchKernelGGL((conv3d_k3_cout1<TD, TH, 4>), dim3((unsigned)nt, 1, B), dim3(256), 0, st, in, wpack, scale,
                           shift, out, Cin, D, H, W, tiles_w, tiles_h, relu);
        return ss::check_launch();
    }
    const bool wide = Cout > 32;     // 64 output channels per workgroup when there are that many
    if (k == 3 && stride == 1) {
        if (wide) return launch_conv<3, 1, 2, 2, 1, 8, 4>(in, wpack, scale, shift, residual, out, B, Cin, D, H, W, Cout, relu, st);
        return launch_conv<3, 1, 1, 4, 2, 8, 8>(in, wpack, scale, shift, residual, out, B, Cin, D, H, W, Cout, relu, st);
    }
    if (k == 3 && stride == 2) {
        if (wide) return launch_conv<3, 2, 2, 2, 1, 8, 2>(in, wpack, scale, shift, residual, out, B, Cin, D, H, W, Cout, relu, st);
        return launch_conv<3, 2, 1, 2, 1, 8, 4>(in, wpack, scale, shift, residual, out, B, Cin, D, H, W, Cout, relu, st);
    }
    if (k == 1 && stride == 1) {
        if (wide) return launch_conv<1, 1, 2, 2, 1, 8, 16>(in, wpack, scale, shift, residual, out, B, Cin, D, H, W, Cout, relu, st);
        return launch_conv<1, 1, 1, 4, 2, 8, 16>(in, wpack, scale, shift, residual, out, B, Cin, D, H, W, Cout, relu, st);
    }
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
    const long long blocks = ss::ceil_div_ll(total, 256);
    if (blocks > 0x7fffffffLL) return SS_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(depthwise_patch_kernel, dim3((unsigned)blocks), dim3(256), 0, ss::as_stream(stream), in, w, gate,
                       out, C, D, H, W, total);
    return ss::check_launch();
}
