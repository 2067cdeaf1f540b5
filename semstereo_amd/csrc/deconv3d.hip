// ConvTranspose3d(k=3, s=2, p=1, output_padding=1, bias=False) + folded BatchNorm3d, fused with the
// 1x1x1 skip projection and ReLU of hourglass.forward (reference models/SemStereo.py:124-130,
// 141-142 / 163-169, 180-181) on the fp32 matrix cores (gfx950 v_mfma_f32_32x32x2_f32).
//
//   out[co, o] = sum_{ci, k : o = 2i - 1 + k} Wt[ci, co, k] * in[ci, i]   (+ sum_cs Ws[cs, co] * skip[cs, o])
//
// Per dimension an even output o = 2j takes only (k=1, i=j); an odd output o = 2j+1 takes
// (k=0, i=j+1) and (k=2, i=j).  So the 8 output parity classes are 8 small stride-1 convolutions
// over the INPUT grid with 1..8 taps each (27 in total: no multiplications by inserted zeros).
// A workgroup owns one (pd, ph) class of an input-space tile; every wave accumulates both pw
// parities of its rows, so each lane ends up with the output pair (2j, 2j+1) and stores 8 bytes:
// full 256-B row segments per wave store.  GEMM mapping as in conv3d.hip (M = Cout, N = 32 input
// columns, k-pair = channels ci, ci+1).
#include <algorithm>

#include "common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int MT, int NT, int TD, int TH, int CIT, int CST>
struct DCfg {
    static constexpr int ID = TD + 1, IH = TH + 1, IW = 33;
    static constexpr int CS = ID * IH * IW;
    static constexpr int CO_T = MT * 32;
    static constexpr int IN_FLOATS = CIT * CS;
    static constexpr int W_FLOATS = CIT * 27 * CO_T;
    static constexpr int SK_CS = TD * TH * 64;           // floats per staged skip channel
    static constexpr int SK_FLOATS = CST * SK_CS + CST * CO_T;
    static constexpr int LDS_FLOATS = (IN_FLOATS + W_FLOATS) > SK_FLOATS ? (IN_FLOATS + W_FLOATS) : SK_FLOATS;
    static_assert(TD * TH == 4 * NT && TH % NT == 0 && CIT % 2 == 0 && CST % 2 == 0, "tile shape");
};

// taps of one dimension for output parity P: (kernel index, input offset)
__host__ __device__ constexpr int ntaps(int parity) { return parity == 0 ? 1 : 2; }
__host__ __device__ constexpr int tap_k(int parity, int a) { return parity == 0 ? 1 : (a == 0 ? 0 : 2); }
__host__ __device__ constexpr int tap_off(int parity, int a) { return parity == 0 ? 0 : (a == 0 ? 1 : 0); }

template <int PD, int PH, int MT, int NT, int TD, int TH, int CIT, int CST>
__device__ __forceinline__ void deconv_body(const float* __restrict__ in, const float* __restrict__ wpack,
                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                            const float* __restrict__ skip, const float* __restrict__ skip_w,
                                            float* __restrict__ out, float* lds, int Cin, int D, int H, int W, int Cout,
                                            int Cs, int id0, int ih0, int iw0, int co0, int b, int relu) {
    using C = DCfg<MT, NT, TD, TH, CIT, CST>;
    float* ilds = lds;
    float* wlds = lds + C::IN_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int dzw = (wave * NT) / TH, hy0 = (wave * NT) % TH;
    const int lane_b = half * C::CS + (dzw * C::IH + hy0) * C::IW + l31;
    const int lane_a = half * 27 * C::CO_T + l31;

    f32x16 acc[MT][NT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[mt][i][0][r] = 0.f; acc[mt][i][1][r] = 0.f; }

    const size_t in_plane = (size_t)H * W;
    const float* inb = in + (size_t)b * Cin * D * in_plane;
    for (int ci0 = 0; ci0 < Cin; ci0 += CIT) {
        for (int e = tid; e < C::IN_FLOATS; e += 256) {
            const int wx = e % C::IW;
            int r = e / C::IW;
            const int hy = r % C::IH; r /= C::IH;
            const int dz = r % C::ID;
            const int ci = r / C::ID;
            const int gw = iw0 + wx, gh = ih0 + hy, gd = id0 + dz, gc = ci0 + ci;
            float v = 0.f;
            if (gc < Cin && gd < D && gh < H && gw < W) v = inb[((size_t)gc * D + gd) * in_plane + (size_t)gh * W + gw];
            ilds[e] = v;
        }
        for (int e = tid; e < C::W_FLOATS / 4; e += 256) {
            const int q = e % (C::CO_T / 4);
            const int row = e / (C::CO_T / 4);
            const int ci = row / 27;
            const int co = co0 + q * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ci0 + ci < Cin) {
                const float* wp = wpack + ((size_t)(ci0 + ci) * 27 + (row - ci * 27)) * Cout + co;
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
#pragma unroll 1
        for (int cp = 0; cp < CIT / 2; ++cp) {
            const float* ap = wlds + lane_a + cp * 2 * 27 * C::CO_T;
            const float* bp = ilds + lane_b + cp * 2 * C::CS;
#pragma unroll
            for (int a = 0; a < ntaps(PD); ++a)
#pragma unroll
                for (int c = 0; c < ntaps(PH); ++c) {
                    const int kd = tap_k(PD, a), od = tap_off(PD, a), kh = tap_k(PH, c), oh = tap_off(PH, c);
                    float w0[MT], w1[MT], w2[MT], x0[NT], x1[NT];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        w0[mt] = ap[((kd * 3 + kh) * 3 + 0) * C::CO_T + mt * 32];
                        w1[mt] = ap[((kd * 3 + kh) * 3 + 1) * C::CO_T + mt * 32];
                        w2[mt] = ap[((kd * 3 + kh) * 3 + 2) * C::CO_T + mt * 32];
                    }
#pragma unroll
                    for (int i = 0; i < NT; ++i) {
                        x0[i] = bp[(od * C::IH + oh + i) * C::IW];
                        x1[i] = bp[(od * C::IH + oh + i) * C::IW + 1];
                    }
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int i = 0; i < NT; ++i) {
                            acc[mt][i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[mt], x0[i], acc[mt][i][0], 0, 0, 0);
                            acc[mt][i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0[mt], x1[i], acc[mt][i][1], 0, 0, 0);
                            acc[mt][i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w2[mt], x0[i], acc[mt][i][1], 0, 0, 0);
                        }
                }
        }
        __syncthreads();
    }

    const int Do = 2 * D, Ho = 2 * H, Wo = 2 * W;
    const size_t out_plane = (size_t)Ho * Wo;
    if (skip != nullptr) {
        // ---- 1x1x1 projection of the skip tensor at this class's output positions ----
        float* slds = lds;                          // [CST][TD][TH][64]
        float* swl = lds + CST * C::SK_CS;          // [CST][CO_T]
        const float* skb = skip + (size_t)b * Cs * Do * out_plane;
        const int lane_s = half * C::SK_CS + (dzw * TH + hy0) * 64 + 2 * l31;
        for (int cs0 = 0; cs0 < Cs; cs0 += CST) {
            for (int e = tid; e < CST * C::SK_CS; e += 256) {
                const int wx = e % 64;
                int r = e / 64;
                const int hy = r % TH; r /= TH;
                const int dz = r % TD;
                const int cs = r / TD;
                const int gw = 2 * iw0 + wx, gh = 2 * (ih0 + hy) + PH, gd = 2 * (id0 + dz) + PD, gc = cs0 + cs;
                float v = 0.f;
                if (gc < Cs && gd < Do && gh < Ho && gw < Wo) v = skb[((size_t)gc * Do + gd) * out_plane + (size_t)gh * Wo + gw];
                slds[e] = v;
            }
            for (int e = tid; e < CST * C::CO_T; e += 256) {
                const int co = co0 + e % C::CO_T, cs = cs0 + e / C::CO_T;
                swl[e] = (cs < Cs && co < Cout) ? skip_w[(size_t)cs * Cout + co] : 0.f;
            }
            __syncthreads();
#pragma unroll
            for (int cp = 0; cp < CST / 2; ++cp) {
                float wv[MT], x0[NT], x1[NT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) wv[mt] = swl[(cp * 2 + half) * C::CO_T + mt * 32 + l31];
#pragma unroll
                for (int i = 0; i < NT; ++i) {
                    x0[i] = slds[lane_s + cp * 2 * C::SK_CS + i * 64];
                    x1[i] = slds[lane_s + cp * 2 * C::SK_CS + i * 64 + 1];
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int i = 0; i < NT; ++i) {
                        acc[mt][i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[mt], x0[i], acc[mt][i][0], 0, 0, 0);
                        acc[mt][i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[mt], x1[i], acc[mt][i][1], 0, 0, 0);
                    }
            }
            __syncthreads();
        }
    }

    // ---- epilogue: each lane owns the output pair (2j, 2j+1) of its rows ----
    const int jw = iw0 + l31, jd = id0 + dzw;
    if (jw >= W || jd >= D) return;
    const int od = 2 * jd + PD;
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
                const int jh = ih0 + hy0 + i;
                if (jh >= H) continue;
                const int oh = 2 * jh + PH;
                float v0 = ss::add_rn(ss::mul_rn(acc[mt][i][0][r], sc), sh);
                float v1 = ss::add_rn(ss::mul_rn(acc[mt][i][1][r], sc), sh);
                if (relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
                float* op = out + (((size_t)b * Cout + co) * Do + od) * out_plane + (size_t)oh * Wo + 2 * jw;
                *reinterpret_cast<float2*>(op) = make_float2(v0, v1);
            }
        }
    }
}

template <int MT, int NT, int TD, int TH, int CIT, int CST>
__global__ __launch_bounds__(256) void deconv3d_mfma(const float* __restrict__ in, const float* __restrict__ wpack,
                                                      const float* __restrict__ scale, const float* __restrict__ shift,
                                                      const float* __restrict__ skip, const float* __restrict__ skip_w,
                                                      float* __restrict__ out, int Cin, int D, int H, int W, int Cout,
                                                      int Cs, int tiles_w, int tiles_h, int relu) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    int t = blockIdx.x;
    const int cls = t & 3; t >>= 2;               // (pd, ph) parity class of this workgroup
    const int tw = t % tiles_w; t /= tiles_w;
    const int th = t % tiles_h; t /= tiles_h;
    const int iw0 = tw * 32, ih0 = th * TH, id0 = t * TD;
    const int co0 = blockIdx.y * MT * 32;
    const int b = blockIdx.z;
#define SS_DECONV_CASE(PD, PH)                                                                                       \
    deconv_body<PD, PH, MT, NT, TD, TH, CIT, CST>(in, wpack, scale, shift, skip, skip_w, out, lds, Cin, D, H, W, Cout, \
                                                  Cs, id0, ih0, iw0, co0, b, relu)
    switch (cls) {
        case 0: SS_DECONV_CASE(0, 0); break;
        case 1: SS_DECONV_CASE(0, 1); break;
        case 2: SS_DECONV_CASE(1, 0); break;
        default: SS_DECONV_CASE(1, 1); break;
    }
#undef SS_DECONV_CASE
}

template <int MT, int NT, int TD, int TH, int CIT, int CST>
int launch_deconv(const float* in, const float* wpack, const float* scale, const float* shift, const float* skip,
                  const float* skip_w, float* out, int B, int Cin, int D, int H, int W, int Cout, int Cs, int relu,
                  hipStream_t st) {
    using C = DCfg<MT, NT, TD, TH, CIT, CST>;
    const int tiles_w = ss::ceil_div(W, 32), tiles_h = ss::ceil_div(H, TH), tiles_d = ss::ceil_div(D, TD);
    const long long nt = 4LL * tiles_w * tiles_h * tiles_d;
    if (nt > 0x7fffffffLL || B > 65535) return SS_ERR_UNSUPPORTED;
    auto kern = deconv3d_mfma<MT, NT, TD, TH, CIT, CST>;
    const size_t lds = (size_t)C::LDS_FLOATS * 4;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { ss::note_hip_error(e); return SS_ERR_LAUNCH; }
    }
    dim3 grid((unsigned)nt, ss::ceil_div(Cout, C::CO_T), B);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, in, wpack, scale, shift, skip, skip_w, out, Cin, D, H, W, Cout, Cs,
                       tiles_w, tiles_h, relu);
    return ss::check_launch();
}

}  // namespace

extern "C" int ss_deconv3d_fwd(const float* in, const float* wpack, const float* scale, const float* shift,
                               const float* skip, const float* skip_wpack, const float* skip_scale,
                               const float* skip_shift, float* out, int B, int Cin, int D, int H, int W, int Cout, int Cs,
                               int relu, ss_stream_t stream) {
    SS_REQUIRE(in && wpack && out);
    SS_REQUIRE(B > 0 && Cin > 0 && D > 0 && H > 0 && W > 0 && Cout > 0);
    SS_REQUIRE((skip == nullptr) || (skip_wpack != nullptr && Cs > 0));
    // One accumulator serves both products, so per-branch affines must already be folded into the
    // packed weights by the caller (semstereo_amd/modules.py does); only a common affine is applied here.
    if (skip_scale != nullptr || skip_shift != nullptr) return SS_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(out) & 7) != 0) return SS_ERR_INVALID;
    hipStream_t st = ss::as_stream(stream);
    if (Cout > 32)
        return launch_deconv<2, 1, 1, 4, 4, 8>(in, wpack, scale, shift, skip, skip_wpack, out, B, Cin, D, H, W, Cout, Cs, relu, st);
    return launch_deconv<1, 2, 1, 8, 4, 8>(in, wpack, scale, shift, skip, skip_wpack, out, B, Cin, D, H, W, Cout, Cs, relu, st);
}
