// ConvTranspose3d(k=3, s=2, p=1, output_padding=1, bias=False) + folded BatchNorm3d, fused with the
// 1x1x1 skip projection and ReLU of hourglass.forward (reference models/SemStereo.py:124-130,
// 141-142 / 163-169, 180-181) on the fp32 matrix cores (gfx950 v_mfma_f32_32x32x2_f32).
//
//   out[co, o] = sum_{ci, k : o = 2i - 1 + k} Wt[ci, co, k] * in[ci, i]   (+ sum_cs Ws[cs, co] * skip[cs, o])
//
// Per dimension an even output o = 2j takes only (k=1, i=j); an odd output o = 2j+1 takes
// (k=0, i=j+1) and (k=2, i=j).  So every one of the 27 taps feeds exactly ONE of the 8 output
// parity classes (pd,ph,pw) = (kd!=1, kh!=1, kw!=1), reading the input at offset
// (kd==0, kh==0, kw==0): no multiplications by inserted zeros, and the whole transposed
// convolution is 27 MFMAs per channel pair over an INPUT-space tile, exactly like a stride-1 conv.
// A wave owns one row of 32 input columns and keeps all 8 parity accumulators (128 registers), so
// each lane ends with its 2x2x2 output cube and stores (2j, 2j+1) pairs: 256-B row segments per
// wave store.  GEMM mapping as in conv3d.hip (M = 32 output channels, N = 32 input columns,
// k-pair = channels ci, ci+1); staging is register-prefetched one chunk ahead.
#include <algorithm>
#include <stdlib.h>

#include "common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int TD, int TH, int CIT>
struct DCfg {
    static constexpr int ID = TD + 1, IH = TH + 1, IW = 33;
    static constexpr int CS = ID * IH * IW;
    static constexpr int IN_FLOATS = CIT * CS;
    static constexpr int W_FLOATS = CIT * 27 * 32;
    static constexpr int LDS_FLOATS = IN_FLOATS + W_FLOATS;
    static_assert(TD * TH == 4 && CIT % 2 == 0, "4 waves x one row each");
};

// PD = -1: all 8 output parity classes; PD = 0 / 1: only the classes of even / odd output planes (9 / 18 of
// the 27 taps, 4 accumulators) -- used to double the number of workgroups of layers too small to fill the chip
template <int TD, int TH, int CIT, bool HAS_SKIP, int PD>
__device__ __forceinline__ void deconv3d_body(const int co_tile, const float* __restrict__ in, const float* __restrict__ wpack,
                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                         const float* __restrict__ skip, const float* __restrict__ skip_w,
                                                         float* __restrict__ out, int Cin, int D, int H, int W, int Cout,
                                                         int Cs, int tiles_w, int tiles_h, int relu) {
    using C = DCfg<TD, TH, CIT>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* ilds = lds;                    // [CIT][ID][IH][IW]
    float* wlds = lds + C::IN_FLOATS;     // [CIT][27][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    int t = blockIdx.x;
    const int tw = t % tiles_w; t /= tiles_w;
    const int th = t % tiles_h; t /= tiles_h;
    const int iw0 = tw * 32, ih0 = th * TH, id0 = t * TD;     // input-space tile origin
    const int co0 = co_tile * 32;
    const int b = blockIdx.z;
    const int dzw = wave / TH, hyw = wave % TH;               // this wave's input row
    const int lane_b = half * C::CS + (dzw * C::IH + hyw) * C::IW + l31;
    const int lane_a = half * 27 * 32 + l31;

    f32x16 acc[8];                        // index pd*4 + ph*2 + pw
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;

    const size_t in_plane = (size_t)H * W;
    const float* inb = in + (size_t)b * Cin * D * in_plane;

    constexpr int NIN = (C::IN_FLOATS + 255) / 256;
    constexpr int NWQ = (C::W_FLOATS / 4 + 255) / 256;
    static_assert(NIN <= 32 && NWQ <= 32, "validity masks are 32 bits");
    unsigned ioff[NIN], woff[NWQ], imask = 0u, wmask = 0u;
#pragma unroll
    for (int i = 0; i < NIN; ++i) {
        const int e = tid + 256 * i;
        const int wx = e % C::IW;
        int r = e / C::IW;
        const int hy = r % C::IH; r /= C::IH;
        const int dz = r % C::ID;
        const int ci = r / C::ID;
        const int gw = iw0 + wx, gh = ih0 + hy, gd = id0 + dz;
        const bool ok = (e < C::IN_FLOATS) && gd < D && gh < H && gw < W;
        ioff[i] = ok ? (unsigned)((((size_t)ci * D + gd) * in_plane + (size_t)gh * W + gw) * 4) : 0u;
        imask |= (unsigned)ok << i;
    }
#pragma unroll
    for (int j = 0; j < NWQ; ++j) {
        const int e = tid + 256 * j;
        const int q = e % 8;
        const int row = e / 8;                     // ci * 27 + tap
        const int co = co0 + q * 4;
        const bool ok = (e < C::W_FLOATS / 4) && (co + 3 < Cout);
        woff[j] = ok ? (unsigned)(((size_t)row * Cout + co) * 4) : 0u;
        wmask |= (unsigned)ok << j;
    }
    float rin[NIN];
    float4 rw[NWQ];
    unsigned ilive = imask, wlive = wmask;
    auto prefetch = [&](int ci0) {
        ilive = imask;
        wlive = wmask;
        if (ci0 + CIT > Cin) {
#pragma unroll
            for (int i = 0; i < NIN; ++i)
                if (ci0 + (tid + 256 * i) / C::CS >= Cin) ilive &= ~(1u << i);
#pragma unroll
            for (int j = 0; j < NWQ; ++j)
                if (ci0 + (tid + 256 * j) / (27 * 8) >= Cin) wlive &= ~(1u << j);
        }
        const char* ib = reinterpret_cast<const char*>(inb + (size_t)ci0 * D * in_plane);
        const char* wb = reinterpret_cast<const char*>(wpack + (size_t)ci0 * 27 * Cout);
#pragma unroll
        for (int i = 0; i < NIN; ++i) rin[i] = *reinterpret_cast<const float*>(ib + (((ilive >> i) & 1u) ? ioff[i] : 0u));
#pragma unroll
        for (int j = 0; j < NWQ; ++j) rw[j] = *reinterpret_cast<const float4*>(wb + (((wlive >> j) & 1u) ? woff[j] : 0u));
    };

    prefetch(0);
    for (int ci0 = 0; ci0 < Cin; ci0 += CIT) {
#pragma unroll
        for (int i = 0; i < NIN; ++i) {
            const int e = tid + 256 * i;
            if (e < C::IN_FLOATS) ilds[e] = ((ilive >> i) & 1u) ? rin[i] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < NWQ; ++j) {
            const int e = tid + 256 * j;
            if (e < C::W_FLOATS / 4)
                *reinterpret_cast<float4*>(&wlds[e * 4]) = ((wlive >> j) & 1u) ? rw[j] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();
        if (ci0 + CIT < Cin) prefetch(ci0 + CIT);
#pragma unroll 1
        for (int cp = 0; cp < CIT / 2; ++cp) {
            const float* ap = wlds + lane_a + cp * 2 * 27 * 32;
            const float* bp = ilds + lane_b + cp * 2 * C::CS;
            float x[8];                   // the 2x2x2 input neighbourhood of this lane's position
#pragma unroll
            for (int o = 0; o < 8; ++o) x[o] = bp[(((o >> 2) & 1) * C::IH + ((o >> 1) & 1)) * C::IW + (o & 1)];
#pragma unroll
            for (int kd = 0; kd < 3; ++kd)
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        if (PD >= 0 && (kd != 1) != PD) continue;          // compile-time after unrolling
                        const int cls = (kd != 1) * 4 + (kh != 1) * 2 + (kw != 1);
                        const int off = (kd == 0) * 4 + (kh == 0) * 2 + (kw == 0);
                        acc[cls] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[((kd * 3 + kh) * 3 + kw) * 32], x[off], acc[cls], 0, 0, 0);
                    }
        }
        __syncthreads();
    }

    const int Do = 2 * D, Ho = 2 * H, Wo = 2 * W;
    const size_t out_plane = (size_t)Ho * Wo;
    if (HAS_SKIP) {
        // ---- 1x1x1 projection of the skip tensor at the 8 output positions of every lane ----
        // No LDS: the 32x32x2 B operand is one value per lane (k = channel cs + half, n = this lane's
        // column), so each lane reads its own 2x2x2 cube of the skip tensor straight from global memory
        // as four float2 (a wave reads 256-byte row segments), two channel pairs per group, the next
        // group in flight while the current one is multiplied.
        // Buffer loads: SGPR descriptor + one 32-bit offset per lane, and out-of-range offsets (channel
        // pairs beyond Cs) read as zero without a branch.  Lanes outside the volume read garbage into
        // their own accumulator column only, which the epilogue never stores.
        const size_t schan = (size_t)Do * out_plane;
        const int jw_ = min(iw0 + l31, W - 1), jd_ = min(id0 + dzw, D - 1), jh_ = min(ih0 + hyw, H - 1);
        const __amdgpu_buffer_rsrc_t sres = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(skip + (size_t)b * Cs * schan), 0, (int)((size_t)Cs * schan * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(skip_w), 0, Cs * Cout * 4, 0x00020000);
        const unsigned lane_s = (unsigned)(((size_t)half * schan + (size_t)(2 * jd_) * out_plane + (size_t)(2 * jh_) * Wo + 2 * jw_) * 4);
        const unsigned lane_w = (unsigned)((half * Cout + min(co0 + l31, Cout - 1)) * 4);
        struct Group { float2 s[2][4]; float w[2]; };
        auto load_group = [&](Group& g, int cs0) {
#pragma unroll
            for (int cp = 0; cp < 2; ++cp) {
                const unsigned cb = (unsigned)((size_t)(cs0 + 2 * cp) * schan * 4);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned qo = (unsigned)(((size_t)(q >> 1) * out_plane + (size_t)(q & 1) * Wo) * 4);
                    g.s[cp][q] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(sres, (int)(lane_s + cb + qo), 0, 0));
                }
                g.w[cp] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                    wres, (int)(lane_w + (unsigned)((cs0 + 2 * cp) * Cout * 4)), 0, 0));
            }
        };
        auto mul_group = [&](const Group& g) {
#pragma unroll
            for (int cp = 0; cp < 2; ++cp)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (PD >= 0 && (q >> 1) != PD) continue;
                    acc[q * 2 + 0] = __builtin_amdgcn_mfma_f32_32x32x2f32(g.w[cp], g.s[cp][q].x, acc[q * 2 + 0], 0, 0, 0);
                    acc[q * 2 + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(g.w[cp], g.s[cp][q].y, acc[q * 2 + 1], 0, 0, 0);
                }
        };
        Group ga, gb;
        load_group(ga, 0);
#pragma unroll 1
        for (int cs0 = 0; cs0 < Cs; cs0 += 4) {
            load_group(gb, cs0 + 4);
            mul_group(ga);
            ga = gb;
        }
    }

    // ---- epilogue: each lane owns the 2x2x2 output cube of its input position ----
    const int jw = iw0 + l31, jd = id0 + dzw, jh = ih0 + hyw;
    if (jw >= W || jd >= D || jh >= H) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = co0 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (co >= Cout) continue;
        const float sc = scale ? scale[co] : 1.0f;
        const float sh = shift ? shift[co] : 0.0f;
#pragma unroll
        for (int pdh = 0; pdh < 4; ++pdh) {
            if (PD >= 0 && (pdh >> 1) != PD) continue;
            const int od = 2 * jd + (pdh >> 1), oh = 2 * jh + (pdh & 1);
            float v0 = ss::add_rn(ss::mul_rn(acc[pdh * 2 + 0][r], sc), sh);
            float v1 = ss::add_rn(ss::mul_rn(acc[pdh * 2 + 1][r], sc), sh);
            if (relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
            float* op = out + (((size_t)b * Cout + co) * Do + od) * out_plane + (size_t)oh * Wo + 2 * jw;
            *reinterpret_cast<float2*>(op) = make_float2(v0, v1);
        }
    }
}

template <int TD, int TH, int CIT, bool HAS_SKIP, bool SPLIT>
__global__ __launch_bounds__(256, 2) void deconv3d_mfma(const float* __restrict__ in, const float* __restrict__ wpack,
                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                         const float* __restrict__ skip, const float* __restrict__ skip_w,
                                                         float* __restrict__ out, int Cin, int D, int H, int W, int Cout,
                                                         int Cs, int tiles_w, int tiles_h, int relu) {
    if (SPLIT) {        // blockIdx.y = 2 * (tile of 32 output channels) + (parity of the output plane)
        if (blockIdx.y & 1)
            deconv3d_body<TD, TH, CIT, HAS_SKIP, 1>(blockIdx.y >> 1, in, wpack, scale, shift, skip, skip_w, out, Cin, D, H, W,
                                                    Cout, Cs, tiles_w, tiles_h, relu);
        else
            deconv3d_body<TD, TH, CIT, HAS_SKIP, 0>(blockIdx.y >> 1, in, wpack, scale, shift, skip, skip_w, out, Cin, D, H, W,
                                                    Cout, Cs, tiles_w, tiles_h, relu);
    } else {
        deconv3d_body<TD, TH, CIT, HAS_SKIP, -1>(blockIdx.y, in, wpack, scale, shift, skip, skip_w, out, Cin, D, H, W, Cout, Cs,
                                                 tiles_w, tiles_h, relu);
    }
}

template <int TD, int TH, int CIT, bool HAS_SKIP, bool SPLIT>
int launch_deconv_split(const float* in, const float* wpack, const float* scale, const float* shift, const float* skip,
                        const float* skip_w, float* out, int B, int Cin, int D, int H, int W, int Cout, int Cs, int relu,
                        long long nt, int tiles_w, int tiles_h, hipStream_t st) {
    using C = DCfg<TD, TH, CIT>;
    auto kern = deconv3d_mfma<TD, TH, CIT, HAS_SKIP, SPLIT>;
    const size_t lds = (size_t)C::LDS_FLOATS * 4;
    if (lds > 64 * 1024) {
        if (ss::ensure_dynamic_lds(reinterpret_cast<const void*>(kern), (int)lds) != SS_OK) return SS_ERR_LAUNCH;
    }
    dim3 grid((unsigned)nt, ss::ceil_div(Cout, 32) * (SPLIT ? 2 : 1), B);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, in, wpack, scale, shift, skip, skip_w, out, Cin, D, H, W, Cout, Cs,
                       tiles_w, tiles_h, relu);
    return ss::check_launch();
}

template <int TD, int TH, int CIT, bool HAS_SKIP>
int launch_deconv_as(const float* in, const float* wpack, const float* scale, const float* shift, const float* skip,
                     const float* skip_w, float* out, int B, int Cin, int D, int H, int W, int Cout, int Cs, int relu,
                     hipStream_t st) {
    const int tiles_w = ss::ceil_div(W, 32), tiles_h = ss::ceil_div(H, TH), tiles_d = ss::ceil_div(D, TD);
    const long long nt = (long long)tiles_w * tiles_h * tiles_d;
    if (nt > 0x7fffffffLL || B > 65535) return SS_ERR_UNSUPPORTED;
    // fewer workgroups than CUs: split every workgroup into its even- and odd-plane halves (measured on the
    // bench shapes: 128 workgroups 87 -> 67 us; at 384 workgroups the split form is already 7 % slower)
    bool split = nt * ss::ceil_div(Cout, 32) * B < 256;
    if (ss::tuning().deconv_split >= 0) split = ss::tuning().deconv_split == 1;      // tuning aid
    if (split)
        return launch_deconv_split<TD, TH, CIT, HAS_SKIP, true>(in, wpack, scale, shift, skip, skip_w, out, B, Cin, D, H, W, Cout,
                                                                Cs, relu, nt, tiles_w, tiles_h, st);
    return launch_deconv_split<TD, TH, CIT, HAS_SKIP, false>(in, wpack, scale, shift, skip, skip_w, out, B, Cin, D, H, W, Cout, Cs,
                                                             relu, nt, tiles_w, tiles_h, st);
}

template <int TD, int TH, int CIT>
int launch_deconv(const float* in, const float* wpack, const float* scale, const float* shift, const float* skip,
                  const float* skip_w, float* out, int B, int Cin, int D, int H, int W, int Cout, int Cs, int relu,
                  hipStream_t st) {
    if (skip != nullptr)
        return launch_deconv_as<TD, TH, CIT, true>(in, wpack, scale, shift, skip, skip_w, out, B, Cin, D, H, W, Cout, Cs, relu, st);
    return launch_deconv_as<TD, TH, CIT, false>(in, wpack, scale, shift, skip, skip_w, out, B, Cin, D, H, W, Cout, Cs, relu, st);
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
    if (Cout % 4 != 0) return SS_ERR_UNSUPPORTED;      // weight slabs are moved as float4
    if (skip != nullptr && (size_t)(Cs + 10) * 8 * D * H * W * 4 >= ((size_t)1 << 32))
        return SS_ERR_UNSUPPORTED;                      // the skip tensor of one pair is addressed with 32-bit offsets
    hipStream_t st = ss::as_stream(stream);
    // 2 planes x 2 rows when the volume has fewer than 4 rows (keeps the tile inside the volume)
    if (D >= 2 && H < 4)
        return launch_deconv<2, 2, 8>(in, wpack, scale, shift, skip, skip_wpack, out, B, Cin, D, H, W, Cout, Cs, relu, st);
    return launch_deconv<1, 4, 8>(in, wpack, scale, shift, skip, skip_wpack, out, B, Cin, D, H, W, Cout, Cs, relu, st);
}
