// The left half of concat_stem without its 87 GFLOP (gfx950).
//
// The sparse concat volume of SemStereo.forward (reference models/SemStereo.py:241-244, 316-318) is
//   vol[c, j, h, w] = att[j, h, w] * left[c, h, w]            c <  C   (the left features, BROADCAST over the candidates j)
//   vol[C + c, j, h, w] = att[j, h, w] * warp(right)[c, j, h, w]
// and concat_stem (:319) is a 3x3x3 convolution over its 2C channels.  The convolution is linear, so the left half is
//   sum_{c<C, tap} W[co, c, tap] * att[pos + tap] * left[c, (pos + tap).hw]
//     = sum_tap att[pos + tap] * Q[tap, co, (pos + tap).hw],      Q[tap, co, y, x] = sum_{c<C} W[co, c, tap] * left[c, y, x]
// Q is a 1x1 convolution of the 2-D left feature map (C -> 27 * Cout channels, 3.6 GFLOP instead of 87: one launch of
// ss_conv3d_pointwise_bf16s_fwd, with the BatchNorm scale folded into W), and what remains -- this kernel -- is 27
// multiply-adds per output element.  The result enters the right half's convolution as its `residual` operand.
//
// One workgroup: 8 rows x 32 columns x all nd candidates x 4 output channels; the att tile (+1 halo in h, w) is parked in
// LDS, every thread keeps its 4 x nd outputs in registers and streams the 27 Q planes of its 4 channels past them.
#include <algorithm>

#include "common.h"

namespace {

constexpr int TH = 8, TW = 32, COG = 4;

template <int ND>
__global__ __launch_bounds__(256) void stem_left_kernel(const float* __restrict__ q, const float* __restrict__ att,
                                                         float* __restrict__ out, int Cout, int H, int W) {
    __shared__ float atile[ND][TH + 2][TW + 2];
    const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;
    const int w0 = blockIdx.x * TW, h0 = blockIdx.y * TH;
    const int ngroups = Cout / COG;
    const int b = blockIdx.z / ngroups, co0 = (blockIdx.z % ngroups) * COG;
    const size_t plane = (size_t)H * W;
    const float* ab = att + (size_t)b * ND * plane;
    for (int e = tid; e < ND * (TH + 2) * (TW + 2); e += 256) {
        const int x = e % (TW + 2);
        int r = e / (TW + 2);
        const int y = r % (TH + 2), j = r / (TH + 2);
        const int gh = h0 + y - 1, gw = w0 + x - 1;
        atile[j][y][x] = ((unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W) ? ab[(size_t)j * plane + (size_t)gh * W + gw] : 0.f;
    }
    __syncthreads();
    const int h = h0 + ty, w = w0 + tx;
    if (h >= H || w >= W) return;
    float acc[COG][ND];
#pragma unroll
    for (int c = 0; c < COG; ++c)
#pragma unroll
        for (int j = 0; j < ND; ++j) acc[c][j] = 0.f;
    const float* qb = q + ((size_t)b * 27 * Cout + co0) * plane;
    // The 12 Q values of a (kh, kw) shift are fetched one shift ahead.  Shifts that leave the image read a clamped
    // (valid) address: their att column in LDS is all zeros, so the value does not matter -- no branch in the loop.
    float qv[2][3][COG];
    auto load_q = [&](float (&dst)[3][COG], int s) {
        const int kh = s / 3, kw = s % 3;
        const int y = min(max(h + kh - 1, 0), H - 1), x = min(max(w + kw - 1, 0), W - 1);
        const float* p = qb + (size_t)y * W + x;
#pragma unroll
        for (int kd = 0; kd < 3; ++kd)
#pragma unroll
            for (int c = 0; c < COG; ++c) dst[kd][c] = p[((size_t)((kd * 3 + kh) * 3 + kw) * Cout + c) * plane];
    };
    load_q(qv[0], 0);
#pragma unroll
    for (int s = 0; s < 9; ++s) {
        if (s + 1 < 9) load_q(qv[(s + 1) & 1], s + 1);
        const int kh = s / 3, kw = s % 3;
        float a[ND + 2];                 // a[1 + j] = att[j, y, x]; a[0] = a[ND + 1] = 0 (zero padding along the candidates)
        a[0] = 0.f;
        a[ND + 1] = 0.f;
#pragma unroll
        for (int j = 0; j < ND; ++j) a[1 + j] = atile[j][ty + kh][tx + kw];
#pragma unroll
        for (int kd = 0; kd < 3; ++kd)
#pragma unroll
            for (int c = 0; c < COG; ++c) {
                const float v = qv[s & 1][kd][c];
#pragma unroll
                for (int j = 0; j < ND; ++j) acc[c][j] = fmaf(a[j + kd], v, acc[c][j]);      // att[j + kd - 1]
            }
    }
    float* ob = out + ((size_t)b * Cout + co0) * ND * plane + (size_t)h * W + w;
#pragma unroll
    for (int c = 0; c < COG; ++c)
#pragma unroll
        for (int j = 0; j < ND; ++j) ob[((size_t)c * ND + j) * plane] = acc[c][j];
}

}  // namespace

extern "C" int ss_stem_left_fwd(const float* q, const float* att, float* out, int B, int Cout, int nd, int H, int W,
                                ss_stream_t stream) {
    SS_REQUIRE(q && att && out);
    SS_REQUIRE(B > 0 && Cout > 0 && nd > 0 && H > 0 && W > 0);
    if (Cout % COG != 0 || (long long)B * (Cout / COG) > 65535) return SS_ERR_UNSUPPORTED;
    dim3 grid(ss::ceil_div(W, TW), ss::ceil_div(H, TH), B * (Cout / COG));
    hipStream_t st = ss::as_stream(stream);
    // nd is the number of kept candidates: 24 in the reference (models/SemStereo.py:301; "32" in its comment)
    if (nd == 24)
        hipLaunchKernelGGL(stem_left_kernel<24>, grid, dim3(256), 0, st, q, att, out, Cout, H, W);
    else if (nd == 32)
        hipLaunchKernelGGL(stem_left_kernel<32>, grid, dim3(256), 0, st, q, att, out, Cout, H, W);
    else if (nd == 6)
        hipLaunchKernelGGL(stem_left_kernel<6>, grid, dim3(256), 0, st, q, att, out, Cout, H, W);
    else
        return SS_ERR_UNSUPPORTED;
    return ss::check_launch();
}

// ---------------------------------------------------------------------------------------------------------------------
// The same result with Q never leaving the CU: one workgroup owns 4 x 32 output positions (all candidates) and walks
// over the output channels two at a time.  Per channel pair the 54 rows (27 taps x 2 channels) of Q on the 6 x 34 halo
// tile are one 64 x 224 matrix product on the bf16 matrix core (split-bf16, K = C = 32 left channels: the left map's
// tile is read straight from global memory and split ONCE per workgroup, the weights are the pointwise-packed rows
// [pair][64][C]), parked in LDS, and consumed by the 27 multiply-adds per output as above.  Saves the 226 MB write and
// ~300 MB read of Q.
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) short;
using bf16x2_t = __attribute__((ext_vector_type(2))) __bf16;
using f32x2_t = __attribute__((ext_vector_type(2))) float;

__device__ __forceinline__ unsigned cvt_pk_bf16(float x0, float x1) {
    const f32x2_t v = {x0, x1};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ void split3_pk(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    h = cvt_pk_bf16(x0, x1);
    const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
    m = cvt_pk_bf16(r0, r1);
    const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
    l = cvt_pk_bf16(s0, s1);
}

constexpr int FTH = 4, FTW = 32;                    // output tile
constexpr int HH = FTH + 2, HW = FTW + 2;           // halo tile 6 x 34
constexpr int NPOSH = HH * HW;                      // 204 positions
constexpr int NNT = (NPOSH + 31) / 32;              // 7 N-tiles of 32 positions
constexpr int QCOLS = NNT * 32;                     // 224

template <int ND, int NTERMS>      // C = 32 left channels (2 K-steps of 16)
__global__ __launch_bounds__(256, 2) void stem_left_fused(const float* __restrict__ left, const uint4* __restrict__ wsplit,
                                                           const float* __restrict__ att, float* __restrict__ out,
                                                           int Cout, int H, int W) {
    constexpr int NC = (NTERMS == 6) ? 3 : 2, KS = 2, C = 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float (*atile)[HH][HW] = reinterpret_cast<float (*)[HH][HW]>(smem) + 1;          // [-1 .. ND][6][34]: planes -1 and ND are zero (the padding along the candidates)
    float (*qtile)[QCOLS] = reinterpret_cast<float (*)[QCOLS]>(smem + (ND + 2) * HH * HW);   // [64][224]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int w0 = blockIdx.x * FTW, h0 = blockIdx.y * FTH, b = blockIdx.z;
    const size_t plane = (size_t)H * W;

    // the att halo tile: every thread's loads are issued together, unconditionally, through a buffer descriptor (positions
    // outside the map: an offset beyond the buffer, which reads 0).  As a rolled loop of conditional loads this was 19
    // exposed round trips per workgroup.
    {
        const __amdgpu_buffer_rsrc_t ares = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(att + (size_t)b * ND * plane), 0, (int)min((long long)ND * (long long)plane * 4, 0x7fffffffLL), 0x00020000);
        constexpr int NE = (ND * NPOSH + 255) / 256;
        float av[NE];
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            const int e = tid + 256 * k;
            const int x = e % HW;
            int r = e / HW;
            const int y = r % HH, j = r / HH;
            const int gh = h0 + y - 1, gw = w0 + x - 1;
            const bool ok = e < ND * NPOSH && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W;
            av[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                  ares, (int)(ok ? (unsigned)(((size_t)j * plane + (size_t)gh * W + gw) * 4) : 0x80000000u), 0, 0));
        }
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            const int e = tid + 256 * k;
            if (e < ND * NPOSH) (&atile[0][0][0])[e] = av[k];
        }
        if (tid < NPOSH) {
            (&atile[-1][0][0])[tid] = 0.f;
            (&atile[ND][0][0])[tid] = 0.f;
        }
    }

    // this wave's N-tiles of the halo tile (wave, wave + 4): the left map there, split once
    const __amdgpu_buffer_rsrc_t lres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(left + (size_t)b * C * plane), 0, (int)min((long long)C * (long long)plane * 4, 0x7fffffffLL), 0x00020000);
    const int chan_b = (int)(plane * 4);
    bf16x8 bfrag[2][KS][NC];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int nt = wave + 4 * t;
        const int p = nt * 32 + l31;
        const int py = p / HW, px = p % HW;
        const int gh = h0 + py - 1, gw = w0 + px - 1;
        const bool ok = nt < NNT && p < NPOSH && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W;
        const unsigned off = ok ? (unsigned)((8LL * half * plane + (size_t)gh * W + gw) * 4) : 0x80000000u;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                x[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(lres, (int)off, (ks * 16 + j) * chan_b, 0));
            unsigned bh[4], bm[4], bl[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) split3_pk(x[2 * j], x[2 * j + 1], bh[j], bm[j], bl[j]);
            bfrag[t][ks][0] = __builtin_bit_cast(bf16x8, make_uint4(bh[0], bh[1], bh[2], bh[3]));
            bfrag[t][ks][1] = __builtin_bit_cast(bf16x8, make_uint4(bm[0], bm[1], bm[2], bm[3]));
            if (NC == 3) bfrag[t][ks][NC - 1] = __builtin_bit_cast(bf16x8, make_uint4(bl[0], bl[1], bl[2], bl[3]));
        }
    }

    // multiply-add phase: a thread owns one position, BOTH channels of the pair and half of the candidates (r05: it owned one channel and
    // all candidates -- the same 24 accumulators, but the two channels of a position share the attention operand, so their 27 x 12
    // multiply-adds are v_pk_fma_f32 pairs: half the VALU instructions, 180 instead of 243 LDS reads per thread, same sums in the same order)
    const int jh = tid >> 7, pos = tid & 127, ty = pos >> 5, tx = pos & 31;
    constexpr int NJ = ND / 2;                       // candidates per thread
    static_assert(ND % 2 == 0, "the candidates split over two thread halves");
    const int h = h0 + ty, w = w0 + tx;
    const bool inside = h < H && w < W;
    const int npairs = Cout / 2;
    for (int g = 0; g < npairs; ++g) {
        // ---- Q rows of channel pair g on the halo tile: [64 rows][224 positions] ----
        bf16x8 a[2][KS][NC];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int c = 0; c < NC; ++c)
                    a[m][ks][c] = __builtin_bit_cast(bf16x8, wsplit[((((size_t)(g * 2 + m) * KS + ks) * 3 + c) * 2 + half) * 32 + l31]);
        __syncthreads();                 // the previous pair's multiply-adds are done with qtile (and atile is staged)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int nt = wave + 4 * t;
            if (nt >= NNT) continue;     // wave-uniform
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    if (NTERMS == 6) {
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][ks][1], bfrag[t][ks][1], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][ks][0], bfrag[t][ks][NC - 1], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][ks][NC - 1], bfrag[t][ks][0], acc, 0, 0, 0);
                    }
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][ks][0], bfrag[t][ks][1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][ks][1], bfrag[t][ks][0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][ks][0], bfrag[t][ks][0], acc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    qtile[m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half][nt * 32 + l31] = acc[r];
            }
        }
        __syncthreads();
        // ---- 27 multiply-adds per output: Q row of (tap, channel cl of the pair) = tap * 2 + cl ----
        f32x2_t o[NJ];                   // (channel 2 g, channel 2 g + 1) of candidate jh * NJ + j
#pragma unroll
        for (int j = 0; j < NJ; ++j) o[j] = f32x2_t{0.f, 0.f};
        const int j0 = jh * NJ;
#pragma unroll
        for (int s = 0; s < 9; ++s) {
            const int kh = s / 3, kw = s % 3;
            const int hp = (ty + kh) * HW + tx + kw;
            float av[NJ + 2];            // av[1 + j] = att[j0 + j]; av[0], av[NJ + 1]: the neighbours (zero padding along the candidates)
#pragma unroll
            for (int j = -1; j <= NJ; ++j) av[1 + j] = atile[j0 + j][ty + kh][tx + kw];
#pragma unroll
            for (int kd = 0; kd < 3; ++kd) {
                const int qrow = ((kd * 3 + kh) * 3 + kw) * 2;
                const f32x2_t v = {qtile[qrow][hp], qtile[qrow + 1][hp]};
#pragma unroll
                for (int j = 0; j < NJ; ++j) o[j] = __builtin_elementwise_fma(f32x2_t{av[j + kd], av[j + kd]}, v, o[j]);
            }
        }
        if (inside) {
            float* ob = out + (((size_t)b * Cout + 2 * g) * ND + j0) * plane + (size_t)h * W + w;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                ob[(size_t)j * plane] = o[j][0];
                ob[((size_t)ND + j) * plane] = o[j][1];
            }
        }
    }
}

template <int ND>
int launch_fused(const float* left, const void* wsplit, const float* att, float* out, int B, int Cout, int H, int W,
                 int nterms, hipStream_t st) {
    const size_t lds = ((size_t)(ND + 2) * HH * HW + 64 * QCOLS) * sizeof(float);
    dim3 grid(ss::ceil_div(W, FTW), ss::ceil_div(H, FTH), B);
    if (nterms == 6) {
        auto kern = stem_left_fused<ND, 6>;
        if (lds > 64 * 1024) {
            if (ss::ensure_dynamic_lds(reinterpret_cast<const void*>(kern), (int)lds) != SS_OK) return SS_ERR_LAUNCH;
        }
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, left, reinterpret_cast<const uint4*>(wsplit), att, out, Cout, H, W);
    } else {
        auto kern = stem_left_fused<ND, 3>;
        if (lds > 64 * 1024) {
            if (ss::ensure_dynamic_lds(reinterpret_cast<const void*>(kern), (int)lds) != SS_OK) return SS_ERR_LAUNCH;
        }
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, left, reinterpret_cast<const uint4*>(wsplit), att, out, Cout, H, W);
    }
    return ss::check_launch();
}

}  // namespace

extern "C" int ss_stem_left_fused_fwd(const float* left, const void* wsplit, const float* att, float* out, int B, int C,
                                      int Cout, int nd, int H, int W, int nterms, ss_stream_t stream) {
    SS_REQUIRE(left && wsplit && att && out);
    SS_REQUIRE(B > 0 && C > 0 && Cout > 0 && nd > 0 && H > 0 && W > 0 && (nterms == 3 || nterms == 6));
    SS_REQUIRE((reinterpret_cast<uintptr_t>(wsplit) & 15) == 0);
    if (C != 32 || Cout % 2 != 0 || B > 65535) return SS_ERR_UNSUPPORTED;
    if ((long long)C * H * W * 4 >= 0x7fffffffLL || (long long)nd * H * W * 4 >= 0x7fffffffLL) return SS_ERR_UNSUPPORTED;   // 32-bit buffer offsets
    hipStream_t st = ss::as_stream(stream);
    if (nd == 24) return launch_fused<24>(left, wsplit, att, out, B, Cout, H, W, nterms, st);
    if (nd == 32) return launch_fused<32>(left, wsplit, att, out, B, Cout, H, W, nterms, st);
    if (nd == 6) return launch_fused<6>(left, wsplit, att, out, B, Cout, H, W, nterms, st);
    return SS_ERR_UNSUPPORTED;
}
