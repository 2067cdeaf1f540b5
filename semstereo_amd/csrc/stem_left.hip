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
#include <type_traits>

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

// phase stamps: no-ops here; tools/stem_left_timing.hip defines them for tools/wg_phases_stem_left.py
#ifndef SL_STAMP_DECL
#define SL_STAMP_VARS do {} while (0)
#define SL_STAMP_DECL() do {} while (0)
#define SL_STAMP(k) do {} while (0)
#define SL_STAMP_FINISH() do {} while (0)
#endif

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
    float (*atile)[HH][HW] = reinterpret_cast<float (*)[HH][HW]>(smem);              // [ND][6][34]
    float (*qtile)[QCOLS] = reinterpret_cast<float (*)[QCOLS]>(smem + ND * HH * HW);   // [64][224]
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

    const int cl = tid >> 7, pos = tid & 127, ty = pos >> 5, tx = pos & 31;     // multiply-add phase: (channel of the pair, position)
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
        float o[ND];
#pragma unroll
        for (int j = 0; j < ND; ++j) o[j] = 0.f;
#pragma unroll
        for (int s = 0; s < 9; ++s) {
            const int kh = s / 3, kw = s % 3;
            const int hp = (ty + kh) * HW + tx + kw;
            float av[ND + 2];
            av[0] = 0.f;
            av[ND + 1] = 0.f;
#pragma unroll
            for (int j = 0; j < ND; ++j) av[1 + j] = atile[j][ty + kh][tx + kw];
#pragma unroll
            for (int kd = 0; kd < 3; ++kd) {
                const float v = qtile[((kd * 3 + kh) * 3 + kw) * 2 + cl][hp];
#pragma unroll
                for (int j = 0; j < ND; ++j) o[j] = fmaf(av[j + kd], v, o[j]);
            }
        }
        if (inside) {
            float* ob = out + (((size_t)b * Cout + 2 * g + cl) * ND) * plane + (size_t)h * W + w;
#pragma unroll
            for (int j = 0; j < ND; ++j) ob[(size_t)j * plane] = o[j];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same computation with its two phases overlapped and the att tile in REGISTERS (r03).  In stem_left_fused a channel pair
// is first projected on the matrix core (Q rows -> LDS), then consumed by the 27 multiply-adds per output, and both
// workgroups of a CU run the phases in lock-step (tools/wg_phases_stem_left.py: 62 % of the loop in the multiply-adds, 29 % in
// the projection).  Overlapping the two alone (projection of channel c + 1 issued between the multiply-adds of channel c)
// measured 88.3 / 90.4 us against 88.7 / 89.2: the multiply-add phase re-reads the att values of its nine shifts from LDS for
// every channel -- 400 KB per channel and CU, as many LDS clocks as the multiply-adds take VALU clocks.  Here a workgroup is
// EIGHT waves, one per CU: a thread owns (position, quarter of the candidates) and keeps the 9 x (ND / 4 + 2) att values it ever
// needs in registers (read once); per channel it reads only its 27 Q values.  The workgroup walks the output channels one at
// a time with two Q buffers: wave w projects N-tile w of channel c + 1 (12 matrix instructions, issued between the
// multiply-adds of channel c) into the other buffer; one barrier per channel.  M tile = the 27 taps of one channel (rows
// 27-31 zero): weights packed [channel][32 rows][C] by the pointwise packer.  Per output the same fused multiply-adds in
// the same order as stem_left_fused: bit-identical results.
constexpr int att_stride(int nd) { return (nd + 3) / 4 * 4 + 4; }     // dwords per position of the staged att tile: [0, att_0 .. att_{ND-1}, 0, pad]

template <int ND, int NTERMS>
__global__ __launch_bounds__(512) void stem_left_overlap(const float* __restrict__ left, const uint4* __restrict__ wsplit,
                                                          const float* __restrict__ att, float* __restrict__ out,
                                                          int Cout, int H, int W) {
    constexpr int NC = (NTERMS == 6) ? 3 : 2, KS = 2, C = 32;
    constexpr int NP = (NTERMS == 6) ? 6 : 3;                 // products per K-step
    SL_STAMP_VARS;
    constexpr int NPART = (ND % 4 == 0) ? 4 : ((ND % 2 == 0) ? 2 : 1);      // candidate groups per position (4 x 128 threads)
    constexpr int AS = att_stride(ND), NJ = ND / NPART, NWIN = NJ + 2;
    static_assert(NNT <= 8, "one N-tile per wave");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // (the Q buffers first: every Q access of the channel loop is then one base register + a 16-bit immediate; behind the
    // att tile they end beyond 64 KB and the compiler kept ~30 row addresses in registers -- and spilled them)
    float (*qbuf)[32][QCOLS] = reinterpret_cast<float (*)[32][QCOLS]>(smem);           // [2][32 rows][224]
    float* atile = smem + 2 * 32 * QCOLS;                                             // [204 positions][AS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int w0 = blockIdx.x * FTW, h0 = blockIdx.y * FTH, b = blockIdx.z;
    const size_t plane = (size_t)H * W;

    {   // the att halo tile (all loads issued together, unconditionally: see stem_left_fused), then the zero ends of every row
        const __amdgpu_buffer_rsrc_t ares = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(att + (size_t)b * ND * plane), 0, (int)min((long long)ND * (long long)plane * 4, 0x7fffffffLL), 0x00020000);
        constexpr int NE = (ND * NPOSH + 511) / 512;
        float av[NE];
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            const int e = tid + 512 * k;
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
            const int e = tid + 512 * k;
            if (e < ND * NPOSH) atile[(e % NPOSH) * AS + 1 + e / NPOSH] = av[k];
        }
        if (tid < NPOSH) {
            atile[tid * AS] = 0.f;
#pragma unroll
            for (int k = ND + 1; k < AS; ++k) atile[tid * AS + k] = 0.f;
        }
    }

    // this wave's N-tile of the halo tile (the 8th does not exist: zeros, computed and dropped): the left map there, split once
    const __amdgpu_buffer_rsrc_t lres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(left + (size_t)b * C * plane), 0, (int)min((long long)C * (long long)plane * 4, 0x7fffffffLL), 0x00020000);
    const int chan_b = (int)(plane * 4);
    bf16x8 bfrag[KS][NC];
    {
        const int p = wave * 32 + l31;
        const int py = p / HW, px = p % HW;
        const int gh = h0 + py - 1, gw = w0 + px - 1;
        const bool ok = wave < NNT && p < NPOSH && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W;
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
            bfrag[ks][0] = __builtin_bit_cast(bf16x8, make_uint4(bh[0], bh[1], bh[2], bh[3]));
            bfrag[ks][1] = __builtin_bit_cast(bf16x8, make_uint4(bm[0], bm[1], bm[2], bm[3]));
            if (NC == 3) bfrag[ks][NC - 1] = __builtin_bit_cast(bf16x8, make_uint4(bl[0], bl[1], bl[2], bl[3]));
        }
    }

    // Weight rows are fetched one channel ahead, at the top of a channel, i.e. BEFORE that channel's stores: the memory counter
    // retires in order and counts stores too, so a wait for rows requested after stores is a wait for those stores to reach
    // memory.  Two register sets in fixed roles per unrolled pass (rotated by copies, every pass waits for its own request).
    bf16x8 wa[KS][NC], wb[KS][NC];
    // (through a buffer descriptor: one 32-bit lane offset and a wave-uniform scalar offset per fragment -- as 64-bit per-lane
    // pointers the six addresses were hoisted out of the channel loop and spilled around it)
    const __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint4*>(wsplit), 0, (int)min((long long)Cout * KS * 3 * 2 * 32 * 16, 0x7fffffffLL), 0x00020000);
    const int wlane = (half * 32 + l31) * 16;
    auto load_a = [&](bf16x8 (&dst)[KS][NC], int c) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int q = 0; q < NC; ++q)
                dst[ks][q] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wres, wlane, ((c * KS + ks) * 3 + q) * 2 * 32 * 16, 0));
    };
    // the k-th matrix-core instruction of a channel's projection: product k % NP of K-step k / NP, smallest cross terms first
    constexpr int NMF = KS * NP;
    f32x16 acc;
    auto mfma_k = [&](const bf16x8 (&a)[KS][NC], int k) {
        constexpr int pa[6] = {1, 0, NC - 1, 0, 1, 0}, pb[6] = {1, NC - 1, 0, 1, 0, 0};
        const int ks = k / NP, p = 6 - NP + k % NP;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][pa[p]], bfrag[ks][pb[p]], acc, 0, 0, 0);
    };
    auto clear_acc = [&]() {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    };
    auto park_q = [&](int buf) {                                  // D layout: column = position, row (r & 3) + 8 (r >> 2) + 4 half = tap
        if (wave < NNT) {                                         // wave-uniform
#pragma unroll
            for (int r = 0; r < 16; ++r) qbuf[buf][(r & 3) + 8 * (r >> 2) + 4 * half][wave * 32 + l31] = acc[r];
        }
    };

    // channel 0's projection, plainly
    load_a(wb, 0);
    load_a(wa, min(1, Cout - 1));
    clear_acc();
#pragma unroll
    for (int k = 0; k < NMF; ++k) mfma_k(wb, k);
    park_q(0);
    __syncthreads();                                              // atile and Q(0) are in place

    const int jq = tid >> 7, pos = tid & 127, ty = pos >> 5, tx = pos & 31;      // multiply-adds: (group of candidates, position); jq is wave-uniform
    const int h = h0 + ty, w = w0 + tx;
    const bool active = jq < NPART, inside = active && h < H && w < W;
    // results leave through a buffer descriptor: a thread outside the image (or without a candidate group) stores beyond the
    // buffer, i.e. nothing -- no branch around the multiply-adds, which share their block with the matrix instructions
    const __amdgpu_buffer_rsrc_t ores = __builtin_amdgcn_make_buffer_rsrc(
        out + (size_t)b * Cout * ND * plane, 0, (int)min((long long)Cout * ND * (long long)plane * 4, 0x7fffffffLL), 0x00020000);
    const unsigned ooff = inside ? (unsigned)((((size_t)jq * NJ) * plane + (size_t)h * W + w) * 4) : 0x80000000u;
    const int plane_b = (int)(plane * 4);
    // ar[s][k] = att[jq * NJ - 1 + k] at the position shifted by (kh, kw) = (s / 3, s % 3): zero beyond both ends of the candidates
    float ar[9][NWIN];
#pragma unroll
    for (int s = 0; s < 9; ++s) {
        const int hp = (ty + s / 3) * HW + tx + s % 3;
        const float* row = atile + hp * AS + (active ? jq : 0) * NJ;
        if constexpr (NJ % 2 == 0) {                             // 8-byte aligned windows
#pragma unroll
            for (int k = 0; k < NWIN; k += 2) {
                const float2 t2 = *reinterpret_cast<const float2*>(row + k);
                ar[s][k] = t2.x;
                ar[s][k + 1] = t2.y;
            }
        } else {
#pragma unroll
            for (int k = 0; k < NWIN; ++k) ar[s][k] = row[k];
        }
    }
    const int hp0 = ty * HW + tx;
    // one channel: `a` holds the rows of channel c + 1 (projected now), `freed` -- the set the previous channel projected from --
    // takes the request for channel c + 2
    // A CU's store path drains ~19 bytes per clock whatever the width (tools/exp_store_rate.hip: 13.4 clocks per 256-byte store
    // instruction into L2, 27.7 into HBM): a channel's 48 store instructions per CU are 650-1300 clocks, as long as its
    // multiply-adds.  Issued together at the end of the channel they were paid in full by every wave, the phases being in
    // lock-step across the workgroup (tools/wg_phases_stem_left.py: 1280 clocks of multiply-adds, 554 issuing stores, 740 at the
    // barrier); so a channel's results wait in registers and leave one store per shift of the NEXT channel's multiply-adds.
    float op[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) op[j] = 0.f;
    // ... and the two waves of a SIMD (w, w + 4) take turns: waves 0-3 store during the first four shifts, waves 4-7 during
    // the last four.  The store queue of a CU is short: a wave that stores into a full queue stands still, and when both waves
    // of a SIMD store at the same points of the program -- as they did, in lock-step -- the SIMD issues nothing meanwhile.
    auto channel = [&](auto late_tag, int c, const bf16x8 (&a)[KS][NC], bf16x8 (&freed)[KS][NC]) {
        constexpr int S0 = decltype(late_tag)::value ? 5 : 0;     // first shift of this wave's store window (4 shifts)
        const int cur = c & 1;
        const unsigned ooff_prev = c > 0 ? ooff : 0x80000000u;    // (nothing to store beside channel 0)
        SL_STAMP(0);
        load_a(freed, min(c + 2, Cout - 1));
        clear_acc();
        float v[27];
#pragma unroll
        for (int t = 0; t < 27; ++t) v[t] = qbuf[cur][t][hp0 + ((t / 3) % 3) * HW + t % 3];      // tap t = (kd * 3 + kh) * 3 + kw
        float o[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) o[j] = 0.f;
#pragma unroll
        for (int s = 0; s < 9; ++s) {
            // this shift's share of the next channel's projection, then its multiply-adds (same order per output as stem_left_fused)
            constexpr int PER = (NMF + 8) / 9;
#pragma unroll
            for (int k = s * PER; k < (s + 1) * PER && k < NMF; ++k) mfma_k(a, k);
#pragma unroll
            for (int kd = 0; kd < 3; ++kd)
#pragma unroll
                for (int j = 0; j < NJ; ++j) o[j] = fmaf(ar[s][j + kd], v[kd * 9 + s], o[j]);
            // the previous channel's results: stores (s - S0) * NJ / 4 .. (s - S0 + 1) * NJ / 4 - 1 of its NJ
            if (s >= S0 && s < S0 + 4) {
#pragma unroll
                for (int j = (s - S0) * NJ / 4; j < (s - S0 + 1) * NJ / 4; ++j)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, op[j]), ores, (int)ooff_prev, ((c - 1) * ND + j) * plane_b, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) op[j] = o[j];
        SL_STAMP(1);
        if (c + 1 < Cout) park_q(cur ^ 1);                        // (the last pass projected a clamped channel: dropped)
        SL_STAMP(2);
        __syncthreads();
        SL_STAMP(3);
    };
    SL_STAMP_DECL();
    if (wave < 4) {                                               // (wave-uniform: both copies run the same barriers)
        for (int c = 0; c < Cout; c += 2) {
            channel(std::false_type{}, c, wa, wb);
            if (c + 1 >= Cout) break;
            channel(std::false_type{}, c + 1, wb, wa);
        }
    } else {
        for (int c = 0; c < Cout; c += 2) {
            channel(std::true_type{}, c, wa, wb);
            if (c + 1 >= Cout) break;
            channel(std::true_type{}, c + 1, wb, wa);
        }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j)                                  // the last channel's results
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, op[j]), ores, (int)ooff, ((Cout - 1) * ND + j) * plane_b, 0);
    SL_STAMP_FINISH();
}

template <int ND>
int launch_overlap(const float* left, const void* wsplit, const float* att, float* out, int B, int Cout, int H, int W,
                   int nterms, hipStream_t st) {
    const size_t lds = ((size_t)att_stride(ND) * NPOSH + 2 * 32 * QCOLS) * sizeof(float);
    dim3 grid(ss::ceil_div(W, FTW), ss::ceil_div(H, FTH), B);
    if (nterms == 6) {
        auto kern = stem_left_overlap<ND, 6>;
        if (lds > 64 * 1024) {
            if (ss::ensure_dynamic_lds(reinterpret_cast<const void*>(kern), (int)lds) != SS_OK) return SS_ERR_LAUNCH;
        }
        hipLaunchKernelGGL(kern, grid, dim3(512), lds, st, left, reinterpret_cast<const uint4*>(wsplit), att, out, Cout, H, W);
    } else {
        auto kern = stem_left_overlap<ND, 3>;
        if (lds > 64 * 1024) {
            if (ss::ensure_dynamic_lds(reinterpret_cast<const void*>(kern), (int)lds) != SS_OK) return SS_ERR_LAUNCH;
        }
        hipLaunchKernelGGL(kern, grid, dim3(512), lds, st, left, reinterpret_cast<const uint4*>(wsplit), att, out, Cout, H, W);
    }
    return ss::check_launch();
}

template <int ND>
int launch_fused(const float* left, const void* wsplit, const float* att, float* out, int B, int Cout, int H, int W,
                 int nterms, hipStream_t st) {
    const size_t lds = ((size_t)ND * HH * HW + 64 * QCOLS) * sizeof(float);
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

// the overlapped form: `wsplit` = ss_pack_pointwise_weights_bf16s of [Cout * 32, C] with row co * 32 + tap (rows 27-31 of a channel zero)
extern "C" int ss_stem_left_overlap_fwd(const float* left, const void* wsplit, const float* att, float* out, int B, int C,
                                        int Cout, int nd, int H, int W, int nterms, ss_stream_t stream) {
    SS_REQUIRE(left && wsplit && att && out);
    SS_REQUIRE(B > 0 && C > 0 && Cout > 0 && nd > 0 && H > 0 && W > 0 && (nterms == 3 || nterms == 6));
    SS_REQUIRE((reinterpret_cast<uintptr_t>(wsplit) & 15) == 0);
    if (C != 32 || B > 65535) return SS_ERR_UNSUPPORTED;
    if ((long long)C * H * W * 4 >= 0x7fffffffLL || (long long)nd * H * W * 4 >= 0x7fffffffLL) return SS_ERR_UNSUPPORTED;   // 32-bit buffer offsets
    if ((long long)Cout * nd * H * W * 4 >= 0x7fffffffLL) return SS_ERR_UNSUPPORTED;      // ... also into one batch element's output
    hipStream_t st = ss::as_stream(stream);
    if (nd == 24) return launch_overlap<24>(left, wsplit, att, out, B, Cout, H, W, nterms, st);
    if (nd == 32) return launch_overlap<32>(left, wsplit, att, out, B, Cout, H, W, nterms, st);
    if (nd == 6) return launch_overlap<6>(left, wsplit, att, out, B, Cout, H, W, nterms, st);
    return SS_ERR_UNSUPPORTED;
}
