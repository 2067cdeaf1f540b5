// 3x3x3 Conv3d with ONE output channel (the classifier heads classif.2 / classif_att_.2, reference
// models/SemStereo.py:228-234: nn.Conv3d(32, 1, 3, padding=1, bias=False)) on the bf16 matrix core with
// split-bf16 fp32 emulation (see conv3d_bf16s.hip for the arithmetic).
//
// A single output channel wastes 31/32 of an M = 32 tile, so the roles are turned around: the 27 TAPS are
// the M rows.  For one input row (d', h') of 32 consecutive columns
//
//     P[tap][x] = sum_c  w[c, tap] * in[c, d', h', x]            (M = 27 taps, N = 32 columns, K = Cin)
//
// is two K-steps of v_mfma_f32_32x32x16_bf16 per 16 channels, with the weights resident in registers
// and the B operand read straight from global memory (lane n holds 8 channels of column n: no LDS
// staging at all).  The output is the shifted sum  out[d, h, x] = sum_{kd,kh,kw} P[(kd,kh,kw)][d+kd-1,
// h+kh-1, x+kw-1]:  the kw part is a +-1 lane shift of accumulator registers (DPP wave_shr/wave_shl;
// taps are assigned to accumulator rows so that the three kw taps of a (kd,kh) group sit in the same
// lane half), which leaves 9 partial rows S[(kd,kh)][d', h'][x] per input row.  Those go to LDS, and a
// second phase adds the 9 of every output position in a fixed order (deterministic).  A workgroup owns
// TD x TH x 30 outputs; the 32-lane tile carries one halo column on each side.
#include <algorithm>

#include "common.h"
#include "split_f16.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) short;
using bf16x2_t = __attribute__((ext_vector_type(2))) __bf16;
using f32x2_t = __attribute__((ext_vector_type(2))) float;

constexpr int TWO = 30;           // output columns per tile (32 lanes - 2 halo columns)

__device__ __forceinline__ unsigned cvt_pk_bf16(float x0, float x1) {       // lo16 = bf16(x0), hi16 = bf16(x1), RNE
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
__device__ __forceinline__ float from_lane_below(float v) {   // lane n <- lane n-1
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float from_lane_above(float v) {   // lane n <- lane n+1
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}

// accumulator row (MFMA M index) -> tap, or -1.  Row i lives in register (i&3) + 4*(i>>3) of the lanes of
// half (i>>2)&1; lane half 0 carries the (kd,kh) groups 0-4 in registers 3g+kw, half 1 the groups 5-8.
__host__ __device__ inline int head_row_tap(int i) {
    const int half = (i >> 2) & 1, reg = (i & 3) + 4 * (i >> 3);
    if (half == 0) return reg < 15 ? (reg / 3) * 3 + reg % 3 : -1;
    return reg < 12 ? (5 + reg / 3) * 3 + reg % 3 : -1;
}

// CL: the input is CHANNELS-LAST [D][H][W][Cin] (written so by ss_conv3d_bf16s_cl_fwd, the first layer of the same classifier):
// a lane's 8 channels of a position are 32 consecutive bytes, two 16-byte loads instead of eight 4-byte ones -- the plain
// layout keeps the CU's address path busy for 16 cycles per 4-byte wave load, 40 us of this kernel's 78 with as many again
// for the matrix work (tools/_build ablations).
template <int TD, int TH, int KS, int NTERMS, bool CL = false>     // KS = Cin / 16
__global__ __launch_bounds__(256, 2) void conv3d_head_bf16s(const float* __restrict__ in, const uint4* __restrict__ wsplit,
                                                             const float* __restrict__ scale, const float* __restrict__ shift,
                                                             float* __restrict__ out, int D, int H, int W, int tiles_w,
                                                             int tiles_h, int relu) {
    // NTERMS = F16X3 (19): TWO fp16 terms, three products (half the matrix work of the six bf16 ones), block floating point as
    // in conv3d_bf16s.hip: the weights carry ONE power-of-two scale (stored behind the packed terms), every input row is scaled
    // by the power of two of its own maximum (a wave-wide reduce) and its partial rows are un-scaled as they go to LDS.
    constexpr bool F16 = (NTERMS == F16X3);
    constexpr int NC = (NTERMS == 6) ? 3 : 2;
    constexpr int IH = TH + 2, NR = (TD + 2) * IH, Cin = KS * 16;
    extern __shared__ __attribute__((aligned(16))) float S[];     // [9][NR][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    int t = blockIdx.x;
    const int tw = t % tiles_w; t /= tiles_w;
    const int th = t % tiles_h; t /= tiles_h;
    const int w0 = tw * TWO, h0 = th * TH, d0 = t * TD;
    const int b = blockIdx.y;
    const size_t plane = (size_t)H * W, chan = (size_t)D * plane;

    bf16x8 a[KS][NC];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int c = 0; c < NC; ++c) a[ks][c] = __builtin_bit_cast(bf16x8, wsplit[((ks * (F16 ? 2 : 3) + c) * 2 + half) * 32 + l31]);
    // f16 form: 2^-(weight scale), one float behind the KS * 2 * 2 * 32 fragment slots
    const float w_unscale = F16 ? *reinterpret_cast<const float*>(wsplit + KS * 2 * 2 * 32) : 1.0f;

    const __amdgpu_buffer_rsrc_t ires = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(in + (size_t)b * Cin * chan), 0, (int)min((long long)Cin * (long long)chan * 4, 0x7fffffffLL), 0x00020000);
    const int chan_b = (int)(chan * 4);
    const int gw = w0 - 1 + l31;
    const bool col_ok = (unsigned)gw < (unsigned)W;
    // lane part of the offset: this lane's 8-channel block and column (beyond the buffer when outside the row)
    const unsigned lane_off = !col_ok ? 0x80000000u
                              : CL ? (unsigned)((size_t)gw * Cin * 4 + 32 * half) : (unsigned)((size_t)(8 * half) * chan * 4 + (size_t)gw * 4);

    // This wave's input rows r = wave + 4*i, a ring of PD rows ahead of the one being multiplied: the loads of row i + PD are
    // issued right before row i's arithmetic and the scheduler is fenced per row, so every wave keeps PD * KS * 8 loads per
    // lane in flight THROUGH its arithmetic.  (Loaded in groups of 5 rows the compiler hoisted all of a wave's loads to the top
    // of the kernel: the workgroups of a round, started together, first all waited for memory and then all multiplied -- with
    // the loads or the MFMAs removed the kernel took 40 us either way, with both 78.)
    constexpr int NRW = NR / 4, PD = (KS <= 2) ? 3 : 2;
    static_assert(NR % 4 == 0, "rows split evenly over the 4 waves");
    auto load_row = [&](float (&x)[KS][8], int i) {
        const int r = wave + 4 * i;
        // rows outside the volume are requested beyond the buffer (no access, zeros) instead of skipped: a load under a
        // branch makes the compiler's wait-count pass fall back to vmcnt(0) at the merge, i.e. drain the whole prefetch
        const int gd = d0 - 1 + r / IH, gh = h0 - 1 + r % IH;
        const bool ok = (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && col_ok;
        const unsigned off = ok ? lane_off + (unsigned)(((size_t)gd * H + gh) * W * 4 * (CL ? Cin : 1)) : 0x80000000u;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (CL) {
#pragma unroll
                for (int j4 = 0; j4 < 2; ++j4) {
                    const float4 q = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ires, (int)off, ks * 64 + j4 * 16, 0));
                    x[ks][4 * j4] = q.x; x[ks][4 * j4 + 1] = q.y; x[ks][4 * j4 + 2] = q.z; x[ks][4 * j4 + 3] = q.w;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    x[ks][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ires, (int)off, (ks * 16 + j) * chan_b, 0));
            }
        }
    };
    auto process_row = [&](float (&x)[KS][8], int i) {
        const int r = wave + 4 * i;
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.f;
        float row_unscale = 1.0f;
        if constexpr (F16) {
            float m = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(x[ks][j]));
            const int e = max((int)(wave_max_bits(__float_as_uint(m)) >> 23), E_MIN);       // wave-uniform; inf / NaN: 255
            const float in_scale = __uint_as_float((unsigned)(127 + E_ONE - e) << 23);
            row_unscale = __uint_as_float((unsigned)(127 - E_ONE + e) << 23) * w_unscale;     // powers of two: exact
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                unsigned bh[4], bl[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) split2_pk_f16(x[ks][2 * j] * in_scale, x[ks][2 * j + 1] * in_scale, bh[j], bl[j]);
                const f16x8 h8 = __builtin_bit_cast(f16x8, make_uint4(bh[0], bh[1], bh[2], bh[3]));
                const f16x8 l8 = __builtin_bit_cast(f16x8, make_uint4(bl[0], bl[1], bl[2], bl[3]));
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[ks][0]), l8, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[ks][1]), h8, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[ks][0]), h8, acc, 0, 0, 0);
            }
        } else
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {                   // (rows outside the volume were read as zeros)
            unsigned bh[4], bm[4], bl[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) split3_pk(x[ks][2 * j], x[ks][2 * j + 1], bh[j], bm[j], bl[j]);
            const bf16x8 h8 = __builtin_bit_cast(bf16x8, make_uint4(bh[0], bh[1], bh[2], bh[3]));
            const bf16x8 m8 = __builtin_bit_cast(bf16x8, make_uint4(bm[0], bm[1], bm[2], bm[3]));
            if (NTERMS == 6) {
                const bf16x8 l8 = __builtin_bit_cast(bf16x8, make_uint4(bl[0], bl[1], bl[2], bl[3]));
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][1], m8, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][0], l8, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][NC - 1], h8, acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][0], m8, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][1], h8, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][0], h8, acc, 0, 0, 0);
        }
        // kw = 0 comes from the column to the left, kw = 2 from the column to the right
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            float sv = ss::add_rn(ss::add_rn(from_lane_below(acc[3 * q]), acc[3 * q + 1]), from_lane_above(acc[3 * q + 2]));
            if (F16) sv *= row_unscale;
            if (half == 0 || q < 4) S[((half ? 5 + q : q) * NR + r) * 32 + l31] = sv;
        }
    };
    float xr[PD + 1][KS][8];
#pragma unroll
    for (int k = 0; k < PD && k < NRW; ++k) load_row(xr[k], k);
#pragma unroll
    for (int i = 0; i < NRW; ++i) {
        if (i + PD < NRW) load_row(xr[(i + PD) % (PD + 1)], i + PD);
        process_row(xr[i % (PD + 1)], i);
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();

    // ---- phase 2: out[d,h,x] = sum over the 9 (kd,kh) partial rows, fixed order ----
    const float sc = scale ? scale[0] : 1.0f, sh = shift ? shift[0] : 0.0f;
    float* ob = out + (size_t)b * chan;
#pragma unroll
    for (int i = 0; i < TD * TH * 32 / 256; ++i) {
        const int o = tid + 256 * i;
        const int n = o & 31, hh = (o >> 5) % TH, dd = (o >> 5) / TH;
        const int ow = w0 - 1 + n, oh = h0 + hh, od = d0 + dd;
        if (n < 1 || n > TWO || ow >= W || oh >= H || od >= D) continue;
        float v = 0.f;
#pragma unroll
        for (int g = 0; g < 9; ++g) v = ss::add_rn(v, S[(g * NR + (dd + g / 3) * IH + hh + g % 3) * 32 + n]);
        v = ss::add_rn(ss::mul_rn(v, sc), sh);
        if (relu) v = fmaxf(v, 0.f);
        ob[(size_t)od * plane + (size_t)oh * W + ow] = v;
    }
}

// [1,Cin,3,3,3] fp32 -> [Cin/16][3 terms][2 k-halves][32 rows][8] bf16 (rows = taps in accumulator-row order)
__global__ void pack_head_weights_kernel(const float* __restrict__ w, unsigned short* __restrict__ wsplit, int Cin, int total) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int j = i % 8;
    int r = i / 8;
    const int row = r % 32; r /= 32;
    const int hk = r % 2; r /= 2;
    const int term = r % 3;
    const int ks = r / 3;
    const int tap = head_row_tap(row), c = ks * 16 + 8 * hk + j;
    const float x = (tap >= 0 && c < Cin) ? w[(size_t)c * 27 + tap] : 0.f;
    unsigned h, m, l;
    split3_pk(x, 0.f, h, m, l);
    wsplit[i] = (unsigned short)((term == 0 ? h : (term == 1 ? m : l)) & 0xffffu);
}

// the two-term fp16 form: [Cin/16][2 terms][2 k-halves][32 rows][8] fp16 of w * 2^k, k from max |w| (into [2^14, 2^15)), then one
// float 2^-k.  One workgroup: the weights are 27 * Cin <= 1728 values.
__global__ __launch_bounds__(256) void pack_head_weights_f16s_kernel(const float* __restrict__ w, unsigned short* __restrict__ wsplit,
                                                                      int Cin, int total) {
    __shared__ unsigned wmax[4];
    float m = 0.f;
    for (int i = threadIdx.x; i < Cin * 27; i += 256) m = fmaxf(m, fabsf(w[i]));
    const unsigned wm = wave_max_bits(__float_as_uint(m));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = wm;
    __syncthreads();
    const int e = max((int)(max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3])) >> 23), E_MIN);
    const float unscale = __uint_as_float((unsigned)(127 - E_ONE + e) << 23);
    if (threadIdx.x == 0) *reinterpret_cast<float*>(wsplit + total) = unscale;
    for (int i = threadIdx.x; i < total; i += 256) {
        const int j = i % 8;
        int r = i / 8;
        const int row = r % 32; r /= 32;
        const int hk = r % 2; r /= 2;
        const int term = r % 2;
        const int ks = r / 2;
        const int tap = head_row_tap(row), c = ks * 16 + 8 * hk + j;
        const float x = (tap >= 0 && c < Cin) ? w[(size_t)c * 27 + tap] / unscale : 0.f;         // exact: a power of two
        const _Float16 h = (_Float16)x;
        const _Float16 l = (_Float16)(x - (float)h);
        wsplit[i] = __builtin_bit_cast(unsigned short, term == 0 ? h : l);
    }
}

template <int TD, int TH, int KS, int NTERMS, bool CL>
int launch_head(const float* in, const void* wsplit, const float* scale, const float* shift, float* out, int B, int D,
                int H, int W, int relu, hipStream_t st) {
    const int tiles_w = ss::ceil_div(W, TWO), tiles_h = ss::ceil_div(H, TH), tiles_d = ss::ceil_div(D, TD);
    const long long nt = (long long)tiles_w * tiles_h * tiles_d;
    if (nt > 0x7fffffffLL || B > 65535) return SS_ERR_UNSUPPORTED;
    auto kern = conv3d_head_bf16s<TD, TH, KS, NTERMS, CL>;
    const size_t lds = (size_t)9 * (TD + 2) * (TH + 2) * 32 * sizeof(float);
    if (lds > 64 * 1024) {
        if (ss::ensure_dynamic_lds(reinterpret_cast<const void*>(kern), (int)lds) != SS_OK) return SS_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)nt, B), dim3(256), lds, st, in, reinterpret_cast<const uint4*>(wsplit), scale, shift,
                       out, D, H, W, tiles_w, tiles_h, relu);
    return ss::check_launch();
}

template <int KS, int NTERMS, bool CL = false>
int launch_head_tile(const float* in, const void* wsplit, const float* scale, const float* shift, float* out, int B, int D,
                     int H, int W, int relu, hipStream_t st) {
    // 4 planes x 8 rows per workgroup unless that leaves the chip short of workgroups (small volumes)
    const long long big = (long long)ss::ceil_div(W, TWO) * ss::ceil_div(H, 8) * ss::ceil_div(D, 4) * B;
    if (big >= 1024 && D >= 4)
        return launch_head<4, 8, KS, NTERMS, CL>(in, wsplit, scale, shift, out, B, D, H, W, relu, st);
    return launch_head<2, 8, KS, NTERMS, CL>(in, wsplit, scale, shift, out, B, D, H, W, relu, st);
}

}  // namespace

extern "C" int ss_pack_conv3d_head_weights_bf16s(const float* w, void* wsplit, int Cin, ss_stream_t stream) {
    SS_REQUIRE(w && wsplit && Cin > 0 && Cin % 16 == 0);
    const int total = (Cin / 16) * 3 * 2 * 32 * 8;
    hipLaunchKernelGGL(pack_head_weights_kernel, dim3(ss::ceil_div(total, 256)), dim3(256), 0, ss::as_stream(stream), w,
                       reinterpret_cast<unsigned short*>(wsplit), Cin, total);
    return ss::check_launch();
}

extern "C" int ss_pack_conv3d_head_weights_f16s(const float* w, void* wsplit, int Cin, ss_stream_t stream) {
    SS_REQUIRE(w && wsplit && Cin > 0 && Cin % 16 == 0);
    const int total = (Cin / 16) * 2 * 2 * 32 * 8;
    hipLaunchKernelGGL(pack_head_weights_f16s_kernel, dim3(1), dim3(256), 0, ss::as_stream(stream), w,
                       reinterpret_cast<unsigned short*>(wsplit), Cin, total);
    return ss::check_launch();
}

extern "C" int ss_conv3d_head_bf16s_fwd(const float* in, const void* wsplit, const float* scale, const float* shift,
                                        float* out, int B, int Cin, int D, int H, int W, int relu, int nterms,
                                        ss_stream_t stream) {
    SS_REQUIRE(in && wsplit && out);
    SS_REQUIRE(B > 0 && Cin > 0 && D > 0 && H > 0 && W > 0 && (nterms == 3 || nterms == 6 || nterms == F16X3));
    SS_REQUIRE((reinterpret_cast<uintptr_t>(wsplit) & 15) == 0);
    if ((long long)Cin * D * H * W * 4 >= 0x7fffffffLL) return SS_ERR_UNSUPPORTED;     // 32-bit buffer offsets per pair
    hipStream_t st = ss::as_stream(stream);
#define SS_HEAD(KSV)                                                                                                  \
    if (Cin == 16 * KSV)                                                                                              \
        return nterms == 6 ? launch_head_tile<KSV, 6>(in, wsplit, scale, shift, out, B, D, H, W, relu, st)            \
             : nterms == 3 ? launch_head_tile<KSV, 3>(in, wsplit, scale, shift, out, B, D, H, W, relu, st)            \
                           : launch_head_tile<KSV, F16X3>(in, wsplit, scale, shift, out, B, D, H, W, relu, st);
    SS_HEAD(1)
    SS_HEAD(2)
    SS_HEAD(4)
#undef SS_HEAD
    return SS_ERR_UNSUPPORTED;
}

// ---------------------------------------------------------------------------------------------------
// 1x1x1 Conv3d / Linear over channels (+ per-channel affine = bias, ReLU) on the same split-bf16
// arithmetic: qkv_3d and final1x1 of attention_block (reference models/submodule_other.py:804, 835).
//   out[co, p] = sum_ci W[co, ci] * in[ci, p]        p = flattened (d, h, w)
// A wave owns 32 output channels with ALL their weights resident in registers (Cin <= 128: 8 K-steps x
// 3 terms x 4 registers) and walks over tiles of 32 consecutive positions; the activation operand is
// read straight from global memory (lane n = position, 8 channels per lane half), split in registers.
// No LDS, no barriers.
namespace {

template <int KS, int NTERMS>     // KS = Cin / 16
__global__ __launch_bounds__(256, 2) void pointwise_bf16s(const float* __restrict__ in, const uint4* __restrict__ wsplit,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           float* __restrict__ out, int Cout, long long npos,
                                                           int tiles_per_wave, int relu) {
    constexpr int NC = (NTERMS == 6) ? 3 : 2;
    constexpr int Cin = KS * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int mt = blockIdx.y * 4 + wave;                     // this wave's tile of 32 output channels
    const int b = blockIdx.z;
    if (mt * 32 >= Cout) return;
    bf16x8 a[KS][NC];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int c = 0; c < NC; ++c)
            a[ks][c] = __builtin_bit_cast(bf16x8, wsplit[(((size_t)mt * KS + ks) * 3 + c) * 64 + half * 32 + l31]);

    const __amdgpu_buffer_rsrc_t ires = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(in + (size_t)b * Cin * npos), 0, (int)min((long long)Cin * npos * 4, 0x7fffffffLL), 0x00020000);
    const int chan_b = (int)(npos * 4);
    const __amdgpu_buffer_rsrc_t ores = __builtin_amdgcn_make_buffer_rsrc(
        out + (size_t)b * Cout * npos, 0, (int)min((long long)Cout * npos * 4, 0x7fffffffLL), 0x00020000);
    // this wave's 32 (scale, shift) pairs, parked in LDS now (written and read by the same wave: no barrier): fetched
    // after the MFMAs of every tile they were an exposed round trip to L2 per tile
    __shared__ float aff[4][64];
    if (lane < 32) {
        const int co = min(mt * 32 + lane, Cout - 1);
        aff[wave][lane] = scale ? scale[co] : 1.0f;
        aff[wave][32 + lane] = shift ? shift[co] : 0.0f;
    }
    const float floor_v = relu ? 0.f : -__builtin_inff();
    const long long t0 = (long long)blockIdx.x * tiles_per_wave;
    float x[KS][8];
    auto lane_offset = [&](long long tile) {
        const long long p = tile * 32 + l31;
        // positions beyond the volume get an offset beyond the buffer -> zeros
        return (p < npos) ? (unsigned)((8LL * half * npos + p) * 4) : 0x80000000u;
    };
    auto load_step = [&](unsigned off, int ks) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            x[ks][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ires, (int)off, (ks * 16 + j) * chan_b, 0));
    };
    {
        const unsigned off0 = lane_offset(t0);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) load_step(off0, ks);
    }
#pragma unroll 1
    for (int it = 0; it < tiles_per_wave; ++it) {
        const long long tile = t0 + it;
        if (tile * 32 >= npos) break;
        // the next tile's loads are issued unconditionally (a load under a branch costs a vmcnt(0) at the merge): beyond
        // npos, or after the last tile, every lane reads zeros from beyond the buffer
        const unsigned offn = (it + 1 < tiles_per_wave) ? lane_offset(tile + 1) : 0x80000000u;
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            // split this K-step's 8 channels, then refill its registers with the next tile's
            unsigned bh[4], bm[4], bl[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) split3_pk(x[ks][2 * j], x[ks][2 * j + 1], bh[j], bm[j], bl[j]);
            load_step(offn, ks);
            const bf16x8 h8 = __builtin_bit_cast(bf16x8, make_uint4(bh[0], bh[1], bh[2], bh[3]));
            const bf16x8 m8 = __builtin_bit_cast(bf16x8, make_uint4(bm[0], bm[1], bm[2], bm[3]));
            if (NTERMS == 6) {
                const bf16x8 l8 = __builtin_bit_cast(bf16x8, make_uint4(bl[0], bl[1], bl[2], bl[3]));
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][1], m8, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][0], l8, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][NC - 1], h8, acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][0], m8, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][1], h8, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][0], h8, acc, 0, 0, 0);
        }
        const long long p = tile * 32 + l31;
        const unsigned vo = (p < npos) ? (unsigned)((4LL * half * npos + p) * 4) : 0x80000000u;     // beyond the buffer: dropped
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cl = (r & 3) + 8 * (r >> 2);
            const float v = fmaxf(ss::add_rn(ss::mul_rn(acc[r], aff[wave][cl + 4 * half]), aff[wave][32 + cl + 4 * half]), floor_v);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ores,
                                                  (int)((mt * 32 + cl + 4 * half < Cout) ? vo : 0x80000000u), (mt * 32 + cl) * chan_b, 0);
        }
    }
}

// [Cout,Cin] fp32 -> [ceil(Cout/32)][Cin/16][3 terms][2 k-halves][32 rows][8] bf16
__global__ void pack_pointwise_weights_kernel(const float* __restrict__ w, unsigned short* __restrict__ wsplit, int Cout,
                                              int Cin, int total) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int j = i % 8;
    int r = i / 8;
    const int row = r % 32; r /= 32;
    const int hk = r % 2; r /= 2;
    const int term = r % 3; r /= 3;
    const int KS = Cin / 16;
    const int ks = r % KS;
    const int mt = r / KS;
    const int co = mt * 32 + row, c = ks * 16 + 8 * hk + j;
    const float x = (co < Cout) ? w[(size_t)co * Cin + c] : 0.f;
    unsigned h, m, l;
    split3_pk(x, 0.f, h, m, l);
    wsplit[i] = (unsigned short)((term == 0 ? h : (term == 1 ? m : l)) & 0xffffu);
}

template <int KS, int NTERMS>
int launch_pointwise(const float* in, const void* wsplit, const float* scale, const float* shift, float* out, int B,
                     int Cout, long long npos, int relu, hipStream_t st) {
    const long long ntiles = (npos + 31) / 32;
    const int mgroups = ss::ceil_div(ss::ceil_div(Cout, 32), 4);
    // tiles of 32 positions per wave: as many as keeps >= 1024 workgroups (4 per CU), at most 8 --
    // the weights are loaded once per wave (24 KB), so more tiles per wave amortise them better
    int tpw = 8;
    while (tpw > 1 && ((ntiles + tpw - 1) / tpw) * mgroups * B < 1024) tpw >>= 1;
    const long long gx = (ntiles + tpw - 1) / tpw;
    if (gx > 0x7fffffffLL || B > 65535 || mgroups > 65535) return SS_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((pointwise_bf16s<KS, NTERMS>), dim3((unsigned)gx, mgroups, B), dim3(256), 0, st, in,
                       reinterpret_cast<const uint4*>(wsplit), scale, shift, out, Cout, npos, tpw, relu);
    return ss::check_launch();
}

}  // namespace

extern "C" int ss_pack_pointwise_weights_bf16s(const float* w, void* wsplit, int Cout, int Cin, ss_stream_t stream) {
    SS_REQUIRE(w && wsplit && Cout > 0 && Cin > 0 && Cin % 16 == 0);
    const int total = ss::ceil_div(Cout, 32) * (Cin / 16) * 3 * 2 * 32 * 8;
    hipLaunchKernelGGL(pack_pointwise_weights_kernel, dim3(ss::ceil_div(total, 256)), dim3(256), 0, ss::as_stream(stream), w,
                       reinterpret_cast<unsigned short*>(wsplit), Cout, Cin, total);
    return ss::check_launch();
}

extern "C" int ss_conv3d_pointwise_bf16s_fwd(const float* in, const void* wsplit, const float* scale, const float* shift,
                                             float* out, int B, int Cin, int Cout, long long npos, int relu, int nterms,
                                             ss_stream_t stream) {
    SS_REQUIRE(in && wsplit && out);
    SS_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && npos > 0 && (nterms == 3 || nterms == 6));
    SS_REQUIRE((reinterpret_cast<uintptr_t>(wsplit) & 15) == 0);
    if ((long long)Cin * npos * 4 >= 0x7fffffffLL) return SS_ERR_UNSUPPORTED;     // 32-bit buffer offsets per pair
    if ((long long)Cout * npos * 4 >= 0x7fffffffLL) return SS_ERR_UNSUPPORTED;
    hipStream_t st = ss::as_stream(stream);
#define SS_PW(KSV)                                                                                                    \
    if (Cin == 16 * KSV)                                                                                              \
        return nterms == 6 ? launch_pointwise<KSV, 6>(in, wsplit, scale, shift, out, B, Cout, npos, relu, st)         \
                           : launch_pointwise<KSV, 3>(in, wsplit, scale, shift, out, B, Cout, npos, relu, st);
    SS_PW(2)
    SS_PW(4)
    SS_PW(8)
#undef SS_PW
    return SS_ERR_UNSUPPORTED;
}

extern "C" int ss_conv3d_head_bf16s_cl_fwd(const float* in, const void* wsplit, const float* scale, const float* shift,
                                           float* out, int B, int Cin, int D, int H, int W, int relu, int nterms,
                                           ss_stream_t stream) {
    SS_REQUIRE(in && wsplit && out);
    SS_REQUIRE(B > 0 && Cin > 0 && D > 0 && H > 0 && W > 0 && (nterms == 3 || nterms == 6 || nterms == F16X3));
    SS_REQUIRE(((reinterpret_cast<uintptr_t>(wsplit) | reinterpret_cast<uintptr_t>(in)) & 15) == 0);
    if ((long long)Cin * D * H * W * 4 >= 0x7fffffffLL) return SS_ERR_UNSUPPORTED;
    hipStream_t st = ss::as_stream(stream);
    if (Cin == 32)
        return nterms == 6 ? launch_head_tile<2, 6, true>(in, wsplit, scale, shift, out, B, D, H, W, relu, st)
             : nterms == 3 ? launch_head_tile<2, 3, true>(in, wsplit, scale, shift, out, B, D, H, W, relu, st)
                           : launch_head_tile<2, F16X3, true>(in, wsplit, scale, shift, out, B, D, H, W, relu, st);
    return SS_ERR_UNSUPPORTED;
}
