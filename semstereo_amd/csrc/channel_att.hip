// channelAtt.im_att (models/SemStereo.py:89-100) as ONE kernel:
//   logits[cv, p] = W2[cv, :] . relu( bn( W1 . im[:, p] ) ) + b2[cv]          (optionally sigmoid(logits))
// a 1x1 Conv2d (no bias) -> BatchNorm2d(eval) -> ReLU (BasicConv, models/submodule.py:89-116) -> 1x1 Conv2d (+bias) over
// the 2-D image features, whose sigmoid gates the cost volume (:101-102).  The reference runs it as two cuBLAS/MIOpen
// GEMM calls, a BatchNorm kernel, a clamp and (later) a sigmoid; here both projections run back to back on the matrix
// core with the hidden map never leaving the CU.
//
// Arithmetic: fp32 operands split exactly into three bf16 terms, six cross products on v_mfma_f32_32x32x16_bf16, fp32
// accumulation -- the same fp32-class arithmetic as the 1x1x1 projections of conv3d_head.hip (weights come packed by
// ss_pack_pointwise_weights_bf16s: [ceil(Cout/32)][Cin/16][3 terms][2 k-halves][32 rows][8] bf16).
//
// Workgroup = 4 waves, 64 consecutive positions (two column tiles of 32).  Stage 1: wave w owns 32 hidden channels
// (CMID = 128: one row tile per wave, both column tiles; CMID = 64: row tile w & 1, column tile w >> 1); the image operand
// is read straight from global memory (lane = position, 8 channels per lane half; the 4 waves' re-reads hit the L1), the
// weight fragments stream from L2 one K-step ahead.  The hidden tile goes through LDS as fp32 [CMID][64]; stage 2 (32
// output channels) is done by waves 0 and 1, one column tile each.  ~1.3 GFLOP per call: the kernel is bound by the read
// of the image features (16.8 MB at 1/8 scale, 33.5 MB at 1/4 scale).
#include "common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) short;
using f32x2_t = __attribute__((ext_vector_type(2))) float;
using bf16x2_t = __attribute__((ext_vector_type(2))) __bf16;

__device__ __forceinline__ unsigned cvt_pk_bf16(float x0, float x1) {
    const f32x2_t v = {x0, x1};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
// (x0, x1) -> packed (hi, mid, lo) bf16 pairs with hi + mid + lo == x up to 2^-25 |x|
__device__ __forceinline__ void split3_pk(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    h = cvt_pk_bf16(x0, x1);
    const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
    m = cvt_pk_bf16(r0, r1);
    const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
    l = cvt_pk_bf16(s0, s1);
}

// six cross products of (a.hi, a.mid, a.lo) x (b.hi, b.mid, b.lo), smallest first
__device__ __forceinline__ f32x16 mfma6(const uint4 (&a)[3], const float (&x)[8], f32x16 acc) {
    unsigned bh[4], bm[4], bl[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) split3_pk(x[2 * j], x[2 * j + 1], bh[j], bm[j], bl[j]);
    const bf16x8 h8 = __builtin_bit_cast(bf16x8, make_uint4(bh[0], bh[1], bh[2], bh[3]));
    const bf16x8 m8 = __builtin_bit_cast(bf16x8, make_uint4(bm[0], bm[1], bm[2], bm[3]));
    const bf16x8 l8 = __builtin_bit_cast(bf16x8, make_uint4(bl[0], bl[1], bl[2], bl[3]));
    const bf16x8 a0 = __builtin_bit_cast(bf16x8, a[0]), a1 = __builtin_bit_cast(bf16x8, a[1]), a2 = __builtin_bit_cast(bf16x8, a[2]);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, m8, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, l8, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, h8, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, m8, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, h8, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, h8, acc, 0, 0, 0);
    return acc;
}

constexpr int NP = 64;            // positions per workgroup
constexpr int LDP = NP + 4;       // LDS row pitch of the hidden tile (floats)

template <int CIN, int CMID>
__global__ __launch_bounds__(256, 2) void channel_att_kernel(const float* __restrict__ im, const uint4* __restrict__ w1,
                                                             const float* __restrict__ scale1, const float* __restrict__ shift1,
                                                             const uint4* __restrict__ w2, const float* __restrict__ bias2,
                                                             float* __restrict__ out, int npos, int sigmoid) {
    static_assert(CMID == 128 || CMID == 64, "hidden width of the reference's two gates");
    constexpr int KS1 = CIN / 16, KS2 = CMID / 16;
    constexpr int NTW = (CMID == 128) ? 2 : 1;               // column tiles per wave in stage 1
    __shared__ float hid[CMID * LDP];
    __shared__ float aff[2 * CMID + 32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int b = blockIdx.y;
    const int p0 = blockIdx.x * NP;
    for (int i = tid; i < CMID; i += 256) {
        aff[i] = scale1 ? scale1[i] : 1.0f;
        aff[CMID + i] = shift1 ? shift1[i] : 0.0f;
    }
    if (tid < 32) aff[2 * CMID + tid] = bias2 ? bias2[tid] : 0.0f;

    const int mt = (CMID == 128) ? wave : (wave & 1);         // this wave's tile of 32 hidden channels
    const int nt0 = (CMID == 128) ? 0 : (wave >> 1);          // its first column tile
    const __amdgpu_buffer_rsrc_t ires = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(im + (size_t)b * CIN * npos), 0, (int)min((long long)CIN * npos * 4, 0x7fffffffLL), 0x00020000);
    const int chan_b = npos * 4;
    unsigned voff[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        const int p = p0 + (nt0 + t) * 32 + l31;
        voff[t] = (p < npos) ? (unsigned)(((size_t)8 * half * npos + p) * 4) : 0x80000000u;   // beyond the buffer: zeros
    }
    auto load_x = [&](float (&x)[NTW][8], int ks) {
#pragma unroll
        for (int t = 0; t < NTW; ++t)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                x[t][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ires, (int)voff[t], (ks * 16 + j) * chan_b, 0));
    };
    auto load_w1 = [&](uint4 (&a)[3], int ks) {
#pragma unroll
        for (int c = 0; c < 3; ++c) a[c] = w1[(((size_t)mt * KS1 + ks) * 3 + c) * 64 + lane];
    };
    f32x16 acc[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;
    // ---- stage 1: hidden = W1 . im, operands one K-step ahead (all loads unconditional: past the end they re-request
    // the last step, see the wait-count note in conv3d_bf16s.hip) ----
    float xc[NTW][8], xn[NTW][8];
    uint4 ac[3], an[3];
    load_x(xc, 0);
    load_w1(ac, 0);
#pragma unroll 2
    for (int ks = 0; ks < KS1; ++ks) {
        const int nx = min(ks + 1, KS1 - 1);
        load_x(xn, nx);
        load_w1(an, nx);
#pragma unroll
        for (int t = 0; t < NTW; ++t) acc[t] = mfma6(ac, xc[t], acc[t]);
#pragma unroll
        for (int t = 0; t < NTW; ++t)
#pragma unroll
            for (int j = 0; j < 8; ++j) xc[t][j] = xn[t][j];
#pragma unroll
        for (int c = 0; c < 3; ++c) ac[c] = an[c];
    }
    __syncthreads();                                          // aff[] is in place (and, on later reuse, hid[] is free)
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cm = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            // BatchNorm(eval) as x * scale + shift (two roundings, like ATen's inference path), then ReLU
            const float v = fmaxf(ss::add_rn(ss::mul_rn(acc[t][r], aff[cm]), aff[CMID + cm]), 0.f);
            hid[cm * LDP + (nt0 + t) * 32 + l31] = v;
        }
    __syncthreads();
    // ---- stage 2: 32 output channels, waves 0 and 1 take one column tile each ----
    if (wave < 2) {
        f32x16 o;
#pragma unroll
        for (int q = 0; q < 16; ++q) o[q] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks) {
            uint4 a2[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) a2[c] = w2[((size_t)ks * 3 + c) * 64 + lane];
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = hid[(ks * 16 + half * 8 + j) * LDP + wave * 32 + l31];
            o = mfma6(a2, x, o);
        }
        const int p = p0 + wave * 32 + l31;
        if (p < npos) {
            float* ob = out + (size_t)b * 32 * npos + p;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cv = (r & 3) + 8 * (r >> 2) + 4 * half;
                float v = ss::add_rn(o[r], aff[2 * CMID + cv]);
                if (sigmoid) v = 1.0f / (1.0f + expf(-v));
                ob[(size_t)cv * npos] = v;
            }
        }
    }
}

template <int CIN, int CMID>
int launch_catt(const float* im, const void* w1, const float* scale1, const float* shift1, const void* w2, const float* bias2,
                float* out, int B, int npos, int sigmoid, hipStream_t st) {
    const dim3 grid(ss::ceil_div(npos, NP), B);
    hipLaunchKernelGGL((channel_att_kernel<CIN, CMID>), grid, dim3(256), 0, st, im, reinterpret_cast<const uint4*>(w1), scale1,
                       shift1, reinterpret_cast<const uint4*>(w2), bias2, out, npos, sigmoid);
    return ss::check_launch();
}

}  // namespace

extern "C" int ss_channel_att_logits_fwd(const float* im, const void* w1_split, const float* scale1, const float* shift1,
                                         const void* w2_split, const float* bias2, float* out, int B, int Cin, int Cmid,
                                         int Cout, int H, int W, int sigmoid, ss_stream_t stream) {
    SS_REQUIRE(im && w1_split && w2_split && out);
    SS_REQUIRE(B > 0 && Cin > 0 && Cmid > 0 && Cout > 0 && H > 0 && W > 0);
    SS_REQUIRE(((reinterpret_cast<uintptr_t>(w1_split) | reinterpret_cast<uintptr_t>(w2_split)) & 15) == 0);
    const long long npos = (long long)H * W;
    if (Cout != 32 || B > 65535 || (long long)Cin * npos * 4 >= 0x7fffffffLL) return SS_ERR_UNSUPPORTED;
    hipStream_t st = ss::as_stream(stream);
    if (Cin == 256 && Cmid == 128)
        return launch_catt<256, 128>(im, w1_split, scale1, shift1, w2_split, bias2, out, B, (int)npos, sigmoid, st);
    if (Cin == 128 && Cmid == 64)
        return launch_catt<128, 64>(im, w1_split, scale1, shift1, w2_split, bias2, out, B, (int)npos, sigmoid, st);
    return SS_ERR_UNSUPPORTED;
}
