// Weight gradient of the 3x3x3 convolutions of the aggregation stack on the bf16 matrix core (training: main_us3d.py:186-222
// back-propagates through convbn_3d / BasicConv / the hourglasses' ConvTranspose3d layers, models/SemStereo.py:106-182, 228-236):
//
//   dW[co, ci, kd, kh, kw] = sum_{b, od, oh, ow} gout[b, co, od, oh, ow] * in[b, ci, od*S + kd - 1, oh*S + kh - 1, ow*S + kw - 1]
//
// (zero padding 1, stride S = 1 or 2; ConvTranspose3d(k3, s2, p1, op1): the S = 2 form with the roles swapped, see
// conv3d_wgrad.hip, whose exact-fp32 kernel this one replaces: measured r06 at 13 TFLOP/s, 66 % of the training step).
//
// A GEMM with M = Cout, N = Cin per tap and K = every output position.  Both operands are rows of an NCDHW tensor, so "8
// consecutive k per lane" -- what v_mfma_f32_32x32x16_bf16 wants of A (lane = output channel) and B (lane = input channel) -- are 8
// consecutive W positions of one channel row: the natural layout.  fp32 is kept by the split of conv3d_bf16s.hip: x = hi + mid +
// lo bf16 terms, six cross products, fp32 accumulation (error below the exact-fp32 MFMA's; bf16 has fp32's exponent range, so
// gradients of any magnitude need no scaling).
//
// Three kernels share that arithmetic (ss_conv3d_wgrad_bf16s_fwd picks): `conv3d_wgrad_bf16s_coop` for stride 1 (nine waves walk down a
// column of chunks and share what they stage: the form every big layer runs, further down), `conv3d_wgrad_bf16s<S>` below for stride 2 and
// the transposed convs' weights (and for stride 1 under SS_WGRAD_COOP=0), `conv3d_wgrad_head_bf16s` for a single output channel.
//
// conv3d_wgrad_bf16s<S>: one WAVE is one unit of work, nothing is shared between waves (no barrier in the kernel): a (32 co x 32 ci) tile pair, ONE
// kernel row (kd, kh) (3 taps = 3 accumulator tiles: 48 registers, so three waves share a SIMD and one's staging hides under the
// others' MFMAs -- all 9 taps of a depth plane in one wave need 144 and spilled) and a contiguous range of 32-position chunks of
// output rows.  Per chunk:
//   * A (gout): each lane loads its channel's 2 x 8 positions straight from global memory and splits them in registers -- the two
//     fragments serve the 3 taps x 6 products;
//   * B (in): the wave stages its input row's 32 ci x (32 + halo) positions -- coalesced 4-byte loads
//     along W, register-prefetched one chunk ahead, split, written as bf16 pairs -- into its private LDS tile [term][ci][w] and reads
//     the fragments of the three kw taps from it: kw = 1 is the aligned 16-byte word, kw = 0 / 2 are that word shifted by one
//     element (v_alignbit on the word and one neighbouring dword).  Stride 2: even and odd input columns are staged as two rows, so
//     kw = 1 / 2 are aligned words of the even / odd row and kw = 0 the odd row shifted by one.
// Row strides of 112 / 80 bytes make the lane = channel fragment reads conflict-free.  The 3 x 16 accumulator registers of a wave
// are added into a workspace laid out [tile][tap][register][lane] (coalesced fp32 atomics: 256 bytes per instruction; lanes of dW
// itself would be 108 bytes apart), which a second small kernel re-orders into dW [Cout, Cin, 27].
#include <algorithm>
#include <type_traits>

#include "common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) short;
using f32x2_w = __attribute__((ext_vector_type(2))) float;
using bf16x2_w = __attribute__((ext_vector_type(2))) __bf16;

__device__ __forceinline__ unsigned cvt_pk_bf16_w(float x0, float x1) {     // lo16 = bf16(x0), hi16 = bf16(x1), RNE
    const f32x2_w v = {x0, x1};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_w));
}
__device__ __forceinline__ void split3_pk_w(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    h = cvt_pk_bf16_w(x0, x1);
    const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
    m = cvt_pk_bf16_w(r0, r1);
    const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
    l = cvt_pk_bf16_w(s0, s1);
}

constexpr int CW = 32;                                    // output positions per chunk (two K-steps of 16)
template <int S>
struct WG {
    // S = 1: one row [8 pad | 32 | 8 pad] bf16 per (term, ci): 96 bytes, stride 112; S = 2: rows E [32] and O [8 pad | 32]: stride 80
    static constexpr int RS = (S == 1) ? 112 : 80;
    static constexpr int PLANE = 32 * RS;                 // bytes per (row kind, term)
    static constexpr int NPL = (S == 1) ? 3 : 6;
    static constexpr int TILE = NPL * PLANE;              // bytes per wave
    static constexpr int NLD = (S == 1) ? 2 : 4;          // loads per lane and channel iteration of the staging
};

template <int S>
__global__ __launch_bounds__(192, S == 1 ? 3 : 2) void conv3d_wgrad_bf16s(const float* __restrict__ gout, const float* __restrict__ in,
                                                              float* __restrict__ ws, int Cin, int Cout, int D, int H, int W, int Do,
                                                              int Ho, int Wo, int chunks_per_row, int total_chunks, int chunks_per_unit,
                                                              int nsplit, int ci_tiles) {
    using C = WG<S>;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    // (the wave index as a SCALAR: everything derived from it -- the unit, its chunk walk, the row offsets of the buffer loads -- is then
    // scalar arithmetic; as a per-lane value every buffer load's scalar offset needed a waterfall loop: 164 of them in the first build)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    // a workgroup = (chunk range `split`, kernel-depth plane kd), its three waves the kernel rows kh.  The nine waves of a chunk range
    // read the same gout chunks and overlapping input rows: the three workgroups of a range get block indices 8 apart, i.e. the SAME
    // XCD (workgroups go round-robin over the 8 XCDs), a few dispatches from each other, so that what one fetches the others find in
    // that XCD's L2.  (First build: consecutive waves = consecutive kernel rows, spread over 2-3 XCDs -- every XCD fetched its own copy
    // over the fabric, 1.9 GB for a 0.4 GB layer, and a chunk took ~20 000 cycles of a wave's life.)
    const int kh = wave;
    const int kd = (blockIdx.x / 8) % 3, split = (blockIdx.x / 24) * 8 + (blockIdx.x % 8);
    if (split >= nsplit) return;                           // (no barrier anywhere: a wave may leave)
    const int c_begin = split * chunks_per_unit, c_end = min(c_begin + chunks_per_unit, total_chunks);
    const int co0 = (blockIdx.y / ci_tiles) * 32, ci0 = (blockIdx.y % ci_tiles) * 32;
    const int b = blockIdx.z;
    unsigned char* tile = lds_raw + wave * C::TILE;

    const long long ochan = (long long)Do * Ho * Wo, ichan = (long long)D * H * W;
    const __amdgpu_buffer_rsrc_t gres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(gout + (long long)b * Cout * ochan), 0, (int)min((long long)Cout * ochan * 4, 0x7fffffffLL), 0x00020000);
    const __amdgpu_buffer_rsrc_t ires = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(in + (long long)b * Cin * ichan), 0, (int)min((long long)Cin * ichan * 4, 0x7fffffffLL), 0x00020000);
    // A: lane = (output channel l31, positions 8 * half ..)
    const unsigned a_lane = (co0 + l31 < Cout) ? (unsigned)((co0 + l31) * ochan * 4) + 32u * half : 0x80000000u;
    // B staging: lane = (position pair q = lane & 15, channel 4 i + (lane >> 4))
    const int q = lane & 15, csub = lane >> 4;
    // B fragments: lane = (input channel l31, word half)
    const int frag_base = l31 * C::RS;

    f32x16 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    float rin[8 * C::NLD], rhalo = 0.f;
    // ---- issue the loads of input row (id, ih), chunk at output column w0 (values outside the row: masked when they are used) ----
    auto issue_in = [&](int id, int ih, int w0) {
        const unsigned row_b = (unsigned)((((long long)id * H + ih) * W) * 4);
        const int iw0 = (w0 + 2 * q) * S;                  // first input column of this lane's pair
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = ci0 + 4 * i + csub;
            const unsigned ch = (c < Cin) ? (unsigned)(c * ichan * 4) : 0x80000000u;
#pragma unroll
            for (int e = 0; e < C::NLD; ++e)
                rin[i * C::NLD + e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ires, (int)(ch + (unsigned)((iw0 + e) * 4)), (int)row_b, 0));
        }
        // halo: S = 1: columns w0 - 1 (lanes 0-31) and w0 + 32 (lanes 32-63) of channel l31; S = 2: column 2 w0 - 1 (lanes 0-31)
        const int hw = (S == 1) ? (half ? w0 + CW : w0 - 1) : 2 * w0 - 1;
        const bool hok = (ci0 + l31 < Cin) && hw >= 0 && hw < W && (S == 1 || half == 0);
        rhalo = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                              ires, (int)(hok ? (unsigned)((ci0 + l31) * ichan * 4) + (unsigned)(hw * 4) : 0x80000000u), (int)row_b, 0));
    };
    // ---- split the loaded row and write it to this wave's LDS tile ----
    auto store_in = [&](int w0, auto whole_tag) {           // WHOLE: every column this chunk reads lies inside the row (wave-uniform): no masks
        constexpr bool WHOLE = decltype(whole_tag)::value;
        const int iw0 = (w0 + 2 * q) * S;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float v[C::NLD];
#pragma unroll
            for (int e = 0; e < C::NLD; ++e) v[e] = (WHOLE || iw0 + e < W) ? rin[i * C::NLD + e] : 0.f;      // (beyond the row: the next row's data)
            const int c = 4 * i + csub;
            if constexpr (S == 1) {
                unsigned h, m, l;
                split3_pk_w(v[0], v[1], h, m, l);
                unsigned char* p = tile + c * C::RS + 16 + 4 * q;
                *reinterpret_cast<unsigned*>(p) = h;
                *reinterpret_cast<unsigned*>(p + C::PLANE) = m;
                *reinterpret_cast<unsigned*>(p + 2 * C::PLANE) = l;
            } else {
                unsigned h, m, l;
                split3_pk_w(v[0], v[2], h, m, l);          // even columns: E[j], E[j + 1]
                unsigned char* p = tile + c * C::RS + 4 * q;
                *reinterpret_cast<unsigned*>(p) = h;
                *reinterpret_cast<unsigned*>(p + C::PLANE) = m;
                *reinterpret_cast<unsigned*>(p + 2 * C::PLANE) = l;
                split3_pk_w(v[1], v[3], h, m, l);          // odd columns: O[j], O[j + 1]
                p = tile + 3 * C::PLANE + c * C::RS + 16 + 4 * q;
                *reinterpret_cast<unsigned*>(p) = h;
                *reinterpret_cast<unsigned*>(p + C::PLANE) = m;
                *reinterpret_cast<unsigned*>(p + 2 * C::PLANE) = l;
            }
        }
        if (S == 1 || half == 0) {                         // the halo element(s) of channel l31: element 7 (left) / 40 (right) of the row
            unsigned h, m, l;
            split3_pk_w(rhalo, 0.f, h, m, l);
            unsigned char* p = tile + (S == 1 ? 0 : 3 * C::PLANE) + l31 * C::RS + ((S == 1 && half) ? 80 : 14);
            *reinterpret_cast<unsigned short*>(p) = (unsigned short)h;
            *reinterpret_cast<unsigned short*>(p + C::PLANE) = (unsigned short)m;
            *reinterpret_cast<unsigned short*>(p + 2 * C::PLANE) = (unsigned short)l;
        }
    };
    auto rd128 = [&](int off) { return *reinterpret_cast<const uint4*>(tile + off); };
    auto rd32 = [&](int off) { return *reinterpret_cast<const unsigned*>(tile + off); };

    // the chunks of this unit whose input row exists, walked with the loads one chunk ahead
    int nc = c_begin - 1, nod = 0, noh = 0, nw0 = 0;        // the NEXT chunk to be loaded
    auto advance = [&]() -> bool {                          // -> nc (and nod, noh, nw0) at the next chunk with an input row, false: none left
        for (++nc; nc < c_end; ++nc) {
            const int row = nc / chunks_per_row;
            nod = row / Ho; noh = row - nod * Ho; nw0 = (nc - row * chunks_per_row) * CW;
            if ((unsigned)(nod * S + kd - 1) < (unsigned)D && (unsigned)(noh * S + kh - 1) < (unsigned)H) return true;
        }
        return false;
    };
    float av[2][8];                                         // A of the chunk loaded last: 2 K-steps x 8 positions of channel l31
    // (lane = channel: a load instruction touches 32 cache lines whatever its width -- 16 bytes per lane where the rows are 16-byte
    // aligned, i.e. Wo % 4 == 0, every layer of the model: 4 instructions per chunk instead of 16)
    const bool a_wide = (Wo & 3) == 0;
    auto issue_a = [&]() {
        const unsigned a_row = (unsigned)((((long long)nod * Ho + noh) * Wo + nw0) * 4);
        if (a_wide) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int e = 0; e < 8; e += 4) {
                    const float4 v4 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(gres, (int)(a_lane + (unsigned)((16 * s + e) * 4)), (int)a_row, 0));
                    av[s][e] = v4.x; av[s][e + 1] = v4.y; av[s][e + 2] = v4.z; av[s][e + 3] = v4.w;
                }
        } else {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    av[s][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(gres, (int)(a_lane + (unsigned)((16 * s + e) * 4)), (int)a_row, 0));
        }
    };
    bool have = advance();
    if (have) { issue_a(); issue_in(nod * S + kd - 1, noh * S + kh - 1, nw0); }
#ifdef SS_EXP_WG_NOSTAGE
    bool first_chunk = true;
#endif
    while (have) {
        const int w0 = nw0;
        // ---- A fragments of this chunk: 2 K-steps x 8 positions of channel l31, split in registers (loaded a chunk ago) ----
        uint4 af[2][3];                                     // [K-step][term]
        // (a chunk that lies inside its rows -- all but the last of a row, and that one too when the widths are multiples of 32 -- takes
        // the copies without the per-element "inside the row" selects: a quarter of the staging's vector instructions)
        const bool whole = w0 + CW <= Wo && (w0 + CW) * S + 1 <= W;
        auto split_a = [&](auto whole_tag) {
            constexpr bool WHOLE = decltype(whole_tag)::value;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                unsigned h[4], m[4], l[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int w = w0 + 16 * s + 8 * half + 2 * e;
                    split3_pk_w((WHOLE || w < Wo) ? av[s][2 * e] : 0.f, (WHOLE || w + 1 < Wo) ? av[s][2 * e + 1] : 0.f, h[e], m[e], l[e]);
                }
                af[s][0] = make_uint4(h[0], h[1], h[2], h[3]);
                af[s][1] = make_uint4(m[0], m[1], m[2], m[3]);
                af[s][2] = make_uint4(l[0], l[1], l[2], l[3]);
            }
        };
#ifdef SS_EXP_WG_NOSTAGE          // (timing experiment, wrong results: split + LDS staging only for a wave's first chunk)
        if (first_chunk)
#endif
        {
        if (whole) { split_a(std::true_type{}); store_in(w0, std::true_type{}); }
        else { split_a(std::false_type{}); store_in(w0, std::false_type{}); }
        }
#ifdef SS_EXP_WG_NOSTAGE
        first_chunk = false;
#endif
        __builtin_amdgcn_wave_barrier();
        // the next chunk's loads fly under this chunk's MFMAs
        have = advance();
#ifndef SS_EXP_WG_NOLOAD          // (timing experiment, wrong results: no global loads after a wave's first chunk)
        if (have) { issue_a(); issue_in(nod * S + kd - 1, noh * S + kh - 1, nw0); }
#endif
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            uint4 bfr[3][3];                                // [kw][term]
            if (S == 1) {
                const int off = frag_base + 16 * (1 + 2 * s + half);
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const uint4 x = rd128(t * C::PLANE + off);
                    const unsigned pl = rd32(t * C::PLANE + off - 4), nf = rd32(t * C::PLANE + off + 16);
                    const unsigned a0 = __builtin_amdgcn_alignbit(x.x, pl, 16), a1 = __builtin_amdgcn_alignbit(x.y, x.x, 16);
                    const unsigned a2 = __builtin_amdgcn_alignbit(x.z, x.y, 16), a3 = __builtin_amdgcn_alignbit(x.w, x.z, 16);
                    const unsigned a4 = __builtin_amdgcn_alignbit(nf, x.w, 16);
                    bfr[0][t] = make_uint4(a0, a1, a2, a3);
                    bfr[1][t] = x;
                    bfr[2][t] = make_uint4(a1, a2, a3, a4);
                }
            } else {
                const int offe = frag_base + 16 * (2 * s + half), offo = 3 * C::PLANE + frag_base + 16 * (1 + 2 * s + half);
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const uint4 xe = rd128(t * C::PLANE + offe), xo = rd128(t * C::PLANE + offo);
                    const unsigned pl = rd32(t * C::PLANE + offo - 4);
                    bfr[0][t] = make_uint4(__builtin_amdgcn_alignbit(xo.x, pl, 16), __builtin_amdgcn_alignbit(xo.y, xo.x, 16),
                                           __builtin_amdgcn_alignbit(xo.z, xo.y, 16), __builtin_amdgcn_alignbit(xo.w, xo.z, 16));
                    bfr[1][t] = xe;
                    bfr[2][t] = xo;
                }
            }
            // six cross products per tap, smallest first: (m,m) (h,l) (l,h) (h,m) (m,h) (h,h)
            constexpr int pa[6] = {1, 0, 2, 0, 1, 0}, pb[6] = {1, 2, 0, 1, 0, 0};
#ifdef SS_EXP_WG_NOMFMA           // (timing experiment, wrong results: one MFMA per tap and K-step instead of six)
#pragma unroll
            for (int p = 5; p < 6; ++p)
#else
#pragma unroll
            for (int p = 0; p < 6; ++p)
#endif
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
                    acc[kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[s][pa[p]]), __builtin_bit_cast(bf16x8, bfr[kw][pb[p]]),
                                                                      acc[kw], 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
    }
    // ---- the wave's 3 tiles -> workspace [tile][tap][register][lane] ----
    float* wt = ws + ((long long)blockIdx.y * 27 + kd * 9 + kh * 3) * 1024 + lane;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) unsafeAtomicAdd(wt + (t * 16 + r) * 64, acc[t][r]);
}

// ---- the cooperative form (stride 1): the nine waves of a workgroup are the nine kernel rows (kd, kh) and WALK DOWN a column of
// chunks: fixed (od, w0), oh = oh0 ... oh1.  The phase ablation of the per-wave kernel above (profiles/r06_h_wgrad_ablation.txt: 751
// us as is, 386 without its global loads, 709 with a sixth of its MFMAs) says what it is bound by: every wave fetching its own 4 KB
// of gout and 4.3 KB of input per 36 MFMAs, 3.6 GB through L2 for a 0.4 GB layer.  Walking down a column, step oh needs the input
// rows oh - 1, oh, oh + 1 of each depth plane: ONE new row per plane and step, staged once -- a third of it by each of the plane's
// three waves -- into a ring of four LDS tiles that all three read their fragments from; the gout chunk is staged once per step
// by the nine waves together (double-buffered) instead of nine times: 17 KB per 324 MFMAs instead of 75, and a fifth of the
// conversion work.  One barrier per step: at step t the workgroup writes row t + 1 into ring slot (t + 1) & 3 and gout chunk t into
// buffer t & 1 (loaded a step ago), meets, issues the loads of step t + 1 and multiplies -- a wave still multiplying step t - 1
// reads slots t - 2, t - 1, t and buffer (t - 1) & 1, none of which is being written.
constexpr int CO_RING = 4;
constexpr int CO_RSA = 80;                                // gout tile: [32 positions] bf16 per (term, co), 64 bytes, stride 80
constexpr int CO_TILE = WG<1>::TILE;                      // one input-row tile: 3 terms x 32 ci x 112 bytes
constexpr int CO_ATILE = 3 * 32 * CO_RSA;
constexpr int CO_LDS = 3 * CO_RING * CO_TILE + 2 * CO_ATILE;      // 144 384 bytes: one workgroup of nine waves per CU

__global__ __launch_bounds__(576, 1) void conv3d_wgrad_bf16s_coop(const float* __restrict__ gout, const float* __restrict__ in,
                                                                   float* __restrict__ ws, int Cin, int Cout, int D, int H, int W,
                                                                   int chunks_per_row, int seg_len, int nseg, int ci_tiles) {
    using C = WG<1>;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int kd = wave / 3, kh = wave - 3 * kd;
    const int col = blockIdx.x / nseg, seg = blockIdx.x - col * nseg;
    const int od = col / chunks_per_row, w0 = (col - od * chunks_per_row) * CW;
    const int oh0 = seg * seg_len, oh1 = min(oh0 + seg_len, H);           // output rows [oh0, oh1) (stride 1: Ho = H, Wo = W)
    const int co0 = (blockIdx.y / ci_tiles) * 32, ci0 = (blockIdx.y % ci_tiles) * 32;
    const int b = blockIdx.z;
    const int id = od + kd - 1;
    const bool plane_ok = (unsigned)id < (unsigned)D;                      // (wave-uniform) this wave's depth plane exists
    unsigned char* ring = lds_raw + kd * (CO_RING * CO_TILE);
    unsigned char* abuf = lds_raw + 3 * CO_RING * CO_TILE;

    const long long chan = (long long)D * H * W;
    const __amdgpu_buffer_rsrc_t gres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(gout + (long long)b * Cout * chan), 0, (int)min((long long)Cout * chan * 4, 0x7fffffffLL), 0x00020000);
    const __amdgpu_buffer_rsrc_t ires = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(in + (long long)b * Cin * chan), 0, (int)min((long long)Cin * chan * 4, 0x7fffffffLL), 0x00020000);
    const int q = lane & 15, csub = lane >> 4;             // staging: lane = (position pair q, channel 4 i + csub)
    const bool whole = w0 + CW + 1 <= W;                    // every column this column of chunks reads lies inside the rows: no masks
    // this wave's share of the staging: iterations [i0, i1) of the new input row of its plane (+ the halo: kh == 2), iteration `wave`
    // of the gout chunk (wave 8: none)
    const int i0 = 3 * kh, i1 = min(3 * kh + 3, 8);

    f32x16 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    float rin[6], rhalo = 0.f, ra[2] = {0.f, 0.f};
    auto issue = [&](int t) {                               // the loads of step t: input row t + 1 (this wave's share), gout row t
        const int ih = t + 1;
        const bool row_ok = plane_ok && (unsigned)ih < (unsigned)H && ih <= oh1;
        const unsigned row_b = (unsigned)((((long long)id * H + ih) * W) * 4);
        const unsigned dead = row_ok ? 0u : 0x80000000u;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int i = i0 + k;
            const int c = ci0 + 4 * i + csub;
            const unsigned ch = ((c < Cin && i < i1) ? (unsigned)(c * chan * 4) : 0x80000000u) | dead;
#pragma unroll
            for (int e = 0; e < 2; ++e)
                rin[k * 2 + e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ires, (int)(ch + (unsigned)((w0 + 2 * q + e) * 4)), (int)row_b, 0));
        }
        const int hw = half ? w0 + CW : w0 - 1;
        const bool hok = row_ok && kh == 2 && (ci0 + l31 < Cin) && hw >= 0 && hw < W;
        rhalo = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                              ires, (int)(hok ? (unsigned)((ci0 + l31) * chan * 4) + (unsigned)(hw * 4) : 0x80000000u), (int)row_b, 0));
        const bool a_ok = wave < 8 && t >= oh0 && t < oh1;
        const int ca = co0 + 4 * wave + csub;
        const unsigned cha = (a_ok && ca < Cout) ? (unsigned)(ca * chan * 4) : 0x80000000u;
        const unsigned arow_b = (unsigned)((((long long)od * H + max(t, 0)) * W) * 4);
#pragma unroll
        for (int e = 0; e < 2; ++e)
            ra[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(gres, (int)(cha + (unsigned)((w0 + 2 * q + e) * 4)), (int)arow_b, 0));
    };
    auto stage = [&](int t, auto whole_tag) {               // write what `issue(t)` loaded: input row t + 1 -> ring, gout row t -> buffer t & 1
        constexpr bool WHOLE = decltype(whole_tag)::value;
        const int ih = t + 1;
        if (plane_ok && (unsigned)ih < (unsigned)H && ih <= oh1) {
            unsigned char* tile = ring + ((ih + 4) & 3) * CO_TILE;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int i = i0 + k;
                if (i < i1) {                                // (wave-uniform)
                    const float v0 = (WHOLE || w0 + 2 * q < W) ? rin[k * 2] : 0.f, v1 = (WHOLE || w0 + 2 * q + 1 < W) ? rin[k * 2 + 1] : 0.f;
                    unsigned h, m, l;
                    split3_pk_w(v0, v1, h, m, l);
                    unsigned char* p = tile + (4 * i + csub) * C::RS + 16 + 4 * q;
                    *reinterpret_cast<unsigned*>(p) = h;
                    *reinterpret_cast<unsigned*>(p + C::PLANE) = m;
                    *reinterpret_cast<unsigned*>(p + 2 * C::PLANE) = l;
                }
            }
            if (kh == 2) {                                   // the halo elements of channel l31: element 7 (left) / 40 (right) of the row
                unsigned h, m, l;
                split3_pk_w(rhalo, 0.f, h, m, l);
                unsigned char* p = tile + l31 * C::RS + (half ? 80 : 14);
                *reinterpret_cast<unsigned short*>(p) = (unsigned short)h;
                *reinterpret_cast<unsigned short*>(p + C::PLANE) = (unsigned short)m;
                *reinterpret_cast<unsigned short*>(p + 2 * C::PLANE) = (unsigned short)l;
            }
        }
        if (wave < 8 && t >= oh0 && t < oh1) {
            const float v0 = (WHOLE || w0 + 2 * q < W) ? ra[0] : 0.f, v1 = (WHOLE || w0 + 2 * q + 1 < W) ? ra[1] : 0.f;
            unsigned h, m, l;
            split3_pk_w(v0, v1, h, m, l);
            unsigned char* p = abuf + (t & 1) * CO_ATILE + (4 * wave + csub) * CO_RSA + 4 * q;
            *reinterpret_cast<unsigned*>(p) = h;
            *reinterpret_cast<unsigned*>(p + 32 * CO_RSA) = m;
            *reinterpret_cast<unsigned*>(p + 64 * CO_RSA) = l;
        }
    };

    // pseudo-steps t = oh0 - 2, oh0 - 1 only stage (rows oh0 - 1, oh0); steps oh0 ... oh1 - 1 stage row t + 1 and multiply
    issue(oh0 - 2);
    for (int t = oh0 - 2; t < oh1; ++t) {
        if (whole) stage(t, std::true_type{});
        else stage(t, std::false_type{});
        __syncthreads();
        if (t + 1 < oh1) issue(t + 1);
        const int ih = t + kh - 1;
        if (t >= oh0 && plane_ok && (unsigned)ih < (unsigned)H) {          // (wave-uniform)
            const unsigned char* tile = ring + ((ih + 4) & 3) * CO_TILE;
            const unsigned char* at = abuf + (t & 1) * CO_ATILE;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                uint4 af[3], bfr[3][3];
                const int offa = l31 * CO_RSA + 16 * (2 * s + half), off = l31 * C::RS + 16 * (1 + 2 * s + half);
#pragma unroll
                for (int tm = 0; tm < 3; ++tm) {
                    af[tm] = *reinterpret_cast<const uint4*>(at + tm * 32 * CO_RSA + offa);
                    const uint4 x = *reinterpret_cast<const uint4*>(tile + tm * C::PLANE + off);
                    const unsigned pl = *reinterpret_cast<const unsigned*>(tile + tm * C::PLANE + off - 4);
                    const unsigned nf = *reinterpret_cast<const unsigned*>(tile + tm * C::PLANE + off + 16);
                    const unsigned a0 = __builtin_amdgcn_alignbit(x.x, pl, 16), a1 = __builtin_amdgcn_alignbit(x.y, x.x, 16);
                    const unsigned a2 = __builtin_amdgcn_alignbit(x.z, x.y, 16), a3 = __builtin_amdgcn_alignbit(x.w, x.z, 16);
                    const unsigned a4 = __builtin_amdgcn_alignbit(nf, x.w, 16);
                    bfr[0][tm] = make_uint4(a0, a1, a2, a3);
                    bfr[1][tm] = x;
                    bfr[2][tm] = make_uint4(a1, a2, a3, a4);
                }
                constexpr int pa[6] = {1, 0, 2, 0, 1, 0}, pb[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
                for (int p = 0; p < 6; ++p)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw)
                        acc[kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[pa[p]]), __builtin_bit_cast(bf16x8, bfr[kw][pb[p]]),
                                                                          acc[kw], 0, 0, 0);
            }
        }
    }
    float* wt = ws + ((long long)blockIdx.y * 27 + kd * 9 + kh * 3) * 1024 + lane;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
#ifdef SS_EXP_COOP_NOATOM                                   // (timing-only ablation: wrong results)
        for (int r = 0; r < 16; ++r) wt[(t * 16 + r) * 64] = acc[t][r];
#else
        for (int r = 0; r < 16; ++r) unsafeAtomicAdd(wt + (t * 16 + r) * 64, acc[t][r]);
#endif
}

// ---- stride 2 with the gout chunk SHARED (r06, last): the per-wave form above is bound by its global loads (271 us as is, 135 without
// them, 109 with neither loads nor staging: profiles/EXPERIMENTS.md F.23), and the loads that hurt are the gout fragments -- lane =
// channel, 32 cache lines per instruction, and the same chunk fetched by all nine kernel rows.  Here a workgroup is the nine kernel rows
// (kd, kh) of ONE chunk range: each wave keeps the per-wave form's private input tile (the nine rows 2 od + kd - 1, 2 oh + kh - 1 of a
// chunk are nine different rows: nothing to share), and the gout chunk is staged once by eight of the waves together -- coalesced,
// lane = (position pair, channel), as in the cooperative stride-1 kernel -- into a double-buffered LDS tile all nine read their A
// fragments from.  One barrier per chunk.  (The cooperative stride-2 form that also shared the input rows needed 16-position chunks to
// fit its ring in LDS and lost, F.12; this one keeps 32.)
constexpr int S2A_LDS = 9 * WG<2>::TILE + 2 * CO_ATILE;   // 153 600 bytes: one workgroup of nine waves per CU

__global__ __launch_bounds__(576, 1) void conv3d_wgrad_bf16s_s2a(const float* __restrict__ gout, const float* __restrict__ in,
                                                                  float* __restrict__ ws, int Cin, int Cout, int D, int H, int W, int Do,
                                                                  int Ho, int Wo, int chunks_per_row, int total_chunks, int chunks_per_unit,
                                                                  int ci_tiles) {
    using C = WG<2>;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int kd = wave / 3, kh = wave - 3 * kd;
#ifdef SS_S2A_XCD
    // consecutive chunk ranges (neighbouring depth planes: they read the same input planes) on the SAME XCD, i.e. behind one L2:
    // workgroups go round-robin over the 8 XCDs, so XCD x takes the contiguous x-th eighth of the ranges
    const int unit = (gridDim.x % 8 == 0) ? (int)(blockIdx.x % 8) * (int)(gridDim.x / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
#else
    const int unit = blockIdx.x;
#endif
    const int c_begin = unit * chunks_per_unit, c_end = min(c_begin + chunks_per_unit, total_chunks);
    const int co0 = (blockIdx.y / ci_tiles) * 32, ci0 = (blockIdx.y % ci_tiles) * 32;
    const int b = blockIdx.z;
    unsigned char* tile = lds_raw + wave * C::TILE;
    unsigned char* abuf = lds_raw + 9 * C::TILE;

    const long long ochan = (long long)Do * Ho * Wo, ichan = (long long)D * H * W;
    const __amdgpu_buffer_rsrc_t gres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(gout + (long long)b * Cout * ochan), 0, (int)min((long long)Cout * ochan * 4, 0x7fffffffLL), 0x00020000);
    const __amdgpu_buffer_rsrc_t ires = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(in + (long long)b * Cin * ichan), 0, (int)min((long long)Cin * ichan * 4, 0x7fffffffLL), 0x00020000);
    const int q = lane & 15, csub = lane >> 4;             // staging: lane = (position pair q, channel 4 i + csub)
    const int frag_base = l31 * C::RS;
    const int ca = co0 + 4 * wave + csub;                  // the gout channel this lane stages (waves 0-7)
    const unsigned a_chan = (wave < 8 && ca < Cout) ? (unsigned)(ca * ochan * 4) : 0x80000000u;

    f32x16 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    float rin[8 * C::NLD], rhalo = 0.f, ra[2] = {0.f, 0.f};
    auto issue_in = [&](int id, int ih, int w0) {
        const unsigned row_b = (unsigned)((((long long)id * H + ih) * W) * 4);
        const int iw0 = (w0 + 2 * q) * 2;                  // first input column of this lane's pair
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = ci0 + 4 * i + csub;
            const unsigned ch = (c < Cin) ? (unsigned)(c * ichan * 4) : 0x80000000u;
#pragma unroll
            for (int e = 0; e < C::NLD; ++e)
                rin[i * C::NLD + e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ires, (int)(ch + (unsigned)((iw0 + e) * 4)), (int)row_b, 0));
        }
        const int hw = 2 * w0 - 1;                          // halo: column 2 w0 - 1 of channel l31 (lanes 0-31)
        const bool hok = (ci0 + l31 < Cin) && hw >= 0 && hw < W && half == 0;
        rhalo = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                              ires, (int)(hok ? (unsigned)((ci0 + l31) * ichan * 4) + (unsigned)(hw * 4) : 0x80000000u), (int)row_b, 0));
    };
    auto store_in = [&](int w0, auto whole_tag) {
        constexpr bool WHOLE = decltype(whole_tag)::value;
        const int iw0 = (w0 + 2 * q) * 2;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float v[C::NLD];
#pragma unroll
            for (int e = 0; e < C::NLD; ++e) v[e] = (WHOLE || iw0 + e < W) ? rin[i * C::NLD + e] : 0.f;
            const int c = 4 * i + csub;
            unsigned h, m, l;
            split3_pk_w(v[0], v[2], h, m, l);              // even columns: E[j], E[j + 1]
            unsigned char* p = tile + c * C::RS + 4 * q;
            *reinterpret_cast<unsigned*>(p) = h;
            *reinterpret_cast<unsigned*>(p + C::PLANE) = m;
            *reinterpret_cast<unsigned*>(p + 2 * C::PLANE) = l;
            split3_pk_w(v[1], v[3], h, m, l);              // odd columns: O[j], O[j + 1]
            p = tile + 3 * C::PLANE + c * C::RS + 16 + 4 * q;
            *reinterpret_cast<unsigned*>(p) = h;
            *reinterpret_cast<unsigned*>(p + C::PLANE) = m;
            *reinterpret_cast<unsigned*>(p + 2 * C::PLANE) = l;
        }
        if (half == 0) {
            unsigned h, m, l;
            split3_pk_w(rhalo, 0.f, h, m, l);
            unsigned char* p = tile + 3 * C::PLANE + l31 * C::RS + 14;
            *reinterpret_cast<unsigned short*>(p) = (unsigned short)h;
            *reinterpret_cast<unsigned short*>(p + C::PLANE) = (unsigned short)m;
            *reinterpret_cast<unsigned short*>(p + 2 * C::PLANE) = (unsigned short)l;
        }
    };
    auto rd128 = [&](int off) { return *reinterpret_cast<const uint4*>(tile + off); };
    auto rd32 = [&](int off) { return *reinterpret_cast<const unsigned*>(tile + off); };
    auto where = [&](int c, int& od, int& oh, int& w0) {
        const int row = c / chunks_per_row;
        od = row / Ho; oh = row - od * Ho; w0 = (c - row * chunks_per_row) * CW;
    };
    auto issue = [&](int c) {                               // the loads of chunk c: this wave's input row (if it exists), its share of gout
        int od, oh, w0;
        where(c, od, oh, w0);
        const int id = 2 * od + kd - 1, ih = 2 * oh + kh - 1;
        if ((unsigned)id < (unsigned)D && (unsigned)ih < (unsigned)H) issue_in(id, ih, w0);
        const unsigned a_row = (unsigned)((((long long)od * Ho + oh) * Wo + w0) * 4);
#pragma unroll
        for (int e = 0; e < 2; ++e)
            ra[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(gres, (int)(a_chan + (unsigned)((2 * q + e) * 4)), (int)a_row, 0));
    };

    if (c_begin < c_end) issue(c_begin);
    int par = 0;
    for (int c = c_begin; c < c_end; ++c, par ^= 1) {
        int od, oh, w0;
        where(c, od, oh, w0);
        const bool mine = (unsigned)(2 * od + kd - 1) < (unsigned)D && (unsigned)(2 * oh + kh - 1) < (unsigned)H;      // (wave-uniform)
        const bool whole = w0 + CW <= Wo && (w0 + CW) * 2 + 1 <= W;
        if (mine) {
            if (whole) store_in(w0, std::true_type{});
            else store_in(w0, std::false_type{});
        }
        if (wave < 8) {
            const float v0 = (whole || w0 + 2 * q < Wo) ? ra[0] : 0.f, v1 = (whole || w0 + 2 * q + 1 < Wo) ? ra[1] : 0.f;
            unsigned h, m, l;
            split3_pk_w(v0, v1, h, m, l);
            unsigned char* p = abuf + par * CO_ATILE + (4 * wave + csub) * CO_RSA + 4 * q;
            *reinterpret_cast<unsigned*>(p) = h;
            *reinterpret_cast<unsigned*>(p + 32 * CO_RSA) = m;
            *reinterpret_cast<unsigned*>(p + 64 * CO_RSA) = l;
        }
        __syncthreads();
        if (c + 1 < c_end) issue(c + 1);                    // the next chunk's loads fly under this chunk's MFMAs
        if (mine) {
            const unsigned char* at = abuf + par * CO_ATILE;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                uint4 af[3], bfr[3][3];
                const int offa = l31 * CO_RSA + 16 * (2 * s + half);
                const int offe = frag_base + 16 * (2 * s + half), offo = 3 * C::PLANE + frag_base + 16 * (1 + 2 * s + half);
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    af[t] = *reinterpret_cast<const uint4*>(at + t * 32 * CO_RSA + offa);
                    const uint4 xe = rd128(t * C::PLANE + offe), xo = rd128(t * C::PLANE + offo);
                    const unsigned pl = rd32(t * C::PLANE + offo - 4);
                    bfr[0][t] = make_uint4(__builtin_amdgcn_alignbit(xo.x, pl, 16), __builtin_amdgcn_alignbit(xo.y, xo.x, 16),
                                           __builtin_amdgcn_alignbit(xo.z, xo.y, 16), __builtin_amdgcn_alignbit(xo.w, xo.z, 16));
                    bfr[1][t] = xe;
                    bfr[2][t] = xo;
                }
                constexpr int pa[6] = {1, 0, 2, 0, 1, 0}, pb[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
                for (int p = 0; p < 6; ++p)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw)
                        acc[kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[pa[p]]), __builtin_bit_cast(bf16x8, bfr[kw][pb[p]]),
                                                                          acc[kw], 0, 0, 0);
            }
        }
    }
    float* wt = ws + ((long long)blockIdx.y * 27 + kd * 9 + kh * 3) * 1024 + lane;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) unsafeAtomicAdd(wt + (t * 16 + r) * 64, acc[t][r]);
}

// ---- a single output channel (the classifiers' 32 -> 1 heads, models/SemStereo.py:228-234): dW[ci, tap] = sum_p' x[ci, p'] * g[p' - tap + 1].
// With Cout = 1 the tile forms above spend 27 accumulator tiles on one live matrix row each (the head's weight gradient cost what a
// 32 -> 32 layer's costs, 0.48 ms at [24,256,256]).  Here the 27 TAPS are the MFMA's N columns: A = x (lane = input channel, 8
// consecutive input positions, read exactly once), B[k = position][n = tap] = the gradient row the tap points at, shifted by kw --
// every lane reads its own (row, shift) from a small LDS tile of the 9 neighbouring gradient rows.  One accumulator tile per wave.
__global__ __launch_bounds__(256) void conv3d_wgrad_head_bf16s(const float* __restrict__ g, const float* __restrict__ x, float* __restrict__ dw,
                                                                int Cin, int D, int H, int W, int chunks_per_row, int total_chunks,
                                                                int chunks_per_wave) {
    __shared__ float gt_all[4][9][40];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int unit = blockIdx.x * 4 + wave;
    const int c_begin = unit * chunks_per_wave, c_end = min(c_begin + chunks_per_wave, total_chunks);
    const int ci0 = blockIdx.y * 32;
    const int b = blockIdx.z;
    float (*gt)[40] = gt_all[wave];
    const long long chan = (long long)D * H * W;
    const __amdgpu_buffer_rsrc_t gres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(g + (long long)b * chan), 0, (int)min(chan * 4, 0x7fffffffLL), 0x00020000);
    const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(x + (long long)b * Cin * chan), 0, (int)min((long long)Cin * chan * 4, 0x7fffffffLL), 0x00020000);
    const unsigned a_lane = (ci0 + l31 < Cin) ? (unsigned)((ci0 + l31) * chan * 4) + 32u * half : 0x80000000u;
    const bool tap_ok = l31 < 27;
    const int trow = tap_ok ? l31 / 3 : 0, kw = tap_ok ? l31 % 3 : 0;       // row kd * 3 + kh of the gradient tile, column shift
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float av[2][8], gv[9];
    auto issue = [&](int c) {                               // the loads of chunk c: its 9 gradient rows (lanes < 34) and this lane's 2 x 8 inputs
        const int row = c / chunks_per_row, w0 = (c - row * chunks_per_row) * CW;
        const int id = row / H, ih = row - id * H;
        const int col = w0 - 1 + lane;
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            const int od = id - r / 3 + 1, oh = ih - r % 3 + 1;
            const bool ok = lane < 34 && (unsigned)od < (unsigned)D && (unsigned)oh < (unsigned)H && (unsigned)col < (unsigned)W;
            gv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                  gres, (int)(ok ? (unsigned)((((long long)od * H + oh) * W + col) * 4) : 0x80000000u), 0, 0));
        }
        const unsigned x_row = (unsigned)((((long long)id * H + ih) * W + w0) * 4);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int e = 0; e < 8; ++e)
                av[s][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xres, (int)(a_lane + (unsigned)((16 * s + e) * 4)), (int)x_row, 0));
    };
    if (c_begin < c_end) issue(c_begin);
    for (int c = c_begin; c < c_end; ++c) {
        const int row = c / chunks_per_row, w0 = (c - row * chunks_per_row) * CW;
        // this chunk's operands out of the prefetch registers: the gradient rows -> LDS, the inputs -> three bf16 terms
        if (lane < 34) {
#pragma unroll
            for (int r = 0; r < 9; ++r) gt[r][lane] = gv[r];
        }
        uint4 af[2][3];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            unsigned h[4], m[4], l[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int w = w0 + 16 * s + 8 * half + 2 * e;
                split3_pk_w(w < W ? av[s][2 * e] : 0.f, w + 1 < W ? av[s][2 * e + 1] : 0.f, h[e], m[e], l[e]);
            }
            af[s][0] = make_uint4(h[0], h[1], h[2], h[3]); af[s][1] = make_uint4(m[0], m[1], m[2], m[3]); af[s][2] = make_uint4(l[0], l[1], l[2], l[3]);
        }
        __builtin_amdgcn_wave_barrier();
        if (c + 1 < c_end) issue(c + 1);                    // the next chunk's loads fly under this chunk's arithmetic
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            uint4 bf[3];
            unsigned h[4], m[4], l[4];
            const float* gp = &gt[trow][16 * s + 8 * half + 2 - kw];
#pragma unroll
            for (int e = 0; e < 4; ++e) split3_pk_w(tap_ok ? gp[2 * e] : 0.f, tap_ok ? gp[2 * e + 1] : 0.f, h[e], m[e], l[e]);
            bf[0] = make_uint4(h[0], h[1], h[2], h[3]); bf[1] = make_uint4(m[0], m[1], m[2], m[3]); bf[2] = make_uint4(l[0], l[1], l[2], l[3]);
            constexpr int pa[6] = {1, 0, 2, 0, 1, 0}, pb[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
            for (int p = 0; p < 6; ++p)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[s][pa[p]]), __builtin_bit_cast(bf16x8, bf[pb[p]]), acc, 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
    }
    // the four waves' tiles are summed through LDS (fixed order), then ONE wave adds the workgroup's tile to grad_w: every workgroup hits
    // the same Cin x 27 addresses, and 8 192 waves doing so one by one took 0.4 ms of a 0.03 ms kernel
    __shared__ float red[4][16][64];
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
    __syncthreads();
    // D layout: register r of lane (l31, half) = row (r & 3) + 8 (r >> 2) + 4 half (input channel), column l31 (tap)
    if (wave == 0 && tap_ok) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = ci0 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const float v = (red[0][r][lane] + red[1][r][lane]) + (red[2][r][lane] + red[3][r][lane]);
            if (ci < Cin) unsafeAtomicAdd(dw + (long long)ci * 27 + l31, v);
        }
    }
}

// workspace [tile][tap][register][lane] -> dW [Cout][Cin][27]: register r of lane (l31, half) = row (r & 3) + 8 (r >> 2) + 4 half
// (output channel), column l31 (input channel)
__global__ void wgrad_reorder_kernel(const float* __restrict__ ws, float* __restrict__ dw, int Cout, int Cin, int ci_tiles, long long total) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int tap = (int)(i % 27);
    const long long r_ = i / 27;
    const int ci = (int)(r_ % Cin), co = (int)(r_ / Cin);
    const int tile = (co / 32) * ci_tiles + ci / 32, row = co % 32;
    const int half = (row >> 2) & 1, r = (row & 3) + 4 * (row >> 3);
    dw[i] = ws[(((long long)tile * 27 + tap) * 16 + r) * 64 + half * 32 + (ci % 32)];
}

}  // namespace

extern "C" int ss_conv3d_wgrad_bf16s_fwd(const float* grad_out, const float* in, float* grad_w, float* workspace, int B, int Cin, int D,
                                         int H, int W, int Cout, int stride, ss_stream_t stream) {
    SS_REQUIRE(grad_out && in && grad_w && workspace);
    SS_REQUIRE(B > 0 && Cin > 0 && D > 0 && H > 0 && W > 0 && Cout > 0 && (stride == 1 || stride == 2));
    const int Do = (D - 1) / stride + 1, Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    if ((long long)Cin * D * H * W * 4 >= 0x7fffffffLL || (long long)Cout * Do * Ho * Wo * 4 >= 0x7fffffffLL || B > 65535)
        return SS_ERR_UNSUPPORTED;                                          // 32-bit buffer offsets per batch element
    hipStream_t st = ss::as_stream(stream);
    const int ci_tiles = ss::ceil_div(Cin, 32), tiles = ci_tiles * ss::ceil_div(Cout, 32);
    if (tiles > 65535) return SS_ERR_UNSUPPORTED;
    if (Cout == 1 && stride == 1 && ss::tuning().wgrad_coop != 0) {
        // a single output channel: the taps as the matrix columns (conv3d_wgrad_head_bf16s), straight into grad_w
        if (hipMemsetAsync(grad_w, 0, (size_t)Cin * 27 * sizeof(float), st) != hipSuccess) return SS_ERR_LAUNCH;
        const int cpr = ss::ceil_div(W, CW);
        const long long total_ll = (long long)D * H * cpr;
        if (total_ll > 0x7fffffffLL) return SS_ERR_UNSUPPORTED;
        const int total = (int)total_ll;
        const int waves = std::max(4, 2048 / (ci_tiles * B));
        const int per_wave = std::max(1, ss::ceil_div(total, waves));
        const int nwaves = ss::ceil_div(total, per_wave);
        hipLaunchKernelGGL(conv3d_wgrad_head_bf16s, dim3(ss::ceil_div(nwaves, 4), ci_tiles, B), dim3(256), 0, st, grad_out, in, grad_w, Cin, D, H, W,
                           cpr, total, per_wave);
        return ss::check_launch();
    }
    if (hipMemsetAsync(workspace, 0, (size_t)tiles * 27 * 1024 * sizeof(float), st) != hipSuccess) return SS_ERR_LAUNCH;
    const int chunks_per_row = ss::ceil_div(Wo, CW);
    const long long total_ll = (long long)Do * Ho * chunks_per_row;
    if (total_ll > 0x7fffffffLL) return SS_ERR_UNSUPPORTED;
    const int total = (int)total_ll;
    // ~8 waves per CU on 256 CUs; a unit (= wave) is (kernel-depth plane, tile pair, batch element, chunk range)
    int nsplit = std::max(1, 3072 / (9 * tiles * B));             // ~12 waves per CU on 256 CUs
    int per_unit = std::max(chunks_per_row >= 4 ? 4 : 1, ss::ceil_div(total, nsplit));
    nsplit = ss::ceil_div(total, per_unit);
    const dim3 grid(ss::ceil_div(nsplit, 8) * 24, tiles, B);          // (split, kd) -> block index ((split / 8) * 3 + kd) * 8 + split % 8
    if (stride == 1 && ss::tuning().wgrad_coop != 0) {
        // the cooperative form: a workgroup walks down a column (od, w0) of chunks, the columns cut into segments so that ~3 workgroups
        // per CU exist
        auto kern = conv3d_wgrad_bf16s_coop;
        if (ss::ensure_dynamic_lds(reinterpret_cast<const void*>(kern), CO_LDS) != SS_OK) return SS_ERR_LAUNCH;
        const long long cols = (long long)Do * chunks_per_row;
#ifndef SS_COOP_WGS
#define SS_COOP_WGS 768
#endif
        int nseg_c = (int)std::min<long long>(std::max<long long>(1, SS_COOP_WGS / std::max<long long>(1, cols * tiles * B)), std::max(1, Ho / 8));
        const int seg_len = ss::ceil_div(Ho, nseg_c);
        nseg_c = ss::ceil_div(Ho, seg_len);
        if (cols * nseg_c > 0x7fffffffLL) return SS_ERR_UNSUPPORTED;
        hipLaunchKernelGGL(kern, dim3((unsigned)(cols * nseg_c), tiles, B), dim3(576), CO_LDS, st, grad_out, in, workspace, Cin, Cout, D, H, W,
                           chunks_per_row, seg_len, nseg_c, ci_tiles);
    } else if (stride == 1) {
        auto kern = conv3d_wgrad_bf16s<1>;
        hipLaunchKernelGGL(kern, grid, dim3(192), 3 * WG<1>::TILE, st, grad_out, in, workspace, Cin, Cout, D, H, W, Do, Ho, Wo, chunks_per_row,
                           total, per_unit, nsplit, ci_tiles);
    } else if (ss::tuning().wgrad_coop != 0) {
        // stride 2 with the gout chunk shared: a workgroup = the nine kernel rows of a chunk range; ONE workgroup per CU's worth of ranges (each
        // ends with 27 648 atomics into the workspace: 768 ranges 234 us, 256 ranges 203, 3072 ranges 368 on hourglass.conv1's layer)
        auto kern = conv3d_wgrad_bf16s_s2a;
        if (ss::ensure_dynamic_lds(reinterpret_cast<const void*>(kern), S2A_LDS) != SS_OK) return SS_ERR_LAUNCH;
#ifndef SS_S2A_WGS
#define SS_S2A_WGS 256
#endif
        int ns = std::max(1, SS_S2A_WGS / (tiles * B));
        int pu = std::max(chunks_per_row >= 4 ? 4 : 1, ss::ceil_div(total, ns));
        ns = ss::ceil_div(total, pu);
        hipLaunchKernelGGL(kern, dim3(ns, tiles, B), dim3(576), S2A_LDS, st, grad_out, in, workspace, Cin, Cout, D, H, W, Do, Ho, Wo, chunks_per_row,
                           total, pu, ci_tiles);
    } else {
        auto kern = conv3d_wgrad_bf16s<2>;
        if (ss::ensure_dynamic_lds(reinterpret_cast<const void*>(kern), 3 * WG<2>::TILE) != SS_OK) return SS_ERR_LAUNCH;
        hipLaunchKernelGGL(kern, grid, dim3(192), 3 * WG<2>::TILE, st, grad_out, in, workspace, Cin, Cout, D, H, W, Do, Ho, Wo, chunks_per_row,
                           total, per_unit, nsplit, ci_tiles);
    }
    if (ss::check_launch() != SS_OK) return SS_ERR_LAUNCH;
    const long long n = (long long)Cout * Cin * 27;
    hipLaunchKernelGGL(wgrad_reorder_kernel, dim3((unsigned)ss::ceil_div_ll(n, 256)), dim3(256), 0, st, workspace, grad_w, Cout, Cin, ci_tiles, n);
    return ss::check_launch();
}
