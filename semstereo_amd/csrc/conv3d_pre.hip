// 3x3x3 stride-1 Conv3d (+ partial sum, folded BatchNorm, ReLU, channelAtt gate) whose input arrives PRE-SPLIT: the producer
// (warp.hip: the warped half of the sparse concat volume, models/SemStereo.py:316-318) has already written every value as
// the two fp16 terms of the matrix-core form of conv3d_bf16s.hip (x * 2^(E_ONE - e) = hi + lo, e one block exponent per
// batch element), 8 channels x 2 terms = 32 bytes per position, in exactly the 16-byte slots this kernel's LDS tile holds:
//
//     xs [B][Cin/8][2 terms][D][H][W][8 channels] fp16        xexp [B] int: the biased exponent e
// (term-major: a DMA instruction's 64 lanes then read 64 consecutive 16-byte slots of a row, whole cache lines)
//
// The operand is then staged by LDS-DMA loads (buffer_load_dwordx4 ... lds): no prefetch registers, no conversion, no
// per-chunk maximum, no accumulator rescale -- the K loop is MFMAs, LDS fragment reads, the weight-fragment ring and one DMA
// instruction per step that brings the NEXT chunk into the other half of a double-buffered tile (2 x 2 terms x 1224 slots =
// 78 KB: two workgroups per CU).  Same GEMM mapping, weight packing (ss_pack_conv3d_weights_f16s), epilogue and result
// contract as conv3d_bf16s<1, 4, 4, 4, F16X3, GATED> -- whose per-chunk VALU work (32 instructions per staged position +
// the running maximum: ~180 of ~480 per chunk and wave beside 168 MFMAs, ISA count r03) is what this form removes.
#include <algorithm>
#include <stdlib.h>
#include <type_traits>

#include "common.h"
#include "split_f16.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

constexpr int P_TD = 4, P_TH = 4, P_NT = 4;                  // 4 planes x 4 rows x 32 columns per workgroup; wave = plane
constexpr int P_ID = 6, P_IH = 6, P_IW = 34, P_CS = P_ID * P_IH * P_IW;       // halo tile: 1224 positions
constexpr int P_NPOS = (P_CS + 255) / 256;                   // DMA instructions per term and thread
constexpr int P_KSTEPS = 14, P_AP = 2, P_AR = P_AP + 1;
constexpr int P_ZSLOT = 4 * P_CS;                            // [buffer][term][position] slots, then the all-zero slot, then the affine
constexpr size_t P_LDS_BYTES = (size_t)(4 * P_CS + 2 + 48) * 16;
static_assert(P_LDS_BYTES <= 80 * 1024, "two workgroups per CU");

template <bool GATED>
__global__ __launch_bounds__(256, 2) void conv3d_pre(const uint4* __restrict__ xs, const int* __restrict__ xexp,
                                                      const uint4* __restrict__ wsplit, const float* __restrict__ scale,
                                                      const float* __restrict__ shift, const float* __restrict__ residual,
                                                      const float* __restrict__ gate, float* __restrict__ out, int Cin, int D, int H,
                                                      int W, int Cout, int tiles_w, int tiles_h, int ntiles, int relu) {
    extern __shared__ __attribute__((aligned(16))) uint4 lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int co0 = blockIdx.y * 32, b = blockIdx.z;
    const int lane_pos = wave * P_IH * P_IW + l31;             // slot of this lane's first row, tap (0,0,0)
    auto tile_origin = [&](int tile, int& ow0, int& oh0, int& od0) {
        int t = tile;
        const int tw = t % tiles_w; t /= tiles_w;
        const int th = t % tiles_h; t /= tiles_h;
        ow0 = tw * 32; oh0 = th * P_TH; od0 = t * P_TD;
    };
    const bool res_pre = (relu & 2) != 0 && residual != nullptr;      // `residual`: a partial sum of the same convolution
    const int nchunks = Cin / 8;
    const float* wunscale = reinterpret_cast<const float*>(reinterpret_cast<const char*>(wsplit) + (size_t)nchunks * P_KSTEPS * (2 * 2 * Cout * 16));
    const size_t plane = (size_t)H * W;
    const unsigned ochan_b = (unsigned)((size_t)D * plane * 4), gchan_b = (unsigned)(plane * 4);
    const int obytes = (int)min((long long)Cout * (long long)ochan_b, 0x7fffffffLL);
    unsigned vout[P_NT], vgate[P_NT];
    auto set_outputs = [&](int tile) {
        int ow0, oh0, od0;
        tile_origin(tile, ow0, oh0, od0);
        const int ow_ = ow0 + l31, od_ = od0 + wave;
#pragma unroll
        for (int i = 0; i < P_NT; ++i) {
            const int oh_ = oh0 + i;
            const bool ok = ow_ < W && od_ < D && oh_ < H;
            vout[i] = ok ? (unsigned)((((size_t)od_ * H + oh_) * W + ow_) * 4) + 4u * half * ochan_b : 0x80000000u;
            vgate[i] = ok ? (unsigned)(((size_t)oh_ * W + ow_) * 4) + 4u * half * gchan_b : 0x80000000u;
        }
    };
    auto cbase = [&](int r) { return co0 + (r & 3) + 8 * (r >> 2); };      // this lane's channel of fragment register r: + 4 * half
    f32x16 acc[P_NT];

    // staging plan: DMA instruction i of a term covers the 64 positions [p0, p0 + 64) of the halo tile, one per lane; the last
    // group of the last wave is shifted back so that it ENDS at the tile's last position (re-loading 56 positions of its
    // neighbour: no lane ever writes past a term's slots, no EXEC masking, no padding between the terms)
    const bool tail = (wave == 3);
    auto slot0 = [&](int i) { return (i == P_NPOS - 1 && tail) ? P_CS - 64 : 256 * i + 64 * wave; };
    auto make_poff = [&](int tile, unsigned (&po)[P_NPOS]) {
        int ow0, oh0, od0;
        tile_origin(tile, ow0, oh0, od0);
#pragma unroll
        for (int i = 0; i < P_NPOS; ++i) {
            const int p = slot0(i) + lane;
            const int wx = p % P_IW;
            const int r = p / P_IW;
            const int gw = ow0 - 1 + wx, gh = oh0 - 1 + r % P_IH, gd = od0 - 1 + r / P_IH;
            const bool ok = (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W;
            po[i] = ok ? (unsigned)(((size_t)gd * plane + (size_t)gh * W + gw) * 16) : 0x80000000u;     // outside: zeros
        }
    };
    unsigned poff[P_NPOS];
    make_poff(blockIdx.x, poff);
    const long long chunk_bytes = (long long)D * (long long)plane * 32;
    const int term_b = (int)(chunk_bytes / 2);                  // bytes of one term of one chunk
    const __amdgpu_buffer_rsrc_t ires = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint4*>(xs) + (size_t)b * nchunks * D * plane * 2, 0, (int)min((long long)nchunks * chunk_bytes, 0x7fffffffLL), 0x00020000);
    // DMA instruction k = 2 * i + term of chunk `q` into buffer `buf` (`dead`: nothing follows -- an offset beyond the buffer)
    auto dma = [&](int k, int buf, int q, unsigned dead) {
        const int i = k >> 1, term = k & 1;
        lds_dma16(ires, &lds[(buf * 2 + term) * P_CS + slot0(i)], (int)(poff[i] | dead), (2 * q + term) * term_b);
    };
    if (tid == 0) lds[P_ZSLOT] = make_uint4(0u, 0u, 0u, 0u);
    float* aff = reinterpret_cast<float*>(&lds[P_ZSLOT + 2]);   // scale, shift, 2^-(weight scale) of the workgroup's 32 channels
    if (tid < 32) {
        const int co = min(co0 + tid, Cout - 1);
        aff[tid] = scale ? scale[co] : 1.0f;
        aff[64 + tid] = shift ? shift[co] : 0.0f;
        aff[128 + tid] = wunscale[co];
    }
    // weight fragments: the packed two-term layout of conv3d_bf16s.hip, streamed from L2 two K-steps ahead
    const int wlane = (half * Cout + min(co0 + l31, Cout - 1)) * 16;
    const int wstep = 2 * 2 * Cout * 16;
    const int G = nchunks * P_KSTEPS;
    const __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint4*>(wsplit), 0, (int)min((long long)G * wstep, 0x7fffffffLL), 0x00020000);
    auto load_a = [&](int g, int c) {
        return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(wres, wlane, g * wstep + c * 2 * Cout * 16, 0));
    };
    uint4 aq[P_AR][2];
#pragma unroll
    for (int k = 0; k < P_AP; ++k)
#pragma unroll
        for (int c = 0; c < 2; ++c) aq[k][c] = load_a(min(k, G - 1), c);
    // the first chunk of the first tile: the only exposed round trip of the workgroup's life
#pragma unroll
    for (int k = 0; k < 2 * P_NPOS; ++k) dma(k, 0, 0, 0u);
    int cur = 0;
    // the accumulators' scale is the input's block exponent (one per batch element) times the per-channel weight scale
    const int e_in = xexp[b];
    const float acc_unscale = __uint_as_float((unsigned)(127 - E_ONE + e_in) << 23);
    __builtin_amdgcn_s_waitcnt(0x0F70);                          // vmcnt(0)
    __syncthreads();

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const bool has_next = tile + (int)gridDim.x < ntiles;
#pragma unroll
    for (int i = 0; i < P_NT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int q = 0, g0 = 0; q < nchunks; ++q, g0 += P_KSTEPS) {
        const bool more = q + 1 < nchunks;
        if (!more) make_poff(tile + (int)gridDim.x, poff);        // the next tile's first chunk (index arithmetic under a uniform branch)
        const int q_next = more ? q + 1 : 0;
        const unsigned dead = (more || has_next) ? 0u : 0x80000000u;
        const int nxt = cur ^ 1;
        uint4 bcur[2], bnxt[2];
        auto read_b = [&](uint4 (&dst)[2], int s, int i) {
            const int ta = 2 * s, tb = 2 * s + 1;
            const int offa = ((ta / 9) * P_IH + (ta / 3) % 3) * P_IW + ta % 3;
            const int offb = (tb < 27) ? ((tb / 9) * P_IH + (tb / 3) % 3) * P_IW + tb % 3 : 0;
            const int slot = lane_pos + i * P_IW + (half ? offb : offa);
#pragma unroll
            for (int c = 0; c < 2; ++c) dst[c] = lds[(tb >= 27 && half) ? P_ZSLOT : (cur * 2 + c) * P_CS + slot];
        };
        read_b(bcur, 0, 0);
#pragma unroll
        for (int s = 0; s < P_KSTEPS; ++s) {
            // (no vector-memory instruction of the K loop sits under a branch: see conv3d_bf16s.hip)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int gw_ = g0 + s + P_AP;
                aq[(s + P_AP) % P_AR][c] = load_a(gw_ < G ? gw_ : gw_ - G, c);
            }
            if (s < 2 * P_NPOS) dma(s, nxt, q_next, dead);         // (compile-time condition: the loop is unrolled)
            uint4 a[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) a[c] = aq[s % P_AR][c];
#pragma unroll
            for (int i0 = 0; i0 < P_NT; ++i0) {
                if (i0 + 1 < P_NT) read_b(bnxt, s, i0 + 1);
                else if (s + 1 < P_KSTEPS) read_b(bnxt, s + 1, 0);
                // cross terms (weight term, operand term) in the order of conv3d_bf16s.hip: hi*lo, lo*hi, hi*hi
                acc[i0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[0]), __builtin_bit_cast(f16x8, bcur[1]), acc[i0], 0, 0, 0);
                acc[i0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[1]), __builtin_bit_cast(f16x8, bcur[0]), acc[i0], 0, 0, 0);
                acc[i0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[0]), __builtin_bit_cast(f16x8, bcur[0]), acc[i0], 0, 0, 0);
                bcur[0] = bnxt[0]; bcur[1] = bnxt[1];
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);      // the next row's fragment reads BEFORE this row's MFMAs
                __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        {   // steps 14, 15 of this chunk are steps 0, 1 of the next: re-base the fragment ring
            uint4 tq[P_AP][2];
#pragma unroll
            for (int k = 0; k < P_AP; ++k)
#pragma unroll
                for (int c = 0; c < 2; ++c) tq[k][c] = aq[(P_KSTEPS + k) % P_AR][c];
#pragma unroll
            for (int k = 0; k < P_AP; ++k)
#pragma unroll
                for (int c = 0; c < 2; ++c) aq[k][c] = tq[k][c];
        }
        // the next chunk has landed in the other buffer: its DMA loads were issued in steps 0 .. 9, and vmcnt retires in order, so
        // "at most the 8 weight-fragment loads of steps 10 .. 13 outstanding" is enough -- vmcnt(0) would also wait for those
        // (an exposed L2 round trip per chunk: 315 vs 296 us for the on-the-fly form, r03_d)
        static_assert(2 * P_NPOS == 10 && P_KSTEPS == 14, "vmcnt(8): 4 steps x 2 fragment loads after the last DMA");
        __builtin_amdgcn_s_waitcnt(0x0F78);                      // vmcnt(8)
        __syncthreads();
        cur = nxt;
    }

    // ---- epilogue (as conv3d_bf16s.hip: 32x32 D layout, column = lane & 31, row = channel) ----
    set_outputs(tile);
    const float* obase = out + (size_t)b * Cout * D * plane;
    const __amdgpu_buffer_rsrc_t ores = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(obase), 0, obytes, 0x00020000);
    const bool res_epi = residual != nullptr && !res_pre;
    const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(residual ? residual + (size_t)b * Cout * D * plane : obase), 0, residual ? obytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t gres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(GATED ? gate + (size_t)b * Cout * plane : obase), 0,
        GATED ? (int)min((long long)Cout * (long long)gchan_b, 0x7fffffffLL) : 0, 0x00020000);
    const float floor_v = (relu & 1) ? 0.f : -__builtin_inff();
    auto epilogue = [&](auto all_channels) {
    constexpr bool ALLC = decltype(all_channels)::value;
    constexpr int EG = 4;
#pragma unroll
    for (int r0 = 0; r0 < 16; r0 += EG) {
        float sc[EG], sh[EG], un[EG], gv[EG][P_NT], rv[EG][P_NT];
#pragma unroll
        for (int k = 0; k < EG; ++k) {
            const int cb = cbase(r0 + k);
            const bool cok = ALLC || cb + 4 * half < Cout;
            sc[k] = aff[cb - co0 + 4 * half];
            sh[k] = aff[64 + cb - co0 + 4 * half];
            un[k] = aff[128 + cb - co0 + 4 * half] * acc_unscale;          // powers of two: exact
#pragma unroll
            for (int i = 0; i < P_NT; ++i)
                if (GATED) gv[k][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                                      gres, (int)(cok ? vgate[i] : 0x80000000u), cb * (int)gchan_b, 0));
        }
        if (residual != nullptr) {
#pragma unroll
            for (int k = 0; k < EG; ++k) {
                const int cb = cbase(r0 + k);
                const bool cok = ALLC || cb + 4 * half < Cout;
#pragma unroll
                for (int i = 0; i < P_NT; ++i)
                    rv[k][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                             rres, (int)(cok ? vout[i] : 0x80000000u), cb * (int)ochan_b, 0));
            }
        } else {
#pragma unroll
            for (int k = 0; k < EG; ++k)
#pragma unroll
                for (int i = 0; i < P_NT; ++i) rv[k][i] = 0.f;
        }
#pragma unroll
        for (int k = 0; k < EG; ++k) {
            const int cb = cbase(r0 + k);
            const bool cok = ALLC || cb + 4 * half < Cout;
#pragma unroll
            for (int i = 0; i < P_NT; ++i) {
                float a0 = acc[i][r0 + k] * un[k];
                if (res_pre) a0 = ss::add_rn(a0, rv[k][i]);
                float v = ss::add_rn(ss::mul_rn(a0, sc[k]), sh[k]);
                if (res_epi) v = ss::add_rn(v, rv[k][i]);
                v = fmaxf(v, floor_v);
                if (GATED) v = ss::mul_rn(gv[k][i], v);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ores, (int)(cok ? vout[i] : 0x80000000u),
                                                      cb * (int)ochan_b, 0);
            }
        }
    }
    };
    if (co0 + 32 <= Cout) epilogue(std::true_type{});
    else epilogue(std::false_type{});
    }
}

template <bool GATED>
int launch_pre(const void* xs, const int* xexp, const void* wsplit, const float* scale, const float* shift, const float* residual,
               const float* gate, float* out, int B, int Cin, int D, int H, int W, int Cout, int relu, hipStream_t st) {
    const int tiles_w = ss::ceil_div(W, 32), tiles_h = ss::ceil_div(H, P_TH), tiles_d = ss::ceil_div(D, P_TD);
    const long long nt = (long long)tiles_w * tiles_h * tiles_d;
    if (nt > 0x7fffffffLL || B > 65535) return SS_ERR_UNSUPPORTED;
    auto kern = conv3d_pre<GATED>;
    if (ss::ensure_dynamic_lds(reinterpret_cast<const void*>(kern), (int)P_LDS_BYTES) != SS_OK) return SS_ERR_LAUNCH;
    const int groups = ss::ceil_div(Cout, 32) * B;
    const long long cap = std::max<long long>(1, ss::resident_workgroups(reinterpret_cast<const void*>(kern), 256, (int)P_LDS_BYTES) / groups);
    const long long rounds = ss::ceil_div_ll(nt, cap);
    const long long gx = ss::ceil_div_ll(nt, rounds);
    dim3 grid((unsigned)gx, ss::ceil_div(Cout, 32), B);
    hipLaunchKernelGGL(kern, grid, dim3(256), P_LDS_BYTES, st, reinterpret_cast<const uint4*>(xs), xexp, reinterpret_cast<const uint4*>(wsplit),
                       scale, shift, residual, gate, out, Cin, D, H, W, Cout, tiles_w, tiles_h, (int)nt, relu);
    return ss::check_launch();
}

}  // namespace

// Conv3d(k3, s1, p1, bias=False) over a PRE-SPLIT input (see the top of this file; producer: ss_concat_sampled_presplit_fwd)
// [+ partial sum before the affine] + per-channel affine + optional ReLU + optional channelAtt gate (its sigmoid, [B,Cout,H,W]).
// wsplit: ss_pack_conv3d_weights_f16s.  Replaces models/SemStereo.py:319-320 on the warped half of the volume.
extern "C" int ss_conv3d_presplit_fwd(const void* xs, const int* xexp, const void* wsplit, const float* partial, const float* scale,
                                      const float* shift, const float* gate, float* out, int B, int Cin, int D, int H, int W,
                                      int Cout, int relu, ss_stream_t stream) {
    SS_REQUIRE(xs && xexp && wsplit && out);
    SS_REQUIRE(B > 0 && Cin > 0 && Cin % 8 == 0 && D > 0 && H > 0 && W > 0 && Cout > 0);
    SS_REQUIRE(((reinterpret_cast<uintptr_t>(wsplit) | reinterpret_cast<uintptr_t>(xs)) & 15) == 0);
    if ((long long)Cin * D * H * W * 4 >= 0x7fffffffLL || (long long)Cout * D * H * W * 4 >= 0x7fffffffLL) return SS_ERR_UNSUPPORTED;
    const int r = (relu ? 1 : 0) | (partial ? 2 : 0);
    hipStream_t st = ss::as_stream(stream);
    if (gate != nullptr) return launch_pre<true>(xs, xexp, wsplit, scale, shift, partial, gate, out, B, Cin, D, H, W, Cout, r, st);
    return launch_pre<false>(xs, xexp, wsplit, scale, shift, partial, gate, out, B, Cin, D, H, W, Cout, r, st);
}
