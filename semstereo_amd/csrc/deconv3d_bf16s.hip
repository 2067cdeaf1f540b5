// ConvTranspose3d(k=3, s=2, p=1, output_padding=1) + folded BatchNorm + 1x1x1 skip projection + ReLU of
// hourglass.forward (reference models/SemStereo.py:124-130, 141-142 / 163-169, 180-181) on the bf16 matrix core
// with split-bf16 fp32 emulation: the parity-class formulation of deconv3d.hip (every one of the 27 taps feeds
// exactly one of the 8 output parity classes at a fixed input offset, an INPUT-space tile, 8 accumulators per
// wave) with the operand machinery of conv3d_bf16s.hip (x = hi + mid + lo bf16 terms, 6 cross products on
// v_mfma_f32_32x32x16_bf16, fp32 accumulation; error below the exact-fp32 MFMA's).
//
// A K-step is ONE tap x 16 input channels (lanes 0-31: channels 0-7, lanes 32-63: channels 8-15), so a chunk of 16
// channels is 27 K-steps; the taps are visited grouped by their input offset (8 groups), so the activation fragment
// of an offset is read from LDS once per chunk and feeds up to 8 taps, whose MFMAs go to 8 different accumulators.
// LDS image: [term][channel half][position] 16-byte slots of the (TD+1) x (TH+1) x 33 input halo tile.  Weight
// fragments stream from L2 two K-steps ahead (buffer loads), the next chunk's activations are register-prefetched in
// slices over the K-steps.  The skip projection (redir) is two more K-steps per 16 skip channels with the operand
// read straight from global memory (each lane's 2x2x2 output cube as four float2 per channel) and split in registers.
#include <algorithm>
#include <type_traits>

#include "common.h"
#include "split_f16.h"


namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) short;
using bf16x2_t = __attribute__((ext_vector_type(2))) __bf16;
using ss_u32x2 = __attribute__((ext_vector_type(2))) unsigned;

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

// taps (kd*9 + kh*3 + kw) grouped by input offset o = (kd==0)*4 + (kh==0)*2 + (kw==0); class = (kd!=1)*4 + (kh!=1)*2 + (kw!=1)
__host__ __device__ constexpr int tap_of(int s) {
    constexpr int T[27] = {13, 14, 16, 17, 22, 23, 25, 26, 12, 15, 21, 24, 10, 11, 19, 20, 9, 18, 4, 5, 7, 8, 3, 6, 1, 2, 0};
    return T[s];
}
__host__ __device__ constexpr int cls_of(int s) {
    constexpr int T[27] = {0, 1, 2, 3, 4, 5, 6, 7, 1, 3, 5, 7, 2, 3, 6, 7, 3, 7, 4, 5, 6, 7, 5, 7, 6, 7, 7};
    return T[s];
}
__host__ __device__ constexpr int off_of(int s) {
    constexpr int T[27] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 4, 4, 4, 4, 5, 5, 6, 6, 7};
    return T[s];
}

constexpr int KST = 27;           // K-steps per 16-channel chunk
// STREAM (template argument of the kernel): the output goes out with NONTEMPORAL stores (buffer aux 2).  Measured r05 on
// hourglass2.conv6 (201 MB written, alone): 128 -> 108 us; sc0 + nt the same; nontemporal loads of the skip tensor 119, both 124
// (profiles/r05_f_deconv_forms.txt).  The launcher streams outputs of >= 192 MB per launch (the rule of the gwc volume kernel:
// smaller outputs are handed to the next kernel by the 256 MB Infinity Cache); SS_DECONV_STREAM=0/1 forces it.
#ifndef SS_DECONV_XCD
#define SS_DECONV_XCD 0           // 1: every XCD walks a contiguous eighth of the units (see the kernel's last lines)
#endif
#ifndef SS_DECONV_SKIP_AUX
#define SS_DECONV_SKIP_AUX 0      // ... of the skip tensor's loads
#endif

// ---- parity-class groups (r05) ----
// GRP < 0: a workgroup computes all 8 parity classes of its tile (8 accumulators = 128 registers per wave: two waves per SIMD).
// GRP = 0 / 1: the classes are split over TWO workgroups per tile -- {0,1,6,7} (15 taps) and {2,3,4,5} (12 taps): each group's
// output cube I/O is two whole (plane, row) pairs of float2 -- so a wave keeps 4 accumulators, the chunk's weight slab in LDS
// is the group's taps only (30 / 24 KB instead of 55), and a CU holds three workgroups instead of two; the freed registers
// also hold the second accumulator set of the chunk-blocked summation (ACCB, see conv3d_bf16s.hip).  Both workgroups stage
// the same input tile (it is read twice from L2: 50 MB of the layer's 453 MB), everything else is disjoint.
struct GrpTable { int ns; int step[KST]; int next_off[KST]; };
constexpr bool grp_has(int grp, int cls) { return grp < 0 || (grp == 0 ? (cls < 2 || cls >= 6) : (cls >= 2 && cls < 6)); }
constexpr GrpTable make_grp_table(int grp) {
    GrpTable t{};
    for (int s = 0; s < KST; ++s)
        if (grp_has(grp, cls_of(s))) t.step[t.ns++] = s;
    for (int k = t.ns; k < KST; ++k) t.step[k] = KST - 1;
    // the input offset of the next RUN of equal offsets after position k of the group's step list (-1: none)
    for (int k = 0; k < KST; ++k) {
        t.next_off[k] = -1;
        for (int j = k + 1; j < t.ns; ++j)
            if (off_of(t.step[j]) != off_of(t.step[k])) { t.next_off[k] = off_of(t.step[j]); break; }
    }
    return t;
}
template <int GRP>
struct Grp {
    static constexpr int NA = GRP < 0 ? 8 : 4;                  // accumulators per wave
    static constexpr int NQ2 = GRP < 0 ? 4 : 2;                 // (pd, ph) rows of the output cube
    static constexpr int idx(int cls) { return GRP < 0 ? cls : (GRP == 0 ? (cls < 2 ? cls : cls - 4) : cls - 2); }
    static constexpr int q(int j) { return GRP < 0 ? j : (GRP == 0 ? (j == 0 ? 0 : 3) : (j == 0 ? 1 : 2)); }      // pd * 2 + ph of row j
    static constexpr GrpTable T = make_grp_table(GRP);
    static constexpr int NS = T.ns;                             // 27 / 15 / 12
};
// the groups' K-steps as a run-time table (the LDS-DMA of the weight slab picks its steps by a wave-dependent index)
struct GroupSteps { int s[2][16]; };
constexpr GroupSteps make_group_steps() {
    GroupSteps t{};
    for (int k = 0; k < 16; ++k) { t.s[0][k] = Grp<0>::T.step[k]; t.s[1][k] = Grp<1>::T.step[k]; }
    return t;
}
__device__ const GroupSteps kGroupSteps = make_group_steps();

template <int TD, int TH, int LT = 3, bool WLDS = false, int WSTEPS = 27>      // LT: operand terms kept in LDS; WLDS: + WSTEPS K-steps of a chunk's weights
struct DB {
    static constexpr int ID = TD + 1, IH = TH + 1, IW = 33;
    static constexpr int CS = ID * IH * IW;                    // positions of the halo tile
    static constexpr int NPOS = (CS + 255) / 256;              // positions per thread
    // + the four waves' maxima (fp16 form) + the per-channel epilogue constants of the workgroup's 32 channels
    static constexpr int WSLOTS = WLDS ? WSTEPS * 2 * 64 : 0;  // [K-steps][2 terms][2 halves x 32 channels] 16-byte slots
    static constexpr size_t LDS_BYTES = (size_t)(LT * 2 * CS + 1 + 16 + WSLOTS) * 16;
    static_assert(TD * TH == 4, "4 waves x one input row each");
};

// SPLIT: two workgroups per tile, one per parity-class group (blockIdx.x = 2 * tile + group; fp16 form only); ACCB: chunk-blocked
// accumulation (fp16 form, SPLIT only: it lives in the registers the split frees)
template <int TD, int TH, int NTERMS, bool HAS_SKIP, bool SPLIT = false, bool ACCB = false, bool STREAM = false>
__global__ __launch_bounds__(256, (SPLIT && !ACCB) ? 3 : 2) void deconv3d_bf16s(const float* __restrict__ in, const uint4* __restrict__ wsplit,
                                                          const float* __restrict__ shift, const float* __restrict__ skip,
                                                          const uint4* __restrict__ skip_wsplit, float* __restrict__ out,
                                                          int Cin, int D, int H, int W, int Cout, int Cs, int tiles_w,
                                                          int tiles_h, int relu, int nunits) {
    static_assert(!SPLIT || NTERMS == F16X3, "the class-group split exists for the fp16 form (weights through LDS)");
    static_assert(!ACCB || SPLIT, "chunk-blocked accumulation: in the registers the split frees");
    // NTERMS = 19 (F16X3): the main loop on two fp16 terms with block-floating operands (split_f16.h; the scheme of
    // conv3d_bf16s.hip); the skip projection, whose operand never passes through LDS, stays on three bf16 terms
    constexpr bool F16 = (NTERMS == F16X3);
    constexpr int NC = (NTERMS == 6) ? 3 : 2;
    constexpr int NCW = F16 ? 2 : 3;                           // terms in the packed main weights
    using C = DB<TD, TH, NC, F16, SPLIT ? 15 : 27>;
    constexpr int MSLOT = NC * 2 * C::CS;                      // LDS slot of the waves' maxima
    // fp16 form: the chunk's weight fragments (27 K-steps x 2 terms, 55 KB) are brought into LDS once per workgroup by
    // LDS-DMA loads (buffer_load_dwordx4 ... lds: no registers) instead of being fetched by each of the four waves: a K-step
    // has only three MFMAs per wave here, and eight waves per CU re-reading 2 KB of weights every 96 cycles is 85 B/clk
    // against the 64 B/clk of the vector L1 -- the main loop ran at the L1's pace (tools: phases of this kernel)
    constexpr int WL = MSLOT + 1 + 16;
    extern __shared__ __attribute__((aligned(16))) uint4 lds[];   // [NC terms][2 channel halves][CS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    // a workgroup's unit of work: a tile (SPLIT: a tile and one of its two class groups, unit = 2 * tile + group)
    int iw0 = 0, ih0 = 0, id0 = 0, grp = -1;                    // origin (and SPLIT: class group, wave-uniform) of the current unit
    auto unit_origin = [&](int u, int& w0, int& h0, int& d0) {
        int t = SPLIT ? (u >> 1) : u;
        const int tw = t % tiles_w; t /= tiles_w;
        const int th = t % tiles_h; t /= tiles_h;
        w0 = tw * 32; h0 = th * TH; d0 = t * TD;
    };
    const int co0 = blockIdx.y * 32;
    const int b = blockIdx.z;
    const int dzw = wave / TH, hyw = wave % TH;
    const int lane_pos = (dzw * C::IH + hyw) * C::IW + l31;        // this lane's position, offset (0,0,0)

    const size_t in_plane = (size_t)H * W, chan = (size_t)D * in_plane;
    const float* inb = in + (size_t)b * Cin * chan;

    // staging plan: this thread owns positions p = tid + 256*i of the halo tile, all 16 channels of the chunk
    unsigned poff[C::NPOS];
    auto make_poff = [&](int u) {                                  // (a unit that does not exist: every position beyond the buffer)
        int w0, h0, d0;
        unit_origin(u, w0, h0, d0);
#pragma unroll
        for (int i = 0; i < C::NPOS; ++i) {
            const int p = tid + 256 * i;
            const int wx = p % C::IW;
            int r = p / C::IW;
            const int hy = r % C::IH, dz = r / C::IH;
            const int gw = w0 + wx, gh = h0 + hy, gd = d0 + dz;
            const bool ok = (p < C::CS) && gd < D && gh < H && gw < W && u < nunits;
            poff[i] = ok ? (unsigned)(((size_t)gd * in_plane + (size_t)gh * W + gw) * 4) : 0x80000000u;     // beyond the buffer: reads 0
        }
    };
    constexpr int NQ = 16 * C::NPOS;
    float rin[NQ];

    const int nchunks = (Cin + 15) / 16;
    const int G = nchunks * KST;
    const int wlane = (half * Cout + min(co0 + l31, Cout - 1)) * 16;
    const int wstep = NCW * 2 * Cout * 16;                     // bytes per K-step of the main weights
    const int swstep = 3 * 2 * Cout * 16;                      // ... of the skip projection's (always three bf16 terms)
    const float* wunscale = reinterpret_cast<const float*>(reinterpret_cast<const char*>(wsplit) + (size_t)G * wstep);
    const __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint4*>(wsplit), 0, (int)min((long long)G * wstep, 0x7fffffffLL), 0x00020000);
    const __amdgpu_buffer_rsrc_t ires = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(inb), 0, (int)min((long long)Cin * (long long)chan * 4, 0x7fffffffLL), 0x00020000);
    const int chan_b = (int)(chan * 4);
    auto load_a = [&](int g, int c) {
        return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(wres, wlane, g * wstep + c * 2 * Cout * 16, 0));
    };
    // channels beyond Cin (ragged last chunk) lie beyond the buffer for the LAST positions only; clamp and zero instead
    auto load_in = [&](int ch, int i) {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ires, (int)poff[i], min(ch, Cin - 1) * chan_b, 0));
    };
    auto load_in_masked = [&](int ch, int i, unsigned mask) {          // mask 0x80000000: beyond the buffer, reads 0
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ires, (int)(poff[i] | mask), min(ch, Cin - 1) * chan_b, 0));
    };
    constexpr int AQ = F16 ? 2 : 3;       // fragment ring: fp16 form (from LDS) one step ahead, bf16 forms (from L2) two
    uint4 aq[AQ][NC];                     // aq[step % AQ]

    const int Do = 2 * D, Ho = 2 * H, Wo = 2 * W;
    const size_t out_plane = (size_t)Ho * Wo;
    // per-channel epilogue constants, fetched now and parked in LDS (read after the K loop they are an exposed round trip
    // to L2/HBM per workgroup): aff[c] = shift, aff[32 + c] = 2^-(weight scale) (fp16 form)
    float* aff = reinterpret_cast<float*>(&lds[MSLOT + 1]);
    if (tid < 32) {
        const int co = min(co0 + tid, Cout - 1);
        aff[tid] = shift ? shift[co] : 0.0f;
        aff[32 + tid] = F16 ? wunscale[co] : 1.0f;
    }
    // Half of the workgroups project the skip tensor BEFORE the main loop, half after it: a grid whose workgroups all
    // start together otherwise alternates between a phase where every CU multiplies and one where every CU waits for
    // HBM (measured on the last layer of hourglass2: 173 us of main loop + 61 us of skip reads + 29 us of stores = the
    // 263 us of the whole kernel, nothing overlapped).  Same sums, deterministic per workgroup.
    constexpr int SNT = F16 ? 6 : NTERMS, SNC = (SNT == 6) ? 3 : 2;      // cross products / terms of the skip projection
    // ---- everything from here on is written once for a compile-time class group (Grp<GRP>) and instantiated for the one(s) this
    // kernel serves; `acc[j * 2 + pw]` holds class Grp::q(j) * 2 + pw ----
    auto body = [&](auto grp_tag, const int unit) {
    constexpr int GRP = decltype(grp_tag)::value;
    using GP = Grp<GRP>;
    constexpr int NA = GP::NA, NS = GP::NS;
    // XPF: the next unit's first chunk prefetched under this unit's last chunk (as conv3d_bf16s does).  Off: the prefetched
    // registers would stay live through the skip projection and the epilogue beside all accumulators (73 spilled registers
    // measured at the 256 of two workgroups per CU); the next unit's first loads are issued at its top instead, while this
    // unit's stores drain.
    constexpr bool XPF = false;
    const bool has_next = XPF && unit + (int)gridDim.x < nunits;
    if (!XPF) {
        make_poff(unit);
#pragma unroll
        for (int q = 0; q < NQ; ++q) rin[q] = load_in(q / C::NPOS, q % C::NPOS);
    }
    if (!F16) {                           // bf16 forms: the first two K-steps' weight fragments of this unit
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int c = 0; c < NC; ++c) aq[k][c] = load_a(min(k, G - 1), c);
    }
    f32x16 acc[NA];
#pragma unroll
    for (int p = 0; p < NA; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
    auto skip_phase = [&]() {
#ifdef SS_EXP_DECONV_NOSKIP           // (timing experiment, wrong results)
        return;
#endif
        // ---- 1x1x1 projection of the skip tensor at the 8 output positions of every lane: per 16 skip channels one K-step
        // per parity class; the operand (8 channels x this lane's 2x2x2 cube) comes straight from global memory ----
        const size_t schan = (size_t)Do * out_plane;
        const int jw_ = min(iw0 + l31, W - 1), jd_ = min(id0 + dzw, D - 1), jh_ = min(ih0 + hyw, H - 1);
        const __amdgpu_buffer_rsrc_t sres = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(skip + (size_t)b * Cs * schan), 0, (int)min((long long)Cs * (long long)schan * 4, 0x7fffffffLL), 0x00020000);
        const int nks = (Cs + 15) / 16;
        const __amdgpu_buffer_rsrc_t swres = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint4*>(skip_wsplit), 0, nks * swstep, 0x00020000);
        const unsigned lane_s = (unsigned)(((size_t)(2 * jd_) * out_plane + (size_t)(2 * jh_) * Wo + 2 * jw_) * 4);
        const unsigned schan_b = (unsigned)(schan * 4);
        // one batch = two (pd, ph) rows of the cube x 8 channels of this lane's half: 16 eight-byte loads (32 registers)
        auto batch_load = [&](float2 (&v)[2][8], int ks, auto q0_tag, unsigned dead) {
            constexpr int q0 = decltype(q0_tag)::value;
#pragma unroll
            for (int q = 0; q < 2; ++q) {                  // row q0 + q of the group = (pd, ph); the float2 holds pw = 0, 1
                const unsigned qo = (unsigned)(((size_t)(GP::q(q0 + q) >> 1) * out_plane + (size_t)(GP::q(q0 + q) & 1) * Wo) * 4);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int cs = ks * 16 + 8 * half + j;     // channels beyond Cs: a clamped (valid) address, value zeroed
                    v[q][j] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(
                                                             sres, (int)((lane_s + qo + (unsigned)min(cs, Cs - 1) * schan_b) | dead), 0, SS_DECONV_SKIP_AUX));
                }
            }
        };
        auto batch_mfma = [&](float2 (&v)[2][8], int ks, auto q0_tag, const bf16x8 (&a)[SNC]) {
            constexpr int q0 = decltype(q0_tag)::value;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (ks * 16 + 8 * half + j >= Cs) v[q][j] = make_float2(0.f, 0.f);
#pragma unroll
                for (int pw = 0; pw < 2; ++pw) {
                    unsigned bh[4], bm[4], bl[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        split3_pk(pw ? v[q][2 * c].y : v[q][2 * c].x, pw ? v[q][2 * c + 1].y : v[q][2 * c + 1].x, bh[c], bm[c], bl[c]);
                    const bf16x8 h8 = __builtin_bit_cast(bf16x8, make_uint4(bh[0], bh[1], bh[2], bh[3]));
                    const bf16x8 m8 = __builtin_bit_cast(bf16x8, make_uint4(bm[0], bm[1], bm[2], bm[3]));
                    const int cls = (q0 + q) * 2 + pw;
                    if (SNT == 6) {
                        const bf16x8 l8 = __builtin_bit_cast(bf16x8, make_uint4(bl[0], bl[1], bl[2], bl[3]));
                        acc[cls] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], m8, acc[cls], 0, 0, 0);
                        acc[cls] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], l8, acc[cls], 0, 0, 0);
                        acc[cls] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[SNC - 1], h8, acc[cls], 0, 0, 0);
                    }
                    acc[cls] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], m8, acc[cls], 0, 0, 0);
                    acc[cls] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], h8, acc[cls], 0, 0, 0);
                    acc[cls] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], h8, acc[cls], 0, 0, 0);
                }
            }
        };
        using Q0 = std::integral_constant<int, 0>;
        using Q2 = std::integral_constant<int, 2>;
#pragma unroll 1
        for (int ks = 0; ks < nks; ++ks) {
            bf16x8 a[SNC];
#pragma unroll
            for (int c = 0; c < SNC; ++c)
                a[c] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(swres, wlane, ks * swstep + c * 2 * Cout * 16, 0));
            // all 16 eight-byte loads of a batch first, then its arithmetic: one exposed round trip per batch (interleaved by the
            // compiler, each (pd, ph) row waited for its own 8 loads)
            float2 v[2][8];
            batch_load(v, ks, Q0{}, 0u);
            __builtin_amdgcn_sched_barrier(0);
            batch_mfma(v, ks, Q0{}, a);
            if constexpr (GP::NQ2 == 4) {
                batch_load(v, ks, Q2{}, 0u);
                __builtin_amdgcn_sched_barrier(0);
                batch_mfma(v, ks, Q2{}, a);
            }
        }
    };


    // (SPLIT: always after the main loop -- before it the prefetched first chunk is live beside the skip operands, 92 spilled
    // registers at the 168 of three workgroups per CU; the two class groups of a tile and the third workgroup of the CU already
    // spread the phases)
    // (more than two populations -- some of the skip steps before the main loop, the rest after -- and the next batch's loads ahead of
    // this batch's MFMAs were both measured in r05 and change nothing: profiles/EXPERIMENTS.md part E)
    const bool skip_first = !SPLIT && HAS_SKIP && ((unit ^ blockIdx.y) & 1);
    if (HAS_SKIP && skip_first) {
        skip_phase();
        if (F16) {      // the main loop's accumulators carry the channel's weight scale (a power of two: exact)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                // (from global memory: `aff` is not visible before the first barrier)
                const float ws = __uint_as_float((254u << 23) - __float_as_uint(wunscale[min(co0 + (r & 3) + 8 * (r >> 2) + 4 * half, Cout - 1)]));
#pragma unroll
                for (int p = 0; p < NA; ++p) acc[p][r] *= ws;
            }
        }
    }
    // fp16 form: block-floating scale of the staged chunk (see conv3d_bf16s.hip)
    int e_cur = E_ONE, e_run = E_MIN;
    auto publish_max = [&](float m) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) m = fmaxf(m, fabsf(rin[q]));
        const unsigned wm = wave_max_bits(__float_as_uint(m));
        if (lane == 0) reinterpret_cast<unsigned*>(&lds[MSLOT])[wave] = wm;
    };
    if (F16) {
        float m0 = 0.f;
        if (HAS_SKIP && skip_first) {                          // see E_INIT_SHIFT (split_f16.h)
#pragma unroll
            for (int p = 0; p < NA; ++p)
#pragma unroll
                for (int r = 0; r < 16; ++r) m0 = fmaxf(m0, fabsf(acc[p][r]));
            m0 *= __uint_as_float((unsigned)(127 - E_INIT_SHIFT) << 23);
        }
        publish_max(m0);
        __syncthreads();
    }
    const int nchunks_run = nchunks;
    for (int ck = 0, g0 = 0; ck < nchunks_run; ++ck, g0 += KST) {
        const int ci0 = ck * 16;
        // ---- split + transpose: registers -> [term][half][position] ----
        if (F16) {          // this chunk's weights: wave w issues the (K-step, term) pairs i = w, w + 4, ...; lane -> (half, channel)
            // (the wave index as a SCALAR made opaque once per chunk: the slab entries' LDS addresses and source offsets are then
            // scalar arithmetic recomputed here -- as per-lane values they are loop invariants the compiler hoists out of the chunk
            // loop and, under the register pressure of the class-blocked loop, spills: 130 registers, 112 -> 450 us)
            int wv = __builtin_amdgcn_readfirstlane(wave);
            asm volatile("" : "+s"(wv));
#pragma unroll
            for (int k = 0; k < (NS * 2 + 3) / 4; ++k) {
                const int i = wv + 4 * k;                      // wave-uniform; slab entry i = (the group's step i / 2, term i & 1)
                if (i < NS * 2) {
                    const int sg = GRP < 0 ? i / 2 : __builtin_amdgcn_readfirstlane(kGroupSteps.s[GRP < 0 ? 0 : GRP][i / 2]);
                    lds_dma16(wres, &lds[WL + i * 64], wlane, (g0 + sg) * wstep + (i & 1) * 2 * Cout * 16);
                }
            }
        }
        float in_scale = 1.f;
        if (F16) {
            const uint4 wm = lds[MSLOT];
            const int e_new = max(e_run, (int)(max(max(wm.x, wm.y), max(wm.z, wm.w)) >> 23));
            e_run = e_new;
            if (e_new != e_cur) {                              // wave-uniform; exact power-of-two rescale
                const float ratio = __uint_as_float((unsigned)max(127 + e_cur - e_new, 0) << 23);
#pragma unroll
                for (int p = 0; p < NA; ++p)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[p][r] *= ratio;
                e_cur = e_new;
            }
            in_scale = __uint_as_float((unsigned)(127 + E_ONE - e_cur) << 23);
        }
#pragma unroll
        for (int i = 0; i < C::NPOS; ++i) {
            const int p = tid + 256 * i;
            if (p >= C::CS) continue;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                unsigned hh[4], mm[4], ll[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int c0 = hf * 8 + 2 * c;
                    const float x0 = (ci0 + c0 < Cin) ? rin[c0 * C::NPOS + i] : 0.f;
                    const float x1 = (ci0 + c0 + 1 < Cin) ? rin[(c0 + 1) * C::NPOS + i] : 0.f;
                    if (F16) split2_pk_f16(x0 * in_scale, x1 * in_scale, hh[c], mm[c]);
                    else split3_pk(x0, x1, hh[c], mm[c], ll[c]);
                }
                lds[(0 * 2 + hf) * C::CS + p] = make_uint4(hh[0], hh[1], hh[2], hh[3]);
                lds[(1 * 2 + hf) * C::CS + p] = make_uint4(mm[0], mm[1], mm[2], mm[3]);
                if (NC == 3) lds[(2 * 2 + hf) * C::CS + p] = make_uint4(ll[0], ll[1], ll[2], ll[3]);
            }
        }
#ifndef SS_EXP_DECONV_NOWAIT      // (timing experiment, wrong results: the upper bound of what a look-ahead of the weight DMA can buy)
        if (F16) __builtin_amdgcn_s_waitcnt(0x0F70);           // vmcnt(0): the LDS-DMA weight loads have landed
#endif
        __syncthreads();
        const bool more = ck + 1 < nchunks;
        // what the K-steps prefetch: the next chunk of this unit, or (last chunk) the first chunk of the workgroup's next unit
        if (XPF && !more) make_poff(unit + (int)gridDim.x);     // pure index arithmetic under a wave-uniform branch
        const unsigned nomore = (more || has_next) ? 0u : 0x80000000u;
        const int ch_next = more ? ci0 + 16 : 0;

        uint4 bcur[NC], bnxt[NC];
        auto read_b = [&](uint4 (&dst)[NC], int o) {
            const int slot = lane_pos + (((o >> 2) & 1) * C::IH + ((o >> 1) & 1)) * C::IW + (o & 1);
#pragma unroll
            for (int c = 0; c < NC; ++c) dst[c] = lds[(c * 2 + half) * C::CS + slot];
        };
        constexpr int QSG = (NQ + (NS * 20 / 27) - 1) / (NS * 20 / 27);     // the next chunk's loads: over the first ~3/4 of the steps
        read_b(bcur, off_of(GP::T.step[0]));
        if (F16) {
#pragma unroll
            for (int c = 0; c < NC; ++c) aq[0][c] = lds[WL + c * 64 + lane];
        }
        // ACCB: the chunk's MFMAs accumulate from ZERO in a second register set that joins `acc` once per chunk (two-level
        // blocked summation, as the CPU GEMMs the reference runs on: conv3d_bf16s.hip has the measurements)
        f32x16 tacc[ACCB ? NA : 1];
        if constexpr (ACCB) {
#pragma unroll
            for (int p = 0; p < NA; ++p)
#pragma unroll
                for (int r = 0; r < 16; ++r) tacc[p][r] = 0.f;
        }
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int s = GP::T.step[k];                       // (compile-time after unrolling)
            // no vector-memory instruction under a branch (see conv3d_bf16s.hip: the wait-count pass falls back to vmcnt(0)
            // at control-flow merges): past the end the last fragment is requested again, the input loads go beyond the buffer
            if (!F16) {
#pragma unroll
                for (int c = 0; c < NC; ++c) aq[(s + 2) % AQ][c] = load_a(min(g0 + s + 2, G - 1), c);
            } else if (k + 1 < NS) {                           // next step's fragments from the LDS copy
#pragma unroll
                for (int c = 0; c < NC; ++c) aq[(k + 1) % AQ][c] = lds[WL + ((k + 1) * 2 + c) * 64 + lane];
            }
#pragma unroll
            for (int q = k * QSG; q < (k + 1) * QSG && q < NQ; ++q) rin[q] = load_in_masked(ch_next + q / C::NPOS, q % C::NPOS, nomore);
            // first tap of a run of equal input offsets: fetch the next run's activation fragment
            if ((k == 0 || off_of(s) != off_of(GP::T.step[k > 0 ? k - 1 : 0])) && GP::T.next_off[k] >= 0) read_b(bnxt, GP::T.next_off[k]);
            const int ai = GP::idx(cls_of(s));
            f32x16& dst = ACCB ? tacc[ACCB ? ai : 0] : acc[ai];
            if (F16) {
                f16x8 a[NC], bq[NC];
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    a[c] = __builtin_bit_cast(f16x8, aq[k % AQ][c]);
                    bq[c] = __builtin_bit_cast(f16x8, bcur[c]);
                }
                dst = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], bq[1], dst, 0, 0, 0);
                dst = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], bq[0], dst, 0, 0, 0);
                dst = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], bq[0], dst, 0, 0, 0);
            } else {
                bf16x8 a[NC], bq[NC];
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    a[c] = __builtin_bit_cast(bf16x8, aq[s % AQ][c]);
                    bq[c] = __builtin_bit_cast(bf16x8, bcur[c]);
                }
                if (NTERMS == 6) {
                    dst = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], bq[1], dst, 0, 0, 0);
                    dst = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bq[NC - 1], dst, 0, 0, 0);
                    dst = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[NC - 1], bq[0], dst, 0, 0, 0);
                }
                dst = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bq[1], dst, 0, 0, 0);
                dst = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], bq[0], dst, 0, 0, 0);
                dst = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bq[0], dst, 0, 0, 0);
            }
            // last tap of a run: the prefetched fragment becomes current
            if (k + 1 < NS && off_of(GP::T.step[k + 1 < NS ? k + 1 : k]) != off_of(s)) {
#pragma unroll
                for (int c = 0; c < NC; ++c) bcur[c] = bnxt[c];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (ACCB) {
#pragma unroll
            for (int p = 0; p < NA; ++p)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[p][r] += tacc[p][r];
        }
        // steps 27, 28 of this chunk are steps 0, 1 of the next: re-base the fragment ring (27 % 3 == 0: already in place)
        if (F16 && more) publish_max(0.f);                        // of the chunk staged next (a next UNIT's first chunk: at that unit's top)
        __syncthreads();
    }
    if (F16) {          // back to plain values: 2^-(activation scale) x the channel's 2^-(weight scale), exact
        const float au = __uint_as_float((unsigned)(127 - E_ONE + e_cur) << 23);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float un = au * aff[32 + (r & 3) + 8 * (r >> 2) + 4 * half];
#pragma unroll
            for (int p = 0; p < NA; ++p) acc[p][r] *= un;
        }
    }

    if (HAS_SKIP && !skip_first) skip_phase();

    // ---- epilogue: each lane owns the 2x2x2 output cube of its input position.  Buffer stores: a 32-bit per-lane offset
    // per (plane, row) pair of the cube (positions outside the volume parked beyond the buffer: the store is dropped) and
    // a scalar offset per channel -- no 64-bit per-lane arithmetic, no branches ----
    const int jw = iw0 + l31, jd = id0 + dzw, jh = ih0 + hyw;
    const unsigned ochan_b = (unsigned)((size_t)Do * out_plane * 4);
    const __amdgpu_buffer_rsrc_t ores = __builtin_amdgcn_make_buffer_rsrc(
        out + (size_t)b * Cout * Do * out_plane, 0, (int)min((long long)Cout * (long long)ochan_b, 0x7fffffffLL), 0x00020000);
    const bool ok = jw < W && jd < D && jh < H;
    unsigned vo[GP::NQ2];
#pragma unroll
    for (int j = 0; j < GP::NQ2; ++j)
        vo[j] = ok ? (unsigned)((((size_t)(2 * jd + (GP::q(j) >> 1)) * Ho + 2 * jh + (GP::q(j) & 1)) * Wo + 2 * jw) * 4) + 4u * half * ochan_b
                   : 0x80000000u;
    const float floor_v = relu ? 0.f : -__builtin_inff();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int cb = co0 + (r & 3) + 8 * (r >> 2);
        const bool cok = cb + 4 * half < Cout;
        const float sh = aff[(r & 3) + 8 * (r >> 2) + 4 * half];
#pragma unroll
        for (int j = 0; j < GP::NQ2; ++j) {
            const float v0 = fmaxf(ss::add_rn(acc[j * 2 + 0][r], sh), floor_v);
            const float v1 = fmaxf(ss::add_rn(acc[j * 2 + 1][r], sh), floor_v);
#ifdef SS_EXP_DECONV_NOSTORE          // (timing experiment: one store per lane instead of 64)
            if (r == 0 && j == 0)
#endif
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(ss_u32x2, make_float2(v0, v1)), ores,
                                                  (int)(cok ? vo[j] : 0x80000000u), cb * (int)ochan_b, STREAM ? 2 : 0);
        }
    }
    };      // body
    // ONE unit per workgroup.  (The persistent form -- a workgroup walking units blockIdx.x, blockIdx.x + gridDim.x, ... so that a
    // unit's stores drain under the next one's loads -- was built and measured in r05: the loop costs 65 spilled registers at the
    // 256 of two workgroups per CU, 129 -> 177 us on hourglass2.conv6, and with the spills equalised persistence itself bought
    // 4 %: profiles/r05_e_deconv_persist.txt.)
#if SS_DECONV_XCD
    // XCD-aware unit order (r06): workgroups are dispatched round-robin over the 8 XCDs, so with unit = blockIdx.x a tile and its
    // neighbours in h (4 indices away) and d never share an L2 and every halo row / plane is fetched from the fabric again (PMC r05:
    // 537 MB for 453 algorithmic on hourglass2.conv6, the layer that runs at the copy rate of the bytes it moves).  Here XCD k walks
    // the contiguous k-th eighth of the units: a bijection of [0, nunits) (XCD x gets nunits / 8 units, the first nunits % 8 XCDs one more).
    const int unit = [&]() {
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3, q = nunits >> 3, r = nunits & 7;
        return x * q + min(x, r) + j;
    }();
#else
    const int unit = blockIdx.x;
#endif
    unit_origin(unit, iw0, ih0, id0);
    if constexpr (!SPLIT) {
        body(std::integral_constant<int, -1>{}, unit);
    } else {
        grp = unit & 1;
        if (grp == 0) body(std::integral_constant<int, 0>{}, unit);
        else body(std::integral_constant<int, 1>{}, unit);
    }
}

// wpack [Cin][27][Cout] fp32 (ss_pack_conv3d_weights, transposed form, BN scale folded by the caller) ->
// [ceil(Cin/16)][27 K-steps in offset-grouped tap order][3 terms][2 channel halves][Cout][8] bf16
__global__ void pack_deconv_weights_kernel(const float* __restrict__ wpack, unsigned short* __restrict__ wsplit, int Cin,
                                           int Cout, int ntaps, long long total) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int j = (int)(i % 8);
    long long r = i / 8;
    const int co = (int)(r % Cout); r /= Cout;
    const int hf = (int)(r % 2); r /= 2;
    const int term = (int)(r % 3); r /= 3;
    const int s = (int)(r % ntaps);
    const int chunk = (int)(r / ntaps);
    const int ci = chunk * 16 + hf * 8 + j;
    const int tap = (ntaps == 27) ? tap_of(s) : 0;
    const float x = (ci < Cin) ? wpack[((size_t)ci * ntaps + tap) * Cout + co] : 0.f;
    unsigned h, m, l;
    split3_pk(x, 0.f, h, m, l);
    wsplit[i] = (unsigned short)((term == 0 ? h : (term == 1 ? m : l)) & 0xffffu);
}

// fp16 form, pass 1: per output channel the power of two bringing max |w| into [2^14, 2^15); its inverse goes behind the terms
__global__ __launch_bounds__(256) void deconv_weight_unscale_f16s_kernel(const float* __restrict__ wpack, float* __restrict__ wunscale,
                                                                          int Cout, int rows) {
    __shared__ unsigned wmax[4];
    float m = 0.f;
    for (int i = threadIdx.x; i < rows; i += 256) m = fmaxf(m, fabsf(wpack[(size_t)i * Cout + blockIdx.x]));
    const unsigned wm = wave_max_bits(__float_as_uint(m));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = wm;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int e = max((int)(max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3])) >> 23), E_MIN);
        wunscale[blockIdx.x] = __uint_as_float((unsigned)(127 - E_ONE + e) << 23);
    }
}
// pass 2: -> [ceil(Cin/16)][27 K-steps][2 terms][2 channel halves][Cout][8] fp16 of w / wunscale[co]
__global__ void pack_deconv_weights_f16s_kernel(const float* __restrict__ wpack, unsigned short* __restrict__ wsplit,
                                                const float* __restrict__ wunscale, int Cin, int Cout, long long total) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int j = (int)(i % 8);
    long long r = i / 8;
    const int co = (int)(r % Cout); r /= Cout;
    const int hf = (int)(r % 2); r /= 2;
    const int term = (int)(r % 2); r /= 2;
    const int s = (int)(r % 27);
    const int chunk = (int)(r / 27);
    const int ci = chunk * 16 + hf * 8 + j;
    const float x = (ci < Cin) ? wpack[((size_t)ci * 27 + tap_of(s)) * Cout + co] / wunscale[co] : 0.f;
    const _Float16 h = (_Float16)x;
    const _Float16 l = (_Float16)(x - (float)h);
    wsplit[i] = __builtin_bit_cast(unsigned short, term == 0 ? h : l);
}

template <int TD, int TH, int NTERMS, bool HAS_SKIP, bool SPLIT = false, bool ACCB = false, bool STREAM = false>
int launch_db(const float* in, const void* wsplit, const float* shift, const float* skip, const void* skip_wsplit, float* out,
              int B, int Cin, int D, int H, int W, int Cout, int Cs, int relu, hipStream_t st) {
    using C = DB<TD, TH, (NTERMS == 6) ? 3 : 2, NTERMS == F16X3, SPLIT ? 15 : 27>;
    const int tiles_w = ss::ceil_div(W, 32), tiles_h = ss::ceil_div(H, TH), tiles_d = ss::ceil_div(D, TD);
    const long long nt = (long long)tiles_w * tiles_h * tiles_d;
    if (nt > 0x7fffffffLL || B > 65535) return SS_ERR_UNSUPPORTED;
    auto kern = deconv3d_bf16s<TD, TH, NTERMS, HAS_SKIP, SPLIT, ACCB, STREAM>;
    if (C::LDS_BYTES > 64 * 1024) {
        if (ss::ensure_dynamic_lds(reinterpret_cast<const void*>(kern), (int)C::LDS_BYTES) != SS_OK) return SS_ERR_LAUNCH;
    }
    if (nt > 0x3fffffffLL) return SS_ERR_UNSUPPORTED;
    const long long nunits = SPLIT ? 2 * nt : nt;
    dim3 grid((unsigned)nunits, ss::ceil_div(Cout, 32), B);
    hipLaunchKernelGGL(kern, grid, dim3(256), C::LDS_BYTES, st, in, reinterpret_cast<const uint4*>(wsplit), shift, skip,
                       reinterpret_cast<const uint4*>(skip_wsplit), out, Cin, D, H, W, Cout, Cs, tiles_w, tiles_h, relu, (int)nunits);
    return ss::check_launch();
}

template <int TD, int TH>
int launch_db_all(const float* in, const void* wsplit, const float* shift, const float* skip, const void* skip_wsplit,
                  float* out, int B, int Cin, int D, int H, int W, int Cout, int Cs, int relu, int nterms, hipStream_t st) {
#define SS_DB(NTERMS, HAS_SKIP) launch_db<TD, TH, NTERMS, HAS_SKIP>(in, wsplit, shift, skip, skip_wsplit, out, B, Cin, D, H, W, Cout, Cs, relu, st)
#define SS_DF(HAS_SKIP, SPLIT, ACCB, STREAM) \
    launch_db<TD, TH, F16X3, HAS_SKIP, SPLIT, ACCB, STREAM>(in, wsplit, shift, skip, skip_wsplit, out, B, Cin, D, H, W, Cout, Cs, relu, st)
    if (nterms == F16X3) {
        // Parity-class groups (r05, Grp<> above), measured alone on the four transposed convs of the 1024^2 pair (hourglass2.conv6 /
        // .conv5, hourglass_att.conv6 / .conv5; profiles/r05_f_deconv_forms.txt): all 8 classes per workgroup 128 / 55 / 38 / 39 us;
        // two class groups at three workgroups per CU 148 / 56 / 55 / 36; two groups + chunk-blocked accumulation 163 / 75 / 55 / 36.
        // The second staging of the input tile costs more than the third workgroup per CU returns wherever the layer fills the
        // chip; it pays -- and brings the blocked sum's accuracy -- on a layer whose tiles number fewer than the chip's CUs.
        // That is a property of the LAYER (tiles x channel tiles of ONE pair), so a pair gets the same bits at every batch size.
        // SS_DECONV_GROUPS=0 / 1 / 2 forces the form (2 = groups + chunk-blocked accumulation).
        const long long per_pair = (long long)ss::ceil_div(W, 32) * ss::ceil_div(H, TH) * ss::ceil_div(D, TD) * ss::ceil_div(Cout, 32);
        int groups = ss::tuning().deconv_groups;
        if (groups < 0) groups = per_pair < 256 ? 2 : 0;
        const int stream_env = ss::tuning().deconv_stream;
        const bool stream = stream_env >= 0 ? stream_env != 0 : (long long)B * Cout * 8 * D * H * W * 4 >= (192LL << 20);
        if (groups == 2) return skip != nullptr ? SS_DF(true, true, true, false) : SS_DF(false, true, true, false);
        if (groups == 1) return skip != nullptr ? SS_DF(true, true, false, false) : SS_DF(false, true, false, false);
        if (stream) return skip != nullptr ? SS_DF(true, false, false, true) : SS_DF(false, false, false, true);
        return skip != nullptr ? SS_DF(true, false, false, false) : SS_DF(false, false, false, false);
    }
    if (skip != nullptr) return nterms == 6 ? SS_DB(6, true) : SS_DB(3, true);
    return nterms == 6 ? SS_DB(6, false) : SS_DB(3, false);
#undef SS_DF
#undef SS_DB
}

}  // namespace

extern "C" int ss_pack_deconv3d_weights_bf16s(const float* wpack, void* wsplit, int Cin, int Cout, int ntaps, ss_stream_t stream) {
    SS_REQUIRE(wpack && wsplit && Cin > 0 && Cout > 0 && (ntaps == 27 || ntaps == 1));
    const long long total = (long long)((Cin + 15) / 16) * ntaps * 3 * 2 * Cout * 8;
    hipLaunchKernelGGL(pack_deconv_weights_kernel, dim3((unsigned)ss::ceil_div_ll(total, 256)), dim3(256), 0, ss::as_stream(stream),
                       wpack, reinterpret_cast<unsigned short*>(wsplit), Cin, Cout, ntaps, total);
    return ss::check_launch();
}

// the two-term fp16 form of the main weights (nterms = 19 of ss_deconv3d_bf16s_fwd; the skip projection keeps the bf16 form)
extern "C" int ss_pack_deconv3d_weights_f16s(const float* wpack, void* wsplit, int Cin, int Cout, ss_stream_t stream) {
    SS_REQUIRE(wpack && wsplit && Cin > 0 && Cout > 0);
    const long long total = (long long)((Cin + 15) / 16) * 27 * 2 * 2 * Cout * 8;
    float* wunscale = reinterpret_cast<float*>(reinterpret_cast<char*>(wsplit) + total * 2);
    hipLaunchKernelGGL(deconv_weight_unscale_f16s_kernel, dim3(Cout), dim3(256), 0, ss::as_stream(stream), wpack, wunscale, Cout, Cin * 27);
    hipLaunchKernelGGL(pack_deconv_weights_f16s_kernel, dim3((unsigned)ss::ceil_div_ll(total, 256)), dim3(256), 0, ss::as_stream(stream),
                       wpack, reinterpret_cast<unsigned short*>(wsplit), wunscale, Cin, Cout, total);
    return ss::check_launch();
}

extern "C" int ss_deconv3d_bf16s_fwd(const float* in, const void* wsplit, const float* shift, const float* skip,
                                     const void* skip_wsplit, float* out, int B, int Cin, int D, int H, int W, int Cout,
                                     int Cs, int relu, int nterms, ss_stream_t stream) {
    SS_REQUIRE(in && wsplit && out);
    SS_REQUIRE(B > 0 && Cin > 0 && D > 0 && H > 0 && W > 0 && Cout > 0 && (nterms == 3 || nterms == 6 || nterms == F16X3));
    SS_REQUIRE((skip == nullptr) || (skip_wsplit != nullptr && Cs > 0));
    SS_REQUIRE((reinterpret_cast<uintptr_t>(wsplit) & 15) == 0 && (reinterpret_cast<uintptr_t>(skip_wsplit) & 15) == 0);
    if ((reinterpret_cast<uintptr_t>(out) & 7) != 0) return SS_ERR_INVALID;
    if ((long long)Cin * D * H * W * 4 >= 0x7fffffffLL) return SS_ERR_UNSUPPORTED;
    if (skip != nullptr && (long long)Cs * 8 * D * H * W * 4 >= 0x7fffffffLL) return SS_ERR_UNSUPPORTED;
    if ((long long)Cout * 8 * D * H * W * 4 >= 0x7fffffffLL) return SS_ERR_UNSUPPORTED;       // 32-bit output offsets per batch element
    hipStream_t st = ss::as_stream(stream);
    if (D >= 2 && H < 4)
        return launch_db_all<2, 2>(in, wsplit, shift, skip, skip_wsplit, out, B, Cin, D, H, W, Cout, Cs, relu, nterms, st);
    return launch_db_all<1, 4>(in, wsplit, shift, skip, skip_wsplit, out, B, Cin, D, H, W, Cout, Cs, relu, nterms, st);
}
