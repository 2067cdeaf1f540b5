// 3x3x3 stride-1 Conv3d (+ folded BatchNorm, partial sum / residual, ReLU, channelAtt gate) in the two-term fp16 form
// (split_f16.h) as ONE software pipeline per CU: the large layers' instantiation of conv3d_bf16s.hip's kernel
// (same GEMM mapping, same packed weights; results equal up to fp32 summation order).
//
// Why a second kernel: with half the MFMAs of the bf16 form, a workgroup of conv3d_bf16s<1,4,2,8,19> spends 40 % of its
// life before its first and after its last MFMA and a fifth of its K loop between chunks (tools/wg_phases.py).  Here
// one 8-wave workgroup owns a CU (two waves per SIMD, 256 VGPRs each, 131 KB of LDS) and
//   * walks the depth tiles (4 planes x 8 rows x 32 columns) of its (row, column) position, so set-up is paid once,
//     the first chunk of the next tile is fetched under the last chunk of this one and a tile's stores drain under
//     the next tile's K loop;
//   * double-buffers the LDS image: the next chunk is fetched under K-steps 0-3, its block-floating maximum agreed on
//     at K-step 8 (one barrier), it is split and written to the other image under K-steps 9-12, all between MFMAs
//     -- a chunk boundary is one barrier with nothing left to do behind it;
//   * a partial sum of the same convolution (concat_stem's broadcast half) joins the accumulator in the epilogue,
//     before the affine, instead of initialising it (no prologue loads, no overflow bound to keep: split_f16.h).
// Measured (32 -> 32 on [24,256,256]): 255 us against 285-300 us for the tiled kernel; with partial sum and gate
// (concat_stem) 317 against 320 -- that instantiation spills 24 registers and its epilogue waits four times per tile
// for its side inputs.  A first version with one 4-wave workgroup per CU (512 VGPRs per wave, gate and next partial sum
// prefetched into registers) was parity-green and slower (304-380 us): with one wave per SIMD every vector-memory
// instruction that waits for a queue slot stalls the MFMAs behind it (48 input loads per chunk: 2.4 k cycles; the
// split's 150 VALU operations: 1.5 k), where a second wave simply runs.
#include <algorithm>
#include <stdlib.h>

#include "common.h"
#include "split_f16.h"

#ifdef SS_TIMING     // phase times of every workgroup (tools/wg_phases.py pipe); not part of the product build
__device__ unsigned long long ss_dbgp_t[8 * 4096];
extern "C" int ss_debug_read_p(unsigned long long* dst, int n) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(ss_dbgp_t), (size_t)n * 8) == hipSuccess ? 0 : -1;
}
#define SS_T(var) const long long var = __builtin_readcyclecounter()
#define SS_TADD(acc, t0) acc += (long long)__builtin_readcyclecounter() - (t0)
#else
#define SS_T(var) do {} while (0)
#define SS_TADD(acc, t0) do {} while (0)
#endif

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int NW = 8, NTHR = NW * 64;             // waves / threads per workgroup (two waves per SIMD, one workgroup per CU)
constexpr int NT = 4, TD = 4, TH = 8, KSTEPS = 14, NC = 2, KT = 27;
static_assert(TD * TH == NW * NT, "NW waves x NT rows tile TD x TH");
constexpr int ID = TD + 2, IH = TH + 2, IW = 34;
constexpr int CS = ID * IH * IW;                  // 1360 positions in the halo tile
constexpr int NPOS = (CS + NTHR - 1) / NTHR;      // 4 positions per thread
constexpr int NQ = 8 * NPOS;                      // prefetch registers (8 channels per chunk)
constexpr int IMG = NC * CS;                      // 16-byte slots of one image [term][position][8 ch]
constexpr int ZSLOT = 2 * IMG;                    // the all-zero slot (the 28th half-step)
constexpr int MSLOT = ZSLOT + 1, PSLOT = ZSLOT + 3;     // the 8 waves' maxima of the staged chunk (2 slots; PSLOT: spare)
#ifndef SS_P_AHEAD
#define SS_P_AHEAD 6
#endif
#ifndef SS_P_AHEAD_GATED
#define SS_P_AHEAD_GATED 2
#endif
constexpr int AFF = ZSLOT + 5;                    // 24 slots: scale[32], shift[32], 2^-(weight scale)[32]
constexpr int SLOTS = AFF + 24;
constexpr size_t LDS_BYTES = (size_t)SLOTS * 16;
constexpr int IN_STEPS = 4, QS = NQ / IN_STEPS;   // the next chunk's loads are issued under K-steps 0-3
constexpr int MSTEP = 8;                          // ... its maximum is agreed on at K-step 8, it is split under 9-12
// Weight fragments are fetched AP K-steps ahead into a ring of AR = 7 (14 % 7 == 0: a step's slot is the same in every
// chunk, no re-basing).  vmcnt retires in order: a wait for a fragment also waits for every load issued before it, and
// with one wave per SIMD nothing hides that, so no fragment may be requested after a slice of HBM loads whose latency
// it cannot afford -- six steps ahead, every fragment consumed under K-steps 0-9 was requested before the chunk's
// input loads or at least five steps after them.
constexpr int AR = 7;
static_assert(KSTEPS % AR == 0 && SS_P_AHEAD < AR, "ring slots repeat per chunk");
constexpr int BR = 3;                             // activation fragments: a ring of 3, read two rows ahead of their MFMAs
static_assert(NQ % IN_STEPS == 0, "whole slices");

// No vector-memory instruction of the K loop sits under a branch: where control flow merges, the compiler's wait-count
// pass can no longer tell how many loads are younger than the one it waits for and falls back to vmcnt(0) -- every
// K-step then waits for the weight fragments it has just requested (seen in the ISA of the first version).  Loads
// that are not needed are issued with an offset beyond the buffer instead (no memory access, the value is unused).
template <bool GATED, bool RES_PRE>
__global__ __launch_bounds__(NTHR, 1) void conv3d_f16p(const float* __restrict__ in, const uint4* __restrict__ wsplit,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       const float* __restrict__ residual, const float* __restrict__ gate,
                                                       float* __restrict__ out, int Cin, int D, int H, int W, int Cout,
                                                       int tiles_w, int tiles_h, int tiles_d, int tpw, int relu) {
    constexpr int AP = GATED ? SS_P_AHEAD_GATED : SS_P_AHEAD;     // (the gated epilogue needs the registers)
    extern __shared__ __attribute__((aligned(16))) uint4 lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef SS_TIMING
    long long t_a = 0, t_mid = 0, t_b = 0, t_end = 0, t_epi = 0, t_init = 0;
#endif
    SS_T(t_start);
    const int l31 = lane & 31, half = lane >> 5;
    int t = blockIdx.x;
    const int tw = t % tiles_w; t /= tiles_w;
    const int th = t % tiles_h; t /= tiles_h;
    const int td0 = t * tpw, ntile = min(tpw, tiles_d - td0);
    const int ow0 = tw * 32, oh0 = th * TH;
    const int co0 = blockIdx.y * 32;
    const int b = blockIdx.z;
    const int iw0 = ow0 - 1, ih0 = oh0 - 1;
    const int dzw = (wave * NT) / TH, hy0 = (wave * NT) % TH;
    const int lane_pos = (dzw * IH + hy0) * IW + l31;          // slot of this lane's first row, tap (0,0,0)

    // relu bit 0: ReLU; bit 1: `residual` is the initial value of the accumulators (a partial sum of the same convolution)
    constexpr bool res_pre = RES_PRE;
    const int G = ((Cin + 7) / 8) * KSTEPS;
    const int wstep = NC * 2 * Cout * 16;                      // bytes per K-step of packed weights
    const float* wunscale = reinterpret_cast<const float*>(reinterpret_cast<const char*>(wsplit) + (size_t)G * wstep);

    const size_t plane = (size_t)H * W, chan = (size_t)D * plane;          // stride 1: output geometry = input geometry
    const unsigned ochan_b = (unsigned)(chan * 4), gchan_b = (unsigned)(plane * 4);
    const int obytes = (int)min((long long)Cout * (long long)ochan_b, 0x7fffffffLL);
    const float* obase = out + (size_t)b * Cout * chan;
    const __amdgpu_buffer_rsrc_t ores = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(obase), 0, obytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(residual ? residual + (size_t)b * Cout * chan : obase), 0, residual ? obytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t ires = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(in + (size_t)b * Cin * chan), 0, (int)min((long long)Cin * (long long)chan * 4, 0x7fffffffLL), 0x00020000);
    const __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint4*>(wsplit), 0, (int)min((long long)G * wstep, 0x7fffffffLL), 0x00020000);
    const int chan_b = (int)(chan * 4);
    const float floor_v = (relu & 1) ? 0.f : -__builtin_inff();
    // this lane's channel of fragment register r: cbase(r) + 4 * half
    auto cbase = [&](int r) { return co0 + (r & 3) + 8 * (r >> 2); };
    // offset of (row i, this lane's column) in a plane, or beyond every buffer (loads give 0, stores are dropped)
    unsigned vrow[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int oh_ = oh0 + hy0 + i, ow_ = ow0 + l31;
        vrow[i] = (ow_ < W && oh_ < H) ? (unsigned)(((size_t)oh_ * W + ow_) * 4) : 0x80000000u;
    }
    auto out_off = [&](int i, int od) {     // ... + the plane + this half-wave's channel offset
        return (vrow[i] != 0x80000000u && od < D) ? vrow[i] + (unsigned)((size_t)od * plane * 4) + 4u * half * ochan_b : 0x80000000u;
    };

    if (tid == 0) lds[ZSLOT] = make_uint4(0u, 0u, 0u, 0u);
    float* aff = reinterpret_cast<float*>(&lds[AFF]);
    if (tid < 32) {
        const int co = min(co0 + tid, Cout - 1);
        aff[tid] = scale ? scale[co] : 1.0f;
        aff[32 + tid] = shift ? shift[co] : 0.0f;
        aff[64 + tid] = wunscale[co];
    }
    const __amdgpu_buffer_rsrc_t gres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(GATED ? gate + (size_t)b * Cout * plane : obase), 0,
        GATED ? (int)min((long long)Cout * (long long)gchan_b, 0x7fffffffLL) : 0, 0x00020000);

    // ---- staging plan: this thread owns positions p = tid + 256*i of the halo tile, all 8 channels of a chunk ----
    unsigned poff[NPOS];                  // byte offsets in the tile being FETCHED (beyond the buffer: reads 0)
    auto set_poff = [&](int id0) {
#pragma unroll
        for (int i = 0; i < NPOS; ++i) {
            const int p = tid + NTHR * i;
            const int wx = p % IW;
            int r = p / IW;
            const int hy = r % IH, dz = r / IH;
            const int gw = iw0 + wx, gh = ih0 + hy, gd = id0 + dz;
            const bool ok = (p < CS) && (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W;
            poff[i] = ok ? (unsigned)(((size_t)gd * plane + (size_t)gh * W + gw) * 4) : 0x80000000u;
        }
    };
    auto load_in = [&](int ch, int i) {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ires, (int)poff[i], ch * chan_b, 0));
    };
    const int wlane = (half * Cout + min(co0 + l31, Cout - 1)) * 16;
    auto load_a = [&](int g, int c) {
        return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(wres, wlane, g * wstep + c * 2 * Cout * 16, 0));
    };
    float rin[NQ];
    auto rin_max = [&]() {
        float m = 0.f;
#pragma unroll
        for (int q = 0; q < NQ; ++q) m = fmaxf(m, fabsf(rin[q]));
        return wave_max_bits(__float_as_uint(m));
    };
    auto agreed_exponent = [&](int e_floor) {
        const uint4 w0 = lds[MSLOT], w1 = lds[MSLOT + 1];
        const unsigned m = max(max(max(w0.x, w0.y), max(w0.z, w0.w)), max(max(w1.x, w1.y), max(w1.z, w1.w)));
        return max(e_floor, (int)(m >> 23));
    };
    // split position slot i of the prefetched chunk (nl live channels) with scale sc into image `buf`
    auto split_pos = [&](int i, int nl, float sc, int buf) {
        const int p = tid + NTHR * i;
        if (p >= CS) return;
        unsigned hh[4], ll[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float x0 = (2 * c < nl) ? rin[(2 * c) * NPOS + i] : 0.f;
            const float x1 = (2 * c + 1 < nl) ? rin[(2 * c + 1) * NPOS + i] : 0.f;
            split2_pk_f16(x0 * sc, x1 * sc, hh[c], ll[c]);
        }
        lds[buf * IMG + p] = make_uint4(hh[0], hh[1], hh[2], hh[3]);
        lds[buf * IMG + CS + p] = make_uint4(ll[0], ll[1], ll[2], ll[3]);
    };

    // ---- the first chunk of the first tile: fetched, agreed on and staged in the open ----
    int nlive = min(8, Cin);
    uint4 aq[AR][NC];
#pragma unroll
    for (int k = 0; k < AP; ++k)
#pragma unroll
        for (int c = 0; c < NC; ++c) aq[k][c] = load_a(min(k, G - 1), c);
    set_poff(td0 * TD - 1);
#pragma unroll
    for (int q = 0; q < NQ; ++q) rin[q] = load_in(min(q / NPOS, nlive - 1), q % NPOS);
    {
        const unsigned wm = rin_max();
        if (lane == 0) reinterpret_cast<unsigned*>(&lds[MSLOT])[wave] = wm;
    }
    __syncthreads();
    int e_stage = agreed_exponent(E_MIN);              // exponent the chunk in the current image is scaled for
    int e_run = e_stage;                               // running maximum of the tile being computed (monotone)
    {
        const float sc = __uint_as_float((unsigned)(127 + E_ONE - e_stage) << 23);
#pragma unroll
        for (int i = 0; i < NPOS; ++i) split_pos(i, nlive, sc, 0);
    }
    __syncthreads();
    int cur = 0;

    f32x16 acc[NT];
    for (int it = 0; it < ntile; ++it) {
        const int od = (td0 + it) * TD + dzw;
        const bool more_tile = it + 1 < ntile;
        SS_T(ti0);
        int e_cur = e_stage;
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

        SS_TADD(t_init, ti0);
        for (int ci0 = 0, g0 = 0; ci0 < Cin; ci0 += 8, g0 += KSTEPS) {
            SS_T(tc0);
#ifdef SS_TIMING
            long long tm0 = 0, tm1 = 0;
#endif
            // image `cur` holds chunk ci0 of this tile, scaled for e_stage
            if (e_stage != e_cur) {                    // wave-uniform; exact power-of-two rescale
                const float ratio = __uint_as_float((unsigned)max(127 + e_cur - e_stage, 0) << 23);
#pragma unroll
                for (int i = 0; i < NT; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][r] *= ratio;
                e_cur = e_stage;
            }
            // what is fetched and staged under this chunk: the next chunk of this tile, or the first of the next tile
            const bool more_chunk = ci0 + 8 < Cin;
            const bool new_tile = !more_chunk && more_tile;
            const bool has_next = more_chunk || more_tile;
            const int nci = more_chunk ? ci0 + 8 : 0;
            const int nlive_next = min(8, Cin - nci);
            if (new_tile) set_poff((td0 + it + 1) * TD - 1);
            if (!has_next) {
#pragma unroll
                for (int i = 0; i < NPOS; ++i) poff[i] = 0x80000000u;
            }
            int e_next = e_stage;
            float sc_next = 1.f;
            const int ib = cur * IMG, nb = cur ^ 1;

            uint4 bq[BR][NC];                          // fragment of (step s, row i) in slot (s * NT + i) % BR
            auto read_b = [&](uint4 (&dst)[NC], int s, int i) {
                const int ta = 2 * s, tb = 2 * s + 1;
                const int offa = ((ta / 9) * IH + (ta / 3) % 3) * IW + ta % 3;
                const int offb = (tb < KT) ? ((tb / 9) * IH + (tb / 3) % 3) * IW + tb % 3 : 0;
                const int slot = lane_pos + i * IW + (half ? offb : offa);
#pragma unroll
                for (int c = 0; c < NC; ++c) dst[c] = lds[(tb >= KT && half) ? ZSLOT : ib + c * CS + slot];
            };
            read_b(bq[0], 0, 0);
            read_b(bq[1], 0, 1);
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) {
                {   // weight fragments AP steps ahead (the same for every tile: the ring wraps)
                    const int gi = g0 + s + AP;
#ifndef SS_ABLP_A
#pragma unroll
                    for (int c = 0; c < NC; ++c) aq[(s + AP) % AR][c] = load_a(gi < G ? gi : min(gi - G, G - 1), c);
#endif
                }
#ifndef SS_ABLP_IN
                if (s < IN_STEPS) {
#pragma unroll
                    for (int q = s * QS; q < (s + 1) * QS; ++q) rin[q] = load_in(nci + min(q / NPOS, nlive_next - 1), q % NPOS);
                }
#endif
                // (unconditional also under the last chunk of the last tile: zeros are staged and never read)
                if (s == MSTEP) {
#ifdef SS_TIMING
                    tm0 = __builtin_readcyclecounter();
#endif
                    const unsigned wm = rin_max();
                    if (lane == 0) reinterpret_cast<unsigned*>(&lds[MSLOT])[wave] = wm;
                    __syncthreads();
                    e_next = agreed_exponent(new_tile ? E_MIN : e_run);
                    sc_next = __uint_as_float((unsigned)(127 + E_ONE - e_next) << 23);
#ifdef SS_TIMING
                    tm1 = __builtin_readcyclecounter();
#endif
                }
#ifndef SS_ABLP_SPLIT
                if (s > MSTEP && s <= MSTEP + NPOS) split_pos(s - MSTEP - 1, nlive_next, sc_next, nb);     // 4 positions over K-steps 9-12
#endif
                uint4 a[NC];
#pragma unroll
                for (int c = 0; c < NC; ++c) a[c] = aq[s % AR][c];
#pragma unroll
                for (int i = 0; i < NT; ++i) {
                    const int n = s * NT + i, n2 = n + 2;              // the fragment read now feeds the MFMAs two rows on
#ifndef SS_ABLP_B
                    if (n2 < KSTEPS * NT) read_b(bq[n2 % BR], n2 / NT, n2 % NT);
#endif
                    const uint4 (&bc)[NC] = bq[n % BR];
                    // cross terms (weight term, activation term), smallest first: lo*hi, hi*lo, hi*hi
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[1]), __builtin_bit_cast(f16x8, bc[0]), acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[0]), __builtin_bit_cast(f16x8, bc[1]), acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[0]), __builtin_bit_cast(f16x8, bc[0]), acc[i], 0, 0, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, NC, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                }
                __builtin_amdgcn_sched_barrier(0);     // keep each step's loads and staging inside the step
            }
            if (has_next) {
                nlive = nlive_next;
                e_stage = e_next;
                e_run = e_next;
            }
            SS_T(te0);
            __syncthreads();                           // the other image is complete; this one is free
            cur ^= 1;
#ifdef SS_TIMING
            t_a += tm0 - tc0; t_mid += tm1 - tm0; t_b += te0 - tm1; t_end += (long long)__builtin_readcyclecounter() - te0;
#endif
        }

        // ---- epilogue: 32x32 D layout (col = lane & 31 = output column, row = channel).  A partial sum of the same
        // convolution (RES_PRE) joins the accumulator here, before the affine: out = gate * relu(scale * (partial + conv) + shift) ----
        const float acc_unscale = __uint_as_float((unsigned)(127 - E_ONE + e_cur) << 23);
        SS_T(tp0);
        constexpr int EG = GATED ? 4 : 8;      // fragment rows whose side inputs are fetched together (registers: 2 x EG x NT when gated)
#pragma unroll
        for (int r0 = 0; r0 < 16; r0 += EG) {
            float rv[EG][NT], gv[EG][NT];
            if (residual != nullptr) {                             // one uniform branch per group, none per element
#pragma unroll
                for (int q = 0; q < EG; ++q)
#pragma unroll
                    for (int i = 0; i < NT; ++i)
                        rv[q][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                                 rres, (int)((cbase(r0 + q) + 4 * half < Cout) ? out_off(i, od) : 0x80000000u),
                                                                 cbase(r0 + q) * (int)ochan_b, 0));
            } else {
#pragma unroll
                for (int q = 0; q < EG; ++q)
#pragma unroll
                    for (int i = 0; i < NT; ++i) rv[q][i] = 0.f;
            }
            if (GATED) {
                // the gate does not depend on depth, i.e. on the tile: left alone, the compiler hoists these 64 loads out of
                // the tile loop and keeps 64 registers for them (spills).  An opaque zero ties the addresses to the iteration.
                unsigned tie = 0;
                asm volatile("" : "+v"(tie));
#pragma unroll
                for (int q = 0; q < EG; ++q)
#pragma unroll
                    for (int i = 0; i < NT; ++i)
                        gv[q][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                                 gres, (int)(((cbase(r0 + q) + 4 * half < Cout) ? vrow[i] + 4u * half * gchan_b : 0x80000000u) + tie),
                                                                 cbase(r0 + q) * (int)gchan_b, 0));
            }
#pragma unroll
            for (int q = 0; q < EG; ++q) {
                const int r = r0 + q, cb = cbase(r);
                const bool cok = cb + 4 * half < Cout;
                const float un = aff[64 + cb - co0 + 4 * half] * acc_unscale;   // powers of two: exact
                const float sc = aff[cb - co0 + 4 * half], sh = aff[32 + cb - co0 + 4 * half];
#pragma unroll
                for (int i = 0; i < NT; ++i) {
                    float a0 = acc[i][r] * un;
                    if (res_pre) a0 = ss::add_rn(a0, rv[q][i]);
                    float v = ss::add_rn(ss::mul_rn(a0, sc), sh);
                    if (!res_pre) v = ss::add_rn(v, rv[q][i]);
                    v = fmaxf(v, floor_v);
                    if (GATED) v = ss::mul_rn(gv[q][i], v);                // channelAtt gate, broadcast over D
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ores, (int)(cok ? out_off(i, od) : 0x80000000u),
                                                          cb * (int)ochan_b, 0);
                }
            }
        }
        SS_TADD(t_epi, tp0);
    }
#ifdef SS_TIMING
    if (tid == 0 && blockIdx.x < 4096 && blockIdx.y == 0 && blockIdx.z == 0) {
        unsigned long long* d = ss_dbgp_t + blockIdx.x * 8;
        d[0] = (unsigned long long)((long long)__builtin_readcyclecounter() - t_start);
        d[1] = t_a; d[2] = t_mid; d[3] = t_b; d[4] = t_end; d[5] = t_epi; d[6] = t_init; d[7] = ntile;
    }
#endif
}

int cu_count() {
    static const int cus = [] {
        int dev = 0, n = 256;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        return n;
    }();
    return cus;
}

// depth tiles per workgroup: as many as leave one workgroup per CU (the longest pipelines that still fill the chip)
int tiles_per_workgroup(int B, int D, int H, int W, int Cout) {
    const int tiles_w = ss::ceil_div(W, 32), tiles_h = ss::ceil_div(H, TH), tiles_d = ss::ceil_div(D, TD);
    const long long columns = (long long)tiles_w * tiles_h * ss::ceil_div(Cout, 32) * B;
    int groups = tiles_d;                                      // depth groups: the smallest divisor of tiles_d that fills the chip
    for (int g = tiles_d; g >= 1; --g)
        if (tiles_d % g == 0 && columns * g >= cu_count()) groups = g;
    return tiles_d / groups;
}

template <bool GATED, bool RES_PRE>
int launch_p(const float* in, const void* wsplit, const float* scale, const float* shift, const float* residual,
             const float* gate, float* out, int B, int Cin, int D, int H, int W, int Cout, int relu, hipStream_t st) {
    const int tiles_w = ss::ceil_div(W, 32), tiles_h = ss::ceil_div(H, TH), tiles_d = ss::ceil_div(D, TD);
    auto kern = conv3d_f16p<GATED, RES_PRE>;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                       (int)LDS_BYTES);
    if (attr != hipSuccess) { ss::note_hip_error(attr); return SS_ERR_LAUNCH; }
    static const int tpw_forced = getenv("SS_CONV_TPW") ? atoi(getenv("SS_CONV_TPW")) : 0;
    int tpw = tiles_per_workgroup(B, D, H, W, Cout);
    if (tpw_forced > 0) tpw = std::min(tpw_forced, tiles_d);
    const long long nx = (long long)tiles_w * tiles_h * ss::ceil_div(tiles_d, tpw);
    if (nx > 0x7fffffffLL || B > 65535) return SS_ERR_UNSUPPORTED;
    dim3 grid((unsigned)nx, ss::ceil_div(Cout, 32), B);
    hipLaunchKernelGGL(kern, grid, dim3(NTHR), LDS_BYTES, st, in, reinterpret_cast<const uint4*>(wsplit), scale, shift, residual,
                       gate, out, Cin, D, H, W, Cout, tiles_w, tiles_h, tiles_d, tpw, relu);
    return ss::check_launch();
}

}  // namespace

// A pipeline of fewer than three tiles does not pay for its set-up and leaves the chip unevenly filled (hourglass2.conv2,
// 64 -> 64 on [12,128,128]: 141 us here with one tile per workgroup, 120 us on the tiled kernel).
bool ss_conv3d_f16p_applicable(int B, int D, int H, int W, int Cout) { return tiles_per_workgroup(B, D, H, W, Cout) >= 3; }

// Called by conv3d_bf16s.hip's dispatcher for nterms = 19, stride 1, layers large enough (see above).
int ss_conv3d_f16p_launch(const float* in, const void* wsplit, const float* scale, const float* shift, const float* residual,
                          const float* gate, float* out, int B, int Cin, int D, int H, int W, int Cout, int relu,
                          hipStream_t st) {
    const bool res_pre = (relu & 2) != 0 && residual != nullptr;
#define SS_P(G, R) launch_p<G, R>(in, wsplit, scale, shift, residual, gate, out, B, Cin, D, H, W, Cout, relu, st)
    if (gate != nullptr) return res_pre ? SS_P(true, true) : SS_P(true, false);
    return res_pre ? SS_P(false, true) : SS_P(false, false);
#undef SS_P
}
