// 3x3x3 Conv3d (+ folded BatchNorm, residual, ReLU) with fp32 operands emulated on the bf16 matrix
// core: "split-bf16" engine for the same layers as conv3d.hip (gfx950 v_mfma_f32_32x32x16_bf16).
//
// Every fp32 value is split exactly into three bf16 terms x = hi + mid + lo (24+ significand bits);
// a product a*b is the sum of its cross terms.  Keeping the six terms down to 2^-24 relative
// (hh, hm, mh, mm, hl, lh) on bf16 MFMAs with fp32 accumulation gives a GEMM whose measured error
// (tools/exp_split_bf16.hip on MI355X: 1.1e-7 of sum|a*b| at K = 864..3456) is BELOW that of the exact
// fp32 MFMA (1.8e-7), at 6/16 of its matrix-core time; the three-term form (hh, hm, mh: 3/16 of the
// time) measures 3-6e-7.  NTERMS selects the form; products of bf16 are exact in fp32, so the only
// rounding is the accumulation, as in any fp32 GEMM.
//
// GEMM mapping: M = 32 output channels, N = 32 consecutive output columns of one row (lanes),
// K-step of 16 = lanes 0-31: 8 input channels of tap 2s, lanes 32-63: the same 8 channels of tap
// 2s+1 (27 taps -> 14 steps, the 28th half is zero).  The bf16 B operand wants 8 consecutive k per
// lane, so the LDS halo tile is CHANNEL-INNERMOST: [term][position][8 channels] bf16 = one 16-byte
// slot per (term, position); a wave's fragment read is 64 consecutive slots (conflict-free
// ds_read_b128).  The NCDHW fp32 input is transposed/split while it is staged: each thread owns
// whole positions, loads their 8 channels (coalesced along W per channel, register-prefetched one
// chunk ahead), splits them and writes three 16-byte slots.  Weights are pre-split and pre-packed in
// fragment order by ss_pack_conv3d_weights_bf16s and streamed from L2 (every wave of a workgroup
// reads the same 16 B per lane per term and step; LDS is left to the activations).
#include <algorithm>
#include <stdlib.h>
#include <type_traits>

#include "common.h"
#include "split_f16.h"

// Phase stamps of a workgroup: no-ops here.  tools/conv_timing.hip defines SS_STAMP / SS_STAMP_STEPS_* before including this
// file to build the instrumented library tools/wg_phases.py reads (tools/build_timing.sh); the product build has no
// instrumentation in it.
#ifndef SS_STAMP
#define SS_STAMP(k) do {} while (0)
#define SS_STAMP_STEPS_BEGIN() do {} while (0)
#define SS_STAMP_STEPS_END() do {} while (0)
#define SS_STAMP_FINISH() do {} while (0)
#endif

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) short;

constexpr int KSTEPS = 14;        // ceil(27 taps / 2): the 3-D kernels (BCfg::KSTEPS is the general form)
// measured-best settings (each was swept on the bench shapes)
#ifndef SS_IN_STEPS
#define SS_IN_STEPS 10            // K-steps over which the next chunk's input loads are issued
#endif
#ifndef SS_IN_STEPS_S2
#define SS_IN_STEPS_S2 10         // ... in the stride-2 form (tools/build_variant.sh sweeps: 5: 94.0, 7: 89.1, 10: 87.9, 14: 86.5 us)
#endif
#ifndef SS_A_AHEAD_S2
#define SS_A_AHEAD_S2 2           // SS_A_AHEAD / SS_ROW_PAIR of the stride-2 form (3, 4 steps ahead / row pairs: all within +-1.5 us of 90)
#endif
#ifndef SS_S2_MS_MIN_WGS
#define SS_S2_MS_MIN_WGS 256      // stride 2: fewest 64-channel workgroups for which the waves split the channels (below: one 32-channel tile per workgroup; 128 on the 128-workgroup layer: 29.5 vs 26.4 us)
#endif
#ifndef SS_ROW_PAIR_S2
#define SS_ROW_PAIR_S2 1
#endif
#ifndef SS_S2_DEINT
#define SS_S2_DEINT 0             // stride 2: the halo rows in LDS de-interleaved by column parity (r06; 1: a thread owns its SLOT's column, 2: a thread owns a column and writes the permuted slot; see lane_pos -- both measured slower than the natural order, 0)
#endif
constexpr int SS_IN_AUX = 0;      // cache policy bits of the activation loads (buffer_load aux: 1 = sc0, 2 = nt)
#ifndef SS_IN_AUX_GATED
#define SS_IN_AUX_GATED 0         // ... of the gated launch (concat_stem), whose inputs -- warped half, partial sum -- are dead after it (nt / sc0+nt measured on the whole step: 479.5 / 476 against 489.6 pairs/s on one box: the launch itself 284 -> 334 / 344 us)
#endif
constexpr int SS_ROW_PAIR = 1;    // rows whose MFMAs alternate; 2 measured 1 % slower: the other wave of the SIMD already fills the gaps
#ifndef SS_A_AHEAD
#define SS_A_AHEAD 2              // K-steps between the load of a weight fragment and its MFMAs
#endif
#ifndef SS_F16_WGS
#define SS_F16_WGS 2              // workgroups per CU the fp16 form is compiled for (3 = 168 VGPRs: spills, +29 %)
#endif

__device__ __forceinline__ unsigned bf16_rne(float x) {       // finite inputs
    unsigned u = __float_as_uint(x);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float bf16_up(unsigned b) { return __uint_as_float(b << 16); }

// x -> (hi, mid, lo) bf16 bit patterns with hi + mid + lo == x up to 2^-25 |x|
__device__ __forceinline__ void split3(float x, unsigned& h, unsigned& m, unsigned& l) {
    h = bf16_rne(x);
    const float r1 = x - bf16_up(h);
    m = bf16_rne(r1);
    const float r2 = r1 - bf16_up(m);
    l = bf16_rne(r2);
}

// the same split for two values at once on gfx950's packed converter: v_cvt_pk_bf16_f32 (RNE) gives
// lo16 = bf16(x0), hi16 = bf16(x1) -- exactly the LDS slot layout -- and the residuals are one v_pk_add_f32
using bf16x2_t = __attribute__((ext_vector_type(2))) __bf16;
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

// KD: kernel depth, 3 (3x3x3) or 1 (3x3 over [B,C,H,W] maps seen as depth-1 volumes: 9 taps, 5 K-steps)
// WSL: 16-byte LDS slots of a chunk's weight fragments (0: every wave fetches its own)
// MS: waves that share a row group and split the workgroup's output channels (1: every wave owns NT rows of all of them)
// XSL: further 16-byte slots behind everything else (the gather form's candidate / weight words)
template <int S, int NT, int TD, int TH, int KD = 3, int LT = 3, int WSL = 0, int MS = 1, int XSL = 0>      // LT: operand terms kept in LDS
struct BCfg {
    static constexpr int KT = KD * 9, KSTEPS = (KT + 1) / 2;
    static constexpr int ID = (TD - 1) * S + KD, IH = (TH - 1) * S + 3, IW = 31 * S + 3;
    static_assert(KD == 3 || (KD == 1 && TD == 1), "2-D form: depth-1 tiles");
    static constexpr int CS = ID * IH * IW;                    // positions in the halo tile
    static constexpr int NPOS = (CS + 255) / 256;              // positions per thread
    // + one all-zero slot (the 28th half-step) + the waves' maxima (f16 form) + the affine of the workgroup's (<= 64) channels
    static constexpr int SLOTS = LT * CS + 2 + 48 + WSL + XSL;
    static constexpr int XS0 = LT * CS + 2 + 48 + WSL;         // first extra slot
    static constexpr size_t LDS_BYTES = (size_t)SLOTS * 16;
    static_assert(TD * TH * MS == 4 * NT && TH % NT == 0 && (MS == 1 || MS == 2), "4 / MS wave groups x NT rows tile TD x TH");
};

#ifndef SS_WLDS_NT4
#define SS_WLDS_NT4 0             // 1: the 4-row tile also takes its weight fragments through LDS (variant builds)
#endif
constexpr bool wlds_form(int S, int NT, int NTERMS, int MT, int KD, bool gather = false) {
    // (the gather form's tile has no LDS left for the 28 KB slab at two workgroups per CU)
    return NTERMS == 19 && S == 1 && (NT == 1 || (SS_WLDS_NT4 && NT == 4 && !gather)) && MT == 1 && KD == 3;       // (NT = 2: 43 B/clk, measured +3 %: left alone)
}
// gather form: per halo position one candidate word and two attention words (this tile's, the next tile's), each thread's
// own positions p = tid + 256 i -> 3 x NPOS x 256 floats
constexpr int gather_slots(bool gather, int S, int TD, int TH, int KD) {
    return gather ? 3 * ((((TD - 1) * S + KD) * ((TH - 1) * S + 3) * (31 * S + 3) + 255) / 256) * 64 : 0;
}
// head form (HEAD of the kernel): the 32 -> 1 head's weight fragments, [3 bf16 terms][2 K-steps][64 lanes] 16-byte slots
constexpr int HEAD_WSLOTS = 3 * 2 * 64;
constexpr int HEAD_PATCH = 6 * 6 * 34;        // a 4 x 4 x 32 tile's contributions to the head's output: its positions and one ring around them

// Chunk-blocked accumulation (ACCB of the kernel) is a property of the LAYER, never of the tile a launch happens to get: the
// tile candidates depend on the batch size, and a pair must get the same bits alone and in a batch
// (test_hot_segment_batch_invariance_at_the_sharded_batch_sizes).  fp16 form only.  It is on for every stride-2 layer, every
// 2-D layer, and the stride-1 3-D layers that at batch 1 are too small for the 4-row tile -- the deep, narrow layers with the
// longest K (conv2 / conv4 of the hourglasses: K = 1728 / 3456): there the second accumulator set fits the registers of the 1- and
// 2-row tiles they run on at small batch; when a larger batch moves them onto the 4-row tile that variant spills 16 registers
// (+7 %, measured on the stem shape).  The big 4-row layers (concat_stem, classif.0, hourglass2.conv2) keep the single chain.
#ifndef SS_ACC_BLOCKED
#define SS_ACC_BLOCKED 1                  // 0: single chains everywhere (tools/build_variant.sh, for A/B measurements)
#endif

// GATED: the channelAtt gate is fused into the epilogue (only concat_stem has one, so its launches also carry
// their own kernel symbol in a profile: conv3d_bf16s<..., true>)
// MT: 32-channel output tiles per wave (the activation fragments of a row then feed MT x 6 MFMAs: used by the stride-2
// layers, whose staging is 8x dearer per MFMA and whose 2-4 output tiles would otherwise each stage the same input)
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

// MS = 2 (stride-2 layers): the four waves are 2 row groups x 2 halves of the workgroup's 64 * MT output channels.  A wave then
// fetches the weight fragments of ITS channels only (the vector L1 carried every fragment four times per workgroup: 32 KB
// per K-step and CU beside 8 KB of activations, 640 clocks at its 64 B/clk for 384 clocks of MFMA issue -- tools/wg_phases_s2.py)
// and reads the activation fragments of two rows from LDS instead of one.
//
// GATHER (r05; SURVEY.md section 8 f1, second half: sparse concat -> x att -> concat_stem in ONE launch): `in` is the 2-D right
// feature map [B,Cin,H,W]; the operand of the convolution, x[c, j, h, w] = att[j,h,w] * in[c, h, w - cand[j,h,w]] (zero where
// the column leaves the image) -- the warped half of the reference's sparse concat volume times the attention weights
// (models/SemStereo.py:241-244, 316-318; models/submodule.py:265-288 for INTEGER candidates, which is what :299-305 produce) --
// is formed while the halo tile is staged: no [B,Cin,nd,H,W] volume exists.  Per halo position the candidate and the weight
// are fetched once per TILE by LDS-DMA (no registers: each thread's own positions, parked in LDS) a chunk ahead of their
// first use; the gathered loads take the place of the volume loads one for one (same prefetch registers, same slices).
// D = the number of candidates.  Cin % 8 == 0 and Cin >= 16 (the look-ahead needs two chunks per tile).
//
// HEAD (r05, late): nn.Sequential(convbn_3d(32,32,3,1,1), ReLU, Conv3d(32,1,3,p1)) -- `classif` / `classif_att_`, models/SemStereo.py:228-234 -- in
// ONE pass over the volume: the 32-channel intermediate never leaves the CU.  After a tile's K loop its y = ReLU(BN(conv)) sits in the
// accumulators in exactly the B-operand layout of the next contraction (lane = position, 8 consecutive registers = 8 channels of a
// K-step), so t[tap][position] = sum_c w2[c][tap] y[c][position] is 12 more MFMAs per row (27 taps as matrix rows, three bf16 terms, six
// products); the 27 shifted sums out[q] = sum_tap t[tap][q + tap - 1] are done in three deterministic stages: over kh in registers
// (a wave owns 4 rows of one plane, a lane half whole kh-triples), over (kd, kw) through a 30 KB LDS buffer that reuses the dead
// activation tile -- giving the tile's contribution to the 6 x 6 x 34 output positions it touches, written to `out` as a PATCH
// [B][tiles][6][6][34] -- and over the <= 8 tiles that touch an output position in classifier_patch_sum_kernel.  201 MB written +
// 288 MB read per classifier become 15 + 15 MB; the head's own launch disappears.  `cand` carries the head's fragments
// (ss_pack_classifier_head_weights), `out` the patch buffer.
template <int S, int NT, int TD, int TH, int NTERMS, bool GATED, int MT, int KD = 3, int MS = 1, bool ACCB = false, bool GATHER = false, bool HEAD = false>
__global__ __launch_bounds__(256, (NTERMS == F16X3) ? SS_F16_WGS : 2) void conv3d_bf16s(const float* __restrict__ in, const uint4* __restrict__ wsplit,
                                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                                        const float* __restrict__ residual, const float* __restrict__ gate,
                                                        float* __restrict__ out,
                                                        int Cin, int D, int H, int W, int Cout, int Do, int Ho, int Wo,
                                                        int tiles_w, int tiles_h, int ntiles, int relu,
                                                        const float* __restrict__ cand, const float* __restrict__ catt,
                                                        const float* __restrict__ in2, int bsplit) {
    static_assert(!GATHER || (S == 1 && MT == 1 && KD == 3 && MS == 1), "gather form: plain stride-1 3-D tiles");
    static_assert(!HEAD || (S == 1 && NT == 4 && TD == 4 && TH == 4 && MT == 1 && KD == 3 && MS == 1 && !GATED && !GATHER && NTERMS == F16X3),
                  "head form: the 4 x 4 x 32 tile of the fp16 engine, one wave per plane");
    constexpr bool F16 = (NTERMS == F16X3);
    constexpr int NC = (NTERMS == 6) ? 3 : 2;                  // operand terms actually read
    constexpr int NCW = F16 ? 2 : 3;                           // terms in the packed weights
    // Small tiles in the fp16 form: a K-step has only 3 MFMAs per row and wave, and eight waves per CU each re-reading the
    // step's 2 KB of weight fragments every 96-192 cycles ask the vector L1 for 43-85 B/clk of its 64.  The chunk's fragments
    // (14 steps x 2 terms, 28 KB) are then brought into LDS once per workgroup by LDS-DMA loads (no registers) and read
    // from there by the four waves (deconv3d_bf16s.hip has the measurement: -11 %).
    constexpr bool WLDS = wlds_form(S, NT, NTERMS, MT, KD, GATHER);
    static_assert(!ACCB || NTERMS == F16X3, "chunk-blocked accumulation: fp16 form only");
    static_assert(!WLDS || MS == 1, "the LDS copy of the weights is one channel tile's");
    using C = BCfg<S, NT, TD, TH, KD, NC, WLDS ? ((KD * 9 + 1) / 2) * 2 * 64 : 0, MS, gather_slots(GATHER, S, TD, TH, KD) + (HEAD ? HEAD_WSLOTS : 0)>;
    constexpr int WL = NC * C::CS + 2 + 48;                    // first slot of the weight fragments
    constexpr int KSTEPS = C::KSTEPS;                          // shadows the 3-D constant
    constexpr int ZSLOT = NC * C::CS;                          // the all-zero slot; ZSLOT + 1: the four waves' maxima
    extern __shared__ __attribute__((aligned(16))) uint4 lds[];   // [NC][CS] slots + zero slot + maxima

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    SS_STAMP(0);
    const int l31 = lane & 31, half = lane >> 5;
    // PERSISTENT workgroups: a workgroup walks the tiles blockIdx.x, blockIdx.x + gridDim.x, ... of its (channel tile, batch
    // element).  While the LAST chunk of a tile is multiplied, the prefetch registers -- idle there until now -- fetch the
    // FIRST chunk of the next tile, so only the very first tile of a workgroup pays the exposed round trip to HBM that used
    // to open every workgroup's life (12 k of ~100 k cycles, tools/wg_phases.py); the epilogue's stores then drain under the
    // next tile's first K-steps.
    const int co0 = blockIdx.y * 32 * MT * MS;                  // the workgroup's first channel
    const int wrow = wave / MS;                                 // row group of this wave
    const int cow = co0 + (wave % MS) * 32 * MT;                // this wave's first channel
    const int b = blockIdx.z;
    const int dzw = (wrow * NT) / TH, hy0 = (wrow * NT) % TH;
    // Stride 2: with a row's columns in their natural order the 32 lanes of a fragment read are 32 bytes apart, a 2-way bank conflict
    // per ds_read_b128 lane group (r05: SQ_LDS_BANK_CONFLICT 5.3 M cycles on the largest layer, the only conv family with any).  DEINT
    // (r06): a row of the halo tile is stored even columns first (33 slots), then odd (32), and a thread OWNS the position of its slot
    // (make_poff maps slot -> column), so nothing is permuted on the way in: tap kw of output column l31 is input column 2 l31 + kw =
    // slot l31 (kw 0), 33 + l31 (kw 1), l31 + 1 (kw 2) -- 64 consecutive slots per fragment read as in the stride-1 forms.  (r03 measured
    // a de-interleaved build of the MT = 2 form as neutral, 88.7 / 89.6 us against 89.0 / 86.4; this is the MS = 2 form with scalar
    // addressing.)  Measured r06, same box, interleaved (profiles/r06_e_ab_conv_s2_deint.txt): SQ_LDS_BANK_CONFLICT 5.3 M -> 0, but the
    // largest layer 88.0 -> 94.8 us and the [16,64,64] one 26.4 -> 32.1: the lanes of a staging load are then 8 bytes apart (two
    // instructions per cache line instead of one) and that costs more than the conflicts did (wait_inst_lds was 1.4 % of wave cycles).
    // DEINT_W (SS_S2_DEINT=2): columns stay with their natural owners (coalesced loads) and the permutation is applied to the LDS WRITE
    // address instead (2-way conflicts on 3 writes per position instead of on every fragment read).
    constexpr bool DEINT_W = (S == 2) && SS_S2_DEINT == 2;
    constexpr bool DEINT = (S == 2) && SS_S2_DEINT != 0;
    const int lane_pos = (dzw * S * C::IH + hy0 * S) * C::IW + (DEINT ? l31 : l31 * S);     // slot of this lane's first row, tap (0,0,0)
    auto tile_origin = [&](int tile, int& ow0, int& oh0, int& od0) {
        int t = tile;
        const int tw = t % tiles_w; t /= tiles_w;
        const int th = t % tiles_h; t /= tiles_h;
        ow0 = tw * 32; oh0 = th * TH; od0 = t * TD;
    };
    // relu bit 0: ReLU; bit 1: `residual` is added BEFORE the affine: a partial sum of the same convolution computed
    // elsewhere (stem_left.hip).  It joins the accumulator in the epilogue, fetched together with the gate (as initial
    // accumulators its 64 loads per lane were 14 k cycles of every workgroup's prologue: tools/wg_phases.py).
    const bool res_pre = (relu & 2) != 0 && residual != nullptr;
    // bit 2: CHANNELS-LAST output [Do][Ho][Wo][Cout] (the hand-off to the 32 -> 1 head of the same classifier, conv3d_head.hip,
    // which wants 8 consecutive channels of a position per lane): a lane's 4 consecutive channels of a fragment register
    // group go out as one 16-byte store instead of four 4-byte ones.  Plain stride-1 form only, Cout % 8 == 0.
    constexpr bool CAN_CL = !GATED && MT == 1 && S == 1 && KD == 3;
    const bool out_cl = CAN_CL && (relu & 4) != 0;
    // f16 form: float[Cout] of 2^-(weight scale of the channel), stored behind the packed terms
    const float* wunscale = reinterpret_cast<const float*>(
        reinterpret_cast<const char*>(wsplit) + (size_t)((Cin + 7) / 8) * C::KSTEPS * ((NTERMS == F16X3 ? 2 : 3) * 2 * Cout * 16));
    // Output-side addressing (partial sum in, result out, gate, residual): buffer descriptors per batch element with a
    // 32-bit per-lane offset per row and a scalar offset per channel, positions outside the volume parked beyond the
    // buffer (loads give 0, stores are dropped): no 64-bit per-lane arithmetic and no branches around the 64 stores of a
    // lane (they were ~35 instructions and 3.5 branches per store).
    const size_t out_plane = (size_t)Ho * Wo;
    const unsigned ochan_b = (unsigned)((size_t)Do * out_plane * 4), gchan_b = (unsigned)(out_plane * 4);
    const int obytes = (int)min((long long)Cout * (long long)ochan_b, 0x7fffffffLL);
    unsigned vout[NT], vgate[NT];
    auto set_outputs = [&](int tile) {
        int ow0, oh0, od0;
        tile_origin(tile, ow0, oh0, od0);
        const int ow_ = ow0 + l31, od_ = od0 + dzw;
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int oh_ = oh0 + hy0 + i;
            const bool ok = ow_ < Wo && od_ < Do && oh_ < Ho;
            vout[i] = ok ? (unsigned)((((size_t)od_ * Ho + oh_) * Wo + ow_) * 4) + 4u * half * ochan_b : 0x80000000u;
            if (CAN_CL && out_cl) vout[i] = ok ? (unsigned)((((size_t)od_ * Ho + oh_) * Wo + ow_) * Cout * 4) + 16u * half : 0x80000000u;
            vgate[i] = ok ? (unsigned)(((size_t)oh_ * Wo + ow_) * 4) + 4u * half * gchan_b : 0x80000000u;
        }
    };
    // this lane's channel of fragment register r of output tile mt: cbase(mt, r) + 4 * half
    auto cbase = [&](int mt, int r) { return cow + mt * 32 + (r & 3) + 8 * (r >> 2); };
    f32x16 acc[MT * NT];                  // index mt * NT + row

    const size_t in_plane = (size_t)H * W, chan = GATHER ? in_plane : (size_t)D * in_plane;     // (gather: channels of a 2-D map)
    // (in2: batch elements bsplit, bsplit + 1, ... of the launch come from a SECOND input tensor -- the left and right views of
    // concat_feature in one launch without a torch.cat in front, ss_conv2d_bf16s_pair_fwd)
    const float* inb = (in2 != nullptr && b >= bsplit) ? in2 + (size_t)(b - bsplit) * Cin * chan : in + (size_t)b * Cin * chan;

    // staging plan of a tile: this thread owns positions p = tid + 256*i of the halo tile, all 8 channels
    auto make_poff = [&](int tile, unsigned (&po)[C::NPOS]) {
        int ow0, oh0, od0;
        tile_origin(tile, ow0, oh0, od0);
        const int iw0 = ow0 * S - 1, ih0 = oh0 * S - 1, id0 = od0 * S - KD / 2;
#pragma unroll
        for (int i = 0; i < C::NPOS; ++i) {
            const int p = tid + 256 * i;
            const int sx = p % C::IW;
            const int wx = (DEINT && !DEINT_W) ? (sx < (C::IW + 1) / 2 ? 2 * sx : 2 * (sx - (C::IW + 1) / 2) + 1) : sx;      // (DEINT: slot -> column)
            int r = p / C::IW;
            const int hy = r % C::IH;
            const int dz = r / C::IH;
            const int gw = iw0 + wx, gh = ih0 + hy, gd = id0 + dz;
            const bool ok = (p < C::CS) && (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W;
            // halo positions outside the volume get an offset beyond the buffer's num_records: the buffer
            // load returns 0 for them, no select needed
            po[i] = ok ? (unsigned)(((size_t)gd * in_plane + (size_t)gh * W + gw) * 4) : 0x80000000u;
        }
    };
    unsigned poff[C::NPOS];
    if constexpr (!GATHER) make_poff(blockIdx.x, poff);
    // ---- gather form: candidates and attention weights of a tile's halo positions, parked in LDS ----
    float* gcand = reinterpret_cast<float*>(&lds[C::XS0]);                       // [NPOS * 256]: candidates of the tile set up next
    float* gatt = gcand + C::NPOS * 256;                                        // [2][NPOS * 256]: weights of this tile / the next
    const __amdgpu_buffer_rsrc_t cres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(GATHER ? cand + (size_t)b * D * in_plane : in), 0, GATHER ? (int)min((long long)D * (long long)in_plane * 4, 0x7fffffffLL) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t tres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(GATHER ? catt + (size_t)b * D * in_plane : in), 0, GATHER ? (int)min((long long)D * (long long)in_plane * 4, 0x7fffffffLL) : 0, 0x00020000);
    // request (LDS-DMA, 4 bytes per lane, no registers) the candidate and weight words of `tile`'s halo positions; positions
    // outside the volume -- and every position of a tile that does not exist -- read beyond the buffers and get zeros
    auto gather_request = [&](int tile, int buf) {
        unsigned po[C::NPOS];
        make_poff(tile, po);                                   // offsets into [D][H][W]: the candidates' and the weights' own layout
        const unsigned dead = tile < ntiles ? 0u : 0x80000000u;
        const int wbase = __builtin_amdgcn_readfirstlane(wave * 64);
#pragma unroll
        for (int i = 0; i < C::NPOS; ++i) {
            lds_dma4(cres, gcand + 256 * i + wbase, (int)(po[i] | dead), 0);
            lds_dma4(tres, gatt + buf * (C::NPOS * 256) + 256 * i + wbase, (int)(po[i] | dead), 0);
        }
    };
    // offsets of the gathered elements of `tile` (its candidates have landed in gcand): row h, column w - candidate of the 2-D map
    auto gather_offsets = [&](int tile, unsigned (&po)[C::NPOS]) {
        __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0): the LDS-DMA words have landed
        int ow0, oh0, od0;
        tile_origin(tile, ow0, oh0, od0);
        const int iw0 = ow0 - 1, ih0 = oh0 - 1, id0 = od0 - 1;
#pragma unroll
        for (int i = 0; i < C::NPOS; ++i) {
            const int p = tid + 256 * i;
            const int wx = p % C::IW;
            int r = p / C::IW;
            const int hy = r % C::IH;
            const int dz = r / C::IH;
            const int gw = iw0 + wx, gh = ih0 + hy, gd = id0 + dz;
            const int col = gw - (int)lds_read4(gcand + p);      // integer candidates (the ABI's contract)
            const bool ok = (p < C::CS) && (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W &&
                            (unsigned)col < (unsigned)W && tile < ntiles;
            po[i] = ok ? (unsigned)(((size_t)gh * W + col) * 4) : 0x80000000u;
        }
    };
    if constexpr (GATHER) {
        gather_request(blockIdx.x, 0);
        gather_offsets(blockIdx.x, poff);
    }
    // input prefetch registers, flattened q = c * NPOS + i, loaded in KSTEPS slices spread over the
    // K-steps of the previous chunk so that a wait for a weight fragment never drains them all
    constexpr int NQ = 8 * C::NPOS;
    // ... all of them within the first SS_IN_STEPS steps: the split phase at the end of the chunk waits
    // for the youngest slice, which needs a few K-steps (HBM latency) to land
    constexpr int IN_STEPS = ((S == 2 ? SS_IN_STEPS_S2 : SS_IN_STEPS) < KSTEPS) ? (S == 2 ? SS_IN_STEPS_S2 : SS_IN_STEPS) : KSTEPS;
    constexpr int QS = (NQ + IN_STEPS - 1) / IN_STEPS;
    float rin[NQ];
    int nlive = min(8, Cin), nlive_next = 8;                 // channels that exist in the staged / prefetched chunk
    if (tid == 0) lds[ZSLOT] = make_uint4(0u, 0u, 0u, 0u);
    // per-channel epilogue constants, fetched now and parked in LDS: read after the K loop they cost two exposed round
    // trips to L2/HBM per workgroup (tools/wg_phases.py).  aff[c] = scale, aff[64 + c] = shift, aff[128 + c] = 2^-(weight scale)
    float* aff = reinterpret_cast<float*>(&lds[ZSLOT + 2]);
    if constexpr (HEAD) {             // the head's weight fragments: parked in LDS once per workgroup
        const uint4* hw = reinterpret_cast<const uint4*>(cand);
        for (int i = tid; i < HEAD_WSLOTS; i += 256) lds[C::XS0 + i] = hw[i];
    }
    if (tid < 32 * MT * MS) {
        const int co = min(co0 + tid, Cout - 1);
        aff[tid] = scale ? scale[co] : 1.0f;
        aff[64 + tid] = shift ? shift[co] : 0.0f;
        aff[128 + tid] = F16 ? wunscale[co] : 1.0f;            // 2^-(weight scale of the channel)
    }

    // weight fragments: [global K-step g = blk*14 + s][term][half][Cout][8 bf16] as uint4 slots; lanes of
    // output channels beyond Cout read a clamped (valid) column and are dropped in the epilogue
    // Both operands are fetched with buffer loads: wave-uniform base (SGPR descriptor) + uniform
    // scalar offset + one 32-bit per-lane offset, so no 64-bit per-lane addresses are kept live.
    int wlane[MT];                                                            // byte offset of this lane's column, per output tile
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) wlane[mt] = (half * Cout + min(cow + mt * 32 + l31, Cout - 1)) * 16;
    const int wstep = NCW * 2 * Cout * 16;                                    // bytes per K-step
    const __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint4*>(wsplit), 0, (int)min((long long)((Cin + 7) / 8) * KSTEPS * wstep, 0x7fffffffLL), 0x00020000);
    const __amdgpu_buffer_rsrc_t ires = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(inb), 0, (int)min((long long)Cin * (long long)chan * 4, 0x7fffffffLL), 0x00020000);
    const int chan_b = (int)(chan * 4);                                       // bytes per input channel
    auto load_a = [&](int g, int c, int mt) {
        return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(wres, wlane[mt], g * wstep + c * 2 * Cout * 16, 0));
    };
    auto load_in = [&](int ch, int i) {                                       // channel ch (absolute), position slot i
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ires, (int)poff[i], ch * chan_b, GATED ? SS_IN_AUX_GATED : SS_IN_AUX));
    };
    const int G = ((Cin + 7) / 8) * KSTEPS;
    // Ring of weight fragments, AP steps ahead: vmcnt retires in order, so a wait for a fragment also
    // waits for every input (HBM) load issued before it -- the distance must cover HBM latency, not L2's.
    constexpr int AP = (S == 2) ? SS_A_AHEAD_S2 : SS_A_AHEAD, AR = AP + 1;
    uint4 aq[AR][MT][NC];                                      // aq[s % AR] = fragments of step s of the chunk
    if (!WLDS) {
#pragma unroll
        for (int k = 0; k < AP; ++k)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int c = 0; c < NC; ++c) aq[k][mt][c] = load_a(min(k, G - 1), c, mt);
    }
    {   // first chunk: plain load of every slice
#pragma unroll
        for (int q = 0; q < NQ; ++q) rin[q] = load_in(min(q / C::NPOS, nlive - 1), q % C::NPOS);
    }
    // f16 form: block-floating scale of the staged chunk.  e_cur = biased exponent the accumulators are scaled for
    // (scale 2^(E_ONE - e)); e_run = that of the running maximum of the tile (monotone: the accumulators only scale DOWN
    // after the first chunk, so they cannot overflow)
    int e_cur = E_ONE, e_run = E_MIN;
    // gather form: the prefetched chunk times the attention weights of its positions (models/SemStereo.py:318), rounded to fp32 as
    // the reference's materialised product is
    auto apply_att = [&](int buf) {
#pragma unroll
        for (int i = 0; i < C::NPOS; ++i) {
            const float a = lds_read4(gatt + buf * (C::NPOS * 256) + tid + 256 * i);
#pragma unroll
            for (int c = 0; c < 8; ++c) rin[c * C::NPOS + i] = ss::mul_rn(rin[c * C::NPOS + i], a);
        }
    };
    if constexpr (GATHER) apply_att(0);
    auto publish_max = [&](float m) {                                         // this wave's max(m, |rin|) -> LDS
#pragma unroll
        for (int q = 0; q < NQ; ++q) m = fmaxf(m, fabsf(rin[q]));
        const unsigned wm = wave_max_bits(__float_as_uint(m));
        if (lane == 0) reinterpret_cast<unsigned*>(&lds[ZSLOT + 1])[wave] = wm;
    };
    if (F16) {
        publish_max(0.f);
        __syncthreads();
    }
    SS_STAMP(1);

    int gbuf = 0;                                               // gather form: which half of gatt holds this tile's weights
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, gbuf ^= 1) {
    const bool has_next = tile + (int)gridDim.x < ntiles;
#pragma unroll
    for (int i = 0; i < MT * NT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    e_cur = E_ONE; e_run = E_MIN;
    nlive = min(8, Cin);
    for (int ci0 = 0, g0 = 0; ci0 < Cin; ci0 += 8, g0 += KSTEPS) {
        // ---- split + transpose: registers -> [term][position][8 ch] ----
        if constexpr (WLDS) {         // this chunk's weights: wave w issues the (K-step, term) pairs i = w, w + 4, ...; lane -> (half, channel)
            // (the wave index as a scalar, opaque once per chunk: the entries' LDS addresses and source offsets are scalar arithmetic
            // done here instead of per-lane loop invariants hoisted out of the chunk loop -- deconv3d_bf16s.hip, r05)
            int wv = __builtin_amdgcn_readfirstlane(wave);
            asm volatile("" : "+s"(wv));
#pragma unroll
            for (int k = 0; k < (KSTEPS * 2 + 3) / 4; ++k) {
                const int i = wv + 4 * k;                      // wave-uniform
                if (i < KSTEPS * 2)
                    lds_dma16(wres, &lds[WL + i * 64], wlane[0], (g0 + i / 2) * wstep + (i & 1) * 2 * Cout * 16);
            }
        }
        float in_scale = 1.f;
        if (F16) {
            const uint4 wm = lds[ZSLOT + 1];
            const int e_new = max(e_run, (int)(max(max(wm.x, wm.y), max(wm.z, wm.w)) >> 23));     // inf/NaN: 255
            e_run = e_new;
            if (e_new != e_cur) {                              // wave-uniform; exact power-of-two rescale
                const float ratio = __uint_as_float((unsigned)max(127 + e_cur - e_new, 0) << 23);
#pragma unroll
                for (int i = 0; i < MT * NT; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][r] *= ratio;
                e_cur = e_new;
            }
            in_scale = __uint_as_float((unsigned)(127 + E_ONE - e_cur) << 23);
        }
        // (a chunk of 8 live channels -- every chunk when Cin % 8 == 0 -- takes the copy of this loop without the
        // per-channel selects: 7 of its 32 VALU instructions per position, ISA count r03)
        auto stage = [&](auto whole_chunk) {
            constexpr bool WHOLE = decltype(whole_chunk)::value;
#pragma unroll
            for (int i = 0; i < C::NPOS; ++i) {
                const int p = tid + 256 * i;
                if (p >= C::CS) continue;
                unsigned hh[4], mm[4], ll[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float x0 = (WHOLE || 2 * c < nlive) ? rin[(2 * c) * C::NPOS + i] : 0.f;
                    const float x1 = (WHOLE || 2 * c + 1 < nlive) ? rin[(2 * c + 1) * C::NPOS + i] : 0.f;
                    if (F16) split2_pk_f16(x0 * in_scale, x1 * in_scale, hh[c], mm[c]);
                    else split3_pk(x0, x1, hh[c], mm[c], ll[c]);
                }
                int ps = p;
                if constexpr (DEINT_W) {                       // column wx of its row -> slot (wx >> 1) + (wx & 1) * 33
                    const int row = p / C::IW, wx = p - row * C::IW;
                    ps = row * C::IW + (wx >> 1) + (wx & 1) * ((C::IW + 1) / 2);
                }
                lds[0 * C::CS + ps] = make_uint4(hh[0], hh[1], hh[2], hh[3]);
                lds[1 * C::CS + ps] = make_uint4(mm[0], mm[1], mm[2], mm[3]);
                if (NC == 3) lds[2 * C::CS + ps] = make_uint4(ll[0], ll[1], ll[2], ll[3]);
            }
        };
#ifdef SS_EXP_CONV_NOSTAGE        // (timing experiment, wrong results: the split / transpose / LDS writes done once per workgroup)
        if (tile == (int)blockIdx.x && ci0 == 0)
#endif
        {
        if (nlive == 8) stage(std::true_type{});
        else stage(std::false_type{});
        }
        if (WLDS) __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0): the LDS-DMA weight loads have landed
        __syncthreads();
        const bool more = ci0 + 8 < Cin;                       // another chunk of THIS tile follows
        // what the K-steps prefetch: the next chunk of this tile, or (last chunk) the first chunk of the workgroup's next tile
        // (this tile's own offsets are not needed past this point: its last chunk is already staged)
        if constexpr (GATHER) {
            // second-to-last chunk: request the next tile's candidates / weights; last chunk: they have landed (every load of the
            // chunk in between was issued behind them) -- the prefetch below then gathers the next tile's first chunk
            if (more && ci0 + 16 >= Cin) gather_request(tile + (int)gridDim.x, gbuf ^ 1);
            if (!more) gather_offsets(tile + (int)gridDim.x, poff);
        } else if (!more) make_poff(tile + (int)gridDim.x, poff);          // pure index arithmetic under a wave-uniform branch
        nlive_next = more ? min(8, Cin - ci0 - 8) : min(8, Cin);
        const int ch_next = more ? ci0 + 8 : 0;
        const unsigned nomore = (more || has_next) ? 0u : 0x80000000u;
        SS_STAMP_STEPS_BEGIN();
        // ACCB: the chunk's 14 x 3 MFMAs accumulate from ZERO in a second register set that is added to `acc` once per chunk
        // (two-level blocked summation).  One chain over all of K = Cin x 27 products rounds a growing partial sum 3 K / 16
        // times: measured 3.3e-7 ... 6.2e-7 of the output's rms for K = 864 ... 3456 against float64 (tools/err_stages.py),
        // 1.2 - 3.5x the fp32 CPU convolution of the reference, which sums in blocks too; chunk-blocked it is 1.9e-7 ... 2.3e-7
        // whatever K, at or below the CPU's.  Where the second set fits the register budget it costs no time (stride-2
        // forms 85.0 vs 85.5 us, the 1 x 4 tile 68.8 vs 68.2); the 4-row tiles would spill (+7 %) and keep the single chain.
        f32x16 tacc[ACCB ? MT * NT : 1];
        if constexpr (ACCB) {
#pragma unroll
            for (int i = 0; i < MT * NT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) tacc[i][r] = 0.f;
        }

        // B fragments are read one row GROUP ahead of their MFMAs.  With RP = 2 the MFMAs of two rows
        // alternate so that no two consecutive ones share an accumulator (SS_ROW_PAIR; no gain measured).
        constexpr int RP = (NT >= 2) ? ((S == 2) ? SS_ROW_PAIR_S2 : SS_ROW_PAIR) : 1;
        static_assert(NT % RP == 0, "rows are processed in whole groups");
        uint4 bcur[RP][NC], bnxt[RP][NC];
        auto read_b = [&](uint4 (&dst)[NC], int s, int i) {
            const int ta = 2 * s, tb = 2 * s + 1;
            constexpr int kwo[3] = {0, DEINT ? (C::IW + 1) / 2 : 1, DEINT ? 1 : 2};            // slot offset of tap column kw
            const int offa = ((ta / 9) * C::IH + (ta / 3) % 3) * C::IW + kwo[ta % 3];
            const int offb = (tb < C::KT) ? ((tb / 9) * C::IH + (tb / 3) % 3) * C::IW + kwo[tb % 3] : 0;
            const int slot = lane_pos + i * S * C::IW + (half ? offb : offa);
#pragma unroll
            for (int c = 0; c < NC; ++c) dst[c] = lds[(tb >= C::KT && half) ? ZSLOT : c * C::CS + slot];
        };
#pragma unroll
        for (int r = 0; r < RP; ++r) read_b(bcur[r], 0, r);
        if (WLDS) {
#pragma unroll
            for (int c = 0; c < NC; ++c) aq[0][0][c] = lds[WL + c * 64 + lane];
        }
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            // weight fragments two steps ahead, then this step's slice of the next chunk's input
            // No vector-memory instruction of the K loop sits under a branch: where control flow merges, the compiler's
            // wait-count pass cannot tell how many loads are younger than the one it waits for and falls back to
            // vmcnt(0/1) -- every K-step then waited for the input loads it had just issued (seen in the ISA).  Past the
            // end, the last fragment is requested again and the input loads get an offset beyond the buffer (no access).
            if (!WLDS) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int c = 0; c < NC; ++c) {              // (past the tile's last step: steps 0, 1, .. of the next tile)
                        const int gw_ = g0 + s + AP;
                        aq[(s + AP) % AR][mt][c] = load_a(gw_ < G ? gw_ : gw_ - G, c, mt);
                    }
            } else if (s + 1 < KSTEPS) {                       // next step's fragments from the LDS copy
#pragma unroll
                for (int c = 0; c < NC; ++c) aq[(s + 1) % AR][0][c] = lds[WL + ((s + 1) * 2 + c) * 64 + lane];
            }
#ifndef SS_EXP_CONV_KEEP          // (timing experiment, wrong results: no input loads after a workgroup's first chunk -- the staged data keeps its statistics)
#pragma unroll
            for (int q = s * QS; q < (s + 1) * QS && q < NQ; ++q)
                rin[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                       ires, (int)(poff[q % C::NPOS] | nomore), (ch_next + min(q / C::NPOS, max(nlive_next, 1) - 1)) * chan_b, GATED ? SS_IN_AUX_GATED : SS_IN_AUX));
#endif
            uint4 a[MT][NC];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int c = 0; c < NC; ++c) a[mt][c] = aq[s % AR][mt][c];
#pragma unroll
            for (int i0 = 0; i0 < NT; i0 += RP) {
#pragma unroll
                for (int r = 0; r < RP; ++r) {
                    if (i0 + RP < NT) read_b(bnxt[r], s, i0 + RP + r);
                    else if (s + 1 < KSTEPS) read_b(bnxt[r], s + 1, r);
                }
                // cross terms (a term, b term), smallest first; rows of the group alternate
                constexpr int NP = (NTERMS == 6) ? 6 : 3;
                constexpr int pa[6] = {1, 0, NC - 1, 0, 1, 0}, pb[6] = {1, NC - 1, 0, 1, 0, 0};
#pragma unroll
                for (int p = 6 - NP; p < 6; ++p)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int r = 0; r < RP; ++r) {
                            f32x16& dst = ACCB ? tacc[ACCB ? mt * NT + i0 + r : 0] : acc[mt * NT + i0 + r];
                            if (F16)
                                dst = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                                    __builtin_bit_cast(f16x8, a[mt][pa[p]]), __builtin_bit_cast(f16x8, bcur[r][pb[p]]), dst, 0, 0, 0);
                            else
                                dst = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                                    __builtin_bit_cast(bf16x8, a[mt][pa[p]]), __builtin_bit_cast(bf16x8, bcur[r][pb[p]]), dst, 0, 0, 0);
                        }
#pragma unroll
                for (int r = 0; r < RP; ++r)
#pragma unroll
                    for (int c = 0; c < NC; ++c) bcur[r][c] = bnxt[r][c];
                // pin the software pipeline: the next group's fragment reads are issued BEFORE this
                // group's MFMAs (the scheduler otherwise sinks them next to their use and every row
                // starts with an exposed LDS latency)
                __builtin_amdgcn_sched_group_barrier(0x100, NC * RP, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, NP * RP * MT, 0);
            }
            __builtin_amdgcn_sched_barrier(0);     // keep each step's loads inside the step
        }
        // steps 14 .. 14+AP-1 of this chunk are steps 0 .. AP-1 of the next: re-base the fragment ring
        if (!WLDS) {
            uint4 tq[AP][MT][NC];
#pragma unroll
            for (int k = 0; k < AP; ++k)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int c = 0; c < NC; ++c) tq[k][mt][c] = aq[(KSTEPS + k) % AR][mt][c];
#pragma unroll
            for (int k = 0; k < AP; ++k)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int c = 0; c < NC; ++c) aq[k][mt][c] = tq[k][mt][c];
        }
        if constexpr (ACCB) {
#pragma unroll
            for (int i = 0; i < MT * NT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] += tacc[i][r];
        }
        nlive = nlive_next;
        SS_STAMP_STEPS_END();
        if (GATHER && (more || has_next)) apply_att(more ? gbuf : gbuf ^ 1);
        if (F16 && (more || has_next)) publish_max(0.f);       // of the chunk staged next (its loads were issued >= 4 K-steps ago)
        __syncthreads();
    }

    SS_STAMP(2);
    if constexpr (HEAD) {
        int ow0, oh0, od0;
        tile_origin(tile, ow0, oh0, od0);
        const float hunscale = __uint_as_float((unsigned)(127 - E_ONE + e_cur) << 23);
        const float hfloor = (relu & 1) ? 0.f : -__builtin_inff();
        const bool colok = ow0 + l31 < Wo && od0 + dzw < Do;
        // ---- t[tap row][position] of this wave's rows: the accumulators become the B operand as they are ----
#ifndef SS_EXP_HEAD_NOT           // (timing experiment, wrong results: the head's contraction skipped)
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const bool ok = colok && oh0 + hy0 + i < Ho;
            float y[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cl = (r & 3) + 8 * (r >> 2) + 4 * half;
                const float v = fmaxf(ss::add_rn(ss::mul_rn(acc[i][r] * (aff[128 + cl] * hunscale), aff[cl]), aff[64 + cl]), hfloor);
                y[r] = ok ? v : 0.f;            // positions outside the volume: the head's zero padding
            }
            f32x16 tt = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                unsigned bh[4], bm[4], bl[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) split3_pk(y[8 * ks + 2 * c], y[8 * ks + 2 * c + 1], bh[c], bm[c], bl[c]);
                const bf16x8 h8 = __builtin_bit_cast(bf16x8, make_uint4(bh[0], bh[1], bh[2], bh[3]));
                const bf16x8 m8 = __builtin_bit_cast(bf16x8, make_uint4(bm[0], bm[1], bm[2], bm[3]));
                const bf16x8 l8 = __builtin_bit_cast(bf16x8, make_uint4(bl[0], bl[1], bl[2], bl[3]));
                const bf16x8 ah = __builtin_bit_cast(bf16x8, lds[C::XS0 + (0 * 2 + ks) * 64 + lane]);
                const bf16x8 am = __builtin_bit_cast(bf16x8, lds[C::XS0 + (1 * 2 + ks) * 64 + lane]);
                const bf16x8 al = __builtin_bit_cast(bf16x8, lds[C::XS0 + (2 * 2 + ks) * 64 + lane]);
                tt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, m8, tt, 0, 0, 0);      // smallest cross terms first
                tt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, l8, tt, 0, 0, 0);
                tt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, h8, tt, 0, 0, 0);
                tt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, m8, tt, 0, 0, 0);
                tt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, h8, tt, 0, 0, 0);
                tt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, h8, tt, 0, 0, 0);
            }
            acc[i] = tt;                        // register 3 * ps + kh of this lane half: (kd, kw) pair ps + 5 * half, tap row kh
            __builtin_amdgcn_sched_barrier(0);  // one row at a time (all four interleaved: 43 spilled registers)
        }
#endif
        // ---- over kh, in registers: U[ps][ph] = sum_row t[row][3 ps + (row + 2 - ph)], ph = 0..5 the patch row (fixed order) ----
        float* ubuf = reinterpret_cast<float*>(lds);      // [10 pairs][6 patch rows][4 planes][32 columns], over the dead activation tile
        float* ub = ubuf + (5 * half * 6 * 4 + wave) * 32 + l31;      // (one per-lane address, the rest immediate offsets)
#pragma unroll
        for (int ps = 0; ps < 5; ++ps) {
#pragma unroll
            for (int ph = 0; ph < 6; ++ph) {
                float u = 0.f;
#pragma unroll
                for (int row = 0; row < 4; ++row) {
                    const int kh = row + 2 - ph;
                    if (kh >= 0 && kh < 3) u = ss::add_rn(u, acc[row][3 * ps + kh]);
                }
                ub[(ps * 6 + ph) * 4 * 32] = u;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
#ifndef SS_EXP_HEAD_NOP           // (timing experiment, wrong results: the patch positions' sums skipped)
        // ---- over (kd, kw): each thread owns patch positions q = tid + 256 i ----
        float* pout = out + ((size_t)b * ntiles + tile) * HEAD_PATCH;
        int ptid = tid;                // (opaque: the positions' index arithmetic is redone here per tile instead of living in registers -- or scratch -- through the K loop)
        asm volatile("" : "+v"(ptid));
#pragma unroll
        for (int i = 0; i < (HEAD_PATCH + 255) / 256; ++i) {
            const int q = ptid + 256 * i;
            if (q < HEAD_PATCH) {
                const int pw = q % 34, ph = (q / 34) % 6, pd = q / (34 * 6);
                // (all nine reads issued before the first add, from clamped -- always valid -- addresses: under a branch each the reads
                // were 37 dependent LDS round trips per thread)
                float uv[9];
#pragma unroll
                for (int kd = 0; kd < 3; ++kd)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        const int pl = min(max(pd - 2 + kd, 0), 3), n = min(max(pw - 2 + kw, 0), 31);
                        uv[kd * 3 + kw] = ubuf[(((kd * 3 + kw) * 6 + ph) * 4 + pl) * 32 + n];
                    }
                float pv = 0.f;
#pragma unroll
                for (int kd = 0; kd < 3; ++kd)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        const bool in = (unsigned)(pd - 2 + kd) < 4u && (unsigned)(pw - 2 + kw) < 32u;
                        pv = ss::add_rn(pv, in ? uv[kd * 3 + kw] : 0.f);
                    }
                pout[q] = pv;
            }
        }
#endif
        __syncthreads();               // the next tile's first chunk is staged over ubuf
    } else {
    set_outputs(tile);                 // (output addressing is derived here, not kept live through the K loop)
    // ---- epilogue: 32x32 D layout (col = lane & 31 = output column, row = channel, see cbase) ----
    // f16 form: 2^-(activation scale); the per-channel 2^-(weight scale) is stored behind the packed weights
    const float acc_unscale = __uint_as_float((unsigned)(127 - E_ONE + e_cur) << 23);
    const float* obase = out + (size_t)b * Cout * Do * out_plane;
    const __amdgpu_buffer_rsrc_t ores = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(obase), 0, obytes, 0x00020000);
    // the residual: added after the affine, or (res_pre) a partial sum added before it
    const bool res_epi = residual != nullptr && !res_pre;
    const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(residual ? residual + (size_t)b * Cout * Do * out_plane : obase), 0, residual ? obytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t gres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(GATED ? gate + (size_t)b * Cout * out_plane : obase), 0,
        GATED ? (int)min((long long)Cout * (long long)gchan_b, 0x7fffffffLL) : 0, 0x00020000);
    const float floor_v = (relu & 1) ? 0.f : -__builtin_inff();
    // the side inputs (affine, gate, residual) of a group of EG fragment rows are fetched first so that their latencies
    // overlap instead of chaining; every group costs one exposed round trip (load -> store -> the next group's loads)
    // (the next tile's first chunk occupies the prefetch registers during the epilogue: with both a gate and a residual to
    // fetch, groups of 4 rows keep the kernel out of scratch)
    constexpr int EG = (GATED || NT * MT >= 4) ? 4 : 8;
    // (a workgroup whose 32 * MT channels all exist -- every one when Cout % 32 == 0 -- takes the copy of the epilogue without
    // the per-element "channel exists" selects on the load / store offsets: 4 of ~11 VALU instructions per element, ISA count r03)
    auto epilogue = [&](auto all_channels) {
    constexpr bool ALLC = decltype(all_channels)::value;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r0 = 0; r0 < 16; r0 += EG) {
        float sc[EG], sh[EG], un[EG], gv[EG][NT], rv[EG][NT];
#pragma unroll
        for (int q = 0; q < EG; ++q) {
            const int cb = cbase(mt, r0 + q);
            const bool cok = ALLC || cb + 4 * half < Cout;
            sc[q] = aff[cb - co0 + 4 * half];
            sh[q] = aff[64 + cb - co0 + 4 * half];
            un[q] = F16 ? aff[128 + cb - co0 + 4 * half] * acc_unscale : 1.0f;      // powers of two: acc * un is exact
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                if (GATED) gv[q][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                                      gres, (int)(cok ? vgate[i] : 0x80000000u), cb * (int)gchan_b, 0));
            }
        }
        if (residual != nullptr) {                             // one uniform branch per group, none per element
#pragma unroll
            for (int q = 0; q < EG; ++q) {
                const int cb = cbase(mt, r0 + q);
                const bool cok = ALLC || cb + 4 * half < Cout;
#pragma unroll
                for (int i = 0; i < NT; ++i)
                    rv[q][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                             rres, (int)(cok ? vout[i] : 0x80000000u), cb * (int)ochan_b, GATED ? SS_IN_AUX_GATED : 0));
            }
        } else {
#pragma unroll
            for (int q = 0; q < EG; ++q)
#pragma unroll
                for (int i = 0; i < NT; ++i) rv[q][i] = 0.f;
        }
        float vv[EG][NT];
#pragma unroll
        for (int q = 0; q < EG; ++q) {
            const int r = r0 + q;
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                float a0 = F16 ? acc[mt * NT + i][r] * un[q] : acc[mt * NT + i][r];
                if (res_pre) a0 = ss::add_rn(a0, rv[q][i]);                 // the partial sum of the same convolution
                float v = ss::add_rn(ss::mul_rn(a0, sc[q]), sh[q]);
                if (res_epi) v = ss::add_rn(v, rv[q][i]);
                v = fmaxf(v, floor_v);
                if (GATED) v = ss::mul_rn(gv[q][i], v);     // channelAtt gate, broadcast over D
                vv[q][i] = v;
            }
        }
        if (CAN_CL && out_cl) {
#pragma unroll
            for (int g4 = 0; g4 < EG; g4 += 4) {
                const int cb = cbase(mt, r0 + g4);           // channels cb + 4 * half .. + 3 sit in registers r0 + g4 .. + 3
                const bool cok = ALLC || cb + 4 * half < Cout;
#pragma unroll
                for (int i = 0; i < NT; ++i)
                    __builtin_amdgcn_raw_buffer_store_b128(
                        __builtin_bit_cast(u32x4_t, make_float4(vv[g4][i], vv[g4 + 1][i], vv[g4 + 2][i], vv[g4 + 3][i])), ores,
                        (int)(cok ? vout[i] : 0x80000000u), cb * 4, 0);
            }
        } else {
#pragma unroll
            for (int q = 0; q < EG; ++q) {
                const int cb = cbase(mt, r0 + q);
                const bool cok = ALLC || cb + 4 * half < Cout;
#pragma unroll
                for (int i = 0; i < NT; ++i)
#ifdef SS_EXP_CONV_NOSTORE        // (timing experiment: one store per lane instead of 64)
                    if (r0 + q + i == 0)
#endif
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vv[q][i]), ores,
                                                          (int)(cok ? vout[i] : 0x80000000u), cb * (int)ochan_b, 0);
            }
        }
    }
    };
    if (co0 + 32 * MT * MS <= Cout) epilogue(std::true_type{});
    else epilogue(std::false_type{});
    }       // !HEAD
    SS_STAMP(3);
    }       // tiles of this workgroup (the next one's first chunk is already in the prefetch registers)
    SS_STAMP_FINISH();
}

#ifndef SS_CONV_GATHER_TU
// [Cout,Cin,3,3,3] fp32 -> [ceil(Cin/8)][14 steps][3 terms][2 halves][Cout][8] bf16 (zero padded)
__global__ void pack_weights_bf16s_kernel(const float* __restrict__ w, unsigned short* __restrict__ wsplit, int Cout,
                                          int Cin, int ktaps, long long total) {
    const int KSTEPS = (ktaps + 1) / 2;                        // 14 (3x3x3) or 5 (3x3)
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int j = (int)(i % 8);
    long long r = i / 8;
    const int co = (int)(r % Cout); r /= Cout;
    const int half = (int)(r % 2); r /= 2;
    const int term = (int)(r % 3); r /= 3;
    const int s = (int)(r % KSTEPS);
    const int blk = (int)(r / KSTEPS);
    const int tap = 2 * s + half, ci = blk * 8 + j;
    float x = 0.f;
    if (tap < ktaps && ci < Cin) x = w[((long long)co * Cin + ci) * ktaps + tap];
    unsigned h, m, l;
    split3(x, h, m, l);
    wsplit[i] = (unsigned short)(term == 0 ? h : (term == 1 ? m : l));
}

// f16 form: per output channel, the power of two that brings max |w| into [2^14, 2^15) -- its inverse is stored behind the packed terms
// (float[Cout]) for the conv kernel's epilogue -- then [Cout,Cin,taps] fp32 -> [ceil(Cin/8)][steps][2 terms][2 halves][Cout][8] fp16 of
// w / wunscale[co].  One workgroup per output channel does both (r06: they were two launches, and training re-packs every layer's
// weights every step: ~100 small launches per step; same arithmetic on the same values, bit-identical output).
__global__ __launch_bounds__(256) void pack_weights_f16s_fused_kernel(const float* __restrict__ w, unsigned short* __restrict__ wsplit,
                                                                       float* __restrict__ wunscale, int Cout, int Cin, int ktaps) {
    __shared__ unsigned wmax[4];
    __shared__ float unscale_s;
    const int co = blockIdx.x, per_co = Cin * ktaps;
    const float* wc = w + (size_t)co * per_co;
    float m = 0.f;
    for (int i = threadIdx.x; i < per_co; i += 256) m = fmaxf(m, fabsf(wc[i]));
    const unsigned wm = wave_max_bits(__float_as_uint(m));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = wm;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int e = max((int)(max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3])) >> 23), E_MIN);
        const float u = __uint_as_float((unsigned)(127 - E_ONE + e) << 23);
        wunscale[co] = u;
        unscale_s = u;
    }
    __syncthreads();
    const float u = unscale_s;
    const int KSTEPS = (ktaps + 1) / 2, nblk = (Cin + 7) / 8;
    const int n = nblk * KSTEPS * 4 * 8;                      // this channel's slots: (blk, s, term, half, j)
    for (int e = threadIdx.x; e < n; e += 256) {
        const int j = e % 8;
        int r = e / 8;
        const int half = r % 2; r /= 2;
        const int term = r % 2; r /= 2;
        const int s = r % KSTEPS;
        const int blk = r / KSTEPS;
        const int tap = 2 * s + half, ci = blk * 8 + j;
        float x = 0.f;
        if (tap < ktaps && ci < Cin) x = wc[(size_t)ci * ktaps + tap] / u;       // exact: a power of two
        const _Float16 h = (_Float16)x;
        const _Float16 l = (_Float16)(x - (float)h);
        const long long i = ((((long long)(blk * KSTEPS + s) * 2 + term) * 2 + half) * Cout + co) * 8 + j;
        wsplit[i] = __builtin_bit_cast(unsigned short, term == 0 ? h : l);
    }
}

#endif  // !SS_CONV_GATHER_TU

template <int S, int NT, int TD, int TH, int NTERMS, bool GATED, int MT, int KD = 3, int MS = 1, bool ACCB = false, bool GATHER = false, bool HEAD = false>
int launch_bgm(const float* in, const void* wsplit, const float* scale, const float* shift, const float* residual,
              const float* gate, float* out, int B, int Cin, int D, int H, int W, int Cout, int relu, hipStream_t st,
              const float* cand = nullptr, const float* catt = nullptr, const float* in2 = nullptr, int bsplit = 0) {
    using C = BCfg<S, NT, TD, TH, KD, (NTERMS == 6) ? 3 : 2, wlds_form(S, NT, NTERMS, MT, KD, GATHER) ? ((KD * 9 + 1) / 2) * 2 * 64 : 0, MS,
                   gather_slots(GATHER, S, TD, TH, KD) + (HEAD ? HEAD_WSLOTS : 0)>;
    const int Do = (D + 2 * (KD / 2) - KD) / S + 1, Ho = (H - 1) / S + 1, Wo = (W - 1) / S + 1;
    const int tiles_w = ss::ceil_div(Wo, 32), tiles_h = ss::ceil_div(Ho, TH), tiles_d = ss::ceil_div(Do, TD);
    const long long nt = (long long)tiles_w * tiles_h * tiles_d;
    if (nt > 0x7fffffffLL || B > 65535) return SS_ERR_UNSUPPORTED;
    auto kern = conv3d_bf16s<S, NT, TD, TH, NTERMS, GATED, MT, KD, MS, ACCB, GATHER, HEAD>;
    if (C::LDS_BYTES > 64 * 1024) {
        if (ss::ensure_dynamic_lds(reinterpret_cast<const void*>(kern), (int)C::LDS_BYTES) != SS_OK) return SS_ERR_LAUNCH;
    }
    // persistent workgroups: as many as the chip holds at once (2 - 4 per CU depending on the tile), each walking an equal
    // share of the tiles
    const int groups = ss::ceil_div(Cout, 32 * MT * MS) * B;
    const long long cap = std::max<long long>(1, ss::resident_workgroups(reinterpret_cast<const void*>(kern), 256, (int)C::LDS_BYTES) / groups);
    const long long rounds = ss::ceil_div_ll(nt, cap);
    const long long gx = ss::ceil_div_ll(nt, rounds);
    dim3 grid((unsigned)gx, ss::ceil_div(Cout, 32 * MT * MS), B);
    // (Workgroups of equal duration that all start together stay in lock-step -- every CU stages, multiplies and stores at
    // the same time.  Starting the first round's workgroups spread over 0.5-1.5 estimated lifetimes, in 2-16 groups, was
    // measured: no gain, -0 .. -8 %.)
    hipLaunchKernelGGL(kern, grid, dim3(256), C::LDS_BYTES, st, in, reinterpret_cast<const uint4*>(wsplit), scale, shift,
                       residual, gate, out, Cin, D, H, W, Cout, Do, Ho, Wo, tiles_w, tiles_h, (int)nt, relu, cand, catt, in2, bsplit);
    return ss::check_launch();
}

#ifndef SS_CONV_GATHER_TU
template <int S, int NT, int TD, int TH, int NTERMS, bool GATED, bool ACCB>
int launch_bg(const float* in, const void* wsplit, const float* scale, const float* shift, const float* residual,
              const float* gate, float* out, int B, int Cin, int D, int H, int W, int Cout, int relu, hipStream_t st) {
    // stride 2: two output tiles per wave (the activation staging is then shared: 154 vs 191 us on the largest layer)
    // unless that leaves fewer workgroups than CUs (57 vs 41 us on the smallest)
    const int Do = (D - 1) / S + 1, Ho = (H - 1) / S + 1, Wo = (W - 1) / S + 1;
    const long long wg2 = (long long)ss::ceil_div(Wo, 32) * ss::ceil_div(Ho, TH) * ss::ceil_div(Do, TD) * ss::ceil_div(Cout, 64) * B;
    // ... and, since r03, the 64 channels split over the waves (MS = 2: wave = (row pair, 32 channels)) instead of two channel
    // tiles per wave: the same MFMAs and staging with half the weight-fragment fetches (SS_CONV_S2_MT1=0: the r02 form)
    if constexpr (S == 2 && NT == 1 && TD * TH == 4 && !GATED) {      // (no layer gates a stride-2 conv; its MS form would spill)
        if (Cout > 32 && wg2 >= SS_S2_MS_MIN_WGS && ss::tuning().conv_s2_mt1 < 0)
            return launch_bgm<S, 2, TD, TH, NTERMS, GATED, 1, 3, 2, ACCB>(in, wsplit, scale, shift, residual, gate, out, B, Cin, D, H, W, Cout, relu, st);
    }
    if (S == 2 && Cout > 32 && wg2 >= 256 && ss::tuning().conv_s2_mt1 <= 0)
        return launch_bgm<S, NT, TD, TH, NTERMS, GATED, (S == 2) ? 2 : 1, 3, 1, ACCB>(in, wsplit, scale, shift, residual, gate, out, B, Cin, D, H, W,
                                                                                      Cout, relu, st);
    return launch_bgm<S, NT, TD, TH, NTERMS, GATED, 1, 3, 1, ACCB>(in, wsplit, scale, shift, residual, gate, out, B, Cin, D, H, W, Cout, relu, st);
}

template <int S, int NT, int TD, int TH, int NTERMS>
int launch_b(const float* in, const void* wsplit, const float* scale, const float* shift, const float* residual,
             const float* gate, float* out, int B, int Cin, int D, int H, int W, int Cout, int relu, bool accb, hipStream_t st) {
    if constexpr (NTERMS == F16X3) {
        if (accb) {
            if (gate != nullptr)
                return launch_bg<S, NT, TD, TH, NTERMS, true, true>(in, wsplit, scale, shift, residual, gate, out, B, Cin, D, H, W, Cout, relu, st);
            return launch_bg<S, NT, TD, TH, NTERMS, false, true>(in, wsplit, scale, shift, residual, gate, out, B, Cin, D, H, W, Cout, relu, st);
        }
    }
    if (gate != nullptr)
        return launch_bg<S, NT, TD, TH, NTERMS, true, false>(in, wsplit, scale, shift, residual, gate, out, B, Cin, D, H, W, Cout, relu, st);
    return launch_bg<S, NT, TD, TH, NTERMS, false, false>(in, wsplit, scale, shift, residual, gate, out, B, Cin, D, H, W, Cout, relu, st);
}

#endif  // !SS_CONV_GATHER_TU

}  // namespace

#ifndef SS_CONV_GATHER_TU
static int conv3d_bf16s_impl(const float* in, const void* wsplit, const float* scale, const float* shift,
                             const float* residual, const float* gate, float* out, int B, int Cin, int D, int H,
                             int W, int Cout, int stride, int relu, int nterms, ss_stream_t stream);

extern "C" int ss_conv3d_bf16s_fwd(const float* in, const void* wsplit, const float* scale, const float* shift,
                                   const float* residual, const float* gate, float* out, int B, int Cin, int D, int H,
                                   int W, int Cout, int stride, int relu, int nterms, ss_stream_t stream) {
    return conv3d_bf16s_impl(in, wsplit, scale, shift, residual, gate, out, B, Cin, D, H, W, Cout, stride, relu ? 1 : 0, nterms,
                             stream);
}

extern "C" int ss_conv3d_bf16s_partial_fwd(const float* in, const void* wsplit, const float* partial, const float* scale,
                                           const float* shift, const float* gate, float* out, int B, int Cin, int D,
                                           int H, int W, int Cout, int relu, int nterms, ss_stream_t stream) {
    SS_REQUIRE(partial != nullptr);
    return conv3d_bf16s_impl(in, wsplit, scale, shift, partial, gate, out, B, Cin, D, H, W, Cout, 1, (relu ? 1 : 0) | 2, nterms,
                             stream);
}

extern "C" int ss_conv3d_bf16s_cl_fwd(const float* in, const void* wsplit, const float* scale, const float* shift, float* out,
                                      int B, int Cin, int D, int H, int W, int Cout, int relu, int nterms, ss_stream_t stream) {
    SS_REQUIRE(Cout > 0 && Cout % 8 == 0);
    return conv3d_bf16s_impl(in, wsplit, scale, shift, nullptr, nullptr, out, B, Cin, D, H, W, Cout, 1, (relu ? 1 : 0) | 4, nterms,
                             stream);
}

static int conv3d_bf16s_impl(const float* in, const void* wsplit, const float* scale, const float* shift,
                                   const float* residual, const float* gate, float* out, int B, int Cin, int D, int H,
                                   int W, int Cout, int stride, int relu, int nterms, ss_stream_t stream) {
    SS_REQUIRE(in && wsplit && out);
    SS_REQUIRE(B > 0 && Cin > 0 && D > 0 && H > 0 && W > 0 && Cout > 0);
    SS_REQUIRE((stride == 1 || stride == 2) && (nterms == 3 || nterms == 6 || nterms == F16X3));
    SS_REQUIRE((reinterpret_cast<uintptr_t>(wsplit) & 15) == 0);
    // operands are addressed through 32-bit buffer offsets: one batch element's input must stay below 2 GiB
    if ((long long)Cin * D * H * W * 4 >= 0x7fffffffLL) return SS_ERR_UNSUPPORTED;
    hipStream_t st = ss::as_stream(stream);
    const int Do = (D - 1) / stride + 1, Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    if ((long long)Cout * Do * Ho * Wo * 4 >= 0x7fffffffLL) return SS_ERR_UNSUPPORTED;      // ... and so must its output
    auto blocks = [&](int td, int th) {
        return (long long)ss::ceil_div(Wo, 32) * ss::ceil_div(Ho, th) * ss::ceil_div(Do, td) * ss::ceil_div(Cout, 32) * B;
    };
    const int forced = ss::tuning().conv_tile;
    int tile = (blocks(2, 8) >= 512) ? 0 : ((blocks(1, 8) >= 512) ? 1 : 2);
    if (forced >= 0 && forced <= 2) tile = forced;
    // chunk-blocked accumulation: decided by the LAYER (what ONE pair of it offers the chip), not by this launch's batch or tile
    const bool small_layer = blocks(2, 8) / B < 512;
    const bool accb = SS_ACC_BLOCKED && nterms == F16X3 && (stride == 2 || small_layer);
#define SS_B(S, NT, TD, TH)                                                                                              \
    return (nterms == 6) ? launch_b<S, NT, TD, TH, 6>(in, wsplit, scale, shift, residual, gate, out, B, Cin, D, H, W, Cout, relu, accb, st) \
         : (nterms == 3) ? launch_b<S, NT, TD, TH, 3>(in, wsplit, scale, shift, residual, gate, out, B, Cin, D, H, W, Cout, relu, accb, st) \
                         : launch_b<S, NT, TD, TH, F16X3>(in, wsplit, scale, shift, residual, gate, out, B, Cin, D, H, W, Cout, relu, accb, st)
    // stride 2 needs a 65-column halo tile per row.  A 1 x 4 output tile is 84 KB of split operands: one workgroup per
    // CU, slower than the exact-fp32 kernel (283 vs 229 us on the largest layer).  A 2 x 2 tile is 78 KB: two
    // workgroups per CU, faster on every stride-2 layer of the model (190 / 95 / 68 / 41 us vs 229 / 122 / 81 / 62).
    if (stride == 2) { SS_B(2, 1, 2, 2); }     // 5 x 5 x 65 halo positions: 78 KB of split operands, two workgroups per CU
    // 4 planes x 4 rows: the most compact halo (6 x 6 x 34 = 1224 positions against 1360 for 2 x 8: -3 % time, 8 fewer
    // prefetch registers); 2 x 8 when the depth is not a multiple of 4
    if (tile == 0 && Do % 4 == 0) { SS_B(1, 4, 4, 4); }
    if (tile == 0) { SS_B(1, 4, 2, 8); }
    if (tile == 1) { SS_B(1, 2, 1, 8); }
    SS_B(1, 1, 1, 4);
#undef SS_B
}

extern "C" int ss_pack_conv3d_weights_bf16s(const float* w, void* wsplit, int Cout, int Cin, ss_stream_t stream) {
    SS_REQUIRE(w && wsplit && Cout > 0 && Cin > 0);
    const long long total = (long long)ss::ceil_div(Cin, 8) * KSTEPS * 3 * 2 * Cout * 8;
    hipLaunchKernelGGL(pack_weights_bf16s_kernel, dim3((unsigned)ss::ceil_div_ll(total, 256)), dim3(256), 0,
                       ss::as_stream(stream), w, reinterpret_cast<unsigned short*>(wsplit), Cout, Cin, 27, total);
    return ss::check_launch();
}

// the two-term fp16 form of the same weights (nterms = 19 of ss_conv3d_bf16s_fwd): see the top of this file
extern "C" int ss_pack_conv3d_weights_f16s(const float* w, void* wsplit, int Cout, int Cin, ss_stream_t stream) {
    SS_REQUIRE(w && wsplit && Cout > 0 && Cin > 0);
    const long long total = (long long)ss::ceil_div(Cin, 8) * KSTEPS * 2 * 2 * Cout * 8;
    float* wunscale = reinterpret_cast<float*>(reinterpret_cast<char*>(wsplit) + total * 2);
    hipLaunchKernelGGL(pack_weights_f16s_fused_kernel, dim3(Cout), dim3(256), 0, ss::as_stream(stream), w, reinterpret_cast<unsigned short*>(wsplit),
                       wunscale, Cout, Cin, 27);
    return ss::check_launch();
}

// ---- the 2-D form: Conv2d(k3, s1, p1, bias=False) + affine + optional residual + ReLU on [B,C,H,W] maps (concat_feature of
// the reference model, models/SemStereo.py:222-226): the same kernel with a depth-1 volume and 9 taps (5 K-steps) ----
extern "C" int ss_pack_conv2d_weights_bf16s(const float* w, void* wsplit, int Cout, int Cin, ss_stream_t stream) {
    SS_REQUIRE(w && wsplit && Cout > 0 && Cin > 0);
    const long long total = (long long)ss::ceil_div(Cin, 8) * 5 * 3 * 2 * Cout * 8;
    hipLaunchKernelGGL(pack_weights_bf16s_kernel, dim3((unsigned)ss::ceil_div_ll(total, 256)), dim3(256), 0,
                       ss::as_stream(stream), w, reinterpret_cast<unsigned short*>(wsplit), Cout, Cin, 9, total);
    return ss::check_launch();
}

extern "C" int ss_pack_conv2d_weights_f16s(const float* w, void* wsplit, int Cout, int Cin, ss_stream_t stream) {
    SS_REQUIRE(w && wsplit && Cout > 0 && Cin > 0);
    const long long total = (long long)ss::ceil_div(Cin, 8) * 5 * 2 * 2 * Cout * 8;
    float* wunscale = reinterpret_cast<float*>(reinterpret_cast<char*>(wsplit) + total * 2);
    hipLaunchKernelGGL(pack_weights_f16s_fused_kernel, dim3(Cout), dim3(256), 0, ss::as_stream(stream), w, reinterpret_cast<unsigned short*>(wsplit),
                       wunscale, Cout, Cin, 9);
    return ss::check_launch();
}

static int conv2d_bf16s_impl(const float* in, const float* in2, int bsplit, const void* wsplit, const float* scale, const float* shift,
                            const float* residual, float* out, int B, int Cin, int H, int W, int Cout, int relu,
                            int nterms, ss_stream_t stream);

extern "C" int ss_conv2d_bf16s_fwd(const float* in, const void* wsplit, const float* scale, const float* shift,
                                   const float* residual, float* out, int B, int Cin, int H, int W, int Cout, int relu,
                                   int nterms, ss_stream_t stream) {
    return conv2d_bf16s_impl(in, nullptr, 0, wsplit, scale, shift, residual, out, B, Cin, H, W, Cout, relu, nterms, stream);
}

extern "C" int ss_conv2d_bf16s_pair_fwd(const float* in_a, const float* in_b, const void* wsplit, const float* scale,
                                        const float* shift, float* out, int B, int Cin, int H, int W, int Cout, int relu,
                                        int nterms, ss_stream_t stream) {
    SS_REQUIRE(in_a && in_b);
    return conv2d_bf16s_impl(in_a, in_b, B, wsplit, scale, shift, nullptr, out, 2 * B, Cin, H, W, Cout, relu, nterms, stream);
}

static int conv2d_bf16s_impl(const float* in, const float* in2, int bsplit, const void* wsplit, const float* scale, const float* shift,
                            const float* residual, float* out, int B, int Cin, int H, int W, int Cout, int relu,
                            int nterms, ss_stream_t stream) {
    SS_REQUIRE(in && wsplit && out);
    SS_REQUIRE(B > 0 && Cin > 0 && H > 0 && W > 0 && Cout > 0 && (nterms == 3 || nterms == 6 || nterms == F16X3));
    SS_REQUIRE((reinterpret_cast<uintptr_t>(wsplit) & 15) == 0);
    if ((long long)Cin * H * W * 4 >= 0x7fffffffLL) return SS_ERR_UNSUPPORTED;
    hipStream_t st = ss::as_stream(stream);
    auto blocks = [&](int th) { return (long long)ss::ceil_div(W, 32) * ss::ceil_div(H, th) * ss::ceil_div(Cout, 32) * B; };
    const int r = relu ? 1 : 0;
#define SS_B2(NT, TH)                                                                                                      \
    return (nterms == 6) ? launch_bgm<1, NT, 1, TH, 6, false, 1, 1>(in, wsplit, scale, shift, residual, nullptr, out, B, Cin, 1, H, W, Cout, r, st, nullptr, nullptr, in2, bsplit) \
         : (nterms == 3) ? launch_bgm<1, NT, 1, TH, 3, false, 1, 1>(in, wsplit, scale, shift, residual, nullptr, out, B, Cin, 1, H, W, Cout, r, st, nullptr, nullptr, in2, bsplit) \
                         : launch_bgm<1, NT, 1, TH, F16X3, false, 1, 1, 1, SS_ACC_BLOCKED != 0>(in, wsplit, scale, shift, residual, nullptr, out, B, Cin, 1, H, W, Cout, r, st, nullptr, nullptr, in2, bsplit)
    if (blocks(16) >= 512) { SS_B2(4, 16); }
    if (blocks(8) >= 512) { SS_B2(2, 8); }
    SS_B2(1, 4);
#undef SS_B2
}
#endif  // !SS_CONV_GATHER_TU
