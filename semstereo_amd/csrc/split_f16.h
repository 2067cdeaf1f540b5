// Shared pieces of the two-term fp16 operand form of the matrix-core convolutions (conv3d_bf16s.hip,
// deconv3d_bf16s.hip).  Device code only; included inside each file's anonymous namespace users.
#pragma once
#include "common.h"

namespace {

using f32x2_t = __attribute__((ext_vector_type(2))) float;
// ---- the two-term fp16 form ("f16x3", NTERMS = 19) ----
// fp16 carries 11 significand bits, so x = hi + lo leaves |x - hi - lo| <= 2^-23 |x| (one fp32 ulp) and THREE
// products (hh, hl, lh; ll <= 2^-22 is dropped) reach the accuracy of the six bf16 ones at half the matrix-core
// time -- provided both terms stay NORMAL fp16 numbers (5 exponent bits).  Weights are pre-scaled per output
// channel by a power of two at pack time (undone in the epilogue); activations are block floating point: each
// staged chunk (8 channels x halo tile) is multiplied by 2^k with k from the running |max| of the tile, and the
// fp32 accumulators are re-scaled (exactly: a power of two) whenever k changes.  Everything within 2^-17 of the
// tile's maximum keeps full precision; below that the absolute error is <= 2^-39 of that maximum.
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x2_t = __attribute__((ext_vector_type(2))) _Float16;
__device__ __forceinline__ void split2_pk_f16(float x0, float x1, unsigned& h, unsigned& l) {
    const f32x2_t v = {x0, x1};
    const f16x2_t hv = __builtin_convertvector(v, f16x2_t);
    const f32x2_t r = {x0 - (float)hv[0], x1 - (float)hv[1]};
    h = __builtin_bit_cast(unsigned, hv);
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2_t));
}
// max over the wave of non-negative floats (as their bit patterns), wave-uniform result
__device__ __forceinline__ unsigned wave_max_bits(unsigned x) {
    x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true));      // row_shr:1, 0 shifted in
    x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true));      // row_shr:2
    x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true));      // row_shr:4
    x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true));      // row_shr:8 -> lane 15 of each row
    return max(max((unsigned)__builtin_amdgcn_readlane((int)x, 15), (unsigned)__builtin_amdgcn_readlane((int)x, 31)),
               max((unsigned)__builtin_amdgcn_readlane((int)x, 47), (unsigned)__builtin_amdgcn_readlane((int)x, 63)));
}
// One wave-wide LDS-DMA load (buffer_load_dwordx4 ... lds): 64 lanes x 16 bytes from (voffset per lane, soffset) to 1 KB of
// LDS at `dst`, no registers.  (Kept in a helper: called directly inside some __global__ templates the builtin makes
// hipcc 7.2 drop the kernel's host stub without a diagnostic.)
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t res, uint4* dst, int voffset, int soffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(res, (__attribute__((address_space(3))) void*)dst, 16, voffset, soffset, 0, 0);
}
// ... and 4 bytes per lane (buffer_load_dword ... lds): 64 lanes -> 256 bytes of LDS at `dst`
__device__ __forceinline__ void lds_dma4(__amdgpu_buffer_rsrc_t res, float* dst, int voffset, int soffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(res, (__attribute__((address_space(3))) void*)dst, 4, voffset, soffset, 0, 0);
}
// a 4-byte LDS read that stays a ds_read_b32 on data the compiler cannot see being written (LDS-DMA); `p` points into LDS
__device__ __forceinline__ float lds_read4(const float* p) {
    return *(const volatile float __attribute__((address_space(3)))*)(p);
}
constexpr int F16X3 = 19;                                       // the ABI's `nterms` code of this form
// biased fp32 exponents.  A maximum with exponent e is scaled by 2^(E_ONE - e) into [2^14, 2^15); E_MIN floors e so that
// every scale and its inverse stay normal fp32 numbers (values below 2^-111 are flushed).  Both operands being normalised,
// an accumulator never exceeds K * 2^30 whatever the scales.  A value that INITIALISES an accumulator (a partial sum, the
// skip projection) enters the running maximum 2^-E_INIT_SHIFT-fold: it then sits below 2^100 in the scaled domain, and
// whenever that bound is what sets the scale, the products it pushes out of fp16's range lie below the value's own ulp.
constexpr int E_MIN = 16, E_ONE = 141, E_INIT_SHIFT = 85;


}  // namespace
