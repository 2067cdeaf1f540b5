// Shared helpers for the gfx950 kernels of libsemstereo_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/semstereo_hip.h"

namespace ss {

constexpr int kWave = 64;  // CDNA wavefront

// Records the text of the last launch failure on this thread (see ss_last_hip_error()).
void note_hip_error(hipError_t e);

inline int check_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        note_hip_error(e);
        return SS_ERR_LAUNCH;
    }
    return SS_OK;
}

inline hipStream_t as_stream(ss_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Tuning switches (A/B measurements and the tests that cover both forms of a kernel): read from the environment ONCE
// per process -- a launch costs no getenv() and an environment change mid-process cannot silently change which kernel
// runs -- and again only when ss_reload_tuning() is called.  -1 = not set (the measured-best automatic choice).
struct Tuning {
    int conv_tile;        // SS_CONV_TILE   0..2: force a tile candidate of the conv kernels
    int conv_s2_mt1;      // SS_CONV_S2_MT1 (tuning aid): unset = the waves of a workgroup split its 64 channels; 0 = two channel tiles per wave (r02); 1 = one
    int gwc_stream;       // SS_GWC_STREAM  0/1: plain / nontemporal stores of the gwc volume
    int warp_stream;      // SS_WARP_STREAM 0/1
    int warp_vec4;        // SS_WARP_VEC=4
    int warp_generic;     // SS_WARP_GENERIC set: the generic warp kernel for the live form too
    int deconv_split;     // SS_DECONV_SPLIT 0/1: even/odd-plane split of the exact-fp32 transposed conv
    int deconv_groups;    // SS_DECONV_GROUPS (fp16 transposed convs): 0 = all 8 parity classes per workgroup, 1 = two class groups, 2 = + chunk-blocked accumulation; unset: by layer size
    int deconv_stream;    // SS_DECONV_STREAM 0/1: plain / nontemporal stores of the fp16 transposed convs' output; unset: by output size
    int wgrad_coop;       // SS_WGRAD_COOP=0: the per-wave form of the stride-1 bf16 weight gradient instead of the cooperative one (r06)
};
const Tuning& tuning();

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device) instead of once per launch.
int ensure_dynamic_lds(const void* kernel, int bytes);
// Workgroups of `kernel` (block size, dynamic LDS bytes) the hardware keeps resident per CU, and the CU count of the
// current device (hipOccupancyMaxActiveBlocksPerMultiprocessor / hipDeviceGetAttribute, asked once and cached): the grid
// size of the persistent kernels.
int resident_workgroups(const void* kernel, int block, int lds_bytes);

__host__ __device__ inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// Disparity ranges.  Every volume / regression entry point takes (dmin, ndisp): plane p holds disparity dmin + p.  The
// reference's signed op set (models/submodule.py) is dmin = -maxdisp, ndisp = 2 * maxdisp; the unsigned one
// (models/submodule_.py, the form models/SemStereo_WHU.py is written for) is dmin = 0, ndisp = maxdisp.
// range_halo: columns of halo per side an LDS tile needs (the largest |shift|), rounded up to a multiple of 4.
inline int range_halo(int dmin, int ndisp) {
    const int a = -dmin, b = dmin + ndisp - 1;
    const int m = a > b ? a : b;
    return m <= 0 ? 0 : (m + 3) / 4 * 4;
}
__host__ __device__ inline long long ceil_div_ll(long long a, long long b) { return (a + b - 1) / b; }

// Unfused multiply / add (the reference materialises the product tensor, i.e. rounds it,
// before reducing; keeping the two roundings makes several kernels bit-identical to it).
// hipcc's __fmul_rn / __fadd_rn are plain `a * b` / `a + b` (clang's __clang_hip_math.h without OCML_BASIC_ROUNDED_OPERATIONS)
// and HIP compiles with -ffp-contract=fast-honor-pragmas: through them a product was still fused into the add or subtract
// that consumed it (found r04 in the ISA of the warp kernels: `ix - floor(ix)` became fma(t, half_w, -floor(ix)) on the
// UNROUNDED product, which moved the bilinear weights by ~1e-5 at W = 256 and put the HIP path 2-3x further from the float64
// answer than the reference's own CPU arithmetic).  The contract(off) pragma takes the `contract` flag off these two
// instructions, so neither can become half of an fma after inlining.
__device__ __forceinline__ float mul_rn(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ float add_rn(float a, float b) {
#pragma clang fp contract(off)
    return a + b;
}
__device__ __forceinline__ float sub_rn(float a, float b) {
#pragma clang fp contract(off)
    return a - b;
}
// exp(x) for finite x <= ~88 to ~1.5 ulp in 6 VALU ops: 2^(x*log2e) on the hardware exp2 (v_exp_f32, 1 ulp),
// with the rounding error of the product and the low bits of log2(e) carried as a first-order correction
// (a plain __expf loses |x| * 6e-8 relative there).  Not for x = -inf (use expf).
__device__ __forceinline__ float exp_fast(float x) {
    const float L = 1.44269502162933349609375f, Llo = 1.925962991126617e-8f;    // log2(e) = L + Llo
    const float t = __fmul_rn(x, L);
    const float e = __fmaf_rn(x, L, -t) + x * Llo;
    const float p = __builtin_amdgcn_exp2f(t);
    return __fmaf_rn(p, e * 0.693147180559945f, p);
}

// fp32 add into LDS as ONE ds_add_f32 (no return value).  `p` must point into LDS: through a generic pointer the atomic is a FLAT
// instruction, and pointer arithmetic that leaves the LDS aperture on the way (a negative intermediate offset) faults the queue
// (HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION, r06).
// (No product kernel uses it any more -- ss::lds_owned_add* below -- it stays for the timing ablations that measured why.)
__device__ __forceinline__ void lds_add(float* p, float v) {
    typedef float __attribute__((address_space(3))) * lds_ptr_t;
    __hip_atomic_fetch_add((lds_ptr_t)(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// fp32 sums into an LDS buffer that ONE WAVE alone writes, without ds_add_f32.  On gfx950 that instruction retires 0.38 lanes per CU
// clock -- 170 clocks per wave instruction, the rate of the GLOBAL fp32 atomic, whatever the addresses (ds_add_u32: 14.8 lanes per
// clock; a plain read-add-write: 8-10; tools/exp_lds_atomic.hip, profiles/r06_lds_atomic.txt).  Lanes of one call may aim at the same
// slot, so each first writes its lane number to tag[k] and reads it back (the LDS executes a wave's instructions in order): the lane
// whose number survived adds, the others go round again.  The loop condition is a
// BALLOT on purpose: with a per-lane `while (pending)` the optimiser, which reasons per thread, sinks the update below the loop --
// every lane spins until it has won the tag and then all of them add at once, colliding lanes losing updates (seen in the ISA, r06).
// Every access to such a buffer goes through the volatile forms below for the same reason (a lane's own earlier store would be forwarded
// past another lane's add).  Pointers into LDS only.
typedef volatile float __attribute__((address_space(3))) * lds_vf_t;
typedef volatile int __attribute__((address_space(3))) * lds_vi_t;
// (the value is read in the same round trip as the tag: nobody else writes slot k in a round this lane wins)
__device__ __forceinline__ void lds_owned_add(int* tag, unsigned k, bool take, float* row, float v) {
    const int me = (int)__lane_id();
    bool pending = take;
    while (__builtin_amdgcn_ballot_w64(pending) != 0) {
        if (pending) {
            ((lds_vi_t)tag)[k] = me;
            const int won = ((lds_vi_t)tag)[k];
            const float a = ((lds_vf_t)row)[k];
            if (won == me) {
                ((lds_vf_t)row)[k] = a + v;
                pending = false;
            }
        }
    }
}
// ... into the same slot of two rows (`two` wave-uniform: false leaves row_b alone)
__device__ __forceinline__ void lds_owned_add2(int* tag, unsigned k, bool take, float* row_a, float va, float* row_b, float vb, bool two) {
    const int me = (int)__lane_id();
    bool pending = take;
    while (__builtin_amdgcn_ballot_w64(pending) != 0) {
        if (pending) {
            ((lds_vi_t)tag)[k] = me;
            const int won = ((lds_vi_t)tag)[k];
            const float a = ((lds_vf_t)row_a)[k];
            const float b = two ? ((lds_vf_t)row_b)[k] : 0.f;
            if (won == me) {
                ((lds_vf_t)row_a)[k] = a + va;
                if (two) ((lds_vf_t)row_b)[k] = b + vb;
                pending = false;
            }
        }
    }
}
// ... into the same slot of N rows `stride` floats apart (rows 0 .. nlive - 1 written; nlive wave-uniform)
template <int N>
__device__ __forceinline__ void lds_owned_addn(int* tag, unsigned k, bool take, float* row0, int stride, const float (&v)[N], int nlive) {
    const int me = (int)__lane_id();
    bool pending = take;
    if (nlive == N) {                                      // (kept apart: every `n < nlive` below is a scalar branch otherwise)
        while (__builtin_amdgcn_ballot_w64(pending) != 0) {
            if (pending) {
                ((lds_vi_t)tag)[k] = me;
                const int won = ((lds_vi_t)tag)[k];
                float a[N];
#pragma unroll
                for (int n = 0; n < N; ++n) a[n] = ((lds_vf_t)(row0 + n * stride))[k];
                if (won == me) {
#pragma unroll
                    for (int n = 0; n < N; ++n) ((lds_vf_t)(row0 + n * stride))[k] = a[n] + v[n];
                    pending = false;
                }
            }
        }
        return;
    }
    while (__builtin_amdgcn_ballot_w64(pending) != 0) {
        if (pending) {
            ((lds_vi_t)tag)[k] = me;
            const int won = ((lds_vi_t)tag)[k];
            float a[N];
#pragma unroll
            for (int n = 0; n < N; ++n) a[n] = (n < nlive) ? ((lds_vf_t)(row0 + n * stride))[k] : 0.f;
            if (won == me) {
#pragma unroll
                for (int n = 0; n < N; ++n)
                    if (n < nlive) ((lds_vf_t)(row0 + n * stride))[k] = a[n] + v[n];
                pending = false;
            }
        }
    }
}
// a plain (non-volatile) read through an explicit LDS pointer: keeps the optimiser from merging it with a global load of the other
// arm of a branch into one flat access (which the gfx950 backend then fails to select: "Operand has incorrect register class", r06)
__device__ __forceinline__ float lds_ld(const float* p) { return *(const float __attribute__((address_space(3)))*)p; }
__device__ __forceinline__ float lds_get(const float* p) { return *(lds_vf_t)p; }
__device__ __forceinline__ void lds_put(float* p, float v) { *(lds_vf_t)p = v; }

// 16-byte LDS read that stays ONE ds_read_b128.  Through a plain float4 the optimiser splits the load into scalars,
// drops unused lanes and re-merges the rest as 4- and 8-byte reads (gwc_patch_gate_v4: 56 LDS instructions per channel
// block instead of 24, and at a 16-byte lane stride those are bank conflicts -- 47 -> 37 us, tools/pmc_sq.sh).
// `p` must point into LDS and be 16-byte aligned.
typedef float lds_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void lds_read16(const float* p, float* dst) {
    typedef const volatile lds_v4f __attribute__((address_space(3))) * ptr_t;
    const lds_v4f q = *(ptr_t)(p);
    dst[0] = q.x; dst[1] = q.y; dst[2] = q.z; dst[3] = q.w;
}

}  // namespace ss

// attention_tail.hip: wave-split softmax+regression+variance; returns non-zero if D is out of its range
int ss_softmax_regress_split_launch(const float* logits, float* prob, float* disp, float* var, int B, int dmin, int ndisp,
                                    int H, int W, hipStream_t st);

#define SS_REQUIRE(cond)              \
    do {                                \
        if (!(cond)) return SS_ERR_INVALID; \
    } while (0)
