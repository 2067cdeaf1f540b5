// Instrumented build of stem_left.hip for tools/wg_phases_stem_left.py: per workgroup (thread 0), cycles spent per phase of the
// channel loop of stem_left_overlap, summed over the channels: [0] Q reads + multiply-adds (+ the interleaved matrix instructions),
// [1] issuing the stores + parking Q, [2] waiting at the barrier, [4] whole loop.  NOT part of the product
// library: tools/build_variant.sh-style link (tools/build_timing_stem_left.sh).
#include <hip/hip_runtime.h>

__device__ unsigned long long sl_dbg_t[8 * 4096];
extern "C" int ss_debug_read_sl(unsigned long long* dst, int n) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(sl_dbg_t), (size_t)n * 8) == hipSuccess ? 0 : -1;
}
extern "C" int ss_debug_reset_sl() {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(sl_dbg_t)) != hipSuccess) return -1;
    return hipMemset(p, 0, sizeof(unsigned long long) * 8 * 4096) == hipSuccess ? 0 : -1;
}
#define SL_STAMP_DECL() sl_begin = __builtin_readcyclecounter()
#define SL_STAMP(k) do { sl_t[k] = __builtin_readcyclecounter(); if ((k) > 0) sl_acc[(k) - 1] += sl_t[k] - sl_t[(k) - 1]; } while (0)
#define SL_STAMP_FINISH() do { const int wg = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x; \
    if (threadIdx.x == 0 && wg < 4096) { for (int i = 0; i < 4; ++i) sl_dbg_t[wg * 8 + i] = (unsigned long long)sl_acc[i]; \
    sl_dbg_t[wg * 8 + 4] = (unsigned long long)(__builtin_readcyclecounter() - sl_begin); sl_dbg_t[wg * 8 + 5] = (unsigned long long)sl_begin; } } while (0)

#define SL_STAMP_VARS long long sl_t[5] = {0, 0, 0, 0, 0}, sl_acc[4] = {0, 0, 0, 0}, sl_begin = 0
#include "../semstereo_amd/csrc/stem_left.hip"
