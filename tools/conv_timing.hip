// Instrumented build of the tiled conv kernel for tools/wg_phases.py: cycle-counter stamps at the phase boundaries of every
// workgroup (prologue / K loop / epilogue, and the K-steps alone), read back through ss_debug_read.  NOT part of the product
// library: tools/build_timing.sh compiles this file INSTEAD of semstereo_amd/csrc/conv3d_bf16s.hip into tools/_build/lib_timing.so.
#include <hip/hip_runtime.h>

__device__ unsigned long long ss_dbg_t[8 * 16384];
__device__ long long ss_dbg_steps_scratch;
extern "C" int ss_debug_read(unsigned long long* dst, int n) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(ss_dbg_t), (size_t)n * 8) == hipSuccess ? 0 : -1;
}
extern "C" int ss_debug_read_place(unsigned long long* dst, int n);
extern "C" int ss_debug_reset() {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(ss_dbg_t)) != hipSuccess) return -1;
    return hipMemset(p, 0, sizeof(unsigned long long) * 8 * 16384) == hipSuccess ? 0 : -1;
}
#define SS_STAMP_ON (threadIdx.x == 0 && blockIdx.x < 16384 && blockIdx.y == 0 && blockIdx.z == 0)
#define SS_STAMP(k) do { if (SS_STAMP_ON) ss_dbg_t[blockIdx.x * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
#define SS_STAMP_STEPS_BEGIN() const long long ss_tk0 = __builtin_readcyclecounter()
#define SS_STAMP_STEPS_END() do { if (SS_STAMP_ON) ss_dbg_t[blockIdx.x * 8 + 5] += (unsigned long long)(__builtin_readcyclecounter() - ss_tk0); } while (0)
// slot 4: HW_ID (wave / SIMD / CU / SH / SE of the stamping wave), slot 6: XCC_ID; slot 7 (EVERY workgroup of the launch, indexed
// by its linear id, in a second table): the same two packed, for tools/wg_placement.py
__device__ unsigned long long ss_dbg_place[65536];
#define SS_STAMP_FINISH() do { if (threadIdx.x == 0) { unsigned hw, xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); \
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); \
    const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z); \
    if (lin < 65536) ss_dbg_place[lin] = ((unsigned long long)xcc << 32) | hw | (1ull << 63); \
    if (SS_STAMP_ON) { ss_dbg_t[blockIdx.x * 8 + 4] = hw; ss_dbg_t[blockIdx.x * 8 + 6] = xcc; } } } while (0)

#include "../semstereo_amd/csrc/conv3d_bf16s.hip"
extern "C" int ss_debug_read_place(unsigned long long* dst, int n) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(ss_dbg_place), (size_t)n * 8) == hipSuccess ? 0 : -1;
}
extern "C" int ss_debug_reset_place() {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(ss_dbg_place)) != hipSuccess) return -1;
    return hipMemset(p, 0, sizeof(unsigned long long) * 65536) == hipSuccess ? 0 : -1;
}
