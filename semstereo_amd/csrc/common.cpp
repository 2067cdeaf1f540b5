// Status strings and the per-thread HIP error note of libsemstereo_hip.so.
#include "common.h"

#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <unordered_map>

namespace {
thread_local char g_last_error[256] = "";

int env_int(const char* name) {            // -1: unset; otherwise its integer value (a set but non-numeric variable: 1)
    const char* v = getenv(name);
    if (v == nullptr) return -1;
    return ((v[0] >= '0' && v[0] <= '9') || (v[0] == '-' && v[1] >= '0' && v[1] <= '9')) ? atoi(v) : 1;      // ("-1": as if unset)
}

ss::Tuning read_tuning() {
    ss::Tuning t;
    t.conv_tile = env_int("SS_CONV_TILE");
    t.conv_s2_mt1 = env_int("SS_CONV_S2_MT1");
    t.gwc_stream = env_int("SS_GWC_STREAM");
    t.warp_stream = env_int("SS_WARP_STREAM");
    t.warp_vec4 = env_int("SS_WARP_VEC") == 4 ? 1 : 0;
    t.warp_generic = env_int("SS_WARP_GENERIC");
    t.deconv_split = env_int("SS_DECONV_SPLIT");
    t.deconv_groups = env_int("SS_DECONV_GROUPS");
    if (t.deconv_groups > 2) t.deconv_groups = -1;
    t.deconv_stream = env_int("SS_DECONV_STREAM");
    t.wgrad_coop = env_int("SS_WGRAD_COOP");
    return t;
}

// two slots, published by index: a reload never writes the slot a concurrent launch may be reading
ss::Tuning g_tuning[2];
std::atomic<int> g_tuning_slot{-1};
std::mutex g_mutex;
std::unordered_map<unsigned long long, int> g_lds_set;      // (kernel address ^ device) -> bytes granted
}

namespace ss {
void note_hip_error(hipError_t e) {
    const char* s = hipGetErrorString(e);
    strncpy(g_last_error, s ? s : "unknown hip error", sizeof(g_last_error) - 1);
    g_last_error[sizeof(g_last_error) - 1] = 0;
}

const Tuning& tuning() {
    int slot = g_tuning_slot.load(std::memory_order_acquire);
    if (slot < 0) {
        std::lock_guard<std::mutex> lock(g_mutex);
        slot = g_tuning_slot.load(std::memory_order_relaxed);
        if (slot < 0) {
            g_tuning[0] = read_tuning();
            g_tuning_slot.store(slot = 0, std::memory_order_release);
        }
    }
    return g_tuning[slot];
}

int ensure_dynamic_lds(const void* kernel, int bytes) {
    if (bytes <= 64 * 1024) return SS_OK;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    const unsigned long long key = (unsigned long long)reinterpret_cast<uintptr_t>(kernel) * 64ull + (unsigned)dev;
    std::lock_guard<std::mutex> lock(g_mutex);
    auto it = g_lds_set.find(key);
    if (it != g_lds_set.end() && it->second >= bytes) return SS_OK;
    hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
        note_hip_error(e);
        return SS_ERR_LAUNCH;
    }
    g_lds_set[key] = bytes;
    return SS_OK;
}

int resident_workgroups(const void* kernel, int block, int lds_bytes) {
    static std::unordered_map<unsigned long long, int> cache;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    const unsigned long long key = (unsigned long long)reinterpret_cast<uintptr_t>(kernel) * 64ull + (unsigned)dev;
    std::lock_guard<std::mutex> lock(g_mutex);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block, (size_t)lds_bytes) != hipSuccess || per_cu < 1) per_cu = 2;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
    (void)hipGetLastError();
    return cache[key] = per_cu * cus;
}
}  // namespace ss

extern "C" int ss_reload_tuning(void) {
    std::lock_guard<std::mutex> lock(g_mutex);
    const int cur = g_tuning_slot.load(std::memory_order_relaxed);
    const int nxt = cur == 0 ? 1 : 0;
    g_tuning[nxt] = read_tuning();
    g_tuning_slot.store(nxt, std::memory_order_release);
    return SS_OK;
}

extern "C" int ss_abi_version(void) { return 19; }   // 19: ss_batchnorm_bwd_pg takes the forward's bias (ReLU mask from x when y is NULL); 18: + ss_group_normalise_fwd / _bwd; 17: + ss_concat_sampled_bwd (the sparse concat volume's backward in one pass); 16: + ss_batchnorm_train_fwd_rs, ss_batchnorm_bwd_pg (running statistics and float parameter gradients inside the kernels); 15: + ss_sample_strength_bwd_ws; 14: + ss_regression_topk_patched_fwd (the classifier's patch sum folded into the top-2 soft-argmax); 13: + ss_conv3d_wgrad_bf16s_fwd (weight gradients on the bf16 matrix core); 12: + ss_conv3d_gather_fwd (sparse concat formed inside concat_stem), ss_ssr_upsample2_fwd; 11: + training leftovers (ss_batchnorm_train_res_fwd/_bwd: residual + ReLU inside the BatchNorm apply; ss_window_attention_core_pad_bwd; the attention-tail backward kernels); 10: + training side (ss_batchnorm_train_fwd/_bwd, ss_channel_sum_fwd, ss_depthwise_patch_wgrad_fwd, ss_channel_gate_bwd_logits, ss_window_attention_core_bwd), ss_tool_copy_fwd; 9: + ss_concat_sampled_presplit_fwd, ss_conv3d_presplit_fwd (pre-split operands, LDS-DMA staging); 8: channels-last hand-off inside the classifiers (ss_conv3d_bf16s_cl_fwd, ss_conv3d_head_bf16s_cl_fwd); 7: disparity ranges (dmin, ndisp) instead of maxdisp, ss_conv3d_wgrad_fwd, backward entry points; 6: + ss_channel_att_logits_fwd, ss_upsample_softmax_regression_fwd (5: ss_reload_tuning)

extern "C" const char* ss_status_string(int status) {
    switch (status) {
        case SS_OK: return "ok";
        case SS_ERR_INVALID: return "invalid argument (null pointer, non-positive size, or C % groups != 0)";
        case SS_ERR_UNSUPPORTED: return "shape not supported by the gfx950 kernels";
        case SS_ERR_LAUNCH: return "kernel launch failed (see ss_last_hip_error)";
        default: return "unknown status";
    }
}

extern "C" const char* ss_last_hip_error(void) { return g_last_error; }
