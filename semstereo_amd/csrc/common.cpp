// Status strings and the per-thread HIP error note of libsemstereo_hip.so.
#include "common.h"

#include <string.h>

namespace {
thread_local char g_last_error[256] = "";
}

namespace ss {
void note_hip_error(hipError_t e) {
    const char* s = hipGetErrorString(e);
    strncpy(g_last_error, s ? s : "unknown hip error", sizeof(g_last_error) - 1);
    g_last_error[sizeof(g_last_error) - 1] = 0;
}
}  // namespace ss

extern "C" int ss_abi_version(void) { return 4; }   // 4: + two-term fp16 forms (nterms = 19, ss_pack_*_f16s), 2-D conv

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
