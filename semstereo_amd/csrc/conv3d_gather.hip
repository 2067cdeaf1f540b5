// concat_stem with the sparse concat volume formed INSIDE its staging (second half; reference
// models/SemStereo.py:241-244, 316-320, models/submodule.py:265-288): the GATHER form of conv3d_bf16s, instantiated in its own
// translation unit so that the two files compile side by side.  See the kernel's header comment in conv3d_bf16s.hip.
#define SS_CONV_GATHER_TU 1
#include "conv3d_bf16s.hip"

namespace {

template <int NT, int TD, int TH>
int launch_gather(const float* right, const float* cand, const float* att, const void* wsplit, const float* partial,
                  const float* scale, const float* shift, const float* gate, float* out, int B, int Cin, int nd, int H, int W,
                  int Cout, int relu, bool accb, hipStream_t st) {
#define SS_G(GATED, ACCB)                                                                                                    \
    return launch_bgm<1, NT, TD, TH, F16X3, GATED, 1, 3, 1, ACCB, true>(right, wsplit, scale, shift, partial, gate, out, B, Cin, nd, \
                                                                       H, W, Cout, relu, st, cand, att)
    if (gate != nullptr) {
        if (accb) { SS_G(true, true); }
        SS_G(true, false);
    }
    if (accb) { SS_G(false, true); }
    SS_G(false, false);
#undef SS_G
}

}  // namespace

extern "C" int ss_conv3d_gather_fwd(const float* right, const float* cand, const float* att, const void* wsplit,
                                    const float* partial, const float* scale, const float* shift, const float* gate, float* out,
                                    int B, int Cin, int nd, int H, int W, int Cout, int relu, int nterms, ss_stream_t stream) {
    SS_REQUIRE(right && cand && att && wsplit && out);
    SS_REQUIRE(B > 0 && Cin >= 16 && Cin % 8 == 0 && nd > 0 && H > 0 && W > 0 && Cout > 0);
    SS_REQUIRE((reinterpret_cast<uintptr_t>(wsplit) & 15) == 0);
    if (nterms != F16X3) return SS_ERR_UNSUPPORTED;             // the two-term fp16 engine (the default) only
    if ((long long)Cin * H * W * 4 >= 0x7fffffffLL || (long long)nd * H * W * 4 >= 0x7fffffffLL) return SS_ERR_UNSUPPORTED;
    if ((long long)Cout * nd * H * W * 4 >= 0x7fffffffLL) return SS_ERR_UNSUPPORTED;
    hipStream_t st = ss::as_stream(stream);
    // tile choice and chunk-blocked accumulation: the rules of ss_conv3d_bf16s_fwd for a stride-1 layer of this output shape
    auto blocks = [&](int td, int th) {
        return (long long)ss::ceil_div(W, 32) * ss::ceil_div(H, th) * ss::ceil_div(nd, td) * ss::ceil_div(Cout, 32) * B;
    };
    const int forced = ss::tuning().conv_tile;
    int tile = (blocks(2, 8) >= 512) ? 0 : ((blocks(1, 8) >= 512) ? 1 : 2);
    if (forced >= 0 && forced <= 2) tile = forced;
    const bool accb = SS_ACC_BLOCKED && blocks(2, 8) / B < 512;
    const int r = (relu ? 1 : 0) | (partial ? 2 : 0);
#define SS_T(NT, TD, TH) return launch_gather<NT, TD, TH>(right, cand, att, wsplit, partial, scale, shift, gate, out, B, Cin, nd, H, W, Cout, r, accb, st)
    if (tile == 0 && nd % 4 == 0) { SS_T(4, 4, 4); }
    if (tile == 0) { SS_T(4, 2, 8); }
    if (tile == 1) { SS_T(2, 1, 8); }
    SS_T(1, 1, 4);
#undef SS_T
}
