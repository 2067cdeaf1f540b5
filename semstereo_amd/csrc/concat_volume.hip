// Dense concatenation cost volume over a disparity range [dmin, dmin + D) (gfx950): plane p holds disparity s = dmin + p.
//
//   out[b,   c, p,y,x] = ref[b,c,y,x]            if 0 <= x-s < W else 0          (unmasked with mask_left == 0)
//   out[b, C+c, p,y,x] = tgt[b,c,y,x-s]          if 0 <= x-s < W else 0
// The reference's signed form (models/submodule.py:173-187) is dmin = -maxdisp, D = 2*maxdisp, both halves masked; its
// unsigned form (models/submodule_.py:166-177, the op set models/SemStereo_WHU.py needs) is dmin = 0, D = maxdisp with the
// left half copied unmasked.  Below, mh = the halo of the LDS tile per side, off = mh - dmin.
//
// Replaces build_concat_volume (reference models/submodule.py:173-187).  Pure data movement:
// 4*(2*C + 2*C*2m)*H*W bytes per pair, all of it coalesced 16-B-per-lane traffic: one workgroup
// owns (b, c, 8 rows, 128 columns), parks the right-image rows (+ zero halo) in LDS once and
// streams the 2m shifted copies out, 8 disparities per three ds_read_b128.
#include <algorithm>

#include "common.h"

namespace {

constexpr int XT = 128, RT = 8;

__global__ __launch_bounds__(256) void concat_volume_v4(const float* __restrict__ ref, const float* __restrict__ tgt,
                                                         float* __restrict__ out, int C, int H, int W, int mh, int dmin,
                                                         int D, int mask_left) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [RT][LW]
    const int m = mh, off = mh - dmin;
    const int LW = XT + 2 * m, LQ = LW / 4;
    const int tid = threadIdx.x;
    const int xt0 = blockIdx.x * XT, y0 = blockIdx.y * RT;
    const int b = blockIdx.z / C, c = blockIdx.z % C;
    const size_t plane = (size_t)H * W;
    const float* refp = ref + ((size_t)b * C + c) * plane;
    const float* tgtp = tgt + ((size_t)b * C + c) * plane;
    for (int q = tid; q < RT * LQ; q += 256) {
        const int row = q / LQ, qi = q - row * LQ;
        const int col0 = xt0 - m + qi * 4, y = y0 + row;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (y < H && col0 >= 0 && col0 < W) v = *reinterpret_cast<const float4*>(tgtp + (size_t)y * W + col0);
        *reinterpret_cast<float4*>(&lds[row * LW + qi * 4]) = v;
    }
    const int tx = tid & 31, ty = tid >> 5;
    const int x0 = xt0 + tx * 4, y = y0 + ty;
    const bool active = (y < H) && (x0 < W);
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (active) r = *reinterpret_cast<const float4*>(refp + (size_t)y * W + x0);
    __syncthreads();
    if (!active) return;
    float* outl = out + ((((size_t)b * 2 * C + c) * D) * H + y) * W + x0;
    float* outr = out + ((((size_t)b * 2 * C + C + c) * D) * H + y) * W + x0;
    for (int d0 = 0; d0 < D; d0 += 8) {
        const float* lp = &lds[ty * LW + tx * 4 + off - d0 - 8];
        float w[12];
        *reinterpret_cast<float4*>(&w[0]) = *reinterpret_cast<const float4*>(lp);
        *reinterpret_cast<float4*>(&w[4]) = *reinterpret_cast<const float4*>(lp + 4);
        *reinterpret_cast<float4*>(&w[8]) = *reinterpret_cast<const float4*>(lp + 8);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int d = d0 + i;
            const int col = x0 - (d + dmin);
            const bool v0 = (unsigned)(col + 0) < (unsigned)W, v1 = (unsigned)(col + 1) < (unsigned)W;
            const bool v2 = (unsigned)(col + 2) < (unsigned)W, v3 = (unsigned)(col + 3) < (unsigned)W;
            float4 a, t;
            a.x = (v0 || !mask_left) ? r.x : 0.f; a.y = (v1 || !mask_left) ? r.y : 0.f;
            a.z = (v2 || !mask_left) ? r.z : 0.f; a.w = (v3 || !mask_left) ? r.w : 0.f;
            t.x = v0 ? w[8 - i] : 0.f; t.y = v1 ? w[9 - i] : 0.f; t.z = v2 ? w[10 - i] : 0.f; t.w = v3 ? w[11 - i] : 0.f;
            *reinterpret_cast<float4*>(outl + (size_t)d * plane) = a;
            *reinterpret_cast<float4*>(outr + (size_t)d * plane) = t;
        }
    }
}

// any W / maxdisp: one element per thread, grid-stride over the OUTPUT.
__global__ void concat_volume_generic(const float* __restrict__ ref, const float* __restrict__ tgt,
                                      float* __restrict__ out, int C, int H, int W, int dmin, int D, int mask_left,
                                      long long total) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        long long t = i / W;
        const int y = (int)(t % H); t /= H;
        const int d = (int)(t % D); t /= D;
        const int c2 = (int)(t % (2 * C));
        const long long b = t / (2 * C);
        const int col = x - (d + dmin);
        float v = 0.f;
        if (c2 < C) {
            if ((unsigned)col < (unsigned)W || !mask_left) v = ref[((b * C + c2) * H + y) * W + x];
        } else if ((unsigned)col < (unsigned)W) {
            v = tgt[((b * C + (c2 - C)) * H + y) * W + col];
        }
        out[i] = v;
    }
}

__global__ void concat_volume_bwd_kernel(const float* __restrict__ gout, float* __restrict__ gref,
                                         float* __restrict__ gtgt, int C, int H, int W, int dmin, int D, int mask_left,
                                         long long total) {
    const long long plane = (long long)H * W;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        const long long row = i / W;
        const int y = (int)(row % H);
        const long long bc = row / H;
        const int c = (int)(bc % C);
        const long long b = bc / C;
        const float* gl = gout + ((b * 2 * C + c) * D) * plane + (long long)y * W;
        const float* gr = gout + ((b * 2 * C + C + c) * D) * plane + (long long)y * W;
        float ar = 0.f, at = 0.f;
        for (int d = 0; d < D; ++d) {
            const int s = d + dmin;
            if ((unsigned)(x - s) < (unsigned)W || !mask_left) ar += gl[d * plane + x];
            if ((unsigned)(x + s) < (unsigned)W) at += gr[d * plane + x + s];
        }
        gref[i] = ar;
        gtgt[i] = at;
    }
}

}  // namespace

// ---- HBM calibration: a plain 16-byte-per-lane device copy, what SURVEY.md section 8(d) asks the bandwidth fractions to be read
// against on THIS box (torch's copy_ reaches 64 % of the 8 TB/s spec, a float4 copy 79 %: MI355X_MICROARCH.md) ----
namespace {
typedef unsigned copy_u4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void copy16_kernel(const copy_u4* __restrict__ src, copy_u4* __restrict__ dst, long long n) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const copy_u4 v = __builtin_nontemporal_load(src + i);
        __builtin_nontemporal_store(v, dst + i);
    }
}
}  // namespace

extern "C" int ss_tool_copy_fwd(const void* src, void* dst, long long bytes, ss_stream_t stream) {
    SS_REQUIRE(src && dst && bytes > 0 && bytes % 16 == 0);
    SS_REQUIRE(((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0);
    const long long n = bytes / 16;
    const long long blocks = std::min<long long>(ss::ceil_div_ll(n, 256), 256LL * 32);       // grid-stride: 32 workgroups per CU
    hipLaunchKernelGGL(copy16_kernel, dim3((unsigned)blocks), dim3(256), 0, ss::as_stream(stream),
                       reinterpret_cast<const copy_u4*>(src), reinterpret_cast<copy_u4*>(dst), n);
    return ss::check_launch();
}

extern "C" int ss_concat_volume_fwd(const float* ref, const float* tgt, float* out, int B, int C, int H, int W,
                                    int dmin, int ndisp, int mask_left, ss_stream_t stream) {
    SS_REQUIRE(ref && tgt && out);
    SS_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && ndisp > 0);
    hipStream_t st = ss::as_stream(stream);
    const int mh = ss::range_halo(dmin, ndisp);
    const bool aligned = ((reinterpret_cast<uintptr_t>(ref) | reinterpret_cast<uintptr_t>(tgt) |
                           reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    if (aligned && W % 4 == 0 && dmin % 4 == 0 && ndisp % 8 == 0 && (long long)B * C <= 65535 &&
        (size_t)RT * (XT + 2 * mh) * 4 <= 64 * 1024) {
        dim3 grid(ss::ceil_div(W, XT), ss::ceil_div(H, RT), B * C);
        hipLaunchKernelGGL(concat_volume_v4, grid, dim3(256), (size_t)RT * (XT + 2 * mh) * sizeof(float), st, ref, tgt,
                           out, C, H, W, mh, dmin, ndisp, mask_left);
        return ss::check_launch();
    }
    const long long total = (long long)B * 2 * C * ndisp * H * W;
    const int blocks = (int)std::min<long long>(ss::ceil_div_ll(total, 256), 256 * 32);
    hipLaunchKernelGGL(concat_volume_generic, dim3(blocks), dim3(256), 0, st, ref, tgt, out, C, H, W, dmin, ndisp, mask_left, total);
    return ss::check_launch();
}

extern "C" int ss_concat_volume_bwd(const float* grad_out, float* grad_ref, float* grad_tgt, int B, int C, int H,
                                    int W, int dmin, int ndisp, int mask_left, ss_stream_t stream) {
    SS_REQUIRE(grad_out && grad_ref && grad_tgt);
    SS_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && ndisp > 0);
    const long long total = (long long)B * C * H * W;
    const int blocks = (int)std::min<long long>(ss::ceil_div_ll(total, 256), 256 * 32);
    hipLaunchKernelGGL(concat_volume_bwd_kernel, dim3(blocks), dim3(256), 0, ss::as_stream(stream), grad_out, grad_ref,
                       grad_tgt, C, H, W, dmin, ndisp, mask_left, total);
    return ss::check_launch();
}
