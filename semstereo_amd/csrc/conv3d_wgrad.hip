// Weight gradient of the 3x3x3 convolutions of the aggregation stack (training: main_us3d.py:186-222 back-propagates through
// convbn_3d / BasicConv / the hourglasses' ConvTranspose3d layers, models/SemStereo.py:106-182, 228-236):
//
//   dW[co, ci, kd, kh, kw] = sum_{b, od, oh, ow} gout[b, co, od, oh, ow] * in[b, ci, od*S + kd - 1, oh*S + kh - 1, ow*S + kw - 1]
//
// (zero padding 1, stride S = 1 or 2).  The same kernel serves ConvTranspose3d(k3, s2, p1, op1): its weight gradient is the
// S = 2 form with the roles swapped (gout := the layer's INPUT, in := the gradient of its output), which lands directly in
// the [Cin, Cout, 3, 3, 3] layout of that weight.
//
// A GEMM with M = Cout, N = Cin per tap and a very long K (every output position): exact-fp32 MFMA
// (v_mfma_f32_32x32x2_f32, K = 2 positions per instruction) -- gradients of a training step do not need the split-operand
// machinery of the forward kernels, and the fp32 matrix pipe keeps them exact to accumulation order.  A wave owns one
// kernel-depth plane (9 taps = 9 accumulator tiles of 32 x 32) of one (32 output channels x 32 input channels) tile and
// walks its share of the output rows two positions at a time; operands come straight from global memory through buffer
// loads (lane = channel, so a wave touches 32 cache lines per load that the next iterations re-use from the L1; taps and
// positions outside the tensors read zeros from beyond the buffer).  The K range is split over the grid and the partial
// tiles are combined with hardware fp32 atomics (summation order, hence the last bits, vary from run to run -- like the
// vendor libraries' default weight-gradient algorithms).
#include <algorithm>

#include "common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int S>
__global__ __launch_bounds__(192) void conv3d_wgrad_k3(const float* __restrict__ gout, const float* __restrict__ in,
                                                        float* __restrict__ dw, int Cin, int Cout, int D, int H, int W, int Do,
                                                        int Ho, int Wo, int rows_per_wg, int total_rows, int ci_tiles) {
    const int lane = threadIdx.x & 63, kd = threadIdx.x >> 6;               // wave = kernel depth plane
    const int l31 = lane & 31, half = lane >> 5;
    const int co0 = (blockIdx.y / ci_tiles) * 32, ci0 = (blockIdx.y % ci_tiles) * 32;
    const int b = blockIdx.z;
    const long long ochan = (long long)Do * Ho * Wo, ichan = (long long)D * H * W;
    const __amdgpu_buffer_rsrc_t gres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(gout + (long long)b * Cout * ochan), 0, (int)min((long long)Cout * ochan * 4, 0x7fffffffLL), 0x00020000);
    const __amdgpu_buffer_rsrc_t ires = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(in + (long long)b * Cin * ichan), 0, (int)min((long long)Cin * ichan * 4, 0x7fffffffLL), 0x00020000);
    const bool co_ok = co0 + l31 < Cout, ci_ok = ci0 + l31 < Cin;
    const unsigned a_lane = co_ok ? (unsigned)((co0 + l31) * ochan * 4) : 0x80000000u;       // + position
    const unsigned b_lane = ci_ok ? (unsigned)((ci0 + l31) * ichan * 4) : 0x80000000u;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int r0 = blockIdx.x * rows_per_wg, r1 = min(r0 + rows_per_wg, total_rows);
    for (int row = r0; row < r1; ++row) {
        const int od = row / Ho, oh = row - od * Ho;
        const int id = od * S + kd - 1;
        if ((unsigned)id >= (unsigned)D) continue;                          // this wave's depth tap is in the padding: zero
        const unsigned a_row = (unsigned)(((long long)od * Ho + oh) * Wo * 4);
        for (int p0 = 0; p0 < Wo; p0 += 8) {                                // four position pairs per pass: 40 loads in flight
            float a[4], bv[4][9];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int ow = p0 + 2 * q + half;
                const bool ow_ok = ow < Wo;
                a[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                     gres, (int)((ow_ok ? a_lane : 0x80000000u) + (unsigned)(ow * 4)), (int)a_row, 0));
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    const int ih = oh * S + kh - 1;
                    const bool ih_ok = (unsigned)ih < (unsigned)H;          // wave-uniform
                    const unsigned b_row = (unsigned)((((long long)id * H + (ih_ok ? ih : 0)) * W) * 4);
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        const int iw = ow * S + kw - 1;
                        const bool ok = ih_ok && ow_ok && (unsigned)iw < (unsigned)W;
                        bv[q][kh * 3 + kw] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                                           ires, (int)((ok ? b_lane : 0x80000000u) + (unsigned)(max(iw, 0) * 4)),
                                                                           (int)b_row, 0));
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], bv[q][t], acc[t], 0, 0, 0);
        }
    }
    // D layout of the 32x32 tile: register r of lane (l31, half) = row (r & 3) + 8 * (r >> 2) + 4 * half (output channel), column l31
    if (ci_ok) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (co < Cout) unsafeAtomicAdd(dw + ((long long)co * Cin + ci0 + l31) * 27 + kd * 9 + t, acc[t][r]);
            }
    }
}

// ---- 1x1x1 convolutions (redir1 / redir2, attention_block.qkv_3d / final1x1, channelAtt.im_att):
//   dW[co, ci] = sum_{b, pos} gout[b, co, pos] * in[b, ci, pos]
// A [32 x 32] tile per workgroup, the (long) K range = positions split over the grid.  Both operands are staged 128 positions at
// a time through LDS with coalesced 16-byte loads along the position axis (row stride 129 floats: the MFMA's lane = channel
// reads are then conflict-free across channels); each of the 4 waves multiplies 32 of the 128 positions
// (v_mfma_f32_32x32x2_f32, two positions per instruction) and adds its tile to dW with hardware fp32 atomics.
__global__ __launch_bounds__(256) void conv_wgrad_k1(const float* __restrict__ gout, const float* __restrict__ in,
                                                      float* __restrict__ dw, int Cin, int Cout, long long npos, long long pos_per_wg,
                                                      int ci_tiles) {
    constexpr int LS = 129;
    __shared__ float ga[32 * LS], xa[32 * LS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int co0 = (blockIdx.y / ci_tiles) * 32, ci0 = (blockIdx.y % ci_tiles) * 32;
    const int b = blockIdx.z;
    const float* gb = gout + (long long)b * Cout * npos;
    const float* xb = in + (long long)b * Cin * npos;
    const long long p0 = (long long)blockIdx.x * pos_per_wg, p1 = min(p0 + pos_per_wg, npos);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int row = tid >> 3, col = (tid & 7) * 16;                 // this thread stages 16 consecutive positions of one channel row
    // (r06) whole 128-position blocks of 16-byte-aligned rows: four 16-byte loads per operand, the NEXT block's issued before this block's
    // MFMAs (the scalar form below -- 32 four-byte loads between two barriers, nothing in flight across them -- ran at 2.8 TB/s of
    // traffic: 1.7 ms of the 1024^2 training step)
    const bool vec = (npos & 3) == 0 && (p0 & 3) == 0 && ((reinterpret_cast<uintptr_t>(gout) | reinterpret_cast<uintptr_t>(in)) & 15) == 0;
    if (vec) {
        const bool gok = co0 + row < Cout, xok = ci0 + row < Cin;
        const float* gr = gb + (long long)min(co0 + row, Cout - 1) * npos;
        const float* xr = xb + (long long)min(ci0 + row, Cin - 1) * npos;
        float4 gq[4], xq[4];
        auto fetch = [&](long long p) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const long long pp = p + col + 4 * k;
                const bool ok = pp < p1;                            // (p1 and pp are multiples of 4: a word is inside or outside as a whole)
                gq[k] = (ok && gok) ? *reinterpret_cast<const float4*>(gr + pp) : make_float4(0.f, 0.f, 0.f, 0.f);
                xq[k] = (ok && xok) ? *reinterpret_cast<const float4*>(xr + pp) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };
        fetch(p0);
        for (long long p = p0; p < p1; p += 128) {
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float* gd = &ga[row * LS + col + 4 * k];
                float* xd = &xa[row * LS + col + 4 * k];
                gd[0] = gq[k].x; gd[1] = gq[k].y; gd[2] = gq[k].z; gd[3] = gq[k].w;
                xd[0] = xq[k].x; xd[1] = xq[k].y; xd[2] = xq[k].z; xd[3] = xq[k].w;
            }
            __syncthreads();
            if (p + 128 < p1) fetch(p + 128);
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int c = 32 * wave + 2 * k + half;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[l31 * LS + c], xa[l31 * LS + c], acc, 0, 0, 0);
            }
        }
    } else
    for (long long p = p0; p < p1; p += 128) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const long long pp = p + col + k;
            const bool ok = pp < p1;
            ga[row * LS + col + k] = (ok && co0 + row < Cout) ? gb[(long long)(co0 + row) * npos + pp] : 0.f;
            xa[row * LS + col + k] = (ok && ci0 + row < Cin) ? xb[(long long)(ci0 + row) * npos + pp] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int c = 32 * wave + 2 * k + half;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[l31 * LS + c], xa[l31 * LS + c], acc, 0, 0, 0);
        }
    }
    if (ci0 + l31 < Cin) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (co < Cout) unsafeAtomicAdd(dw + (long long)co * Cin + ci0 + l31, acc[r]);
        }
    }
}

}  // namespace

// Weight gradient of a 1x1(x1) convolution: grad_out [B,Cout,npos], in [B,Cin,npos] -> grad_w [Cout,Cin].
extern "C" int ss_conv_k1_wgrad_fwd(const float* grad_out, const float* in, float* grad_w, int B, int Cin, int Cout, long long npos,
                                    ss_stream_t stream) {
    SS_REQUIRE(grad_out && in && grad_w);
    SS_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && npos > 0 && B <= 65535);
    hipStream_t st = ss::as_stream(stream);
    if (hipMemsetAsync(grad_w, 0, (size_t)Cout * Cin * sizeof(float), st) != hipSuccess) return SS_ERR_LAUNCH;
    const int ci_tiles = ss::ceil_div(Cin, 32), tiles = ci_tiles * ss::ceil_div(Cout, 32);
    if (tiles > 65535) return SS_ERR_UNSUPPORTED;
    long long splits = std::max<long long>(1, 2048 / ((long long)tiles * B));
    long long pos_per_wg = std::max<long long>(128, ss::ceil_div_ll(ss::ceil_div_ll(npos, splits), 128) * 128);
    const long long gx = ss::ceil_div_ll(npos, pos_per_wg);
    if (gx > 0x7fffffffLL) return SS_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(conv_wgrad_k1, dim3((unsigned)gx, tiles, B), dim3(256), 0, st, grad_out, in, grad_w, Cin, Cout, npos, pos_per_wg,
                       ci_tiles);
    return ss::check_launch();
}

extern "C" int ss_conv3d_wgrad_fwd(const float* grad_out, const float* in, float* grad_w, int B, int Cin, int D, int H, int W,
                                   int Cout, int stride, ss_stream_t stream) {
    SS_REQUIRE(grad_out && in && grad_w);
    SS_REQUIRE(B > 0 && Cin > 0 && D > 0 && H > 0 && W > 0 && Cout > 0 && (stride == 1 || stride == 2));
    const int Do = (D - 1) / stride + 1, Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    if ((long long)Cin * D * H * W * 4 >= 0x7fffffffLL || (long long)Cout * Do * Ho * Wo * 4 >= 0x7fffffffLL || B > 65535)
        return SS_ERR_UNSUPPORTED;                                          // 32-bit buffer offsets per batch element
    hipStream_t st = ss::as_stream(stream);
    if (hipMemsetAsync(grad_w, 0, (size_t)Cout * Cin * 27 * sizeof(float), st) != hipSuccess) return SS_ERR_LAUNCH;
    const int ci_tiles = ss::ceil_div(Cin, 32), tiles = ci_tiles * ss::ceil_div(Cout, 32);
    const int total_rows = Do * Ho;
    // split the rows so that the grid has ~2048 workgroups (8 per CU), at least 2 rows per workgroup
    int splits = max(1, 2048 / (tiles * B));
    int rows_per_wg = max(2, ss::ceil_div(total_rows, splits));
    const dim3 grid(ss::ceil_div(total_rows, rows_per_wg), tiles, B);
    if (stride == 1)
        hipLaunchKernelGGL(conv3d_wgrad_k3<1>, grid, dim3(192), 0, st, grad_out, in, grad_w, Cin, Cout, D, H, W, Do, Ho, Wo, rows_per_wg,
                           total_rows, ci_tiles);
    else
        hipLaunchKernelGGL(conv3d_wgrad_k3<2>, grid, dim3(192), 0, st, grad_out, in, grad_w, Cin, Cout, D, H, W, Do, Ho, Wo, rows_per_wg,
                           total_rows, ci_tiles);
    return ss::check_launch();
}
