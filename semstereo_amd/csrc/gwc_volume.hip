// Group-wise correlation cost volume over a signed disparity range (gfx950).
//
//   out[b,g,d,y,x] = (1/Cg) * sum_c  R[b,g,c,y,x] * T[b,g,c,y,x-(d-m)]      d in [0,2m)
//
// R/T are the left/right feature maps, optionally L2-normalised over each group's Cg
// channels (x / (||x||_2 + 1e-5)); the value is 0 where x-(d-m) leaves the image.
// Replaces build_gwc_volume / build_gwc_volume_norm / groupwise_correlation[_norm]
// (reference models/submodule.py:190-238), which run a Python loop over d of ~10 ATen
// kernels each and re-normalise the same pixels 2m times.
//
// HBM-bound: algorithmic traffic = 4*(2*C + G*2m)*H*W bytes per pair, 2.9 flop/byte.
// One workgroup owns (b, g, 4 rows, 128 columns): the right-image tile (+ m columns of
// halo each side, zero outside the image) is normalised ONCE and parked in LDS, the
// left-image pixels stay in registers, and every thread then produces all 2m disparities
// of its 4 consecutive columns, 8 disparities at a time from three ds_read_b128 per channel.
// All global traffic is 16 B per lane, coalesced along W.
#include <algorithm>
#include <stdlib.h>
#include <type_traits>

#include "common.h"

namespace {

constexpr int XT = 128;   // columns per workgroup tile (32 lanes x float4)
constexpr float kEps = 1e-05f;

constexpr int RT = 4;     // rows per workgroup tile (measured against 8: never slower, +5 % at B=8)

// x / (||x||_2 + 1e-5) over the CG channels of 4 pixels (models/submodule.py:200-205).  The CG quotients of a pixel share
// their divisor: ONE IEEE reciprocal r = RN(1/s) per pixel, then per channel q = v * r (faithful), e = v - s * q (exact, one
// fma), q' = RN(q + e * r) -- Markstein's correction step, which returns the correctly rounded v / s (what the plain
// division computes with ~10 instructions per quotient instead of 3) unless the significand of s is all ones or
// |v| < 2^-100 s (then it may differ in the last bit).  Pixels whose divisor is not finite take the plain division.
template <int CG>
__device__ __forceinline__ void l2_normalise4(float4 (&v)[CG]) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int c = 0; c < CG; ++c) {
        s.x = ss::add_rn(s.x, ss::mul_rn(v[c].x, v[c].x));
        s.y = ss::add_rn(s.y, ss::mul_rn(v[c].y, v[c].y));
        s.z = ss::add_rn(s.z, ss::mul_rn(v[c].z, v[c].z));
        s.w = ss::add_rn(s.w, ss::mul_rn(v[c].w, v[c].w));
    }
    s.x = sqrtf(s.x) + kEps; s.y = sqrtf(s.y) + kEps;
    s.z = sqrtf(s.z) + kEps; s.w = sqrtf(s.w) + kEps;
    const float big = 3.0e38f;
    if (s.x < big && s.y < big && s.z < big && s.w < big) {              // (NaN compares false)
        const float4 r = make_float4(1.0f / s.x, 1.0f / s.y, 1.0f / s.z, 1.0f / s.w);
        auto quot = [](float a, float d, float rd) {
            const float q = ss::mul_rn(a, rd);
            return __fmaf_rn(__fmaf_rn(-d, q, a), rd, q);
        };
#pragma unroll
        for (int c = 0; c < CG; ++c) {
            v[c].x = quot(v[c].x, s.x, r.x); v[c].y = quot(v[c].y, s.y, r.y);
            v[c].z = quot(v[c].z, s.z, r.z); v[c].w = quot(v[c].w, s.w, r.w);
        }
    } else {
#pragma unroll
        for (int c = 0; c < CG; ++c) {
            v[c].x = v[c].x / s.x; v[c].y = v[c].y / s.y;
            v[c].z = v[c].z / s.z; v[c].w = v[c].w / s.w;
        }
    }
}

// STREAM: write the volume with nontemporal stores.  Kernel alone: -10 % time at every size; but a
// volume that fits the 256 MB infinity cache is re-read from there by the next kernel of the path,
// and streaming it past the cache costs the consumer more than it saves here (measured on the live
// shape, B=1: 24.5 -> 21.9 us for this kernel, +45 us for the step) -- so only large volumes stream.
template <int CG, bool NORM, bool STREAM>  // RT rows x 128 columns per workgroup, 32*RT threads
__global__ __launch_bounds__(32 * RT) void gwc_volume_v4(const float* __restrict__ ref,
                                                      const float* __restrict__ tgt,
                                                      float* __restrict__ out,
                                                      int C, int H, int W, int m, int G, int dmin, int D) {
    // m: halo columns per side of the LDS tile (ss::range_halo); plane p holds disparity dmin + p; off = m - dmin
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [CG][RT][LW]
    const int off = m - dmin;
    const int LW = XT + 2 * m;
    const int tid = threadIdx.x;
    const int xt0 = blockIdx.x * XT;
    const int y0 = blockIdx.y * RT;
    const int b = blockIdx.z / G, g = blockIdx.z % G;
    const size_t plane = (size_t)H * W;
    const float* refg = ref + ((size_t)b * C + (size_t)g * CG) * plane;
    const float* tgtg = tgt + ((size_t)b * C + (size_t)g * CG) * plane;

    // ---- stage the (normalised) right-image tile, zero-extended by m columns per side ----
    // (the 128 central columns: one aligned quad per thread; then the m / 4 halo quads per side and row)
    auto stage_right = [&](int row, int qi) {
        const int col0 = xt0 - m + qi * 4;
        const int y = y0 + row;
        float4 v[CG];
        if (y < H && col0 >= 0 && col0 < W) {
            const float* p = tgtg + (size_t)y * W + col0;
#pragma unroll
            for (int c = 0; c < CG; ++c) v[c] = *reinterpret_cast<const float4*>(p + c * plane);
            if (NORM) l2_normalise4<CG>(v);
        } else {
#pragma unroll
            for (int c = 0; c < CG; ++c) v[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int c = 0; c < CG; ++c)
            *reinterpret_cast<float4*>(&lds[(c * RT + row) * LW + qi * 4]) = v[c];
    };
    const int HQ = m / 4;
    stage_right(tid >> 5, HQ + (tid & 31));
    for (int q = tid; q < RT * 2 * HQ; q += 32 * RT) {
        const int row = q / (2 * HQ), k = q - row * 2 * HQ;
        stage_right(row, k < HQ ? k : k + XT / 4);
    }

    // ---- this thread's 4 left-image pixels, all CG channels, in registers ----
    const int tx = tid & 31, ty = tid >> 5;
    const int x0 = xt0 + tx * 4;
    const int y = y0 + ty;
    const bool active = (y < H) && (x0 < W);
    float r[CG][4];
    if (active) {
        const float* p = refg + (size_t)y * W + x0;
        float4 v[CG];
#pragma unroll
        for (int c = 0; c < CG; ++c) v[c] = *reinterpret_cast<const float4*>(p + c * plane);
        if (NORM) l2_normalise4<CG>(v);
#pragma unroll
        for (int c = 0; c < CG; ++c) { r[c][0] = v[c].x; r[c][1] = v[c].y; r[c][2] = v[c].z; r[c][3] = v[c].w; }
    } else {
#pragma unroll
        for (int c = 0; c < CG; ++c) { r[c][0] = r[c][1] = r[c][2] = r[c][3] = 0.f; }
    }
    __syncthreads();
    if (!active) return;

    // ---- all D disparities of these 4 columns, 8 at a time ----
    // LDS column of image column xx is xx - xt0 + m, so output (plane d, x0+j), disparity dmin + d, reads index
    // tx*4 + j + m - dmin - d = base + (j + 8 - i) with d = d0 + i, base = tx*4 + off - d0 - 8.
    float* outp = out + (((size_t)(b * G + g) * D) * H + y) * W + x0;
    const float den = (float)CG;
    for (int d0 = 0; d0 < D; d0 += 8) {
        float acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i) { acc[i][0] = acc[i][1] = acc[i][2] = acc[i][3] = 0.f; }
        const int base = tx * 4 + off - d0 - 8;
#pragma unroll
        for (int c = 0; c < CG; ++c) {
            const float* lp = &lds[(c * RT + ty) * LW + base];
            float w[12];
            ss::lds_read16(lp, &w[0]);
            ss::lds_read16(lp + 4, &w[4]);
            ss::lds_read16(lp + 8, &w[8]);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = fmaf(r[c][j], w[j + 8 - i], acc[i][j]);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int d = d0 + i;
            const int col = x0 - (d + dmin);       // partner column of x0; of x0+j it is col+j
            float4 o;
            o.x = ((unsigned)(col + 0) < (unsigned)W) ? acc[i][0] / den : 0.f;
            o.y = ((unsigned)(col + 1) < (unsigned)W) ? acc[i][1] / den : 0.f;
            o.z = ((unsigned)(col + 2) < (unsigned)W) ? acc[i][2] / den : 0.f;
            o.w = ((unsigned)(col + 3) < (unsigned)W) ? acc[i][3] / den : 0.f;
            if (STREAM) {
                typedef float v4f __attribute__((ext_vector_type(4)));
                v4f ov = {o.x, o.y, o.z, o.w};
                __builtin_nontemporal_store(ov, reinterpret_cast<v4f*>(outp + (size_t)d * plane));
            } else {
                *reinterpret_cast<float4*>(outp + (size_t)d * plane) = o;
            }
        }
    }
}

// Any W / maxdisp / Cg: one column per lane, both normalised tiles in LDS.
// blockDim = (GX, GR); LDS = Cg * GR * (GX + GX + 2m) floats.
template <bool NORM>
__global__ void gwc_volume_generic(const float* __restrict__ ref, const float* __restrict__ tgt,
                                   float* __restrict__ out, int C, int H, int W, int m, int G, int dmin, int D) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int GX = blockDim.x, GR = blockDim.y;
    const int Cg = C / G;
    const int LW = GX + 2 * m;
    float* rn = lds;                       // [Cg][GR][GX]
    float* tn = lds + (size_t)Cg * GR * GX;  // [Cg][GR][LW]
    const int xt0 = blockIdx.x * GX, y0 = blockIdx.y * GR;
    const int b = blockIdx.z / G, g = blockIdx.z % G;
    const size_t plane = (size_t)H * W;
    const float* refg = ref + ((size_t)b * C + (size_t)g * Cg) * plane;
    const float* tgtg = tgt + ((size_t)b * C + (size_t)g * Cg) * plane;
    const int tid = threadIdx.y * GX + threadIdx.x, nthr = GX * GR;

    for (int q = tid; q < GR * LW; q += nthr) {
        const int row = q / LW, li = q - row * LW;
        const int col = xt0 - m + li, y = y0 + row;
        const bool in = (y < H) && (col >= 0) && (col < W);
        float den = 1.f;
        if (in && NORM) {
            float s = 0.f;
            for (int c = 0; c < Cg; ++c) { float v = tgtg[c * plane + (size_t)y * W + col]; s = ss::add_rn(s, ss::mul_rn(v, v)); }
            den = sqrtf(s) + kEps;
        }
        for (int c = 0; c < Cg; ++c) {
            float v = in ? tgtg[c * plane + (size_t)y * W + col] : 0.f;
            tn[((size_t)c * GR + row) * LW + li] = NORM ? v / den : v;
        }
    }
    for (int q = tid; q < GR * GX; q += nthr) {
        const int row = q / GX, li = q - row * GX;
        const int col = xt0 + li, y = y0 + row;
        const bool in = (y < H) && (col < W);
        float den = 1.f;
        if (in && NORM) {
            float s = 0.f;
            for (int c = 0; c < Cg; ++c) { float v = refg[c * plane + (size_t)y * W + col]; s = ss::add_rn(s, ss::mul_rn(v, v)); }
            den = sqrtf(s) + kEps;
        }
        for (int c = 0; c < Cg; ++c) {
            float v = in ? refg[c * plane + (size_t)y * W + col] : 0.f;
            rn[((size_t)c * GR + row) * GX + li] = NORM ? v / den : v;
        }
    }
    __syncthreads();
    const int x = xt0 + threadIdx.x, y = y0 + threadIdx.y;
    if (x >= W || y >= H) return;
    const float den = (float)Cg;
    float* outp = out + (((size_t)(b * G + g) * D) * H + y) * W + x;
    for (int d = 0; d < D; ++d) {
        const int li = threadIdx.x + m - dmin - d;       // = (x - (dmin + d)) - (xt0 - m)
        float acc = 0.f;
        for (int c = 0; c < Cg; ++c)
            acc = fmaf(rn[((size_t)c * GR + threadIdx.y) * GX + threadIdx.x], tn[((size_t)c * GR + threadIdx.y) * LW + li], acc);   // the v4 kernels' chain
        const int col = x - (d + dmin);
        outp[(size_t)d * plane] = ((unsigned)col < (unsigned)W) ? acc / den : 0.f;
    }
}

// groupwise_correlation: no shift, one output per (b,g,pixel).
template <bool NORM>
__global__ void group_corr_kernel(const float* __restrict__ f1, const float* __restrict__ f2,
                                  float* __restrict__ out, int C, int G, long long plane, long long total) {
    const int Cg = C / G;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long pix = i % plane;
        const long long bg = i / plane;          // b*G + g
        const long long b = bg / G, g = bg % G;
        const float* p1 = f1 + (b * C + g * Cg) * plane + pix;
        const float* p2 = f2 + (b * C + g * Cg) * plane + pix;
        float d1 = 1.f, d2 = 1.f;
        if (NORM) {
            float s1 = 0.f, s2 = 0.f;
            for (int c = 0; c < Cg; ++c) {
                float a = p1[c * plane], bb = p2[c * plane];
                s1 = ss::add_rn(s1, ss::mul_rn(a, a));
                s2 = ss::add_rn(s2, ss::mul_rn(bb, bb));
            }
            d1 = sqrtf(s1) + kEps; d2 = sqrtf(s2) + kEps;
        }
        float acc = 0.f;
        for (int c = 0; c < Cg; ++c) {
            float a = p1[c * plane], bb = p2[c * plane];
            if (NORM) { a = a / d1; bb = bb / d2; }
            acc = fmaf(a, bb, acc);          // the same fused chain as the volume kernels: the zero-disparity plane of a volume IS this op
        }
        out[i] = acc / (float)Cg;
    }
}

// Backward of the un-normalised volume: one thread per input element, loop over d.
__global__ void gwc_volume_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ ref,
                                      const float* __restrict__ tgt, float* __restrict__ gref,
                                      float* __restrict__ gtgt, int C, int H, int W, int dmin, int D, int G,
                                      long long total) {
    const int Cg = C / G;
    const long long plane = (long long)H * W;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        const long long row = i / W;                 // (b*C + c)*H + y
        const int yy = (int)(row % H);
        const long long bc = row / H;
        const int c = (int)(bc % C);
        const long long b = bc / C;
        const int g = c / Cg;
        const float* go = gout + ((b * G + g) * D) * plane + (long long)yy * W;
        const float* rrow = ref + row * W;
        const float* trow = tgt + row * W;
        float ar = 0.f, at = 0.f;
        for (int d = 0; d < D; ++d) {
            const int s = d + dmin;
            const int xr = x - s;                    // partner of left column x
            if ((unsigned)xr < (unsigned)W) ar += go[d * plane + x] * trow[xr];
            const int xl = x + s;                    // left column whose partner is right column x
            if ((unsigned)xl < (unsigned)W) at += go[d * plane + xl] * rrow[xl];
        }
        gref[i] = ar / (float)Cg;
        gtgt[i] = at / (float)Cg;
    }
}

// ... row by row (r06; W <= GB_WMAX, D <= GB_DMAX, Cg <= 8: every live shape): a workgroup owns one image row of one (batch element,
// group) and stages that row of the gradient's D planes and of the group's Cg channels of both maps in LDS once -- the thread-per-element
// form above fetches every gradient value Cg times and every feature value 2 D times through the caches (442 us at batch 4 of the
// 1024^2 training shape for 200 MB of tensors).  Same sums in the same order as above (d ascending), so the results are identical.
constexpr int GB_WMAX = 512, GB_DMAX = 48;
__global__ __launch_bounds__(256) void gwc_volume_bwd_rows_kernel(const float* __restrict__ gout, const float* __restrict__ ref,
                                                                   const float* __restrict__ tgt, float* __restrict__ gref,
                                                                   float* __restrict__ gtgt, int C, int H, int W, int dmin, int D, int G) {
    extern __shared__ float gb_lds[];                // [D][W] gradient | [Cg][W] ref | [Cg][W] tgt
    const int Cg = C / G;
    float* gs = gb_lds;
    float* rs = gs + (size_t)D * W;
    float* ts = rs + (size_t)Cg * W;
    const int yy = blockIdx.x, g = blockIdx.y;
    const long long b = blockIdx.z, plane = (long long)H * W;
    const float* go = gout + ((b * G + g) * D) * plane + (long long)yy * W;
    const long long fbase = ((b * C + (long long)g * Cg) * H + yy) * W;       // channel c of the group: + c * plane
    for (int i = threadIdx.x; i < D * W; i += 256) gs[i] = go[(long long)(i / W) * plane + (i % W)];
    for (int i = threadIdx.x; i < Cg * W; i += 256) {
        const long long o = fbase + (long long)(i / W) * plane + (i % W);
        rs[i] = ref[o];
        ts[i] = tgt[o];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < Cg * W; i += 256) {
        const int c = i / W, x = i % W;
        float ar = 0.f, at = 0.f;
        for (int d = 0; d < D; ++d) {
            const int s = d + dmin;
            const int xr = x - s;
            if ((unsigned)xr < (unsigned)W) ar += gs[d * W + x] * ts[c * W + xr];
            const int xl = x + s;
            if ((unsigned)xl < (unsigned)W) at += gs[d * W + xl] * rs[c * W + xl];
        }
        const long long o = fbase + (long long)c * plane + x;
        gref[o] = ar / (float)Cg;
        gtgt[o] = at / (float)Cg;
    }
}


// ---- models/SemStereo.py:273-276 in one kernel: build_gwc_volume_norm -> `patch` (depthwise (1,3,3) Conv3d) ->
// channelAtt gate -- the volume never reaches HBM un-stenciled (67 MB write + 67 MB read per 1024^2 pair saved).
//   out[b,g,d,y,x] = sigmoid(gate[b,g,y,x]) * sum_{ky,kx} wp[g,ky,kx] * V[b,g,d,y+ky-1,x+kx-1]       (V = 0 outside the image)
// A workgroup owns (b, g, FRO = 6 output rows, 128 columns) and computes V on FRC = 8 rows (one halo row each side) exactly
// as gwc_volume_v4 does (same tile of the normalised right image in LDS, same fmaf order), 8 disparities at a time,
// into an LDS tile [8 d][8 rows][136]; the two seam columns (x = xt0 - 1, xt0 + 128) are computed by 128 threads from two
// parked columns of the left image.  The 3x3 stencil and the gate are then applied from LDS in depthwise_patch_v4's own
// order (rows outside the image skipped, columns outside contribute fmaf(w, 0)), so the result is bit-identical to the
// two-kernel form.
constexpr int FRO = 6, FRC = FRO + 2, FVP = XT + 8;


template <int CG, bool NORM, bool STREAM>
__global__ __launch_bounds__(32 * FRC) void gwc_patch_gate_v4(const float* __restrict__ ref, const float* __restrict__ tgt,
                                                             const float* __restrict__ wpatch, const float* __restrict__ gate,
                                                             float* __restrict__ out, int C, int H, int W, int m, int G,
                                                             int dmin, int D) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int off = m - dmin;                                   // (m: halo per side, plane p = disparity dmin + p; see gwc_volume_v4)
    const int LW = XT + 2 * m + 8;                              // right-image tile: image columns xt0 - m - 4 .. xt0 + 128 + m + 3
    float* tn = lds;                                            // [CG][FRC][LW]
    float* vt = lds + CG * FRC * LW;                            // [8][FRC][FVP]: index 4 + j <-> image column xt0 + j
    float* rh = vt + 8 * FRC * FVP;                             // [2][FRC][CG]: the left image at the two seam columns
    float* sgt = rh + 2 * FRC * CG;                             // [FRO][XT]: sigmoid of the gate logits of the output pixels
    const int tid = threadIdx.x;
    const int xt0 = blockIdx.x * XT;
    const int y0 = blockIdx.y * FRO - 1;                        // first COMPUTED row (the halo row above the output rows)
    const int b = blockIdx.z / G, g = blockIdx.z % G;
    const size_t plane = (size_t)H * W;
    const float* refg = ref + ((size_t)b * C + (size_t)g * CG) * plane;
    const float* tgtg = tgt + ((size_t)b * C + (size_t)g * CG) * plane;

    // The tile's 128 central columns are one aligned quad per thread (as the left image below); the (m + 4) / 4 halo quads per
    // side and row follow in a second pass of (m + 4) / 2 * FRC threads.  (As one loop over all FRC * LW / 4 quads the first two
    // waves normalised two quads each -- the second mostly out-of-image zeros -- while the others waited at the barrier.)
    auto stage_right = [&](int row, int qi) {
        const int col0 = xt0 - m - 4 + qi * 4;
        const int y = y0 + row;
        float4 v[CG];
        if ((unsigned)y < (unsigned)H && col0 >= 0 && col0 < W) {
            const float* p = tgtg + (size_t)y * W + col0;
#pragma unroll
            for (int c = 0; c < CG; ++c) v[c] = *reinterpret_cast<const float4*>(p + c * plane);
            if (NORM) l2_normalise4<CG>(v);
        } else {
#pragma unroll
            for (int c = 0; c < CG; ++c) v[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int c = 0; c < CG; ++c) *reinterpret_cast<float4*>(&tn[(c * FRC + row) * LW + qi * 4]) = v[c];
    };
    const int HQ = (m + 4) / 4;                                 // halo quads per side
    stage_right(tid >> 5, HQ + (tid & 31));
    for (int q = tid; q < FRC * 2 * HQ; q += 32 * FRC) {
        const int row = q / (2 * HQ), k = q - row * 2 * HQ;
        stage_right(row, k < HQ ? k : k + XT / 4);
    }
    if (tid < 2 * FRC) {                                        // the left image at x = xt0 - 1 and x = xt0 + 128
        // (normalised as the aligned quad that holds the column, through the same code as every other pixel: a scalar
        // restatement of the normalisation compiled to a result one ulp away)
        const int side = tid / FRC, row = tid % FRC;
        const int xq = side ? xt0 + XT : xt0 - 4, y = y0 + row;
        float4 v[CG];
        if ((unsigned)y < (unsigned)H && xq >= 0 && xq < W) {
            const float* p = refg + (size_t)y * W + xq;
#pragma unroll
            for (int c = 0; c < CG; ++c) v[c] = *reinterpret_cast<const float4*>(p + c * plane);
            if (NORM) l2_normalise4<CG>(v);
        } else {
#pragma unroll
            for (int c = 0; c < CG; ++c) v[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int c = 0; c < CG; ++c) rh[(side * FRC + row) * CG + c] = side ? v[c].x : v[c].w;
    }
    const int tx = tid & 31, ty = tid >> 5;
    const int x0 = xt0 + tx * 4;
    const int y = y0 + ty;
    const bool active = ((unsigned)y < (unsigned)H) && (x0 < W);
    float r[CG][4];
    {
        float4 v[CG];
        if (active) {
            const float* p = refg + (size_t)y * W + x0;
#pragma unroll
            for (int c = 0; c < CG; ++c) v[c] = *reinterpret_cast<const float4*>(p + c * plane);
            if (NORM) l2_normalise4<CG>(v);
        } else {
#pragma unroll
            for (int c = 0; c < CG; ++c) v[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int c = 0; c < CG; ++c) { r[c][0] = v[c].x; r[c][1] = v[c].y; r[c][2] = v[c].z; r[c][3] = v[c].w; }
    }
    float wv[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wv[k] = wpatch[g * 9 + k];
    __syncthreads();

    const float den = (float)CG;
    float* outg = out + ((size_t)(b * G + g) * D) * plane;
    const float* gateg = gate ? gate + (size_t)(b * G + g) * plane : nullptr;
    // the gate of the tile's 6 x 128 output pixels: ONE sigmoid per pixel (not one per output), parked in LDS
    if (gateg && ty < FRO) {
        const int yo = y0 + 1 + ty;
        float4 sg = make_float4(0.f, 0.f, 0.f, 0.f);
        if (yo < H && x0 < W) {
            const float4 gl = *reinterpret_cast<const float4*>(gateg + (size_t)yo * W + x0);
            sg = make_float4(1.0f / (1.0f + expf(-gl.x)), 1.0f / (1.0f + expf(-gl.y)), 1.0f / (1.0f + expf(-gl.z)),
                             1.0f / (1.0f + expf(-gl.w)));
        }
        *reinterpret_cast<float4*>(&sgt[ty * XT + tx * 4]) = sg;
    }
    for (int d0 = 0; d0 < D; d0 += 8) {
        // ---- V for this thread's row and 4 columns, 8 disparities (gwc_volume_v4's arithmetic) ----
        float acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i) { acc[i][0] = acc[i][1] = acc[i][2] = acc[i][3] = 0.f; }
        const int base = tx * 4 + off - d0 - 8 + 4;
#pragma unroll
        for (int c = 0; c < CG; ++c) {
            const float* lp = &tn[(c * FRC + ty) * LW + base];
            float w[12];
            ss::lds_read16(lp, &w[0]);
            ss::lds_read16(lp + 4, &w[4]);
            ss::lds_read16(lp + 8, &w[8]);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(r[c][j], w[j + 8 - i], acc[i][j]);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int col = x0 - (d0 + i + dmin);
            float4 o;
            o.x = (active && (unsigned)(col + 0) < (unsigned)W) ? acc[i][0] / den : 0.f;
            o.y = (active && (unsigned)(col + 1) < (unsigned)W) ? acc[i][1] / den : 0.f;
            o.z = (active && (unsigned)(col + 2) < (unsigned)W) ? acc[i][2] / den : 0.f;
            o.w = (active && (unsigned)(col + 3) < (unsigned)W) ? acc[i][3] / den : 0.f;
            *reinterpret_cast<float4*>(&vt[(i * FRC + ty) * FVP + 4 + tx * 4]) = o;
        }
        if (tid < 2 * FRC * 8) {                                // the seam columns: (side, row, disparity) per thread
            const int i = tid & 7, row = (tid >> 3) % FRC, side = tid / (8 * FRC);
            const int xx = side ? xt0 + XT : xt0 - 1;
            const int col = xx - (d0 + i + dmin);
            float a = 0.f;
#pragma unroll
            for (int c = 0; c < CG; ++c) a = fmaf(rh[(side * FRC + row) * CG + c], tn[(c * FRC + row) * LW + (col - xt0 + m + 4)], a);
            const bool ok = (unsigned)xx < (unsigned)W && (unsigned)(y0 + row) < (unsigned)H && (unsigned)col < (unsigned)W;
            vt[(i * FRC + row) * FVP + (side ? 4 + XT : 3)] = ok ? a / den : 0.f;
        }
        __syncthreads();
        // ---- depthwise 3x3 + gate from LDS: thread (ty, tx) owns disparity d0 + ty, 4 columns, ALL 6 output rows: the 8
        // computed rows slide through registers (one 16-byte + two 4-byte LDS reads per row instead of nine 16-byte reads
        // per output quad: this phase was bound by LDS bandwidth) ----
        // Two copies: tiles whose 8 computed rows and 128 columns all lie inside the image (all but the first and last row of
        // tiles) run without a single bounds test; the border tiles take the checked form.
        auto stencil = [&](auto checked) {
            constexpr bool CHECK = decltype(checked)::value;
            // a row of the window as aligned register PAIRS, in both phases: wa[k] = (x[2k], x[2k+1]), ws[k] = (x[2k+1], x[2k+2]) with
            // x[0..5] = columns x0 - 1 .. x0 + 4
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 wa[3][3], ws[3][2];                              // [slot][pair]
#pragma unroll
            for (int rr = 0; rr < FRC; ++rr) {
                const float* vrow = &vt[(ty * FRC + rr) * FVP];
                float mq[4];
                ss::lds_read16(vrow + 4 + tx * 4, mq);
                // the two neighbouring columns come from the neighbouring LANES (a +-1 wave shift of M.w / M.x); the first and
                // last lane of a row take the seam columns, which every lane of the row reads from the same LDS address (a
                // broadcast, no branch).  (As two 4-byte LDS reads per lane at a 16-byte stride they were 8-way bank
                // conflicts: 43 % of the kernel's LDS time, tools/pmc_sq.sh.)
                const float sl_ = vrow[3], sr_ = vrow[4 + XT];
                float l = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, mq[3]), 0x138, 0xf, 0xf, false));   // lane n <- n - 1
                float rgt = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, mq[0]), 0x130, 0xf, 0xf, false)); // lane n <- n + 1
                l = (tx == 0) ? sl_ : l;
                rgt = (tx == 31) ? sr_ : rgt;
                const int slot = rr % 3;
                wa[slot][0] = f2{l, mq[0]}; wa[slot][1] = f2{mq[1], mq[2]}; wa[slot][2] = f2{mq[3], rgt};
                ws[slot][0] = f2{mq[0], mq[1]}; ws[slot][1] = f2{mq[2], mq[3]};
                if (rr < 2) continue;
                const int orow = rr - 2, yo = y0 + 1 + orow;      // output row whose window is complete
                if (CHECK && (yo >= H || x0 >= W)) continue;
                f2 o01 = {0.f, 0.f}, o23 = {0.f, 0.f};
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    if (CHECK && (unsigned)(yo + ky - 1) >= (unsigned)H) continue;        // (as depthwise_patch_v4: the row is skipped)
                    const int sl = (orow + ky) % 3;
                    const f2 w0 = {wv[ky * 3], wv[ky * 3]}, w1 = {wv[ky * 3 + 1], wv[ky * 3 + 1]}, w2 = {wv[ky * 3 + 2], wv[ky * 3 + 2]};
                    o01 = __builtin_elementwise_fma(w0, wa[sl][0], o01); o23 = __builtin_elementwise_fma(w0, wa[sl][1], o23);
                    o01 = __builtin_elementwise_fma(w1, ws[sl][0], o01); o23 = __builtin_elementwise_fma(w1, ws[sl][1], o23);
                    o01 = __builtin_elementwise_fma(w2, wa[sl][1], o01); o23 = __builtin_elementwise_fma(w2, wa[sl][2], o23);
                }
                float o[4] = {o01.x, o01.y, o23.x, o23.y};
                if (gateg) {
                    float sg[4];
                    ss::lds_read16(&sgt[orow * XT + tx * 4], sg);
                    o[0] = ss::mul_rn(sg[0], o[0]); o[1] = ss::mul_rn(sg[1], o[1]);
                    o[2] = ss::mul_rn(sg[2], o[2]); o[3] = ss::mul_rn(sg[3], o[3]);
                }
                float* op = outg + (size_t)(d0 + ty) * plane + (size_t)yo * W + x0;
                if (STREAM) {
                    typedef float v4f __attribute__((ext_vector_type(4)));
                    v4f ov = {o[0], o[1], o[2], o[3]};
                    __builtin_nontemporal_store(ov, reinterpret_cast<v4f*>(op));
                } else {
                    *reinterpret_cast<float4*>(op) = make_float4(o[0], o[1], o[2], o[3]);
                }
            }
        };
        if (y0 >= 0 && y0 + FRC <= H && xt0 + XT <= W) stencil(std::false_type{});
        else stencil(std::true_type{});
        __syncthreads();
    }
}

template <int CG, bool NORM, bool STREAM>
int launch_gpg(const float* ref, const float* tgt, const float* wpatch, const float* gate, float* out, int B, int C, int H,
               int W, int m, int G, int dmin, int D, hipStream_t st) {
    dim3 grid(ss::ceil_div(W, XT), ss::ceil_div(H, FRO), B * G);
    const size_t lds = ((size_t)CG * FRC * (XT + 2 * m + 8) + 8 * FRC * FVP + 2 * FRC * CG + FRO * XT) * sizeof(float);
    auto kern = gwc_patch_gate_v4<CG, NORM, STREAM>;
    if (ss::ensure_dynamic_lds(reinterpret_cast<const void*>(kern), (int)lds) != SS_OK) return SS_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, grid, dim3(32 * FRC), lds, st, ref, tgt, wpatch, gate, out, C, H, W, m, G, dmin, D);
    return ss::check_launch();
}

template <int CG, bool NORM, bool STREAM>
int launch_v4_as(const float* ref, const float* tgt, float* out, int B, int C, int H, int W, int m, int G, int dmin, int D,
                 hipStream_t st) {
    dim3 grid(ss::ceil_div(W, XT), ss::ceil_div(H, RT), B * G);
    size_t lds = (size_t)CG * RT * (XT + 2 * m) * sizeof(float);
    hipLaunchKernelGGL((gwc_volume_v4<CG, NORM, STREAM>), grid, dim3(32 * RT), lds, st, ref, tgt, out, C, H, W, m, G, dmin, D);
    return ss::check_launch();
}

template <int CG>
int launch_v4(const float* ref, const float* tgt, float* out, int B, int C, int H, int W, int m, int G, int dmin, int D,
              int normalize, bool stream_out, hipStream_t st) {
    if (normalize)
        return stream_out ? launch_v4_as<CG, true, true>(ref, tgt, out, B, C, H, W, m, G, dmin, D, st)
                          : launch_v4_as<CG, true, false>(ref, tgt, out, B, C, H, W, m, G, dmin, D, st);
    return stream_out ? launch_v4_as<CG, false, true>(ref, tgt, out, B, C, H, W, m, G, dmin, D, st)
                      : launch_v4_as<CG, false, false>(ref, tgt, out, B, C, H, W, m, G, dmin, D, st);
}

}  // namespace

extern "C" int ss_gwc_volume_fwd(const float* ref, const float* tgt, float* out, int B, int C, int H, int W,
                                 int dmin, int ndisp, int groups, int normalize, ss_stream_t stream) {
    SS_REQUIRE(ref && tgt && out);
    SS_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && ndisp > 0 && groups > 0);
    SS_REQUIRE(C % groups == 0);
    SS_REQUIRE((long long)B * groups <= 65535);
    hipStream_t st = ss::as_stream(stream);
    const int Cg = C / groups, m = ss::range_halo(dmin, ndisp);
    const bool aligned = ((reinterpret_cast<uintptr_t>(ref) | reinterpret_cast<uintptr_t>(tgt) |
                           reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    if (aligned && W % 4 == 0 && dmin % 4 == 0 && ndisp % 8 == 0 && (size_t)Cg * RT * (XT + 2 * m) * 4 <= 64 * 1024) {
        // volumes beyond the 256 MB infinity cache cannot stay resident for the consumer: stream them
        const size_t out_bytes = (size_t)B * groups * ndisp * H * W * sizeof(float);
        bool stream_out = out_bytes > ((size_t)192 << 20);
        if (ss::tuning().gwc_stream >= 0) stream_out = ss::tuning().gwc_stream == 1;   // tuning aid
        if (Cg == 8) return launch_v4<8>(ref, tgt, out, B, C, H, W, m, groups, dmin, ndisp, normalize, stream_out, st);
        if (Cg == 4) return launch_v4<4>(ref, tgt, out, B, C, H, W, m, groups, dmin, ndisp, normalize, stream_out, st);
    }
    // generic path: shrink the tile until both normalised tiles fit in 64 KiB of LDS
    int gx = 64, gr = 4;
    const int mg = ss::range_halo(dmin, ndisp);
    auto bytes = [&](int x, int r) { return (size_t)Cg * r * (2 * x + 2 * mg) * sizeof(float); };
    while (gr > 1 && bytes(gx, gr) > 64 * 1024) gr >>= 1;
    while (gx > 16 && bytes(gx, gr) > 64 * 1024) gx >>= 1;
    if (bytes(gx, gr) > 64 * 1024) return SS_ERR_UNSUPPORTED;
    dim3 grid(ss::ceil_div(W, gx), ss::ceil_div(H, gr), B * groups), block(gx, gr);
    if (normalize)
        hipLaunchKernelGGL(gwc_volume_generic<true>, grid, block, bytes(gx, gr), st, ref, tgt, out, C, H, W, mg, groups, dmin, ndisp);
    else
        hipLaunchKernelGGL(gwc_volume_generic<false>, grid, block, bytes(gx, gr), st, ref, tgt, out, C, H, W, mg, groups, dmin, ndisp);
    return ss::check_launch();
}

extern "C" int ss_gwc_patch_gate_fwd(const float* ref, const float* tgt, const float* patch_w, const float* gate_logits,
                                     float* out, int B, int C, int H, int W, int dmin, int ndisp, int groups, int normalize,
                                     ss_stream_t stream) {
    SS_REQUIRE(ref && tgt && patch_w && out);
    SS_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && ndisp > 0 && groups > 0);
    SS_REQUIRE(C % groups == 0);
    SS_REQUIRE((long long)B * groups <= 65535);
    const int Cg = C / groups, m = ss::range_halo(dmin, ndisp);
    const uintptr_t bits = reinterpret_cast<uintptr_t>(ref) | reinterpret_cast<uintptr_t>(tgt) | reinterpret_cast<uintptr_t>(out) |
                           reinterpret_cast<uintptr_t>(gate_logits);
    const size_t lds = ((size_t)Cg * FRC * (XT + 2 * m + 8) + 8 * FRC * FVP + 2 * FRC * Cg + FRO * XT) * sizeof(float);
    if ((bits & 15) != 0 || W % 4 != 0 || dmin % 4 != 0 || ndisp % 8 != 0 || (Cg != 8 && Cg != 4) || lds > 150 * 1024)
        return SS_ERR_UNSUPPORTED;
    hipStream_t st = ss::as_stream(stream);
    const size_t out_bytes = (size_t)B * groups * ndisp * H * W * sizeof(float);
    bool stream_out = out_bytes > ((size_t)192 << 20);           // as ss_gwc_volume_fwd: beyond the infinity cache
    if (ss::tuning().gwc_stream >= 0) stream_out = ss::tuning().gwc_stream == 1;
#define SS_GPG(CGV)                                                                                                           \
    if (Cg == CGV) {                                                                                                          \
        if (normalize)                                                                                                        \
            return stream_out ? launch_gpg<CGV, true, true>(ref, tgt, patch_w, gate_logits, out, B, C, H, W, m, groups, dmin, ndisp, st)   \
                              : launch_gpg<CGV, true, false>(ref, tgt, patch_w, gate_logits, out, B, C, H, W, m, groups, dmin, ndisp, st); \
        return stream_out ? launch_gpg<CGV, false, true>(ref, tgt, patch_w, gate_logits, out, B, C, H, W, m, groups, dmin, ndisp, st)      \
                          : launch_gpg<CGV, false, false>(ref, tgt, patch_w, gate_logits, out, B, C, H, W, m, groups, dmin, ndisp, st);    \
    }
    SS_GPG(8)
    SS_GPG(4)
#undef SS_GPG
    return SS_ERR_UNSUPPORTED;
}

extern "C" int ss_groupwise_correlation_fwd(const float* fea1, const float* fea2, float* out, int B, int C, int H,
                                            int W, int groups, int normalize, ss_stream_t stream) {
    SS_REQUIRE(fea1 && fea2 && out);
    SS_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && groups > 0);
    SS_REQUIRE(C % groups == 0);
    const long long plane = (long long)H * W, total = (long long)B * groups * plane;
    const int blocks = (int)std::min<long long>(ss::ceil_div_ll(total, 256), 256 * 16);
    if (normalize)
        hipLaunchKernelGGL(group_corr_kernel<true>, dim3(blocks), dim3(256), 0, ss::as_stream(stream), fea1, fea2, out, C, groups, plane, total);
    else
        hipLaunchKernelGGL(group_corr_kernel<false>, dim3(blocks), dim3(256), 0, ss::as_stream(stream), fea1, fea2, out, C, groups, plane, total);
    return ss::check_launch();
}

extern "C" int ss_gwc_volume_bwd(const float* grad_out, const float* ref, const float* tgt, float* grad_ref,
                                 float* grad_tgt, int B, int C, int H, int W, int dmin, int ndisp, int groups,
                                 ss_stream_t stream) {
    SS_REQUIRE(grad_out && ref && tgt && grad_ref && grad_tgt);
    SS_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && ndisp > 0 && groups > 0);
    SS_REQUIRE(C % groups == 0);
    const long long total = (long long)B * C * H * W;
    const int cg = C / groups;
    const size_t lds = ((size_t)ndisp + 2 * (size_t)cg) * W * sizeof(float);
    if (W <= GB_WMAX && ndisp <= GB_DMAX && cg <= 8 && lds <= 64 * 1024 && H <= 65535 && groups <= 65535 && B <= 65535) {
        hipLaunchKernelGGL(gwc_volume_bwd_rows_kernel, dim3(H, groups, B), dim3(256), lds, ss::as_stream(stream), grad_out, ref, tgt, grad_ref,
                           grad_tgt, C, H, W, dmin, ndisp, groups);
        return ss::check_launch();
    }
    const int blocks = (int)std::min<long long>(ss::ceil_div_ll(total, 256), 256 * 32);
    hipLaunchKernelGGL(gwc_volume_bwd_kernel, dim3(blocks), dim3(256), 0, ss::as_stream(stream), grad_out, ref, tgt,
                       grad_ref, grad_tgt, C, H, W, dmin, ndisp, groups, total);
    return ss::check_launch();
}
