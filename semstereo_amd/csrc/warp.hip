// Disparity-candidate warping of the right feature map (gfx950).
//
//   warp(y)[b,c,j,h,w] = bilinear( y[b,c], row ~ h, col ~ w - disp[b,j,h,w] )   zeros padding
//
// Replaces SpatialTransformer_grid (reference models/submodule.py:265-288) and two of its
// consumers in SemStereo.forward: the sparse concat volume `att_topk * cat(left, warp(right))`
// (models/SemStereo.py:241-244, 316-318) and the 5-sample matching-strength probe
// `mean_c(left * warp(right))` (:291-292).  The reference materialises a [B,nd,H,W,2] grid, a
// repeated copy of the left map, the cat and the product: four full passes over the volume; here
// each output element is written exactly once (one column per lane; a float4-per-lane form is kept behind SS_WARP_VEC=4).
//
// Coordinates follow the reference's fp32 round trip exactly: gx = (w - disp)/((W-1)/2) - 1,
// then grid_sample's align_corners=True un-normalisation ix = (gx + 1) * ((W-1)/2) (the ATen
// CPU form), floor, the four weights (1-fx)(1-fy).. and a nw+ne+sw+se sum with separate
// roundings; taps outside the image contribute 0 * weight.  Because of the round trip ix, iy are
// only approximately (w - disp, h): rows h-1..h+1 can carry ~1e-5 weights, which we keep.
#include <algorithm>

#include <stdlib.h>

#include "common.h"

namespace {

struct Taps {
    int o_nw, o_ne, o_sw, o_se;     // element offsets into a [H][W] plane, -1 = outside (value 0)
    float w_nw, w_ne, w_sw, w_se;
};

__device__ __forceinline__ Taps make_taps(float disp, int h, int w, int H, int W, float half_w, float half_h) {
    const float gx = ((float)w - disp) / half_w - 1.0f;
    const float gy = (float)h / half_h - 1.0f;
    const float ix = ss::mul_rn(gx + 1.0f, half_w);
    const float iy = ss::mul_rn(gy + 1.0f, half_h);
    const float xw = floorf(ix), yn = floorf(iy);
    const float fw = ix - xw, fe = 1.0f - fw, fn = iy - yn, fs = 1.0f - fn;
    const float xe = xw + 1.0f, ys = yn + 1.0f;
    const bool mw = (xw > -1.0f) && (xw < (float)W), me = (xe > -1.0f) && (xe < (float)W);
    const bool mn = (yn > -1.0f) && (yn < (float)H), ms = (ys > -1.0f) && (ys < (float)H);
    const int ixw = (int)xw, iyn = (int)yn;
    Taps t;
    t.w_nw = ss::mul_rn(fs, fe); t.w_ne = ss::mul_rn(fs, fw);
    t.w_sw = ss::mul_rn(fn, fe); t.w_se = ss::mul_rn(fn, fw);
    t.o_nw = (mn && mw) ? iyn * W + ixw : -1;
    t.o_ne = (mn && me) ? iyn * W + ixw + 1 : -1;
    t.o_sw = (ms && mw) ? (iyn + 1) * W + ixw : -1;
    t.o_se = (ms && me) ? (iyn + 1) * W + ixw + 1 : -1;
    return t;
}

// the two taps of a row are neighbours in memory: one 8-byte load (4-byte aligned) when both are inside
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
__device__ __forceinline__ void row_pair(const float* __restrict__ plane, int o_w, int o_e, float& vw, float& ve) {
    if (o_w >= 0 && o_e >= 0) {
        const f2u v = *reinterpret_cast<const f2u*>(plane + o_w);
        vw = v.x;
        ve = v.y;
    } else {
        vw = (o_w >= 0) ? plane[o_w] : 0.f;
        ve = (o_e >= 0) ? plane[o_e] : 0.f;
    }
}

__device__ __forceinline__ float sample(const float* __restrict__ plane, const Taps& t) {
    float a, b, c, d;
    row_pair(plane, t.o_nw, t.o_ne, a, b);
    row_pair(plane, t.o_sw, t.o_se, c, d);
    float r = ss::mul_rn(a, t.w_nw);
    r = ss::add_rn(r, ss::mul_rn(b, t.w_ne));
    r = ss::add_rn(r, ss::mul_rn(c, t.w_sw));
    r = ss::add_rn(r, ss::mul_rn(d, t.w_se));
    return r;
}

template <int VEC> struct Vec;
template <> struct Vec<1> { using type = float; };
template <> struct Vec<4> { using type = float4; };

template <int VEC> __device__ __forceinline__ void load_vec(const float* p, float (&v)[VEC]);
template <> __device__ __forceinline__ void load_vec<1>(const float* p, float (&v)[1]) { v[0] = *p; }
template <> __device__ __forceinline__ void load_vec<4>(const float* p, float (&v)[4]) {
    const float4 q = *reinterpret_cast<const float4*>(p);
    v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
}
// nt: nontemporal store (volumes larger than the 256 MB infinity cache gain nothing from allocating in it)
template <int VEC> __device__ __forceinline__ void store_vec(float* p, const float (&v)[VEC], int nt);
template <> __device__ __forceinline__ void store_vec<1>(float* p, const float (&v)[1], int nt) {
    if (nt) __builtin_nontemporal_store(v[0], p); else *p = v[0];
}
template <> __device__ __forceinline__ void store_vec<4>(float* p, const float (&v)[4], int nt) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    if (nt) {
        v4f q = {v[0], v[1], v[2], v[3]};
        __builtin_nontemporal_store(q, reinterpret_cast<v4f*>(p));
    } else {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

enum { MODE_WARP = 0, MODE_CONCAT = 1, MODE_CORR = 2 };

// One thread: VEC consecutive columns of one (b, j, h); loops over the C channels.
//  MODE_WARP   out0 = y_warped [B,C,nd,H,W], out1 = x_warped (or null)
//  MODE_CONCAT out0 = volume [B,2C,nd,H,W], gate = att [B,nd,H,W] (or null)
//  MODE_CORR   out0 = [B,nd,H,W] = mean_c x * warp(y)
template <int MODE, int VEC>
__global__ __launch_bounds__(256) void warp_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                    const float* __restrict__ disp, const float* __restrict__ gate,
                                                    float* __restrict__ out0, float* __restrict__ out1,
                                                    int C, int H, int W, int nd, float half_w, float half_h,
                                                    long long total, int nt) {
    const long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int WQ = W / VEC;
    const int wq = (int)(idx % WQ);
    long long t = idx / WQ;
    const int h = (int)(t % H); t /= H;
    const int j = (int)(t % nd);
    const long long b = t / nd;
    const int w0 = wq * VEC;
    const long long plane = (long long)H * W;
    const long long pix = (long long)h * W + w0;

    float dv[VEC];
    load_vec<VEC>(disp + (b * nd + j) * plane + pix, dv);
    Taps tp[VEC];
#pragma unroll
    for (int p = 0; p < VEC; ++p) tp[p] = make_taps(dv[p], h, w0 + p, H, W, half_w, half_h);

    if (MODE == MODE_CORR) {
        float acc[VEC];
#pragma unroll
        for (int p = 0; p < VEC; ++p) acc[p] = 0.f;
#pragma unroll 4
        for (int c = 0; c < C; ++c) {
            const float* yp = y + (b * C + c) * plane;
            float xv[VEC];
            load_vec<VEC>(x + (b * C + c) * plane + pix, xv);
#pragma unroll
            for (int p = 0; p < VEC; ++p) acc[p] = ss::add_rn(acc[p], ss::mul_rn(xv[p], sample(yp, tp[p])));
        }
#pragma unroll
        for (int p = 0; p < VEC; ++p) acc[p] = acc[p] / (float)C;
        store_vec<VEC>(out0 + (b * nd + j) * plane + pix, acc, 0);
        return;
    }

    float g[VEC];
    if (MODE == MODE_CONCAT && gate != nullptr) load_vec<VEC>(gate + (b * nd + j) * plane + pix, g);
    const bool both = (MODE == MODE_CONCAT) && (x != nullptr);     // MODE_CONCAT with x == null: the gated right half alone
    const int CO = both ? 2 * C : C;
#pragma unroll 4
    for (int c = 0; c < C; ++c) {
        const float* yp = y + (b * C + c) * plane;
        float wv[VEC];
#pragma unroll
        for (int p = 0; p < VEC; ++p) wv[p] = sample(yp, tp[p]);
        if (MODE == MODE_WARP) {
            store_vec<VEC>(out0 + ((b * CO + c) * nd + j) * plane + pix, wv, nt);
            if (out1 != nullptr) {
                float xv[VEC];
                load_vec<VEC>(x + (b * C + c) * plane + pix, xv);
                store_vec<VEC>(out1 + ((b * CO + c) * nd + j) * plane + pix, xv, nt);
            }
        } else {
            if (gate != nullptr) {
#pragma unroll
                for (int p = 0; p < VEC; ++p) wv[p] = ss::mul_rn(g[p], wv[p]);
            }
            if (both) {
                float xv[VEC];
                load_vec<VEC>(x + (b * C + c) * plane + pix, xv);
                if (gate != nullptr) {
#pragma unroll
                    for (int p = 0; p < VEC; ++p) xv[p] = ss::mul_rn(g[p], xv[p]);
                }
                store_vec<VEC>(out0 + ((b * CO + c) * nd + j) * plane + pix, xv, nt);
            }
            store_vec<VEC>(out0 + ((b * CO + (both ? C : 0) + c) * nd + j) * plane + pix, wv, nt);
        }
    }
}

// The live form of the concat volume (MODE_CONCAT without the left half, one column per lane) with no branch and no
// 64-bit per-lane arithmetic in the channel loop: both taps of a row come from ONE unconditional 8-byte buffer load at a
// clamped column (rows or whole pairs outside the image: an offset beyond the buffer, which reads zeros), the two values
// are then picked by selects; stores go through a buffer descriptor with a scalar channel offset.  (The generic kernel
// takes a divergent branch per row and channel around its loads: the compiler then waits with vmcnt(0) after each.)
template <bool NT>
__global__ __launch_bounds__(256) void warp_right_gated(const float* __restrict__ y, const float* __restrict__ disp,
                                                         const float* __restrict__ gate, float* __restrict__ out,
                                                         int C, int H, int W, int nd, float half_w, float half_h) {
    const int w = blockIdx.x * 64 + (threadIdx.x & 63);
    const int h = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int j = blockIdx.z % nd, b = blockIdx.z / nd;
    if (h >= H) return;                                        // wave-uniform
    const bool inside = w < W;
    const long long plane = (long long)H * W;
    const int pix = h * W + min(w, W - 1);
    const float dv = disp[((long long)b * nd + j) * plane + pix];
    const float g = gate ? gate[((long long)b * nd + j) * plane + pix] : 1.0f;
    // the reference's coordinate round trip (see make_taps)
    const float gx = ((float)w - dv) / half_w - 1.0f, gy = (float)h / half_h - 1.0f;
    const float ix = ss::mul_rn(gx + 1.0f, half_w), iy = ss::mul_rn(gy + 1.0f, half_h);
    const float xw = floorf(ix), yn = floorf(iy);
    const float fw = ix - xw, fe = 1.0f - fw, fn = iy - yn, fs = 1.0f - fn;
    const float w_nw = ss::mul_rn(fs, fe), w_ne = ss::mul_rn(fs, fw), w_sw = ss::mul_rn(fn, fe), w_se = ss::mul_rn(fn, fw);
    // columns: west tap xi, east tap xi + 1; the pair is fetched at xc = clamp(xi, 0, W - 2)
    const bool xfin = (xw >= -1.0f) && (xw <= (float)(W - 1));      // at least one tap of the pair inside
    const int xi = xfin ? (int)xw : -2;
    const int xc = min(max(xi, 0), W - 2);
    const bool west_is_y = (xi == xc + 1);                     // xi = W - 1: the west tap is the pair's second element
    const bool east_is_x = (xi == xc - 1);                     // xi = -1: the east tap is the pair's first element
    const bool west_ok = xfin && xi >= 0, east_ok = xfin && xi <= W - 2;
    const int yi = (int)yn;
    const bool n_ok = (yn > -1.0f) && (yn < (float)H), s_ok = (yn + 1.0f > -1.0f) && (yn + 1.0f < (float)H);
    const unsigned off_n = (inside && xfin && n_ok) ? (unsigned)((yi * W + xc) * 4) : 0x80000000u;
    const unsigned off_s = (inside && xfin && s_ok) ? (unsigned)(((yi + 1) * W + xc) * 4) : 0x80000000u;
    const __amdgpu_buffer_rsrc_t yres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(y + (long long)b * C * plane), 0, (int)min((long long)C * plane * 4, 0x7fffffffLL), 0x00020000);
    const __amdgpu_buffer_rsrc_t ores = __builtin_amdgcn_make_buffer_rsrc(
        out + (long long)b * C * nd * plane, 0, (int)min((long long)C * nd * plane * 4, 0x7fffffffLL), 0x00020000);
    const unsigned off_o = inside ? (unsigned)(((long long)j * plane + pix) * 4) : 0x80000000u;
    const int ystep = (int)(plane * 4), ostep = (int)((long long)nd * plane * 4);
    typedef float f2 __attribute__((ext_vector_type(2)));
    // The row coordinate depends on h alone, and h is wave-uniform here (one row per wave): on the 3 rows in 4 whose round trip
    // lands exactly on the integer (fn == 0: both south weights are 0) the south pair is not fetched at all -- half of the kernel's
    // loads.  Same value for finite features (x + 0 * s == x); a NaN / inf in the skipped row no longer reaches the output
    // through 0 * inf, which is the one documented difference from F.grid_sample.
    if (__builtin_amdgcn_readfirstlane(__float_as_int(fn)) == 0) {
#pragma unroll 8
        for (int c = 0; c < C; ++c) {
            const f2 pn = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(yres, (int)off_n, c * ystep, 0));
            const float a = west_ok ? (west_is_y ? pn.y : pn.x) : 0.f, bq = east_ok ? (east_is_x ? pn.x : pn.y) : 0.f;
            float r = ss::add_rn(ss::mul_rn(a, w_nw), ss::mul_rn(bq, w_ne));
            if (gate) r = ss::mul_rn(g, r);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, r), ores, (int)off_o, c * ostep, NT ? 2 : 0);
        }
        return;
    }
#pragma unroll 8
    for (int c = 0; c < C; ++c) {
        const f2 pn = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(yres, (int)off_n, c * ystep, 0));
        const f2 ps = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(yres, (int)off_s, c * ystep, 0));
        const float a = west_ok ? (west_is_y ? pn.y : pn.x) : 0.f, bq = east_ok ? (east_is_x ? pn.x : pn.y) : 0.f;
        const float cq = west_ok ? (west_is_y ? ps.y : ps.x) : 0.f, d = east_ok ? (east_is_x ? ps.x : ps.y) : 0.f;
        float r = ss::mul_rn(a, w_nw);
        r = ss::add_rn(r, ss::mul_rn(bq, w_ne));
        r = ss::add_rn(r, ss::mul_rn(cq, w_sw));
        r = ss::add_rn(r, ss::mul_rn(d, w_se));
        if (gate) r = ss::mul_rn(g, r);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, r), ores, (int)off_o, c * ostep, NT ? 2 : 0);
    }
}

// ---- the warped half of the sparse concat volume written PRE-SPLIT for the matrix-core stem (conv3d_pre.hip): every value as
// the two fp16 terms of x * 2^(E_ONE - e), 8 channels per 16-byte slot, [B][C/8][2 terms][nd][H][W][8].  One block
// exponent e per batch element from a BOUND of the volume, max|y[b]| * max|att[b]| (bilinear weights sum to 1, so no element
// exceeds it): known before the first element is produced, and everything within 2^-17 of it keeps all 24 bits (below that
// the absolute error is 2^-39 of the bound -- the contract of the on-the-fly split in conv3d_bf16s.hip, with the tensor's
// bound in place of the tile's maximum).  Same arithmetic per element as warp_right_gated.

// maxima of |y[b]| and |att[b]| as the bit patterns of non-negative floats: amax[2 * b], amax[2 * b + 1] (zeroed before)
__global__ __launch_bounds__(256) void absmax2_kernel(const float* __restrict__ y, long long ny, const float* __restrict__ att,
                                                       long long na, unsigned* __restrict__ amax) {
    const int b = blockIdx.y;
    const float* yb = y + (long long)b * ny;
    float m = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < ny; i += (long long)gridDim.x * 256) m = fmaxf(m, fabsf(yb[i]));
    unsigned mb = __float_as_uint(m);
    for (int o = 32; o > 0; o >>= 1) mb = max(mb, (unsigned)__shfl_xor((int)mb, o));
    if ((threadIdx.x & 63) == 0 && mb) atomicMax(&amax[2 * b], mb);
    if (att != nullptr) {
        const float* ab = att + (long long)b * na;
        float ma = 0.f;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < na; i += (long long)gridDim.x * 256) ma = fmaxf(ma, fabsf(ab[i]));
        unsigned ab_ = __float_as_uint(ma);
        for (int o = 32; o > 0; o >>= 1) ab_ = max(ab_, (unsigned)__shfl_xor((int)ab_, o));
        if ((threadIdx.x & 63) == 0 && ab_) atomicMax(&amax[2 * b + 1], ab_);
    }
}

__device__ __forceinline__ void split2_pk_f16_w(float x0, float x1, unsigned& h, unsigned& l) {     // (split_f16.h's, local copy)
    typedef float f32x2_w __attribute__((ext_vector_type(2)));
    typedef _Float16 f16x2_w __attribute__((ext_vector_type(2)));
    const f32x2_w v = {x0, x1};
    const f16x2_w hv = __builtin_convertvector(v, f16x2_w);
    const f32x2_w r = {x0 - (float)hv[0], x1 - (float)hv[1]};
    h = __builtin_bit_cast(unsigned, hv);
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2_w));
}

constexpr int W_E_MIN = 16, W_E_ONE = 141;                     // split_f16.h: E_MIN, E_ONE

__global__ __launch_bounds__(256) void warp_right_presplit(const float* __restrict__ y, const float* __restrict__ disp,
                                                            const float* __restrict__ gate, const unsigned* __restrict__ amax,
                                                            uint4* __restrict__ xs, int* __restrict__ xexp, int C, int H, int W, int nd,
                                                            float half_w, float half_h) {
    const int w = blockIdx.x * 64 + (threadIdx.x & 63);
    const int h = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int j = blockIdx.z % nd, b = blockIdx.z / nd;
    // the block exponent of this batch element (every workgroup derives the same value; one of them publishes it)
    const float bound = __uint_as_float(amax[2 * b]) * (gate ? __uint_as_float(amax[2 * b + 1]) : 1.0f);
    const int e = min(max((int)(__float_as_uint(bound) >> 23) & 0xff, W_E_MIN), 254);
    if (blockIdx.x == 0 && blockIdx.y == 0 && j == 0 && threadIdx.x == 0) xexp[b] = e;
    const float in_scale = __uint_as_float((unsigned)(127 + W_E_ONE - e) << 23);
    if (h >= H) return;                                        // wave-uniform
    const bool inside = w < W;
    const long long plane = (long long)H * W;
    const int pix = h * W + min(w, W - 1);
    const float dv = disp[((long long)b * nd + j) * plane + pix];
    const float g = gate ? gate[((long long)b * nd + j) * plane + pix] : 1.0f;
    const float gx = ((float)w - dv) / half_w - 1.0f, gy = (float)h / half_h - 1.0f;
    const float ix = ss::mul_rn(gx + 1.0f, half_w), iy = ss::mul_rn(gy + 1.0f, half_h);
    const float xw = floorf(ix), yn = floorf(iy);
    const float fw = ix - xw, fe = 1.0f - fw, fn = iy - yn, fs = 1.0f - fn;
    const float w_nw = ss::mul_rn(fs, fe), w_ne = ss::mul_rn(fs, fw), w_sw = ss::mul_rn(fn, fe), w_se = ss::mul_rn(fn, fw);
    const bool xfin = (xw >= -1.0f) && (xw <= (float)(W - 1));
    const int xi = xfin ? (int)xw : -2;
    const int xc = min(max(xi, 0), W - 2);
    const bool west_is_y = (xi == xc + 1), east_is_x = (xi == xc - 1);
    const bool west_ok = xfin && xi >= 0, east_ok = xfin && xi <= W - 2;
    const int yi = (int)yn;
    const bool n_ok = (yn > -1.0f) && (yn < (float)H), s_ok = (yn + 1.0f > -1.0f) && (yn + 1.0f < (float)H);
    const unsigned off_n = (inside && xfin && n_ok) ? (unsigned)((yi * W + xc) * 4) : 0x80000000u;
    const unsigned off_s = (inside && xfin && s_ok) ? (unsigned)(((yi + 1) * W + xc) * 4) : 0x80000000u;
    const __amdgpu_buffer_rsrc_t yres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(y + (long long)b * C * plane), 0, (int)min((long long)C * plane * 4, 0x7fffffffLL), 0x00020000);
    const __amdgpu_buffer_rsrc_t ores = __builtin_amdgcn_make_buffer_rsrc(
        xs + (long long)b * (C / 8) * nd * plane * 2, 0, (int)min((long long)C * nd * plane * 4, 0x7fffffffLL), 0x00020000);
    const unsigned off_o = inside ? (unsigned)pix * 16u : 0x80000000u;
    const int ystep = (int)(plane * 4);
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    for (int c0 = 0; c0 < C; c0 += 8) {
        float r8[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int c = c0 + k;
            const f2 pn = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(yres, (int)off_n, c * ystep, 0));
            const f2 ps = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(yres, (int)off_s, c * ystep, 0));
            const float a = west_ok ? (west_is_y ? pn.y : pn.x) : 0.f, bq = east_ok ? (east_is_x ? pn.x : pn.y) : 0.f;
            const float cq = west_ok ? (west_is_y ? ps.y : ps.x) : 0.f, d = east_ok ? (east_is_x ? ps.x : ps.y) : 0.f;
            float r = ss::mul_rn(a, w_nw);
            r = ss::add_rn(r, ss::mul_rn(bq, w_ne));
            r = ss::add_rn(r, ss::mul_rn(cq, w_sw));
            r = ss::add_rn(r, ss::mul_rn(d, w_se));
            if (gate) r = ss::mul_rn(g, r);
            r8[k] = r * in_scale;                              // exact: a power of two
        }
        unsigned hh[4], ll[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) split2_pk_f16_w(r8[2 * k], r8[2 * k + 1], hh[k], ll[k]);
        const int soff_h = (int)((((long long)(c0 / 8) * 2 * nd + j) * plane) * 16), soff_l = soff_h + (int)((long long)nd * plane * 16);
        const u4 hv = {hh[0], hh[1], hh[2], hh[3]}, lv = {ll[0], ll[1], ll[2], ll[3]};
        __builtin_amdgcn_raw_buffer_store_b128(hv, ores, (int)off_o, soff_h, 0);
        __builtin_amdgcn_raw_buffer_store_b128(lv, ores, (int)off_o, soff_l, 0);
    }
}

// ---- backward of SpatialTransformer_grid (autograd of the reference's meshgrid -> normalise -> F.grid_sample composition,
// models/submodule.py:265-288): one thread per (b, j, h, w), loop over the channels.
//   grad_y[b,c,tap] += w_tap * g[b,c,j,h,w]                         (scatter through hardware fp32 atomics)
//   grad_disp[b,j,h,w] = -(half_w * sum_c gix_c) / half_w           gix as ATen's grid_sampler backward accumulates it
// (the row coordinate does not depend on the disparity).  grad_x = sum_j grad_x_warped is the second kernel.
__global__ __launch_bounds__(256) void warp_bwd_kernel(const float* __restrict__ gyw, const float* __restrict__ y,
                                                        const float* __restrict__ disp, float* __restrict__ gy,
                                                        float* __restrict__ gdisp, int C, int H, int W, int nd, float half_w,
                                                        float half_h, long long total) {
    const long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int w = (int)(idx % W);
    long long t = idx / W;
    const int h = (int)(t % H); t /= H;
    const int j = (int)(t % nd);
    const long long b = t / nd;
    const long long plane = (long long)H * W;
    const long long pix = (long long)h * W + w;
    const float dv = disp[(b * nd + j) * plane + pix];
    const Taps tp = make_taps(dv, h, w, H, W, half_w, half_h);
    // fs = iy_se - iy, fn = iy - iy_nw (make_taps' names: w_nw = fs * fe ...): recover the row fractions
    const float gyc = (float)h / half_h - 1.0f;
    const float iy = ss::mul_rn(gyc + 1.0f, half_h);
    const float fn = iy - floorf(iy), fs = 1.0f - fn;
    float gix = 0.f;
    for (int c = 0; c < C; ++c) {
        const float g = gyw[((b * C + c) * nd + j) * plane + pix];
        const float* yp = y + (b * C + c) * plane;
        float* gp = gy + (b * C + c) * plane;
        const float a = tp.o_nw >= 0 ? yp[tp.o_nw] : 0.f, bq = tp.o_ne >= 0 ? yp[tp.o_ne] : 0.f;
        const float cq = tp.o_sw >= 0 ? yp[tp.o_sw] : 0.f, d = tp.o_se >= 0 ? yp[tp.o_se] : 0.f;
        if (gy != nullptr) {
            // (taps of weight zero add nothing: with the INTEGER candidates of models/SemStereo.py:299-305 that is three taps in four on most
            // columns and rows -- 3.3 ms of the 1024^2 training step were 201 M unconditional atomics, measured r06)
            if (tp.o_nw >= 0 && tp.w_nw != 0.f) unsafeAtomicAdd(gp + tp.o_nw, tp.w_nw * g);
            if (tp.o_ne >= 0 && tp.w_ne != 0.f) unsafeAtomicAdd(gp + tp.o_ne, tp.w_ne * g);
            if (tp.o_sw >= 0 && tp.w_sw != 0.f) unsafeAtomicAdd(gp + tp.o_sw, tp.w_sw * g);
            if (tp.o_se >= 0 && tp.w_se != 0.f) unsafeAtomicAdd(gp + tp.o_se, tp.w_se * g);
        }
        gix -= a * fs * g;
        gix += bq * fs * g;
        gix -= cq * fn * g;
        gix += d * fn * g;
    }
    if (gdisp != nullptr) gdisp[(b * nd + j) * plane + pix] = -((half_w * gix) / half_w);
}

// grad_y alone (no disparity gradient: the candidates of models/SemStereo.py:316 are indices), row by row: a workgroup owns output row h
// of WB_CC channels and sums every candidate's contributions to ROW h of grad_y in LDS, then adds the row to memory once -- C * H * W
// global atomics instead of up to 4 * C * nd * H * W (r06: 65 M after the zero-weight taps were dropped, 1.33 ms of the 1024^2 training
// step at the ~50 G atomics/s the part sustains).  The south taps of the rows whose coordinate is not exact land in row h + 1: those
// few go to memory directly.  Each of the four waves owns TWO channel rows and walks the whole row of pixels for them, so that its sums
// need no ds_add_f32 (170 clocks per wave instruction on gfx950; ss::lds_owned_add2, common.h: the 50 M of them were 0.25 of the kernel's
// 0.43 ms): one claim per tap serves both channels, whose taps are the same.
constexpr int WB_CC = 8, WB_WAVES = 4;
__global__ __launch_bounds__(64 * WB_WAVES) void warp_bwd_rows_kernel(const float* __restrict__ gyw, const float* __restrict__ disp,
                                                                      float* __restrict__ gy, int C, int H, int W, int nd, float half_w,
                                                                      float half_h) {
    extern __shared__ float wb_row[];                     // [WB_CC][W] sums, then [WB_WAVES][W] tags
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = blockIdx.x, c0 = blockIdx.y * WB_CC;
    const long long b = blockIdx.z;
    const long long plane = (long long)H * W;
    const int ca = c0 + wave, cb = ca + WB_WAVES;         // this wave's channels
    if (ca >= C) return;                                  // (wave-uniform; no workgroup barrier below)
    const bool two = cb < C;
    float* ra = wb_row + (size_t)wave * W;
    float* rbw = wb_row + (size_t)(wave + WB_WAVES) * W;
    int* tag = reinterpret_cast<int*>(wb_row + (size_t)WB_CC * W) + (size_t)wave * W;
    for (int i = lane; i < W; i += 64) { ss::lds_put(ra + i, 0.f); ss::lds_put(rbw + i, 0.f); }
    __builtin_amdgcn_wave_barrier();
    const float* ga_p = gyw + (b * C + ca) * nd * plane;
    const float* gb_p = gyw + (b * C + (two ? cb : ca)) * nd * plane;
    float* gpa = gy + (b * C + ca) * plane;
    float* gpb = gy + (b * C + (two ? cb : ca)) * plane;
    const int base = h * W;
    constexpr int JU = 4;                                 // candidates whose loads are issued together (the wave is alone with its latency)
    for (int w = lane; w < W; w += 64) {
        const long long pix = (long long)h * W + w;
        for (int j0 = 0; j0 < nd; j0 += JU) {
            float dv[JU], gav[JU], gbv[JU];
#pragma unroll
            for (int u = 0; u < JU; ++u) {
                const int j = min(j0 + u, nd - 1);
                dv[u] = disp[(b * nd + j) * plane + pix];
                gav[u] = ga_p[j * plane + pix];
                gbv[u] = two ? gb_p[j * plane + pix] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < JU; ++u) {
                if (j0 + u >= nd) break;
                const Taps tp = make_taps(dv[u], h, w, H, W, half_w, half_h);
                const float ga = gav[u], gb = gbv[u];
                // (column of a tap inside its row; the north taps lie in row h wherever the row coordinate is exact -- checked, not assumed)
                auto add = [&](int o, float wt) {
                    const bool any = o >= 0 && wt != 0.f;
                    const bool in_row = any && o >= base && o < base + W;
                    if (any && !in_row) {
                        unsafeAtomicAdd(gpa + o, wt * ga);
                        if (two) unsafeAtomicAdd(gpb + o, wt * gb);
                    }
                    ss::lds_owned_add2(tag, (unsigned)(o - base), in_row, ra, wt * ga, rbw, wt * gb, two);
                };
                add(tp.o_nw, tp.w_nw); add(tp.o_ne, tp.w_ne); add(tp.o_sw, tp.w_sw); add(tp.o_se, tp.w_se);
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < W; i += 64) {
        const float va = ss::lds_get(ra + i), vb = ss::lds_get(rbw + i);
        if (va != 0.f) unsafeAtomicAdd(gpa + (long long)h * W + i, va);
        if (two && vb != 0.f) unsafeAtomicAdd(gpb + (long long)h * W + i, vb);
    }
}

// ... and where a row is whole 64-pixel blocks (W % 64 == 0: every shape of the path), pixel block by pixel block: a wave owns 64
// pixels of row h and WBC_CH channels, and sums all nd candidates' contributions in TWO private windows -- the 64 columns +- WBC_M
// (enough for |disparity| <= 64 at quarter resolution) of row h and of the one other row its taps reach -- which it then adds to memory
// once; a tap beyond them goes to memory directly.  The other row is not an exception: the reference's coordinate round trip
// iy = ((h / half_h - 1) + 1) * half_h leaves iy = h -+ ~1e-5 on most rows (and ix likewise), so a candidate that is an exact integer
// still has four live taps, two of them in row h + 1 or h - 1 with weights ~1e-5 -- which the row kernel above sends to memory one
// atomic at a time (ablation r06, profiles/EXPERIMENTS.md F.16: 40 us of this kernel's 167 were its loads and sums, 125 those atomics).
#ifndef SS_WBC_CH
#define SS_WBC_CH 4
#endif
#ifndef SS_WBC_NW
#define SS_WBC_NW 4
#endif
constexpr int WBC_NW = SS_WBC_NW, WBC_CH = SS_WBC_CH, WBC_M = 64, WBC_RB = 64 + 2 * WBC_M + 2, WBC_RS = WBC_RB + 62, WBC_JU = 2;
__global__ __launch_bounds__(64 * WBC_NW) void warp_bwd_blocks_kernel(const float* __restrict__ gyw, const float* __restrict__ disp,
                                                                      float* __restrict__ gy, int C, int H, int W, int nd, float half_w,
                                                                      float half_h) {
    __shared__ float rowbuf[WBC_NW][2][WBC_CH][WBC_RS];        // [wave][row h / the other row][channel][column - xb]
    __shared__ int rowtag[WBC_NW][2][WBC_RS];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long plane = (long long)H * W;
    const long long i = (long long)blockIdx.x * 64 + lane;    // over B * H * W (whole blocks: W % 64 == 0)
    const int w = (int)(i % W), h = (int)((i / W) % H);
    const long long b = i / plane, pix = (long long)h * W + w;
    const int c0 = (blockIdx.y * WBC_NW + wave) * WBC_CH;
    if (c0 >= C) return;                                      // (wave-uniform; no workgroup barrier in this kernel)
    const int nlive = min(WBC_CH, C - c0);                    // (wave-uniform)
    float* rows = &rowbuf[wave][0][0][0];
    int* tags = &rowtag[wave][0][0];
#pragma unroll
    for (int n = 0; n < 2 * WBC_CH; ++n)
#pragma unroll
        for (int k = 0; k < 4; ++k) ss::lds_put(&rows[n * WBC_RS + lane + 64 * k], 0.f);
    const float* g_p = gyw + (b * C + c0) * nd * plane;
    float* gp = gy + (b * C + c0) * plane;
    const int xb = (int)(((long long)blockIdx.x * 64) % W) - WBC_M;      // column of slot 0 of the windows
    // the rows of the taps (the same for every candidate and lane of this wave): yn = floor(iy) and yn + 1, one of them h
    const int yn = (int)floorf(ss::mul_rn(((float)h / half_h - 1.0f) + 1.0f, half_h));
    const int other = (yn == h) ? h + 1 : yn;
    for (int j0 = 0; j0 < nd; j0 += WBC_JU) {
        float dv[WBC_JU], gv[WBC_JU][WBC_CH];
#pragma unroll
        for (int u = 0; u < WBC_JU; ++u) {
            const int j = min(j0 + u, nd - 1);
            dv[u] = disp[(b * nd + j) * plane + pix];
#pragma unroll
            for (int n = 0; n < WBC_CH; ++n) gv[u][n] = (n < nlive) ? g_p[((long long)n * nd + j) * plane + pix] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < WBC_JU; ++u) {
            if (j0 + u >= nd) break;
            const Taps tp = make_taps(dv[u], h, w, H, W, half_w, half_h);
            auto add = [&](int o, int row, float wt) {
                const bool any = o >= 0 && wt != 0.f;
                if (__builtin_amdgcn_ballot_w64(any) == 0) return;             // (one scalar branch for a tap nobody has)
                const int r = (row == h) ? 0 : 1;
                const unsigned k = (unsigned)(o - row * W - xb);               // the tap's column inside the window
                const bool in_win = any && (row == h || row == other) && k < (unsigned)WBC_RB;
                float v[WBC_CH];
#pragma unroll
                for (int n = 0; n < WBC_CH; ++n) v[n] = wt * gv[u][n];
                if (__builtin_amdgcn_ballot_w64(any && !in_win) != 0) {        // (beyond the windows: |disparity| > 64, or a row that is neither)
                    for (int n = 0; n < nlive; ++n)
                        if (any && !in_win) unsafeAtomicAdd(gp + n * plane + o, v[n]);
                }
                ss::lds_owned_addn<WBC_CH>(tags + r * WBC_RS, k, in_win, rows + r * (WBC_CH * WBC_RS), WBC_RS, v, nlive);
            };
            // (the row of a tap from its offset: the north pair lies in row yn_t, the south pair below it -- per candidate, not assumed)
            const int row_n = (tp.o_nw >= 0 ? tp.o_nw : tp.o_ne) / W, row_s = (tp.o_sw >= 0 ? tp.o_sw : tp.o_se) / W;
            add(tp.o_nw, row_n, tp.w_nw); add(tp.o_ne, row_n, tp.w_ne); add(tp.o_sw, row_s, tp.w_sw); add(tp.o_se, row_s, tp.w_se);
        }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int row = r == 0 ? h : other;
        if (row < 0 || row >= H) continue;                    // (wave-uniform)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int idx = lane + 64 * k, col = xb + idx;
            if (idx < WBC_RB && col >= 0 && col < W) {
#pragma unroll
                for (int n = 0; n < WBC_CH; ++n) {
                    const float v = ss::lds_get(&rows[(r * WBC_CH + n) * WBC_RS + idx]);
                    if (n < nlive && v != 0.f) unsafeAtomicAdd(gp + n * plane + (long long)row * W + col, v);
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void sum_over_candidates_kernel(const float* __restrict__ gxw, float* __restrict__ gx, int nd,
                                                                   long long plane, long long total) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;     // over B*C*H*W
    if (i >= total) return;
    const long long bc = i / plane, pix = i % plane;
    float s = 0.f;
    for (int j = 0; j < nd; ++j) s += gxw[(bc * nd + j) * plane + pix];
    gx[i] = s;
}

template <int MODE>
int launch(const float* x, const float* y, const float* disp, const float* gate, float* out0, float* out1, int B,
           int C, int H, int W, int nd, hipStream_t st) {
    // the reference divides by the Python float (W-1)/2 computed in double; cast once to fp32
    const float half_w = (float)((W - 1.0) / 2.0), half_h = (float)((H - 1.0) / 2.0);
    uintptr_t bits = reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(disp) |
                     reinterpret_cast<uintptr_t>(out0) | reinterpret_cast<uintptr_t>(out1) |
                     reinterpret_cast<uintptr_t>(gate);
    // One column per lane by default: a wave's tap loads then touch 2 cache lines instead of 8, which
    // measures ~25 us faster on the live shape than the float4 form (SS_WARP_VEC=4) despite 4-byte stores.
    bool v4 = false;
    if (ss::tuning().warp_vec4) v4 = (W % 4 == 0) && ((bits & 15) == 0);
    const long long total = (long long)B * nd * H * (v4 ? W / 4 : W);
    const long long blocks = ss::ceil_div_ll(total, 256);
    if (blocks > 0x7fffffffLL) return SS_ERR_UNSUPPORTED;
    const int planes = (MODE == MODE_CORR) ? 1 : ((MODE == MODE_CONCAT && x != nullptr) || out1 != nullptr) ? 2 * C : C;
    int nt = (size_t)B * planes * nd * H * W * sizeof(float) > ((size_t)192 << 20);
    if (ss::tuning().warp_stream >= 0) nt = ss::tuning().warp_stream == 1;      // tuning aid
    const bool fast_ok = ss::tuning().warp_generic < 0;
    if (MODE == MODE_CONCAT && x == nullptr && !v4 && fast_ok && W >= 2 && (long long)C * nd * H * W * 4 < 0x7fffffffLL &&
        (long long)B * nd <= 65535) {
        const dim3 grid(ss::ceil_div(W, 64), ss::ceil_div(H, 4), B * nd);
        if (nt) hipLaunchKernelGGL(warp_right_gated<true>, grid, dim3(256), 0, st, y, disp, gate, out0, C, H, W, nd, half_w, half_h);
        else hipLaunchKernelGGL(warp_right_gated<false>, grid, dim3(256), 0, st, y, disp, gate, out0, C, H, W, nd, half_w, half_h);
        return ss::check_launch();
    }
    if (v4)
        hipLaunchKernelGGL((warp_kernel<MODE, 4>), dim3((unsigned)blocks), dim3(256), 0, st, x, y, disp, gate, out0,
                           out1, C, H, W, nd, half_w, half_h, total, nt);
    else
        hipLaunchKernelGGL((warp_kernel<MODE, 1>), dim3((unsigned)blocks), dim3(256), 0, st, x, y, disp, gate, out0,
                           out1, C, H, W, nd, half_w, half_h, total, nt);
    return ss::check_launch();
}

}  // namespace

extern "C" int ss_warp_sampled_fwd(const float* x, const float* y, const float* disp, float* y_warped,
                                   float* x_warped, int B, int C, int H, int W, int nd, ss_stream_t stream) {
    SS_REQUIRE(x && y && disp && y_warped);
    SS_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && nd > 0);
    return launch<MODE_WARP>(x, y, disp, nullptr, y_warped, x_warped, B, C, H, W, nd, ss::as_stream(stream));
}

extern "C" int ss_warp_sampled_bwd(const float* grad_y_warped, const float* grad_x_warped, const float* y, const float* disp,
                                   float* grad_x, float* grad_y, float* grad_disp, int B, int C, int H, int W, int nd,
                                   ss_stream_t stream) {
    SS_REQUIRE(y && disp);
    SS_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && nd > 0);
    hipStream_t st = ss::as_stream(stream);
    const long long plane = (long long)H * W;
    if (grad_y_warped != nullptr && (grad_y != nullptr || grad_disp != nullptr)) {
        if (grad_y != nullptr && hipMemsetAsync(grad_y, 0, (size_t)B * C * plane * sizeof(float), st) != hipSuccess) return SS_ERR_LAUNCH;
        const float half_w = (float)((W - 1.0) / 2.0), half_h = (float)((H - 1.0) / 2.0);
        const long long total = (long long)B * nd * plane, blocks = ss::ceil_div_ll(total, 256);
        if (blocks > 0x7fffffffLL) return SS_ERR_UNSUPPORTED;
        if (grad_disp == nullptr && W % 64 == 0 && total / nd / 64 <= 0x7fffffffLL && ss::ceil_div(C, WBC_NW * WBC_CH) <= 65535)
            hipLaunchKernelGGL(warp_bwd_blocks_kernel, dim3((unsigned)(total / nd / 64), ss::ceil_div(C, WBC_NW * WBC_CH)), dim3(64 * WBC_NW), 0, st,
                               grad_y_warped, disp, grad_y, C, H, W, nd, half_w, half_h);
        else if (grad_disp == nullptr && W <= 1024 && H <= 65535 && B <= 65535 && ss::ceil_div(C, WB_CC) <= 65535)
            hipLaunchKernelGGL(warp_bwd_rows_kernel, dim3(H, ss::ceil_div(C, WB_CC), B), dim3(64 * WB_WAVES), (size_t)(WB_CC + WB_WAVES) * W * sizeof(float), st,
                               grad_y_warped, disp, grad_y, C, H, W, nd, half_w, half_h);
        else
            hipLaunchKernelGGL(warp_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, st, grad_y_warped, y, disp, grad_y, grad_disp, C, H,
                               W, nd, half_w, half_h, total);
    }
    if (grad_x_warped != nullptr && grad_x != nullptr) {
        const long long total = (long long)B * C * plane, blocks = ss::ceil_div_ll(total, 256);
        if (blocks > 0x7fffffffLL) return SS_ERR_UNSUPPORTED;
        hipLaunchKernelGGL(sum_over_candidates_kernel, dim3((unsigned)blocks), dim3(256), 0, st, grad_x_warped, grad_x, nd, plane, total);
    }
    return ss::check_launch();
}

extern "C" int ss_concat_sampled_fwd(const float* left, const float* right, const float* disp, const float* att,
                                     float* out, int B, int C, int H, int W, int nd, ss_stream_t stream) {
    SS_REQUIRE(right && disp && out);                    // left == NULL: only the right half, out [B,C,nd,H,W]
    SS_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && nd > 0);
    return launch<MODE_CONCAT>(left, right, disp, att, out, nullptr, B, C, H, W, nd, ss::as_stream(stream));
}

// ---- backward of ss_concat_sampled_fwd (training: models/SemStereo.py:316-318, `att_topk * cat(left broadcast, warp(right))`, as ONE
// pass over the gradient of the volume instead of the backward of a warp, a cat and a multiply: r06, those were 402 MB x 6 of traffic
// per pair at 1024^2).  The candidates are INDICES (no gradient).  A wave owns 64 pixels of row h and CSB_CH = 4 channels (8 per wave: 169 registers, two waves per SIMD, 300 us against 239; 2: 244); per candidate j:
//   grad_left[c]   += att[j] * gL[c,j]                                  (registers, one plain store at the end)
//   grad_att[j]    += sum_c gL[c,j] * left[c] + gR[c,j] * warp(right)[c,j]   (the workgroup's waves meet in LDS once per pair of candidates)
//   grad_right[c]  <- att[j] * w_tap * gR[c,j] at the four taps           (the two row windows of warp_bwd_blocks_kernel)
// with gL / gR the two halves of the volume's gradient.  `margin`: the |disparity| the windows cover (a hint: taps beyond go to
// memory one atomic at a time, still right).
#ifndef SS_CSB_CH
#define SS_CSB_CH 4
#endif
#ifndef SS_CSB_NW
#define SS_CSB_NW 8
#endif
#ifndef SS_CSB_JU
#define SS_CSB_JU 2
#endif
constexpr int CSB_NW = SS_CSB_NW, CSB_CH = SS_CSB_CH, CSB_JU = SS_CSB_JU;
__global__ __launch_bounds__(64 * CSB_NW) void concat_sampled_bwd_kernel(const float* __restrict__ gvol, const float* __restrict__ left,
                                                                         const float* __restrict__ right, const float* __restrict__ disp,
                                                                         const float* __restrict__ att, float* __restrict__ g_left,
                                                                         float* __restrict__ g_right, float* __restrict__ g_att, int C,
                                                                         int H, int W, int nd, float half_w, float half_h, int margin,
                                                                         int rs, int att_atomic) {
    extern __shared__ float csb_lds[];   // [NW][2][CH][rs] gradient windows | [NW][2][CH][rs] windows of `right` | [NW][2][rs] tags | [2][NW][JU][64] partial grad_att
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* rows = csb_lds + (size_t)wave * (2 * CSB_CH * rs);
    float* rwin = csb_lds + (size_t)(CSB_NW + wave) * (2 * CSB_CH * rs);
    int* tags = reinterpret_cast<int*>(csb_lds + (size_t)2 * CSB_NW * (2 * CSB_CH * rs)) + (size_t)wave * (2 * rs);
    float* red = csb_lds + (size_t)2 * CSB_NW * (2 * CSB_CH * rs) + (size_t)CSB_NW * (2 * rs);
    const int rb = 64 + 2 * margin + 2;                       // live slots of a window
    const long long plane = (long long)H * W;
    const long long i = (long long)blockIdx.x * 64 + lane;    // over B * H * W (whole blocks: W % 64 == 0)
    const int w = (int)(i % W), h = (int)((i / W) % H);
    const long long b = i / plane, pix = (long long)h * W + w;
    const int c0 = (blockIdx.y * CSB_NW + wave) * CSB_CH;
    const int nlive = max(0, min(CSB_CH, C - c0));            // (wave-uniform; 0: the wave only keeps the barriers)
    for (int k = lane; k < 2 * CSB_CH * rs; k += 64) ss::lds_put(&rows[k], 0.f);
    const float* gl_p = gvol + (b * 2 * C + c0) * nd * plane;         // gradient of the broadcast half, channel c0
    const float* gr_p = gvol + (b * 2 * C + C + c0) * nd * plane;     // ... of the warped half
    const float* rp = right + (b * C + c0) * plane;
    float* grp = g_right ? g_right + (b * C + c0) * plane : nullptr;
    const int xb = (int)(((long long)blockIdx.x * 64) % W) - margin;
    const int yn = (int)floorf(ss::mul_rn(((float)h / half_h - 1.0f) + 1.0f, half_h));
    const int other = (yn == h) ? h + 1 : yn;
    float cl[CSB_CH], gcl[CSB_CH];
#pragma unroll
    for (int n = 0; n < CSB_CH; ++n) {
        cl[n] = (n < nlive) ? left[(b * C + c0 + n) * plane + pix] : 0.f;
        gcl[n] = 0.f;
    }
    // the same two windows of `right` itself (zeros outside the image): the forward's warp(right)[c, j], which grad_att needs, is four
    // taps per channel and candidate -- as global gathers they were 100 of the kernel's 246 us (ablation r06, EXPERIMENTS.md F.20)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int row = r == 0 ? h : other;
        const bool row_ok = row >= 0 && row < H;
        for (int idx = lane; idx < rs; idx += 64) {
            const int col = xb + idx;
            const bool ok = row_ok && idx < rb && col >= 0 && col < W;
#pragma unroll
            for (int n = 0; n < CSB_CH; ++n)
                ss::lds_put(&rwin[(r * CSB_CH + n) * rs + idx], (ok && n < nlive) ? rp[n * plane + (long long)row * W + col] : 0.f);
        }
    }
    __builtin_amdgcn_wave_barrier();
    int par = 0;
    for (int j0 = 0; j0 < nd; j0 += CSB_JU, par ^= 1) {
        float dv[CSB_JU], av[CSB_JU], gL[CSB_JU][CSB_CH], gR[CSB_JU][CSB_CH], part[CSB_JU];
#pragma unroll
        for (int u = 0; u < CSB_JU; ++u) {
            const int j = min(j0 + u, nd - 1);
            dv[u] = disp[(b * nd + j) * plane + pix];
            av[u] = att ? att[(b * nd + j) * plane + pix] : 1.f;
#pragma unroll
            for (int n = 0; n < CSB_CH; ++n) {
                gL[u][n] = (n < nlive) ? gl_p[((long long)n * nd + j) * plane + pix] : 0.f;
                gR[u][n] = (n < nlive) ? gr_p[((long long)n * nd + j) * plane + pix] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < CSB_JU; ++u) {
            part[u] = 0.f;
            if (j0 + u >= nd) continue;
            const Taps tp = make_taps(dv[u], h, w, H, W, half_w, half_h);
            const float a = av[u];
            float s = 0.f;
            const int row_n = (tp.o_nw >= 0 ? tp.o_nw : tp.o_ne) / W, row_s = (tp.o_sw >= 0 ? tp.o_sw : tp.o_se) / W;
            // a tap's value: from the window it lies in, from memory beyond (the same arithmetic as warp.hip::sample, tap by tap)
            auto tap = [&](int n, int o, int row) -> float {
                const unsigned k = (unsigned)(o - row * W - xb);
                const bool inw = o >= 0 && (row == h || row == other) && k < (unsigned)rb;
                float v = 0.f;
                if (inw) v = ss::lds_ld(&rwin[(((row == h) ? 0 : 1) * CSB_CH + n) * rs + k]);
                if (o >= 0 && !inw) v = __builtin_nontemporal_load(rp + n * plane + o);
                return v;
            };
            auto wval = [&](int n) -> float {
                float r = ss::mul_rn(tap(n, tp.o_nw, row_n), tp.w_nw);
                r = ss::add_rn(r, ss::mul_rn(tap(n, tp.o_ne, row_n), tp.w_ne));
                r = ss::add_rn(r, ss::mul_rn(tap(n, tp.o_sw, row_s), tp.w_sw));
                r = ss::add_rn(r, ss::mul_rn(tap(n, tp.o_se, row_s), tp.w_se));
                return r;
            };
#pragma unroll
            for (int n = 0; n < CSB_CH; ++n) {
                gcl[n] = fmaf(a, gL[u][n], gcl[n]);
                const float val = (n < nlive) ? wval(n) : 0.f;                          // the forward's warp(right)[c, j]
                s = fmaf(gL[u][n], cl[n], s);
                s = fmaf(gR[u][n], val, s);
            }
            part[u] = s;
            if (grp != nullptr) {
                auto add = [&](int o, int row, float wt) {
                    const bool any = o >= 0 && wt != 0.f;
                    if (__builtin_amdgcn_ballot_w64(any) == 0) return;
                    const int r = (row == h) ? 0 : 1;
                    const unsigned k = (unsigned)(o - row * W - xb);
                    const bool in_win = any && (row == h || row == other) && k < (unsigned)rb;
                    float v[CSB_CH];
#pragma unroll
                    for (int n = 0; n < CSB_CH; ++n) v[n] = (a * wt) * gR[u][n];
                    if (__builtin_amdgcn_ballot_w64(any && !in_win) != 0) {
                        for (int n = 0; n < nlive; ++n)
                            if (any && !in_win) unsafeAtomicAdd(grp + n * plane + o, v[n]);
                    }
                    ss::lds_owned_addn<CSB_CH>(tags + r * rs, k, in_win, rows + r * (CSB_CH * rs), rs, v, nlive);
                };
                add(tp.o_nw, row_n, tp.w_nw); add(tp.o_ne, row_n, tp.w_ne); add(tp.o_sw, row_s, tp.w_sw); add(tp.o_se, row_s, tp.w_se);
            }
        }
        if (g_att != nullptr) {          // (kernel-uniform) the waves' partial sums over their channels -> one value per (candidate, pixel)
#pragma unroll
            for (int u = 0; u < CSB_JU; ++u) red[((par * CSB_NW + wave) * CSB_JU + u) * 64 + lane] = part[u];
            __syncthreads();
            if (wave == 0) {
#pragma unroll
                for (int u = 0; u < CSB_JU; ++u) {
                    if (j0 + u >= nd) continue;
                    float t = 0.f;
#pragma unroll
                    for (int wv = 0; wv < CSB_NW; ++wv) t += red[((par * CSB_NW + wv) * CSB_JU + u) * 64 + lane];
                    float* dst = g_att + (b * nd + j0 + u) * plane + pix;
                    if (att_atomic) unsafeAtomicAdd(dst, t); else *dst = t;
                }
            }
        }
    }
    if (g_left != nullptr) {
#pragma unroll
        for (int n = 0; n < CSB_CH; ++n)
            if (n < nlive) g_left[(b * C + c0 + n) * plane + pix] = gcl[n];
    }
    if (grp == nullptr) return;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int row = r == 0 ? h : other;
        if (row < 0 || row >= H) continue;                    // (wave-uniform)
        for (int idx = lane; idx < rb; idx += 64) {
            const int col = xb + idx;
            if (col < 0 || col >= W) continue;
#pragma unroll
            for (int n = 0; n < CSB_CH; ++n) {
                const float v = ss::lds_get(&rows[(r * CSB_CH + n) * rs + idx]);
                if (n < nlive && v != 0.f) unsafeAtomicAdd(grp + n * plane + (long long)row * W + col, v);
            }
        }
    }
}

extern "C" int ss_concat_sampled_bwd(const float* grad_out, const float* left, const float* right, const float* disp, const float* att,
                                     float* grad_left, float* grad_right, float* grad_att, int B, int C, int H, int W, int nd, int margin,
                                     ss_stream_t stream) {
    SS_REQUIRE(grad_out && left && right && disp && B > 0 && C > 0 && H > 0 && W > 0 && nd > 0 && margin >= 0);
    SS_REQUIRE(grad_att == nullptr || att != nullptr);
    if (W % 64 != 0 || (long long)B * H * W / 64 > 0x7fffffffLL || ss::ceil_div(C, CSB_NW * CSB_CH) > 65535) return SS_ERR_UNSUPPORTED;
    hipStream_t st = ss::as_stream(stream);
    const long long plane = (long long)H * W;
    margin = std::min(margin, 96);
    const int rs = 64 + 2 * margin + 2 + 6;                   // (+ 6: the row stride off the 32-bank period)
    const size_t lds = ((size_t)2 * CSB_NW * 2 * CSB_CH * rs + (size_t)CSB_NW * 2 * rs + (size_t)2 * CSB_NW * CSB_JU * 64) * sizeof(float);
    if (lds > 160 * 1024) return SS_ERR_UNSUPPORTED;
    const int gy = ss::ceil_div(C, CSB_NW * CSB_CH);
    if (grad_right && hipMemsetAsync(grad_right, 0, (size_t)B * C * plane * sizeof(float), st) != hipSuccess) return SS_ERR_LAUNCH;
    if (grad_att && gy > 1 && hipMemsetAsync(grad_att, 0, (size_t)B * nd * plane * sizeof(float), st) != hipSuccess) return SS_ERR_LAUNCH;
    const float half_w = (float)((W - 1.0) / 2.0), half_h = (float)((H - 1.0) / 2.0);
    if (lds > 64 * 1024 && ss::ensure_dynamic_lds(reinterpret_cast<const void*>(concat_sampled_bwd_kernel), (int)lds) != SS_OK) return SS_ERR_LAUNCH;
    hipLaunchKernelGGL(concat_sampled_bwd_kernel, dim3((unsigned)((long long)B * plane / 64), gy), dim3(64 * CSB_NW), lds, st, grad_out, left,
                       right, disp, att, grad_left, grad_right, grad_att, C, H, W, nd, half_w, half_h, margin, rs, gy > 1 ? 1 : 0);
    return ss::check_launch();
}

// The warped half of `att * cat(left broadcast, warp(right))` (models/SemStereo.py:241-244, 316-318) in the pre-split operand
// form of ss_conv3d_presplit_fwd: xs [B][C/8][nd][H][W][2][8] fp16 (C % 8 == 0), xexp int[3 * B]: [0, B) the block exponents,
// [B, 3B) scratch of the maxima.  att may be NULL.
extern "C" int ss_concat_sampled_presplit_fwd(const float* right, const float* disp, const float* att, void* xs, int* xexp, int B,
                                              int C, int H, int W, int nd, ss_stream_t stream) {
    SS_REQUIRE(right && disp && xs && xexp);
    SS_REQUIRE(B > 0 && C > 0 && C % 8 == 0 && H > 0 && W >= 2 && nd > 0);
    SS_REQUIRE((reinterpret_cast<uintptr_t>(xs) & 15) == 0);
    if ((long long)C * nd * H * W * 4 >= 0x7fffffffLL || (long long)B * nd > 65535) return SS_ERR_UNSUPPORTED;
    hipStream_t st = ss::as_stream(stream);
    unsigned* amax = reinterpret_cast<unsigned*>(xexp + B);
    if (hipMemsetAsync(amax, 0, (size_t)2 * B * sizeof(unsigned), st) != hipSuccess) return SS_ERR_LAUNCH;
    const long long ny = (long long)C * H * W, na = (long long)nd * H * W;
    hipLaunchKernelGGL(absmax2_kernel, dim3(128, B), dim3(256), 0, st, right, ny, att, na, amax);
    const float half_w = (float)((W - 1.0) / 2.0), half_h = (float)((H - 1.0) / 2.0);
    const dim3 grid(ss::ceil_div(W, 64), ss::ceil_div(H, 4), B * nd);
    hipLaunchKernelGGL(warp_right_presplit, grid, dim3(256), 0, st, right, disp, att, amax, reinterpret_cast<uint4*>(xs), xexp, C, H, W,
                       nd, half_w, half_h);
    return ss::check_launch();
}

extern "C" int ss_warp_correlation_fwd(const float* x, const float* y, const float* disp, float* out, int B, int C,
                                       int H, int W, int nd, ss_stream_t stream) {
    SS_REQUIRE(x && y && disp && out);
    SS_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && nd > 0);
    return launch<MODE_CORR>(x, y, disp, nullptr, out, nullptr, B, C, H, W, nd, ss::as_stream(stream));
}
