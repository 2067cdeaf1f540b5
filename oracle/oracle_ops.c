/*
 * oracle_ops.c -- plain-C, single-threaded restatement of the op-library functions on the
 * SemStereo hot path.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): it is a second,
 * torch-free statement of the same algorithms, checked against the same golden fixtures as
 * oracle/ops.py, and is what `bench.py --cpu-c-port` can time as a 1-core scalar port.
 * Nothing under semstereo_amd/ links or loads it.
 *
 * Layouts: float32, contiguous, NCHW feature maps / NCDHW volumes, as in the reference.
 * Each function cites the reference lines it follows (/root/reference/models/submodule.py).
 */
#include <math.h>
#include <stddef.h>
#include <string.h>

#define IDX4(b, c, y, x, C, H, W) ((((size_t)(b) * (C) + (c)) * (H) + (y)) * (W) + (x))

/* groupwise_correlation[_norm] over one column pair (x_l of the left map, x_r of the right map):
 * models/submodule.py:190-196 / 213-221 */
static float group_corr(const float* l, const float* r, size_t plane, int cg, int normalize) {
    float nl = 1.f, nr = 1.f, acc = 0.f;
    if (normalize) {
        float sl = 0.f, sr = 0.f;
        for (int c = 0; c < cg; ++c) { sl += l[c * plane] * l[c * plane]; sr += r[c * plane] * r[c * plane]; }
        nl = sqrtf(sl) + 1e-05f; nr = sqrtf(sr) + 1e-05f;
    }
    for (int c = 0; c < cg; ++c) acc += (l[c * plane] / nl) * (r[c * plane] / nr);
    return acc / (float)cg;
}

/* build_gwc_volume / build_gwc_volume_norm: models/submodule.py:198-211 / 224-238 */
void orc_gwc_volume(const float* ref, const float* tgt, float* out, int B, int C, int H, int W, int m, int G,
                    int normalize) {
    const int cg = C / G, D = 2 * m;
    const size_t plane = (size_t)H * W;
    memset(out, 0, sizeof(float) * (size_t)B * G * D * plane);
    for (int b = 0; b < B; ++b)
        for (int g = 0; g < G; ++g)
            for (int d = 0; d < D; ++d) {
                const int s = d - m;                       /* signed disparity */
                for (int y = 0; y < H; ++y)
                    for (int x = 0; x < W; ++x) {
                        const int xr = x - s;
                        if (xr < 0 || xr >= W) continue;
                        out[((((size_t)b * G + g) * D + d) * H + y) * W + x] =
                            group_corr(ref + IDX4(b, g * cg, y, x, C, H, W), tgt + IDX4(b, g * cg, y, xr, C, H, W),
                                       plane, cg, normalize);
                    }
            }
}

/* build_concat_volume: models/submodule.py:173-187 */
void orc_concat_volume(const float* ref, const float* tgt, float* out, int B, int C, int H, int W, int m) {
    const int D = 2 * m;
    memset(out, 0, sizeof(float) * (size_t)B * 2 * C * D * H * W);
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c)
            for (int d = 0; d < D; ++d)
                for (int y = 0; y < H; ++y)
                    for (int x = 0; x < W; ++x) {
                        const int xr = x - (d - m);
                        if (xr < 0 || xr >= W) continue;
                        out[((((size_t)b * 2 * C + c) * D + d) * H + y) * W + x] = ref[IDX4(b, c, y, x, C, H, W)];
                        out[((((size_t)b * 2 * C + C + c) * D + d) * H + y) * W + x] = tgt[IDX4(b, c, y, xr, C, H, W)];
                    }
}

/* disparity_regression: models/submodule.py:164-170 ; disparity_variance (disp != NULL): :257-263 */
void orc_disparity_regression(const float* prob, const float* disp, float* out, int B, int m, int H, int W) {
    const int D = 2 * m;
    const size_t plane = (size_t)H * W;
    for (int b = 0; b < B; ++b)
        for (size_t p = 0; p < plane; ++p) {
            float acc = 0.f;
            for (int d = 0; d < D; ++d) {
                float w = (float)(d - m);
                if (disp) { w -= disp[b * plane + p]; w = w * w; }
                acc += prob[((size_t)b * D + d) * plane + p] * w;
            }
            out[b * plane + p] = acc;
        }
}

/* regression_topk: models/submodule.py:434-442 (ties: lower index first) */
void orc_regression_topk(const float* cost, const float* samples, float* out, int B, int nd, int H, int W, int k) {
    const size_t plane = (size_t)H * W;
    for (int b = 0; b < B; ++b)
        for (size_t p = 0; p < plane; ++p) {
            const float* c = cost + (size_t)b * nd * plane + p;
            const float* s = samples + (size_t)b * nd * plane + p;
            float sel_v[64]; int sel_i[64];
            float pv = INFINITY; int pi = -1;
            for (int t = 0; t < k; ++t) {
                float bv = -INFINITY; int bi = -1;
                for (int j = 0; j < nd; ++j) {
                    const float v = c[j * plane];
                    const int after = (v < pv) || (v == pv && j > pi);
                    const int better = (bi < 0) || (v > bv);
                    if (after && better) { bv = v; bi = j; }
                }
                sel_v[t] = bv; sel_i[t] = bi; pv = bv; pi = bi;
            }
            float sum = 0.f, acc = 0.f;
            for (int t = 0; t < k; ++t) sum += expf(sel_v[t] - sel_v[0]);
            for (int t = 0; t < k; ++t) acc += s[sel_i[t] * plane] * (expf(sel_v[t] - sel_v[0]) / sum);
            out[b * plane + p] = acc;
        }
}

/* SpatialTransformer_grid: models/submodule.py:265-288 (bilinear, zeros padding, align_corners=True,
 * coordinates through the normalise / un-normalise round trip).  x_warped may be NULL. */
void orc_warp_sampled(const float* x, const float* y, const float* disp, float* y_warped, float* x_warped, int B,
                      int C, int H, int W, int nd) {
    const float half_w = (float)((W - 1.0) / 2.0), half_h = (float)((H - 1.0) / 2.0);
    const size_t plane = (size_t)H * W;
    for (int b = 0; b < B; ++b)
        for (int j = 0; j < nd; ++j)
            for (int h = 0; h < H; ++h)
                for (int w = 0; w < W; ++w) {
                    const float dv = disp[(((size_t)b * nd + j) * H + h) * W + w];
                    const float gx = ((float)w - dv) / half_w - 1.0f, gy = (float)h / half_h - 1.0f;
                    const float ix = (gx + 1.0f) * half_w, iy = (gy + 1.0f) * half_h;
                    const float xw = floorf(ix), yn = floorf(iy);
                    const float fw = ix - xw, fe = 1.0f - fw, fn = iy - yn, fs = 1.0f - fn;
                    const int in_w = xw > -1.0f && xw < (float)W, in_e = xw + 1.0f > -1.0f && xw + 1.0f < (float)W;
                    const int in_n = yn > -1.0f && yn < (float)H, in_s = yn + 1.0f > -1.0f && yn + 1.0f < (float)H;
                    for (int c = 0; c < C; ++c) {
                        const float* yp = y + ((size_t)b * C + c) * plane;
                        const float a = (in_n && in_w) ? yp[(int)yn * W + (int)xw] : 0.f;
                        const float bb = (in_n && in_e) ? yp[(int)yn * W + (int)xw + 1] : 0.f;
                        const float cc = (in_s && in_w) ? yp[((int)yn + 1) * W + (int)xw] : 0.f;
                        const float dd = (in_s && in_e) ? yp[((int)yn + 1) * W + (int)xw + 1] : 0.f;
                        const size_t o = ((((size_t)b * C + c) * nd + j) * H + h) * W + w;
                        y_warped[o] = a * (fs * fe) + bb * (fs * fw) + cc * (fn * fe) + dd * (fn * fw);
                        if (x_warped) x_warped[o] = x[((size_t)b * C + c) * plane + (size_t)h * W + w];
                    }
                }
}
