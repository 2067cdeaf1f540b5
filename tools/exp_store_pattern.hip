// How fast can a workgroup write the conv epilogue's 64 KB tile?  A: 64 dword stores per lane in the MFMA D layout
// (32 consecutive lanes = 128 B of one (channel, row); the other half-wave 4 channels away); B: the same bytes as 16
// dwordx4 stores per lane (8 lanes = 128 B of one (channel, row)).  Output [32 ch][24 d][256 h][256 w] fp32 like concat_stem.
// build: hipcc -O3 --offload-arch=gfx950 tools/exp_store_pattern.hip -o tools/_build/exp_store ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int C = 32, D = 24, H = 256, W = 256, TD = 2, TH = 8, NT = 4;
__global__ __launch_bounds__(256) void store_dword(float* out, int tiles_w, int tiles_h, int reps) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    int t = blockIdx.x; const int tw = t % tiles_w; t /= tiles_w; const int th = t % tiles_h; t /= tiles_h;
    const int ow = tw * 32 + l31, od = t * TD + (wave * NT) / TH, oh0 = th * TH + (wave * NT) % TH;
    const size_t plane = (size_t)H * W, chan = (size_t)D * plane;
    for (int rep = 0; rep < reps; ++rep)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = (r & 3) + 8 * (r >> 2) + 4 * half;
#pragma unroll
        for (int i = 0; i < NT; ++i) out[co * chan + od * plane + (size_t)(oh0 + i) * W + ow] = (float)(r + i + rep);
    }
}
__global__ __launch_bounds__(256) void store_x4(float* out, int tiles_w, int tiles_h, int reps) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int t = blockIdx.x; const int tw = t % tiles_w; t /= tiles_w; const int th = t % tiles_h; t /= tiles_h;
    const int od = t * TD + (wave * NT) / TH, oh0 = th * TH + (wave * NT) % TH;
    const size_t plane = (size_t)H * W, chan = (size_t)D * plane;
    // 128 (channel, row) pairs per wave, 8 per instruction: pair = k * 8 + lane / 8, 4 columns per lane
    for (int rep = 0; rep < reps; ++rep)
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int pair = k * 8 + (lane >> 3), co = pair / NT, i = pair % NT;
        float4 v = make_float4((float)k, (float)rep, 0.f, 1.f);
        *reinterpret_cast<float4*>(out + co * chan + od * plane + (size_t)(oh0 + i) * W + tw * 32 + (lane & 7) * 4) = v;
    }
}
int main() {
    float* out; hipMalloc(&out, sizeof(float) * C * D * H * W);
    const int tiles_w = W / 32, tiles_h = H / TH, tiles_d = D / TD, n = tiles_w * tiles_h * tiles_d;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {256, 512, n}) for (int which = 0; which < 2; ++which) {
        for (int it = 0; it < 200; ++it) { if (which) hipLaunchKernelGGL(store_x4, dim3(n), dim3(256), 0, 0, out, tiles_w, tiles_h, 1); else hipLaunchKernelGGL(store_dword, dim3(n), dim3(256), 0, 0, out, tiles_w, tiles_h, 1); }
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int it = 0; it < 50; ++it) { if (which) hipLaunchKernelGGL(store_x4, dim3(grid), dim3(256), 0, 0, out, tiles_w, tiles_h, 1); else hipLaunchKernelGGL(store_dword, dim3(grid), dim3(256), 0, 0, out, tiles_w, tiles_h, 1); }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%s grid %5d: %7.2f us/launch  %.0f GB/s\n", which ? "dwordx4" : "dword  ", grid, ms * 20, grid * 65536.0 / (ms / 50 * 1e-3) / 1e9);
    }
    return 0;
}
