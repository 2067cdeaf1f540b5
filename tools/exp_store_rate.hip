// How many clocks does a CU's memory pipeline spend per wave-wide store instruction, by width?  512 workgroups of 8 waves (one
// per CU and round), every wave issues `n` stores of 4 / 8 / 16 bytes per lane, whole 128-byte lines, into a window of `win_mb`
// MB (4: stays in L2; 1024: goes to HBM).  Prints clocks per store instruction and CU (2.4 GHz nominal) and the byte rate.
// build + run (GPU box): hipcc -O3 --offload-arch=gfx950 tools/exp_store_rate.hip -o tools/_build/exp_store_rate && tools/_build/exp_store_rate
#include <hip/hip_runtime.h>
#include <cstdio>

template <int WIDTH>      // dwords per lane
__global__ __launch_bounds__(512) void store_loop(float* __restrict__ out, int n, unsigned win_mask) {
    const unsigned wave = (blockIdx.x * 8 + (threadIdx.x >> 6));
    const unsigned lane = threadIdx.x & 63;
    float v = (float)threadIdx.x;
    for (int i = 0; i < n; ++i) {
        // one instruction = 64 lanes x WIDTH dwords, consecutive: WIDTH * 2 whole lines
        const unsigned off = ((wave * 977u + (unsigned)i * 131u) * (64u * WIDTH) * 4u) & win_mask;
        float* p = reinterpret_cast<float*>(reinterpret_cast<char*>(out) + off) + lane * WIDTH;
        if (WIDTH == 1) *p = v;
        if (WIDTH == 2) *reinterpret_cast<float2*>(p) = make_float2(v, v);
        if (WIDTH == 4) *reinterpret_cast<float4*>(p) = make_float4(v, v, v, v);
        v += 1.f;
    }
}

int main() {
    float* out;
    hipMalloc(&out, (size_t)1 << 30);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int n = 2048, blocks = 512;
    for (int win_mb : {4, 1024}) {
        const unsigned mask = (unsigned)((size_t)win_mb << 20) - 1u;
        for (int width : {1, 2, 4}) {
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                if (width == 1) hipLaunchKernelGGL(store_loop<1>, dim3(blocks), dim3(512), 0, 0, out, n, mask & ~255u);
                if (width == 2) hipLaunchKernelGGL(store_loop<2>, dim3(blocks), dim3(512), 0, 0, out, n, mask & ~511u);
                if (width == 4) hipLaunchKernelGGL(store_loop<4>, dim3(blocks), dim3(512), 0, 0, out, n, mask & ~1023u);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                best = ms < best ? ms : best;
            }
            const double instr_per_cu = (double)blocks * 8 * n / 256;
            printf("window %4d MB, %2d B/lane: %.3f ms, %.1f clocks per store instruction and CU, %.2f TB/s\n", win_mb, 4 * width, best,
                   best * 1e-3 * 2.4e9 / instr_per_cu, (double)blocks * 8 * n * 256.0 * width / (best * 1e-3) / 1e12);
        }
    }
    return 0;
}
