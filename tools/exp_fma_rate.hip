// What rate do v_fma_f32 chains of the broadcast-half kernel's shape reach?  Per thread: 9 x 8 "att" registers, 27 "q" values
// per pass (loaded from LDS, as the kernel does, or kept in registers), 6 outputs x 27 fused multiply-adds per pass -- no matrix
// instructions, no global traffic.  Prints FMA/clk/SIMD (the VALU peak for scalar fp32 FMAs is 16).
// build + run (GPU box): hipcc -O3 --offload-arch=gfx950 tools/exp_fma_rate.hip -o tools/_build/exp_fma_rate && tools/_build/exp_fma_rate
#include <hip/hip_runtime.h>
#include <cstdio>

template <bool LDSQ>
__global__ __launch_bounds__(512) void fma_loop(const float* __restrict__ in, float* __restrict__ out, int passes) {
    __shared__ float q[2][32][224];
    const int tid = threadIdx.x;
    float ar[9][8];
#pragma unroll
    for (int s = 0; s < 9; ++s)
#pragma unroll
        for (int k = 0; k < 8; ++k) ar[s][k] = in[(s * 8 + k) * 512 + tid];
    for (int e = tid; e < 2 * 32 * 224; e += 512) (&q[0][0][0])[e] = in[e % 4096];
    __syncthreads();
    float tot = 0.f;
    const int hp0 = (tid & 127) + (tid >> 7);
    for (int c = 0; c < passes; ++c) {
        float v[27];
#pragma unroll
        for (int t = 0; t < 27; ++t) v[t] = LDSQ ? q[c & 1][t][hp0 + (t % 9) * 3] : ar[t % 9][t % 8] + (float)c;
        float o[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 9; ++s)
#pragma unroll
            for (int kd = 0; kd < 3; ++kd)
#pragma unroll
                for (int j = 0; j < 6; ++j) o[j] = fmaf(ar[s][j + kd], v[kd * 9 + s], o[j]);
        tot += o[0] + o[1] + o[2] + o[3] + o[4] + o[5];
    }
    out[blockIdx.x * 512 + tid] = tot;
}

int main() {
    float *in, *out;
    hipMalloc(&in, 1 << 22);
    hipMalloc(&out, 1 << 22);
    hipMemset(in, 0, 1 << 22);
    const int passes = 4096, blocks = 512;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int variant = 0; variant < 2; ++variant) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (variant) hipLaunchKernelGGL(fma_loop<true>, dim3(blocks), dim3(512), 0, 0, in, out, passes);
            else hipLaunchKernelGGL(fma_loop<false>, dim3(blocks), dim3(512), 0, 0, in, out, passes);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double fma = (double)blocks * 512 * passes * 162;
            // 512 workgroups of 8 waves on 256 CUs: 2 rounds of 2 waves per SIMD
            printf("%s: %.3f ms, %.2f T FMA/s = %.2f FMA/clk/SIMD at 2.4 GHz (1024 SIMDs)\n", variant ? "q from LDS" : "q in registers", ms,
                   fma / ms / 1e9, fma / (ms * 1e-3) / 1024 / 2.4e9);
        }
    }
    return 0;
}
