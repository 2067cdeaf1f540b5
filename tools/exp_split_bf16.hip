// Experiment (not part of the product): how accurate is an fp32 GEMM emulated on the bf16 matrix core
// by splitting every fp32 operand into 2 or 3 bf16 terms (x = hi + mid + lo) and summing the
// 3 or 6 dominant cross products with v_mfma_f32_32x32x16_bf16, compared with the exact-fp32
// v_mfma_f32_32x32x2_f32 and an fp64 host reference?   hipcc --offload-arch=gfx950 -O2 -o exp exp_split_bf16.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) short;

__device__ __forceinline__ unsigned short f2bf(float x) {   // round-to-nearest-even, finite inputs
    unsigned u = __float_as_uint(x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf2f(unsigned short b) { return __uint_as_float(((unsigned)b) << 16); }

// one wave: C[32x32] = A[32xK] * B[Kx32];  A row-major [32][K], B row-major [K][32]
__global__ void gemm_variants(const float* A, const float* B, float* C32, float* C3, float* C6, int K) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    f32x16 acc = {0};
    for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * K + k + h], B[(k + h) * 32 + r], acc, 0, 0, 0);
    f32x16 a3 = {0}, a6 = {0};
    for (int k0 = 0; k0 < K; k0 += 16) {
        bf16x8 ah, am, al, bh, bm, bl;
        for (int j = 0; j < 8; ++j) {
            const float a = A[r * K + k0 + 8 * h + j], b = B[(k0 + 8 * h + j) * 32 + r];
            unsigned short x;
            float rem;
            x = f2bf(a); ah[j] = (short)x; rem = a - bf2f(x);
            x = f2bf(rem); am[j] = (short)x; rem = rem - bf2f(x);
            x = f2bf(rem); al[j] = (short)x;
            x = f2bf(b); bh[j] = (short)x; rem = b - bf2f(x);
            x = f2bf(rem); bm[j] = (short)x; rem = rem - bf2f(x);
            x = f2bf(rem); bl[j] = (short)x;
        }
        // smallest terms first so they are not absorbed by a large accumulator
        a6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, a6, 0, 0, 0);
        a6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, a6, 0, 0, 0);
        a6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, a6, 0, 0, 0);
        a6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, a6, 0, 0, 0);
        a6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, a6, 0, 0, 0);
        a6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, a6, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, a3, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, a3, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, a3, 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
        C32[row * 32 + r] = acc[i]; C3[row * 32 + r] = a3[i]; C6[row * 32 + r] = a6[i];
    }
}

int main() {
    for (int K : {32, 864, 1728, 3456}) {
        std::vector<float> A(32 * K), B(K * 32);
        srand(1234 + K);
        for (auto& v : A) v = (float)rand() / RAND_MAX * 2.f - 1.f;
        for (auto& v : B) v = (float)rand() / RAND_MAX * 2.f - 1.f;
        float *dA, *dB, *d32, *d3, *d6;
        hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4);
        hipMalloc(&d32, 4096); hipMalloc(&d3, 4096); hipMalloc(&d6, 4096);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(gemm_variants, dim3(1), dim3(64), 0, 0, dA, dB, d32, d3, d6, K);
        std::vector<float> c32(1024), c3(1024), c6(1024);
        hipMemcpy(c32.data(), d32, 4096, hipMemcpyDeviceToHost);
        hipMemcpy(c3.data(), d3, 4096, hipMemcpyDeviceToHost);
        hipMemcpy(c6.data(), d6, 4096, hipMemcpyDeviceToHost);
        double e32 = 0, e3 = 0, e6 = 0, scale = 0;
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                double ref = 0, mag = 0;
                for (int k = 0; k < K; ++k) { ref += (double)A[i * K + k] * B[k * 32 + j]; mag += fabs((double)A[i * K + k] * B[k * 32 + j]); }
                e32 = fmax(e32, fabs(c32[i * 32 + j] - ref) / mag);
                e3 = fmax(e3, fabs(c3[i * 32 + j] - ref) / mag);
                e6 = fmax(e6, fabs(c6[i * 32 + j] - ref) / mag);
                scale = fmax(scale, mag);
            }
        printf("K=%5d  max |err| / sum|a*b|:  fp32-mfma %.3e   bf16x3 %.3e   bf16x6 %.3e\n", K, e32, e3, e6);
        hipFree(dA); hipFree(dB); hipFree(d32); hipFree(d3); hipFree(d6);
    }
    return 0;
}
