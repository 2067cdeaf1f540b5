// How fast is ds_add_f32 (no return) against a plain LDS store and a global fp32 atomic, on distinct and on shared addresses?
// (r06: the training step's scatter kernels sum through LDS row buffers.)   hipcc -O3 --offload-arch=gfx950 -o exp_lds_atomic exp_lds_atomic.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, int stride) {
    __shared__ float buf[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) buf[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x;
    float v = 1.0f + lane;
    for (int it = 0; it < iters; ++it) {
        const int idx = ((lane * stride) + it * 7) & 4095;
        if (MODE == 0) buf[idx] = v;                                                          // plain store
        else if (MODE == 1) __hip_atomic_fetch_add((float __attribute__((address_space(3)))*)(&buf[idx]), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (MODE == 2) buf[idx] += v;                                                    // read-modify-write, not atomic
        else if (MODE == 3) unsafeAtomicAdd(out + (blockIdx.x * 4096 + idx), v);              // global atomic, distinct addresses
        else if (MODE == 4) __hip_atomic_fetch_add((unsigned __attribute__((address_space(3)))*)(&buf[idx]), (unsigned)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else {                                                                                // tagged read-add-write: the wave owns its buffer (each wave a quarter here), lanes that collide take turns
            volatile int __attribute__((address_space(3)))* tag = (volatile int __attribute__((address_space(3)))*)(buf + 2048);
            const int k = (idx & 511) + 512 * (threadIdx.x >> 6);
            bool pending = true;                                                              // (ballot, not `while (pending)`: the optimiser sinks the add below a per-lane loop)
            while (__builtin_amdgcn_ballot_w64(pending) != 0) {
                if (pending) {
                    tag[k] = lane & 63;
                    if (tag[k] == (lane & 63)) {
                        volatile float __attribute__((address_space(3)))* q = (volatile float __attribute__((address_space(3)))*)(buf + k);
                        *q = *q + v;
                        pending = false;
                    }
                }
            }
        }
        v += 1.0f;
    }
    __syncthreads();
    if (MODE != 3) out[blockIdx.x * 256 + threadIdx.x] = buf[threadIdx.x];
}
template <int MODE>
void run(const char* name, float* out, int stride) {
    const int iters = 4096, blocks = 2048;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, stride);
    hipEventRecord(a); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, stride); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double ops = (double)blocks * 256 * iters;
    printf("%-34s stride %2d: %7.2f ms, %7.1f G lane-ops/s, %.2f lane-ops per CU-clock (256 CUs, 2.1 GHz)\n", name, stride, ms, ops / ms / 1e6, ops / (ms * 1e-3) / 256 / 2.1e9);
}
int main() {
    float* out; hipMalloc(&out, 2048 * 4096 * 4); hipMemset(out, 0, 2048 * 4096 * 4);
    for (int stride : {1, 2, 32}) {
        run<0>("LDS store", out, stride); run<1>("LDS ds_add_f32", out, stride); run<2>("LDS read-add-write", out, stride);
        run<4>("LDS ds_add_u32", out, stride); run<5>("LDS tagged read-add-write", out, stride);
    }
    run<3>("global atomic add (distinct)", out, 1);
    return 0;
}
