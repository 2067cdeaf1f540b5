// What does a byte cost?  One streaming pattern for ~4 s over a 1 GiB window (far beyond L2 and the 256 MiB Infinity Cache) while
// tools/power_of.sh samples the socket power beside it: read-only (16 B per lane), write-only (4 / 16 B per lane, default / nontemporal),
// copy.  usage (GPU box): tools/_build/exp_mem_energy <read|read4|write4|write16|write16nt|copy|idle>      prints the achieved rate
// build: hipcc -O3 --offload-arch=gfx950 tools/exp_mem_energy.hip -o tools/_build/exp_mem_energy
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_read(const f32x4* __restrict__ in, float* __restrict__ sink, size_t n4) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) s += __builtin_nontemporal_load(in + i);
    if (s.x + s.y + s.z + s.w == 12345.678f) *sink = s.x;
}
__global__ __launch_bounds__(256) void k_read4(const float* __restrict__ in, float* __restrict__ sink, size_t n) {
    float s = 0.f;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += __builtin_nontemporal_load(in + i);
    if (s == 12345.678f) *sink = s;
}
__global__ __launch_bounds__(256) void k_write4(float* __restrict__ out, size_t n) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = (float)i;
}
template <bool NT>
__global__ __launch_bounds__(256) void k_write16(f32x4* __restrict__ out, size_t n4) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const f32x4 v = {(float)i, 1.f, 2.f, 3.f};
        if (NT) __builtin_nontemporal_store(v, out + i);
        else out[i] = v;
    }
}
__global__ __launch_bounds__(256) void k_copy(const f32x4* __restrict__ in, f32x4* __restrict__ out, size_t n4) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) out[i] = in[i];
}
__global__ void k_spin(float* sink, int n) {       // every CU busy with scalar / VALU work, no memory traffic: the "static + clocks" floor
    float v = threadIdx.x;
    for (int i = 0; i < n; ++i) v = __builtin_fmaf(v, 1.0000001f, 1e-9f);
    if (v == 12345.678f) *sink = v;
}

int main(int argc, char** argv) {
    const char* mode = argc > 1 ? argv[1] : "read";
    const size_t bytes = (size_t)1 << 30, n = bytes / 4, n4 = bytes / 16;
    float *a, *b, *sink;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&sink, 4);
    hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
    const int grid = 256 * 8;
    auto t0 = std::chrono::steady_clock::now();
    long launches = 0;
    double per_launch = strcmp(mode, "copy") == 0 ? 2.0 * bytes : (double)bytes;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 4.0) {
        for (int r = 0; r < 10; ++r) {
            if (!strcmp(mode, "read4")) hipLaunchKernelGGL(k_read4, dim3(grid), dim3(256), 0, 0, (const float*)a, sink, n);
            else if (!strcmp(mode, "read")) hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, (const f32x4*)a, sink, n4);
            else if (!strcmp(mode, "write4")) hipLaunchKernelGGL(k_write4, dim3(grid), dim3(256), 0, 0, b, n);
            else if (!strcmp(mode, "write16")) hipLaunchKernelGGL(k_write16<false>, dim3(grid), dim3(256), 0, 0, (f32x4*)b, n4);
            else if (!strcmp(mode, "write16nt")) hipLaunchKernelGGL(k_write16<true>, dim3(grid), dim3(256), 0, 0, (f32x4*)b, n4);
            else if (!strcmp(mode, "copy")) hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, (const f32x4*)a, (f32x4*)b, n4);
            else { hipLaunchKernelGGL(k_spin, dim3(grid), dim3(256), 0, 0, sink, 200000); per_launch = 0; }
            ++launches;
        }
        hipDeviceSynchronize();
    }
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("%s: %.2f TB/s (%ld launches in %.2f s)\n", mode, per_launch * launches / s / 1e12, launches, s);
    return 0;
}
