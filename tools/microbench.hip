// Calibration microbenchmarks for the MI355X box: kernel cadence, shader clock, dependent-load latency.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 1024) p[0] = 1; }

typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void clock_kernel(unsigned long long* out, int iters) {
    f32x4 acc = {0, 0, 0, 0};
    float a = threadIdx.x, b = 1.0f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
    if (acc[0] == 12345.f) out[2] = 1;
}

__global__ void chase_kernel(const int* next, int steps, int* out, unsigned long long* cyc) {
    int i = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int s = 0; s < steps; ++s) i = next[i];
    unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    out[0] = i; cyc[0] = t1 - t0;
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int* d; CK(hipMalloc(&d, 4));
    float ms;
    for (int grid : {1, 256, 2048}) {
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(256), 0, st, d);
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < 1000; ++i) hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(256), 0, st, d);
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("eager empty kernel grid=%d: %.2f us per launch\n", grid, ms);
    }
    // graph of 32 empty kernels
    {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int i = 0; i < 32; ++i) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, st, d);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < 200; ++i) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("graph of 32 empty kernels: %.2f us per replay = %.2f us per node\n", ms * 1000 / 200, ms * 1000 / 200 / 32);
    }
    unsigned long long* dc; CK(hipMalloc(&dc, 64));
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(clock_kernel, dim3(1024), dim3(256), 0, st, dc, 20000);
        CK(hipStreamSynchronize(st));
        unsigned long long h[2]; CK(hipMemcpy(h, dc, 16, hipMemcpyDeviceToHost));
        printf("clock: %llu shader cycles in %llu x10ns -> %.3f GHz ; %.1f cycles per mfma_f32_16x16x4\n", h[0], h[1],
               (double)h[0] / ((double)h[1] * 10.0), (double)h[0] / 20000.0);
    }
    // pointer chase, different footprints
    for (size_t n : {(size_t)1 << 10, (size_t)1 << 18, (size_t)1 << 24, (size_t)1 << 27}) {
        std::vector<int> h(n);
        size_t stride = 4099;  // elements; odd -> permutation
        for (size_t i = 0; i < n; ++i) h[i] = (int)((i + stride * 16) % n);
        int* dn; CK(hipMalloc(&dn, n * 4)); CK(hipMemcpy(dn, h.data(), n * 4, hipMemcpyHostToDevice));
        int steps = 2000;
        hipLaunchKernelGGL(chase_kernel, dim3(1), dim3(1), 0, st, dn, steps, d, dc);
        hipLaunchKernelGGL(chase_kernel, dim3(1), dim3(1), 0, st, dn, steps, d, dc);
        CK(hipStreamSynchronize(st));
        unsigned long long c; CK(hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost));
        printf("pointer chase footprint %zu KB: %.1f ns per dependent load\n", n * 4 / 1024, (double)c * 10.0 / steps);
        CK(hipFree(dn));
    }
    return 0;
}
