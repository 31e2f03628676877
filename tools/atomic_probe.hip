// How long does a grid-wide sum through memory-side 64-bit integer atomics take?  G workgroups each add 200 (or 228) int64
// partials to the same 200 addresses (agent scope, no return value) -- against the same grid writing G separate records.
//   hipcc --offload-arch=gfx950 -O3 tools/atomic_probe.hip -o tools/atomic_probe && tools/atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ void k_atomic(long long* acc, int n) {
    const int t = threadIdx.x;
    if (t < n) __hip_atomic_fetch_add(acc + t, (long long)(blockIdx.x * 131 + t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void k_records(float* rec, int n) {
    const int t = threadIdx.x;
    if (t < n) __hip_atomic_store(rec + (size_t)blockIdx.x * 256 + t, (float)(blockIdx.x + t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void k_read_records(const float* rec, int G, int n, float* out) {
    const int t = threadIdx.x & 255, part = threadIdx.x >> 8;
    double a = 0;
    const int Gq = (G + 3) / 4, g0 = part * Gq, g1 = min(G, g0 + Gq);
    if (t < n)
        for (int g = g0; g < g1; ++g) a += rec[(size_t)g * 256 + t];
    __shared__ double s[4][256];
    s[part][t] = a;
    __syncthreads();
    if (threadIdx.x < n) out[blockIdx.x * 256 + threadIdx.x] = (float)(s[0][t] + s[1][t] + s[2][t] + s[3][t]);
}
__global__ void k_read_acc(const long long* acc, int n, float* out) {
    if ((int)threadIdx.x < n) out[blockIdx.x * 256 + threadIdx.x] = (float)acc[threadIdx.x] * 1e-9f;
}
__global__ void k_empty() {}

template <class F>
float time_us(F f, int reps = 200) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (int i = 0; i < 10; ++i) f();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / reps;
}

int main() {
    long long* acc;
    float *rec, *out;
    hipMalloc(&acc, 4096);
    hipMalloc(&rec, 1024 * 256 * 4);
    hipMalloc(&out, 1024 * 256 * 4);
    hipMemset(acc, 0, 4096);
    const float base = time_us([&] { hipLaunchKernelGGL(k_empty, dim3(124), dim3(256), 0, 0); });
    printf("empty launch (back to back)            %6.2f us\n", base);
    for (int G : {62, 124, 248}) {
        for (int n : {200, 228}) {
            const float ta = time_us([&] { hipLaunchKernelGGL(k_atomic, dim3(G), dim3(256), 0, 0, acc, n); });
            const float tr = time_us([&] { hipLaunchKernelGGL(k_records, dim3(G), dim3(256), 0, 0, rec, n); });
            const float rr = time_us([&] { hipLaunchKernelGGL(k_read_records, dim3(G), dim3(1024), 0, 0, rec, G, n, out); });
            const float ra = time_us([&] { hipLaunchKernelGGL(k_read_acc, dim3(G), dim3(256), 0, 0, acc, n, out); });
            printf("G=%3d n=%3d: atomics %6.2f us | records %6.2f us || consumer sums records %6.2f us | reads accumulator %6.2f us\n", G, n, ta, tr, rr, ra);
        }
    }
    return 0;
}
