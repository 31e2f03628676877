// Finding 30: how many wait states does gfx950 need between an 8-pass MFMA (v_mfma_f32_16x16x4_f32) writing its D registers
// and a DS instruction reading them?  hipcc's hazard recognizer pads 10 states when the store sits in another basic block
// (csrc/dgcn_ops.hip brgcn_bwd_target_tile_kernel: s_and_saveexec + s_cbranch_execz + s_nop 7); the CDNA ISA table says 12.
// Every variant overwrites the accumulator (holding a POISON product) with a second MFMA, waits N states, stores the tile
// to LDS with ds_write_b128 and compares with the fully waited result.
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_hazard.hip -o tools/mfma_hazard && tools/mfma_hazard
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define VARIANT(NAME, PAD)                                                                                         \
    __global__ void NAME(float* out, int iters) {                                                                  \
        __shared__ __attribute__((aligned(16))) float lds[64 * 4];                                                 \
        const int lane = threadIdx.x;                                                                              \
        const unsigned addr = (unsigned)(lane * 16);                                                               \
        for (int it = 0; it < iters; ++it) {                                                                       \
            const float a1 = 1000.0f + lane, a2 = (float)(lane % 7) + 0.25f * it, b = 1.0f;                        \
            f32x4 acc;                                                                                             \
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %2, %3, 0\n\ts_nop 15\n\ts_nop 15\n\t"                        \
                         "v_mfma_f32_16x16x4_f32 %0, %4, %3, 0\n\t" PAD "ds_write_b128 %1, %0\n\t"                 \
                         "s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15"                                            \
                         : "=&v"(acc)                                                                              \
                         : "v"(addr), "v"(a1), "v"(b), "v"(a2)                                                     \
                         : "memory");                                                                              \
            for (int j = 0; j < 4; ++j) out[((size_t)it * 64 + lane) * 4 + j] = lds[lane * 4 + j];                 \
        }                                                                                                          \
    }

VARIANT(k0, "")
VARIANT(k2, "s_nop 1\n\t")
VARIANT(k4, "s_nop 3\n\t")
VARIANT(k6, "s_nop 5\n\t")
VARIANT(k8, "s_nop 7\n\t")
VARIANT(k9, "s_nop 8\n\t")
VARIANT(k10, "s_nop 9\n\t")
VARIANT(k11, "s_nop 10\n\t")
VARIANT(k12, "s_nop 11\n\t")
VARIANT(k14, "s_nop 13\n\t")
VARIANT(kref, "s_nop 15\n\ts_nop 15\n\t")

int main() {
    const int iters = 2000;
    const size_t n = (size_t)iters * 64 * 4;
    float *d, *ref = (float*)malloc(n * 4), *got = (float*)malloc(n * 4);
    hipMalloc(&d, n * 4);
    hipLaunchKernelGGL(kref, dim3(1), dim3(64), 0, 0, d, iters);
    hipMemcpy(ref, d, n * 4, hipMemcpyDeviceToHost);
    struct { const char* name; void (*k)(float*, int); int states; } v[] = {
        {"0", k0, 0}, {"2", k2, 2}, {"4", k4, 4}, {"6", k6, 6}, {"8", k8, 8}, {"9", k9, 9}, {"10", k10, 10},
        {"11", k11, 11}, {"12", k12, 12}, {"14", k14, 14}};
    for (auto& e : v) {
        size_t bad = 0;
        for (int rep = 0; rep < 5; ++rep) {
            hipMemset(d, 0, n * 4);
            hipLaunchKernelGGL(e.k, dim3(1), dim3(64), 0, 0, d, iters);
            hipMemcpy(got, d, n * 4, hipMemcpyDeviceToHost);
            for (size_t i = 0; i < n; ++i) bad += got[i] != ref[i];
        }
        printf("wait states between the MFMA and ds_write_b128 of its result: %2d -> %zu wrong words of %zu\n", e.states, bad, 5 * n);
    }
    return 0;
}
