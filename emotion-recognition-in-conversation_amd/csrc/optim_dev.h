// Device-side pieces of the optimizer shared by csrc/optim.hip (the optimizer launch) and csrc/wgrad_bf16.hip (the update
// applied by the weight-gradient launch's last arrivers): the bf16 shadow table and the Adam / AdamW element update.
#pragma once
#include "erc_common.h"

namespace {

// bf16 shadow copies of parameter ranges (operands of the bf16 matrix-core products), written by the optimizer
// kernel itself so that no extra launch keeps them in sync.  Element i of [src_off, src_off + n_el) of the flat
// buffer, idx = i - src_off, is split into digits d0 = idx % n0, d1 = (idx / n0) % n1, d2 = idx / (n0 n1); the digits
// give the element's coordinates in the LOGICAL B operand of its product, n = sum d_i sn_i (output column) and
// k = sum d_i sk_i (reduction index), and the layout places it:
//   mode 0: shadow[dst_off + n * ld + k]                                   (row-major [n][k]: identity copies)
//   mode 1: the fragment order of v_mfma_f32_16x16x32_bf16's B operand, ld = number of 32-deep K blocks: the 512
//           elements of (column tile n / 16, K block k / 32) are contiguous, lane (r = n % 16, g = (k % 32) / 8) at
//           [(g * 16 + r) * 8, +8) -- a wavefront's fragment load is ONE contiguous 1 KB run (8 full cache lines;
//           the row-major layout cost 16 half-used lines per load and the L1 miss path, not bytes, set the time).
//   terms > 1 (the split compute modes f32x2 / f32x3, csrc/split_dev.h): the range is stored as `terms` bf16 PLANES of the same
//   layout, plane t at + t * plane_stride elements, holding term t of the parameter's bf16 expansion
//   p = t0 + t1 (+ t2),  t0 = bf16(p), t1 = bf16(p - t0), t2 = bf16(p - t0 - t1)   (round to nearest even; the remainders
//   are exact in fp32), so that products against all planes are fp32-class on the bf16 matrix cores.
struct ShadowDesc {
    int64_t src_off, n_el, dst_off;
    int32_t n0, n1, sn0, sn1, sn2, sk0, sk1, sk2, ld, mode;
    int64_t plane_stride;
    int32_t terms, pad_;
};
constexpr int SHADOW_MAX = 8;
struct ShadowTab {
    int32_t n, flags;
    ShadowDesc d[SHADOW_MAX];
};

__device__ __forceinline__ int64_t shadow_dst(const ShadowDesc& d, int32_t n, int32_t k) {
    if (d.mode == 0) return d.dst_off + (int64_t)n * d.ld + k;
    return d.dst_off + ((((int64_t)(n >> 4) * d.ld + (k >> 5)) * 64 + ((k & 31) >> 3) * 16 + (n & 15)) << 3) + (k & 7);
}
__device__ __forceinline__ unsigned short f2bf_u16(float f) {
    const __bf16 h = (__bf16)f;
    return __builtin_bit_cast(unsigned short, h);
}
// term t of the bf16 expansion of x (see ShadowDesc.terms); `r` carries the remainder from term to term
__device__ __forceinline__ unsigned short bf_term_next(float& r) {
    const __bf16 h = (__bf16)r;
    r -= (float)h;
    return __builtin_bit_cast(unsigned short, h);
}

// Four consecutive elements (i0 % 4 == 0) at once: with src_off, n0 multiples of 4 (checked by the host; the flat buffer
// aligns every group to 64 floats) they share d1 and d2, so one index decomposition serves the quad, and when they run
// along k (sn0 = 0, sk0 = 1, k % 4 == 0) the four bf16 are one 8-byte store in either layout.
__device__ __forceinline__ void shadow_store4_desc(unsigned short* __restrict__ shadow, const ShadowDesc& d, const bool k_quads,
                                                   int64_t i0, float4 pn) {
    const int64_t idx = i0 - d.src_off;
    if (idx >= 0 && idx < d.n_el) {
        const int32_t x = (int32_t)idx;
        const int32_t q = x / d.n0, d0 = x - q * d.n0, d2 = q / d.n1, d1 = q - d2 * d.n1;
        const int32_t n = d0 * d.sn0 + d1 * d.sn1 + d2 * d.sn2, k = d0 * d.sk0 + d1 * d.sk1 + d2 * d.sk2;
        float r0 = pn.x, r1 = pn.y, r2 = pn.z, r3 = pn.w;
        unsigned short* sp = shadow;
#pragma unroll 1
        for (int t = 0; t < d.terms; ++t, sp += d.plane_stride) {
            const unsigned short h0 = bf_term_next(r0), h1 = bf_term_next(r1), h2 = bf_term_next(r2), h3 = bf_term_next(r3);
            if (d.sn0 == 0 && d.sk0 == 1 && k_quads) {   // k_quads (host flag): k % 4 == 0 and 8-byte aligned destinations
                *reinterpret_cast<uint2*>(sp + shadow_dst(d, n, k)) =
                    make_uint2((uint32_t)h0 | ((uint32_t)h1 << 16), (uint32_t)h2 | ((uint32_t)h3 << 16));
            } else {
                sp[shadow_dst(d, n, k)] = h0;
                sp[shadow_dst(d, n + d.sn0, k + d.sk0)] = h1;
                sp[shadow_dst(d, n + 2 * d.sn0, k + 2 * d.sk0)] = h2;
                sp[shadow_dst(d, n + 3 * d.sn0, k + 3 * d.sk0)] = h3;
            }
        }
    }
}
__device__ __forceinline__ void shadow_store4(unsigned short* __restrict__ shadow, const ShadowTab& tab, int64_t i0, float4 pn) {
#pragma unroll
    for (int t = 0; t < SHADOW_MAX; ++t)
        if (t < tab.n) shadow_store4_desc(shadow, tab.d[t], (tab.flags & (2 << t)) != 0, i0, pn);
}

__device__ __forceinline__ void shadow_store_desc(unsigned short* __restrict__ shadow, const ShadowDesc& d, int64_t i, float pn) {
    const int64_t idx = i - d.src_off;
    if (idx >= 0 && idx < d.n_el) {
        const int32_t x = (int32_t)idx;
        const int32_t q = x / d.n0, d0 = x - q * d.n0, d2 = q / d.n1, d1 = q - d2 * d.n1;
        const int64_t dst = shadow_dst(d, d0 * d.sn0 + d1 * d.sn1 + d2 * d.sn2, d0 * d.sk0 + d1 * d.sk1 + d2 * d.sk2);
        float r = pn;
#pragma unroll 1
        for (int t = 0; t < d.terms; ++t) shadow[dst + t * d.plane_stride] = bf_term_next(r);
    }
}
__device__ __forceinline__ void shadow_store(unsigned short* __restrict__ shadow, const ShadowTab& tab, int64_t i, float pn) {
#pragma unroll
    for (int t = 0; t < SHADOW_MAX; ++t)
        if (t < tab.n) shadow_store_desc(shadow, tab.d[t], i, pn);
}

// torch.optim.Adam / AdamW update (torch/optim/adam.py single-tensor path), coefficients of one step:
//   g += wd*p (Adam)  |  p *= 1 - lr*wd (AdamW)
//   m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2
//   p -= lr/(1-b1^t) * m / ( sqrt(v)/sqrt(1-b2^t) + eps )
struct AdamCoef {
    float decay, l2, b1, b2, step_size, inv_sqrt_bc2, eps, gs;
    __device__ __forceinline__ void init(float lr, float b1_, float b2_, float eps_, float wd, int decoupled, float grad_scale,
                                         int64_t step) {
        const float bc1 = 1.0f - powf(b1_, (float)step), bc2 = 1.0f - powf(b2_, (float)step);
        b1 = b1_, b2 = b2_, eps = eps_, gs = grad_scale;
        step_size = lr / bc1, inv_sqrt_bc2 = 1.0f / sqrtf(bc2);
        decay = decoupled ? 1.0f - lr * wd : 1.0f, l2 = decoupled ? 0.f : wd;
    }
    // (no fused multiply-adds: the update is inlined into several kernels and contexts -- 16-byte and scalar paths of the
    //  optimizer launch, the weight-gradient launch's epilogue -- and must round identically in all of them)
    __device__ __forceinline__ void upd(float& pi, float gi, float& mi, float& vi) const {
#pragma clang fp contract(off)
        pi *= decay;
        gi = gi * gs + l2 * pi;
        mi = b1 * mi + (1.f - b1) * gi;
        vi = b2 * vi + (1.f - b2) * gi * gi;
        pi = pi - step_size * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
    }
};

}  // namespace
