// Split compute modes (f32x2 / f32x3): fp32-class dense products on the bf16 matrix cores.
//
// north_star asks for logits within 1e-4 of the reference's fp32 arithmetic (track_mm/cogmen.py:61-74,116-122,179-195);
// v_mfma_f32_16x16x4_f32 runs at 1/16 of the bf16 instruction's rate.  Every fp32 operand value x is therefore expanded
// into NT bf16 terms,
//     x = t0 + t1 (+ t2),   t0 = bf16(x), t1 = bf16(x - t0), t2 = bf16(x - t0 - t1)      (round to nearest even;
//                                                                                          the remainders are exact in fp32)
// and a product A B is the sum of the term products t_i(A) t_j(B) with i + j < NT -- the ones whose weight 2^(-8 (i + j)) is
// above the expansion's own resolution -- accumulated in fp32 on v_mfma_f32_16x16x32_bf16, small terms first:
//     NT = 2:  3 products, operands to 2^-17 (16 significant bits + sign of the remainder): measured on COGMEN config 2 against
//              float64: logits 7.6e-6, gradients 6.8e-5 of a tensor's scale (tests/study_split_numerics.py)
//     NT = 3:  6 products, operands to 2^-25: indistinguishable from fp32 arithmetic (1.8e-7 / 7.5e-6; plain fp32: 3.7e-7 / 5.4e-6)
// No exponent-range caveat (bf16 has fp32's exponent), no loss scaling.
#pragma once
#include <hip/hip_runtime.h>

namespace {

typedef short sp_bf16x8 __attribute__((ext_vector_type(8)));
typedef float sp_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int sp_u32x4 __attribute__((ext_vector_type(4)));

// {bf16(lo), bf16(hi)} as one dword, lo in the low half: v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned sp_pack(float lo, float hi) {
    const __bf16 a = (__bf16)lo, b = (__bf16)hi;
    return (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
}
__device__ __forceinline__ float sp_lo(unsigned d) { return __builtin_bit_cast(float, d << 16); }
__device__ __forceinline__ float sp_hi(unsigned d) { return __builtin_bit_cast(float, d & 0xffff0000u); }

// the NT packed term dwords of a pair of values (6 VALU operations for NT = 2, 11 for NT = 3)
template <int NT>
__device__ __forceinline__ void sp_split2(float x0, float x1, unsigned (&t)[NT]) {
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        t[i] = sp_pack(x0, x1);
        if (i + 1 < NT) x0 -= sp_lo(t[i]), x1 -= sp_hi(t[i]);
    }
}

// acc += sum over (i, j), i + j < NT, of a[i] b[j] -- small terms first, the leading product last
template <int NT>
__device__ __forceinline__ sp_f32x4 sp_mfma(const sp_u32x4 (&a)[NT], const sp_u32x4 (&b)[NT], sp_f32x4 acc) {
#pragma unroll
    for (int s = NT - 1; s >= 0; --s)
#pragma unroll
        for (int i = 0; i <= s; ++i)
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(sp_bf16x8, a[i]), __builtin_bit_cast(sp_bf16x8, b[s - i]), acc, 0, 0, 0);
    return acc;
}

}  // namespace
