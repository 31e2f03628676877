// K2 (skinny shapes): register-streaming MFMA GEMM, no LDS staging, no barrier in the K loop.
//
// The matrices of this workload are skinny (M ~ 2000 nodes, N ~ 100, K 100..2000) and L2-resident, so a
// tiled/LDS GEMM spends its time in the global->LDS->barrier chain of each K chunk (measured: ~1.9 us per
// 32-deep chunk with 124 workgroups).  Here every wavefront owns a 16 x 32 output tile, loads its MFMA
// fragments STRAIGHT from global memory (one 16-byte load per fragment and 16-deep K block, several blocks in
// flight), and the wavefronts of a workgroup split K among themselves (in-workgroup split-K, reduced once through
// LDS).  Weight gradients (K = #nodes) therefore need no partial slabs and no reduce pass.
//
// K permutation: v_mfma_f32_16x16x4_f32 sums over its 4 k-slots g = lane>>4.  For a 16-deep block at kb, MFMA t
// (t < 4) uses k = kb + 4g + t for BOTH operands, so a lane's four A values are the contiguous A[row][kb+4g..+3]
// (one float4) and likewise for a K-contiguous B; the sum over t and g covers the 16 k's exactly once.
#include "erc_common.h"
#include <stdlib.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));

struct T {
    static constexpr bool value = true;
};
struct Fx {
    static constexpr bool value = false;
};

struct StreamP {
    const void* A;
    const void* B;
    float* C;
    const int32_t* a_gather;
    const int32_t* b_gather;
    const float* bias;
    const float* aux;
    float* bias_out;
    const uint64_t* rng;
    int64_t c_slab, bias_slab;
    int lda, ldb, ldc, ldaux;
    int M, N, K;
    int kblocks_per_split;
    int ones, act, accumulate;
    int a_vec, b_vec;
    float act_scale, drop_p;
    float alpha;           // act 4 (GCNII layer tail)
    uint64_t rng_stream;   // act 4: xor-ed into the RNG seed (one dropout stream per layer)
};

__device__ __forceinline__ float4 ld4g(const float* p, int valid, bool vec) {
    if (valid >= 4 && vec) return *reinterpret_cast<const float4*>(p);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid > 0) v.x = p[0];
    if (valid > 1) v.y = p[1];
    if (valid > 2) v.z = p[2];
    if (valid > 3) v.w = p[3];
    return v;
}

template <int NW, int RM, int CF>
__device__ __forceinline__ void reduce_and_store(const StreamP& p, f32x4 (&acc)[RM][CF], float* red, int m0, int n0, int z) {
    constexpr int SLOTS = RM * CF * 4;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    float* __restrict__ C = p.C + (int64_t)z * p.c_slab;
    uint64_t rng_off = 0, rng_seed = 0;
    if (p.act == 3 || (p.act == 4 && p.drop_p > 0.f)) rng_off = p.rng[0], rng_seed = p.rng[1] ^ p.rng_stream;
    if (NW > 1) {
#pragma unroll
        for (int h = 0; h < RM; ++h)
#pragma unroll
            for (int f = 0; f < CF; ++f)
#pragma unroll
                for (int i = 0; i < 4; ++i) red[(w * SLOTS + (h * CF + f) * 4 + i) * 64 + lane] = acc[h][f][i];
        __syncthreads();
        if (w != 0) return;
#pragma unroll
        for (int h = 0; h < RM; ++h)
#pragma unroll
            for (int f = 0; f < CF; ++f)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float s = 0.f;
#pragma unroll
                    for (int ww = 0; ww < NW; ++ww) s += red[(ww * SLOTS + (h * CF + f) * 4 + i) * 64 + lane];
                    acc[h][f][i] = s;
                }
    }
    // bias of this lane's columns, requested once (a guarded load per stored element costs a round trip each)
    float bv[CF];
#pragma unroll
    for (int f = 0; f < CF; ++f) bv[f] = 0.f;
    if (p.bias) {  // uniform
#pragma unroll
        for (int f = 0; f < CF; ++f) bv[f] = p.bias[min(n0 + 16 * f + r, p.N - 1)];
    }
#pragma unroll
    for (int h = 0; h < RM; ++h)
#pragma unroll
        for (int f = 0; f < CF; ++f) {
            const int col = n0 + 16 * f + r;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = m0 + 16 * h + 4 * g + i;
                float v = acc[h][f][i];
                if (row >= p.M) {
                    if (row == p.M && p.ones == 2 && col < p.N && p.bias_out) p.bias_out[(int64_t)z * p.bias_slab + col] = v;
                    continue;
                }
                if (col < p.N) {
                    float* dst = C + (int64_t)row * p.ldc + col;
                    v += bv[f];
                    if (p.accumulate) v += *dst;
                    if (p.act == 1)
                        v = fmaxf(v, 0.f);
                    else if (p.act == 2)
                        v = p.aux[(int64_t)row * p.ldaux + col] > 0.f ? v * p.act_scale : 0.f;
                    else if (p.act == 3) {
                        const float u = erc_uniform(rng_seed, rng_off, (uint64_t)row * (uint64_t)p.N + col);
                        v = (u >= p.drop_p) ? fmaxf(v, 0.f) * p.act_scale : 0.f;
                    } else if (p.act == 4) {   // GCNII layer tail: aux = [hi | h0] rows (pitch ldaux, h0 at column N)
                        const float* ar = p.aux + (int64_t)row * p.ldaux + col;
                        v = p.act_scale * v + (1.f - p.act_scale) * ((1.f - p.alpha) * ar[0] + p.alpha * ar[p.N]);
                        v = fmaxf(v, 0.f);
                        if (p.drop_p > 0.f) {
                            const float u = erc_uniform(rng_seed, rng_off, (uint64_t)row * (uint64_t)p.N + col);
                            v = (u >= p.drop_p) ? v * (1.0f / (1.0f - p.drop_p)) : 0.f;
                        }
                    }
                    *dst = v;
                } else if (col == p.N && p.ones == 1 && p.bias_out) {
                    p.bias_out[(int64_t)z * p.bias_slab + row] = v;
                }
            }
        }
}

// Wave tile (16 RM) x (16 CF): RM x CF MFMA tiles share the RM + CF fragments of a K block.  The skinny products of
// this workload are bound by the per-CU L2 path (~70 GB/s), so bigger wave tiles (fewer fragment bytes per MFMA) win
// as long as enough workgroups remain to fill the chip; the host picks (1,2), (1,4) or (2,4).
template <int A_MODE, int B_MODE, int NW, bool GA, bool GB, bool VEC, int RM, int CF>
__device__ __forceinline__ void gemm_f32_stream_body(const StreamP& p, float* red, const int bx, const int by, const int z) {
    const float* __restrict__ A = (const float*)p.A;
    const float* __restrict__ B = (const float*)p.B;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int n0 = bx * 16 * CF, m0 = by * 16 * RM;
    const int nkb = (p.K + 15) / 16;
    const int kb_begin = z * p.kblocks_per_split;
    const int kb_end = min(nkb, kb_begin + p.kblocks_per_split);
    const int kb_full_end = min(kb_end, p.K / 16);  // blocks entirely inside K

    // Out-of-range rows / columns / k are CLAMPED to a valid address and the loaded value is replaced afterwards
    // by a select: every load is unconditional, so no branch and no s_waitcnt separates them.
    int64_t a_row[RM];
    bool m_ok[RM];
    float a_fill[RM];
#pragma unroll
    for (int h = 0; h < RM; ++h) {
        const int m = m0 + 16 * h + r;
        m_ok[h] = m < p.M;
        a_fill[h] = (m == p.M && p.ones == 2) ? 1.f : 0.f;
        const int mc = min(m, p.M - 1);
        if (A_MODE == 0)
            a_row[h] = (GA ? (int64_t)p.a_gather[mc] : (int64_t)mc) * p.lda;
        else
            a_row[h] = mc;
    }
    int nc[CF];
    bool n_ok[CF];
    float b_fill[CF];
#pragma unroll
    for (int f = 0; f < CF; ++f) {
        const int n = n0 + 16 * f + r;
        n_ok[f] = n < p.N;
        b_fill[f] = (n == p.N && p.ones == 1) ? 1.f : 0.f;
        nc[f] = min(n, p.N - 1);
    }

    f32x4 acc[RM][CF];
#pragma unroll
    for (int h = 0; h < RM; ++h)
#pragma unroll
        for (int f = 0; f < CF; ++f) acc[h][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // FULL: the block lies inside K (vector loads allowed, no k masking)
    auto load = [&](const int kb, float (&a)[RM][4], float (&b)[CF][4], auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        const int k = kb * 16 + 4 * g;
        int kc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) kc[j] = FULL ? k + j : min(k + j, p.K - 1);
#pragma unroll
        for (int h = 0; h < RM; ++h) {
            if (A_MODE == 0) {
                if (FULL && VEC) {
                    const float4 v = *reinterpret_cast<const float4*>(A + a_row[h] + k);
                    a[h][0] = v.x, a[h][1] = v.y, a[h][2] = v.z, a[h][3] = v.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) a[h][j] = A[a_row[h] + kc[j]];
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) a[h][j] = A[(int64_t)kc[j] * p.lda + a_row[h]];
            }
        }
        if (B_MODE == 0) {
#pragma unroll
            for (int f = 0; f < CF; ++f) {
                const float* bp = B + (int64_t)nc[f] * p.ldb;
                if (FULL && VEC) {
                    const float4 v = *reinterpret_cast<const float4*>(bp + k);
                    b[f][0] = v.x, b[f][1] = v.y, b[f][2] = v.z, b[f][3] = v.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) b[f][j] = bp[kc[j]];
                }
            }
        } else {
            int64_t brow[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) brow[j] = (GB ? (int64_t)p.b_gather[kc[j]] : (int64_t)kc[j]) * p.ldb;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int f = 0; f < CF; ++f) b[f][j] = B[brow[j] + nc[f]];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool kin = FULL || (k + j < p.K);
#pragma unroll
            for (int h = 0; h < RM; ++h) a[h][j] = kin ? (m_ok[h] ? a[h][j] : a_fill[h]) : 0.f;
#pragma unroll
            for (int f = 0; f < CF; ++f) b[f][j] = kin ? (n_ok[f] ? b[f][j] : b_fill[f]) : 0.f;
        }
    };
    auto mma = [&](const float (&a)[RM][4], const float (&b)[CF][4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int h = 0; h < RM; ++h)
#pragma unroll
                for (int f = 0; f < CF; ++f) acc[h][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[h][j], b[f][j], acc[h][f], 0, 0, 0);
    };
    // UB independent K blocks per iteration: all their fragment loads are issued before the first MFMA
    constexpr int UB = (RM * CF >= 8) ? 2 : 4;
    int kb = kb_begin + w;
    for (; kb + (UB - 1) * NW < kb_full_end; kb += UB * NW) {
        float a[UB][RM][4], b[UB][CF][4];
#pragma unroll
        for (int u = 0; u < UB; ++u) load(kb + u * NW, a[u], b[u], T{});
#pragma unroll
        for (int u = 0; u < UB; ++u) mma(a[u], b[u]);
    }
    // the 1 .. UB-1 full blocks that remain for this wavefront: one batch (one round trip), not one per block
    const int rem = kb < kb_full_end ? (kb_full_end - kb + NW - 1) / NW : 0;  // wave-uniform
    if (rem == 3) {
        float a[3][RM][4], b[3][CF][4];
#pragma unroll
        for (int u = 0; u < 3; ++u) load(kb + u * NW, a[u], b[u], T{});
#pragma unroll
        for (int u = 0; u < 3; ++u) mma(a[u], b[u]);
    } else if (rem == 2) {
        float a[2][RM][4], b[2][CF][4];
#pragma unroll
        for (int u = 0; u < 2; ++u) load(kb + u * NW, a[u], b[u], T{});
#pragma unroll
        for (int u = 0; u < 2; ++u) mma(a[u], b[u]);
    } else if (rem == 1) {
        float a0[RM][4], b0[CF][4];
        load(kb, a0, b0, T{});
        mma(a0, b0);
    }
    kb += rem * NW;
    if (kb < kb_end) {  // the one partial block of this split (k-clamped scalar loads)
        float a0[RM][4], b0[CF][4];
        load(kb, a0, b0, Fx{});
        mma(a0, b0);
    }
    reduce_and_store<NW, RM, CF>(p, acc, red, m0, n0, z);
}

template <int A_MODE, int B_MODE, int NW, bool GA, bool GB, bool VEC, int RM, int CF>
__global__ __launch_bounds__(64 * NW) void gemm_f32_stream_kernel(StreamP p) {
    __shared__ float red[NW > 1 ? NW * RM * CF * 4 * 64 : 1];
    gemm_f32_stream_body<A_MODE, B_MODE, NW, GA, GB, VEC, RM, CF>(p, red, blockIdx.x, blockIdx.y, blockIdx.z);
}

// bf16 feature block as the A operand (rows K-contiguous, optional gather), fp32 B [N,K] rounded to bf16 on the fly:
// the forward input projection.  v_mfma_f32_16x16x32_bf16: lane (r,g) holds k = kb*32 + 8g + j, j < 8.
__device__ __forceinline__ short f2bf_s(float f) {
    const __bf16 h = (__bf16)f;
    return __builtin_bit_cast(short, h);
}

// Wave tile 32 rows x 32 columns (2 x 2 MFMA tiles: every B fragment is used for two row blocks, which halves the
// W traffic through the per-CU L2 path -- the measured limiter of this kernel, ~70 GB/s per CU), 8 wavefronts
// split K.  WB = true: W is given as a bf16 shadow copy (refreshed by the optimizer kernel), else fp32 rounded
// while loaded.
// UB = K blocks per load batch: 6 (one batch covers K = 1380 / 8 wavefronts: fewest round trips, 188 VGPRs, one
// workgroup per CU).
template <int NW, bool WB, int UB>
__global__ __launch_bounds__(64 * NW) void gemm_bf16a_stream_kernel(StreamP p) {
    __shared__ float red[NW > 1 ? NW * 16 * 64 : 1];
    const unsigned short* __restrict__ A = (const unsigned short*)p.A;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    // XCD-aware tile mapping (speed only): workgroups are dealt round-robin over the 8 XCDs, so blocks b and b+8
    // share an L2.  All column tiles of one 32-row group are given to the same XCD, which then fetches that
    // group's feature rows from HBM once instead of once per column tile (measured: FETCH_SIZE 4x -> ~1x).
    const int n_ct = (p.N + 31) / 32, n_rg = (p.M + 31) / 32;
    const int L = blockIdx.x, xcd = L & 7, j = L >> 3;
    const int rg = xcd + 8 * (j / n_ct), ct = j % n_ct;
    if (rg >= n_rg) return;
    const int n0 = ct * 32, m0 = rg * 32;
    const int nkb = (p.K + 31) / 32;
    const int kb_full_end = p.K / 32;
    int64_t a_row[2];
    bool m_ok[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int m = m0 + 16 * h + r;
        m_ok[h] = m < p.M;
        const int mc = min(m, p.M - 1);
        a_row[h] = (p.a_gather ? (int64_t)p.a_gather[mc] : (int64_t)mc) * p.lda;
    }
    int nc[2];
    bool n_ok[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        const int n = n0 + 16 * f + r;
        n_ok[f] = n < p.N;
        nc[f] = min(n, p.N - 1);
    }
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    f32x4 acc[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int f = 0; f < 2; ++f) acc[h][f] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // every load is unconditional (clamped address + select afterwards): no branch, no wait between loads
    auto load = [&](const int kb, bf16x8 (&a)[2], bf16x8 (&b)[2], auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        const int k = kb * 32 + 8 * g;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const unsigned short* s = A + a_row[h];
            if (FULL && p.a_vec == 2) {
                a[h] = *reinterpret_cast<const bf16x8*>(s + k);
            } else if (FULL && p.a_vec == 1) {
                const bf16x4 lo = *reinterpret_cast<const bf16x4*>(s + k), hi = *reinterpret_cast<const bf16x4*>(s + k + 4);
                a[h] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            } else if (FULL && p.a_vec == 4) {   // rows only 4-byte aligned (odd multiples of 2 elements, e.g. D = 1242)
                typedef short bf16x2 __attribute__((ext_vector_type(2)));
                const bf16x2 q0 = *reinterpret_cast<const bf16x2*>(s + k), q1 = *reinterpret_cast<const bf16x2*>(s + k + 2);
                const bf16x2 q2 = *reinterpret_cast<const bf16x2*>(s + k + 4), q3 = *reinterpret_cast<const bf16x2*>(s + k + 6);
                a[h] = (bf16x8){q0[0], q0[1], q1[0], q1[1], q2[0], q2[1], q3[0], q3[1]};
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const short v = (short)s[min(k + j, p.K - 1)];
                    a[h][j] = (FULL || k + j < p.K) ? v : (short)0;
                }
            }
            a[h] = m_ok[h] ? a[h] : zero8;
        }
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            bf16x8 v;
            if (WB) {
                const unsigned short* bp = (const unsigned short*)p.B + (int64_t)nc[f] * p.ldb;
                if (FULL && p.b_vec == 2) {
                    v = *reinterpret_cast<const bf16x8*>(bp + k);
                } else if (FULL && p.b_vec == 1) {
                    const bf16x4 lo = *reinterpret_cast<const bf16x4*>(bp + k), hi = *reinterpret_cast<const bf16x4*>(bp + k + 4);
                    v = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const short t = (short)bp[min(k + j, p.K - 1)];
                        v[j] = (FULL || k + j < p.K) ? t : (short)0;
                    }
                }
            } else {
                const float* bp = (const float*)p.B + (int64_t)nc[f] * p.ldb;
                float t[8];
                if (FULL && p.b_vec == 1) {
                    const float4 lo = *reinterpret_cast<const float4*>(bp + k), hi = *reinterpret_cast<const float4*>(bp + k + 4);
                    t[0] = lo.x, t[1] = lo.y, t[2] = lo.z, t[3] = lo.w, t[4] = hi.x, t[5] = hi.y, t[6] = hi.z, t[7] = hi.w;
                } else if (FULL && p.b_vec == 3) {   // rows only 8-byte aligned
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float2 v2 = *reinterpret_cast<const float2*>(bp + k + 2 * j);
                        t[2 * j] = v2.x, t[2 * j + 1] = v2.y;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float x = bp[min(k + j, p.K - 1)];
                        t[j] = (FULL || k + j < p.K) ? x : 0.f;
                    }
                }
                v = (bf16x8){f2bf_s(t[0]), f2bf_s(t[1]), f2bf_s(t[2]), f2bf_s(t[3]),
                             f2bf_s(t[4]), f2bf_s(t[5]), f2bf_s(t[6]), f2bf_s(t[7])};
            }
            b[f] = n_ok[f] ? v : zero8;
        }
    };
    auto mma = [&](const bf16x8 (&a)[2], const bf16x8 (&b)[2]) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int f = 0; f < 2; ++f) acc[h][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[h], b[f], acc[h][f], 0, 0, 0);
    };
    // bias of this lane's two columns, requested up front (a guarded load in the epilogue costs a round trip per use)
    float bv[2] = {0.f, 0.f};
    if (p.bias) {  // uniform
#pragma unroll
        for (int f = 0; f < 2; ++f) bv[f] = p.bias[nc[f]];
    }
    // the one partial K block belongs to wavefront (kb_full_end % NW): loaded first, multiplied last
    const bool has_tail = nkb > kb_full_end && (kb_full_end % NW) == w;
    bf16x8 ta[2] = {zero8, zero8}, tb[2] = {zero8, zero8};
    if (has_tail) load(kb_full_end, ta, tb, Fx{});
    // batches of UB K blocks per wavefront, all loads of a batch in flight; blocks past the end are clamped to a valid
    // one and their A fragments ANDed with zero (no select on a load result: hipcc would branch around the load)
    for (int kb = w; kb < kb_full_end; kb += UB * NW) {
        bf16x8 a[UB][2], b[UB][2];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const bool valid = kb + u * NW < kb_full_end;
            load(valid ? kb + u * NW : kb, a[u], b[u], T{});
            const short mk = valid ? (short)-1 : (short)0;
            const bf16x8 m8 = {mk, mk, mk, mk, mk, mk, mk, mk};
            a[u][0] &= m8, a[u][1] &= m8;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < UB; ++u) mma(a[u], b[u]);
    }
    if (has_tail) mma(ta, tb);
    // in-workgroup split-K reduction, then bias / relu epilogue
    if (NW > 1) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int i = 0; i < 4; ++i) red[(w * 16 + h * 8 + f * 4 + i) * 64 + lane] = acc[h][f][i];
        __syncthreads();
        // every wavefront finalises 16 / NW of the 16 accumulator slots (sum over the wavefronts in order) and stores them
#pragma unroll
        for (int s = 0; s < 16 / NW; ++s) {
            const int slot = w + NW * s, h = slot >> 3, f = (slot >> 2) & 1, i = slot & 3;
            float sum = 0.f;
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) sum += red[(ww * 16 + slot) * 64 + lane];
            const int row = m0 + 16 * h + 4 * g + i, col = n0 + 16 * f + r;
            if (row < p.M && col < p.N) {
                float v = sum + (f ? bv[1] : bv[0]);
                if (p.act == 1) v = fmaxf(v, 0.f);
                p.C[(int64_t)row * p.ldc + col] = v;
            }
        }
        return;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            const int col = n0 + 16 * f + r;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = m0 + 16 * h + 4 * g + i;
                if (row < p.M && col < p.N) {
                    float v = acc[h][f][i] + bv[f];
                    if (p.act == 1) v = fmaxf(v, 0.f);
                    p.C[(int64_t)row * p.ldc + col] = v;
                }
            }
        }
}

// Large row counts (many row groups per CU): persistent variant with the WEIGHTS RESIDENT IN REGISTERS.
// The streaming kernel above re-reads its 32 x K slice of W for every 32 x 32 output tile through the per-CU L2 path
// (~70 GB/s): at M = 32 k rows that path, not HBM, sets the time (118 us for 106 MB = 0.9 TB/s).  Here a workgroup of
// 8 wavefronts splits K once (wavefront w owns the 32-deep K blocks w, w + 8, ...; at most 6), loads the W fragments of
// FOUR column tiles of its K blocks into registers (6 x 4 fragments = 96 VGPRs; all 7 tiles would spill) and then walks
// over 16-row groups: per group each lane loads its 6 A fragments -- the only memory traffic --, 24 MFMAs, the 8
// partial 16 x 64 tiles are summed through LDS, bias / relu, store.  Two workgroups (column tiles 0-3 and 4-6) take the
// same row groups; their ids are equal mod 8, i.e. they sit on one XCD and the second reader of a feature row hits L2.
// N <= 112, bf16 W.
constexpr int PJ_NB = 6;    // K blocks per wavefront (8 wavefronts: K <= 1536)
constexpr int PJ_NT = 4;    // column tiles of 16 per workgroup

__global__ __launch_bounds__(512) void gemm_bf16a_persist_kernel(StreamP p) {
    __shared__ float red[8 * PJ_NT * 4 * 64];   // 32 KB: partial tiles of the 8 wavefronts
    const unsigned short* __restrict__ A = (const unsigned short*)p.A;
    const unsigned short* __restrict__ W = (const unsigned short*)p.B;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 15, g = lane >> 4;
    const int id = blockIdx.x, half = (id >> 3) & 1, pair = (id & 7) + 8 * (id >> 4), n_pairs = (int)gridDim.x >> 1;
    const int n_base = half * 16 * PJ_NT;            // first column of this workgroup
    const int nkb = (p.K + 31) / 32;
    // ---- this wavefront's K blocks.  A block that would run past K is shifted back to end at K; the k it then
    //      shares with the previous block are zeroed in the W fragment (K >= 32, K % 4 == 0).
    int k0[PJ_NB];
    bf16x8 wf[PJ_NB][PJ_NT];
#pragma unroll
    for (int s = 0; s < PJ_NB; ++s) {
        const int kb = w + 8 * s;
        const bool live = kb < nkb;
        const int kstart = kb * 32;
        k0[s] = live ? min(kstart, p.K - 32) : 0;
        const int k = k0[s] + 8 * g;                    // this lane's 8 consecutive k
#pragma unroll
        for (int nt = 0; nt < PJ_NT; ++nt) {
            const int n = n_base + 16 * nt + r;
            const unsigned short* wr = W + (int64_t)min(n, p.N - 1) * p.ldb + k;
            const bf16x4 lo = *reinterpret_cast<const bf16x4*>(wr), hi = *reinterpret_cast<const bf16x4*>(wr + 4);
            bf16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (live && n < p.N && k + j >= kstart) ? v[j] : (short)0;
            wf[s][nt] = v;
        }
    }
    const int n_rg = (p.M + 15) / 16;
    // (Tried: the A fragments of two row groups per iteration, i.e. twice the bytes per memory round trip -- no change at
    // B = 512, 40 vs 38 us: the ~2.3 us per row group are not the load latency.)
    auto load_a = [&](int rg, bf16x8 (&af)[PJ_NB]) {
        const int mc = min(rg * 16 + r, p.M - 1);
        const int64_t arow = (p.a_gather ? (int64_t)p.a_gather[mc] : (int64_t)mc) * p.lda;
#pragma unroll
        for (int s = 0; s < PJ_NB; ++s) {
            const unsigned short* ar = A + arow + k0[s] + 8 * g;
            const bf16x4 lo = *reinterpret_cast<const bf16x4*>(ar), hi = *reinterpret_cast<const bf16x4*>(ar + 4);
            af[s] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
    };
    auto group = [&](int rg, const bf16x8 (&af)[PJ_NB]) {
        f32x4 acc[PJ_NT];
#pragma unroll
        for (int nt = 0; nt < PJ_NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < PJ_NB; ++s)
#pragma unroll
            for (int nt = 0; nt < PJ_NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[s], wf[s][nt], acc[nt], 0, 0, 0);
        __syncthreads();   // previous group's partials consumed
#pragma unroll
        for (int nt = 0; nt < PJ_NT; ++nt)
#pragma unroll
            for (int i = 0; i < 4; ++i) red[((w * PJ_NT + nt) * 4 + i) * 64 + lane] = acc[nt][i];
        __syncthreads();
        // 16 x 64 outputs over 512 threads: element e = tid + 512 u -> row e / 64, col e % 64
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = tid + 512 * u, row = e >> 6, col = e & 63;
            const int idx = (((col >> 4) * 4) + (row & 3)) * 64 + 16 * (row >> 2) + (col & 15);
            float sum = 0.f;
#pragma unroll
            for (int ww = 0; ww < 8; ++ww) sum += red[ww * PJ_NT * 4 * 64 + idx];
            const int mrow = rg * 16 + row, ncol = n_base + col;
            if (mrow < p.M && ncol < p.N) {
                float v = sum + (p.bias ? p.bias[ncol] : 0.f);
                if (p.act == 1) v = fmaxf(v, 0.f);
                p.C[(int64_t)mrow * p.ldc + ncol] = v;
            }
        }
    };
    for (int rg = pair; rg < n_rg; rg += n_pairs) {
        bf16x8 a0[PJ_NB];
        load_a(rg, a0);
        group(rg, a0);
    }
}

bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

// Same contract as erc_gemm_f32 (ercgraft.h); selected by the host for skinny problems.
extern "C" int erc_gemm_f32_stream(const float* A, int lda, int a_kmajor, const int32_t* a_gather, const float* B, int ldb,
                                   int b_kmajor, const int32_t* b_gather, float* C, int ldc, int M, int N, int K,
                                   int split_k, int64_t c_slab, int ones_col, float* bias_out, int64_t bias_slab,
                                   const float* bias, int act, const float* aux, int ldaux, float act_scale, float drop_p,
                                   const uint64_t* rng_state, int accumulate, void* stream) {
    ERC_REQUIRE(A && B && C, "gemm_f32_stream: null operand");
    ERC_REQUIRE(M > 0 && N > 0 && K > 0, "gemm_f32_stream: bad shape M=%d N=%d K=%d", M, N, K);
    ERC_REQUIRE(split_k >= 1, "gemm_f32_stream: split_k must be >= 1");
    ERC_REQUIRE(!(a_kmajor && !b_kmajor), "gemm_f32_stream: (A k-major, B k-contiguous) is not built");
    ERC_REQUIRE(split_k == 1 || (!bias && act == 0 && !accumulate), "gemm_f32_stream: epilogue requires split_k == 1");
    ERC_REQUIRE(act >= 0 && act <= 3 && (act != 2 || aux) && (act != 3 || rng_state), "gemm_f32_stream: bad epilogue");
    ERC_REQUIRE(!(a_gather && a_kmajor) && !(b_gather && !b_kmajor), "gemm_f32_stream: gather on a contiguous-K index only");
    ERC_REQUIRE(ones_col >= 0 && ones_col <= 2, "gemm_f32_stream: ones mode %d", ones_col);
    StreamP p{};
    p.A = A; p.B = B; p.C = C; p.a_gather = a_gather; p.b_gather = b_gather; p.bias = bias; p.aux = aux;
    p.bias_out = bias_out; p.rng = rng_state; p.c_slab = c_slab; p.bias_slab = bias_slab;
    p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldaux = ldaux; p.M = M; p.N = N; p.K = K;
    const int nkb = erc_cdiv(K, 16);
    ERC_REQUIRE(split_k <= nkb, "gemm_f32_stream: split_k %d exceeds the %d K-blocks", split_k, nkb);
    p.kblocks_per_split = erc_cdiv(nkb, split_k);
    p.ones = ones_col; p.act = act; p.accumulate = accumulate;
    p.a_vec = al16(A) && (lda % 4 == 0);
    p.b_vec = al16(B) && (ldb % 4 == 0);
    p.act_scale = act_scale; p.drop_p = drop_p;
    const int Nlog = N + (ones_col == 1 ? 1 : 0), Mlog = M + (ones_col == 2 ? 1 : 0);
    const int kb_split = p.kblocks_per_split;
    // wavefronts per workgroup: enough to cut the K chain, not more than there are blocks
    const int nw = kb_split >= 32 ? 8 : (kb_split >= 6 ? 4 : 1);   // (1 wavefront for K = 100 measured slower: 11.2 vs 10.0 us)
    // Wave tile 16 x 32.  Measured on MI355X (COGMEN B=32 shapes, M = 1982): 32x64 / 16x64 tiles (the body is generic in
    // RM, CF) were SLOWER -- 14.1 vs 10.5 us (K=100, N=400/900), 13.9 vs 12.5 (K=900, N=100), 11.7 vs 7.9 (K=400,
    // N=100): these products are bound by the length of the per-wavefront chain, not by fragment traffic.
    // Re-checked on every module's step (16x64 / 32x64 whenever >= 300..700 workgroups remain): MMGCN 8.18 -> 8.41..8.90 ms,
    // COGMEN 0.154 -> 0.160..0.169 ms, DialogueGCN 0.746 -> 0.749..0.874 ms.
    dim3 grid(erc_cdiv(Nlog, 32), erc_cdiv(Mlog, 16), split_k);
    hipStream_t st = (hipStream_t)stream;
    const bool ga = a_gather != nullptr, gb = b_gather != nullptr;
    const bool vec = (a_kmajor || p.a_vec) && (b_kmajor || p.b_vec);
    // Long K over many rows (MMGCN's dH0 = DG [6 300, 12 800] x U^T [12 800, 200]): with 16 x 32 wave tiles every workgroup
    // streams 2.4 MB of operands through the ~70 GB/s per-CU L2 path for 0.8 MFLOP x K/16 -- 6.7 GB in all, which is the
    // launch's 381 us.  32 x 64 wave tiles halve the bytes per flop; these products are long enough that the longer
    // per-wavefront chain (what made this tile lose on the skinny COGMEN shapes) does not matter.  4 wavefronts (8 spill
    // 30 VGPRs at this tile); 64 x 64 tiles leave too few workgroups (MMGCN step 4.07 / 4.00 / 4.07 ms for 16x32 / 32x64 / 64x64).
    static int big_tile = -1;
    if (big_tile < 0) {
        const char* e = getenv("ERC_GEMM_BIG_TILE");
        big_tile = e ? atoi(e) : 1;
    }
    if (big_tile && !a_kmajor && !b_kmajor && !ga && !gb && vec && nw == 8 && K >= 4096 && Mlog >= 2048 && Nlog >= 64) {
        dim3 grid2(erc_cdiv(Nlog, 64), erc_cdiv(Mlog, 32), split_k);
        hipLaunchKernelGGL((gemm_f32_stream_kernel<0, 0, 4, false, false, true, 2, 4>), grid2, dim3(256), 0, st, p);
        ERC_LAUNCH_CHECK("gemm_f32_stream");
        return ERC_OK;
    }
#define ERC_SL4(AM, BM_, NW_, GA_, GB_, V_) \
    hipLaunchKernelGGL((gemm_f32_stream_kernel<AM, BM_, NW_, GA_, GB_, V_, 1, 2>), grid, dim3(64 * NW_), 0, st, p)
#define ERC_SL3(AM, BM_, GA_, GB_, V_)                        \
    do {                                                      \
        if (nw == 8) ERC_SL4(AM, BM_, 8, GA_, GB_, V_);       \
        else if (nw == 4) ERC_SL4(AM, BM_, 4, GA_, GB_, V_);  \
        else ERC_SL4(AM, BM_, 1, GA_, GB_, V_);               \
    } while (0)
#define ERC_SL2(AM, BM_, GA_, GB_)              \
    do {                                        \
        if (vec) ERC_SL3(AM, BM_, GA_, GB_, true); \
        else ERC_SL3(AM, BM_, GA_, GB_, false);    \
    } while (0)
    if (!a_kmajor && !b_kmajor) {
        if (ga) ERC_SL2(0, 0, true, false); else ERC_SL2(0, 0, false, false);
    } else if (!a_kmajor && b_kmajor) {
        if (ga && gb) ERC_SL2(0, 1, true, true);
        else if (ga) ERC_SL2(0, 1, true, false);
        else if (gb) ERC_SL2(0, 1, false, true);
        else ERC_SL2(0, 1, false, false);
    } else {
        if (gb) ERC_SL3(1, 1, false, true, false); else ERC_SL3(1, 1, false, false, false);
    }
#undef ERC_SL2
#undef ERC_SL3
#undef ERC_SL4
    ERC_LAUNCH_CHECK("gemm_f32_stream");
    return ERC_OK;
}

// C[M,N] = act(X[gather(m), :K] (bf16) * W[N,K]^T + bias), fp32 accumulate: forward input projection.
// w_is_bf16 != 0: W is a bf16 copy [N, ldw] (kept in sync by erc_adam_step's shadow output), else fp32.
extern "C" int erc_gemm_bf16a_stream(const void* X, int ldx, const int32_t* gather, const void* W, int ldw, int w_is_bf16,
                                     float* C, int ldc, int M, int N, int K, const float* bias, int act, void* stream) {
    ERC_REQUIRE(X && W && C && M > 0 && N > 0 && K > 0, "gemm_bf16a_stream: bad arguments");
    ERC_REQUIRE(act == 0 || act == 1, "gemm_bf16a_stream: act %d", act);
    StreamP p{};
    p.A = X; p.B = W; p.C = C; p.a_gather = gather; p.bias = bias; p.lda = ldx; p.ldb = ldw; p.ldc = ldc;
    p.M = M; p.N = N; p.K = K; p.act = act;
    const int nkb = erc_cdiv(K, 32);
    p.kblocks_per_split = nkb;
    // widest legal access per 8-element fragment: 2 = 16 B, 1 = 8 B, 4 = 4 B (bf16 operands); fp32 W: 1 = 16 B, 3 = 8 B
    auto bvec = [](const void* q, int ld) { return !al16(q) ? 0 : (ld % 8 == 0 ? 2 : (ld % 4 == 0 ? 1 : (ld % 2 == 0 ? 4 : 0))); };
    p.a_vec = bvec(X, ldx);
    p.b_vec = w_is_bf16 ? bvec(W, ldw) : (!al16(W) ? 0 : (ldw % 4 == 0 ? 1 : (ldw % 2 == 0 ? 3 : 0)));
    dim3 grid(8 * erc_cdiv(erc_cdiv(M, 32), 8) * erc_cdiv(N, 32), 1, 1);  // see the XCD-aware mapping in the kernel
    hipStream_t st = (hipStream_t)stream;
    // weights resident in registers, every feature row read once (see the kernel); ERC_PERSIST_MIN_M overrides the
    // row count from which it is used
    static int persist_min_m = -1;
    if (persist_min_m < 0) {
        const char* e = getenv("ERC_PERSIST_MIN_M");
        persist_min_m = e ? atoi(e) : 1024;   // measured crossover: the persistent kernel wins from ~2 k rows (6.9 vs 7.1 us) up
    }
    if (w_is_bf16 && M >= persist_min_m && N <= 32 * PJ_NT && K >= 32 && K <= 32 * 8 * PJ_NB && K % 4 == 0 && ldx % 4 == 0 && ldw % 4 == 0 &&
        ((uintptr_t)X & 7) == 0 && ((uintptr_t)W & 7) == 0) {
        hipLaunchKernelGGL(gemm_bf16a_persist_kernel, dim3(256), dim3(512), 0, st, p);   // 128 pairs of workgroups
        ERC_LAUNCH_CHECK("gemm_bf16a_persist");
        return ERC_OK;
    }
    // UB = 3 with two workgroups per CU (<= 128 VGPRs) was tried for big grids: it spills (42 VGPRs) and ran 192 vs 118 us at
    // B = 512; with 173 VGPRs and one workgroup per CU 143 us.  The single 6-block batch is kept for every size.
    const bool big = false;
#define ERC_BA(NW_, WB_)                                                                                          \
    do {                                                                                                          \
        if (big) hipLaunchKernelGGL((gemm_bf16a_stream_kernel<NW_, WB_, 3>), grid, dim3(64 * NW_), 0, st, p);     \
        else hipLaunchKernelGGL((gemm_bf16a_stream_kernel<NW_, WB_, 6>), grid, dim3(64 * NW_), 0, st, p);         \
    } while (0)
    if (nkb >= 16) {
        if (w_is_bf16) ERC_BA(8, true); else ERC_BA(8, false);
    } else if (nkb >= 4) {
        if (w_is_bf16) ERC_BA(4, true); else ERC_BA(4, false);
    } else {
        if (w_is_bf16) ERC_BA(1, true); else ERC_BA(1, false);
    }
#undef ERC_BA
    ERC_LAUNCH_CHECK("gemm_bf16a_stream");
    return ERC_OK;
}

// One GCNII layer's dense part with its tail in the epilogue (mmgcn_models.py GraphConvolution + GCNII.forward):
//   hd = dropout(relu(theta * ([hi | h0] W) + (1 - theta) * ((1 - alpha) hi + alpha h0)))
// A = [hi | h0] rows of pitch lda (hi = adj-propagated features, h0 = the initial residual, both F wide), W [2F, F]
// stored [in, out].  Replaces two GEMMs (hi W[:F], h0 W[F:]) and the elementwise combine launch.
extern "C" int erc_gcnii_layer_fwd(const float* hih0, int lda, const float* W, int ldw, float theta, float alpha, float drop_p,
                                   const uint64_t* rng_state, uint64_t rng_stream, float* hd, int ldo, int rows, int F,
                                   void* stream) {
    ERC_REQUIRE(hih0 && W && hd && rows > 0 && F > 0 && lda >= 2 * F && ldw >= F && ldo >= F, "gcnii_layer_fwd: bad arguments");
    ERC_REQUIRE(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || rng_state), "gcnii_layer_fwd: drop_p=%f", (double)drop_p);
    StreamP p{};
    p.A = hih0; p.B = W; p.C = hd; p.aux = hih0; p.rng = rng_state; p.rng_stream = rng_stream;
    p.lda = lda; p.ldb = ldw; p.ldc = ldo; p.ldaux = lda; p.M = rows; p.N = F; p.K = 2 * F;
    const int nkb = erc_cdiv(p.K, 16);
    p.kblocks_per_split = nkb;
    p.act = 4; p.act_scale = theta; p.alpha = alpha; p.drop_p = drop_p;
    p.a_vec = al16(hih0) && (lda % 4 == 0);
    p.b_vec = 0;
    dim3 grid(erc_cdiv(F, 32), erc_cdiv(rows, 16), 1);
    hipStream_t st = (hipStream_t)stream;
    if (p.a_vec) {
        if (nkb >= 32) hipLaunchKernelGGL((gemm_f32_stream_kernel<0, 1, 8, false, false, true, 1, 2>), grid, dim3(512), 0, st, p);
        else hipLaunchKernelGGL((gemm_f32_stream_kernel<0, 1, 4, false, false, true, 1, 2>), grid, dim3(256), 0, st, p);
    } else {
        if (nkb >= 32) hipLaunchKernelGGL((gemm_f32_stream_kernel<0, 1, 8, false, false, false, 1, 2>), grid, dim3(512), 0, st, p);
        else hipLaunchKernelGGL((gemm_f32_stream_kernel<0, 1, 4, false, false, false, 1, 2>), grid, dim3(256), 0, st, p);
    }
    ERC_LAUNCH_CHECK("gcnii_layer_fwd");
    return ERC_OK;
}
