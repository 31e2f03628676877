// fp32-class GEMM on the bf16 matrix cores: C[M, N] (+ split-K slabs) = A[M, K] B[N, K]^T with both operands fp32 in memory,
// K contiguous.  Every fp32 value is split into three bf16 terms while it is staged into LDS, x = h + m + l
// (h = bf16(x), m = bf16(x - h), l = bf16(x - h - m): 24 significant bits), and the six cross products that matter
// (hh, hm, mh, hl, lh, mm; the dropped ml, lm, ll are below 2^-24 relative) are accumulated in fp32 on
// v_mfma_f32_16x16x32_bf16.  Six bf16 MFMAs move 6 x 16 x 16 x 32 MACs in 96 cycles, the exact-fp32 MFMA
// (v_mfma_f32_16x16x4_f32) needs 256 cycles for the same tile-k volume: 2.7 x the fp32 roof.
//
// Used for the two big products of MMGCN's re-associated GCNII chain (track_mm/mmgcn_models.py:373-394: out_l = theta [hi ||
// h0] W_l + ...): Call = h0 [U_1 .. U_64] ([3N, 200] x [200, 12800], 163 MB of output) and dh0 = dCall [U_1 .. U_64]^T
// ([3N, 12800] x [12800, 200]), 16.3 GFLOP each, which the generic exact-fp32 tiles (64 x 128, csrc/gemm.hip) ran at 36 % of
// the fp32 matrix peak (287 / 305 us).  The split is amortised here: a staged element is reused by 128 outputs (the
// register-streaming weight-gradient form of the same idea, erc_wgrad_table_x3, reuses nothing and gained 1 %: finding 41).
//
// Geometry: 256 threads = 2 x 2 wavefronts of 64 x 64 outputs, workgroup tile 128 x 128, K in chunks of 32 through ONE LDS
// stage (6 bf16 planes of 128 x 32, 60 KB: two workgroups per CU -- one multiplies while the other stages) with the next
// chunk's global loads in registers.  LDS rows are 40 bf16 (80 B): the 16 rows of a fragment read start in 16 different
// 4-bank groups.
#include "erc_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int XT = 128;          // tile rows / columns
constexpr int XK = 32;           // K per chunk
constexpr int XS = XK + 8;       // LDS row pitch (bf16 elements)
constexpr int XPLANE = XT * XS;  // elements of one plane

struct X3P {
    const float* A;
    const float* B;
    float* C;
    int64_t c_slab;
    int lda, ldb, ldc, M, N, K, chunks_per_split;
};

typedef __bf16 bfx2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// two floats -> one packed bf16 pair (ONE v_cvt_pk_bf16_f32, round to nearest even) and back
__device__ __forceinline__ unsigned pk_bf(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, bfx2));
}
__device__ __forceinline__ float lo_val(unsigned pk) { return __builtin_bit_cast(float, pk << 16); }
__device__ __forceinline__ float hi_val(unsigned pk) { return __builtin_bit_cast(float, pk & 0xffff0000u); }

// four consecutive k of one row -> the three planes (8 bytes each): 13 VALU operations per pair of values
__device__ __forceinline__ void split_store(unsigned short* dst, const f32x4 x) {
    unsigned h[2], m[2], l[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const float a = x[2 * t], b = x[2 * t + 1];
        h[t] = pk_bf(a, b);
        const float ra = a - lo_val(h[t]), rb = b - hi_val(h[t]);
        m[t] = pk_bf(ra, rb);
        l[t] = pk_bf(ra - lo_val(m[t]), rb - hi_val(m[t]));
    }
    *reinterpret_cast<u32x2*>(dst) = (u32x2){h[0], h[1]};
    *reinterpret_cast<u32x2*>(dst + XPLANE) = (u32x2){m[0], m[1]};
    *reinterpret_cast<u32x2*>(dst + 2 * XPLANE) = (u32x2){l[0], l[1]};
}

__device__ __forceinline__ void gemm_x3_body(const X3P& p, const int n0, const int m0, const int z, unsigned short* lds) {
    unsigned short* const As = lds;
    unsigned short* const Bs = lds + 3 * XPLANE;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 15, g = lane >> 4;
    const int wm = w >> 1, wn = w & 1;
    const int nchunk = (p.K + XK - 1) / XK;
    const int c_begin = z * p.chunks_per_split, c_end = min(nchunk, c_begin + p.chunks_per_split);

    // staging map: float4 q = tid + 256 j (j < 4) of a 128 x 32 tile: row q >> 3, k = 4 (q & 7); a row is 128 contiguous bytes
    const int srow = tid >> 3, skq = 4 * (tid & 7);
    f32x4 pa[4], pb[4];
    auto fetch = [&](int c) {
        const int k = c * XK + skq;
        const bool kv = k < p.K;                       // K % 4 == 0: a quad is inside or outside
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ra = m0 + srow + 32 * j, rb = n0 + srow + 32 * j;
            const f32x4 va = *reinterpret_cast<const f32x4*>(p.A + (int64_t)min(ra, p.M - 1) * p.lda + (kv ? k : 0));
            const f32x4 vb = *reinterpret_cast<const f32x4*>(p.B + (int64_t)min(rb, p.N - 1) * p.ldb + (kv ? k : 0));
            const float ma = (kv && ra < p.M) ? 1.f : 0.f, mb = (kv && rb < p.N) ? 1.f : 0.f;
            pa[j] = va * ma, pb[j] = vb * mb;
        }
    };
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (c_begin < c_end) fetch(c_begin);
    for (int c = c_begin; c < c_end; ++c) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            split_store(As + (srow + 32 * j) * XS + skq, pa[j]);
            split_store(Bs + (srow + 32 * j) * XS + skq, pb[j]);
        }
        __syncthreads();
        if (c + 1 < c_end) fetch(c + 1);          // in flight during the products
        // the B fragments of this wavefront's four column tiles, all three planes
        bf16x8 bh[4], bm[4], bl[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned short* q = Bs + (64 * wn + 16 * j + r) * XS + 8 * g;
            bh[j] = *reinterpret_cast<const bf16x8*>(q), bm[j] = *reinterpret_cast<const bf16x8*>(q + XPLANE);
            bl[j] = *reinterpret_cast<const bf16x8*>(q + 2 * XPLANE);
        }
        bf16x8 ah[4], am[4], al[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned short* q = As + (64 * wm + 16 * i + r) * XS + 8 * g;
            ah[i] = *reinterpret_cast<const bf16x8*>(q), am[i] = *reinterpret_cast<const bf16x8*>(q + XPLANE);
            al[i] = *reinterpret_cast<const bf16x8*>(q + 2 * XPLANE);
        }
        // six passes over the 16 tiles, small terms first: consecutive MFMAs write different accumulators (a dependent chain
        // of six per tile would leave the matrix pipe idle between them)
#define X3_PASS(AP, BP)                                                                                   \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j)           \
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(AP[i], BP[j], acc[i][j], 0, 0, 0);
        X3_PASS(al, bh)
        X3_PASS(ah, bl)
        X3_PASS(am, bm)
        X3_PASS(am, bh)
        X3_PASS(ah, bm)
        X3_PASS(ah, bh)
#undef X3_PASS
        __syncthreads();
    }
    // C / D layout of the 16 x 16 tile: register q of lane (r, g) is C[4 g + q][r]
    float* const C = p.C + (int64_t)z * p.c_slab;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = m0 + 64 * wm + 16 * i + 4 * g + q;
            if (row >= p.M) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = n0 + 64 * wn + 16 * j + r;
                if (col < p.N) C[(int64_t)row * p.ldc + col] = acc[i][j][q];
            }
        }
}

__global__ __launch_bounds__(256, 2) void gemm_x3_kernel(const X3P p) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[6 * XPLANE];   // A: planes 0..2, B: planes 3..5
    gemm_x3_body(p, blockIdx.x * XT, blockIdx.y * XT, blockIdx.z, lds);
}

// Grouped form (erc_gemm_f32_grouped form 1): block (dialogue b, modality m) of a block-diagonal result,
// Blk[L, L] = A_rows[L, K] B_rows[L, K]^T over the node rows m * n_nodes + [node_off[b], node_off[b + 1]) -- one 128 x 128 tile
// per block (L <= 128), split-K slabs.
struct X3Group {
    const int32_t* node_off;
    int n_mod, n_nodes, pitch, split;
};
__global__ __launch_bounds__(256, 2) void gemm_x3_grouped_kernel(X3P p, const X3Group g) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[6 * XPLANE];
    const int item = (int)blockIdx.x / g.split, sp = (int)blockIdx.x % g.split, b = item / g.n_mod, m = item % g.n_mod;
    const int off = g.node_off[b], L = g.node_off[b + 1] - off;
    if (L <= 0) return;
    const int64_t row0 = (int64_t)m * g.n_nodes + off;
    p.A += row0 * p.lda, p.B += row0 * p.ldb, p.C += (int64_t)item * g.pitch * g.pitch;
    p.M = p.N = min(L, XT);
    gemm_x3_body(p, 0, 0, sp, lds);
}

}  // namespace

// C[M, N] = A[M, K] B[N, K]^T, fp32 in and out, three-term bf16 split inside (fp32-class: ~2^-23 relative per product).
// split_k > 1: split s writes its partial product to C + s * c_slab (the caller adds the slabs: erc_slab_reduce).
// lda, ldb, K multiples of 4, A and B 16-byte aligned.
extern "C" int erc_gemm_x3(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K, int split_k,
                           int64_t c_slab, void* stream) {
    ERC_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0 && split_k >= 1, "gemm_x3: bad arguments");
    ERC_REQUIRE(K % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && lda >= K && ldb >= K && ldc >= N && (((uintptr_t)A | (uintptr_t)B) & 15) == 0,
                "gemm_x3: K, lda, ldb must be multiples of 4 and the operands 16-byte aligned");
    const int nchunk = erc_cdiv(K, XK);
    const int cps = erc_cdiv(nchunk, split_k), splits = erc_cdiv(nchunk, cps);
    ERC_REQUIRE(splits == split_k || split_k == 1, "gemm_x3: split_k = %d leaves an empty split (K = %d: use %d)", split_k, K, splits);
    ERC_REQUIRE(split_k == 1 || c_slab >= (int64_t)M * ldc, "gemm_x3: c_slab too small");
    X3P p{A, B, C, c_slab, lda, ldb, ldc, M, N, K, cps};
    hipLaunchKernelGGL(gemm_x3_kernel, dim3(erc_cdiv(N, XT), erc_cdiv(M, XT), splits), dim3(256), 0, (hipStream_t)stream, p);
    ERC_LAUNCH_CHECK("gemm_x3");
    return ERC_OK;
}

// erc_gemm_f32_grouped(form 1) with the three-term split: dialogue blocks of at most 128 rows, C = [n_dlg * n_mod] blocks of
// pitch x pitch floats (+ split-K slabs of c_slab floats), rows of block (b, m) = m * n_nodes + [node_off[b], node_off[b + 1]).
extern "C" int erc_gemm_x3_grouped(const float* A, int lda, const float* B, int ldb, float* C, int pitch, const int32_t* node_off,
                                   int n_dlg, int n_mod, int n_nodes, int max_rows, int K, int split_k, int64_t c_slab, void* stream) {
    ERC_REQUIRE(A && B && C && node_off && n_dlg > 0 && n_mod > 0 && K > 0 && split_k >= 1, "gemm_x3_grouped: bad arguments");
    ERC_REQUIRE(max_rows <= XT && pitch >= max_rows, "gemm_x3_grouped: blocks of at most %d rows (got %d, pitch %d)", XT, max_rows, pitch);
    ERC_REQUIRE(K % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && lda >= K && ldb >= K && (((uintptr_t)A | (uintptr_t)B) & 15) == 0,
                "gemm_x3_grouped: K, lda, ldb must be multiples of 4 and the operands 16-byte aligned");
    const int nchunk = erc_cdiv(K, XK);
    const int cps = erc_cdiv(nchunk, split_k), splits = erc_cdiv(nchunk, cps);
    ERC_REQUIRE(splits == split_k, "gemm_x3_grouped: split_k = %d leaves an empty split (K = %d: use %d)", split_k, K, splits);
    ERC_REQUIRE(split_k == 1 || c_slab >= (int64_t)n_dlg * n_mod * pitch * pitch, "gemm_x3_grouped: c_slab too small");
    X3P p{A, B, C, c_slab, lda, ldb, pitch, 0, 0, K, cps};
    X3Group g{node_off, n_mod, n_nodes, pitch, splits};
    hipLaunchKernelGGL(gemm_x3_grouped_kernel, dim3(n_dlg * n_mod * splits), dim3(256), 0, (hipStream_t)stream, p, g);
    ERC_LAUNCH_CHECK("gemm_x3_grouped");
    return ERC_OK;
}
